// glue.hip -- the small kernels either side of the hot path (gfx950): everything the reference frontend does
// between two kernel calls, as device kernels, so that a device-resident driver never bounces a 3-D array to the
// host.  All are elementwise or short per-column loops with lanes = columns.
//
// Reference C ABI symbols (scalars by address):
//   rte_compute_Planck_source_1D / _2D   rte/kernels/api/mo_gas_optics_utils.F90:6-34 (impl rte/kernels/mo_gas_optics_utils.F90:36-95)
//   rte_sum_byband, rte_net_byband_full, net_byband_precalc   rte/extensions/mo_fluxes_byband.F90:156-209
// Library-extension symbols (rte_hip_*, scalars by value) -- device versions of frontend loops that are not behind
// the reference's C API:
//   rte_hip_get_layer_number / _mass      rte/kernels/mo_gas_optics_utils.F90:99-152 (no bind(C) in the reference: the
//                                         Fortran shim shim/rte_hip_fortran_shim.F90 forwards to these)
//   rte_hip_col_gas_fill                  rrtmgp/frontend/mo_gas_optics_rrtmgp.F90:594-609
//   rte_hip_tlev_interp                   :893-912
//   rte_hip_compute_optimal_angles        :1536-1561
//   rte_hip_combine_abs_and_rayleigh_1scl / _nstr   :1966-1979, :2003-2035
//   rte_hip_expand_and_transpose          rte/frontend/mo_rte_lw.F90:478-501
//   rte_hip_secants_fill                  rte/frontend/mo_rte_lw.F90:346-365
//   rte_hip_rfmip_sw_toa_renorm, _rfmip_sw_mu0, _broadcast_cols, _mask_columns
//                                         examples/rfmip-clear-sky/rrtmgp_rfmip_sw.F90:273-337
#include <math.h>

#include "common.h"

namespace {
using rte::cdiv;

// constants of rte/kernels/mo_gas_optics_constants.F90:17-35
constexpr double kBoltzmann = 1.380649e-23, kPlanckH = 6.626075540e-34, kLightspeed = 2.99792458e8;
constexpr double kMH2O = 0.018016, kAvogad = 6.02214076e23;

// B_nu, rte/kernels/mo_gas_optics_utils.F90:31-35 (same association, left to right)
__device__ __forceinline__ Float B_nu(Float T, Float nu) {
  const Float nu100 = nu * (Float)100;
  const Float c2 = (Float)kLightspeed * (Float)kLightspeed;
  const Float num = (Float)100 * (Float)2 * (Float)kPlanckH * (nu100 * nu100 * nu100) * c2;
  return num / (exp(((Float)kPlanckH * (Float)kLightspeed * nu * (Float)100) / ((Float)kBoltzmann * T)) - (Float)1);
}

// source(i, inu) = B_nu(T(i), nus(inu)) * dnus(inu), i over ncol [x nlay]
__global__ void __launch_bounds__(256)
planck_nu_kernel(size_t n, const Float* __restrict__ nus, const Float* __restrict__ dnus, const Float* __restrict__ T,
                 Float* __restrict__ source) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  const int inu = blockIdx.y;
  if (i < n) source[i + n * inu] = B_nu(T[i], nus[inu]) * dnus[inu];
}

// get_layer_number :127-152
__global__ void __launch_bounds__(256)
layer_number_kernel(int ncol, int nlay, const Float* __restrict__ vmr_h2o, const Float* __restrict__ plev, Float m_dry,
                    Float grav, Float* __restrict__ col_dry) {
  const int icol = blockIdx.x * 256 + threadIdx.x, ilay = blockIdx.y;
  if (icol >= ncol) return;
  const size_t i = icol + (size_t)ncol * ilay;
  const Float delta_plev = fabs(plev[i] - plev[i + ncol]);
  const Float fact = (Float)1 / ((Float)1 + vmr_h2o[i]);
  const Float m_air = (m_dry + (Float)kMH2O * vmr_h2o[i]) * fact;
  col_dry[i] = (Float)10 * delta_plev * (Float)kAvogad * fact / ((Float)1000 * m_air * (Float)100 * grav);
}
// get_layer_mass :99-125; vmr, layer_mass are (ngas, ncol, nlay)
__global__ void __launch_bounds__(256)
layer_mass_kernel(int ncol, int nlay, int ngas, const Float* __restrict__ vmr, const Float* __restrict__ plev,
                  const Float* __restrict__ mol_weights, Float m_dry, Float grav, Float* __restrict__ layer_mass) {
  const size_t n = (size_t)ngas * ncol * nlay;
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const int igas = (int)(i % ngas);
  const size_t cl = i / ngas;  // icol + ncol * ilay
  layer_mass[i] = vmr[i] * (mol_weights[igas] / m_dry) * fabs(plev[cl + ncol] - plev[cl]) / grav;
}
// col_gas(:,:,0) = col_dry ; col_gas(:,:,igas) = vmr(:,:,igas) * col_dry
__global__ void __launch_bounds__(256)
col_gas_fill_kernel(size_t ncl, const Float* __restrict__ vmr, const Float* __restrict__ col_dry, Float* __restrict__ col_gas) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  const int igas = blockIdx.y;  // 0 .. ngas
  if (i >= ncl) return;
  col_gas[i + ncl * igas] = igas == 0 ? col_dry[i] : vmr[i + ncl * (igas - 1)] * col_dry[i];
}
// tlev from tlay, pressure weighted :893-912
__global__ void __launch_bounds__(256)
tlev_interp_kernel(int ncol, int nlay, const Float* __restrict__ play, const Float* __restrict__ plev,
                   const Float* __restrict__ tlay, Float* __restrict__ tlev) {
  const int icol = blockIdx.x * 256 + threadIdx.x, ilev = blockIdx.y;  // 0 .. nlay
  if (icol >= ncol) return;
  auto P = [&](const Float* a, int l) { return a[icol + (size_t)ncol * l]; };  // 0-based layer / level
  Float v;
  if (ilev == 0)
    v = P(tlay, 0) + (P(plev, 0) - P(play, 0)) * (P(tlay, 1) - P(tlay, 0)) / (P(play, 1) - P(play, 0));
  else if (ilev == nlay)
    v = P(tlay, nlay - 1) +
        (P(plev, nlay) - P(play, nlay - 1)) * (P(tlay, nlay - 1) - P(tlay, nlay - 2)) / (P(play, nlay - 1) - P(play, nlay - 2));
  else
    v = (P(play, ilev - 1) * P(tlay, ilev - 1) * (P(plev, ilev) - P(play, ilev)) +
         P(play, ilev) * P(tlay, ilev) * (P(play, ilev - 1) - P(plev, ilev))) /
        (P(plev, ilev) * (P(play, ilev - 1) - P(play, ilev)));
  tlev[icol + (size_t)ncol * ilev] = v;
}
// optimal_angles(c,g) = fit(1,band(g)) * exp(-sum_l tau(c,l,g)) + fit(2,band(g)) :1536-1561
__global__ void __launch_bounds__(256)
optimal_angles_kernel(int ncol, int nlay, int nbnd, const int* __restrict__ band_lims, const Float* __restrict__ tau,
                      const Float* __restrict__ fit, Float* __restrict__ out) {
  const int icol = blockIdx.x * 256 + threadIdx.x, igpt = blockIdx.y;
  if (icol >= ncol) return;
  int bnd = 0;
  for (int b = 0; b < nbnd; ++b)
    if (igpt + 1 >= band_lims[2 * b] && igpt + 1 <= band_lims[2 * b + 1]) bnd = b;
  const Float* t = tau + icol + (size_t)ncol * nlay * igpt;
  Float s = 0;
  for (int l = 0; l < nlay; ++l) s = s + t[(size_t)ncol * l];
  out[icol + (size_t)ncol * igpt] = fit[2 * bnd] * exp(-s) + fit[2 * bnd + 1];
}
__global__ void __launch_bounds__(256)
combine_1scl_kernel(size_t n, const Float* __restrict__ a, const Float* __restrict__ r, Float* __restrict__ tau) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i < n) tau[i] = a[i] + r[i];
}
// nstr: tau, ssa as 2str; p(nmom, ...) = 0 except p(2, ...) = 0.1
__global__ void __launch_bounds__(256)
combine_nstr_kernel(size_t n, int nmom, const Float* __restrict__ a, const Float* __restrict__ r, Float* __restrict__ tau,
                    Float* __restrict__ ssa, Float* __restrict__ p) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const Float t = a[i] + r[i];
#ifdef RTE_USE_SP
  const Float tiny2 = (Float)2 * 1.17549435e-38f;
#else
  const Float tiny2 = (Float)2 * 2.2250738585072014e-308;
#endif
  ssa[i] = t > tiny2 ? r[i] / t : (Float)0;
  tau[i] = t;
  for (int m = 0; m < nmom; ++m) p[m + (size_t)nmom * i] = m == 1 ? (Float)0.1 : (Float)0;
}
// arr_out(icol, igpt) = arr_in(band(igpt), icol)
__global__ void __launch_bounds__(256)
expand_transpose_kernel(int ncol, int nbnd, const int* __restrict__ band_lims, const Float* __restrict__ in,
                        Float* __restrict__ out) {
  const int icol = blockIdx.x * 256 + threadIdx.x, ibnd = blockIdx.y;
  if (icol >= ncol) return;
  const Float v = in[ibnd + (size_t)nbnd * icol];
  for (int g = band_lims[2 * ibnd] - 1; g <= band_lims[2 * ibnd + 1] - 1; ++g) out[icol + (size_t)ncol * g] = v;
}
__global__ void __launch_bounds__(256)
secants_fill_kernel(size_t ncg, const Float* __restrict__ Ds, Float* __restrict__ secants) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i < ncg) secants[i + ncg * blockIdx.y] = Ds[blockIdx.y];
}
// toa_flux(c,g) *= tsi(c) / sum_g toa_flux(c,g)  (sum sequential over g)
__global__ void __launch_bounds__(256)
toa_renorm_kernel(int ncol, int ngpt, const Float* __restrict__ tsi, Float* __restrict__ toa) {
  const int icol = blockIdx.x * 256 + threadIdx.x;
  if (icol >= ncol) return;
  Float s = 0;
  for (int g = 0; g < ngpt; ++g) s = s + toa[icol + (size_t)ncol * g];
  const Float tv = tsi[icol];
  for (int g = 0; g < ngpt; ++g) toa[icol + (size_t)ncol * g] = toa[icol + (size_t)ncol * g] * tv / s;
}
__global__ void __launch_bounds__(256)
rfmip_mu0_kernel(int ncol, const Float* __restrict__ sza, const Bool* __restrict__ usecol, Float* __restrict__ mu0) {
  const int icol = blockIdx.x * 256 + threadIdx.x;
  if (icol >= ncol) return;
  const Float deg_to_rad = acos(-(Float)1) / (Float)180;  // rrtmgp_rfmip_sw.F90:113
  mu0[icol] = usecol[icol] ? cos(sza[icol] * deg_to_rad) : (Float)1;
}
// out(i, icol) = in(icol), i < n  (per-band copies of a per-column value: sfc_alb_spec)
__global__ void __launch_bounds__(256)
broadcast_cols_kernel(int n, int ncol, const Float* __restrict__ in, Float* __restrict__ out) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i < (size_t)n * ncol) out[i] = in[i / n];
}
__global__ void __launch_bounds__(256)
mask_columns_kernel(int ncol, int nlev, const Bool* __restrict__ usecol, Float* __restrict__ up, Float* __restrict__ dn) {
  const int icol = blockIdx.x * 256 + threadIdx.x, ilev = blockIdx.y;
  if (icol >= ncol || usecol[icol]) return;
  up[icol + (size_t)ncol * ilev] = 0;
  dn[icol + (size_t)ncol * ilev] = 0;
}
// by-band sums, sequential over the band's g-points exactly like the reference
__global__ void __launch_bounds__(256)
sum_byband_kernel(size_t n2, const int* __restrict__ band_lims, const Float* __restrict__ spectral, Float* __restrict__ out) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  const int ibnd = blockIdx.y;
  if (i >= n2) return;
  const int gS = band_lims[2 * ibnd] - 1, gE = band_lims[2 * ibnd + 1] - 1;
  Float s = spectral[i + n2 * (size_t)gS];
  for (int g = gS + 1; g <= gE; ++g) s = s + spectral[i + n2 * (size_t)g];
  out[i + n2 * (size_t)ibnd] = s;
}
__global__ void __launch_bounds__(256)
net_byband_full_kernel(size_t n2, const int* __restrict__ band_lims, const Float* __restrict__ dn, const Float* __restrict__ up,
                       Float* __restrict__ out) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  const int ibnd = blockIdx.y;
  if (i >= n2) return;
  const int gS = band_lims[2 * ibnd] - 1, gE = band_lims[2 * ibnd + 1] - 1;
  Float s = dn[i + n2 * (size_t)gS] - up[i + n2 * (size_t)gS];
  for (int g = gS + 1; g <= gE; ++g) s = s + dn[i + n2 * (size_t)g] - up[i + n2 * (size_t)g];  // (s + dn) - up, :194-196
  out[i + n2 * (size_t)ibnd] = s;
}
__global__ void __launch_bounds__(256)
sub_kernel(size_t n, const Float* __restrict__ a, const Float* __restrict__ b, Float* __restrict__ out) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i < n) out[i] = a[i] - b[i];
}

}  // namespace

extern "C" {

// ---- reference C ABI ---------------------------------------------------------------------------
void rte_compute_Planck_source_2D(const int* ncol, const int* nlay, const int* nnu, const Float* nus, const Float* dnus,
                                  const Float* T, Float* source) {
  const size_t n = (size_t)*ncol * *nlay;
  if (n == 0 || *nnu <= 0) return;
  RTE_TRY
  rte::Call c("rte_compute_Planck_source_2D");
  const Float *dn = c.in(nus, (size_t)*nnu), *dd = c.in(dnus, (size_t)*nnu), *dT = c.in(T, n);
  Float* ds = c.out(source, n * *nnu);
  rte::ProfScope p("planck_nu_kernel");
  hipLaunchKernelGGL(planck_nu_kernel, dim3(cdiv(n, 256), *nnu), dim3(256), 0, rte::stream(), n, dn, dd, dT, ds);
  RTE_CATCH("rte_compute_Planck_source_2D")
}
void rte_compute_Planck_source_1D(const int* ncol, const int* nnu, const Float* nus, const Float* dnus, const Float* T,
                                  Float* source) {
  const size_t n = (size_t)*ncol;
  if (n == 0 || *nnu <= 0) return;
  RTE_TRY
  rte::Call c("rte_compute_Planck_source_1D");
  const Float *dn = c.in(nus, (size_t)*nnu), *dd = c.in(dnus, (size_t)*nnu), *dT = c.in(T, n);
  Float* ds = c.out(source, n * *nnu);
  rte::ProfScope p("planck_nu_kernel");
  hipLaunchKernelGGL(planck_nu_kernel, dim3(cdiv(n, 256), *nnu), dim3(256), 0, rte::stream(), n, dn, dd, dT, ds);
  RTE_CATCH("rte_compute_Planck_source_1D")
}
void rte_sum_byband(const int* ncol, const int* nlev, const int* ngpt, const int* nbnd, const int* band_lims,
                    const Float* spectral_flux, Float* byband_flux) {
  const size_t n2 = (size_t)*ncol * *nlev;
  if (n2 == 0 || *nbnd <= 0) return;
  RTE_TRY
  rte::Call c("rte_sum_byband");
  const int* bl = c.in(band_lims, (size_t)2 * *nbnd);
  const Float* s = c.in(spectral_flux, n2 * *ngpt);
  Float* o = c.out(byband_flux, n2 * *nbnd);
  rte::ProfScope p("sum_byband_kernel");
  hipLaunchKernelGGL(sum_byband_kernel, dim3(cdiv(n2, 256), *nbnd), dim3(256), 0, rte::stream(), n2, bl, s, o);
  RTE_CATCH("rte_sum_byband")
}
void rte_net_byband_full(const int* ncol, const int* nlev, const int* ngpt, const int* nbnd, const int* band_lims,
                         const Float* spectral_flux_dn, const Float* spectral_flux_up, Float* byband_flux_net) {
  const size_t n2 = (size_t)*ncol * *nlev;
  if (n2 == 0 || *nbnd <= 0) return;
  RTE_TRY
  rte::Call c("rte_net_byband_full");
  const int* bl = c.in(band_lims, (size_t)2 * *nbnd);
  const Float *d = c.in(spectral_flux_dn, n2 * *ngpt), *u = c.in(spectral_flux_up, n2 * *ngpt);
  Float* o = c.out(byband_flux_net, n2 * *nbnd);
  rte::ProfScope p("net_byband_full_kernel");
  hipLaunchKernelGGL(net_byband_full_kernel, dim3(cdiv(n2, 256), *nbnd), dim3(256), 0, rte::stream(), n2, bl, d, u, o);
  RTE_CATCH("rte_net_byband_full")
}
void net_byband_precalc(const int* ncol, const int* nlev, const int* nbnd, const Float* byband_flux_dn,
                        const Float* byband_flux_up, Float* byband_flux_net) {
  const size_t n = (size_t)*ncol * *nlev * *nbnd;
  if (n == 0) return;
  RTE_TRY
  rte::Call c("net_byband_precalc");
  const Float *d = c.in(byband_flux_dn, n), *u = c.in(byband_flux_up, n);
  Float* o = c.out(byband_flux_net, n);
  rte::ProfScope p("sub_kernel");
  hipLaunchKernelGGL(sub_kernel, dim3(cdiv(n, 256)), dim3(256), 0, rte::stream(), n, d, u, o);
  RTE_CATCH("net_byband_precalc")
}

// ---- extension symbols (scalars by value) -------------------------------------------------------
int rte_hip_get_layer_number(int ncol, int nlay, const Float* vmr_h2o, const Float* plev, double m_dry, double grav,
                             Float* col_dry) {
  const size_t n = (size_t)ncol * nlay;
  if (n == 0) return 0;
  RTE_TRY
  rte::Call c("rte_hip_get_layer_number");
  const Float *v = c.in(vmr_h2o, n), *pl = c.in(plev, (size_t)ncol * (nlay + 1));
  Float* o = c.out(col_dry, n);
  rte::ProfScope p("layer_number_kernel");
  hipLaunchKernelGGL(layer_number_kernel, dim3(cdiv(ncol, 256), nlay), dim3(256), 0, rte::stream(), ncol, nlay, v, pl,
                     (Float)m_dry, (Float)grav, o);
  return 0;
  RTE_CATCH("rte_hip_get_layer_number")
  return -1;
}
int rte_hip_get_layer_mass(int ncol, int nlay, int ngas, const Float* vmr, const Float* plev, const Float* mol_weights,
                           double m_dry, double grav, Float* layer_mass) {
  const size_t n = (size_t)ngas * ncol * nlay;
  if (n == 0) return 0;
  RTE_TRY
  rte::Call c("rte_hip_get_layer_mass");
  const Float *v = c.in(vmr, n), *pl = c.in(plev, (size_t)ncol * (nlay + 1)), *mw = c.in(mol_weights, (size_t)ngas);
  Float* o = c.out(layer_mass, n);
  rte::ProfScope p("layer_mass_kernel");
  hipLaunchKernelGGL(layer_mass_kernel, dim3(cdiv(n, 256)), dim3(256), 0, rte::stream(), ncol, nlay, ngas, v, pl, mw,
                     (Float)m_dry, (Float)grav, o);
  return 0;
  RTE_CATCH("rte_hip_get_layer_mass")
  return -1;
}
int rte_hip_col_gas_fill(int ncol, int nlay, int ngas, const Float* vmr, const Float* col_dry, Float* col_gas) {
  const size_t ncl = (size_t)ncol * nlay;
  if (ncl == 0) return 0;
  RTE_TRY
  rte::Call c("rte_hip_col_gas_fill");
  const Float *v = c.in(vmr, ncl * ngas), *cd = c.in(col_dry, ncl);
  Float* o = c.out(col_gas, ncl * (ngas + 1));
  rte::ProfScope p("col_gas_fill_kernel");
  hipLaunchKernelGGL(col_gas_fill_kernel, dim3(cdiv(ncl, 256), ngas + 1), dim3(256), 0, rte::stream(), ncl, v, cd, o);
  return 0;
  RTE_CATCH("rte_hip_col_gas_fill")
  return -1;
}
int rte_hip_tlev_interp(int ncol, int nlay, const Float* play, const Float* plev, const Float* tlay, Float* tlev) {
  if (ncol <= 0 || nlay < 2) return nlay < 2 ? -1 : 0;
  const size_t ncl = (size_t)ncol * nlay;
  RTE_TRY
  rte::Call c("rte_hip_tlev_interp");
  const Float *pa = c.in(play, ncl), *pe = c.in(plev, ncl + ncol), *tl = c.in(tlay, ncl);
  Float* o = c.out(tlev, ncl + ncol);
  rte::ProfScope p("tlev_interp_kernel");
  hipLaunchKernelGGL(tlev_interp_kernel, dim3(cdiv(ncol, 256), nlay + 1), dim3(256), 0, rte::stream(), ncol, nlay, pa, pe, tl, o);
  return 0;
  RTE_CATCH("rte_hip_tlev_interp")
  return -1;
}
int rte_hip_compute_optimal_angles(int ncol, int nlay, int ngpt, int nbnd, const int* band_lims, const Float* tau,
                                   const Float* optimal_angle_fit, Float* optimal_angles) {
  if (ncol <= 0 || ngpt <= 0) return 0;
  RTE_TRY
  rte::Call c("rte_hip_compute_optimal_angles");
  const int* bl = c.in(band_lims, (size_t)2 * nbnd);
  const Float *t = c.in(tau, (size_t)ncol * nlay * ngpt), *f = c.in(optimal_angle_fit, (size_t)2 * nbnd);
  Float* o = c.out(optimal_angles, (size_t)ncol * ngpt);
  rte::ProfScope p("optimal_angles_kernel");
  hipLaunchKernelGGL(optimal_angles_kernel, dim3(cdiv(ncol, 256), ngpt), dim3(256), 0, rte::stream(), ncol, nlay, nbnd, bl, t, f, o);
  return 0;
  RTE_CATCH("rte_hip_compute_optimal_angles")
  return -1;
}
int rte_hip_combine_abs_and_rayleigh_1scl(int ncol, int nlay, int ngpt, const Float* tau_abs, const Float* tau_ray, Float* tau) {
  const size_t n = (size_t)ncol * nlay * ngpt;
  if (n == 0) return 0;
  RTE_TRY
  rte::Call c("rte_hip_combine_abs_and_rayleigh_1scl");
  const Float *a = c.in(tau_abs, n), *r = c.in(tau_ray, n);
  Float* t = c.out(tau, n);
  rte::ProfScope p("combine_1scl_kernel");
  hipLaunchKernelGGL(combine_1scl_kernel, dim3(cdiv(n, 256)), dim3(256), 0, rte::stream(), n, a, r, t);
  return 0;
  RTE_CATCH("rte_hip_combine_abs_and_rayleigh_1scl")
  return -1;
}
int rte_hip_combine_abs_and_rayleigh_nstr(int ncol, int nlay, int ngpt, int nmom, const Float* tau_abs, const Float* tau_ray,
                                          Float* tau, Float* ssa, Float* p) {
  const size_t n = (size_t)ncol * nlay * ngpt;
  if (n == 0) return 0;
  RTE_TRY
  rte::Call c("rte_hip_combine_abs_and_rayleigh_nstr");
  const Float *a = c.in(tau_abs, n), *r = c.in(tau_ray, n);
  Float *t = c.out(tau, n), *s = c.out(ssa, n), *pp = c.out(p, n * nmom);
  rte::ProfScope pr("combine_nstr_kernel");
  hipLaunchKernelGGL(combine_nstr_kernel, dim3(cdiv(n, 256)), dim3(256), 0, rte::stream(), n, nmom, a, r, t, s, pp);
  return 0;
  RTE_CATCH("rte_hip_combine_abs_and_rayleigh_nstr")
  return -1;
}
int rte_hip_expand_and_transpose(int ncol, int nbnd, int ngpt, const int* band_lims, const Float* arr_in, Float* arr_out) {
  if (ncol <= 0 || nbnd <= 0) return 0;
  RTE_TRY
  rte::Call c("rte_hip_expand_and_transpose");
  const int* bl = c.in(band_lims, (size_t)2 * nbnd);
  const Float* in = c.in(arr_in, (size_t)nbnd * ncol);
  Float* out = c.out(arr_out, (size_t)ncol * ngpt);
  rte::ProfScope p("expand_transpose_kernel");
  hipLaunchKernelGGL(expand_transpose_kernel, dim3(cdiv(ncol, 256), nbnd), dim3(256), 0, rte::stream(), ncol, nbnd, bl, in, out);
  return 0;
  RTE_CATCH("rte_hip_expand_and_transpose")
  return -1;
}
int rte_hip_secants_fill(int ncol, int ngpt, int nmus, const Float* Ds, Float* secants) {
  const size_t ncg = (size_t)ncol * ngpt;
  if (ncg == 0 || nmus <= 0) return 0;
  RTE_TRY
  rte::Call c("rte_hip_secants_fill");
  const Float* d = c.in(Ds, (size_t)nmus);
  Float* o = c.out(secants, ncg * nmus);
  rte::ProfScope p("secants_fill_kernel");
  hipLaunchKernelGGL(secants_fill_kernel, dim3(cdiv(ncg, 256), nmus), dim3(256), 0, rte::stream(), ncg, d, o);
  return 0;
  RTE_CATCH("rte_hip_secants_fill")
  return -1;
}
int rte_hip_rfmip_sw_toa_renorm(int ncol, int ngpt, const Float* total_solar_irradiance, Float* toa_flux) {
  if (ncol <= 0 || ngpt <= 0) return 0;
  RTE_TRY
  rte::Call c("rte_hip_rfmip_sw_toa_renorm");
  const Float* tsi = c.in(total_solar_irradiance, (size_t)ncol);
  Float* toa = c.inout(toa_flux, (size_t)ncol * ngpt);
  rte::ProfScope p("toa_renorm_kernel");
  hipLaunchKernelGGL(toa_renorm_kernel, dim3(cdiv(ncol, 256)), dim3(256), 0, rte::stream(), ncol, ngpt, tsi, toa);
  return 0;
  RTE_CATCH("rte_hip_rfmip_sw_toa_renorm")
  return -1;
}
int rte_hip_rfmip_sw_mu0(int ncol, const Float* solar_zenith_angle, const Bool* usecol, Float* mu0) {
  if (ncol <= 0) return 0;
  RTE_TRY
  rte::Call c("rte_hip_rfmip_sw_mu0");
  const Float* z = c.in(solar_zenith_angle, (size_t)ncol);
  const Bool* u = c.in(usecol, (size_t)ncol);
  Float* m = c.out(mu0, (size_t)ncol);
  rte::ProfScope p("rfmip_mu0_kernel");
  hipLaunchKernelGGL(rfmip_mu0_kernel, dim3(cdiv(ncol, 256)), dim3(256), 0, rte::stream(), ncol, z, u, m);
  return 0;
  RTE_CATCH("rte_hip_rfmip_sw_mu0")
  return -1;
}
int rte_hip_broadcast_cols(int n, int ncol, const Float* per_col, Float* out) {
  if (n <= 0 || ncol <= 0) return 0;
  RTE_TRY
  rte::Call c("rte_hip_broadcast_cols");
  const Float* in = c.in(per_col, (size_t)ncol);
  Float* o = c.out(out, (size_t)n * ncol);
  rte::ProfScope p("broadcast_cols_kernel");
  hipLaunchKernelGGL(broadcast_cols_kernel, dim3(cdiv((size_t)n * ncol, 256)), dim3(256), 0, rte::stream(), n, ncol, in, o);
  return 0;
  RTE_CATCH("rte_hip_broadcast_cols")
  return -1;
}
int rte_hip_mask_columns(int ncol, int nlev, const Bool* usecol, Float* flux_up, Float* flux_dn) {
  if (ncol <= 0 || nlev <= 0) return 0;
  RTE_TRY
  rte::Call c("rte_hip_mask_columns");
  const Bool* u = c.in(usecol, (size_t)ncol);
  Float *up = c.inout(flux_up, (size_t)ncol * nlev), *dn = c.inout(flux_dn, (size_t)ncol * nlev);
  rte::ProfScope p("mask_columns_kernel");
  hipLaunchKernelGGL(mask_columns_kernel, dim3(cdiv(ncol, 256), nlev), dim3(256), 0, rte::stream(), ncol, nlev, u, up, dn);
  return 0;
  RTE_CATCH("rte_hip_mask_columns")
  return -1;
}

}  // extern "C"
