/*
 * glue_oracle.c -- CPU restatement (plain C) of the small computations either side of the hot path: the Planck
 * function on a wavenumber grid, by-band flux reductions, and the frontend's glue loops (dry-air column amounts,
 * column gas amounts, level temperatures, optimal transport angles, band -> g-point expansion, RFMIP-SW boundary
 * conditions).  Compiled into liboracle[_sp].so next to rte_rrtmgp_oracle.c.
 *
 * THIS IS TEST INFRASTRUCTURE, NOT PRODUCT CODE (see rte_rrtmgp_oracle.c).  Each function cites the reference lines it
 * follows, expression by expression.  Pin: rte_compute_Planck_source_1D/2D, get_layer_number and get_layer_mass are
 * checked against the reference build (oracle/_ref/librefkernels.so and its wrapper symbols rte_ref_get_layer_*,
 * oracle/ref_wrappers.F90) by tests/test_glue.py; the frontend loops and the by-band sums live inside Fortran
 * type-bound procedures of the reference that need its whole frontend + netCDF-backed set-up to call, so they are
 * pinned only by the explicit formulas restated in numpy in that test ("parity unpinned" against a reference run).
 *
 * Reference C ABI symbols take scalars by address; the rte_hip_* names are this project's extension entry points
 * (scalars by value) and are exported here under the same names so that one driver runs both libraries.
 */
#include <math.h>
#include <float.h>
#include <stddef.h>
#include "../include/rte_rrtmgp_kernels.h"

#ifdef RTE_USE_SP
#define EXP(x) expf(x)
#define FABS(x) fabsf(x)
#define COS(x) cosf(x)
#define ACOS(x) acosf(x)
#define FTINY FLT_MIN
#else
#define EXP(x) exp(x)
#define FABS(x) fabs(x)
#define COS(x) cos(x)
#define ACOS(x) acos(x)
#define FTINY DBL_MIN
#endif

/* rte/kernels/mo_gas_optics_constants.F90:17-35 */
static const Float boltzmann_k = (Float)1.380649e-23, planck_h = (Float)6.626075540e-34, lightspeed = (Float)2.99792458e8;
static const Float m_h2o = (Float)0.018016, avogad = (Float)6.02214076e23;

/* rte/kernels/mo_gas_optics_utils.F90:31-35 */
static Float B_nu(Float T, Float nu) {
  const Float nu100 = nu * (Float)100;
  return (Float)100 * (Float)2 * planck_h * (nu100 * nu100 * nu100) * (lightspeed * lightspeed) /
         (EXP((planck_h * lightspeed * nu * (Float)100) / (boltzmann_k * T)) - (Float)1);
}
/* :36-64 */
void rte_compute_Planck_source_2D(const int* ncol, const int* nlay, const int* nnu, const Float* nus, const Float* dnus,
                                  const Float* T, Float* source) {
  const size_t n = (size_t)*ncol * *nlay;
  for (int inu = 0; inu < *nnu; ++inu)
    for (size_t i = 0; i < n; ++i) source[i + n * inu] = B_nu(T[i], nus[inu]) * dnus[inu];
}
/* :66-95 */
void rte_compute_Planck_source_1D(const int* ncol, const int* nnu, const Float* nus, const Float* dnus, const Float* T,
                                  Float* source) {
  const size_t n = (size_t)*ncol;
  for (int inu = 0; inu < *nnu; ++inu)
    for (size_t i = 0; i < n; ++i) source[i + n * inu] = B_nu(T[i], nus[inu]) * dnus[inu];
}
/* rte/extensions/mo_fluxes_byband.F90:156-174 */
void rte_sum_byband(const int* ncol, const int* nlev, const int* ngpt, const int* nbnd, const int* band_lims,
                    const Float* spectral_flux, Float* byband_flux) {
  const size_t n2 = (size_t)*ncol * *nlev;
  (void)ngpt;
  for (int b = 0; b < *nbnd; ++b)
    for (size_t i = 0; i < n2; ++i) {
      Float s = spectral_flux[i + n2 * (size_t)(band_lims[2 * b] - 1)];
      for (int g = band_lims[2 * b]; g <= band_lims[2 * b + 1] - 1; ++g) s = s + spectral_flux[i + n2 * (size_t)g];
      byband_flux[i + n2 * (size_t)b] = s;
    }
}
/* :179-201 */
void rte_net_byband_full(const int* ncol, const int* nlev, const int* ngpt, const int* nbnd, const int* band_lims,
                         const Float* dn, const Float* up, Float* net) {
  const size_t n2 = (size_t)*ncol * *nlev;
  (void)ngpt;
  for (int b = 0; b < *nbnd; ++b)
    for (size_t i = 0; i < n2; ++i) {
      size_t o = i + n2 * (size_t)(band_lims[2 * b] - 1);
      Float s = dn[o] - up[o];
      for (int g = band_lims[2 * b]; g <= band_lims[2 * b + 1] - 1; ++g) {
        o = i + n2 * (size_t)g;
        s = s + dn[o] - up[o];
      }
      net[i + n2 * (size_t)b] = s;
    }
}
/* :203-209 */
void net_byband_precalc(const int* ncol, const int* nlev, const int* nbnd, const Float* dn, const Float* up, Float* net) {
  const size_t n = (size_t)*ncol * *nlev * *nbnd;
  for (size_t i = 0; i < n; ++i) net[i] = dn[i] - up[i];
}

/* get_layer_number, rte/kernels/mo_gas_optics_utils.F90:127-152 (m_dry, grav are run-time settable there) */
int rte_hip_get_layer_number(int ncol, int nlay, const Float* vmr_h2o, const Float* plev, double m_dry_, double grav_,
                             Float* col_dry) {
  const Float m_dry = (Float)m_dry_, grav = (Float)grav_;
  for (int l = 0; l < nlay; ++l)
    for (int c = 0; c < ncol; ++c) {
      const size_t i = c + (size_t)ncol * l;
      const Float delta_plev = FABS(plev[i] - plev[i + ncol]);
      const Float fact = (Float)1 / ((Float)1 + vmr_h2o[i]);
      const Float m_air = (m_dry + m_h2o * vmr_h2o[i]) * fact;
      col_dry[i] = (Float)10 * delta_plev * avogad * fact / ((Float)1000 * m_air * (Float)100 * grav);
    }
  return 0;
}
/* get_layer_mass :99-125; vmr, layer_mass (ngas, ncol, nlay) */
int rte_hip_get_layer_mass(int ncol, int nlay, int ngas, const Float* vmr, const Float* plev, const Float* mol_weights,
                           double m_dry_, double grav_, Float* layer_mass) {
  const Float m_dry = (Float)m_dry_, grav = (Float)grav_;
  for (int l = 0; l < nlay; ++l)
    for (int c = 0; c < ncol; ++c)
      for (int g = 0; g < ngas; ++g) {
        const size_t cl = c + (size_t)ncol * l, i = g + (size_t)ngas * cl;
        layer_mass[i] = vmr[i] * (mol_weights[g] / m_dry) * FABS(plev[cl + ncol] - plev[cl]) / grav;
      }
  return 0;
}
/* rrtmgp/frontend/mo_gas_optics_rrtmgp.F90:594-609; vmr (ncol, nlay, ngas), col_gas (ncol, nlay, 0:ngas) */
int rte_hip_col_gas_fill(int ncol, int nlay, int ngas, const Float* vmr, const Float* col_dry, Float* col_gas) {
  const size_t ncl = (size_t)ncol * nlay;
  for (size_t i = 0; i < ncl; ++i) col_gas[i] = col_dry[i];
  for (int g = 1; g <= ngas; ++g)
    for (size_t i = 0; i < ncl; ++i) col_gas[i + ncl * g] = vmr[i + ncl * (g - 1)] * col_dry[i];
  return 0;
}
/* :893-912 */
int rte_hip_tlev_interp(int ncol, int nlay, const Float* play, const Float* plev, const Float* tlay, Float* tlev) {
  if (nlay < 2) return -1;
#define A2(a, c, l) (a)[(size_t)(c) + (size_t)ncol * (size_t)(l)]
  for (int c = 0; c < ncol; ++c) {
    A2(tlev, c, 0) = A2(tlay, c, 0) + (A2(plev, c, 0) - A2(play, c, 0)) * (A2(tlay, c, 1) - A2(tlay, c, 0)) /
                                          (A2(play, c, 1) - A2(play, c, 0));
    A2(tlev, c, nlay) = A2(tlay, c, nlay - 1) + (A2(plev, c, nlay) - A2(play, c, nlay - 1)) *
                                                    (A2(tlay, c, nlay - 1) - A2(tlay, c, nlay - 2)) /
                                                    (A2(play, c, nlay - 1) - A2(play, c, nlay - 2));
  }
  for (int l = 1; l < nlay; ++l)
    for (int c = 0; c < ncol; ++c)
      A2(tlev, c, l) = (A2(play, c, l - 1) * A2(tlay, c, l - 1) * (A2(plev, c, l) - A2(play, c, l)) +
                        A2(play, c, l) * A2(tlay, c, l) * (A2(play, c, l - 1) - A2(plev, c, l))) /
                       (A2(plev, c, l) * (A2(play, c, l - 1) - A2(play, c, l)));
  return 0;
}
/* :1536-1561; band of a g-point from band_lims (convert_gpt2band, rte/frontend/mo_optical_props.F90) */
int rte_hip_compute_optimal_angles(int ncol, int nlay, int ngpt, int nbnd, const int* band_lims, const Float* tau,
                                   const Float* fit, Float* out) {
  for (int g = 0; g < ngpt; ++g) {
    int bnd = 0;
    for (int b = 0; b < nbnd; ++b)
      if (g + 1 >= band_lims[2 * b] && g + 1 <= band_lims[2 * b + 1]) bnd = b;
    for (int c = 0; c < ncol; ++c) {
      Float t = 0;
      for (int l = 0; l < nlay; ++l) t = t + tau[c + (size_t)ncol * (l + (size_t)nlay * g)];
      const Float trans_total = EXP(-t);
      out[c + (size_t)ncol * g] = fit[2 * bnd] * trans_total + fit[2 * bnd + 1];
    }
  }
  return 0;
}
/* :1966-1979 */
int rte_hip_combine_abs_and_rayleigh_1scl(int ncol, int nlay, int ngpt, const Float* tau_abs, const Float* tau_ray, Float* tau) {
  const size_t n = (size_t)ncol * nlay * ngpt;
  for (size_t i = 0; i < n; ++i) tau[i] = tau_abs[i] + tau_ray[i];
  return 0;
}
/* :2003-2035 */
int rte_hip_combine_abs_and_rayleigh_nstr(int ncol, int nlay, int ngpt, int nmom, const Float* tau_abs, const Float* tau_ray,
                                          Float* tau, Float* ssa, Float* p) {
  const size_t n = (size_t)ncol * nlay * ngpt;
  for (size_t i = 0; i < n; ++i) {
    const Float t = tau_abs[i] + tau_ray[i];
    ssa[i] = t > (Float)2 * FTINY ? tau_ray[i] / t : (Float)0;
    tau[i] = t;
    for (int m = 0; m < nmom; ++m) p[m + (size_t)nmom * i] = m == 1 ? (Float)0.1 : (Float)0;
  }
  return 0;
}
/* rte/frontend/mo_rte_lw.F90:478-501; arr_in (nbnd, ncol) -> arr_out (ncol, ngpt) */
int rte_hip_expand_and_transpose(int ncol, int nbnd, int ngpt, const int* band_lims, const Float* arr_in, Float* arr_out) {
  (void)ngpt;
  for (int b = 0; b < nbnd; ++b)
    for (int c = 0; c < ncol; ++c)
      for (int g = band_lims[2 * b] - 1; g <= band_lims[2 * b + 1] - 1; ++g)
        arr_out[c + (size_t)ncol * g] = arr_in[b + (size_t)nbnd * c];
  return 0;
}
/* rte/frontend/mo_rte_lw.F90:357-365 */
int rte_hip_secants_fill(int ncol, int ngpt, int nmus, const Float* Ds, Float* secants) {
  const size_t ncg = (size_t)ncol * ngpt;
  for (int m = 0; m < nmus; ++m)
    for (size_t i = 0; i < ncg; ++i) secants[i + ncg * m] = Ds[m];
  return 0;
}
/* examples/rfmip-clear-sky/rrtmgp_rfmip_sw.F90:273-300 */
int rte_hip_rfmip_sw_toa_renorm(int ncol, int ngpt, const Float* tsi, Float* toa) {
  for (int c = 0; c < ncol; ++c) {
    Float def_tsi = 0;
    for (int g = 0; g < ngpt; ++g) def_tsi = def_tsi + toa[c + (size_t)ncol * g];
    for (int g = 0; g < ngpt; ++g) toa[c + (size_t)ncol * g] = toa[c + (size_t)ncol * g] * tsi[c] / def_tsi;
  }
  return 0;
}
/* :312-317 */
int rte_hip_rfmip_sw_mu0(int ncol, const Float* sza, const Bool* usecol, Float* mu0) {
  const Float deg_to_rad = ACOS(-(Float)1) / (Float)180;
  for (int c = 0; c < ncol; ++c) mu0[c] = usecol[c] ? COS(sza[c] * deg_to_rad) : (Float)1;
  return 0;
}
/* :303-308: sfc_alb_spec(ibnd, icol) = surface_albedo(icol) */
int rte_hip_broadcast_cols(int n, int ncol, const Float* per_col, Float* out) {
  for (int c = 0; c < ncol; ++c)
    for (int i = 0; i < n; ++i) out[i + (size_t)n * c] = per_col[c];
  return 0;
}
/* :331-337 */
int rte_hip_mask_columns(int ncol, int nlev, const Bool* usecol, Float* flux_up, Float* flux_dn) {
  for (int c = 0; c < ncol; ++c)
    if (!usecol[c])
      for (int l = 0; l < nlev; ++l) { flux_up[c + (size_t)ncol * l] = 0; flux_dn[c + (size_t)ncol * l] = 0; }
  return 0;
}
