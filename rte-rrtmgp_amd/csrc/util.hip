// util.hip -- array utilities, flux reductions and frontend-glue kernels (gfx950).
//
// C ABI: zero_array_*, set_to_scalar_* (reference rte/kernels/mo_rte_util_array.F90:32-132),
//        rte_sum_broadband, rte_net_broadband_full, rte_net_broadband_precalc
//        (reference rte/kernels/mo_fluxes_broadband_kernels.F90:32-128).
// Extension symbols (rte_hip_*): device versions of frontend glue loops that are not behind the
// reference's C API but must run on the device in a device-resident driver.
#include <string.h>

#include "common.h"

namespace {
using rte::cdiv;

__global__ void __launch_bounds__(256) fill_kernel(Float* __restrict__ a, size_t n, Float v) {
  // 2 elements per thread, grid-stride: wide coalesced stores
  size_t i = ((size_t)blockIdx.x * blockDim.x + threadIdx.x);
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  for (; i < n; i += stride) a[i] = v;
}

// rank: of the array as the caller declared it.  Host arrays (what the unchanged Fortran frontend passes): a zero fill of
// a 1-D or 2-D array is done where the array lives, on the host -- callers read such arrays there right away (the RFMIP
// driver accumulates into def_tsi, rrtmgp_rfmip_sw.F90:274-283; the simple spectral model fills part of vmr,
// ssm/mo_optics_ssm.F90:606-612) -- and so is every fill outside host-mirror mode.  Only the 3-D / 4-D arrays of the
// tau / ssa / g path are recorded on a device copy in host-mirror mode (the frontend hands them straight to the next kernel:
// mo_gas_optics_rrtmgp.F90:637,679; the mode's contract).
void fill(const char* name, Float* a, size_t n, Float v, int rank) {
  if (n == 0) return;
  if (v == (Float)0 && rte::defer_zero_enabled() && rte::is_device_memory(a)) {
    rte::defer_zero(a, n * sizeof(Float));  // materialised by the next library call unless consumed
    return;
  }
  RTE_TRY
  rte::Call c(name);
  if (v == (Float)0 && rank >= 3 && c.lazy_zero(a, n * sizeof(Float))) return;  // host-mirror mode: recorded on the device copy
  if (v == (Float)0 && !rte::is_device_pointer(a)) {  // pageable host array: zeroed in place (a mirror of it, if any, loses its canaries)
    memset(a, 0, n * sizeof(Float));
    return;
  }
  Float* d = v == (Float)0 ? c.out_lazy(a, n) : c.out(a, n);
  rte::ProfScope p("fill_kernel");
  if (v == (Float)0) {
    HIP_CHECK(hipMemsetAsync(d, 0, n * sizeof(Float), rte::stream()));
  } else {
    const unsigned blocks = (unsigned)((n + 255) / 256 < 16384 ? (n + 255) / 256 : 16384);
    hipLaunchKernelGGL(fill_kernel, dim3(blocks), dim3(256), 0, rte::stream(), d, n, v);
  }
  RTE_CATCH(name)
}

// sequential sum over g-points, exactly the reference's order
__global__ void __launch_bounds__(256)
sum_broadband_kernel(size_t n2, int ngpt, const Float* __restrict__ spectral, Float* __restrict__ out) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n2) return;
  Float s = 0;
  for (int g = 0; g < ngpt; ++g) s = s + spectral[i + n2 * (size_t)g];
  out[i] = s;
}
__global__ void __launch_bounds__(256)
net_broadband_full_kernel(size_t n2, int ngpt, const Float* __restrict__ dn, const Float* __restrict__ up,
                          Float* __restrict__ out) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n2) return;
  Float s = dn[i] - up[i];
  for (int g = 1; g < ngpt; ++g) s = s + (dn[i + n2 * (size_t)g] - up[i + n2 * (size_t)g]);
  out[i] = s;
}
__global__ void __launch_bounds__(256)
net_precalc_kernel(size_t n2, const Float* __restrict__ dn, const Float* __restrict__ up, Float* __restrict__ out) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n2) out[i] = dn[i] - up[i];
}

// combine_abs_and_rayleigh, 2-stream branch: reference rrtmgp/frontend/mo_gas_optics_rrtmgp.F90:1983-2002
__global__ void __launch_bounds__(256)
combine_2str_kernel(size_t n, const Float* __restrict__ tau_abs, const Float* __restrict__ tau_ray,
                    Float* __restrict__ tau, Float* __restrict__ ssa, Float* __restrict__ g) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const Float tr = tau_ray[i];
  const Float t = tau_abs[i] + tr;
#ifdef RTE_USE_SP
  const Float tiny2 = (Float)2 * (Float)1.17549435e-38f;
#else
  const Float tiny2 = (Float)2 * (Float)2.2250738585072014e-308;
#endif
  ssa[i] = (t > tiny2) ? tr / t : (Float)0;
  tau[i] = t;
  g[i] = 0;
}
// out(icol, igpt) = per_gpt(igpt): toa_src broadcast, reference mo_gas_optics_rrtmgp.F90:405-411
// Frontend glue of cloud_optics (rrtmgp/frontend/mo_cloud_optics_rrtmgp.F90:334-341): masks = water path > 0
__global__ void __launch_bounds__(256)
cloud_masks_kernel(size_t n, const Float* __restrict__ clwp, const Float* __restrict__ ciwp, Bool* __restrict__ liqmsk,
                   Bool* __restrict__ icemsk) {
  const size_t i = blockIdx.x * (size_t)256 + threadIdx.x;
  if (i >= n) return;
  liqmsk[i] = clwp[i] > (Float)0;
  icemsk[i] = ciwp[i] > (Float)0;
}

// ... and its liquid + ice combination (:392-425): absorption optical depth for 1scl, (tau, ssa, g) for 2str
template <bool TWOSTR>
__global__ void __launch_bounds__(256)
cloud_combine_kernel(size_t n, const Float* __restrict__ ltau, const Float* __restrict__ ltaussa,
                     const Float* __restrict__ ltaussag, const Float* __restrict__ itau, const Float* __restrict__ itaussa,
                     const Float* __restrict__ itaussag, Float* __restrict__ tau, Float* __restrict__ ssa,
                     Float* __restrict__ g) {
  const size_t i = blockIdx.x * (size_t)256 + threadIdx.x;
  if (i >= n) return;
  if (!TWOSTR) {
    tau[i] = (ltau[i] - ltaussa[i]) + (itau[i] - itaussa[i]);  // (1 - ssa) tau = tau - taussa
  } else {
    const Float t = ltau[i] + itau[i];
    const Float ts = ltaussa[i] + itaussa[i];
#ifdef RTE_USE_SP
    const Float eps = 1.1920929e-07f;  // epsilon(tau)
#else
    const Float eps = 2.220446049250313e-16;  // epsilon(tau)
#endif
    g[i] = (ltaussag[i] + itaussag[i]) / fmax(eps, ts);
    ssa[i] = ts / fmax(eps, t);
    tau[i] = t;
  }
}

__global__ void __launch_bounds__(256)
broadcast_gpt_kernel(int ncol, int ngpt, const Float* __restrict__ per_gpt, Float* __restrict__ out) {
  const int icol = blockIdx.x * blockDim.x + threadIdx.x;
  const int g = blockIdx.y;
  if (icol < ncol) out[icol + (size_t)ncol * g] = per_gpt[g];
}
}  // namespace

extern "C" {
void zero_array_1D(const int* ni, Float* a) { fill("zero_array_1D", a, (size_t)*ni, 0, 1); }
void zero_array_2D(const int* ni, const int* nj, Float* a) { fill("zero_array_2D", a, (size_t)*ni * *nj, 0, 2); }
void zero_array_3D(const int* ni, const int* nj, const int* nk, Float* a) {
  fill("zero_array_3D", a, (size_t)*ni * *nj * *nk, 0, 3);
}
void zero_array_4D(const int* ni, const int* nj, const int* nk, const int* nl, Float* a) {
  fill("zero_array_4D", a, (size_t)*ni * *nj * *nk * *nl, 0, 4);
}
void set_to_scalar_1D(const int* ni, Float* a, const Float* v) { fill("set_to_scalar_1D", a, (size_t)*ni, *v, 1); }
void set_to_scalar_2D(const int* ni, const int* nj, Float* a, const Float* v) {
  fill("set_to_scalar_2D", a, (size_t)*ni * *nj, *v, 2);
}
void set_to_scalar_3D(const int* ni, const int* nj, const int* nk, Float* a, const Float* v) {
  fill("set_to_scalar_3D", a, (size_t)*ni * *nj * *nk, *v, 3);
}
void set_to_scalar_4D(const int* ni, const int* nj, const int* nk, const int* nl, Float* a, const Float* v) {
  fill("set_to_scalar_4D", a, (size_t)*ni * *nj * *nk * *nl, *v, 4);
}

void rte_sum_broadband(const int* ncol, const int* nlev, const int* ngpt, const Float* spectral_flux,
                       Float* broadband_flux) {
  const size_t n2 = (size_t)*ncol * *nlev;
  if (n2 == 0) return;
  RTE_TRY
  rte::Call c("rte_sum_broadband");
  const Float* s = c.in(spectral_flux, n2 * *ngpt);
  Float* o = c.out(broadband_flux, n2);
  rte::ProfScope p("sum_broadband_kernel");
  hipLaunchKernelGGL(sum_broadband_kernel, dim3(cdiv(n2, 256)), dim3(256), 0, rte::stream(), n2, *ngpt, s, o);
  RTE_CATCH("rte_sum_broadband")
}
void rte_net_broadband_full(const int* ncol, const int* nlev, const int* ngpt, const Float* spectral_flux_dn,
                            const Float* spectral_flux_up, Float* broadband_flux_net) {
  const size_t n2 = (size_t)*ncol * *nlev;
  if (n2 == 0) return;
  RTE_TRY
  rte::Call c("rte_net_broadband_full");
  const Float* d = c.in(spectral_flux_dn, n2 * *ngpt);
  const Float* u = c.in(spectral_flux_up, n2 * *ngpt);
  Float* o = c.out(broadband_flux_net, n2);
  rte::ProfScope p("net_broadband_full_kernel");
  hipLaunchKernelGGL(net_broadband_full_kernel, dim3(cdiv(n2, 256)), dim3(256), 0, rte::stream(), n2, *ngpt, d, u, o);
  RTE_CATCH("rte_net_broadband_full")
}
void rte_net_broadband_precalc(const int* ncol, const int* nlev, const Float* flux_dn, const Float* flux_up,
                               Float* broadband_flux_net) {
  const size_t n2 = (size_t)*ncol * *nlev;
  if (n2 == 0) return;
  RTE_TRY
  rte::Call c("rte_net_broadband_precalc");
  const Float* d = c.in(flux_dn, n2);
  const Float* u = c.in(flux_up, n2);
  Float* o = c.out(broadband_flux_net, n2);
  rte::ProfScope p("net_precalc_kernel");
  hipLaunchKernelGGL(net_precalc_kernel, dim3(cdiv(n2, 256)), dim3(256), 0, rte::stream(), n2, d, u, o);
  RTE_CATCH("rte_net_broadband_precalc")
}

// ---- extension symbols (scalars by value) ---------------------------------------------------
int rte_hip_combine_abs_and_rayleigh_2str(int ncol, int nlay, int ngpt, const Float* tau_abs, const Float* tau_ray,
                                          Float* tau, Float* ssa, Float* g) {
  const size_t n = (size_t)ncol * nlay * ngpt;
  if (n == 0) return 0;
  RTE_TRY
  rte::Call c("rte_hip_combine_abs_and_rayleigh_2str");
  const Float* a = c.in(tau_abs, n);
  const Float* r = c.in(tau_ray, n);
  Float *t = c.out(tau, n), *s = c.out(ssa, n), *gg = c.out(g, n);
  rte::ProfScope p("combine_2str_kernel");
  hipLaunchKernelGGL(combine_2str_kernel, dim3(cdiv(n, 256)), dim3(256), 0, rte::stream(), n, a, r, t, s, gg);
  return 0;
  RTE_CATCH("rte_hip_combine_abs_and_rayleigh_2str")
  return -1;
}
int rte_hip_broadcast_gpt(int ncol, int ngpt, const Float* per_gpt, Float* out) {
  if (ncol <= 0 || ngpt <= 0) return 0;
  RTE_TRY
  rte::Call c("rte_hip_broadcast_gpt");
  const Float* pg = c.in(per_gpt, (size_t)ngpt);
  Float* o = c.out(out, (size_t)ncol * ngpt);
  rte::ProfScope p("broadcast_gpt_kernel");
  hipLaunchKernelGGL(broadcast_gpt_kernel, dim3(cdiv(ncol, 256), ngpt), dim3(256), 0, rte::stream(), ncol, ngpt, pg, o);
  return 0;
  RTE_CATCH("rte_hip_broadcast_gpt")
  return -1;
}

// liqmsk = clwp > 0, icemsk = ciwp > 0 (mo_cloud_optics_rrtmgp.F90:334-341)
int rte_hip_cloud_masks(int ncol, int nlay, const Float* clwp, const Float* ciwp, Bool* liqmsk, Bool* icemsk) {
  const size_t n = (size_t)ncol * nlay;
  if (n == 0) return 0;
  RTE_TRY
  rte::Call c("rte_hip_cloud_masks");
  const Float *l = c.in(clwp, n), *i = c.in(ciwp, n);
  Bool *lm = c.out(liqmsk, n), *im = c.out(icemsk, n);
  rte::ProfScope p("cloud_masks_kernel");
  hipLaunchKernelGGL(cloud_masks_kernel, dim3(cdiv(n, 256)), dim3(256), 0, rte::stream(), n, l, i, lm, im);
  return 0;
  RTE_CATCH("rte_hip_cloud_masks")
  return -1;
}
// liquid + ice -> cloud optical properties (mo_cloud_optics_rrtmgp.F90:392-425); twostr = 0: tau only
int rte_hip_cloud_combine(int ncol, int nlay, int nspec, int twostr, const Float* ltau, const Float* ltaussa,
                          const Float* ltaussag, const Float* itau, const Float* itaussa, const Float* itaussag,
                          Float* tau, Float* ssa, Float* g) {
  const size_t n = (size_t)ncol * nlay * nspec;
  if (n == 0) return 0;
  RTE_TRY
  rte::Call c("rte_hip_cloud_combine");
  const Float *a0 = c.in(ltau, n), *a1 = c.in(ltaussa, n), *a2 = c.in(ltaussag, n);
  const Float *b0 = c.in(itau, n), *b1 = c.in(itaussa, n), *b2 = c.in(itaussag, n);
  Float* t = c.out(tau, n);
  Float* s_ = twostr ? c.out(ssa, n) : nullptr;
  Float* g_ = twostr ? c.out(g, n) : nullptr;
  rte::ProfScope p("cloud_combine_kernel");
  if (twostr)
    hipLaunchKernelGGL((cloud_combine_kernel<true>), dim3(cdiv(n, 256)), dim3(256), 0, rte::stream(), n, a0, a1, a2, b0, b1, b2, t, s_, g_);
  else
    hipLaunchKernelGGL((cloud_combine_kernel<false>), dim3(cdiv(n, 256)), dim3(256), 0, rte::stream(), n, a0, a1, a2, b0, b1, b2, t, s_, g_);
  return 0;
  RTE_CATCH("rte_hip_cloud_combine")
  return -1;
}
}
