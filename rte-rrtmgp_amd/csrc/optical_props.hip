// optical_props.hip -- optical-properties arithmetic and cloud look-up-table optics (gfx950).
//
// C ABI (reference rte/kernels/api/mo_optical_props_kernels.F90,
//        rrtmgp/kernels/api/mo_cloud_optics_rrtmgp_kernels.F90):
//   rte_delta_scale_2str_k, rte_delta_scale_2str_f_k, 9 x rte_increment_*, 9 x rte_inc_*_bybnd,
//   rte_extract_subset_{dim1_3d,dim2_4d,absorption_tau}, rrtmgp_compute_cld_from_table.
// All are elementwise over (col, lay, gpt): one thread per (column*layer), blockIdx.y = g-point, so
// lanes are consecutive columns (coalesced 512-byte wave accesses) and the band of a g-point is
// block-uniform.  Arithmetic follows the reference default kernels
// (rte/kernels/mo_optical_props_kernels.F90:44-778, rrtmgp/kernels/mo_cloud_optics_rrtmgp_kernels.F90:24-65).
#include <math.h>

#include "common.h"

namespace {
using rte::cdiv;

#ifdef RTE_USE_SP
__device__ constexpr Float kEps = (Float)3 * (Float)1.17549435e-38f;
#else
__device__ constexpr Float kEps = (Float)3 * (Float)2.2250738585072014e-308;
#endif

enum Op { OP_1S_1S, OP_1S_2S, OP_2S_1S, OP_2S_2S, OP_NS_G2, OP_NS_P2 };

struct IncArgs {
  int ncl /* ncol*nlay */, ngpt, nbnd, nmom1, nmom2;
  const int* lims;  // (2,nbnd) or null for g-point operands
  Float *tau1, *ssa1, *g1 /* or p1 */;
  const Float *tau2, *ssa2, *g2 /* g2, or p2 with leading dimension nmom2 */;
};

template <int OP>
__device__ __forceinline__ void increment_one(const IncArgs& a, const size_t i, const size_t i2) {
  if (OP == OP_1S_1S) {
    a.tau1[i] = a.tau1[i] + a.tau2[i2];
  } else if (OP == OP_1S_2S) {
    a.tau1[i] = a.tau1[i] + a.tau2[i2] * ((Float)1 - a.ssa2[i2]);
  } else if (OP == OP_2S_1S) {
    const Float t1 = a.tau1[i];
    const Float tau12 = t1 + a.tau2[i2];
    a.ssa1[i] = t1 * a.ssa1[i] / fmax(kEps, tau12);
    a.tau1[i] = tau12;
  } else if (OP == OP_2S_2S) {
    const Float t1 = a.tau1[i], s1 = a.ssa1[i], t2 = a.tau2[i2], s2 = a.ssa2[i2];
    const Float tau12 = t1 + t2;
    const Float tauscat12 = t1 * s1 + t2 * s2;
    a.g1[i] = (t1 * s1 * a.g1[i] + t2 * s2 * a.g2[(size_t)a.nmom2 * i2]) / fmax(kEps, tauscat12);
    a.ssa1[i] = tauscat12 / fmax(kEps, tau12);
    a.tau1[i] = tau12;
  } else {
    const Float t1 = a.tau1[i], s1 = a.ssa1[i], t2 = a.tau2[i2], s2 = a.ssa2[i2];
    const Float tau12 = t1 + t2;
    const Float tauscat12 = t1 * s1 + t2 * s2;
    const int mom_lim = OP == OP_NS_G2 ? a.nmom1 : min(a.nmom1, a.nmom2);
    Float tm = 1;
    for (int im = 0; im < mom_lim; ++im) {
      Float m2;
      if (OP == OP_NS_G2) {
        tm = (im == 0) ? a.g2[i2] : tm * a.g2[i2];
        m2 = tm;
      } else {
        m2 = a.g2[(size_t)a.nmom2 * i2 + im];
      }
      Float* p = a.g1 + (size_t)a.nmom1 * i + im;
      *p = (t1 * s1 * *p + t2 * s2 * m2) / fmax(kEps, tauscat12);
    }
    a.ssa1[i] = tauscat12 / fmax(kEps, tau12);
    a.tau1[i] = tau12;
  }
}

// One thread = one (column, layer) and, when operand 2 is given by band, all g-points of that band (its values
// are then the same for the whole run and stay in L1; g-points that no band covers are never visited, as in the
// reference); one g-point otherwise (measured: a run of 16 planes per thread is slower there, 7.1 vs 6.2 ms).
template <int OP>
__global__ void __launch_bounds__(256) increment_kernel(IncArgs a) {
  const int cl = blockIdx.x * blockDim.x + threadIdx.x;
  if (cl >= a.ncl) return;
  int gS, gE;
  if (a.lims) {
    gS = a.lims[2 * blockIdx.y] - 1; gE = a.lims[2 * blockIdx.y + 1] - 1;
  } else {
    gS = gE = blockIdx.y;
  }
  const size_t i2b = (size_t)cl + (size_t)a.ncl * blockIdx.y;
#pragma unroll 4
  for (int g = gS; g <= gE; ++g) {
    const size_t i = (size_t)cl + (size_t)a.ncl * g;
    increment_one<OP>(a, i, a.lims ? i2b : i);
  }
}

template <int OP>
void increment(const char* name, int ncol, int nlay, int ngpt, Float* tau1, Float* ssa1, Float* g1, int nmom1,
               const Float* tau2, const Float* ssa2, const Float* g2, int nmom2, int nbnd, const int* lims) {
  if (ncol <= 0 || nlay <= 0 || ngpt <= 0) return;
  RTE_TRY
  rte::Call c(name);
  const size_t n1 = (size_t)ncol * nlay * ngpt, n2 = (size_t)ncol * nlay * (lims ? nbnd : ngpt);
  IncArgs a;
  a.ncl = ncol * nlay; a.ngpt = ngpt; a.nbnd = nbnd; a.nmom1 = nmom1 > 0 ? nmom1 : 1; a.nmom2 = nmom2 > 0 ? nmom2 : 1;
  a.lims = lims ? c.in(lims, (size_t)2 * nbnd) : nullptr;
  // (lazy: in host-mirror mode the incremented optical properties stay on the device for the solver)
  a.tau1 = c.inout_lazy(tau1, n1);
  a.ssa1 = ssa1 ? c.inout_lazy(ssa1, n1) : nullptr;
  a.g1 = g1 ? c.inout_lazy(g1, n1 * a.nmom1) : nullptr;
  a.tau2 = c.in(tau2, n2);
  a.ssa2 = ssa2 ? c.in(ssa2, n2) : nullptr;
  a.g2 = g2 ? c.in(g2, n2 * a.nmom2) : nullptr;
  rte::ProfScope p("increment_kernel");
  hipLaunchKernelGGL(increment_kernel<OP>, dim3(cdiv(a.ncl, 256), lims ? nbnd : ngpt), dim3(256), 0,
                     rte::stream(), a);
  RTE_CATCH(name)
}

// :44-98
__global__ void __launch_bounds__(256)
delta_scale_kernel(size_t n, Float* __restrict__ tau, Float* __restrict__ ssa, Float* __restrict__ g,
                   const Float* __restrict__ f_in) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const Float gi = g[i], si = ssa[i];
  const Float f = f_in ? f_in[i] : gi * gi;
  const Float wf = si * f;
  tau[i] = ((Float)1 - wf) * tau[i];
  ssa[i] = (si - wf) / fmax(kEps, ((Float)1 - wf));
  g[i] = (gi - f) / fmax(kEps, ((Float)1 - f));
}

void delta_scale(const char* name, int ncol, int nlay, int ngpt, Float* tau, Float* ssa, Float* g, const Float* f) {
  const size_t n = (size_t)ncol * nlay * ngpt;
  if (n == 0) return;
  RTE_TRY
  rte::Call c(name);
  Float *dt = c.inout_lazy(tau, n), *ds = c.inout_lazy(ssa, n), *dg = c.inout_lazy(g, n);
  const Float* df = f ? c.in(f, n) : nullptr;
  rte::ProfScope p("delta_scale_kernel");
  hipLaunchKernelGGL(delta_scale_kernel, dim3(cdiv(n, 256)), dim3(256), 0, rte::stream(), n, dt, ds, dg, df);
  RTE_CATCH(name)
}

// :713-778; nmom = leading dimension (1 for 3-D arrays); ssa_in non-null: absorption optical depth
__global__ void __launch_bounds__(256)
extract_subset_kernel(int nmom, int ncol, int nc, int colS0, size_t nk, const Float* __restrict__ in,
                      const Float* __restrict__ ssa_in, Float* __restrict__ out) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;  // over nmom*nc
  const size_t k = blockIdx.y;                                       // over nlay*ngpt
  if (i >= (size_t)nmom * nc || k >= nk) return;
  const int im = (int)(i % nmom), ic = (int)(i / nmom);
  const size_t src = im + (size_t)nmom * ((size_t)(colS0 + ic) + (size_t)ncol * k);
  const Float v = in[src];
  out[i + (size_t)nmom * nc * k] = ssa_in ? v * ((Float)1 - ssa_in[src]) : v;
}

void extract_subset(const char* name, int nmom, int ncol, int nlay, int ngpt, const Float* in, const Float* ssa_in,
                    int colS, int colE, Float* out) {
  const int nc = colE - colS + 1;
  const size_t nk = (size_t)nlay * ngpt;
  if (nc <= 0 || nk == 0) return;
  RTE_TRY
  rte::Call c(name);
  const Float* d_in = c.in(in, (size_t)nmom * ncol * nk);
  const Float* d_ssa = ssa_in ? c.in(ssa_in, (size_t)nmom * ncol * nk) : nullptr;
  Float* d_out = c.out(out, (size_t)nmom * nc * nk);
  rte::ProfScope p("extract_subset_kernel");
  // grid.y is limited to 65535: fold the (nlay*ngpt) dimension
  const size_t ky = nk < 65535 ? nk : 65535;
  for (size_t k0 = 0; k0 < nk; k0 += ky) {
    const size_t kn = nk - k0 < ky ? nk - k0 : ky;
    hipLaunchKernelGGL(extract_subset_kernel, dim3(cdiv((size_t)nmom * nc, 256), (unsigned)kn), dim3(256), 0,
                       rte::stream(), nmom, ncol, nc, colS - 1, kn, d_in + (size_t)nmom * ncol * k0,
                       d_ssa ? d_ssa + (size_t)nmom * ncol * k0 : nullptr, d_out + (size_t)nmom * nc * k0);
  }
  RTE_CATCH(name)
}

// rrtmgp/kernels/mo_cloud_optics_rrtmgp_kernels.F90:40-64
__global__ void __launch_bounds__(256)
cld_from_table_kernel(int ncl, int nsteps, Float step_size, Float offset, const Bool* __restrict__ mask,
                      const Float* __restrict__ lwp, const Float* __restrict__ re, const Float* __restrict__ tau_table,
                      const Float* __restrict__ ssa_table, const Float* __restrict__ asy_table, Float* __restrict__ tau,
                      Float* __restrict__ taussa, Float* __restrict__ taussag) {
  const int cl = blockIdx.x * blockDim.x + threadIdx.x;
  const int igpt = blockIdx.y;
  if (cl >= ncl) return;
  const size_t i = (size_t)cl + (size_t)ncl * igpt;
  if (mask[cl]) {
    const Float x = (re[cl] - offset) / step_size;
    const int index = min((int)floor(x) + 1, nsteps - 1);  // 1-based
    const Float fint = x - (Float)(index - 1);
    const size_t o = (size_t)nsteps * igpt + (index - 1);
    const Float t = lwp[cl] * (tau_table[o] + fint * (tau_table[o + 1] - tau_table[o]));
    const Float ts = t * (ssa_table[o] + fint * (ssa_table[o + 1] - ssa_table[o]));
    taussag[i] = ts * (asy_table[o] + fint * (asy_table[o + 1] - asy_table[o]));
    taussa[i] = ts;
    tau[i] = t;
  } else {
    tau[i] = 0;
    taussa[i] = 0;
    taussag[i] = 0;
  }
}

// ---- fused cloud optics (library extension): cloud_optics of ty_cloud_optics_rrtmgp, look-up-table branch
// (rrtmgp/frontend/mo_cloud_optics_rrtmgp.F90:334-425) = masks, compute_cld_from_table for liquid and for ice, their
// combination, and optionally the all-sky driver's delta scaling (rte_delta_scale_2str_k), in ONE pass: the six
// (ncol, nlay, nbnd) intermediates of the separate kernels never go to memory.  Every value is formed by the same
// operations in the same order as in those kernels (cld_from_table_kernel, cloud_combine_kernel, delta_scale_kernel).
struct CldFused {
  int ncl, nbnd, liq_nsteps, ice_nsteps;
  Float liq_step, liq_off, ice_step, ice_off;
  const Float *clwp, *ciwp, *reliq, *deice;
  const Float *lext, *lssa, *lasy, *iext, *issa, *iasy;  // (nsteps, nbnd)
  Float *tau, *ssa, *g;
};
// index and interpolation weight of one phase's table for this cell (:334-341 of the frontend's masks, compute_cld_from_table)
struct CldIdx { bool on; Float wp, fint; int o; };
__device__ __forceinline__ CldIdx cld_index(Float wp, Float re, Float offset, Float step_size, int nsteps) {
  CldIdx r;
  r.on = wp > (Float)0; r.wp = wp;
  const Float x = (re - offset) / step_size;
  const int index = r.on ? min((int)floor(x) + 1, nsteps - 1) : 1;  // 1-based (a cell without this phase reads row 1: never used)
  r.fint = x - (Float)(index - 1);
  r.o = index - 1;
  return r;
}
__device__ __forceinline__ void cld_lookup(const CldIdx& q, int nsteps, int ibnd, const Float* __restrict__ te, const Float* __restrict__ ts_,
                                           const Float* __restrict__ ta, Float& tau, Float& taussa, Float& taussag) {
  tau = 0; taussa = 0; taussag = 0;
  if (q.on) {
    const size_t o = (size_t)nsteps * ibnd + q.o;
    const Float t = q.wp * (te[o] + q.fint * (te[o + 1] - te[o]));
    const Float ts = t * (ts_[o] + q.fint * (ts_[o + 1] - ts_[o]));
    taussag = ts * (ta[o] + q.fint * (ta[o + 1] - ta[o]));
    taussa = ts;
    tau = t;
  }
}
// thread = one (column, layer) cell, the bands walked inside it: the four cloud fields are read once (as a grid dimension the
// bands' blocks ran far apart and re-read them per band: 784 B moved per cell for 368 B of inputs and outputs), the table row
// and weight of a phase are formed once, and a wave without cloud in any of its cells only stores
template <bool TWOSTR, bool DELTA>
__global__ void __launch_bounds__(256) cloud_optics_fused_kernel(CldFused a) {
  const int cl = blockIdx.x * blockDim.x + threadIdx.x;
  if (cl >= a.ncl) return;
  const CldIdx ql = cld_index(a.clwp[cl], a.reliq[cl], a.liq_off, a.liq_step, a.liq_nsteps);
  const CldIdx qi = cld_index(a.ciwp[cl], a.deice[cl], a.ice_off, a.ice_step, a.ice_nsteps);
#ifdef RTE_USE_SP
  const Float eps = 1.1920929e-07f;  // epsilon(tau)
#else
  const Float eps = 2.220446049250313e-16;
#endif
  for (int ibnd = 0; ibnd < a.nbnd; ++ibnd) {
    Float lt, lts, ltsg, it, its, itsg;
    cld_lookup(ql, a.liq_nsteps, ibnd, a.lext, a.lssa, a.lasy, lt, lts, ltsg);
    cld_lookup(qi, a.ice_nsteps, ibnd, a.iext, a.issa, a.iasy, it, its, itsg);
    const size_t i = (size_t)cl + (size_t)a.ncl * ibnd;
    if (!TWOSTR) {
      rte::store_stream(a.tau + i, (lt - lts) + (it - its));
      continue;
    }
    const Float t = lt + it;
    const Float ts = lts + its;
    Float g = (ltsg + itsg) / fmax(eps, ts);
    Float ssa = ts / fmax(eps, t);
    Float tau = t;
    if (DELTA) {  // delta_scale_2str_k, rte/kernels/mo_optical_props_kernels.F90:76-98
      const Float f = g * g;
      const Float wf = ssa * f;
      tau = ((Float)1 - wf) * tau;
      const Float ssa_new = (ssa - wf) / fmax(kEps, ((Float)1 - wf));
      g = (g - f) / fmax(kEps, ((Float)1 - f));
      ssa = ssa_new;
    }
    rte::store_stream(a.tau + i, tau); rte::store_stream(a.ssa + i, ssa); rte::store_stream(a.g + i, g);
  }
}
}  // namespace

extern "C" {
void rte_delta_scale_2str_f_k(const int* ncol, const int* nlay, const int* ngpt, Float* tau, Float* ssa, Float* g,
                              const Float* f) {
  delta_scale("rte_delta_scale_2str_f_k", *ncol, *nlay, *ngpt, tau, ssa, g, f);
}
void rte_delta_scale_2str_k(const int* ncol, const int* nlay, const int* ngpt, Float* tau, Float* ssa, Float* g) {
  delta_scale("rte_delta_scale_2str_k", *ncol, *nlay, *ngpt, tau, ssa, g, nullptr);
}
#define N ncol, nlay, ngpt
#define D *ncol, *nlay, *ngpt
void rte_increment_1scalar_by_1scalar(const int* ncol, const int* nlay, const int* ngpt, Float* tau1, const Float* tau2) {
  increment<OP_1S_1S>("rte_increment_1scalar_by_1scalar", D, tau1, nullptr, nullptr, 1, tau2, nullptr, nullptr, 1, 0, nullptr);
}
void rte_increment_1scalar_by_2stream(const int* ncol, const int* nlay, const int* ngpt, Float* tau1, const Float* tau2,
                                      const Float* ssa2) {
  increment<OP_1S_2S>("rte_increment_1scalar_by_2stream", D, tau1, nullptr, nullptr, 1, tau2, ssa2, nullptr, 1, 0, nullptr);
}
void rte_increment_1scalar_by_nstream(const int* ncol, const int* nlay, const int* ngpt, Float* tau1, const Float* tau2,
                                      const Float* ssa2) {
  increment<OP_1S_2S>("rte_increment_1scalar_by_nstream", D, tau1, nullptr, nullptr, 1, tau2, ssa2, nullptr, 1, 0, nullptr);
}
void rte_increment_2stream_by_1scalar(const int* ncol, const int* nlay, const int* ngpt, Float* tau1, Float* ssa1,
                                      const Float* tau2) {
  increment<OP_2S_1S>("rte_increment_2stream_by_1scalar", D, tau1, ssa1, nullptr, 1, tau2, nullptr, nullptr, 1, 0, nullptr);
}
void rte_increment_2stream_by_2stream(const int* ncol, const int* nlay, const int* ngpt, Float* tau1, Float* ssa1,
                                      Float* g1, const Float* tau2, const Float* ssa2, const Float* g2) {
  increment<OP_2S_2S>("rte_increment_2stream_by_2stream", D, tau1, ssa1, g1, 1, tau2, ssa2, g2, 1, 0, nullptr);
}
void rte_increment_2stream_by_nstream(const int* ncol, const int* nlay, const int* ngpt, const int* nmom2, Float* tau1,
                                      Float* ssa1, Float* g1, const Float* tau2, const Float* ssa2, const Float* p2) {
  increment<OP_2S_2S>("rte_increment_2stream_by_nstream", D, tau1, ssa1, g1, 1, tau2, ssa2, p2, *nmom2, 0, nullptr);
}
void rte_increment_nstream_by_1scalar(const int* ncol, const int* nlay, const int* ngpt, Float* tau1, Float* ssa1,
                                      const Float* tau2) {
  increment<OP_2S_1S>("rte_increment_nstream_by_1scalar", D, tau1, ssa1, nullptr, 1, tau2, nullptr, nullptr, 1, 0, nullptr);
}
void rte_increment_nstream_by_2stream(const int* ncol, const int* nlay, const int* ngpt, const int* nmom1, Float* tau1,
                                      Float* ssa1, Float* p1, const Float* tau2, const Float* ssa2, const Float* g2) {
  increment<OP_NS_G2>("rte_increment_nstream_by_2stream", D, tau1, ssa1, p1, *nmom1, tau2, ssa2, g2, 1, 0, nullptr);
}
void rte_increment_nstream_by_nstream(const int* ncol, const int* nlay, const int* ngpt, const int* nmom1,
                                      const int* nmom2, Float* tau1, Float* ssa1, Float* p1, const Float* tau2,
                                      const Float* ssa2, const Float* p2) {
  increment<OP_NS_P2>("rte_increment_nstream_by_nstream", D, tau1, ssa1, p1, *nmom1, tau2, ssa2, p2, *nmom2, 0, nullptr);
}
void rte_inc_1scalar_by_1scalar_bybnd(const int* ncol, const int* nlay, const int* ngpt, Float* tau1, const Float* tau2,
                                      const int* nbnd, const int* gpt_lims) {
  increment<OP_1S_1S>("rte_inc_1scalar_by_1scalar_bybnd", D, tau1, nullptr, nullptr, 1, tau2, nullptr, nullptr, 1, *nbnd, gpt_lims);
}
void rte_inc_1scalar_by_2stream_bybnd(const int* ncol, const int* nlay, const int* ngpt, Float* tau1, const Float* tau2,
                                      const Float* ssa2, const int* nbnd, const int* gpt_lims) {
  increment<OP_1S_2S>("rte_inc_1scalar_by_2stream_bybnd", D, tau1, nullptr, nullptr, 1, tau2, ssa2, nullptr, 1, *nbnd, gpt_lims);
}
void rte_inc_1scalar_by_nstream_bybnd(const int* ncol, const int* nlay, const int* ngpt, Float* tau1, const Float* tau2,
                                      const Float* ssa2, const int* nbnd, const int* gpt_lims) {
  increment<OP_1S_2S>("rte_inc_1scalar_by_nstream_bybnd", D, tau1, nullptr, nullptr, 1, tau2, ssa2, nullptr, 1, *nbnd, gpt_lims);
}
void rte_inc_2stream_by_1scalar_bybnd(const int* ncol, const int* nlay, const int* ngpt, Float* tau1, Float* ssa1,
                                      const Float* tau2, const int* nbnd, const int* gpt_lims) {
  increment<OP_2S_1S>("rte_inc_2stream_by_1scalar_bybnd", D, tau1, ssa1, nullptr, 1, tau2, nullptr, nullptr, 1, *nbnd, gpt_lims);
}
void rte_inc_2stream_by_2stream_bybnd(const int* ncol, const int* nlay, const int* ngpt, Float* tau1, Float* ssa1,
                                      Float* g1, const Float* tau2, const Float* ssa2, const Float* g2, const int* nbnd,
                                      const int* gpt_lims) {
  increment<OP_2S_2S>("rte_inc_2stream_by_2stream_bybnd", D, tau1, ssa1, g1, 1, tau2, ssa2, g2, 1, *nbnd, gpt_lims);
}
void rte_inc_2stream_by_nstream_bybnd(const int* ncol, const int* nlay, const int* ngpt, const int* nmom2, Float* tau1,
                                      Float* ssa1, Float* g1, const Float* tau2, const Float* ssa2, const Float* p2,
                                      const int* nbnd, const int* gpt_lims) {
  increment<OP_2S_2S>("rte_inc_2stream_by_nstream_bybnd", D, tau1, ssa1, g1, 1, tau2, ssa2, p2, *nmom2, *nbnd, gpt_lims);
}
void rte_inc_nstream_by_1scalar_bybnd(const int* ncol, const int* nlay, const int* ngpt, Float* tau1, Float* ssa1,
                                      const Float* tau2, const int* nbnd, const int* gpt_lims) {
  increment<OP_2S_1S>("rte_inc_nstream_by_1scalar_bybnd", D, tau1, ssa1, nullptr, 1, tau2, nullptr, nullptr, 1, *nbnd, gpt_lims);
}
void rte_inc_nstream_by_2stream_bybnd(const int* ncol, const int* nlay, const int* ngpt, const int* nmom1, Float* tau1,
                                      Float* ssa1, Float* p1, const Float* tau2, const Float* ssa2, const Float* g2,
                                      const int* nbnd, const int* gpt_lims) {
  increment<OP_NS_G2>("rte_inc_nstream_by_2stream_bybnd", D, tau1, ssa1, p1, *nmom1, tau2, ssa2, g2, 1, *nbnd, gpt_lims);
}
void rte_inc_nstream_by_nstream_bybnd(const int* ncol, const int* nlay, const int* ngpt, const int* nmom1,
                                      const int* nmom2, Float* tau1, Float* ssa1, Float* p1, const Float* tau2,
                                      const Float* ssa2, const Float* p2, const int* nbnd, const int* gpt_lims) {
  increment<OP_NS_P2>("rte_inc_nstream_by_nstream_bybnd", D, tau1, ssa1, p1, *nmom1, tau2, ssa2, p2, *nmom2, *nbnd, gpt_lims);
}
#undef N
#undef D
void rte_extract_subset_dim1_3d(const int* ncol, const int* nlay, const int* ngpt, const Float* array_in,
                                const int* colS, const int* colE, Float* array_out) {
  extract_subset("rte_extract_subset_dim1_3d", 1, *ncol, *nlay, *ngpt, array_in, nullptr, *colS, *colE, array_out);
}
void rte_extract_subset_dim2_4d(const int* nmom, const int* ncol, const int* nlay, const int* ngpt,
                                const Float* array_in, const int* colS, const int* colE, Float* array_out) {
  extract_subset("rte_extract_subset_dim2_4d", *nmom, *ncol, *nlay, *ngpt, array_in, nullptr, *colS, *colE, array_out);
}
void rte_extract_subset_absorption_tau(const int* ncol, const int* nlay, const int* ngpt, const Float* tau_in,
                                       const Float* ssa_in, const int* colS, const int* colE, Float* tau_out) {
  extract_subset("rte_extract_subset_absorption_tau", 1, *ncol, *nlay, *ngpt, tau_in, ssa_in, *colS, *colE, tau_out);
}

void rrtmgp_compute_cld_from_table(const int* ncol_, const int* nlay_, const int* ngpt_, const Bool* mask,
                                   const Float* lwp, const Float* re, const int* nsteps_, const Float* step_size,
                                   const Float* offset, const Float* tau_table, const Float* ssa_table,
                                   const Float* asy_table, Float* tau, Float* taussa, Float* taussag) {
  const int ncol = *ncol_, nlay = *nlay_, ngpt = *ngpt_, nsteps = *nsteps_;
  if (ncol <= 0 || nlay <= 0 || ngpt <= 0) return;
  RTE_TRY
  rte::Call c("rrtmgp_compute_cld_from_table");
  const size_t ncl = (size_t)ncol * nlay, n = ncl * ngpt, nt = (size_t)nsteps * ngpt;
  const Bool* d_mask = c.in(mask, ncl);
  const Float *d_lwp = c.in(lwp, ncl), *d_re = c.in(re, ncl);
  const Float *d_tt = c.in(tau_table, nt), *d_st = c.in(ssa_table, nt), *d_at = c.in(asy_table, nt);
  Float *d_tau = c.out(tau, n), *d_ts = c.out(taussa, n), *d_tsg = c.out(taussag, n);
  rte::ProfScope p("cld_from_table_kernel");
  hipLaunchKernelGGL(cld_from_table_kernel, dim3(cdiv(ncl, 256), ngpt), dim3(256), 0, rte::stream(), (int)ncl, nsteps,
                     *step_size, *offset, d_mask, d_lwp, d_re, d_tt, d_st, d_at, d_tau, d_ts, d_tsg);
  RTE_CATCH("rrtmgp_compute_cld_from_table")
}
// cloud_optics (look-up tables) in one pass; twostr = 0: absorption optical depth only (tau); delta_scale != 0: the
// two-stream result is delta-scaled with f = g^2.  Tables are (nsteps, nbnd); outputs (ncol, nlay, nbnd).
int rte_hip_cloud_optics_fused(int ncol, int nlay, int nbnd, int twostr, int delta_scale, const Float* clwp, const Float* ciwp,
                               const Float* reliq, const Float* deice, int liq_nsteps, double liq_step_size, double radliq_lwr,
                               const Float* extliq, const Float* ssaliq, const Float* asyliq, int ice_nsteps,
                               double ice_step_size, double diamice_lwr, const Float* extice, const Float* ssaice,
                               const Float* asyice, Float* tau, Float* ssa, Float* g) {
  const size_t ncl = (size_t)ncol * nlay;
  if (ncl == 0 || nbnd <= 0) return 0;
  RTE_TRY
  rte::Call c("rte_hip_cloud_optics_fused");
  CldFused a;
  a.ncl = (int)ncl; a.nbnd = nbnd; a.liq_nsteps = liq_nsteps; a.ice_nsteps = ice_nsteps;
  a.liq_step = (Float)liq_step_size; a.liq_off = (Float)radliq_lwr; a.ice_step = (Float)ice_step_size; a.ice_off = (Float)diamice_lwr;
  a.clwp = c.in(clwp, ncl); a.ciwp = c.in(ciwp, ncl); a.reliq = c.in(reliq, ncl); a.deice = c.in(deice, ncl);
  a.lext = c.in(extliq, (size_t)liq_nsteps * nbnd); a.lssa = c.in(ssaliq, (size_t)liq_nsteps * nbnd); a.lasy = c.in(asyliq, (size_t)liq_nsteps * nbnd);
  a.iext = c.in(extice, (size_t)ice_nsteps * nbnd); a.issa = c.in(ssaice, (size_t)ice_nsteps * nbnd); a.iasy = c.in(asyice, (size_t)ice_nsteps * nbnd);
  a.tau = c.out(tau, ncl * nbnd);
  a.ssa = twostr ? c.out(ssa, ncl * nbnd) : nullptr;
  a.g = twostr ? c.out(g, ncl * nbnd) : nullptr;
  rte::ProfScope p("cloud_optics_fused_kernel");
  const dim3 grid(cdiv(ncl, 256)), blk(256);
  if (!twostr) hipLaunchKernelGGL((cloud_optics_fused_kernel<false, false>), grid, blk, 0, rte::stream(), a);
  else if (delta_scale) hipLaunchKernelGGL((cloud_optics_fused_kernel<true, true>), grid, blk, 0, rte::stream(), a);
  else hipLaunchKernelGGL((cloud_optics_fused_kernel<true, false>), grid, blk, 0, rte::stream(), a);
  return 0;
  RTE_CATCH("rte_hip_cloud_optics_fused")
  return -1;
}
}
