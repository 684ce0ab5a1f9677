// solvers.hip -- RTE solver kernels for gfx950 (MI355X), hand-written HIP.
//
// Entry points (C ABI = reference rte/kernels/api/mo_rte_solver_kernels.F90):
//   rte_lw_solver_noscat, rte_lw_solver_2stream, rte_sw_solver_noscat, rte_sw_solver_2stream
// Arithmetic follows the reference `default` CPU kernels (rte/kernels/mo_rte_solver_kernels.F90)
// expression by expression.  Mapping: one thread per (column, g-point); wave lanes = 64
// consecutive columns, blockIdx.y = g-point; the vertical recurrences run sequentially inside
// the thread.
//
// Two families of kernels live here:
//   * lw_noscat_seg_kernel -- the production path for the no-scattering LW solver without
//     rescaling: the column is cut into vertical segments owned by different waves of a block;
//     transmissivities and sources live in registers, segment composites are exchanged through
//     LDS, and broadband sums are accumulated in registers over the block's g-points
//     ("K4").
//   * the *_generic kernels -- every other solver/flag combination; per-layer intermediates that
//     a second sweep needs are parked in a device scratch slab, g-points processed in chunks.
#include <math.h>

#include <type_traits>

#include <atomic>

#include "common.h"
#include "fastmath.h"

namespace {

using rte::cdiv;

#ifdef RTE_USE_SP
#define RTE_EPS 1.1920929e-07f
#else
#define RTE_EPS 2.220446049250313e-16
#endif
__device__ constexpr Float kPi = (Float)3.14159265358979323846264338327950288;

// ---------------------------------------------------------------------------------------------
// shared pieces
// ---------------------------------------------------------------------------------------------
// lw_source_noscat for one layer: reference :652-663.  lev_dec / lev_inc are the Planck sources
// at level ilay and ilay+1 (array order); returns the sources emitted toward increasing / decreasing index
__device__ __forceinline__ void lw_source_layer(Float tau_loc, Float trans, Float lay, Float lev_lo, Float lev_hi,
                                                Float& src_inc, Float& src_dec) {
#pragma clang fp contract(fast)  // well-conditioned sums: a*b+c may fuse (a few ulp from the reference)
  const Float tau_thresh = sqrt(sqrt((Float)RTE_EPS));
  Float fact;
  if (tau_loc > tau_thresh)
    fact = ((Float)1 - trans) / tau_loc - trans;
  else
    fact = tau_loc * ((Float)0.5 + tau_loc * (-(Float)1 / (Float)3 + tau_loc * (Float)1 / (Float)8));
  src_inc = ((Float)1 - trans) * lev_hi + (Float)2 * fact * (lay - lev_hi);
  src_dec = ((Float)1 - trans) * lev_lo + (Float)2 * fact * (lay - lev_lo);
}

// the same for the segmented kernels: both branches evaluated and selected (no divergent branch), the quotient by
// reciprocal + Newton (fastmath.h); tau_loc > tau_thresh ~ 1.2e-4 where the quotient is taken
__device__ __forceinline__ void lw_source_layer_fast(Float tau_loc, Float trans, Float lay, Float lev_lo, Float lev_hi,
                                                     Float& src_inc, Float& src_dec) {
#pragma clang fp contract(fast)
  const Float tau_thresh = sqrt(sqrt((Float)RTE_EPS));
  const Float omt = (Float)1 - trans;
  const Float big = rte::div_nr(omt, tau_loc > tau_thresh ? tau_loc : (Float)1) - trans;
  const Float small = tau_loc * ((Float)0.5 + tau_loc * (-(Float)1 / (Float)3 + tau_loc * (Float)1 / (Float)8));
  const Float fact2 = (Float)2 * (tau_loc > tau_thresh ? big : small);
  src_inc = omt * lev_hi + fact2 * (lay - lev_hi);
  src_dec = omt * lev_lo + fact2 * (lay - lev_lo);
}

// broadband(c,l) = [prev +] scale * sum_g spectral(c,l,g)   (sequential over g like the reference)
__global__ void __launch_bounds__(256)
sum_gpt_kernel(size_t n2, int ngpt, const Float* __restrict__ spectral, Float* __restrict__ out, Float scale,
               int mode /*0: out = s ; 1: out += s ; 2: out = scale*(out + s) ; 3: out = scale*s*/) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n2) return;
  Float s = (mode == 1 || mode == 2) ? out[i] : (Float)0;
  for (int g = 0; g < ngpt; ++g) s = s + spectral[i + n2 * (size_t)g];
  out[i] = (mode >= 2) ? scale * s : s;
}

__global__ void __launch_bounds__(256)
axpy_kernel(size_t n, const Float* __restrict__ x, Float* __restrict__ y, int first) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) y[i] = first ? x[i] : y[i] + x[i];
}

// ---------------------------------------------------------------------------------------------
// LW no-scattering, generic: reference lw_solver_noscat_oneangle :51-240
// radn_up / radn_dn: (ncol, nlay+1, gchunk) intensities for this chunk of g-points (either the
// caller's spectral flux arrays or scratch).  jac: (ncol, nlay+1, gchunk) scratch or null.
// ---------------------------------------------------------------------------------------------
struct LwArgs {
  int ncol, nlay, ngpt, g_begin, gchunk;
  bool top_at_1, do_jac, do_rescaling, scale_out, accumulate;
  Float weight;
  const Float *D, *tau, *lay_source, *lev_source, *sfc_emis, *sfc_src, *inc_flux, *sfc_srcJac, *ssa, *g;
  Float *radn_up, *radn_dn, *jac;
};

__global__ void __launch_bounds__(256) lw_noscat_generic_kernel(LwArgs a) {
  const int icol = blockIdx.x * blockDim.x + threadIdx.x;
  const int gl = blockIdx.y;  // g-point within the chunk
  if (icol >= a.ncol) return;
  const int igpt = a.g_begin + gl;
  const int ncol = a.ncol, nlay = a.nlay, nlev = nlay + 1;
  const size_t ncl = (size_t)ncol * nlay, nclv = (size_t)ncol * nlev;
  const size_t cg = icol + (size_t)ncol * igpt;
  const Float* tau = a.tau + icol + ncl * igpt;
  const Float* lay_source = a.lay_source + icol + ncl * igpt;
  const Float* lev_source = a.lev_source + icol + nclv * igpt;
  const Float* ssa = a.do_rescaling ? a.ssa + icol + ncl * igpt : nullptr;
  const Float* gg = a.do_rescaling ? a.g + icol + ncl * igpt : nullptr;
  Float* up = a.radn_up + icol + nclv * gl;
  Float* dn = a.radn_dn + icol + nclv * gl;
  Float* jac = a.do_jac ? a.jac + icol + nclv * gl : nullptr;
  const Float D = a.D[cg];
  const Float piw = kPi * a.weight;
  const int top_level = a.top_at_1 ? 0 : nlay, sfc_level = a.top_at_1 ? nlay : 0;

  // per-layer quantities recomputed by every sweep (no cross-sweep storage besides the radiances)
  auto layer = [&](int ilay, Float& trans, Float& src_dn, Float& src_up, Float& An, Float& Cn) {
    const Float t = tau[(size_t)ncol * ilay];
    Float tau_loc;
    if (a.do_rescaling) {  // :154-178
      const Float ssal = ssa[(size_t)ncol * ilay];
      const Float wb = ssal * ((Float)1 - gg[(size_t)ncol * ilay]) * (Float)0.5;
      const Float scaleTau = ((Float)1 - ssal + wb);
      Cn = (Float)0.4 * wb / scaleTau;
      tau_loc = t * D * scaleTau;
      trans = exp(-tau_loc);
      An = ((Float)1 - trans * trans);
    } else {  // :180-183
      tau_loc = t * D;
      trans = exp(-tau_loc);
      An = Cn = 0;
    }
    Float s_inc, s_dec;
    lw_source_layer(tau_loc, trans, lay_source[(size_t)ncol * ilay], lev_source[(size_t)ncol * ilay],
                    lev_source[(size_t)ncol * (ilay + 1)], s_inc, s_dec);
    src_dn = a.top_at_1 ? s_inc : s_dec;
    src_up = a.top_at_1 ? s_dec : s_inc;
  };

  Float trans, src_dn, src_up, An, Cn;
  // ---- transport down :681-708
  Float r = a.inc_flux[cg] / piw;  // :144
  dn[(size_t)ncol * top_level] = r;
  if (a.top_at_1) {
    for (int ilev = 1; ilev < nlev; ++ilev) {
      layer(ilev - 1, trans, src_dn, src_up, An, Cn);
      r = trans * r + src_dn;
      dn[(size_t)ncol * ilev] = r;
    }
  } else {
    for (int ilev = nlay - 1; ilev >= 0; --ilev) {
      layer(ilev, trans, src_dn, src_up, An, Cn);
      r = trans * r + src_dn;
      dn[(size_t)ncol * ilev] = r;
    }
  }
  // ---- surface :198-202
  const Float emis = a.sfc_emis[cg];
  const Float sfc_albedo = (Float)1 - emis;
  Float u = r * sfc_albedo + emis * a.sfc_src[cg];
  up[(size_t)ncol * sfc_level] = u;
  Float j = 0;
  if (a.do_jac) {
    j = emis * a.sfc_srcJac[cg];
    jac[(size_t)ncol * sfc_level] = j;
  }
  // ---- transport up (:710-745) or up + second down with rescaling (:753-844)
  if (a.top_at_1) {
    for (int ilev = nlay - 1; ilev >= 0; --ilev) {
      layer(ilev, trans, src_dn, src_up, An, Cn);
      if (a.do_rescaling) {
        const Float adj = Cn * (An * dn[(size_t)ncol * ilev] - trans * src_dn - src_up);
        u = trans * u + src_up + adj;
      } else {
        u = trans * u + src_up;
      }
      up[(size_t)ncol * ilev] = u;
      if (a.do_jac) { j = trans * j; jac[(size_t)ncol * ilev] = j; }
    }
    if (a.do_rescaling) {
      r = dn[0];
      for (int ilev = 0; ilev < nlay; ++ilev) {
        layer(ilev, trans, src_dn, src_up, An, Cn);
        const Float adj = Cn * (An * up[(size_t)ncol * ilev] - trans * src_up - src_dn);
        r = trans * r + src_dn + adj;
        dn[(size_t)ncol * (ilev + 1)] = r;
      }
    }
  } else {
    for (int ilev = 0; ilev < nlay; ++ilev) {
      layer(ilev, trans, src_dn, src_up, An, Cn);
      if (a.do_rescaling) {
        const Float adj = Cn * (An * dn[(size_t)ncol * (ilev + 1)] - trans * src_dn - src_up);
        u = trans * u + src_up + adj;
      } else {
        u = trans * u + src_up;
      }
      up[(size_t)ncol * (ilev + 1)] = u;
      if (a.do_jac) { j = trans * j; jac[(size_t)ncol * (ilev + 1)] = j; }
    }
    if (a.do_rescaling) {
      r = dn[(size_t)ncol * nlay];
      for (int ilev = nlay - 1; ilev >= 0; --ilev) {
        layer(ilev, trans, src_dn, src_up, An, Cn);
        const Float adj = Cn * (An * up[(size_t)ncol * ilev] - trans * src_up - src_dn);
        r = trans * r + src_dn + adj;
        dn[(size_t)ncol * ilev] = r;
      }
    }
  }
  // ---- spectral output: intensity -> flux (:223-224)
  if (a.scale_out)
    for (int ilev = 0; ilev < nlev; ++ilev) {
      dn[(size_t)ncol * ilev] = piw * dn[(size_t)ncol * ilev];
      up[(size_t)ncol * ilev] = piw * up[(size_t)ncol * ilev];
    }
}

// ---------------------------------------------------------------------------------------------
// LW no-scattering, segmented production kernel (broadband, no rescaling).
//
// block = S waves x 64 columns; wave s owns layers [s*L, (s+1)*L) counted FROM THE TOP of the
// atmosphere ("position" p; array layer index = p if top_at_1 else nlay-1-p).  Per g-point:
//   pass 1  each thread loads tau / lay / lev of its L layers, computes trans, src_dn, src_up
//           (kept in registers) and the segment composites  Td = prod t,  Sd, Su  (the radiance
//           leaving the segment for zero radiance entering it);
//   exchange composites through LDS, chain them: radiance entering every segment from above
//           (down) and from below (up, after the surface reflection);
//   pass 2  re-sweep the segment from registers with the correct entering radiances and add the
//           level radiances to register accumulators (levels p = s*L .. s*L+L-1, the last
//           segment also owns the surface level).
// Within a segment the recurrence is the reference's; across segments the entering radiance is
// formed from composites (same mathematics, different rounding: ~1e-16 relative).
// After the block's g-points: acc * pi * weight -> partial broadband slab for this g-group.
// ---------------------------------------------------------------------------------------------

template <int L>
struct SegTile {  // one g-point's inputs for one thread's segment
  Float tau[L], lay[L], lev[L + 1], D, emis, ssrc, inc, sjac;
};

// Loop-invariant 32-bit BYTE offsets of a thread's rows inside one g-point plane: every load of the g-point loop is
// then (scalar plane base, advanced per g-point) + (VGPR offset) -- the saddr form of global_load, no per-load
// address arithmetic.  Needs 8 * ncol * (nlay + 1) < 2^32 (checked by the host, else the generic kernel runs).
template <int L>
struct SegOffsets {
  unsigned lay[L], lev[L + 1], cg;
};

template <int L, bool FACT = false>
__device__ __forceinline__ void seg_offsets(SegOffsets<L>& o, int c, int ncol, int nlay, int p0, int np, bool top_at_1) {
  if constexpr (FACT) {
    // factored sources (see lw_noscat_seg_kernel): lev[0], lev[1] are the LAYER rows just above and just below the segment
    // (clamped to the column), whose Planck fractions enter the geometric means at the segment's outer levels
    const int pa = max(p0 - 1, 0), pb = min(p0 + L, nlay - 1);
    o.lev[0] = ((unsigned)c + (unsigned)ncol * (unsigned)(top_at_1 ? pa : nlay - 1 - pa)) * (unsigned)sizeof(Float);
    o.lev[1] = ((unsigned)c + (unsigned)ncol * (unsigned)(top_at_1 ? pb : nlay - 1 - pb)) * (unsigned)sizeof(Float);
    asm volatile("" : "+v"(o.lev[0]), "+v"(o.lev[1]));
  }
#pragma unroll
  for (int i = 0; i < L; ++i) {
    // out-of-segment slots (only the last segment can have them) re-read a valid layer and are then
    // made NEUTRAL (see seg_load)
    const int p = p0 + min(i, np - 1);
    const int ilay = top_at_1 ? p : nlay - 1 - p;
    o.lay[i] = ((unsigned)c + (unsigned)ncol * (unsigned)ilay) * (unsigned)sizeof(Float);
    asm volatile("" : "+v"(o.lay[i]));  // keep it a 32-bit VGPR value (not re-derived per load in 64 bits)
  }
  if constexpr (!FACT) {
#pragma unroll
    for (int i = 0; i <= L; ++i) {
      const int p = p0 + min(i, np);
      o.lev[i] = ((unsigned)c + (unsigned)ncol * (unsigned)(top_at_1 ? p : nlay - p)) * (unsigned)sizeof(Float);
      asm volatile("" : "+v"(o.lev[i]));
    }
  }
  o.cg = (unsigned)c * (unsigned)sizeof(Float);
  asm volatile("" : "+v"(o.cg));
}

template <int L, bool do_jac, bool SFC = true, bool FACT = false>
__device__ __forceinline__ void seg_load(SegTile<L>& t, SegOffsets<L>& o, int igpt, int ncol, int nlay, int np,
                                         const Float* __restrict__ Dsec,
                                         const Float* __restrict__ tau_, const Float* __restrict__ lay_source_,
                                         const Float* __restrict__ lev_source_, const Float* __restrict__ sfc_emis,
                                         const Float* __restrict__ sfc_src, const Float* __restrict__ inc_flux,
                                         const Float* __restrict__ sfc_srcJac, const int i0 = 0, const int i1 = L) {
  // [i0, i1): the layer slots whose rows this call requests (the whole tile by default; lw_noscat_seg_kernel spreads a g-point's
  // requests over pass 1 of the previous one, two slots at a time); the level row below the last slot (FACT: the layer row below the
  // segment) goes with the last group, the per-(column, g-point) arrays (FACT: and the layer row above the segment) with the first
  const size_t ncl = (size_t)ncol * nlay, nclv = (size_t)ncol * (nlay + 1), ncg = (size_t)ncol * igpt;
  // the row offsets are made opaque IN PLACE, once per g-point: their 64-bit extension cannot be hoisted out of the g-point loop
  // (as a by-value copy per load every offset was first moved into a scratch register: 25 v_mov per g-point and wave)
#pragma unroll
  for (int i = 0; i < L; ++i)
    if (i >= i0 && i < i1) asm volatile("" : "+v"(o.lay[i]));
#pragma unroll
  for (int i = 0; i <= (FACT ? 1 : L); ++i)
    if (FACT ? (i == 0 ? i0 == 0 : i1 == L) : (i >= i0 && (i < i1 || i1 == L))) asm volatile("" : "+v"(o.lev[i]));
  if (SFC && i0 == 0) asm volatile("" : "+v"(o.cg));
  auto at = [](const Float* plane, unsigned off) {  // plane is wave-uniform
    // every byte of tau / lay_source / lev_source is read exactly once: non-temporal loads (6.92 -> 6.87 ms in one
    // process, within the noise of that comparison but never slower)
#ifdef RTE_NO_NT_LOADS
    return *reinterpret_cast<const Float*>(reinterpret_cast<const char*>(plane) + off);
#else
    return __builtin_nontemporal_load(reinterpret_cast<const Float*>(reinterpret_cast<const char*>(plane) + off));
#endif
  };
  const int igp = igpt;
  const Float* tau = tau_ + ncl * igp;
  const Float* lay = lay_source_ + ncl * igp;
  const Float* lev = lev_source_ + nclv * igp;
#pragma unroll
  for (int i = 0; i < L; ++i) {
    if (i < i0 || i >= i1) continue;
    // tau = 0 alone makes an out-of-segment slot NEUTRAL whatever finite sources it carries: trans = 1,
    // 1 - trans = 0, fact = 0 -> layer sources = 0, so the sweeps need no predication and the slot after
    // the last real layer naturally receives the surface values
    // (the consumer applies `i < np ? tau : 0`, where the value is used: selected here, at the request, the selects -- and the
    //  waits for the rows -- were scheduled at the END of the g-point that requested them, a few hundred cycles after the request)
    t.tau[i] = at(tau, o.lay[i]);
    t.lay[i] = at(lay, o.lay[i]);
  }
  if constexpr (FACT) {  // lay_source_ is the Planck fraction (ncol, nlay, ngpt): t.lay = the segment's, t.lev[0 / 1] = the rows above / below
    if (i0 == 0) t.lev[0] = at(lay, o.lev[0]);  // (the row above is consumed first, the row below last: refreshed in place)
    if (i1 == L) t.lev[1] = at(lay, o.lev[1]);
  } else {
#pragma unroll
    for (int i = 0; i <= L; ++i)
      if (i >= i0 && (i < i1 || i1 == L)) t.lev[i] = at(lev, o.lev[i]);
  }
  if (SFC && i0 == 0) {
    t.D = at(Dsec + ncg, o.cg);
    t.emis = at(sfc_emis + ncg, o.cg);
    t.ssrc = at(sfc_src + ncg, o.cg);
    t.inc = at(inc_flux + ncg, o.cg);
    t.sjac = do_jac ? at(sfc_srcJac + ncg, o.cg) : (Float)0;
  }
}

// SFCLDS (8-wave blocks only): the per-(column, g-point) arrays -- secant, surface emissivity and source, incident flux
// [, surface source Jacobian] -- are fetched ONCE per block in chunks of 16 g-points, each wave loading two g-points'
// worth a chunk ahead, and handed to all waves through LDS; otherwise every wave loads all of them for every g-point
// (8 waves x 4 arrays of the same 512 bytes: 32 of the block's 232 load instructions per g-point, 28 of them redundant).
// SPEC: spectral output (the interface's flux_up / flux_dn (ncol, nlay+1, ngpt), what rte_lw asks for with any ty_fluxes
// other than ty_fluxes_broadband, rte/frontend/mo_rte_lw.F90:297-321): every wave stores pi * weight * radiance at the
// levels it owns, per g-point, instead of accumulating; `spec_add` (angles after the first, :343-361) adds to what is
// there.  The Jacobian stays a broadband quantity (partial slabs as before).
// BYBAND (extension rte_hip_lw_solver_noscat_byband): one block per (column tile, BAND) -- grid.y = band, the g-point range
// comes from band_lims_gpt -- and the block's sums leave as the by-band fluxes themselves, pi * weight applied (what
// rte_sum_byband of the spectral arrays gives, rte/extensions/mo_fluxes_byband.F90:46-137, without the spectral arrays).
// FACT (extension rte_hip_lw_solver_noscat_factored): the sources arrive FACTORED, as rte_hip_compute_Planck_source_factored
// leaves them -- the Planck fraction (ncol, nlay, ngpt) in lay_source_, the band's Planck function at the layer and level
// temperatures (ncol, nlay, nbnd) in plk_lay and (ncol, nlay + 1, nbnd) in lev_source_ -- and the kernel forms
// lay_source = pfrac * planck_lay (:674) and lev_source = sqrt(pfrac(above) * pfrac(below)) * planck_lev (:695-705, the fraction
// itself at the column's two ends) with the operations of compute_Planck_source: the same bits, a third less to read
// (8 + 10 instead of 8 + 8 + 9 rows per g-point and wave) and 26 GB the gas optics no longer write at 1e5 x 60 x 256.
// The body of lw_noscat_seg_kernel for ONE wave, whose segment -- up to L layers -- starts p0 layers below the top: the kernel
// proper (every wave L layers, p0 = s L) and lw_noscat_seg_mixed_kernel (segments of two lengths) are thin wrappers.
template <int L, bool do_jac, bool SFCLDS, bool SPEC, bool BYBAND, bool FACT>
__device__ __forceinline__ void
lw_noscat_seg_wave(const int p0, int ncol, int nlay, int ngpt, int S, int g_per_block, bool top_at_1, Float weight,
                   const Float* __restrict__ Dsec, const Float* __restrict__ tau_,
                   const Float* __restrict__ lay_source_, const Float* __restrict__ lev_source_,
                   const Float* __restrict__ sfc_emis, const Float* __restrict__ sfc_src,
                   const Float* __restrict__ inc_flux, const Float* __restrict__ sfc_srcJac,
                   Float* __restrict__ part_up, Float* __restrict__ part_dn, Float* __restrict__ part_jac,
                   Float* __restrict__ spec_up, Float* __restrict__ spec_dn, bool spec_add,
                   const int* __restrict__ band_lims, const Float* __restrict__ plk_lay) {
#pragma clang fp contract(fast)  // this kernel is fp64-issue bound: fuse the recurrences' a*b+c
  extern __shared__ Float lds[];  // [2 buffers][3 (Td,Sd,Su)][MAXS][64]
  const int lane = threadIdx.x & 63;
  // the wave index is wave-uniform: tell the compiler, so layer offsets live in SGPRs and every load is
  // (uniform 64-bit base) + (32-bit lane offset) instead of per-lane 64-bit address arithmetic
  const int s = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int icol = blockIdx.x * 64 + lane;
  const bool active = icol < ncol;
  const int c = active ? icol : ncol - 1;  // clamp: inactive lanes compute on a valid column, never store
  const int nlev = nlay + 1;
  const size_t nclv = (size_t)ncol * nlev;
  const int np = min(L, nlay - p0);  // layers in this segment (>= 1 by construction)
  const bool last = (s == S - 1);
  const Float piw = kPi * weight;
  const Float inv_piw = (Float)1 / piw;
  const int g_begin = BYBAND ? band_lims[2 * blockIdx.y] - 1 : blockIdx.y * g_per_block;
  const int g_end = BYBAND ? band_lims[2 * blockIdx.y + 1] : min(ngpt, g_begin + g_per_block);
  constexpr int MAXS = 8;  // waves per block at most
  constexpr int CH = 16, NA = do_jac ? 5 : 4, RPW = 2 * NA;  // chunk of g-points, arrays, rows loaded per wave and chunk
  Float* const SFCB = lds + 2 * 3 * MAXS * 64;  // [2 buffers][CH][NA][64]
  // neutral composites (Td, Sd, Su) = (1, 0, 0) for the segment slots no wave owns (read after the first barrier)
  for (int i = threadIdx.x; i < 2 * 3 * MAXS * 64; i += blockDim.x) {
    const int q = (i >> 6) % MAXS, k = (i >> 6) / MAXS % 3;
    if (q >= S) lds[i] = k == 0 ? (Float)1 : (Float)0;
  }

  Float acc_dn[SPEC ? 1 : L + 1], acc_up[SPEC ? 1 : L + 1], acc_j[do_jac ? L + 1 : 1];
#pragma unroll
  for (int i = 0; i <= L; ++i) { if (!SPEC) { acc_dn[i] = 0; acc_up[i] = 0; } if (do_jac) acc_j[i] = 0; }
  SegOffsets<L> offs;
  seg_offsets<L, FACT>(offs, c, ncol, nlay, p0, np, top_at_1);
  // FACT: the band's Planck function at this wave's layers and levels (reloaded when the g-point loop enters the next band)
  // and, per level slot, whether the rows above and below it are the same row (the column's ends, and the repeated bottom
  // level of a partial last segment): the source there is the fraction itself, :695 / :705
  // ... in registers where they fit beside the single in-place tile (6.45 -> 6.12 ms against the thread's own LDS slots, which the
  // 10-layer variant with Jacobians keeps: it would spill)
  constexpr bool PLKREG = L <= 9 || !do_jac;
  Float plk_r[FACT && PLKREG ? 2 * L + 1 : 1];
  Float* const plk_l = SFCB + (SFCLDS ? 2 * CH * NA * 64 : 0) + (size_t)s * (2 * L + 1) * 64 + lane;
  struct PlkRef {  // element i * 64 as the LDS layout has it: ply[i] at [i * 64], plv[i] at [(L + i) * 64]
    Float *r, *l;
    __device__ __forceinline__ Float& operator[](int i) const { return PLKREG ? r[i / 64] : l[i]; }
  };
  const PlkRef PLK{plk_r, plk_l};
  int band = 0, band_end = 0;  // FACT: current band (0-based) and the first g-point after it
  auto load_band = [&](int igpt) {
    while (band_lims[2 * band + 1] <= igpt) ++band;  // (1-based inclusive limits)
    band_end = band_lims[2 * band + 1];
    // rows of the (ncol, nlay) / (ncol, nlay + 1) band planes: the layer offsets of the g-point loads; a level row is its layer's
    // row or the one after it, by orientation (slot i = top of layer i) and past the last layer
    const char* pl = reinterpret_cast<const char*>(plk_lay + (size_t)ncol * nlay * band);
    const char* pv = reinterpret_cast<const char*>(lev_source_ + nclv * band);
#pragma unroll
    for (int i = 0; i < L; ++i) PLK[i * 64] = *reinterpret_cast<const Float*>(pl + offs.lay[i]);
#pragma unroll
    for (int i = 0; i <= L; ++i) {
      const unsigned step = ((i < np) != top_at_1) ? (unsigned)ncol * (unsigned)sizeof(Float) : 0u;  // wave-uniform
      PLK[(L + i) * 64] = *reinterpret_cast<const Float*>(pv + (offs.lay[i < L ? i : L - 1] + step));
    }
  };
  // level slot i of this wave: accumulate (broadband) or store the flux of g-point `ig` (spectral; owned levels only)
  auto put = [&](Float* acc, Float* __restrict__ spec, int i, Float v, int ig) {
    if constexpr (SPEC) {
      if (active && (i < np || (last && i == np))) {
        Float* q = reinterpret_cast<Float*>(reinterpret_cast<char*>(spec + nclv * ig) + offs.lev[i]);
        const Float f = v * piw;
        if (spec_add) *q = *q + f; else rte::store_stream(q, f);
      }
    } else {
      acc[i] += v;
    }
  };

  // One g-point of work on tile `cur` (padded slots are neutral, see seg_load: no predication)
  // `request(i0, i1)`: the next g-point's rows of slots [i0, i1), issued BETWEEN the layer pairs of pass 1 (fenced on both sides).
  // Issued in one burst at the top of the g-point, the block's 200-264 row requests kept every wave stalled at its load
  // instructions for 2400-4400 cycles of the 8400 a g-point took (tools/time_lw_phases.py: the CU's memory pipeline takes
  // ~24 cycles per 512-byte row, whether the rows come from HBM or from the caches), with only the SIMD's other wave computing.
  auto process = [&](SegTile<L>& cur_, int buf, int gl, int cbuf, int ig, auto&& request) {
#pragma clang fp contract(fast)
    struct Sfc { Float D, emis, ssrc, inc, sjac; } cur;
    if (SFCLDS) {
      const Float* q = SFCB + ((size_t)(cbuf * CH + gl) * NA) * 64 + lane;
      cur.D = q[0]; cur.emis = q[64]; cur.ssrc = q[128]; cur.inc = q[192]; cur.sjac = do_jac ? q[(NA - 1) * 64] : (Float)0;
    } else {
      cur.D = cur_.D; cur.emis = cur_.emis; cur.ssrc = cur_.ssrc; cur.inc = cur_.inc; cur.sjac = cur_.sjac;
    }
    Float t[L], sd[L], su[L];
    // FACT: the level source at slot i of this g-point, :695 / :699 / :705 (a rounded product, as the array element it replaces:
    // no contraction into its users)
    auto level_src = [&](int i) {
      const Float pa = i == 0 ? cur_.lev[0] : cur_.lay[i - 1], pb = i == L ? cur_.lev[1] : cur_.lay[i];
      // at the column's ends (and on the repeated bottom level of a partial last segment) both rows are the same row, and the
      // correctly rounded root of the rounded square is the fraction itself (:695 / :705) for every value whose square neither
      // underflows nor overflows -- no select (measured: 18 v_cndmask per g-point and wave in a kernel bound by its issue)
      const Float gm = rte::sqrt_cr0(pa * pb);
      Float v = gm * PLK[(L + i) * 64];
      asm volatile("" : "+v"(v));
      return v;
    };
    Float lv_hi = 0;
    if constexpr (FACT) lv_hi = level_src(0);
    // ---- pass 1: layer transmissivities and sources (:180-190), segment composites
    Float Td = 1, Sd = 0;
#pragma unroll
    for (int i = 0; i < L; ++i) {
      const Float tau_loc = (i < np ? cur_.tau[i] : (Float)0) * cur.D;  // neutral slot (see seg_load)
      const Float tr = rte::exp_nonpos(-tau_loc);
      // lw_source_layer(lo, hi) returns (inc: uses hi, dec: uses lo): "toward bottom" uses the
      // bottom level source, "toward top" the top level source
      if constexpr (FACT) {
        Float ls = cur_.lay[i] * PLK[i * 64];  // :674
        asm volatile("" : "+v"(ls));
        const Float lv_lo = level_src(i + 1);
        lw_source_layer_fast(tau_loc, tr, ls, lv_hi, lv_lo, sd[i], su[i]);
        lv_hi = lv_lo;
      } else
      lw_source_layer_fast(tau_loc, tr, cur_.lay[i], cur_.lev[i], cur_.lev[i + 1], sd[i], su[i]);
      t[i] = tr;
      Sd = tr * Sd + sd[i];
      Td = Td * tr;
      if (i & 1) {
        __builtin_amdgcn_sched_barrier(0);  // bound the interleaving (register pressure) to 2 layers
        request(i - 1, i + 1 == L ? L : i + 1);
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    if constexpr ((L & 1) != 0) {
      request(L - 1, L);
      __builtin_amdgcn_sched_barrier(0);
    }
    Float Su = 0;
#pragma unroll
    for (int i = L - 1; i >= 0; --i) Su = t[i] * Su + su[i];
    // ---- exchange segment composites: slots [buffer][Td, Sd, Su][MAXS segments][64 lanes]; slots of segments
    // that do not exist hold the neutral composite (1, 0, 0), so the chains below have a fixed length and all
    // their LDS reads are issued back to back (a read per chain step costs one LDS latency per step)
    Float* X = lds + (size_t)buf * 3 * MAXS * 64;
    X[(0 * MAXS + s) * 64 + lane] = Td;
    X[(1 * MAXS + s) * 64 + lane] = Sd;
    X[(2 * MAXS + s) * 64 + lane] = Su;
    __syncthreads();
    Float r = cur.inc * inv_piw;  // radiance entering segment 0 from above (:144)
    Float r_in = r;
    Float u, jv;
    if constexpr (L <= 8 && !do_jac) {  // the register budget allows all composites at once
      Float Tq[MAXS], Sq[MAXS];
#pragma unroll
      for (int q = 0; q < MAXS; ++q) { Tq[q] = X[(0 * MAXS + q) * 64 + lane]; Sq[q] = X[(1 * MAXS + q) * 64 + lane]; }
#pragma unroll
      for (int q = 0; q < MAXS; ++q) {
        r_in = (q == s) ? r : r_in;
        r = Tq[q] * r + Sq[q];
      }
#pragma unroll
      for (int q = 1; q < MAXS; ++q) Sq[q] = X[(2 * MAXS + q) * 64 + lane];  // Su of the segments below
      u = r * ((Float)1 - cur.emis) + cur.emis * cur.ssrc;  // :198-200
      jv = 0;
#pragma unroll
      for (int q = MAXS - 1; q >= 1; --q)
        if (q > s) u = Tq[q] * u + Sq[q];  // wave-uniform
    } else {
      for (int q = 0; q < S; ++q) {
        if (q == s) r_in = r;
        r = X[(0 * MAXS + q) * 64 + lane] * r + X[(1 * MAXS + q) * 64 + lane];
      }
      u = r * ((Float)1 - cur.emis) + cur.emis * cur.ssrc;  // :198-200
      jv = do_jac ? cur.emis * cur.sjac : (Float)0;
      for (int q = S - 1; q > s; --q) {
        const Float Tq = X[(0 * MAXS + q) * 64 + lane];
        u = Tq * u + X[(2 * MAXS + q) * 64 + lane];
        jv = Tq * jv;
      }
    }
    // the level below the segment's last slot (used only by a FULL last segment: the surface)
    asm volatile("" : "+v"(u), "+v"(r_in));
    put(acc_up, spec_up, L, u, ig);
    if (do_jac) acc_j[L] += jv;
    // ---- pass 2: down; slot i is the level at the top of layer i.  In a partial last segment the
    // neutral slots i >= np all see the surface radiance, so slot np receives the surface value.
    r = r_in;
#pragma unroll
    for (int i = 0; i < L; ++i) {
      put(acc_dn, spec_dn, i, r, ig);
      r = t[i] * r + sd[i];
    }
    put(acc_dn, spec_dn, L, r, ig);
    // ---- pass 2: up (+ Jacobian, :729-743); neutral slots leave u at the surface value, which is
    // exactly what slot np of a partial last segment must accumulate
#pragma unroll
    for (int i = L - 1; i >= 0; --i) {
      u = t[i] * u + su[i];
      put(acc_up, spec_up, i, u, ig);
      if (do_jac) { jv = t[i] * jv; acc_j[i] += jv; }
    }
  };
  auto load = [&](SegTile<L>& tile, int igpt, int i0 = 0, int i1 = L) {
    seg_load<L, do_jac, !SFCLDS, FACT>(tile, offs, min(igpt, g_end - 1), ncol, nlay, np, Dsec, tau_, lay_source_, lev_source_,
                                       sfc_emis, sfc_src, inc_flux, sfc_srcJac, i0, i1);
  };
  // surface-array chunks: wave s fetches g-points 2s, 2s+1 of a chunk (row i: array i % NA of g-point 2s + i / NA)
  Float sfcpf[RPW];
  auto sfc_fetch = [&](int chunk) {
#pragma unroll
    for (int i = 0; i < RPW; ++i) {
      const int g = min(g_begin + chunk * CH + 2 * s + i / NA, g_end - 1);
      const Float* base = (i % NA) == 0 ? Dsec : (i % NA) == 1 ? sfc_emis : (i % NA) == 2 ? sfc_src : (i % NA) == 3 ? inc_flux : sfc_srcJac;
      unsigned off = offs.cg;
      asm volatile("" : "+v"(off));
      sfcpf[i] = *reinterpret_cast<const Float*>(reinterpret_cast<const char*>(base + (size_t)ncol * g) + off);
    }
  };
  auto sfc_publish = [&](int cbuf) {
#pragma unroll
    for (int i = 0; i < RPW; ++i) SFCB[((size_t)(cbuf * CH + 2 * s + i / NA) * NA + (i % NA)) * 64 + lane] = sfcpf[i];
  };
  if (SFCLDS) {
    sfc_fetch(0);
    sfc_publish(0);
    __syncthreads();
  }
  // software prefetch: the next g-point's loads are in flight while this one is computed.  (A ring of tiles
  // unrolled so that no tile is copied measures the same and needs 45 more registers; two g-points ahead spill.)
  // (round 3: without this prefetch the kernel needs 179 instead of 230 registers and is 1-4 % slower, 7.5-7.7 against
  // 7.4 ms at 1e5 x 60 x 256 -- the two waves of a SIMD cover most of each other's waits; kept, it fits)
  // ONE tile, refreshed in place: the rows of slots (i - 1, i) are requested for the next g-point right after pass 1 has consumed
  // them for this one (`request` in process), into the same registers.  (Round 1-3 kept a second tile, filled by a burst of requests
  // at the top of the g-point and copied at its end: 50 registers more, and the burst stalled the waves -- see process.)
  SegTile<L> cur;
  load(cur, g_begin);
  int buf = 0, gl = 0, chunk = 0;
  for (int igpt = g_begin; igpt < g_end;) {
    int g_stop = g_end;
    if constexpr (FACT) {  // band by band: the band's Planck functions are loaded between the g-point loops, not inside them
      load_band(igpt);
      g_stop = min(g_end, band_end);
    }
#pragma unroll 1
    for (; igpt < g_stop; ++igpt, buf ^= 1) {
      if (SFCLDS && gl == 0) sfc_fetch(chunk + 1);           // a chunk ahead, behind this g-point's barrier ...
      if (SFCLDS && gl == 1) sfc_publish((chunk + 1) & 1);   // ... written an iteration later, read 14 barriers later
      process(cur, buf, gl, chunk & 1, igpt, [&](int i0, int i1) { load(cur, igpt + 1, i0, i1); });
      if (++gl == CH) { gl = 0; ++chunk; }
    }
  }
  // ---- partial broadband for this g-group: (ncol, nlev, ngroups)
  if (active && (!SPEC || do_jac)) {
    const size_t base = icol + nclv * blockIdx.y;
#pragma unroll
    for (int i = 0; i <= L; ++i) {
      if (i < np || (last && i == np)) {
        const int p = p0 + i;  // level position from the top
        const int ilev = top_at_1 ? p : nlay - p;
        if constexpr (BYBAND) {  // the by-band fluxes themselves (angles after the first add)
          Float* qd = part_dn + base + (size_t)ncol * ilev;
          Float* qu = part_up + base + (size_t)ncol * ilev;
          *qd = spec_add ? *qd + acc_dn[i] * piw : acc_dn[i] * piw;
          *qu = spec_add ? *qu + acc_up[i] * piw : acc_up[i] * piw;
        } else if constexpr (!SPEC) {
          part_dn[base + (size_t)ncol * ilev] = acc_dn[i];
          part_up[base + (size_t)ncol * ilev] = acc_up[i];
        }
        if (do_jac) part_jac[base + (size_t)ncol * ilev] = acc_j[i];
      }
    }
  }
}

template <int L, bool do_jac, bool SFCLDS, bool SPEC = false, bool BYBAND = false, bool FACT = false>
__global__ void __launch_bounds__(64 * 8)
lw_noscat_seg_kernel(int ncol, int nlay, int ngpt, int S, int g_per_block, bool top_at_1, Float weight,
                     const Float* __restrict__ Dsec, const Float* __restrict__ tau_,
                     const Float* __restrict__ lay_source_, const Float* __restrict__ lev_source_,
                     const Float* __restrict__ sfc_emis, const Float* __restrict__ sfc_src,
                     const Float* __restrict__ inc_flux, const Float* __restrict__ sfc_srcJac,
                     Float* __restrict__ part_up, Float* __restrict__ part_dn, Float* __restrict__ part_jac,
                     Float* __restrict__ spec_up = nullptr, Float* __restrict__ spec_dn = nullptr, bool spec_add = false,
                     const int* __restrict__ band_lims = nullptr, const Float* __restrict__ plk_lay = nullptr) {
  const int s = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  lw_noscat_seg_wave<L, do_jac, SFCLDS, SPEC, BYBAND, FACT>(s * L, ncol, nlay, ngpt, S, g_per_block, top_at_1, weight, Dsec, tau_, lay_source_,
                                                           lev_source_, sfc_emis, sfc_src, inc_flux, sfc_srcJac, part_up, part_dn, part_jac,
                                                           spec_up, spec_dn, spec_add, band_lims, plk_lay);
}

// Segments of TWO lengths (broadband, 8 waves): waves 0-3 own LA layers each, waves 4-7 LB -- 4 x 7 + 4 x 8 = 60, where eight
// layers per wave leave the eighth wave four layers and four neutral slots whose rows it requests all the same (clamped: 200
// row requests per g-point and block for 188 rows; the kernel is bound by its row requests, see process).  As in
// sw_2stream_seg_mixed_kernel.
// (FACT: the factored-source instance, so that the deferred / factored step keeps the segments -- and with them the bits -- of
//  the plain one; at 7 and 8 layers per wave the band's Planck functions are in registers, no per-wave LDS slots to lay out)
template <int LA, int LB, bool do_jac, bool SFCLDS, bool FACT = false>
__global__ void __launch_bounds__(64 * 8)
lw_noscat_seg_mixed_kernel(int ncol, int nlay, int ngpt, int S, int g_per_block, bool top_at_1, Float weight,
                           const Float* __restrict__ Dsec, const Float* __restrict__ tau_,
                           const Float* __restrict__ lay_source_, const Float* __restrict__ lev_source_,
                           const Float* __restrict__ sfc_emis, const Float* __restrict__ sfc_src,
                           const Float* __restrict__ inc_flux, const Float* __restrict__ sfc_srcJac,
                           Float* __restrict__ part_up, Float* __restrict__ part_dn, Float* __restrict__ part_jac,
                           const int* __restrict__ band_lims = nullptr, const Float* __restrict__ plk_lay = nullptr) {
  static_assert(LA <= 9 && LB <= 9, "PLKREG: the band's Planck functions in registers");
  const int s = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  if (s < 4)
    lw_noscat_seg_wave<LA, do_jac, SFCLDS, false, false, FACT>(s * LA, ncol, nlay, ngpt, S, g_per_block, top_at_1, weight, Dsec, tau_, lay_source_,
                                                               lev_source_, sfc_emis, sfc_src, inc_flux, sfc_srcJac, part_up, part_dn, part_jac,
                                                               nullptr, nullptr, false, band_lims, plk_lay);
  else
    lw_noscat_seg_wave<LB, do_jac, SFCLDS, false, false, FACT>(4 * LA + (s - 4) * LB, ncol, nlay, ngpt, S, g_per_block, top_at_1, weight, Dsec, tau_,
                                                               lay_source_, lev_source_, sfc_emis, sfc_src, inc_flux, sfc_srcJac, part_up, part_dn,
                                                               part_jac, nullptr, nullptr, false, band_lims, plk_lay);
}

// seg_load with the row offsets formed where they are used: base + (clamped slot) * step, two integer operations per load
// instead of a register per row (the two-sub-segment kernel below has no registers to spare for 4 L + 2 offsets)
template <int L, bool do_jac, bool SFC>
__device__ __forceinline__ void seg_load2(SegTile<L>& t, unsigned lay0, unsigned lev0, unsigned step, unsigned cg, int np_off, int np,
                                          int igpt, int ncol, size_t ncl, size_t nclv, const Float* __restrict__ Dsec,
                                          const Float* __restrict__ tau_, const Float* __restrict__ lay_source_,
                                          const Float* __restrict__ lev_source_, const Float* __restrict__ sfc_emis,
                                          const Float* __restrict__ sfc_src, const Float* __restrict__ inc_flux,
                                          const Float* __restrict__ sfc_srcJac) {
  // ncl, nclv: elements between the g-point planes of the layer / level arrays (a WINDOW of layers keeps the column's)
  const size_t ncg = (size_t)ncol * igpt;
  auto at = [](const Float* plane, unsigned off) {  // plane is wave-uniform
    asm volatile("" : "+v"(off));
    return __builtin_nontemporal_load(reinterpret_cast<const Float*>(reinterpret_cast<const char*>(plane) + off));
  };
  const Float* tau = tau_ + ncl * igpt;
  const Float* lay = lay_source_ + ncl * igpt;
  const Float* lev = lev_source_ + nclv * igpt;
#pragma unroll
  for (int i = 0; i < L; ++i) {
    const unsigned o = lay0 + (unsigned)min(i, np_off - 1) * step;
    const Float tv = at(tau, o);
    t.tau[i] = i < np ? tv : (Float)0;  // neutral slot (see seg_load)
    t.lay[i] = at(lay, o);
  }
#pragma unroll
  for (int i = 0; i <= L; ++i) t.lev[i] = at(lev, lev0 + (unsigned)min(i, np_off) * step);
  if (SFC) {
    t.D = at(Dsec + ncg, cg);
    t.emis = at(sfc_emis + ncg, cg);
    t.ssrc = at(sfc_src + ncg, cg);
    t.inc = at(inc_flux + ncg, cg);
    t.sjac = do_jac ? at(sfc_srcJac + ncg, cg) : (Float)0;
  }
}

// ---------------------------------------------------------------------------------------------
// LW no-scattering, segmented, TWO sub-segments per wave: 81 ... 160 layers (host models at 91 / 128 / 137 levels).
//
// The reference has no layer limit (:697-743); the kernel above holds one segment's (trans, src_dn, src_up) in
// registers and cannot take more than 10 layers per wave without spilling.  Here wave s owns 2 L consecutive layers as
// sub-segments A (upper) and B (lower): pass 1 evaluates A, parks its 3 L values per thread in the thread's own LDS
// slots (98 ... 123 KB per block: the reason the surface arrays are not shared through LDS here), evaluates B into the
// registers A just left, and publishes the composite of A followed by B; after the exchange pass 2 sweeps A from LDS
// and B from registers.  Same arithmetic per layer as the one-segment kernel, same composites across waves.
// ---------------------------------------------------------------------------------------------
// WIN (columns of 177 ... 352 layers, solved as an upper and a lower window of at most 176 layers; host: lw_noscat_windows):
// the kernel works on a window of nlay layers of arrays whose g-point planes are the whole column's (pointers offset to the
// window's first row), takes the radiance entering at its top as it is, and leaves the downward radiance at its bottom and
// the upward radiance (and its Jacobian) at its top per (column, g-point).
struct LwWin {
  size_t plane_lay, plane_lev;  // elements between g-point planes of the layer / level arrays
  bool inc_is_radiance;         // inc_flux holds a radiance (the upper window's out_dn_bot), not a flux
  Float *out_dn_bot, *out_up_top, *out_jv_top;  // (ncol, ngpt), each optional
};
template <int L, bool do_jac, bool SPEC, bool WIN = false>
__global__ void __launch_bounds__(64 * 8)
lw_noscat_seg2_kernel(int ncol, int nlay, int ngpt, int S, int g_per_block, bool top_at_1, Float weight,
                      const Float* __restrict__ Dsec, const Float* __restrict__ tau_,
                      const Float* __restrict__ lay_source_, const Float* __restrict__ lev_source_,
                      const Float* __restrict__ sfc_emis, const Float* __restrict__ sfc_src,
                      const Float* __restrict__ inc_flux, const Float* __restrict__ sfc_srcJac,
                      Float* __restrict__ part_up, Float* __restrict__ part_dn, Float* __restrict__ part_jac,
                      Float* __restrict__ spec_up, Float* __restrict__ spec_dn, bool spec_add, LwWin win) {
#pragma clang fp contract(fast)
  constexpr int LT = 2 * L, MAXS = 8;
  // input prefetch (see process): kept where it fits; the widest variant and the Jacobian variants spilled 39-103 registers
  // with it and run without
  constexpr bool PREF2 = false;
  extern __shared__ Float lds[];  // [2 buffers][3 (Td,Sd,Su)][MAXS][64], then A's parked values [3][L][512]
  const int lane = threadIdx.x & 63;
  const int s = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int icol = blockIdx.x * 64 + lane;
  const bool active = icol < ncol;
  const int c = active ? icol : ncol - 1;
  const int nlev = nlay + 1;
  const size_t nclv = (size_t)ncol * nlev;  // one partial slab (the window's levels)
  const size_t pl_lay = WIN ? win.plane_lay : (size_t)ncol * nlay, pl_lev = WIN ? win.plane_lev : nclv;
  const int p0 = s * LT;
  const int np = min(LT, nlay - p0);  // layers of this wave (>= 1 by construction)
  const bool last = (s == S - 1);
  const Float piw = kPi * weight;
  const Float inv_piw = (Float)1 / piw;
  const int g_begin = blockIdx.y * g_per_block;
  const int g_end = min(ngpt, g_begin + g_per_block);
  Float* const PARK = lds + 2 * 3 * MAXS * 64 + threadIdx.x;  // value k of layer i at PARK[(k * L + i) * 512]
  for (int i = threadIdx.x; i < 2 * 3 * MAXS * 64; i += blockDim.x) {
    const int q = (i >> 6) % MAXS, k = (i >> 6) / MAXS % 3;
    if (q >= S) lds[i] = k == 0 ? (Float)1 : (Float)0;  // neutral composites of the segment slots no wave owns
  }
  // the wider variants are a dozen registers over the budget: the LAST NL upward-flux accumulators live in the thread's own
  // LDS slots (ds_add_f64), as many as the LDS left beside the parked values holds (L = 9: 6 of 38, L = 10: 3 of 42)
  constexpr int NL = SPEC ? 0 : (L == 9 ? 6 : (L == 10 ? 3 : 0));
  Float* const ACCL = lds + 2 * 3 * MAXS * 64 + 3 * L * 512 + threadIdx.x;  // slot k at ACCL[k * 512]
  Float acc_dn[SPEC ? 1 : LT + 1], acc_up[SPEC ? 1 : LT + 1 - NL], acc_j[do_jac ? LT + 1 : 1];
#pragma unroll
  for (int i = 0; i <= LT; ++i) {
    if (!SPEC) { acc_dn[i] = 0; if (i <= LT - NL) acc_up[i] = 0; else ACCL[(i - (LT + 1 - NL)) * 512] = 0; }
    if (do_jac) acc_j[i] = 0;
  }
  auto put_up = [&](int i, Float v) {  // broadband upward accumulation (compile-time i)
    if (i <= LT - NL) acc_up[i <= LT - NL ? i : 0] += v;
    else atomicAdd(&ACCL[(i - (LT + 1 - NL)) * 512], v);
  };
  // sub-segment A: layers p0 .. p0 + L - 1, B: the L after it.  A B without layers (the wave's last layers all in A)
  // points at valid rows and is neutral (tau = 0, see seg_load)
  const int npA = min(L, np), npB = max(0, np - L);
  // byte offsets inside one g-point plane: the wave's first level row, the step to the next row (layers and levels
  // advance alike), first layer / level rows of A and B
  const unsigned dlev = (top_at_1 ? (unsigned)ncol : 0u - (unsigned)ncol) * (unsigned)sizeof(Float);
  auto lay_row = [&](int p) { return ((unsigned)c + (unsigned)ncol * (unsigned)(top_at_1 ? p : nlay - 1 - p)) * (unsigned)sizeof(Float); };
  auto lev_row = [&](int p) { return ((unsigned)c + (unsigned)ncol * (unsigned)(top_at_1 ? p : nlay - p)) * (unsigned)sizeof(Float); };
  const int pB = min(p0 + L, nlay - 1);
  unsigned olev0 = lev_row(p0), layA = lay_row(p0), layB = lay_row(pB), levB = lev_row(pB), ocg = (unsigned)c * (unsigned)sizeof(Float);
  asm volatile("" : "+v"(olev0), "+v"(layA), "+v"(layB), "+v"(levB), "+v"(ocg));
  auto put = [&](Float* acc, Float* __restrict__ spec, int i, Float v, int ig) {
    if constexpr (SPEC) {
      if (active && (i < np || (last && i == np))) {
        Float* q = reinterpret_cast<Float*>(reinterpret_cast<char*>(spec + nclv * ig) + (olev0 + (unsigned)i * dlev));
        const Float f = v * piw;
        if (spec_add) *q = *q + f; else rte::store_stream(q, f);
      }
    } else {
      if (acc == acc_up) put_up(i, v); else acc[i] += v;
    }
  };
  auto loadA = [&](SegTile<L>& tile, int igpt) {
    seg_load2<L, do_jac, true>(tile, layA, olev0, dlev, ocg, npA, npA, min(igpt, g_end - 1), ncol, pl_lay, pl_lev, Dsec, tau_, lay_source_,
                               lev_source_, sfc_emis, sfc_src, inc_flux, sfc_srcJac);
  };
  auto loadB = [&](SegTile<L>& tile, int igpt) {
    seg_load2<L, do_jac, false>(tile, layB, levB, dlev, ocg, max(npB, 1), npB, min(igpt, g_end - 1), ncol, pl_lay, pl_lev, Dsec, tau_,
                                lay_source_, lev_source_, sfc_emis, sfc_src, inc_flux, sfc_srcJac);
  };
  // One g-point.  The inputs of a sub-segment are dead after its pass 1: the next g-point's are requested into the same
  // registers right there (in flight during the rest of this g-point) -- one set of input registers, not two
  auto process = [&](SegTile<L>& ta, SegTile<L>& tb, int buf, int ig) {
#pragma clang fp contract(fast)
    if constexpr (!PREF2) loadA(ta, ig);
    Float t[L], sd[L], su[L];
    const Float D = ta.D, emis = ta.emis, ssrc = ta.ssrc, inc = ta.inc, sjac = ta.sjac;
    auto pass1 = [&](const SegTile<L>& x, Float& Td, Float& Sd, Float& Su) {
      Td = 1; Sd = 0;
#pragma unroll
      for (int i = 0; i < L; ++i) {
        const Float tau_loc = x.tau[i] * D;
        const Float tr = rte::exp_nonpos(-tau_loc);
        lw_source_layer_fast(tau_loc, tr, x.lay[i], x.lev[i], x.lev[i + 1], sd[i], su[i]);
        t[i] = tr;
        Sd = tr * Sd + sd[i];
        Td = Td * tr;
        __builtin_amdgcn_sched_barrier(0);  // one layer at a time: this kernel has no registers for interleaved layers
      }
      Su = 0;
#pragma unroll
      for (int i = L - 1; i >= 0; --i) Su = t[i] * Su + su[i];
    };
    Float TdA, SdA, SuA, TdB, SdB, SuB;
    // Register budget (a double is two VGPRs): accumulators 4 L + 2, a sub-segment's inputs 3 L + 1, its (t, sd, su) 3 L.
    // Only ONE set of inputs may be in flight besides the one being consumed: A's next are requested after pass 1 of B
    // (in flight during the exchange and pass 2), B's next at the very end (in flight during the next pass 1 of A).
    pass1(ta, TdA, SdA, SuA);
#pragma unroll
    for (int i = 0; i < L; ++i) { PARK[(0 * L + i) * 512] = t[i]; PARK[(1 * L + i) * 512] = sd[i]; PARK[(2 * L + i) * 512] = su[i]; }
    __builtin_amdgcn_sched_barrier(0);
    if constexpr (!PREF2) loadB(tb, ig);
    pass1(tb, TdB, SdB, SuB);
    __builtin_amdgcn_sched_barrier(0);
    if constexpr (PREF2) loadA(ta, ig + 1);
    // A above B: down through A then B; up through B then A
    Float* X = lds + (size_t)buf * 3 * MAXS * 64;
    X[(0 * MAXS + s) * 64 + lane] = TdA * TdB;
    X[(1 * MAXS + s) * 64 + lane] = TdB * SdA + SdB;
    X[(2 * MAXS + s) * 64 + lane] = TdA * SuB + SuA;
    __syncthreads();
    Float r = (WIN && win.inc_is_radiance) ? inc : inc * inv_piw;  // radiance entering segment 0 from above (:144)
    Float r_in = r;
    for (int q = 0; q < S; ++q) {
      if (q == s) r_in = r;
      r = X[(0 * MAXS + q) * 64 + lane] * r + X[(1 * MAXS + q) * 64 + lane];
    }
    if constexpr (WIN) {  // (r: the downward radiance below the last wave's last layer)
      if (win.out_dn_bot && s == 0 && active) win.out_dn_bot[icol + (size_t)ncol * ig] = r;
    }
    Float u = r * ((Float)1 - emis) + emis * ssrc;  // :198-200
    Float jv = do_jac ? emis * sjac : (Float)0;
    for (int q = S - 1; q > s; --q) {
      const Float Tq = X[(0 * MAXS + q) * 64 + lane];
      u = Tq * u + X[(2 * MAXS + q) * 64 + lane];
      jv = Tq * jv;
    }
    put(acc_up, spec_up, LT, u, ig);  // the level below the wave's last slot (a FULL last wave: the surface)
    if (do_jac) acc_j[LT] += jv;
    // ---- pass 2, down: A from LDS, B from registers (neutral slots carry the value on: slot np gets the surface value)
    r = r_in;
#pragma unroll
    for (int i = 0; i < L; ++i) {
      put(acc_dn, spec_dn, i, r, ig);
      r = PARK[(0 * L + i) * 512] * r + PARK[(1 * L + i) * 512];
    }
#pragma unroll
    for (int i = 0; i < L; ++i) {
      put(acc_dn, spec_dn, L + i, r, ig);
      r = t[i] * r + sd[i];
    }
    put(acc_dn, spec_dn, LT, r, ig);
    // ---- pass 2, up (+ Jacobian, :729-743): B, then A
#pragma unroll
    for (int i = L - 1; i >= 0; --i) {
      u = t[i] * u + su[i];
      put(acc_up, spec_up, L + i, u, ig);
      if (do_jac) { jv = t[i] * jv; acc_j[L + i] += jv; }
    }
#pragma unroll
    for (int i = L - 1; i >= 0; --i) {
      const Float ta_ = PARK[(0 * L + i) * 512];
      u = ta_ * u + PARK[(2 * L + i) * 512];
      put(acc_up, spec_up, i, u, ig);
      if (do_jac) { jv = ta_ * jv; acc_j[i] += jv; }
    }
    if constexpr (WIN) {  // (wave 0: u, jv are at the window's top level)
      if (s == 0 && active) {
        if (win.out_up_top) win.out_up_top[icol + (size_t)ncol * ig] = u;
        if (do_jac && win.out_jv_top) win.out_jv_top[icol + (size_t)ncol * ig] = jv;
      }
    }
    __builtin_amdgcn_sched_barrier(0);
    if constexpr (PREF2) loadB(tb, ig + 1);
  };
  SegTile<L> ta, tb;
  if constexpr (PREF2) { loadA(ta, g_begin); loadB(tb, g_begin); }
  int buf = 0;
#pragma unroll 1
  for (int igpt = g_begin; igpt < g_end; ++igpt, buf ^= 1) process(ta, tb, buf, igpt);
  if (active && (!SPEC || do_jac)) {
    const size_t base = icol + nclv * blockIdx.y;
#pragma unroll
    for (int i = 0; i <= LT; ++i) {
      if (i < np || (last && i == np)) {
        const int p = p0 + i;
        const int ilev = top_at_1 ? p : nlay - p;
        if constexpr (!SPEC) {
          part_dn[base + (size_t)ncol * ilev] = acc_dn[i];
          part_up[base + (size_t)ncol * ilev] = i <= LT - NL ? acc_up[i <= LT - NL ? i : 0] : ACCL[(i - (LT + 1 - NL)) * 512];
        }
        if (do_jac) part_jac[base + (size_t)ncol * ilev] = acc_j[i];
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------
// LW two-stream, generic: reference :377-440 (lw_two_stream :854-909, lw_source_2str :917-967,
// adding :1135-1245).  ws: 4 layer slabs (ncol, nlay, gchunk): Rdif, Tdif, src_dn, denom.
// flux_up temporarily holds `src`, flux_dn holds `albedo` (each level is read before it is
// overwritten in the final downward pass).
// ---------------------------------------------------------------------------------------------
struct Lw2Args {
  int ncol, nlay, ngpt, g_begin;
  bool top_at_1, lev_gpt1;
  const Float *tau, *ssa, *g, *lay_source, *lev_source, *sfc_emis, *sfc_src, *inc_flux;
  Float *flux_up, *flux_dn, *ws;
};

__global__ void __launch_bounds__(256) lw_2stream_generic_kernel(Lw2Args a) {
  const int icol = blockIdx.x * blockDim.x + threadIdx.x;
  const int gl = blockIdx.y;
  if (icol >= a.ncol) return;
  const int igpt = a.g_begin + gl;
  const int ncol = a.ncol, nlay = a.nlay, nlev = nlay + 1;
  const size_t ncl = (size_t)ncol * nlay, nclv = (size_t)ncol * nlev;
  const size_t cg = icol + (size_t)ncol * igpt;
  const Float* tau = a.tau + icol + ncl * igpt;
  const Float* ssa = a.ssa + icol + ncl * igpt;
  const Float* gg = a.g + icol + ncl * igpt;
  const Float* lev_source = a.lev_source + icol + nclv * (a.lev_gpt1 ? 0 : igpt);
  Float* fup = a.flux_up + icol + nclv * igpt;
  Float* fdn = a.flux_dn + icol + nclv * igpt;
  const size_t slab = ncl * gridDim.y;
  Float* wR = a.ws + icol + ncl * gl;
  Float* wT = wR + slab;
  Float* wSd = wT + slab;
  Float* wDen = wSd + slab;
  const Float LW_diff_sec = (Float)1.66f;  // :870: default-real literal widened to wp
  const Float emis = a.sfc_emis[cg];
  Float albedo = (Float)1 - emis;                 // :428
  Float src = kPi * emis * a.sfc_src[cg];         // :965
  const int sfc_level = a.top_at_1 ? nlay : 0, top_level = a.top_at_1 ? 0 : nlay;
  fdn[(size_t)ncol * sfc_level] = albedo;
  fup[(size_t)ncol * sfc_level] = src;
  // ---- bottom -> top: layer properties, sources, albedo / source recurrences (adding :1174-1186 / :1214-1226)
  for (int k = 0; k < nlay; ++k) {
    const int ilay = a.top_at_1 ? nlay - 1 - k : k;
    const size_t o = (size_t)ncol * ilay;
    const Float t = tau[o], w0 = ssa[o], g = gg[o];
    const Float gamma1 = LW_diff_sec * ((Float)1 - (Float)0.5 * w0 * ((Float)1 + g));
    const Float gamma2 = LW_diff_sec * (Float)0.5 * w0 * ((Float)1 - g);
    const Float kk = sqrt(fmax((gamma1 - gamma2) * (gamma1 + gamma2), (Float)1.e-12));
    const Float e1 = exp(-t * kk);
    const Float e2 = e1 * e1;
    const Float RT = (Float)1 / (kk * ((Float)1 + e2) + gamma1 * ((Float)1 - e2));
    const Float Rdif = RT * gamma2 * ((Float)1 - e2);
    const Float Tdif = RT * (Float)2 * kk * e1;
    const Float lev_a = lev_source[(size_t)ncol * ilay], lev_b = lev_source[(size_t)ncol * (ilay + 1)];
    const Float lev_top = a.top_at_1 ? lev_a : lev_b, lev_bot = a.top_at_1 ? lev_b : lev_a;
    Float s_up, s_dn;
    if (t > (Float)1.0e-8) {
      const Float Z = (lev_bot - lev_top) / (t * (gamma1 + gamma2));
      const Float Zup_top = Z + lev_top, Zup_bottom = Z + lev_bot;
      const Float Zdn_top = -Z + lev_top, Zdn_bottom = -Z + lev_bot;
      s_up = kPi * (Zup_top - Rdif * Zdn_top - Tdif * Zup_bottom);
      s_dn = kPi * (Zdn_bottom - Rdif * Zup_bottom - Tdif * Zdn_top);
    } else {
      s_up = 0;
      s_dn = 0;
    }
    const Float denom = (Float)1 / ((Float)1 - Rdif * albedo);
    const Float src_new = s_up + Tdif * denom * (src + albedo * s_dn);
    const Float alb_new = Rdif + Tdif * Tdif * albedo * denom;
    albedo = alb_new;
    src = src_new;
    const int lev_above = a.top_at_1 ? ilay : ilay + 1;
    fdn[(size_t)ncol * lev_above] = albedo;
    fup[(size_t)ncol * lev_above] = src;
    wR[o] = Rdif; wT[o] = Tdif; wSd[o] = s_dn; wDen[o] = denom;
  }
  // ---- top boundary and top -> bottom fluxes (:1188-1202 / :1228-1243)
  Float fd = a.inc_flux[cg];  // :432
  fdn[(size_t)ncol * top_level] = fd;
  fup[(size_t)ncol * top_level] = fd * albedo + src;
  for (int k = 0; k < nlay; ++k) {
    const int ilay = a.top_at_1 ? k : nlay - 1 - k;
    const int lev_below = a.top_at_1 ? ilay + 1 : ilay;
    const size_t o = (size_t)ncol * ilay, ol = (size_t)ncol * lev_below;
    const Float alb = fdn[ol], sr = fup[ol];
    fd = (wT[o] * fd + wR[o] * sr + wSd[o]) * wDen[o];
    fdn[ol] = fd;
    fup[ol] = fd * alb + sr;
  }
}

// ---------------------------------------------------------------------------------------------
// SW direct beam only: reference :450-494
// ---------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256)
sw_noscat_kernel(int ncol, int nlay, int ngpt, bool top_at_1, const Float* __restrict__ tau,
                 const Float* __restrict__ mu0, const Float* __restrict__ inc_flux_dir, Float* __restrict__ flux_dir) {
  const int icol = blockIdx.x * blockDim.x + threadIdx.x;
  const int igpt = blockIdx.y;
  if (icol >= ncol) return;
  const size_t ncl = (size_t)ncol * nlay, nclv = (size_t)ncol * (nlay + 1);
  const Float* t = tau + icol + ncl * igpt;
  Float* f = flux_dir + icol + nclv * igpt;
  if (top_at_1) {
    Float v = inc_flux_dir[icol + (size_t)ncol * igpt] * mu0[icol];
    f[0] = v;
    for (int ilev = 1; ilev <= nlay; ++ilev) {
      v = v * exp(-t[(size_t)ncol * (ilev - 1)] / mu0[icol + (size_t)ncol * (ilev - 1)]);
      f[(size_t)ncol * ilev] = v;
    }
  } else {
    Float v = inc_flux_dir[icol + (size_t)ncol * igpt] * mu0[icol + (size_t)ncol * (nlay - 1)];
    f[(size_t)ncol * nlay] = v;
    for (int ilev = nlay - 1; ilev >= 0; --ilev) {
      v = v * exp(-t[(size_t)ncol * ilev] / mu0[icol + (size_t)ncol * ilev]);
      f[(size_t)ncol * ilev] = v;
    }
  }
}

// ---------------------------------------------------------------------------------------------
// SW two-stream, generic: reference :503-609 (sw_dif_and_source :985-1127, adding :1135-1245)
// up/dn/dir: (ncol, nlay+1, gchunk) slabs (the caller's spectral arrays, or scratch when
// broadband).  ws: 5 layer slabs (ncol, nlay, gchunk): Rdif, Tdif, src_up, src_dn, denom.
// `up` temporarily holds src, `dn` holds albedo (as in the LW two-stream kernel).
// ---------------------------------------------------------------------------------------------
struct Sw2Args {
  int ncol, nlay, ngpt, g_begin;
  bool top_at_1, has_dif_bc, add_dir_to_dn;
  const Float *tau, *ssa, *g, *mu0, *sfc_alb_dir, *sfc_alb_dif, *inc_flux_dir, *inc_flux_dif;
  Float *up, *dn, *dir, *ws;
};

__global__ void __launch_bounds__(256) sw_2stream_generic_kernel(Sw2Args a) {
  const int icol = blockIdx.x * blockDim.x + threadIdx.x;
  const int gl = blockIdx.y;
  if (icol >= a.ncol) return;
  const int igpt = a.g_begin + gl;
  const int ncol = a.ncol, nlay = a.nlay, nlev = nlay + 1;
  const size_t ncl = (size_t)ncol * nlay, nclv = (size_t)ncol * nlev;
  const size_t cg = icol + (size_t)ncol * igpt;
  const Float* tau = a.tau + icol + ncl * igpt;
  const Float* ssa = a.ssa + icol + ncl * igpt;
  const Float* gg = a.g + icol + ncl * igpt;
  const Float* mu0 = a.mu0 + icol;
  Float* fup = a.up + icol + nclv * gl;
  Float* fdn = a.dn + icol + nclv * gl;
  Float* fdir = a.dir + icol + nclv * gl;
  const size_t slab = ncl * gridDim.y;
  Float* wR = a.ws + icol + ncl * gl;
  Float* wT = wR + slab;
  Float* wSu = wT + slab;
  Float* wSd = wSu + slab;
  Float* wDen = wSd + slab;
  const Float min_k = (Float)1.e4 * (Float)RTE_EPS;
  const Float min_mu0 = sqrt((Float)RTE_EPS);
  const int top_level = a.top_at_1 ? 0 : nlay, top_layer = a.top_at_1 ? 0 : nlay - 1;
  const int sfc_level = a.top_at_1 ? nlay : 0, sfc_layer = a.top_at_1 ? nlay - 1 : 0;

  // ---- top -> bottom: layer R/T, direct beam, sources (:1015-1114)
  Float dir = a.inc_flux_dir[cg] * mu0[(size_t)ncol * top_layer];  // :575
  fdir[(size_t)ncol * top_level] = dir;
  for (int k = 0; k < nlay; ++k) {
    const int ilay = a.top_at_1 ? k : nlay - 1 - k;
    const size_t o = (size_t)ncol * ilay;
    const Float tau_s = tau[o], w0_s = ssa[o], g_s = gg[o];
    const Float gamma1 = ((Float)8 - w0_s * ((Float)5 + (Float)3 * g_s)) * (Float).25;
    const Float gamma2 = (Float)3 * (w0_s * ((Float)1 - g_s)) * (Float).25;
    const Float kk = sqrt(fmax((gamma1 - gamma2) * (gamma1 + gamma2), min_k));
    const Float e1 = exp(-tau_s * kk);
    const Float e2 = e1 * e1;
    Float RT = (Float)1 / (kk * ((Float)1 + e2) + gamma1 * ((Float)1 - e2));
    const Float Rdif = RT * gamma2 * ((Float)1 - e2);
    const Float Tdif = RT * (Float)2 * kk * e1;
    const Float mu0_raw = mu0[o];
    const Float mu0_s = fmax(min_mu0, mu0_raw);
    const Float k_mu = kk * mu0_s;
    const Float om = (Float)1 - k_mu * k_mu;
    RT = w0_s * RT / (fabs(om) >= (Float)RTE_EPS ? om : (Float)RTE_EPS);
    const Float gamma3 = ((Float)2 - (Float)3 * mu0_s * g_s) * (Float).25;
    const Float gamma4 = (Float)1 - gamma3;
    const Float alpha1 = gamma1 * gamma4 + gamma2 * gamma3;
    const Float alpha2 = gamma1 * gamma3 + gamma2 * gamma4;
    const Float k_gamma3 = kk * gamma3, k_gamma4 = kk * gamma4;
    const Float Tnoscat = exp(-tau_s / mu0_s);
    Float Rdir = RT * (((Float)1 - k_mu) * (alpha2 + k_gamma3) - ((Float)1 + k_mu) * (alpha2 - k_gamma3) * e2 -
                       (Float)2.0 * (k_gamma3 - alpha2 * k_mu) * e1 * Tnoscat);
    Float Tdir = -RT * (((Float)1 + k_mu) * (alpha1 + k_gamma4) * Tnoscat -
                        ((Float)1 - k_mu) * (alpha1 - k_gamma4) * e2 * Tnoscat -
                        (Float)2.0 * (k_gamma4 + alpha1 * k_mu) * e1);
    Rdir = fmax((Float)0, fmin(Rdir, ((Float)1 - Tnoscat)));
    Tdir = fmax((Float)0, fmin(Tdir, ((Float)1 - Tnoscat - Rdir)));
    Float s_up = Rdir * dir, s_dn = Tdir * dir;
    dir = Tnoscat * dir;
    if (mu0_raw <= (Float)0) { s_up = 0; s_dn = 0; }  // :1122-1125
    fdir[(size_t)ncol * (a.top_at_1 ? ilay + 1 : ilay)] = dir;
    wR[o] = Rdif; wT[o] = Tdif; wSu[o] = s_up; wSd[o] = s_dn;
  }
  // :1120-1121
  Float src = (mu0[(size_t)ncol * sfc_layer] > (Float)0) ? dir * a.sfc_alb_dir[cg] : (Float)0;
  Float albedo = a.sfc_alb_dif[cg];
  fdn[(size_t)ncol * sfc_level] = albedo;
  fup[(size_t)ncol * sfc_level] = src;
  // ---- bottom -> top: adding recurrences (:1174-1186 / :1214-1226)
  for (int k = 0; k < nlay; ++k) {
    const int ilay = a.top_at_1 ? nlay - 1 - k : k;
    const size_t o = (size_t)ncol * ilay;
    const Float Rdif = wR[o], Tdif = wT[o];
    const Float denom = (Float)1 / ((Float)1 - Rdif * albedo);
    const Float src_new = wSu[o] + Tdif * denom * (src + albedo * wSd[o]);
    const Float alb_new = Rdif + Tdif * Tdif * albedo * denom;
    albedo = alb_new;
    src = src_new;
    const int lev_above = a.top_at_1 ? ilay : ilay + 1;
    fdn[(size_t)ncol * lev_above] = albedo;
    fup[(size_t)ncol * lev_above] = src;
    wDen[o] = denom;
  }
  // ---- top boundary, top -> bottom fluxes (:1188-1202 / :1228-1243); dn gets direct added (:603,:606)
  Float fd = a.has_dif_bc ? a.inc_flux_dif[cg] : (Float)0;  // :579-583
  {
    const size_t ol = (size_t)ncol * top_level;
    fup[ol] = fd * albedo + src;
    fdn[ol] = a.add_dir_to_dn ? fd + fdir[ol] : fd;
  }
  for (int k = 0; k < nlay; ++k) {
    const int ilay = a.top_at_1 ? k : nlay - 1 - k;
    const int lev_below = a.top_at_1 ? ilay + 1 : ilay;
    const size_t o = (size_t)ncol * ilay, ol = (size_t)ncol * lev_below;
    const Float alb = fdn[ol], sr = fup[ol];
    fd = (wT[o] * fd + wR[o] * sr + wSd[o]) * wDen[o];
    fup[ol] = fd * alb + sr;
    fdn[ol] = a.add_dir_to_dn ? fd + fdir[ol] : fd;
  }
}

// ---------------------------------------------------------------------------------------------
// SW two-stream, segmented (broadband output, nlay <= 64): the layout of lw_noscat_seg_kernel applied to
// sw_dif_and_source + adding (reference :503-609, :985-1127, :1135-1245).
// Block = S waves x 64 columns; wave s owns layers [8s, 8s+8) counted from the top; per g-point
//   (1) each wave evaluates the two-stream coefficients of its layers, the direct-beam attenuation
//       RELATIVE to the beam entering the segment, and the segment composite of the adding recurrence.
//       One adding step (albedo a, source c below -> above: a' = R + T^2 a/(1-Ra), c' = su + T(c + a sd)/(1-Ra))
//       is the projective map (a, c, 1) -> M (a, c, 1), M = [[T^2-R^2, 0, R], [T sd - su R, T, su], [-R, 0, 1]];
//       products keep the zero pattern (7 entries), and the source entries m10, m12 are linear in the beam
//       entering the segment, so they are formed with the relative beam and scaled after the exchange;
//   (2) barrier; every wave chains the composites from the surface up to its own lower edge, then re-sweeps
//       its layers from registers with the reference's own expressions (albedo, source, denom per level) and
//       forms the affine maps of the downward diffuse flux, fd' = alpha fd + beta, and their composite;
//   (3) barrier; chain from the top, final sweep, broadband accumulation in registers (partial slabs per
//       g-group, reduced deterministically by reduce_parts_kernel).
// Across segments the boundary values come from composites -- same mathematics, different rounding.
// ---------------------------------------------------------------------------------------------
struct Sw2SegArgs {
  int ncol, nlay, ngpt, S, g_per_block;
  bool top_at_1, has_dif_bc;
  const Float *tau, *ssa, *g, *mu0, *sfc_alb_dir, *sfc_alb_dif, *inc_flux_dir, *inc_flux_dif;
  Float *part_up, *part_dn, *part_dir;  // (ncol, nlev, ngroups)
  Float *spec_up, *spec_dn, *spec_dir;  // SPEC: the interface's spectral flux arrays (ncol, nlev, ngpt)
  const int* band_lims;                 // non-null (rte_hip_sw_solver_2stream_byband): grid.y = band, part_* are the by-band fluxes
  // WIN (columns of more than 96 layers, solved as an upper and a lower part): the kernel works on a WINDOW of nlay layers of
  // arrays whose g-point planes have plane_lay / plane_lev elements (pointers already offset to the window's first row)
  size_t plane_lay, plane_lev, plane_part;
  int beam_mode;        // 0: beam at the window's top = inc_flux_dir * mu0 (top of the column); 1: unit beam; 2: inc_flux_dir as it is
  bool sfc_src_given;   // the "surface" is the part below: source = beam * sfc_alb_dir without the sun test
  bool skip_first_level;  // the window's first level belongs to the part above
  Float *out_alb, *out_src;  // (ncol, ngpt): albedo and (relative) source at the window's top
  Float *out_fd, *out_dir;   // (ncol, ngpt): diffuse and direct downward flux at the window's bottom
};

// SPEC: spectral output (rte_sw with a ty_fluxes other than ty_fluxes_broadband, rte/frontend/mo_rte_sw.F90): every wave
// stores the three fluxes of the levels it owns per g-point instead of accumulating them

// G0: the asymmetry parameter is zero everywhere and its array does not exist (g == NULL in rte_sw_solver_2stream: clear-sky optical
// properties as rte_hip_gas_optics_sw_2str leaves them without a g array) -- nothing is read for it and the terms it multiplies are
// dropped as written (5 + 3 * 0, 1 - 0, 0.75 mu0 * 0): the same bits as with an array of zeros.
// (sw_seg_wave below restates this body per wave for segments of two lengths: changes to the arithmetic go to both)
template <int L, bool SPEC = false, bool WIN = false, bool G0 = false>
__global__ void __launch_bounds__(64 * 8) sw_2stream_seg_kernel(Sw2SegArgs a) {
#pragma clang fp contract(fast)  // VALU-bound: fuse a*b+c (the segment composites already differ from the reference's rounding)
  constexpr int SMAX = 8, NC1 = 8, NC2 = 2;
  extern __shared__ Float lds[];  // X1[NC1][SMAX][64] (P, m00, m02, m10, m11, m12, m20, m22), X2[NC2][SMAX][64] (A, B)
  Float* const X1 = lds;
  Float* const X2 = lds + NC1 * SMAX * 64;
  Float* const MU = X2 + NC2 * SMAX * 64;  // of this wave's layers: max(min_mu0, mu0) and its reciprocal, [SMAX][2][L][64]
  const int lane = threadIdx.x & 63;
  const int s = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int S = a.S, ncol = a.ncol, nlay = a.nlay;
  const int icol = blockIdx.x * 64 + lane;
  const bool active = icol < ncol;
  const int c = active ? icol : ncol - 1;  // clamp: inactive lanes compute on a valid column, never store
  const size_t ncl = WIN ? a.plane_lay : (size_t)ncol * nlay, nclv = WIN ? a.plane_lev : (size_t)ncol * (nlay + 1);
  const size_t nclp = WIN ? a.plane_part : nclv;  // one partial slab
  const int p0 = s * L;
  const int np = min(L, nlay - p0);  // layers in this segment (>= 1 by construction)
  const bool last = (s == S - 1);
  const int g_begin = a.band_lims ? a.band_lims[2 * blockIdx.y] - 1 : blockIdx.y * a.g_per_block;
  const int g_end = a.band_lims ? a.band_lims[2 * blockIdx.y + 1] : min(a.ngpt, g_begin + a.g_per_block);
  const Float min_k = (Float)1.e4 * (Float)RTE_EPS;
  const Float min_mu0 = sqrt((Float)RTE_EPS);
  auto layer_of = [&](int i) {  // array index of the segment's i-th layer from the top (clamped to a valid one)
    const int p = p0 + min(i, np - 1);
    return a.top_at_1 ? p : nlay - 1 - p;
  };

  // PREF: the next g-point's inputs are requested into the registers of the current ones as soon as pass (1) has consumed
  // them (3 L + 4 doubles in flight during passes (2) and (3)).  Worth 3 % where it fits (8 and 9 layers per wave, broadband:
  // 12.1-12.2 against 12.5-12.6 ms at 1e5 x 60 x 224); the kernel is bound by its arithmetic and the two waves of a SIMD cover
  // most of each other's wait for a g-point's first inputs.  Where it does not fit it costs far more than it gives: the
  // spectral-output variants spilled (17.7 -> 15.6 ms at 60 layers, 31.1 -> 17.9 ms at 72 without it), and 10 ... 12 layers
  // per wave (73 ... 96 layers) spilled 61-112 registers and ran at 244-320 us per layer -- without the prefetch they run at
  // the 100-108 us per layer of the others.
  constexpr bool PREF = L <= 9 && !SPEC && !WIN;
  constexpr bool DIRLDS = (L <= 9 || L >= 11) && !SPEC;  // the direct-flux accumulators in LDS (ds_add_f64 on the thread's own slots): 2L+2 registers
  // L == 9 (72 layers) is 15 registers over: the upward-flux accumulators go to LDS as well, and to make room there
  // only ONE value per layer is parked (the reciprocal is formed again per g-point, 6 instructions per layer)
  constexpr bool UPLDS = L == 9 && !SPEC;
  constexpr bool ONEMU = UPLDS || (L >= 11 && !SPEC);  // one parked value per layer (L >= 11: room for the direct-flux slots)
  constexpr int NMU = ONEMU ? 1 : 2;
  Float* const dirs = MU + (size_t)SMAX * NMU * L * 64 + (size_t)s * (L + 1) * 64 + lane;  // acc_dir slot i at dirs[i * 64]
  Float* const ups = MU + (size_t)SMAX * NMU * L * 64 + (size_t)SMAX * (L + 1) * 64 + (size_t)s * (L + 1) * 64 + lane;
  // per-layer cosine of the solar zenith angle and what depends on it alone: independent of the g-point, parked in
  // LDS (lane-private slots) -- the clamped value (:1046) and its reciprocal (tau / mu0 becomes a product), the
  // reciprocal (UPLDS: the clamped value) carrying "mu0 > 0" (:1122) in its sign
  Float* const mu0s = MU + (size_t)s * NMU * L * 64 + lane;  // element i at mu0s[i * 64]
  Float* const mu0i = mu0s + L * 64;
#pragma unroll
  for (int i = 0; i < L; ++i) {
    const Float m = a.mu0[c + (size_t)ncol * layer_of(i)];
    const Float ms = fmax(min_mu0, m);
    if constexpr (ONEMU) {
      mu0s[i * 64] = m > (Float)0 ? ms : -ms;
    } else {
      mu0s[i * 64] = ms;
      mu0i[i * 64] = m > (Float)0 ? rte::rcp_nr(ms) : -rte::rcp_nr(ms);
    }
  }
  const Float mu0_top = a.mu0[c + (size_t)ncol * (a.top_at_1 ? 0 : nlay - 1)];
  const Float mu0_sfc = a.mu0[c + (size_t)ncol * (a.top_at_1 ? nlay - 1 : 0)];

  Float acc_up[UPLDS || SPEC ? 1 : L + 1], acc_dn[SPEC ? 1 : L + 1], acc_dir[DIRLDS || SPEC ? 1 : L + 1];
  // SPEC: byte offset of the wave's first level row inside one g-point plane and the step to the next level (two
  // registers instead of L + 1: the stores are the only users)
  unsigned olev0 = ((unsigned)c + (unsigned)ncol * (unsigned)(a.top_at_1 ? p0 : nlay - p0)) * (unsigned)sizeof(Float);
  const unsigned dlev = (a.top_at_1 ? (unsigned)ncol : 0u - (unsigned)ncol) * (unsigned)sizeof(Float);
  int gcur = g_begin;               // SPEC: the g-point being processed
  if constexpr (SPEC) {
    asm volatile("" : "+v"(olev0));
  } else {
#pragma unroll
    for (int i = 0; i <= L; ++i) {
      acc_dn[i] = 0;
      if constexpr (UPLDS) ups[i * 64] = 0; else acc_up[i] = 0;
      if constexpr (DIRLDS) dirs[i * 64] = 0; else acc_dir[i] = 0;
    }
  }
  auto spec_store = [&](Float* __restrict__ arr, int i, Float v) {
    if (active && (i < np || (last && i == np)) && !(WIN && a.skip_first_level && s == 0 && i == 0))
      rte::store_stream(reinterpret_cast<Float*>(reinterpret_cast<char*>(arr + nclv * gcur) + (olev0 + (unsigned)i * dlev)), v);
  };
  auto add_dir = [&](int i, Float v) {
    if constexpr (SPEC) spec_store(a.spec_dir, i, v);
    else if constexpr (DIRLDS) atomicAdd(&dirs[i * 64], v);
    else acc_dir[i] += v;
  };
  auto add_up = [&](int i, Float v) {
    if constexpr (SPEC) spec_store(a.spec_up, i, v);
    else if constexpr (UPLDS) atomicAdd(&ups[i * 64], v);
    else acc_up[i] += v;
  };
  auto add_dn = [&](int i, Float v) {
    if constexpr (SPEC) spec_store(a.spec_dn, i, v);
    else acc_dn[i] += v;
  };

  struct In { Float tau[L], ssa[L], g[L], inc_dir, alb_dir, alb_dif, inc_dif; };
  // loads as (wave-uniform plane base, advanced per g-point) + (32-bit byte offset of the lane's row): the saddr form
  // of global_load, no 64-bit address arithmetic per load (the host guarantees 8 * ncol * nlay < 2^32)
  unsigned orow[L];
#pragma unroll
  for (int i = 0; i < L; ++i) {
    orow[i] = ((unsigned)c + (unsigned)ncol * (unsigned)layer_of(i)) * (unsigned)sizeof(Float);
    asm volatile("" : "+v"(orow[i]));
  }
  unsigned ocg = (unsigned)c * (unsigned)sizeof(Float);
  asm volatile("" : "+v"(ocg));
  auto at = [](const Float* plane, unsigned off) {  // plane is wave-uniform
    return *reinterpret_cast<const Float*>(reinterpret_cast<const char*>(plane) + off);
  };
  auto load = [&](In& x, int igpt_) {
    const int igpt = min(igpt_, g_end - 1);
    const Float *ptau = a.tau + ncl * igpt, *pssa = a.ssa + ncl * igpt, *pg = G0 ? nullptr : a.g + ncl * igpt;
    // the row offsets are made opaque IN PLACE once per g-point (their 64-bit extension cannot be hoisted out of the loop into
    // registers of its own); as a by-value copy per load the compiler moved every offset into a scratch register first: 27
    // v_mov per g-point and wave
#pragma unroll
    for (int i = 0; i < L; ++i) {
      asm volatile("" : "+v"(orow[i]));
      x.tau[i] = at(ptau, orow[i]); x.ssa[i] = at(pssa, orow[i]);
      if constexpr (!G0) x.g[i] = at(pg, orow[i]);
    }
    asm volatile("" : "+v"(ocg));
    const size_t cg = (size_t)ncol * igpt;
    x.inc_dir = at(a.inc_flux_dir + cg, ocg); x.alb_dir = at(a.sfc_alb_dir + cg, ocg); x.alb_dif = at(a.sfc_alb_dif + cg, ocg);
    x.inc_dif = a.has_dif_bc ? at(a.inc_flux_dif + cg, ocg) : (Float)0;  // :579-583
  };

  auto process = [&](In& x, int igpt_next) {
#pragma clang fp contract(fast)
    Float R[L], T[L], su[L], sd[L];  // Rdif, Tdif, source up / down (relative to the beam entering the segment)
    Float Tn[L];                      // direct-beam transmission of the layer
    Float P = 1;                      // direct-beam transmission of the layers above layer i in this segment
    // ---- (1) two-stream coefficients and relative sources, top -> bottom (:1015-1114)
#pragma unroll
    for (int i = 0; i < L; ++i) {
      {
        // a slot past the segment's last layer (partial last segment only) is made NEUTRAL by tau = 0: e1 = e2 =
        // Tnoscat = 1, so Rdif = 0, the clamps give Rdir = Tdir = 0 exactly, and Tdif (1 up to rounding) is set to 1
        // -- the identity in every recurrence, without a branch around the layer
        const Float tau_s = i < np ? x.tau[i] : (Float)0, w0_s = x.ssa[i], g_s = G0 ? (Float)0 : x.g[i];
        const Float gamma1 = G0 ? ((Float)8 - w0_s * (Float)5) * (Float).25 : ((Float)8 - w0_s * ((Float)5 + (Float)3 * g_s)) * (Float).25;
        const Float gamma2 = G0 ? (Float)3 * w0_s * (Float).25 : (Float)3 * (w0_s * ((Float)1 - g_s)) * (Float).25;
        // (the reference's own operation order: gamma1 - gamma2 cancels for conservative scattering, and a differently rounded
        //  pair moves the fluxes of cloudy columns by 1e-8 relative to the reference's -- measured, all-sky at 1e5 columns)
        const Float kk = rte::sqrt_pos(fmax((gamma1 - gamma2) * (gamma1 + gamma2), min_k));
        const Float e1 = rte::exp_nonpos(-tau_s * kk);
        const Float e2 = e1 * e1;
        // RT = 1 / x (:1031) and w0 RT / om (:1054) from ONE reciprocal, of x om
        const Float xden = kk * ((Float)1 + e2) + gamma1 * ((Float)1 - e2);
        const Float mu0_s = ONEMU ? fabs(mu0s[i * 64]) : mu0s[i * 64];
        const Float k_mu = kk * mu0_s;
        const Float om = (Float)1 - k_mu * k_mu;
        const Float om_s = fabs(om) >= (Float)RTE_EPS ? om : (Float)RTE_EPS;
        const Float inv = rte::rcp_nr(xden * om_s);
        Float RT = om_s * inv;
        R[i] = RT * gamma2 * ((Float)1 - e2);
        T[i] = i < np ? RT * (Float)2 * kk * e1 : (Float)1;
        RT = w0_s * inv;
        const Float gamma3 = G0 ? (Float).5 : (Float).5 - ((Float).75 * mu0_s) * g_s;  // (2 - 3 mu0 g) / 4
        // alpha1 = gamma1 gamma4 + gamma2 gamma3, alpha2 = gamma1 gamma3 + gamma2 gamma4 with gamma4 = 1 - gamma3 (:1078-1081)
        const Float dgam = gamma1 - gamma2;
        const Float alpha1 = gamma1 - gamma3 * dgam;
        const Float alpha2 = gamma2 + gamma3 * dgam;
        const Float k_gamma3 = kk * gamma3, k_gamma4 = kk - k_gamma3;
        Float imu;
        if constexpr (ONEMU) { const Float r_ = rte::rcp_nr(mu0_s); imu = mu0s[i * 64] > (Float)0 ? r_ : -r_; }
        else imu = mu0i[i * 64];
        const Float Tnoscat = rte::exp_nonpos(-tau_s * fabs(imu));
        // Rdir, Tdir (:1097-1105, Meador & Weaver eq. 14-15 as the reference rearranges them), multiplied out in the
        // products with (1 -+ k mu0) and collected by 1 - e2 and 1 + e2, which the diffuse part has formed already:
        //   Rdir / RT  = (1 - k mu0)(alpha2 + k gamma3) - (1 + k mu0)(alpha2 - k gamma3) e2 - 2 (k gamma3 - alpha2 k mu0) e1 Tn
        //              = u (1 - e2) + v (1 + e2 - 2 e1 Tn),        u = alpha2 - k mu0 k gamma3,  v = k gamma3 - k mu0 alpha2
        //   Tdir / -RT = (1 + k mu0)(alpha1 + k gamma4) Tn - (1 - k mu0)(alpha1 - k gamma4) e2 Tn - 2 (k gamma4 + alpha1 k mu0) e1
        //              = Tn (p (1 - e2) + q (1 + e2)) - 2 e1 q,   p = alpha1 + k mu0 k gamma4,  q = k gamma4 + k mu0 alpha1
        // -- 15 fp64 operations for the pair instead of 32 (the kernel is bound by its fp64 issue)
        const Float om2 = (Float)1 - e2, op2 = (Float)1 + e2;
        const Float u_ = alpha2 - k_mu * k_gamma3, v_ = k_gamma3 - k_mu * alpha2;
        const Float p_ = alpha1 + k_mu * k_gamma4, q_ = k_gamma4 + k_mu * alpha1;
        Float Rdir = RT * (u_ * om2 + v_ * (op2 - (Float)2 * (e1 * Tnoscat)));
        Float Tdir = -RT * (Tnoscat * (p_ * om2 + q_ * op2) - (Float)2 * (e1 * q_));
        Rdir = fmax((Float)0, fmin(Rdir, ((Float)1 - Tnoscat)));
        Tdir = fmax((Float)0, fmin(Tdir, ((Float)1 - Tnoscat - Rdir)));
        const bool sun = imu > (Float)0;  // :1122-1125
        const Float Ps = sun ? P : (Float)0;  // (one select instead of two: Rdir, Tdir are finite)
        su[i] = Rdir * Ps;
        sd[i] = Tdir * Ps;
        Tn[i] = Tnoscat;
        P = Tnoscat * P;
      }
    }
    // the layer inputs are dead: request the next g-point's into the same registers (in flight during (2), (3);
    // spreading the requests over pass (1), layer by layer, was tried: the register allocator spills)
    const Float inc_dir = x.inc_dir, alb_dir = x.alb_dir, alb_dif = x.alb_dif, inc_dif = x.inc_dif;
    if constexpr (PREF) load(x, igpt_next);
    // ---- segment composite of the adding recurrence (bottom -> top product of the per-layer maps)
    // the sources of the adding chain are carried RELATIVE to the beam entering the segment below (sigma = src / beam): with
    // beam(q + 1) = beam(q) P(q) the step reads  sigma' = (c10 alb + (c11 P) sigma + c12) w  -- the wave publishes m11 P and the
    // chain needs no beam per segment (was: a prefix product over all segments in every wave, two more products per step).
    // (Skipping the top segment's composite, which nobody reads, was measured: no gain -- that wave waits at the barrier anyway.)
    X1[(0 * SMAX + s) * 64 + lane] = P;
    // (the product starts FROM the lowest layer's map instead of multiplying it into the identity: the compiler may not drop
    //  x * 0 and x + 0, and kept all 17 operations of that step)
    Float m00 = T[L - 1] * T[L - 1] - R[L - 1] * R[L - 1], m02 = R[L - 1], m10 = T[L - 1] * sd[L - 1] - su[L - 1] * R[L - 1],
          m11 = T[L - 1], m12 = su[L - 1], m20 = -R[L - 1], m22 = 1;
#pragma unroll
    for (int i = L - 2; i >= 0; --i) {
      const Float q00 = T[i] * T[i] - R[i] * R[i], q02 = R[i], q10 = T[i] * sd[i] - su[i] * R[i], q11 = T[i], q12 = su[i],
                  q20 = -R[i];  // q22 = 1, q01 = q21 = 0
      const Float n00 = q00 * m00 + q02 * m20, n02 = q00 * m02 + q02 * m22;
      const Float n10 = q10 * m00 + q11 * m10 + q12 * m20, n11 = q11 * m11, n12 = q10 * m02 + q11 * m12 + q12 * m22;
      const Float n20 = q20 * m00 + m20, n22 = q20 * m02 + m22;
      m00 = n00; m02 = n02; m10 = n10; m11 = n11; m12 = n12; m20 = n20; m22 = n22;
    }
    X1[(1 * SMAX + s) * 64 + lane] = m00;
    X1[(2 * SMAX + s) * 64 + lane] = m02;
    X1[(3 * SMAX + s) * 64 + lane] = m10;
    X1[(4 * SMAX + s) * 64 + lane] = m11 * P;
    X1[(5 * SMAX + s) * 64 + lane] = m12;
    X1[(6 * SMAX + s) * 64 + lane] = m20;
    X1[(7 * SMAX + s) * 64 + lane] = m22;
    __syncthreads();
    // comparisons with the wave's segment number stay scalar instructions inside the loop: hoisted out of it they became 64-bit
    // masks in spilled scalar registers (two v_readlane in front of every branch)
    int s_u = s, S_u = S;
    asm volatile("" : "+s"(s_u), "+s"(S_u));
    // ---- (2) beam entering this segment (all transmissions are requested at once: one wait, not one per segment)
    const Float dir_toa = !WIN || a.beam_mode == 0 ? inc_dir * mu0_top : (a.beam_mode == 1 ? (Float)1 : inc_dir);  // :575
    Float pq[SMAX - 1];
#pragma unroll
    for (int q = 0; q < SMAX - 1; ++q) pq[q] = X1[(0 * SMAX + q) * 64 + lane];
    const Float P_own = X1[(0 * SMAX + s) * 64 + lane];
    Float dir_in = dir_toa;
#pragma unroll
    for (int q = 0; q < SMAX - 1; ++q)
      if (q < s_u) dir_in = dir_in * pq[q];
    // ---- adding chain from the surface up to this segment's lower edge
    Float alb = alb_dif;                                                  // :1121
    Float sig = (mu0_sfc > (Float)0 || (WIN && a.sfc_src_given)) ? alb_dir : (Float)0;     // :1120, per unit of beam at the surface
#pragma unroll
    for (int q = SMAX - 1; q > 0; --q) {
      if (q < S_u && q > s_u) {  // wave-uniform
        const Float c00 = X1[(1 * SMAX + q) * 64 + lane], c02 = X1[(2 * SMAX + q) * 64 + lane];
        const Float c10 = X1[(3 * SMAX + q) * 64 + lane], c11 = X1[(4 * SMAX + q) * 64 + lane];
        const Float c12 = X1[(5 * SMAX + q) * 64 + lane];
        const Float c20 = X1[(6 * SMAX + q) * 64 + lane], c22 = X1[(7 * SMAX + q) * 64 + lane];
        const Float w = rte::rcp_nr(c20 * alb + c22);
        const Float a_new = (c00 * alb + c02) * w;
        sig = (c10 * alb + c11 * sig + c12) * w;
        alb = a_new;
      }
    }
    Float src = sig * (dir_in * P_own);  // the beam leaving this segment
    asm volatile("" : "+v"(src), "+v"(alb));
    // ---- own layers, bottom -> top, the reference's expressions (:1174-1186 / :1214-1226); the beam at the
    // levels is accumulated on the way (direct flux, and the direct part of flux_dn, :603,:606)
    Float al[L + 1], sr[L + 1];  // albedo and source at the levels of the segment (slot i = top of layer i)
    al[L] = alb; sr[L] = src;
    Float fa[L], fb[L];           // downward diffuse flux below layer i = fa * (flux above) + fb
#pragma unroll
    for (int i = L - 1; i >= 0; --i) {
      const Float sui = su[i] * dir_in, sdi = sd[i] * dir_in;
      const Float denom = rte::rcp_nr((Float)1 - R[i] * alb);
      const Float src_new = sui + T[i] * denom * (src + alb * sdi);
      const Float alb_new = R[i] + T[i] * T[i] * alb * denom;
      fa[i] = T[i] * denom;
      fb[i] = (R[i] * src + sdi) * denom;
      alb = alb_new; src = src_new;
      al[i] = alb; sr[i] = src;
    }
    if constexpr (WIN) {
      if (a.out_alb && s == 0 && active) {  // (wave 0: dir_in is the beam entering the window)
        a.out_alb[icol + (size_t)ncol * gcur] = al[0];
        a.out_src[icol + (size_t)ncol * gcur] = sr[0];
      }
    }
    Float A = 1, B = 0;
#pragma unroll
    for (int i = 0; i < L; ++i) { B = fa[i] * B + fb[i]; A = fa[i] * A; }
    X2[(0 * SMAX + s) * 64 + lane] = A;
    X2[(1 * SMAX + s) * 64 + lane] = B;
    __syncthreads();
    // ---- (3) diffuse flux entering the segment from above, final sweep (:1188-1202 / :1228-1243)
    Float fd = inc_dif;
    {
      Float qa[SMAX - 1], qb[SMAX - 1];  // requested at once, whatever the segment
#pragma unroll
      for (int q = 0; q < SMAX - 1; ++q) { qa[q] = X2[(0 * SMAX + q) * 64 + lane]; qb[q] = X2[(1 * SMAX + q) * 64 + lane]; }
#pragma unroll
      for (int q = 0; q < SMAX - 1; ++q)
        if (q < s_u) fd = qa[q] * fd + qb[q];
    }
    Float dirl = dir_in;  // beam at the segment's levels
#pragma unroll
    for (int i = 0; i < L; ++i) {
      add_up(i, fd * al[i] + sr[i]);
      add_dn(i, fd + dirl);
      add_dir(i, dirl);
      fd = fa[i] * fd + fb[i];
      dirl = Tn[i] * dirl;
    }
    add_up(L, fd * al[L] + sr[L]);
    add_dn(L, fd + dirl);
    add_dir(L, dirl);
    if constexpr (WIN) {
      if (a.out_fd && last && active) {  // (neutral slots pass the values on: these are the fluxes at the window's bottom)
        a.out_fd[icol + (size_t)ncol * gcur] = fd;
        a.out_dir[icol + (size_t)ncol * gcur] = dirl;
      }
    }
  };

  In cur;
  if constexpr (PREF) load(cur, g_begin);
  for (int igpt = g_begin; igpt < g_end; ++igpt) {
    gcur = igpt;
    if constexpr (!PREF) load(cur, igpt);
    process(cur, igpt + 1);
  }
  if constexpr (SPEC) return;
  if (active) {
    const size_t base = icol + nclp * blockIdx.y;
#pragma unroll
    for (int i = 0; i <= L; ++i) {
      if ((i < np || (last && i == np)) && !(WIN && a.skip_first_level && s == 0 && i == 0)) {
        const int p = p0 + i;  // level position from the top
        const int ilev = a.top_at_1 ? p : nlay - p;
        if constexpr (UPLDS) a.part_up[base + (size_t)ncol * ilev] = ups[i * 64];
        else a.part_up[base + (size_t)ncol * ilev] = acc_up[i];
        a.part_dn[base + (size_t)ncol * ilev] = acc_dn[i];
        if constexpr (DIRLDS) a.part_dir[base + (size_t)ncol * ilev] = dirs[i * 64];
        else a.part_dir[base + (size_t)ncol * ilev] = acc_dir[i];
      }
    }
  }
}


// ---------------------------------------------------------------------------------------------
// sw_2stream_seg_kernel with segments of TWO lengths (broadband output, whole column, 57 ... 60 layers): waves 0-3 own LA
// layers each, waves 4-7 LB -- 4 x 7 + 4 x 8 = 60.  With 8 layers per wave the eighth wave owns four layers and computes four
// neutral slots: 16 slots per SIMD (waves s and s + 4 share one) for 15 layers.  Measured at 1e5 x 60 x 224, two rounds in one
// process each: 8 + 8 10.66-10.70 ms, 9 + 6 10.57-10.58, 8 + 7 10.26-10.29, 7 + 8 10.23-10.26, 6 + 9 10.20-10.24 (256 registers).
// (The idea it started from -- the wave a SIMD serves first, s < 4, should own MORE layers so that both reach the barrier
//  together instead of the younger one finishing alone at a single wave's issue rate -- is refuted by 9 + 6: what pays is the
//  sixteenth slot not computed, and the split that gives the later wave the longer segment is, if anything, the better one.)
// sw_seg_wave is the body of sw_2stream_seg_kernel for ONE wave with its segment's length as the template parameter (same
// expressions in the same order: see the comments there -- the two must be kept in step; the broadband whole-column case only,
// always with the input prefetch and the direct-flux accumulators in LDS, i.e. what that kernel does for L <= 9); a wave's private LDS slots (64 lanes each) are, from `priv`:
// NMU x L values of mu0, L + 1 direct-flux accumulators, and with UPLDS L + 1 upward-flux accumulators.
// ---------------------------------------------------------------------------------------------
template <int L, bool G0, bool UPLDS, bool ONEMU>
__device__ __forceinline__ void sw_seg_wave(const Sw2SegArgs& a, Float* __restrict__ const X1, Float* __restrict__ const X2,
                                            Float* __restrict__ const priv, const int s, const int p0, const int np, const bool last) {
#pragma clang fp contract(fast)
  constexpr int SMAX = 8, NMU = ONEMU ? 1 : 2;
  const int lane = threadIdx.x & 63;
  const int S = a.S, ncol = a.ncol, nlay = a.nlay;
  const int icol = blockIdx.x * 64 + lane;
  const bool active = icol < ncol;
  const int c = active ? icol : ncol - 1;
  const size_t ncl = (size_t)ncol * nlay, nclv = (size_t)ncol * (nlay + 1);
  const int g_begin = blockIdx.y * a.g_per_block;
  const int g_end = min(a.ngpt, g_begin + a.g_per_block);
  const Float min_k = (Float)1.e4 * (Float)RTE_EPS;
  const Float min_mu0 = sqrt((Float)RTE_EPS);
  auto layer_of = [&](int i) {
    const int p = p0 + min(i, np - 1);
    return a.top_at_1 ? p : nlay - 1 - p;
  };
  Float* const mu0s = priv + lane;
  Float* const mu0i = mu0s + L * 64;
  Float* const dirs = priv + NMU * L * 64 + lane;
  Float* const ups = dirs + (L + 1) * 64;
#pragma unroll
  for (int i = 0; i < L; ++i) {
    const Float m = a.mu0[c + (size_t)ncol * layer_of(i)];
    const Float ms = fmax(min_mu0, m);
    if constexpr (ONEMU) {
      mu0s[i * 64] = m > (Float)0 ? ms : -ms;
    } else {
      mu0s[i * 64] = ms;
      mu0i[i * 64] = m > (Float)0 ? rte::rcp_nr(ms) : -rte::rcp_nr(ms);
    }
  }
  const Float mu0_top = a.mu0[c + (size_t)ncol * (a.top_at_1 ? 0 : nlay - 1)];
  const Float mu0_sfc = a.mu0[c + (size_t)ncol * (a.top_at_1 ? nlay - 1 : 0)];
  Float acc_up[UPLDS ? 1 : L + 1], acc_dn[L + 1];
#pragma unroll
  for (int i = 0; i <= L; ++i) {
    acc_dn[i] = 0;
    if constexpr (UPLDS) ups[i * 64] = 0; else acc_up[i] = 0;
    dirs[i * 64] = 0;
  }
  auto add_dir = [&](int i, Float v) { atomicAdd(&dirs[i * 64], v); };
  auto add_up = [&](int i, Float v) {
    if constexpr (UPLDS) atomicAdd(&ups[i * 64], v);
    else acc_up[i] += v;
  };
  auto add_dn = [&](int i, Float v) { acc_dn[i] += v; };

  struct In { Float tau[L], ssa[L], g[L], inc_dir, alb_dir, alb_dif, inc_dif; };
  unsigned orow[L];
#pragma unroll
  for (int i = 0; i < L; ++i) {
    orow[i] = ((unsigned)c + (unsigned)ncol * (unsigned)layer_of(i)) * (unsigned)sizeof(Float);
    asm volatile("" : "+v"(orow[i]));
  }
  unsigned ocg = (unsigned)c * (unsigned)sizeof(Float);
  asm volatile("" : "+v"(ocg));
  auto at = [](const Float* plane, unsigned off) {
    return *reinterpret_cast<const Float*>(reinterpret_cast<const char*>(plane) + off);
  };
  auto load = [&](In& x, int igpt_) {
    const int igpt = min(igpt_, g_end - 1);
    const Float *ptau = a.tau + ncl * igpt, *pssa = a.ssa + ncl * igpt, *pg = G0 ? nullptr : a.g + ncl * igpt;
#pragma unroll
    for (int i = 0; i < L; ++i) {
      asm volatile("" : "+v"(orow[i]));
      x.tau[i] = at(ptau, orow[i]); x.ssa[i] = at(pssa, orow[i]);
      if constexpr (!G0) x.g[i] = at(pg, orow[i]);
    }
    asm volatile("" : "+v"(ocg));
    const size_t cg = (size_t)ncol * igpt;
    x.inc_dir = at(a.inc_flux_dir + cg, ocg); x.alb_dir = at(a.sfc_alb_dir + cg, ocg); x.alb_dif = at(a.sfc_alb_dif + cg, ocg);
    x.inc_dif = a.has_dif_bc ? at(a.inc_flux_dif + cg, ocg) : (Float)0;
  };

  auto process = [&](In& x, int igpt_next) {
#pragma clang fp contract(fast)
    Float R[L], T[L], su[L], sd[L];
    Float Tn[L];
    Float P = 1;
#pragma unroll
    for (int i = 0; i < L; ++i) {
      {
        const Float tau_s = i < np ? x.tau[i] : (Float)0, w0_s = x.ssa[i], g_s = G0 ? (Float)0 : x.g[i];
        const Float gamma1 = G0 ? ((Float)8 - w0_s * (Float)5) * (Float).25 : ((Float)8 - w0_s * ((Float)5 + (Float)3 * g_s)) * (Float).25;
        const Float gamma2 = G0 ? (Float)3 * w0_s * (Float).25 : (Float)3 * (w0_s * ((Float)1 - g_s)) * (Float).25;
        const Float kk = rte::sqrt_pos(fmax((gamma1 - gamma2) * (gamma1 + gamma2), min_k));
        const Float e1 = rte::exp_nonpos(-tau_s * kk);
        const Float e2 = e1 * e1;
        const Float xden = kk * ((Float)1 + e2) + gamma1 * ((Float)1 - e2);
        const Float mu0_s = ONEMU ? fabs(mu0s[i * 64]) : mu0s[i * 64];
        const Float k_mu = kk * mu0_s;
        const Float om = (Float)1 - k_mu * k_mu;
        const Float om_s = fabs(om) >= (Float)RTE_EPS ? om : (Float)RTE_EPS;
        const Float inv = rte::rcp_nr(xden * om_s);
        Float RT = om_s * inv;
        R[i] = RT * gamma2 * ((Float)1 - e2);
        T[i] = i < np ? RT * (Float)2 * kk * e1 : (Float)1;
        RT = w0_s * inv;
        const Float gamma3 = G0 ? (Float).5 : (Float).5 - ((Float).75 * mu0_s) * g_s;
        const Float dgam = gamma1 - gamma2;
        const Float alpha1 = gamma1 - gamma3 * dgam;
        const Float alpha2 = gamma2 + gamma3 * dgam;
        const Float k_gamma3 = kk * gamma3, k_gamma4 = kk - k_gamma3;
        Float imu;
        if constexpr (ONEMU) { const Float r_ = rte::rcp_nr(mu0_s); imu = mu0s[i * 64] > (Float)0 ? r_ : -r_; }
        else imu = mu0i[i * 64];
        const Float Tnoscat = rte::exp_nonpos(-tau_s * fabs(imu));
        const Float om2 = (Float)1 - e2, op2 = (Float)1 + e2;
        const Float u_ = alpha2 - k_mu * k_gamma3, v_ = k_gamma3 - k_mu * alpha2;
        const Float p_ = alpha1 + k_mu * k_gamma4, q_ = k_gamma4 + k_mu * alpha1;
        Float Rdir = RT * (u_ * om2 + v_ * (op2 - (Float)2 * (e1 * Tnoscat)));
        Float Tdir = -RT * (Tnoscat * (p_ * om2 + q_ * op2) - (Float)2 * (e1 * q_));
        Rdir = fmax((Float)0, fmin(Rdir, ((Float)1 - Tnoscat)));
        Tdir = fmax((Float)0, fmin(Tdir, ((Float)1 - Tnoscat - Rdir)));
        const bool sun = imu > (Float)0;
        const Float Ps = sun ? P : (Float)0;
        su[i] = Rdir * Ps;
        sd[i] = Tdir * Ps;
        Tn[i] = Tnoscat;
        P = Tnoscat * P;
      }
    }
    const Float inc_dir = x.inc_dir, alb_dir = x.alb_dir, alb_dif = x.alb_dif, inc_dif = x.inc_dif;
    load(x, igpt_next);
    X1[(0 * SMAX + s) * 64 + lane] = P;
    Float m00 = T[L - 1] * T[L - 1] - R[L - 1] * R[L - 1], m02 = R[L - 1], m10 = T[L - 1] * sd[L - 1] - su[L - 1] * R[L - 1],
          m11 = T[L - 1], m12 = su[L - 1], m20 = -R[L - 1], m22 = 1;
#pragma unroll
    for (int i = L - 2; i >= 0; --i) {
      const Float q00 = T[i] * T[i] - R[i] * R[i], q02 = R[i], q10 = T[i] * sd[i] - su[i] * R[i], q11 = T[i], q12 = su[i],
                  q20 = -R[i];
      const Float n00 = q00 * m00 + q02 * m20, n02 = q00 * m02 + q02 * m22;
      const Float n10 = q10 * m00 + q11 * m10 + q12 * m20, n11 = q11 * m11, n12 = q10 * m02 + q11 * m12 + q12 * m22;
      const Float n20 = q20 * m00 + m20, n22 = q20 * m02 + m22;
      m00 = n00; m02 = n02; m10 = n10; m11 = n11; m12 = n12; m20 = n20; m22 = n22;
    }
    X1[(1 * SMAX + s) * 64 + lane] = m00;
    X1[(2 * SMAX + s) * 64 + lane] = m02;
    X1[(3 * SMAX + s) * 64 + lane] = m10;
    X1[(4 * SMAX + s) * 64 + lane] = m11 * P;
    X1[(5 * SMAX + s) * 64 + lane] = m12;
    X1[(6 * SMAX + s) * 64 + lane] = m20;
    X1[(7 * SMAX + s) * 64 + lane] = m22;
    __syncthreads();
    int s_u = s, S_u = S;
    asm volatile("" : "+s"(s_u), "+s"(S_u));
    const Float dir_toa = inc_dir * mu0_top;
    Float pq[SMAX - 1];
#pragma unroll
    for (int q = 0; q < SMAX - 1; ++q) pq[q] = X1[(0 * SMAX + q) * 64 + lane];
    const Float P_own = X1[(0 * SMAX + s) * 64 + lane];
    Float dir_in = dir_toa;
#pragma unroll
    for (int q = 0; q < SMAX - 1; ++q)
      if (q < s_u) dir_in = dir_in * pq[q];
    Float alb = alb_dif;
    Float sig = mu0_sfc > (Float)0 ? alb_dir : (Float)0;
#pragma unroll
    for (int q = SMAX - 1; q > 0; --q) {
      if (q < S_u && q > s_u) {
        const Float c00 = X1[(1 * SMAX + q) * 64 + lane], c02 = X1[(2 * SMAX + q) * 64 + lane];
        const Float c10 = X1[(3 * SMAX + q) * 64 + lane], c11 = X1[(4 * SMAX + q) * 64 + lane];
        const Float c12 = X1[(5 * SMAX + q) * 64 + lane];
        const Float c20 = X1[(6 * SMAX + q) * 64 + lane], c22 = X1[(7 * SMAX + q) * 64 + lane];
        const Float w = rte::rcp_nr(c20 * alb + c22);
        const Float a_new = (c00 * alb + c02) * w;
        sig = (c10 * alb + c11 * sig + c12) * w;
        alb = a_new;
      }
    }
    Float src = sig * (dir_in * P_own);
    asm volatile("" : "+v"(src), "+v"(alb));
    Float al[L + 1], sr[L + 1];
    al[L] = alb; sr[L] = src;
    Float fa[L], fb[L];
#pragma unroll
    for (int i = L - 1; i >= 0; --i) {
      const Float sui = su[i] * dir_in, sdi = sd[i] * dir_in;
      const Float denom = rte::rcp_nr((Float)1 - R[i] * alb);
      const Float src_new = sui + T[i] * denom * (src + alb * sdi);
      const Float alb_new = R[i] + T[i] * T[i] * alb * denom;
      fa[i] = T[i] * denom;
      fb[i] = (R[i] * src + sdi) * denom;
      alb = alb_new; src = src_new;
      al[i] = alb; sr[i] = src;
    }
    Float A = 1, B = 0;
#pragma unroll
    for (int i = 0; i < L; ++i) { B = fa[i] * B + fb[i]; A = fa[i] * A; }
    X2[(0 * SMAX + s) * 64 + lane] = A;
    X2[(1 * SMAX + s) * 64 + lane] = B;
    __syncthreads();
    Float fd = inc_dif;
    {
      Float qa[SMAX - 1], qb[SMAX - 1];
#pragma unroll
      for (int q = 0; q < SMAX - 1; ++q) { qa[q] = X2[(0 * SMAX + q) * 64 + lane]; qb[q] = X2[(1 * SMAX + q) * 64 + lane]; }
#pragma unroll
      for (int q = 0; q < SMAX - 1; ++q)
        if (q < s_u) fd = qa[q] * fd + qb[q];
    }
    Float dirl = dir_in;
#pragma unroll
    for (int i = 0; i < L; ++i) {
      add_up(i, fd * al[i] + sr[i]);
      add_dn(i, fd + dirl);
      add_dir(i, dirl);
      fd = fa[i] * fd + fb[i];
      dirl = Tn[i] * dirl;
    }
    add_up(L, fd * al[L] + sr[L]);
    add_dn(L, fd + dirl);
    add_dir(L, dirl);
  };

  In cur;
  load(cur, g_begin);
  for (int igpt = g_begin; igpt < g_end; ++igpt) process(cur, igpt + 1);
  if (active) {
    const size_t base = icol + nclv * blockIdx.y;
#pragma unroll
    for (int i = 0; i <= L; ++i) {
      if (i < np || (last && i == np)) {
        const int p = p0 + i;
        const int ilev = a.top_at_1 ? p : nlay - p;
        if constexpr (UPLDS) a.part_up[base + (size_t)ncol * ilev] = ups[i * 64];
        else a.part_up[base + (size_t)ncol * ilev] = acc_up[i];
        a.part_dn[base + (size_t)ncol * ilev] = acc_dn[i];
        a.part_dir[base + (size_t)ncol * ilev] = dirs[i * 64];
      }
    }
  }
}

// LDS: X1[8][8][64], X2[2][8][64], then the waves' private slots: 2 L values of mu0 (clamped, reciprocal), L + 1 direct-flux
// accumulators and, from 9 layers on, L + 1 upward-flux accumulators
constexpr int sw_mixed_slots(int L) { return 2 * L + (L + 1) + (L >= 9 ? L + 1 : 0); }
template <int LA, int LB>
constexpr size_t sw_mixed_lds_bytes() { return sizeof(Float) * 64 * (8 * 8 + 2 * 8 + 4 * sw_mixed_slots(LA) + 4 * sw_mixed_slots(LB)); }

template <int LA, int LB, bool G0>
__global__ void __launch_bounds__(64 * 8) sw_2stream_seg_mixed_kernel(Sw2SegArgs a) {
  extern __shared__ Float lds[];
  Float* const X1 = lds;
  Float* const X2 = lds + 8 * 8 * 64;
  Float* const PV = X2 + 2 * 8 * 64;
  constexpr int SLOTS_A = sw_mixed_slots(LA), SLOTS_B = sw_mixed_slots(LB);
  const int s = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  if (s < 4) {
    const int p0 = s * LA;
    sw_seg_wave<LA, G0, (LA >= 9), false>(a, X1, X2, PV + (size_t)s * SLOTS_A * 64, s, p0, min(LA, a.nlay - p0), false);
  } else {
    const int p0 = 4 * LA + (s - 4) * LB;
    sw_seg_wave<LB, G0, (LB >= 9), false>(a, X1, X2, PV + ((size_t)4 * SLOTS_A + (size_t)(s - 4) * SLOTS_B) * 64, s, p0, min(LB, a.nlay - p0), s == 7);
  }
}

// ---------------------------------------------------------------------------------------------
// LW two-stream, segmented (spectral output as the interface defines it, nlay <= 64): the scheme of
// sw_2stream_seg_kernel without the direct beam (reference :377-440, lw_two_stream :854-909,
// lw_source_2str :917-967, adding :1135-1245).  The layer sources are absolute here, so one exchange
// carries the projective composites of the adding recurrence and a second one the affine composites of
// the downward flux; every wave writes the fluxes at the levels it owns.
// ---------------------------------------------------------------------------------------------
struct Lw2SegArgs {
  int ncol, nlay, ngpt, S, g_per_block;
  bool top_at_1, lev_gpt1;
  const Float *tau, *ssa, *g, *lev_source, *sfc_emis, *sfc_src, *inc_flux;
  Float *flux_up, *flux_dn;  // (ncol, nlev, ngpt)
  // WIN (more than 96 layers, solved as an upper and a lower part; see sw_2stream_seg_kernel): window of nlay layers inside
  // arrays with these plane sizes; the "surface" given as albedo and source (in sfc_emis, sfc_src); side outputs
  size_t plane_lay, plane_lev;
  bool sfc_given, skip_first_level;
  Float *out_alb, *out_src, *out_fd;  // (ncol, ngpt)
};

template <int L, bool WIN = false>
__global__ void __launch_bounds__(64 * 8) lw_2stream_seg_kernel(Lw2SegArgs a) {
  constexpr bool PREF = L <= 10 && !WIN;  // input prefetch while it fits (see sw_2stream_seg_kernel); 11 and 12 layers per wave without
  constexpr int SMAX = 8, NC1 = 7;
  extern __shared__ Float lds[];  // X1[NC1][SMAX][64] (m00, m02, m10, m11, m12, m20, m22), X2[2][SMAX][64] (A, B)
  Float* const X1 = lds;
  Float* const X2 = lds + NC1 * SMAX * 64;
  const int lane = threadIdx.x & 63;
  const int s = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int S = a.S, ncol = a.ncol, nlay = a.nlay;
  const int icol = blockIdx.x * 64 + lane;
  const bool active = icol < ncol;
  const int c = active ? icol : ncol - 1;  // clamp: inactive lanes compute on a valid column, never store
  const size_t ncl = WIN ? a.plane_lay : (size_t)ncol * nlay, nclv = WIN ? a.plane_lev : (size_t)ncol * (nlay + 1);
  const int p0 = s * L;
  const int np = min(L, nlay - p0);  // layers in this segment (>= 1 by construction)
  const bool last = (s == S - 1);
  const int g_begin = blockIdx.y * a.g_per_block;
  const int g_end = min(a.ngpt, g_begin + a.g_per_block);
  const Float LW_diff_sec = (Float)1.66f;  // :870: default-real literal widened to wp

  struct In { Float tau[L], ssa[L], g[L], lev[L + 1], emis, ssrc, inc; };
  auto load = [&](In& x, int igpt_) {
    const int igpt = min(igpt_, g_end - 1);
    const size_t cg = c + (size_t)ncol * igpt;
    const Float* lev = a.lev_source + c + nclv * (a.lev_gpt1 ? 0 : igpt);
#pragma unroll
    for (int i = 0; i < L; ++i) {
      const int p = p0 + min(i, np - 1);
      const size_t o = c + (size_t)ncol * (a.top_at_1 ? p : nlay - 1 - p) + ncl * igpt;
      x.tau[i] = a.tau[o]; x.ssa[i] = a.ssa[o]; x.g[i] = a.g[o];
    }
#pragma unroll
    for (int i = 0; i <= L; ++i) {
      const int p = p0 + min(i, np);
      x.lev[i] = lev[(size_t)ncol * (a.top_at_1 ? p : nlay - p)];
    }
    x.emis = a.sfc_emis[cg]; x.ssrc = a.sfc_src[cg]; x.inc = a.inc_flux[cg];
  };

  auto process = [&](In& x, int igpt, int igpt_next) {
    // (no FMA contraction here: lw_two_stream's differences of nearly equal terms move by 4e-10 relative when fused)
    Float R[L], T[L], su[L], sd[L];
    // ---- (1) two-stream coefficients and sources of this segment's layers (slot i: top level i, bottom i+1)
#pragma unroll
    for (int i = 0; i < L; ++i) {
      if (i < np) {  // wave-uniform
        const Float t = x.tau[i], w0 = x.ssa[i], g = x.g[i];
        const Float gamma1 = LW_diff_sec * ((Float)1 - (Float)0.5 * w0 * ((Float)1 + g));
        const Float gamma2 = LW_diff_sec * (Float)0.5 * w0 * ((Float)1 - g);
        const Float kk = rte::sqrt_pos(fmax((gamma1 - gamma2) * (gamma1 + gamma2), (Float)1.e-12));
        const Float e1 = rte::exp_nonpos(-t * kk);
        const Float e2 = e1 * e1;
        const Float RT = rte::rcp_nr(kk * ((Float)1 + e2) + gamma1 * ((Float)1 - e2));
        const Float Rdif = RT * gamma2 * ((Float)1 - e2);
        const Float Tdif = RT * (Float)2 * kk * e1;
        const Float lev_top = x.lev[i], lev_bot = x.lev[i + 1];
        Float s_up = 0, s_dn = 0;
        if (t > (Float)1.0e-8) {
          const Float Z = rte::div_nr(lev_bot - lev_top, t * (gamma1 + gamma2));
          const Float Zup_top = Z + lev_top, Zup_bottom = Z + lev_bot;
          const Float Zdn_top = -Z + lev_top, Zdn_bottom = -Z + lev_bot;
          s_up = kPi * (Zup_top - Rdif * Zdn_top - Tdif * Zup_bottom);
          s_dn = kPi * (Zdn_bottom - Rdif * Zup_bottom - Tdif * Zdn_top);
        }
        R[i] = Rdif; T[i] = Tdif; su[i] = s_up; sd[i] = s_dn;
      } else {  // neutral layer: identity in every recurrence
        R[i] = 0; T[i] = 1; su[i] = 0; sd[i] = 0;
      }
    }
    // ---- segment composite of the adding recurrence (see sw_2stream_seg_kernel)
    Float m00 = 1, m02 = 0, m10 = 0, m11 = 1, m12 = 0, m20 = 0, m22 = 1;
#pragma unroll
    for (int i = L - 1; i >= 0; --i) {
      const Float q00 = T[i] * T[i] - R[i] * R[i], q02 = R[i], q10 = T[i] * sd[i] - su[i] * R[i], q11 = T[i], q12 = su[i],
                  q20 = -R[i];
      const Float n00 = q00 * m00 + q02 * m20, n02 = q00 * m02 + q02 * m22;
      const Float n10 = q10 * m00 + q11 * m10 + q12 * m20, n11 = q11 * m11, n12 = q10 * m02 + q11 * m12 + q12 * m22;
      const Float n20 = q20 * m00 + m20, n22 = q20 * m02 + m22;
      m00 = n00; m02 = n02; m10 = n10; m11 = n11; m12 = n12; m20 = n20; m22 = n22;
    }
    const Float emis = x.emis, ssrc = x.ssrc, inc = x.inc;
    if constexpr (PREF) load(x, igpt_next);  // the layer inputs are dead: the next g-point's go into the same registers
    X1[(0 * SMAX + s) * 64 + lane] = m00;
    X1[(1 * SMAX + s) * 64 + lane] = m02;
    X1[(2 * SMAX + s) * 64 + lane] = m10;
    X1[(3 * SMAX + s) * 64 + lane] = m11;
    X1[(4 * SMAX + s) * 64 + lane] = m12;
    X1[(5 * SMAX + s) * 64 + lane] = m20;
    X1[(6 * SMAX + s) * 64 + lane] = m22;
    __syncthreads();
    // ---- (2) adding chain from the surface up to this segment's lower edge, then the own layers
    Float alb = (WIN && a.sfc_given) ? emis : (Float)1 - emis;        // :428
    Float src = (WIN && a.sfc_given) ? ssrc : kPi * emis * ssrc;      // :965
#pragma unroll
    for (int q = SMAX - 1; q > 0; --q) {
      if (q < S && q > s) {  // wave-uniform
        const Float c00 = X1[(0 * SMAX + q) * 64 + lane], c02 = X1[(1 * SMAX + q) * 64 + lane];
        const Float c10 = X1[(2 * SMAX + q) * 64 + lane], c11 = X1[(3 * SMAX + q) * 64 + lane];
        const Float c12 = X1[(4 * SMAX + q) * 64 + lane];
        const Float c20 = X1[(5 * SMAX + q) * 64 + lane], c22 = X1[(6 * SMAX + q) * 64 + lane];
        const Float w = rte::rcp_nr(c20 * alb + c22);
        const Float a_new = (c00 * alb + c02) * w;
        const Float s_new = (c10 * alb + c11 * src + c12) * w;
        alb = a_new; src = s_new;
      }
    }
    Float al[L + 1], sr[L + 1], fa[L], fb[L];
    al[L] = alb; sr[L] = src;
#pragma unroll
    for (int i = L - 1; i >= 0; --i) {  // :1174-1186 / :1214-1226
      const Float denom = rte::rcp_nr((Float)1 - R[i] * alb);
      const Float src_new = su[i] + T[i] * denom * (src + alb * sd[i]);
      const Float alb_new = R[i] + T[i] * T[i] * alb * denom;
      fa[i] = T[i] * denom;
      fb[i] = (R[i] * src + sd[i]) * denom;
      alb = alb_new; src = src_new;
      al[i] = alb; sr[i] = src;
    }
    if constexpr (WIN) {
      if (a.out_alb && s == 0 && active) { a.out_alb[icol + (size_t)ncol * igpt] = al[0]; a.out_src[icol + (size_t)ncol * igpt] = sr[0]; }
    }
    Float A = 1, B = 0;
#pragma unroll
    for (int i = 0; i < L; ++i) { B = fa[i] * B + fb[i]; A = fa[i] * A; }
    X2[(0 * SMAX + s) * 64 + lane] = A;
    X2[(1 * SMAX + s) * 64 + lane] = B;
    __syncthreads();
    // ---- (3) flux entering the segment from above, final sweep (:1188-1202 / :1228-1243)
    Float fd = inc;  // :432
#pragma unroll
    for (int q = 0; q < SMAX - 1; ++q)
      if (q < s) fd = X2[(0 * SMAX + q) * 64 + lane] * fd + X2[(1 * SMAX + q) * 64 + lane];
    Float* fup = a.flux_up + icol + nclv * igpt;
    Float* fdn = a.flux_dn + icol + nclv * igpt;
#pragma unroll
    for (int i = 0; i <= L; ++i) {
      if (active && (i < np || (last && i == np)) && !(WIN && a.skip_first_level && s == 0 && i == 0)) {
        const int p = p0 + i;  // level position from the top
        const size_t ol = (size_t)ncol * (a.top_at_1 ? p : nlay - p);
        fup[ol] = fd * al[i] + sr[i];
        fdn[ol] = fd;
      }
      if (i < L) fd = fa[i] * fd + fb[i];
    }
    if constexpr (WIN) {
      if (a.out_fd && last && active) a.out_fd[icol + (size_t)ncol * igpt] = fd;  // (neutral slots pass it on: the window's bottom)
    }
  };

  In cur;
  if constexpr (PREF) load(cur, g_begin);
  for (int igpt = g_begin; igpt < g_end; ++igpt) {
    if constexpr (!PREF) load(cur, igpt);
    process(cur, igpt, igpt + 1);
  }
}


// ---------------------------------------------------------------------------------------------
// LW no-scattering solver WITH rescaling (Tang et al. 2018; reference :154-178, lw_transport_1rescl :753-844),
// segmented, broadband output.  Three affine sweeps per g-point, each chained across the segments through
// LDS: (1) down with the rescaled optical depth; (2) up from the surface with the source adjusted by the
// downward radiance of sweep 1 at the layer top; (3) down again with the source adjusted by the upward
// radiance (at the layer's top level when top_at_1, at its bottom level otherwise -- the reference's own
// indexing).  All three share the segment transmission; three barriers per g-point.
// ---------------------------------------------------------------------------------------------
struct LwRescArgs {
  int ncol, nlay, ngpt, S, g_per_block;
  bool top_at_1, do_jac;
  Float weight;
  const Float *D, *tau, *ssa, *g, *lay_source, *lev_source, *sfc_emis, *sfc_src, *inc_flux, *sfc_srcJac;
  Float *part_up, *part_dn, *part_jac;  // (ncol, nlev, ngroups)
};

template <int L, bool do_jac>
__global__ void __launch_bounds__(64 * 8) lw_noscat_rescale_seg_kernel(LwRescArgs a) {
#pragma clang fp contract(fast)
  constexpr bool PREF = L <= 9;  // input prefetch (5 L + 6 doubles) while it fits; 10 ... 12 layers per wave without
  constexpr int SMAX = 8;
  extern __shared__ Float lds[];  // X[2 buffers (g-point parity)][4 (T, Sd1, Su2, Sd3)][SMAX][64]
  const int lane = threadIdx.x & 63;
  const int s = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int S = a.S, ncol = a.ncol, nlay = a.nlay;
  const int icol = blockIdx.x * 64 + lane;
  const bool active = icol < ncol;
  const int c = active ? icol : ncol - 1;
  const size_t ncl = (size_t)ncol * nlay, nclv = (size_t)ncol * (nlay + 1);
  const int p0 = s * L;
  const int np = min(L, nlay - p0);
  const bool last = (s == S - 1);
  const Float piw = kPi * a.weight;
  const int g_begin = blockIdx.y * a.g_per_block;
  const int g_end = min(a.ngpt, g_begin + a.g_per_block);

  // the level accumulators live in LDS (ds_add_f64 on the thread's own slots): 4L+4 registers the three sweeps need
  constexpr int AW = L >= 14 ? 1 : 2;  // values per level slot in LDS (from 14 layers per wave on: the upward one only)
  Float* const ACC = lds + 2 * 4 * SMAX * 64 + (size_t)s * (L + 1) * AW * 64 + lane;  // [wave][slot][dn, up][64]
  // (from 14 layers per wave on the LDS holds only the upward ones; the downward ones stay in registers)
  constexpr bool DNREG = L >= 14;
  Float acc_j[do_jac ? L + 1 : 1], acc_dn[DNREG ? L + 1 : 1];
  auto add_dn = [&](int i, Float v) { if constexpr (DNREG) acc_dn[i] += v; else atomicAdd(&ACC[(2 * i) * 64], v); };
  auto add_up = [&](int i, Float v) { atomicAdd(&ACC[(AW * i + AW - 1) * 64], v); };
#pragma unroll
  for (int i = 0; i <= L; ++i) {
    if constexpr (DNREG) acc_dn[i] = 0; else ACC[(2 * i) * 64] = 0;
    ACC[(AW * i + AW - 1) * 64] = 0;
    if (do_jac) acc_j[i] = 0;
  }

  struct In { Float tau[L], ssa[L], g[L], lay[L], lev[L + 1], D, emis, ssrc, inc, sjac; };
  auto load = [&](In& x, int igpt_) {
    const int igpt = min(igpt_, g_end - 1);
    const size_t cg = c + (size_t)ncol * igpt;
#pragma unroll
    for (int i = 0; i < L; ++i) {
      const int p = p0 + min(i, np - 1);
      const size_t o = c + (size_t)ncol * (a.top_at_1 ? p : nlay - 1 - p) + ncl * igpt;
      x.tau[i] = a.tau[o]; x.ssa[i] = a.ssa[o]; x.g[i] = a.g[o]; x.lay[i] = a.lay_source[o];
    }
#pragma unroll
    for (int i = 0; i <= L; ++i) {
      const int p = p0 + min(i, np);
      x.lev[i] = a.lev_source[c + (size_t)ncol * (a.top_at_1 ? p : nlay - p) + nclv * igpt];
    }
    x.D = a.D[cg]; x.emis = a.sfc_emis[cg]; x.ssrc = a.sfc_src[cg]; x.inc = a.inc_flux[cg];
    x.sjac = do_jac ? a.sfc_srcJac[cg] : (Float)0;
  };

  auto process = [&](In& x, int igpt_next, int buf) {
#pragma clang fp contract(fast)
    // composites of g-point g live in buffer g & 1: a wave may start g+1 while another still chains g
    auto X = [&](int kind, int q) -> Float& { return lds[((buf * 4 + kind) * SMAX + q) * 64 + lane]; };
    Float t[L], sd[L], su[L], Cn[L];
    // ---- pass 1: rescaled optical depth, transmissivity, Clough sources (:154-190); composite of sweep 1
    Float Td = 1, Sd = 0;
#pragma unroll
    for (int i = 0; i < L; ++i) {
      if (i < np) {  // wave-uniform
        const Float ssal = x.ssa[i];
        const Float wb = ssal * ((Float)1 - x.g[i]) * (Float)0.5;
        const Float scaleTau = ((Float)1 - ssal + wb);
        Cn[i] = rte::div_nr((Float)0.4 * wb, scaleTau);
        const Float tau_loc = x.tau[i] * x.D * scaleTau;
        const Float tr = rte::exp_nonpos(-tau_loc);
        lw_source_layer_fast(tau_loc, tr, x.lay[i], x.lev[i], x.lev[i + 1], sd[i], su[i]);
        t[i] = tr;
      } else {  // neutral layer
        t[i] = 1; sd[i] = 0; su[i] = 0; Cn[i] = 0;
      }
      Sd = t[i] * Sd + sd[i];
      Td = Td * t[i];
    }
    const Float emis = x.emis, ssrc = x.ssrc, inc = x.inc, sjac = x.sjac;
    if constexpr (PREF) load(x, igpt_next);  // the layer inputs are dead: the next g-point's go into the same registers
    X(0, s) = Td; X(1, s) = Sd;
    __syncthreads();
    // ---- sweep 1 (down) across the segments, then inside this one
    Float r = inc / piw;  // :144
    const Float r_top = r;
    Float r_in = r;
    for (int q = 0; q < S; ++q) {
      if (q == s) r_in = r;
      r = X(0, q) * r + X(1, q);
    }
    const Float u_sfc = r * ((Float)1 - emis) + emis * ssrc;  // :198-200
    // own levels of sweep 1 and the adjusted upward sources (:771-775 / :806-810): An = 1 - trans^2
    Float su2[L];
    {
      Float d = r_in;
#pragma unroll
      for (int i = 0; i < L; ++i) {
        su2[i] = su[i] + Cn[i] * (((Float)1 - t[i] * t[i]) * d - t[i] * sd[i] - su[i]);
        d = t[i] * d + sd[i];
      }
    }
    Float Su = 0;
#pragma unroll
    for (int i = L - 1; i >= 0; --i) Su = t[i] * Su + su2[i];
    X(2, s) = Su;
    __syncthreads();
    // ---- sweep 2 (up) from the surface
    Float u = u_sfc;
    Float jv = do_jac ? emis * sjac : (Float)0;
    for (int q = S - 1; q > s; --q) {
      const Float Tq = X(0, q);
      u = Tq * u + X(2, q);
      jv = Tq * jv;
    }
    Float ul[L + 1];
    ul[L] = u;
    add_up(L, u);
    if (do_jac) acc_j[L] += jv;
#pragma unroll
    for (int i = L - 1; i >= 0; --i) {
      u = t[i] * u + su2[i];
      ul[i] = u;
      add_up(i, u);
      if (do_jac) { jv = t[i] * jv; acc_j[i] += jv; }
    }
    // adjusted downward sources of sweep 3 (:787-791 / :822-826): the upward radiance at the layer's top level
    // when top_at_1, at its bottom level otherwise (the reference indexes radn_up(ilev) in both branches)
    Float Sd3 = 0;
#pragma unroll
    for (int i = 0; i < L; ++i) {
      const Float usel = a.top_at_1 ? ul[i] : ul[i + 1];
      su2[i] = sd[i] + Cn[i] * (((Float)1 - t[i] * t[i]) * usel - t[i] * su[i] - sd[i]);  // reused as sd3
      Sd3 = t[i] * Sd3 + su2[i];
    }
    X(3, s) = Sd3;
    __syncthreads();
    // ---- sweep 3 (down again)
    r = r_top;
    for (int q = 0; q < s; ++q) r = X(0, q) * r + X(3, q);
#pragma unroll
    for (int i = 0; i < L; ++i) {
      add_dn(i, r);
      r = t[i] * r + su2[i];
    }
    add_dn(L, r);
  };

  In cur;
  if constexpr (PREF) load(cur, g_begin);
  for (int igpt = g_begin; igpt < g_end; ++igpt) {
    if constexpr (!PREF) load(cur, igpt);
    process(cur, igpt + 1, igpt & 1);
  }
  if (active) {
    const size_t base = icol + nclv * blockIdx.y;
#pragma unroll
    for (int i = 0; i <= L; ++i) {
      if (i < np || (last && i == np)) {
        const int p = p0 + i;
        const int ilev = a.top_at_1 ? p : nlay - p;
        if constexpr (DNREG) a.part_dn[base + (size_t)ncol * ilev] = acc_dn[i]; else a.part_dn[base + (size_t)ncol * ilev] = ACC[(2 * i) * 64];
        a.part_up[base + (size_t)ncol * ilev] = ACC[(AW * i + AW - 1) * 64];
        if (do_jac) a.part_jac[base + (size_t)ncol * ilev] = acc_j[i];
      }
    }
  }
}


// out(c,l) = sum over the ngroups partial slabs (in order), times scale; optionally accumulate
__global__ void __launch_bounds__(256)
reduce_parts_kernel(size_t n2, int ngroups, const Float* __restrict__ parts, Float* __restrict__ out, Float scale,
                    int accumulate) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n2) return;
  Float s = 0;
  for (int q = 0; q < ngroups; ++q) s = s + parts[i + n2 * (size_t)q];
  s = scale * s;
  out[i] = accumulate ? out[i] + s : s;
}

// the same for a run of rows of the slabs: out(i) = sum_q parts[first + i + stride q], i < n
__global__ void __launch_bounds__(256)
reduce_parts_rows_kernel(size_t n, size_t first, size_t stride, int ngroups, const Float* __restrict__ parts, Float* __restrict__ out,
                         Float scale, int accumulate) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  Float s = 0;
  for (int q = 0; q < ngroups; ++q) s = s + parts[first + i + stride * (size_t)q];
  s = scale * s;
  out[i] = accumulate ? out[i] + s : s;
}
__global__ void __launch_bounds__(256) fill_kernel(Float* __restrict__ p, size_t n, Float v) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) p[i] = v;
}

size_t pick_gchunk(size_t bytes_per_g, int ngpt) {
  const size_t budget = (size_t)3 << 30;  // scratch budget for the generic solvers
  size_t c = budget / (bytes_per_g ? bytes_per_g : 1);
  if (c < 1) c = 1;
  if (c > (size_t)ngpt) c = ngpt;
  return c;
}

}  // namespace

// bug-compat switch for rte_lw_solver_2stream (see DESIGN.md / SURVEY.md section 9-1):
// 0 (default) = each g-point uses its own level source (what the reference accel kernel and
// the physics intend); 1 = replicate the reference default CPU kernel, which passes the 3-D
// lev_source to a 2-D dummy and therefore uses g-point 1's level source everywhere.
static std::atomic<int> g_lw2str_gpt1_levsource{0};
static std::atomic<int> g_lw_force_generic{0};
static std::atomic<int> g_sw_force_generic{0};
static int env_switch(const char* name, int dflt) { const char* e = getenv(name); return e ? atoi(e) : dflt; }
static std::atomic<int> g_lw_mixed{env_switch("RTE_LW_MIXED", 1)};  // lw_solver_noscat, 57 ... 60 layers, broadband: lw_noscat_seg_mixed_kernel (rte_hip_lw_mixed_segments)
static std::atomic<int> g_sw_mixed{env_switch("RTE_SW_MIXED", 1)};  // 57 ... 60 layers on sw_2stream_seg_mixed_kernel (rte_hip_sw_mixed_segments)
static std::atomic<int> g_lw_sfc_lds{1};  // surface arrays of the LW segmented solver through LDS chunks (rte_hip_lw_sfc_lds)
static std::atomic<int> g_seg_groups{0};  // > 0: g-point groups per column tile of the segmented solvers (rte_hip_seg_groups; 0 = automatic;
                              // < 0: g-points per block given directly)
// g-points per block of the segmented broadband solvers (grid = column tiles x groups, each block accumulates its
// g-points into a partial slab).  A few LONG groups and one SHORT last group -- grid.y is the slow launch index, so the
// short blocks run last: the long blocks keep the partial slabs few, the short ones fill the tail of the launch.
// 1e5 columns: 120 + 120 + 16 of 256 g-points (lw_noscat: 6.96-7.03 ms against 7.10-7.17 for 86 + 86 + 84 and
// 7.07-7.16 for 4 x 64), 104 + 104 + 16 of 224 (sw_2stream: 12.41-12.44 against 12.56-12.64 for 4 x 56); same results
// (tools/sweep_seg_groups.py, bench.py --seg-groups).
static int seg_g_per_block(int col_tiles, int ngpt) {
  const int tail_g = ngpt >= 64 ? 16 : 0;
  int nlong = 1;
  while (nlong < 15 && (size_t)col_tiles * nlong < 3072 && (ngpt - tail_g) / (nlong + 1) >= 8) ++nlong;  // >= 12 long blocks per CU
  int gpb = ((ngpt - tail_g + nlong - 1) / nlong + 7) / 8 * 8;
  if (gpb > ngpt) gpb = ngpt;
  const int sg = g_seg_groups;
  if (sg > 0) { const int n = sg < ngpt ? sg : ngpt; gpb = (ngpt + n - 1) / n; }
  if (sg < 0) gpb = -sg < ngpt ? -sg : ngpt;
  return gpb;
}

extern "C" {

int rte_hip_set_lw2str_bugcompat(int on) { g_lw2str_gpt1_levsource = on; return 0; }
int rte_hip_force_generic_lw(int on) { g_lw_force_generic = on; return 0; }
int rte_hip_force_generic_sw(int on) { g_sw_force_generic = on; return 0; }
int rte_hip_sw_mixed_segments(int on) { g_sw_mixed = on; return 0; }
int rte_hip_lw_mixed_segments(int on) { g_lw_mixed = on; return 0; }
int rte_hip_seg_groups(int n) { g_seg_groups = n; return 0; }
int rte_hip_lw_sfc_lds(int on) { g_lw_sfc_lds = on; return 0; }

int rte_hip_lw_solver_noscat_factored(int ncol, int nlay, int ngpt, int nbnd, int top_at_1, int nmus, const Float* Ds,
                                      const Float* weights, const int* band_lims_gpt, const Float* tau, const Float* pfrac,
                                      const Float* planck_lay, const Float* planck_lev, const Float* sfc_emis, const Float* sfc_src,
                                      const Float* inc_flux, Float* broadband_up, Float* broadband_dn, int do_jac,
                                      const Float* sfc_srcJac, Float* flux_upJac);
void rte_lw_solver_noscat(const int* ncol_, const int* nlay_, const int* ngpt_, const Bool* top_at_1,
                          const int* nmus_, const Float* Ds, const Float* weights, const Float* tau,
                          const Float* lay_source, const Float* lev_source, const Float* sfc_emis,
                          const Float* sfc_src, const Float* inc_flux, Float* flux_up,
                          Float* flux_dn, const Bool* do_broadband_, Float* broadband_up,
                          Float* broadband_dn, const Bool* do_Jacobians_, const Float* sfc_srcJac,
                          Float* flux_upJac, const Bool* do_rescaling_, const Float* ssa,
                          const Float* g) {
  const int ncol = *ncol_, nlay = *nlay_, ngpt = *ngpt_, nmus = *nmus_, nlev = nlay + 1;
  const bool do_broadband = *do_broadband_, do_jac = *do_Jacobians_, do_rescaling = *do_rescaling_;
  if (ncol <= 0 || nlay <= 0 || ngpt <= 0 || nmus <= 0) return;
  {
    // deferred sources (rte_hip_defer_sources; runtime.hip, planck.hip): lay_source holds the Planck fraction and a record
    // names the bands' Planck functions -- solve from the factors (bit-identical fluxes), then leave the record in place so
    // that a later use of these arrays still finds them expanded
    rte::PendingSources ps;
    if (rte::take_pending_sources(lay_source, lev_source, &ps)) {
      int rc = -2;
      if (do_broadband && !do_rescaling && ps.ncol == ncol && ps.nlay == nlay && ps.ngpt == ngpt)
        rc = rte_hip_lw_solver_noscat_factored(ncol, nlay, ngpt, ps.nbnd, *top_at_1 ? 1 : 0, nmus, Ds, weights, ps.band_lims, tau,
                                               lay_source, (const Float*)ps.plk_lay, (const Float*)ps.plk_lev, sfc_emis, sfc_src,
                                               inc_flux, broadband_up, broadband_dn, do_jac ? 1 : 0, sfc_srcJac, flux_upJac);
      // back on the list: consumed here (from now on the record is used only while lay_source still holds the fraction:
      // common.h) or expanded by Call::in below
      if (rc == 0) { rte::sources_consumed(ps); return; }
      rte::defer_sources(ps, nullptr);
    }
  }
  RTE_TRY
  rte::Call c("rte_lw_solver_noscat");
  const size_t ncl = (size_t)ncol * nlay, nclv = (size_t)ncol * nlev, ncg = (size_t)ncol * ngpt;
  const Float* w_h = c.host(weights, (size_t)nmus);
  const Float* d_Ds = c.in(Ds, ncg * nmus);
  const Float* d_tau = c.in(tau, ncl * ngpt);
  const Float* d_lay = c.in(lay_source, ncl * ngpt);
  const Float* d_lev = c.in(lev_source, nclv * ngpt);
  const Float* d_emis = c.in(sfc_emis, ncg);
  const Float* d_sfc = c.in(sfc_src, ncg);
  const Float* d_inc = c.in(inc_flux, ncg);
  const Float* d_srcJac = do_jac ? c.in(sfc_srcJac, ncg) : nullptr;
  const Float* d_ssa = do_rescaling ? c.in(ssa, ncl * ngpt) : nullptr;
  const Float* d_g = do_rescaling ? c.in(g, ncl * ngpt) : nullptr;
  Float* d_flux_up = do_broadband ? nullptr : c.out(flux_up, nclv * ngpt);
  Float* d_flux_dn = do_broadband ? nullptr : c.out(flux_dn, nclv * ngpt);
  Float* d_bb_up = do_broadband ? c.out(broadband_up, nclv) : nullptr;
  Float* d_bb_dn = do_broadband ? c.out(broadband_dn, nclv) : nullptr;
  Float* d_jac = do_jac ? c.out(flux_upJac, nclv) : nullptr;
  hipStream_t st = rte::stream();

  // ------------------------------------------------------------------ production path
  // layers per segment (8 waves per block): 8, 9 or 10 -- up to 80 layers.  Wider segments spill registers and
  // lose to the generic kernel (measured: 12 layers per wave 20.6 vs 18.4 ms, 16 per wave 44 vs 19 ms at 1e5 x 128)
  const int L = nlay <= 64 ? 8 : nlay <= 72 ? 9 : 10;
  const int S = (nlay + L - 1) / L;
  if (do_broadband && do_rescaling && nlay <= 144 && !g_lw_force_generic) {
    // ------------------------------------------------------------------ production path with rescaling
    const int Lr = nlay <= 64 ? 8 : nlay <= 72 ? 9 : nlay <= 80 ? 10 : nlay <= 88 ? 11 : nlay <= 96 ? 12 : nlay <= 112 ? 14 : nlay <= 128 ? 16 : 18;
    const int Sr = (nlay + Lr - 1) / Lr;
    const int col_tiles = cdiv(ncol, 64);
    const int g_per_block = seg_g_per_block(col_tiles, ngpt);
    const int ngroups = (ngpt + g_per_block - 1) / g_per_block;
    LwRescArgs q;
    q.ncol = ncol; q.nlay = nlay; q.ngpt = ngpt; q.S = Sr; q.g_per_block = g_per_block; q.top_at_1 = *top_at_1; q.do_jac = do_jac;
    q.tau = d_tau; q.ssa = d_ssa; q.g = d_g; q.lay_source = d_lay; q.lev_source = d_lev; q.sfc_emis = d_emis; q.sfc_src = d_sfc;
    q.inc_flux = d_inc; q.sfc_srcJac = d_srcJac;
    q.part_up = (Float*)rte::scratch(sizeof(Float) * nclv * ngroups * (do_jac ? 3 : 2));
    q.part_dn = q.part_up + nclv * ngroups;
    q.part_jac = do_jac ? q.part_dn + nclv * ngroups : nullptr;
    // composites + level accumulators (from 14 layers per wave on: the upward ones only, in the same [slot][2] layout)
    const size_t lds_bytes = sizeof(Float) * (2 * 4 * 8 * 64 + 8 * (Lr + 1) * (Lr >= 14 ? 1 : 2) * 64);
    for (int imu = 0; imu < nmus; ++imu) {
      q.weight = w_h[imu]; q.D = d_Ds + ncg * imu;
      {
        rte::ProfScope p("lw_noscat_rescale_seg_kernel");
#define RTE_LAUNCH_RESC(LL, JJ) \
  hipLaunchKernelGGL((lw_noscat_rescale_seg_kernel<LL, JJ>), dim3(col_tiles, ngroups), dim3(64 * Sr), lds_bytes, st, q)
        if (Lr == 8)       { if (do_jac) RTE_LAUNCH_RESC(8, true); else RTE_LAUNCH_RESC(8, false); }
        else if (Lr == 9)  { if (do_jac) RTE_LAUNCH_RESC(9, true); else RTE_LAUNCH_RESC(9, false); }
        else if (Lr == 10) { if (do_jac) RTE_LAUNCH_RESC(10, true); else RTE_LAUNCH_RESC(10, false); }
        else if (Lr == 11) { if (do_jac) RTE_LAUNCH_RESC(11, true); else RTE_LAUNCH_RESC(11, false); }
        else if (Lr == 12) { if (do_jac) RTE_LAUNCH_RESC(12, true); else RTE_LAUNCH_RESC(12, false); }
        else if (Lr == 14) { if (do_jac) RTE_LAUNCH_RESC(14, true); else RTE_LAUNCH_RESC(14, false); }
        else if (Lr == 16) { if (do_jac) RTE_LAUNCH_RESC(16, true); else RTE_LAUNCH_RESC(16, false); }
        else               { if (do_jac) RTE_LAUNCH_RESC(18, true); else RTE_LAUNCH_RESC(18, false); }
#undef RTE_LAUNCH_RESC
      }
      rte::ProfScope p("lw_reduce_parts");
      const Float piw = (Float)3.14159265358979323846264338327950288 * w_h[imu];
      hipLaunchKernelGGL(reduce_parts_kernel, dim3(cdiv(nclv, 256)), dim3(256), 0, st, nclv, ngroups, q.part_up, d_bb_up, piw, imu > 0);
      hipLaunchKernelGGL(reduce_parts_kernel, dim3(cdiv(nclv, 256)), dim3(256), 0, st, nclv, ngroups, q.part_dn, d_bb_dn, piw, imu > 0);
      if (do_jac)
        hipLaunchKernelGGL(reduce_parts_kernel, dim3(cdiv(nclv, 256)), dim3(256), 0, st, nclv, ngroups, q.part_jac, d_jac, piw, imu > 0);
    }
    return;
  }
  if (do_broadband && !do_rescaling && nlay <= 80 && !g_lw_force_generic && nclv < ((size_t)1 << 29)) {  // 32-bit in-plane byte offsets
    // g-points per block: enough blocks to fill the chip several times over, few enough partial slabs
    const int col_tiles = cdiv(ncol, 64);
    const int g_per_block = seg_g_per_block(col_tiles, ngpt);
    const int ngroups = (ngpt + g_per_block - 1) / g_per_block;
    Float* part_up = (Float*)rte::scratch(sizeof(Float) * nclv * ngroups * (do_jac ? 3 : 2));
    Float* part_dn = part_up + nclv * ngroups;
    Float* part_jac = do_jac ? part_dn + nclv * ngroups : nullptr;
    const bool sfclds = g_lw_sfc_lds && S == 8 && (L == 8 || !do_jac);  // (the wider variants with Jacobians run out of registers)
    const size_t lds_bytes = sizeof(Float) * (2 * 3 * 8 * 64 + (sfclds ? 2 * 16 * (do_jac ? 5 : 4) * 64 : 0));
    for (int imu = 0; imu < nmus; ++imu) {
      {
        rte::ProfScope p("lw_noscat_seg_kernel");
#define RTE_LAUNCH_SEG(LL, JJ)                                                                                  \
  do {                                                                                                          \
    if (sfclds)                                                                                                 \
      hipLaunchKernelGGL((lw_noscat_seg_kernel<LL, JJ, true>), dim3(col_tiles, ngroups), dim3(64 * S), lds_bytes, st, ncol, \
                         nlay, ngpt, S, g_per_block, (bool)*top_at_1, w_h[imu], d_Ds + ncg * imu, d_tau, d_lay, d_lev, \
                         d_emis, d_sfc, d_inc, d_srcJac, part_up, part_dn, part_jac);                           \
    else                                                                                                        \
      hipLaunchKernelGGL((lw_noscat_seg_kernel<LL, JJ, false>), dim3(col_tiles, ngroups), dim3(64 * S), lds_bytes, st, ncol, \
                         nlay, ngpt, S, g_per_block, (bool)*top_at_1, w_h[imu], d_Ds + ncg * imu, d_tau, d_lay, d_lev, \
                         d_emis, d_sfc, d_inc, d_srcJac, part_up, part_dn, part_jac);                           \
  } while (0)
        // 57 ... 60 layers: segments of 7 and 8 layers (lw_noscat_seg_mixed_kernel); rte_hip_lw_mixed_segments(0) for A/B
        if (g_lw_mixed && sfclds && S == 8 && nlay >= 57 && nlay <= 60) {
          if (do_jac)
            hipLaunchKernelGGL((lw_noscat_seg_mixed_kernel<7, 8, true, true>), dim3(col_tiles, ngroups), dim3(64 * 8), lds_bytes, st, ncol, nlay,
                               ngpt, S, g_per_block, (bool)*top_at_1, w_h[imu], d_Ds + ncg * imu, d_tau, d_lay, d_lev, d_emis, d_sfc, d_inc,
                               d_srcJac, part_up, part_dn, part_jac);
          else
            hipLaunchKernelGGL((lw_noscat_seg_mixed_kernel<7, 8, false, true>), dim3(col_tiles, ngroups), dim3(64 * 8), lds_bytes, st, ncol, nlay,
                               ngpt, S, g_per_block, (bool)*top_at_1, w_h[imu], d_Ds + ncg * imu, d_tau, d_lay, d_lev, d_emis, d_sfc, d_inc,
                               d_srcJac, part_up, part_dn, part_jac);
        } else
        if (L == 8) { if (do_jac) RTE_LAUNCH_SEG(8, true); else RTE_LAUNCH_SEG(8, false); }
        else if (L == 9) { if (do_jac) RTE_LAUNCH_SEG(9, true); else RTE_LAUNCH_SEG(9, false); }
        else { if (do_jac) RTE_LAUNCH_SEG(10, true); else RTE_LAUNCH_SEG(10, false); }
#undef RTE_LAUNCH_SEG
      }
      rte::ProfScope p("lw_reduce_parts");
      const Float piw = (Float)3.14159265358979323846264338327950288 * w_h[imu];
      hipLaunchKernelGGL(reduce_parts_kernel, dim3(cdiv(nclv, 256)), dim3(256), 0, st, nclv, ngroups, part_up,
                         d_bb_up, piw, imu > 0);
      hipLaunchKernelGGL(reduce_parts_kernel, dim3(cdiv(nclv, 256)), dim3(256), 0, st, nclv, ngroups, part_dn,
                         d_bb_dn, piw, imu > 0);
      if (do_jac)
        hipLaunchKernelGGL(reduce_parts_kernel, dim3(cdiv(nclv, 256)), dim3(256), 0, st, nclv, ngroups, part_jac,
                           d_jac, piw, imu > 0);
    }
    return;
  }

  if (!do_broadband && !do_rescaling && nlay <= 80 && !g_lw_force_generic && nclv < ((size_t)1 << 29)) {
    // ---------------------------------------------------------------- production path, spectral output
    // the same segmented kernel; every wave stores the fluxes of the levels it owns per g-point (25 GB of stores at
    // 1e5 x 60 x 256: the kernel is bound by them); angles after the first add to the arrays
    const int col_tiles = cdiv(ncol, 64);
    const int g_per_block = seg_g_per_block(col_tiles, ngpt);
    const int ngroups = (ngpt + g_per_block - 1) / g_per_block;
    Float* part_jac = do_jac ? (Float*)rte::scratch(sizeof(Float) * nclv * ngroups) : nullptr;
    const size_t lds_bytes = sizeof(Float) * (2 * 3 * 8 * 64);
    for (int imu = 0; imu < nmus; ++imu) {
      {
        rte::ProfScope p("lw_noscat_seg_spectral_kernel");
#define RTE_LAUNCH_SEGS(LL, JJ)                                                                                        \
  hipLaunchKernelGGL((lw_noscat_seg_kernel<LL, JJ, false, true>), dim3(col_tiles, ngroups), dim3(64 * S), lds_bytes, st, ncol, \
                     nlay, ngpt, S, g_per_block, (bool)*top_at_1, w_h[imu], d_Ds + ncg * imu, d_tau, d_lay, d_lev,    \
                     d_emis, d_sfc, d_inc, d_srcJac, (Float*)nullptr, (Float*)nullptr, part_jac, d_flux_up, d_flux_dn, imu > 0)
        if (L == 8) { if (do_jac) RTE_LAUNCH_SEGS(8, true); else RTE_LAUNCH_SEGS(8, false); }
        else if (L == 9) { if (do_jac) RTE_LAUNCH_SEGS(9, true); else RTE_LAUNCH_SEGS(9, false); }
        else { if (do_jac) RTE_LAUNCH_SEGS(10, true); else RTE_LAUNCH_SEGS(10, false); }
#undef RTE_LAUNCH_SEGS
      }
      if (do_jac) {
        rte::ProfScope p("lw_reduce_parts");
        const Float piw = (Float)3.14159265358979323846264338327950288 * w_h[imu];
        hipLaunchKernelGGL(reduce_parts_kernel, dim3(cdiv(nclv, 256)), dim3(256), 0, st, nclv, ngroups, part_jac, d_jac, piw, imu > 0);
      }
    }
    return;
  }

  if (!do_rescaling && nlay > 80 && nlay <= 176 && !g_lw_force_generic && nclv < ((size_t)1 << 29)) {
    // ---------------------------------------------------------------- production path, 81 ... 160 layers: two sub-segments
    // of 8 ... 11 layers per wave (lw_noscat_seg2_kernel), broadband or spectral output
    const int L2 = nlay <= 128 ? 8 : nlay <= 144 ? 9 : nlay <= 160 ? 10 : 11;
    const int S2 = (nlay + 2 * L2 - 1) / (2 * L2);
    const int col_tiles = cdiv(ncol, 64);
    const int g_per_block = seg_g_per_block(col_tiles, ngpt);
    const int ngroups = (ngpt + g_per_block - 1) / g_per_block;
    const int nparts = (do_broadband ? 2 : 0) + (do_jac ? 1 : 0);
    Float* parts = nparts ? (Float*)rte::scratch(sizeof(Float) * nclv * ngroups * nparts) : nullptr;
    Float* part_up = do_broadband ? parts : nullptr;
    Float* part_dn = do_broadband ? parts + nclv * ngroups : nullptr;
    Float* part_jac = do_jac ? parts + nclv * ngroups * (do_broadband ? 2 : 0) : nullptr;
    const size_t lds_bytes = sizeof(Float) * (2 * 3 * 8 * 64 + 3 * L2 * 512 + (L2 == 9 ? 6 : L2 == 10 ? 3 : 0) * 512);
    for (int imu = 0; imu < nmus; ++imu) {
      {
        rte::ProfScope p("lw_noscat_seg2_kernel");
#define RTE_LAUNCH_SEG2(LL, JJ, SS)                                                                                       \
  do {                                                                                                                    \
    HIP_CHECK(hipFuncSetAttribute((const void*)lw_noscat_seg2_kernel<LL, JJ, SS>, hipFuncAttributeMaxDynamicSharedMemorySize, \
                                  (int)lds_bytes));                                                                       \
    hipLaunchKernelGGL((lw_noscat_seg2_kernel<LL, JJ, SS>), dim3(col_tiles, ngroups), dim3(64 * S2), lds_bytes, st, ncol, \
                       nlay, ngpt, S2, g_per_block, (bool)*top_at_1, w_h[imu], d_Ds + ncg * imu, d_tau, d_lay, d_lev,     \
                       d_emis, d_sfc, d_inc, d_srcJac, part_up, part_dn, part_jac, d_flux_up, d_flux_dn, imu > 0, LwWin{}); \
  } while (0)
#define RTE_LAUNCH_SEG2_(LL)                                                                     \
  do {                                                                                           \
    if (do_broadband) { if (do_jac) RTE_LAUNCH_SEG2(LL, true, false); else RTE_LAUNCH_SEG2(LL, false, false); } \
    else              { if (do_jac) RTE_LAUNCH_SEG2(LL, true, true); else RTE_LAUNCH_SEG2(LL, false, true); }   \
  } while (0)
        if (L2 == 8) RTE_LAUNCH_SEG2_(8); else if (L2 == 9) RTE_LAUNCH_SEG2_(9); else if (L2 == 10) RTE_LAUNCH_SEG2_(10); else RTE_LAUNCH_SEG2_(11);
#undef RTE_LAUNCH_SEG2_
#undef RTE_LAUNCH_SEG2
      }
      rte::ProfScope p("lw_reduce_parts");
      const Float piw = (Float)3.14159265358979323846264338327950288 * w_h[imu];
      if (do_broadband) {
        hipLaunchKernelGGL(reduce_parts_kernel, dim3(cdiv(nclv, 256)), dim3(256), 0, st, nclv, ngroups, part_up, d_bb_up, piw, imu > 0);
        hipLaunchKernelGGL(reduce_parts_kernel, dim3(cdiv(nclv, 256)), dim3(256), 0, st, nclv, ngroups, part_dn, d_bb_dn, piw, imu > 0);
      }
      if (do_jac)
        hipLaunchKernelGGL(reduce_parts_kernel, dim3(cdiv(nclv, 256)), dim3(256), 0, st, nclv, ngroups, part_jac, d_jac, piw, imu > 0);
    }
    return;
  }

  constexpr int kWinLay = 128, kWinMax = 8;  // layers per window (8 per sub-segment: no spills), windows per column
  if (!do_rescaling && do_broadband && nlay > 176 && nlay <= kWinLay * kWinMax && !g_lw_force_generic && nclv < ((size_t)1 << 29)) {
    // ---------------------------------------------------------------- 177 ... 1024 layers: the column as K = ceil(nlay / 128)
    // WINDOWS of layers on the two-sub-segment kernel.  Per angle, top to bottom: (A) every window but the last for the
    // downward radiance at its bottom, which enters the next one; (B) the last window with the real surface: its fluxes, and the
    // upward radiance (and Jacobian) at its top; then bottom to top (C) every other window once more over a "surface" of
    // emissivity 1 that emits the radiance of the window below: its fluxes.  All windows but the last are read twice
    // ((2 K - 1) / K of the traffic of one pass; the generic kernel reads everything twice at a third of the speed).
    const int K = (nlay + kWinLay - 1) / kWinLay;
    int wn[kWinMax], wrow[kWinMax];  // layers of window w (from the top), first array row of its layers and levels
    for (int w = 0, pos = 0; w < K; ++w) {
      wn[w] = nlay / K + (w < nlay % K ? 1 : 0);
      wrow[w] = *top_at_1 ? pos : nlay - pos - wn[w];
      pos += wn[w];
    }
    const int col_tiles = cdiv(ncol, 64);
    const int g_per_block = seg_g_per_block(col_tiles, ngpt);
    const int ngroups = (ngpt + g_per_block - 1) / g_per_block;
    const size_t nclv_w = (size_t)ncol * (wn[0] + 1);  // (the first window is the largest)
    const int nparts = 2 + (do_jac ? 1 : 0);
    Float* parts = (Float*)rte::scratch(sizeof(Float) * nclv_w * ngroups * nparts);
    // (column, g-point) arrays: the downward radiance entering windows 1 ... K - 1, the upward radiance and its Jacobian at an
    // interface (two each: a window reads the one below it and writes the one above it), ones
    Float* edge = (Float*)rte::scratch(sizeof(Float) * ncg * (K + 4));
    Float *e_dn = edge, *e_up = edge + ncg * (K - 1), *e_jv = e_up + 2 * ncg, *e_one = e_jv + 2 * ncg;
    hipLaunchKernelGGL(fill_kernel, dim3(cdiv(ncg, 256)), dim3(256), 0, st, e_one, ncg, (Float)1);
    // store: 0 nothing, 1 all of the window's levels, 2 all but its top level (the window above owns an interface)
    auto run_window = [&](int nw, int lay0, int lev0, Float weight, const Float* Dsec, const Float* emis_, const Float* sfc_, const Float* inc_,
                          const Float* jac_, LwWin win, int store, bool accumulate) {
      const int L2 = nw <= 128 ? 8 : nw <= 144 ? 9 : nw <= 160 ? 10 : 11;
      const int S2 = (nw + 2 * L2 - 1) / (2 * L2);
      const size_t nclv_k = (size_t)ncol * (nw + 1);
      Float* part_up = parts;
      Float* part_dn = parts + nclv_k * ngroups;
      Float* part_jac = do_jac ? parts + nclv_k * ngroups * 2 : nullptr;
      const size_t lds_bytes = sizeof(Float) * (2 * 3 * 8 * 64 + 3 * L2 * 512 + (L2 == 9 ? 6 : L2 == 10 ? 3 : 0) * 512);
      win.plane_lay = ncl; win.plane_lev = nclv;
      const Float* t_ = d_tau + (size_t)ncol * lay0;
      const Float* l_ = d_lay + (size_t)ncol * lay0;
      const Float* v_ = d_lev + (size_t)ncol * lev0;
      {
        rte::ProfScope p("lw_noscat_seg2_kernel");
#define RTE_LAUNCH_WIN(LL, JJ)                                                                                            \
  do {                                                                                                                    \
    HIP_CHECK(hipFuncSetAttribute((const void*)lw_noscat_seg2_kernel<LL, JJ, false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, \
                                  (int)lds_bytes));                                                                       \
    hipLaunchKernelGGL((lw_noscat_seg2_kernel<LL, JJ, false, true>), dim3(col_tiles, ngroups), dim3(64 * S2), lds_bytes, st, ncol, \
                       nw, ngpt, S2, g_per_block, (bool)*top_at_1, weight, Dsec, t_, l_, v_, emis_, sfc_, inc_, jac_, part_up,     \
                       part_dn, part_jac, (Float*)nullptr, (Float*)nullptr, false, win);                                  \
  } while (0)
#define RTE_LAUNCH_WIN_(LL) do { if (do_jac) RTE_LAUNCH_WIN(LL, true); else RTE_LAUNCH_WIN(LL, false); } while (0)
        if (L2 == 8) RTE_LAUNCH_WIN_(8); else if (L2 == 9) RTE_LAUNCH_WIN_(9); else if (L2 == 10) RTE_LAUNCH_WIN_(10); else RTE_LAUNCH_WIN_(11);
#undef RTE_LAUNCH_WIN_
#undef RTE_LAUNCH_WIN
      }
      if (!store) return;
      rte::ProfScope p("lw_reduce_parts");
      const Float piw = (Float)3.14159265358979323846264338327950288 * weight;
      // the window's levels are a contiguous run of rows of the (ncol, nlev) outputs; the interface level is the lower window's
      // first row (top_at_1) or last row
      const size_t first = (store == 2 && *top_at_1) ? (size_t)ncol : 0, n = nclv_k - (store == 2 ? (size_t)ncol : 0);
      const size_t o = (size_t)ncol * lev0 + first;
      const int acc = accumulate ? 1 : 0;
      hipLaunchKernelGGL(reduce_parts_rows_kernel, dim3(cdiv(n, 256)), dim3(256), 0, st, n, first, nclv_k, ngroups, (const Float*)part_up, d_bb_up + o, piw, acc);
      hipLaunchKernelGGL(reduce_parts_rows_kernel, dim3(cdiv(n, 256)), dim3(256), 0, st, n, first, nclv_k, ngroups, (const Float*)part_dn, d_bb_dn + o, piw, acc);
      if (do_jac)
        hipLaunchKernelGGL(reduce_parts_rows_kernel, dim3(cdiv(n, 256)), dim3(256), 0, st, n, first, nclv_k, ngroups, (const Float*)part_jac, d_jac + o, piw, acc);
    };
    for (int imu = 0; imu < nmus; ++imu) {
      const Float* Dsec = d_Ds + ncg * imu;
      auto inc_of = [&](int w) { return w == 0 ? d_inc : (const Float*)(e_dn + ncg * (w - 1)); };
      for (int w = 0; w + 1 < K; ++w) {  // (A)
        LwWin wa{}; wa.inc_is_radiance = w > 0; wa.out_dn_bot = e_dn + ncg * w;
        run_window(wn[w], wrow[w], wrow[w], w_h[imu], Dsec, e_one, d_sfc, inc_of(w), do_jac ? d_srcJac : nullptr, wa, 0, false);
      }
      int cur = 0;  // which of the two (e_up, e_jv) pairs holds the radiance below the window being solved
      {             // (B)
        LwWin wb{}; wb.inc_is_radiance = true; wb.out_up_top = e_up + ncg * cur; wb.out_jv_top = do_jac ? e_jv + ncg * cur : nullptr;
        run_window(wn[K - 1], wrow[K - 1], wrow[K - 1], w_h[imu], Dsec, d_emis, d_sfc, inc_of(K - 1), do_jac ? d_srcJac : nullptr, wb, 2, imu > 0);
      }
      for (int w = K - 2; w >= 0; --w, cur ^= 1) {  // (C)
        LwWin wc{}; wc.inc_is_radiance = w > 0;
        if (w > 0) { wc.out_up_top = e_up + ncg * (cur ^ 1); wc.out_jv_top = do_jac ? e_jv + ncg * (cur ^ 1) : nullptr; }
        run_window(wn[w], wrow[w], wrow[w], w_h[imu], Dsec, e_one, e_up + ncg * cur, inc_of(w), do_jac ? e_jv + ncg * cur : nullptr, wc,
                   w > 0 ? 2 : 1, imu > 0);
      }
    }
    return;
  }

  // ------------------------------------------------------------------ generic path
  // spectral intensities for a chunk of g-points: in the caller's arrays (single angle, spectral
  // output) or in scratch; then reduce / accumulate.
  const bool direct = !do_broadband && nmus == 1;
  const int nslab = (direct ? 0 : 2) + (do_jac ? 1 : 0);
  const size_t gchunk = direct && !do_jac ? (size_t)ngpt : pick_gchunk(sizeof(Float) * nclv * (nslab ? nslab : 1), ngpt);
  Float* ws = nslab ? (Float*)rte::scratch(sizeof(Float) * nclv * gchunk * nslab) : nullptr;
  Float* bb_tmp = (do_broadband || do_jac) ? (Float*)rte::scratch(sizeof(Float) * nclv * 3) : nullptr;
  for (int imu = 0; imu < nmus; ++imu) {
    const Float piw = (Float)3.14159265358979323846264338327950288 * w_h[imu];
    for (int g0 = 0; g0 < ngpt; g0 += (int)gchunk) {
      const int gc = (int)((size_t)(ngpt - g0) < gchunk ? (size_t)(ngpt - g0) : gchunk);
      LwArgs a;
      a.ncol = ncol; a.nlay = nlay; a.ngpt = ngpt; a.g_begin = g0; a.gchunk = gc;
      a.top_at_1 = *top_at_1; a.do_jac = do_jac; a.do_rescaling = do_rescaling;
      a.scale_out = !do_broadband; a.accumulate = false; a.weight = w_h[imu];
      a.D = d_Ds + ncg * imu; a.tau = d_tau; a.lay_source = d_lay; a.lev_source = d_lev;
      a.sfc_emis = d_emis; a.sfc_src = d_sfc; a.inc_flux = d_inc; a.sfc_srcJac = d_srcJac;
      a.ssa = d_ssa; a.g = d_g;
      Float* w = ws;
      if (direct) {
        a.radn_up = d_flux_up + nclv * g0;
        a.radn_dn = d_flux_dn + nclv * g0;
      } else {
        a.radn_up = w; w += nclv * gchunk;
        a.radn_dn = w; w += nclv * gchunk;
      }
      a.jac = do_jac ? w : nullptr;
      {
        rte::ProfScope p("lw_noscat_generic_kernel");
        hipLaunchKernelGGL(lw_noscat_generic_kernel, dim3(cdiv(ncol, 256), gc), dim3(256), 0, st, a);
      }
      rte::ProfScope p("lw_generic_reduce");
      const bool first_chunk = g0 == 0, last_chunk = g0 + gc >= ngpt;
      if (do_broadband) {
        // per-angle unscaled sums in bb_tmp, scaled and added to the outputs after the last chunk
        hipLaunchKernelGGL(sum_gpt_kernel, dim3(cdiv(nclv, 256)), dim3(256), 0, st, nclv, gc, a.radn_up, bb_tmp,
                           (Float)1, first_chunk ? 0 : 1);
        hipLaunchKernelGGL(sum_gpt_kernel, dim3(cdiv(nclv, 256)), dim3(256), 0, st, nclv, gc, a.radn_dn,
                           bb_tmp + nclv, (Float)1, first_chunk ? 0 : 1);
        if (last_chunk) {
          hipLaunchKernelGGL(reduce_parts_kernel, dim3(cdiv(nclv, 256)), dim3(256), 0, st, nclv, 1, bb_tmp, d_bb_up,
                             piw, imu > 0);
          hipLaunchKernelGGL(reduce_parts_kernel, dim3(cdiv(nclv, 256)), dim3(256), 0, st, nclv, 1, bb_tmp + nclv,
                             d_bb_dn, piw, imu > 0);
        }
      } else if (!direct) {
        // spectral output with several angles: flux(:,:,g) (+)= this angle's flux
        hipLaunchKernelGGL(axpy_kernel, dim3(cdiv(nclv * gc, 256)), dim3(256), 0, st, nclv * gc, a.radn_up,
                           d_flux_up + nclv * g0, imu == 0);
        hipLaunchKernelGGL(axpy_kernel, dim3(cdiv(nclv * gc, 256)), dim3(256), 0, st, nclv * gc, a.radn_dn,
                           d_flux_dn + nclv * g0, imu == 0);
      }
      if (do_jac) {
        hipLaunchKernelGGL(sum_gpt_kernel, dim3(cdiv(nclv, 256)), dim3(256), 0, st, nclv, gc, a.jac,
                           bb_tmp + 2 * nclv, (Float)1, first_chunk ? 0 : 1);
        if (last_chunk)
          hipLaunchKernelGGL(reduce_parts_kernel, dim3(cdiv(nclv, 256)), dim3(256), 0, st, nclv, 1,
                             bb_tmp + 2 * nclv, d_jac, piw, imu > 0);
      }
    }
  }
  RTE_CATCH("rte_lw_solver_noscat")
}

void rte_lw_solver_2stream(const int* ncol_, const int* nlay_, const int* ngpt_, const Bool* top_at_1,
                           const Float* tau, const Float* ssa, const Float* g,
                           const Float* lay_source, const Float* lev_source, const Float* sfc_emis,
                           const Float* sfc_src, const Float* inc_flux, Float* flux_up,
                           Float* flux_dn) {
  const int ncol = *ncol_, nlay = *nlay_, ngpt = *ngpt_, nlev = nlay + 1;
  if (ncol <= 0 || nlay <= 0 || ngpt <= 0) return;
  RTE_TRY
  rte::Call c("rte_lw_solver_2stream");
  const size_t ncl = (size_t)ncol * nlay, nclv = (size_t)ncol * nlev, ncg = (size_t)ncol * ngpt;
  Lw2Args a;
  a.ncol = ncol; a.nlay = nlay; a.ngpt = ngpt; a.top_at_1 = *top_at_1; a.lev_gpt1 = g_lw2str_gpt1_levsource != 0;
  a.tau = c.in(tau, ncl * ngpt); a.ssa = c.in(ssa, ncl * ngpt); a.g = c.in(g, ncl * ngpt);
  a.lay_source = c.in(lay_source, ncl * ngpt); a.lev_source = c.in(lev_source, nclv * ngpt);
  a.sfc_emis = c.in(sfc_emis, ncg); a.sfc_src = c.in(sfc_src, ncg); a.inc_flux = c.in(inc_flux, ncg);
  a.flux_up = c.out(flux_up, nclv * ngpt); a.flux_dn = c.out(flux_dn, nclv * ngpt);
  // ------------------------------------------------------------------ production path (nlay <= 64)
  if (nlay <= 96 && !g_lw_force_generic) {
    const int L = nlay <= 64 ? 8 : (nlay <= 72 ? 9 : nlay <= 80 ? 10 : nlay <= 88 ? 11 : 12);  // layers per wave (always 8 waves), see rte_sw_solver_2stream
    const int S = (nlay + L - 1) / L;
    const int col_tiles = cdiv(ncol, 64);
    const int g_per_block = seg_g_per_block(col_tiles, ngpt);
    const int ngroups = (ngpt + g_per_block - 1) / g_per_block;
    Lw2SegArgs q;
    q.ncol = ncol; q.nlay = nlay; q.ngpt = ngpt; q.S = S; q.g_per_block = g_per_block;
    q.top_at_1 = *top_at_1; q.lev_gpt1 = a.lev_gpt1;
    q.tau = a.tau; q.ssa = a.ssa; q.g = a.g; q.lev_source = a.lev_source; q.sfc_emis = a.sfc_emis; q.sfc_src = a.sfc_src;
    q.inc_flux = a.inc_flux; q.flux_up = a.flux_up; q.flux_dn = a.flux_dn;
    rte::ProfScope p("lw_2stream_seg_kernel");
    const size_t lds_bytes = sizeof(Float) * 64 * 8 * (7 + 2);
    if (L == 8) hipLaunchKernelGGL((lw_2stream_seg_kernel<8>), dim3(col_tiles, ngroups), dim3(64 * S), lds_bytes, rte::stream(), q);
    else if (L == 9) hipLaunchKernelGGL((lw_2stream_seg_kernel<9>), dim3(col_tiles, ngroups), dim3(64 * S), lds_bytes, rte::stream(), q);
    else if (L == 10) hipLaunchKernelGGL((lw_2stream_seg_kernel<10>), dim3(col_tiles, ngroups), dim3(64 * S), lds_bytes, rte::stream(), q);
    else if (L == 11) hipLaunchKernelGGL((lw_2stream_seg_kernel<11>), dim3(col_tiles, ngroups), dim3(64 * S), lds_bytes, rte::stream(), q);
    else hipLaunchKernelGGL((lw_2stream_seg_kernel<12>), dim3(col_tiles, ngroups), dim3(64 * S), lds_bytes, rte::stream(), q);
    return;
  }
  constexpr int kLw2WinLay = 88, kLw2WinMax = 8;  // layers per window (11 per wave), windows
  if (nlay > 96 && nlay <= kLw2WinLay * kLw2WinMax && !g_lw_force_generic) {
    // ------------------------------------------------------------------ 97 ... 704 layers: K windows of layers on the segmented
    // kernel, coupled through the adding method as in rte_sw_solver_2stream: bottom to top every window but the first, alone
    // over the albedo and source of the windows below it, gives its albedo and source at its top; top to bottom every window is
    // solved with the flux the window above leaves at its bottom entering and the window below as its surface
    const int K = (nlay + kLw2WinLay - 1) / kLw2WinLay;
    int wn[kLw2WinMax], wpos[kLw2WinMax];
    for (int w = 0, pos = 0; w < K; ++w) { wn[w] = nlay / K + (w < nlay % K ? 1 : 0); wpos[w] = pos; pos += wn[w]; }
    const int col_tiles = cdiv(ncol, 64);
    const int g_per_block = seg_g_per_block(col_tiles, ngpt);
    const int ngroups = (ngpt + g_per_block - 1) / g_per_block;
    Float* side = (Float*)rte::scratch(sizeof(Float) * ncg * (2 * (K - 1) + 2));  // albedo / source at the top of windows 1 ... K - 1, two flux arrays
    Float* const fd_pair = side + ncg * 2 * (K - 1);
    const bool top = *top_at_1;
    auto run = [&](int w, int phase) {  // phase 1: alone (-> albedo, source at its top); 2: the final solve
      const int lay0 = wpos[w], nl = wn[w];
      Lw2SegArgs q{};
      const int L = nl <= 64 ? 8 : (nl <= 72 ? 9 : nl <= 80 ? 10 : nl <= 88 ? 11 : 12);
      q.ncol = ncol; q.nlay = nl; q.ngpt = ngpt; q.S = (nl + L - 1) / L; q.g_per_block = g_per_block; q.top_at_1 = top;
      q.lev_gpt1 = a.lev_gpt1;
      const size_t off = (size_t)ncol * (top ? lay0 : nlay - lay0 - nl);
      q.tau = a.tau + off; q.ssa = a.ssa + off; q.g = a.g + off; q.lev_source = a.lev_source + off;
      q.flux_up = a.flux_up + off; q.flux_dn = a.flux_dn + off;
      q.plane_lay = ncl; q.plane_lev = nclv;
      q.sfc_emis = a.sfc_emis; q.sfc_src = a.sfc_src; q.inc_flux = a.inc_flux;
      if (w + 1 < K) { q.sfc_emis = side + ncg * 2 * w; q.sfc_src = side + ncg * (2 * w + 1); q.sfc_given = true; }  // the window below
      if (phase == 1) { q.out_alb = side + ncg * 2 * (w - 1); q.out_src = side + ncg * (2 * (w - 1) + 1); }
      else {
        if (w > 0) { q.inc_flux = fd_pair + ncg * ((w - 1) & 1); q.skip_first_level = true; }
        if (w + 1 < K) q.out_fd = fd_pair + ncg * (w & 1);
      }
      const size_t lds_bytes = sizeof(Float) * 64 * 8 * (7 + 2);
      rte::ProfScope p("lw_2stream_seg_kernel");
      const dim3 grid(col_tiles, ngroups), blk(64 * q.S);
      if (L == 8) hipLaunchKernelGGL((lw_2stream_seg_kernel<8, true>), grid, blk, lds_bytes, rte::stream(), q);
      else if (L == 9) hipLaunchKernelGGL((lw_2stream_seg_kernel<9, true>), grid, blk, lds_bytes, rte::stream(), q);
      else if (L == 10) hipLaunchKernelGGL((lw_2stream_seg_kernel<10, true>), grid, blk, lds_bytes, rte::stream(), q);
      else if (L == 11) hipLaunchKernelGGL((lw_2stream_seg_kernel<11, true>), grid, blk, lds_bytes, rte::stream(), q);
      else hipLaunchKernelGGL((lw_2stream_seg_kernel<12, true>), grid, blk, lds_bytes, rte::stream(), q);
    };
    for (int w = K - 1; w >= 1; --w) run(w, 1);
    for (int w = 0; w < K; ++w) run(w, 2);
    return;
  }
  const size_t gchunk = pick_gchunk(sizeof(Float) * ncl * 4, ngpt);
  a.ws = (Float*)rte::scratch(sizeof(Float) * ncl * 4 * gchunk);
  rte::ProfScope p("lw_2stream_generic_kernel");
  for (int g0 = 0; g0 < ngpt; g0 += (int)gchunk) {
    const int gc = (int)((size_t)(ngpt - g0) < gchunk ? (size_t)(ngpt - g0) : gchunk);
    a.g_begin = g0;
    hipLaunchKernelGGL(lw_2stream_generic_kernel, dim3(cdiv(ncol, 256), gc), dim3(256), 0, rte::stream(), a);
  }
  RTE_CATCH("rte_lw_solver_2stream")
}

void rte_sw_solver_noscat(const int* ncol_, const int* nlay_, const int* ngpt_, const Bool* top_at_1,
                          const Float* tau, const Float* mu0, const Float* inc_flux_dir,
                          Float* flux_dir) {
  const int ncol = *ncol_, nlay = *nlay_, ngpt = *ngpt_;
  if (ncol <= 0 || nlay <= 0 || ngpt <= 0) return;
  RTE_TRY
  rte::Call c("rte_sw_solver_noscat");
  const size_t ncl = (size_t)ncol * nlay;
  const Float* d_tau = c.in(tau, ncl * ngpt);
  const Float* d_mu0 = c.in(mu0, ncl);
  const Float* d_inc = c.in(inc_flux_dir, (size_t)ncol * ngpt);
  Float* d_dir = c.out(flux_dir, (size_t)ncol * (nlay + 1) * ngpt);
  rte::ProfScope p("sw_noscat_kernel");
  hipLaunchKernelGGL(sw_noscat_kernel, dim3(cdiv(ncol, 256), ngpt), dim3(256), 0, rte::stream(), ncol, nlay, ngpt,
                     (bool)*top_at_1, d_tau, d_mu0, d_inc, d_dir);
  RTE_CATCH("rte_sw_solver_noscat")
}

void rte_sw_solver_2stream(const int* ncol_, const int* nlay_, const int* ngpt_, const Bool* top_at_1,
                           const Float* tau, const Float* ssa, const Float* g, const Float* mu0,
                           const Float* sfc_alb_dir, const Float* sfc_alb_dif,
                           const Float* inc_flux_dir, Float* flux_up, Float* flux_dn,
                           Float* flux_dir, const Bool* has_dif_bc, const Float* inc_flux_dif,
                           const Bool* do_broadband_, Float* broadband_up, Float* broadband_dn,
                           Float* broadband_dir) {
  const int ncol = *ncol_, nlay = *nlay_, ngpt = *ngpt_, nlev = nlay + 1;
  const bool do_broadband = *do_broadband_;
  if (ncol <= 0 || nlay <= 0 || ngpt <= 0) return;
  RTE_TRY
  rte::Call c("rte_sw_solver_2stream");
  const size_t ncl = (size_t)ncol * nlay, nclv = (size_t)ncol * nlev, ncg = (size_t)ncol * ngpt;
  Sw2Args a;
  a.ncol = ncol; a.nlay = nlay; a.ngpt = ngpt; a.top_at_1 = *top_at_1; a.has_dif_bc = *has_dif_bc;
  a.add_dir_to_dn = true;
  a.tau = c.in(tau, ncl * ngpt); a.ssa = c.in(ssa, ncl * ngpt);
  // extension: g == NULL means "g = 0 everywhere" (what rte_hip_gas_optics_sw_2str leaves when it is given no g array).  The
  // broadband segmented kernel at 8 / 9 layers per wave has an instance that reads nothing for it; every other path gets zeros.
  const bool g0_kernel = !g && do_broadband && nlay <= 72 && !g_sw_force_generic && ncl < ((size_t)1 << 29);
  if (g) a.g = c.in(g, ncl * ngpt);
  else if (g0_kernel) a.g = nullptr;
  else {
    Float* z = (Float*)rte::scratch(sizeof(Float) * ncl * ngpt);
    HIP_CHECK(hipMemsetAsync(z, 0, sizeof(Float) * ncl * ngpt, rte::stream()));
    a.g = z;
  }
  a.mu0 = c.in(mu0, ncl);
  a.sfc_alb_dir = c.in(sfc_alb_dir, ncg); a.sfc_alb_dif = c.in(sfc_alb_dif, ncg);
  a.inc_flux_dir = c.in(inc_flux_dir, ncg);
  a.inc_flux_dif = *has_dif_bc ? c.in(inc_flux_dif, ncg) : nullptr;
  Float *d_up = nullptr, *d_dn = nullptr, *d_dir = nullptr, *d_bu = nullptr, *d_bd = nullptr, *d_bdir = nullptr;
  if (do_broadband) {
    d_bu = c.out(broadband_up, nclv); d_bd = c.out(broadband_dn, nclv); d_bdir = c.out(broadband_dir, nclv);
  } else {
    d_up = c.out(flux_up, nclv * ngpt); d_dn = c.out(flux_dn, nclv * ngpt); d_dir = c.out(flux_dir, nclv * ngpt);
  }
  hipStream_t st0 = rte::stream();
  // ------------------------------------------------------------------ production path (broadband, nlay <= 64)
  constexpr int kSwMaxLay = 96;  // 8 waves x up to 12 layers (11: no spill, 12: 7 spilled registers); above: generic kernel
  if (do_broadband && nlay <= kSwMaxLay && !g_sw_force_generic && ncl < ((size_t)1 << 29)) {  // 32-bit in-plane byte offsets
    // layers per wave: 8 up to 64 layers, then 9 (72 layers: the all-sky configuration) or 10 -- always 8 waves
    const int L = nlay <= 64 ? 8 : (nlay <= 72 ? 9 : nlay <= 80 ? 10 : nlay <= 88 ? 11 : 12);
    const int S = (nlay + L - 1) / L;
    const int col_tiles = cdiv(ncol, 64);
    const int g_per_block = seg_g_per_block(col_tiles, ngpt);
    const int ngroups = (ngpt + g_per_block - 1) / g_per_block;
    Sw2SegArgs q;
    q.ncol = ncol; q.nlay = nlay; q.ngpt = ngpt; q.S = S; q.g_per_block = g_per_block;
    q.top_at_1 = *top_at_1; q.has_dif_bc = *has_dif_bc;
    q.tau = a.tau; q.ssa = a.ssa; q.g = a.g; q.mu0 = a.mu0; q.sfc_alb_dir = a.sfc_alb_dir; q.sfc_alb_dif = a.sfc_alb_dif;
    q.inc_flux_dir = a.inc_flux_dir; q.inc_flux_dif = a.inc_flux_dif;
    q.part_up = (Float*)rte::scratch(sizeof(Float) * nclv * ngroups * 3);
    q.part_dn = q.part_up + nclv * ngroups;
    q.part_dir = q.part_dn + nclv * ngroups;
    q.spec_up = q.spec_dn = q.spec_dir = nullptr; q.band_lims = nullptr;
    // composites, flux maps, mu0 (clamped, reciprocal; L == 9: one value), direct-flux (L == 9: and upward-flux) accumulators
    const size_t lds_bytes = sizeof(Float) * 64 * (8 * 8 + 2 * 8 + ((L == 9 || L >= 11) ? 1 : 2) * 8 * L + ((L <= 9 || L >= 11) ? 8 * (L + 1) : 0) + (L == 9 ? 8 * (L + 1) : 0));
    {
      rte::ProfScope p("sw_2stream_seg_kernel");
      // 57 ... 60 layers: segments of 7 and 8 layers, no neutral slots (sw_2stream_seg_mixed_kernel); rte_hip_sw_mixed_segments(0) for A/B
      if (g_sw_mixed && S == 8 && nlay >= 57 && nlay <= 60) {
        const size_t lds_mixed = sw_mixed_lds_bytes<7, 8>();
        if (g0_kernel) hipLaunchKernelGGL((sw_2stream_seg_mixed_kernel<7, 8, true>), dim3(col_tiles, ngroups), dim3(64 * 8), lds_mixed, st0, q);
        else hipLaunchKernelGGL((sw_2stream_seg_mixed_kernel<7, 8, false>), dim3(col_tiles, ngroups), dim3(64 * 8), lds_mixed, st0, q);
      } else
      if (g0_kernel && L == 8) hipLaunchKernelGGL((sw_2stream_seg_kernel<8, false, false, true>), dim3(col_tiles, ngroups), dim3(64 * S), lds_bytes, st0, q);
      else if (g0_kernel) hipLaunchKernelGGL((sw_2stream_seg_kernel<9, false, false, true>), dim3(col_tiles, ngroups), dim3(64 * S), lds_bytes, st0, q);
      else if (L == 8) hipLaunchKernelGGL((sw_2stream_seg_kernel<8>), dim3(col_tiles, ngroups), dim3(64 * S), lds_bytes, st0, q);
      else if (L == 9) hipLaunchKernelGGL((sw_2stream_seg_kernel<9>), dim3(col_tiles, ngroups), dim3(64 * S), lds_bytes, st0, q);
      else if (L == 10) hipLaunchKernelGGL((sw_2stream_seg_kernel<10>), dim3(col_tiles, ngroups), dim3(64 * S), lds_bytes, st0, q);
      else if (L == 11) hipLaunchKernelGGL((sw_2stream_seg_kernel<11>), dim3(col_tiles, ngroups), dim3(64 * S), lds_bytes, st0, q);
      else hipLaunchKernelGGL((sw_2stream_seg_kernel<12>), dim3(col_tiles, ngroups), dim3(64 * S), lds_bytes, st0, q);
    }
    rte::ProfScope p("sw_reduce_parts");
    hipLaunchKernelGGL(reduce_parts_kernel, dim3(cdiv(nclv, 256)), dim3(256), 0, st0, nclv, ngroups, q.part_up, d_bu, (Float)1, false);
    hipLaunchKernelGGL(reduce_parts_kernel, dim3(cdiv(nclv, 256)), dim3(256), 0, st0, nclv, ngroups, q.part_dn, d_bd, (Float)1, false);
    hipLaunchKernelGGL(reduce_parts_kernel, dim3(cdiv(nclv, 256)), dim3(256), 0, st0, nclv, ngroups, q.part_dir, d_bdir, (Float)1, false);
    return;
  }
  constexpr int kSwWinLay = 88, kSwWinMax = 8;  // layers per window (11 per wave: the widest variant without spills), windows
  if (nlay > kSwMaxLay && nlay <= kSwWinLay * kSwWinMax && !g_sw_force_generic && nclv < ((size_t)1 << 29)) {
    // ------------------------------------------------------------------ 97 ... 704 layers: the column as K windows of layers,
    // each on the segmented kernel (WIN).  The adding method composes: bottom to top (phase 1) every window but the first,
    // alone under a unit beam, over the albedo and source of the windows below it (the last: the surface), gives the albedo
    // and the source per unit of beam it presents at its top; top to bottom every window is then solved with the diffuse and
    // direct flux the window above leaves at its bottom as its top boundary (the first: the column's) and the window below as
    // its "surface".  Every window but the first is evaluated twice.
    const int K = (nlay + kSwWinLay - 1) / kSwWinLay;
    int wn[kSwWinMax], wpos[kSwWinMax];  // layers of window w, its first layer's position from the top
    for (int w = 0, pos = 0; w < K; ++w) { wn[w] = nlay / K + (w < nlay % K ? 1 : 0); wpos[w] = pos; pos += wn[w]; }
    const int col_tiles = cdiv(ncol, 64);
    const int g_per_block = seg_g_per_block(col_tiles, ngpt);
    const int ngroups = (ngpt + g_per_block - 1) / g_per_block;
    Float* parts = do_broadband ? (Float*)rte::scratch(sizeof(Float) * nclv * ngroups * 3) : nullptr;
    // (column, g-point) arrays: albedo / source at the top of windows 1 ... K - 1, diffuse / direct flux at a window's bottom (two
    // pairs: a window reads the pair of the window above and writes its own)
    Float* side = (Float*)rte::scratch(sizeof(Float) * ncg * (2 * (K - 1) + 4));
    Float* const flux_pair = side + ncg * 2 * (K - 1);
    const bool top = *top_at_1;
    // phase 1: alone under a unit beam (-> albedo, source at its top); phase 2: the final solve.  Window w's "surface" is window
    // w + 1 (the last: the surface), its top boundary what window w - 1 left (the first: the column's)
    auto run = [&](int w, int phase) {
      const int lay0 = wpos[w], nl = wn[w];
      Sw2SegArgs q{};
      // (a short lower part gets 4 layers per wave: all eight waves -- all four SIMDs -- work instead of two or three)
      const int L = nl <= 32 ? 4 : nl <= 64 ? 8 : (nl <= 72 ? 9 : nl <= 80 ? 10 : nl <= 88 ? 11 : 12);
      q.ncol = ncol; q.nlay = nl; q.ngpt = ngpt; q.S = (nl + L - 1) / L; q.g_per_block = g_per_block; q.top_at_1 = top;
      // first row of the window: layer position lay0 from the top (layers), the same for levels
      const size_t off = (size_t)ncol * (top ? lay0 : nlay - lay0 - nl);
      q.tau = a.tau + off; q.ssa = a.ssa + off; q.g = a.g + off; q.mu0 = a.mu0 + off;
      q.plane_lay = ncl; q.plane_lev = nclv; q.plane_part = nclv;
      if (do_broadband) { q.part_up = parts + off; q.part_dn = q.part_up + nclv * ngroups; q.part_dir = q.part_dn + nclv * ngroups; }
      else { q.spec_up = d_up + off; q.spec_dn = d_dn + off; q.spec_dir = d_dir + off; }  // (spectral: phase 1 writes B's levels too, phases 2 and 3 overwrite them)
      q.sfc_alb_dir = a.sfc_alb_dir; q.sfc_alb_dif = a.sfc_alb_dif; q.inc_flux_dir = a.inc_flux_dir; q.inc_flux_dif = a.inc_flux_dif;
      q.has_dif_bc = *has_dif_bc;
      if (w + 1 < K) {  // the window below as the surface
        q.sfc_alb_dif = side + ncg * 2 * w; q.sfc_alb_dir = side + ncg * (2 * w + 1); q.sfc_src_given = true;
      }
      if (phase == 1) {
        q.beam_mode = 1; q.has_dif_bc = false; q.out_alb = side + ncg * 2 * (w - 1); q.out_src = side + ncg * (2 * (w - 1) + 1);
      } else {
        if (w > 0) {  // what the window above left at its bottom
          Float* in = flux_pair + ncg * 2 * ((w - 1) & 1);
          q.beam_mode = 2; q.inc_flux_dif = in; q.inc_flux_dir = in + ncg; q.has_dif_bc = true; q.skip_first_level = true;
        }
        if (w + 1 < K) { Float* out = flux_pair + ncg * 2 * (w & 1); q.out_fd = out; q.out_dir = out + ncg; }
      }
      const size_t lds_bytes = do_broadband
          ? sizeof(Float) * 64 * (8 * 8 + 2 * 8 + ((L == 9 || L >= 11) ? 1 : 2) * 8 * L + ((L <= 9 || L >= 11) ? 8 * (L + 1) : 0) + (L == 9 ? 8 * (L + 1) : 0))
          : sizeof(Float) * 64 * (8 * 8 + 2 * 8 + 2 * 8 * L);
      rte::ProfScope p(do_broadband ? "sw_2stream_seg_kernel" : "sw_2stream_seg_spectral_kernel");
      const dim3 grid(col_tiles, ngroups), blk(64 * q.S);
      if (!do_broadband) {
        if (L == 4) hipLaunchKernelGGL((sw_2stream_seg_kernel<4, true, true>), grid, blk, lds_bytes, st0, q);
        else if (L == 8) hipLaunchKernelGGL((sw_2stream_seg_kernel<8, true, true>), grid, blk, lds_bytes, st0, q);
        else if (L == 9) hipLaunchKernelGGL((sw_2stream_seg_kernel<9, true, true>), grid, blk, lds_bytes, st0, q);
        else if (L == 10) hipLaunchKernelGGL((sw_2stream_seg_kernel<10, true, true>), grid, blk, lds_bytes, st0, q);
        else if (L == 11) hipLaunchKernelGGL((sw_2stream_seg_kernel<11, true, true>), grid, blk, lds_bytes, st0, q);
        else hipLaunchKernelGGL((sw_2stream_seg_kernel<12, true, true>), grid, blk, lds_bytes, st0, q);
        return;
      }
      if (L == 4) hipLaunchKernelGGL((sw_2stream_seg_kernel<4, false, true>), grid, blk, lds_bytes, st0, q);
      else if (L == 8) hipLaunchKernelGGL((sw_2stream_seg_kernel<8, false, true>), grid, blk, lds_bytes, st0, q);
      else if (L == 9) hipLaunchKernelGGL((sw_2stream_seg_kernel<9, false, true>), grid, blk, lds_bytes, st0, q);
      else if (L == 10) hipLaunchKernelGGL((sw_2stream_seg_kernel<10, false, true>), grid, blk, lds_bytes, st0, q);
      else if (L == 11) hipLaunchKernelGGL((sw_2stream_seg_kernel<11, false, true>), grid, blk, lds_bytes, st0, q);
      else hipLaunchKernelGGL((sw_2stream_seg_kernel<12, false, true>), grid, blk, lds_bytes, st0, q);
    };
    for (int w = K - 1; w >= 1; --w) run(w, 1);
    for (int w = 0; w < K; ++w) run(w, 2);
    if (!do_broadband) return;
    rte::ProfScope p("sw_reduce_parts");
    hipLaunchKernelGGL(reduce_parts_kernel, dim3(cdiv(nclv, 256)), dim3(256), 0, st0, nclv, ngroups, parts, d_bu, (Float)1, false);
    hipLaunchKernelGGL(reduce_parts_kernel, dim3(cdiv(nclv, 256)), dim3(256), 0, st0, nclv, ngroups, parts + nclv * ngroups, d_bd, (Float)1, false);
    hipLaunchKernelGGL(reduce_parts_kernel, dim3(cdiv(nclv, 256)), dim3(256), 0, st0, nclv, ngroups, parts + 2 * nclv * ngroups, d_bdir, (Float)1, false);
    return;
  }
  if (!do_broadband && nlay <= kSwMaxLay && !g_sw_force_generic && nclv < ((size_t)1 << 29)) {
    // ---------------------------------------------------------------- production path, spectral output: the same
    // segmented kernel, every wave storing the three fluxes of the levels it owns per g-point
    const int L = nlay <= 64 ? 8 : (nlay <= 72 ? 9 : nlay <= 80 ? 10 : nlay <= 88 ? 11 : 12);
    const int S = (nlay + L - 1) / L;
    const int col_tiles = cdiv(ncol, 64);
    const int g_per_block = seg_g_per_block(col_tiles, ngpt);
    const int ngroups = (ngpt + g_per_block - 1) / g_per_block;
    Sw2SegArgs q;
    q.ncol = ncol; q.nlay = nlay; q.ngpt = ngpt; q.S = S; q.g_per_block = g_per_block;
    q.top_at_1 = *top_at_1; q.has_dif_bc = *has_dif_bc;
    q.tau = a.tau; q.ssa = a.ssa; q.g = a.g; q.mu0 = a.mu0; q.sfc_alb_dir = a.sfc_alb_dir; q.sfc_alb_dif = a.sfc_alb_dif;
    q.inc_flux_dir = a.inc_flux_dir; q.inc_flux_dif = a.inc_flux_dif;
    q.part_up = q.part_dn = q.part_dir = nullptr;
    q.spec_up = d_up; q.spec_dn = d_dn; q.spec_dir = d_dir; q.band_lims = nullptr;
    const size_t lds_bytes = sizeof(Float) * 64 * (8 * 8 + 2 * 8 + 2 * 8 * L);  // composites, flux maps, mu0 (clamped, reciprocal)
    rte::ProfScope p("sw_2stream_seg_spectral_kernel");
    if (L == 8) hipLaunchKernelGGL((sw_2stream_seg_kernel<8, true>), dim3(col_tiles, ngroups), dim3(64 * S), lds_bytes, st0, q);
    else if (L == 9) hipLaunchKernelGGL((sw_2stream_seg_kernel<9, true>), dim3(col_tiles, ngroups), dim3(64 * S), lds_bytes, st0, q);
    else if (L == 10) hipLaunchKernelGGL((sw_2stream_seg_kernel<10, true>), dim3(col_tiles, ngroups), dim3(64 * S), lds_bytes, st0, q);
    else if (L == 11) hipLaunchKernelGGL((sw_2stream_seg_kernel<11, true>), dim3(col_tiles, ngroups), dim3(64 * S), lds_bytes, st0, q);
    else hipLaunchKernelGGL((sw_2stream_seg_kernel<12, true>), dim3(col_tiles, ngroups), dim3(64 * S), lds_bytes, st0, q);
    return;
  }
  // ------------------------------------------------------------------ generic path
  const size_t per_g = sizeof(Float) * (ncl * 5 + (do_broadband ? nclv * 3 : 0));
  const size_t gchunk = pick_gchunk(per_g, ngpt);
  a.ws = (Float*)rte::scratch(sizeof(Float) * ncl * 5 * gchunk);
  Float* slabs = do_broadband ? (Float*)rte::scratch(sizeof(Float) * nclv * 3 * gchunk) : nullptr;
  hipStream_t st = rte::stream();
  for (int g0 = 0; g0 < ngpt; g0 += (int)gchunk) {
    const int gc = (int)((size_t)(ngpt - g0) < gchunk ? (size_t)(ngpt - g0) : gchunk);
    a.g_begin = g0;
    if (do_broadband) {
      a.up = slabs; a.dn = slabs + nclv * gchunk; a.dir = slabs + 2 * nclv * gchunk;
    } else {
      a.up = d_up + nclv * g0; a.dn = d_dn + nclv * g0; a.dir = d_dir + nclv * g0;
    }
    {
      rte::ProfScope p("sw_2stream_generic_kernel");
      hipLaunchKernelGGL(sw_2stream_generic_kernel, dim3(cdiv(ncol, 256), gc), dim3(256), 0, st, a);
    }
    if (do_broadband) {
      rte::ProfScope p("sw_generic_reduce");
      const int mode = g0 == 0 ? 0 : 1;
      hipLaunchKernelGGL(sum_gpt_kernel, dim3(cdiv(nclv, 256)), dim3(256), 0, st, nclv, gc, a.up, d_bu, (Float)1, mode);
      hipLaunchKernelGGL(sum_gpt_kernel, dim3(cdiv(nclv, 256)), dim3(256), 0, st, nclv, gc, a.dn, d_bd, (Float)1, mode);
      hipLaunchKernelGGL(sum_gpt_kernel, dim3(cdiv(nclv, 256)), dim3(256), 0, st, nclv, gc, a.dir, d_bdir, (Float)1, mode);
    }
  }
  RTE_CATCH("rte_sw_solver_2stream")
}

// ---- by-band fluxes straight from the segmented kernels (extensions; what rte_lw / rte_sw + ty_fluxes_byband%reduce produce
//      through the spectral arrays and rte_sum_byband, rte/extensions/mo_fluxes_byband.F90:46-137, without those arrays)
int rte_hip_lw_solver_noscat_byband(int ncol, int nlay, int ngpt, int nbnd, int top_at_1, int nmus, const Float* Ds,
                                    const Float* weights, const int* band_lims_gpt, const Float* tau, const Float* lay_source,
                                    const Float* lev_source, const Float* sfc_emis, const Float* sfc_src, const Float* inc_flux,
                                    Float* byband_up, Float* byband_dn) {
  if (ncol <= 0 || nlay <= 0 || ngpt <= 0 || nbnd <= 0 || nmus <= 0) return 0;
  const int nlev = nlay + 1;
  const size_t ncl = (size_t)ncol * nlay, nclv = (size_t)ncol * nlev, ncg = (size_t)ncol * ngpt;
  if (nlay > 80 || nclv >= ((size_t)1 << 29)) return -2;  // (callers fall back to spectral output + rte_sum_byband)
  RTE_TRY
  rte::Call c("rte_hip_lw_solver_noscat_byband");
  const Float* w_h = c.host(weights, (size_t)nmus);
  const Float* d_Ds = c.in(Ds, ncg * nmus);
  const int* d_bl = c.in(band_lims_gpt, (size_t)2 * nbnd);
  const Float *d_tau = c.in(tau, ncl * ngpt), *d_lay = c.in(lay_source, ncl * ngpt), *d_lev = c.in(lev_source, nclv * ngpt);
  const Float *d_emis = c.in(sfc_emis, ncg), *d_sfc = c.in(sfc_src, ncg), *d_inc = c.in(inc_flux, ncg);
  Float *d_up = c.out(byband_up, nclv * nbnd), *d_dn = c.out(byband_dn, nclv * nbnd);
  hipStream_t st = rte::stream();
  const int L = nlay <= 64 ? 8 : nlay <= 72 ? 9 : 10;
  const int S = (nlay + L - 1) / L;
  const size_t lds_bytes = sizeof(Float) * (2 * 3 * 8 * 64);
  rte::ProfScope p("lw_noscat_seg_byband_kernel");
  for (int imu = 0; imu < nmus; ++imu) {
#define RTE_LAUNCH_SEGB(LL)                                                                                              \
  hipLaunchKernelGGL((lw_noscat_seg_kernel<LL, false, false, false, true>), dim3(cdiv(ncol, 64), nbnd), dim3(64 * S), lds_bytes, st, \
                     ncol, nlay, ngpt, S, 0, top_at_1 != 0, w_h[imu], d_Ds + ncg * imu, d_tau, d_lay, d_lev, d_emis, d_sfc, \
                     d_inc, (const Float*)nullptr, d_up, d_dn, (Float*)nullptr, (Float*)nullptr, (Float*)nullptr, imu > 0, d_bl)
    if (L == 8) RTE_LAUNCH_SEGB(8); else if (L == 9) RTE_LAUNCH_SEGB(9); else RTE_LAUNCH_SEGB(10);
#undef RTE_LAUNCH_SEGB
  }
  return 0;
  RTE_CATCH("rte_hip_lw_solver_noscat_byband")
  return -1;
}

// ---- rte_lw_solver_noscat on FACTORED sources (extension): what rte_hip_compute_Planck_source_factored leaves -- the Planck
//      fraction per g-point and the Planck function per band -- instead of lay_source / lev_source (ncol, nlay[+1], ngpt).  Broadband
//      output, no rescaling, [Jacobian], nlay <= 80: lw_noscat_seg_kernel<..., FACT> forms the sources per g-point with the operations
//      of compute_Planck_source (results bit-identical to the two ABI calls; 26 GB less written and 13 GB less read at 1e5 x 60 x 256).
//      Returns -2 for what it does not cover: the caller expands the sources (rte_hip_expand_factored_sources) and calls the ABI.
int rte_hip_lw_solver_noscat_factored(int ncol, int nlay, int ngpt, int nbnd, int top_at_1, int nmus, const Float* Ds,
                                      const Float* weights, const int* band_lims_gpt, const Float* tau, const Float* pfrac,
                                      const Float* planck_lay, const Float* planck_lev, const Float* sfc_emis, const Float* sfc_src,
                                      const Float* inc_flux, Float* broadband_up, Float* broadband_dn, int do_jac,
                                      const Float* sfc_srcJac, Float* flux_upJac) {
  if (ncol <= 0 || nlay <= 0 || ngpt <= 0 || nbnd <= 0 || nmus <= 0) return 0;
  const int nlev = nlay + 1;
  const size_t ncl = (size_t)ncol * nlay, nclv = (size_t)ncol * nlev, ncg = (size_t)ncol * ngpt;
  if (nlay > 80 || nclv >= ((size_t)1 << 29) || g_lw_force_generic) return -2;
  RTE_TRY
  rte::Call c("rte_hip_lw_solver_noscat_factored");
  const Float* w_h = c.host(weights, (size_t)nmus);
  const Float* d_Ds = c.in(Ds, ncg * nmus);
  const int* d_bl = c.in(band_lims_gpt, (size_t)2 * nbnd);
  const Float *d_tau = c.in(tau, ncl * ngpt), *d_pf = c.in(pfrac, ncl * ngpt);
  const Float *d_ply = c.in(planck_lay, ncl * nbnd), *d_plv = c.in(planck_lev, nclv * nbnd);
  const Float *d_emis = c.in(sfc_emis, ncg), *d_sfc = c.in(sfc_src, ncg), *d_inc = c.in(inc_flux, ncg);
  const Float* d_srcJac = do_jac ? c.in(sfc_srcJac, ncg) : nullptr;
  Float *d_bb_up = c.out(broadband_up, nclv), *d_bb_dn = c.out(broadband_dn, nclv);
  Float* d_jac = do_jac ? c.out(flux_upJac, nclv) : nullptr;
  hipStream_t st = rte::stream();
  const int L = nlay <= 64 ? 8 : nlay <= 72 ? 9 : 10;
  const int S = (nlay + L - 1) / L;
  const int col_tiles = cdiv(ncol, 64);
  const int g_per_block = seg_g_per_block(col_tiles, ngpt);
  const int ngroups = (ngpt + g_per_block - 1) / g_per_block;
  Float* part_up = (Float*)rte::scratch(sizeof(Float) * nclv * ngroups * (do_jac ? 3 : 2));
  Float* part_dn = part_up + nclv * ngroups;
  Float* part_jac = do_jac ? part_dn + nclv * ngroups : nullptr;
  // (the 10-layer Jacobian variant parks the band's Planck functions in LDS beside the composites; the shared surface arrays only where both fit in 160 KB)
  const size_t lds_plk = (L <= 9 || !do_jac) ? 0 : 8 * (2 * L + 1) * 64, lds_sfc = 2 * 16 * (do_jac ? 5 : 4) * 64;  // (PLKREG in the kernel)
  const bool sfclds = g_lw_sfc_lds && S == 8 && (L == 8 || !do_jac) && sizeof(Float) * (2 * 3 * 8 * 64 + lds_sfc + lds_plk) <= 160 * 1024;
  const size_t lds_bytes = sizeof(Float) * (2 * 3 * 8 * 64 + (sfclds ? lds_sfc : 0) + lds_plk);
  for (int imu = 0; imu < nmus; ++imu) {
    {
      rte::ProfScope p("lw_noscat_seg_factored_kernel");
#define RTE_LAUNCH_SEGF(LL, JJ, SS)                                                                                       \
  hipLaunchKernelGGL((lw_noscat_seg_kernel<LL, JJ, SS, false, false, true>), dim3(col_tiles, ngroups), dim3(64 * S), lds_bytes, st, \
                     ncol, nlay, ngpt, S, g_per_block, top_at_1 != 0, w_h[imu], d_Ds + ncg * imu, d_tau, d_pf, d_plv, d_emis, \
                     d_sfc, d_inc, d_srcJac, part_up, part_dn, part_jac, (Float*)nullptr, (Float*)nullptr, false, d_bl, d_ply)
#define RTE_LAUNCH_SEGF_(LL)                                                                     \
  do {                                                                                           \
    if (sfclds) { if (do_jac) RTE_LAUNCH_SEGF(LL, true, true); else RTE_LAUNCH_SEGF(LL, false, true); } \
    else        { if (do_jac) RTE_LAUNCH_SEGF(LL, true, false); else RTE_LAUNCH_SEGF(LL, false, false); } \
  } while (0)
      // (the same segments as rte_lw_solver_noscat on the same shape: the fluxes stay bit-identical to the plain chain's)
      if (g_lw_mixed && sfclds && S == 8 && nlay >= 57 && nlay <= 60) {
        if (do_jac)
          hipLaunchKernelGGL((lw_noscat_seg_mixed_kernel<7, 8, true, true, true>), dim3(col_tiles, ngroups), dim3(64 * 8), lds_bytes, st, ncol, nlay,
                             ngpt, S, g_per_block, top_at_1 != 0, w_h[imu], d_Ds + ncg * imu, d_tau, d_pf, d_plv, d_emis, d_sfc, d_inc, d_srcJac,
                             part_up, part_dn, part_jac, d_bl, d_ply);
        else
          hipLaunchKernelGGL((lw_noscat_seg_mixed_kernel<7, 8, false, true, true>), dim3(col_tiles, ngroups), dim3(64 * 8), lds_bytes, st, ncol, nlay,
                             ngpt, S, g_per_block, top_at_1 != 0, w_h[imu], d_Ds + ncg * imu, d_tau, d_pf, d_plv, d_emis, d_sfc, d_inc, d_srcJac,
                             part_up, part_dn, part_jac, d_bl, d_ply);
      } else
      if (L == 8) RTE_LAUNCH_SEGF_(8); else if (L == 9) RTE_LAUNCH_SEGF_(9); else RTE_LAUNCH_SEGF_(10);
#undef RTE_LAUNCH_SEGF_
#undef RTE_LAUNCH_SEGF
    }
    rte::ProfScope p("lw_reduce_parts");
    const Float piw = (Float)3.14159265358979323846264338327950288 * w_h[imu];
    hipLaunchKernelGGL(reduce_parts_kernel, dim3(cdiv(nclv, 256)), dim3(256), 0, st, nclv, ngroups, part_up, d_bb_up, piw, imu > 0);
    hipLaunchKernelGGL(reduce_parts_kernel, dim3(cdiv(nclv, 256)), dim3(256), 0, st, nclv, ngroups, part_dn, d_bb_dn, piw, imu > 0);
    if (do_jac)
      hipLaunchKernelGGL(reduce_parts_kernel, dim3(cdiv(nclv, 256)), dim3(256), 0, st, nclv, ngroups, part_jac, d_jac, piw, imu > 0);
  }
  return 0;
  RTE_CATCH("rte_hip_lw_solver_noscat_factored")
  return -1;
}

int rte_hip_sw_solver_2stream_byband(int ncol, int nlay, int ngpt, int nbnd, int top_at_1, const int* band_lims_gpt,
                                     const Float* tau, const Float* ssa, const Float* g, const Float* mu0,
                                     const Float* sfc_alb_dir, const Float* sfc_alb_dif, const Float* inc_flux_dir,
                                     int has_dif_bc, const Float* inc_flux_dif, Float* byband_up, Float* byband_dn,
                                     Float* byband_dir) {
  if (ncol <= 0 || nlay <= 0 || ngpt <= 0 || nbnd <= 0) return 0;
  const size_t ncl = (size_t)ncol * nlay, nclv = (size_t)ncol * (nlay + 1), ncg = (size_t)ncol * ngpt;
  if (nlay > 96 || nclv >= ((size_t)1 << 29)) return -2;
  RTE_TRY
  rte::Call c("rte_hip_sw_solver_2stream_byband");
  Sw2SegArgs q;
  const int L = nlay <= 64 ? 8 : (nlay <= 72 ? 9 : nlay <= 80 ? 10 : nlay <= 88 ? 11 : 12);
  q.ncol = ncol; q.nlay = nlay; q.ngpt = ngpt; q.S = (nlay + L - 1) / L; q.g_per_block = 0;
  q.top_at_1 = top_at_1 != 0; q.has_dif_bc = has_dif_bc != 0;
  q.band_lims = c.in(band_lims_gpt, (size_t)2 * nbnd);
  q.tau = c.in(tau, ncl * ngpt); q.ssa = c.in(ssa, ncl * ngpt); q.mu0 = c.in(mu0, ncl);
  if (g) q.g = c.in(g, ncl * ngpt);
  else {  // g == NULL: g = 0 everywhere, as in rte_sw_solver_2stream
    Float* z = (Float*)rte::scratch(sizeof(Float) * ncl * ngpt);
    HIP_CHECK(hipMemsetAsync(z, 0, sizeof(Float) * ncl * ngpt, rte::stream()));
    q.g = z;
  }
  q.sfc_alb_dir = c.in(sfc_alb_dir, ncg); q.sfc_alb_dif = c.in(sfc_alb_dif, ncg); q.inc_flux_dir = c.in(inc_flux_dir, ncg);
  q.inc_flux_dif = has_dif_bc ? c.in(inc_flux_dif, ncg) : nullptr;
  q.part_up = c.out(byband_up, nclv * nbnd); q.part_dn = c.out(byband_dn, nclv * nbnd); q.part_dir = c.out(byband_dir, nclv * nbnd);
  q.spec_up = q.spec_dn = q.spec_dir = nullptr;
  const size_t lds_bytes = sizeof(Float) * 64 * (8 * 8 + 2 * 8 + ((L == 9 || L >= 11) ? 1 : 2) * 8 * L + ((L <= 9 || L >= 11) ? 8 * (L + 1) : 0) + (L == 9 ? 8 * (L + 1) : 0));
  rte::ProfScope p("sw_2stream_seg_byband_kernel");
  hipStream_t st = rte::stream();
  const dim3 grid(cdiv(ncol, 64), nbnd), blk(64 * q.S);
  if (L == 8) hipLaunchKernelGGL((sw_2stream_seg_kernel<8>), grid, blk, lds_bytes, st, q);
  else if (L == 9) hipLaunchKernelGGL((sw_2stream_seg_kernel<9>), grid, blk, lds_bytes, st, q);
  else if (L == 10) hipLaunchKernelGGL((sw_2stream_seg_kernel<10>), grid, blk, lds_bytes, st, q);
  else if (L == 11) hipLaunchKernelGGL((sw_2stream_seg_kernel<11>), grid, blk, lds_bytes, st, q);
  else hipLaunchKernelGGL((sw_2stream_seg_kernel<12>), grid, blk, lds_bytes, st, q);
  return 0;
  RTE_CATCH("rte_hip_sw_solver_2stream_byband")
  return -1;
}

}  // extern "C"


