// gas_optics.hip -- RRTMGP gas-optics kernels for gfx950 (MI355X), hand-written HIP.
//
// Entry points (C ABI = the reference's bind(C) interface, include/rte_rrtmgp_kernels.h):
//   rrtmgp_interpolation, rrtmgp_compute_tau_absorption, rrtmgp_compute_tau_rayleigh,
//   rrtmgp_compute_Planck_source
// Semantics follow the reference `default` CPU kernels
// (rrtmgp/kernels/mo_gas_optics_rrtmgp_kernels.F90), NOT its OpenACC variant; the loop structure
// is this library's own:
//   * lanes of a wavefront = 64 consecutive columns (unit stride on every (ncol,...) array);
//   * blockIdx.y = layer, blockIdx.z = flavor (interpolation) or band (tau / Rayleigh);
//   * each thread owns one (column, layer, band) and walks the band's g-points in register
//     chunks, so tau is read-modify-written exactly once per call although major, lower-minor
//     and upper-minor contributions are all accumulated (same summation order as the reference:
//     major, then lower minors in interval order, then upper minors);
//   * Planck: each thread owns one (column, band) and walks the layers sequentially so the
//     geometric mean of adjacent layers' Planck fractions needs no second gather.
#include <math.h>

#include <vector>

#include <type_traits>

#include <atomic>

#include "common.h"

namespace {

using rte::cdiv;
using rte::store_stream;

constexpr int GC = 16;  // g-points held in registers per chunk

// OR over the 64 lanes of a wave, result returned as a wave-uniform value: inclusive scan inside each row of 16
// lanes (row_shr 1, 2, 4, 8), then row 0 -> row 1 and row 2 -> row 3 (row_bcast:15), then rows 0-1 -> rows 2-3
// (row_bcast:31); lane 63 holds the total
__device__ __forceinline__ unsigned wave_or(unsigned v) {
  int x = (int)v;
  x |= __builtin_amdgcn_update_dpp(0, x, 0x111, 0xf, 0xf, false);  // row_shr:1
  x |= __builtin_amdgcn_update_dpp(0, x, 0x112, 0xf, 0xf, false);  // row_shr:2
  x |= __builtin_amdgcn_update_dpp(0, x, 0x114, 0xf, 0xf, false);  // row_shr:4
  x |= __builtin_amdgcn_update_dpp(0, x, 0x118, 0xf, 0xf, false);  // row_shr:8
  x |= __builtin_amdgcn_update_dpp(0, x, 0x142, 0xa, 0xf, false);  // row_bcast:15 into rows 1 and 3
  x |= __builtin_amdgcn_update_dpp(0, x, 0x143, 0xc, 0xf, false);  // row_bcast:31 into rows 2 and 3
  return (unsigned)__builtin_amdgcn_readlane(x, 63);
}

constexpr int MAXFLAV = 32;

// -------------------------------------------------------------------------------------------
// interpolation: reference mo_gas_optics_rrtmgp_kernels.F90:37-170
// -------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256)
interpolation_kernel(int ncol, int nlay, int ngas, int nflav, int neta, int npres, int ntemp,
                     const int* __restrict__ flavor, const Float* __restrict__ temp_ref,
                     const Float* __restrict__ press_ref_log, Float press_ref_log_delta_inv, Float temp_ref_min,
                     Float temp_ref_delta, Float temp_ref_delta_inv, Float press_ref_trop,
                     const Float* __restrict__ vmr_ref, const Float* __restrict__ play,
                     const Float* __restrict__ tlay, const Float* __restrict__ col_gas,
                     int* __restrict__ jtemp, Float* __restrict__ fmajor, Float* __restrict__ fminor,
                     Float* __restrict__ col_mix, Bool* __restrict__ tropo, int* __restrict__ jeta,
                     int* __restrict__ jpress, unsigned* __restrict__ masks, int cg_lds) {
  // block = (256 columns, one layer); the flavors are walked INSIDE the block: pressure / temperature terms (one log)
  // are formed once per (column, layer), and play, tlay and the column amounts are read once instead of once per flavor
  // (as a grid dimension the flavors' blocks ran far apart: 1.9 GB of reads for 0.5 GB of inputs)
  const int icol_raw = blockIdx.x * blockDim.x + threadIdx.x;
  const int ilay = blockIdx.y;
  const bool in_range = icol_raw < ncol;
  const int icol = in_range ? icol_raw : ncol - 1;  // ragged last block: compute on a valid column, store nothing
  const size_t ncl = (size_t)ncol * nlay;
  const size_t cl = icol + (size_t)ncol * ilay;
  const Float T = tlay[cl], P = play[cl];
  // :106-108 (INT truncates toward zero; ftemp uses the unclamped index)
  const int jtemp_ = (int)((T - (temp_ref_min - temp_ref_delta)) * temp_ref_delta_inv);
  const int jt = min(ntemp - 1, max(1, jtemp_));
  const int jt_read = min(ntemp, max(1, jtemp_));  // reference reads out of bounds outside the table
  const Float ftemp = (T - temp_ref[jt_read - 1]) * temp_ref_delta_inv;
  // :111-114
  const Float locpress = (Float)1 + (log(P) - press_ref_log[0]) * press_ref_log_delta_inv;
  const Float jpress_aint = fmin((Float)(npres - 1), fmax((Float)1, trunc(locpress)));
  const Float fpress = locpress - jpress_aint;
  const bool trop = P > press_ref_trop;  // :117
  if (in_range) {
    jtemp[cl] = jt;
    jpress[cl] = (int)jpress_aint;
    tropo[cl] = trop;
  }
  const int itropo = trop ? 0 : 1;
  // masks != nullptr: the block also leaves bit masks of the LUT rows its columns touch (temperature, pressure, regime,
  // and per flavor and regime the eta rows) -- what tile_geom2_kernel would otherwise derive by reading jtemp, jpress,
  // tropo and all of jeta again in the compute_tau_absorption call that follows (InterpMasks below)
  __shared__ unsigned s_mask[4 + 2 * MAXFLAV];
  const int mask_w = 4 + 2 * nflav;
  if (masks) {
    if ((int)threadIdx.x < mask_w) s_mask[threadIdx.x] = 0;
    __syncthreads();
    const int jp = (int)jpress_aint + itropo + 1;
    const unsigned long long pm = in_range ? (3ull << (jp - 1)) : 0ull;
    const unsigned tm = wave_or(in_range ? (3u << jt) : 0u);
    const unsigned p0 = wave_or((unsigned)pm), p1 = wave_or((unsigned)(pm >> 32));
    const unsigned rg = wave_or(in_range ? (trop ? 1u : 2u) : 0u);
    if ((threadIdx.x & 63) == 0) { atomicOr(&s_mask[0], tm); atomicOr(&s_mask[1], p0); atomicOr(&s_mask[2], p1); atomicOr(&s_mask[3], rg); }
  }
  // this column's amounts of every gas, parked in LDS (lane-private slots; the flavor's two gases are block-uniform indices)
  // (tables with many gases -- the real files have ~20 -- would need more LDS than a block may have beside the transpose
  //  buffers: `cg_lds` == 0 then reads the two amounts of a flavor from global memory, L2-resident after the first touch)
  extern __shared__ Float s_cg[];  // [ngas + 1][256]
  const int t = threadIdx.x;
  if (cg_lds)
    for (int ig = 0; ig <= ngas; ++ig) s_cg[ig * 256 + t] = col_gas[cl + ncl * ig];
  // The outputs are interleaved records per column (8, 4, 2, 2 values): written straight from the
  // registers every store instruction would scatter 8-16 bytes per lane over kilobytes.  Transpose
  // through LDS instead, so each store instruction of the block writes one contiguous 2-4 KB run.
  __shared__ Float s_fmj[256 * 9], s_fmn[256 * 5], s_cm[256 * 3];
  __shared__ int s_je[256 * 3];
  const int c0 = blockIdx.x * blockDim.x;
  const int nc = min((int)blockDim.x, ncol - c0);  // columns of this block
#pragma unroll 1
  for (int iflav = 0; iflav < nflav; ++iflav) {
    // :121-168
    const int igas_1 = flavor[2 * iflav], igas_2 = flavor[2 * iflav + 1];
    const Float cg1 = cg_lds ? s_cg[igas_1 * 256 + t] : col_gas[cl + ncl * igas_1];
    const Float cg2 = cg_lds ? s_cg[igas_2 * 256 + t] : col_gas[cl + ncl * igas_2];
    Float fmn[4], fmj[8], cm[2];
    int je[2];
#pragma unroll
    for (int itemp = 0; itemp < 2; ++itemp) {
      const int tt = jt + itemp;  // 1-based
      const size_t v = (size_t)itropo + 2 * ((size_t)0 + (size_t)(ngas + 1) * (tt - 1));
      const Float ratio_eta_half = vmr_ref[v + 2 * (size_t)igas_1] / vmr_ref[v + 2 * (size_t)igas_2];
      const Float c = cg1 + ratio_eta_half * cg2;
      cm[itemp] = c;
      Float eta;
#ifdef RTE_USE_SP
      if (c > (Float)2 * (Float)1.17549435e-38f)
#else
      if (c > (Float)2 * (Float)2.2250738585072014e-308)
#endif
        eta = cg1 / c;
      else
        eta = (Float)0.5;
      const Float loceta = eta * (Float)(neta - 1);
      je[itemp] = min((int)loceta + 1, neta - 1);
      const Float feta = loceta - trunc(loceta);
      const Float ftemp_term = ((Float)(1 - itemp) + (Float)(2 * itemp - 1) * ftemp);
      const Float f1 = ((Float)1 - feta) * ftemp_term;
      const Float f2 = feta * ftemp_term;
      fmn[0 + 2 * itemp] = f1;
      fmn[1 + 2 * itemp] = f2;
      fmj[0 + 4 * itemp] = ((Float)1 - fpress) * f1;
      fmj[1 + 4 * itemp] = ((Float)1 - fpress) * f2;
      fmj[2 + 4 * itemp] = fpress * f1;
      fmj[3 + 4 * itemp] = fpress * f2;
    }
    if (masks) {
      const unsigned m = (3u << je[0]) | (3u << je[1]);  // rows eta, eta + 1 of both temperature corners
      const unsigned w0 = wave_or(in_range && trop ? m : 0u), w1 = wave_or(in_range && !trop ? m : 0u);
      if ((t & 63) == 0) { atomicOr(&s_mask[4 + 2 * iflav], w0); atomicOr(&s_mask[5 + 2 * iflav], w1); }
    }
    __syncthreads();  // the previous flavor's records have been stored
#pragma unroll
    for (int i = 0; i < 8; ++i) s_fmj[t * 9 + i] = fmj[i];
#pragma unroll
    for (int i = 0; i < 4; ++i) s_fmn[t * 5 + i] = fmn[i];
    s_cm[t * 3] = cm[0]; s_cm[t * 3 + 1] = cm[1];
    s_je[t * 3] = je[0]; s_je[t * 3 + 1] = je[1];
    __syncthreads();
    const size_t rec0 = (size_t)c0 + (size_t)ncol * ilay + ncl * iflav;  // record index of the block's first column
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const int e = t + 256 * k;
      if (e < 8 * nc) fmajor[8 * rec0 + e] = s_fmj[(e >> 3) * 9 + (e & 7)];
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int e = t + 256 * k;
      if (e < 4 * nc) fminor[4 * rec0 + e] = s_fmn[(e >> 2) * 5 + (e & 3)];
    }
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      const int e = t + 256 * k;
      if (e < 2 * nc) {
        col_mix[2 * rec0 + e] = s_cm[(e >> 1) * 3 + (e & 1)];
        jeta[2 * rec0 + e] = s_je[(e >> 1) * 3 + (e & 1)];
      }
    }
  }
  if (masks) {
    __syncthreads();
    if (t < mask_w) masks[((size_t)blockIdx.x + (size_t)gridDim.x * ilay) * mask_w + t] = s_mask[t];
  }
}

// -------------------------------------------------------------------------------------------
// Plan guards.  The host-side plans of the production kernels (band / minor-interval metadata, stage width) are
// cached per table address; device-resident tables cannot be inspected by the host without draining the stream.
// Instead of trusting the addresses, every call re-checks the tables ON THE DEVICE against the cached plan: an
// order-independent weighted checksum of the index tables (tau) or the band alignment (Planck, Rayleigh).  On a
// mismatch the guard flag is raised, the production kernels return at once and the direct kernels -- which read the
// caller's tables themselves -- do the call; the host learns about it at its next plan look-up.  So tables changed in
// place, or re-uploaded at the same addresses, give correct results without rte_hip_invalidate_plans().
// -------------------------------------------------------------------------------------------
__host__ __device__ inline unsigned guard_term(unsigned value, unsigned index) {
  return (value + 0x9e3779b9u) * (2u * index + 1u);
}
struct GuardTables {
  const int* ip[10];    // int tables
  int in[10];
  const Bool* bp[4];    // logical tables
  int bn[4];
};
__device__ __forceinline__ void tables_guard_body(const GuardTables& t, unsigned expected, int* __restrict__ flag,
                                                  int* __restrict__ stale) {
  __shared__ unsigned acc;
  if (threadIdx.x == 0) acc = 0;
  __syncthreads();
  unsigned h = 0, base = 0;
  for (int a = 0; a < 10; ++a) {
    for (int i = threadIdx.x; i < t.in[a]; i += 256) h += guard_term((unsigned)t.ip[a][i], base + (unsigned)i);
    base += (unsigned)t.in[a];
  }
  for (int a = 0; a < 4; ++a) {
    for (int i = threadIdx.x; i < t.bn[a]; i += 256) h += guard_term(t.bp[a][i] ? 1u : 0u, base + (unsigned)i);
    base += (unsigned)t.bn[a];
  }
  atomicAdd(&acc, h);
  __syncthreads();
  if (threadIdx.x == 0 && acc != expected) { *flag = 1; *stale = 1; }
}
__global__ void __launch_bounds__(256) tables_guard_kernel(GuardTables t, unsigned expected, int* __restrict__ flag,
                                                           int* __restrict__ stale) {
  tables_guard_body(t, expected, flag, stale);
}
// band limits: whole chunks of gw g-points, ngpt a multiple of gw (what the stage loops of the production kernels assume)
__global__ void bands_guard_kernel(int nbnd, int ngpt, const int* __restrict__ band_lims, int gw, int* __restrict__ flag,
                                   int* __restrict__ stale) {
  bool ok = gw > 0 && ngpt % gw == 0;
  for (int b = threadIdx.x; b < nbnd; b += 64) ok = ok && (band_lims[2 * b] - 1) % gw == 0 && band_lims[2 * b + 1] % gw == 0;
  if (!ok) { *flag = 1; *stale = 1; }
}

// -------------------------------------------------------------------------------------------
// layer limits of the lower / upper atmosphere: reference :274-285 (minloc/maxloc with mask,
// first extremal location; 0 = no such layer)
// -------------------------------------------------------------------------------------------
__device__ __forceinline__ void tropo_limits_body(unsigned bx, int ncol, int nlay, const Float* __restrict__ play,
                                                  const Bool* __restrict__ tropo, int* __restrict__ lim /*(ncol,4)*/,
                                                  int* __restrict__ overlap, int* __restrict__ irregular) {
  const int icol = bx * blockDim.x + threadIdx.x;
  if (icol >= ncol) return;
  const bool top_at_1 = play[0] < play[(size_t)ncol * (nlay - 1)];
  int minloc_t = 0, maxloc_n = 0;
  int first_t = 0, last_t = 0, first_n = 0, last_n = 0;  // first / last layer (1-based) with / without the tropo flag
  Float pmin = 0, pmax = 0;
  // twelve layers requested at a time (one load after the other, the 60 layers of a column were 60 memory latencies)
  constexpr int B = 12;
  for (int l0 = 0; l0 < nlay; l0 += B) {
    Float pb[B];
    bool tb[B];
#pragma unroll
    for (int k = 0; k < B; ++k) {
      const size_t cl = icol + (size_t)ncol * min(l0 + k, nlay - 1);
      pb[k] = play[cl];
      tb[k] = tropo[cl];
    }
#pragma unroll
    for (int k = 0; k < B; ++k) {
      const int ilay = l0 + k;
      const Float p = pb[k];
      if (ilay < nlay) {
        if (tb[k]) {
          if (minloc_t == 0 || p < pmin) { minloc_t = ilay + 1; pmin = p; }
          if (first_t == 0) first_t = ilay + 1;
          last_t = ilay + 1;
        } else {
          if (maxloc_n == 0 || p > pmax) { maxloc_n = ilay + 1; pmax = p; }
          if (first_n == 0) first_n = ilay + 1;
          last_n = ilay + 1;
        }
      }
    }
  }
  int lo1, lo2, up1, up2;
  if (top_at_1) { lo1 = minloc_t; lo2 = nlay; up1 = 1; up2 = maxloc_n; }
  else          { lo1 = 1; lo2 = minloc_t; up1 = maxloc_n; up2 = nlay; }
  // the reference tests layer_limits(icol,1) > 0 only (:450,456); fold "no layers" into lo1/up1
  lim[icol] = lo1;
  lim[icol + ncol] = lo2;
  lim[icol + 2 * (size_t)ncol] = up1;
  lim[icol + 3 * (size_t)ncol] = up2;
  // a layer that lies in BOTH ranges gets both regimes' minor absorbers in the reference (possible
  // only for non-monotone pressure profiles); the production kernel does not handle that
  if (lo1 > 0 && up1 > 0 && max(lo1, up1) <= min(lo2, up2)) *overlap = 1;
  // "regular": every layer lies in exactly the range of its own flag (lower <=> tropo), which is what a pressure
  // profile monotone in the layer index gives.  Only then are masks keyed by the tropo flag alone (those the
  // interpolation call leaves, InterpMasks) the masks tile_geom2_kernel derives from these limits.
  bool regular;
  if (top_at_1) regular = (first_t == 0 || first_t == minloc_t) && (last_n == 0 || last_n == maxloc_n) && (first_t == 0 || last_n == 0 || last_n < first_t);
  else          regular = (last_t == 0 || last_t == minloc_t) && (first_n == 0 || first_n == maxloc_n) && (last_t == 0 || first_n == 0 || last_t < first_n);
  if (!regular) *irregular = 1;
}
__global__ void tropo_limits_kernel(int ncol, int nlay, const Float* __restrict__ play,
                                    const Bool* __restrict__ tropo, int* __restrict__ lim /*(ncol,4)*/,
                                    int* __restrict__ overlap, int* __restrict__ irregular) {
  tropo_limits_body(blockIdx.x, ncol, nlay, play, tropo, lim, overlap, irregular);
}

// Per band, the ordered list of minor intervals whose g-point range intersects the band
// (one wave; ordered compaction by ballot so the reference's interval order is preserved).
__device__ __forceinline__ void plan_minor_body(int nbnd, const int* __restrict__ band_lims_gpt, int nminor,
                                                const int* __restrict__ minor_limits_gpt, int* __restrict__ cnt /*(nbnd)*/,
                                                int* __restrict__ list /*(nminor,nbnd)*/) {
  if (threadIdx.x >= RTE_WAVE) return;  // one wave
  const int lane = threadIdx.x;
  for (int ibnd = 0; ibnd < nbnd; ++ibnd) {
    const int bS = band_lims_gpt[2 * ibnd], bE = band_lims_gpt[2 * ibnd + 1];
    int n = 0;
    for (int base = 0; base < nminor; base += RTE_WAVE) {
      const int i = base + lane;
      bool hit = false;
      if (i < nminor) hit = minor_limits_gpt[2 * i] <= bE && minor_limits_gpt[2 * i + 1] >= bS;
      const unsigned long long m = __ballot(hit);
      if (hit) list[(size_t)ibnd * nminor + n + __popcll(m & ((1ull << lane) - 1ull))] = i;
      n += __popcll(m);
    }
    if (lane == 0) cnt[ibnd] = n;
  }
}
__global__ void plan_minor_kernel(int nbnd, const int* __restrict__ band_lims_gpt, int nminor,
                                  const int* __restrict__ minor_limits_gpt, int* __restrict__ cnt /*(nbnd)*/,
                                  int* __restrict__ list /*(nminor,nbnd)*/) {
  plan_minor_body(nbnd, band_lims_gpt, nminor, minor_limits_gpt, cnt, list);
}

struct MinorTables {
  const Float* kminor;
  const int* limits;       // (2,nminor)
  const Bool* scales_with_density;
  const Bool* scale_by_complement;
  const int* idx_minor;
  const int* idx_minor_scaling;
  const int* kminor_start;
  const int* cnt;          // per band
  const int* list;         // (nminor, nbnd)
  int nminor;
};

// column amount of a minor absorber with its optional scalings: reference :461-480
__device__ __forceinline__ Float minor_scaling(const MinorTables& mt, int imnr, size_t ncl, size_t cl, int idx_h2o, Float P, Float T,
                                               const Float* __restrict__ col_gas) {
  Float scaling = col_gas[cl + ncl * mt.idx_minor[imnr]];
  if (mt.scales_with_density[imnr]) {
    scaling = scaling * ((Float)0.01 * P / T);
    const int isc = mt.idx_minor_scaling[imnr];
    if (isc > 0) {
      const Float vmr_fact = (Float)1 / col_gas[cl];
      const Float dry_fact = (Float)1 / ((Float)1 + col_gas[cl + ncl * idx_h2o] * vmr_fact);
      const Float cgs = col_gas[cl + ncl * isc];
      if (mt.scale_by_complement[imnr])
        scaling = scaling * ((Float)1 - cgs * vmr_fact * dry_fact);
      else
        scaling = scaling * (cgs * vmr_fact * dry_fact);
    }
  }
  return scaling;
}

// contribution of one regime's minor absorbers to the register chunk acc[0..GC)
__device__ __forceinline__ void minor_chunk(const MinorTables& mt, int flav_row, int ibnd, int g0, int gEnd,
                                            int ncol, size_t ncl, size_t cl, int ntemp, int neta, int idx_h2o,
                                            Float P, Float T, int jT, const Float* __restrict__ col_gas,
                                            const Float* __restrict__ fminor, const int* __restrict__ jeta,
                                            const int* __restrict__ gpoint_flavor, Float (&acc)[GC]) {
  const int n = mt.cnt[ibnd];
  for (int k = 0; k < n; ++k) {
    const int imnr = mt.list[(size_t)ibnd * mt.nminor + k];
    const int mS = mt.limits[2 * imnr] - 1, mE = mt.limits[2 * imnr + 1] - 1;  // 0-based
    if (mE < g0 || mS >= g0 + GC) continue;
    const Float scaling = minor_scaling(mt, imnr, ncl, cl, idx_h2o, P, T, col_gas);
    // :485-494
    const int iflav = gpoint_flavor[flav_row + 2 * mS] - 1;
    const size_t clf = cl + ncl * iflav;
    const Float f0 = fminor[4 * clf], f1 = fminor[4 * clf + 1], f2 = fminor[4 * clf + 2], f3 = fminor[4 * clf + 3];
    const int je1 = jeta[2 * clf], je2 = jeta[2 * clf + 1];
    const size_t tn = (size_t)ntemp * neta;
    const size_t o1 = (size_t)(jT - 1) + (size_t)ntemp * (je1 - 1);
    const size_t o2 = (size_t)jT + (size_t)ntemp * (je2 - 1);
    const size_t kb = (size_t)(mt.kminor_start[imnr] - 1);
#pragma unroll
    for (int j = 0; j < GC; ++j) {
      const int g = g0 + j;
      if (g >= mS && g <= mE && g <= gEnd) {
        const Float* kk = mt.kminor + tn * (kb + (size_t)(g - mS));
        const Float tau_minor = f0 * kk[o1] + f1 * kk[o1 + ntemp] + f2 * kk[o2] + f3 * kk[o2 + ntemp];
        acc[j] = acc[j] + scaling * tau_minor;
      }
    }
  }
}

// -------------------------------------------------------------------------------------------
// compute_tau_absorption: reference :176-338 (driver), :345-396 (major), :402-501 (minor)
// -------------------------------------------------------------------------------------------
struct alignas(2 * sizeof(Float)) Float2 { Float x, y; };

// Output planes are written once and never read by the kernel that writes them: stored non-temporally they do not
// push the interpolation weights and index arrays, which the bands of a tile share, out of the 4 MB L2 of the XCD.
// Measured (PMC FETCH_SIZE, 1e5 columns): compute_Planck_source reads 7.65 -> 5.33 GB (4.5 GB is the algorithmic
// minimum) and runs 5.37 -> 5.06 ms; compute_tau_absorption 14.1 -> 12.1 GB, 5.3 -> 5.2 ms.  (Before the wait-count
// fixes of round 2 the same change made no difference: the kernels were stalled on their own stores then.)
struct TauArgs {
  int ncol, nlay, ngpt, neta, npres, ntemp, idx_h2o;
  const int *gpoint_flavor, *band_lims_gpt;
  const Float* kmajor;
  MinorTables lower, upper;
  const int* run_if;  // when non-null the kernel does nothing unless *run_if != 0
  bool overwrite;     // tau is known to be zero (deferred zero_array): do not read it
  const int* lim;
  const Bool* tropo;
  const Float *col_mix, *fmajor, *fminor, *play, *tlay, *col_gas;
  const int *jeta, *jtemp, *jpress;
  Float* tau;
  const Float* add_bybnd;  // (ncol, nlay, nbnd) or nullptr: added to every g-point of its band after the gas terms
};

// direct-gather version for one (column, layer, band): reads the native tables through L1/L2
__device__ __forceinline__ void tau_direct_column(const TauArgs& a, int icol, int ilay, int ibnd) {
  const int ncol = a.ncol, nlay = a.nlay, neta = a.neta, ntemp = a.ntemp;
  const size_t ncl = (size_t)ncol * nlay;
  const size_t cl = icol + (size_t)ncol * ilay;
  const int gptS = a.band_lims_gpt[2 * ibnd] - 1, gptE = a.band_lims_gpt[2 * ibnd + 1] - 1;
  const int itropo = a.tropo[cl] ? 0 : 1;
  const int iflav = a.gpoint_flavor[itropo + 2 * gptS] - 1;
  const size_t clf = cl + ncl * iflav;
  const int jT = a.jtemp[cl];
  const int jp = a.jpress[cl] + itropo + 1;  // "jpress + itropo": levels jp-1 and jp (1-based)
  const int je1 = a.jeta[2 * clf], je2 = a.jeta[2 * clf + 1];
  const Float cm1 = a.col_mix[2 * clf], cm2 = a.col_mix[2 * clf + 1];
  Float fm[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) fm[i] = a.fmajor[8 * clf + i];
  const size_t tn = (size_t)ntemp * neta;
  const size_t gstride = tn * (a.npres + 1);
  // corner offsets (without the g-point term) into kmajor(ntemp,neta,npres+1,ngpt)
  const size_t a0 = (size_t)(jT - 1) + (size_t)ntemp * (je1 - 1) + tn * (size_t)(jp - 2);
  const size_t b0 = (size_t)jT + (size_t)ntemp * (je2 - 1) + tn * (size_t)(jp - 2);
  const Float P = a.play[cl], T = a.tlay[cl];
  const int lay1 = ilay + 1;
  const int lo1 = a.lim[icol], lo2 = a.lim[icol + ncol];
  const int up1 = a.lim[icol + 2 * (size_t)ncol], up2 = a.lim[icol + 3 * (size_t)ncol];
  const bool in_lower = lo1 > 0 && lay1 >= lo1 && lay1 <= lo2;
  const bool in_upper = up1 > 0 && lay1 >= up1 && lay1 <= up2;

  for (int g0 = gptS; g0 <= gptE; g0 += GC) {
    Float acc[GC];
#pragma unroll
    for (int j = 0; j < GC; ++j) acc[j] = (g0 + j <= gptE && !a.overwrite) ? a.tau[cl + ncl * (size_t)(g0 + j)] : (Float)0;
#pragma unroll
    for (int j = 0; j < GC; ++j) {
      if (g0 + j <= gptE) {
        const Float* ka = a.kmajor + gstride * (size_t)(g0 + j) + a0;
        const Float* kb = a.kmajor + gstride * (size_t)(g0 + j) + b0;
        // :791-801
        const Float tau_major =
            cm1 * (fm[0] * ka[0] + fm[1] * ka[ntemp] + fm[2] * ka[tn] + fm[3] * ka[tn + ntemp]) +
            cm2 * (fm[4] * kb[0] + fm[5] * kb[ntemp] + fm[6] * kb[tn] + fm[7] * kb[tn + ntemp]);
        acc[j] = acc[j] + tau_major;
      }
    }
    if (in_lower)
      minor_chunk(a.lower, 0, ibnd, g0, gptE, ncol, ncl, cl, ntemp, neta, a.idx_h2o, P, T, jT, a.col_gas, a.fminor,
                  a.jeta, a.gpoint_flavor, acc);
    if (in_upper)
      minor_chunk(a.upper, 1, ibnd, g0, gptE, ncol, ncl, cl, ntemp, neta, a.idx_h2o, P, T, jT, a.col_gas, a.fminor,
                  a.jeta, a.gpoint_flavor, acc);
    if (a.add_bybnd) {  // increment_1scalar_by_1scalar_bybnd fused in: tau = tau_gas + tau_2(band)
      const Float addv = a.add_bybnd[cl + ncl * (size_t)ibnd];
#pragma unroll
      for (int j = 0; j < GC; ++j) acc[j] = acc[j] + addv;
    }
#pragma unroll
    for (int j = 0; j < GC; ++j)
      if (g0 + j <= gptE) a.tau[cl + ncl * (size_t)(g0 + j)] = acc[j];
  }
}

// ---- the same column from the g-point-fastest table copies of the production path (worklist entries only) ----------
// One 16-byte load brings a corner's coefficients for two g-points, and a band's 16 g-points of a corner share one
// cache line: half the load instructions of the native layout and 1/16 of its cache lines (the worklist kernel is
// bound by the texture addresser: one lane-private line per clock).  Valid where the production path runs: bands and
// minor intervals are whole aligned chunks of 8 or 16 g-points, k-offsets and row lengths are even.  Every g-point
// is formed by the same expression as in tau_direct_column: bit-identical results.
struct GfastTabs { const Float *kmaj, *klo, *kup; int nkl, nku; };

__device__ __forceinline__ void minor_chunk_g(const MinorTables& mt, const Float* __restrict__ kg, int nk, int flav_row, int ibnd,
                                              int g0, int gEnd, size_t ncl, size_t cl, int ntemp, int idx_h2o, Float P, Float T,
                                              int jT, const Float* __restrict__ col_gas, const Float* __restrict__ fminor,
                                              const int* __restrict__ jeta, const int* __restrict__ gpoint_flavor,
                                              Float (&acc)[GC]) {
  const int n = mt.cnt[ibnd];
  for (int k = 0; k < n; ++k) {
    const int imnr = mt.list[(size_t)ibnd * mt.nminor + k];
    const int mS = mt.limits[2 * imnr] - 1, mE = mt.limits[2 * imnr + 1] - 1;  // 0-based
    if (mE < g0 || mS >= g0 + GC) continue;
    const Float scaling = minor_scaling(mt, imnr, ncl, cl, idx_h2o, P, T, col_gas);
    // :485-494
    const int iflav = gpoint_flavor[flav_row + 2 * mS] - 1;
    const size_t clf = cl + ncl * iflav;
    const Float f0 = fminor[4 * clf], f1 = fminor[4 * clf + 1], f2 = fminor[4 * clf + 2], f3 = fminor[4 * clf + 3];
    const int je1 = jeta[2 * clf], je2 = jeta[2 * clf + 1];
    const size_t kb = (size_t)(mt.kminor_start[imnr] - 1);
    // rows (temperature, eta) of the g-fastest copy: [te][nk]
    const Float* r0 = kg + ((size_t)(jT - 1) + (size_t)ntemp * (je1 - 1)) * nk + kb;
    const Float* r1 = r0 + (size_t)ntemp * nk;
    const Float* r2 = kg + ((size_t)jT + (size_t)ntemp * (je2 - 1)) * nk + kb;
    const Float* r3 = r2 + (size_t)ntemp * nk;
#pragma unroll
    for (int j = 0; j < GC; j += 2) {
      const int g = g0 + j;
      if (g >= mS && g <= mE && g <= gEnd) {
        const int c = g - mS;
        const Float2 v0 = *reinterpret_cast<const Float2*>(r0 + c), v1 = *reinterpret_cast<const Float2*>(r1 + c);
        const Float2 v2 = *reinterpret_cast<const Float2*>(r2 + c), v3 = *reinterpret_cast<const Float2*>(r3 + c);
        const Float ta = f0 * v0.x + f1 * v1.x + f2 * v2.x + f3 * v3.x;
        const Float tb = f0 * v0.y + f1 * v1.y + f2 * v2.y + f3 * v3.y;
        acc[j] = acc[j] + scaling * ta;
        acc[j + 1] = acc[j + 1] + scaling * tb;
      }
    }
  }
}

__device__ __forceinline__ void tau_direct_column_g(const TauArgs& a, const GfastTabs& t, int icol, int ilay, int ibnd) {
  constexpr int GH = 8;  // g-points per register chunk here (bands are whole chunks of 8 or 16 on this path)
  const int ncol = a.ncol, nlay = a.nlay, neta = a.neta, ntemp = a.ntemp, ngpt = a.ngpt;
  const size_t ncl = (size_t)ncol * nlay;
  const size_t cl = icol + (size_t)ncol * ilay;
  const int gptS = a.band_lims_gpt[2 * ibnd] - 1, gptE = a.band_lims_gpt[2 * ibnd + 1] - 1;
  const int itropo = a.tropo[cl] ? 0 : 1;
  const int iflav = a.gpoint_flavor[itropo + 2 * gptS] - 1;
  const size_t clf = cl + ncl * iflav;
  const int jT = a.jtemp[cl];
  const int jp = a.jpress[cl] + itropo + 1;  // "jpress + itropo": levels jp-1 and jp (1-based)
  const int je1 = a.jeta[2 * clf], je2 = a.jeta[2 * clf + 1];
  const Float cm1 = a.col_mix[2 * clf], cm2 = a.col_mix[2 * clf + 1];
  Float fm[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) fm[i] = a.fmajor[8 * clf + i];
  const unsigned TE = (unsigned)ntemp * neta;
  // rows [pressure level][eta][temperature] x ngpt of the g-fastest copy, as 32-bit element offsets (the table has
  // (npres + 1) * TE * ngpt < 2^31 elements: checked where the copy is made)
  const unsigned oA = ((unsigned)(jp - 2) * TE + (unsigned)(jT - 1) + (unsigned)ntemp * (je1 - 1)) * (unsigned)ngpt;
  const unsigned oB = ((unsigned)(jp - 2) * TE + (unsigned)jT + (unsigned)ntemp * (je2 - 1)) * (unsigned)ngpt;
  const unsigned dE = (unsigned)ntemp * ngpt, dP = TE * (unsigned)ngpt;  // next eta row, next pressure level
  const Float P = a.play[cl], T = a.tlay[cl];
  const int lay1 = ilay + 1;
  const int lo1 = a.lim[icol], lo2 = a.lim[icol + ncol];
  const int up1 = a.lim[icol + 2 * (size_t)ncol], up2 = a.lim[icol + 3 * (size_t)ncol];
  const bool in_lower = lo1 > 0 && lay1 >= lo1 && lay1 <= lo2;
  const bool in_upper = up1 > 0 && lay1 >= up1 && lay1 <= up2;
  auto row2 = [&](unsigned off) { return *reinterpret_cast<const Float2*>(t.kmaj + off); };

  for (int g0 = gptS; g0 <= gptE; g0 += GH) {
    Float acc[GC];  // (minor_chunk_g works on GC-wide chunks: the upper half stays unused here)
#pragma unroll
    for (int j = 0; j < GC; ++j) acc[j] = (Float)0;
#pragma unroll
    for (int j = 0; j < GH; ++j) acc[j] = a.overwrite ? (Float)0 : a.tau[cl + ncl * (size_t)(g0 + j)];
#pragma unroll
    for (int j = 0; j < GH; j += 2) {
      const unsigned g = (unsigned)(g0 + j);
      const Float2 a00 = row2(oA + g), a01 = row2(oA + dE + g), a10 = row2(oA + dP + g), a11 = row2(oA + dP + dE + g);
      const Float2 b00 = row2(oB + g), b01 = row2(oB + dE + g), b10 = row2(oB + dP + g), b11 = row2(oB + dP + dE + g);
      // :791-801
      const Float ta = cm1 * (fm[0] * a00.x + fm[1] * a01.x + fm[2] * a10.x + fm[3] * a11.x) +
                       cm2 * (fm[4] * b00.x + fm[5] * b01.x + fm[6] * b10.x + fm[7] * b11.x);
      const Float tb = cm1 * (fm[0] * a00.y + fm[1] * a01.y + fm[2] * a10.y + fm[3] * a11.y) +
                       cm2 * (fm[4] * b00.y + fm[5] * b01.y + fm[6] * b10.y + fm[7] * b11.y);
      acc[j] = acc[j] + ta;
      acc[j + 1] = acc[j + 1] + tb;
    }
    const int gEnd = g0 + GH - 1;  // this chunk only (the upper half of acc is not a g-point here)
    if (in_lower)
      minor_chunk_g(a.lower, t.klo, t.nkl, 0, ibnd, g0, gEnd, ncl, cl, ntemp, a.idx_h2o, P, T, jT, a.col_gas, a.fminor, a.jeta,
                    a.gpoint_flavor, acc);
    if (in_upper)
      minor_chunk_g(a.upper, t.kup, t.nku, 1, ibnd, g0, gEnd, ncl, cl, ntemp, a.idx_h2o, P, T, jT, a.col_gas, a.fminor, a.jeta,
                    a.gpoint_flavor, acc);
    if (a.add_bybnd) {  // increment_1scalar_by_1scalar_bybnd fused in: tau = tau_gas + tau_2(band)
      const Float addv = a.add_bybnd[cl + ncl * (size_t)ibnd];
#pragma unroll
      for (int j = 0; j < GH; ++j) acc[j] = acc[j] + addv;
    }
#pragma unroll
    for (int j = 0; j < GH; ++j) a.tau[cl + ncl * (size_t)(g0 + j)] = acc[j];
  }
}

// direct kernel over all (column tile, layer, band) triples, grid-stride
__global__ void __launch_bounds__(256) tau_absorption_kernel(TauArgs a, int nbnd) {
  if (a.run_if && *a.run_if == 0) return;
  const unsigned tiles_x = (a.ncol + 255) / 256;
  const size_t total = (size_t)tiles_x * a.nlay * nbnd;
  for (size_t w = blockIdx.x; w < total; w += gridDim.x) {
    const int tx = (int)(w % tiles_x);
    const int ilay = (int)((w / tiles_x) % a.nlay);
    const int ibnd = (int)(w / ((size_t)tiles_x * a.nlay));
    const int icol = tx * 256 + threadIdx.x;
    if (icol < a.ncol) tau_direct_column(a, icol, ilay, ibnd);
  }
}

// -------------------------------------------------------------------------------------------
// LUT re-layout (per call, into the scratch arena): (TE = ntemp*neta, nouter, ng) with the
// (temperature, eta) plane fastest  ->  rows of g-points: out[(o*TE + te)*ng + g].
// A band's g-points of one (T, eta, p) corner become one contiguous 128-byte row, which is what
// the LDS staging below copies.  ~35 MB moved per call (L2 / Infinity-Cache resident): ~10 us.
// -------------------------------------------------------------------------------------------
__device__ __forceinline__ void relayout_gfast_body(unsigned bx, unsigned by, int TE, int nouter, int ng,
                                                    const Float* __restrict__ in, Float* __restrict__ out) {
  extern __shared__ Float tile[];  // [TE][33]
  const int g0 = bx * 32, o = by;
  const int ngc = min(32, ng - g0);
  for (int idx = threadIdx.x; idx < TE * ngc; idx += blockDim.x) {
    const int te = idx % TE, gg = idx / TE;
    tile[te * 33 + gg] = in[(size_t)te + (size_t)TE * ((size_t)o + (size_t)nouter * (g0 + gg))];
  }
  __syncthreads();
  for (int idx = threadIdx.x; idx < TE * ngc; idx += blockDim.x) {
    const int gg = idx % ngc, te = idx / ngc;
    out[((size_t)o * TE + te) * ng + g0 + gg] = tile[te * 33 + gg];
  }
}
__global__ void __launch_bounds__(256)
relayout_gfast_kernel(int TE, int nouter, int ng, const Float* __restrict__ in, Float* __restrict__ out) {
  relayout_gfast_body(blockIdx.x, blockIdx.y, TE, nouter, ng, in, out);
}

// Everything compute_tau_absorption's production path prepares before its geometry pre-pass, in ONE launch: the
// blocks take roles by index -- layer limits per column, the two minor-interval plans of the stand-by direct kernel,
// the g-fastest copies of up to five tables, the plan guard.  The roles do not depend on each other; as seven
// launches of 5-40 us each they cost their sum (0.08 ms) plus the gaps between dependent launches.
struct TauSetupArgs {
  int ncol, nlay, nbnd, TE;
  const Float* play; const Bool* tropo; int *lim, *overlap, *irregular;
  const int* band_lims;
  int nminor[2]; const int* minor_limits[2]; int* cnt[2]; int* list[2];
  int ntab; int nouter[5], ng[5], first_block[6]; const Float* tin[5]; Float* tout[5];  // tables to re-lay out
  GuardTables gt; unsigned guard_expected; int* stale;
  unsigned b_plan, b_tab, b_guard;  // first block of each role after the layer limits
};
__global__ void __launch_bounds__(256) tau_setup_kernel(TauSetupArgs a) {
  const unsigned b = blockIdx.x;
  if (b < a.b_plan) {
    tropo_limits_body(b, a.ncol, a.nlay, a.play, a.tropo, a.lim, a.overlap, a.irregular);
  } else if (b < a.b_tab) {
    const int r = b - a.b_plan;
    plan_minor_body(a.nbnd, a.band_lims, a.nminor[r], a.minor_limits[r], a.cnt[r], a.list[r]);
  } else if (b < a.b_guard) {
    const unsigned q = b - a.b_tab;
    int t = 0;
    while (t + 1 < a.ntab && q >= (unsigned)a.first_block[t + 1]) ++t;
    const unsigned local = q - a.first_block[t];
    const unsigned nbx = (a.ng[t] + 31) / 32;
    relayout_gfast_body(local % nbx, local / nbx, a.TE, a.nouter[t], a.ng[t], a.tin[t], a.tout[t]);
  } else {
    tables_guard_body(a.gt, a.guard_expected, a.overlap, a.stale);
  }
}

// -------------------------------------------------------------------------------------------
// compute_tau_absorption, production kernels (LDS slab).
//
// What the measurements on MI355X said (tools/membench.hip): kernels that gather
// LUT values straight from global memory with lanes = columns run at ~20 ms per 1e5 columns whatever the
// table layout, because each lane pulls its own cache line and the vector L1 retires about one distinct
// line per clock.  Both kernels below therefore
//   * copy the tables to a g-point-fastest layout per call (relayout_gfast_kernel), so that the 16 g-points
//     of a stage are one 128-byte row piece;
//   * stage, per (column tile, layer, band), the BOUNDING BOX of the rows the tile's columns need --
//     pressure x temperature x eta ranges for kmajor, temperature x eta per minor interval -- into an LDS
//     slab with a row stride of 18 doubles;
//   * keep lanes = columns: every thread gathers its 8 major + 4-per-interval minor corner rows with
//     16-byte LDS reads (two g-points per read) and writes tau with coalesced 512-byte wave stores.
// tau_absorption_v7_kernel does all of it with one kind of wave and two barriers per band;
// tau_absorption_v9_kernel (default) splits the roles: loader waves stage the next stage's slab into the
// other half of a double-buffered slab while compute waves gather, one barrier per stage.
// Tiles whose box does not fit the slab go to a worklist for the direct-gather kernel.
// Arithmetic: the same products and sums as the reference (:791-801, :757-760) evaluated with
// fused multiply-adds and col_mix folded into the major weights; differences from the reference
// association are a few ulp (tests: 1e-12 relative).
// -------------------------------------------------------------------------------------------
constexpr int MAXM = 12;   // minor intervals per (band, regime) handled by the production kernels; more -> native kernel
constexpr int MAXB = 32;   // bands

struct MinorMeta {  // one minor interval
  int mS, mE, idx_minor, idx_scaling, kstart, flags /*1: scales with density, 2: by complement*/;
};
struct BandMeta {  // built on the host from the small index tables, uploaded per call
  int cnt[2];
  int gS, gE;   // g-point range of the band (0-based)
  int flav[2];  // flavor (0-based) of the band per tropo regime: gpoint_flavor(:, gS)
  MinorMeta m[2][MAXM];  // [0]: lower-regime intervals of the band, [1]: upper
};

// combine_abs_and_rayleigh, 2-stream branch (rrtmgp/frontend/mo_gas_optics_rrtmgp.F90:1983-2002), applied to one value,
// optionally followed by increment_2stream_by_2stream_bybnd (rte/kernels/mo_optical_props_kernels.F90: the by-band
// form of :159-181) with a second set of 2-stream properties given per band (clouds): the same operations in the same
// order as the separate kernels, on values that are doubles in registers instead of doubles in memory -- bit-identical.
struct RaylCombine {
  const Float* tau_abs;  // nullptr: plain compute_tau_rayleigh
  Float *tau, *ssa, *g;  // tau may alias tau_abs
  const Float *cld_tau, *cld_ssa, *cld_g;  // (ncol, nlay, nbnd) or nullptr
};
#ifdef RTE_USE_SP
#define RTE_TINY 1.17549435e-38f
#else
#define RTE_TINY 2.2250738585072014e-308
#endif
__device__ __forceinline__ void rayl_finish(Float ta, Float tr, bool cld, Float t2, Float s2, Float g2, Float& tau, Float& ssa,
                                            Float& g) {
  const Float tiny2 = (Float)2 * (Float)RTE_TINY;
  const Float t = ta + tr;
  ssa = t > tiny2 ? tr / t : (Float)0;
  tau = t;
  g = (Float)0;
  if (cld) {
    const Float eps = (Float)3 * (Float)RTE_TINY;  // mo_optical_props_kernels.F90:38
    const Float tau12 = tau + t2;
    const Float tauscat12 = tau * ssa + t2 * s2;
    g = (tau * ssa * g + t2 * s2 * g2) / fmax(eps, tauscat12);
    ssa = tauscat12 / fmax(eps, tau12);
    tau = tau12;
  }
}
// compute_tau_absorption fused with compute_tau_rayleigh and the 2-stream combine (rte_hip_gas_optics_sw_2str): the
// Rayleigh table rows are staged like one more pair of minor planes, and a stage writes tau, ssa, g instead of tau_abs
struct RaylFuse {
  const Float* krayl_g[2];  // g-fastest copies of krayl(:, :, :, regime)
  const Float* col_dry;
  const Float *cld_tau, *cld_ssa, *cld_g;  // (ncol, nlay, nbnd) or nullptr: increment by band-wise 2-stream properties
  Float *ssa, *g;           // (tau goes to TauV5::tau)
};

struct TauV5 {
  int ncol, nlay, ngpt, nbnd, ntemp, TE, idx_h2o, nk_lo, nk_up;
  const int* band_lims;      // (2,nbnd)
  const int* gpoint_flavor;  // (2,ngpt)
  const BandMeta* bmeta;     // [nbnd]
  const Float *kmaj, *klo, *kup;  // g-fastest tables
  const int *lim, *jeta, *jtemp, *jpress;
  const Bool* tropo;
  const Float *col_mix, *fmajor, *fminor, *play, *tlay, *col_gas;
  Float* tau;
  const int* skip_if;  // device flag: some column has overlapping regimes -> the fallback kernel does the call
  int* worklist;       // [0] = count, then (tile, layer, band) triples for tau_absorption_worklist_kernel
  bool overwrite;      // tau is known to be zero (deferred zero_array): do not read it
  bool atomic_ok;      // tau is device memory proper: hardware floating-point atomics are defined on it (not on host-visible memory)
  const Float* add_bybnd;  // (ncol, nlay, nbnd) or nullptr: see TauArgs
  RaylFuse rf;             // used by the RAYL instantiations only
#ifdef EXP_CLOCKS
  unsigned long long* clocks;
#endif
};

// wave-wide min / max by butterfly shuffles (LDS atomics on one address serialise lane by lane)
__device__ __forceinline__ int wave_min(int v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = min(v, __shfl_xor(v, o));
  return v;
}
__device__ __forceinline__ Float2 ld2(const Float* p) { return *reinterpret_cast<const Float2*>(p); }
__device__ __forceinline__ int wave_max(int v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = max(v, __shfl_xor(v, o));
  return v;
}

// LDS slab row stride in Floats: 18 = nine 16-byte quads.  Rows are read with ds_read_b128 (two g-points per
// read), which the LDS serves in groups of 16 lanes x 4 banks: an odd quad stride puts the rows of a group on
// distinct banks (MI355X_MICROARCH.md, LDS), and b128 reaches the LDS peak with one wave per SIMD where
// 8-byte reads need four.
constexpr int RS = GC + 2;
constexpr int SLAB_FLOATS = 8704;  // 68 KB of LUT slab per block (2 blocks per CU); tiles that need more go to the direct kernel

// lanes = columns; block = (256 columns, one layer), walks the bands.  Per band the block stages the
// bounding box of LUT rows its columns need (pressure x temperature x eta ranges of the tile) from the
// g-fastest tables into LDS -- each 128-byte row piece is one coalesced line -- and every thread then
// gathers its 8 major + 4-per-interval minor corner rows with 16-byte LDS reads.  The staging pieces of a
// band are requested back to back (one L2 latency per batch); the eta indices of band b+1 are requested
// before band b is computed.  A tile whose bounding box does not fit the slab is appended to a worklist
// for tau_absorption_worklist_kernel.
template <int BS, int MINW, int HW, int SLAB>
__global__ void __launch_bounds__(BS, MINW) tau_absorption_v7_kernel(TauV5 a) {
  __shared__ int rng[6];      // Tmin, Tmax, Pmin, Pmax, has_lower, has_upper
  __shared__ int erng[2][2];  // eta range of the band (ping-pong between bands)
  __shared__ __align__(16) Float slab[SLAB];
  extern __shared__ BandMeta bm[];  // [nbnd]
  if (*a.skip_if) return;
  const int tid = threadIdx.x;
  const unsigned ncol = a.ncol, nlay = a.nlay;
  const unsigned ilay = blockIdx.y;
  const unsigned ncl = ncol * nlay;  // host guarantees < 2^31
  const int ntemp = a.ntemp, TE = a.TE, ngpt = a.ngpt, nbnd = a.nbnd;
  if (tid == 0) {
    rng[0] = 1 << 30; rng[1] = -1; rng[2] = 1 << 30; rng[3] = -1; rng[4] = 0; rng[5] = 0;
    erng[0][0] = 1 << 30; erng[0][1] = -1; erng[1][0] = 1 << 30; erng[1][1] = -1;
  }
  {  // band metadata -> LDS (a few KB, coalesced)
    const int* src = reinterpret_cast<const int*>(a.bmeta);
    int* dst = reinterpret_cast<int*>(bm);
    const int nw = nbnd * (int)(sizeof(BandMeta) / sizeof(int));
    for (int i = tid; i < nw; i += BS) dst[i] = src[i];
  }
  __syncthreads();
  // ---- band-independent state of this thread's column
  const unsigned icol = blockIdx.x * BS + tid;
  const bool valid = icol < ncol;
  const unsigned ic = min(icol, ncol - 1);
  const unsigned cl = ic + ncol * ilay;
  const int itropo = a.tropo[cl] ? 0 : 1;
  const int jT = a.jtemp[cl];
  const int jp = a.jpress[cl] + itropo + 1;  // levels jp-1, jp (1-based)
  int regime;
  {
    const int lay1 = ilay + 1;
    const int lo1 = a.lim[ic], lo2 = a.lim[ic + ncol];
    const int up1 = a.lim[ic + 2 * (size_t)ncol], up2 = a.lim[ic + 3 * (size_t)ncol];
    regime = ((lo1 > 0 && lay1 >= lo1 && lay1 <= lo2) ? 1 : 0) | ((up1 > 0 && lay1 >= up1 && lay1 <= up2) ? 2 : 0);
  }
  const int rsel = regime == 2 ? 1 : 0;
  const Float P = a.play[cl], T = a.tlay[cl];
  const Float dens = (Float)0.01 * P / T;                                                             // :469
  const Float vmr_fact = (Float)1 / a.col_gas[cl];                                                    // :471
  const Float dry_fact = (Float)1 / ((Float)1 + a.col_gas[cl + (size_t)ncl * a.idx_h2o] * vmr_fact);  // :472
  {
    const int big = 1 << 30;
    const int a0 = wave_min(valid ? jT : big), a1 = wave_max(valid ? jT + 1 : -1);
    const int a2 = wave_min(valid ? jp - 1 : big), a3 = wave_max(valid ? jp : -1);
    const int a4 = wave_max(valid ? (regime & 1) : 0), a5 = wave_max(valid ? (regime & 2) : 0);
    if ((tid & 63) == 0) {
      atomicMin(&rng[0], a0); atomicMax(&rng[1], a1); atomicMin(&rng[2], a2); atomicMax(&rng[3], a3);
      if (a4) rng[4] = 1;
      if (a5) rng[5] = 1;
    }
  }

  // eta indices of band b (major flavor and the minor regime's flavor): prefetched one band ahead, they
  // define the slab's bounding box; the weights are requested while the slab is being staged
  auto load_idx = [&](int b, int2& je, int2& em) {
    const int gptS = a.band_lims[2 * b] - 1;
    const int iflav = a.gpoint_flavor[itropo + 2 * gptS] - 1;
    const int iflav_m = a.gpoint_flavor[rsel + 2 * gptS] - 1;  // minor absorbers use THEIR regime's flavor (:487)
    je = *reinterpret_cast<const int2*>(a.jeta + 2 * (cl + (size_t)ncl * iflav));
    em = *reinterpret_cast<const int2*>(a.jeta + 2 * (cl + (size_t)ncl * iflav_m));
  };
  int2 nje, nem;
  load_idx(0, nje, nem);

  for (int ibnd = 0; ibnd < nbnd; ++ibnd) {
    const int gptS = a.band_lims[2 * ibnd] - 1, gptE = a.band_lims[2 * ibnd + 1] - 1;
    int* er = erng[ibnd & 1];
    const int je1 = nje.x, je2 = nje.y, em1 = nem.x, em2 = nem.y;
    {
      const int e0 = wave_min(valid ? min(min(je1, je2), min(em1, em2)) : (1 << 30));
      const int e1 = wave_max(valid ? max(max(je1, je2), max(em1, em2)) + 1 : -1);
      if ((tid & 63) == 0) { atomicMin(&er[0], e0); atomicMax(&er[1], e1); }
    }
    __syncthreads();  // ranges complete; previous band's compute finished (slab is free)
    const int Tmin = rng[0], nT = rng[1] - rng[0] + 1, Pmin = rng[2], nP = rng[3] - rng[2] + 1;
    const int n_lo = rng[4] ? bm[ibnd].cnt[0] : 0, n_up = rng[5] ? bm[ibnd].cnt[1] : 0;
    const int emin = er[0], nE = er[1] - er[0] + 1;
    const float inv_nE = 1.0f / (float)nE, inv_nT = 1.0f / (float)nT;
    const int rowsMaj = nP * nT * nE, rowsLo = n_lo * nT * nE, rowsUp = n_up * nT * nE;
    const bool use_lds = (rowsMaj + rowsLo + rowsUp) * RS <= SLAB && regime != 3;
    if (tid == 0) {
      erng[(ibnd + 1) & 1][0] = 1 << 30; erng[(ibnd + 1) & 1][1] = -1;
      if (!use_lds) {  // hand (tile, layer, band) to the direct kernel
        const int w = atomicAdd(&a.worklist[0], 1);
        a.worklist[1 + 3 * w] = blockIdx.x; a.worklist[2 + 3 * w] = ilay; a.worklist[3 + 3 * w] = ibnd;
      }
    }
    if (ibnd + 1 < nbnd) load_idx(ibnd + 1, nje, nem);
    if (!use_lds) continue;  // block-uniform
    // weights and minor column amounts of this band: requested now, used after the staging
    Float2 fm[4], fn[2], cm;
    {
      const int iflav = a.gpoint_flavor[itropo + 2 * gptS] - 1;
      const size_t clf = cl + (size_t)ncl * iflav;
      const Float2* fmp = reinterpret_cast<const Float2*>(a.fmajor + 8 * clf);
#pragma unroll
      for (int i = 0; i < 4; ++i) fm[i] = fmp[i];
      cm = *reinterpret_cast<const Float2*>(a.col_mix + 2 * clf);
      const int iflav_m = a.gpoint_flavor[rsel + 2 * gptS] - 1;
      const Float2* fnp = reinterpret_cast<const Float2*>(a.fminor + 4 * (cl + (size_t)ncl * iflav_m));
      fn[0] = fnp[0]; fn[1] = fnp[1];
    }
    const int n_my = regime > 0 ? bm[ibnd].cnt[rsel] : 0;
    Float w0 = 0, w1 = 0, w2 = 0, w3 = 0, w4 = 0, w5 = 0, w6 = 0, w7 = 0, f0 = 0, f1 = 0, f2 = 0, f3 = 0;

#pragma unroll 1
    for (int g0 = gptS; g0 <= gptE; g0 += GC) {  // host guarantees whole, 16-aligned chunks
      if (g0 != gptS) __syncthreads();
      // ---- stage the slab; rows ordered [p][t][eta] (+ minor: [interval][t][eta]); 16-byte pieces.
      // Up to SB pieces per thread are requested back to back and only then written to LDS, so a tile
      // pays the L2 latency once per batch, not once per piece.
#ifdef EXP_NOSTAGE
      if (a.ncol == -12345)
#endif
      {
        constexpr int SB = 8;
        const int nMaj = rowsMaj * (GC / 2), nAll = (rowsMaj + rowsLo + rowsUp) * (GC / 2);
        auto piece = [&](int idx) -> Float2 {
          const int j = idx & 7, r = idx >> 3;
          if (idx < nMaj) {
            const int rest = (int)(((float)r + 0.5f) * inv_nE), e = r - rest * nE;  // rows < 2^12: exact
            const int p_l = (int)(((float)rest + 0.5f) * inv_nT), t_l = rest - p_l * nT;
            return *reinterpret_cast<const Float2*>(
                a.kmaj + ((size_t)((Pmin - 1 + p_l) * TE + (emin - 1 + e) * ntemp + (Tmin - 1 + t_l)) * ngpt + g0 + 2 * j));
          }
          const int rm = r - rowsMaj;
          const bool up = rm >= rowsLo;
          const int rr = up ? rm - rowsLo : rm;
          const int rest = (int)(((float)rr + 0.5f) * inv_nE), e = rr - rest * nE;
          const int q = (int)(((float)rest + 0.5f) * inv_nT), t_l = rest - q * nT;
          const MinorMeta& m = bm[ibnd].m[up ? 1 : 0][q];
          Float2 v{0, 0};
          if (m.mS <= g0 && m.mE >= g0) {
            const Float* kg = up ? a.kup : a.klo;
            const unsigned nk = up ? a.nk_up : a.nk_lo;
            v = *reinterpret_cast<const Float2*>(
                kg + ((size_t)((emin - 1 + e) * ntemp + (Tmin - 1 + t_l)) * nk + (unsigned)m.kstart + (g0 - m.mS) + 2 * j));
          }
          return v;
        };
#pragma unroll 1
        for (int base = tid; base < nAll; base += SB * BS) {
          Float2 v[SB];
#pragma unroll
          for (int u = 0; u < SB; ++u) {
            v[u] = Float2{0, 0};
            if (base + u * BS < nAll) v[u] = piece(base + u * BS);
          }
#pragma unroll
          for (int u = 0; u < SB; ++u) {
            const int idx = base + u * BS;
            if (idx < nAll) *reinterpret_cast<Float2*>(slab + (idx >> 3) * RS + 2 * (idx & 7)) = v[u];
          }
        }
      }
      __syncthreads();
      if (!valid) continue;
#ifdef EXP_NOCOMPUTE
      if (a.ncol != -12345) continue;
#endif
      if (g0 == gptS) {
        // col_mix folded into the major weights
        w0 = cm.x * fm[0].x; w1 = cm.x * fm[0].y; w2 = cm.x * fm[1].x; w3 = cm.x * fm[1].y;
        w4 = cm.y * fm[2].x; w5 = cm.y * fm[2].y; w6 = cm.y * fm[3].x; w7 = cm.y * fm[3].y;
        f0 = fn[0].x; f1 = fn[0].y; f2 = fn[1].x; f3 = fn[1].y;
      }
      const Float* A0_ = slab + (((jp - 1 - Pmin) * nT + (jT - Tmin)) * nE + (je1 - emin)) * RS;
      const Float* B0_ = slab + (((jp - 1 - Pmin) * nT + (jT + 1 - Tmin)) * nE + (je2 - emin)) * RS;
      const int sP = nT * nE * RS;
      const Float* M0_ = slab + (rowsMaj + (regime == 2 ? rowsLo : 0)) * RS;
      const Float *A0 = A0_, *B0 = B0_, *M0 = M0_;
#ifdef EXP_BCAST
      A0 = slab + (a.ncol == -12345 ? tid : 0) * RS; B0 = A0 + 2 * RS; M0 = A0 - ((jT - Tmin) * nE + (em1 - emin)) * RS;
#endif
#pragma unroll 1
      for (int h = 0; h < GC; h += HW) {  // HW g-points at a time: bounded register footprint
        Float acc[HW];
        Float* tp = a.tau + cl + (size_t)ncl * (g0 + h);
        if (a.overwrite) {
#pragma unroll
          for (int j = 0; j < HW; ++j) acc[j] = 0;
        } else {
#pragma unroll
          for (int j = 0; j < HW; ++j) acc[j] = tp[(size_t)ncl * j];
        }
#pragma unroll
        for (int j = 0; j < HW; j += 2) {
          // :791-801 with col_mix folded into the weights; one 16-byte LDS read feeds two g-points
          const Float2 k0 = ld2(A0 + h + j), k1 = ld2(A0 + RS + h + j), k2 = ld2(A0 + sP + h + j),
                       k3 = ld2(A0 + sP + RS + h + j), k4 = ld2(B0 + h + j), k5 = ld2(B0 + RS + h + j),
                       k6 = ld2(B0 + sP + h + j), k7 = ld2(B0 + sP + RS + h + j);
          Float m = w0 * k0.x, n = w0 * k0.y;
          m = fma(w1, k1.x, m); n = fma(w1, k1.y, n);
          m = fma(w2, k2.x, m); n = fma(w2, k2.y, n);
          m = fma(w3, k3.x, m); n = fma(w3, k3.y, n);
          m = fma(w4, k4.x, m); n = fma(w4, k4.y, n);
          m = fma(w5, k5.x, m); n = fma(w5, k5.y, n);
          m = fma(w6, k6.x, m); n = fma(w6, k6.y, n);
          m = fma(w7, k7.x, m); n = fma(w7, k7.y, n);
          acc[j] = acc[j] + m;
          acc[j + 1] = acc[j + 1] + n;
          if ((j & 2) != 0) __builtin_amdgcn_sched_barrier(0);  // at most 16 row reads (64 VGPRs) in flight
        }
        // minor absorbers of this regime; the column amounts of interval k+1 are requested while k is computed
        Float amt = 0, amt_s = 0, amt_n = 0, amt_sn = 0;
        auto load_amounts = [&](int k, Float& x, Float& xs) {
          const MinorMeta& m = bm[ibnd].m[rsel][k];
          x = a.col_gas[cl + (size_t)ncl * m.idx_minor];
          xs = ((m.flags & 1) && m.idx_scaling > 0) ? a.col_gas[cl + (size_t)ncl * m.idx_scaling] : (Float)0;
        };
        if (n_my > 0) load_amounts(0, amt_n, amt_sn);
#pragma unroll 1
        for (int k = 0; k < n_my; ++k) {
          amt = amt_n; amt_s = amt_sn;
          if (k + 1 < n_my) load_amounts(k + 1, amt_n, amt_sn);
          const MinorMeta& mm = bm[ibnd].m[rsel][k];
          if (mm.mE < g0 || mm.mS > g0) continue;  // intervals are whole 16-aligned chunks inside the band
          Float scaling = amt;  // :461-480
          if (mm.flags & 1) {
            scaling = scaling * dens;  // :469
            if (mm.idx_scaling > 0) {  // :470-478
              if (mm.flags & 2)
                scaling = scaling * ((Float)1 - amt_s * vmr_fact * dry_fact);
              else
                scaling = scaling * (amt_s * vmr_fact * dry_fact);
            }
          }
          const Float* r1 = M0 + ((k * nT + (jT - Tmin)) * nE + (em1 - emin)) * RS + h;
          const Float* r2 = M0 + ((k * nT + (jT + 1 - Tmin)) * nE + (em2 - emin)) * RS + h;
#pragma unroll
          for (int j = 0; j < HW; j += 2) {
            // :757-760, :493
            const Float2 q0 = ld2(r1 + j), q1 = ld2(r1 + RS + j), q2 = ld2(r2 + j), q3 = ld2(r2 + RS + j);
            Float s_ = f0 * q0.x, t_ = f0 * q0.y;
            s_ = fma(f1, q1.x, s_); t_ = fma(f1, q1.y, t_);
            s_ = fma(f2, q2.x, s_); t_ = fma(f2, q2.y, t_);
            s_ = fma(f3, q3.x, s_); t_ = fma(f3, q3.y, t_);
            acc[j] = fma(scaling, s_, acc[j]);
            acc[j + 1] = fma(scaling, t_, acc[j + 1]);
          }
        }
#ifdef EXP_NOSTORE
        Float sum = 0;
#pragma unroll
        for (int j = 0; j < HW; ++j) sum += acc[j];
        if (sum == (Float)-12345.678) tp[0] = sum;
#else
#pragma unroll
        for (int j = 0; j < HW; ++j) tp[(size_t)ncl * j] = acc[j];
#endif
      }
    }
  }
}


#ifndef V9_NCW  // shape of the specialised-wave kernel (overridable for experiments: tools/variants.py)
#define V9_NCW 8
#define V9_NLW 2
#endif
#ifndef V9_SB
#define V9_SB 8
#endif
#ifndef V9_SLAB
#define V9_SLAB 8704   // floats per slab buffer (two buffers per block)
#endif
#ifndef V9_MINW
#define V9_MINW ((V9_NCW + V9_NLW + 3) / 4)  // waves per SIMD the register budget must allow
#endif
// -------------------------------------------------------------------------------------------
// compute_tau_absorption, specialised-wave kernel ("v9").
//
// Measured on the slab kernel above (tools/variants.py, per-phase cycle counters): staging, compute and
// the tau stores of a stage run back to back -- vector-memory operations of a wave retire in order, so a
// staging load issued after the previous stage's stores waits for them, and the range reductions, the
// dependent index loads and two barriers per stage sit on the same critical path.  Here the work is split:
//   * tau_geom_kernel (tiny pre-pass) computes, per (column tile, layer), the bounding box of LUT rows
//     every band needs, and sends oversized (tile, layer, band) triples to the direct-gather worklist;
//   * the main kernel runs one block per CU with NCW compute waves (lanes = columns) and NLW loader
//     waves.  The loaders know the whole schedule from the geometry table: they stage the slab of stage
//     s+1 into the other half of a double-buffered LDS slab while the compute waves work on stage s, with
//     ONE barrier per stage.  The loaders' memory queue holds only table reads; the compute waves' queue
//     holds weights (requested one stage ahead) and tau stores, so neither waits for the other's traffic.
// -------------------------------------------------------------------------------------------
struct TileGeom {   // one per (column tile, layer)
  int Tmin, nT, Pmin, nP, has_lo, has_up, pad0, pad1;
  int2 eg[MAXB];    // per band: (emin, nE); nE = 0 -> band handled by the direct kernel (or no work)
};

template <int TILE, int G>
__global__ void __launch_bounds__(TILE) tau_geom_kernel(TauV5 a, TileGeom* __restrict__ geom, int slab_floats) {
  constexpr int RS = G + 2;

  __shared__ int rng[6];
  __shared__ int erng[MAXB][2];
  __shared__ BandMeta bm[MAXB];
  if (*a.skip_if) return;
  const int tid = threadIdx.x;
  const unsigned ncol = a.ncol, nlay = a.nlay, ilay = blockIdx.y;
  const unsigned ncl = ncol * nlay;
  const int nbnd = a.nbnd;
  if (tid == 0) { rng[0] = 1 << 30; rng[1] = -1; rng[2] = 1 << 30; rng[3] = -1; rng[4] = 0; rng[5] = 0; }
  if (tid < MAXB) { erng[tid][0] = 1 << 30; erng[tid][1] = -1; }
  {
    const int* src = reinterpret_cast<const int*>(a.bmeta);
    int* dst = reinterpret_cast<int*>(bm);
    const int nw = nbnd * (int)(sizeof(BandMeta) / sizeof(int));
    for (int i = tid; i < nw; i += TILE) dst[i] = src[i];
  }
  __syncthreads();
  const unsigned icol = blockIdx.x * TILE + tid;
  const bool valid = icol < ncol;
  const unsigned ic = min(icol, ncol - 1);
  const unsigned cl = ic + ncol * ilay;
  const int itropo = a.tropo[cl] ? 0 : 1;
  const int jT = a.jtemp[cl];
  const int jp = a.jpress[cl] + itropo + 1;
  int regime;
  {
    const int lay1 = ilay + 1;
    const int lo1 = a.lim[ic], lo2 = a.lim[ic + ncol];
    const int up1 = a.lim[ic + 2 * (size_t)ncol], up2 = a.lim[ic + 3 * (size_t)ncol];
    regime = ((lo1 > 0 && lay1 >= lo1 && lay1 <= lo2) ? 1 : 0) | ((up1 > 0 && lay1 >= up1 && lay1 <= up2) ? 2 : 0);
  }
  const int rsel = regime == 2 ? 1 : 0;
  const int big = 1 << 30;
  {
    const int a0 = wave_min(valid ? jT : big), a1 = wave_max(valid ? jT + 1 : -1);
    const int a2 = wave_min(valid ? jp - 1 : big), a3 = wave_max(valid ? jp : -1);
    const int a4 = wave_max(valid ? (regime & 1) : 0), a5 = wave_max(valid ? (regime & 2) : 0);
    if ((tid & 63) == 0) {
      atomicMin(&rng[0], a0); atomicMax(&rng[1], a1); atomicMin(&rng[2], a2); atomicMax(&rng[3], a3);
      if (a4) rng[4] = 1;
      if (a5) rng[5] = 1;
    }
  }
  for (int b = 0; b < nbnd; ++b) {
    const int2 je = *reinterpret_cast<const int2*>(a.jeta + 2 * (cl + (size_t)ncl * bm[b].flav[itropo]));
    const int2 em = *reinterpret_cast<const int2*>(a.jeta + 2 * (cl + (size_t)ncl * bm[b].flav[rsel]));
    const int e0 = wave_min(valid ? min(min(je.x, je.y), min(em.x, em.y)) : big);
    const int e1 = wave_max(valid ? max(max(je.x, je.y), max(em.x, em.y)) + 1 : -1);
    if ((tid & 63) == 0) { atomicMin(&erng[b][0], e0); atomicMax(&erng[b][1], e1); }
  }
  __syncthreads();
  TileGeom* out = geom + (blockIdx.x + (size_t)gridDim.x * ilay);
  const int nT = rng[1] - rng[0] + 1, nP = rng[3] - rng[2] + 1;
  if (tid == 0) {
    out->Tmin = rng[0]; out->nT = nT; out->Pmin = rng[2]; out->nP = nP; out->has_lo = rng[4]; out->has_up = rng[5];
    out->pad0 = 0; out->pad1 = 0;
  }
  if (tid < nbnd) {
    const int emin = erng[tid][0], nE = erng[tid][1] - erng[tid][0] + 1;
    const int n_lo = rng[4] ? bm[tid].cnt[0] : 0, n_up = rng[5] ? bm[tid].cnt[1] : 0;
    const int rows = (nP + n_lo + n_up) * nT * nE;
    const bool fits = rows * RS <= slab_floats;
    if (!fits) {  // hand (tile, layer, band) to the direct kernel
      const int w = atomicAdd(&a.worklist[0], 1);
      a.worklist[1 + 3 * w] = blockIdx.x; a.worklist[2 + 3 * w] = ilay; a.worklist[3 + 3 * w] = tid;
    }
    out->eg[tid] = make_int2(emin, fits ? nE : 0);
  }
}

// -------------------------------------------------------------------------------------------
// Tile geometry by bit masks ("geom2"): the pre-pass of both specialised-wave kernels.
//
// The first version walked the bands, loading the band's eta indices and reducing them with 12 cross-lane
// shuffles per band -- 16 dependent load -> reduce steps per block (0.31 + 0.27 ms per step of the LW chain).
// Here every thread requests the eta indices of ALL flavors up front (4 at a time), turns each index pair into a
// bit mask of the LUT rows it touches (row r -> bit r; neta, ntemp < 31, npres + 1 < 63 checked by the host),
// and masks are OR-reduced: six DPP steps inside the wave (no LDS traffic), one LDS atomic per wave and word.
// A band's eta range is then the span of the masks of its two flavors, keyed by the regime of the columns
// that use them -- the same box as before.
// -------------------------------------------------------------------------------------------
struct Geom2Args {
  int ncol, nlay, nbnd, nflav, slab_floats;
  bool planck;               // Planck: box = pressure x temperature x eta of pfrac; no minor rows, no regime ranges
  const int* lim;            // (ncol, 4) regime layer limits (tau only)
  const int *jeta, *jtemp, *jpress;
  const Bool* tropo;
  const BandMeta* bmeta;     // tau: band flavors and minor counts
  const int *band_lims, *gpoint_flavor;  // Planck: band flavors
  const int* skip_if;        // tau: the direct kernel does the whole call
  const int* skip_if2;       // Planck: the geometry left by the compute_tau_absorption call before is valid (shared)
  int* valid_out;            // tau: set to 1 once this geometry is (being) written, for a Planck call that shares it
  int extra_planes;          // tau: more (T, eta) planes staged per stage (2 with the fused Rayleigh rows)
  int* worklist;             // tau: (tile, layer, band) triples; Planck: (tile, band) pairs
  int* flags;                // Planck: one worklist entry per (tile, band)
  const unsigned* imask;     // tau: masks per (256-column block, layer) left by the interpolation call (InterpMasks), or nullptr
  int imask_nblk;            //      blocks per layer
  const int* irregular;      //      != 0: some column's layer ranges are not those of its tropo flags -> derive the masks here
  int* stat;                 //      rte_hip_stat(2): 1 = masks taken from the interpolation call, 2 = derived here
};

template <int TILE, int G>
__global__ void __launch_bounds__(TILE) tile_geom2_kernel(Geom2Args a, TileGeom* __restrict__ geom) {
  constexpr int RS = G + 2;
  __shared__ unsigned mT, mP[2], mReg;
  __shared__ unsigned mE[MAXFLAV][2];
  __shared__ int flav[MAXB][2], cnt[MAXB][2];
  if (a.skip_if && *a.skip_if) return;
  if (a.skip_if2 && *a.skip_if2) return;
  const int tid = threadIdx.x, lane = tid & 63;
  const unsigned ncol = a.ncol, nlay = a.nlay, ilay = blockIdx.y;
  const unsigned ncl = ncol * nlay;
  const int nbnd = a.nbnd, nflav = a.nflav;
  if (a.valid_out && tid == 0 && blockIdx.x == 0 && blockIdx.y == 0) *a.valid_out = 1;  // (read by later launches only)
  if (tid == 0) { mT = 0; mP[0] = 0; mP[1] = 0; mReg = 0; }
  if (tid < 2 * MAXFLAV) mE[tid >> 1][tid & 1] = 0;
  if (tid < 2 * nbnd) {
    const int b = tid >> 1, r = tid & 1;
    if (a.planck) {
      flav[b][r] = a.gpoint_flavor[r + 2 * (a.band_lims[2 * b] - 1)] - 1;
      cnt[b][r] = 0;
    } else {
      flav[b][r] = a.bmeta[b].flav[r];
      cnt[b][r] = a.bmeta[b].cnt[r];
    }
  }
  __syncthreads();
  static_assert(TILE % 256 == 0, "the interpolation kernel leaves one mask record per 256 columns");
  const bool pre = a.imask != nullptr && *a.irregular == 0;
  if (a.stat && tid == 0 && blockIdx.x == 0 && blockIdx.y == 0) *a.stat = pre ? 1 : 2;
  if (pre && tid >= 128) return;  // two waves do the rest (4 + 2 * nflav <= 68 words; finished waves no longer count at the barrier)
  if (pre) {
    const int W = 4 + 2 * nflav;
    if (tid < W) {
      unsigned m = 0;
      for (int k = 0; k < TILE / 256; ++k) {
        const unsigned blk = blockIdx.x * (TILE / 256) + k;
        if (blk < (unsigned)a.imask_nblk) m |= a.imask[((size_t)blk + (size_t)a.imask_nblk * ilay) * W + tid];
      }
      if (tid == 0) mT = m;
      else if (tid == 1) mP[0] = m;
      else if (tid == 2) mP[1] = m;
      else if (tid == 3) mReg = m;
      else mE[(tid - 4) >> 1][(tid - 4) & 1] = m;
    }
  } else {
  const unsigned icol = blockIdx.x * TILE + tid;
  const bool valid = icol < ncol;
  const unsigned ic = min(icol, ncol - 1);
  const unsigned cl = ic + ncol * ilay;
  const int itropo = a.tropo[cl] ? 0 : 1;
  const int jT = a.jtemp[cl];
  const int jp = a.jpress[cl] + itropo + 1;  // levels jp-1, jp (1-based)
  int regime = 0;
  if (!a.planck) {
    const int lay1 = ilay + 1;
    const int lo1 = a.lim[ic], lo2 = a.lim[ic + ncol];
    const int up1 = a.lim[ic + 2 * (size_t)ncol], up2 = a.lim[ic + 3 * (size_t)ncol];
    regime = ((lo1 > 0 && lay1 >= lo1 && lay1 <= lo2) ? 1 : 0) | ((up1 > 0 && lay1 >= up1 && lay1 <= up2) ? 2 : 0);
  }
  const int rsel = regime == 2 ? 1 : 0;  // regime whose flavor the minor absorbers use
  // a column's eta rows count for the regimes whose flavor table it uses: itropo (major species, Planck
  // fractions) and rsel (minor species; differs from itropo only for non-contiguous tropo masks)
  const bool key0 = valid && (itropo == 0 || (!a.planck && rsel == 0));
  const bool key1 = valid && (itropo == 1 || (!a.planck && rsel == 1));
  {
    const unsigned long long pm = valid ? (3ull << (jp - 1)) : 0ull;
    const unsigned t = wave_or(valid ? (3u << jT) : 0u);
    const unsigned p0 = wave_or((unsigned)pm), p1 = wave_or((unsigned)(pm >> 32));
    const unsigned rg = wave_or(valid ? (unsigned)regime : 0u);
    if (lane == 0) { atomicOr(&mT, t); atomicOr(&mP[0], p0); atomicOr(&mP[1], p1); atomicOr(&mReg, rg); }
  }
  for (int f0 = 0; f0 < nflav; f0 += 4) {
    int2 je[4];
#pragma unroll
    for (int k = 0; k < 4; ++k)
      je[k] = *reinterpret_cast<const int2*>(a.jeta + 2 * (cl + (size_t)ncl * min(f0 + k, nflav - 1)));
    unsigned w[4][2];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const unsigned m = (3u << je[k].x) | (3u << je[k].y);  // rows eta, eta + 1 of both temperature corners
      w[k][0] = wave_or(key0 ? m : 0u);
      w[k][1] = wave_or(key1 ? m : 0u);
    }
    if (lane == 0) {
#pragma unroll
      for (int k = 0; k < 4; ++k)
        if (f0 + k < nflav) { atomicOr(&mE[f0 + k][0], w[k][0]); atomicOr(&mE[f0 + k][1], w[k][1]); }
    }
  }
  }  // !pre
  __syncthreads();
  TileGeom* out = geom + (blockIdx.x + (size_t)gridDim.x * ilay);
  const int Tmin = __ffs(mT) - 1, nT = (32 - __clz(mT)) - Tmin;
  const unsigned long long pmask = ((unsigned long long)mP[1] << 32) | mP[0];
  const int Pmin = __ffsll((long long)pmask) - 1, nP = (64 - __clzll((long long)pmask)) - Pmin;
  const int has_lo = mReg & 1, has_up = (mReg >> 1) & 1;
  if (tid == 0) {
    out->Tmin = Tmin; out->nT = nT; out->Pmin = Pmin; out->nP = nP; out->has_lo = has_lo; out->has_up = has_up;
    out->pad0 = 0; out->pad1 = 0;
  }
  if (tid < nbnd) {
    const unsigned me = mE[flav[tid][0]][0] | mE[flav[tid][1]][1];
    const int emin = me ? __ffs(me) - 1 : 1, nE = me ? (32 - __clz(me)) - emin : 0;
    const int n_lo = has_lo ? cnt[tid][0] : 0, n_up = has_up ? cnt[tid][1] : 0;
    const int rows = (nP + n_lo + n_up + (a.planck ? 0 : a.extra_planes)) * nT * nE;
    const bool fits = rows * RS <= a.slab_floats;
    if (a.planck) {
      if (!fits && atomicCAS(&a.flags[blockIdx.x * nbnd + tid], 0, 1) == 0) {  // once per (tile, band)
        const int w = atomicAdd(&a.worklist[0], 1);
        a.worklist[1 + 2 * w] = blockIdx.x; a.worklist[2 + 2 * w] = tid;
      }
      out->eg[tid] = make_int2(emin, nE);
    } else {
      if (!fits) {  // hand (tile, layer, band) to the direct kernel
        const int w = atomicAdd(&a.worklist[0], 1);
        a.worklist[1 + 3 * w] = blockIdx.x; a.worklist[2 + 3 * w] = ilay; a.worklist[3 + 3 * w] = tid;
      }
      // (nE <= 0: not a stage of the slab kernel; the magnitude is kept for a Planck call that shares this geometry)
      out->eg[tid] = make_int2(emin, fits ? nE : -nE);
    }
  }
}

// MM = minor intervals per (band, regime) whose column amounts are kept in registers a stage ahead (4: what the register
// budget allows without spills -- an MM = 8 instantiation spilled 4 registers, 52 in the fused SW variant).  A band with
// more intervals (the real tables are ragged: 1 ... 9 per band and regime) runs its first MM this way and the rest in a
// tail pass whose column amounts are requested where they are used (their latency is exposed, for those bands only).
// ADDB: a band-wise operand is added (rte_hip_compute_tau_absorption_inc_bybnd) -- a template parameter, not a run-time
// test: a conditional load changes the number of outstanding memory operations from path to path, and the compiler
// then waits for (nearly) all of them, i.e. for the previous stage's stores, at the top of every stage.
// RAYL: fused with compute_tau_rayleigh + combine_abs_and_rayleigh (2-stream) [+ by-band 2-stream increment]: see RaylFuse
template <int NCW, int NLW, int SLAB, bool OVERWRITE, int G, int MM, bool ADDB, int RAYL = 0 /* 1: fused, 2: + by-band clouds */>
__global__ void __launch_bounds__((NCW + NLW) * 64, V9_MINW)
tau_absorption_v9_kernel(TauV5 a, const TileGeom* __restrict__ geom) {
  constexpr int TILE = NCW * 64, NLT = NLW * 64, NT = TILE + NLT;
  constexpr int RS = G + 2, PPR = G / 2, PSH = G == 16 ? 3 : 2;  // row stride, 16-byte pieces per row, log2(PPR)
  __shared__ __align__(16) Float slab[2][SLAB];
  __shared__ TileGeom tg;
  extern __shared__ BandMeta bm[];  // [nbnd]
  if (*a.skip_if) return;
  const int tid = threadIdx.x;
  const unsigned ncol = a.ncol, nlay = a.nlay, ilay = blockIdx.y;
  const unsigned ncl = ncol * nlay;  // host guarantees < 2^31
  const int ntemp = a.ntemp, TE = a.TE, ngpt = a.ngpt, nbnd = a.nbnd;
  {
    const int* src = reinterpret_cast<const int*>(a.bmeta);
    int* dst = reinterpret_cast<int*>(bm);
    const int nw = nbnd * (int)(sizeof(BandMeta) / sizeof(int));
    for (int i = tid; i < nw; i += NT) dst[i] = src[i];
    const int* gs = reinterpret_cast<const int*>(geom + (blockIdx.x + (size_t)gridDim.x * ilay));
    int* gd = reinterpret_cast<int*>(&tg);
    for (int i = tid; i < (int)(sizeof(TileGeom) / sizeof(int)); i += NT) gd[i] = gs[i];
  }
  __syncthreads();
  const int Tmin = tg.Tmin, nT = tg.nT, Pmin = tg.Pmin, nP = tg.nP;
  const bool has_lo = tg.has_lo != 0, has_up = tg.has_up != 0;
  const int nstage = ngpt / G;  // host guarantees whole, G-aligned chunks per band

  if (tid >= TILE) {
    // ================================ loader waves ================================
    // the loaders issue little and mostly wait for memory: a raised issue priority lets their requests and LDS writes go
    // out ahead of the eight compute waves' FMAs, so that the next slab is complete a little earlier (tau 5.34 -> 5.28 ms
    // in one process, no change for Planck)
    __builtin_amdgcn_s_setprio(1);
    const int lt = tid - TILE;
    const float inv_nT = 1.0f / (float)nT;
    constexpr int SB = V9_SB;  // 16-byte pieces per lane requested back to back
    int ibnd = 0;
#pragma unroll 1
    for (int s = 0; s < nstage; ++s) {
      const int g0 = s * G;
      while (ibnd + 1 < nbnd && bm[ibnd].gE < g0) ++ibnd;
      const int emin = tg.eg[ibnd].x, nE = tg.eg[ibnd].y;
      if (nE > 0) {
        const float inv_nE = 1.0f / (float)nE;
        const int n_lo = has_lo ? bm[ibnd].cnt[0] : 0, n_up = has_up ? bm[ibnd].cnt[1] : 0;
        const int rowsMaj = nP * nT * nE, rowsLo = n_lo * nT * nE, rowsUp = n_up * nT * nE;
        const int rowsRay = RAYL ? 2 * nT * nE : 0;  // [regime][t][eta] rows of the Rayleigh table, behind the minor planes
        const int nAll = (rowsMaj + rowsLo + rowsUp + rowsRay) * (G / 2);
        Float* sl = slab[s & 1];
        // rows ordered [p][t][eta] (+ minor: [interval][t][eta]); piece = 16 bytes of a 128-byte row chunk
        auto piece = [&](int idx) -> Float2 {
          const int j = idx & (PPR - 1), r = idx >> PSH;
          if (r < rowsMaj) {
            const int rest = (int)(((float)r + 0.5f) * inv_nE), e = r - rest * nE;  // rows < 2^12: exact
            const int p_l = (int)(((float)rest + 0.5f) * inv_nT), t_l = rest - p_l * nT;
            return *reinterpret_cast<const Float2*>(
                a.kmaj + ((size_t)((Pmin - 1 + p_l) * TE + (emin - 1 + e) * ntemp + (Tmin - 1 + t_l)) * ngpt + g0 + 2 * j));
          }
          const int rm = r - rowsMaj;
          if (RAYL && rm >= rowsLo + rowsUp) {
            const int rr = rm - rowsLo - rowsUp;
            const int rest = (int)(((float)rr + 0.5f) * inv_nE), e = rr - rest * nE;
            const int k = (int)(((float)rest + 0.5f) * inv_nT), t_l = rest - k * nT;  // k: regime
            return *reinterpret_cast<const Float2*>(
                a.rf.krayl_g[k] + ((size_t)((emin - 1 + e) * ntemp + (Tmin - 1 + t_l)) * ngpt + g0 + 2 * j));
          }
          const bool up = rm >= rowsLo;
          const int rr = up ? rm - rowsLo : rm;
          const int rest = (int)(((float)rr + 0.5f) * inv_nE), e = rr - rest * nE;
          const int k = (int)(((float)rest + 0.5f) * inv_nT), t_l = rest - k * nT;
          const MinorMeta& m = bm[ibnd].m[up ? 1 : 0][k];
          const bool on = m.mS <= g0 && m.mE >= g0;  // off: any valid address, the row is never read
          const Float* kg = up ? a.kup : a.klo;
          const unsigned nk = up ? a.nk_up : a.nk_lo;
          return *reinterpret_cast<const Float2*>(
              kg + ((size_t)((emin - 1 + e) * ntemp + (Tmin - 1 + t_l)) * nk + (unsigned)m.kstart + (on ? g0 - m.mS : 0) + 2 * j));
        };
#ifdef X9_NOSTAGE
        if (a.ncol < 0)
#endif
#pragma unroll 1
        for (int base = lt; base < nAll; base += SB * NLT) {
          Float2 v[SB];
#pragma unroll
          for (int u = 0; u < SB; ++u) v[u] = piece(min(base + u * NLT, nAll - 1));
#pragma unroll
          for (int u = 0; u < SB; ++u) {
            const int idx = base + u * NLT;
            if (idx < nAll) *reinterpret_cast<Float2*>(sl + (idx >> PSH) * RS + 2 * (idx & (PPR - 1))) = v[u];
          }
        }
      }
      __syncthreads();  // B(s): slab(s) complete; the compute waves are done with the other buffer
    }
    return;
  }

  // ================================ compute waves (lanes = columns) ================================
  const unsigned icol = blockIdx.x * TILE + tid;
  const bool valid = icol < ncol;
  const unsigned ic = min(icol, ncol - 1);
  const unsigned cl = ic + ncol * ilay;
  const unsigned cl8 = cl * (unsigned)sizeof(Float);
  const int itropo = a.tropo[cl] ? 0 : 1;
  const int jT = a.jtemp[cl];
  const int jp = a.jpress[cl] + itropo + 1;  // levels jp-1, jp (1-based)
  int regime;
  {
    const int lay1 = ilay + 1;
    const int lo1 = a.lim[ic], lo2 = a.lim[ic + ncol];
    const int up1 = a.lim[ic + 2 * (size_t)ncol], up2 = a.lim[ic + 3 * (size_t)ncol];
    regime = ((lo1 > 0 && lay1 >= lo1 && lay1 <= lo2) ? 1 : 0) | ((up1 > 0 && lay1 >= up1 && lay1 <= up2) ? 2 : 0);
  }
  const int rsel = regime == 2 ? 1 : 0;
  const Float P = a.play[cl], T = a.tlay[cl];
  const Float dens = (Float)0.01 * P / T;                                                             // :469
  const Float vmr_fact = (Float)1 / a.col_gas[cl];                                                    // :471
  const Float dry_fact = (Float)1 / ((Float)1 + a.col_gas[cl + (size_t)ncl * a.idx_h2o] * vmr_fact);  // :472
  Float wray = 0;  // Rayleigh: column amount of moist air (:553)
  if (RAYL) wray = a.col_gas[cl + (size_t)ncl * a.idx_h2o] + a.rf.col_dry[cl];
  // The fused variants are a few registers over the budget, and a register spilled to scratch is reloaded with
  // `s_waitcnt vmcnt(0)` -- in the middle of the minor pass that drains the previous stage's 48 stores (vector memory
  // retires in order).  These three per-column factors are used a few times per stage only: park them in the thread's
  // own LDS slots instead (a `ds_read` waits on lgkmcnt).  The unfused variants keep them in registers.
  constexpr bool PARK = RAYL != 0;  // (the variants that would otherwise spill)
  constexpr int NPARK = PARK ? 3 : 0;
  __shared__ Float s_park[NPARK ? NPARK : 1][NPARK ? TILE : 1];
  unsigned park_at = 0;  // LDS byte address of this thread's first slot (the low half of the generic address)
  if constexpr (PARK) {
    s_park[0][tid] = dens; s_park[1][tid] = vmr_fact; s_park[2][tid] = dry_fact;
    park_at = (unsigned)(uintptr_t)&s_park[0][tid];
  }
  // read back with an explicit ds_read (a volatile access from inside the stage lambdas becomes a flat load, and an
  // ordinary one is hoisted back into a register)
  auto parked = [](unsigned at, int i) -> Float {
    Float v;
    if constexpr (sizeof(Float) == 8)
      asm volatile("ds_read_b64 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(at + (unsigned)(i * NCW * 64 * sizeof(Float))) : "memory");
    else
      asm volatile("ds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(at + (unsigned)(i * NCW * 64 * sizeof(Float))) : "memory");
    return v;
  };
#define RTE_PARKED(i, in_register) (PARK ? parked(park_at, i) : (in_register))

  // major weights + eta indices of band b (requested one stage ahead)
  struct Major { Float2 fm[4], cm; int2 je; };
  auto load_major = [&](int flav, Major& x) {
    const size_t clf = cl + (size_t)ncl * flav;
    const Float2* fmp = reinterpret_cast<const Float2*>(a.fmajor + 8 * clf);
#pragma unroll
    for (int i = 0; i < 4; ++i) x.fm[i] = fmp[i];
    x.cm = *reinterpret_cast<const Float2*>(a.col_mix + 2 * clf);
    x.je = *reinterpret_cast<const int2*>(a.jeta + 2 * clf);
  };
  // minor column amounts, weights and eta indices of one stage
  struct Minor { Float sc[MM], cgs[MM]; Float2 fn0, fn1; int2 em; Float addv; };
  // What the requests of a stage's minor inputs need from the band table in LDS: which gases, which flavor.  Read at
  // the TOP of the stage before (peek_minor), so that at its end the requests go out back to back: looked up there,
  // each request waited for its own LDS round trip -- eleven in a row, with nothing else left to issue (0.8 ms).
  struct MinorIdx { int idx[MM], isc[MM], flav, flav_major, n; };
  auto peek_minor = [&](int b, int n, MinorIdx& q) {
    q.n = n;
#pragma unroll
    for (int k = 0; k < MM; ++k) {
      const MinorMeta& m = bm[b].m[rsel][k];  // (slots past the band's count are zero-filled: never used)
      q.idx[k] = m.idx_minor;
      q.isc[k] = ((m.flags & 1) && m.idx_scaling > 0) ? m.idx_scaling : -1;
      asm volatile("" : "+v"(q.idx[k]), "+v"(q.isc[k]));  // looked up here, not where they are used
    }
    q.flav = bm[b].flav[rsel];  // minor absorbers use THEIR regime's flavor (:487)
    q.flav_major = bm[b].flav[itropo];
    asm volatile("" : "+v"(q.flav), "+v"(q.flav_major));
  };
  auto load_minor = [&](int b, const MinorIdx& q, Minor& x) {
    x.addv = ADDB ? a.add_bybnd[cl + (size_t)ncl * b] : (Float)0;

#pragma unroll
    for (int k = 0; k < MM; ++k) {
      x.sc[k] = 0; x.cgs[k] = 0;
      if (k < q.n) {
        x.sc[k] = a.col_gas[cl + (size_t)ncl * q.idx[k]];
        if (q.isc[k] >= 0) x.cgs[k] = a.col_gas[cl + (size_t)ncl * q.isc[k]];
      }
    }
  };
  // the minor interpolation weights and eta indices go with the major ones (after the major pass), into registers of
  // their own: left to the end of the stage with the column amounts they were on the stage's critical path
  auto load_minor_w = [&](const MinorIdx& q, Minor& x) {
    const size_t clm = cl + (size_t)ncl * q.flav;
    const Float2* fnp = reinterpret_cast<const Float2*>(a.fminor + 4 * clm);
    x.fn0 = fnp[0]; x.fn1 = fnp[1];
    x.em = *reinterpret_cast<const int2*>(a.jeta + 2 * clm);
  };
  auto n_minor = [&](int b) { return (tg.eg[b].y > 0 && regime > 0) ? bm[b].cnt[rsel] : 0; };
  // Vector-memory operations of a wave retire IN ORDER, stores included: a request issued after a stage's 16 tau
  // stores is served only when those have drained.  The minor weights and column amounts of stage s+1 are
  // therefore requested at the END of stage s, just BEFORE its stores (the stores then drain behind them while
  // stage s+1 gathers its major species), and the major weights of stage s+1 after the major pass of stage s.
  // The order below is only kept by the compiler's wait-count pass when every path through the loop issues the
  // same memory operations: blocks in which some band goes to the direct kernel (no stores for its stages) run a
  // second instance of the loop (ALLRUN = false) that pays the drains.
  bool all_run = true;
  for (int b = 0; b < nbnd; ++b) all_run = all_run && tg.eg[b].y > 0;
  auto run_stages = [&](auto allrun_tag) {
  constexpr bool ALLRUN = decltype(allrun_tag)::value;
  Major mj;
  Minor mn;
  MinorIdx nq;
  Minor mw;  // (only fn0, fn1, em are used: the next stage's)
  peek_minor(0, n_minor(0), nq);
  load_major(nq.flav_major, mj);
  load_minor(0, nq, mn);
  load_minor_w(nq, mw);
  // Nothing outstanding when the loop is entered: the wait counts inside it are then those of the steady state
  // (requests of stage s+1, then the stores of stage s) and not the merge with this prologue, which made every stage
  // wait for all but three of the previous stage's stores.
  __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0)
  int ibnd = 0;
#pragma unroll 1
  for (int s = 0; s < nstage; ++s) {
    const int g0 = s * G;
    while (ibnd + 1 < nbnd && bm[ibnd].gE < g0) ++ibnd;
    const int emin = tg.eg[ibnd].x, nE = tg.eg[ibnd].y;
    int ibnd_n = ibnd;
    if (s + 1 < nstage) while (ibnd_n + 1 < nbnd && bm[ibnd_n].gE < g0 + G) ++ibnd_n;
    const bool run = nE > 0;  // block-uniform
    const int n_my = n_minor(ibnd);
    Float sc[MM], cgs[MM];
#pragma unroll
    for (int k = 0; k < MM; ++k) { sc[k] = mn.sc[k]; cgs[k] = mn.cgs[k]; }
    const Float2 fn0 = mw.fn0, fn1 = mw.fn1;
    const int2 em = mw.em;
    const Float addv = mn.addv;
    // RAYL: what only the end of the stage needs -- the Rayleigh interpolation weights (fminor of the MAJOR species'
    // flavor, :548-551) and the band's cloud properties -- is requested here, at the top of its own stage
    Float2 fr0{}, fr1{};
    Float cld_t = 0, cld_s = 0, cld_g = 0;
    if (RAYL) {
      const int flav_cur = bm[ibnd].flav[itropo];
      const Float2* frp = reinterpret_cast<const Float2*>(a.fminor + 4 * (cl + (size_t)ncl * flav_cur));
      fr0 = frp[0]; fr1 = frp[1];
    }
    if (RAYL == 2) {
      cld_t = a.rf.cld_tau[cl + (size_t)ncl * ibnd]; cld_s = a.rf.cld_ssa[cl + (size_t)ncl * ibnd];
      cld_g = a.rf.cld_g[cl + (size_t)ncl * ibnd];
    }
    // this stage's major weights into locals (col_mix folded in)
    const Float w0 = mj.cm.x * mj.fm[0].x, w1 = mj.cm.x * mj.fm[0].y, w2 = mj.cm.x * mj.fm[1].x, w3 = mj.cm.x * mj.fm[1].y,
                w4 = mj.cm.y * mj.fm[2].x, w5 = mj.cm.y * mj.fm[2].y, w6 = mj.cm.y * mj.fm[3].x, w7 = mj.cm.y * mj.fm[3].y;
    const int je1 = mj.je.x, je2 = mj.je.y;
    __syncthreads();  // B(s): slab(s) is complete
    peek_minor(ibnd_n, n_minor(ibnd_n), nq);
    if (!ALLRUN && !run) {
      load_major(nq.flav_major, mj);
      load_minor_w(nq, mw);
      load_minor(ibnd_n, nq, mn);
      continue;
    }
    const Float* sl = slab[s & 1];
    const int rowsMaj = nP * nT * nE;
    const int rowsLo = (has_lo ? bm[ibnd].cnt[0] : 0) * nT * nE;
    const Float* A0 = sl + (((jp - 1 - Pmin) * nT + (jT - Tmin)) * nE + (je1 - emin)) * RS;
    const Float* B0 = sl + (((jp - 1 - Pmin) * nT + (jT + 1 - Tmin)) * nE + (je2 - emin)) * RS;
    const int sP = nT * nE * RS;
    const Float* M0 = sl + (rowsMaj + (regime == 2 ? rowsLo : 0)) * RS;
    Float acc[G];
    // tau(:, :, g) = scalar plane base + this column's 32-bit byte offset (host guarantees 8*ncol*nlay < 2^32)
    char* const tplane = reinterpret_cast<char*>(a.tau + (size_t)ncl * g0);
    const size_t gstride = (size_t)ncl * sizeof(Float);
    unsigned toff = cl8;
    asm volatile("" : "+v"(toff));  // keep the 64-bit address out of the loop-invariant registers
    auto tau_at = [&](int j) { return reinterpret_cast<Float*>(tplane + gstride * j + toff); };
#pragma unroll
    for (int j = 0; j < G; ++j) acc[j] = 0;
#pragma unroll
    for (int j = 0; j < G; j += 2) {
      // :791-801 with col_mix folded into the weights; one 16-byte LDS read feeds two g-points
#ifdef X9_NOGATHER
      const Float2 k0{w1, w2}, k1{w2, w3}, k2{w3, w4}, k3{w4, w5}, k4{w5, w6}, k5{w6, w7}, k6{w7, w0}, k7{w0, w1};
#else
      const Float2 k0 = ld2(A0 + j), k1 = ld2(A0 + RS + j), k2 = ld2(A0 + sP + j), k3 = ld2(A0 + sP + RS + j),
                   k4 = ld2(B0 + j), k5 = ld2(B0 + RS + j), k6 = ld2(B0 + sP + j), k7 = ld2(B0 + sP + RS + j);
#endif
      Float m = w0 * k0.x, n = w0 * k0.y;
      m = fma(w1, k1.x, m); n = fma(w1, k1.y, n);
      m = fma(w2, k2.x, m); n = fma(w2, k2.y, n);
      m = fma(w3, k3.x, m); n = fma(w3, k3.y, n);
      m = fma(w4, k4.x, m); n = fma(w4, k4.y, n);
      m = fma(w5, k5.x, m); n = fma(w5, k5.y, n);
      m = fma(w6, k6.x, m); n = fma(w6, k6.y, n);
      m = fma(w7, k7.x, m); n = fma(w7, k7.y, n);
      acc[j] = acc[j] + m;
      acc[j + 1] = acc[j + 1] + n;
      // pin the accumulation here: otherwise the FMA chains are sunk below the whole loop and all 64 reads stay live
      asm volatile("" : "+v"(acc[j]), "+v"(acc[j + 1]));
      if ((j & 2) != 0) __builtin_amdgcn_sched_barrier(0);  // at most 16 row reads (64 VGPRs) in flight
    }
    // next stage's major weights: their registers are free now, and the request is a minor pass ahead of its use
    // (requested with the minor weights at the end of the stage, their latency is exposed: 5.5 -> 5.9 ms)
#if defined(X9_NOLOAD) || defined(X9_NOLOAD_MAJ)
    if (a.ncol < 0)
#endif
    load_major(nq.flav_major, mj);
#if defined(X9_NOLOAD) || defined(X9_NOLOAD_MIN)
    if (a.ncol < 0)
#endif
    load_minor_w(nq, mw);
    __builtin_amdgcn_sched_barrier(0);
    // ---- minor absorbers of this regime; scalings (:461-480)
#pragma unroll
    for (int k = 0; k < MM; ++k) {
      if (k < n_my) {
        const MinorMeta& m = bm[ibnd].m[rsel][k];
        if (m.flags & 1) {
          sc[k] = sc[k] * RTE_PARKED(0, dens);  // :469
          if (m.idx_scaling > 0) {          // :470-478
            if (m.flags & 2)
              sc[k] = sc[k] * ((Float)1 - cgs[k] * RTE_PARKED(1, vmr_fact) * RTE_PARKED(2, dry_fact));
            else
              sc[k] = sc[k] * (cgs[k] * RTE_PARKED(1, vmr_fact) * RTE_PARKED(2, dry_fact));
          }
        }
      }
    }
#ifdef X9_EARLY_SC
#if defined(X9_NOLOAD) || defined(X9_NOLOAD_MIN)
    if (a.ncol < 0)
#endif
    load_minor(ibnd_n, nq, mn);  // this stage's amounts are scaled copies by now
    __builtin_amdgcn_sched_barrier(0);
#endif
    const Float f0 = fn0.x, f1 = fn0.y, f2 = fn1.x, f3 = fn1.y;
    // one minor interval's contribution (:757-760, :493): 4 corner rows of its plane, 2 g-points per LDS read
    auto minor_rows = [&](int k, Float scaling) {
      const Float* r1 = M0 + ((k * nT + (jT - Tmin)) * nE + (em.x - emin)) * RS;
      const Float* r2 = M0 + ((k * nT + (jT + 1 - Tmin)) * nE + (em.y - emin)) * RS;
#pragma unroll
      for (int j = 0; j < G; j += 2) {
#ifdef X9_NOGATHER
        const Float2 q0{f1, f2}, q1{f2, f3}, q2{f3, scaling}, q3{scaling, f0};
        (void)r1; (void)r2;
#else
        const Float2 q0 = ld2(r1 + j), q1 = ld2(r1 + RS + j), q2 = ld2(r2 + j), q3 = ld2(r2 + RS + j);
#endif
        Float s_ = f0 * q0.x, t_ = f0 * q0.y;
        s_ = fma(f1, q1.x, s_); t_ = fma(f1, q1.y, t_);
        s_ = fma(f2, q2.x, s_); t_ = fma(f2, q2.y, t_);
        s_ = fma(f3, q3.x, s_); t_ = fma(f3, q3.y, t_);
        acc[j] = fma(scaling, s_, acc[j]);
        acc[j + 1] = fma(scaling, t_, acc[j + 1]);
        asm volatile("" : "+v"(acc[j]), "+v"(acc[j + 1]));
        if ((j & 6) == 6) __builtin_amdgcn_sched_barrier(0);  // at most 16 row reads in flight
      }
    };
    const int n_reg = n_my < MM ? n_my : MM;
#pragma unroll 1
    for (int k = 0; k < n_reg; ++k) {
      const MinorMeta& mm = bm[ibnd].m[rsel][k];
      if (mm.mE < g0 || mm.mS > g0) continue;  // intervals are whole 16-aligned chunks inside the band
      Float scaling = sc[0];
#pragma unroll
      for (int q = 1; q < MM; ++q) scaling = (k == q) ? sc[q] : scaling;
      minor_rows(k, scaling);
    }
    if (n_my > MM) {
      // the band's intervals beyond the MM held in registers: amounts requested here, same expressions (:461-480)
#pragma unroll 1
      for (int k = MM; k < n_my; ++k) {
        const MinorMeta& mm = bm[ibnd].m[rsel][k];
        if (mm.mE < g0 || mm.mS > g0) continue;
        Float scaling = a.col_gas[cl + (size_t)ncl * mm.idx_minor];
        if (mm.flags & 1) {
          scaling = scaling * RTE_PARKED(0, dens);  // :469
          if (mm.idx_scaling > 0) {                 // :470-478
            const Float cg = a.col_gas[cl + (size_t)ncl * mm.idx_scaling];
            if (mm.flags & 2)
              scaling = scaling * ((Float)1 - cg * RTE_PARKED(1, vmr_fact) * RTE_PARKED(2, dry_fact));
            else
              scaling = scaling * (cg * RTE_PARKED(1, vmr_fact) * RTE_PARKED(2, dry_fact));
          }
        }
        minor_rows(k, scaling);
      }
    }
#ifndef X9_EARLY_SC
#if defined(X9_NOLOAD) || defined(X9_NOLOAD_MIN)
    if (a.ncol < 0)
#endif
    load_minor(ibnd_n, nq, mn);
#endif
    __builtin_amdgcn_sched_barrier(0);  // keep these requests ahead of the stores that follow
#ifdef X9_NOSTORE
    {
      Float t_ = 0;
#pragma unroll
      for (int j = 0; j < G; ++j) t_ += acc[j];
      if (t_ == (Float)-1.2345) *tau_at(0) = t_;
    }
    if (a.ncol < 0)
#endif
    if constexpr (RAYL != 0) {
      // compute_tau_rayleigh (:548-555: interpolate2D with the reference's association) on the staged table rows,
      // combine_abs_and_rayleigh and the optional by-band increment on the values in registers (rayl_finish), and
      // the stage's 3 x G stores.  Rows [regime][t][eta] behind the minor planes; unconditional stores as below.
      const int rowsUp_ = (has_up ? bm[ibnd].cnt[1] : 0) * nT * nE;
      const Float* R1 = sl + (rowsMaj + rowsLo + rowsUp_ + ((itropo * nT + (jT - Tmin)) * nE + (je1 - emin))) * RS;
      const Float* R2 = sl + (rowsMaj + rowsLo + rowsUp_ + ((itropo * nT + (jT + 1 - Tmin)) * nE + (je2 - emin))) * RS;
      char* const splane = reinterpret_cast<char*>(a.rf.ssa + (size_t)ncl * g0);
      char* const gplane = reinterpret_cast<char*>(a.rf.g + (size_t)ncl * g0);
#pragma unroll
      for (int j = 0; j < G; j += 2) {
        const Float2 a0 = ld2(R1 + j), a1 = ld2(R1 + RS + j), b0 = ld2(R2 + j), b1 = ld2(R2 + RS + j);
        const Float ka = fr0.x * a0.x + fr0.y * a1.x + fr1.x * b0.x + fr1.y * b1.x;
        const Float kb = fr0.x * a0.y + fr0.y * a1.y + fr1.x * b0.y + fr1.y * b1.y;
#pragma unroll
        for (int u = 0; u < 2; ++u) {
          Float t_, s_, g_;
          rayl_finish(acc[j + u], (u == 0 ? ka : kb) * wray, RAYL == 2, cld_t, cld_s, cld_g, t_, s_, g_);
          store_stream(tau_at(j + u), t_);
          store_stream(reinterpret_cast<Float*>(splane + gstride * (j + u) + toff), s_);
          store_stream(reinterpret_cast<Float*>(gplane + gstride * (j + u) + toff), g_);
        }
      }
    } else if (OVERWRITE) {
      if (ADDB) {  // by-band increment fused in (tau = tau_gas + tau_2 of the band)
#pragma unroll
        for (int j = 0; j < G; ++j) acc[j] = acc[j] + addv;
      }
      // lanes past the last column repeat it (ic is clamped): same values to the same addresses.  Unconditional
      // stores keep the count of outstanding memory operations static (counted waits instead of drains).
#pragma unroll
      for (int j = 0; j < G; ++j) {
        store_stream(tau_at(j), acc[j]);
      }
    } else if (valid) {
      // tau is inout (the reference accumulates onto it, :637,:679).  The stage's sum is added to the incoming
      // value at the end: identical to the reference when tau comes in as zero (always, in the frontend),
      // otherwise the same terms in a different order (1 ulp)
      // ... as a hardware floating-point atomic add performed in L2 (global_atomic_add, no return value): the same
      // single addition tau_in + sum, but the wave neither waits for tau_in nor holds it in registers (a load - add -
      // store sequence needed `vmcnt(0)` 15 times per stage).  Every element is touched by exactly one thread of
      // one block per call, so the result does not depend on any order.
      if (ADDB) {
#pragma unroll
        for (int j = 0; j < G; ++j) acc[j] = acc[j] + addv;
      }
#ifdef X9_RMW
      const bool use_atomics = false;
#else
      const bool use_atomics = a.atomic_ok;  // host-visible (pinned / managed) buffers: load - add - store
#endif
      if (use_atomics) {
#pragma unroll
        for (int j = 0; j < G; ++j) unsafeAtomicAdd(tau_at(j), acc[j]);
      } else {
#pragma unroll
        for (int j = 0; j < G; ++j) acc[j] = *tau_at(j) + acc[j];
#pragma unroll
        for (int j = 0; j < G; ++j) *tau_at(j) = acc[j];
      }
    }
  }
  };
  if (all_run) run_stages(std::true_type{}); else run_stages(std::false_type{});
}

// the few flag / counter words a call needs zeroed, in ONE launch (each hipMemsetAsync is a launch of its own)
__global__ void __launch_bounds__(256) zero_words_kernel(int* a, unsigned na, int* b, unsigned nb, int* c, unsigned nc) {
  for (unsigned i = blockIdx.x * 256 + threadIdx.x; i < na + nb + nc; i += gridDim.x * 256) {
    if (i < na) a[i] = 0;
    else if (i < na + nb) b[i - na] = 0;
    else c[i - na - nb] = 0;
  }
}

// (tile, layer, band) triples the slab kernel could not hold, done by the direct-gather code
// (work item = one entry x one 64-column chunk, taken by waves in grid stride: the few hundred entries of a call
// spread over all CUs instead of one block walking an entry's 512 columns)
// (<= 168 registers: three waves per SIMD, so that a wave of it fits beside two waves of the slab kernel's blocks)
template <bool GFAST>
__global__ void __launch_bounds__(256, 3) tau_absorption_worklist_kernel(TauArgs a, GfastTabs gt, const int* __restrict__ worklist,
                                                                      int tile, int* __restrict__ stat) {
  const int n = worklist[0];
  if (blockIdx.x == 0 && threadIdx.x == 0) *stat = n;  // rte_hip_stat(0)
  const int chunks = tile / 64;
  const int items = n * chunks;
  const int wpb = blockDim.x >> 6;  // 4 waves per block after the slab kernel, 1 beside it (to fit next to its blocks)
  for (int it = blockIdx.x * wpb + (threadIdx.x >> 6); it < items; it += gridDim.x * wpb) {
    const int w = it / chunks, ch = it - w * chunks;
    const int icol = worklist[1 + 3 * w] * tile + ch * 64 + (threadIdx.x & 63);
    if (icol >= a.ncol) continue;
    if constexpr (GFAST) tau_direct_column_g(a, gt, icol, worklist[2 + 3 * w], worklist[3 + 3 * w]);
    else tau_direct_column(a, icol, worklist[2 + 3 * w], worklist[3 + 3 * w]);
  }
}

// -------------------------------------------------------------------------------------------
// compute_tau_rayleigh: reference :506-565
// -------------------------------------------------------------------------------------------
__device__ __forceinline__ void rayl_store(const RaylCombine& cb, Float* tau_rayleigh, size_t idx, size_t idx_bnd, Float tr) {
  if (cb.tau_abs == nullptr) { tau_rayleigh[idx] = tr; return; }
  const bool cld = cb.cld_tau != nullptr;
  Float t, s_, g_;
  rayl_finish(cb.tau_abs[idx], tr, cld, cld ? cb.cld_tau[idx_bnd] : (Float)0, cld ? cb.cld_ssa[idx_bnd] : (Float)0,
              cld ? cb.cld_g[idx_bnd] : (Float)0, t, s_, g_);
  cb.ssa[idx] = s_;
  cb.tau[idx] = t;
  cb.g[idx] = g_;
}

// direct kernel: work items (column tile, layer, band) in grid stride (a small grid when it only stands by for the plan guard)
__global__ void __launch_bounds__(256)
tau_rayleigh_kernel(int ncol, int nlay, int nbnd, int ngpt, int neta, int ntemp, int idx_h2o,
                    const int* __restrict__ gpoint_flavor, const int* __restrict__ band_lims_gpt,
                    const Float* __restrict__ krayl, const Float* __restrict__ col_dry,
                    const Float* __restrict__ col_gas, const Float* __restrict__ fminor,
                    const int* __restrict__ jeta, const Bool* __restrict__ tropo,
                    const int* __restrict__ jtemp, Float* __restrict__ tau_rayleigh, RaylCombine cb,
                    const int* __restrict__ run_if, const int* __restrict__ worklist = nullptr, int wl_tile = 0) {
  if (run_if && *run_if == 0) return;
  const unsigned tiles_x = (ncol + 255) / 256;
  // worklist != nullptr: only the (tile of wl_tile columns, layer, band) triples listed (the entries the fused gas-optics
  // kernel left to the direct-gather code)
  const int chunks = worklist ? wl_tile / 256 : 1;
  const size_t total = worklist ? (size_t)worklist[0] * chunks : (size_t)tiles_x * nlay * nbnd;
  for (size_t wi = blockIdx.x; wi < total; wi += gridDim.x) {
    int icol, ilay, ibnd;
    if (worklist) {
      const size_t w = wi / chunks;
      icol = worklist[1 + 3 * w] * wl_tile + (int)(wi - w * chunks) * 256 + threadIdx.x;
      ilay = worklist[2 + 3 * w]; ibnd = worklist[3 + 3 * w];
    } else {
      icol = (int)(wi % tiles_x) * 256 + threadIdx.x;
      ilay = (int)((wi / tiles_x) % nlay); ibnd = (int)(wi / ((size_t)tiles_x * nlay));
    }
    if (icol >= ncol) continue;
    const size_t ncl = (size_t)ncol * nlay;
    const size_t cl = icol + (size_t)ncol * ilay;
    const int gptS = band_lims_gpt[2 * ibnd] - 1, gptE = band_lims_gpt[2 * ibnd + 1] - 1;
    const int itropo = tropo[cl] ? 0 : 1;
    const int iflav = gpoint_flavor[itropo + 2 * gptS] - 1;
    const size_t clf = cl + ncl * iflav;
    const Float f0 = fminor[4 * clf], f1 = fminor[4 * clf + 1], f2 = fminor[4 * clf + 2], f3 = fminor[4 * clf + 3];
    const int je1 = jeta[2 * clf], je2 = jeta[2 * clf + 1];
    const int jT = jtemp[cl];
    const size_t tn = (size_t)ntemp * neta;
    const Float* kr = krayl + tn * ngpt * (size_t)itropo;
    const size_t o1 = (size_t)(jT - 1) + (size_t)ntemp * (je1 - 1);
    const size_t o2 = (size_t)jT + (size_t)ntemp * (je2 - 1);
    const Float w = col_gas[cl + ncl * idx_h2o] + col_dry[cl];
    for (int g = gptS; g <= gptE; ++g) {
      const Float* kk = kr + tn * (size_t)g;
      const Float k = f0 * kk[o1] + f1 * kk[o1 + ntemp] + f2 * kk[o2] + f3 * kk[o2 + ntemp];
      rayl_store(cb, tau_rayleigh, cl + ncl * (size_t)g, cl + ncl * (size_t)ibnd, k * w);
    }
  }
}

// -------------------------------------------------------------------------------------------
// compute_Planck_source: reference :568-710 (+ interpolate1D :715-737)
// -------------------------------------------------------------------------------------------
__device__ __forceinline__ Float planck_1d(Float val, Float offset, Float delta_r, const Float* __restrict__ table,
                                           int ntab) {
  const Float val0 = (val - offset) * delta_r;
  const Float frac = val0 - trunc(val0);
  const int index = min(ntab - 1, max(1, (int)val0 + 1));  // 1-based
  const Float t0 = table[index - 1], t1 = table[index];
  return t0 + frac * (t1 - t0);
}

struct PlanckArgs {
  int ncol, nlay, ngpt, neta, npres, ntemp, nPlanckTemp, sfc_lay;
  const Float *tlay, *tlev, *tsfc, *fmajor;
  const int* jeta;
  const Bool* tropo;
  const int *jtemp, *jpress, *band_lims_gpt;
  const Float* pfracin;
  Float temp_ref_min, totplnk_delta_r;
  const Float* totplnk;
  const int* gpoint_flavor;
  Float *sfc_src, *lay_src, *lev_src, *sfc_source_Jac;
};

// one column, one band, native table layout: always applicable
__device__ __forceinline__ void planck_direct_column(const PlanckArgs& q, const int icol, const int ibnd) {
  const int ncol = q.ncol, nlay = q.nlay, neta = q.neta, npres = q.npres, ntemp = q.ntemp,
            nPlanckTemp = q.nPlanckTemp, sfc_lay = q.sfc_lay;
  const Float *tlay = q.tlay, *tlev = q.tlev, *tsfc = q.tsfc, *fmajor = q.fmajor, *pfracin = q.pfracin, *totplnk = q.totplnk;
  const int *jeta = q.jeta, *jtemp = q.jtemp, *jpress = q.jpress, *band_lims_gpt = q.band_lims_gpt,
            *gpoint_flavor = q.gpoint_flavor;
  const Bool* tropo = q.tropo;
  const Float temp_ref_min = q.temp_ref_min, totplnk_delta_r = q.totplnk_delta_r;
  Float *sfc_src = q.sfc_src, *lay_src = q.lay_src, *lev_src = q.lev_src, *sfc_source_Jac = q.sfc_source_Jac;
  const size_t ncl = (size_t)ncol * nlay, nclv = (size_t)ncol * (nlay + 1);
  const int gptS = band_lims_gpt[2 * ibnd] - 1, gptE = band_lims_gpt[2 * ibnd + 1] - 1;
  const Float* tp = totplnk + (size_t)nPlanckTemp * ibnd;
  const size_t tn = (size_t)ntemp * neta;
  const size_t gstride = tn * (npres + 1);
  // :641-656 surface Planck function at tsfc and tsfc + 1 K
  const Float pl_sfc = planck_1d(tsfc[icol], temp_ref_min, totplnk_delta_r, tp, nPlanckTemp);
  const Float pl_sfc1 = planck_1d(tsfc[icol] + (Float)1, temp_ref_min, totplnk_delta_r, tp, nPlanckTemp);

  for (int g0 = gptS; g0 <= gptE; g0 += GC) {
    Float pf_prev[GC];
#pragma unroll
    for (int j = 0; j < GC; ++j) pf_prev[j] = 0;
    for (int ilay = 0; ilay < nlay; ++ilay) {
      const size_t cl = icol + (size_t)ncol * ilay;
      const int itropo = tropo[cl] ? 0 : 1;
      const int iflav = gpoint_flavor[itropo + 2 * gptS] - 1;
      const size_t clf = cl + ncl * iflav;
      const int jT = jtemp[cl];
      const int jp = jpress[cl] + itropo + 1;
      const int je1 = jeta[2 * clf], je2 = jeta[2 * clf + 1];
      Float fm[8];
#pragma unroll
      for (int i = 0; i < 8; ++i) fm[i] = fmajor[8 * clf + i];
      const size_t a0 = (size_t)(jT - 1) + (size_t)ntemp * (je1 - 1) + tn * (size_t)(jp - 2);
      const size_t b0 = (size_t)jT + (size_t)ntemp * (je2 - 1) + tn * (size_t)(jp - 2);
      const Float pl_lay = planck_1d(tlay[cl], temp_ref_min, totplnk_delta_r, tp, nPlanckTemp);
      const Float pl_lev = planck_1d(tlev[icol + (size_t)ncol * ilay], temp_ref_min, totplnk_delta_r, tp, nPlanckTemp);
#pragma unroll
      for (int j = 0; j < GC; ++j) {
        const int g = g0 + j;
        if (g <= gptE) {
          const Float* ka = pfracin + gstride * (size_t)g + a0;
          const Float* kb = pfracin + gstride * (size_t)g + b0;
          // interpolate3D_byflav with scaling = (1,1), :791-801
          const Float pf =
              (Float)1 * (fm[0] * ka[0] + fm[1] * ka[ntemp] + fm[2] * ka[tn] + fm[3] * ka[tn + ntemp]) +
              (Float)1 * (fm[4] * kb[0] + fm[5] * kb[ntemp] + fm[6] * kb[tn] + fm[7] * kb[tn + ntemp]);
          lay_src[cl + ncl * (size_t)g] = pf * pl_lay;                                   // :674
          const Float lv = (ilay == 0) ? pf : sqrt(pf_prev[j] * pf);                      // :695,:699
          lev_src[icol + (size_t)ncol * ilay + nclv * (size_t)g] = lv * pl_lev;
          if (ilay == sfc_lay - 1) {                                                      // :651-653
            sfc_src[icol + (size_t)ncol * g] = pf * pl_sfc;
            sfc_source_Jac[icol + (size_t)ncol * g] = pf * (pl_sfc1 - pl_sfc);
          }
          pf_prev[j] = pf;
        }
      }
    }
    const Float pl_top = planck_1d(tlev[icol + (size_t)ncol * nlay], temp_ref_min, totplnk_delta_r, tp, nPlanckTemp);
#pragma unroll
    for (int j = 0; j < GC; ++j)
      if (g0 + j <= gptE) lev_src[icol + (size_t)ncol * nlay + nclv * (size_t)(g0 + j)] = pf_prev[j] * pl_top;  // :705
  }
}

__global__ void __launch_bounds__(256) planck_source_kernel(PlanckArgs q, const int* __restrict__ run_if) {
  if (run_if && *run_if == 0) return;
  const int icol = blockIdx.x * blockDim.x + threadIdx.x;
  if (icol < q.ncol) planck_direct_column(q, icol, blockIdx.y);
}

// (tile, band) pairs the slab kernel handed over (worklist[0] = count)
__global__ void __launch_bounds__(256)
planck_source_worklist_kernel(PlanckArgs q, const int* __restrict__ worklist, int tile, int* __restrict__ stat) {
  const int n = worklist[0];
  if (blockIdx.x == 0 && threadIdx.x == 0) *stat = n;  // rte_hip_stat(1)
  for (int w = blockIdx.x; w < n; w += gridDim.x)
    for (int c = threadIdx.x; c < tile; c += 256) {
      const int icol = worklist[1 + 2 * w] * tile + c;
      if (icol < q.ncol) planck_direct_column(q, icol, worklist[2 + 2 * w]);
    }
}

// -------------------------------------------------------------------------------------------
// compute_Planck_source, production kernel: same scheme as tau_absorption_v7_kernel.
// block = (256 columns, one band) and walks the LAYERS, so the previous layer's Planck fractions
// stay in registers for the geometric mean at the interface (:699).  Per layer the tile's
// bounding box of pfrac rows is staged in LDS from the g-fastest table; the band's totplnk column
// sits in LDS for the whole block.  Interpolation state of layer l+1 is requested while layer l
// is computed (two-deep: indices two layers ahead, flavor-dependent weights one layer ahead).
// -------------------------------------------------------------------------------------------
struct PlanckV7 {
  int ncol, nlay, ngpt, ntemp, TE, nPlanckTemp, sfc_lay;
  Float temp_ref_min, totplnk_delta_r;
  const int *band_lims, *gpoint_flavor, *jeta, *jtemp, *jpress;
  const Bool* tropo;
  const Float *pf_g, *totplnk, *fmajor, *tlay, *tlev, *tsfc;
  Float *sfc_src, *lay_src, *lev_src, *sfc_jac;
  int* worklist;  // [0] = count, then (tile, band) pairs for planck_source_worklist_kernel
  const int* skip_if;  // plan guard raised: the direct kernel does the call
#ifdef EXP_CLOCKS
  unsigned long long* clocks;
#endif
};

template <int BS>
__global__ void __launch_bounds__(BS, 2) planck_source_v7_kernel(PlanckV7 a) {
  __shared__ int rng[2][6];  // per layer (ping-pong): Tmin, Tmax, Pmin, Pmax, emin, emax
  constexpr int PSLAB = 8704;  // 68 KB: no minor tables here and 2 blocks per CU, so the slab can be larger
  __shared__ __align__(16) Float slab[PSLAB];
  extern __shared__ Float tpl[];  // totplnk(:, ibnd)
  if (*a.skip_if) return;
  const int tid = threadIdx.x;
  const int ibnd = blockIdx.y;
  const unsigned ncol = a.ncol, nlay = a.nlay;
  const unsigned ncl = ncol * nlay, nclv = ncol * (nlay + 1);  // host guarantees < 2^31
  const int ntemp = a.ntemp, TE = a.TE, ngpt = a.ngpt, nPT = a.nPlanckTemp;
  const int gptS = a.band_lims[2 * ibnd] - 1, gptE = a.band_lims[2 * ibnd + 1] - 1;
  for (int i = tid; i < nPT; i += BS) tpl[i] = a.totplnk[(size_t)nPT * ibnd + i];
  if (tid < 12) rng[tid / 6][tid % 6] = (tid % 2 == 0) ? (1 << 30) : -1;
  __syncthreads();
  const unsigned icol = blockIdx.x * BS + tid;
  const bool valid = icol < ncol;
  const unsigned ic = min(icol, ncol - 1);
  auto planck = [&](Float t) {  // interpolate1D :715-737 on the LDS copy of the band's column
    const Float val0 = (t - a.temp_ref_min) * a.totplnk_delta_r;
    const Float frac = val0 - trunc(val0);
    const int index = min(nPT - 1, max(1, (int)val0 + 1));
    const Float t0 = tpl[index - 1], t1 = tpl[index];
    return t0 + frac * (t1 - t0);
  };
  const Float pl_sfc = planck(a.tsfc[ic]);
  const Float pl_sfc1 = planck(a.tsfc[ic] + (Float)1);

  struct Idx { int itropo, jT, jp; Float tlay, tlev; };
  struct Wts { Float2 fm[4]; int je1, je2; };
  auto load_idx = [&](unsigned l, Idx& x) {
    const unsigned cl = ic + ncol * l;
    x.itropo = a.tropo[cl] ? 0 : 1;
    x.jT = a.jtemp[cl];
    x.jp = a.jpress[cl] + x.itropo + 1;
    x.tlay = a.tlay[cl];
    x.tlev = a.tlev[cl];
  };
  auto load_wts = [&](unsigned l, const Idx& x, Wts& w) {
    const unsigned cl = ic + ncol * l;
    const int iflav = a.gpoint_flavor[x.itropo + 2 * gptS] - 1;
    const size_t clf = cl + (size_t)ncl * iflav;
    const Float2* fmp = reinterpret_cast<const Float2*>(a.fmajor + 8 * clf);
#pragma unroll
    for (int i = 0; i < 4; ++i) w.fm[i] = fmp[i];
    const int2 je = *reinterpret_cast<const int2*>(a.jeta + 2 * clf);
    w.je1 = je.x; w.je2 = je.y;
  };
  Idx x0, x1;   // layers l and l+1
  Wts w0;       // layer l
  load_idx(0, x0);
  load_idx(min(1u, nlay - 1), x1);
  load_wts(0, x0, w0);

  for (int g0 = gptS; g0 <= gptE; g0 += GC) {  // host guarantees whole, 16-aligned chunks (one pass per 16 g)
    if (g0 != gptS) {  // restart the layer walk for the next chunk of a wide band
      load_idx(0, x0); load_idx(min(1u, nlay - 1), x1); load_wts(0, x0, w0);
    }
    Float prev[GC];
#pragma unroll
    for (int j = 0; j < GC; ++j) prev[j] = 0;
    for (unsigned l = 0; l < nlay; ++l) {
      int* r = rng[l & 1];
      {
        const int big = 1 << 30;
        const int a0 = wave_min(valid ? x0.jT : big), a1 = wave_max(valid ? x0.jT + 1 : -1);
        const int a2 = wave_min(valid ? x0.jp - 1 : big), a3 = wave_max(valid ? x0.jp : -1);
        const int a4 = wave_min(valid ? min(w0.je1, w0.je2) : big), a5 = wave_max(valid ? max(w0.je1, w0.je2) + 1 : -1);
        if ((tid & 63) == 0) {
          atomicMin(&r[0], a0); atomicMax(&r[1], a1); atomicMin(&r[2], a2); atomicMax(&r[3], a3);
          atomicMin(&r[4], a4); atomicMax(&r[5], a5);
        }
      }
      __syncthreads();  // ranges complete; previous layer's compute finished (slab is free)
      const int Tmin = r[0], nT = r[1] - r[0] + 1, Pmin = r[2], nP = r[3] - r[2] + 1, emin = r[4], nE = r[5] - r[4] + 1;
      const int rows = nP * nT * nE;
      if (rows * RS > PSLAB) {  // block-uniform: this (tile, band) goes to the direct kernel as a whole
        if (tid == 0) {
          const int w = atomicAdd(&a.worklist[0], 1);
          a.worklist[1 + 2 * w] = blockIdx.x; a.worklist[2 + 2 * w] = ibnd;
        }
        return;
      }
      constexpr bool use_lds = true;
      if (tid < 6) rng[(l + 1) & 1][tid] = (tid % 2 == 0) ? (1 << 30) : -1;
      if (use_lds) {
        // 16-byte pieces of the bounding box, SB per thread requested back to back (index clamped, so the
        // count is fixed): the tile pays the L2 latency once per batch
        constexpr int SB = 4;
        const int nAll = rows * (GC / 2);
        const float inv_nE = 1.0f / (float)nE, inv_nT = 1.0f / (float)nT;
        auto piece = [&](int idx) -> Float2 {
          const int j = idx & 7, rr = idx >> 3;
          const int rest = (int)(((float)rr + 0.5f) * inv_nE), e = rr - rest * nE;  // rows < 2^12: exact
          const int p_l = (int)(((float)rest + 0.5f) * inv_nT), t_l = rest - p_l * nT;
          return *reinterpret_cast<const Float2*>(
              a.pf_g + ((size_t)((Pmin - 1 + p_l) * TE + (emin - 1 + e) * ntemp + (Tmin - 1 + t_l)) * ngpt + g0 + 2 * j));
        };
#pragma unroll 1
        for (int base = tid; base < nAll; base += SB * BS) {
          Float2 v[SB];
#pragma unroll
          for (int u = 0; u < SB; ++u) v[u] = piece(min(base + u * BS, nAll - 1));
#pragma unroll
          for (int u = 0; u < SB; ++u) {
            const int idx = base + u * BS;
            if (idx < nAll) *reinterpret_cast<Float2*>(slab + (idx >> 3) * RS + 2 * (idx & 7)) = v[u];
          }
        }
      }
      // this layer's values into locals, then request the following layers' inputs
      const Float f0 = w0.fm[0].x, f1 = w0.fm[0].y, f2 = w0.fm[1].x, f3 = w0.fm[1].y, f4 = w0.fm[2].x, f5 = w0.fm[2].y,
                  f6 = w0.fm[3].x, f7 = w0.fm[3].y;
      const int je1 = w0.je1, je2 = w0.je2, jT = x0.jT, jp = x0.jp;
      const Float tl = x0.tlay, tv = x0.tlev;
      x0 = x1;
      if (l + 1 < nlay) load_wts(l + 1, x0, w0);
      if (l + 2 < nlay) load_idx(l + 2, x1);
      __syncthreads();
      if (!valid) continue;
      const Float pl_lay = planck(tl), pl_lev = planck(tv);
      const unsigned cl = ic + ncol * l;
      Float* lay = a.lay_src + cl + (size_t)ncl * g0;
      Float* lev = a.lev_src + (ic + ncol * l) + (size_t)nclv * g0;
      const bool sfc = (int)l == a.sfc_lay - 1;
      // one body, instantiated separately for LDS and for global rows (a merged pointer would be a
      // generic one and every gather a slow flat load)
      auto body = [&](const Float* __restrict__ A0, const Float* __restrict__ B0, const int sE, const int sP) {
#pragma unroll
        for (int jj = 0; jj < GC; jj += 2) {
          // interpolate3D_byflav with scaling (1,1), :791-801; one 16-byte read feeds two g-points
          const Float2 k0 = ld2(A0 + jj), k1 = ld2(A0 + sE + jj), k2 = ld2(A0 + sP + jj), k3 = ld2(A0 + sP + sE + jj),
                       k4 = ld2(B0 + jj), k5 = ld2(B0 + sE + jj), k6 = ld2(B0 + sP + jj), k7 = ld2(B0 + sP + sE + jj);
          Float pfv[2], pgv[2];
          pfv[0] = f0 * k0.x; pfv[1] = f0 * k0.y;
          pfv[0] = fma(f1, k1.x, pfv[0]); pfv[1] = fma(f1, k1.y, pfv[1]);
          pfv[0] = fma(f2, k2.x, pfv[0]); pfv[1] = fma(f2, k2.y, pfv[1]);
          pfv[0] = fma(f3, k3.x, pfv[0]); pfv[1] = fma(f3, k3.y, pfv[1]);
          pgv[0] = f4 * k4.x; pgv[1] = f4 * k4.y;
          pgv[0] = fma(f5, k5.x, pgv[0]); pgv[1] = fma(f5, k5.y, pgv[1]);
          pgv[0] = fma(f6, k6.x, pgv[0]); pgv[1] = fma(f6, k6.y, pgv[1]);
          pgv[0] = fma(f7, k7.x, pgv[0]); pgv[1] = fma(f7, k7.y, pgv[1]);
#pragma unroll
          for (int u = 0; u < 2; ++u) {
            const int j = jj + u;
            const Float pf = pfv[u] + pgv[u];
            lay[(size_t)ncl * j] = pf * pl_lay;                                  // :674
            lev[(size_t)nclv * j] = (l == 0 ? pf : sqrt(prev[j] * pf)) * pl_lev;  // :695,:699
            if (sfc) {                                                           // :651-653
              a.sfc_src[ic + (size_t)ncol * (g0 + j)] = pf * pl_sfc;
              a.sfc_jac[ic + (size_t)ncol * (g0 + j)] = pf * (pl_sfc1 - pl_sfc);
            }
            prev[j] = pf;
          }
          asm volatile("" : "+v"(prev[jj]), "+v"(prev[jj + 1]));  // keep the pair's arithmetic here (see tau kernel)
          if ((jj & 2) != 0) __builtin_amdgcn_sched_barrier(0);  // at most 16 row reads (64 VGPRs) in flight
        }
      };
      body(slab + (((jp - 1 - Pmin) * nT + (jT - Tmin)) * nE + (je1 - emin)) * RS,
           slab + (((jp - 1 - Pmin) * nT + (jT + 1 - Tmin)) * nE + (je2 - emin)) * RS, RS, nT * nE * RS);
    }
    if (valid) {
      const Float pl_top = planck(a.tlev[ic + ncol * nlay]);
#pragma unroll
      for (int j = 0; j < GC; ++j) a.lev_src[ic + ncol * nlay + (size_t)nclv * (g0 + j)] = prev[j] * pl_top;  // :705
    }
    __syncthreads();
  }
}


// -------------------------------------------------------------------------------------------
// compute_Planck_source, specialised-wave kernel: the loader / compute split of tau_absorption_v9_kernel.
// Block = (NCW*64 columns, one band): NCW compute waves (lanes = columns) walk the LAYERS, so the previous
// layer's Planck fractions stay in registers for the geometric mean at the interface (:699); NLW loader
// waves stage the bounding box of pfrac rows of layer l+1 into the other half of a double-buffered LDS slab
// while layer l is computed; one barrier per layer.  planck_geom_kernel provides the boxes and sends
// (tile, band) pairs that do not fit the slab at some layer to the direct kernel.
// -------------------------------------------------------------------------------------------
template <int TILE, int G>
__global__ void __launch_bounds__(TILE) planck_geom_kernel(PlanckV7 a, int nbnd, TileGeom* __restrict__ geom,
                                                           int* __restrict__ flags, int slab_floats) {
  constexpr int RS = G + 2;
  __shared__ int rng[4];
  __shared__ int erng[MAXB][2];
  __shared__ int flav[MAXB][2];  // flavor (0-based) of band b per tropo regime
  if (*a.skip_if) return;
  const int tid = threadIdx.x;
  const unsigned ncol = a.ncol, nlay = a.nlay, ilay = blockIdx.y;
  const unsigned ncl = ncol * nlay;
  if (tid == 0) { rng[0] = 1 << 30; rng[1] = -1; rng[2] = 1 << 30; rng[3] = -1; }
  if (tid < MAXB) { erng[tid][0] = 1 << 30; erng[tid][1] = -1; }
  if (tid < 2 * nbnd) flav[tid >> 1][tid & 1] = a.gpoint_flavor[(tid & 1) + 2 * (a.band_lims[2 * (tid >> 1)] - 1)] - 1;
  __syncthreads();
  const unsigned icol = blockIdx.x * TILE + tid;
  const bool valid = icol < ncol;
  const unsigned ic = min(icol, ncol - 1);
  const unsigned cl = ic + ncol * ilay;
  const int itropo = a.tropo[cl] ? 0 : 1;
  const int jT = a.jtemp[cl];
  const int jp = a.jpress[cl] + itropo + 1;
  const int big = 1 << 30;
  {
    const int a0 = wave_min(valid ? jT : big), a1 = wave_max(valid ? jT + 1 : -1);
    const int a2 = wave_min(valid ? jp - 1 : big), a3 = wave_max(valid ? jp : -1);
    if ((tid & 63) == 0) { atomicMin(&rng[0], a0); atomicMax(&rng[1], a1); atomicMin(&rng[2], a2); atomicMax(&rng[3], a3); }
  }
  for (int b = 0; b < nbnd; ++b) {
    const int2 je = *reinterpret_cast<const int2*>(a.jeta + 2 * (cl + (size_t)ncl * flav[b][itropo]));
    const int e0 = wave_min(valid ? min(je.x, je.y) : big), e1 = wave_max(valid ? max(je.x, je.y) + 1 : -1);
    if ((tid & 63) == 0) { atomicMin(&erng[b][0], e0); atomicMax(&erng[b][1], e1); }
  }
  __syncthreads();
  TileGeom* out = geom + (blockIdx.x + (size_t)gridDim.x * ilay);
  const int nT = rng[1] - rng[0] + 1, nP = rng[3] - rng[2] + 1;
  if (tid == 0) {
    out->Tmin = rng[0]; out->nT = nT; out->Pmin = rng[2]; out->nP = nP; out->has_lo = 0; out->has_up = 0;
    out->pad0 = 0; out->pad1 = 0;
  }
  if (tid < nbnd) {
    const int emin = erng[tid][0], nE = erng[tid][1] - erng[tid][0] + 1;
    const bool fits = nP * nT * nE * RS <= slab_floats;
    if (!fits && atomicCAS(&flags[blockIdx.x * nbnd + tid], 0, 1) == 0) {  // once per (tile, band)
      const int w = atomicAdd(&a.worklist[0], 1);
      a.worklist[1 + 2 * w] = blockIdx.x; a.worklist[2 + 2 * w] = tid;
    }
    out->eg[tid] = make_int2(emin, nE);
  }
}

// Planck on a geometry left by compute_tau_absorption (rte_hip_share_geometry): which (tile, band) pairs do not fit
// the slab at some layer.  One wave per pair, lanes = layers (one thread walking the layers was 60 dependent latencies).
__global__ void __launch_bounds__(256)
planck_flags_kernel(const TileGeom* __restrict__ geom, int tiles, int nlay, int nbnd, int slab_floats, int RS,
                    int* __restrict__ flags, int* __restrict__ worklist, const int* __restrict__ valid,
                    const int* __restrict__ guard) {
  if (!*valid || *guard) return;
  const int i = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (i >= tiles * nbnd) return;
  const int tile = i / nbnd, b = i - tile * nbnd;
  bool fits = true;
  for (int l = lane; l < nlay; l += 64) {
    const TileGeom* g = geom + (tile + (size_t)tiles * l);
    fits = fits && g->nP * g->nT * abs(g->eg[b].y) * RS <= slab_floats;
  }
  if (__ballot(!fits) != 0ull && lane == 0) {
    flags[i] = 1;
    const int w = atomicAdd(&worklist[0], 1);
    worklist[1 + 2 * w] = tile; worklist[2 + 2 * w] = b;
  }
}

template <int NCW, int NLW, int SLAB, int G>
__global__ void __launch_bounds__((NCW + NLW) * 64, (NCW + NLW + 3) / 4)
planck_source_v9_kernel(PlanckV7 a, int nbnd, unsigned ntiles, const TileGeom* __restrict__ geom,
                        const int* __restrict__ flags) {
  constexpr int TILE = NCW * 64, NLT = NLW * 64, NT = TILE + NLT;
  constexpr int RS = G + 2, PPR = G / 2, PSH = G == 16 ? 3 : 2;  // row stride, 16-byte pieces per row, log2(PPR)
  constexpr int MAXL = 256;  // layers per block held in the LDS geometry table (host checks nlay <= MAXL)
  __shared__ __align__(16) Float slab[2][SLAB];
  __shared__ int gl[MAXL][6];       // per layer: Tmin, nT, Pmin, nP, emin, nE
  extern __shared__ Float tpl[];    // totplnk(:, ibnd)
  if (*a.skip_if) return;
  const int tid = threadIdx.x;
  // the bands of one column tile are neighbours in launch order (band = fast grid index): they run at about the
  // same time and share the tile's index arrays and, per flavor, its interpolation weights in the caches
  // XCD-aware: workgroups go to the 8 XCDs round-robin by linear id, so (id % 8) picks the XCD and the
  // sequence id / 8 on one XCD walks the bands of one tile before the next tile
  const unsigned lin = blockIdx.x, xcd = lin % 8, seq = lin / 8;
  const int ibnd = (int)(seq % (unsigned)nbnd);
  const unsigned tile = (seq / (unsigned)nbnd) * 8 + xcd;
  if (tile >= ntiles) return;  // block-uniform (grid padded to a multiple of 8 tiles)
  if (flags[tile * nbnd + ibnd]) return;  // block-uniform: the direct kernel does this (tile, band)
  const unsigned ncol = a.ncol, nlay = a.nlay;
  const unsigned ncl = ncol * nlay, nclv = ncol * (nlay + 1);  // host guarantees 8 * nclv < 2^32
  const int ntemp = a.ntemp, TE = a.TE, ngpt = a.ngpt, nPT = a.nPlanckTemp;
  const int gptS = a.band_lims[2 * ibnd] - 1, gptE = a.band_lims[2 * ibnd + 1] - 1;
  for (int i = tid; i < nPT; i += NT) tpl[i] = a.totplnk[(size_t)nPT * ibnd + i];
  for (int l = tid; l < (int)nlay; l += NT) {
    const TileGeom* g = geom + (tile + (size_t)ntiles * l);
    gl[l][0] = g->Tmin; gl[l][1] = g->nT; gl[l][2] = g->Pmin; gl[l][3] = g->nP;
    gl[l][4] = g->eg[ibnd].x; gl[l][5] = abs(g->eg[ibnd].y);  // (negative in a geometry shared with compute_tau_absorption)
  }
  __syncthreads();
  const int nchunk = (gptE - gptS + 1) / G;  // host guarantees whole, 16-aligned chunks
  // stages of a chunk: the layers in order, then -- unless the surface layer is the last one, whose Planck
  // fractions are still in registers -- the surface layer once more for sfc_source (keeps those stores and
  // their addresses out of the layer loop)
  const int lsfc = a.sfc_lay - 1;
  const int spc = (int)nlay + (lsfc == (int)nlay - 1 ? 0 : 1);
  const int nstage = nchunk * spc;

  if (tid >= TILE) {
    // ================================ loader waves ================================
    // the loaders issue little and mostly wait for memory: a raised issue priority lets their requests and LDS writes go
    // out ahead of the eight compute waves' FMAs, so that the next slab is complete a little earlier (tau 5.34 -> 5.28 ms
    // in one process, no change for Planck)
    __builtin_amdgcn_s_setprio(1);
    const int lt = tid - TILE;
    constexpr int SB = 8;  // 16-byte pieces per lane requested back to back
#pragma unroll 1
    for (int s = 0; s < nstage; ++s) {
      const int ls = s % spc, l = ls < (int)nlay ? ls : lsfc, g0 = gptS + (s / spc) * G;
      const int Tmin = gl[l][0], nT = gl[l][1], Pmin = gl[l][2], nP = gl[l][3], emin = gl[l][4], nE = gl[l][5];
      const float inv_nE = 1.0f / (float)nE, inv_nT = 1.0f / (float)nT;
      const int nAll = nP * nT * nE * (G / 2);
      Float* sl = slab[s & 1];
      auto piece = [&](int idx) -> Float2 {  // rows ordered [p][t][eta]
        const int j = idx & (PPR - 1), r = idx >> PSH;
        const int rest = (int)(((float)r + 0.5f) * inv_nE), e = r - rest * nE;  // rows < 2^12: exact
        const int p_l = (int)(((float)rest + 0.5f) * inv_nT), t_l = rest - p_l * nT;
        return *reinterpret_cast<const Float2*>(
            a.pf_g + ((size_t)((Pmin - 1 + p_l) * TE + (emin - 1 + e) * ntemp + (Tmin - 1 + t_l)) * ngpt + g0 + 2 * j));
      };
#pragma unroll 1
      for (int base = lt; base < nAll; base += SB * NLT) {
        Float2 v[SB];
#pragma unroll
        for (int u = 0; u < SB; ++u) v[u] = piece(min(base + u * NLT, nAll - 1));
#pragma unroll
        for (int u = 0; u < SB; ++u) {
          const int idx = base + u * NLT;
          if (idx < nAll) *reinterpret_cast<Float2*>(sl + (idx >> PSH) * RS + 2 * (idx & (PPR - 1))) = v[u];
        }
      }
      __syncthreads();  // B(s): slab(s) complete; the compute waves are done with the other buffer
    }
    return;
  }

  // ================================ compute waves (lanes = columns) ================================
  const unsigned icol = tile * TILE + tid;
  const bool valid = icol < ncol;
  const unsigned ic = min(icol, ncol - 1);
  const int flav0 = a.gpoint_flavor[2 * gptS] - 1, flav1 = a.gpoint_flavor[1 + 2 * gptS] - 1;
  auto planck = [&](Float t) {  // interpolate1D :715-737 on the LDS copy of the band's column
    const Float val0 = (t - a.temp_ref_min) * a.totplnk_delta_r;
    const Float frac = val0 - trunc(val0);
    const int index = min(nPT - 1, max(1, (int)val0 + 1));
    const Float t0 = tpl[index - 1], t1 = tpl[index];
    return t0 + frac * (t1 - t0);
  };
  const Float pl_sfc = planck(a.tsfc[ic]);
  const Float pl_sfc1 = planck(a.tsfc[ic] + (Float)1);

  struct Idx { Bool tropo; int jT, jpress; Float tlay, tlev; };  // raw loaded values: nothing is derived at load
  struct Wts { Float2 fm[4]; int je1, je2; };                     // time, so no request waits for another
  auto load_idx = [&](unsigned l, Idx& x) {
    const unsigned cl = ic + ncol * l;
    x.tropo = a.tropo[cl];
    x.jT = a.jtemp[cl];
    x.jpress = a.jpress[cl];
    x.tlay = a.tlay[cl];
    x.tlev = a.tlev[cl];
  };
  auto load_wts = [&](unsigned l, const Idx& x, Wts& w) {
    const size_t clf = (ic + ncol * l) + (size_t)ncl * (x.tropo ? flav0 : flav1);
    const Float2* fmp = reinterpret_cast<const Float2*>(a.fmajor + 8 * clf);
#pragma unroll
    for (int i = 0; i < 4; ++i) w.fm[i] = fmp[i];
    const int2 je = *reinterpret_cast<const int2*>(a.jeta + 2 * clf);
    w.je1 = je.x; w.je2 = je.y;
  };
  Idx x0, x1;   // layers l and l+1
  Wts w0;       // layer l
  Float prev[G];
  int s = 0;
#ifdef EXP_CLOCKS
  unsigned long long tk = clock64(), tacc[4] = {0, 0, 0, 0};
#undef TICK
#define TICK(i) do { const unsigned long long t2_ = clock64(); tacc[i] += t2_ - tk; tk = t2_; } while (0)
#else
#undef TICK
#define TICK(i)
#endif
#pragma unroll 1
  for (int g0 = gptS; g0 <= gptE; g0 += G) {
    load_idx(0, x0);
    load_idx(min(1u, nlay - 1), x1);
    load_wts(0, x0, w0);
#pragma unroll
    for (int j = 0; j < G; ++j) prev[j] = 0;
    // nothing outstanding at loop entry: the wait counts inside are then those of the steady state (requests of
    // the following layers, then this layer's 32 stores), not their merge with this prologue
    __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0)
#pragma unroll 1
    for (unsigned l = 0; l < nlay; ++l, ++s) {
      TICK(0);
      // this layer's values into locals, then request the following layers' inputs
      const Float f0 = w0.fm[0].x, f1 = w0.fm[0].y, f2 = w0.fm[1].x, f3 = w0.fm[1].y, f4 = w0.fm[2].x, f5 = w0.fm[2].y,
                  f6 = w0.fm[3].x, f7 = w0.fm[3].y;
      const int je1 = w0.je1, je2 = w0.je2, jT = x0.jT, jp = x0.jpress + (x0.tropo ? 0 : 1) + 1;  // levels jp-1, jp
      const Float tl = x0.tlay, tv = x0.tlev;
      x0 = x1;
      // unconditional (the last layers repeat the last one): a request made on some paths only makes the number of
      // outstanding memory operations path-dependent, and the compiler then drains them all -- this layer's requests
      // and the previous layer's 32 stores -- in front of every barrier
      load_wts(min(l + 1, nlay - 1), x0, w0);
      load_idx(min(l + 2, nlay - 1), x1);
      const int Tmin = gl[l][0], nT = gl[l][1], Pmin = gl[l][2], emin = gl[l][4], nE = gl[l][5];
      const Float pl_lay = planck(tl), pl_lev = planck(tv);
      TICK(1);
      __syncthreads();  // B(s): slab(s) is complete
      TICK(2);
      const Float* sl = slab[s & 1];
      const Float* A0 = sl + (((jp - 1 - Pmin) * nT + (jT - Tmin)) * nE + (je1 - emin)) * RS;
      const Float* B0 = sl + (((jp - 1 - Pmin) * nT + (jT + 1 - Tmin)) * nE + (je2 - emin)) * RS;
      const int sP = nT * nE * RS;
      // byte offsets of this column in the (col, lay, g) / (col, lev, g) planes; scalar plane bases
      unsigned olay = (ic + ncol * l) * (unsigned)sizeof(Float);
      asm volatile("" : "+v"(olay));  // keep 64-bit addresses out of the loop-invariant registers
      char* const play_ = reinterpret_cast<char*>(a.lay_src + (size_t)ncl * g0);
      char* const plev_ = reinterpret_cast<char*>(a.lev_src + (size_t)nclv * g0);
      const size_t slay = (size_t)ncl * sizeof(Float), slev = (size_t)nclv * sizeof(Float);
#pragma unroll
      for (int jj = 0; jj < G; jj += 2) {
        // interpolate3D_byflav with scaling (1,1), :791-801; one 16-byte read feeds two g-points
        const Float2 k0 = ld2(A0 + jj), k1 = ld2(A0 + RS + jj), k2 = ld2(A0 + sP + jj), k3 = ld2(A0 + sP + RS + jj),
                     k4 = ld2(B0 + jj), k5 = ld2(B0 + RS + jj), k6 = ld2(B0 + sP + jj), k7 = ld2(B0 + sP + RS + jj);
        Float pfv[2], pgv[2];
        pfv[0] = f0 * k0.x; pfv[1] = f0 * k0.y;
        pfv[0] = fma(f1, k1.x, pfv[0]); pfv[1] = fma(f1, k1.y, pfv[1]);
        pfv[0] = fma(f2, k2.x, pfv[0]); pfv[1] = fma(f2, k2.y, pfv[1]);
        pfv[0] = fma(f3, k3.x, pfv[0]); pfv[1] = fma(f3, k3.y, pfv[1]);
        pgv[0] = f4 * k4.x; pgv[1] = f4 * k4.y;
        pgv[0] = fma(f5, k5.x, pgv[0]); pgv[1] = fma(f5, k5.y, pgv[1]);
        pgv[0] = fma(f6, k6.x, pgv[0]); pgv[1] = fma(f6, k6.y, pgv[1]);
        pgv[0] = fma(f7, k7.x, pgv[0]); pgv[1] = fma(f7, k7.y, pgv[1]);
#pragma unroll
        for (int u = 0; u < 2; ++u) {
          const int j = jj + u;
          const Float pf = pfv[u] + pgv[u];
          const Float vlay = pf * pl_lay;                                      // :674
          const Float vlev = (l == 0 ? pf : sqrt(prev[j] * pf)) * pl_lev;      // :695,:699
          // lanes past the last column repeat it (ic is clamped) and store the same values to the same
          // addresses: unconditional stores keep the number of outstanding memory operations static, so the
          // wait for the next layer's weights is a counted one instead of a drain of these stores
          store_stream(reinterpret_cast<Float*>(play_ + slay * j + olay), vlay);
          store_stream(reinterpret_cast<Float*>(plev_ + slev * j + olay), vlev);  // level l of (ncol, nlay+1): same column offset
          prev[j] = pf;
        }
        asm volatile("" : "+v"(prev[jj]), "+v"(prev[jj + 1]));  // keep the pair's arithmetic here
        __builtin_amdgcn_sched_barrier(0);   // at most 8 row reads (32 VGPRs) in flight
      }
      TICK(3);
    }
    if (valid) {
      const Float pl_top = planck(a.tlev[ic + ncol * nlay]);
#pragma unroll
      for (int j = 0; j < G; ++j) a.lev_src[ic + ncol * nlay + (size_t)nclv * (g0 + j)] = prev[j] * pl_top;  // :705
    }
    // ---- surface source (:651-653) from the Planck fractions of the surface layer
    if (lsfc != (int)nlay - 1) {
      load_idx(lsfc, x0);
      load_wts(lsfc, x0, w0);
      const int Tmin = gl[lsfc][0], nT = gl[lsfc][1], Pmin = gl[lsfc][2], emin = gl[lsfc][4], nE = gl[lsfc][5];
      __syncthreads();  // B(s): the surface layer's slab is complete
      const Float* sl = slab[s & 1];
      ++s;
      const int jps = x0.jpress + (x0.tropo ? 0 : 1) + 1;
      const Float* A0 = sl + (((jps - 1 - Pmin) * nT + (x0.jT - Tmin)) * nE + (w0.je1 - emin)) * RS;
      const Float* B0 = sl + (((jps - 1 - Pmin) * nT + (x0.jT + 1 - Tmin)) * nE + (w0.je2 - emin)) * RS;
      const int sP = nT * nE * RS;
#pragma unroll
      for (int jj = 0; jj < G; jj += 2) {
        const Float2 k0 = ld2(A0 + jj), k1 = ld2(A0 + RS + jj), k2 = ld2(A0 + sP + jj), k3 = ld2(A0 + sP + RS + jj),
                     k4 = ld2(B0 + jj), k5 = ld2(B0 + RS + jj), k6 = ld2(B0 + sP + jj), k7 = ld2(B0 + sP + RS + jj);
        Float pa = w0.fm[0].x * k0.x, pb = w0.fm[0].x * k0.y, qa = w0.fm[2].x * k4.x, qb = w0.fm[2].x * k4.y;
        pa = fma(w0.fm[0].y, k1.x, pa); pb = fma(w0.fm[0].y, k1.y, pb); qa = fma(w0.fm[2].y, k5.x, qa); qb = fma(w0.fm[2].y, k5.y, qb);
        pa = fma(w0.fm[1].x, k2.x, pa); pb = fma(w0.fm[1].x, k2.y, pb); qa = fma(w0.fm[3].x, k6.x, qa); qb = fma(w0.fm[3].x, k6.y, qb);
        pa = fma(w0.fm[1].y, k3.x, pa); pb = fma(w0.fm[1].y, k3.y, pb); qa = fma(w0.fm[3].y, k7.x, qa); qb = fma(w0.fm[3].y, k7.y, qb);
        prev[jj] = pa + qa; prev[jj + 1] = pb + qb;
        asm volatile("" : "+v"(prev[jj]), "+v"(prev[jj + 1]));
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    if (valid) {
#pragma unroll
      for (int j = 0; j < G; ++j) {
        a.sfc_src[ic + (size_t)ncol * (g0 + j)] = prev[j] * pl_sfc;
        a.sfc_jac[ic + (size_t)ncol * (g0 + j)] = prev[j] * (pl_sfc1 - pl_sfc);
      }
    }
  }
#ifdef EXP_CLOCKS
  if (tid == 0)
    for (int i = 0; i < 4; ++i) atomicAdd(&a.clocks[i], tacc[i]);
#endif
}


// -------------------------------------------------------------------------------------------
// compute_tau_rayleigh, production kernel.  The Rayleigh table has no pressure dimension: the whole
// (T, eta) plane of a band's 16 g-points for both tropo regimes is 2 x ntemp*neta rows of 128 bytes
// (32 KB), so a block = (256 columns, 16 g-points) stages it ONCE, walks the layers and gathers its
// four corner rows from LDS with 16-byte reads (reference :506-565).  Inputs of layer l+1 are requested
// while layer l is computed; no barrier in the layer loop.
// -------------------------------------------------------------------------------------------
struct RaylArgs {
  const int* skip_if;  // plan guard raised: the direct kernel does the call
  RaylCombine cb;      // cb.tau_abs != nullptr: fused with combine_abs_and_rayleigh (2-stream)
  int nbnd;
  const int* band_lims;
  int ncol, nlay, ngpt, neta, ntemp, idx_h2o;
  const int *gpoint_flavor, *jeta, *jtemp;
  const Float *krayl, *col_dry, *col_gas, *fminor;
  const Bool* tropo;
  Float* tau_rayleigh;
};

template <int BS, int G, bool COMBINE>
__global__ void __launch_bounds__(BS) tau_rayleigh_slab_kernel(RaylArgs a) {
  constexpr int RS = G + 2;
  extern __shared__ __align__(16) Float rslab[];  // [2 tropo][neta][ntemp] rows of RS Floats
  if (*a.skip_if) return;
  const int tid = threadIdx.x;
  // the g-point chunk is the fast grid index: the chunks of one column tile run together and share its inputs in cache
  // (pinning a tile's chunks to one XCD, as planck_source_v9_kernel does, measured slower here: 2.8 vs 2.45 ms)
  const int g0 = blockIdx.x * G;  // host guarantees whole, G-aligned chunks per band
  const unsigned ncol = a.ncol, nlay = a.nlay;
  const unsigned ncl = ncol * nlay;  // host guarantees 8 * ncl < 2^32
  const int ntemp = a.ntemp, tn = a.ntemp * a.neta;
  // stage: native layout (ntemp, neta, ngpt, 2) is contiguous along (T, eta) for a fixed g-point -> coalesced reads
  for (int idx = tid; idx < 2 * G * tn; idx += BS) {
    const int te = idx % tn, gj = (idx / tn) % G, r = idx / (tn * G);
    rslab[(r * tn + te) * RS + gj] = a.krayl[(size_t)te + (size_t)tn * ((g0 + gj) + (size_t)a.ngpt * r)];
  }
  __syncthreads();
  const unsigned icol = blockIdx.y * BS + tid;
  const unsigned ic = min(icol, ncol - 1);  // lanes past the last column repeat it (same values, same addresses)
  const int flav0 = a.gpoint_flavor[2 * g0] - 1, flav1 = a.gpoint_flavor[1 + 2 * g0] - 1;
  const bool cld = COMBINE && a.cb.cld_tau != nullptr;
  int ibnd_blk = 0;  // band of this block's g-point chunk (by-band cloud operand)
  if (cld)
    for (int b = 0; b < a.nbnd; ++b)
      if (g0 + 1 >= a.band_lims[2 * b] && g0 + 1 <= a.band_lims[2 * b + 1]) ibnd_blk = b;
  struct In { Bool tropo; int jT; Float h2o, dry; };
  struct Wt { Float2 f01, f23; int2 je; };
  auto load_in = [&](unsigned l, In& x) {
    const unsigned cl = ic + ncol * l;
    x.tropo = a.tropo[cl]; x.jT = a.jtemp[cl];
    x.h2o = a.col_gas[cl + (size_t)ncl * a.idx_h2o]; x.dry = a.col_dry[cl];
  };
  auto load_wt = [&](unsigned l, const In& x, Wt& w) {
    const size_t clf = (ic + ncol * l) + (size_t)ncl * (x.tropo ? flav0 : flav1);
    const Float2* fp = reinterpret_cast<const Float2*>(a.fminor + 4 * clf);
    w.f01 = fp[0]; w.f23 = fp[1];
    w.je = *reinterpret_cast<const int2*>(a.jeta + 2 * clf);
  };
  In x0, x1;
  Wt w0;
  load_in(0, x0);
  load_in(min(1u, nlay - 1), x1);
  load_wt(0, x0, w0);
  char* const plane0 = reinterpret_cast<char*>(a.tau_rayleigh + (size_t)ncl * g0);
  const size_t gstride = (size_t)ncl * sizeof(Float);
#pragma unroll 1
  for (unsigned l = 0; l < nlay; ++l) {
    const Float f0 = w0.f01.x, f1 = w0.f01.y, f2 = w0.f23.x, f3 = w0.f23.y;
    const int je1 = w0.je.x, je2 = w0.je.y, jT = x0.jT, r = x0.tropo ? 0 : 1;
    const Float w = x0.h2o + x0.dry;  // :553
    x0 = x1;
    load_wt(min(l + 1, nlay - 1), x0, w0);
    load_in(min(l + 2, nlay - 1), x1);
    const Float* k1 = rslab + (r * tn + (jT - 1) + ntemp * (je1 - 1)) * RS;
    const Float* k2 = rslab + (r * tn + jT + ntemp * (je2 - 1)) * RS;
    unsigned off = (ic + ncol * l) * (unsigned)sizeof(Float);
    asm volatile("" : "+v"(off));  // keep 64-bit store addresses out of the loop-invariant registers
    Float ta[COMBINE ? G : 1];
    Float ct = 0, cs = 0, cg = 0;
    if (COMBINE && cld) {  // the band's cloud properties of this (column, layer)
      const size_t ob = (size_t)ic + (size_t)ncol * l + (size_t)ncl * ibnd_blk;
      ct = a.cb.cld_tau[ob]; cs = a.cb.cld_ssa[ob]; cg = a.cb.cld_g[ob];
    }
    if (COMBINE) {  // this layer's absorption optical depths, requested before the table arithmetic
#pragma unroll
      for (int j = 0; j < G; ++j)
        ta[j] = *reinterpret_cast<const Float*>(reinterpret_cast<const char*>(a.cb.tau_abs) + (size_t)ncl * (g0 + j) * sizeof(Float) + off);
    }
#pragma unroll
    for (int j = 0; j < G; j += 2) {
      // interpolate2D :757-760 with the reference's association, then :555
      const Float2 a0 = ld2(k1 + j), a1 = ld2(k1 + ntemp * RS + j), b0 = ld2(k2 + j), b1 = ld2(k2 + ntemp * RS + j);
      const Float ka = f0 * a0.x + f1 * a1.x + f2 * b0.x + f3 * b1.x;
      const Float kb = f0 * a0.y + f1 * a1.y + f2 * b0.y + f3 * b1.y;
      if (!COMBINE) {
        *reinterpret_cast<Float*>(plane0 + gstride * j + off) = ka * w;
        *reinterpret_cast<Float*>(plane0 + gstride * (j + 1) + off) = kb * w;
      } else {
        // combine_abs_and_rayleigh (2-stream branch, mo_gas_optics_rrtmgp.F90:1983-2002) on the value just formed:
        // tau = tau_abs + tau_rayleigh, ssa = tau_rayleigh / tau, g = 0 -- tau_rayleigh never goes to memory -- and,
        // with clouds given by band, their increment_2stream_by_2stream_bybnd
#pragma unroll
        for (int u = 0; u < 2; ++u) {
          const size_t po = (size_t)ncl * (g0 + j + u) * sizeof(Float) + off;
          Float t, s_, g_;
          rayl_finish(ta[j + u], (u == 0 ? ka : kb) * w, cld, ct, cs, cg, t, s_, g_);
          if (icol < ncol) {  // tau may alias tau_abs: the clamped lanes past the last column must not update it again
            *reinterpret_cast<Float*>(reinterpret_cast<char*>(a.cb.ssa) + po) = s_;
            *reinterpret_cast<Float*>(reinterpret_cast<char*>(a.cb.tau) + po) = t;
            *reinterpret_cast<Float*>(reinterpret_cast<char*>(a.cb.g) + po) = g_;
          }
        }
      }
    }
  }
}

}  // namespace

// ===============================================================================================
// C ABI
// ===============================================================================================
// process-wide tuning switches (set from any thread: relaxed atomics)
static std::atomic<int> g_tau_force_direct{0};
static std::atomic<int> g_tau_variant{9};
static const bool g_worklist_native = getenv("RTE_WORKLIST_NATIVE") != nullptr;  // A/B: worklist entries from the native-layout tables
static std::atomic<int> g_planck_variant{9};  // 9: specialised-wave kernel, 7: single-role slab kernel (rte_hip_planck_variant)  // 9: specialised-wave kernel, 7: single-role slab kernel (rte_hip_tau_variant)
static std::atomic<int> g_geom_variant{2};  // 2: bit-mask pre-pass (tile_geom2_kernel), 1: the band-walking pre-passes (rte_hip_geom_variant)
// Every piece of mutable host-side state of this file lives in the calling thread's current CONTEXT (runtime.hip):
// plan caches, the geometry shared between consecutive calls, the guards' flag words.  Tuning switches (rte_hip_*_variant)
// are process-wide.
namespace {
struct TauPlanCache {
  const void* key[14] = {};
  int dims[7] = {};
  int epoch = -1;
  bool fast_ok = false;
  int gw = 0;  // g-points per stage of the production kernels (16 or 8)
  bool uploads_pending = false;  // bands changed since the last upload to the device
  unsigned guard = 0;            // checksum of the index tables the plan was built from (tables_guard_kernel)
  std::vector<BandMeta> bands;
  bool matches(const void* const* k, const int* d, int e) const {
    if (e != epoch) return false;
    for (int i = 0; i < 14; ++i)
      if (k[i] != key[i]) return false;
    for (int i = 0; i < 7; ++i)
      if (d[i] != dims[i]) return false;
    return true;
  }
  void set(const void* const* k, const int* d, int e) {
    for (int i = 0; i < 14; ++i) key[i] = k[i];
    for (int i = 0; i < 7; ++i) dims[i] = d[i];
    epoch = e;
  }
};
}  // namespace

// Geometry shared between compute_tau_absorption and the compute_Planck_source call that directly follows it
// (opt-in, rte_hip_share_geometry): both derive the same per-(tile, layer) bounding boxes from the same interpolation
// indices, each by reading all of jeta (0.13 ms).  Like the deferred zero fill, for callers that touch the
// interpolation arrays only through this library between the two calls; keyed by the arrays' addresses, the
// dimensions and the library's call sequence (the Planck call must be the very next one).
struct SharedGeom {
  const void *jeta = nullptr, *jtemp = nullptr, *jpress = nullptr, *tropo = nullptr;
  int ncol = 0, nlay = 0, nflav = 0, nbnd = 0, gw = 0;
  long seq = -1;            // call sequence number of the compute_tau_absorption call that wrote it
  TileGeom* geom = nullptr;  // persistent: lives across calls
  int* valid = nullptr;      // device word: 1 once that call's geometry kernel ran (it does not when the call is rerouted)
  size_t cap = 0;
};
// The same option also lets rrtmgp_interpolation leave, per (256-column block, layer), the bit masks of the LUT rows
// its columns touch (it has every index in registers), and the compute_tau_absorption call that is the very next
// library call on the same interpolation arrays builds its tile geometry from these few megabytes instead of reading
// jtemp, jpress, tropo and all of jeta again (0.13 ms).  Masks are keyed by the tropo flag; the geometry kernel's own
// are keyed by the layer ranges derived from it, which is the same thing unless a column's pressure is not monotone in
// the layer index -- tropo_limits_kernel raises `irregular` then and the geometry kernel derives its masks itself.
struct InterpMasks {
  const void *jeta = nullptr, *jtemp = nullptr, *jpress = nullptr, *tropo = nullptr;
  int ncol = 0, nlay = 0, nflav = 0;
  long seq = -1;             // call sequence number of the interpolation call that wrote them
  unsigned* buf = nullptr;   // persistent
  size_t cap = 0;
};
// "are this table's bands whole aligned chunks of 16 or 8 g-points" -- checked once per table pointer and contents
struct BandCheck {
  const void* key = nullptr;
  int n = -1, epoch = -1;
  bool ok = false;
  int gw = 0;
  unsigned fp_seen = 0;
};
constexpr int NPLAN = 4;  // a few plans are kept (e.g. an LW and an SW k-distribution used alternately), least recently built evicted
struct GasState {
  int plan_epoch = 0;  // bumped by rte_hip_invalidate_plans(): forget cached host-side plans
  // Raised ON THE DEVICE by the plan guards when a cached plan no longer matches the caller's tables: one int in pinned,
  // device-mapped host memory that the guard kernels write directly.  The host looks at it at every plan look-up and then
  // drops the cached plans, so that they are rebuilt instead of the direct kernels doing every later call.
  volatile int* stale_host = nullptr;
  int* stale_dev = nullptr;
  int* stats_dev = nullptr;  // diagnostics: entries handed to the direct-gather worklists by the last tau / Planck call (rte_hip_stat)
  int share_geom = 0;        // 0 off, 1 on; 2 = tau -> Planck only, 3 = interpolation -> tau only (A/B)
  SharedGeom shared;
  InterpMasks imask;
  TauPlanCache plans[NPLAN];
  int plan_next = 0;
  BandCheck rayl_bands, planck_bands;
};
static std::atomic<int> g_share_geom_default{0};  // what a context starts with (the last rte_hip_share_geometry of any context)
static void* make_gas_state() {
  auto* g = new GasState();
  g->share_geom = g_share_geom_default;
  return g;
}
static void free_gas_state(void* p) {
  auto* g = (GasState*)p;
  if (g->shared.geom) (void)hipFree(g->shared.geom);
  if (g->shared.valid) (void)hipFree(g->shared.valid);
  if (g->imask.buf) (void)hipFree(g->imask.buf);
  if (g->stats_dev) (void)hipFree(g->stats_dev);
  if (g->stale_host) (void)hipHostFree((void*)g->stale_host);
  delete g;
}
static GasState& gs() { return *(GasState*)rte::gas_state(make_gas_state, free_gas_state); }
static int* stats_dev() {
  GasState& g = gs();
  if (!g.stats_dev) {
    HIP_CHECK(hipMalloc((void**)&g.stats_dev, 4 * sizeof(int)));
    HIP_CHECK(hipMemset(g.stats_dev, 0, 4 * sizeof(int)));
  }
  return g.stats_dev;
}
static int* stale_flag() {
  GasState& g = gs();
  if (!g.stale_dev) {
    HIP_CHECK(hipHostMalloc((void**)&g.stale_host, sizeof(int), hipHostMallocMapped));
    *g.stale_host = 0;
    HIP_CHECK(hipHostGetDevicePointer((void**)&g.stale_dev, (void*)g.stale_host, 0));
  }
  return g.stale_dev;
}
static void stale_poll() {
  (void)stale_flag();
  GasState& g = gs();
  if (*g.stale_host) {  // a guard fired in an earlier call: forget every plan
    ++g.plan_epoch;
    *g.stale_host = 0;
  }
}
static bool share_boxes() { const int v = gs().share_geom; return v == 1 || v == 2; }
static bool share_masks() { const int v = gs().share_geom; return v == 1 || v == 3; }

extern "C" {

int rte_hip_share_geometry(int on) { g_share_geom_default = on; gs().share_geom = on; gs().shared.seq = -1; gs().imask.seq = -1; return 0; }
int rte_hip_force_direct_gather(int on) { g_tau_force_direct = on; return 0; }
int rte_hip_tau_variant(int v) { g_tau_variant = v; return 0; }
int rte_hip_planck_variant(int v) { g_planck_variant = v; return 0; }
int rte_hip_invalidate_plans(void) { ++gs().plan_epoch; return 0; }
int rte_hip_geom_variant(int v) { g_geom_variant = v; return 0; }
// diagnostics (synchronises): 0 = (column tile, layer, band) triples the last compute_tau_absorption call handed to the
// direct-gather worklist, 1 = (column tile, band) pairs of the last compute_Planck_source call
int rte_hip_stat(int which) {
  if (which < 0 || which > 3) return -1;
  int v = 0;
  HIP_CHECK(hipStreamSynchronize(rte::stream()));
  HIP_CHECK(hipMemcpy(&v, stats_dev() + which, sizeof(int), hipMemcpyDeviceToHost));
  return v;
}


void rrtmgp_interpolation(const int* ncol_, const int* nlay_, const int* ngas_, const int* nflav_,
                          const int* neta_, const int* npres_, const int* ntemp_, const int* flavor,
                          const Float* press_ref_log, const Float* temp_ref,
                          const Float* press_ref_log_delta, const Float* temp_ref_min,
                          const Float* temp_ref_delta, const Float* press_ref_trop_log,
                          const Float* vmr_ref, const Float* play, const Float* tlay,
                          const Float* col_gas, int* jtemp, Float* fmajor, Float* fminor,
                          Float* col_mix, Bool* tropo, int* jeta, int* jpress) {
  const int ncol = *ncol_, nlay = *nlay_, ngas = *ngas_, nflav = *nflav_, neta = *neta_,
            npres = *npres_, ntemp = *ntemp_;
  if (ncol <= 0 || nlay <= 0 || nflav <= 0) return;
  RTE_TRY
  rte::Call c("rrtmgp_interpolation");
  const size_t ncl = (size_t)ncol * nlay;
  // scalar preparation exactly as reference :99-102
  const Float press_ref_trop = exp(*press_ref_trop_log);
  const Float temp_ref_delta_inv = (Float)1 / *temp_ref_delta;
  const Float press_ref_log_delta_inv = (Float)1 / *press_ref_log_delta;
  const int* d_flavor = c.in(flavor, (size_t)2 * nflav);
  const Float* d_temp_ref = c.in(temp_ref, (size_t)ntemp);
  const Float* d_press_ref_log = c.in(press_ref_log, (size_t)npres);
  const Float* d_vmr_ref = c.in(vmr_ref, (size_t)2 * (ngas + 1) * ntemp);
  const Float* d_play = c.in(play, ncl);
  const Float* d_tlay = c.in(tlay, ncl);
  const Float* d_col_gas = c.in(col_gas, ncl * (ngas + 1));
  int* d_jtemp = c.out_lazy(jtemp, ncl);  // (lazy: host-mirror mode keeps the interpolation state on the device)
  Float* d_fmajor = c.out_lazy(fmajor, 8 * ncl * nflav);
  Float* d_fminor = c.out_lazy(fminor, 4 * ncl * nflav);
  Float* d_col_mix = c.out_lazy(col_mix, 2 * ncl * nflav);
  Bool* d_tropo = c.out_lazy(tropo, ncl);
  int* d_jeta = c.out_lazy(jeta, 2 * ncl * nflav);
  int* d_jpress = c.out_lazy(jpress, ncl);
  dim3 grid(cdiv(ncol, 256), nlay), block(256);
  // masks for the compute_tau_absorption call that follows (InterpMasks): row numbers must fit the mask words
  unsigned* d_masks = nullptr;
  gs().imask.seq = -1;
  if (share_masks() && !c.any_host() && rte::is_device_memory(jeta) && nflav <= MAXFLAV && neta < 31 && ntemp < 31 && npres + 1 < 63) {
    const size_t need = sizeof(unsigned) * (size_t)grid.x * nlay * (4 + 2 * nflav);
    if (gs().imask.cap < need) {
      HIP_CHECK(hipStreamSynchronize(rte::stream()));
      if (gs().imask.buf) HIP_CHECK(hipFree(gs().imask.buf));
      HIP_CHECK(hipMalloc((void**)&gs().imask.buf, need));
      gs().imask.cap = need;
    }
    d_masks = gs().imask.buf;
    gs().imask.jeta = jeta; gs().imask.jtemp = jtemp; gs().imask.jpress = jpress; gs().imask.tropo = tropo;
    gs().imask.ncol = ncol; gs().imask.nlay = nlay; gs().imask.nflav = nflav;
    gs().imask.seq = rte::call_seq();
  }
  rte::ProfScope p("interpolation_kernel");
  // the gas amounts of a (column, layer) in LDS while they fit beside the 38 KB of transpose buffers in the 64 KB a block
  // gets without asking for more (ngas <= 11); larger tables read them through L2
  const int cg_lds = (ngas + 1) <= 12 ? 1 : 0;
  hipLaunchKernelGGL(interpolation_kernel, grid, block, cg_lds ? sizeof(Float) * 256 * (ngas + 1) : 0, rte::stream(), ncol, nlay, ngas, nflav, neta,
                     npres, ntemp, d_flavor, d_temp_ref, d_press_ref_log, press_ref_log_delta_inv,
                     *temp_ref_min, *temp_ref_delta, temp_ref_delta_inv, press_ref_trop, d_vmr_ref, d_play,
                     d_tlay, d_col_gas, d_jtemp, d_fmajor, d_fminor, d_col_mix, d_tropo, d_jeta, d_jpress, d_masks, cg_lds);
  RTE_CATCH("rrtmgp_interpolation")
}

}  // extern "C"
// compute_tau_absorption; add_bybnd != nullptr: the band-wise increment of the result by a second optical depth given
// per band (clouds as absorbers) is applied in the same pass
// fused SW gas optics (rte_hip_gas_optics_sw_2str): what compute_tau_rayleigh and the combine need besides the
// arguments of compute_tau_absorption; tau is then an output only
struct RaylHost {
  const Float *krayl, *col_dry, *cld_tau, *cld_ssa, *cld_g;
  Float *ssa, *g;
};
static void tau_absorption_impl(
    const char* api_name, int ncol, int nlay, int nbnd, int ngpt, int ngas, int nflav, int neta, int npres, int ntemp,
    int nlo, int nkl_, int nup, int nku_, int idx_h2o, const int* gpoint_flavor,
    const int* band_lims_gpt, const Float* kmajor, const Float* kminor_lower,
    const Float* kminor_upper, const int* minor_limits_gpt_lower, const int* minor_limits_gpt_upper,
    const Bool* minor_scales_with_density_lower, const Bool* minor_scales_with_density_upper,
    const Bool* scale_by_complement_lower, const Bool* scale_by_complement_upper,
    const int* idx_minor_lower, const int* idx_minor_upper, const int* idx_minor_scaling_lower,
    const int* idx_minor_scaling_upper, const int* kminor_start_lower, const int* kminor_start_upper,
    const Bool* tropo, const Float* col_mix, const Float* fmajor, const Float* fminor,
    const Float* play, const Float* tlay, const Float* col_gas, const int* jeta, const int* jtemp,
    const int* jpress, Float* tau, const Float* add_bybnd, const RaylHost* rh = nullptr) {
  const int* nminorklower_ = &nkl_;
  const int* nminorkupper_ = &nku_;
  const int* idx_h2o_ = &idx_h2o;
  if (ncol <= 0 || nlay <= 0 || ngpt <= 0) return;
  // a deferred zero_array on exactly this buffer turns the accumulate into an overwrite
  bool overwrite = rh ? true : rte::take_pending_zero(tau, sizeof(Float) * (size_t)ncol * nlay * ngpt);
  RTE_TRY
  rte::Call c(api_name);
  const size_t ncl = (size_t)ncol * nlay;
  const size_t tn = (size_t)ntemp * neta;
  const Float* d_add = add_bybnd ? c.in(add_bybnd, ncl * nbnd) : nullptr;
  const int* d_gpoint_flavor = c.in(gpoint_flavor, (size_t)2 * ngpt);
  const int* d_band_lims = c.in(band_lims_gpt, (size_t)2 * nbnd);
  const Float* d_kmajor = c.in(kmajor, tn * (npres + 1) * ngpt);
  MinorTables lo{c.in(kminor_lower, tn * *nminorklower_), c.in(minor_limits_gpt_lower, (size_t)2 * nlo),
                 c.in(minor_scales_with_density_lower, (size_t)nlo), c.in(scale_by_complement_lower, (size_t)nlo),
                 c.in(idx_minor_lower, (size_t)nlo), c.in(idx_minor_scaling_lower, (size_t)nlo),
                 c.in(kminor_start_lower, (size_t)nlo), nullptr, nullptr, nlo};
  MinorTables up{c.in(kminor_upper, tn * *nminorkupper_), c.in(minor_limits_gpt_upper, (size_t)2 * nup),
                 c.in(minor_scales_with_density_upper, (size_t)nup), c.in(scale_by_complement_upper, (size_t)nup),
                 c.in(idx_minor_upper, (size_t)nup), c.in(idx_minor_scaling_upper, (size_t)nup),
                 c.in(kminor_start_upper, (size_t)nup), nullptr, nullptr, nup};
  const Bool* d_tropo = c.in(tropo, ncl);
  const Float* d_col_mix = c.in(col_mix, 2 * ncl * nflav);
  const Float* d_fmajor = c.in(fmajor, 8 * ncl * nflav);
  const Float* d_fminor = c.in(fminor, 4 * ncl * nflav);
  const Float* d_play = c.in(play, ncl);
  const Float* d_tlay = c.in(tlay, ncl);
  const Float* d_col_gas = c.in(col_gas, ncl * (ngas + 1));
  const int* d_jeta = c.in(jeta, 2 * ncl * nflav);
  const int* d_jtemp = c.in(jtemp, ncl);
  const int* d_jpress = c.in(jpress, ncl);
  // (host-mirror mode: tau stays on the device; a zero_array recorded on its device copy makes this an overwrite too)
  bool zero_recorded = false;
  Float* d_tau = rh ? c.out_lazy(tau, ncl * ngpt) : c.inout_lazy(tau, ncl * ngpt, &zero_recorded);
  overwrite = overwrite || zero_recorded;
  RaylCombine cb{nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
  const Float *d_krayl = nullptr, *d_col_dry = nullptr;
  if (rh) {
    d_krayl = c.in(rh->krayl, (size_t)ntemp * neta * ngpt * 2);
    d_col_dry = c.in(rh->col_dry, ncl);
    if (rh->cld_tau) { cb.cld_tau = c.in(rh->cld_tau, ncl * nbnd); cb.cld_ssa = c.in(rh->cld_ssa, ncl * nbnd); cb.cld_g = c.in(rh->cld_g, ncl * nbnd); }
    cb.tau_abs = d_tau; cb.tau = d_tau;  // the direct kernels combine in place
    cb.ssa = c.out_lazy(rh->ssa, ncl * ngpt);
    cb.g = c.out_lazy(rh->g, ncl * ngpt);
  }
  hipStream_t st = rte::stream();
  if (!rh && !c.any_host() && rte::is_device_memory(d_tau)) rte::fork_point(d_tau, sizeof(Float) * ncl * ngpt);
  // layer limits of the two regimes per column (:274-285) + "regimes overlap somewhere" flag
  int* lim = (int*)rte::scratch(sizeof(int) * (4 * (size_t)ncol + 2));
  int* overlap = lim + 4 * (size_t)ncol;
  int* irregular = overlap + 1;  // some column's layer ranges are not those of its tropo flags (see tropo_limits_kernel)
  const size_t wl_cap = (size_t)cdiv(ncol, 256) * nlay * nbnd;  // tiles are at least 256 columns wide
  int* const worklist = (int*)rte::scratch(sizeof(int) * (1 + 3 * wl_cap));
  int* const valid_word = share_boxes() ? gs().shared.valid : nullptr;  // (null until the first sharing call has allocated it)
  {
    rte::ProfScope p("tau_absorption_setup");
    // (the validity word of a geometry shared with compute_Planck_source is cleared here too: it is set again only if this
    //  call's geometry kernel runs)
    hipLaunchKernelGGL(zero_words_kernel, dim3(1), dim3(256), 0, st, overlap, 2u, worklist, 1u, valid_word, valid_word ? 1u : 0u);
  }  // (layer limits: tropo_limits_kernel below, or a role of tau_setup_kernel on the production path)
  int* d_stale = stale_flag();
  stale_poll();
  // ---- host-side plan from the small index tables (cached while the caller's table pointers and
  // dimensions do not change; rte_hip_release() drops the cache)
  // a few plans are kept (e.g. an LW and an SW k-distribution used alternately), least recently built evicted
  TauPlanCache* const plans = gs().plans;
  int& plan_next = gs().plan_next;
  const void* key[14] = {gpoint_flavor, band_lims_gpt, minor_limits_gpt_lower, minor_limits_gpt_upper, kminor_start_lower,
                         kminor_start_upper, idx_minor_lower, idx_minor_upper, idx_minor_scaling_lower,
                         idx_minor_scaling_upper, minor_scales_with_density_lower, minor_scales_with_density_upper,
                         scale_by_complement_lower, scale_by_complement_upper};
  // Tables in HOST memory (what the Fortran frontend passes) are fingerprinted, so a plan is never reused for
  // different contents at the same address.  Device-resident tables cannot be inspected without draining the
  // stream: their owner calls rte_hip_invalidate_plans() after (re)uploading tables (frontend.GasOptics does).
  auto fnv = [](unsigned h, const void* p, size_t bytes) {
    const unsigned char* b = (const unsigned char*)p;
    for (size_t i = 0; i < bytes; ++i) h = (h ^ b[i]) * 16777619u;
    return h;
  };
  unsigned fp = 2166136261u;
  if (!rte::is_device_pointer(band_lims_gpt)) {
    fp = fnv(fp, band_lims_gpt, sizeof(int) * 2 * nbnd);
    fp = fnv(fp, gpoint_flavor, sizeof(int) * 2 * ngpt);
    fp = fnv(fp, minor_limits_gpt_lower, sizeof(int) * 2 * nlo);
    fp = fnv(fp, minor_limits_gpt_upper, sizeof(int) * 2 * nup);
    fp = fnv(fp, kminor_start_lower, sizeof(int) * nlo);
    fp = fnv(fp, kminor_start_upper, sizeof(int) * nup);
    fp = fnv(fp, idx_minor_lower, sizeof(int) * nlo);
    fp = fnv(fp, idx_minor_upper, sizeof(int) * nup);
    fp = fnv(fp, idx_minor_scaling_lower, sizeof(int) * nlo);
    fp = fnv(fp, idx_minor_scaling_upper, sizeof(int) * nup);
    fp = fnv(fp, minor_scales_with_density_lower, sizeof(Bool) * nlo);
    fp = fnv(fp, minor_scales_with_density_upper, sizeof(Bool) * nup);
    fp = fnv(fp, scale_by_complement_lower, sizeof(Bool) * nlo);
    fp = fnv(fp, scale_by_complement_upper, sizeof(Bool) * nup);
  }
  const int dims[7] = {nbnd, ngpt, nlo, nup, *nminorklower_, *nminorkupper_, (int)fp};
  int plan_slot = -1;
  for (int i = 0; i < NPLAN; ++i)
    if (plans[i].matches(key, dims, gs().plan_epoch)) plan_slot = i;
  const bool plan_hit = plan_slot >= 0;
  if (!plan_hit) { plan_slot = plan_next; plan_next = (plan_next + 1) % NPLAN; }
  TauPlanCache& cache = plans[plan_slot];
  if (!plan_hit) {
    cache.set(key, dims, gs().plan_epoch);
    const int* bl = c.host(band_lims_gpt, (size_t)2 * nbnd);
    const int* ml[2] = {c.host(minor_limits_gpt_lower, (size_t)2 * nlo), c.host(minor_limits_gpt_upper, (size_t)2 * nup)};
    const int* ks[2] = {c.host(kminor_start_lower, (size_t)nlo), c.host(kminor_start_upper, (size_t)nup)};
    const int* im[2] = {c.host(idx_minor_lower, (size_t)nlo), c.host(idx_minor_upper, (size_t)nup)};
    const int* is[2] = {c.host(idx_minor_scaling_lower, (size_t)nlo), c.host(idx_minor_scaling_upper, (size_t)nup)};
    const Bool* sd[2] = {c.host(minor_scales_with_density_lower, (size_t)nlo), c.host(minor_scales_with_density_upper, (size_t)nup)};
    const Bool* sc[2] = {c.host(scale_by_complement_lower, (size_t)nlo), c.host(scale_by_complement_upper, (size_t)nup)};
    const int nn[2] = {nlo, nup};
    const int nk2[2] = {*nminorklower_, *nminorkupper_};
    // Eligibility of the production kernels: every band and every minor interval is made of whole,
    // aligned chunks of gw g-points and lies inside one band, k-offsets are even, at most MAXM intervals
    // per (band, regime).
    // The stage width gw is 16 g-points when everything is 16-aligned (g256 / g224 tables), else 8 (g128 / g112).
    auto aligned = [&](int w) {
      bool al_ = ngpt % w == 0;
      for (int b = 0; b < nbnd; ++b) al_ = al_ && (bl[2 * b] - 1) % w == 0 && bl[2 * b + 1] % w == 0;
      for (int r = 0; r < 2; ++r)
        for (int i = 0; i < nn[r]; ++i) al_ = al_ && (ml[r][2 * i] - 1) % w == 0 && ml[r][2 * i + 1] % w == 0;
      return al_;
    };
    const int gw = aligned(16) ? 16 : (aligned(8) ? 8 : 0);
    bool ok = gw > 0 && nbnd <= MAXB;
    cache.bands.assign(nbnd > 0 ? nbnd : 1, BandMeta{});
    {
      const int* gf = c.host(gpoint_flavor, (size_t)2 * ngpt);
      for (int b = 0; b < nbnd; ++b) {
        BandMeta& bmh = cache.bands[b];
        bmh.gS = bl[2 * b] - 1; bmh.gE = bl[2 * b + 1] - 1;
        bmh.flav[0] = gf[2 * bmh.gS] - 1; bmh.flav[1] = gf[2 * bmh.gS + 1] - 1;
      }
    }
    for (int r = 0; r < 2 && ok; ++r) {
      ok = ok && (nn[r] == 0 || nk2[r] % 2 == 0);
      for (int i = 0; i < nn[r] && ok; ++i) {
        ok = ok && (ks[r][i] - 1) % 2 == 0;
        int band = -1;
        for (int b = 0; b < nbnd; ++b)
          if (ml[r][2 * i] >= bl[2 * b] && ml[r][2 * i + 1] <= bl[2 * b + 1]) band = b;
        ok = ok && band >= 0;
        if (!ok) break;
        BandMeta& bmh = cache.bands[band];
        if (bmh.cnt[r] >= MAXM) { ok = false; break; }
        MinorMeta& m = bmh.m[r][bmh.cnt[r]++];  // interval order is preserved (ascending i)
        m.mS = ml[r][2 * i] - 1; m.mE = ml[r][2 * i + 1] - 1;
        m.idx_minor = im[r][i]; m.idx_scaling = is[r][i]; m.kstart = ks[r][i] - 1;
        m.flags = (sd[r][i] ? 1 : 0) | (sc[r][i] ? 2 : 0);
      }
    }
    {  // checksum of everything the plan depends on, in the order tables_guard_kernel walks it
      const int* gf = c.host(gpoint_flavor, (size_t)2 * ngpt);
      const int* ia[10] = {gf, bl, ml[0], ml[1], ks[0], ks[1], im[0], im[1], is[0], is[1]};
      const int in[10] = {2 * ngpt, 2 * nbnd, 2 * nlo, 2 * nup, nlo, nup, nlo, nup, nlo, nup};
      const Bool* ba[4] = {sd[0], sd[1], sc[0], sc[1]};
      const int bn[4] = {nlo, nup, nlo, nup};
      unsigned h = 0, base = 0;
      for (int a_ = 0; a_ < 10; ++a_) {
        for (int i = 0; i < in[a_]; ++i) h += guard_term((unsigned)ia[a_][i], base + (unsigned)i);
        base += (unsigned)in[a_];
      }
      for (int a_ = 0; a_ < 4; ++a_) {
        for (int i = 0; i < bn[a_]; ++i) h += guard_term(ba[a_][i] ? 1u : 0u, base + (unsigned)i);
        base += (unsigned)bn[a_];
      }
      cache.guard = h;
    }
    cache.fast_ok = ok;
    cache.gw = ok ? gw : 0;
    cache.uploads_pending = true;
  }
  // A deferred zero fill turns the accumulate into an overwrite of the g-points the bands cover; if the bands do not
  // tile 1..ngpt the fill is executed after all (zero_array would have zeroed the uncovered g-points too)
  bool overwrite_ok = overwrite;
  if (overwrite) {
    std::vector<char> covered((size_t)ngpt, 0);
    for (const BandMeta& bmh : cache.bands)
      for (int g = bmh.gS; g <= bmh.gE && g < ngpt; ++g)
        if (g >= 0) covered[g] = 1;
    for (int g = 0; g < ngpt; ++g) overwrite_ok = overwrite_ok && covered[g];
    if (!overwrite_ok) HIP_CHECK(hipMemsetAsync(d_tau, 0, sizeof(Float) * ncl * ngpt, st));
  }
  auto al = [](const void* q, size_t n) { return ((uintptr_t)q % n) == 0; };
  const bool fast = cache.fast_ok && ncol >= 512 && !g_tau_force_direct && ncl < ((size_t)1 << 29) &&
                    al(d_fmajor, 16) && al(d_fminor, 16) && al(d_col_mix, 16) && al(d_jeta, 8) &&
                    (!rh || (overwrite_ok && g_geom_variant == 2 && nflav <= MAXFLAV && neta < 31 && ntemp < 31 &&
                             npres + 1 < 63));  // (fused: the bands tile the g-points -- else tau was zero-filled above --
                                                //  and the bit-mask geometry, which counts the Rayleigh rows)
  // the direct Rayleigh + combine kernel of the fused entry: everything (run_if == nullptr), only when the guard
  // fired (run_if = the flag), or the worklist entries
  auto rayleigh_direct = [&](const int* run_if, const int* wl, int wl_tile) {
    const size_t items = (size_t)cdiv(ncol, 256) * nlay * nbnd;
    const unsigned blocks = (unsigned)((run_if || wl) ? (items < 2048 ? items : 2048) : (items < 262144 ? items : 262144));
    hipLaunchKernelGGL(tau_rayleigh_kernel, dim3(blocks), dim3(256), 0, st, ncol, nlay, nbnd, ngpt, neta, ntemp,
                       *idx_h2o_, d_gpoint_flavor, d_band_lims, d_krayl, d_col_dry, d_col_gas, d_fminor, d_jeta,
                       d_tropo, d_jtemp, (Float*)nullptr, cb, run_if, wl, wl_tile);
  };

  // native-layout direct kernel: always correct; the whole call when the fast path does not apply,
  // otherwise armed only if some column has overlapping regimes (device-side flag)
  int* plan = (int*)rte::scratch(sizeof(int) * ((size_t)2 * nbnd + (size_t)nbnd * (nlo + nup) + 2));
  lo.cnt = plan; up.cnt = plan + nbnd;
  lo.list = plan + 2 * nbnd; up.list = plan + 2 * nbnd + (size_t)nbnd * nlo;
  TauArgs a;
  a.ncol = ncol; a.nlay = nlay; a.ngpt = ngpt; a.neta = neta; a.npres = npres; a.ntemp = ntemp;
  a.idx_h2o = *idx_h2o_;
  a.gpoint_flavor = d_gpoint_flavor; a.band_lims_gpt = d_band_lims;
  a.kmajor = d_kmajor; a.lower = lo; a.upper = up;
  a.lim = lim; a.tropo = d_tropo; a.col_mix = d_col_mix; a.fmajor = d_fmajor; a.fminor = d_fminor;
  a.play = d_play; a.tlay = d_tlay; a.col_gas = d_col_gas; a.jeta = d_jeta; a.jtemp = d_jtemp; a.jpress = d_jpress;
  a.tau = d_tau; a.overwrite = overwrite_ok; a.add_bybnd = d_add;
  a.run_if = fast ? overlap : nullptr;
  if (!fast) {
    rte::ProfScope p("tau_absorption_kernel");
    hipLaunchKernelGGL(tropo_limits_kernel, dim3(cdiv(ncol, 256)), dim3(256), 0, st, ncol, nlay, d_play, d_tropo, lim,
                       overlap, irregular);
    hipLaunchKernelGGL(plan_minor_kernel, dim3(1), dim3(RTE_WAVE), 0, st, nbnd, d_band_lims, nlo, lo.limits,
                       (int*)lo.cnt, (int*)lo.list);
    hipLaunchKernelGGL(plan_minor_kernel, dim3(1), dim3(RTE_WAVE), 0, st, nbnd, d_band_lims, nup, up.limits,
                       (int*)up.cnt, (int*)up.list);
    const size_t tiles = (size_t)cdiv(ncol, 256) * nlay * nbnd;
    hipLaunchKernelGGL(tau_absorption_kernel, dim3((unsigned)(tiles < 1048576 ? tiles : 1048576)), dim3(256), 0, st, a,
                       nbnd);
    if (rh) rayleigh_direct(nullptr, nullptr, 0);
    return;
  }
  // ---- production path: g-fastest copies of the three tables (scratch, this call only)
  const int TE = ntemp * neta, nkl = *nminorklower_, nku = *nminorkupper_;
  Float* kmaj_g = (Float*)rte::scratch(sizeof(Float) * tn * (npres + 1) * ngpt);
  Float* klo_g = (Float*)rte::scratch(sizeof(Float) * tn * (nkl > 0 ? nkl : 1));
  Float* kup_g = (Float*)rte::scratch(sizeof(Float) * tn * (nku > 0 ? nku : 1));
  Float* kray_g = nullptr;
  // band metadata lives in a persistent device buffer and is uploaded only when the host plan was rebuilt
  // (a per-call copy from pageable host memory stalls the submitting thread)
  bool bm_fresh = false;
  BandMeta* d_bm = (BandMeta*)rte::persistent(plan_slot, sizeof(BandMeta) * MAXB, &bm_fresh);
  {
    rte::ProfScope p("relayout_gfast_kernel");
    if (bm_fresh || cache.uploads_pending) {
      HIP_CHECK(hipMemcpyAsync(d_bm, cache.bands.data(), sizeof(BandMeta) * nbnd, hipMemcpyHostToDevice, st));
      HIP_CHECK(hipStreamSynchronize(st));  // cache.bands is host memory that the next rebuild overwrites
      cache.uploads_pending = false;
    }
  }
  {  // one launch: layer limits, minor-interval plans, g-fastest table copies, plan guard (tau_setup_kernel)
    TauSetupArgs sa{};
    sa.ncol = ncol; sa.nlay = nlay; sa.nbnd = nbnd; sa.TE = TE;
    sa.play = d_play; sa.tropo = d_tropo; sa.lim = lim; sa.overlap = overlap; sa.irregular = irregular;
    sa.band_lims = d_band_lims;
    sa.nminor[0] = nlo; sa.minor_limits[0] = lo.limits; sa.cnt[0] = (int*)lo.cnt; sa.list[0] = (int*)lo.list;
    sa.nminor[1] = nup; sa.minor_limits[1] = up.limits; sa.cnt[1] = (int*)up.cnt; sa.list[1] = (int*)up.list;
    unsigned nb = 0;
    auto table = [&](const Float* in, Float* out, int nouter, int ng) {
      if (ng <= 0) return;
      const int t = sa.ntab++;
      sa.tin[t] = in; sa.tout[t] = out; sa.nouter[t] = nouter; sa.ng[t] = ng; sa.first_block[t] = (int)nb;
      nb += (unsigned)cdiv(ng, 32) * nouter;
    };
    table(d_kmajor, kmaj_g, npres + 1, ngpt);
    table(lo.kminor, klo_g, 1, nkl);
    table(up.kminor, kup_g, 1, nku);
    if (rh) {  // the Rayleigh table (ntemp, neta, ngpt, 2): one g-fastest copy per regime
      kray_g = (Float*)rte::scratch(sizeof(Float) * tn * ngpt * 2);
      for (int r = 0; r < 2; ++r) table(d_krayl + tn * ngpt * r, kray_g + tn * ngpt * r, 1, ngpt);
    }
    sa.first_block[sa.ntab] = (int)nb;
    // plan guard: the tables on the device must be the ones the cached plan was built from
    const int* ia[10] = {d_gpoint_flavor, d_band_lims, lo.limits, up.limits, lo.kminor_start, up.kminor_start,
                         lo.idx_minor, up.idx_minor, lo.idx_minor_scaling, up.idx_minor_scaling};
    const int in[10] = {2 * ngpt, 2 * nbnd, 2 * nlo, 2 * nup, nlo, nup, nlo, nup, nlo, nup};
    const Bool* ba[4] = {lo.scales_with_density, up.scales_with_density, lo.scale_by_complement, up.scale_by_complement};
    const int bn[4] = {nlo, nup, nlo, nup};
    for (int i = 0; i < 10; ++i) { sa.gt.ip[i] = ia[i]; sa.gt.in[i] = in[i]; }
    for (int i = 0; i < 4; ++i) { sa.gt.bp[i] = ba[i]; sa.gt.bn[i] = bn[i]; }
    sa.guard_expected = cache.guard; sa.stale = d_stale;
    sa.b_plan = (unsigned)cdiv(ncol, 256); sa.b_tab = sa.b_plan + 2; sa.b_guard = sa.b_tab + nb;
    rte::ProfScope p("tau_absorption_setup");
    hipLaunchKernelGGL(tau_setup_kernel, dim3(sa.b_guard + 1), dim3(256), sizeof(Float) * TE * 33, st, sa);
  }
  TauV5 v;
  v.ncol = ncol; v.nlay = nlay; v.ngpt = ngpt; v.nbnd = nbnd; v.ntemp = ntemp; v.TE = TE; v.idx_h2o = *idx_h2o_;
  v.nk_lo = nkl; v.nk_up = nku;
  v.band_lims = d_band_lims; v.gpoint_flavor = d_gpoint_flavor; v.bmeta = d_bm;
  v.kmaj = kmaj_g; v.klo = klo_g; v.kup = kup_g;
  v.lim = lim; v.jeta = d_jeta; v.jtemp = d_jtemp; v.jpress = d_jpress; v.tropo = d_tropo;
  v.col_mix = d_col_mix; v.fmajor = d_fmajor; v.fminor = d_fminor; v.play = d_play; v.tlay = d_tlay;
  v.col_gas = d_col_gas; v.tau = d_tau; v.skip_if = overlap; v.overwrite = overwrite_ok; v.add_bybnd = d_add;
  v.atomic_ok = rte::is_device_memory(d_tau);
  v.rf = RaylFuse{};
  if (rh) {
    v.rf.krayl_g[0] = kray_g; v.rf.krayl_g[1] = kray_g + tn * ngpt; v.rf.col_dry = d_col_dry;
    v.rf.cld_tau = cb.cld_tau; v.rf.cld_ssa = cb.cld_ssa; v.rf.cld_g = cb.cld_g; v.rf.ssa = cb.ssa; v.rf.g = cb.g;
  }
#ifndef V7_BS
#define V7_BS 256
#define V7_MINW 2
#define V7_HW 16
#define V7_SLAB SLAB_FLOATS
#endif
  constexpr int BS = V7_BS;
  v.worklist = worklist;
  hipStream_t aux = nullptr;
  const bool use_v9 = g_tau_variant == 9 || cache.gw != 16 || d_add != nullptr || rh != nullptr;  // the single-role kernel exists for 16-wide stages only
  if (use_v9) {
#ifdef EXP_CLOCKS
    v.clocks = (unsigned long long*)rte::scratch(64);
    HIP_CHECK(hipMemsetAsync(v.clocks, 0, 64, st));
#endif
    constexpr int NCW = V9_NCW, NLW = V9_NLW, SLAB9 = V9_SLAB;  // compute + loader waves, 2 x 68 KB slab: one block per CU
    const unsigned tiles = cdiv(ncol, NCW * 64);
    const bool geom2 = g_geom_variant == 2 && nflav <= MAXFLAV && neta < 31 && ntemp < 31 && npres + 1 < 63;
    const bool share = share_boxes() && geom2 && NCW * 64 == 512 && !c.any_host() && !rh;
    TileGeom* d_geom;
    gs().shared.seq = -1;
    if (share) {  // the geometry outlives this call: a compute_Planck_source call right behind it may use it
      const size_t need = sizeof(TileGeom) * (size_t)tiles * nlay;
      if (gs().shared.cap < need) {
        HIP_CHECK(hipStreamSynchronize(st));
        if (gs().shared.geom) HIP_CHECK(hipFree(gs().shared.geom));
        HIP_CHECK(hipMalloc((void**)&gs().shared.geom, need));
        if (!gs().shared.valid) HIP_CHECK(hipMalloc((void**)&gs().shared.valid, sizeof(int)));
        gs().shared.cap = need;
      }
      d_geom = gs().shared.geom;
      if (valid_word == nullptr)  // just allocated (otherwise it was cleared with this call's other flag words)
        hipLaunchKernelGGL(zero_words_kernel, dim3(1), dim3(64), 0, st, gs().shared.valid, 1u, (int*)nullptr, 0u, (int*)nullptr, 0u);
      gs().shared.jeta = jeta; gs().shared.jtemp = jtemp; gs().shared.jpress = jpress; gs().shared.tropo = tropo;
      gs().shared.ncol = ncol; gs().shared.nlay = nlay; gs().shared.nflav = nflav; gs().shared.nbnd = nbnd; gs().shared.gw = cache.gw;
      gs().shared.seq = rte::call_seq();
    } else {
      d_geom = (TileGeom*)rte::scratch(sizeof(TileGeom) * (size_t)tiles * nlay);
    }
    const dim3 grid(tiles, nlay), blk((NCW + NLW) * 64);
    const size_t dyn = sizeof(BandMeta) * nbnd;
    const TileGeom* cg = d_geom;
    Geom2Args ga{};
    ga.ncol = ncol; ga.nlay = nlay; ga.nbnd = nbnd; ga.nflav = nflav; ga.slab_floats = SLAB9; ga.planck = false;
    ga.lim = lim; ga.jeta = d_jeta; ga.jtemp = d_jtemp; ga.jpress = d_jpress; ga.tropo = d_tropo; ga.bmeta = d_bm;
    ga.skip_if = overlap; ga.worklist = v.worklist; ga.valid_out = share ? gs().shared.valid : nullptr;
    ga.extra_planes = rh ? 2 : 0;
    ga.irregular = irregular;
    ga.stat = stats_dev() + 2;
    if (share_masks() && gs().imask.seq >= 0 && gs().imask.seq + 1 == rte::call_seq() && gs().imask.jeta == jeta && gs().imask.jtemp == jtemp &&
        gs().imask.jpress == jpress && gs().imask.tropo == tropo && gs().imask.ncol == ncol && gs().imask.nlay == nlay &&
        gs().imask.nflav == nflav && !c.any_host()) {
      ga.imask = gs().imask.buf;
      ga.imask_nblk = cdiv(ncol, 256);
    }
#define RTE_LAUNCH_TAU9R_(GW, MMV, RV) \
  hipLaunchKernelGGL((tau_absorption_v9_kernel<NCW, NLW, SLAB9, true, GW, MMV, false, RV>), grid, blk, dyn, st, v, cg)
#define RTE_LAUNCH_TAU9_(GW, AB)                                                                                  \
  do {                                                                                                            \
    if (overwrite_ok) hipLaunchKernelGGL((tau_absorption_v9_kernel<NCW, NLW, SLAB9, true, GW, 4, AB>), grid, blk, dyn, st, v, cg); \
    else hipLaunchKernelGGL((tau_absorption_v9_kernel<NCW, NLW, SLAB9, false, GW, 4, AB>), grid, blk, dyn, st, v, cg); \
  } while (0)
#define RTE_LAUNCH_TAU9(GW)                                                                                       \
  do {                                                                                                            \
    {                                                                                                             \
      rte::ProfScope p("tau_absorption_setup");                                                                   \
      if (geom2) hipLaunchKernelGGL((tile_geom2_kernel<NCW * 64, GW>), grid, dim3(NCW * 64), 0, st, ga, d_geom);  \
      else hipLaunchKernelGGL((tau_geom_kernel<NCW * 64, GW>), grid, dim3(NCW * 64), 0, st, v, d_geom, SLAB9);    \
    }                                                                                                             \
    aux = rte::aux_fork(); /* the worklist is complete: its kernel may run beside the slab kernel */             \
    rte::ProfScope p("tau_absorption_kernel");                                                                    \
    if (rh) {                                                                                                     \
      if (cb.cld_tau) RTE_LAUNCH_TAU9R_(GW, 4, 2);                                                                \
      else            RTE_LAUNCH_TAU9R_(GW, 4, 1);                                                                \
    } else if (d_add) RTE_LAUNCH_TAU9_(GW, true); else RTE_LAUNCH_TAU9_(GW, false);                               \
  } while (0)
    if (cache.gw == 16) RTE_LAUNCH_TAU9(16); else RTE_LAUNCH_TAU9(8);
#undef RTE_LAUNCH_TAU9
#undef RTE_LAUNCH_TAU9_
#undef RTE_LAUNCH_TAU9R_
  } else {
    rte::ProfScope p("tau_absorption_kernel");
    // <min waves per SIMD, g-points per register chunk>: measured best of {2,3} x {4,8,16} on MI355X
    hipLaunchKernelGGL((tau_absorption_v7_kernel<BS, V7_MINW, V7_HW, V7_SLAB>), dim3(cdiv(ncol, BS), nlay), dim3(BS), sizeof(BandMeta) * nbnd, st,
                       v);
  }
#ifdef EXP_CLOCKS
  if (use_v9) {
    unsigned long long h[8];
    HIP_CHECK(hipMemcpyAsync(h, v.clocks, 64, hipMemcpyDeviceToHost, st));
    HIP_CHECK(hipStreamSynchronize(st));
    const double n = (double)cdiv(ncol, V9_NCW * 64) * nlay * (ngpt / 16);
    fprintf(stderr, "clocks/stage: looptop %.0f requests %.0f barrier %.0f major %.0f minor %.0f stores %.0f\n", h[0] / n, h[1] / n, h[2] / n,
            h[3] / n, h[4] / n, h[5] / n);
  }
#endif
  {
    // runs only when *overlap != 0 (some column's lower and upper layer ranges intersect)
    rte::ProfScope p("tau_absorption_fallback");
    hipLaunchKernelGGL(tau_absorption_kernel, dim3(2048), dim3(256), 0, st, a, nbnd);
    // tiles whose LUT bounding box exceeded the LDS slab
    TauArgs aw = a;
    aw.run_if = nullptr;
    if (rh) rayleigh_direct(overlap, nullptr, 0);  // the same (column, layer, band) hold tau_abs in tau: Rayleigh + combine in place
    hipStream_t main_st = st;
    if (aux) st = aux;
    GfastTabs gft{};
    // (tau_direct_column_g addresses the g-fastest copies with 32-bit element offsets: tables beyond 2^31 elements take the
    //  native-layout worklist kernel)
    const bool offsets_fit = (size_t)(npres + 1) * TE * ngpt < ((size_t)1 << 31) && (size_t)TE * nkl < ((size_t)1 << 31) &&
                             (size_t)TE * nku < ((size_t)1 << 31);
    if (!g_worklist_native && offsets_fit) { gft.kmaj = kmaj_g; gft.klo = klo_g; gft.kup = kup_g; gft.nkl = nkl; gft.nku = nku; }
    if (gft.kmaj)
      hipLaunchKernelGGL(tau_absorption_worklist_kernel<true>, dim3(aux ? 16384 : 4096), dim3(aux ? 64 : 256), 0, st, aw, gft,
                         (const int*)v.worklist, use_v9 ? V9_NCW * 64 : BS, stats_dev() + 0);
    else
      hipLaunchKernelGGL(tau_absorption_worklist_kernel<false>, dim3(aux ? 16384 : 4096), dim3(aux ? 64 : 256), 0, st, aw, gft,
                         (const int*)v.worklist, use_v9 ? V9_NCW * 64 : BS, stats_dev() + 0);
    if (rh) rayleigh_direct(nullptr, (const int*)v.worklist, V9_NCW * 64);  // (lambda launches on st)
    st = main_st;
    if (aux) rte::aux_join();
  }
  RTE_CATCH(api_name)
}

extern "C" {
void rrtmgp_compute_tau_absorption(
    const int* ncol_, const int* nlay_, const int* nbnd_, const int* ngpt_, const int* ngas_,
    const int* nflav_, const int* neta_, const int* npres_, const int* ntemp_,
    const int* nminorlower_, const int* nminorklower_, const int* nminorupper_,
    const int* nminorkupper_, const int* idx_h2o_, const int* gpoint_flavor,
    const int* band_lims_gpt, const Float* kmajor, const Float* kminor_lower,
    const Float* kminor_upper, const int* minor_limits_gpt_lower, const int* minor_limits_gpt_upper,
    const Bool* minor_scales_with_density_lower, const Bool* minor_scales_with_density_upper,
    const Bool* scale_by_complement_lower, const Bool* scale_by_complement_upper,
    const int* idx_minor_lower, const int* idx_minor_upper, const int* idx_minor_scaling_lower,
    const int* idx_minor_scaling_upper, const int* kminor_start_lower, const int* kminor_start_upper,
    const Bool* tropo, const Float* col_mix, const Float* fmajor, const Float* fminor,
    const Float* play, const Float* tlay, const Float* col_gas, const int* jeta, const int* jtemp,
    const int* jpress, Float* tau) {
  tau_absorption_impl("rrtmgp_compute_tau_absorption", *ncol_, *nlay_, *nbnd_, *ngpt_, *ngas_, *nflav_, *neta_, *npres_,
                      *ntemp_, *nminorlower_, *nminorklower_, *nminorupper_, *nminorkupper_, *idx_h2o_, gpoint_flavor,
                      band_lims_gpt, kmajor, kminor_lower, kminor_upper, minor_limits_gpt_lower, minor_limits_gpt_upper,
                      minor_scales_with_density_lower, minor_scales_with_density_upper, scale_by_complement_lower,
                      scale_by_complement_upper, idx_minor_lower, idx_minor_upper, idx_minor_scaling_lower,
                      idx_minor_scaling_upper, kminor_start_lower, kminor_start_upper, tropo, col_mix, fmajor, fminor, play,
                      tlay, col_gas, jeta, jtemp, jpress, tau, nullptr);
}
// Library extension (scalars by value): compute_tau_absorption followed, in the same pass, by the band-wise increment
// tau(:,:,g) += tau_bybnd(:,:,band(g)) -- rte_inc_1scalar_by_1scalar_bybnd (rte/kernels/mo_optical_props_kernels.F90), what
// the all-sky driver does with its absorbing clouds (examples/all-sky/rrtmgp_allsky.F90:374); saves a read and a write of tau.
int rte_hip_compute_tau_absorption_inc_bybnd(
    int ncol, int nlay, int nbnd, int ngpt, int ngas, int nflav, int neta, int npres, int ntemp, int nminorlower,
    int nminorklower, int nminorupper, int nminorkupper, int idx_h2o, const int* gpoint_flavor,
    const int* band_lims_gpt, const Float* kmajor, const Float* kminor_lower,
    const Float* kminor_upper, const int* minor_limits_gpt_lower, const int* minor_limits_gpt_upper,
    const Bool* minor_scales_with_density_lower, const Bool* minor_scales_with_density_upper,
    const Bool* scale_by_complement_lower, const Bool* scale_by_complement_upper,
    const int* idx_minor_lower, const int* idx_minor_upper, const int* idx_minor_scaling_lower,
    const int* idx_minor_scaling_upper, const int* kminor_start_lower, const int* kminor_start_upper,
    const Bool* tropo, const Float* col_mix, const Float* fmajor, const Float* fminor,
    const Float* play, const Float* tlay, const Float* col_gas, const int* jeta, const int* jtemp,
    const int* jpress, Float* tau, const Float* tau_bybnd) {
  tau_absorption_impl("rte_hip_compute_tau_absorption_inc_bybnd", ncol, nlay, nbnd, ngpt, ngas, nflav, neta, npres, ntemp,
                      nminorlower, nminorklower, nminorupper, nminorkupper, idx_h2o, gpoint_flavor, band_lims_gpt, kmajor,
                      kminor_lower, kminor_upper, minor_limits_gpt_lower, minor_limits_gpt_upper,
                      minor_scales_with_density_lower, minor_scales_with_density_upper, scale_by_complement_lower,
                      scale_by_complement_upper, idx_minor_lower, idx_minor_upper, idx_minor_scaling_lower,
                      idx_minor_scaling_upper, kminor_start_lower, kminor_start_upper, tropo, col_mix, fmajor, fminor, play,
                      tlay, col_gas, jeta, jtemp, jpress, tau, tau_bybnd);
  return 0;
}
// Library extension (scalars by value): the SW gas optics of one call -- compute_tau_absorption, compute_tau_rayleigh and
// combine_abs_and_rayleigh (2-stream branch, mo_gas_optics_rrtmgp.F90:1983-2002), optionally followed by the band-wise
// increment by 2-stream cloud properties -- in ONE pass over (column, layer, g-point): the absorption optical depth
// never goes to memory (-21.5 GB per step at 1e5 x 60 x 224).  Same operations in the same order on the same doubles
// as the chain compute_tau_absorption -> rte_hip_tau_rayleigh_combine_2str: bit-identical tau, ssa, g.
int rte_hip_gas_optics_sw_2str(
    int ncol, int nlay, int nbnd, int ngpt, int ngas, int nflav, int neta, int npres, int ntemp, int nminorlower,
    int nminorklower, int nminorupper, int nminorkupper, int idx_h2o, const int* gpoint_flavor,
    const int* band_lims_gpt, const Float* kmajor, const Float* kminor_lower,
    const Float* kminor_upper, const int* minor_limits_gpt_lower, const int* minor_limits_gpt_upper,
    const Bool* minor_scales_with_density_lower, const Bool* minor_scales_with_density_upper,
    const Bool* scale_by_complement_lower, const Bool* scale_by_complement_upper,
    const int* idx_minor_lower, const int* idx_minor_upper, const int* idx_minor_scaling_lower,
    const int* idx_minor_scaling_upper, const int* kminor_start_lower, const int* kminor_start_upper,
    const Bool* tropo, const Float* col_mix, const Float* fmajor, const Float* fminor,
    const Float* play, const Float* tlay, const Float* col_gas, const int* jeta, const int* jtemp,
    const int* jpress, const Float* krayl, const Float* col_dry, Float* tau, Float* ssa, Float* g,
    const Float* cld_tau, const Float* cld_ssa, const Float* cld_g) {
  RaylHost rh{krayl, col_dry, cld_tau, cld_ssa, cld_g, ssa, g};
  tau_absorption_impl("rte_hip_gas_optics_sw_2str", ncol, nlay, nbnd, ngpt, ngas, nflav, neta, npres, ntemp,
                      nminorlower, nminorklower, nminorupper, nminorkupper, idx_h2o, gpoint_flavor, band_lims_gpt, kmajor,
                      kminor_lower, kminor_upper, minor_limits_gpt_lower, minor_limits_gpt_upper,
                      minor_scales_with_density_lower, minor_scales_with_density_upper, scale_by_complement_lower,
                      scale_by_complement_upper, idx_minor_lower, idx_minor_upper, idx_minor_scaling_lower,
                      idx_minor_scaling_upper, kminor_start_lower, kminor_start_upper, tropo, col_mix, fmajor, fminor, play,
                      tlay, col_gas, jeta, jtemp, jpress, tau, nullptr, &rh);
  return 0;
}
}  // extern "C"
// compute_tau_rayleigh, optionally fused with combine_abs_and_rayleigh (tau_abs != nullptr: tau_rayleigh is not written)
static void tau_rayleigh_impl(const char* api_name, int ncol, int nlay, int nbnd, int ngpt, int ngas, int nflav, int neta,
                              int ntemp, const int* gpoint_flavor, const int* band_lims_gpt, const Float* krayl,
                              int idx_h2o, const Float* col_dry, const Float* col_gas, const Float* fminor,
                              const int* jeta, const Bool* tropo, const int* jtemp, Float* tau_rayleigh,
                              const Float* tau_abs, Float* tau, Float* ssa, Float* g, const Float* cld_tau = nullptr,
                              const Float* cld_ssa = nullptr, const Float* cld_g = nullptr) {
  if (ncol <= 0 || nlay <= 0 || ngpt <= 0) return;
  RTE_TRY
  rte::Call c(api_name);
  const bool combine = tau_abs != nullptr;
  const size_t ncl = (size_t)ncol * nlay;
  const int* d_gpoint_flavor = c.in(gpoint_flavor, (size_t)2 * ngpt);
  const int* d_band_lims = c.in(band_lims_gpt, (size_t)2 * nbnd);
  const Float* d_krayl = c.in(krayl, (size_t)ntemp * neta * ngpt * 2);
  const Float* d_col_dry = c.in(col_dry, ncl);
  const Float* d_col_gas = c.in(col_gas, ncl * (ngas + 1));
  const Float* d_fminor = c.in(fminor, 4 * ncl * nflav);
  const int* d_jeta = c.in(jeta, 2 * ncl * nflav);
  const Bool* d_tropo = c.in(tropo, ncl);
  const int* d_jtemp = c.in(jtemp, ncl);
  Float* d_tau = combine ? nullptr : c.out(tau_rayleigh, ncl * ngpt);
  // the reference ABI call: the frontend combines tau and tau_rayleigh on the HOST next (combine_abs_and_rayleigh,
  // mo_gas_optics_rrtmgp.F90:666-678, :1954-2036), so in host-mirror mode the absorption optical depth the preceding
  // compute_tau_absorption call left on the device goes back to its host array with this call's output
  if (!combine) c.writeback_produced_by("rrtmgp_compute_tau_absorption");
  RaylCombine cb{nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
  if (combine && cld_tau) {
    cb.cld_tau = c.in(cld_tau, ncl * nbnd); cb.cld_ssa = c.in(cld_ssa, ncl * nbnd); cb.cld_g = c.in(cld_g, ncl * nbnd);
  }
  if (combine) {
    if (tau == tau_abs) { cb.tau = c.inout(tau, ncl * ngpt); cb.tau_abs = cb.tau; }
    else { cb.tau_abs = c.in(tau_abs, ncl * ngpt); cb.tau = c.out(tau, ncl * ngpt); }
    cb.ssa = c.out(ssa, ncl * ngpt);
    cb.g = c.out(g, ncl * ngpt);
  }
  stale_poll();
  // production kernel: 16-aligned whole-chunk bands (checked once per table pointer), aligned inputs
  BandCheck& bc_ = gs().rayl_bands;
  const void*& bl_key = bc_.key;
  int &bl_n = bc_.n, &bl_epoch = bc_.epoch;
  bool& bl_ok = bc_.ok;
  int& bl_gw = bc_.gw;
  unsigned bl_fp = 0;
  if (!rte::is_device_pointer(band_lims_gpt))  // host tables: fingerprint the contents (see compute_tau_absorption)
    for (int i = 0; i < 2 * nbnd; ++i) bl_fp = (bl_fp ^ (unsigned)band_lims_gpt[i]) * 16777619u;
  unsigned& bl_fp_seen = bc_.fp_seen;
  if (bl_key != (const void*)band_lims_gpt || bl_n != nbnd || bl_epoch != gs().plan_epoch || bl_fp != bl_fp_seen) {
    bl_fp_seen = bl_fp;
    const int* bl = c.host(band_lims_gpt, (size_t)2 * nbnd);
    auto aligned = [&](int w) {
      bool al_ = ngpt % w == 0;
      for (int b = 0; b < nbnd; ++b) al_ = al_ && (bl[2 * b] - 1) % w == 0 && bl[2 * b + 1] % w == 0;
      return al_;
    };
    bl_gw = aligned(16) ? 16 : (aligned(8) ? 8 : 0);  // g-points per stage of the production kernel
    bl_ok = bl_gw > 0;
    bl_key = band_lims_gpt; bl_n = nbnd; bl_epoch = gs().plan_epoch;
  }
  const size_t slab_bytes = sizeof(Float) * 2 * (size_t)ntemp * neta * (bl_gw + 2);
  hipStream_t st = rte::stream();
  const bool fast = bl_ok && ncol >= 512 && !g_tau_force_direct && ncl < ((size_t)1 << 29) && slab_bytes <= 64 * 1024 &&
                    ((uintptr_t)d_fminor % 16) == 0 && ((uintptr_t)d_jeta % 8) == 0;
  int* guard = nullptr;
  if (fast) {
    // plan guard: the band limits on the device must have the alignment the cached stage width assumes
    guard = (int*)rte::scratch(sizeof(int));
    HIP_CHECK(hipMemsetAsync(guard, 0, sizeof(int), st));
    hipLaunchKernelGGL(bands_guard_kernel, dim3(1), dim3(64), 0, st, nbnd, ngpt, d_band_lims, bl_gw, guard, stale_flag());
    RaylArgs q;
    q.skip_if = guard;
    q.cb = cb;
    q.nbnd = nbnd; q.band_lims = d_band_lims;
    q.ncol = ncol; q.nlay = nlay; q.ngpt = ngpt; q.neta = neta; q.ntemp = ntemp; q.idx_h2o = idx_h2o;
    q.gpoint_flavor = d_gpoint_flavor; q.jeta = d_jeta; q.jtemp = d_jtemp; q.krayl = d_krayl; q.col_dry = d_col_dry;
    q.col_gas = d_col_gas; q.fminor = d_fminor; q.tropo = d_tropo; q.tau_rayleigh = d_tau;
    rte::ProfScope p(combine ? "tau_rayleigh_combine_kernel" : "tau_rayleigh_kernel");
    const dim3 g16(ngpt / 16, cdiv(ncol, 256)), g8(ngpt / 8, cdiv(ncol, 256));
    if (bl_gw == 16) {
      if (combine) hipLaunchKernelGGL((tau_rayleigh_slab_kernel<256, 16, true>), g16, dim3(256), slab_bytes, st, q);
      else hipLaunchKernelGGL((tau_rayleigh_slab_kernel<256, 16, false>), g16, dim3(256), slab_bytes, st, q);
    } else {
      if (combine) hipLaunchKernelGGL((tau_rayleigh_slab_kernel<256, 8, true>), g8, dim3(256), slab_bytes, st, q);
      else hipLaunchKernelGGL((tau_rayleigh_slab_kernel<256, 8, false>), g8, dim3(256), slab_bytes, st, q);
    }
  }
  {
    // the direct kernel: the whole call when the production kernel does not apply, otherwise only if the guard fired
    rte::ProfScope p(fast ? "tau_rayleigh_fallback" : (combine ? "tau_rayleigh_combine_kernel" : "tau_rayleigh_kernel"));
    const size_t items = (size_t)cdiv(ncol, 256) * nlay * nbnd;
    const unsigned blocks = (unsigned)(fast ? (items < 2048 ? items : 2048) : (items < 262144 ? items : 262144));
    hipLaunchKernelGGL(tau_rayleigh_kernel, dim3(blocks), dim3(256), 0, st, ncol, nlay, nbnd, ngpt, neta, ntemp,
                       idx_h2o, d_gpoint_flavor, d_band_lims, d_krayl, d_col_dry, d_col_gas, d_fminor, d_jeta,
                       d_tropo, d_jtemp, d_tau, cb, (const int*)guard);
  }
  RTE_CATCH(api_name)
}
extern "C" {
void rrtmgp_compute_tau_rayleigh(const int* ncol_, const int* nlay_, const int* nbnd_,
                                 const int* ngpt_, const int* ngas_, const int* nflav_,
                                 const int* neta_, const int* npres_, const int* ntemp_,
                                 const int* gpoint_flavor, const int* band_lims_gpt,
                                 const Float* krayl, const int* idx_h2o_, const Float* col_dry,
                                 const Float* col_gas, const Float* fminor, const int* jeta,
                                 const Bool* tropo, const int* jtemp, Float* tau_rayleigh) {
  (void)npres_;
  tau_rayleigh_impl("rrtmgp_compute_tau_rayleigh", *ncol_, *nlay_, *nbnd_, *ngpt_, *ngas_, *nflav_, *neta_, *ntemp_,
                    gpoint_flavor, band_lims_gpt, krayl, *idx_h2o_, col_dry, col_gas, fminor, jeta, tropo, jtemp,
                    tau_rayleigh, nullptr, nullptr, nullptr, nullptr);
}
// Library extension (scalars by value): compute_tau_rayleigh FUSED with the 2-stream branch of the frontend's
// combine_abs_and_rayleigh (rrtmgp/frontend/mo_gas_optics_rrtmgp.F90:666-678, :1983-2002): tau = tau_abs + tau_rayleigh,
// ssa = tau_rayleigh / tau, g = 0, without the tau_rayleigh array's round trip through memory.  tau may be tau_abs.
// With cloud properties by band it also performs the all-sky driver's band-wise increment
// (examples/all-sky/rrtmgp_allsky.F90:395), saving a read and a write of the three gas arrays.
int rte_hip_tau_rayleigh_combine_2str(int ncol, int nlay, int nbnd, int ngpt, int ngas, int nflav, int neta, int ntemp,
                                      const int* gpoint_flavor, const int* band_lims_gpt, const Float* krayl, int idx_h2o,
                                      const Float* col_dry, const Float* col_gas, const Float* fminor, const int* jeta,
                                      const Bool* tropo, const int* jtemp, const Float* tau_abs, Float* tau, Float* ssa,
                                      Float* g, const Float* cld_tau, const Float* cld_ssa, const Float* cld_g) {
  // cld_* (ncol, nlay, nbnd), all three or none: additionally increment_2stream_by_2stream_bybnd with these properties
  tau_rayleigh_impl("rte_hip_tau_rayleigh_combine_2str", ncol, nlay, nbnd, ngpt, ngas, nflav, neta, ntemp, gpoint_flavor,
                    band_lims_gpt, krayl, idx_h2o, col_dry, col_gas, fminor, jeta, tropo, jtemp, nullptr, tau_abs, tau, ssa, g,
                    cld_tau, cld_tau ? cld_ssa : nullptr, cld_tau ? cld_g : nullptr);
  return 0;
}

void rrtmgp_compute_Planck_source(const int* ncol_, const int* nlay_, const int* nbnd_,
                                  const int* ngpt_, const int* nflav_, const int* neta_,
                                  const int* npres_, const int* ntemp_, const int* nPlanckTemp_,
                                  const Float* tlay, const Float* tlev, const Float* tsfc,
                                  const int* sfc_lay_, const Float* fmajor, const int* jeta,
                                  const Bool* tropo, const int* jtemp, const int* jpress,
                                  const int* gpoint_bands, const int* band_lims_gpt,
                                  const Float* pfracin, const Float* temp_ref_min,
                                  const Float* totplnk_delta, const Float* totplnk,
                                  const int* gpoint_flavor, Float* sfc_src, Float* lay_src,
                                  Float* lev_src, Float* sfc_source_Jac) {
  const int ncol = *ncol_, nlay = *nlay_, nbnd = *nbnd_, ngpt = *ngpt_, nflav = *nflav_, neta = *neta_,
            npres = *npres_, ntemp = *ntemp_, nPlanckTemp = *nPlanckTemp_;
  (void)gpoint_bands;
  if (ncol <= 0 || nlay <= 0 || ngpt <= 0) return;
  RTE_TRY
  rte::Call c("rrtmgp_compute_Planck_source");
  const size_t ncl = (size_t)ncol * nlay;
  const Float* d_tlay = c.in(tlay, ncl);
  const Float* d_tlev = c.in(tlev, (size_t)ncol * (nlay + 1));
  const Float* d_tsfc = c.in(tsfc, (size_t)ncol);
  const Float* d_fmajor = c.in(fmajor, 8 * ncl * nflav);
  const int* d_jeta = c.in(jeta, 2 * ncl * nflav);
  const Bool* d_tropo = c.in(tropo, ncl);
  const int* d_jtemp = c.in(jtemp, ncl);
  const int* d_jpress = c.in(jpress, ncl);
  const int* d_band_lims = c.in(band_lims_gpt, (size_t)2 * nbnd);
  const Float* d_pfracin = c.in(pfracin, (size_t)ntemp * neta * (npres + 1) * ngpt);
  const Float* d_totplnk = c.in(totplnk, (size_t)nPlanckTemp * nbnd);
  const int* d_gpoint_flavor = c.in(gpoint_flavor, (size_t)2 * ngpt);
  Float* d_sfc_src = c.out_lazy(sfc_src, (size_t)ncol * ngpt);  // (lazy: host-mirror mode keeps the sources on the device)
  Float* d_lay_src = c.out_lazy(lay_src, ncl * ngpt);
  Float* d_lev_src = c.out_lazy(lev_src, (size_t)ncol * (nlay + 1) * ngpt);
  Float* d_sfc_jac = c.out_lazy(sfc_source_Jac, (size_t)ncol * ngpt);
  const Float totplnk_delta_r = (Float)1 / *totplnk_delta;  // :636
  {
    const void* outs[4] = {d_sfc_src, d_lay_src, d_lev_src, d_sfc_jac};
    const size_t ob[4] = {sizeof(Float) * (size_t)ncol * ngpt, sizeof(Float) * ncl * ngpt,
                          sizeof(Float) * (size_t)ncol * (nlay + 1) * ngpt, sizeof(Float) * (size_t)ncol * ngpt};
    c.try_fork(outs, ob, 4);  // opt-in: concurrently with the compute_tau_absorption call this one follows
  }
  hipStream_t st = rte::stream();
  int* d_stale = stale_flag();
  stale_poll();
  // production kernel: 16-aligned whole-chunk bands (checked once per table pointer), aligned inputs
  BandCheck& bc_ = gs().planck_bands;
  const void*& bl_key = bc_.key;
  int &bl_n = bc_.n, &bl_epoch = bc_.epoch;
  bool& bl_ok = bc_.ok;
  int& bl_gw = bc_.gw;
  unsigned bl_fp = 0;
  if (!rte::is_device_pointer(band_lims_gpt))  // host tables: fingerprint the contents (see compute_tau_absorption)
    for (int i = 0; i < 2 * nbnd; ++i) bl_fp = (bl_fp ^ (unsigned)band_lims_gpt[i]) * 16777619u;
  unsigned& bl_fp_seen = bc_.fp_seen;
  if (bl_key != (const void*)band_lims_gpt || bl_n != nbnd || bl_epoch != gs().plan_epoch || bl_fp != bl_fp_seen) {
    bl_fp_seen = bl_fp;
    const int* bl = c.host(band_lims_gpt, (size_t)2 * nbnd);
    auto aligned = [&](int w) {
      bool al_ = ngpt % w == 0;
      for (int b = 0; b < nbnd; ++b) al_ = al_ && (bl[2 * b] - 1) % w == 0 && bl[2 * b + 1] % w == 0;
      return al_;
    };
    bl_gw = aligned(16) ? 16 : (aligned(8) ? 8 : 0);  // g-points per stage of the production kernel
    bl_ok = bl_gw > 0;
    bl_key = band_lims_gpt; bl_n = nbnd; bl_epoch = gs().plan_epoch;
  }
  auto al = [](const void* q, size_t n) { return ((uintptr_t)q % n) == 0; };
  const bool fast = bl_ok && ncol >= 512 && !g_tau_force_direct && (size_t)ncol * (nlay + 1) < ((size_t)1 << 31) &&
                    al(d_fmajor, 16) && al(d_jeta, 8);
  PlanckArgs q;
  q.ncol = ncol; q.nlay = nlay; q.ngpt = ngpt; q.neta = neta; q.npres = npres; q.ntemp = ntemp; q.nPlanckTemp = nPlanckTemp;
  q.sfc_lay = *sfc_lay_; q.tlay = d_tlay; q.tlev = d_tlev; q.tsfc = d_tsfc; q.fmajor = d_fmajor; q.jeta = d_jeta;
  q.tropo = d_tropo; q.jtemp = d_jtemp; q.jpress = d_jpress; q.band_lims_gpt = d_band_lims; q.pfracin = d_pfracin;
  q.temp_ref_min = *temp_ref_min; q.totplnk_delta_r = totplnk_delta_r; q.totplnk = d_totplnk;
  q.gpoint_flavor = d_gpoint_flavor; q.sfc_src = d_sfc_src; q.lay_src = d_lay_src; q.lev_src = d_lev_src;
  q.sfc_source_Jac = d_sfc_jac;
  if (!fast) {
    rte::ProfScope p("planck_source_kernel");
    hipLaunchKernelGGL(planck_source_kernel, dim3(cdiv(ncol, 256), nbnd), dim3(256), 0, st, q, (const int*)nullptr);
    return;
  }
  // plan guard: the band limits on the device must have the alignment the cached stage width assumes
  int* guard = (int*)rte::scratch(sizeof(int));
  constexpr int BS = 256;
  int* const worklist = (int*)rte::scratch(sizeof(int) * (1 + 2 * (size_t)cdiv(ncol, BS) * nbnd));
  const unsigned nflags = cdiv(ncol, 512) * (unsigned)nbnd;  // (512-column tile, band) flags of the specialised-wave kernel
  int* const d_flags = (int*)rte::scratch(sizeof(int) * (size_t)nflags);
  hipLaunchKernelGGL(zero_words_kernel, dim3(cdiv(nflags + 2, 256)), dim3(256), 0, st, guard, 1u, worklist, 1u, d_flags, nflags);
  hipLaunchKernelGGL(bands_guard_kernel, dim3(1), dim3(64), 0, st, nbnd, ngpt, d_band_lims, bl_gw, guard, d_stale);
  const int TE = ntemp * neta;
  Float* pf_g = (Float*)rte::scratch(sizeof(Float) * (size_t)TE * (npres + 1) * ngpt);
  {
    rte::ProfScope p("relayout_gfast_kernel");
    hipLaunchKernelGGL(relayout_gfast_kernel, dim3(cdiv(ngpt, 32), npres + 1), dim3(256), sizeof(Float) * TE * 33, st,
                       TE, npres + 1, ngpt, d_pfracin, pf_g);
  }
  PlanckV7 v;
  v.ncol = ncol; v.nlay = nlay; v.ngpt = ngpt; v.ntemp = ntemp; v.TE = TE; v.nPlanckTemp = nPlanckTemp;
  v.sfc_lay = *sfc_lay_; v.temp_ref_min = *temp_ref_min; v.totplnk_delta_r = totplnk_delta_r;
  v.band_lims = d_band_lims; v.gpoint_flavor = d_gpoint_flavor; v.jeta = d_jeta; v.jtemp = d_jtemp;
  v.jpress = d_jpress; v.tropo = d_tropo; v.pf_g = pf_g; v.totplnk = d_totplnk; v.fmajor = d_fmajor;
  v.tlay = d_tlay; v.tlev = d_tlev; v.tsfc = d_tsfc;
  v.sfc_src = d_sfc_src; v.lay_src = d_lay_src; v.lev_src = d_lev_src; v.sfc_jac = d_sfc_jac;
  v.skip_if = guard;
  v.worklist = worklist;
  int wl_tile = BS;
  const bool planck9 = (g_planck_variant == 9 || bl_gw != 16) && nlay <= 256 && nbnd <= MAXB &&
                       (size_t)ncol * (nlay + 1) < ((size_t)1 << 29);
  if (!planck9 && bl_gw != 16) {  // 8-wide stages exist only in the specialised-wave kernel
    rte::ProfScope p("planck_source_kernel");
    hipLaunchKernelGGL(planck_source_kernel, dim3(cdiv(ncol, 256), nbnd), dim3(256), 0, st, q, (const int*)nullptr);
    return;
  }
  if (planck9) {
    constexpr int NCW = 8, NLW = 2, SLAB9 = 8704;  // 8 compute + 2 loader waves, 2 x 68 KB slab: one block per CU
    wl_tile = NCW * 64;
    const unsigned tiles = cdiv(ncol, NCW * 64);
    // the geometry of the compute_tau_absorption call immediately before this one, if it is for the same arrays
    const bool shared = share_boxes() && gs().shared.seq >= 0 && gs().shared.seq + 1 == rte::call_seq() && gs().shared.jeta == jeta &&
                        gs().shared.jtemp == jtemp && gs().shared.jpress == jpress && gs().shared.tropo == tropo &&
                        gs().shared.ncol == ncol && gs().shared.nlay == nlay && gs().shared.nflav == nflav &&
                        gs().shared.nbnd == nbnd && gs().shared.gw == bl_gw && !c.any_host() &&
                        !c.forked();  // (on the side stream this call does not wait for that call's kernels)
    gs().shared.seq = -1;
    TileGeom* d_geom = shared ? gs().shared.geom : (TileGeom*)rte::scratch(sizeof(TileGeom) * (size_t)tiles * nlay);
    static_assert(NCW * 64 == 512, "d_flags is sized for 512-column tiles");
#ifdef EXP_CLOCKS
    v.clocks = (unsigned long long*)rte::scratch(64);
    HIP_CHECK(hipMemsetAsync(v.clocks, 0, 64, st));
#endif
    const bool geom2 = g_geom_variant == 2 && nflav <= MAXFLAV && neta < 31 && ntemp < 31 && npres + 1 < 63;
    Geom2Args ga{};
    ga.ncol = ncol; ga.nlay = nlay; ga.nbnd = nbnd; ga.nflav = nflav; ga.slab_floats = SLAB9; ga.planck = true;
    ga.jeta = d_jeta; ga.jtemp = d_jtemp; ga.jpress = d_jpress; ga.tropo = d_tropo; ga.band_lims = d_band_lims;
    ga.gpoint_flavor = d_gpoint_flavor; ga.worklist = v.worklist; ga.flags = d_flags; ga.skip_if = guard;
    ga.skip_if2 = shared ? gs().shared.valid : nullptr;  // (set on the device by that call's geometry kernel, if it ran)
#define RTE_LAUNCH_PLANCK9(GW)                                                                                    \
  do {                                                                                                            \
    {                                                                                                             \
      rte::ProfScope p("planck_source_setup");                                                                    \
      if (shared) hipLaunchKernelGGL(planck_flags_kernel, dim3(cdiv(tiles * nbnd, 4)), dim3(256), 0, st, (const TileGeom*)d_geom, \
                                     (int)tiles, nlay, nbnd, SLAB9, GW + 2, d_flags, v.worklist, (const int*)gs().shared.valid, \
                                     (const int*)guard);                                                          \
      if (geom2) hipLaunchKernelGGL((tile_geom2_kernel<NCW * 64, GW>), dim3(tiles, nlay), dim3(NCW * 64), 0, st, ga, d_geom); \
      else hipLaunchKernelGGL((planck_geom_kernel<NCW * 64, GW>), dim3(tiles, nlay), dim3(NCW * 64), 0, st, v, nbnd, d_geom, \
                              d_flags, SLAB9);                                                                    \
    }                                                                                                             \
    rte::ProfScope p("planck_source_kernel");                                                                     \
    hipLaunchKernelGGL((planck_source_v9_kernel<NCW, NLW, SLAB9, GW>), dim3(nbnd * 8 * cdiv(tiles, 8)),          \
                       dim3((NCW + NLW) * 64), sizeof(Float) * nPlanckTemp, st, v, nbnd, tiles,                   \
                       (const TileGeom*)d_geom, (const int*)d_flags);                                             \
  } while (0)
    if (bl_gw == 16) RTE_LAUNCH_PLANCK9(16); else RTE_LAUNCH_PLANCK9(8);
#undef RTE_LAUNCH_PLANCK9
  } else {
    rte::ProfScope p("planck_source_kernel");
    hipLaunchKernelGGL((planck_source_v7_kernel<BS>), dim3(cdiv(ncol, BS), nbnd), dim3(BS), sizeof(Float) * nPlanckTemp, st,
                       v);
  }
#ifdef EXP_CLOCKS
  if (wl_tile != BS) {
    unsigned long long h[8];
    HIP_CHECK(hipMemcpyAsync(h, v.clocks, 64, hipMemcpyDeviceToHost, st));
    HIP_CHECK(hipStreamSynchronize(st));
    const double n = (double)cdiv(ncol, wl_tile) * nbnd * nlay;
    fprintf(stderr, "planck clocks/stage: top %.0f requests %.0f barrier %.0f compute+stores %.0f\n", h[0] / n, h[1] / n, h[2] / n, h[3] / n);
  }
#endif
  {
    // (tile, band) pairs whose pfrac bounding box exceeded the LDS slab at some layer
    rte::ProfScope p("planck_source_fallback");
    hipLaunchKernelGGL(planck_source_worklist_kernel, dim3(1024), dim3(256), 0, st, q, (const int*)v.worklist, wl_tile,
                       stats_dev() + 1);
    // the whole call on the direct kernel if the guard fired
    hipLaunchKernelGGL(planck_source_kernel, dim3(cdiv(ncol, 256), nbnd), dim3(256), 0, st, q, (const int*)guard);
  }
  RTE_CATCH("rrtmgp_compute_Planck_source")
}

}  // extern "C"
