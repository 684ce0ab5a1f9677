// gas_optics_common.h -- what the RRTMGP gas-optics translation units share (interpolation.hip, tau_absorption.hip,
// planck.hip, plans.hip): wave-level helpers, the plan guards, the g-point-fastest table re-layout, the band metadata and
// tile geometry of the production kernels with their bit-mask pre-pass, and the per-context host state (plan caches,
// geometry shared between consecutive calls, guard flag words; runtime.hip owns the contexts).
// Device code sits in an anonymous namespace: every translation unit gets its own copy of the small kernels it launches.
#pragma once
#include <math.h>

#include <atomic>
#include <type_traits>
#include <vector>

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

struct alignas(2 * sizeof(Float)) Float2 { Float x, y; };

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

// -------------------------------------------------------------------------------------------
// compute_tau_absorption / compute_Planck_source / compute_tau_rayleigh, production kernels (LDS slab): shared metadata.
//
// What the measurements on MI355X said (tools/membench.hip): kernels that gather LUT values
// straight from global memory with lanes = columns run at ~20 ms per 1e5 columns whatever the table layout, because
// each lane pulls its own cache line and the vector L1 retires about one distinct line per clock.  The production
// kernels therefore
//   * copy the tables to a g-point-fastest layout per call (relayout_gfast_kernel), so that the 16 (or 8) g-points of a
//     stage are one contiguous row piece;
//   * stage, per (column tile, layer, band), the BOUNDING BOX of the rows the tile's columns need -- pressure x
//     temperature x eta ranges for kmajor / pfrac, temperature x eta per minor interval -- into an LDS slab with an
//     odd row stride in 16-byte pieces (tile_geom2_kernel below derives the boxes);
//   * keep lanes = columns: every thread gathers its corner rows with 16-byte LDS reads (two g-points per read) and
//     writes its outputs with coalesced 512-byte wave stores.
// Tiles whose box does not fit the slab go to a worklist for the direct-gather kernels.  Arithmetic: the same products
// and sums as the reference (:791-801, :757-760) evaluated with fused multiply-adds and col_mix folded into the major
// weights; differences from the reference association are a few ulp (tests: 1e-12 relative).
// -------------------------------------------------------------------------------------------
constexpr int MAXM = 12;   // minor intervals per (band, regime) handled by the production kernels; more -> native kernel
constexpr int MAXB = 32;   // bands
// calls of a few thousand columns: the tau and Planck kernels split a (tile, layer) / (tile, band) pair's work over several
// blocks until about this many blocks are in the grid (two per CU)
#ifndef RTE_SMALL_GRID_BLOCKS
#define RTE_SMALL_GRID_BLOCKS 512u
#endif

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
  // Plain-ABI calls (no deferred zero fill): tau is intent(inout), but the frontend has just zeroed it
  // (mo_gas_optics_rrtmgp.F90:637,679).  tau_is_zero_kernel reads the array once (12 GB at the load ceiling: 2 ms) and leaves
  // "some element is not zero" in *nonzero; the OVERWRITE instance of the kernel then runs if it is 0, the accumulating one
  // (fp64 atomics in L2, 9.8 ms against 4.9) only if it is not.  run_when: 0 always, 1 if *nonzero == 0, 2 if *nonzero != 0.
  const int* nonzero;
  int run_when;
  bool atomic_ok;      // tau is device memory proper: hardware floating-point atomics are defined on it (not on host-visible memory)
  const Float* add_bybnd;  // (ncol, nlay, nbnd) or nullptr: see TauArgs
  RaylFuse rf;             // used by the RAYL instantiations only
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
constexpr int SLAB_FLOATS = 8704;  // values per slab buffer (68 KB in double precision; two buffers per block): boxes that need more go to the direct kernels

#ifndef V9_SB
#define V9_SB 8  // 16-byte pieces a loader lane of planck_source_v9_kernel requests back to back
#endif
struct TileGeom {   // one per (column tile, layer)
  int Tmin, nT, Pmin, nP, has_lo, has_up, pad0, pad1;
  int2 eg[MAXB];    // per band: (emin, nE); nE = 0 -> band handled by the direct kernel (or no work)
};

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
  int row_stride;            // tau: values between slab rows (0: G + 2; the DMA-staged slab pads rows by 16 bytes: G + 4 floats)
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
  if (a.stat && tid == 0 && blockIdx.x == 0 && blockIdx.y == 0) { *a.stat = pre ? 1 : 2; if (!a.planck) a.stat[1] = 9; }  // (rte_hip_stat(3): which tau kernel ran)
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
    const bool fits = rows * (a.row_stride > 0 ? a.row_stride : RS) <= a.slab_floats;
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

// "is any element of a[0 .. n) not zero" (NaNs count as not zero): sets *flag, never clears it
__global__ void __launch_bounds__(256) tau_is_zero_kernel(const Float* __restrict__ a, size_t n, int* __restrict__ flag) {
  typedef Float vec2 __attribute__((ext_vector_type(2)));
  const size_t n2 = n / 2;
  const vec2* a2 = reinterpret_cast<const vec2*>(a);  // (16-byte aligned: checked by the host)
  bool nz = false;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n2; i += (size_t)gridDim.x * 256) {
    const vec2 v = __builtin_nontemporal_load(a2 + i);
    nz = nz || !(v.x == (Float)0) || !(v.y == (Float)0);
  }
  if ((n & 1) && blockIdx.x == 0 && threadIdx.x == 0) nz = nz || !(a[n - 1] == (Float)0);
  if (__ballot(nz) != 0 && (threadIdx.x & 63) == 0) atomicOr(flag, 1);
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
}  // namespace

// process-wide tuning switches (set from any thread: relaxed atomics; defined in plans.hip)
extern std::atomic<int> g_tau_force_direct;
extern std::atomic<int> g_tau_no_zero_check;
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
extern std::atomic<int> g_share_geom_default;  // what a context starts with (the last rte_hip_share_geometry of any context)
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

