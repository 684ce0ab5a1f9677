// tau_slab.h -- rrtmgp_compute_tau_absorption, the production kernel ("slab kernel") for gfx950.
//
// Reference semantics: gas_optical_depths_major + gas_optical_depths_minor (x 2) of
// rrtmgp/kernels/mo_gas_optics_rrtmgp_kernels.F90:345-501 with interpolate3D_byflav / interpolate2D_byflav
// (:741-803); the fused forms add compute_tau_rayleigh (:506-565) and combine_abs_and_rayleigh
// (rrtmgp/frontend/mo_gas_optics_rrtmgp.F90:1954-2036).
//
// One block = 8 waves = 512 consecutive columns of ONE layer; lanes = columns, so every access to an (ncol, nlay, ...)
// array is a unit-stride wave request.  The g-points are walked in STAGES of G (16 or 8) g-points of one band.  Per stage
// the block needs the bounding box of k-table rows its columns touch (tile_geom2_kernel): pressure x temperature x eta
// rows of kmajor, one temperature x eta plane per minor interval of the band, each row = the stage's G g-points = one
// contiguous piece of the g-point-fastest table copies.  The box of stage s+1 is copied into the other half of a
// double-buffered LDS slab while stage s is gathered from this one, ONE barrier per stage:
//   * staging through registers by the compute waves themselves (no loader waves: rounds 1-4 had two, and ten waves
//     allow 168 registers per lane where eight allow 256): every thread requests its 16-byte pieces of slab s+1 right
//     behind the barrier of stage s and writes them to LDS a pass later.  Where each row comes from is planned two stages
//     ahead (plan_rows: one thread per row, a 32-bit count of 16-byte units from the first table), so the request is a
//     table look-up.  The slab keeps a padded row image (row stride = G values + 16 bytes, an odd number of 16-byte pieces:
//     the rows of a 16-lane group of a gather fall on different bank windows).
//   * gathers as ONE rolling pipeline through the stage: a step = 4 row reads of 16 bytes (two g-points of four corner
//     rows) + the FMAs on them; the major species are G steps, every minor interval G / 2; DEPTH steps are in flight all
//     the time, through the end of the major pass into the first minor interval and from one interval into the next.
//   * half of the waves keep a stage's sums in registers across the next barrier and store them then ("rotated"), so the
//     two halves of the block are never in the store phase together; the unrotated half plans the rows while it would
//     otherwise wait at the barrier.
//   * a tile whose columns are all in one regime at its layer walks the bands sorted by that regime's flavor, and a stage
//     of the previous stage's flavor keeps the flavor weights in registers: they are read once per flavor, not per band.
// Vector-memory operations of a wave retire in order, which fixes where things are requested: see the stage loop.
// (tile, layer, band) items whose box exceeds the slab are left to the direct-gather worklist kernel (tau_absorption.hip).
#pragma once
#include "gas_optics_common.h"

#ifndef TAU_DEPTH
#define TAU_DEPTH 2
#endif
namespace {

// 16 bytes of a table row
template <int FP> struct SlabPiece;
template <> struct SlabPiece<2> { using type = Float2; };
struct alignas(16) Float4 { Float x, y, z, w; };
template <> struct SlabPiece<4> { using type = Float4; };

typedef Float SlabVec __attribute__((ext_vector_type(16)));  // the pieces a thread has in flight for the next slab
__device__ __forceinline__ void slab_put(SlabVec& v, int u, const Float2& t) { v[2 * u] = t.x; v[2 * u + 1] = t.y; }
__device__ __forceinline__ void slab_put(SlabVec& v, int u, const Float4& t) { v[4 * u] = t.x; v[4 * u + 1] = t.y; v[4 * u + 2] = t.z; v[4 * u + 3] = t.w; }
__device__ __forceinline__ void slab_get(const SlabVec& v, int u, Float2& t) { t.x = v[2 * u]; t.y = v[2 * u + 1]; }
__device__ __forceinline__ void slab_get(const SlabVec& v, int u, Float4& t) { t.x = v[4 * u]; t.y = v[4 * u + 1]; t.z = v[4 * u + 2]; t.w = v[4 * u + 3]; }

struct SlabStage { int b /* band; bit 8: the lanes keep the previous stage's flavor weights */, emin, nE, rowsMaj, rowsLo, rowsUp, rowsAll, g0; };  // one stage of a (tile, layer): block-uniform
// what a lane of regime r requests for a stage and how it scales it: the band table's entries of the stage's first four minor
// intervals in the order they are used.  bits, per slot k at 4 k: 1 the interval covers the stage's g-points, 2 scales with
// density, 4 has a scaling gas, 8 by its complement
struct alignas(16) SlabPeek { int idx[4], isc[4], n, bits, flav[2]; };
constexpr int SLAB_MAXSTAGE = 32;  // ngpt / G the kernel handles (host-checked)


// OVERWRITE: tau is known to be zero on entry (deferred zero fill, or the zero test said so); otherwise the stage's sums
//   are added to the incoming values with fp64 atomics in L2.
// MM: minor intervals per (band, regime) whose column amounts are requested a stage ahead; a band with more runs the rest
//   in a tail pass that requests them where they are used (the real tables are ragged: 0 ... 9 per band and regime).
// ADDB: a band-wise operand is added before the store (rte_hip_compute_tau_absorption_inc_bybnd).
// RAYL: 1 fused with compute_tau_rayleigh + the 2-stream combine, 2 + by-band cloud increment, 3 as 1 without storing g.
template <int NCW, int SLAB, bool OVERWRITE, int G, int MM, bool ADDB, int RAYL, int DEPTH>
__global__ void __launch_bounds__(NCW * 64, (NCW + 3) / 4)
tau_slab_kernel(TauV5 a, const TileGeom* __restrict__ geom) {
  constexpr int TILE = NCW * 64;
  constexpr int FP = 16 / (int)sizeof(Float);          // values per 16-byte piece
  constexpr int PPR = G / FP, PPRP = PPR + 1;          // pieces per row, per padded row
  constexpr int RS = PPRP * FP;                        // row stride in values
  constexpr int NROW = SLAB / RS;                      // rows a slab buffer holds
  static_assert(PPRP == 9 || PPRP == 5 || PPRP == 3, "stages of 16 or 8 g-points, doubles or floats");
  static_assert(DEPTH == 2 || DEPTH == 4, "the step buffers rotate through G and G / 2 steps");
  static_assert((G / 2) % DEPTH == 0, "whole rotations per minor interval");
  __shared__ __align__(16) Float slab[2][SLAB];
  __shared__ unsigned s_rowoff[2][NROW];  // source of each row's first g-point: 16-byte units from a.kmaj (the g-fastest tables are one allocation)
  __shared__ __align__(16) SlabStage s_stage[SLAB_MAXSTAGE + 2];
  __shared__ SlabPeek s_peek[SLAB_MAXSTAGE + 1][2];
  static_assert(MM == 4, "SlabPeek holds four slots");
  __shared__ TileGeom tg;
  __shared__ int s_key[SLAB_MAXSTAGE], s_order[SLAB_MAXSTAGE + 2];
  extern __shared__ BandMeta bm[];  // [nbnd]
  if (*a.skip_if) return;
  if (a.run_when != 0 && (*a.nonzero != 0) != (a.run_when == 2)) return;  // (plain-ABI calls: see TauV5::nonzero)
  const int tid = threadIdx.x;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
  const unsigned ncol = a.ncol, nlay = a.nlay, ilay = blockIdx.y;
  const unsigned ncl = ncol * nlay;  // host guarantees < 2^29
  const int ntemp = a.ntemp, TE = a.TE, ngpt = a.ngpt, nbnd = a.nbnd;
  {
    const int* src = reinterpret_cast<const int*>(a.bmeta);
    int* dst = reinterpret_cast<int*>(bm);
    const int nw = nbnd * (int)(sizeof(BandMeta) / sizeof(int));
    for (int i = tid; i < nw; i += TILE) dst[i] = src[i];
    const int* gs = reinterpret_cast<const int*>(geom + (blockIdx.x + (size_t)gridDim.x * ilay));
    int* gd = reinterpret_cast<int*>(&tg);
    for (int i = tid; i < (int)(sizeof(TileGeom) / sizeof(int)); i += TILE) gd[i] = gs[i];
  }
  __syncthreads();
  const int Tmin = tg.Tmin, nT = tg.nT, Pmin = tg.Pmin, nP = tg.nP;
  const bool has_lo = tg.has_lo != 0, has_up = tg.has_up != 0;
  const int nstage_all = ngpt / G;  // host guarantees whole, G-aligned chunks per band and nstage_all <= SLAB_MAXSTAGE
  // ---- the order of the stages.  A tile whose columns are all in one regime at this layer walks the bands sorted by that
  // regime's flavor: what a lane requests per stage (fmajor, fminor, col_mix, jeta of the band's flavor) is then the previous
  // stage's for every band but the first of a flavor: the stage record says so (bit 8 of b) and the lanes keep their registers
  // (measured traffic was 1.23 x the algorithmic bytes with the bands in table order -- 16 reads of the weights per layer
  // instead of one per flavor; the kernel itself gains 1 %: it is not bound by these bytes).
  // s_order[position] = the chunk of G g-points done there.
  if (tid < nstage_all) {
    const int g0 = tid * G;
    int b = 0;
    while (b + 1 < nbnd && bm[b].gE < g0) ++b;
    s_key[tid] = (has_lo != has_up ? bm[b].flav[has_up ? 1 : 0] : 0) * SLAB_MAXSTAGE + tid;
  }
  if (tid >= nstage_all && tid < nstage_all + 2) s_order[tid] = tid;
  __syncthreads();
  if (tid < nstage_all) {
    const int key = s_key[tid];
    int pos = 0;
    for (int t = 0; t < nstage_all; ++t) pos += s_key[t] < key ? 1 : 0;
    s_order[pos] = tid;
  }
  __syncthreads();
  // calls of a few thousand columns (fewer (tile, layer) pairs than the chip holds blocks): gridDim.z blocks share a pair,
  // each takes a run of positions of that order; the rest of the kernel sees its own stages 0 ... nstage - 1
  const int sz0 = (int)((blockIdx.z * (unsigned)nstage_all) / gridDim.z);
  const int nstage = (int)(((blockIdx.z + 1) * (unsigned)nstage_all) / gridDim.z) - sz0;
  // ---- the block's schedule: what every stage stages (rows ordered: major [t][eta][p], then one [t][eta] plane per minor
  // interval of the lower, then of the upper regime, RAYL: then the two Rayleigh planes).  Entries nstage, nstage + 1: empty.
  if (tid < nstage + 2) {
    SlabStage si{};
    if (tid < nstage) {
      const int g0 = s_order[sz0 + tid] * G;
      int b = 0;
      while (b + 1 < nbnd && bm[b].gE < g0) ++b;
      si.b = b; si.g0 = g0; si.emin = tg.eg[b].x; si.nE = tg.eg[b].y;
      if (tid > 0) {  // the lanes' weights are the previous stage's: same flavor in every regime the tile has columns in
        const int g0p = s_order[sz0 + tid - 1] * G;
        int bp = 0;
        while (bp + 1 < nbnd && bm[bp].gE < g0p) ++bp;
        const bool same = (!has_lo || bm[bp].flav[0] == bm[b].flav[0]) && (!has_up || bm[bp].flav[1] == bm[b].flav[1]);
        si.b = b | (same ? 256 : 0);
      }
      if (si.nE > 0) {
        const int n_lo = has_lo ? bm[b].cnt[0] : 0, n_up = has_up ? bm[b].cnt[1] : 0;
        si.rowsMaj = nP * nT * si.nE; si.rowsLo = n_lo * nT * si.nE; si.rowsUp = n_up * nT * si.nE;
        si.rowsAll = si.rowsMaj + si.rowsLo + si.rowsUp + (RAYL ? 2 * nT * si.nE : 0);
      } else {
        si.nE = 0;  // (negative: the band's box did not fit, the worklist kernel does it)
      }
    } else {
      si.b = nbnd - 1; si.g0 = tid * G;
    }
    s_stage[tid] = si;
  }
  if (tid < 2 * (nstage + 1)) {
    const int st = tid >> 1, r = tid & 1, g0 = s_order[sz0 + st] * G;
    int b = 0;
    while (b + 1 < nbnd && bm[b].gE < g0) ++b;
    SlabPeek pk{};
    const int n = (st < nstage && tg.eg[b].y > 0) ? bm[b].cnt[r] : 0;
    pk.n = n;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const MinorMeta& m = bm[b].m[r][k];  // (slots past the band's count are zero-filled: never used)
      const bool sd = (m.flags & 1) != 0, hs = sd && m.idx_scaling > 0;
      pk.idx[k] = m.idx_minor;
      pk.isc[k] = hs ? m.idx_scaling : -1;
      const bool on = k < n && !(m.mE < g0 || m.mS > g0);  // intervals are whole G-aligned chunks inside the band
      pk.bits |= ((on ? 1 : 0) | (sd ? 2 : 0) | (hs ? 4 : 0) | ((m.flags & 2) ? 8 : 0)) << (4 * k);
    }
    pk.flav[0] = bm[b].flav[0]; pk.flav[1] = bm[b].flav[1];
    s_peek[st][r] = pk;
  }
  if (tid < 2) s_rowoff[tid][0] = 0;
  bool all_run = true;
  for (int b = 0; b < nbnd; ++b) all_run = all_run && tg.eg[b].y > 0;

  // ---- this lane's column
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

  // uni_reg: EVERY lane of the block is in the same regime at this layer (all lower or all upper, tropo flag in step): what a
  // lane looks up per stage -- the band table's record of its regime: gas indices, slot bits, flavors -- is then one value per
  // wave, kept in scalar registers, and the requests that depend on it are (scalar base) + (the lane's 32-bit offset).  The
  // block-wide AND doubles as the barrier behind the schedule.
  const int all_lo = __syncthreads_and(regime == 1 && itropo == 0);
  const int all_up = __syncthreads_and(regime == 2 && itropo == 1);
#if defined(TAU_NO_UNI)
  const bool uni_reg = false;
#else
  const bool uni_reg = (all_lo | all_up) != 0;
#endif
  auto get_stage = [&](int s) -> SlabStage {  // (wave-uniform: into scalar registers)
    const int4* p = reinterpret_cast<const int4*>(&s_stage[s]);
    const int4 u = p[0], v = p[1];
    SlabStage r;
    r.b = __builtin_amdgcn_readfirstlane(u.x) & 255; r.emin = __builtin_amdgcn_readfirstlane(u.y);
    r.nE = __builtin_amdgcn_readfirstlane(u.z); r.rowsMaj = __builtin_amdgcn_readfirstlane(u.w);
    r.rowsLo = __builtin_amdgcn_readfirstlane(v.x); r.rowsUp = __builtin_amdgcn_readfirstlane(v.y);
    r.rowsAll = __builtin_amdgcn_readfirstlane(v.z); r.g0 = __builtin_amdgcn_readfirstlane(v.w);
    return r;
  };

  // ---- staging.  plan_rows(st, first, step): the planner threads leave the source of row r of stage st in
  // s_rowoff[stage & 1]; called two stages ahead of the gathers, one barrier ahead of stage_load, which reads it.
  const float inv_nT = __builtin_amdgcn_rcpf((float)nT), inv_nP = __builtin_amdgcn_rcpf((float)nP);
  auto plan_rows = [&](const SlabStage& st, int s2, int first, int step) {
    if (st.rowsAll <= 0) return;
    // (v_rcp_f32 is good to 1 ulp: (r + 0.5) / n truncates to r / n exactly for r < 2^12)
    const float inv_nE = __builtin_amdgcn_rcpf((float)st.nE);
    const int nE = st.nE, emin = st.emin, g0 = st.g0;
    for (int r = first; r < st.rowsAll; r += step) {
      const Float* src;
      if (r < st.rowsMaj) {
        // pressure level innermost: the p and p + 1 rows of neighbouring columns then fall into different bank windows
        // (1.15 instead of 1.70 LDS cycles per 16-lane group access with [p][t][eta])
        const int rest = (int)(((float)r + 0.5f) * inv_nP), p_l = r - rest * nP;
        const int t_l = (int)(((float)rest + 0.5f) * inv_nE), e = rest - t_l * nE;
        src = a.kmaj + ((size_t)((Pmin - 1 + p_l) * TE + (emin - 1 + e) * ntemp + (Tmin - 1 + t_l)) * ngpt + g0);
      } else {
        const int rm = r - st.rowsMaj;
        if (RAYL && rm >= st.rowsLo + st.rowsUp) {
          const int rr = rm - st.rowsLo - st.rowsUp;
          const int rest = (int)(((float)rr + 0.5f) * inv_nE), e = rr - rest * nE;
          const int k = (int)(((float)rest + 0.5f) * inv_nT), t_l = rest - k * nT;  // k: regime
          src = a.rf.krayl_g[k] + ((size_t)((emin - 1 + e) * ntemp + (Tmin - 1 + t_l)) * ngpt + g0);
        } else {
          const bool up = rm >= st.rowsLo;
          const int rr = up ? rm - st.rowsLo : rm;
          const int rest = (int)(((float)rr + 0.5f) * inv_nE), e = rr - rest * nE;
          const int k = (int)(((float)rest + 0.5f) * inv_nT), t_l = rest - k * nT;
          const MinorMeta& m = bm[st.b].m[up ? 1 : 0][k];
          const bool on = m.mS <= g0 && m.mE >= g0;  // off: any valid address, the row is never used
          const Float* kg = up ? a.kup : a.klo;
          const unsigned nk = up ? a.nk_up : a.nk_lo;
          src = kg + ((size_t)((emin - 1 + e) * ntemp + (Tmin - 1 + t_l)) * nk + (unsigned)m.kstart + (on ? g0 - m.mS : 0));
        }
      }
      s_rowoff[s2 & 1][r] = (unsigned)((size_t)(src - a.kmaj) / FP);  // (rows start on 16-byte boundaries: host-checked)
    }
  };
  // stage_load / stage_write: the slab of stage s1 goes through registers.  Every thread requests its UP pieces of 16 bytes
  // (piece idx of the unpadded row image: row idx / PPR, 16 bytes idx % PPR of it) right behind the barrier that releases the
  // buffer, and writes them to the padded image a pass later, when they have arrived.  Always UP requests (pieces past the
  // image repeat its last one and are not written): the count of a wave's outstanding memory operations stays static, so
  // the compiler keeps counted waits.  (Round 5 first staged by LDS-DMA -- global_load_lds_dwordx4 from the compute waves --
  // and measured it: parity-clean and 1.3 ms slower, with every source on one hot kilobyte just the same: the DMA path lands
  // ~11 bytes per clock and CU and holds up the other returns meanwhile.  DESIGN 4.2.)
  constexpr int UP = (NROW * PPR + TILE - 1) / TILE;  // pieces per thread and stage
  static_assert(UP * FP <= 16, "the pieces in flight are held in one 16-element vector");
  using Piece = typename SlabPiece<FP>::type;
  // (one vector value, not an array of pieces: under this kernel's register pressure the compiler leaves an array in scratch)
  using PVec = SlabVec;
  const char* const kbase = reinterpret_cast<const char*>(a.kmaj);
  // UP0 pieces per thread cover the usual box (the benchmark atmosphere's average is 4.1 per thread at G = 16); the pieces of
  // a larger one are requested AND written in stage_rest, right away (block-uniform, rare; everything it requests it also
  // waits for, so the counts of outstanding operations behind it are those of the common path)
  constexpr int UP0 = (5 * UP + 7) / 8;
  auto stage_load = [&](int s1, int rowsAll, PVec& v) {
    const int nAll = rowsAll * PPR;
    const unsigned* tab = s_rowoff[s1 & 1];
#pragma unroll
    for (int u = 0; u < UP0; ++u) {
      const int idx = max(min(tid + u * TILE, nAll - 1), 0);
      // (no special case for an empty stage: its one "piece" is row 0 of the table in use, or of a table planned earlier
      //  -- s_rowoff[.][0] starts as 0 -- a valid address either way)
      const unsigned off = tab[idx / PPR] + (unsigned)(idx % PPR);
      const Piece t = *reinterpret_cast<const Piece*>(kbase + 16 * (size_t)off);
      slab_put(v, u, t);
    }
  };
  auto stage_rest = [&](int s1, int rowsAll) {
    const int nAll = rowsAll * PPR;
    if (nAll <= UP0 * TILE) return;  // (block-uniform)
    const unsigned* tab = s_rowoff[s1 & 1];
    Float* sl = slab[s1 & 1];
#pragma unroll
    for (int u = UP0; u < UP; ++u) {
      const int idx = min(tid + u * TILE, nAll - 1);
      const unsigned off = tab[idx / PPR] + (unsigned)(idx % PPR);
      const Piece t = *reinterpret_cast<const Piece*>(kbase + 16 * (size_t)off);
      *reinterpret_cast<Piece*>(sl + (idx / PPR) * RS + (idx % PPR) * FP) = t;  // (pieces past the image rewrite its last one)
    }
  };
  auto stage_write = [&](int s1, int rowsAll, const PVec& v) {
    const int nAll = rowsAll * PPR;
    Float* sl = slab[s1 & 1];
#pragma unroll
    for (int u = 0; u < UP0; ++u) {
      const int idx = tid + u * TILE;
      Piece t;
      slab_get(v, u, t);
      if (idx < nAll) *reinterpret_cast<Piece*>(sl + (idx / PPR) * RS + (idx % PPR) * FP) = t;
    }
  };

  // ---- requests of per-column inputs, a stage ahead
  struct Major { Float2 fm[4], cm; int2 je; };  // fmajor(2,2,2), col_mix(2), jeta(2) of one flavor
  auto load_major = [&](int flav, Major& x) {
    const size_t clf = cl + (size_t)ncl * flav;
    const Float2* fmp = reinterpret_cast<const Float2*>(a.fmajor + 8 * clf);
#pragma unroll
    for (int i = 0; i < 4; ++i) x.fm[i] = fmp[i];
    x.cm = *reinterpret_cast<const Float2*>(a.col_mix + 2 * clf);
    x.je = *reinterpret_cast<const int2*>(a.jeta + 2 * clf);
  };
  struct Minor { Float sc[MM], cgs[MM]; Float2 fn0, fn1; int2 em; Float addv; };
  // What a stage needs from the band table, read a stage ahead as ONE record (s_peek, three 16-byte reads): which gases and
  // flavors to request, and the slots' bits.  Looked up in the band table where they are used, every request waited for
  // its own LDS round trip.
  struct MinorIdx { int idx[MM], isc[MM], flav, flav_major, n, bits; };
  auto peek_minor = [&](auto uni_tag, int st, MinorIdx& q) {
    constexpr bool UNI = decltype(uni_tag)::value;  // (the record is the same for every lane: into scalar registers)
    auto U = [](int v) { return UNI ? __builtin_amdgcn_readfirstlane(v) : v; };
    const int4* p = reinterpret_cast<const int4*>(&s_peek[st][rsel]);
    const int4 u = p[0], v = p[1], w = p[2];
    q.idx[0] = U(u.x); q.idx[1] = U(u.y); q.idx[2] = U(u.z); q.idx[3] = U(u.w);
    q.isc[0] = U(v.x); q.isc[1] = U(v.y); q.isc[2] = U(v.z); q.isc[3] = U(v.w);
    q.n = U(regime > 0 ? w.x : 0);
    q.bits = U(regime > 0 ? w.y : (w.y & ~0x1111));
    q.flav = U(rsel ? w.w : w.z);           // minor absorbers use THEIR regime's flavor (:487)
    q.flav_major = U(itropo ? w.w : w.z);
    // looked up here, not where they are used
    if constexpr (UNI) {
#pragma unroll
      for (int k = 0; k < MM; ++k) asm volatile("" : "+s"(q.idx[k]), "+s"(q.isc[k]));
      asm volatile("" : "+s"(q.flav), "+s"(q.flav_major), "+s"(q.bits), "+s"(q.n));
    } else {
#pragma unroll
      for (int k = 0; k < MM; ++k) asm volatile("" : "+v"(q.idx[k]), "+v"(q.isc[k]));
      asm volatile("" : "+v"(q.flav), "+v"(q.flav_major), "+v"(q.bits), "+v"(q.n));
    }
  };
  // (requested only where the lane has the slot: always 2 MM requests -- a static count of outstanding operations, counted
  //  waits behind them -- was measured: 5.13 against 5.01 ms; every vector-memory instruction costs more than its wait)
  auto load_minor = [&](auto uni_tag, int b, const MinorIdx& q, Minor& x) {
    constexpr bool UNI = decltype(uni_tag)::value;
    // a (ncol, nlay) plane of a 3-D array: wave-uniform plane base + this column's 32-bit byte offset (8 ncol nlay < 2^32)
    auto plane_at = [&](const Float* base, size_t plane) {
      return *reinterpret_cast<const Float*>(reinterpret_cast<const char*>(base + (size_t)ncl * plane) + cl8);
    };
    x.addv = ADDB ? plane_at(a.add_bybnd, (size_t)b) : (Float)0;
#pragma unroll
    for (int k = 0; k < MM; ++k) {
      x.sc[k] = 0; x.cgs[k] = 0;
      if (k < q.n) {
        if constexpr (UNI) {
          x.sc[k] = plane_at(a.col_gas, (size_t)q.idx[k]);
          if (q.isc[k] >= 0) x.cgs[k] = plane_at(a.col_gas, (size_t)q.isc[k]);
        } else {
          x.sc[k] = a.col_gas[cl + (size_t)ncl * q.idx[k]];
          if (q.isc[k] >= 0) x.cgs[k] = a.col_gas[cl + (size_t)ncl * q.isc[k]];
        }
      }
    }
  };
  auto load_minor_w = [&](const MinorIdx& q, Minor& x) {
    const size_t clm = cl + (size_t)ncl * q.flav;
    const Float2* fnp = reinterpret_cast<const Float2*>(a.fminor + 4 * clm);
    x.fn0 = fnp[0]; x.fn1 = fnp[1];
    x.em = *reinterpret_cast<const int2*>(a.jeta + 2 * clm);
  };

  auto run_stages = [&](auto allrun_tag, auto rot_tag, auto rotate_tag, auto uni_tag) {
  constexpr bool ALLRUN = decltype(allrun_tag)::value;
  constexpr bool UNI = decltype(uni_tag)::value;  // one regime in the whole block: see uni_reg
  const int regime_u = UNI ? __builtin_amdgcn_readfirstlane(regime) : regime;
  // ROT: this wave issues a stage's tau stores AFTER the next stage's barrier, while the other half of the block gathers
  constexpr bool ROT = decltype(rot_tag)::value;
  constexpr bool ROTATE = decltype(rotate_tag)::value;  // the block has rotated waves (its upper half)
  static_assert(!ROT || RAYL == 0, "the fused variants finish a stage from its own slab");
  // the unrotated half (which would wait at the barrier for the rotated half, which stores first) plans the rows
  constexpr bool PLANNER = !ROT;
  constexpr int NPLAN = (ROTATE ? NCW / 2 : NCW) * 64;
  Float acc[G];
  bool have_prev = false;
  int g0_prev = 0;
  Float addv_prev = 0;
  // tau(:, :, g) = scalar plane base + this column's 32-bit byte offset (host guarantees 8 * ncol * nlay < 2^32)
  const size_t gstride = (size_t)ncl * sizeof(Float);
  auto flush = [&](int g0f, Float addvf) {  // the stage's G stores (RAYL == 0)
    char* const tplane = reinterpret_cast<char*>(a.tau + (size_t)ncl * g0f);
    unsigned toff = cl8;
    asm volatile("" : "+v"(toff));  // keep the 64-bit address out of the loop-invariant registers
    auto tau_at = [&](int j) { return reinterpret_cast<Float*>(tplane + gstride * j + toff); };
    if (ADDB) {  // by-band increment fused in (tau = tau_gas + tau_2 of the band)
#pragma unroll
      for (int j = 0; j < G; ++j) acc[j] = acc[j] + addvf;
    }
    if (OVERWRITE) {
      // lanes past the last column repeat it (ic is clamped): same values to the same addresses.  Unconditional
      // stores keep the count of outstanding memory operations static (counted waits instead of drains).
#if defined(TAUX_NOSTORE)
      Float t_ = 0;
#pragma unroll
      for (int j = 0; j < G; ++j) t_ += acc[j];
      if (t_ == (Float)1.2345e-300) store_stream(tau_at(0), t_);
#else
#pragma unroll
      for (int j = 0; j < G; ++j) store_stream(tau_at(j), acc[j]);
#endif
    } else if (valid) {
      // tau is inout (the reference accumulates onto it, :637,:679): the stage's sum is added to the incoming value as a
      // hardware floating-point atomic add performed in L2 (no return value) -- the same single addition tau_in + sum,
      // but the wave neither waits for tau_in nor holds it.  Every element is touched by exactly one thread per call.
      // (tau is device memory proper here: the host sends host-visible buffers to the direct kernels)
#pragma unroll
      for (int j = 0; j < G; ++j) unsafeAtomicAdd(tau_at(j), acc[j]);
    }
  };
  Major mj;
  Minor mn;
  MinorIdx nq;
  Minor mw;  // (only fn0, fn1, em are used: the next stage's)
  Float2 fn0{}, fn1{};
  int2 em{};
  bool fresh_cur = true;
  if constexpr (PLANNER) {  // the row plans of stages 0 and 1
    plan_rows(get_stage(0), 0, tid, NPLAN);
    plan_rows(get_stage(1), 1, tid, NPLAN);
  }
  __syncthreads();
  {
    PVec v0;
    const int rows0 = get_stage(0).rowsAll;
    stage_load(0, rows0, v0);
    stage_rest(0, rows0);
    stage_write(0, rows0, v0);
  }
  peek_minor(uni_tag, 0, nq);
  load_major(nq.flav_major, mj);
  load_minor(uni_tag, get_stage(0).b, nq, mn);
  load_minor_w(nq, mw);
  // Nothing outstanding when the loop is entered: the wait counts inside it are then those of the steady state
  __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0)
#pragma unroll 1
  for (int s = 0; s < nstage; ++s) {
    const SlabStage cur = get_stage(s);
    const int g0 = cur.g0, ibnd = cur.b, emin = cur.emin, nE = cur.nE;
    const bool run = nE > 0;  // block-uniform
    const bool fresh = fresh_cur;
    const int cq_bits = nq.bits, cq_n = nq.n, cq_flav_major = nq.flav_major;  // this stage's slots (peeked a stage ago)
    Float sc[MM], cgs[MM];
#pragma unroll
    for (int k = 0; k < MM; ++k) { sc[k] = mn.sc[k]; cgs[k] = mn.cgs[k]; }
    const Float addv = mn.addv;
    if (fresh) {  // the weights requested a stage ago: col_mix folded into fmajor where it landed
      fn0 = mw.fn0; fn1 = mw.fn1; em = mw.em;
      mj.fm[0].x = mj.cm.x * mj.fm[0].x; mj.fm[0].y = mj.cm.x * mj.fm[0].y; mj.fm[1].x = mj.cm.x * mj.fm[1].x; mj.fm[1].y = mj.cm.x * mj.fm[1].y;
      mj.fm[2].x = mj.cm.y * mj.fm[2].x; mj.fm[2].y = mj.cm.y * mj.fm[2].y; mj.fm[3].x = mj.cm.y * mj.fm[3].x; mj.fm[3].y = mj.cm.y * mj.fm[3].y;
    }
    const Float w0 = mj.fm[0].x, w1 = mj.fm[0].y, w2 = mj.fm[1].x, w3 = mj.fm[1].y, w4 = mj.fm[2].x, w5 = mj.fm[2].y, w6 = mj.fm[3].x, w7 = mj.fm[3].y;
    const int je1 = mj.je.x, je2 = mj.je.y;
    // slots the wave walks: up to the last one any of its lanes uses (wave-uniform; a lane without that slot adds 0 x a row it may read)
    int nslot = 0;
#pragma unroll
    for (int k = 0; k < MM; ++k) {
      if constexpr (UNI) { if ((cq_bits >> (4 * k)) & 1) nslot = k + 1; }
      else if (__builtin_amdgcn_ballot_w64(((cq_bits >> (4 * k)) & 1) != 0) != 0) nslot = k + 1;
    }
#if !defined(TAUX_NOBARRIER)
    __syncthreads();  // B(s): slab(s) is complete, and every wave is done with the other buffer
#endif
    const int bw_next = __builtin_amdgcn_readfirstlane(s_stage[s + 1].b), b_next = bw_next & 255;
    const bool fresh_next = (bw_next & 256) == 0;  // (block-uniform) the next stage's flavor weights are not this stage's
    const int rows_next = __builtin_amdgcn_readfirstlane(s_stage[s + 1].rowsAll);
    PVec pv;
#if defined(TAUX_NOSTAGE)
    pv = PVec{};
#else
    stage_load(s + 1, rows_next, pv);  // slab(s+1), for the buffer just released
    stage_rest(s + 1, rows_next);
#endif
    if constexpr (ROT) {
      if (ALLRUN ? s > 0 : have_prev) flush(g0_prev, addv_prev);
      have_prev = false;
    }
    // RAYL: what only the end of the stage needs -- the Rayleigh interpolation weights (fminor of the MAJOR species'
    // flavor, :548-551) and the band's cloud properties -- is requested here, at the top of its own stage
    Float2 fr0{}, fr1{};
    Float cld_t = 0, cld_s = 0, cld_g = 0;
    if (RAYL) {
      const Float2* frp = reinterpret_cast<const Float2*>(a.fminor + 4 * (cl + (size_t)ncl * cq_flav_major));
      fr0 = frp[0]; fr1 = frp[1];
    }
    if (RAYL == 2) {
      cld_t = a.rf.cld_tau[cl + (size_t)ncl * ibnd]; cld_s = a.rf.cld_ssa[cl + (size_t)ncl * ibnd];
      cld_g = a.rf.cld_g[cl + (size_t)ncl * ibnd];
    }
    if (!ALLRUN && !run) {
      peek_minor(uni_tag, s + 1, nq);
      if (fresh_next) {
        load_major(nq.flav_major, mj);
        load_minor_w(nq, mw);
      }
      load_minor(uni_tag, b_next, nq, mn);
      stage_write(s + 1, rows_next, pv);
      if constexpr (PLANNER) plan_rows(get_stage(s + 2), s + 2, tid, NPLAN);
      fresh_cur = fresh_next;
      continue;
    }
    const Float* sl = slab[s & 1];
    const Float* A0 = sl + (((jT - Tmin) * nE + (je1 - emin)) * nP + (jp - 1 - Pmin)) * RS;
    const Float* B0 = sl + (((jT + 1 - Tmin) * nE + (je2 - emin)) * nP + (jp - 1 - Pmin)) * RS;
    constexpr int sP = RS;   // to the row of the next pressure level (innermost)
    const int sE = nP * RS;  // to the row of the next eta
    const Float* A1 = A0 + sE;
    const Float* B1 = B0 + sE;
    const Float* M0 = sl + (cur.rowsMaj + (regime_u == 2 ? cur.rowsLo : 0)) * RS;
    const Float* r1_0 = M0 + ((jT - Tmin) * nE + (em.x - emin)) * RS;
    const Float* r2_0 = M0 + ((jT + 1 - Tmin) * nE + (em.y - emin)) * RS;
    const int plane = nT * nE * RS;
    const int act = cq_bits;
    // rows of slot q for this lane; a lane that does not use the slot reads its major rows instead (always inside the slab)
    auto slot_rows = [&](int q, const Float*& p1, const Float*& p2) {
      const bool on = ((act >> (4 * q)) & 1) != 0;
      p1 = on ? r1_0 + q * plane : A0;
      p2 = on ? r2_0 + q * plane : A0;
    };
    char* const tplane = reinterpret_cast<char*>(a.tau + (size_t)ncl * g0);
    unsigned toff = cl8;
    if (RAYL) asm volatile("" : "+v"(toff));  // keep the 64-bit address out of the loop-invariant registers
    auto tau_at = [&](int j) { return reinterpret_cast<Float*>(tplane + gstride * j + toff); };
#if defined(TAU_NO_FOLD)
#pragma unroll
    for (int j = 0; j < G; ++j) acc[j] = 0;
#endif
    // ================= ONE rolling pipeline of LDS row reads through the stage =================
    const Float f0 = fn0.x, f1 = fn0.y, f2 = fn1.x, f3 = fn1.y;
    Float2 kb[DEPTH][4];
#if defined(TAUX_NOGATHER)  // (attribution builds, results wrong: the stage without its LDS row reads)
    auto rd_major = [&](Float2 (&k)[4], int h) { k[0] = fn0; k[1] = fn1; k[2] = fn0; k[3] = fn1; };
    auto rd_minor = [&](Float2 (&k)[4], const Float* p1, const Float* p2, int j) { k[0] = fn0; k[1] = fn1; k[2] = fn0; k[3] = fn1; };
#else
    auto rd_major = [&](Float2 (&k)[4], int h) {  // h: (g-point pair, lower / upper temperature); :791-801
      const Float* b0 = ((h & 1) ? B0 : A0) + 2 * (h >> 1);
      const Float* b1 = ((h & 1) ? B1 : A1) + 2 * (h >> 1);
      k[0] = ld2(b0); k[1] = ld2(b1); k[2] = ld2(b0 + sP); k[3] = ld2(b1 + sP);
    };
    auto rd_minor = [&](Float2 (&k)[4], const Float* p1, const Float* p2, int j) {  // j: g-point pair; :757-760
      k[0] = ld2(p1 + 2 * j); k[1] = ld2(p1 + RS + 2 * j); k[2] = ld2(p2 + 2 * j); k[3] = ld2(p2 + RS + 2 * j);
    };
#endif
    const Float* c1;
    const Float* c2;
    slot_rows(0, c1, c2);
#pragma unroll
    for (int h = 0; h < DEPTH; ++h) rd_major(kb[h], h);
    {
      Float m = 0, n = 0;
#pragma unroll
      for (int h = 0; h < G; ++h) {
        Float2 (&k)[4] = kb[h % DEPTH];
        if ((h & 1) == 0) {
          m = w0 * k[0].x; n = w0 * k[0].y;
          m = fma(w1, k[1].x, m); n = fma(w1, k[1].y, n);
          m = fma(w2, k[2].x, m); n = fma(w2, k[2].y, n);
          m = fma(w3, k[3].x, m); n = fma(w3, k[3].y, n);
          asm volatile("" : "+v"(m), "+v"(n));
        } else {
          m = fma(w4, k[0].x, m); n = fma(w4, k[0].y, n);
          m = fma(w5, k[1].x, m); n = fma(w5, k[1].y, n);
          m = fma(w6, k[2].x, m); n = fma(w6, k[2].y, n);
          m = fma(w7, k[3].x, m); n = fma(w7, k[3].y, n);
          const int j = h & ~1;
#if defined(TAU_NO_FOLD)
          acc[j] = acc[j] + m;
          acc[j + 1] = acc[j + 1] + n;
#else
          // (the stage's sums START as the major species' term: 0 + m is m, but the compiler may not drop the addition)
          acc[j] = m;
          acc[j + 1] = n;
#endif
          // pin the accumulation here: otherwise the FMA chains are sunk below the whole loop and every read stays live
          asm volatile("" : "+v"(acc[j]), "+v"(acc[j + 1]));
        }
        // the next stage's record is read AHEAD of the minor rows in the LDS queue (reads return in order: behind them its
        // use -- the requests below -- would drain the pipeline)
        if (h == G - DEPTH - 1) peek_minor(uni_tag, s + 1, nq);
        if (h + DEPTH < G) rd_major(k, h + DEPTH);
        else if (nslot > 0) rd_minor(k, c1, c2, h + DEPTH - G);
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    // the minor intervals of this lane: scalings (:461-480), 0 for a slot that is not this lane's or not this stage's
    Float scl[MM];
#pragma unroll
    for (int k = 0; k < MM; ++k) {
      // (selects instead of branches: a factor the reference does not apply is an exact 1)
      const int bits = cq_bits >> (4 * k);
      const Float t_ = cgs[k] * vmr_fact * dry_fact;                          // :470-478
      const Float f_ = (bits & 8) ? (Float)1 - t_ : t_;
      Float v = sc[k] * ((bits & 2) ? dens : (Float)1);                       // :469
      v = v * (((bits & 6) == 6) ? f_ : (Float)1);
      scl[k] = (bits & 1) ? v : (Float)0;
    }
    // everything the next stage needs of this column: its registers are free now, and the requests are a minor pass
    // ahead of their use (requested at the end of the stage their latency is exposed at the barrier; behind the
    // stage's stores they arrive a store drain late)
    // the requests go out at a raised issue priority: a wave that has reached this point is served ahead of the block's other
    // waves, still in their major pass, so its requests have the whole minor pass to arrive (A/B in one process, three rounds:
    // 4.83-4.87 against 4.94-5.01 ms; the same around the slab requests behind the barrier or around the stage's stores: slower
    // or no different -- docs/lab-notebook.md, round 6).  -DTAU_NO_REQ_PRIO for A/B.
#if !defined(TAU_NO_REQ_PRIO)
    __builtin_amdgcn_s_setprio(2);
#endif
#if !defined(TAUX_NOWEIGHTS)
    if (fresh_next) {
      load_major(nq.flav_major, mj);
      load_minor_w(nq, mw);
    }
#if !defined(TAUX_NOAMOUNTS)
    load_minor(uni_tag, b_next, nq, mn);
#endif
#endif
    __builtin_amdgcn_sched_barrier(0);
#if !defined(TAU_NO_REQ_PRIO)
    __builtin_amdgcn_s_setprio(0);
#endif
#pragma unroll 1
    for (int q = 0; q < nslot; ++q) {
      Float scaling = scl[0];
#pragma unroll
      for (int u = 1; u < MM; ++u) scaling = (q == u) ? scl[u] : scaling;
      const bool more = q + 1 < nslot;  // (wave-uniform)
      const Float* n1;
      const Float* n2;
      slot_rows(q + 1, n1, n2);
#if !defined(TAU_NO_FOLD)
      // the interval's scaling folded into the four interpolation weights (4 products per interval and stage): 4 operations per
      // g-point instead of 5; the association differs from the reference's scaling x (interpolated k) by an ulp of the term
      const Float g0_ = scaling * f0, g1_ = scaling * f1, g2_ = scaling * f2, g3_ = scaling * f3;
#endif
#pragma unroll
      for (int j = 0; j < G / 2; ++j) {
        Float2 (&k)[4] = kb[j % DEPTH];
#if defined(TAU_NO_FOLD)
        Float s_ = f0 * k[0].x, t_ = f0 * k[0].y;
        s_ = fma(f1, k[1].x, s_); t_ = fma(f1, k[1].y, t_);
        s_ = fma(f2, k[2].x, s_); t_ = fma(f2, k[2].y, t_);
        s_ = fma(f3, k[3].x, s_); t_ = fma(f3, k[3].y, t_);
        acc[2 * j] = fma(scaling, s_, acc[2 * j]);  // :493
        acc[2 * j + 1] = fma(scaling, t_, acc[2 * j + 1]);
#else
        acc[2 * j] = fma(g0_, k[0].x, acc[2 * j]); acc[2 * j + 1] = fma(g0_, k[0].y, acc[2 * j + 1]);  // :493
        acc[2 * j] = fma(g1_, k[1].x, acc[2 * j]); acc[2 * j + 1] = fma(g1_, k[1].y, acc[2 * j + 1]);
        acc[2 * j] = fma(g2_, k[2].x, acc[2 * j]); acc[2 * j + 1] = fma(g2_, k[2].y, acc[2 * j + 1]);
        acc[2 * j] = fma(g3_, k[3].x, acc[2 * j]); acc[2 * j + 1] = fma(g3_, k[3].y, acc[2 * j + 1]);
#endif
        asm volatile("" : "+v"(acc[2 * j]), "+v"(acc[2 * j + 1]));
        if (j + DEPTH < G / 2) rd_minor(k, c1, c2, j + DEPTH);
        else if (more) rd_minor(k, n1, n2, j + DEPTH - G / 2);
        __builtin_amdgcn_sched_barrier(0);
      }
      c1 = n1; c2 = n2;
    }
    // the next slab's pieces, requested behind the barrier, have had the stage to arrive (written after the major pass --
    // 32 registers fewer through the minor pass -- the wave waits here for pieces that queue behind the previous stage's stores)
    stage_write(s + 1, rows_next, pv);
    if (cq_n > MM) {
      // the band's intervals beyond the MM held in registers: amounts requested here, same expressions (:461-480)
#pragma unroll 1
      for (int k = MM; k < cq_n; ++k) {
        const MinorMeta& mm = bm[ibnd].m[rsel][k];
        if (mm.mE < g0 || mm.mS > g0) continue;
        Float scaling = a.col_gas[cl + (size_t)ncl * mm.idx_minor];
        if (mm.flags & 1) {
          scaling = scaling * dens;  // :469
          if (mm.idx_scaling > 0) {  // :470-478
            const Float cg = a.col_gas[cl + (size_t)ncl * mm.idx_scaling];
            if (mm.flags & 2) scaling = scaling * ((Float)1 - cg * vmr_fact * dry_fact);
            else scaling = scaling * (cg * vmr_fact * dry_fact);
          }
        }
        const Float* p1 = r1_0 + k * plane;
        const Float* p2 = r2_0 + k * plane;
#pragma unroll
        for (int j = 0; j < G; j += 2) {
          const Float2 q0 = ld2(p1 + j), q1 = ld2(p1 + RS + j), q2 = ld2(p2 + j), q3 = ld2(p2 + RS + j);
          Float s_ = f0 * q0.x, t_ = f0 * q0.y;
          s_ = fma(f1, q1.x, s_); t_ = fma(f1, q1.y, t_);
          s_ = fma(f2, q2.x, s_); t_ = fma(f2, q2.y, t_);
          s_ = fma(f3, q3.x, s_); t_ = fma(f3, q3.y, t_);
          acc[j] = fma(scaling, s_, acc[j]);
          acc[j + 1] = fma(scaling, t_, acc[j + 1]);
          asm volatile("" : "+v"(acc[j]), "+v"(acc[j + 1]));
          if ((j & 6) == 6) __builtin_amdgcn_sched_barrier(0);
        }
      }
    }
    __builtin_amdgcn_sched_barrier(0);
    if constexpr (RAYL != 0) {
      // compute_tau_rayleigh (:548-555: interpolate2D with the reference's association) on the staged table rows,
      // combine_abs_and_rayleigh and the optional by-band increment on the values in registers (rayl_finish), and
      // the stage's 3 x G stores.  Rows [regime][t][eta] behind the minor planes; unconditional stores as above.
      const Float* R1 = sl + (cur.rowsMaj + cur.rowsLo + cur.rowsUp + ((itropo * nT + (jT - Tmin)) * nE + (je1 - emin))) * RS;
      const Float* R2 = sl + (cur.rowsMaj + cur.rowsLo + cur.rowsUp + ((itropo * nT + (jT + 1 - Tmin)) * nE + (je2 - emin))) * RS;
      char* const splane = reinterpret_cast<char*>(a.rf.ssa + (size_t)ncl * g0);
      char* const gplane = RAYL == 3 ? nullptr : reinterpret_cast<char*>(a.rf.g + (size_t)ncl * g0);
#pragma unroll
      for (int j = 0; j < G; j += 2) {
        const Float2 a0 = ld2(R1 + j), a1 = ld2(R1 + RS + j), b0 = ld2(R2 + j), b1 = ld2(R2 + RS + j);
        const Float ka = fr0.x * a0.x + fr0.y * a1.x + fr1.x * b0.x + fr1.y * b1.x;
        const Float kb_ = fr0.x * a0.y + fr0.y * a1.y + fr1.x * b0.y + fr1.y * b1.y;
#pragma unroll
        for (int u = 0; u < 2; ++u) {
          Float t_, s_, g_;
          rayl_finish(acc[j + u], (u == 0 ? ka : kb_) * wray, RAYL == 2, cld_t, cld_s, cld_g, t_, s_, g_);
          store_stream(tau_at(j + u), t_);
          store_stream(reinterpret_cast<Float*>(splane + gstride * (j + u) + toff), s_);
          if constexpr (RAYL != 3) store_stream(reinterpret_cast<Float*>(gplane + gstride * (j + u) + toff), g_);
        }
      }
    } else if constexpr (ROT) {
      have_prev = true; g0_prev = g0; addv_prev = addv;
    } else {
      flush(g0, addv);
    }
    // the sources of the rows of stage s + 2, read by stage_load behind the next barrier
    if constexpr (PLANNER) plan_rows(get_stage(s + 2), s + 2, tid, NPLAN);
    fresh_cur = fresh_next;
  }
  if constexpr (ROT) {
    if (ALLRUN ? nstage > 0 : have_prev) flush(g0_prev, addv_prev);
  }
  };
  // the fused variants end a stage with LDS reads of their own slab and cannot rotate
  constexpr bool ROTATE = RAYL == 0;
  // (the rare combination -- a tile with direct-gather items AND one regime -- runs the per-lane instance)
  using T_ = std::true_type;
  using F_ = std::false_type;
  if constexpr (ROTATE) {
    if (wv >= NCW / 2) {
      if (all_run && uni_reg) run_stages(T_{}, T_{}, T_{}, T_{});
      else if (all_run) run_stages(T_{}, T_{}, T_{}, F_{});
      else run_stages(F_{}, T_{}, T_{}, F_{});
      return;
    }
    if (all_run && uni_reg) run_stages(T_{}, F_{}, T_{}, T_{});
    else if (all_run) run_stages(T_{}, F_{}, T_{}, F_{});
    else run_stages(F_{}, F_{}, T_{}, F_{});
  } else {
    if (all_run && uni_reg) run_stages(T_{}, F_{}, F_{}, T_{});
    else if (all_run) run_stages(T_{}, F_{}, F_{}, F_{});
    else run_stages(F_{}, F_{}, F_{}, F_{});
  }
}

}  // namespace
