// tau_absorption.hip -- rrtmgp_compute_tau_absorption and rrtmgp_compute_tau_rayleigh (with their fused extension forms)
// for gfx950 (MI355X), hand-written HIP.  rrtmgp_interpolation: interpolation.hip, rrtmgp_compute_Planck_source: planck.hip.
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
#include "gas_optics_common.h"
#include "tau_mx.h"
#include "tau_slab.h"

// shape of the specialised-wave tau kernel (defaults: that of gas_optics_common.h; overridable for experiments)
#ifndef TAU_NCW
#define TAU_NCW V9_NCW
#define TAU_NLW V9_NLW
#endif
#ifndef TAU_SLAB
#define TAU_SLAB V9_SLAB
#endif
#ifndef TAU_DMA_WAVES  // experiments: which waves issue the DMA pieces (0: all, 1: the rotated half, 2: the other half) ...
#define TAU_DMA_WAVES 0
#endif
#ifndef TAU_DMA_LATE   // ... and when the waves that have just stored do (0: behind the barrier, 1: after the major pass)
#define TAU_DMA_LATE 0
#endif
#ifndef TAU_DEPTH
#define TAU_DEPTH 4  // steps (4 LDS row reads each) in flight per wave in the DMA form of tau_absorption_v9_kernel; 0: the rounds 3-4 gathers
#endif
#ifndef TAU_MINW
#define TAU_MINW ((TAU_NCW + TAU_NLW + 3) / 4)
#endif
static const bool g_worklist_native = getenv("RTE_WORKLIST_NATIVE") != nullptr;  // A/B: worklist entries from the native-layout tables

namespace {
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
  const int* run_if2 = nullptr;  // ... or *run_if2 != 0 (the matrix-core kernel also leaves irregular profiles to this one)
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
  if (a.run_if && *a.run_if == 0 && !(a.run_if2 && *a.run_if2 != 0)) return;
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
#pragma unroll
        for (int j = 0; j < HW; ++j) tp[(size_t)ncl * j] = acc[j];
      }
    }
  }
}


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

// MM = minor intervals per (band, regime) whose column amounts are kept in registers a stage ahead (4: what the register
// budget allows without spills -- an MM = 8 instantiation spilled 4 registers, 52 in the fused SW variant).  A band with
// more intervals (the real tables are ragged: 1 ... 9 per band and regime) runs its first MM this way and the rest in a
// tail pass whose column amounts are requested where they are used (their latency is exposed, for those bands only).
// ADDB: a band-wise operand is added (rte_hip_compute_tau_absorption_inc_bybnd) -- a template parameter, not a run-time
// test: a conditional load changes the number of outstanding memory operations from path to path, and the compiler
// then waits for (nearly) all of them, i.e. for the previous stage's stores, at the top of every stage.
#ifdef TAU_TIMING
// experiment builds only (tools/fastbuild.py taut:tau_absorption.hip=-DTAU_TIMING): s_memtime ticks of the waves of tau_absorption_v9_kernel
// per role and phase of a stage.  Compute waves: [0] waiting at the stage's barrier, [1] the previous stage's stores (ROT) + set-up,
// [2] major gather + FMAs, [3] minor species, rest of the stage.  Loader waves: [4] requesting + waiting for the table pieces,
// [5] writing them to LDS, [6] waiting at the barrier.
// (tau_clk: tau_slab.h)
#define TAU_T(k) do { const unsigned long long t_ = __builtin_amdgcn_s_memtime(); tacc[k] += t_ - tlast; tlast = t_; } while (0)
#else
#define TAU_T(k) do { } while (0)
#endif
// RAYL: fused with compute_tau_rayleigh + combine_abs_and_rayleigh (2-stream) [+ by-band 2-stream increment]: see RaylFuse
template <int NCW, int NLW, int SLAB, bool OVERWRITE, int G, int MM, bool ADDB, int RAYL = 0 /* 1: fused, 2: + by-band clouds, 3: fused, g (all zero) not stored */>
__global__ void __launch_bounds__((NCW + NLW) * 64, TAU_MINW)
tau_absorption_v9_kernel(TauV5 a, const TileGeom* __restrict__ geom) {
  constexpr int TILE = NCW * 64, NLT = NLW * 64, NT = TILE + NLT;
  constexpr int RS = G + 2, PPR = G / 2, PSH = G == 16 ? 3 : 2;  // row stride, 16-byte pieces per row, log2(PPR)
  // DMA (NLW == 0): no loader waves.  The compute waves stage slab s+1 themselves with LDS-DMA (global_load_lds_dwordx4:
  // 64 lanes x 16 bytes land LINEARLY in LDS at a wave-uniform base, the source address is per lane), issued right after
  // the barrier of stage s.  The padded row image (RS = G + 2 doubles = PPR + 1 pieces of 16 bytes) is kept -- lane l of
  // DMA instruction i carries padded piece 64 i + l, the pad piece re-fetches the row's last one -- so the gathers are
  // the loader-wave kernel's, immediate offsets and conflict pattern included.  A block is then 8 waves = two per SIMD:
  // 256 registers per lane instead of the 168 that ten waves allowed.
  constexpr bool DMA = NLW == 0;
  static_assert(!DMA || sizeof(Float) == 8, "the DMA staging moves 16-byte pieces of rows of doubles");
  constexpr int PPRP = PPR + 1;             // 16-byte pieces per padded row
  constexpr int NROW = SLAB / RS;           // rows a slab buffer holds
  __shared__ __align__(16) Float slab[2][SLAB];
  __shared__ unsigned s_rowoff[DMA ? 2 : 1][DMA ? NROW : 1];  // where each row's first g-point is: 16-byte units from a.kmaj (the g-fastest tables are one allocation)
  __shared__ TileGeom tg;
  extern __shared__ BandMeta bm[];  // [nbnd]
  if (*a.skip_if) return;
  if (a.run_when != 0 && (*a.nonzero != 0) != (a.run_when == 2)) return;  // (plain-ABI calls: see TauV5::nonzero)
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

  if constexpr (!DMA) if (tid >= TILE) {
    // ================================ loader waves ================================
    // the loaders issue little and mostly wait for memory: a raised issue priority lets their requests and LDS writes go
    // out ahead of the eight compute waves' FMAs, so that the next slab is complete a little earlier (tau 5.34 -> 5.28 ms
    // in one process, no change for Planck)
    __builtin_amdgcn_s_setprio(1);
    const int lt = tid - TILE;
    const float inv_nT = 1.0f / (float)nT, inv_nP = 1.0f / (float)nP;
    constexpr int SB = V9_SB;  // 16-byte pieces per lane requested back to back
    int ibnd = 0;
#ifdef TAU_TIMING
    unsigned long long tacc[8] = {0, 0, 0, 0, 0, 0, 0, 0}, tlast = __builtin_amdgcn_s_memtime();
#endif
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
#ifdef TAU_ROWS_PTE  // the rounds 1-3 order of the major rows, [p][t][eta] (A/B)
            const int rest = (int)(((float)r + 0.5f) * inv_nE), e = r - rest * nE;  // rows < 2^12: exact
            const int p_l = (int)(((float)rest + 0.5f) * inv_nT), t_l = rest - p_l * nT;
#else
            // major rows ordered [t][eta][p], the pressure level innermost: the rows of a 16-lane group of a gather then fall into
            // different 4-bank windows far more often (tools/lds_conflict_sim.py: 1.15 instead of 1.70 LDS cycles per group access;
            // with [p][t][eta] the p and p + 1 rows of neighbouring columns were often 16 rows apart = the same window)
            const int rest = (int)(((float)r + 0.5f) * inv_nP), p_l = r - rest * nP;  // rows < 2^12: exact
            const int t_l = (int)(((float)rest + 0.5f) * inv_nE), e = rest - t_l * nE;
#endif
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
#pragma unroll 1
        for (int base = lt; base < nAll; base += SB * NLT) {
          Float2 v[SB];
#pragma unroll
          for (int u = 0; u < SB; ++u) v[u] = piece(min(base + u * NLT, nAll - 1));
#ifdef TAU_TIMING
          __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0): the pieces have arrived
          TAU_T(4);
#endif
#pragma unroll
          for (int u = 0; u < SB; ++u) {
            const int idx = base + u * NLT;
            if (idx < nAll) *reinterpret_cast<Float2*>(sl + (idx >> PSH) * RS + 2 * (idx & (PPR - 1))) = v[u];
          }
          TAU_T(5);
        }
      }
      __syncthreads();  // B(s): slab(s) complete; the compute waves are done with the other buffer
      TAU_T(6);
    }
#ifdef TAU_TIMING
    if ((tid & 63) == 0)
      for (int k = 4; k < 7; ++k) atomicAdd(&tau_clk[k], tacc[k]);
#endif
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
  constexpr bool PARK = RAYL != 0 && NLW != 0;  // (the variants that would otherwise spill; with two waves per SIMD -- no loader waves -- none does)
  constexpr int NPARK = PARK ? 3 : 0;
  __shared__ Float s_park[NPARK ? NPARK : 1][NPARK ? TILE : 1];
  // ... and the two row indices (temperature, pressure), packed into one int: the compiler kept address terms derived from
  // them in scratch and reloaded those at the top of every stage (2 KB; the block's LDS is within 1.5 KB of the limit)
  __shared__ int s_parki[1][PARK ? TILE : 1];
  unsigned park_at = 0;  // LDS byte address of this thread's first slot (the low half of the generic address)
  unsigned parki_at = 0;
  if constexpr (PARK) {
    s_park[0][tid] = dens; s_park[1][tid] = vmr_fact; s_park[2][tid] = dry_fact;
    s_parki[0][tid] = jT | (jp << 16);  // (both below 2^15: table dimensions)
    park_at = (unsigned)(uintptr_t)&s_park[0][tid];
    parki_at = (unsigned)(uintptr_t)&s_parki[0][tid];
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
  auto parked_i = [](unsigned at, int i) -> int {  // i = 0: temperature index, 1: pressure index
    int v;
    asm volatile("ds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(at) : "memory");
    return i == 0 ? (v & 0xFFFF) : (v >> 16);
  };
#define RTE_PARKED_I(i, in_register) (PARK ? parked_i(parki_at, i) : (in_register))

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
  // ---- DMA staging (NLW == 0).  Rows of a stage's slab, as the loader waves ordered them: major [t][eta][p], then one
  // [t][eta] plane per minor interval of the lower, then of the upper regime (+ RAYL: the two Rayleigh planes).
  // plan_rows(s2, b2): thread r leaves the source of row r of stage s2 (band b2) in s_rowoff[s2 & 1] -- called two
  // stages ahead of the gathers, one barrier ahead of issue_dma(s2), which reads it.
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
  auto stage_rows = [&](int b, int& rowsMaj, int& rowsLo, int& rowsUp) -> int {  // block-uniform
    const int nE = tg.eg[b].y;
    if (nE <= 0) { rowsMaj = rowsLo = rowsUp = 0; return 0; }
    const int n_lo = has_lo ? bm[b].cnt[0] : 0, n_up = has_up ? bm[b].cnt[1] : 0;
    rowsMaj = nP * nT * nE; rowsLo = n_lo * nT * nE; rowsUp = n_up * nT * nE;
    return rowsMaj + rowsLo + rowsUp + (RAYL ? 2 * nT * nE : 0);
  };
  auto plan_rows = [&](int s2, int b2) {
    if constexpr (DMA) {
      if (s2 >= nstage) return;
      int rowsMaj, rowsLo, rowsUp;
      const int rowsAll = stage_rows(b2, rowsMaj, rowsLo, rowsUp);
      const int g0 = s2 * G;
      const int emin = tg.eg[b2].x, nE = tg.eg[b2].y;
      // (v_rcp_f32 is good to 1 ulp: (r + 0.5) / n truncates to r / n exactly for r < 2^12)
      const float inv_nT = __builtin_amdgcn_rcpf((float)nT), inv_nP = __builtin_amdgcn_rcpf((float)nP),
                  inv_nE = __builtin_amdgcn_rcpf((float)(nE > 0 ? nE : 1));
      for (int r = tid; r < rowsAll; r += TILE) {
        const Float* src;
        if (r < rowsMaj) {
#ifdef TAU_ROWS_PTE
          const int rest = (int)(((float)r + 0.5f) * inv_nE), e = r - rest * nE;  // rows < 2^12: exact
          const int p_l = (int)(((float)rest + 0.5f) * inv_nT), t_l = rest - p_l * nT;
#else
          const int rest = (int)(((float)r + 0.5f) * inv_nP), p_l = r - rest * nP;
          const int t_l = (int)(((float)rest + 0.5f) * inv_nE), e = rest - t_l * nE;
#endif
          src = a.kmaj + ((size_t)((Pmin - 1 + p_l) * TE + (emin - 1 + e) * ntemp + (Tmin - 1 + t_l)) * ngpt + g0);
        } else {
          const int rm = r - rowsMaj;
          if (RAYL && rm >= rowsLo + rowsUp) {
            const int rr = rm - rowsLo - rowsUp;
            const int rest = (int)(((float)rr + 0.5f) * inv_nE), e = rr - rest * nE;
            const int k = (int)(((float)rest + 0.5f) * inv_nT), t_l = rest - k * nT;  // k: regime
            src = a.rf.krayl_g[k] + ((size_t)((emin - 1 + e) * ntemp + (Tmin - 1 + t_l)) * ngpt + g0);
          } else {
            const bool up = rm >= rowsLo;
            const int rr = up ? rm - rowsLo : rm;
            const int rest = (int)(((float)rr + 0.5f) * inv_nE), e = rr - rest * nE;
            const int k = (int)(((float)rest + 0.5f) * inv_nT), t_l = rest - k * nT;
            const MinorMeta& m = bm[b2].m[up ? 1 : 0][k];
            const bool on = m.mS <= g0 && m.mE >= g0;  // off: any valid address, the row is never read
            const Float* kg = up ? a.kup : a.klo;
            const unsigned nk = up ? a.nk_up : a.nk_lo;
            src = kg + ((size_t)((emin - 1 + e) * ntemp + (Tmin - 1 + t_l)) * nk + (unsigned)m.kstart + (on ? g0 - m.mS : 0));
          }
        }
        s_rowoff[s2 & 1][r] = (unsigned)((size_t)(src - a.kmaj) >> 1);  // (rows start on 16-byte boundaries: even g0, kstart, nk)
      }
    }
  };
  // issue_dma(s1, b1): this wave's share of the DMA instructions that fill slab[s1 & 1]: instruction i carries the padded
  // pieces 64 i ... 64 i + 63 to LDS bytes 1024 i ... of the buffer.  Hidden from the compiler (inline assembly: it neither
  // counts them in its s_waitcnt bookkeeping nor orders LDS reads behind them); their completion is waited for explicitly
  // before the barrier that opens stage s1 (dma_wait).
  const unsigned slab_lds = (unsigned)(uintptr_t)&slab[0][0];
  auto issue_dma = [&](int s1, int b1) {
    if constexpr (DMA) {
      if (s1 >= nstage) return;
      int rowsMaj, rowsLo, rowsUp;
      const int rowsAll = stage_rows(b1, rowsMaj, rowsLo, rowsUp);
      const int nD = (rowsAll * PPRP + 63) >> 6;  // (0 when the band does not run here)
      const unsigned* tab = s_rowoff[s1 & 1];
      const unsigned dst0 = slab_lds + (unsigned)((s1 & 1) * SLAB * sizeof(Float));
      constexpr int MAGIC = PPRP == 9 ? 7282 : 13108;  // (P * MAGIC) >> 16 == P / PPRP for P < 2^13
      static_assert(PPRP == 9 || PPRP == 5, "stage widths of 16 and 8 g-points");
      int first = wv, stride = NCW;
      if constexpr (TAU_DMA_WAVES == 1) { if (wv < NCW / 2) return; first = wv - NCW / 2; stride = NCW / 2; }  // the rotated waves only
      if constexpr (TAU_DMA_WAVES == 2) { if (wv >= NCW / 2) return; stride = NCW / 2; }                       // the others only
#pragma unroll 1
      for (int i = first; i < nD; i += stride) {
        const int P = 64 * i + lane;
        int r = (P * MAGIC) >> 16;
        int q = P - r * PPRP;
        r = min(r, rowsAll - 1); q = min(q, PPR - 1);
        const char* src = reinterpret_cast<const char*>(a.kmaj) + 16 * ((size_t)tab[r] + (unsigned)q);
        const unsigned dst = (unsigned)__builtin_amdgcn_readfirstlane((int)(dst0 + 1024u * (unsigned)i));
        unsigned keep;
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                     : "=&s"(keep) : "v"(src), "s"(dst) : "memory");
      }
    }
  };
  auto run_stages = [&](auto allrun_tag, auto rot_tag) {
  constexpr bool ALLRUN = decltype(allrun_tag)::value;
  // ROT: this wave issues a stage's tau stores AFTER the next stage's barrier (while the other half of the compute
  // waves gathers from the LDS) instead of at the end of the stage (when every wave of the block stores)
  constexpr bool ROT = decltype(rot_tag)::value;
  // ROLL: the LDS gathers as a rolling pipeline -- the next four rows are requested BEFORE the FMAs on the four that have
  // arrived, at most 8 row reads in flight per wave (the batched form: 16 reads, wait, 32 FMAs, with the LDS idle during
  // the FMAs and the SIMD idle during the reads).  5.19 -> 5.05 ms at 1e5 x 60 x 256 (-DTAU_BATCHED_GATHER for the A/B).
  // The fused variants are at the register limit and keep the batched form.
#ifdef TAU_BATCHED_GATHER
  constexpr bool ROLL = false;
#else
  constexpr bool ROLL = RAYL == 0 && !ADDB;
#endif
  static_assert(!ROT || RAYL == 0, "the fused variants finish a stage from its own slab");
  constexpr bool DEEP = DMA && TAU_DEPTH > 0;  // one rolling read pipeline through the stage (below); needs the registers of the 8-wave block
  Float acc[G];
  bool have_prev = false;
  int g0_prev = 0;
  Float addv_prev = 0;
  // tau(:, :, g) = scalar plane base + this column's 32-bit byte offset (host guarantees 8*ncol*nlay < 2^32)
  const size_t gstride = (size_t)ncl * sizeof(Float);
  // the stage's G stores (RAYL == 0)
  auto flush = [&](int g0f, Float addvf) {
    char* const tplane = reinterpret_cast<char*>(a.tau + (size_t)ncl * g0f);
    unsigned toff = cl8;
    asm volatile("" : "+v"(toff));  // keep the 64-bit address out of the loop-invariant registers
    auto tau_at = [&](int j) { return reinterpret_cast<Float*>(tplane + gstride * j + toff); };
    if (OVERWRITE) {
      if (ADDB) {  // by-band increment fused in (tau = tau_gas + tau_2 of the band)
#pragma unroll
        for (int j = 0; j < G; ++j) acc[j] = acc[j] + addvf;
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
        for (int j = 0; j < G; ++j) acc[j] = acc[j] + addvf;
      }
      // (tau is device memory proper here: the host sends host-visible buffers to the direct kernels)
#pragma unroll
      for (int j = 0; j < G; ++j) unsafeAtomicAdd(tau_at(j), acc[j]);
    }
  };
  Major mj;
  Minor mn;
  MinorIdx nq;
  Minor mw;  // (only fn0, fn1, em are used: the next stage's)
  if constexpr (DMA) {  // slab 0 on its way, the row addresses of stage 1 in place
    int b1 = 0;
    if (nstage > 1) while (b1 + 1 < nbnd && bm[b1].gE < G) ++b1;
    plan_rows(0, 0);
    plan_rows(1, b1);
    __syncthreads();
    issue_dma(0, 0);
  }
  // what of this wave's memory operations may still be outstanding once its DMA pieces of the next slab have landed: the
  // stage's stores, which are the last thing a wave that stores at the end of the stage issues (everything it requests for
  // the next stage goes out before them); a rotated wave has nothing younger than its requests.  Vector memory operations
  // of a wave retire in order, so vmcnt(that many) means "my pieces are in LDS".
  constexpr int NST = RAYL == 0 ? G : (RAYL == 3 ? 2 * G : 3 * G);
  constexpr int DMA_LEAVE = (ALLRUN && OVERWRITE && !ROT) ? NST : 0;
  static_assert(DMA_LEAVE < 64, "vmcnt has six bits");
  peek_minor(0, n_minor(0), nq);
  load_major(nq.flav_major, mj);
  load_minor(0, nq, mn);
  load_minor_w(nq, mw);
  // Nothing outstanding when the loop is entered: the wait counts inside it are then those of the steady state
  // (requests of stage s+1, then the stores of stage s) and not the merge with this prologue, which made every stage
  // wait for all but three of the previous stage's stores.
  __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0)
  int ibnd = 0;
#ifdef TAU_TIMING
  unsigned long long tacc[8] = {0, 0, 0, 0, 0, 0, 0, 0}, tlast = __builtin_amdgcn_s_memtime();
#endif
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
    TAU_T(DMA ? 6 : 3);
    if constexpr (DMA) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(DMA_LEAVE) : "memory");  // this wave's pieces of slab(s) have landed
    TAU_T(4);
    __syncthreads();  // B(s): slab(s) is complete
    TAU_T(0);
    if constexpr (DMA) {
      // slab(s+1) into the buffer every wave has just finished reading; the row addresses of stage s+2 for the next round
      if constexpr (!(TAU_DMA_LATE && !ROT)) issue_dma(s + 1, ibnd_n);
      int ibnd_nn = ibnd_n;
      if (s + 2 < nstage) while (ibnd_nn + 1 < nbnd && bm[ibnd_nn].gE < g0 + 2 * G) ++ibnd_nn;
      plan_rows(s + 2, ibnd_nn);
      TAU_T(5);
    }
    if constexpr (ROT) {
      if (ALLRUN ? s > 0 : have_prev) flush(g0_prev, addv_prev);
      have_prev = false;
    }
    peek_minor(ibnd_n, n_minor(ibnd_n), nq);
    if (!ALLRUN && !run) {
      if constexpr (DMA && TAU_DMA_LATE && !ROT) issue_dma(s + 1, ibnd_n);
      load_major(nq.flav_major, mj);
      load_minor_w(nq, mw);
      load_minor(ibnd_n, nq, mn);
      continue;
    }
    const Float* sl = slab[s & 1];
    const int rowsMaj = nP * nT * nE;
    const int rowsLo = (has_lo ? bm[ibnd].cnt[0] : 0) * nT * nE;
    const int jT_s = RTE_PARKED_I(0, jT), jp_s = RTE_PARKED_I(1, jp);
#ifdef TAU_ROWS_PTE
    const Float* A0 = sl + (((jp_s - 1 - Pmin) * nT + (jT_s - Tmin)) * nE + (je1 - emin)) * RS;
    const Float* B0 = sl + (((jp_s - 1 - Pmin) * nT + (jT_s + 1 - Tmin)) * nE + (je2 - emin)) * RS;
    const int sP = nT * nE * RS;  // to the row of the next pressure level
    constexpr int sE = RS;        // to the row of the next eta
#else
    const Float* A0 = sl + (((jT_s - Tmin) * nE + (je1 - emin)) * nP + (jp_s - 1 - Pmin)) * RS;
    const Float* B0 = sl + (((jT_s + 1 - Tmin) * nE + (je2 - emin)) * nP + (jp_s - 1 - Pmin)) * RS;
    constexpr int sP = RS;   // to the row of the next pressure level (innermost, see the loader)
    const int sE = nP * RS;  // to the row of the next eta
#endif
    const Float* M0 = sl + (rowsMaj + (regime == 2 ? rowsLo : 0)) * RS;
    // (RAYL: the stage's stores are issued here, see below)
    char* const tplane = reinterpret_cast<char*>(a.tau + (size_t)ncl * g0);
    unsigned toff = cl8;
    if (RAYL) asm volatile("" : "+v"(toff));  // keep the 64-bit address out of the loop-invariant registers
    auto tau_at = [&](int j) { return reinterpret_cast<Float*>(tplane + gstride * j + toff); };
#pragma unroll
    for (int j = 0; j < G; ++j) acc[j] = 0;
    TAU_T(1);
    if constexpr (!DEEP) {
    if constexpr (ROLL) {
      // rolling: the next four rows are requested BEFORE the FMAs on the four that arrived (at most 8 row reads in flight
      // per wave; the LDS serves the other waves' and this wave's next rows while the SIMD works on these)
      Float2 kb[2][4];
      auto rd = [&](Float2 (&k)[4], int h) {  // h: half-step index, (pair, lower / upper temperature)
        const Float* base = ((h & 1) ? B0 : A0) + 2 * (h >> 1);
        k[0] = ld2(base); k[1] = ld2(base + sE); k[2] = ld2(base + sP); k[3] = ld2(base + sP + sE);
      };
      rd(kb[0], 0);
      Float m = 0, n = 0;
#pragma unroll
      for (int h = 0; h < G; ++h) {
        Float2 (&k)[4] = kb[h & 1];
        if (h + 1 < G) rd(kb[(h & 1) ^ 1], h + 1);
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
          acc[j] = acc[j] + m;
          acc[j + 1] = acc[j + 1] + n;
          asm volatile("" : "+v"(acc[j]), "+v"(acc[j + 1]));
        }
        __builtin_amdgcn_sched_barrier(0);
      }
    } else {
#pragma unroll
    for (int j = 0; j < G; j += 2) {
      // :791-801 with col_mix folded into the weights; one 16-byte LDS read feeds two g-points
      const Float2 k0 = ld2(A0 + j), k1 = ld2(A0 + sE + j), k2 = ld2(A0 + sP + j), k3 = ld2(A0 + sP + sE + j),
                   k4 = ld2(B0 + j), k5 = ld2(B0 + sE + j), k6 = ld2(B0 + sP + j), k7 = ld2(B0 + sP + sE + j);
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
    }
    TAU_T(2);
    // next stage's major weights: their registers are free now, and the request is a minor pass ahead of its use
    // (requested with the minor weights at the end of the stage, their latency is exposed: 5.5 -> 5.9 ms)
    load_major(nq.flav_major, mj);
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
    const Float f0 = fn0.x, f1 = fn0.y, f2 = fn1.x, f3 = fn1.y;
    // one minor interval's contribution (:757-760, :493): 4 corner rows of its plane, 2 g-points per LDS read
    auto minor_rows = [&](int k, Float scaling) {
      const int jT_m = RTE_PARKED_I(0, jT);
      const Float* r1 = M0 + ((k * nT + (jT_m - Tmin)) * nE + (em.x - emin)) * RS;
      const Float* r2 = M0 + ((k * nT + (jT_m + 1 - Tmin)) * nE + (em.y - emin)) * RS;
      if constexpr (ROLL) {
      Float2 qb[2][4];  // one g-point pair (4 row reads) per step, the next step's requested ahead
      auto rdm = [&](Float2 (&q)[4], int j) { q[0] = ld2(r1 + j); q[1] = ld2(r1 + RS + j); q[2] = ld2(r2 + j); q[3] = ld2(r2 + RS + j); };
      rdm(qb[0], 0);
#pragma unroll
      for (int j = 0; j < G; j += 2) {
        Float2 (&q)[4] = qb[(j >> 1) & 1];
        if (j + 2 < G) rdm(qb[((j >> 1) & 1) ^ 1], j + 2);
        Float s_ = f0 * q[0].x, t_ = f0 * q[0].y;
        s_ = fma(f1, q[1].x, s_); t_ = fma(f1, q[1].y, t_);
        s_ = fma(f2, q[2].x, s_); t_ = fma(f2, q[2].y, t_);
        s_ = fma(f3, q[3].x, s_); t_ = fma(f3, q[3].y, t_);
        acc[j] = fma(scaling, s_, acc[j]);
        acc[j + 1] = fma(scaling, t_, acc[j + 1]);
        asm volatile("" : "+v"(acc[j]), "+v"(acc[j + 1]));
        __builtin_amdgcn_sched_barrier(0);
      }
      } else {
#pragma unroll
      for (int j = 0; j < G; j += 2) {
        const Float2 q0 = ld2(r1 + j), q1 = ld2(r1 + RS + j), q2 = ld2(r2 + j), q3 = ld2(r2 + RS + j);
        Float s_ = f0 * q0.x, t_ = f0 * q0.y;
        s_ = fma(f1, q1.x, s_); t_ = fma(f1, q1.y, t_);
        s_ = fma(f2, q2.x, s_); t_ = fma(f2, q2.y, t_);
        s_ = fma(f3, q3.x, s_); t_ = fma(f3, q3.y, t_);
        acc[j] = fma(scaling, s_, acc[j]);
        acc[j + 1] = fma(scaling, t_, acc[j + 1]);
        asm volatile("" : "+v"(acc[j]), "+v"(acc[j + 1]));
        if ((j & 6) == 6) __builtin_amdgcn_sched_barrier(0);  // at most 16 row reads in flight
      }
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
    } else {
      // ================= DEEP: ONE rolling pipeline of LDS row reads through the whole stage =================
      // A step = 4 row reads (16 bytes each: two g-points of four corner rows) + the FMAs on them.  The major species are 2 G / 2
      // steps (lower / upper temperature of each g-point pair), every minor interval G / 2.  DEPTH steps are in flight all the
      // time: a step's registers are refilled with the step DEPTH ahead as soon as its FMAs are issued -- through the end of
      // the major pass into the first minor interval and from one interval into the next (the rolling form of rounds 3-4 kept 8
      // reads in flight, restarted at every interval, and had 168 registers).  Same operations per g-point in the
      // same order as before: bit-identical.
      constexpr int DEPTH = TAU_DEPTH;
      static_assert(DEPTH == 2 || DEPTH == 4 || DEPTH == 8, "the buffers rotate through G and G / 2 steps");
      static_assert(G % DEPTH == 0 && (G / 2) % DEPTH == 0 || DEPTH > G / 2, "whole rotations");
      // ---- the minor intervals of this lane: scalings (:461-480), 0 for a slot that is not this lane's or not this stage's
      const int n_reg = n_my < MM ? n_my : MM;
      Float scl[MM];
      unsigned act = 0;
#pragma unroll
      for (int k = 0; k < MM; ++k) {
        scl[k] = 0;
        if (k < n_reg) {
          const MinorMeta& m = bm[ibnd].m[rsel][k];
          if (!(m.mE < g0 || m.mS > g0)) {  // intervals are whole G-aligned chunks inside the band
            Float v = sc[k];
            if (m.flags & 1) {
              v = v * dens;  // :469
              if (m.idx_scaling > 0) {  // :470-478
                if (m.flags & 2) v = v * ((Float)1 - cgs[k] * vmr_fact * dry_fact);
                else v = v * (cgs[k] * vmr_fact * dry_fact);
              }
            }
            scl[k] = v;
            act |= 1u << k;
          }
        }
      }
      // slots the wave walks: up to the last one any of its lanes uses (wave-uniform; lanes without that slot add 0 x a row they may read)
      int nslot = 0;
#pragma unroll
      for (int k = 0; k < MM; ++k)
        if (__builtin_amdgcn_ballot_w64((act >> k) & 1u) != 0) nslot = k + 1;
      const Float f0 = fn0.x, f1 = fn0.y, f2 = fn1.x, f3 = fn1.y;
      const Float* A1 = A0 + sE;
      const Float* B1 = B0 + sE;
      const Float* r1_0 = M0 + ((jT_s - Tmin) * nE + (em.x - emin)) * RS;
      const Float* r2_0 = M0 + ((jT_s + 1 - Tmin) * nE + (em.y - emin)) * RS;
      const int plane = nT * nE * RS;
      // rows of slot q for this lane; a lane that does not use the slot reads its major rows instead (always inside the slab)
      auto slot_rows = [&](int q, const Float*& p1, const Float*& p2) {
        const bool on = ((act >> q) & 1u) != 0;
        p1 = on ? r1_0 + q * plane : A0;
        p2 = on ? r2_0 + q * plane : A0;
      };
      Float2 kb[DEPTH][4];
      auto rd_major = [&](Float2 (&k)[4], int h) {  // h: (g-point pair, lower / upper temperature)
        const Float* b0 = ((h & 1) ? B0 : A0) + 2 * (h >> 1);
        const Float* b1 = ((h & 1) ? B1 : A1) + 2 * (h >> 1);
        k[0] = ld2(b0); k[1] = ld2(b1); k[2] = ld2(b0 + sP); k[3] = ld2(b1 + sP);
      };
      auto rd_minor = [&](Float2 (&k)[4], const Float* p1, const Float* p2, int j) {  // j: g-point pair
        k[0] = ld2(p1 + 2 * j); k[1] = ld2(p1 + RS + 2 * j); k[2] = ld2(p2 + 2 * j); k[3] = ld2(p2 + RS + 2 * j);
      };
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
            acc[j] = acc[j] + m;
            acc[j + 1] = acc[j + 1] + n;
            asm volatile("" : "+v"(acc[j]), "+v"(acc[j + 1]));
          }
          if (h + DEPTH < G) rd_major(k, h + DEPTH);
          else if (nslot > 0) rd_minor(k, c1, c2, h + DEPTH - G);
          __builtin_amdgcn_sched_barrier(0);
        }
      }
      TAU_T(2);
      if constexpr (TAU_DMA_LATE && !ROT) issue_dma(s + 1, ibnd_n);
      // next stage's major weights: their registers are free now, and the request is a minor pass ahead of its use
      load_major(nq.flav_major, mj);
      load_minor_w(nq, mw);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll 1
      for (int q = 0; q < nslot; ++q) {
        Float scaling = scl[0];
#pragma unroll
        for (int u = 1; u < MM; ++u) scaling = (q == u) ? scl[u] : scaling;
        const bool more = q + 1 < nslot;  // (wave-uniform)
        const Float* n1;
        const Float* n2;
        slot_rows(q + 1, n1, n2);
#pragma unroll
        for (int j = 0; j < G / 2; ++j) {
          Float2 (&k)[4] = kb[j % DEPTH];
          Float s_ = f0 * k[0].x, t_ = f0 * k[0].y;
          s_ = fma(f1, k[1].x, s_); t_ = fma(f1, k[1].y, t_);
          s_ = fma(f2, k[2].x, s_); t_ = fma(f2, k[2].y, t_);
          s_ = fma(f3, k[3].x, s_); t_ = fma(f3, k[3].y, t_);
          acc[2 * j] = fma(scaling, s_, acc[2 * j]);
          acc[2 * j + 1] = fma(scaling, t_, acc[2 * j + 1]);
          asm volatile("" : "+v"(acc[2 * j]), "+v"(acc[2 * j + 1]));
          if (j + DEPTH < G / 2) rd_minor(k, c1, c2, j + DEPTH);
          else if (more) rd_minor(k, n1, n2, j + DEPTH - G / 2);
          __builtin_amdgcn_sched_barrier(0);
        }
        c1 = n1; c2 = n2;
      }
      TAU_T(3);
      if (n_my > MM) {
        // the band's intervals beyond the MM held in registers: amounts requested here, same expressions (:461-480)
#pragma unroll 1
        for (int k = MM; k < n_my; ++k) {
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
    }
    load_minor(ibnd_n, nq, mn);
    __builtin_amdgcn_sched_barrier(0);  // keep these requests ahead of the stores that follow
    if constexpr (RAYL != 0) {
      // compute_tau_rayleigh (:548-555: interpolate2D with the reference's association) on the staged table rows,
      // combine_abs_and_rayleigh and the optional by-band increment on the values in registers (rayl_finish), and
      // the stage's 3 x G stores.  Rows [regime][t][eta] behind the minor planes; unconditional stores as below.
      const int rowsUp_ = (has_up ? bm[ibnd].cnt[1] : 0) * nT * nE;
      const int jT_r = RTE_PARKED_I(0, jT);
      const Float wray_s = wray;
      const Float* R1 = sl + (rowsMaj + rowsLo + rowsUp_ + ((itropo * nT + (jT_r - Tmin)) * nE + (je1 - emin))) * RS;
      const Float* R2 = sl + (rowsMaj + rowsLo + rowsUp_ + ((itropo * nT + (jT_r + 1 - Tmin)) * nE + (je2 - emin))) * RS;
      char* const splane = reinterpret_cast<char*>(a.rf.ssa + (size_t)ncl * g0);
      char* const gplane = RAYL == 3 ? nullptr : reinterpret_cast<char*>(a.rf.g + (size_t)ncl * g0);
#pragma unroll
      for (int j = 0; j < G; j += 2) {
        const Float2 a0 = ld2(R1 + j), a1 = ld2(R1 + RS + j), b0 = ld2(R2 + j), b1 = ld2(R2 + RS + j);
        const Float ka = fr0.x * a0.x + fr0.y * a1.x + fr1.x * b0.x + fr1.y * b1.x;
        const Float kb = fr0.x * a0.y + fr0.y * a1.y + fr1.x * b0.y + fr1.y * b1.y;
#pragma unroll
        for (int u = 0; u < 2; ++u) {
          Float t_, s_, g_;
          rayl_finish(acc[j + u], (u == 0 ? ka : kb) * wray_s, RAYL == 2, cld_t, cld_s, cld_g, t_, s_, g_);
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
  }
  if constexpr (ROT) {
    if (ALLRUN ? nstage > 0 : have_prev) flush(g0_prev, addv_prev);
  }
#ifdef TAU_TIMING
  TAU_T(DMA ? 6 : 3);
  if ((tid & 63) == 0)
    for (int k = 0; k < (DMA ? 7 : 4); ++k) atomicAdd(&tau_clk[k + (DMA && ROT ? 8 : 0)], tacc[k]);
#endif
  };
  // (round 3: 5.30 -> 5.19 ms at 1e5 x 60 x 256; -DTAU_NO_ROT for the A/B.  The fused variants end a stage with LDS
  // reads of their own slab and stay as they were.)
#ifdef TAU_NO_ROT
  constexpr bool ROTATE = false;
#else
  constexpr bool ROTATE = RAYL == 0 && !ADDB;  // (the by-band operand's variant would spill)
#endif
  bool rotated = false;
  if constexpr (ROTATE) rotated = tid >= TILE / 2;  // (wave-uniform)
  if constexpr (ROTATE) {
    if (rotated) {
      if (all_run) run_stages(std::true_type{}, std::true_type{}); else run_stages(std::false_type{}, std::true_type{});
      return;
    }
  }
  if (all_run) run_stages(std::true_type{}, std::false_type{}); else run_stages(std::false_type{}, std::false_type{});

}

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
  if (cb.g) cb.g[idx] = g_;  // (nullptr: clear sky, the caller keeps "g = 0" implicit)
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
            if (a.cb.g) *reinterpret_cast<Float*>(reinterpret_cast<char*>(a.cb.g) + po) = g_;
          }
        }
      }
    }
  }
}

}  // namespace

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
  const Float* d_kmajor = c.in_table(kmajor, tn * (npres + 1) * ngpt);
  MinorTables lo{c.in_table(kminor_lower, tn * *nminorklower_), c.in(minor_limits_gpt_lower, (size_t)2 * nlo),
                 c.in(minor_scales_with_density_lower, (size_t)nlo), c.in(scale_by_complement_lower, (size_t)nlo),
                 c.in(idx_minor_lower, (size_t)nlo), c.in(idx_minor_scaling_lower, (size_t)nlo),
                 c.in(kminor_start_lower, (size_t)nlo), nullptr, nullptr, nlo};
  MinorTables up{c.in_table(kminor_upper, tn * *nminorkupper_), c.in(minor_limits_gpt_upper, (size_t)2 * nup),
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
    d_krayl = c.in_table(rh->krayl, (size_t)ntemp * neta * ngpt * 2);
    d_col_dry = c.in(rh->col_dry, ncl);
    if (rh->cld_tau) { cb.cld_tau = c.in(rh->cld_tau, ncl * nbnd); cb.cld_ssa = c.in(rh->cld_ssa, ncl * nbnd); cb.cld_g = c.in(rh->cld_g, ncl * nbnd); }
    cb.tau_abs = d_tau; cb.tau = d_tau;  // the direct kernels combine in place
    cb.ssa = c.out_lazy(rh->ssa, ncl * ngpt);
    cb.g = rh->g ? c.out_lazy(rh->g, ncl * ngpt) : nullptr;  // (nullptr without clouds: g = 0 is not stored)
  }
  hipStream_t st = rte::stream();
  if (!rh && !c.any_host() && rte::is_device_memory(d_tau)) rte::fork_point(d_tau, sizeof(Float) * ncl * ngpt);
  // layer limits of the two regimes per column (:274-285) + "regimes overlap somewhere" flag
  int* lim = (int*)rte::scratch(sizeof(int) * (4 * (size_t)ncol + 3));
  int* overlap = lim + 4 * (size_t)ncol;
  int* nonzero = overlap + 2;    // some element of the incoming tau is not zero (tau_is_zero_kernel, plain-ABI calls)
  int* irregular = overlap + 1;  // some column's layer ranges are not those of its tropo flags (see tropo_limits_kernel)
  const size_t wl_cap = (size_t)cdiv(ncol, 256) * nlay * nbnd;  // tiles are at least 256 columns wide
  int* const worklist = (int*)rte::scratch(sizeof(int) * (1 + 3 * wl_cap));
  int* const valid_word = share_boxes() ? gs().shared.valid : nullptr;  // (null until the first sharing call has allocated it)
  {
    rte::ProfScope p("tau_absorption_setup");
    // (the validity word of a geometry shared with compute_Planck_source is cleared here too: it is set again only if this
    //  call's geometry kernel runs)
    hipLaunchKernelGGL(zero_words_kernel, dim3(1), dim3(256), 0, st, overlap, 3u, worklist, 1u, valid_word, valid_word ? 1u : 0u);
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
#ifdef TAU_FORCE_GW8
    const int gw = aligned(8) ? 8 : 0;
#else
    const int gw = aligned(16) ? 16 : (aligned(8) ? 8 : 0);
#endif
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
      constexpr int PIECE = 16 / (int)sizeof(Float);  // table rows are staged in 16-byte pieces: row starts must be whole pieces
      ok = ok && (nn[r] == 0 || nk2[r] % PIECE == 0);
      for (int i = 0; i < nn[r] && ok; ++i) {
        ok = ok && (ks[r][i] - 1) % PIECE == 0;
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
    // stage list of the matrix-core kernel: bands in order, 16 g-points at a time, 4 minor intervals per sub-stage
    cache.mx_stages.clear();
    if (ok && gw == 16) {
      for (int b = 0; b < nbnd; ++b) {
        const BandMeta& bmh = cache.bands[b];
        const int nmax = bmh.cnt[0] > bmh.cnt[1] ? bmh.cnt[0] : bmh.cnt[1];
        const int nsub = nmax <= 4 ? 1 : (nmax + 3) / 4;
        for (int g0 = bmh.gS; g0 <= bmh.gE; g0 += 16)
          for (int sub = 0; sub < nsub; ++sub) {
            MxStageRec r{};
            r.b = b; r.g0 = g0; r.k0 = 4 * sub; r.flags = (sub == 0 ? 1 : 0) | (sub == nsub - 1 ? 2 : 0);
            for (int q = 0; q < 2; ++q) {
              r.flav[q] = bmh.flav[q];
              for (int j = 0; j < 4; ++j) {
                const int k = r.k0 + j;
                if (k < bmh.cnt[q] && bmh.m[q][k].mS <= g0 && bmh.m[q][k].mE >= g0) {
                  r.act[q] |= 1u << j;
                  r.koff[q][j] = (unsigned)(bmh.m[q][k].kstart + (g0 - bmh.m[q][k].mS));
                }
              }
            }
            cache.mx_stages.push_back(r);
          }
      }
      if ((int)cache.mx_stages.size() > MX_MAXSTAGE) cache.mx_stages.clear();
    }
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
  // (accumulating onto a host-visible -- pinned / managed -- tau: hardware floating-point atomics are not defined there,
  //  the direct kernels' plain read - add - write is)
  const bool fast = cache.fast_ok && ncol >= 512 && !g_tau_force_direct && ncl < ((size_t)1 << 29) &&
                    sizeof(Float) * (tn * (npres + 1) * ngpt + tn * ((size_t)nkl_ + nku_ + 2 * (size_t)ngpt + 4)) < ((size_t)1 << 35) &&  // (32-bit row offsets in 16-byte units)
                    al(d_fmajor, 16) && al(d_fminor, 16) && al(d_col_mix, 16) && al(d_jeta, 8) &&
                    (overwrite_ok || rte::is_device_memory(d_tau)) &&
                    (!rh || (overwrite_ok && g_geom_variant == 2 && nflav <= MAXFLAV && neta < 31 && ntemp < 31 &&
                             npres + 1 < 63 && nbnd <= 16));  // (16: the fused variants' static LDS + the band table, 160 KB)  // (fused: the bands tile the g-points -- else tau was zero-filled above --
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
  // (ONE allocation: the DMA staging of tau_absorption_v9_kernel addresses every table row as a 32-bit count of 16-byte
  //  units from kmaj_g)
  auto even = [](size_t n) { return (n + 3) & ~(size_t)3; };  // (whole 16-byte pieces in either precision)
  const size_t n_maj = even(tn * (npres + 1) * ngpt), n_klo = even(tn * (nkl > 0 ? nkl : 1)), n_kup = even(tn * (nku > 0 ? nku : 1));
  const size_t n_ray = rh ? even(tn * ngpt * 2) : 0;
  Float* kmaj_g = (Float*)rte::scratch(sizeof(Float) * (n_maj + n_klo + n_kup + n_ray));
  Float* klo_g = kmaj_g + n_maj;
  Float* kup_g = klo_g + n_klo;
  Float* kray_g = rh ? kup_g + n_kup : nullptr;
  // band metadata lives in a persistent device buffer and is uploaded only when the host plan was rebuilt
  // (a per-call copy from pageable host memory stalls the submitting thread)
  bool bm_fresh = false;
  BandMeta* d_bm = (BandMeta*)rte::persistent(plan_slot, sizeof(BandMeta) * MAXB + sizeof(MxStageRec) * MX_MAXSTAGE, &bm_fresh);
  MxStageRec* d_mx_stages = (MxStageRec*)(d_bm + MAXB);
  {
    rte::ProfScope p("relayout_gfast_kernel");
    if (bm_fresh || cache.uploads_pending) {
      if (!cache.mx_stages.empty())
        HIP_CHECK(hipMemcpyAsync(d_mx_stages, cache.mx_stages.data(), sizeof(MxStageRec) * cache.mx_stages.size(), hipMemcpyHostToDevice, st));
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
  v.nonzero = nonzero; v.run_when = 0;
  // plain-ABI accumulate onto device memory: find out first whether tau is (still) the zero array the frontend made of it
  const bool zero_check = !overwrite_ok && v.atomic_ok && !g_tau_no_zero_check && al(d_tau, 16) && cache.gw != 0 &&
                          (g_tau_variant == 9 || g_tau_variant == 11 || cache.gw != 16 || d_add != nullptr);
  if (zero_check) {
    rte::ProfScope p("tau_is_zero_kernel");
    hipLaunchKernelGGL(tau_is_zero_kernel, dim3(256 * 16), dim3(256), 0, st, (const Float*)d_tau, ncl * (size_t)ngpt, nonzero);
  }
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
  // ---- the matrix-core kernel (tau_mx.h; rte_hip_tau_variant(10)): double precision, 16-wide stages, no fused Rayleigh
#ifndef RTE_USE_SP
  const bool use_mx = g_tau_variant == 10 && cache.gw == 16 && rh == nullptr && ntemp < 32 && npres + 2 < 64 && neta <= 16 &&
                      2 * ntemp * (npres + 2) <= MX_NB && nflav <= MAXFLAV && nbnd <= MAXB && !cache.mx_stages.empty();
  if (use_mx) {
    constexpr int NW = 8, TILE = NW * 64;
    const unsigned tiles = cdiv(ncol, TILE);
    unsigned* sort_pk = (unsigned*)rte::scratch(sizeof(unsigned) * (size_t)tiles * nlay * nflav * TILE);
    int* n_lo = (int*)rte::scratch(sizeof(int) * (size_t)tiles * nlay);
    {
      rte::ProfScope p("tau_absorption_setup");
      hipLaunchKernelGGL((tau_mx_sort_kernel<TILE>), dim3(tiles, nlay), dim3(TILE), 0, st, ncol, nlay, nflav, neta, d_jtemp, d_jpress,
                         d_tropo, d_jeta, (const int*)overlap, (const int*)irregular, sort_pk, n_lo);
    }
    MxArgs m{};
    m.ncol = ncol; m.nlay = nlay; m.ngpt = ngpt; m.nbnd = nbnd; m.ntemp = ntemp; m.TE = TE; m.idx_h2o = *idx_h2o_;
    m.nk_lo = nkl; m.nk_up = nku; m.nflav = nflav; m.bmeta = d_bm; m.kmaj = kmaj_g; m.klo = klo_g; m.kup = kup_g;
    m.jeta = d_jeta; m.jtemp = d_jtemp; m.jpress = d_jpress; m.tropo = d_tropo; m.col_mix = d_col_mix; m.fmajor = d_fmajor;
    m.fminor = d_fminor; m.play = d_play; m.tlay = d_tlay; m.col_gas = d_col_gas; m.tau = d_tau; m.add_bybnd = d_add;
    m.skip_if = overlap; m.skip_if2 = irregular; m.sort_pk = sort_pk; m.n_lo = n_lo; m.stat = stats_dev() + 3;
    m.stages = d_mx_stages; m.nstage = (int)cache.mx_stages.size();
    const dim3 grid(tiles, nlay), blk(2 * TILE);
    const size_t dyn = sizeof(BandMeta) * nbnd;
    rte::ProfScope p("tau_absorption_kernel");
    if (overwrite_ok) {
      if (d_add) hipLaunchKernelGGL((tau_absorption_mx_kernel<NW, true, true>), grid, blk, dyn, st, m);
      else hipLaunchKernelGGL((tau_absorption_mx_kernel<NW, true, false>), grid, blk, dyn, st, m);
    } else {
      if (d_add) hipLaunchKernelGGL((tau_absorption_mx_kernel<NW, false, true>), grid, blk, dyn, st, m);
      else hipLaunchKernelGGL((tau_absorption_mx_kernel<NW, false, false>), grid, blk, dyn, st, m);
    }
    a.run_if2 = irregular;
  }
#else
  const bool use_mx = false;
#endif
  const bool use_v9 = g_tau_variant == 9 || g_tau_variant == 11 || cache.gw != 16 || d_add != nullptr || rh != nullptr;  // the single-role kernel exists for 16-wide stages only
  // the slab kernel of tau_slab.h (8 waves, LDS-DMA staging): the default; rte_hip_tau_variant(9) = the rounds 1-4 form with loader waves
  const bool use_slab = g_tau_variant != 9 && g_tau_variant != 7 && ngpt / cache.gw <= SLAB_MAXSTAGE;
  if (use_mx) {
    // (launched above)
  } else if (use_v9) {
    constexpr int NCW = TAU_NCW, NLW = TAU_NLW, SLAB9 = TAU_SLAB;  // compute + loader waves, 2 x 68 KB slab: one block per CU
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
    const dim3 grid(tiles, nlay);
    const size_t dyn = sizeof(BandMeta) * nbnd;
    const TileGeom* cg = d_geom;
    Geom2Args ga{};
    ga.ncol = ncol; ga.nlay = nlay; ga.nbnd = nbnd; ga.nflav = nflav; ga.slab_floats = SLAB9; ga.planck = false;
    ga.lim = lim; ga.jeta = d_jeta; ga.jtemp = d_jtemp; ga.jpress = d_jpress; ga.tropo = d_tropo; ga.bmeta = d_bm;
    ga.skip_if = overlap; ga.worklist = v.worklist; ga.valid_out = share ? gs().shared.valid : nullptr;
    ga.extra_planes = rh ? 2 : 0;
    ga.row_stride = use_slab ? cache.gw + 16 / (int)sizeof(Float) : 0;
    ga.irregular = irregular;
    ga.stat = stats_dev() + 2;
    if (share_masks() && gs().imask.seq >= 0 && gs().imask.seq + 1 == rte::call_seq() && gs().imask.jeta == jeta && gs().imask.jtemp == jtemp &&
        gs().imask.jpress == jpress && gs().imask.tropo == tropo && gs().imask.ncol == ncol && gs().imask.nlay == nlay &&
        gs().imask.nflav == nflav && !c.any_host()) {
      ga.imask = gs().imask.buf;
      ga.imask_nblk = cdiv(ncol, 256);
    }
#define RTE_TAU_K(OW, GW, AB, RV)                                                                                \
  do {                                                                                                            \
    if (use_slab) hipLaunchKernelGGL((tau_slab_kernel<NCW, SLAB9, OW, GW, 4, AB, RV, ((RV != 0 || AB) ? 2 : TAU_DEPTH)>), grid, dim3(NCW * 64), dyn, st, vk, cg); \
    else hipLaunchKernelGGL((tau_absorption_v9_kernel<NCW, V9_NLW, SLAB9, OW, GW, 4, AB, RV>), grid, dim3((NCW + V9_NLW) * 64), dyn, st, vk, cg); \
  } while (0)
#define RTE_LAUNCH_TAU9R_(GW, MMV, RV) \
  do { const TauV5& vk = v; RTE_TAU_K(true, GW, false, RV); } while (0)
#define RTE_LAUNCH_TAU9_(GW, AB)                                                                                  \
  do {                                                                                                            \
    if (overwrite_ok) { const TauV5& vk = v; RTE_TAU_K(true, GW, AB, 0); }                                        \
    else if (zero_check) {                                                                                        \
      TauV5 vk = v;                                                                                               \
      vk.run_when = 1; vk.overwrite = true;  /* tau is all zero: the overwriting instance */                      \
      RTE_TAU_K(true, GW, AB, 0);                                                                                 \
      vk.run_when = 2; vk.overwrite = false;  /* it is not: accumulate */                                         \
      RTE_TAU_K(false, GW, AB, 0);                                                                                \
    }                                                                                                             \
    else { const TauV5& vk = v; RTE_TAU_K(false, GW, AB, 0); }                                                    \
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
      else if (cb.g)  RTE_LAUNCH_TAU9R_(GW, 4, 1);                                                                \
      else            RTE_LAUNCH_TAU9R_(GW, 4, 3);                                                                \
    } else if (d_add) RTE_LAUNCH_TAU9_(GW, true); else RTE_LAUNCH_TAU9_(GW, false);                               \
  } while (0)
    if (cache.gw == 16) RTE_LAUNCH_TAU9(16); else RTE_LAUNCH_TAU9(8);
#undef RTE_LAUNCH_TAU9
#undef RTE_LAUNCH_TAU9_
#undef RTE_LAUNCH_TAU9R_
#undef RTE_TAU_K
  } else {
    rte::ProfScope p("tau_absorption_kernel");
    // <min waves per SIMD, g-points per register chunk>: measured best of {2,3} x {4,8,16} on MI355X
    hipLaunchKernelGGL((tau_absorption_v7_kernel<BS, V7_MINW, V7_HW, V7_SLAB>), dim3(cdiv(ncol, BS), nlay), dim3(BS), sizeof(BandMeta) * nbnd, st,
                       v);
  }
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
    if (use_mx) {
      // (no bounding boxes, no worklist)
    } else if (gft.kmaj)
      hipLaunchKernelGGL(tau_absorption_worklist_kernel<true>, dim3(aux ? 16384 : 4096), dim3(aux ? 64 : 256), 0, st, aw, gft,
                         (const int*)v.worklist, use_v9 ? TAU_NCW * 64 : BS, stats_dev() + 0);
    else
      hipLaunchKernelGGL(tau_absorption_worklist_kernel<false>, dim3(aux ? 16384 : 4096), dim3(aux ? 64 : 256), 0, st, aw, gft,
                         (const int*)v.worklist, use_v9 ? TAU_NCW * 64 : BS, stats_dev() + 0);
    if (rh) rayleigh_direct(nullptr, (const int*)v.worklist, TAU_NCW * 64);  // (lambda launches on st)
    st = main_st;
    if (aux) rte::aux_join();
  }
  RTE_CATCH(api_name)
}

#ifdef TAU_TIMING
extern "C" int rte_hip_tau_timing(unsigned long long* out /*[16]*/) {
  (void)hipDeviceSynchronize();
  (void)hipMemcpyFromSymbol(out, HIP_SYMBOL(tau_clk), sizeof(unsigned long long) * 16);
  unsigned long long z[16] = {0};
  (void)hipMemcpyToSymbol(HIP_SYMBOL(tau_clk), z, sizeof(z));
  return 0;
}
#endif
#if defined(MX_TIMING) && !defined(RTE_USE_SP)
extern "C" int rte_hip_mx_timing(unsigned long long* out /*[4]: column waves busy, waiting; matrix waves busy, waiting*/) {
  (void)hipDeviceSynchronize();
  (void)hipMemcpyFromSymbol(out, HIP_SYMBOL(mx_clk), sizeof(unsigned long long) * 4);
  unsigned long long z[4] = {0, 0, 0, 0};
  (void)hipMemcpyToSymbol(HIP_SYMBOL(mx_clk), z, sizeof(z));
  return 0;
}
#endif
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
// g == NULL (clear sky only): combine_abs_and_rayleigh's g = 0 is NOT stored -- 10.75 GB less to write at 1e5 x 60 x 224, and
// rte_sw_solver_2stream takes g == NULL as "g = 0" (same bits as with the array of zeros, nothing to read).
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
  if (!g && cld_tau) return -1;  // g may be omitted (g = 0 stays implicit) only without clouds
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
  const Float* d_krayl = c.in_table(krayl, (size_t)ntemp * neta * ngpt * 2);
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

}  // extern "C"
