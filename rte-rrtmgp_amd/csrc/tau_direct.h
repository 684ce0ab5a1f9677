// tau_direct.h -- rrtmgp_compute_tau_absorption, direct-gather kernels and the small set-up kernels of the call.
//
// Reference: rrtmgp/kernels/mo_gas_optics_rrtmgp_kernels.F90:176-501 (compute_tau_absorption = layer limits :274-285 +
// gas_optical_depths_major :345-396 + gas_optical_depths_minor :402-501, interpolate3D_byflav :765-803 /
// interpolate2D_byflav :741-763).  These kernels read the caller's tables in their native layout (or, for worklist
// entries, the g-point-fastest copies) with the reference's own association: bit-identical to the oracle.  They do
//   * the whole call when the slab kernel (tau_slab.h) does not apply: fewer than 512 columns, tables whose bands or
//     minor intervals are not whole aligned chunks of 16 or 8 g-points, overlapping lower / upper layer ranges;
//   * the (tile, layer, band) items whose table bounding box does not fit the slab (tau_absorption_worklist_kernel).
// lanes of a wavefront = 64 consecutive columns; each thread owns one (column, layer, band) and walks the band's g-points
// in register chunks, so tau is read-modify-written once per call (summation order of the reference: major, lower minors
// in interval order, upper minors).
#pragma once
#include "gas_optics_common.h"

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


}  // namespace
