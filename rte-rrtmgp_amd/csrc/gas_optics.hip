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

#include "common.h"

namespace {

using rte::cdiv;

constexpr int GC = 16;  // g-points held in registers per chunk

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
                     int* __restrict__ jpress) {
  const int icol = blockIdx.x * blockDim.x + threadIdx.x;
  const int ilay = blockIdx.y, iflav = blockIdx.z;
  if (icol >= ncol) return;
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
  if (iflav == 0) {
    jtemp[cl] = jt;
    jpress[cl] = (int)jpress_aint;
    tropo[cl] = trop;
  }
  // :121-168
  const int itropo = trop ? 0 : 1;
  const int igas_1 = flavor[2 * iflav], igas_2 = flavor[2 * iflav + 1];
  const Float cg1 = col_gas[cl + ncl * igas_1], cg2 = col_gas[cl + ncl * igas_2];
  const size_t clf = cl + ncl * iflav;
  Float fmn[4], fmj[8], cm[2];
  int je[2];
#pragma unroll
  for (int itemp = 0; itemp < 2; ++itemp) {
    const int t = jt + itemp;  // 1-based
    const size_t v = (size_t)itropo + 2 * ((size_t)0 + (size_t)(ngas + 1) * (t - 1));
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
  jeta[2 * clf] = je[0];
  jeta[2 * clf + 1] = je[1];
  col_mix[2 * clf] = cm[0];
  col_mix[2 * clf + 1] = cm[1];
#pragma unroll
  for (int i = 0; i < 4; ++i) fminor[4 * clf + i] = fmn[i];
#pragma unroll
  for (int i = 0; i < 8; ++i) fmajor[8 * clf + i] = fmj[i];
}

// -------------------------------------------------------------------------------------------
// layer limits of the lower / upper atmosphere: reference :274-285 (minloc/maxloc with mask,
// first extremal location; 0 = no such layer)
// -------------------------------------------------------------------------------------------
__global__ void tropo_limits_kernel(int ncol, int nlay, const Float* __restrict__ play,
                                    const Bool* __restrict__ tropo, int* __restrict__ lim /*(ncol,4)*/) {
  const int icol = blockIdx.x * blockDim.x + threadIdx.x;
  if (icol >= ncol) return;
  const bool top_at_1 = play[0] < play[(size_t)ncol * (nlay - 1)];
  int minloc_t = 0, maxloc_n = 0;
  Float pmin = 0, pmax = 0;
  for (int ilay = 0; ilay < nlay; ++ilay) {
    const size_t cl = icol + (size_t)ncol * ilay;
    const Float p = play[cl];
    if (tropo[cl]) {
      if (minloc_t == 0 || p < pmin) { minloc_t = ilay + 1; pmin = p; }
    } else {
      if (maxloc_n == 0 || p > pmax) { maxloc_n = ilay + 1; pmax = p; }
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
}

// Per band, the ordered list of minor intervals whose g-point range intersects the band
// (one wave; ordered compaction by ballot so the reference's interval order is preserved).
__global__ void plan_minor_kernel(int nbnd, const int* __restrict__ band_lims_gpt, int nminor,
                                  const int* __restrict__ minor_limits_gpt, int* __restrict__ cnt /*(nbnd)*/,
                                  int* __restrict__ list /*(nminor,nbnd)*/) {
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
    // :461-480
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
__global__ void __launch_bounds__(256)
tau_absorption_kernel(int ncol, int nlay, int ngpt, int neta, int npres, int ntemp, int idx_h2o,
                      const int* __restrict__ gpoint_flavor, const int* __restrict__ band_lims_gpt,
                      const Float* __restrict__ kmajor, MinorTables lower, MinorTables upper,
                      const int* __restrict__ lim, const Bool* __restrict__ tropo,
                      const Float* __restrict__ col_mix, const Float* __restrict__ fmajor,
                      const Float* __restrict__ fminor, const Float* __restrict__ play,
                      const Float* __restrict__ tlay, const Float* __restrict__ col_gas,
                      const int* __restrict__ jeta, const int* __restrict__ jtemp,
                      const int* __restrict__ jpress, Float* __restrict__ tau) {
  const int icol = blockIdx.x * blockDim.x + threadIdx.x;
  const int ilay = blockIdx.y, ibnd = blockIdx.z;
  if (icol >= ncol) return;
  const size_t ncl = (size_t)ncol * nlay;
  const size_t cl = icol + (size_t)ncol * ilay;
  const int gptS = band_lims_gpt[2 * ibnd] - 1, gptE = band_lims_gpt[2 * ibnd + 1] - 1;
  const int itropo = tropo[cl] ? 0 : 1;
  const int iflav = gpoint_flavor[itropo + 2 * gptS] - 1;
  const size_t clf = cl + ncl * iflav;
  const int jT = jtemp[cl];
  const int jp = jpress[cl] + itropo + 1;  // "jpress + itropo": levels jp-1 and jp (1-based)
  const int je1 = jeta[2 * clf], je2 = jeta[2 * clf + 1];
  const Float cm1 = col_mix[2 * clf], cm2 = col_mix[2 * clf + 1];
  Float fm[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) fm[i] = fmajor[8 * clf + i];
  const size_t tn = (size_t)ntemp * neta;
  const size_t gstride = tn * (npres + 1);
  // corner offsets (without the g-point term) into kmajor(ntemp,neta,npres+1,ngpt)
  const size_t a0 = (size_t)(jT - 1) + (size_t)ntemp * (je1 - 1) + tn * (size_t)(jp - 2);
  const size_t b0 = (size_t)jT + (size_t)ntemp * (je2 - 1) + tn * (size_t)(jp - 2);
  const Float P = play[cl], T = tlay[cl];
  const int lay1 = ilay + 1;
  const int lo1 = lim[icol], lo2 = lim[icol + ncol];
  const int up1 = lim[icol + 2 * (size_t)ncol], up2 = lim[icol + 3 * (size_t)ncol];
  const bool in_lower = lo1 > 0 && lay1 >= lo1 && lay1 <= lo2;
  const bool in_upper = up1 > 0 && lay1 >= up1 && lay1 <= up2;

  for (int g0 = gptS; g0 <= gptE; g0 += GC) {
    Float acc[GC];
#pragma unroll
    for (int j = 0; j < GC; ++j) acc[j] = (g0 + j <= gptE) ? tau[cl + ncl * (size_t)(g0 + j)] : (Float)0;
#pragma unroll
    for (int j = 0; j < GC; ++j) {
      if (g0 + j <= gptE) {
        const Float* ka = kmajor + gstride * (size_t)(g0 + j) + a0;
        const Float* kb = kmajor + gstride * (size_t)(g0 + j) + b0;
        // :791-801
        const Float tau_major =
            cm1 * (fm[0] * ka[0] + fm[1] * ka[ntemp] + fm[2] * ka[tn] + fm[3] * ka[tn + ntemp]) +
            cm2 * (fm[4] * kb[0] + fm[5] * kb[ntemp] + fm[6] * kb[tn] + fm[7] * kb[tn + ntemp]);
        acc[j] = acc[j] + tau_major;
      }
    }
    if (in_lower)
      minor_chunk(lower, 0, ibnd, g0, gptE, ncol, ncl, cl, ntemp, neta, idx_h2o, P, T, jT, col_gas, fminor,
                  jeta, gpoint_flavor, acc);
    if (in_upper)
      minor_chunk(upper, 1, ibnd, g0, gptE, ncol, ncl, cl, ntemp, neta, idx_h2o, P, T, jT, col_gas, fminor,
                  jeta, gpoint_flavor, acc);
#pragma unroll
    for (int j = 0; j < GC; ++j)
      if (g0 + j <= gptE) tau[cl + ncl * (size_t)(g0 + j)] = acc[j];
  }
}

// -------------------------------------------------------------------------------------------
// compute_tau_rayleigh: reference :506-565
// -------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256)
tau_rayleigh_kernel(int ncol, int nlay, int ngpt, int neta, int ntemp, int idx_h2o,
                    const int* __restrict__ gpoint_flavor, const int* __restrict__ band_lims_gpt,
                    const Float* __restrict__ krayl, const Float* __restrict__ col_dry,
                    const Float* __restrict__ col_gas, const Float* __restrict__ fminor,
                    const int* __restrict__ jeta, const Bool* __restrict__ tropo,
                    const int* __restrict__ jtemp, Float* __restrict__ tau_rayleigh) {
  const int icol = blockIdx.x * blockDim.x + threadIdx.x;
  const int ilay = blockIdx.y, ibnd = blockIdx.z;
  if (icol >= ncol) return;
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
    tau_rayleigh[cl + ncl * (size_t)g] = k * w;
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

__global__ void __launch_bounds__(256)
planck_source_kernel(int ncol, int nlay, int ngpt, int neta, int npres, int ntemp, int nPlanckTemp,
                     const Float* __restrict__ tlay, const Float* __restrict__ tlev,
                     const Float* __restrict__ tsfc, int sfc_lay, const Float* __restrict__ fmajor,
                     const int* __restrict__ jeta, const Bool* __restrict__ tropo,
                     const int* __restrict__ jtemp, const int* __restrict__ jpress,
                     const int* __restrict__ band_lims_gpt, const Float* __restrict__ pfracin,
                     Float temp_ref_min, Float totplnk_delta_r, const Float* __restrict__ totplnk,
                     const int* __restrict__ gpoint_flavor, Float* __restrict__ sfc_src,
                     Float* __restrict__ lay_src, Float* __restrict__ lev_src,
                     Float* __restrict__ sfc_source_Jac) {
  const int icol = blockIdx.x * blockDim.x + threadIdx.x;
  const int ibnd = blockIdx.y;
  if (icol >= ncol) return;
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

}  // namespace

// ===============================================================================================
// C ABI
// ===============================================================================================
extern "C" {

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
  int* d_jtemp = c.out(jtemp, ncl);
  Float* d_fmajor = c.out(fmajor, 8 * ncl * nflav);
  Float* d_fminor = c.out(fminor, 4 * ncl * nflav);
  Float* d_col_mix = c.out(col_mix, 2 * ncl * nflav);
  Bool* d_tropo = c.out(tropo, ncl);
  int* d_jeta = c.out(jeta, 2 * ncl * nflav);
  int* d_jpress = c.out(jpress, ncl);
  rte::ProfScope p("interpolation_kernel");
  dim3 grid(cdiv(ncol, 256), nlay, nflav), block(256);
  hipLaunchKernelGGL(interpolation_kernel, grid, block, 0, rte::stream(), ncol, nlay, ngas, nflav, neta,
                     npres, ntemp, d_flavor, d_temp_ref, d_press_ref_log, press_ref_log_delta_inv,
                     *temp_ref_min, *temp_ref_delta, temp_ref_delta_inv, press_ref_trop, d_vmr_ref, d_play,
                     d_tlay, d_col_gas, d_jtemp, d_fmajor, d_fminor, d_col_mix, d_tropo, d_jeta, d_jpress);
}

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
  const int ncol = *ncol_, nlay = *nlay_, nbnd = *nbnd_, ngpt = *ngpt_, ngas = *ngas_,
            nflav = *nflav_, neta = *neta_, npres = *npres_, ntemp = *ntemp_;
  const int nlo = *nminorlower_, nup = *nminorupper_;
  if (ncol <= 0 || nlay <= 0 || ngpt <= 0) return;
  rte::Call c("rrtmgp_compute_tau_absorption");
  const size_t ncl = (size_t)ncol * nlay;
  const size_t tn = (size_t)ntemp * neta;
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
  Float* d_tau = c.inout(tau, ncl * ngpt);
  // plans and layer limits (device scratch)
  int* lim = (int*)rte::scratch(sizeof(int) * 4 * (size_t)ncol);
  int* plan = (int*)rte::scratch(sizeof(int) * ((size_t)2 * nbnd + (size_t)nbnd * (nlo + nup) + 2));
  int* cnt_lo = plan;
  int* cnt_up = plan + nbnd;
  int* list_lo = plan + 2 * nbnd;
  int* list_up = list_lo + (size_t)nbnd * nlo;
  lo.cnt = cnt_lo; lo.list = list_lo;
  up.cnt = cnt_up; up.list = list_up;
  {
    rte::ProfScope p("tau_absorption_setup");
    hipLaunchKernelGGL(tropo_limits_kernel, dim3(cdiv(ncol, 256)), dim3(256), 0, rte::stream(), ncol, nlay,
                       d_play, d_tropo, lim);
    hipLaunchKernelGGL(plan_minor_kernel, dim3(1), dim3(RTE_WAVE), 0, rte::stream(), nbnd, d_band_lims, nlo,
                       lo.limits, cnt_lo, list_lo);
    hipLaunchKernelGGL(plan_minor_kernel, dim3(1), dim3(RTE_WAVE), 0, rte::stream(), nbnd, d_band_lims, nup,
                       up.limits, cnt_up, list_up);
  }
  rte::ProfScope p("tau_absorption_kernel");
  dim3 grid(cdiv(ncol, 256), nlay, nbnd), block(256);
  hipLaunchKernelGGL(tau_absorption_kernel, grid, block, 0, rte::stream(), ncol, nlay, ngpt, neta, npres,
                     ntemp, *idx_h2o_, d_gpoint_flavor, d_band_lims, d_kmajor, lo, up, lim, d_tropo, d_col_mix,
                     d_fmajor, d_fminor, d_play, d_tlay, d_col_gas, d_jeta, d_jtemp, d_jpress, d_tau);
}

void rrtmgp_compute_tau_rayleigh(const int* ncol_, const int* nlay_, const int* nbnd_,
                                 const int* ngpt_, const int* ngas_, const int* nflav_,
                                 const int* neta_, const int* npres_, const int* ntemp_,
                                 const int* gpoint_flavor, const int* band_lims_gpt,
                                 const Float* krayl, const int* idx_h2o_, const Float* col_dry,
                                 const Float* col_gas, const Float* fminor, const int* jeta,
                                 const Bool* tropo, const int* jtemp, Float* tau_rayleigh) {
  const int ncol = *ncol_, nlay = *nlay_, nbnd = *nbnd_, ngpt = *ngpt_, ngas = *ngas_,
            nflav = *nflav_, neta = *neta_, ntemp = *ntemp_;
  (void)npres_;
  if (ncol <= 0 || nlay <= 0 || ngpt <= 0) return;
  rte::Call c("rrtmgp_compute_tau_rayleigh");
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
  Float* d_tau = c.out(tau_rayleigh, ncl * ngpt);
  rte::ProfScope p("tau_rayleigh_kernel");
  dim3 grid(cdiv(ncol, 256), nlay, nbnd), block(256);
  hipLaunchKernelGGL(tau_rayleigh_kernel, grid, block, 0, rte::stream(), ncol, nlay, ngpt, neta, ntemp,
                     *idx_h2o_, d_gpoint_flavor, d_band_lims, d_krayl, d_col_dry, d_col_gas, d_fminor, d_jeta,
                     d_tropo, d_jtemp, d_tau);
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
  Float* d_sfc_src = c.out(sfc_src, (size_t)ncol * ngpt);
  Float* d_lay_src = c.out(lay_src, ncl * ngpt);
  Float* d_lev_src = c.out(lev_src, (size_t)ncol * (nlay + 1) * ngpt);
  Float* d_sfc_jac = c.out(sfc_source_Jac, (size_t)ncol * ngpt);
  const Float totplnk_delta_r = (Float)1 / *totplnk_delta;  // :636
  rte::ProfScope p("planck_source_kernel");
  dim3 grid(cdiv(ncol, 256), nbnd), block(256);
  hipLaunchKernelGGL(planck_source_kernel, grid, block, 0, rte::stream(), ncol, nlay, ngpt, neta, npres, ntemp,
                     nPlanckTemp, d_tlay, d_tlev, d_tsfc, *sfc_lay_, d_fmajor, d_jeta, d_tropo, d_jtemp,
                     d_jpress, d_band_lims, d_pfracin, *temp_ref_min, totplnk_delta_r, d_totplnk,
                     d_gpoint_flavor, d_sfc_src, d_lay_src, d_lev_src, d_sfc_jac);
}

}  // extern "C"
