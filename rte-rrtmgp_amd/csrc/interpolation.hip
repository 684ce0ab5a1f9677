// interpolation.hip -- rrtmgp_interpolation (reference rrtmgp/kernels/mo_gas_optics_rrtmgp_kernels.F90:37-170) for gfx950.
// Conventions of the gas-optics kernels: lanes of a wavefront = 64 consecutive columns (unit stride on every (ncol, ...)
// array); semantics follow the reference's `default` CPU kernels, NOT its OpenACC variant.
#include "gas_optics_common.h"

namespace {
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
    // 16 bytes per lane and store, 4 KB contiguous per instruction (8 bytes per lane: 1.37-1.51 against 1.27-1.35 ms)
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int e = 2 * t + 512 * k;
      if (e < 8 * nc) {
        Float2 v; v.x = s_fmj[(e >> 3) * 9 + (e & 7)]; v.y = s_fmj[(e >> 3) * 9 + (e & 7) + 1];
        *reinterpret_cast<Float2*>(fmajor + 8 * rec0 + e) = v;
      }
    }
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      const int e = 2 * t + 512 * k;
      if (e < 4 * nc) {
        Float2 v; v.x = s_fmn[(e >> 2) * 5 + (e & 3)]; v.y = s_fmn[(e >> 2) * 5 + (e & 3) + 1];
        *reinterpret_cast<Float2*>(fminor + 4 * rec0 + e) = v;
      }
    }
    if (t < nc) {
      Float2 v; v.x = s_cm[t * 3]; v.y = s_cm[t * 3 + 1];
      *reinterpret_cast<Float2*>(col_mix + 2 * rec0 + 2 * t) = v;
      int2 w; w.x = s_je[t * 3]; w.y = s_je[t * 3 + 1];
      *reinterpret_cast<int2*>(jeta + 2 * rec0 + 2 * t) = w;
    }
  }
  if (masks) {
    __syncthreads();
    if (t < mask_w) masks[((size_t)blockIdx.x + (size_t)gridDim.x * ilay) * mask_w + t] = s_mask[t];
  }
}

}  // namespace

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
