// planck.hip -- rrtmgp_compute_Planck_source (reference rrtmgp/kernels/mo_gas_optics_rrtmgp_kernels.F90:568-710) for gfx950.
// Each (column tile, band) block walks the layers so that the geometric mean of adjacent layers' Planck fractions needs
// no second gather; the production kernel splits loader and compute waves like the tau kernel (gas_optics_common.h).
#include "gas_optics_common.h"

namespace {
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
  // factored output (rte_hip_compute_Planck_source_factored): lay_src receives the Planck fraction itself, lev_src is not written,
  // and the band's Planck function at the layer / level temperatures goes to plk_lay (ncol, nlay, nbnd) / plk_lev (ncol, nlay+1, nbnd)
  int factored;
  Float *plk_lay, *plk_lev;
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
      if (q.factored && g0 == gptS) {
        q.plk_lay[cl + ncl * (size_t)ibnd] = pl_lay;
        q.plk_lev[icol + (size_t)ncol * ilay + nclv * (size_t)ibnd] = pl_lev;
      }
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
          if (q.factored) {
            lay_src[cl + ncl * (size_t)g] = pf;
          } else {
            lay_src[cl + ncl * (size_t)g] = pf * pl_lay;                                   // :674
            const Float lv = (ilay == 0) ? pf : sqrt(pf_prev[j] * pf);                      // :695,:699
            lev_src[icol + (size_t)ncol * ilay + nclv * (size_t)g] = lv * pl_lev;
          }
          if (ilay == sfc_lay - 1) {                                                      // :651-653
            sfc_src[icol + (size_t)ncol * g] = pf * pl_sfc;
            sfc_source_Jac[icol + (size_t)ncol * g] = pf * (pl_sfc1 - pl_sfc);
          }
          pf_prev[j] = pf;
        }
      }
    }
    const Float pl_top = planck_1d(tlev[icol + (size_t)ncol * nlay], temp_ref_min, totplnk_delta_r, tp, nPlanckTemp);
    if (q.factored) {
      if (g0 == gptS) q.plk_lev[icol + (size_t)ncol * nlay + nclv * (size_t)ibnd] = pl_top;
      continue;
    }
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
// compute_Planck_source, production kernel (planck_source_v9_kernel below).
// Block = (512 columns, one band): 8 compute waves (lanes = columns) walk the LAYERS, so the previous layer's Planck
// fractions stay in registers for the geometric mean at the interface (:699); 2 loader waves stage the bounding box of
// pfrac rows of layer l+1 (from the g-point-fastest table copy) into the other half of a double-buffered LDS slab while
// layer l is computed; one barrier per layer; the band's totplnk column sits in LDS for the whole block.
// tile_geom2_kernel (gas_optics_common.h) provides the boxes and sends (tile, band) pairs that do not fit the slab at
// some layer to the direct kernel.
// -------------------------------------------------------------------------------------------
struct PlanckV7 {
  int ncol, nlay, ngpt, ntemp, TE, nPlanckTemp, sfc_lay;
  Float temp_ref_min, totplnk_delta_r;
  const int *band_lims, *gpoint_flavor, *jeta, *jtemp, *jpress;
  const Bool* tropo;
  const Float *pf_g, *totplnk, *fmajor, *tlay, *tlev, *tsfc;
  Float *sfc_src, *lay_src, *lev_src, *sfc_jac;
  Float *plk_lay, *plk_lev;  // factored output (planck_source_v9_kernel<..., FACT>; see PlanckArgs)
  int* worklist;  // [0] = count, then (tile, band) pairs for planck_source_worklist_kernel
  const int* skip_if;  // plan guard raised: the direct kernel does the call
};

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

// FACT: factored output -- the Planck fraction goes to lay_src as it is (16 stores per stage instead of 32), the band's Planck
// function at the layer and level temperatures to plk_lay / plk_lev (once per band), nothing to lev_src.
// LCH: the layers are dealt to gridDim.y blocks per (tile, band) -- for calls of a few thousand columns, whose (tile, band) pairs
// do not fill the chip and whose time is then one block's walk over all layers (1 024 columns: 32 blocks, 244 us).  A block
// whose range starts inside the column first forms the Planck fractions of the layer above it (they enter the geometric mean at
// its first level, :699) and stores nothing for that layer; the same operations on the same values as the walk in one piece.
template <int NCW, int NLW, int SLAB, int G, bool FACT = false, bool LCH = false>
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
  // (LCH -- few tiles -- takes the blocks as they come: pinned, two tiles would use two of the eight XCDs)
  const unsigned lin = blockIdx.x, xcd = lin % 8, seq = lin / 8;
#if defined(PLANCKX_LINEAR)  // (experiment builds: the blocks as they come, a tile's bands spread over the XCDs)
  constexpr bool LINEAR = true;
#else
  constexpr bool LINEAR = LCH;
#endif
  const int ibnd = (int)((LINEAR ? lin : seq) % (unsigned)nbnd);
  const unsigned tile = LINEAR ? lin / (unsigned)nbnd : (seq / (unsigned)nbnd) * 8 + xcd;
  if (tile >= ntiles) return;  // block-uniform (grid padded to a multiple of 8 tiles)
#if defined(PLANCKX_STAGGER)  // (experiment builds: the bands of a tile start PLANCKX_STAGGER x 64 clocks apart)
  for (int i = 0; i < ibnd * PLANCKX_STAGGER; ++i) __builtin_amdgcn_s_sleep(1);
#endif
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
  // this block's layers [lb, le); lw0: the first layer it walks (lb - 1 for the fractions above its first level)
  const int lb = LCH ? (int)((blockIdx.y * nlay) / gridDim.y) : 0, le = LCH ? (int)(((blockIdx.y + 1) * nlay) / gridDim.y) : (int)nlay;
  const bool lastc = !LCH || blockIdx.y == gridDim.y - 1;  // the block with the last layer: top level, surface source
  const int lw0 = lb > 0 ? lb - 1 : 0, nwalk = le - lw0;
  const int spc = nwalk + ((lastc && lsfc != (int)nlay - 1) ? 1 : 0);
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
      const int ls = s % spc, l = ls < nwalk ? lw0 + ls : lsfc, g0 = gptS + (s / spc) * G;
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
  // (the surface values are formed where they are used, after the layer walk: held across it they were spilled -- 60 bytes of
  //  scratch per lane in the headline instantiation, none of it touched inside the layer loop, but a kernel with scratch
  //  costs 14 us more to launch)

  struct Idx { Bool tropo; int jT, jpress; Float tlay, tlev; };  // raw loaded values: nothing is derived at load
  struct Wts { Float2 fm[4]; int je1, je2; };                     // time, so no request waits for another
  auto load_idx = [&](unsigned c0, unsigned l, Idx& x) {  // c0: the lane's column (ic, or an opaque copy of it)
    const unsigned cl = c0 + ncol * l;
    x.tropo = a.tropo[cl];
    x.jT = a.jtemp[cl];
    x.jpress = a.jpress[cl];
    x.tlay = a.tlay[cl];
    x.tlev = a.tlev[cl];
  };
  auto load_wts = [&](unsigned c0, unsigned l, const Idx& x, Wts& w) {
    const size_t clf = (c0 + ncol * l) + (size_t)ncl * (x.tropo ? flav0 : flav1);
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
#pragma unroll 1
  for (int g0 = gptS; g0 <= gptE; g0 += G) {
    {
      // (an opaque copy of the column index: the prologue's 64-bit addresses are formed here, once per chunk, not hoisted out
      //  of the chunk loop and held -- one of them spilled -- across the layer walk)
      unsigned icp = ic;
      asm volatile("" : "+v"(icp));
      load_idx(icp, (unsigned)lw0, x0);
      load_idx(icp, min((unsigned)lw0 + 1u, nlay - 1), x1);
      load_wts(icp, (unsigned)lw0, x0, w0);
    }
#pragma unroll
    for (int j = 0; j < G; ++j) prev[j] = 0;
    // nothing outstanding at loop entry: the wait counts inside are then those of the steady state (requests of
    // the following layers, then this layer's 32 stores), not their merge with this prologue
    __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0)
#pragma unroll 1
    for (unsigned l = (unsigned)lw0; l < (unsigned)le; ++l, ++s) {
      const bool st_ = !LCH || (int)l >= lb;  // (block-uniform) false: the layer above this block's range, nothing is stored
      // this layer's values into locals, then request the following layers' inputs
      const Float f0 = w0.fm[0].x, f1 = w0.fm[0].y, f2 = w0.fm[1].x, f3 = w0.fm[1].y, f4 = w0.fm[2].x, f5 = w0.fm[2].y,
                  f6 = w0.fm[3].x, f7 = w0.fm[3].y;
      const int je1 = w0.je1, je2 = w0.je2, jT = x0.jT, jp = x0.jpress + (x0.tropo ? 0 : 1) + 1;  // levels jp-1, jp
      const Float tl = x0.tlay, tv = x0.tlev;
      x0 = x1;
      // unconditional (the last layers repeat the last one): a request made on some paths only makes the number of
      // outstanding memory operations path-dependent, and the compiler then drains them all -- this layer's requests
      // and the previous layer's 32 stores -- in front of every barrier
      load_wts(ic, min(l + 1, nlay - 1), x0, w0);
      load_idx(ic, min(l + 2, nlay - 1), x1);
      const int Tmin = gl[l][0], nT = gl[l][1], Pmin = gl[l][2], emin = gl[l][4], nE = gl[l][5];
      const Float pl_lay = planck(tl), pl_lev = planck(tv);
      __syncthreads();  // B(s): slab(s) is complete
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
      if constexpr (FACT) {
        if (g0 == gptS && st_) {  // (block-uniform; unconditional across lanes like the other stores)
          store_stream(reinterpret_cast<Float*>(reinterpret_cast<char*>(a.plk_lay + (size_t)ncl * ibnd) + olay), pl_lay);
          store_stream(reinterpret_cast<Float*>(reinterpret_cast<char*>(a.plk_lev + (size_t)nclv * ibnd) + olay), pl_lev);
        }
      }
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
          // lanes past the last column repeat it (ic is clamped) and store the same values to the same
          // addresses: unconditional stores keep the number of outstanding memory operations static, so the
          // wait for the next layer's weights is a counted one instead of a drain of these stores
          if constexpr (FACT) {
            if (st_) store_stream(reinterpret_cast<Float*>(play_ + slay * j + olay), pf);
          } else {
            const Float vlay = pf * pl_lay;                                      // :674
            const Float vlev = (l == 0 ? pf : sqrt(prev[j] * pf)) * pl_lev;      // :695,:699
            if (st_) {
#if !defined(PLANCKX_NOLAY)
              store_stream(reinterpret_cast<Float*>(play_ + slay * j + olay), vlay);
#endif
#if !defined(PLANCKX_NOLEV)
              store_stream(reinterpret_cast<Float*>(plev_ + slev * j + olay), vlev);  // level l of (ncol, nlay+1): same column offset
#endif
            }
          }
          prev[j] = pf;
        }
        asm volatile("" : "+v"(prev[jj]), "+v"(prev[jj + 1]));  // keep the pair's arithmetic here
        __builtin_amdgcn_sched_barrier(0);   // at most 8 row reads (32 VGPRs) in flight
      }
    }
    if (!lastc) continue;  // (block-uniform)
    // (the column index of this epilogue is made opaque: its 64-bit addresses -- top level, surface layer -- are formed here,
    //  once per chunk, instead of being hoisted out of the chunk loop and spilled across the layer walk)
    unsigned ice = ic;
    asm volatile("" : "+v"(ice));
    if (valid) {
      const Float pl_top = planck(a.tlev[ice + ncol * nlay]);
      if constexpr (FACT) {
        if (g0 == gptS) a.plk_lev[ice + ncol * nlay + (size_t)nclv * ibnd] = pl_top;
      } else {
#pragma unroll
        for (int j = 0; j < G; ++j) a.lev_src[ice + ncol * nlay + (size_t)nclv * (g0 + j)] = prev[j] * pl_top;  // :705
      }
    }
    // ---- surface source (:651-653) from the Planck fractions of the surface layer
    if (lsfc != (int)nlay - 1) {
      {  // (load_idx / load_wts on the opaque column index)
        const unsigned cl = ice + ncol * (unsigned)lsfc;
        x0.tropo = a.tropo[cl]; x0.jT = a.jtemp[cl]; x0.jpress = a.jpress[cl]; x0.tlay = a.tlay[cl]; x0.tlev = a.tlev[cl];
        const size_t clf = cl + (size_t)ncl * (x0.tropo ? flav0 : flav1);
        const Float2* fmp = reinterpret_cast<const Float2*>(a.fmajor + 8 * clf);
#pragma unroll
        for (int i = 0; i < 4; ++i) w0.fm[i] = fmp[i];
        const int2 je = *reinterpret_cast<const int2*>(a.jeta + 2 * clf);
        w0.je1 = je.x; w0.je2 = je.y;
      }
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
      const Float tsfc = a.tsfc[ice];
      const Float pl_sfc = planck(tsfc);
      const Float pl_sfc1 = planck(tsfc + (Float)1);
#pragma unroll
      for (int j = 0; j < G; ++j) {
        a.sfc_src[ice + (size_t)ncol * (g0 + j)] = prev[j] * pl_sfc;
        a.sfc_jac[ice + (size_t)ncol * (g0 + j)] = prev[j] * (pl_sfc1 - pl_sfc);
      }
    }
  }
}



// factored -> lay_source / lev_source (one thread per (column, level, band), the band's g-points in a loop)
__global__ void __launch_bounds__(256)
expand_factored_sources_kernel(int ncol, int nlay, const int* __restrict__ band_lims, const Float* pfrac,
                               const Float* __restrict__ plk_lay, const Float* __restrict__ plk_lev, Float* lay_src,
                               Float* __restrict__ lev_src, int what /* 3: both, 1: lev_src only, 2: lay_src only (in place: pfrac == lay_src, 1 then 2) */) {
  const int icol = blockIdx.x * blockDim.x + threadIdx.x, ilev = blockIdx.y, ibnd = blockIdx.z;
  if (icol >= ncol) return;
  const size_t ncl = (size_t)ncol * nlay, nclv = (size_t)ncol * (nlay + 1);
  const int gS = band_lims[2 * ibnd] - 1, gE = band_lims[2 * ibnd + 1] - 1;
  const Float pl_lev = plk_lev[icol + (size_t)ncol * ilev + nclv * ibnd];
  const Float pl_lay = ilev < nlay ? plk_lay[icol + (size_t)ncol * ilev + ncl * ibnd] : (Float)0;
  for (int g = gS; g <= gE; ++g) {
    const Float* pf = pfrac + ncl * (size_t)g + icol;
    const Float below = pf[(size_t)ncol * min(ilev, nlay - 1)];
    if ((what & 1) != 0) {
      const Float f = (ilev == 0 || ilev == nlay) ? below : sqrt(pf[(size_t)ncol * (ilev - 1)] * below);   // :695, :699, :705
      lev_src[icol + (size_t)ncol * ilev + nclv * g] = f * pl_lev;
    }
    if ((what & 2) != 0 && ilev < nlay) lay_src[icol + (size_t)ncol * ilev + ncl * g] = below * pl_lay;   // :674
  }
}

}  // namespace

// deferred sources: lay_source / lev_source from the factors the record names, in place (runtime.hip calls this with the
// context held, possibly from inside another entry point's Call: launches only)
constexpr int kPlanckDeferSlot = 12;
static void planck_expand_pending(const rte::PendingSources& s) {
  rte::ProfScope p("expand_factored_sources_kernel");
  const dim3 grid(cdiv(s.ncol, 256), s.nlay + 1, s.nbnd);
  Float* lay = (Float*)const_cast<void*>(s.lay);
  Float* lev = (Float*)const_cast<void*>(s.lev);
  hipLaunchKernelGGL(expand_factored_sources_kernel, grid, dim3(256), 0, rte::stream(), s.ncol, s.nlay, s.band_lims, (const Float*)lay,
                     (const Float*)s.plk_lay, (const Float*)s.plk_lev, lay, lev, 1);
  hipLaunchKernelGGL(expand_factored_sources_kernel, grid, dim3(256), 0, rte::stream(), s.ncol, s.nlay, s.band_lims, (const Float*)lay,
                     (const Float*)s.plk_lay, (const Float*)s.plk_lev, lay, lev, 2);
}

// the fingerprint of a consumed record (common.h): 64 values spread over the fraction array; mode 0 keeps them, mode 1
// compares bit for bit and raises the flag behind them
__device__ __forceinline__ bool same_bits(double a, double b) { return __double_as_longlong(a) == __double_as_longlong(b); }
__device__ __forceinline__ bool same_bits(float a, float b) { return __float_as_int(a) == __float_as_int(b); }
__global__ void __launch_bounds__(64) sources_fingerprint_kernel(const Float* __restrict__ lay, size_t n, Float* sample, int mode) {
  const size_t idx = (size_t)((double)(n - 1) * (double)threadIdx.x / 63.0);
  const Float v = lay[idx];
  if (mode == 0) {
    sample[threadIdx.x] = v;
  } else if (!same_bits(v, sample[threadIdx.x])) {
    atomicOr(reinterpret_cast<int*>(sample + 64), 1);
  }
}
static void planck_take_sample(const rte::PendingSources& s) {
  hipLaunchKernelGGL(sources_fingerprint_kernel, dim3(1), dim3(64), 0, rte::stream(), (const Float*)s.lay,
                     (size_t)s.ncol * s.nlay * s.ngpt, (Float*)s.sample, 0);
}
static bool planck_still_factored(const rte::PendingSources& s) {
  int* flag = reinterpret_cast<int*>((Float*)s.sample + 64);
  HIP_CHECK(hipMemsetAsync(flag, 0, sizeof(int), rte::stream()));
  hipLaunchKernelGGL(sources_fingerprint_kernel, dim3(1), dim3(64), 0, rte::stream(), (const Float*)s.lay,
                     (size_t)s.ncol * s.nlay * s.ngpt, (Float*)s.sample, 1);
  int h = 1;
  HIP_CHECK(hipMemcpyAsync(&h, flag, sizeof(int), hipMemcpyDeviceToHost, rte::stream()));
  HIP_CHECK(hipStreamSynchronize(rte::stream()));
  return h == 0;
}
static const rte::PendingSourcesOps kPlanckSourcesOps{planck_expand_pending, planck_take_sample, planck_still_factored};

// the body of rrtmgp_compute_Planck_source and of its factored form (plk_lay != nullptr: see PlanckArgs)
static void planck_source_impl(const char* name, int ncol, int nlay, int nbnd, int ngpt, int nflav, int neta, int npres, int ntemp,
                               int nPlanckTemp, const Float* tlay, const Float* tlev, const Float* tsfc, int sfc_lay,
                               const Float* fmajor, const int* jeta, const Bool* tropo, const int* jtemp, const int* jpress,
                               const int* band_lims_gpt, const Float* pfracin, Float temp_ref_min_v, Float totplnk_delta_v,
                               const Float* totplnk, const int* gpoint_flavor, Float* sfc_src, Float* lay_src, Float* lev_src,
                               Float* sfc_source_Jac, Float* plk_lay, Float* plk_lev, rte::PendingSources* deferred = nullptr) {
  const int* sfc_lay_ = &sfc_lay;
  const Float *temp_ref_min = &temp_ref_min_v, *totplnk_delta = &totplnk_delta_v;
  if (ncol <= 0 || nlay <= 0 || ngpt <= 0) return;
  rte::Call c(name);
  const size_t ncl = (size_t)ncol * nlay;
  // Deferred sources (rte_hip_defer_sources / RTE_HIP_DEFER_SOURCES=1; runtime.hip): the factored form into the caller's own
  // arrays -- the Planck fraction where lay_source goes, nothing to lev_source, the bands' Planck functions into a library
  // buffer -- and a record of it; rte_lw_solver_noscat on these arrays solves from the factors, anything else the library
  // is handed them for finds them expanded first (planck_expand_pending).
  Float* lev_src_deferred = nullptr;
  int* bl_dev = nullptr;
  if (deferred) {
    const size_t nclv = (size_t)ncol * (nlay + 1);
    (void)c.out(lev_src, nclv * ngpt);  // (an earlier record on this array goes)
    // (one buffer for every record's factors: records on OTHER arrays were expanded by the entry point before this call,
    //  rte::flush_pending_sources_except, so growing or overwriting it takes nothing from a record that is still alive)
    plk_lay = (Float*)rte::persistent(kPlanckDeferSlot, sizeof(Float) * ((ncl + nclv) * nbnd + 64 + 2) + sizeof(int) * 2 * nbnd, nullptr);
    plk_lev = plk_lay + ncl * nbnd;
    Float* sample = plk_lev + nclv * nbnd;  // 64 values + the compare flag
    bl_dev = (int*)(sample + 64 + 2);
    HIP_CHECK(hipMemcpyAsync(bl_dev, band_lims_gpt, sizeof(int) * 2 * nbnd, hipMemcpyDefault, rte::stream()));
    if (!rte::is_device_pointer(band_lims_gpt)) HIP_CHECK(hipStreamSynchronize(rte::stream()));  // (pageable source)
    lev_src_deferred = lev_src;
    lev_src = nullptr;
    // (recorded by the entry point once this call has staged its arguments: Call::out on lay_source drops records on it)
    *deferred = rte::PendingSources{lay_src, lev_src_deferred, ncol, nlay, nbnd, ngpt, plk_lay, plk_lev, bl_dev, sample, false};
  }
  const bool factored = plk_lay != nullptr;
  const Float* d_tlay = c.in(tlay, ncl);
  const Float* d_tlev = c.in(tlev, (size_t)ncol * (nlay + 1));
  const Float* d_tsfc = c.in(tsfc, (size_t)ncol);
  const Float* d_fmajor = c.in(fmajor, 8 * ncl * nflav);
  const int* d_jeta = c.in(jeta, 2 * ncl * nflav);
  const Bool* d_tropo = c.in(tropo, ncl);
  const int* d_jtemp = c.in(jtemp, ncl);
  const int* d_jpress = c.in(jpress, ncl);
  const int* d_band_lims = c.in(band_lims_gpt, (size_t)2 * nbnd);
  const Float* d_pfracin = c.in_table(pfracin, (size_t)ntemp * neta * (npres + 1) * ngpt);
  const Float* d_totplnk = c.in(totplnk, (size_t)nPlanckTemp * nbnd);
  const int* d_gpoint_flavor = c.in(gpoint_flavor, (size_t)2 * ngpt);
  Float* d_sfc_src = c.out_lazy(sfc_src, (size_t)ncol * ngpt);  // (lazy: host-mirror mode keeps the sources on the device)
  Float* d_lay_src = c.out_lazy(lay_src, ncl * ngpt);
  Float* d_lev_src = factored ? nullptr : c.out_lazy(lev_src, (size_t)ncol * (nlay + 1) * ngpt);
  Float* d_sfc_jac = c.out_lazy(sfc_source_Jac, (size_t)ncol * ngpt);
  Float* d_plk_lay = factored ? c.out_lazy(plk_lay, ncl * nbnd) : nullptr;
  Float* d_plk_lev = factored ? c.out_lazy(plk_lev, (size_t)ncol * (nlay + 1) * nbnd) : nullptr;
  const Float totplnk_delta_r = (Float)1 / *totplnk_delta;  // :636
  {
    const void* outs[4] = {d_sfc_src, d_lay_src, d_lev_src, d_sfc_jac};
    const size_t ob[4] = {sizeof(Float) * (size_t)ncol * ngpt, sizeof(Float) * ncl * ngpt,
                          sizeof(Float) * (size_t)ncol * (nlay + 1) * ngpt, sizeof(Float) * (size_t)ncol * ngpt};
    if (!factored) c.try_fork(outs, ob, 4);  // opt-in: concurrently with the compute_tau_absorption call this one follows
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
#if defined(PLANCKX_GW8)  // (experiment builds: stages of 8 g-points whatever the bands allow)
    bl_gw = aligned(8) ? 8 : 0;
#else
    bl_gw = aligned(16) ? 16 : (aligned(8) ? 8 : 0);  // g-points per stage of the production kernel
#endif
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
  q.factored = factored ? 1 : 0; q.plk_lay = d_plk_lay; q.plk_lev = d_plk_lev;
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
  v.plk_lay = d_plk_lay; v.plk_lev = d_plk_lev;
  v.skip_if = guard;
  v.worklist = worklist;
  int wl_tile = BS;
  // the production kernel: at most 256 layers (per-layer flags in LDS), the bit-mask geometry pre-pass (table dimensions
  // within its mask words), 32-bit byte offsets into a g-point plane; otherwise the direct kernel
  const bool planck9 = nlay <= 256 && nbnd <= MAXB && (size_t)ncol * (nlay + 1) < ((size_t)1 << 29) && nflav <= MAXFLAV &&
                       neta < 31 && ntemp < 31 && npres + 1 < 63;
  if (!planck9) {
    rte::ProfScope p("planck_source_kernel");
    hipLaunchKernelGGL(planck_source_kernel, dim3(cdiv(ncol, 256), nbnd), dim3(256), 0, st, q, (const int*)nullptr);
    return;
  }
  {
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
    Geom2Args ga{};
    ga.ncol = ncol; ga.nlay = nlay; ga.nbnd = nbnd; ga.nflav = nflav; ga.slab_floats = SLAB9; ga.planck = true;
    ga.jeta = d_jeta; ga.jtemp = d_jtemp; ga.jpress = d_jpress; ga.tropo = d_tropo; ga.band_lims = d_band_lims;
    ga.gpoint_flavor = d_gpoint_flavor; ga.worklist = v.worklist; ga.flags = d_flags; ga.skip_if = guard;
    ga.skip_if2 = shared ? gs().shared.valid : nullptr;  // (set on the device by that call's geometry kernel, if it ran)
    // few (tile, band) pairs (calls of some thousand columns): the layers of a pair go to several blocks, at least 8 each,
    // until about two blocks per CU are in the grid (planck_source_v9_kernel, LCH)
    const int lchunks = (int)std::max(1u, std::min((unsigned)nlay / 8u, RTE_SMALL_GRID_BLOCKS / std::max(1u, tiles * (unsigned)nbnd)));
#define RTE_LAUNCH_PLANCK9(GW)                                                                                    \
  do {                                                                                                            \
    {                                                                                                             \
      rte::ProfScope p("planck_source_setup");                                                                    \
      if (shared) hipLaunchKernelGGL(planck_flags_kernel, dim3(cdiv(tiles * nbnd, 4)), dim3(256), 0, st, (const TileGeom*)d_geom, \
                                     (int)tiles, nlay, nbnd, SLAB9, GW + 2, d_flags, v.worklist, (const int*)gs().shared.valid, \
                                     (const int*)guard);                                                          \
      hipLaunchKernelGGL((tile_geom2_kernel<NCW * 64, GW>), dim3(tiles, nlay), dim3(NCW * 64), 0, st, ga, d_geom); \
    }                                                                                                             \
    rte::ProfScope p(factored ? "planck_source_factored_kernel" : "planck_source_kernel");                        \
    if (lchunks > 1) {                                                                                            \
      const dim3 gch(nbnd * tiles, lchunks);                                                                      \
      if (factored)                                                                                               \
        hipLaunchKernelGGL((planck_source_v9_kernel<NCW, NLW, SLAB9, GW, true, true>), gch, dim3((NCW + NLW) * 64), \
                           sizeof(Float) * nPlanckTemp, st, v, nbnd, tiles, (const TileGeom*)d_geom, (const int*)d_flags); \
      else                                                                                                        \
        hipLaunchKernelGGL((planck_source_v9_kernel<NCW, NLW, SLAB9, GW, false, true>), gch, dim3((NCW + NLW) * 64), \
                           sizeof(Float) * nPlanckTemp, st, v, nbnd, tiles, (const TileGeom*)d_geom, (const int*)d_flags); \
    } else                                                                                                        \
    if (factored)                                                                                                 \
      hipLaunchKernelGGL((planck_source_v9_kernel<NCW, NLW, SLAB9, GW, true>), dim3(nbnd * 8 * cdiv(tiles, 8)),  \
                         dim3((NCW + NLW) * 64), sizeof(Float) * nPlanckTemp, st, v, nbnd, tiles,                 \
                         (const TileGeom*)d_geom, (const int*)d_flags);                                           \
    else                                                                                                          \
    hipLaunchKernelGGL((planck_source_v9_kernel<NCW, NLW, SLAB9, GW>), dim3(nbnd * 8 * cdiv(tiles, 8)),          \
                       dim3((NCW + NLW) * 64), sizeof(Float) * nPlanckTemp, st, v, nbnd, tiles,                   \
                       (const TileGeom*)d_geom, (const int*)d_flags);                                             \
  } while (0)
    if (bl_gw == 16) RTE_LAUNCH_PLANCK9(16); else RTE_LAUNCH_PLANCK9(8);
#undef RTE_LAUNCH_PLANCK9
  }
  {
    // (tile, band) pairs whose pfrac bounding box exceeded the LDS slab at some layer
    rte::ProfScope p("planck_source_fallback");
    hipLaunchKernelGGL(planck_source_worklist_kernel, dim3(1024), dim3(256), 0, st, q, (const int*)v.worklist, wl_tile,
                       stats_dev() + 1);
    // the whole call on the direct kernel if the guard fired
    hipLaunchKernelGGL(planck_source_kernel, dim3(cdiv(ncol, 256), nbnd), dim3(256), 0, st, q, (const int*)guard);
  }
}

extern "C" {
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
  (void)gpoint_bands;
  RTE_TRY
  rte::PendingSources rec{};
  const bool deferred = rte::defer_sources_enabled() && rte::is_device_memory(lay_src) && rte::is_device_memory(lev_src) && *nlay_ <= 80 &&
                        (size_t)*ncol_ * (*nlay_ + 1) < ((size_t)1 << 29) && *nbnd_ > 0;
  if (deferred) rte::flush_pending_sources_except(lay_src, lev_src);  // (the factors of every record share one buffer)
  planck_source_impl("rrtmgp_compute_Planck_source", *ncol_, *nlay_, *nbnd_, *ngpt_, *nflav_, *neta_, *npres_, *ntemp_, *nPlanckTemp_,
                     tlay, tlev, tsfc, *sfc_lay_, fmajor, jeta, tropo, jtemp, jpress, band_lims_gpt, pfracin, *temp_ref_min,
                     *totplnk_delta, totplnk, gpoint_flavor, sfc_src, lay_src, lev_src, sfc_source_Jac, nullptr, nullptr, deferred ? &rec : nullptr);
  if (deferred && rec.lay) rte::defer_sources(rec, &kPlanckSourcesOps);
  RTE_CATCH("rrtmgp_compute_Planck_source")
}

// compute_Planck_source with FACTORED output (extension): the source is the product of the Planck fraction of the g-point and the
// Planck function of its band (:674), and the level source the geometric mean of the neighbouring layers' fractions times the band's
// function at the level temperature (:695-705) -- this entry stores the factors: pfrac (ncol, nlay, ngpt), planck_lay (ncol, nlay, nbnd),
// planck_lev (ncol, nlay+1, nbnd); sfc_src and sfc_source_Jac as in the ABI call.  rte_hip_lw_solver_noscat_factored forms the same
// products per g-point; rte_hip_expand_factored_sources writes lay_source / lev_source from the factors for any other consumer.
// 12.9 + 1.5 GB instead of 26 GB written at 1e5 x 60 x 256.
int rte_hip_compute_Planck_source_factored(int ncol, int nlay, int nbnd, int ngpt, int nflav, int neta, int npres, int ntemp,
                                           int nPlanckTemp, const Float* tlay, const Float* tlev, const Float* tsfc, int sfc_lay,
                                           const Float* fmajor, const int* jeta, const Bool* tropo, const int* jtemp,
                                           const int* jpress, const int* band_lims_gpt, const Float* pfracin, double temp_ref_min,
                                           double totplnk_delta, const Float* totplnk, const int* gpoint_flavor, Float* sfc_src,
                                           Float* pfrac, Float* planck_lay, Float* planck_lev, Float* sfc_source_Jac) {
  if (!planck_lay || !planck_lev || !pfrac) return -1;
  RTE_TRY
  planck_source_impl("rte_hip_compute_Planck_source_factored", ncol, nlay, nbnd, ngpt, nflav, neta, npres, ntemp, nPlanckTemp, tlay, tlev,
                     tsfc, sfc_lay, fmajor, jeta, tropo, jtemp, jpress, band_lims_gpt, pfracin, (Float)temp_ref_min, (Float)totplnk_delta, totplnk,
                     gpoint_flavor, sfc_src, pfrac, nullptr, sfc_source_Jac, planck_lay, planck_lev);
  return 0;
  RTE_CATCH("rte_hip_compute_Planck_source_factored")
  return -1;
}

// lay_source (ncol, nlay, ngpt) and lev_source (ncol, nlay+1, ngpt) from the factored form, with the operations of
// compute_Planck_source (bit-identical to its output)
int rte_hip_expand_factored_sources(int ncol, int nlay, int nbnd, int ngpt, const int* band_lims_gpt, const Float* pfrac,
                                    const Float* planck_lay, const Float* planck_lev, Float* lay_source, Float* lev_source) {
  if (ncol <= 0 || nlay <= 0 || ngpt <= 0) return 0;
  RTE_TRY
  rte::Call c("rte_hip_expand_factored_sources");
  const size_t ncl = (size_t)ncol * nlay, nclv = (size_t)ncol * (nlay + 1);
  const int* d_bl = c.in(band_lims_gpt, (size_t)2 * nbnd);
  const Float *d_pf = c.in(pfrac, ncl * ngpt), *d_ply = c.in(planck_lay, ncl * nbnd), *d_plv = c.in(planck_lev, nclv * nbnd);
  Float *d_lay = c.out(lay_source, ncl * ngpt), *d_lev = c.out(lev_source, nclv * ngpt);
  rte::ProfScope p("expand_factored_sources_kernel");
  const dim3 grid(cdiv(ncol, 256), nlay + 1, nbnd);
  if (d_pf == d_lay) {  // in place: every level source first (they read the neighbouring layers' fractions), then the layer sources
    hipLaunchKernelGGL(expand_factored_sources_kernel, grid, dim3(256), 0, rte::stream(), ncol, nlay, d_bl, d_pf, d_ply, d_plv, d_lay, d_lev, 1);
    hipLaunchKernelGGL(expand_factored_sources_kernel, grid, dim3(256), 0, rte::stream(), ncol, nlay, d_bl, d_pf, d_ply, d_plv, d_lay, d_lev, 2);
  } else {
    hipLaunchKernelGGL(expand_factored_sources_kernel, grid, dim3(256), 0, rte::stream(), ncol, nlay, d_bl, d_pf, d_ply, d_plv, d_lay, d_lev, 3);
  }
  return 0;
  RTE_CATCH("rte_hip_expand_factored_sources")
  return -1;
}

}  // extern "C"

