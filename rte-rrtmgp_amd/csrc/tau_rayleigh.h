// tau_rayleigh.h -- rrtmgp_compute_tau_rayleigh (reference rrtmgp/kernels/mo_gas_optics_rrtmgp_kernels.F90:506-565), optionally
// with combine_abs_and_rayleigh (rrtmgp/frontend/mo_gas_optics_rrtmgp.F90:1954-2036) applied in place: the direct kernel and
// the slab kernel for tables whose bands are whole aligned chunks of 16 or 8 g-points.
#pragma once
#include "gas_optics_common.h"

namespace {
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
