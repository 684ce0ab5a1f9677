// tau_mx.h -- compute_tau_absorption with the LUT gathers on the matrix cores ("mx"; double precision, 16-wide stages).
// Included by tau_absorption.hip (after gas_optics_common.h).  Reference semantics:
// rrtmgp/kernels/mo_gas_optics_rrtmgp_kernels.F90:345-396 (major species, interpolate3D_byflav :765-803) and
// :402-501 (minor species, interpolate2D_byflav :741-763).
//
// Why.  With lanes = columns (tau_absorption_v9_kernel) every column fetches its own 8 + 4-per-minor-interval corner
// rows of 16 g-points from the LDS slab: 192 bytes of LDS reads per (column, g-point), 295 GB per launch at
// 1e5 x 60 x 256, and the LDS is the busiest unit of that kernel (profiles/r03e_lw_sq_summary.csv).  But the rows a
// column needs are determined by its KEY (jtemp, jpress + itropo, jeta(1), jeta(2)) of the band's flavor, and a tile
// of 512 columns holds only 14 (benchmark atmosphere) to 23 (site-like, unordered) distinct keys per band
// (DESIGN.md section 4.2b).  For columns of one key
//     tau(col, g) = sum_k W(col, k) K(k, g),   k = 8 major corners + 4 per minor interval,
// is a small DENSE product with K shared by the columns -- what v_mfma_f64_16x16x4_f64 computes for 16 columns x 16
// g-points x 4 corners per instruction, with ONE 8-byte operand per lane for W and for K.  fp64 MFMA has the rate
// of the fp64 vector FMA on this chip (tools/mfma_f64_bench.hip: 64 cycles per instruction and SIMD, 70 TFLOP/s); the
// point is not flops but operand delivery: 16 B of LDS traffic per (column, g-point) instead of 192.
//
//   * tau_mx_sort_kernel (pre-pass): per (512-column tile, layer, FLAVOR) the tile's columns sorted by key (counting
//     sort on a dense code; regime is the leading key part, so the sort of the band's lower flavor serves the lower
//     columns and that of its upper flavor the upper ones).  Output: 4 bytes per (column, layer, flavor): column | key << 9.
//   * tau_absorption_mx_kernel: block = (512-column tile, layer) = 8 COLUMN waves + 8 MATRIX waves, one barrier per
//     stage (16 g-points of a band), a double-buffered LDS tile of one 128-byte row per column:
//       column waves (lanes = columns; all global traffic is coalesced as before): write the stage's weight row
//         [8 major (col_mix folded in) | 4 fminor | 4 minor scalings]; a stage later read the row
//         back -- now holding tau of the stage's 16 g-points -- and store it (non-temporal, 512 B per wave and plane);
//       matrix waves: 16 sorted positions at a time, per distinct key among them: K rows from the g-fastest tables
//         (L2 / Infinity Cache; one 8-byte load per lane and 4 rows), A = the columns' weights gathered from their
//         LDS rows and masked to the key, 2 + (minor intervals) MFMAs into one accumulator; the result overwrites
//         the 16 rows in place.
//     Rows belong to columns, not to sorted positions, so a column wave touches only its own rows: no hazard between
//     reading tau(s-1) and writing W(s+1) into the same buffer, and no LDS slab, bounding box or worklist -- any
//     atmosphere runs on this kernel, the number of distinct keys only changes the MFMA count.
//   * more than 4 minor intervals in a band and regime: further sub-stages of the same g-points with the next 4
//     scalings; the column waves sum the sub-stages' rows in registers.
// Arithmetic: an MFMA is a k-ordered chain of fp64 FMAs (bit-identical to the host's fma chain, checked by
// tools/mfma_f64_bench.hip); the minor weights fminor x scaling are rounded once more than in the reference
// association -- a few ulp, like the FMA forms of the v9 kernel (tests: 1e-12).
#pragma once
#ifndef RTE_USE_SP

namespace {

typedef double mx_v4d __attribute__((ext_vector_type(4)));
constexpr int MX_NB = 4096;   // bins of one counting-sort pass
constexpr int MX_MINOR = 4;   // minor intervals per sub-stage

// element e (0..15) of column row `row`: 16-byte granules XOR-swizzled by the row so that the 8 lanes of a b128 group
// (consecutive rows, same logical granule) and the 16 rows an MFMA operand gathers spread over the banks
__device__ __forceinline__ int mx_elem(int row, int e) { return row * 16 + ((((e >> 1) ^ (row & 7)) << 1) | (e & 1)); }

// key of a column, 20 bits: jT < 32, jp = jpress + itropo + 1 < 64, eta indices < 16 (checked by the host)
__device__ __forceinline__ unsigned mx_key(int jT, int jp, int itropo, int e1, int e2) {
  return (unsigned)jT | ((unsigned)jp << 5) | ((unsigned)itropo << 11) | ((unsigned)e1 << 12) | ((unsigned)e2 << 16);
}

// inclusive scan over the block of one value per thread (TILE threads); returns the exclusive prefix, *total = block sum.
// Two barriers; `wsum` is LDS scratch of TILE / 64 ints.
template <int TILE>
__device__ __forceinline__ int mx_block_scan(int v, int* wsum, int* total) {
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  int incl = v;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    const int u = __shfl_up(incl, o);
    if (lane >= o) incl += u;
  }
  if (lane == 63) wsum[w] = incl;
  __syncthreads();
  int off = 0, tot = 0;
#pragma unroll
  for (int k = 0; k < TILE / 64; ++k) {
    const int s = wsum[k];
    if (k < w) off += s;
    tot += s;
  }
  *total = tot;
  return off + incl - v;
}

template <int TILE>
__global__ void __launch_bounds__(TILE)
tau_mx_sort_kernel(int ncol, int nlay, int nflav, int neta, const int* __restrict__ jtemp, const int* __restrict__ jpress,
                   const Bool* __restrict__ tropo, const int* __restrict__ jeta, const int* __restrict__ skip_if,
                   const int* __restrict__ skip_if2,
                   unsigned* __restrict__ sort_pk /*[tile][lay][flav][TILE]: column | key << 9*/, int* __restrict__ n_lo_out /*[tile][lay]*/) {
  constexpr int NW = TILE / 64, BPT = MX_NB / TILE;
  static_assert(MX_NB % TILE == 0, "bins per thread");
  __shared__ int bins[2][MX_NB];
  __shared__ int red[5];
  __shared__ int wsum[2][NW];
  if (*skip_if || *skip_if2) return;
  const int tid = threadIdx.x, lane = tid & 63;
  const unsigned ilay = blockIdx.y;
  const unsigned ic = min(blockIdx.x * TILE + (unsigned)tid, (unsigned)ncol - 1u);  // columns past the end repeat the last one
  const size_t ncl = (size_t)ncol * nlay;
  const size_t cl = ic + (size_t)ncol * ilay;
  const int itropo = tropo[cl] ? 0 : 1;
  const int jT = jtemp[cl], jp = jpress[cl] + itropo;
  int2 je = *reinterpret_cast<const int2*>(jeta + 2 * cl);
  if (tid == 0) { red[0] = 1 << 30; red[1] = -1; red[2] = 1 << 30; red[3] = -1; red[4] = 0; }
  for (int i = tid; i < 2 * MX_NB; i += TILE) (&bins[0][0])[i] = 0;
  __syncthreads();
  {
    const int a0 = wave_min(jT), a1 = wave_max(jT), a2 = wave_min(jp), a3 = wave_max(jp);
    const int nl = __popcll(__ballot(itropo == 0));
    if (lane == 0) {
      atomicMin(&red[0], a0); atomicMax(&red[1], a1); atomicMin(&red[2], a2); atomicMax(&red[3], a3);
      atomicAdd(&red[4], nl);
    }
  }
  __syncthreads();
  const int Tmin = red[0], NT = red[1] - red[0] + 1, Pmin = red[2], NP = red[3] - red[2] + 1;
  if (tid == 0) n_lo_out[blockIdx.x + gridDim.x * ilay] = red[4];
  // dense rank of (regime, temperature row, pressure row) among the combinations present in the tile
  // (2 NT NP <= MX_NB: checked by the host from the table dimensions)
  const int bkey = (itropo * NT + (jT - Tmin)) * NP + (jp - Pmin);
  bins[1][bkey] = 1;
  __syncthreads();
  int base_id, nbase;
  {
    int loc[BPT], tot = 0;
#pragma unroll
    for (int j = 0; j < BPT; ++j) { loc[j] = tot; tot += bins[1][BPT * tid + j]; }
    const int ex = mx_block_scan<TILE>(tot, wsum[0], &nbase);
    __syncthreads();  // everybody has read the presence flags
#pragma unroll
    for (int j = 0; j < BPT; ++j) bins[1][BPT * tid + j] = ex + loc[j];
    __syncthreads();
    base_id = bins[1][bkey];
    __syncthreads();
#pragma unroll
    for (int j = 0; j < BPT; ++j) bins[1][BPT * tid + j] = 0;
    __syncthreads();
  }
  const int NE = neta;  // eta indices are 1 .. neta - 1
  const int nbins = nbase * NE * NE;
  unsigned* const out = sort_pk + (size_t)(blockIdx.x + gridDim.x * ilay) * nflav * TILE;
  static_assert(TILE <= 512, "9 bits for the column");
  int it = 0;
  for (int f = 0; f < nflav; ++f) {
    const int2 je_next = *reinterpret_cast<const int2*>(jeta + 2 * (cl + ncl * (size_t)min(f + 1, nflav - 1)));
    const int code = (base_id * NE + min(max(je.x, 0), NE - 1)) * NE + min(max(je.y, 0), NE - 1);
    int base = 0;  // positions taken by earlier passes
    for (int win = 0; win < nbins; win += MX_NB, ++it) {
      int* const B = bins[it & 1];
      int* const Z = bins[(it + 1) & 1];
      const bool in = code >= win && code < win + MX_NB;
      int r = 0;
      if (in) r = atomicAdd(&B[code - win], 1);  // arrival order inside a key: any order is as good as another
      __syncthreads();
      int loc[BPT], tot = 0;
#pragma unroll
      for (int j = 0; j < BPT; ++j) { loc[j] = tot; tot += B[BPT * tid + j]; Z[BPT * tid + j] = 0; }
      int total;
      const int ex = mx_block_scan<TILE>(tot, wsum[it & 1], &total);
#pragma unroll
      for (int j = 0; j < BPT; ++j) B[BPT * tid + j] = base + ex + loc[j];
      __syncthreads();
      if (in) out[(size_t)f * TILE + B[code - win] + r] = (unsigned)tid | (mx_key(jT, jp + 1, itropo, je.x, je.y) << 9);
      base += total;
    }
    je = je_next;
  }
}

#ifdef MX_TIMING
// experiment builds only (tools/fastbuild.py ...:tau_absorption.hip=-DMX_TIMING): clocks the waves of each role spend
// working between two barriers, and waiting at them; rte_hip_mx_timing() reads and clears
__device__ unsigned long long mx_clk[2][2];
#define MX_T0() long long t_prev_ = __builtin_amdgcn_s_memtime(); unsigned long long busy_ = 0, wait_ = 0
#define MX_ARRIVE() const long long t_arr_ = __builtin_amdgcn_s_memtime(); busy_ += (unsigned long long)(t_arr_ - t_prev_)
#define MX_LEAVE() t_prev_ = __builtin_amdgcn_s_memtime(); wait_ += (unsigned long long)(t_prev_ - t_arr_)
#define MX_TEND(role) if (lane == 0) { atomicAdd(&mx_clk[role][0], busy_); atomicAdd(&mx_clk[role][1], wait_); }
#else
#define MX_T0()
#define MX_ARRIVE()
#define MX_LEAVE()
#define MX_TEND(role)
#endif

struct MxArgs {
  int ncol, nlay, ngpt, nbnd, ntemp, TE, idx_h2o, nk_lo, nk_up, nflav;
  const BandMeta* bmeta;
  const Float *kmaj, *klo, *kup;  // g-fastest tables
  const int *jeta, *jtemp, *jpress;
  const Bool* tropo;
  const Float *col_mix, *fmajor, *fminor, *play, *tlay, *col_gas;
  Float* tau;
  const Float* add_bybnd;
  const int *skip_if, *skip_if2;  // overlapping regimes / a stale plan; layer ranges that are not those of the tropo flags
  const unsigned* sort_pk;      // per (tile, layer, flavor): the tile's columns sorted by key, column | key << 9
  const MxStageRec* stages;     // the stage list (plan)
  int nstage;
  const int* n_lo;
  int* stat;  // rte_hip_stat(3) = 10: this kernel did the call
};

// the stage sequence: bands in order, 16 g-points at a time, ceil(intervals / 4) sub-stages each (at least one)
struct MxStage { int b, g0, k0; bool first, last; };
__device__ __forceinline__ int mx_nsub(const BandMeta& m) {
  const int n = max(m.cnt[0], m.cnt[1]);
  return n <= MX_MINOR ? 1 : (n + MX_MINOR - 1) / MX_MINOR;
}
__device__ __forceinline__ void mx_stage_next(const BandMeta* bm, int nbnd, MxStage& s) {
  const int nsub = mx_nsub(bm[s.b]);
  if (s.k0 / MX_MINOR + 1 < nsub) {
    s.k0 += MX_MINOR;
  } else {
    s.k0 = 0;
    if (s.g0 + 16 <= bm[s.b].gE) s.g0 += 16;
    else if (s.b + 1 < nbnd) { ++s.b; s.g0 = bm[s.b].gS; }
    // (past the end: stays on the last stage; callers clamp)
  }
  s.first = s.k0 == 0;
  s.last = s.k0 / MX_MINOR + 1 >= mx_nsub(bm[s.b]);
}
// the same values in scalar registers (they come from the band table in LDS, i.e. through vector registers): the
// matrix waves branch on them around MFMAs, which ignore the execution mask
__device__ __forceinline__ void mx_stage_uniform(MxStage& s) {
  s.b = __builtin_amdgcn_readfirstlane(s.b); s.g0 = __builtin_amdgcn_readfirstlane(s.g0);
  s.k0 = __builtin_amdgcn_readfirstlane(s.k0);
  s.first = __builtin_amdgcn_readfirstlane((int)s.first) != 0; s.last = __builtin_amdgcn_readfirstlane((int)s.last) != 0;
}

template <int NW /* column waves = matrix waves */, bool OVERWRITE, bool ADDB>
__global__ void __launch_bounds__(NW * 128) tau_absorption_mx_kernel(MxArgs a) {
  constexpr int TILE = NW * 64;
  __shared__ __align__(16) double buf[2][TILE * 16];
  extern __shared__ BandMeta bm[];  // [nbnd]
  if (*a.skip_if || *a.skip_if2) return;
  const int tid = threadIdx.x, lane = tid & 63;
  if (tid == 0 && blockIdx.x == 0 && blockIdx.y == 0) *a.stat = 10;
  const unsigned ncol = a.ncol, nlay = a.nlay, ilay = blockIdx.y;
  const unsigned ncl = ncol * nlay;  // host guarantees < 2^29
  const int nbnd = a.nbnd, ntemp = a.ntemp, TE = a.TE, ngpt = a.ngpt;
  {
    const int* src = reinterpret_cast<const int*>(a.bmeta);
    int* dst = reinterpret_cast<int*>(bm);
    const int nw = nbnd * (int)(sizeof(BandMeta) / sizeof(int));
    for (int i = tid; i < nw; i += 2 * TILE) dst[i] = src[i];
  }
  __syncthreads();
  int nstage = 0;
  for (int b = 0; b < nbnd; ++b) nstage += ((bm[b].gE - bm[b].gS + 1) / 16) * mx_nsub(bm[b]);
  nstage = __builtin_amdgcn_readfirstlane(nstage);
  MxStage s0;
  s0.b = 0; s0.g0 = bm[0].gS; s0.k0 = 0; s0.first = true; s0.last = mx_nsub(bm[0]) == 1;

  const bool is_matrix = tid >= TILE;
  const int rtid = tid & (TILE - 1);  // thread index within the role
  if (is_matrix) {
    // ================================ matrix waves ================================
    // Per stage and wave: 64 sorted positions = 4 groups of 16, in RUNS of equal keys.  Everything a stage's MFMAs need except
    // the weight rows is known BEFORE the stage's barrier: the sorted (column, key) list of the band's flavor comes from the
    // pre-pass (requested three stages ahead: under the column waves' stores a request takes about a stage to come back),
    // hence the runs and the K rows of the first MX_PRE runs, whose loads are in flight while the wave waits at the barrier;
    // the stage's plan (band, g-points, active minor intervals, offsets) comes from the plan's stage list by scalar loads.
    // (A ring of K sets reloaded a whole stage ahead -- statically unrolled run sections, counted waits -- was built and
    //  measured as well: no faster.  What the K rows cost is their VOLUME, 12-18 GB per launch through L2 and the fabric,
    //  not their latency: DESIGN.md section 4.2b.)
    constexpr int MX_PRE = 3;
#ifdef MX_PRIO
    __builtin_amdgcn_s_setprio(MX_PRIO);
#endif
    const int p = rtid;  // sorted position this lane looks up
    const int kk = lane >> 4, gl = lane & 15;
    const size_t tl = blockIdx.x + (size_t)gridDim.x * ilay;
    const int n_lo = a.n_lo[tl];
    const bool up = p >= n_lo;  // the regime of the column at this position (lower columns sort first)
    const unsigned* const spk = a.sort_pk + tl * a.nflav * TILE + p;
    const MxStageRec* const ST = a.stages;
    const int nst = a.nstage;
    auto pk_of = [&](int j) -> unsigned {
      const MxStageRec& r = ST[min(j, nst - 1)];
      return spk[(size_t)(up ? r.flav[1] : r.flav[0]) * TILE];
    };
    struct KRun { double K0, K1, Km[MX_MINOR]; };
    unsigned pkA = pk_of(0), pkB = pk_of(1), pkC = pk_of(2);
    __syncthreads();  // (the column waves' prologue barrier)
    MX_T0();
#pragma unroll 1
    for (int i = 0; i <= nstage; ++i) {
      const MxStageRec st = ST[min(i, nst - 1)];
      const int g0 = st.g0;
      const bool first = (st.flags & 1) != 0;
      // ---- keys of my 64 positions, runs of equal keys
      const unsigned key_l = pkA >> 9;
      const int idxA = (int)(pkA & 511u);
      const unsigned key_p = (unsigned)__shfl_up((int)key_l, 1);
      const unsigned long long starts = __ballot(lane == 0 || key_l != key_p);
      auto load_run = [&](unsigned kc, KRun& k) {
#ifdef MX_X_NOKALL
        k.K0 = 1.0; k.K1 = 2.0; k.Km[0] = k.Km[1] = k.Km[2] = k.Km[3] = 0.5;
        return;
#endif
#ifdef MX_X_KFIXED   // experiment: every request goes to the same rows (what do the requests cost when they hit the nearest cache?)
        kc = 2u | (3u << 5) | (1u << 12) | (1u << 16);
#endif
        const int jT = kc & 31, jp = (kc >> 5) & 63, itr = (kc >> 11) & 1, e1 = (kc >> 12) & 15, e2 = (kc >> 16) & 15;
        // major: corner kk = eta offset + 2 x pressure offset, of temperature jT (K0, eta index e1) and jT + 1 (K1, e2)
        const unsigned rp = (unsigned)(jp - 2 + (kk >> 1)) * (unsigned)TE;
        k.K0 = a.kmaj[(size_t)(rp + (unsigned)(e1 - 1 + (kk & 1)) * ntemp + (unsigned)(jT - 1)) * ngpt + g0 + gl];
        k.K1 = a.kmaj[(size_t)(rp + (unsigned)(e2 - 1 + (kk & 1)) * ntemp + (unsigned)jT) * ngpt + g0 + gl];
        // minor: corner kk = eta offset + 2 x temperature offset
        const Float* const kt = itr ? a.kup : a.klo;
        const unsigned nk = itr ? a.nk_up : a.nk_lo;
        const unsigned rm = (unsigned)(((kk >> 1) ? e2 : e1) - 1 + (kk & 1)) * ntemp + (unsigned)(jT - 1 + (kk >> 1));
        const unsigned am = itr ? st.act[1] : st.act[0];
#pragma unroll
        for (int j = 0; j < MX_MINOR; ++j) {
          k.Km[j] = 0;
          if (am & (1u << j)) k.Km[j] = kt[(size_t)rm * nk + (itr ? st.koff[1][j] : st.koff[0][j]) + gl];
        }
      };
      KRun kp[MX_PRE];
      {
        unsigned long long rem = starts;
#pragma unroll
        for (int r = 0; r < MX_PRE; ++r) {
          kp[r].K0 = 0; kp[r].K1 = 0;
#pragma unroll
          for (int j = 0; j < MX_MINOR; ++j) kp[r].Km[j] = 0;
          if (rem != 0 && i < nstage) {
            load_run((unsigned)__builtin_amdgcn_readlane((int)key_l, __builtin_ctzll(rem)), kp[r]);
            rem &= rem - 1;
          }
        }
      }
      const unsigned pkD = pk_of(i + 3);  // (three stages ahead)
      MX_ARRIVE();
      __syncthreads();  // B(i): the weight rows of stage i are in buf[i & 1]; the column waves have tau of stage i - 1
      MX_LEAVE();
      if (i == nstage) break;
#ifdef MX_X_NOMAT
      pkA = pkB; pkB = pkC; pkC = pkD;
      continue;
#endif
      double* const rows = buf[i & 1];
      // ---- state of the group being worked on (a group may span runs: its accumulator lives across the run sections)
      int pos = 0, qcur = -1, rcur = -1;
      unsigned ka = 0, actc = 0;
      int cd[4] = {0, 0, 0, 0};
      double w0 = 0, w1 = 0, wm[MX_MINOR] = {0, 0, 0, 0};
      mx_v4d D = {0, 0, 0, 0};
      KRun kc_;  // the run being worked on
      kc_.K0 = 0; kc_.K1 = 0;
#pragma unroll
      for (int j = 0; j < MX_MINOR; ++j) kc_.Km[j] = 0;
#pragma unroll 1
      while (pos < 64) {
        const int q = pos >> 4;
        if (q != qcur) {  // weights and result rows of the group
          qcur = q;
          const int ca = __shfl(idxA, 16 * q + gl);
          ka = (unsigned)__shfl((int)key_l, 16 * q + gl);
#pragma unroll
          for (int r = 0; r < 4; ++r) cd[r] = __shfl(idxA, 16 * q + kk + 4 * r);
          w0 = rows[mx_elem(ca, kk)]; w1 = rows[mx_elem(ca, 4 + kk)];
          const double wf = rows[mx_elem(ca, 8 + kk)];
#pragma unroll
          for (int j = 0; j < MX_MINOR; ++j) wm[j] = wf * rows[mx_elem(ca, 12 + j)];
          D = mx_v4d{0, 0, 0, 0};
        }
        // the run that holds position `pos`: its index among the wave's runs, its key, where the next one starts
        const int r = __popcll(starts & ((2ull << pos) - 1ull)) - 1;
        const unsigned kc = (unsigned)__builtin_amdgcn_readlane((int)key_l, pos);
        const unsigned long long later = pos < 63 ? (starts >> (pos + 1)) : 0ull;
        const int end = later ? pos + 1 + __builtin_ctzll(later) : 64;
        if (r != rcur) {
          rcur = r;
          actc = ((kc >> 11) & 1) ? st.act[1] : st.act[0];
          if (r == 0) kc_ = kp[0];
          else if (r == 1) kc_ = kp[1];
          else if (r == 2) kc_ = kp[2];
          else load_run(kc, kc_);
        }
        const bool mine = ka == kc;
#ifndef MX_X_NOMFMA
        if (first) {
          D = __builtin_amdgcn_mfma_f64_16x16x4f64(mine ? w0 : 0.0, kc_.K0, D, 0, 0, 0);
          D = __builtin_amdgcn_mfma_f64_16x16x4f64(mine ? w1 : 0.0, kc_.K1, D, 0, 0, 0);
        }
#pragma unroll
        for (int j = 0; j < MX_MINOR; ++j)
          if (actc & (1u << j)) D = __builtin_amdgcn_mfma_f64_16x16x4f64(mine ? wm[j] : 0.0, kc_.Km[j], D, 0, 0, 0);
#else
        D[0] += (mine ? w0 + w1 + wm[0] + wm[1] + wm[2] + wm[3] : 0.0) + kc_.K0 + kc_.K1 + kc_.Km[0] + kc_.Km[1] + kc_.Km[2] + kc_.Km[3];
#endif
        const int gend = 16 * q + 16;
        pos = end < gend ? end : gend;
        if (pos == gend) {  // the group is complete: result rows (lane >> 4) + 4 r, g-point gl, back into the columns' rows, in place
#pragma unroll
          for (int r2 = 0; r2 < 4; ++r2) rows[mx_elem(cd[r2], gl)] = D[r2];
        }
      }
      pkA = pkB; pkB = pkC; pkC = pkD;
    }
    MX_TEND(1);
    return;
  }

  // ================================ column waves (lanes = columns) ================================
  const unsigned icol = blockIdx.x * TILE + rtid;
  const unsigned ic = min(icol, ncol - 1);
  const unsigned cl = ic + ncol * ilay;
  const unsigned cl8 = cl * (unsigned)sizeof(Float);
  const int itropo = a.tropo[cl] ? 0 : 1;
  const Float P = a.play[cl], T = a.tlay[cl];
  const Float dens = (Float)0.01 * P / T;                                                             // :469
  const Float vmr_fact = (Float)1 / a.col_gas[cl];                                                    // :471
  const Float dry_fact = (Float)1 / ((Float)1 + a.col_gas[cl + (size_t)ncl * a.idx_h2o] * vmr_fact);  // :472
  const Float sfact = vmr_fact * dry_fact;
  __syncthreads();  // (prologue barrier, paired with the matrix waves')
  struct In { Float2 fm[4], cm, fn[2]; Float cg[MX_MINOR], cgs[MX_MINOR], addv; };
  auto load_in = [&](const MxStage& s, In& x) {
    const BandMeta& B = bm[s.b];
    const size_t clf = cl + (size_t)ncl * B.flav[itropo];
    const Float2* fmp = reinterpret_cast<const Float2*>(a.fmajor + 8 * clf);
#pragma unroll
    for (int i = 0; i < 4; ++i) x.fm[i] = fmp[i];
    x.cm = *reinterpret_cast<const Float2*>(a.col_mix + 2 * clf);
    const Float2* fnp = reinterpret_cast<const Float2*>(a.fminor + 4 * clf);
    x.fn[0] = fnp[0]; x.fn[1] = fnp[1];
#pragma unroll
    for (int j = 0; j < MX_MINOR; ++j) {
      // (slots past the band's count are zero-filled: gas 0 = dry air, a valid plane; the value is not used)
      const MinorMeta& m = B.m[itropo][min(s.k0 + j, MAXM - 1)];
      x.cg[j] = a.col_gas[cl + (size_t)ncl * m.idx_minor];
      x.cgs[j] = a.col_gas[cl + (size_t)ncl * (((m.flags & 1) && m.idx_scaling > 0) ? m.idx_scaling : 0)];
    }
    x.addv = ADDB ? a.add_bybnd[cl + (size_t)ncl * s.b] : (Float)0;
  };
  auto write_w = [&](const MxStage& s, const In& x, int ib) {
    const BandMeta& B = bm[s.b];
    Float w[16];
    w[0] = x.cm.x * x.fm[0].x; w[1] = x.cm.x * x.fm[0].y; w[2] = x.cm.x * x.fm[1].x; w[3] = x.cm.x * x.fm[1].y;
    w[4] = x.cm.y * x.fm[2].x; w[5] = x.cm.y * x.fm[2].y; w[6] = x.cm.y * x.fm[3].x; w[7] = x.cm.y * x.fm[3].y;
    w[8] = x.fn[0].x; w[9] = x.fn[0].y; w[10] = x.fn[1].x; w[11] = x.fn[1].y;
    const int cnt = B.cnt[itropo];
#pragma unroll
    for (int j = 0; j < MX_MINOR; ++j) {
      const int k = s.k0 + j;
      const MinorMeta& m = B.m[itropo][min(k, MAXM - 1)];
      Float sc = x.cg[j];
      if (m.flags & 1) {       // :469
        sc = sc * dens;
        if (m.idx_scaling > 0) // :470-478
          sc = sc * ((m.flags & 2) ? ((Float)1 - x.cgs[j] * sfact) : (x.cgs[j] * sfact));
      }
      w[12 + j] = (k < cnt && m.mS <= s.g0 && m.mE >= s.g0) ? sc : (Float)0;
    }
    double* const rows = buf[ib];
#pragma unroll
    for (int e = 0; e < 16; e += 2) *reinterpret_cast<Float2*>(rows + mx_elem(rtid, e)) = Float2{w[e], w[e + 1]};
  };
  const size_t gstride = (size_t)ncl * sizeof(Float);
  Float acc[16];
  In in;
  MxStage sw = s0;   // the stage whose weights are written next
  load_in(sw, in);
  write_w(sw, in, 0);
  Float addv_r = 0, addv_w = in.addv;  // the by-band operand of the stage read back next / of the stage after it
  mx_stage_next(bm, nbnd, sw);
  load_in(sw, in);   // stage 1 (or the clamped last stage): consumed in iteration 0
  MxStage sr = s0;   // the stage read back next
  MX_T0();
#pragma unroll 1
  for (int i = 0; i <= nstage; ++i) {
    MX_ARRIVE();
    __syncthreads();  // B(i)
    MX_LEAVE();
    // tau of stage i - 1 (or its sub-stage's share): this column's own row of buf[(i - 1) & 1]
    if (i > 0) {
      const double* const rows = buf[(i - 1) & 1];
#pragma unroll
      for (int e = 0; e < 16; e += 2) {
        const Float2 v = *reinterpret_cast<const Float2*>(rows + mx_elem(rtid, e));
        if (sr.first) { acc[e] = v.x; acc[e + 1] = v.y; }
        else { acc[e] += v.x; acc[e + 1] += v.y; }
      }
    }
    // weights of stage i + 1 into the row just read (same buffer, same row: no other column wave thread touches it)
    Float addv_new = 0;
    if (i + 1 < nstage) {
      write_w(sw, in, (i + 1) & 1);
      addv_new = in.addv;
      mx_stage_next(bm, nbnd, sw);
#ifndef MX_X_NOIN
      load_in(sw, in);  // stage i + 2 (clamped: the last stage again -- a static number of requests per iteration)
#endif
    }
    if (i > 0) {
      // (accumulating: columns past the end repeat the last one, and several read - add - write sequences on one address
      //  would add the incoming value more than once; overwriting, they store the same value to the same address)
      if (sr.last && (OVERWRITE || icol < ncol)) {
        char* const tplane = reinterpret_cast<char*>(a.tau + (size_t)ncl * sr.g0);
        unsigned toff = cl8;
        asm volatile("" : "+v"(toff));
#pragma unroll
        for (int j = 0; j < 16; ++j) {
          Float* const tp = reinterpret_cast<Float*>(tplane + gstride * j + toff);
          Float v = acc[j];
          if (ADDB) v = v + addv_r;
#ifdef MX_X_NOSTORE
          if (v == 1.2345e-300) store_stream(tp, v);
#else
          if (OVERWRITE) store_stream(tp, v);
          else store_stream(tp, *tp + v);
#endif
        }
      }
      mx_stage_next(bm, nbnd, sr);
    }
    addv_r = addv_w; addv_w = addv_new;
  }
  MX_TEND(0);
}

}  // namespace
#endif  // !RTE_USE_SP
