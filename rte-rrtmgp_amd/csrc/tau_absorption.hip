// tau_absorption.hip -- host side of rrtmgp_compute_tau_absorption and rrtmgp_compute_tau_rayleigh (C ABI = the reference's
// bind(C) interface, rrtmgp/kernels/api/mo_gas_optics_rrtmgp_kernels.F90:71-199; include/rte_rrtmgp_kernels.h) and of their
// fused extension forms (include/rte_hip_ext.h), for gfx950 (MI355X).  Semantics follow the reference `default` CPU kernels
// (rrtmgp/kernels/mo_gas_optics_rrtmgp_kernels.F90:176-565), not its OpenACC variant.
//
// Device code: tau_slab.h (the production kernel: LDS slab of table rows per stage), tau_direct.h (direct gathers: small
// calls, irregular tables, the worklist of oversized boxes; set-up kernels), tau_rayleigh.h.  This file: staging of the
// arguments, the plan derived from the small index tables (cached, re-checked on the device every call), the choice of
// kernels and the launches.
#include "gas_optics_common.h"
#include "tau_direct.h"
#include "tau_rayleigh.h"
#include "tau_slab.h"

static const bool g_worklist_native = getenv("RTE_WORKLIST_NATIVE") != nullptr;  // A/B: worklist entries from the native-layout tables

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
  // the slab kernel (tau_slab.h): whole aligned stages, at most SLAB_MAXSTAGE of them, the bit-mask geometry pre-pass
  // (tile_geom2_kernel: table dimensions within its mask words), the band table beside the slab in LDS (nbnd <= 20),
  // 32-bit counts of 16-byte units for the staged table rows, 32-bit byte offsets into a g-point plane
  const bool fast = cache.fast_ok && ncol >= 512 && !g_tau_force_direct && ncl < ((size_t)1 << 29) &&
                    ngpt / cache.gw <= SLAB_MAXSTAGE && nbnd <= 20 && nflav <= MAXFLAV && neta < 31 && ntemp < 31 && npres + 1 < 63 &&
                    sizeof(Float) * (tn * (npres + 1) * ngpt + tn * ((size_t)nkl_ + nku_ + 2 * (size_t)ngpt + 16)) < ((size_t)1 << 35) &&
                    al(d_fmajor, 16) && al(d_fminor, 16) && al(d_col_mix, 16) && al(d_jeta, 8) &&
                    (overwrite_ok || rte::is_device_memory(d_tau)) &&
                    (!rh || (overwrite_ok && nbnd <= 16));  // (fused: the bands tile the g-points -- else tau was zero-filled above)
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
  // (ONE allocation: the slab kernel addresses every table row as a 32-bit count of 16-byte units from kmaj_g)
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
  const bool zero_check = !overwrite_ok && v.atomic_ok && !g_tau_no_zero_check && al(d_tau, 16) && cache.gw != 0;
  if (zero_check) {
    rte::ProfScope p("tau_is_zero_kernel");
    hipLaunchKernelGGL(tau_is_zero_kernel, dim3(256 * 16), dim3(256), 0, st, (const Float*)d_tau, ncl * (size_t)ngpt, nonzero);
  }
  v.rf = RaylFuse{};
  if (rh) {
    v.rf.krayl_g[0] = kray_g; v.rf.krayl_g[1] = kray_g + tn * ngpt; v.rf.col_dry = d_col_dry;
    v.rf.cld_tau = cb.cld_tau; v.rf.cld_ssa = cb.cld_ssa; v.rf.cld_g = cb.cld_g; v.rf.ssa = cb.ssa; v.rf.g = cb.g;
  }
  v.worklist = worklist;
  hipStream_t aux = nullptr;
  constexpr int NCW = 8, SLAB = SLAB_FLOATS;  // 8 waves = 512 columns per block, 2 x 68 KB of slab: one block per CU
  {
    const unsigned tiles = cdiv(ncol, NCW * 64);
    const bool share = share_boxes() && !c.any_host() && !rh;
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
    const dim3 grid(tiles, nlay), blk(NCW * 64);
    // few (tile, layer) pairs (calls of some thousand columns): the stages of a pair go to several blocks, at least 4 each, until
    // about two blocks per CU are in the grid
    const unsigned nz = std::max(1u, std::min((unsigned)(ngpt / cache.gw) / 4u, RTE_SMALL_GRID_BLOCKS / std::max(1u, tiles * (unsigned)nlay)));
    const dim3 gridk(tiles, nlay, nz);
    const size_t dyn = sizeof(BandMeta) * nbnd;
    const TileGeom* cg = d_geom;
    Geom2Args ga{};
    ga.ncol = ncol; ga.nlay = nlay; ga.nbnd = nbnd; ga.nflav = nflav; ga.slab_floats = SLAB; ga.planck = false;
    ga.lim = lim; ga.jeta = d_jeta; ga.jtemp = d_jtemp; ga.jpress = d_jpress; ga.tropo = d_tropo; ga.bmeta = d_bm;
    ga.skip_if = overlap; ga.worklist = v.worklist; ga.valid_out = share ? gs().shared.valid : nullptr;
    ga.extra_planes = rh ? 2 : 0;
    ga.row_stride = cache.gw + 16 / (int)sizeof(Float);  // (the slab's rows are padded by one 16-byte piece)
    ga.irregular = irregular;
    ga.stat = stats_dev() + 2;
    if (share_masks() && gs().imask.seq >= 0 && gs().imask.seq + 1 == rte::call_seq() && gs().imask.jeta == jeta && gs().imask.jtemp == jtemp &&
        gs().imask.jpress == jpress && gs().imask.tropo == tropo && gs().imask.ncol == ncol && gs().imask.nlay == nlay &&
        gs().imask.nflav == nflav && !c.any_host()) {
      ga.imask = gs().imask.buf;
      ga.imask_nblk = cdiv(ncol, 256);
    }
// <overwrite, stage width, by-band operand, fused Rayleigh form>; two steps of LDS row reads in flight per wave
#define RTE_TAU_K(OW, GW, AB, RV) hipLaunchKernelGGL((tau_slab_kernel<NCW, SLAB, OW, GW, 4, AB, RV, TAU_DEPTH>), gridk, blk, dyn, st, vk, cg)
#define RTE_LAUNCH_TAU_(GW, AB)                                                                                   \
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
#define RTE_LAUNCH_TAU(GW)                                                                                        \
  do {                                                                                                            \
    {                                                                                                             \
      rte::ProfScope p("tau_absorption_setup");                                                                   \
      hipLaunchKernelGGL((tile_geom2_kernel<NCW * 64, GW>), grid, blk, 0, st, ga, d_geom);                        \
    }                                                                                                             \
    aux = rte::aux_fork(); /* the worklist is complete: its kernel may run beside the slab kernel */             \
    rte::ProfScope p("tau_absorption_kernel");                                                                    \
    const TauV5& vk = v;                                                                                          \
    if (rh) {                                                                                                     \
      if (cb.cld_tau) RTE_TAU_K(true, GW, false, 2);                                                              \
      else if (cb.g)  RTE_TAU_K(true, GW, false, 1);                                                              \
      else            RTE_TAU_K(true, GW, false, 3);                                                              \
    } else if (d_add) RTE_LAUNCH_TAU_(GW, true); else RTE_LAUNCH_TAU_(GW, false);                                 \
  } while (0)
    if (cache.gw == 16) RTE_LAUNCH_TAU(16); else RTE_LAUNCH_TAU(8);
#undef RTE_LAUNCH_TAU
#undef RTE_LAUNCH_TAU_
#undef RTE_TAU_K
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
    if (gft.kmaj)
      hipLaunchKernelGGL(tau_absorption_worklist_kernel<true>, dim3(aux ? 16384 : 4096), dim3(aux ? 64 : 256), 0, st, aw, gft,
                         (const int*)v.worklist, NCW * 64, stats_dev() + 0);
    else
      hipLaunchKernelGGL(tau_absorption_worklist_kernel<false>, dim3(aux ? 16384 : 4096), dim3(aux ? 64 : 256), 0, st, aw, gft,
                         (const int*)v.worklist, NCW * 64, stats_dev() + 0);
    if (rh) rayleigh_direct(nullptr, (const int*)v.worklist, NCW * 64);  // (lambda launches on st)
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
