/* rte_hip_ext.h -- library-extension entry points of librte_rrtmgp_hip.so (NOT part of the reference kernel interface,
 * which is include/rte_rrtmgp_kernels.h).  A program that only wants the drop-in never needs them; a host model that
 * wants the device rate uses a few (streams, contexts, host-mirror mode, error channel).  Scalars BY VALUE.
 * Every setting belongs to the calling thread's current CONTEXT (see rte_hip_ctx_*); a new context inherits the settings
 * of the context that was current when it was made.  INTEGRATION.md shows the Fortran interface block. */
#ifndef RTE_HIP_EXT_H
#define RTE_HIP_EXT_H
#include "rte_rrtmgp_kernels.h" /* Float, Bool */
#ifdef __cplusplus
extern "C" {
#endif

/* ---- contexts: stream + scratch arena + persistent buffers + plan caches + mirrors, one mutex each ------------------
 * Threads that never choose a context share the process-wide default one (their calls are serialised).  Two threads on two
 * contexts run concurrently, each on its own stream (reference: examples/all-sky/rrtmgp_allsky.F90:331, the intended
 * OpenMP loop over column blocks).  RTE_HIP_THREAD_CONTEXTS=1 in the environment gives every calling thread its own
 * context automatically -- for UNCHANGED host programs. */
void* rte_hip_ctx_create(int device /* -1: current device */, void* hip_stream /* NULL: own non-blocking stream */);
void* rte_hip_ctx_set_current(void* ctx /* NULL: default context */);   /* returns the previous one */
void* rte_hip_ctx_get_current(void);
int   rte_hip_ctx_destroy(void* ctx);
int   rte_hip_set_stream(void* hip_stream);   /* the current context launches on this stream from now on */
int   rte_hip_sync(void);                     /* materialise recorded fills, drain the context's stream */
/* a sequence of library calls captured as ONE hipGraph and replayed with one submission: halves the host time per chain
 * (0.13 -> 0.05 ms); the device time is that of the launches themselves (measured).  Device pointers only, the same arrays at every launch, one uncaptured run of the same sequence
 * beforehand, no value returned to the host inside the region.  All return 0, -1 on a HIP error. */
int   rte_hip_graph_begin(void);
int   rte_hip_graph_end(void** graph_exec);
int   rte_hip_graph_launch(void* graph_exec); /* on the context's stream, ordered with the calls around it.  -4: the graph is
                                                  stale -- it addresses library buffers (scratch arena, persistent slots) that
                                                  were freed or reallocated since the capture (a LARGER call on the context,
                                                  rte_hip_release, dropped host tables): capture again.  -1: HIP error */
int   rte_hip_graph_destroy(void* graph_exec);
int   rte_hip_release(void);                  /* free every device buffer the context holds */
int   rte_hip_device_count(void);

/* ---- error channel: the reference interface has none (void subroutines).  Default: message + abort(). -------------- */
int rte_hip_error_mode(int sticky);                 /* 1: record the first failure, later calls on the context are no-ops */
int rte_hip_last_error(char* buf, int buflen);      /* 0 = none, else hipError_t (or -1); message copied to buf */
int rte_hip_clear_error(void);

/* ---- host arrays (what the unchanged Fortran frontend passes) -------------------------------------------------------
 * Host-mirror mode (also RTE_HIP_HOST_MIRROR=1): outputs the frontend only hands on to the next kernel (interpolation
 * state, tau, Planck sources, incremented optical properties) stay on the device; fluxes and other small results are
 * copied back as usual.  Contract: host code does not read a held array before rte_hip_writeback(ptr) and does not write
 * part of one; the frontend's value checks must be off (rte_config_checks(.false.)).  csrc/runtime.hip documents the
 * canary mechanism that notices reused host memory. */
int       rte_hip_host_mirror(int on);
int       rte_hip_writeback(const void* host_ptr);  /* 1: array copied back, 0: not held on the device */
int       rte_hip_mirror_drop_all(void);
long long rte_hip_mirror_stat(int which);           /* counters, see csrc/runtime.hip (10, 11: unchanged host inputs served from the device, bytes not uploaded) */

/* ---- opt-in modes for drivers that touch the arrays only through this library between two calls -------------------- */
int rte_hip_defer_zero(int on);        /* zero_array_* recorded, folded into compute_tau_absorption (tau write-only) */
int rte_hip_defer_sources(int on);     /* compute_Planck_source leaves factored sources for the rte_lw_solver_noscat that follows
                                          (RTE_HIP_DEFER_SOURCES=1): 26 GB less written, 13 GB less read at 1e5 x 60 x 256 */
int rte_hip_share_geometry(int on);    /* interpolation -> tau -> Planck share the LUT bounding boxes of a column tile */
int rte_hip_overlap_planck(int on);    /* compute_Planck_source beside the compute_tau_absorption call it follows */
int rte_hip_aux_stream(int on);        /* direct-gather worklist beside the slab kernel (default on) */
int rte_hip_invalidate_plans(void);    /* after changing k-distribution tables that live in DEVICE memory */

/* ---- kernel timing with HIP events on the context's stream (bench.py) ---------------------------------------------- */
int rte_hip_profile_enable(int on);
int rte_hip_profile_only(const char* scope);
int rte_hip_profile_reset(void);
int rte_hip_profile_count(void);
int rte_hip_profile_get(int i, char* name, int buflen, long long* launches, double* total_ms);

/* ==== compute extensions: Float / Bool as in rte_rrtmgp_kernels.h, arrays dense column-major, scalars BY VALUE, all return 0.
 * Host or device pointers alike (staged like the reference entry points).  The Python mirror of the frontend
 * (rte-rrtmgp_amd/frontend.py) is the user of all of them; INTEGRATION.md section 4 shows Fortran bindings. ============ */

/* ---- fused / factored forms of the hot path ---------------------------------------------------------------------------
 * compute_tau_absorption with increment(clouds -> gas) by band folded in (tau = tau_gas + tau_bybnd(:, :, band(g))):
 * the argument list of rrtmgp_compute_tau_absorption (scalars by value) + tau_bybnd (ncol, nlay, nbnd). */
int rte_hip_compute_tau_absorption_inc_bybnd(
    int ncol, int nlay, int nbnd, int ngpt, int ngas, int nflav, int neta, int npres, int ntemp, int nminorlower,
    int nminorklower, int nminorupper, int nminorkupper, int idx_h2o, const int* gpoint_flavor, const int* band_lims_gpt,
    const Float* kmajor, const Float* kminor_lower, const Float* kminor_upper, const int* minor_limits_gpt_lower,
    const int* minor_limits_gpt_upper, const Bool* minor_scales_with_density_lower, const Bool* minor_scales_with_density_upper,
    const Bool* scale_by_complement_lower, const Bool* scale_by_complement_upper, const int* idx_minor_lower,
    const int* idx_minor_upper, const int* idx_minor_scaling_lower, const int* idx_minor_scaling_upper,
    const int* kminor_start_lower, const int* kminor_start_upper, const Bool* tropo, const Float* col_mix, const Float* fmajor,
    const Float* fminor, const Float* play, const Float* tlay, const Float* col_gas, const int* jeta, const int* jtemp,
    const int* jpress, Float* tau, const Float* tau_bybnd);
/* SW gas optics in one pass: compute_tau_absorption + compute_tau_rayleigh + combine_abs_and_rayleigh (2-stream)
 * [+ increment by band-wise cloud properties when cld_tau != NULL].  g == NULL (without clouds): g = 0 is not stored. */
int rte_hip_gas_optics_sw_2str(
    int ncol, int nlay, int nbnd, int ngpt, int ngas, int nflav, int neta, int npres, int ntemp, int nminorlower,
    int nminorklower, int nminorupper, int nminorkupper, int idx_h2o, const int* gpoint_flavor, const int* band_lims_gpt,
    const Float* kmajor, const Float* kminor_lower, const Float* kminor_upper, const int* minor_limits_gpt_lower,
    const int* minor_limits_gpt_upper, const Bool* minor_scales_with_density_lower, const Bool* minor_scales_with_density_upper,
    const Bool* scale_by_complement_lower, const Bool* scale_by_complement_upper, const int* idx_minor_lower,
    const int* idx_minor_upper, const int* idx_minor_scaling_lower, const int* idx_minor_scaling_upper,
    const int* kminor_start_lower, const int* kminor_start_upper, const Bool* tropo, const Float* col_mix, const Float* fmajor,
    const Float* fminor, const Float* play, const Float* tlay, const Float* col_gas, const int* jeta, const int* jtemp,
    const int* jpress, const Float* krayl, const Float* col_dry, Float* tau, Float* ssa, Float* g, const Float* cld_tau,
    const Float* cld_ssa, const Float* cld_g);
/* compute_tau_rayleigh + combine (2-stream) on a tau_abs already computed (tau may alias tau_abs) */
int rte_hip_tau_rayleigh_combine_2str(int ncol, int nlay, int nbnd, int ngpt, int ngas, int nflav, int neta, int ntemp,
                                      const int* gpoint_flavor, const int* band_lims_gpt, const Float* krayl, int idx_h2o,
                                      const Float* col_dry, const Float* col_gas, const Float* fminor, const int* jeta,
                                      const Bool* tropo, const int* jtemp, const Float* tau_abs, Float* tau, Float* ssa, Float* g,
                                      const Float* cld_tau, const Float* cld_ssa, const Float* cld_g);
/* Factored LW sources: compute_Planck_source that writes the Planck FRACTION pfrac (ncol, nlay, ngpt) and the band's Planck
 * function at layer / level temperatures planck_lay (ncol, nlay, nbnd), planck_lev (ncol, nlay+1, nbnd) instead of
 * lay_source / lev_source (26 GB less written at 1e5 x 60 x 256); lw_solver_noscat (broadband) that takes them; and the
 * expansion to the reference arrays.  Fluxes are bit-identical to the reference-ABI chain for Planck fractions that are 0
 * or >= 2^-383 (the level source at the column ends is sqrt(p * p) * B: smaller positive fractions underflow in p * p). */
int rte_hip_compute_Planck_source_factored(int ncol, int nlay, int nbnd, int ngpt, int nflav, int neta, int npres, int ntemp,
                                           int nPlanckTemp, const Float* tlay, const Float* tlev, const Float* tsfc, int sfc_lay,
                                           const Float* fmajor, const int* jeta, const Bool* tropo, const int* jtemp,
                                           const int* jpress, const int* band_lims_gpt, const Float* pfracin, double temp_ref_min,
                                           double totplnk_delta, const Float* totplnk, const int* gpoint_flavor, Float* sfc_src,
                                           Float* pfrac, Float* planck_lay, Float* planck_lev, Float* sfc_source_Jac);
int rte_hip_lw_solver_noscat_factored(int ncol, int nlay, int ngpt, int nbnd, int top_at_1, int nmus, const Float* Ds,
                                      const Float* weights, const int* band_lims_gpt, const Float* tau, const Float* pfrac,
                                      const Float* planck_lay, const Float* planck_lev, const Float* sfc_emis, const Float* sfc_src,
                                      const Float* inc_flux, Float* broadband_up, Float* broadband_dn, int do_jac,
                                      const Float* sfc_srcJac, Float* flux_upJac);
int rte_hip_expand_factored_sources(int ncol, int nlay, int nbnd, int ngpt, const int* band_lims_gpt, const Float* pfrac,
                                    const Float* planck_lay, const Float* planck_lev, Float* lay_source, Float* lev_source);
/* By-band fluxes (rte/extensions/mo_fluxes_byband.F90) straight from the solvers: (ncol, nlay+1, nbnd) outputs, no spectral
 * flux arrays in memory.  rte_sw_solver_2stream and rte_hip_sw_solver_2stream_byband accept g == NULL as g = 0 (clear sky). */
int rte_hip_lw_solver_noscat_byband(int ncol, int nlay, int ngpt, int nbnd, int top_at_1, int nmus, const Float* Ds,
                                    const Float* weights, const int* band_lims_gpt, const Float* tau, const Float* lay_source,
                                    const Float* lev_source, const Float* sfc_emis, const Float* sfc_src, const Float* inc_flux,
                                    Float* byband_up, Float* byband_dn);
int rte_hip_sw_solver_2stream_byband(int ncol, int nlay, int ngpt, int nbnd, int top_at_1, const int* band_lims_gpt,
                                     const Float* tau, const Float* ssa, const Float* g, const Float* mu0, const Float* sfc_alb_dir,
                                     const Float* sfc_alb_dif, const Float* inc_flux_dir, int has_dif_bc, const Float* inc_flux_dif,
                                     Float* byband_up, Float* byband_dn, Float* byband_dir);

/* ---- the frontend's glue loops on the device (reference lines in csrc/glue.hip, csrc/util.hip) ---- */
int rte_hip_get_layer_number(int ncol, int nlay, const Float* vmr_h2o, const Float* plev, double m_dry, double grav, Float* col_dry);
int rte_hip_get_layer_mass(int ncol, int nlay, int ngas, const Float* vmr, const Float* plev, const Float* mol_weights, double m_dry,
                           double grav, Float* layer_mass);
int rte_hip_col_gas_fill(int ncol, int nlay, int ngas, const Float* vmr, const Float* col_dry, Float* col_gas);
int rte_hip_tlev_interp(int ncol, int nlay, const Float* play, const Float* plev, const Float* tlay, Float* tlev);
int rte_hip_compute_optimal_angles(int ncol, int nlay, int ngpt, int nbnd, const int* band_lims, const Float* tau,
                                   const Float* optimal_angle_fit, Float* optimal_angles);
int rte_hip_combine_abs_and_rayleigh_1scl(int ncol, int nlay, int ngpt, const Float* tau_abs, const Float* tau_ray, Float* tau);
int rte_hip_combine_abs_and_rayleigh_2str(int ncol, int nlay, int ngpt, const Float* tau_abs, const Float* tau_ray, Float* tau,
                                          Float* ssa, Float* g);
int rte_hip_combine_abs_and_rayleigh_nstr(int ncol, int nlay, int ngpt, int nmom, const Float* tau_abs, const Float* tau_ray,
                                          Float* tau, Float* ssa, Float* p);
int rte_hip_expand_and_transpose(int ncol, int nbnd, int ngpt, const int* band_lims, const Float* arr_in, Float* arr_out);
int rte_hip_secants_fill(int ncol, int ngpt, int nmus, const Float* Ds, Float* secants);
int rte_hip_broadcast_gpt(int ncol, int ngpt, const Float* per_gpt, Float* out);   /* toa_src(:, g) = solar_source(g) */
int rte_hip_broadcast_cols(int n, int ncol, const Float* per_col, Float* out);
int rte_hip_mask_columns(int ncol, int nlev, const Bool* usecol, Float* flux_up, Float* flux_dn);
int rte_hip_rfmip_sw_toa_renorm(int ncol, int ngpt, const Float* total_solar_irradiance, Float* toa_flux);
int rte_hip_rfmip_sw_mu0(int ncol, const Float* solar_zenith_angle, const Bool* usecol, Float* mu0);
/* cloud optics (rrtmgp/frontend/mo_cloud_optics_rrtmgp.F90:373-425): masks, the liquid / ice combination, and all of it with
 * both table look-ups and optional delta scaling in one pass */
int rte_hip_cloud_masks(int ncol, int nlay, const Float* clwp, const Float* ciwp, Bool* liqmsk, Bool* icemsk);
int rte_hip_cloud_combine(int ncol, int nlay, int nspec, int twostr, const Float* ltau, const Float* ltaussa, const Float* ltaussag,
                          const Float* itau, const Float* itaussa, const Float* itaussag, Float* tau, Float* ssa, Float* g);
int rte_hip_cloud_optics_fused(int ncol, int nlay, int nbnd, int twostr, int delta_scale, const Float* clwp, const Float* ciwp,
                               const Float* reliq, const Float* deice, int liq_nsteps, double liq_step_size, double radliq_lwr,
                               const Float* extliq, const Float* ssaliq, const Float* asyliq, int ice_nsteps, double ice_step_size,
                               double diamice_lwr, const Float* extice, const Float* ssaice, const Float* asyice, Float* tau,
                               Float* ssa, Float* g);

/* ---- the exchange step of the column-sharded path for host programs without torch (csrc/collectives.hip) ----------------
 * Columns shard as contiguous ranges per rank, tables replicated, no data-path collective; ranks exchange only broadband
 * flux diagnostics.  nccl_comm: the caller's ncclComm_t (RCCL; from its MPI ranks: INTEGRATION.md section 5), enqueued on
 * the context's stream.  RCCL is resolved at run time: no link-time dependency.  Return -3: no RCCL in this process. */
int rte_hip_rccl_available(void);
int rte_hip_allreduce_mean_profile(void* nccl_comm /* NULL: one rank */, int ncol_local, int nlev, const Float* flux_up,
                                   const Float* flux_dn, long long ncol_global, Float* mean_up /* (nlev) */, Float* mean_dn);
int rte_hip_allgather_columns(void* nccl_comm, int ncol_local, int nlev, const Float* local /* (ncol_local, nlev), device */,
                              Float* global /* (ncol_local * nranks, nlev), device */);
/* slabs of unequal width (shard boundaries on multiples of 64 columns): ncol_slab >= every rank's ncol_local and ncol_global = their
 * sum are the same on every rank; rank r's columns lie behind those of the ranks before it */
int rte_hip_allgatherv_columns(void* nccl_comm, int ncol_local, int nlev, const Float* local /* (ncol_local, nlev), device */,
                               int ncol_slab, long long ncol_global, Float* global /* (ncol_global, nlev), device */);

/* ---- switches for tests and A/B timing (process-wide) ------------------------------------------------------------------ */
int rte_hip_force_direct_gather(int on);   /* gas optics on the direct-gather kernels only */
int rte_hip_force_generic_lw(int on);      /* LW solvers on the generic (any layer count) kernels only */
int rte_hip_force_generic_sw(int on);
int rte_hip_lw_mixed_segments(int on);     /* the same for rte_lw_solver_noscat (broadband, no rescaling) */
int rte_hip_sw_mixed_segments(int on);     /* 1 (default): rte_sw_solver_2stream at 57-60 layers on segments of 7 and 8 layers (no neutral slots); 0: 8 x 8 (A/B) */
int rte_hip_tau_zero_check(int on);        /* plain-ABI compute_tau_absorption looks whether tau is zero (default on) */
int rte_hip_set_lw2str_bugcompat(int on);  /* lw_solver_2stream: g-point 1's lev_source for every g-point, as the reference does */
int rte_hip_seg_groups(int n);             /* g-point groups per column tile of the segmented solvers (0: default) */
int rte_hip_lw_sfc_lds(int on);            /* the LW solver's surface arrays through LDS */
int rte_hip_stat(int which);               /* diagnostics (synchronises): 0 / 1 worklist entries of the last tau / Planck call,
                                              2 where the last tau call's tile geometry came from */

#ifdef __cplusplus
}
#endif
#endif
