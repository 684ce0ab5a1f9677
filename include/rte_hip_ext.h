/* rte_hip_ext.h -- library-extension entry points of librte_rrtmgp_hip.so (NOT part of the reference kernel interface,
 * which is include/rte_rrtmgp_kernels.h).  A program that only wants the drop-in never needs them; a host model that
 * wants the device rate uses a few (streams, contexts, host-mirror mode, error channel).  Scalars BY VALUE.
 * Every setting belongs to the calling thread's current CONTEXT (see rte_hip_ctx_*); a new context inherits the settings
 * of the context that was current when it was made.  INTEGRATION.md shows the Fortran interface block. */
#ifndef RTE_HIP_EXT_H
#define RTE_HIP_EXT_H
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
long long rte_hip_mirror_stat(int which);           /* counters, see csrc/runtime.hip */

/* ---- opt-in modes for drivers that touch the arrays only through this library between two calls -------------------- */
int rte_hip_defer_zero(int on);        /* zero_array_* recorded, folded into compute_tau_absorption (tau write-only) */
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

#ifdef __cplusplus
}
#endif
#endif
