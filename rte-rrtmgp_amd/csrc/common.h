// common.h -- shared declarations of the HIP kernel library (gfx950 only).
//
// Layout conventions used by every kernel in this library:
//   * all arrays are the caller's dense column-major Fortran arrays, column index fastest;
//     a wavefront's 64 lanes always span 64 CONSECUTIVE COLUMNS, so every access to an
//     (ncol, ...) array is a unit-stride 512-byte (fp64) request per wave;
//   * index values stored in integer arrays are 1-based, exactly as the frontend passes them.
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#include "rte_rrtmgp_kernels.h"

#define RTE_WAVE 64

#include <string>

namespace rte {
// A failing HIP call throws rte::Error out of the entry point's body; the entry point's RTE_CATCH hands it to
// rte::on_error, which prints and abort()s (the reference kernel interface has no error channel) unless the context is in
// sticky-error mode (rte_hip_error_mode(1): recorded, readable with rte_hip_last_error; runtime.hip).  An entry point
// without the macros lets the exception reach the extern "C" boundary, i.e. std::terminate -> abort().
struct Error { int code; std::string what; };
[[noreturn]] void fail(hipError_t e, const char* expr, const char* file, int line);
void on_error(const char* entry, const Error& e);
}  // namespace rte

#define HIP_CHECK(expr)                                                  \
  do {                                                                   \
    hipError_t e_ = (expr);                                              \
    if (e_ != hipSuccess) rte::fail(e_, #expr, __FILE__, __LINE__);      \
  } while (0)
#define RTE_TRY try {
#define RTE_CATCH(entry_name) } catch (const rte::Error& err_) { rte::on_error(entry_name, err_); }

namespace rte {

// ---- runtime (runtime.hip) ---------------------------------------------------------------
// All mutable state belongs to the calling thread's current CONTEXT (rte_hip_ctx_*; default: one per process).
hipStream_t stream();
// per-context state owned by another translation unit (the gas-optics plan caches): created on first use with `make`,
// released with `destroy` when the context releases its buffers
void* gas_state(void* (*make)(), void (*destroy)(void*));
// holds the current context's mutex (extension entry points that read or write per-context state outside a Call)
struct CtxLock { CtxLock(); ~CtxLock(); void* ctx_; };
void drop_table_copies();  // host-mirror mode: forget the cached device copies of host tables (rte_hip_invalidate_plans)
// device scratch that lives until the end of the current API call (bump allocator; grows)
void* scratch(size_t bytes);
// persistent named device buffers (LUT re-layouts etc.), keyed by a caller-chosen id
void* persistent(int slot, size_t bytes, bool* fresh);

// Scope object of one API call: serialises calls, resets the scratch arena, and stages host
// arrays.  in()/out()/inout() return a device pointer for `p`: `p` itself when it already is a
// device (or managed/registered) pointer, otherwise a scratch copy (copied back on destruction
// for out/inout, after a stream synchronise).
class Call {
 public:
  explicit Call(const char* name);
  ~Call() noexcept(false);  // reports launch errors of the call (throws rte::Error unless already unwinding)
  template <class T> const T* in(const T* p, size_t n) { return (const T*)stage((void*)p, n * sizeof(T), true, false); }
  template <class T> T* out(T* p, size_t n) { return (T*)stage((void*)p, n * sizeof(T), false, true); }
  template <class T> T* inout(T* p, size_t n) { return (T*)stage((void*)p, n * sizeof(T), true, true); }
  // host-mirror mode (runtime.hip, rte_hip_host_mirror): an output the caller's HOST code does not read before it hands
  // it to the next library call (interpolation state, tau, Planck sources, ...) stays on the device -- no copy back --
  // and is served from there when that host address comes in again.  Without the mode these are out() / inout().
  // `zero_fill`: if non-null, receives "the array is entirely zero by a recorded zero_array call and has not been
  // materialised" -- the caller then overwrites instead of accumulating (compute_tau_absorption).
  template <class T> T* out_lazy(T* p, size_t n) { return (T*)stage((void*)p, n * sizeof(T), false, true, true); }
  template <class T> T* inout_lazy(T* p, size_t n, bool* zero_fill = nullptr) {
    return (T*)stage((void*)p, n * sizeof(T), true, true, true, zero_fill);
  }
  // a k-distribution TABLE (kmajor, kminor_*, krayl, planck_frac, totplnk): in host-mirror mode a host table is uploaded once
  // per context and reused while its address, size and a sampled fingerprint of its contents (head, tail, 256 strided
  // words) stay the same -- the 33 MB of tables are a quarter of what a 4096-column block sends otherwise.  Part of the
  // mode's contract: tables are not modified in place while it is on (rte_hip_invalidate_plans drops the copies).
  // Without the mode: in().
  template <class T> const T* in_table(const T* p, size_t n) { return (const T*)stage_table((const void*)p, n * sizeof(T)); }
  // zero_array on a host array in host-mirror mode: recorded on the device copy (true) or not handled (false)
  bool lazy_zero(void* p, size_t bytes);
  // copy the device-resident arrays last written by the library entry `producer` back to their host addresses at the
  // end of this call (the reference frontend reads them on the host next)
  void writeback_produced_by(const char* producer);
  // small host-side copy of an input array that may live on the device (index tables etc.)
  template <class T> const T* host(const T* p, size_t n) { return (const T*)to_host((const void*)p, n * sizeof(T)); }
  bool any_host() const { return n_back_ > 0 || staged_in_; }
  // opt-in overlap of independent calls (runtime.hip, rte_hip_overlap_planck): run the rest of this call on the side stream
  bool try_fork(const void* const* outs, const size_t* bytes, int n);
  bool forked() const { return forked_; }
  const char* name;

 private:
  void* stage(void* p, size_t bytes, bool copy_in, bool copy_out, bool lazy = false, bool* zero_fill = nullptr);
  const void* stage_table(const void* p, size_t bytes);
  const void* to_host(const void* p, size_t bytes);
  struct Back { void* host; void* dev; size_t bytes; };
  Back back_[16];
  int n_back_ = 0;
  bool staged_in_ = false;
  bool staged_plain_ = false;  // ... by the runtime's own pageable copy (the source may still be in use when it returns)
  bool host_visible_ = false;  // some array is pinned / registered / managed host memory used in place
  void* host_tmp_[24];
  int n_host_tmp_ = 0;
  bool fork_candidate_ = false, forked_ = false;
  void* locked_ = nullptr;  // the context this call holds
  // host-mirror mode: host arrays that get their canaries at the end of the call, device buffers to recycle then
  struct Lazy { void* host; size_t bytes; unsigned long long magic; };
  Lazy lazy_[16];
  int n_lazy_ = 0;
  struct Recycle { void* dev; size_t cap; };
  Recycle recycle_[16];
  int n_recycle_ = 0;
};
long call_seq();  // sequence number of the API call in progress (every entry point counts)
hipStream_t aux_fork();  // second stream inside one call, ordered after everything the call has queued so far (nullptr: off)
void aux_join();         // the call's stream waits for it
void fork_point(const void* out, size_t bytes);  // marks "everything queued so far" at the start of a call others may overlap

bool is_device_pointer(const void* p);  // a kernel can address it (device, or pinned / registered / managed host memory)
bool is_device_memory(const void* p);   // device memory proper: calls on it are asynchronous

// deferred zero fill (runtime.hip)
bool defer_zero_enabled();
void defer_zero(void* p, size_t bytes);
bool take_pending_zero(const void* p, size_t bytes);
void flush_pending_zeros();

// deferred LW sources (runtime.hip; opt-in rte_hip_defer_sources / RTE_HIP_DEFER_SOURCES=1): compute_Planck_source on device
// arrays leaves the FACTORED sources -- the Planck fraction in the caller's lay_source array, the bands' Planck functions
// in library buffers, lev_source untouched -- and records that; rte_lw_solver_noscat on exactly these arrays solves from
// the factors (planck.hip, solvers.hip).  Any other library entry that is handed lay_source or lev_source finds them
// materialised first (Call::in), a new output into them drops the record (Call::out).
// Lifetime of a record: the caller's promise covers the time between compute_Planck_source and the solve.  Until the first
// solve has consumed it the record is trusted.  Afterwards it stays -- a second solve on the same sources (clear-sky and all-sky
// fluxes) and any other use must still find them right -- but it is no longer trusted: when the solve is done a fingerprint of
// the fraction (64 values spread over lay_source) is kept, and every later use compares it first; if the memory has other
// contents by then (the caller wrote or freed and reused it) the record is dropped and the arrays are taken as they are.
// At most one record lives per context: a deferred compute_Planck_source on other arrays first expands what is pending
// (the factors share one library buffer).
struct PendingSources {
  const void *lay, *lev;         // the caller's arrays
  int ncol, nlay, nbnd, ngpt;
  const void *plk_lay, *plk_lev; // (ncol, nlay, nbnd), (ncol, nlay+1, nbnd): library buffers
  const int* band_lims;          // device copy of band_lims_gpt
  void* sample;                  // 64 Floats + one int behind them: the fingerprint and the compare kernel's flag
  bool consumed;                 // a solve has used the record: validate before any further use
};
struct PendingSourcesOps {
  void (*expand)(const PendingSources&);       // lay_source / lev_source from the factors, in place
  void (*take_sample)(const PendingSources&);  // keep the fingerprint of lay_source (asynchronous)
  bool (*still_factored)(const PendingSources&);  // does lay_source still hold it? (synchronises)
};
bool defer_sources_enabled();
void defer_sources(const PendingSources& s, const PendingSourcesOps* ops);
bool take_pending_sources(const void* lay, const void* lev, PendingSources* out);
void sources_consumed(PendingSources s);  // back on the list after a solve from the factors
void flush_pending_sources();
void flush_pending_sources_except(const void* lay, const void* lev);

// kernel-level timing hooks (rte_hip_profile_*): record(name) brackets a launch with events
void prof_begin(const char* kernel);
void prof_end();
struct ProfScope {
  explicit ProfScope(const char* k) { prof_begin(k); }
  ~ProfScope() noexcept(false) { prof_end(); }
};

// write-once output planes: non-temporal stores keep them out of the L2 the kernel's re-read inputs live in
template <typename T>
__device__ __forceinline__ void store_stream(T* p, T v) {
#ifdef RTE_NO_NT_STORES
  *p = v;
#else
  __builtin_nontemporal_store(v, p);
#endif
}

inline unsigned cdiv(size_t a, size_t b) { return (unsigned)((a + b - 1) / b); }

}  // namespace rte
