// runtime.hip -- stream, scratch arena, host-pointer staging and kernel timing hooks.
//
// Drop-in semantics: the reference kernels own nothing -- every buffer is
// the caller's -- and take host arrays from the unchanged Fortran frontend.  This library accepts
// BOTH kinds of pointer on every array argument:
//   device pointer -> the kernel is launched in place, asynchronously, on the library stream;
//   host pointer   -> the array is staged through the scratch arena (H2D before, D2H after) and
//                     the call returns only after the stream has drained (functional mode).
#include <mutex>
#include <string>
#include <vector>

#include "common.h"

namespace rte {

static std::recursive_mutex g_mutex;  // entry points are serialised: stateless for the caller
static hipStream_t g_stream = nullptr;

// ---- side stream (opt-in, rte_hip_overlap_planck) ------------------------------------------------
// compute_tau_absorption and compute_Planck_source of one gas-optics step are independent of each other (both read
// the interpolation state; one writes tau, the other the sources), one is bound by LDS gathers and latency, the other
// by HBM stores, and each leaves a tail of idle CUs.  With the option on, a compute_Planck_source call that directly
// follows a compute_tau_absorption call -- both on device memory, disjoint outputs -- runs on a second stream that
// waits only for the work queued BEFORE the tau call; the library stream then waits for it, so every later call (and
// anything the caller queues afterwards) sees its results.  Like the deferred zero fill: only for callers that queue
// nothing of their own on the library stream between the two calls that writes compute_Planck_source's inputs.
static bool g_overlap = false;
static hipStream_t g_side = nullptr;
static bool g_on_side = false;
static hipEvent_t g_ev_fork = nullptr, g_ev_join = nullptr;
static bool g_fork_valid = false;       // g_ev_fork marks the start of the immediately preceding library call
static const char* g_fork_lo = nullptr; // that call's output range
static const char* g_fork_hi = nullptr;

hipStream_t stream() { return g_on_side ? g_side : g_stream; }

// ---- scratch arena (one per stream) --------------------------------------------------------
struct Block { char* base; size_t size; size_t used; };
static std::vector<Block> g_blocks_main, g_blocks_side;
static std::vector<Block>& blocks() { return g_on_side ? g_blocks_side : g_blocks_main; }

void* scratch(size_t bytes) {
  bytes = (bytes + 255) & ~size_t(255);
  auto& g_blocks = blocks();
  for (auto& b : g_blocks)
    if (b.size - b.used >= bytes) {
      void* p = b.base + b.used;
      b.used += bytes;
      return p;
    }
  size_t sz = bytes < (size_t(64) << 20) ? (size_t(64) << 20) : bytes;
  Block nb{nullptr, sz, bytes};
  HIP_CHECK(hipMalloc((void**)&nb.base, sz));
  g_blocks.push_back(nb);
  return nb.base;
}

static void scratch_reset() {
  auto& g_blocks = blocks();
  // keep one block big enough for the largest call seen so far; drop fragmentation
  if (g_blocks.size() > 1) {
    HIP_CHECK(hipStreamSynchronize(stream()));
    size_t total = 0;
    for (auto& b : g_blocks) { total += b.size; HIP_CHECK(hipFree(b.base)); }
    g_blocks.clear();
    Block nb{nullptr, total, 0};
    HIP_CHECK(hipMalloc((void**)&nb.base, total));
    g_blocks.push_back(nb);
  }
  for (auto& b : g_blocks) b.used = 0;
}

// compute_tau_absorption, before its first launch: everything queued so far is what a following
// compute_Planck_source may depend on
void fork_point(const void* out, size_t bytes) {
  if (!g_overlap) return;
  if (!g_ev_fork) {
    HIP_CHECK(hipEventCreateWithFlags(&g_ev_fork, hipEventDisableTiming));
    HIP_CHECK(hipEventCreateWithFlags(&g_ev_join, hipEventDisableTiming));
    HIP_CHECK(hipStreamCreateWithFlags(&g_side, hipStreamNonBlocking));  // no implicit ordering with the null stream
  }
  HIP_CHECK(hipEventRecord(g_ev_fork, g_stream));
  g_fork_lo = (const char*)out;
  g_fork_hi = g_fork_lo + bytes;
  g_fork_valid = true;
}

// ---- auxiliary stream inside one call --------------------------------------------------------
// compute_tau_absorption's direct-gather worklist (§4.0) is bound by the texture addresser and touches entries the slab
// kernel skips; on a second stream, forked after the geometry pre-pass and joined before the call returns, its single-wave
// blocks run in the register space the slab kernel's 10-wave blocks leave free instead of after it.  Internal to one
// call: whatever follows on the library stream sees both kernels' results.  rte_hip_aux_stream(0) switches it off.
static bool g_aux_on = true;
static hipStream_t g_aux = nullptr;
static hipEvent_t g_ev_aux_fork = nullptr, g_ev_aux_join = nullptr;

hipStream_t aux_fork() {
  if (!g_aux_on) return nullptr;
  if (!g_aux) {
    HIP_CHECK(hipEventCreateWithFlags(&g_ev_aux_fork, hipEventDisableTiming));
    HIP_CHECK(hipEventCreateWithFlags(&g_ev_aux_join, hipEventDisableTiming));
    // lowest priority: its waves take what the library stream's kernel leaves free, not the other way round
    int least = 0, greatest = 0;
    HIP_CHECK(hipDeviceGetStreamPriorityRange(&least, &greatest));
    HIP_CHECK(hipStreamCreateWithPriority(&g_aux, hipStreamNonBlocking, getenv("RTE_AUX_PRIO") ? atoi(getenv("RTE_AUX_PRIO")) : least));
  }
  HIP_CHECK(hipEventRecord(g_ev_aux_fork, stream()));
  HIP_CHECK(hipStreamWaitEvent(g_aux, g_ev_aux_fork, 0));
  return g_aux;
}

void aux_join() {
  HIP_CHECK(hipEventRecord(g_ev_aux_join, g_aux));
  HIP_CHECK(hipStreamWaitEvent(stream(), g_ev_aux_join, 0));
}

// ---- persistent slots ----------------------------------------------------------------------
struct Slot { void* p = nullptr; size_t bytes = 0; };
static Slot g_slots[16];
void* persistent(int slot, size_t bytes, bool* fresh) {
  Slot& s = g_slots[slot];
  if (fresh) *fresh = false;
  if (s.bytes < bytes) {
    if (s.p) { HIP_CHECK(hipStreamSynchronize(g_stream)); HIP_CHECK(hipFree(s.p)); }
    HIP_CHECK(hipMalloc(&s.p, bytes));
    s.bytes = bytes;
    if (fresh) *fresh = true;
  }
  return s.p;
}

// ---- pointer classification ------------------------------------------------------------------
// 0: ordinary host memory (staged through the arena), 1: device memory (launch in place, asynchronously),
// 2: host-VISIBLE memory a kernel can address (pinned / registered host memory, managed memory): launched in place
//    through `dev`, but the host may read it as soon as the call returns, so the call must drain the stream.
static int classify(const void* p, void** dev) {
  *dev = const_cast<void*>(p);
  if (!p) return 1;
  hipPointerAttribute_t a;
  hipError_t e = hipPointerGetAttributes(&a, p);
  if (e != hipSuccess) {
    (void)hipGetLastError();  // plain malloc'ed host memory: "invalid value"
    return 0;
  }
  if (a.type == hipMemoryTypeDevice) return 1;
  if (a.type == hipMemoryTypeManaged) return 2;
  if (a.type == hipMemoryTypeHost && a.devicePointer != nullptr) { *dev = a.devicePointer; return 2; }
  return 0;
}
bool is_device_pointer(const void* p) {
  void* d;
  return classify(p, &d) != 0;
}
bool is_device_memory(const void* p) {
  void* d;
  return p && classify(p, &d) == 1;
}

// ---- deferred zero fill (opt-in, rte_hip_defer_zero) --------------------------------------------
// The frontend zeroes tau and then calls compute_tau_absorption, which accumulates onto it
// (mo_gas_optics_rrtmgp.F90:637,679).  On the device that is a 12 GB memset plus a 12 GB read that only
// exist because the two steps are separate calls.  With the option on, zero_array_* on a device
// buffer is recorded instead of executed; compute_tau_absorption on exactly that buffer consumes the
// record and overwrites; ANY other library entry first materialises all recorded fills.  Only for
// callers that touch the buffer exclusively through this library between the two calls.
struct PendingZero { void* p; size_t bytes; };
static std::vector<PendingZero> g_pending;
static bool g_defer_zero = false;

bool defer_zero_enabled() { return g_defer_zero; }
void defer_zero(void* p, size_t bytes) {
  std::lock_guard<std::recursive_mutex> l(g_mutex);
  g_pending.push_back(PendingZero{p, bytes});
}
bool take_pending_zero(const void* p, size_t bytes) {
  std::lock_guard<std::recursive_mutex> l(g_mutex);
  for (size_t i = 0; i < g_pending.size(); ++i)
    if (g_pending[i].p == p && g_pending[i].bytes == bytes) {
      g_pending.erase(g_pending.begin() + i);
      return true;
    }
  return false;
}
void flush_pending_zeros() {
  std::lock_guard<std::recursive_mutex> l(g_mutex);
  for (auto& z : g_pending) HIP_CHECK(hipMemsetAsync(z.p, 0, z.bytes, g_stream));
  g_pending.clear();
}

// ---- Call ---------------------------------------------------------------------------------------
static long g_seq = 0;
long call_seq() { return g_seq; }  // number of the current (innermost) API call

Call::Call(const char* n) : name(n) {
  g_mutex.lock();
  ++g_seq;
  fork_candidate_ = g_fork_valid;  // the previous call left a fork point (it is consumed or dropped by this call)
  g_fork_valid = false;
  if (!g_pending.empty()) {
    // recorded fills are materialised on the library stream, i.e. BEHIND the previous call's kernels; a call forked to the
    // side stream waits only for what was queued before that previous call, so a fill of one of its outputs could land
    // after its own stores: a call that had to materialise fills is never forked
    fork_candidate_ = false;
    flush_pending_zeros();
  }
  scratch_reset();
}

// Move the rest of this call (launches, scratch, timing events) to the side stream if that is safe: nothing of this
// call has been staged or queued yet, all its arrays are device memory, and its outputs do not touch the range the
// previous call writes.  Must be called after the in()/out() conversions and before the first launch.
bool Call::try_fork(const void* const* outs, const size_t* bytes, int n) {
  if (!g_overlap || !fork_candidate_ || n_back_ > 0 || staged_in_ || host_visible_ || n_host_tmp_ > 0) return false;
  for (int i = 0; i < n; ++i) {
    const char* lo = (const char*)outs[i];
    if (!lo || !is_device_memory(lo)) return false;
    if (lo < g_fork_hi && lo + bytes[i] > g_fork_lo) return false;
  }
  HIP_CHECK(hipStreamWaitEvent(g_side, g_ev_fork, 0));
  g_on_side = true;
  forked_ = true;
  scratch_reset();  // the side arena: its previous user was the previous forked call, which the library stream has joined
  return true;
}

void* Call::stage(void* p, size_t bytes, bool copy_in, bool copy_out) {
  if (!p || bytes == 0) return p;
  void* dv;
  const int kind = classify(p, &dv);
  if (kind == 1) return p;
  if (kind == 2) { host_visible_ = true; return dv; }  // in place, but synchronous for the caller (see ~Call)
  void* d = scratch(bytes);
  if (copy_in) {
    HIP_CHECK(hipMemcpyAsync(d, p, bytes, hipMemcpyHostToDevice, g_stream));
    staged_in_ = true;
  }
  if (copy_out) {
    if (n_back_ >= 16) { fprintf(stderr, "rte_rrtmgp_hip: too many staged outputs\n"); abort(); }
    back_[n_back_++] = Back{p, d, bytes};
  }
  return d;
}

const void* Call::to_host(const void* p, size_t bytes) {
  if (!p || bytes == 0 || !is_device_pointer(p)) return p;
  if (n_host_tmp_ >= (int)(sizeof(host_tmp_) / sizeof(host_tmp_[0]))) {
    fprintf(stderr, "rte_rrtmgp_hip: %s: too many host copies of device tables\n", name);
    abort();
  }
  void* h = malloc(bytes);
  HIP_CHECK(hipMemcpyAsync(h, p, bytes, hipMemcpyDeviceToHost, stream()));
  HIP_CHECK(hipStreamSynchronize(stream()));
  host_tmp_[n_host_tmp_++] = h;
  return h;
}

Call::~Call() {
  for (int i = 0; i < n_back_; ++i)
    HIP_CHECK(hipMemcpyAsync(back_[i].host, back_[i].dev, back_[i].bytes, hipMemcpyDeviceToHost, g_stream));
  // host arrays (staged, or host-visible memory used in place): the caller owns them again when the call returns
  if (n_back_ > 0 || staged_in_ || host_visible_) HIP_CHECK(hipStreamSynchronize(g_stream));
  for (int i = 0; i < n_host_tmp_; ++i) free(host_tmp_[i]);
  if (forked_) {  // join: the library stream (and whatever is queued on it from now on) waits for this call
    HIP_CHECK(hipEventRecord(g_ev_join, g_side));
    HIP_CHECK(hipStreamWaitEvent(g_stream, g_ev_join, 0));
    g_on_side = false;
  }
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) {
    fprintf(stderr, "rte_rrtmgp_hip: %s: launch error: %s\n", name, hipGetErrorString(e));
    abort();
  }
  g_mutex.unlock();
}

// ---- kernel timing ------------------------------------------------------------------------------
struct ProfEntry { std::string name; std::vector<std::pair<hipEvent_t, hipEvent_t>> ev; double ms = 0; long n = 0; };
static bool g_prof_on = false;
static std::string g_prof_only;  // non-empty: only this scope is timed (every event pair costs microseconds on the GPU timeline)
static std::vector<ProfEntry> g_prof;
static ProfEntry* g_cur = nullptr;
static hipEvent_t g_cur_start;

void prof_begin(const char* kernel) {
  g_cur = nullptr;
  if (!g_prof_on) return;
  if (!g_prof_only.empty() && g_prof_only != kernel) return;
  for (auto& e : g_prof)
    if (e.name == kernel) g_cur = &e;
  if (!g_cur) {
    g_prof.push_back(ProfEntry{kernel});
    g_cur = &g_prof.back();
  }
  HIP_CHECK(hipEventCreate(&g_cur_start));
  HIP_CHECK(hipEventRecord(g_cur_start, stream()));
}
void prof_end() {
  if (!g_prof_on || !g_cur) return;
  hipEvent_t stop;
  HIP_CHECK(hipEventCreate(&stop));
  HIP_CHECK(hipEventRecord(stop, stream()));
  g_cur->ev.emplace_back(g_cur_start, stop);
  g_cur = nullptr;
}
static void prof_resolve() {
  for (auto& e : g_prof) {
    for (auto& p : e.ev) {
      HIP_CHECK(hipEventSynchronize(p.second));
      float ms = 0;
      HIP_CHECK(hipEventElapsedTime(&ms, p.first, p.second));
      e.ms += ms;
      e.n += 1;
      HIP_CHECK(hipEventDestroy(p.first));
      HIP_CHECK(hipEventDestroy(p.second));
    }
    e.ev.clear();
  }
}

}  // namespace rte

namespace rte { void release_gas_optics_buffers(); }  // gas_optics.hip

// ---- library-extension entry points (not part of the reference interface) ------------------------
extern "C" {

int rte_hip_set_stream(void* s) {
  std::lock_guard<std::recursive_mutex> l(rte::g_mutex);
  if ((hipStream_t)s == rte::g_stream) return 0;
  // work queued on the old stream still uses the scratch arena, the persistent slots and recorded zero fills:
  // materialise the fills there and drain it before anything is launched on the new stream
  rte::flush_pending_zeros();
  HIP_CHECK(hipStreamSynchronize(rte::g_stream));  // (forked calls have been joined into it)
  rte::g_fork_valid = false;
  rte::g_stream = (hipStream_t)s;
  return 0;
}
int rte_hip_sync(void) {
  rte::flush_pending_zeros();
  HIP_CHECK(hipStreamSynchronize(rte::g_stream));
  return 0;
}
// defer zero_array_* on device buffers until compute_tau_absorption consumes them (see runtime.hip)
int rte_hip_defer_zero(int on) {
  rte::flush_pending_zeros();
  rte::g_defer_zero = on != 0;
  return 0;
}
// run compute_Planck_source concurrently with the compute_tau_absorption call it directly follows (see runtime.hip)
int rte_hip_overlap_planck(int on) {
  std::lock_guard<std::recursive_mutex> l(rte::g_mutex);
  rte::g_overlap = on != 0;
  rte::g_fork_valid = false;
  return 0;
}
// run the direct-gather worklist of compute_tau_absorption on a second stream inside the call (default on)
int rte_hip_aux_stream(int on) {
  std::lock_guard<std::recursive_mutex> l(rte::g_mutex);
  rte::g_aux_on = on != 0;
  return 0;
}
int rte_hip_device_count(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) { (void)hipGetLastError(); return 0; }
  return n;
}
int rte_hip_profile_enable(int on) {
  std::lock_guard<std::recursive_mutex> l(rte::g_mutex);
  rte::g_prof_on = on != 0;
  return 0;
}
// time only the scope of this name (nullptr or "": all scopes)
int rte_hip_profile_only(const char* name) {
  std::lock_guard<std::recursive_mutex> l(rte::g_mutex);
  rte::g_prof_only = name ? name : "";
  return 0;
}
int rte_hip_profile_reset(void) {
  std::lock_guard<std::recursive_mutex> l(rte::g_mutex);
  rte::prof_resolve();
  rte::g_prof.clear();
  return 0;
}
int rte_hip_profile_count(void) {
  std::lock_guard<std::recursive_mutex> l(rte::g_mutex);
  rte::prof_resolve();
  return (int)rte::g_prof.size();
}
// i-th timed kernel: name copied into buf, launches and total milliseconds returned
int rte_hip_profile_get(int i, char* buf, int buflen, long long* launches, double* total_ms) {
  std::lock_guard<std::recursive_mutex> l(rte::g_mutex);
  if (i < 0 || i >= (int)rte::g_prof.size()) return -1;
  snprintf(buf, buflen, "%s", rte::g_prof[i].name.c_str());
  *launches = rte::g_prof[i].n;
  *total_ms = rte::g_prof[i].ms;
  return 0;
}
// release every device buffer held by the library (arena + persistent slots)
int rte_hip_release(void) {
  std::lock_guard<std::recursive_mutex> l(rte::g_mutex);
  rte::flush_pending_zeros();
  HIP_CHECK(hipStreamSynchronize(rte::g_stream));
  if (rte::g_side) HIP_CHECK(hipStreamSynchronize(rte::g_side));
  if (rte::g_aux) HIP_CHECK(hipStreamSynchronize(rte::g_aux));
  rte::release_gas_optics_buffers();
  for (auto* v : {&rte::g_blocks_main, &rte::g_blocks_side}) {
    for (auto& b : *v) HIP_CHECK(hipFree(b.base));
    v->clear();
  }
  for (auto& s : rte::g_slots) {
    if (s.p) HIP_CHECK(hipFree(s.p));
    s = rte::Slot{};
  }
  return 0;
}
}
