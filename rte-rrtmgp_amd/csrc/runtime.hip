// runtime.hip -- contexts (stream, scratch arena, persistent buffers), host-pointer staging, host-mirror mode, error
// channel and kernel timing hooks.
//
// Drop-in semantics: the reference kernels own nothing -- every buffer is the caller's -- and
// take host arrays from the unchanged Fortran frontend.  This library accepts BOTH kinds of pointer on every array
// argument:
//   device pointer -> the kernel is launched in place, asynchronously, on the context's stream;
//   host pointer   -> the array is staged through the scratch arena (H2D before, D2H after) and the call returns only
//                     after the stream has drained (functional mode), or, in host-mirror mode, is kept on the device
//                     between calls (see below).
//
// CONTEXTS.  All mutable state of the library -- stream, arenas, persistent buffers, recorded zero fills, mirrors, plan
// caches of the gas-optics kernels, timing -- belongs to a Context.  Every host thread has a current context
// (rte_hip_ctx_set_current); threads that never set one share the process-wide default context.  A context serialises
// the calls made on it with its own mutex, so two threads on two contexts run concurrently (each on its own stream),
// which is what the reference intends for calls on distinct buffers (examples/all-sky/rrtmgp_allsky.F90:331).
#include <errno.h>
#include <fcntl.h>
#include <string.h>
#include <sys/uio.h>
#include <unistd.h>

#include <atomic>
#include <dlfcn.h>
#include <chrono>
#include <exception>
#include <mutex>
#include <string>
#include <algorithm>
#include <vector>

#include <ctype.h>
#include <sched.h>

#include "common.h"

namespace rte {

struct Block { char* base; size_t size; size_t used; };
struct Slot { void* p = nullptr; size_t bytes = 0; };
struct PendingZero { void* p; size_t bytes; };
struct Mirror {
  char* host; size_t bytes;
  char* dev; size_t cap;
  long last_use;
  unsigned long long magic;
  const char* producer;
  bool zero_pending;  // entirely zero by a recorded zero_array; the device copy has not been filled yet
};
struct FreeBuf { char* dev; size_t cap; };
struct TableCopy { const char* host; size_t bytes; unsigned long long fp; char* dev; long last_use; };
struct ProfEntry { std::string name; std::vector<std::pair<hipEvent_t, hipEvent_t>> ev; double ms = 0; long n = 0; };

struct Context {
  std::recursive_mutex mutex;  // calls on one context are serialised: stateless for the caller
  int device = -1;             // -1: whatever device is current when the context is first used
  hipStream_t stream = nullptr;
  bool own_stream = false;
  // ---- rte_hip_graph_begin / _end: the stream a region is captured from, and the context's stream meanwhile
  hipStream_t graph_stream = nullptr, graph_saved = nullptr;
  bool graph_saved_valid = false, graph_saved_aux = true;
  // a captured graph bakes in addresses of the scratch arena and of the persistent slots: every time one of those buffers is
  // freed or reallocated the epoch moves, and a graph of an earlier epoch refuses to launch (rte_hip_graph_launch: -4)
  long addr_epoch = 0;
  // ---- side stream (opt-in, rte_hip_overlap_planck)
  bool overlap = false;
  hipStream_t side = nullptr;
  bool on_side = false;
  hipEvent_t ev_fork = nullptr, ev_join = nullptr;
  bool fork_valid = false;       // ev_fork marks the start of the immediately preceding library call
  const char* fork_lo = nullptr; // that call's output range
  const char* fork_hi = nullptr;
  // ---- scratch arena (one per stream), persistent slots
  std::vector<Block> blocks_main, blocks_side;
  Slot slots[16];
  // ---- auxiliary stream inside one call
  bool aux_on = true;
  hipStream_t aux = nullptr;
  hipEvent_t ev_aux_fork = nullptr, ev_aux_join = nullptr;
  // ---- deferred zero fill
  std::vector<PendingZero> pending;
  // (RTE_HIP_DEFER_ZERO=1: the opt-in for an unchanged device-pointer binary that cannot call rte_hip_defer_zero)
  bool defer_zero = getenv("RTE_HIP_DEFER_ZERO") && atoi(getenv("RTE_HIP_DEFER_ZERO")) > 0;
  // deferred LW sources (see common.h): at most a few records (one per source-function object in flight)
  std::vector<PendingSources> pending_src;
  PendingSourcesOps src_ops{nullptr, nullptr, nullptr};
  bool defer_sources = getenv("RTE_HIP_DEFER_SOURCES") && atoi(getenv("RTE_HIP_DEFER_SOURCES")) > 0;
  long seq = 0;
  // ---- host-mirror mode
  std::vector<Mirror> mirrors;
  std::vector<FreeBuf> mirror_free;
  std::vector<TableCopy> tables;  // host-mirror mode: device copies of host k-distribution tables
  int mirror_mode = -1;          // -1: take RTE_HIP_HOST_MIRROR at the first call
  size_t mirror_total = 0;       // device bytes held by mirrors and the free list
  size_t mirror_limit = 0;
  long mirror_max_age = 64;
  unsigned long long magic_state = 0x9E3779B97F4A7C15ull;
  hipEvent_t ev_h2d = nullptr;
  // pinned staging ring for host-to-device copies of pageable arrays (h2d())
  static constexpr int kRing = 4;
  static constexpr size_t kRingChunk = size_t(4) << 20;
  char* ring[kRing] = {nullptr, nullptr, nullptr, nullptr};
  hipEvent_t ring_ev[kRing] = {nullptr, nullptr, nullptr, nullptr};
  bool ring_busy[kRing] = {false, false, false, false};
  int ring_cur = -1;       // slot being filled
  size_t ring_fill = 0;    // bytes of it handed to the DMA engine so far
  int ring_mode = -1;  // -1: take RTE_HIP_H2D_RING at the first copy (default on)
  long long mstat[8] = {0, 0, 0, 0, 0, 0, 0, 0};  // hits, mirrors made, H2D bytes, D2H bytes, dropped (host changed), dropped (overlap), aged out, zero fills elided
  long long table_hits = 0, table_uploads = 0;
  double t_call = 0, t_h2d = 0, t_wait = 0, t_find = 0;  // host wall-clock inside the host-array path
  long n_calls = 0;
  std::chrono::steady_clock::time_point call_t0;
  bool report_on = false;        // RTE_HIP_STAGING_REPORT: also keep the wall-clock per entry point
  std::vector<std::pair<const char*, std::pair<double, long>>> t_entry;
  // ---- host-mirror mode: device copies of INPUT arrays the host produced, each with a host-side shadow of what was uploaded
  // (see Call::stage): the frontend hands the same play / tlay / col_gas to three kernel calls in a row and the same constant
  // emissivity / incident-flux / secant fields to every solver call
  struct InputCopy { const char* host; size_t bytes; char* shadow; char* dev; long last_use; };
  std::vector<InputCopy> inputs;
  size_t inputs_total = 0;
  int input_cache = -1;  // RTE_HIP_INPUT_CACHE (default on in host-mirror mode)
  long long input_hits = 0, input_saved = 0, input_made = 0, input_evicted = 0;
  std::vector<std::pair<long long, long long>> b_entry;  // per entry point: bytes copied host-to-device / device-to-host
  long long call_h2d0 = 0, call_d2h0 = 0;
  // ---- timing
  bool prof_on = false;
  std::string prof_only;  // non-empty: only this scope is timed
  std::vector<ProfEntry> prof;
  ProfEntry* cur = nullptr;
  hipEvent_t cur_start = nullptr;
  // ---- error channel
  // rte_hip_error_mode(1) / RTE_HIP_ERROR_MODE=1: record and return instead of abort() (a host model polls rte_hip_last_error)
  bool sticky_errors = getenv("RTE_HIP_ERROR_MODE") && atoi(getenv("RTE_HIP_ERROR_MODE")) > 0;
  int last_error = 0;            // hipError_t of the first failure since rte_hip_clear_error()
  std::string last_error_msg;
  // ---- state of other translation units (plan caches of the gas-optics kernels), created on demand
  void* gas = nullptr;
  void (*gas_free)(void*) = nullptr;
};

static Context& default_context() {
  static Context* c = new Context();  // never destroyed: library calls may come from atexit handlers
  return *c;
}
static thread_local Context* t_ctx = nullptr;
// RTE_HIP_THREAD_CONTEXTS=1: a thread that never chose a context gets one of its own (own stream, arena, mirrors) at its
// first library call instead of sharing the default context -- the threads of an OpenMP loop over column blocks in an
// UNCHANGED host program (which cannot call rte_hip_ctx_create) then run concurrently.  The first thread to call keeps
// the default context.  Such contexts live until the process ends.
static Context* auto_context() {
  static const bool on = getenv("RTE_HIP_THREAD_CONTEXTS") && atoi(getenv("RTE_HIP_THREAD_CONTEXTS")) > 0;
  if (!on) return nullptr;
  static std::mutex m;
  static bool default_taken = false;
  std::lock_guard<std::mutex> l(m);
  if (!default_taken) { default_taken = true; return &default_context(); }
  auto* c = new Context();
  const Context& d = default_context();
  c->overlap = d.overlap; c->aux_on = d.aux_on; c->defer_zero = d.defer_zero; c->defer_sources = d.defer_sources; c->sticky_errors = d.sticky_errors;
  if (hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking) == hipSuccess) c->own_stream = true;
  return c;
}
// RTE_HIP_BIND_NUMA=1 (host-array callers): a thread's first library call pins it to the CPUs of the NUMA node the GPU hangs
// off, so that the thread, its pageable arrays (first touch) and the pinned staging buffers it allocates stay on the socket
// whose PCIe root the device is on.  Pass times of the unchanged Fortran frontend on host arrays were bimodal (0.05 s / 0.3 s,
// whole phases of a run) while the scheduler was free to move the eight OpenMP threads between the sockets of a 256-CPU
// host.  Opt-in: changing a caller's thread affinity is not something a library does unasked.
static void bind_thread_to_gpu_node() {
  static const bool on = getenv("RTE_HIP_BIND_NUMA") && atoi(getenv("RTE_HIP_BIND_NUMA")) > 0;
  static thread_local bool done = false;
  if (!on || done) return;
  done = true;
  int dev = 0;
  char bus[64] = {0};
  if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetPCIBusId(bus, (int)sizeof(bus), dev) != hipSuccess) return;
  for (char* q = bus; *q; ++q) *q = (char)tolower(*q);
  char path[256];
  snprintf(path, sizeof(path), "/sys/bus/pci/devices/%s/numa_node", bus);
  int node = -1;
  if (FILE* f = fopen(path, "r")) { if (fscanf(f, "%d", &node) != 1) node = -1; fclose(f); }
  if (node < 0) return;
  snprintf(path, sizeof(path), "/sys/devices/system/node/node%d/cpulist", node);
  FILE* f = fopen(path, "r");
  if (!f) return;
  cpu_set_t set, allowed;
  CPU_ZERO(&set);
  int a = 0, b = 0;
  char sep = 0;
  while (fscanf(f, "%d", &a) == 1) {  // "0-63,128-191"
    b = a;
    int ch = fgetc(f);
    if (ch == '-') { if (fscanf(f, "%d", &b) != 1) b = a; ch = fgetc(f); }
    for (int k = a; k <= b && k < CPU_SETSIZE; ++k) CPU_SET(k, &set);
    sep = (char)ch;
    if (sep != ',') break;
  }
  fclose(f);
  if (sched_getaffinity(0, sizeof(allowed), &allowed) == 0) CPU_AND(&set, &set, &allowed);  // (never beyond what the process may use)
  if (CPU_COUNT(&set) > 0) (void)sched_setaffinity(0, sizeof(set), &set);
}
Context& ctx() {
  if (t_ctx) return *t_ctx;
  bind_thread_to_gpu_node();
  if (Context* a = auto_context()) { t_ctx = a; return *a; }
  return default_context();
}
#define C ctx()

hipStream_t stream() { Context& c = C; return c.on_side ? c.side : c.stream; }
void* gas_state(void* (*make)(), void (*destroy)(void*)) {
  Context& c = C;
  if (!c.gas) { c.gas = make(); c.gas_free = destroy; }
  return c.gas;
}

CtxLock::CtxLock() : ctx_(&C) { ((Context*)ctx_)->mutex.lock(); }
CtxLock::~CtxLock() { ((Context*)ctx_)->mutex.unlock(); }

// ---- error channel ---------------------------------------------------------------------------------
// The reference kernel interface has no error channel (all entry points are void).  A failing HIP call throws rte::Error;
// the entry point's handler (RTE_CATCH) either prints and abort()s (default: a host model must not continue on garbage) or,
// after rte_hip_error_mode(1), records the error in the context and returns: every later call on that context is then a
// no-op until rte_hip_clear_error(), and the host model polls rte_hip_last_error().
void fail(hipError_t e, const char* expr, const char* file, int line) {
  char buf[512];
  snprintf(buf, sizeof buf, "%s failed at %s:%d: %s", expr, file, line, hipGetErrorString(e));
  if (std::uncaught_exceptions() > 0) {  // a second failure while unwinding from the first
    fprintf(stderr, "rte_rrtmgp_hip: %s\n", buf);
    abort();
  }
  throw Error{(int)e, buf};
}
void on_error(const char* entry, const Error& e) {
  Context& c = C;
  if (!c.sticky_errors) {
    fprintf(stderr, "rte_rrtmgp_hip: %s: %s\n", entry, e.what.c_str());
    abort();
  }
  if (c.last_error == 0) {
    c.last_error = e.code ? e.code : -1;
    c.last_error_msg = std::string(entry) + ": " + e.what;
  }
}

// ---- scratch arena -----------------------------------------------------------------------------------
static std::vector<Block>& blocks() { Context& c = C; return c.on_side ? c.blocks_side : c.blocks_main; }

void* scratch(size_t bytes) {
  bytes = (bytes + 255) & ~size_t(255);
  auto& bl = blocks();
  for (auto& b : bl)
    if (b.size - b.used >= bytes) {
      void* p = b.base + b.used;
      b.used += bytes;
      return p;
    }
  size_t sz = bytes < (size_t(64) << 20) ? (size_t(64) << 20) : bytes;
  Block nb{nullptr, sz, bytes};
  HIP_CHECK(hipMalloc((void**)&nb.base, sz));
  bl.push_back(nb);
  return nb.base;
}

static void scratch_reset() {
  auto& bl = blocks();
  // keep one block big enough for the largest call seen so far; drop fragmentation
  if (bl.size() > 1) {
    HIP_CHECK(hipStreamSynchronize(stream()));
    ++C.addr_epoch;
    size_t total = 0;
    for (auto& b : bl) { total += b.size; HIP_CHECK(hipFree(b.base)); }
    bl.clear();
    Block nb{nullptr, total, 0};
    HIP_CHECK(hipMalloc((void**)&nb.base, total));
    bl.push_back(nb);
  }
  for (auto& b : bl) b.used = 0;
}

// ---- side stream (opt-in, rte_hip_overlap_planck) ------------------------------------------------
// compute_tau_absorption and compute_Planck_source of one gas-optics step are independent of each other (both read
// the interpolation state; one writes tau, the other the sources), one is bound by LDS gathers and latency, the other
// by HBM stores, and each leaves a tail of idle CUs.  With the option on, a compute_Planck_source call that directly
// follows a compute_tau_absorption call -- both on device memory, disjoint outputs -- runs on a second stream that
// waits only for the work queued BEFORE the tau call; the library stream then waits for it, so every later call (and
// anything the caller queues afterwards) sees its results.  Like the deferred zero fill: only for callers that queue
// nothing of their own on the library stream between the two calls that writes compute_Planck_source's inputs.
// compute_tau_absorption, before its first launch: everything queued so far is what a following
// compute_Planck_source may depend on
void fork_point(const void* out, size_t bytes) {
  Context& c = C;
  if (!c.overlap) return;
  if (!c.ev_fork) {
    HIP_CHECK(hipEventCreateWithFlags(&c.ev_fork, hipEventDisableTiming));
    HIP_CHECK(hipEventCreateWithFlags(&c.ev_join, hipEventDisableTiming));
    HIP_CHECK(hipStreamCreateWithFlags(&c.side, hipStreamNonBlocking));  // no implicit ordering with the null stream
  }
  HIP_CHECK(hipEventRecord(c.ev_fork, c.stream));
  c.fork_lo = (const char*)out;
  c.fork_hi = c.fork_lo + bytes;
  c.fork_valid = true;
}

// ---- auxiliary stream inside one call --------------------------------------------------------
// compute_tau_absorption's direct-gather worklist is bound by the texture addresser and touches entries the
// slab kernel skips; on a second stream, forked after the geometry pre-pass and joined before the call returns, its
// single-wave blocks run in the register space the slab kernel's 10-wave blocks leave free instead of after it.
// Internal to one call: whatever follows on the library stream sees both kernels' results.  rte_hip_aux_stream(0)
// switches it off.
hipStream_t aux_fork() {
  Context& c = C;
  if (!c.aux_on) return nullptr;
  if (!c.aux) {
    HIP_CHECK(hipEventCreateWithFlags(&c.ev_aux_fork, hipEventDisableTiming));
    HIP_CHECK(hipEventCreateWithFlags(&c.ev_aux_join, hipEventDisableTiming));
    // lowest priority: its waves take what the library stream's kernel leaves free, not the other way round
    int least = 0, greatest = 0;
    HIP_CHECK(hipDeviceGetStreamPriorityRange(&least, &greatest));
    HIP_CHECK(hipStreamCreateWithPriority(&c.aux, hipStreamNonBlocking, getenv("RTE_AUX_PRIO") ? atoi(getenv("RTE_AUX_PRIO")) : least));
  }
  HIP_CHECK(hipEventRecord(c.ev_aux_fork, stream()));
  HIP_CHECK(hipStreamWaitEvent(c.aux, c.ev_aux_fork, 0));
  return c.aux;
}

void aux_join() {
  Context& c = C;
  HIP_CHECK(hipEventRecord(c.ev_aux_join, c.aux));
  HIP_CHECK(hipStreamWaitEvent(stream(), c.ev_aux_join, 0));
}

// ---- persistent slots ----------------------------------------------------------------------
void* persistent(int slot, size_t bytes, bool* fresh) {
  Context& c = C;
  Slot& s = c.slots[slot];
  if (fresh) *fresh = false;
  if (s.bytes < bytes) {
    if (s.p) { HIP_CHECK(hipStreamSynchronize(c.stream)); HIP_CHECK(hipFree(s.p)); ++c.addr_epoch; }
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
// OpenMP target offload in the CALLER (the reference frontend built with `flang -fopenmp --offload-arch=gfx950` keeps its
// arrays on the device with its own `!$omp target data` regions, rte/frontend/mo_rte_lw.F90:327-365,
// rrtmgp/frontend/mo_gas_optics_rrtmgp.F90:594-609): such a caller passes the HOST address of a MAPPED array.  If the
// process has an OpenMP offload runtime (omp_get_mapped_ptr, OpenMP 5.1, resolved at run time: the library does not link
// libomptarget), the array's device address is used in place -- no staging, no mirror.  RTE_HIP_OMP_MAPPED=0 switches
// the look-up off.
static void* omp_mapped(const void* p) {
  typedef void* (*get_mapped_t)(const void*, int);
  typedef int (*get_dev_t)(void);
  static const get_mapped_t get_mapped = (get_mapped_t)dlsym(RTLD_DEFAULT, "omp_get_mapped_ptr");
  static const get_dev_t get_dev = (get_dev_t)dlsym(RTLD_DEFAULT, "omp_get_default_device");
  static const bool on = get_mapped && get_dev && !(getenv("RTE_HIP_OMP_MAPPED") && atoi(getenv("RTE_HIP_OMP_MAPPED")) == 0);
  static const bool verbose = getenv("RTE_HIP_OMP_MAPPED") && atoi(getenv("RTE_HIP_OMP_MAPPED")) == 2;
  if (verbose) {
    static bool said = false;
    if (!said) { said = true; fprintf(stderr, "rte_rrtmgp_hip: omp_get_mapped_ptr %p, omp_get_default_device %p, look-up %s\n", (void*)get_mapped, (void*)get_dev, on ? "on" : "off"); }
  }
  if (!on) return nullptr;
  void* d = get_mapped(p, get_dev());
  if (verbose) fprintf(stderr, "rte_rrtmgp_hip: omp_get_mapped_ptr(%p) = %p\n", p, d);
  return (d && d != p) ? d : nullptr;
}
static int classify(const void* p, void** dev) {
  *dev = const_cast<void*>(p);
  if (!p) return 1;
  hipPointerAttribute_t a;
  hipError_t e = hipPointerGetAttributes(&a, p);
  if (e != hipSuccess) {
    (void)hipGetLastError();  // plain malloc'ed host memory: "invalid value"
    // ... unless the caller's OpenMP runtime has it mapped: launched in place on the device copy, and (like host-visible
    // memory) the call drains the stream before it returns -- the caller's next target region runs on a queue of its own
    if (void* d = omp_mapped(p)) { *dev = d; return 2; }
    return 0;
  }
  if (a.type == hipMemoryTypeDevice) return 1;
  if (a.type == hipMemoryTypeManaged) return 2;
  if (void* d = omp_mapped(p)) { *dev = d; return 2; }  // (an array the caller's OpenMP runtime has mapped may also be page-locked by it)
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
bool defer_zero_enabled() { return C.defer_zero; }
void defer_zero(void* p, size_t bytes) {
  Context& c = C;
  std::lock_guard<std::recursive_mutex> l(c.mutex);
  c.pending.push_back(PendingZero{p, bytes});
}
bool take_pending_zero(const void* p, size_t bytes) {
  Context& c = C;
  std::lock_guard<std::recursive_mutex> l(c.mutex);
  for (size_t i = 0; i < c.pending.size(); ++i)
    if (c.pending[i].p == p && c.pending[i].bytes == bytes) {
      c.pending.erase(c.pending.begin() + i);
      return true;
    }
  return false;
}
void flush_pending_zeros() {
  Context& c = C;
  std::lock_guard<std::recursive_mutex> l(c.mutex);
  for (auto& z : c.pending) HIP_CHECK(hipMemsetAsync(z.p, 0, z.bytes, c.stream));
  c.pending.clear();
}

// ---- deferred LW sources (opt-in, rte_hip_defer_sources; see common.h) ------------------------------------------------
// compute_Planck_source and lw_solver_noscat are separate calls of the reference interface, with 26 GB of lay_source /
// lev_source written by one and read by the other at 1e5 x 60 x 256; in factored form (Planck fraction per g-point, Planck
// function per band) it is 13.7 GB.  With the option on, an UNCHANGED device-pointer binary gets the factored step: the
// arrays are materialised (in place, bit-identical to the plain call) as soon as the library is handed one of them for
// anything but that solve.  Same promise as the deferred zero fill: between the two calls the caller touches these arrays
// only through this library (rte_hip_sync materialises everything).
bool defer_sources_enabled() { return C.defer_sources; }
void defer_sources(const PendingSources& s, const PendingSourcesOps* ops) {
  Context& c = C;
  std::lock_guard<std::recursive_mutex> l(c.mutex);
  if (ops) c.src_ops = *ops;
  c.pending_src.push_back(s);
}
// may the record still be used?  Unconsumed: the caller's promise holds.  Consumed: only if lay_source still holds the fraction.
// (inside a graph capture the region's contract holds -- library calls only -- and the check, which synchronises, cannot run)
static bool sources_usable(const PendingSources& s) { return !s.consumed || C.graph_saved_valid || C.src_ops.still_factored(s); }
bool take_pending_sources(const void* lay, const void* lev, PendingSources* out) {
  Context& c = C;
  std::lock_guard<std::recursive_mutex> l(c.mutex);
  for (size_t i = 0; i < c.pending_src.size(); ++i)
    if (c.pending_src[i].lay == lay && c.pending_src[i].lev == lev) {
      *out = c.pending_src[i];
      c.pending_src.erase(c.pending_src.begin() + i);
      return sources_usable(*out);  // (a stale record is gone now: the arrays are taken as they are)
    }
  return false;
}
void sources_consumed(PendingSources s) {
  Context& c = C;
  std::lock_guard<std::recursive_mutex> l(c.mutex);
  if (!s.consumed) {
    s.consumed = true;
    c.src_ops.take_sample(s);
  }
  c.pending_src.push_back(s);
}
void flush_pending_sources_except(const void* lay, const void* lev) {
  Context& c = C;
  std::lock_guard<std::recursive_mutex> l(c.mutex);
  for (size_t i = c.pending_src.size(); i-- > 0;) {
    const PendingSources s = c.pending_src[i];
    if (s.lay == lay && s.lev == lev) continue;
    c.pending_src.erase(c.pending_src.begin() + i);
    if (sources_usable(s)) c.src_ops.expand(s);
  }
}
void flush_pending_sources() { flush_pending_sources_except(nullptr, nullptr); }
// a library call is handed `bytes` at p: as an input (or in / out) the array is materialised first, as a pure output the record
// goes (the factors in it are about to be overwritten).  Ranges are matched by their start: the frontend passes whole arrays;
// a consumed record is used only if its fingerprint still matches (see common.h).
static void sources_touch(const void* p, bool copy_in) {
  Context& c = C;
  for (size_t i = 0; i < c.pending_src.size(); ++i)
    if (c.pending_src[i].lay == p || c.pending_src[i].lev == p) {
      const PendingSources s = c.pending_src[i];
      c.pending_src.erase(c.pending_src.begin() + i);
      if (copy_in && sources_usable(s)) c.src_ops.expand(s);
      return;
    }
}

// ---- host-mirror mode (opt-in: rte_hip_host_mirror(1) or RTE_HIP_HOST_MIRROR=1) ---------------------------------
// The unchanged Fortran frontend passes pageable HOST arrays.  Staged naively, every call copies its inputs up and its
// outputs back, so the interpolation state, tau and the Planck sources (0.95 MB per column) cross PCIe twice although
// no host code ever looks at them between gas_optics and rte_lw.  In this mode the outputs that entry points mark as
// lazy (Call::out_lazy / inout_lazy: arrays the reference frontend only hands on to the next kernel) stay on the device:
//   * the device copy ("mirror") is keyed by the host address range; the host array is NOT written;
//   * a later call that receives that range (or a part of it) as an argument is served from the device copy;
//   * to notice that the host reused or overwrote the memory in between (Fortran automatic / allocatable arrays come
//     back at the same addresses), a few 16-byte CANARIES are written into the host array when the mirror is made --
//     its contents are unspecified until a write-back anyway -- and verified, through /proc/self/mem so that a freed
//     and unmapped range cannot fault, before the mirror is trusted.  A host program that filled the array in between
//     has destroyed them: the mirror is dropped and the host contents are staged as usual;
//   * small outputs (fluxes, col_dry, by-band and broadband reductions: everything an entry point does not mark lazy)
//     are copied back before the call returns, exactly as without the mode;
//   * rte_hip_writeback(ptr) copies a mirrored array back to the host on request; mirrors that are neither used nor
//     written back within mirror_max_age library calls are dropped (their host arrays are usually gone by then).
// Contract of the mode: host code does not READ a lazily held array before writing it back, and does not write PART of
// one.  The reference's clear-sky / all-sky LW frontend satisfies it with the value checks off (rte_config_checks); its
// SW gas optics combines tau and tau_rayleigh on the host (mo_gas_optics_rrtmgp.F90:1954-2036), so compute_tau_rayleigh
// writes both back (writeback_produced_by).  Mirrors belong to a context: arrays produced on one context are consumed on it.
constexpr int kCanaries = 34;
constexpr size_t kLazyMinBytes = 4096;
constexpr size_t kInputCacheMin = size_t(64) << 10;  // host-produced inputs from this size on keep a device copy + shadow
static int g_procmem_fd = -2;
// device bytes held by the mirrors (and their free lists) of ALL contexts, and the limit they share: with one context per
// host thread, per-context limits of "60 % of what is free" would add up to several times the device
static std::atomic<size_t> g_mirror_all{0};
static std::atomic<size_t> g_mirror_limit_all{0};
static std::atomic<int> g_staging_contexts{0};  // contexts that have staged a host-produced input in host-mirror mode
static std::mutex g_report_mutex;
static std::vector<Context*> g_report_contexts;

static inline double secs_since(std::chrono::steady_clock::time_point t0) {
  return std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
}
static void staging_report() {
  std::lock_guard<std::mutex> l(g_report_mutex);
  int i = 0;
  for (Context* c : g_report_contexts) {
    fprintf(stderr, "rte_rrtmgp_hip staging report (context %d): %ld calls, %.3f s inside the library (host-to-device copies %.3f s for %.3f GB, "
            "waits + device-to-host %.3f s for %.3f GB, mirror look-ups %.3f s); mirrors made %lld, hits %lld, dropped %lld + %lld, aged %lld, "
            "zero fills elided %lld, device bytes held %.2f GB; host tables uploaded %lld, reused %lld; unchanged inputs served from the device %lld "
            "(%.3f GB not copied; %lld input copies made, %lld evicted)\n", i++, c->n_calls, c->t_call, c->t_h2d, c->mstat[2] * 1e-9, c->t_wait,
            c->mstat[3] * 1e-9, c->t_find, c->mstat[1], c->mstat[0], c->mstat[4], c->mstat[5], c->mstat[6], c->mstat[7], c->mirror_total * 1e-9,
            c->table_uploads, c->table_hits, c->input_hits, c->input_saved * 1e-9, c->input_made, c->input_evicted);
    std::vector<size_t> order(c->t_entry.size());
    for (size_t k = 0; k < order.size(); ++k) order[k] = k;
    std::sort(order.begin(), order.end(), [&](size_t a, size_t b) { return c->t_entry[a].second.first > c->t_entry[b].second.first; });
    for (size_t k = 0; k < order.size() && k < 16; ++k) {
      const auto& e = c->t_entry[order[k]];
      fprintf(stderr, "    %-44s %6ld calls %9.4f s  to device %9.3f MB  to host %9.3f MB\n", e.first, e.second.second, e.second.first,
              c->b_entry[order[k]].first * 1e-6, c->b_entry[order[k]].second * 1e-6);
    }
  }
}
static bool mirror_on() {
  Context& c = C;
  if (c.mirror_mode < 0) {
    const char* e = getenv("RTE_HIP_HOST_MIRROR");
    c.mirror_mode = (e && atoi(e) > 0) ? 1 : 0;
    if (const char* a = getenv("RTE_HIP_MIRROR_MAX_AGE")) c.mirror_max_age = atol(a) > 0 ? atol(a) : c.mirror_max_age;
    if (const char* r = getenv("RTE_HIP_STAGING_REPORT"))
      if (atoi(r) > 0) {
        std::lock_guard<std::mutex> l(g_report_mutex);
        if (g_report_contexts.empty()) atexit(staging_report);
        g_report_contexts.push_back(&c);
        c.report_on = true;
      }
  }
  return c.mirror_mode == 1;
}
static size_t canary_offset(size_t bytes, int k) {
  if (k == kCanaries - 1) return bytes - 16;
  return ((bytes - 16) / (kCanaries - 1) * (size_t)k) & ~size_t(7);
}
static void canary_value(unsigned long long magic, int k, unsigned long long v[2]) {
  v[0] = magic ^ (0xD1B54A32D192ED03ull * (unsigned long long)(k + 1));
  v[1] = ~v[0];
}
static void write_canaries(void* host, size_t bytes, unsigned long long magic) {
  for (int k = 0; k < kCanaries; ++k) {
    unsigned long long v[2];
    canary_value(magic, k, v);
    memcpy((char*)host + canary_offset(bytes, k), v, 16);
  }
}
// are the canaries of `m` still in host memory?  Reads go through /proc/self/mem: a range that has been freed and unmapped
// gives an error instead of a fault.
static bool canaries_intact(const Mirror& m) {
  // one process_vm_readv of our own address space for all of them (33 pread calls were 0.25 ms of every library call of the
  // unchanged frontend); a range that is gone makes the call fail or come back short.  Where the call is not permitted: pread.
  static std::atomic<int> vm_ok{1};
  if (vm_ok.load(std::memory_order_relaxed)) {
    unsigned long long w[kCanaries][2];
    struct iovec loc[kCanaries], rem[kCanaries];
    for (int k = 0; k < kCanaries; ++k) {
      loc[k].iov_base = w[k]; loc[k].iov_len = 16;
      rem[k].iov_base = (void*)(m.host + canary_offset(m.bytes, k)); rem[k].iov_len = 16;
    }
    const ssize_t n = process_vm_readv(getpid(), loc, kCanaries, rem, kCanaries, 0);
    if (n == (ssize_t)(16 * kCanaries)) {
      for (int k = 0; k < kCanaries; ++k) {
        unsigned long long v[2];
        canary_value(m.magic, k, v);
        if (w[k][0] != v[0] || w[k][1] != v[1]) return false;
      }
      return true;
    }
    if (n >= 0 || errno == EFAULT) return false;  // (part of the range is not mapped any more)
    vm_ok.store(0, std::memory_order_relaxed);    // EPERM / ENOSYS: the slower way from now on
  }
  if (g_procmem_fd == -2) g_procmem_fd = open("/proc/self/mem", O_RDONLY | O_CLOEXEC);
  if (g_procmem_fd < 0) return false;  // cannot verify: never trust
  for (int k = 0; k < kCanaries; ++k) {
    unsigned long long v[2], w[2];
    canary_value(m.magic, k, v);
    if (pread(g_procmem_fd, w, 16, (off_t)(uintptr_t)(m.host + canary_offset(m.bytes, k))) != 16) return false;
    if (w[0] != v[0] || w[1] != v[1]) return false;
  }
  return true;
}
// Host-to-device copy of a PAGEABLE array on the context's stream.  For a copy of 1 MB or more the HIP runtime pins the
// caller's pages: fast when the same pages come again (55 GB/s: its pin cache), but the Fortran frontend's arrays are
// automatic / allocatable arrays, freshly mapped for every call -- measured 0.7 ... 40 GB/s and erratic
// (tools/h2d_bench.hip).  Large copies therefore go through the context's own pinned ring: memcpy of a 4 MB chunk while
// the DMA engine takes the chunk before it, 27-38 GB/s whatever the age of the pages.  Like the runtime's pageable copy
// the call returns when the source has been read.  Arrays above 128 MB stay with the runtime: in the frontend those are
// the 3-D members of optical-property objects (SW: tau, ssa, g after the host-side combine_abs_and_rayleigh), allocated
// once per object, and their pages are found pinned again.
static bool h2d(Context& c, void* d, const void* p, size_t bytes) {  // true: the source has been read when it returns
  if (c.ring_mode < 0) {
    const char* e = getenv("RTE_HIP_H2D_RING");
    c.ring_mode = (e && atoi(e) == 0) ? 0 : 1;
  }
  if (bytes == 0) return true;
  if (!c.ring_mode || bytes > (size_t(128) << 20)) {
    HIP_CHECK(hipMemcpyAsync(d, p, bytes, hipMemcpyHostToDevice, c.stream));
    return false;
  }
  // Small arrays are packed into the current slot one after the other (the runtime's pageable copy of some sizes -- 64 KB,
  // 2 MB -- returns only when everything queued on the stream before it has run, i.e. it would make the host wait for the
  // previous calls' kernels: tools/h2d_bench.hip); a slot's event is recorded when the ring moves on, and waited for when
  // the ring comes round to the slot again.
  size_t o = 0;
  while (o < bytes) {
    if (c.ring_cur < 0 || c.ring_fill >= Context::kRingChunk) {
      if (c.ring_cur >= 0) {
        HIP_CHECK(hipEventRecord(c.ring_ev[c.ring_cur], c.stream));
        c.ring_busy[c.ring_cur] = true;
      }
      c.ring_cur = (c.ring_cur + 1) % Context::kRing;
      c.ring_fill = 0;
      if (!c.ring[c.ring_cur]) {
        HIP_CHECK(hipHostMalloc((void**)&c.ring[c.ring_cur], Context::kRingChunk, hipHostMallocDefault));
        HIP_CHECK(hipEventCreateWithFlags(&c.ring_ev[c.ring_cur], hipEventDisableTiming));
      }
      if (c.ring_busy[c.ring_cur]) {
        HIP_CHECK(hipEventSynchronize(c.ring_ev[c.ring_cur]));
        c.ring_busy[c.ring_cur] = false;
      }
    }
    const size_t room = Context::kRingChunk - c.ring_fill;
    const size_t len = bytes - o < room ? bytes - o : room;
    char* slot = c.ring[c.ring_cur] + c.ring_fill;
    memcpy(slot, (const char*)p + o, len);
    HIP_CHECK(hipMemcpyAsync((char*)d + o, slot, len, hipMemcpyHostToDevice, c.stream));
    c.ring_fill += (len + 255) & ~size_t(255);
    o += len;
  }
  return true;
}
// marks "the host-to-device copies queued so far": in host-mirror mode a call with staged inputs only waits for THIS, not
// for its kernels (every in()/out() conversion precedes the call's first launch)
static void mark_h2d() {
  Context& c = C;
  if (!c.ev_h2d) HIP_CHECK(hipEventCreateWithFlags(&c.ev_h2d, hipEventDisableTiming));
  HIP_CHECK(hipEventRecord(c.ev_h2d, c.stream));
}
static void mirror_release_buffer(char* dev, size_t cap) { C.mirror_free.push_back(FreeBuf{dev, cap}); }
static void mirror_trim_free_list() {
  Context& c = C;
  if (c.mirror_free.empty()) return;
  HIP_CHECK(hipStreamSynchronize(c.stream));  // kernels of earlier calls may still use them
  for (auto& f : c.mirror_free) { HIP_CHECK(hipFree(f.dev)); c.mirror_total -= f.cap; g_mirror_all -= f.cap; }
  c.mirror_free.clear();
}
static char* mirror_alloc(size_t bytes, size_t* cap_out) {
  Context& c = C;
  size_t best = (size_t)-1;
  for (size_t i = 0; i < c.mirror_free.size(); ++i) {
    const size_t cap = c.mirror_free[i].cap;
    if (cap >= bytes && cap <= bytes + bytes / 4 + (size_t(1) << 20) && (best == (size_t)-1 || cap < c.mirror_free[best].cap)) best = i;
  }
  if (best != (size_t)-1) {
    FreeBuf f = c.mirror_free[best];
    c.mirror_free.erase(c.mirror_free.begin() + best);
    *cap_out = f.cap;
    return f.dev;
  }
  if (g_mirror_limit_all.load() == 0) {
    size_t lim = 0;
    if (const char* e = getenv("RTE_HIP_MIRROR_MAX_GB")) lim = (size_t)(atof(e) * 1073741824.0);
    if (lim == 0) {
      size_t fr = 0, tot = 0;
      HIP_CHECK(hipMemGetInfo(&fr, &tot));
      lim = fr / 10 * 6;
    }
    size_t expect = 0;
    g_mirror_limit_all.compare_exchange_strong(expect, lim);
  }
  c.mirror_limit = g_mirror_limit_all.load();
  const size_t cap = (bytes + (size_t(2) << 20) - 1) & ~((size_t(2) << 20) - 1);
  if (g_mirror_all.load() + cap > c.mirror_limit) mirror_trim_free_list();  // (this context's idle buffers; others trim theirs in turn)
  char* d = nullptr;
  HIP_CHECK(hipMalloc((void**)&d, cap));
  c.mirror_total += cap;
  g_mirror_all += cap;
  *cap_out = cap;
  return d;
}
static void mirror_drop(size_t i) {
  Context& c = C;
  mirror_release_buffer(c.mirrors[i].dev, c.mirrors[i].cap);
  c.mirrors.erase(c.mirrors.begin() + i);
}
static void mirror_age_out() {
  Context& c = C;
  for (size_t i = c.mirrors.size(); i-- > 0;)
    if (c.seq - c.mirrors[i].last_use > c.mirror_max_age) {
      // An aged-out array whose canaries are still in place was never written back: its host copy holds the canaries and
      // whatever was there before.  Usually the array is simply gone (freed); if the program hands it to the library again
      // it is staged as it is -- say so once (rte_hip_writeback(ptr) or a larger RTE_HIP_MIRROR_MAX_AGE are the remedies).
      static std::atomic<bool> warned{false};
      if (canaries_intact(c.mirrors[i]) && !warned.exchange(true))
        fprintf(stderr, "rte_rrtmgp_hip: host-mirror mode dropped the device copy of a %zu-byte host array at %p unused for %ld calls "
                        "(RTE_HIP_MIRROR_MAX_AGE); its host memory was not written back (rte_hip_writeback); further cases are "
                        "counted only (rte_hip_mirror_stat(6))\n", c.mirrors[i].bytes, (void*)c.mirrors[i].host, c.mirror_max_age);
      mirror_drop(i);
      ++c.mstat[6];
    }
}
static void mirror_drop_all() {
  Context& c = C;
  while (!c.mirrors.empty()) mirror_drop(c.mirrors.size() - 1);
  mirror_trim_free_list();
}
// the mirror that CONTAINS [p, p+bytes) with its canaries intact (index), or -1; mirrors that merely overlap the range, or
// whose host memory was changed, are dropped on the way (the host reused the memory)
static long mirror_find(const char* p, size_t bytes, bool declared = false /* the host program names the array itself (rte_hip_writeback) */) {
  Context& c = C;
  long hit = -1;
  for (size_t i = c.mirrors.size(); i-- > 0;) {
    Mirror& m = c.mirrors[i];
    if (p + bytes <= m.host || m.host + m.bytes <= p) continue;
    const bool contained = m.host <= p && p + bytes <= m.host + m.bytes;
    // A PART of a mirrored array is only served from the device copy if the part itself holds one of the array's canaries:
    // a smaller host array allocated later inside a freed mirrored one, between two canaries, would otherwise be a hit with
    // every canary intact and the kernel would get the stale device copy instead of the host's data.  (Whole arrays --
    // what the frontend hands from one kernel to the next -- always hold all of them.)
    bool guarded = contained && ((bytes == m.bytes && p == m.host) || declared);
    if (contained && !guarded) {
      const size_t lo = (size_t)(p - m.host), hi = lo + bytes;
      for (int k = 0; k < kCanaries && !guarded; ++k) {
        const size_t o = canary_offset(m.bytes, k);
        guarded = o >= lo && o + 16 <= hi;
      }
    }
    if (contained && guarded && hit < 0 && canaries_intact(m)) { hit = (long)i; continue; }
    // A contained part WITHOUT a canary inside it (a slice shorter than about 1/33 of the array: one g-point plane, a column
    // block) of a mirror whose canaries are all intact is most likely a slice of the LIVE array, whose only valid copy is on
    // the device: it is lost with the mirror, and the kernel is staged from host memory that holds canaries and stale data.
    // The guard above cannot tell that from reused memory, so the loss is made visible (once; counted in rte_hip_mirror_stat(4)):
    // a host program that hands slices of a library output back to the library calls rte_hip_writeback(array) first.
    if (contained && !guarded && canaries_intact(m)) {
      static std::atomic<bool> warned{false};
      if (!warned.exchange(true))
        fprintf(stderr, "rte_rrtmgp_hip: host-mirror mode was handed %zu bytes at %p, a part of the %zu-byte host array at %p whose only "
                        "valid copy is on the device; the part holds none of the array's canaries, so the device copy is dropped and "
                        "the call reads the host memory as it is (stale).  Call rte_hip_writeback(array) before passing slices shorter "
                        "than 1/33 of a library output back to the library (INTEGRATION.md); further cases are counted only "
                        "(rte_hip_mirror_stat(4))\n", bytes, (const void*)p, m.bytes, (void*)m.host);
    }
    ++c.mstat[contained ? 4 : 5];
    mirror_drop(i);
    if (hit > (long)i) --hit;
  }
  return hit;
}

// ---- Call ---------------------------------------------------------------------------------------
long call_seq() { return C.seq; }  // number of the current (innermost) API call

Call::Call(const char* n) : name(n) {
  Context& c = C;
  c.mutex.lock();
  locked_ = &c;
  try {
    if (c.last_error != 0)  // sticky error pending: every call on this context is a no-op until it is cleared
      throw Error{0, "skipped: the context holds an earlier error (rte_hip_last_error / rte_hip_clear_error)"};
    if (c.device >= 0) {
      int cur = -1;
      if (hipGetDevice(&cur) != hipSuccess || cur != c.device) HIP_CHECK(hipSetDevice(c.device));
    }
    ++c.seq;
    ++c.n_calls;
    c.call_t0 = std::chrono::steady_clock::now();
    c.call_h2d0 = c.mstat[2]; c.call_d2h0 = c.mstat[3];
    fork_candidate_ = c.fork_valid;  // the previous call left a fork point (it is consumed or dropped by this call)
    c.fork_valid = false;
    if (!c.pending.empty()) {
      // recorded fills are materialised on the library stream, i.e. BEHIND the previous call's kernels; a call forked to
      // the side stream waits only for what was queued before that previous call, so a fill of one of its outputs could
      // land after its own stores: a call that had to materialise fills is never forked
      fork_candidate_ = false;
      flush_pending_zeros();
    }
    scratch_reset();
    if (mirror_on() && !c.mirrors.empty()) mirror_age_out();
  } catch (...) {  // the destructor of a half-constructed object does not run: release the context here
    locked_ = nullptr;
    c.mutex.unlock();
    throw;
  }
}

// Move the rest of this call (launches, scratch, timing events) to the side stream if that is safe: nothing of this
// call has been staged or queued yet, all its arrays are device memory, and its outputs do not touch the range the
// previous call writes.  Must be called after the in()/out() conversions and before the first launch.
bool Call::try_fork(const void* const* outs, const size_t* bytes, int n) {
  Context& c = C;
  if (!c.overlap || !fork_candidate_ || n_back_ > 0 || staged_in_ || host_visible_ || n_host_tmp_ > 0) return false;
  for (int i = 0; i < n; ++i) {
    const char* lo = (const char*)outs[i];
    if (!lo || !is_device_memory(lo)) return false;
    if (lo < c.fork_hi && lo + bytes[i] > c.fork_lo) return false;
  }
  HIP_CHECK(hipStreamWaitEvent(c.side, c.ev_fork, 0));
  c.on_side = true;
  forked_ = true;
  scratch_reset();  // the side arena: its previous user was the previous forked call, which the library stream has joined
  return true;
}

void* Call::stage(void* p, size_t bytes, bool copy_in, bool copy_out, bool lazy, bool* zero_fill) {
  if (zero_fill) *zero_fill = false;
  if (!p || bytes == 0) return p;
  void* dv;
  const int kind = classify(p, &dv);
  if (kind == 1) {
    if (!C.pending_src.empty()) sources_touch(p, copy_in);
    return p;
  }
  if (kind == 2) { host_visible_ = true; return dv; }  // in place, but synchronous for the caller (see ~Call)
  Context& c = C;
  if (mirror_on()) {
    const auto tf = std::chrono::steady_clock::now();
    const long hit = mirror_find((const char*)p, bytes);
    c.t_find += secs_since(tf);
    if (hit >= 0) {
      Mirror& m = c.mirrors[(size_t)hit];
      m.last_use = c.seq;
      ++c.mstat[0];
      char* d = m.dev + ((const char*)p - m.host);
      if (m.zero_pending) {
        if (zero_fill && copy_out && lazy && bytes == m.bytes) { *zero_fill = true; ++c.mstat[7]; }  // the caller overwrites all of it
        else HIP_CHECK(hipMemsetAsync(m.dev, 0, m.bytes, c.stream));
        m.zero_pending = false;
      }
      if (!copy_out) return d;                        // input: served from the device copy
      if (lazy) { m.producer = name; return d; }      // written again on the device; the host copy stays unspecified
      // an output the caller reads on the host lies inside a mirrored range: the whole array goes back and the mirror ends
      if (n_back_ >= 16 || n_recycle_ >= 16) { fprintf(stderr, "rte_rrtmgp_hip: too many staged outputs\n"); abort(); }
      back_[n_back_++] = Back{m.host, m.dev, m.bytes};
      recycle_[n_recycle_++] = Recycle{m.dev, m.cap};
      c.mirrors.erase(c.mirrors.begin() + hit);
      return d;
    }
    if (copy_out && lazy && bytes >= kLazyMinBytes && n_lazy_ < 16) {
      size_t cap = 0;
      char* d = mirror_alloc(bytes, &cap);
      if (copy_in) {
        const auto t0 = std::chrono::steady_clock::now();
        if (!h2d(c, d, p, bytes)) staged_plain_ = true;
        c.t_h2d += secs_since(t0);
        staged_in_ = true;
        c.mstat[2] += (long long)bytes;
        mark_h2d();
      }
      c.magic_state = c.magic_state * 6364136223846793005ull + 1442695040888963407ull;
      Mirror m{(char*)p, bytes, d, cap, c.seq, c.magic_state ^ (unsigned long long)(uintptr_t)p, name, false};
      c.mirrors.push_back(m);
      lazy_[n_lazy_++] = Lazy{p, bytes, m.magic};
      ++c.mstat[1];
      return d;
    }
  }
  if (c.mirror_mode == 1 && copy_in && !copy_out && bytes >= kInputCacheMin && bytes <= (size_t(256) << 20)) {
    // An input the host produced.  The library keeps the device copy AND a host-side shadow of what it uploaded; when the
    // same range comes again with the same bytes (compared in full: memcmp costs the thread what the copy into the pinned
    // staging ring would, and saves the transfer) the device copy is served.  The unchanged frontend passes play, tlay and
    // col_gas to interpolation, compute_tau_absorption and compute_Planck_source in a row, and an emissivity, a zero incident
    // flux and a secant field that do not change from block to block to every solver call: 19.6 KB per column crossed PCIe
    // where 6.3 KB were new.
    // Default: on as soon as a second context stages host arrays (several host threads share the PCIe link, and the link is what
    // binds them: 1.6-1.7 -> 2.2-2.75 M columns/s on eight threads); a single thread is bound by its own passes over the
    // arrays, and compare + shadow cost it 10 % (0.50 -> 0.45 M).  RTE_HIP_INPUT_CACHE=1 / 0 forces it on / off.
    if (c.input_cache < 0) {
      const char* e = getenv("RTE_HIP_INPUT_CACHE");
      c.input_cache = (!e || !*e) ? 2 : (atoi(e) == 0 ? 0 : 1);
      ++g_staging_contexts;
    }
    if (c.input_cache == 1 || (c.input_cache == 2 && g_staging_contexts.load(std::memory_order_relaxed) >= 2)) {
      const auto t0 = std::chrono::steady_clock::now();
      Context::InputCopy* hit = nullptr;  // the entry of this very range, if there is one (it is refilled on a mismatch)
      Context::InputCopy* same = nullptr;  // an entry that holds these bytes
      for (auto& e : c.inputs)
        if (e.host == (const char*)p && e.bytes == bytes) { hit = &e; break; }
      if (hit && memcmp(hit->shadow, p, bytes) == 0) same = hit;
      if (!same) {
        // the same bytes under another address: the frontend's emissivity / incident-flux / secant fields are temporaries that
        // land where the allocator puts them, with the same contents every time (a memcmp of different fields ends at the
        // first word)
        for (auto& e : c.inputs)
          if (&e != hit && e.bytes == bytes && memcmp(e.shadow, p, bytes) == 0) { same = &e; break; }
      }
      if (same) {
        same->last_use = c.seq;
        staged_in_ = true;  // (a call with host inputs stays on the library stream: the next upload into this copy is ordered behind it)
        ++c.input_hits; c.input_saved += (long long)bytes;
        c.t_h2d += secs_since(t0);
        return same->dev;
      }
      // An entry that an EARLIER argument of this very call was served from (last_use == c.seq) is neither refilled nor
      // evicted: its device copy is about to be read by this call's kernel (Fortran temporaries swap addresses between
      // calls, so argument X may have matched entry E by content while argument Y now arrives at E's address with other
      // bytes).  Such an argument is staged through scratch like any uncached input.
      bool cacheable = !(hit && hit->last_use == c.seq);
      if (cacheable && !hit) {
        // (bounded: 48 arrays, 1 GB of shadows per context; the least recently used goes first)
        while (!c.inputs.empty() && (c.inputs.size() >= 48 || c.inputs_total + bytes > (size_t(1) << 30))) {
          size_t v = c.inputs.size();
          for (size_t k = 0; k < c.inputs.size(); ++k)
            if (c.inputs[k].last_use != c.seq && (v == c.inputs.size() || c.inputs[k].last_use < c.inputs[v].last_use)) v = k;
          if (v == c.inputs.size()) { cacheable = false; break; }  // everything held belongs to this call
          HIP_CHECK(hipStreamSynchronize(c.stream));  // kernels of earlier calls may still read the device copy
          HIP_CHECK(hipFree(c.inputs[v].dev)); free(c.inputs[v].shadow);
          c.inputs_total -= c.inputs[v].bytes;
          c.inputs.erase(c.inputs.begin() + v);
          ++c.input_evicted;
        }
      }
      if (cacheable && !hit) {
        Context::InputCopy e{(const char*)p, bytes, (char*)malloc(bytes), nullptr, c.seq};
        if (!e.shadow) throw Error{-1, "out of host memory for an input shadow"};
        const hipError_t rc = hipMalloc((void**)&e.dev, bytes);
        if (rc != hipSuccess) { free(e.shadow); HIP_CHECK(rc); }
        c.inputs.push_back(e);
        c.inputs_total += bytes;
        ++c.input_made;
        hit = &c.inputs.back();
      }
      if (cacheable) {
        hit->last_use = c.seq;
        // (through the pinned ring, stream-ordered behind the kernels that read the device copy's previous contents.  A pinned
        //  shadow that the transfer would start from -- one memcpy instead of two -- was measured: slower, the unpipelined copy
        //  and the pinning of every new shadow cost more than the second memcpy)
        if (!h2d(c, hit->dev, p, bytes)) staged_plain_ = true;
        memcpy(hit->shadow, p, bytes);
        c.t_h2d += secs_since(t0);
        staged_in_ = true;
        c.mstat[2] += (long long)bytes;
        mark_h2d();
        return hit->dev;
      }
      c.t_h2d += secs_since(t0);
    }
  }
  void* d = scratch(bytes);
  if (copy_in) {
    const auto t0 = std::chrono::steady_clock::now();
    if (!h2d(c, d, p, bytes)) staged_plain_ = true;
    c.t_h2d += secs_since(t0);
    staged_in_ = true;
    c.mstat[2] += (long long)bytes;
    if (c.mirror_mode == 1) mark_h2d();
  }
  if (copy_out) {
    if (n_back_ >= 16) { fprintf(stderr, "rte_rrtmgp_hip: too many staged outputs\n"); abort(); }
    back_[n_back_++] = Back{p, d, bytes};
  }
  return d;
}

bool Call::lazy_zero(void* p, size_t bytes) {
  void* dv;
  if (!mirror_on() || !p || bytes < kLazyMinBytes || classify(p, &dv) != 0) return false;
  Context& c = C;
  const long hit = mirror_find((const char*)p, bytes);
  if (hit >= 0) {
    Mirror& m = c.mirrors[(size_t)hit];
    m.last_use = c.seq;
    m.producer = name;
    ++c.mstat[0];
    if (bytes == m.bytes) m.zero_pending = true;  // whole array: recorded, filled only if somebody reads it
    else HIP_CHECK(hipMemsetAsync(m.dev + ((const char*)p - m.host), 0, bytes, c.stream));
    return true;
  }
  if (n_lazy_ >= 16) return false;
  size_t cap = 0;
  char* d = mirror_alloc(bytes, &cap);
  c.magic_state = c.magic_state * 6364136223846793005ull + 1442695040888963407ull;
  Mirror m{(char*)p, bytes, d, cap, c.seq, c.magic_state ^ (unsigned long long)(uintptr_t)p, name, true};
  c.mirrors.push_back(m);
  lazy_[n_lazy_++] = Lazy{p, bytes, m.magic};
  ++c.mstat[1];
  return true;
}

// sampled fingerprint of a host table: 4 KB at either end and 256 eight-byte words in between
static unsigned long long table_fingerprint(const char* p, size_t bytes) {
  unsigned long long h = 1469598103934665603ull ^ bytes;
  auto mix = [&](const char* q, size_t n) {
    for (size_t i = 0; i + 8 <= n; i += 8) { unsigned long long w; memcpy(&w, q + i, 8); h = (h ^ w) * 1099511628211ull; }
  };
  const size_t edge = bytes < 4096 ? bytes : 4096;
  mix(p, edge);
  if (bytes > edge) mix(p + bytes - edge, edge);
  if (bytes > 2 * edge) {
    const size_t step = ((bytes - 2 * edge) / 256) & ~size_t(7);
    if (step) for (size_t i = 0; i < 256; ++i) mix(p + edge + i * step, 8);
  }
  return h;
}
void drop_table_copies() {
  Context& c = C;
  if (c.tables.empty() && c.inputs.empty()) return;
  HIP_CHECK(hipStreamSynchronize(c.stream));
  ++c.addr_epoch;
  for (auto& t : c.tables) HIP_CHECK(hipFree(t.dev));
  c.tables.clear();
  for (auto& e : c.inputs) { HIP_CHECK(hipFree(e.dev)); free(e.shadow); }
  c.inputs.clear();
  c.inputs_total = 0;
}
const void* Call::stage_table(const void* p, size_t bytes) {
  void* dv;
  if (!p || bytes < (size_t(64) << 10) || !mirror_on() || classify(p, &dv) != 0) return stage(const_cast<void*>(p), bytes, true, false);
  Context& c = C;
  const unsigned long long fp = table_fingerprint((const char*)p, bytes);
  for (auto& t : c.tables)
    if (t.host == (const char*)p && t.bytes == bytes && t.fp == fp) { t.last_use = c.seq; ++c.table_hits; return t.dev; }
  for (size_t i = c.tables.size(); i-- > 0;)  // same range, other contents (or more than 24 copies): replace
    if (c.tables[i].host == (const char*)p || c.tables.size() >= 24) {
      size_t victim = i;
      if (c.tables[i].host != (const char*)p) {  // least recently used
        victim = 0;
        for (size_t k = 1; k < c.tables.size(); ++k) if (c.tables[k].last_use < c.tables[victim].last_use) victim = k;
      }
      HIP_CHECK(hipStreamSynchronize(c.stream));
      HIP_CHECK(hipFree(c.tables[victim].dev));
      c.tables.erase(c.tables.begin() + victim);
      break;
    }
  char* d = nullptr;
  HIP_CHECK(hipMalloc((void**)&d, bytes));
  const auto t0 = std::chrono::steady_clock::now();
  if (!h2d(c, d, p, bytes)) staged_plain_ = true;
  c.t_h2d += secs_since(t0);
  staged_in_ = true;
  c.mstat[2] += (long long)bytes;
  mark_h2d();
  c.tables.push_back(TableCopy{(const char*)p, bytes, fp, d, c.seq});
  ++c.table_uploads;
  return d;
}

void Call::writeback_produced_by(const char* producer) {
  if (!mirror_on()) return;
  Context& c = C;
  for (size_t i = c.mirrors.size(); i-- > 0;) {
    Mirror& m = c.mirrors[i];
    if (strcmp(m.producer, producer) != 0 || c.seq - m.last_use > 8) continue;
    if (n_back_ >= 16 || n_recycle_ >= 16) break;
    if (m.zero_pending) { HIP_CHECK(hipMemsetAsync(m.dev, 0, m.bytes, c.stream)); m.zero_pending = false; }
    back_[n_back_++] = Back{m.host, m.dev, m.bytes};
    recycle_[n_recycle_++] = Recycle{m.dev, m.cap};
    c.mirrors.erase(c.mirrors.begin() + i);
  }
}

const void* Call::to_host(const void* p, size_t bytes) {
  if (!p || bytes == 0 || !is_device_pointer(p)) return p;
  if (n_host_tmp_ >= (int)(sizeof(host_tmp_) / sizeof(host_tmp_[0]))) {
    fprintf(stderr, "rte_rrtmgp_hip: %s: too many host copies of device tables\n", name);
    abort();
  }
  void* h = malloc(bytes);
  host_tmp_[n_host_tmp_++] = h;
  HIP_CHECK(hipMemcpyAsync(h, p, bytes, hipMemcpyDeviceToHost, stream()));
  HIP_CHECK(hipStreamSynchronize(stream()));
  return h;
}

Call::~Call() noexcept(false) {
  if (!locked_) return;
  Context& c = *(Context*)locked_;
  locked_ = nullptr;
  const bool unwinding = std::uncaught_exceptions() > 0;  // a HIP call of this entry failed: clean up, queue nothing more
  try {
    if (!unwinding) {
      const bool mirror = c.mirror_mode == 1;
      const auto tw = std::chrono::steady_clock::now();
      for (int i = 0; i < n_back_; ++i) {
        HIP_CHECK(hipMemcpyAsync(back_[i].host, back_[i].dev, back_[i].bytes, hipMemcpyDeviceToHost, c.stream));
        c.mstat[3] += (long long)back_[i].bytes;
      }
      // host arrays (staged, or host-visible memory used in place): the caller owns them again when the call returns.
      // In host-mirror mode a call that staged inputs only waits for those copies (mark_h2d), not for its kernels: they
      // run while the host program prepares the next call.
      if (n_back_ > 0 || host_visible_ || (staged_in_ && !mirror)) HIP_CHECK(hipStreamSynchronize(c.stream));
      else if (staged_plain_) HIP_CHECK(hipEventSynchronize(c.ev_h2d));
      c.t_wait += secs_since(tw);
      for (int i = 0; i < n_lazy_; ++i) write_canaries(lazy_[i].host, lazy_[i].bytes, lazy_[i].magic);
      if (forked_) {  // join: the library stream (and whatever is queued on it from now on) waits for this call
        HIP_CHECK(hipEventRecord(c.ev_join, c.side));
        HIP_CHECK(hipStreamWaitEvent(c.stream, c.ev_join, 0));
      }
      hipError_t e = hipGetLastError();
      if (e != hipSuccess) throw Error{(int)e, std::string("launch error: ") + hipGetErrorString(e)};
    }
  } catch (...) {
    for (int i = 0; i < n_recycle_; ++i) c.mirror_free.push_back(FreeBuf{(char*)recycle_[i].dev, recycle_[i].cap});
    for (int i = 0; i < n_host_tmp_; ++i) free(host_tmp_[i]);
    c.on_side = false;
    c.mutex.unlock();
    throw;
  }
  for (int i = 0; i < n_recycle_; ++i) c.mirror_free.push_back(FreeBuf{(char*)recycle_[i].dev, recycle_[i].cap});
  for (int i = 0; i < n_host_tmp_; ++i) free(host_tmp_[i]);
  c.on_side = false;
  const double dt_call = secs_since(c.call_t0);
  c.t_call += dt_call;
  if (c.report_on) {
    size_t i = 0;
    while (i < c.t_entry.size() && c.t_entry[i].first != name) ++i;  // (entry names are string literals: one address each)
    if (i == c.t_entry.size()) { c.t_entry.push_back({name, {0.0, 0L}}); c.b_entry.push_back({0, 0}); }
    c.t_entry[i].second.first += dt_call;
    c.t_entry[i].second.second += 1;
    c.b_entry[i].first += c.mstat[2] - c.call_h2d0;
    c.b_entry[i].second += c.mstat[3] - c.call_d2h0;
  }
  c.mutex.unlock();
}

// ---- kernel timing ------------------------------------------------------------------------------
void prof_begin(const char* kernel) {
  Context& c = C;
  c.cur = nullptr;
  if (!c.prof_on) return;
  if (!c.prof_only.empty() && c.prof_only != kernel) return;
  for (auto& e : c.prof)
    if (e.name == kernel) c.cur = &e;
  if (!c.cur) {
    c.prof.push_back(ProfEntry{kernel});
    c.cur = &c.prof.back();
  }
  HIP_CHECK(hipEventCreate(&c.cur_start));
  HIP_CHECK(hipEventRecord(c.cur_start, stream()));
}
void prof_end() {
  Context& c = C;
  if (!c.prof_on || !c.cur) return;
  hipEvent_t stop;
  HIP_CHECK(hipEventCreate(&stop));
  HIP_CHECK(hipEventRecord(stop, stream()));
  c.cur->ev.emplace_back(c.cur_start, stop);
  c.cur = nullptr;
}
static void prof_resolve() {
  for (auto& e : C.prof) {
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

// release every device buffer the current context holds (arena, persistent slots, mirrors, gas-optics state)
static void release_context_buffers() {
  Context& c = C;
  flush_pending_zeros();
  if (c.src_ops.expand) flush_pending_sources();  // (a record names factors in a slot that is about to go)
  c.pending_src.clear();
  ++c.addr_epoch;  // graphs captured on this context address buffers that are freed here
  HIP_CHECK(hipStreamSynchronize(c.stream));
  if (c.side) HIP_CHECK(hipStreamSynchronize(c.side));
  if (c.aux) HIP_CHECK(hipStreamSynchronize(c.aux));
  mirror_drop_all();
  drop_table_copies();
  if (c.gas) { c.gas_free(c.gas); c.gas = nullptr; }
  for (auto* v : {&c.blocks_main, &c.blocks_side}) {
    for (auto& b : *v) HIP_CHECK(hipFree(b.base));
    v->clear();
  }
  for (auto& s : c.slots) {
    if (s.p) HIP_CHECK(hipFree(s.p));
    s = Slot{};
  }
}

}  // namespace rte

// ---- library-extension entry points (not part of the reference interface; include/rte_hip_ext.h) ------------------
#define LOCK_CTX std::lock_guard<std::recursive_mutex> l_(rte::ctx().mutex)
extern "C" {

// ---- contexts
// A new context on `device` (-1: the device current at its first use) launching on `stream` (a hipStream_t; NULL: the
// context creates a non-blocking stream of its own).  It inherits the option settings of the calling thread's current
// context.  Make it current on the thread that uses it with rte_hip_ctx_set_current.
void* rte_hip_ctx_create(int device, void* stream) {
  RTE_TRY
  rte::Context& cur = rte::ctx();
  auto* c = new rte::Context();
  c->device = device;
  int prev = -1;
  if (device >= 0) { HIP_CHECK(hipGetDevice(&prev)); HIP_CHECK(hipSetDevice(device)); }
  if (stream) c->stream = (hipStream_t)stream;
  else { HIP_CHECK(hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking)); c->own_stream = true; }
  if (device >= 0 && prev != device) HIP_CHECK(hipSetDevice(prev));
  c->overlap = cur.overlap; c->aux_on = cur.aux_on; c->defer_zero = cur.defer_zero; c->defer_sources = cur.defer_sources; c->mirror_mode = cur.mirror_mode;
  c->mirror_max_age = cur.mirror_max_age; c->sticky_errors = cur.sticky_errors;
  return c;
  RTE_CATCH("rte_hip_ctx_create")
  return nullptr;
}
// the calling thread's current context from now on (NULL: the process-wide default context); returns the previous one
void* rte_hip_ctx_set_current(void* ctx) {
  void* prev = rte::t_ctx;
  rte::t_ctx = (rte::Context*)ctx;
  return prev;
}
void* rte_hip_ctx_get_current(void) { return rte::t_ctx; }
// drains the context's streams, frees everything it holds; it must not be current on any other thread
int rte_hip_ctx_destroy(void* ctx) {
  if (!ctx) return -1;
  RTE_TRY
  auto* c = (rte::Context*)ctx;
  rte::Context* prev = rte::t_ctx;
  rte::t_ctx = c;
  {
    std::lock_guard<std::recursive_mutex> l(c->mutex);
    rte::prof_resolve();
    rte::release_context_buffers();
    if (c->own_stream) HIP_CHECK(hipStreamDestroy(c->stream));
    if (c->side) HIP_CHECK(hipStreamDestroy(c->side));
    if (c->aux) HIP_CHECK(hipStreamDestroy(c->aux));
    for (hipEvent_t e : {c->ev_fork, c->ev_join, c->ev_aux_fork, c->ev_aux_join, c->ev_h2d})
      if (e) HIP_CHECK(hipEventDestroy(e));
    for (int i = 0; i < rte::Context::kRing; ++i) {
      if (c->ring_ev[i]) HIP_CHECK(hipEventDestroy(c->ring_ev[i]));
      if (c->ring[i]) HIP_CHECK(hipHostFree(c->ring[i]));
    }
  }
  {
    std::lock_guard<std::mutex> l(rte::g_report_mutex);
    for (auto& p : rte::g_report_contexts)
      if (p == c) { p = rte::g_report_contexts.back(); rte::g_report_contexts.pop_back(); break; }
  }
  rte::t_ctx = prev == c ? nullptr : prev;
  delete c;
  return 0;
  RTE_CATCH("rte_hip_ctx_destroy")
  return -1;
}

// ---- error channel
// 1: a failing HIP call is recorded in the context and the entry point returns (all later calls on the context are
// no-ops until rte_hip_clear_error); 0 (default): message on stderr and abort()
int rte_hip_error_mode(int sticky) { LOCK_CTX; rte::ctx().sticky_errors = sticky != 0; return 0; }
// 0: no error since the last clear; otherwise the hipError_t of the first failure (-1 if it had none); the message is
// copied into buf if given
int rte_hip_last_error(char* buf, int buflen) {
  LOCK_CTX;
  rte::Context& c = rte::ctx();
  if (buf && buflen > 0) snprintf(buf, buflen, "%s", c.last_error_msg.c_str());
  return c.last_error;
}
int rte_hip_clear_error(void) {
  LOCK_CTX;
  rte::Context& c = rte::ctx();
  c.last_error = 0;
  c.last_error_msg.clear();
  (void)hipGetLastError();
  return 0;
}

int rte_hip_set_stream(void* s) {
  RTE_TRY
  LOCK_CTX;
  rte::Context& c = rte::ctx();
  if ((hipStream_t)s == c.stream) return 0;
  // work queued on the old stream still uses the scratch arena, the persistent slots and recorded zero fills:
  // materialise the fills there and drain it before anything is launched on the new stream
  rte::flush_pending_zeros();
  rte::flush_pending_sources();
  HIP_CHECK(hipStreamSynchronize(c.stream));  // (forked calls have been joined into it)
  c.fork_valid = false;
  if (c.own_stream) { HIP_CHECK(hipStreamDestroy(c.stream)); c.own_stream = false; }
  c.stream = (hipStream_t)s;
  RTE_CATCH("rte_hip_set_stream")
  return 0;
}
int rte_hip_sync(void) {
  RTE_TRY
  LOCK_CTX;
  rte::flush_pending_zeros();
  rte::flush_pending_sources();
  HIP_CHECK(hipStreamSynchronize(rte::ctx().stream));
  RTE_CATCH("rte_hip_sync")
  return 0;
}
// ---- a sequence of library calls as ONE hipGraph: a chain of ~20 launches replayed with one submission.  Measured (ROCm 7.2,
// tools/experiments/graph_replay.py): the host time per chain halves (0.13 -> 0.05 ms); the device-side gaps between the
// kernels do not shrink (0.323 against 0.334 ms per chain at 1 024 columns, 0.90 against 0.92 at 4 096) -- for host programs
// whose issuing thread is the bottleneck, not a faster device path.  Between begin and end the calls
// do their host-side work as always and their device work is captured from the context's stream (the worklist kernels' side
// stream is forked from it and joined into it, so it is captured with it); nothing runs until the graph is launched.  The
// caller's contract: device pointers only, the same arrays at every launch, one chain run uncaptured beforehand (it sizes the
// scratch arena: an allocation cannot be captured), no call inside the region that returns values to the host.
struct GraphHandle { hipGraphExec_t exec; rte::Context* ctx; long epoch; };
int rte_hip_graph_begin(void) {
  RTE_TRY
  LOCK_CTX;
  rte::flush_pending_zeros();
  rte::flush_pending_sources();
  rte::Context& c = rte::ctx();
  if (c.graph_saved_valid) throw rte::Error{-1, "rte_hip_graph_begin: a capture is open on this context"};
  c.fork_valid = false;
  // the region is captured from a stream of its own (the context's may be the null stream, which cannot be captured)
  if (!c.graph_stream) HIP_CHECK(hipStreamCreateWithFlags(&c.graph_stream, hipStreamNonBlocking));
  HIP_CHECK(hipStreamBeginCapture(c.graph_stream, hipStreamCaptureModeRelaxed));
  c.graph_saved = c.stream; c.graph_saved_valid = true;
  c.stream = c.graph_stream;
  // a linear graph: with the worklist kernels' side stream captured as a parallel branch the replay was SLOWER than the
  // launches themselves (1.21 against 0.94 ms per chain at 4 096 columns; linear: 0.90)
  c.graph_saved_aux = c.aux_on; c.aux_on = false;
  return 0;
  RTE_CATCH("rte_hip_graph_begin")
  return -1;
}
int rte_hip_graph_end(void** graph_exec) {
  RTE_TRY
  LOCK_CTX;
  rte::flush_pending_zeros();  // (a fill recorded and not consumed inside the region belongs to the graph)
  rte::flush_pending_sources();
  rte::Context& c = rte::ctx();
  if (!c.graph_saved_valid) throw rte::Error{-1, "rte_hip_graph_end without rte_hip_graph_begin"};
  c.fork_valid = false;
  hipGraph_t g = nullptr;
  const hipError_t rc_end = hipStreamEndCapture(c.graph_stream, &g);
  c.stream = c.graph_saved; c.graph_saved_valid = false;
  c.aux_on = c.graph_saved_aux;
  HIP_CHECK(rc_end);
  hipGraphExec_t e = nullptr;
  const hipError_t rc = hipGraphInstantiate(&e, g, nullptr, nullptr, 0);
  (void)hipGraphDestroy(g);
  HIP_CHECK(rc);
  // (a capture during which the arena grew or a slot moved has baked in addresses that are gone already)
  *graph_exec = (void*)new GraphHandle{e, &c, c.addr_epoch};
  return 0;
  RTE_CATCH("rte_hip_graph_end")
  return -1;
}
// 0; -4 if library buffers the graph addresses were freed or reallocated since the capture (a larger call on this context,
// rte_hip_release, host tables dropped): the graph is stale, capture it again; -1 on a HIP error
int rte_hip_graph_launch(void* graph_exec) {
  RTE_TRY
  LOCK_CTX;
  GraphHandle* h = (GraphHandle*)graph_exec;
  if (!h || !h->exec) return -1;
  rte::Context& c = rte::ctx();
  if (h->ctx != &c || h->epoch != c.addr_epoch) return -4;
  rte::flush_pending_zeros();
  rte::flush_pending_sources();
  HIP_CHECK(hipGraphLaunch(h->exec, c.stream));
  return 0;
  RTE_CATCH("rte_hip_graph_launch")
  return -1;
}
int rte_hip_graph_destroy(void* graph_exec) {
  GraphHandle* h = (GraphHandle*)graph_exec;
  if (h) {
    if (h->exec) (void)hipGraphExecDestroy(h->exec);
    delete h;
  }
  return 0;
}
// compute_Planck_source leaves factored sources for the rte_lw_solver_noscat call that follows (see above)
int rte_hip_defer_sources(int on) {
  RTE_TRY
  LOCK_CTX;
  rte::flush_pending_sources();
  rte::ctx().defer_sources = on != 0;
  RTE_CATCH("rte_hip_defer_sources")
  return 0;
}
// defer zero_array_* on device buffers until compute_tau_absorption consumes them (see above)
int rte_hip_defer_zero(int on) {
  RTE_TRY
  LOCK_CTX;
  rte::flush_pending_zeros();
  rte::ctx().defer_zero = on != 0;
  RTE_CATCH("rte_hip_defer_zero")
  return 0;
}
// run compute_Planck_source concurrently with the compute_tau_absorption call it directly follows (see above)
int rte_hip_overlap_planck(int on) {
  LOCK_CTX;
  rte::ctx().overlap = on != 0;
  rte::ctx().fork_valid = false;
  return 0;
}
// run the direct-gather worklist of compute_tau_absorption on a second stream inside the call (default on)
int rte_hip_aux_stream(int on) {
  LOCK_CTX;
  rte::ctx().aux_on = on != 0;
  return 0;
}
// host-mirror mode (see above): 1 = outputs marked lazy stay on the device, 0 = off (mirrors are dropped, NOT
// written back: call rte_hip_writeback first for arrays the host still needs)
int rte_hip_host_mirror(int on) {
  RTE_TRY
  LOCK_CTX;
  rte::flush_pending_zeros();
  (void)rte::mirror_on();  // (environment defaults, report registration)
  if (!on) rte::mirror_drop_all();
  rte::ctx().mirror_mode = on ? 1 : 0;
  RTE_CATCH("rte_hip_host_mirror")
  return 0;
}
// copy the device-resident array that contains host address `p` back to the host (whole array) and end its mirror;
// returns 1 if one was written, 0 if the address is not mirrored (the host copy is current)
int rte_hip_writeback(const void* p) {
  RTE_TRY
  LOCK_CTX;
  rte::Context& c = rte::ctx();
  const long hit = rte::mirror_find((const char*)p, 1, true);
  if (hit < 0) return 0;
  rte::Mirror m = c.mirrors[(size_t)hit];
  if (m.zero_pending) HIP_CHECK(hipMemsetAsync(m.dev, 0, m.bytes, c.stream));
  HIP_CHECK(hipMemcpyAsync(m.host, m.dev, m.bytes, hipMemcpyDeviceToHost, c.stream));
  HIP_CHECK(hipStreamSynchronize(c.stream));
  c.mstat[3] += (long long)m.bytes;
  rte::mirror_drop((size_t)hit);
  return 1;
  RTE_CATCH("rte_hip_writeback")
  return -1;
}
int rte_hip_mirror_drop_all(void) {
  RTE_TRY
  LOCK_CTX;
  rte::mirror_drop_all();
  RTE_CATCH("rte_hip_mirror_drop_all")
  return 0;
}
// counters of the host-staging path: 0 mirror hits, 1 mirrors made, 2 host-to-device bytes, 3 device-to-host bytes,
// 4 mirrors dropped because the host memory had changed, 5 dropped for overlap, 6 aged out, 7 zero fills elided,
// 8 live mirrors, 9 device bytes held; which < 0 resets
long long rte_hip_mirror_stat(int which) {
  LOCK_CTX;
  rte::Context& c = rte::ctx();
  if (which < 0) { for (auto& v : c.mstat) v = 0; c.input_hits = 0; c.input_saved = 0; return 0; }
  if (which < 8) return c.mstat[which];
  if (which == 8) return (long long)c.mirrors.size();
  if (which == 9) return (long long)c.mirror_total;
  if (which == 10) return c.input_hits;    // host-produced inputs served from their device copy (Call::stage)
  if (which == 11) return c.input_saved;   // bytes not uploaded for them
  return -1;
}
int rte_hip_device_count(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) { (void)hipGetLastError(); return 0; }
  return n;
}
int rte_hip_profile_enable(int on) {
  LOCK_CTX;
  rte::ctx().prof_on = on != 0;
  return 0;
}
// time only the scope of this name (nullptr or "": all scopes)
int rte_hip_profile_only(const char* name) {
  LOCK_CTX;
  rte::ctx().prof_only = name ? name : "";
  return 0;
}
int rte_hip_profile_reset(void) {
  RTE_TRY
  LOCK_CTX;
  rte::prof_resolve();
  rte::ctx().prof.clear();
  RTE_CATCH("rte_hip_profile_reset")
  return 0;
}
int rte_hip_profile_count(void) {
  RTE_TRY
  LOCK_CTX;
  rte::prof_resolve();
  return (int)rte::ctx().prof.size();
  RTE_CATCH("rte_hip_profile_count")
  return 0;
}
// i-th timed kernel: name copied into buf, launches and total milliseconds returned
int rte_hip_profile_get(int i, char* buf, int buflen, long long* launches, double* total_ms) {
  LOCK_CTX;
  rte::Context& c = rte::ctx();
  if (i < 0 || i >= (int)c.prof.size()) return -1;
  snprintf(buf, buflen, "%s", c.prof[i].name.c_str());
  *launches = c.prof[i].n;
  *total_ms = c.prof[i].ms;
  return 0;
}
// release every device buffer held by the current context (arena, persistent slots, mirrors, cached geometry)
int rte_hip_release(void) {
  RTE_TRY
  LOCK_CTX;
  rte::release_context_buffers();
  RTE_CATCH("rte_hip_release")
  return 0;
}
}
