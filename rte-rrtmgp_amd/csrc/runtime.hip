// runtime.hip -- stream, scratch arena, host-pointer staging and kernel timing hooks.
//
// Drop-in semantics: the reference kernels own nothing -- every buffer is
// the caller's -- and take host arrays from the unchanged Fortran frontend.  This library accepts
// BOTH kinds of pointer on every array argument:
//   device pointer -> the kernel is launched in place, asynchronously, on the library stream;
//   host pointer   -> the array is staged through the scratch arena (H2D before, D2H after) and
//                     the call returns only after the stream has drained (functional mode).
#include <fcntl.h>
#include <string.h>
#include <unistd.h>

#include <chrono>
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

static long g_seq = 0;
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
//     written back within g_mirror_max_age library calls are dropped (their host arrays are usually gone by then).
// Contract of the mode: host code does not READ a lazily held array before writing it back, and does not write PART of
// one.  The reference's clear-sky / all-sky LW frontend satisfies it; its SW gas optics combines tau and tau_rayleigh on
// the host (mo_gas_optics_rrtmgp.F90:1954-2036), so compute_tau_rayleigh writes both back (writeback_produced_by).
struct Mirror {
  char* host; size_t bytes;
  char* dev; size_t cap;
  long last_use;
  unsigned long long magic;
  const char* producer;
  bool zero_pending;  // entirely zero by a recorded zero_array; the device copy has not been filled yet
};
static std::vector<Mirror> g_mirrors;
struct FreeBuf { char* dev; size_t cap; };
static std::vector<FreeBuf> g_mirror_free;
static int g_mirror_mode = -1;          // -1: take RTE_HIP_HOST_MIRROR at the first call
static size_t g_mirror_total = 0;       // device bytes held by mirrors and the free list
static size_t g_mirror_limit = 0;
static long g_mirror_max_age = 64;
static int g_procmem_fd = -2;
static unsigned long long g_magic_state = 0x9E3779B97F4A7C15ull;
static hipEvent_t g_ev_h2d = nullptr;
static long long g_mstat[8] = {0, 0, 0, 0, 0, 0, 0, 0};  // hits, mirrors made, H2D bytes, D2H bytes, dropped (host changed), dropped (overlap), aged out, zero fills elided
// host wall-clock spent inside the library's host-array path (RTE_HIP_STAGING_REPORT=1 prints it when the process ends)
static double g_t_call = 0, g_t_h2d = 0, g_t_wait = 0, g_t_find = 0;
static long g_n_calls = 0;
static std::chrono::steady_clock::time_point g_call_t0;
static inline double secs_since(std::chrono::steady_clock::time_point t0) {
  return std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
}
static void staging_report() {
  fprintf(stderr, "rte_rrtmgp_hip staging report: %ld calls, %.3f s inside the library (host-to-device copies %.3f s for %.3f GB, "
          "waits + device-to-host %.3f s for %.3f GB, mirror look-ups %.3f s); mirrors made %lld, hits %lld, dropped %lld + %lld, aged %lld, "
          "zero fills elided %lld, device bytes held %.2f GB\n", g_n_calls, g_t_call, g_t_h2d, g_mstat[2] * 1e-9, g_t_wait, g_mstat[3] * 1e-9,
          g_t_find, g_mstat[1], g_mstat[0], g_mstat[4], g_mstat[5], g_mstat[6], g_mstat[7], g_mirror_total * 1e-9);
}
constexpr int kCanaries = 34;
constexpr size_t kLazyMinBytes = 4096;

static bool mirror_on() {
  if (g_mirror_mode < 0) {
    const char* e = getenv("RTE_HIP_HOST_MIRROR");
    g_mirror_mode = (e && atoi(e) > 0) ? 1 : 0;
    if (const char* r = getenv("RTE_HIP_STAGING_REPORT")) if (atoi(r) > 0) atexit(staging_report);
    if (const char* a = getenv("RTE_HIP_MIRROR_MAX_AGE")) g_mirror_max_age = atol(a) > 0 ? atol(a) : g_mirror_max_age;
  }
  return g_mirror_mode == 1;
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
// marks "the host-to-device copies queued so far": in host-mirror mode a call with staged inputs only waits for THIS, not
// for its kernels (every in()/out() conversion precedes the call's first launch)
static void mark_h2d() {
  if (!g_ev_h2d) HIP_CHECK(hipEventCreateWithFlags(&g_ev_h2d, hipEventDisableTiming));
  HIP_CHECK(hipEventRecord(g_ev_h2d, g_stream));
}
static void mirror_release_buffer(char* dev, size_t cap) { g_mirror_free.push_back(FreeBuf{dev, cap}); }
static void mirror_trim_free_list() {
  if (g_mirror_free.empty()) return;
  HIP_CHECK(hipStreamSynchronize(g_stream));  // kernels of earlier calls may still use them
  for (auto& f : g_mirror_free) { HIP_CHECK(hipFree(f.dev)); g_mirror_total -= f.cap; }
  g_mirror_free.clear();
}
static char* mirror_alloc(size_t bytes, size_t* cap_out) {
  size_t best = (size_t)-1;
  for (size_t i = 0; i < g_mirror_free.size(); ++i) {
    const size_t cap = g_mirror_free[i].cap;
    if (cap >= bytes && cap <= bytes + bytes / 4 + (size_t(1) << 20) && (best == (size_t)-1 || cap < g_mirror_free[best].cap)) best = i;
  }
  if (best != (size_t)-1) {
    FreeBuf f = g_mirror_free[best];
    g_mirror_free.erase(g_mirror_free.begin() + best);
    *cap_out = f.cap;
    return f.dev;
  }
  if (g_mirror_limit == 0) {
    if (const char* e = getenv("RTE_HIP_MIRROR_MAX_GB")) g_mirror_limit = (size_t)(atof(e) * 1073741824.0);
    if (g_mirror_limit == 0) {
      size_t fr = 0, tot = 0;
      HIP_CHECK(hipMemGetInfo(&fr, &tot));
      g_mirror_limit = fr / 10 * 6;
    }
  }
  const size_t cap = (bytes + (size_t(2) << 20) - 1) & ~((size_t(2) << 20) - 1);
  if (g_mirror_total + cap > g_mirror_limit) mirror_trim_free_list();
  char* d = nullptr;
  HIP_CHECK(hipMalloc((void**)&d, cap));
  g_mirror_total += cap;
  *cap_out = cap;
  return d;
}
static void mirror_drop(size_t i) {
  mirror_release_buffer(g_mirrors[i].dev, g_mirrors[i].cap);
  g_mirrors.erase(g_mirrors.begin() + i);
}
static void mirror_age_out() {
  for (size_t i = g_mirrors.size(); i-- > 0;)
    if (g_seq - g_mirrors[i].last_use > g_mirror_max_age) { mirror_drop(i); ++g_mstat[6]; }
}
static void mirror_drop_all() {
  while (!g_mirrors.empty()) mirror_drop(g_mirrors.size() - 1);
  mirror_trim_free_list();
}
// the mirror that CONTAINS [p, p+bytes) with its canaries intact (index), or -1; mirrors that merely overlap the range, or
// whose host memory was changed, are dropped on the way (the host reused the memory)
static long mirror_find(const char* p, size_t bytes) {
  long hit = -1;
  for (size_t i = g_mirrors.size(); i-- > 0;) {
    Mirror& m = g_mirrors[i];
    if (p + bytes <= m.host || m.host + m.bytes <= p) continue;
    const bool contained = m.host <= p && p + bytes <= m.host + m.bytes;
    if (contained && hit < 0 && canaries_intact(m)) { hit = (long)i; continue; }
    ++g_mstat[contained ? 4 : 5];
    mirror_drop(i);
    if (hit > (long)i) --hit;
  }
  return hit;
}

// ---- Call ---------------------------------------------------------------------------------------
long call_seq() { return g_seq; }  // number of the current (innermost) API call

Call::Call(const char* n) : name(n) {
  g_mutex.lock();
  ++g_seq;
  ++g_n_calls;
  g_call_t0 = std::chrono::steady_clock::now();
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
  if (mirror_on() && !g_mirrors.empty()) mirror_age_out();
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

void* Call::stage(void* p, size_t bytes, bool copy_in, bool copy_out, bool lazy, bool* zero_fill) {
  if (zero_fill) *zero_fill = false;
  if (!p || bytes == 0) return p;
  void* dv;
  const int kind = classify(p, &dv);
  if (kind == 1) return p;
  if (kind == 2) { host_visible_ = true; return dv; }  // in place, but synchronous for the caller (see ~Call)
  if (mirror_on()) {
    const auto tf = std::chrono::steady_clock::now();
    const long hit = mirror_find((const char*)p, bytes);
    g_t_find += secs_since(tf);
    if (hit >= 0) {
      Mirror& m = g_mirrors[(size_t)hit];
      m.last_use = g_seq;
      ++g_mstat[0];
      char* d = m.dev + ((const char*)p - m.host);
      if (m.zero_pending) {
        if (zero_fill && copy_out && lazy && bytes == m.bytes) { *zero_fill = true; ++g_mstat[7]; }  // the caller overwrites all of it
        else HIP_CHECK(hipMemsetAsync(m.dev, 0, m.bytes, g_stream));
        m.zero_pending = false;
      }
      if (!copy_out) return d;                        // input: served from the device copy
      if (lazy) { m.producer = name; return d; }      // written again on the device; the host copy stays unspecified
      // an output the caller reads on the host lies inside a mirrored range: the whole array goes back and the mirror ends
      if (n_back_ >= 16 || n_recycle_ >= 16) { fprintf(stderr, "rte_rrtmgp_hip: too many staged outputs\n"); abort(); }
      back_[n_back_++] = Back{m.host, m.dev, m.bytes};
      recycle_[n_recycle_++] = Recycle{m.dev, m.cap};
      g_mirrors.erase(g_mirrors.begin() + hit);
      return d;
    }
    if (copy_out && lazy && bytes >= kLazyMinBytes && n_lazy_ < 16) {
      size_t cap = 0;
      char* d = mirror_alloc(bytes, &cap);
      if (copy_in) {
        const auto t0 = std::chrono::steady_clock::now();
        HIP_CHECK(hipMemcpyAsync(d, p, bytes, hipMemcpyHostToDevice, g_stream));
        g_t_h2d += secs_since(t0);
        staged_in_ = true;
        g_mstat[2] += (long long)bytes;
        mark_h2d();
      }
      g_magic_state = g_magic_state * 6364136223846793005ull + 1442695040888963407ull;
      Mirror m{(char*)p, bytes, d, cap, g_seq, g_magic_state ^ (unsigned long long)(uintptr_t)p, name, false};
      g_mirrors.push_back(m);
      lazy_[n_lazy_++] = Lazy{p, bytes, m.magic};
      ++g_mstat[1];
      return d;
    }
  }
  void* d = scratch(bytes);
  if (copy_in) {
    const auto t0 = std::chrono::steady_clock::now();
    HIP_CHECK(hipMemcpyAsync(d, p, bytes, hipMemcpyHostToDevice, g_stream));
    g_t_h2d += secs_since(t0);
    staged_in_ = true;
    g_mstat[2] += (long long)bytes;
    if (g_mirror_mode == 1) mark_h2d();
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
  const long hit = mirror_find((const char*)p, bytes);
  if (hit >= 0) {
    Mirror& m = g_mirrors[(size_t)hit];
    m.last_use = g_seq;
    m.producer = name;
    ++g_mstat[0];
    if (bytes == m.bytes) m.zero_pending = true;  // whole array: recorded, filled only if somebody reads it
    else HIP_CHECK(hipMemsetAsync(m.dev + ((const char*)p - m.host), 0, bytes, g_stream));
    return true;
  }
  if (n_lazy_ >= 16) return false;
  size_t cap = 0;
  char* d = mirror_alloc(bytes, &cap);
  g_magic_state = g_magic_state * 6364136223846793005ull + 1442695040888963407ull;
  Mirror m{(char*)p, bytes, d, cap, g_seq, g_magic_state ^ (unsigned long long)(uintptr_t)p, name, true};
  g_mirrors.push_back(m);
  lazy_[n_lazy_++] = Lazy{p, bytes, m.magic};
  ++g_mstat[1];
  return true;
}

void Call::writeback_produced_by(const char* producer) {
  if (!mirror_on()) return;
  for (size_t i = g_mirrors.size(); i-- > 0;) {
    Mirror& m = g_mirrors[i];
    if (strcmp(m.producer, producer) != 0 || g_seq - m.last_use > 8) continue;
    if (n_back_ >= 16 || n_recycle_ >= 16) break;
    if (m.zero_pending) { HIP_CHECK(hipMemsetAsync(m.dev, 0, m.bytes, g_stream)); m.zero_pending = false; }
    back_[n_back_++] = Back{m.host, m.dev, m.bytes};
    recycle_[n_recycle_++] = Recycle{m.dev, m.cap};
    g_mirrors.erase(g_mirrors.begin() + i);
  }
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
  const bool mirror = g_mirror_mode == 1;
  const auto tw = std::chrono::steady_clock::now();
  for (int i = 0; i < n_back_; ++i) {
    HIP_CHECK(hipMemcpyAsync(back_[i].host, back_[i].dev, back_[i].bytes, hipMemcpyDeviceToHost, g_stream));
    g_mstat[3] += (long long)back_[i].bytes;
  }
  // host arrays (staged, or host-visible memory used in place): the caller owns them again when the call returns.
  // In host-mirror mode a call that staged inputs only waits for those copies (mark_h2d), not for its kernels: they run
  // while the host program prepares the next call.
  if (n_back_ > 0 || host_visible_ || (staged_in_ && !mirror)) HIP_CHECK(hipStreamSynchronize(g_stream));
  else if (staged_in_) HIP_CHECK(hipEventSynchronize(g_ev_h2d));
  g_t_wait += secs_since(tw);
  for (int i = 0; i < n_lazy_; ++i) write_canaries(lazy_[i].host, lazy_[i].bytes, lazy_[i].magic);
  for (int i = 0; i < n_recycle_; ++i) mirror_release_buffer((char*)recycle_[i].dev, recycle_[i].cap);
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
  g_t_call += secs_since(g_call_t0);
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
// host-mirror mode (see runtime.hip): 1 = outputs marked lazy stay on the device, 0 = off (mirrors are dropped, NOT
// written back: call rte_hip_writeback first for arrays the host still needs)
int rte_hip_host_mirror(int on) {
  std::lock_guard<std::recursive_mutex> l(rte::g_mutex);
  rte::flush_pending_zeros();
  if (!on) rte::mirror_drop_all();
  rte::g_mirror_mode = on ? 1 : 0;
  return 0;
}
// copy the device-resident array that contains host address `p` back to the host (whole array) and end its mirror;
// returns 1 if one was written, 0 if the address is not mirrored (the host copy is current)
int rte_hip_writeback(const void* p) {
  std::lock_guard<std::recursive_mutex> l(rte::g_mutex);
  const long hit = rte::mirror_find((const char*)p, 1);
  if (hit < 0) return 0;
  rte::Mirror m = rte::g_mirrors[(size_t)hit];
  if (m.zero_pending) HIP_CHECK(hipMemsetAsync(m.dev, 0, m.bytes, rte::g_stream));
  HIP_CHECK(hipMemcpyAsync(m.host, m.dev, m.bytes, hipMemcpyDeviceToHost, rte::g_stream));
  HIP_CHECK(hipStreamSynchronize(rte::g_stream));
  rte::g_mstat[3] += (long long)m.bytes;
  rte::mirror_drop((size_t)hit);
  return 1;
}
int rte_hip_mirror_drop_all(void) {
  std::lock_guard<std::recursive_mutex> l(rte::g_mutex);
  rte::mirror_drop_all();
  return 0;
}
// counters of the host-staging path: 0 mirror hits, 1 mirrors made, 2 host-to-device bytes, 3 device-to-host bytes,
// 4 mirrors dropped because the host memory had changed, 5 dropped for overlap, 6 aged out, 7 zero fills elided,
// 8 live mirrors, 9 device bytes held; which < 0 resets
long long rte_hip_mirror_stat(int which) {
  std::lock_guard<std::recursive_mutex> l(rte::g_mutex);
  if (which < 0) { for (auto& v : rte::g_mstat) v = 0; return 0; }
  if (which < 8) return rte::g_mstat[which];
  if (which == 8) return (long long)rte::g_mirrors.size();
  if (which == 9) return (long long)rte::g_mirror_total;
  return -1;
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
  rte::mirror_drop_all();
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
