// GPU-box helper: aggregate host-to-device rate of N host threads, each with its own stream, copying a pageable array that
// it rewrites before every copy -- through a pinned ring (memcpy + async DMA per 4 MB chunk) or with the runtime's
// pageable copy (same address every time: its pin cache hits).
// build: hipcc -O2 --offload-arch=gfx950 tools/h2d_threads.hip -o /tmp/h2d_threads -lpthread
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <chrono>
#include <thread>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
static void worker(int mode, size_t n, int reps, unsigned evflags, double* secs) {
  const size_t MB = 1 << 20, RING = 4, CH = 4 * MB;
  hipStream_t st; CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
  char* dev; CK(hipMalloc(&dev, n));
  char* pin[RING]; hipEvent_t ev[RING];
  for (size_t i = 0; i < RING; ++i) { CK(hipHostMalloc(&pin[i], CH)); CK(hipEventCreateWithFlags(&ev[i], evflags)); }
  char* h = (char*)malloc(n);
  memset(h, 1, n);
  CK(hipMemcpyAsync(dev, h, n, hipMemcpyHostToDevice, st)); CK(hipStreamSynchronize(st));
  const double t0 = now();
  for (int r = 0; r < reps; ++r) {
    memset(h, r, 4096);  // (token write; the rewrite of the whole array is not what is measured)
    if (mode == 0) {
      CK(hipMemcpyAsync(dev, h, n, hipMemcpyHostToDevice, st));
    } else {
      size_t k = 0;
      for (size_t o = 0; o < n; o += CH, ++k) {
        const size_t len = n - o < CH ? n - o : CH, slot = k % RING;
        if (r > 0 || k >= RING) {
          if (mode == 2) { while (hipEventQuery(ev[slot]) == hipErrorNotReady) {} }
          else CK(hipEventSynchronize(ev[slot]));
        }
        memcpy(pin[slot], h + o, len);
        CK(hipMemcpyAsync(dev + o, pin[slot], len, hipMemcpyHostToDevice, st));
        CK(hipEventRecord(ev[slot], st));
      }
    }
    CK(hipStreamSynchronize(st));
  }
  *secs = now() - t0;
}
int main() {
  const size_t n = (size_t)(17.7 * (1 << 20)) & ~size_t(4095);
  const int reps = 20;
  for (int nthr : {1, 2, 4, 8, 16})
    for (int mode = 0; mode < 4; ++mode) {
      std::vector<std::thread> th; std::vector<double> secs(nthr);
      const unsigned evf = mode == 3 ? (hipEventDisableTiming | hipEventBlockingSync) : hipEventDisableTiming;
      const double t0 = now();
      for (int i = 0; i < nthr; ++i) th.emplace_back(worker, mode == 3 ? 1 : mode, n, reps, evf, &secs[i]);
      for (auto& t : th) t.join();
      double mx = 0; for (double s : secs) mx = s > mx ? s : mx;
      printf("threads %2d %-22s %8.2f GB/s aggregate (slowest thread %.3f s, wall %.3f s)\n", nthr,
             mode == 0 ? "plain (pin cache hit)" : mode == 1 ? "ring, event sync" : mode == 2 ? "ring, event query spin" : "ring, blocking events",
             n * (double)reps * nthr / mx * 1e-9, mx, now() - t0);
    }
  return 0;
}
