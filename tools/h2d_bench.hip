// GPU-box helper: how fast do freshly allocated pageable host arrays (what the Fortran frontend hands over: automatic and
// allocatable arrays, mmap'ed per call above malloc's threshold) reach the device?  hipMemcpyAsync as it is, in chunks,
// through a pinned staging ring filled with memcpy (1 or more host threads), and after hipHostRegister.
// build: hipcc -O2 --offload-arch=gfx950 tools/h2d_bench.hip -o gpurun_out/h2d_bench -lpthread
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <chrono>
#include <thread>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

int main(int argc, char** argv) {
  const size_t MB = 1 << 20;
  const double sizes_mb[] = {2.0, 8.0, 17.7, 35.4, 70.8, 141.6};
  hipStream_t st; CK(hipStreamCreate(&st));
  char* dev; CK(hipMalloc(&dev, 160 * MB));
  const size_t RING = 4, CH = 4 * MB;
  char* pin[RING]; hipEvent_t ev[RING];
  for (size_t i = 0; i < RING; ++i) { CK(hipHostMalloc(&pin[i], CH)); CK(hipEventCreateWithFlags(&ev[i], hipEventDisableTiming)); }
  printf("%10s %12s %12s %12s %12s %12s %12s\n", "MB", "plain", "chunk8MB", "ring1thr", "ring2thr", "register", "reused-plain");
  for (double smb : sizes_mb) {
    const size_t n = (size_t)(smb * MB) & ~size_t(4095);
    double t[6] = {0, 0, 0, 0, 0, 0};
    const int reps = 8;
    char* reused = (char*)malloc(n); memset(reused, 1, n);
    for (int mode = 0; mode < 6; ++mode) {
      for (int r = 0; r < reps; ++r) {
        char* h = mode == 5 ? reused : (char*)malloc(n);
        memset(h, r + 1, n);  // the frontend writes the array
        const double t0 = now();
        if (mode == 0 || mode == 5) {
          CK(hipMemcpyAsync(dev, h, n, hipMemcpyHostToDevice, st));
        } else if (mode == 1) {
          for (size_t o = 0; o < n; o += 8 * MB) CK(hipMemcpyAsync(dev + o, h + o, n - o < 8 * MB ? n - o : 8 * MB, hipMemcpyHostToDevice, st));
        } else if (mode == 2 || mode == 3) {
          const int nthr = mode == 2 ? 1 : 2;
          size_t k = 0;
          for (size_t o = 0; o < n; o += CH, ++k) {
            const size_t len = n - o < CH ? n - o : CH;
            const size_t slot = k % RING;
            if (k >= RING) CK(hipEventSynchronize(ev[slot]));
            if (nthr == 1) memcpy(pin[slot], h + o, len);
            else {
              std::thread th([&] { memcpy(pin[slot], h + o, len / 2); });
              memcpy(pin[slot] + len / 2, h + o + len / 2, len - len / 2);
              th.join();
            }
            CK(hipMemcpyAsync(dev + o, pin[slot], len, hipMemcpyHostToDevice, st));
            CK(hipEventRecord(ev[slot], st));
          }
        } else if (mode == 4) {
          CK(hipHostRegister(h, n, hipHostRegisterDefault));
          CK(hipMemcpyAsync(dev, h, n, hipMemcpyHostToDevice, st));
        }
        CK(hipStreamSynchronize(st));
        if (mode == 4) CK(hipHostUnregister(h));
        const double dt = now() - t0;
        if (r > 0) t[mode] += dt;
        if (mode != 5) free(h);
      }
    }
    printf("%10.1f", smb);
    for (int mode = 0; mode < 6; ++mode) printf(" %9.2f GB/s", n * (reps - 1) / t[mode] * 1e-9);
    printf("\n");
    free(reused);
  }
  return 0;
}
