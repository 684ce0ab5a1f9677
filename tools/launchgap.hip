// What does a dependent launch cost on the GPU timeline?  Chains of tiny kernels on one stream, timed with events:
// same kernel / alternating kernels / big kernarg / LDS / scratch / preceded by a kernel that dirties a lot of HBM.
// hipcc --offload-arch=gfx950 -O3 tools/launchgap.hip -o /tmp/launchgap
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
struct Big { double v[60]; int* p; };
__global__ void k_a(int* p) { if (threadIdx.x == 0 && blockIdx.x == 0) p[0] += 1; }
__global__ void k_b(int* p) { if (threadIdx.x == 0 && blockIdx.x == 0) p[1] += 1; }
__global__ void k_big(Big b) { if (threadIdx.x == 0 && blockIdx.x == 0) b.p[2] += (int)b.v[3]; }
__global__ void k_lds(int* p) { extern __shared__ int s[]; s[threadIdx.x] = threadIdx.x; __syncthreads(); if (threadIdx.x == 0 && blockIdx.x == 0) p[3] += s[5]; }
__global__ void k_scratch(int* p, int n) { volatile int a[64]; for (int i = 0; i < 64; ++i) a[i] = i * n; if (threadIdx.x == 0 && blockIdx.x == 0) p[4] += a[n & 63]; }
__global__ void k_fill(double* x, size_t n) { for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) x[i] = 1.0; }
__global__ void k_wide(int* p) { if (threadIdx.x == 0 && blockIdx.x == 0) p[5] += 1; }
int main() {
  int* p; CK(hipMalloc(&p, 64)); CK(hipMemset(p, 0, 64));
  double* x; const size_t n = size_t(1) << 27; CK(hipMalloc(&x, n * 8));
  hipStream_t st; CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const int N = 2000;
  auto run = [&](const char* name, auto&& body, int per) {
    for (int i = 0; i < 50; ++i) body(i);
    hipStreamSynchronize(st);
    hipEventRecord(e0, st);
    for (int i = 0; i < N; ++i) body(i);
    hipEventRecord(e1, st);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("%-44s %7.2f us per launch\n", name, ms * 1e3 / (N * per));
  };
  Big b{}; b.p = p;
  run("same kernel", [&](int) { hipLaunchKernelGGL(k_a, dim3(1), dim3(64), 0, st, p); }, 1);
  run("alternating two kernels", [&](int i) { if (i & 1) hipLaunchKernelGGL(k_a, dim3(1), dim3(64), 0, st, p); else hipLaunchKernelGGL(k_b, dim3(1), dim3(64), 0, st, p); }, 1);
  run("big kernarg (488 B)", [&](int) { hipLaunchKernelGGL(k_big, dim3(1), dim3(64), 0, st, b); }, 1);
  run("dynamic LDS 64 KB", [&](int) { hipLaunchKernelGGL(k_lds, dim3(1), dim3(64), 65536, st, p); }, 1);
  run("alternating LDS 64 KB / none", [&](int i) { if (i & 1) hipLaunchKernelGGL(k_lds, dim3(1), dim3(64), 65536, st, p); else hipLaunchKernelGGL(k_a, dim3(1), dim3(64), 0, st, p); }, 1);
  run("scratch", [&](int i) { hipLaunchKernelGGL(k_scratch, dim3(1), dim3(64), 0, st, p, i); }, 1);
  run("alternating scratch / none", [&](int i) { if (i & 1) hipLaunchKernelGGL(k_scratch, dim3(1), dim3(64), 0, st, p, i); else hipLaunchKernelGGL(k_a, dim3(1), dim3(64), 0, st, p); }, 1);
  run("wide grid (100k blocks x 256)", [&](int) { hipLaunchKernelGGL(k_wide, dim3(100000), dim3(256), 0, st, p); }, 1);
  run("wide grid 640-thread blocks", [&](int) { hipLaunchKernelGGL(k_wide, dim3(12000), dim3(640), 0, st, p); }, 1);
  {
    // a 1 GiB fill followed by a tiny kernel: fill alone, then the pair
    const int M = 200;
    auto t = [&](bool tiny) {
      hipStreamSynchronize(st); hipEventRecord(e0, st);
      for (int i = 0; i < M; ++i) { hipLaunchKernelGGL(k_fill, dim3(4096), dim3(256), 0, st, x, n); if (tiny) hipLaunchKernelGGL(k_a, dim3(1), dim3(64), 0, st, p); }
      hipEventRecord(e1, st); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1); return ms * 1e3 / M;
    };
    t(false);
    const double a = t(false), c = t(true);
    printf("1 GiB fill %.1f us; fill + tiny kernel %.1f us (tiny adds %.2f us)\n", a, c, c - a);
  }
  {
    // with an event record + wait from a second stream between launches (the aux fork/join)
    hipStream_t s2; CK(hipStreamCreateWithFlags(&s2, hipStreamNonBlocking));
    hipEvent_t f, j; CK(hipEventCreateWithFlags(&f, hipEventDisableTiming)); CK(hipEventCreateWithFlags(&j, hipEventDisableTiming));
    run("fork/join around a second-stream kernel", [&](int) {
      hipLaunchKernelGGL(k_a, dim3(1), dim3(64), 0, st, p);
      hipEventRecord(f, st); hipStreamWaitEvent(s2, f, 0);
      hipLaunchKernelGGL(k_b, dim3(1), dim3(64), 0, s2, p);
      hipEventRecord(j, s2); hipStreamWaitEvent(st, j, 0);
    }, 1);
  }
  return 0;
}
