// readbench.hip -- how many bytes must a CU keep in flight to read the LW solver's three arrays at full speed?
// (hipcc --offload-arch=gfx950 -O3 tools/readbench.hip -o tools/readbench)
// The solver's access pattern (block = 64 columns x 8 waves, wave s reads layers [8s, 8s+8) of tau and lay and levels
// [8s, 8s+9) of lev, g-points of a group one after the other) with PF g-points requested ahead, at 1..4 resident
// blocks per CU (occupancy capped by a dynamic LDS allocation), optionally with a block barrier per g-point and a
// dependent fp64 FMA chain of NF steps per loaded value standing in for the solver's arithmetic.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)

template <int PF, int NF, bool BAR>
__global__ void __launch_bounds__(512) r_solver3(const double* __restrict__ tau, const double* __restrict__ lay,
                                                 const double* __restrict__ lev, int ncol, int nlay, int ng, int gpb,
                                                 double* __restrict__ sink) {
  extern __shared__ double lds[];
  const int lane = threadIdx.x & 63, s = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const unsigned col = min(blockIdx.x * 64 + lane, (unsigned)ncol - 1);
  const size_t ncl = (size_t)ncol * nlay, nclv = (size_t)ncol * (nlay + 1);
  double acc = 0;
  const int gb = blockIdx.y * gpb, ge = min(ng, (int)(blockIdx.y + 1) * gpb);
  double buf[PF + 1][25];
  unsigned ol[8], ov[9];
#pragma unroll
  for (int i = 0; i < 8; ++i) ol[i] = (col + (unsigned)ncol * min(8 * s + i, nlay - 1)) * 8u;
#pragma unroll
  for (int i = 0; i < 9; ++i) ov[i] = (col + (unsigned)ncol * min(8 * s + i, nlay)) * 8u;
  auto load = [&](double (&b)[25], int g) {
    g = min(g, ge - 1);
    const char* pt = (const char*)(tau + ncl * g);
    const char* pl = (const char*)(lay + ncl * g);
    const char* pv = (const char*)(lev + nclv * g);
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      unsigned o = ol[i]; asm volatile("" : "+v"(o));
      b[i] = *(const double*)(pt + o);
      b[8 + i] = *(const double*)(pl + o);
    }
#pragma unroll
    for (int i = 0; i < 9; ++i) { unsigned o = ov[i]; asm volatile("" : "+v"(o)); b[16 + i] = *(const double*)(pv + o); }
  };
#pragma unroll
  for (int p = 0; p < PF; ++p) load(buf[p], gb + p);
  for (int g = gb; g < ge; g += PF + 1) {
#pragma unroll
    for (int p = 0; p <= PF; ++p) {
      if (p == 0 || g + p < ge) {
        load(buf[(p + PF) % (PF + 1)], g + p + PF);
        double t = 0;
#pragma unroll
        for (int i = 0; i < 25; ++i) {
          double v = buf[p][i];
#pragma unroll
          for (int f = 0; f < NF; ++f) v = __builtin_fma(v, 1.0000001, 1e-9);
          t += v;
        }
        acc += t;
        if (BAR) { lds[threadIdx.x] = t; __syncthreads(); acc += lds[threadIdx.x ^ 64]; }
      }
    }
  }
  if (acc == -1.2345) sink[0] = acc;
}

__global__ void fill_random(double* p, size_t n, unsigned seed) {
  for (size_t i = blockIdx.x * 256ull + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
    unsigned long long x = (i + seed) * 6364136223846793005ull + 1442695040888963407ull;
    x ^= x >> 29; x *= 0xbf58476d1ce4e5b9ull; x ^= x >> 32;
    p[i] = (double)(x >> 11) * (1.0 / 9007199254740992.0) + 0.5;
  }
}

int main(int argc, char** argv) {
  const int ncol = 100000, nlay = 60, ng = 256;
  const size_t n = (size_t)ncol * nlay * ng, nv = (size_t)ncol * (nlay + 1) * ng;
  double *tau, *lay, *lev, *sink;
  CK(hipMalloc(&tau, n * 8)); CK(hipMalloc(&lay, n * 8)); CK(hipMalloc(&lev, nv * 8)); CK(hipMalloc(&sink, 8));
  if (argc > 1) {  // random contents (zeros read faster: less power, higher clocks)
    hipLaunchKernelGGL(fill_random, dim3(8192), dim3(256), 0, 0, tau, n, 1u);
    hipLaunchKernelGGL(fill_random, dim3(8192), dim3(256), 0, 0, lay, n, 2u);
    hipLaunchKernelGGL(fill_random, dim3(8192), dim3(256), 0, 0, lev, nv, 3u);
    printf("random contents\n");
  } else {
    CK(hipMemset(tau, 0, n * 8)); CK(hipMemset(lay, 0, n * 8)); CK(hipMemset(lev, 0, nv * 8));
  }
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const double bytes = (2.0 * n + nv) * 8;
  auto t3 = [&](const char* name, auto kern, int blocks_per_cu, int groups) {
    const size_t lds = blocks_per_cu == 1 ? 120 * 1024 : blocks_per_cu == 2 ? 72 * 1024 : blocks_per_cu == 3 ? 50 * 1024 : 8 * 1024;
    CK(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 64));
    auto launch = [&] { hipLaunchKernelGGL(kern, dim3((ncol + 63) / 64, groups), dim3(512), lds, 0, tau, lay, lev, ncol, nlay, ng, ng / groups, sink); };
    launch(); CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    for (int r = 0; r < 3; ++r) launch();
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= 3;
    printf("%-44s blocks/CU %d groups %2d  %7.3f ms  %7.1f GB/s\n", name, blocks_per_cu, groups, ms, bytes / (ms * 1e-3) / 1e9);
  };
#define RUN(PF, NF, BAR) for (int bpc = 1; bpc <= 2; ++bpc) t3("PF=" #PF " fma/value=" #NF " barrier=" #BAR, r_solver3<PF, NF, BAR>, bpc, 4);
  RUN(0, 0, false) RUN(1, 0, false) RUN(2, 0, false) RUN(3, 0, false)
  RUN(1, 0, true) RUN(1, 8, true) RUN(1, 16, true) RUN(2, 16, true) RUN(0, 16, true)
  return 0;
}
