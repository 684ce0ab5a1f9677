// v_fma_f64 on gfx950: issue interval and dependent latency (s_memtime around unrolled chains), one and two waves per SIMD,
// one or two independent chains per wave; the same for v_rcp_f64 and v_ldexp_f64.  Answers whether two waves per SIMD
// cover a fully dependent fp64 chain (the Horner steps of exp in the solvers).
// hipcc --offload-arch=gfx950 -O3 tools/fp64_latency.hip -o /tmp/fp64_latency
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

template <int NCHAIN, int OP>
__global__ void __launch_bounds__(1024) chain_kernel(double* out, long long* clk, int iters, double a, double b) {
  double x[NCHAIN];
  for (int i = 0; i < NCHAIN; ++i) x[i] = threadIdx.x * 1e-3 + i;
  __syncthreads();
  const long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 32; ++u) {
#pragma unroll
      for (int i = 0; i < NCHAIN; ++i) {
        if (OP == 0) x[i] = __builtin_fma(x[i], a, b);
        else if (OP == 1) x[i] = __builtin_amdgcn_rcp(x[i]);
        else x[i] = __builtin_amdgcn_ldexp(x[i], 1);
      }
    }
  }
  double s = 0;
  for (int i = 0; i < NCHAIN; ++i) s += x[i];
  asm volatile("" : "+v"(s));
  const long long t1 = __builtin_amdgcn_s_memtime();
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if ((threadIdx.x & 63) == 0) clk[blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64] = t1 - t0;
}

int main() {
  double* out; long long* clk;
  CK(hipMalloc(&out, 1 << 22)); CK(hipMalloc(&clk, 1 << 16));
  const int iters = 400;
  std::vector<long long> h(64);
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  auto run = [&](const char* name, auto kern, int threads, int nchain) {
    hipLaunchKernelGGL(kern, dim3(1), dim3(threads), 0, 0, out, clk, 10, 0.999, 0.001);
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL(kern, dim3(1), dim3(threads), 0, 0, out, clk, iters, 0.999, 0.001);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    CK(hipMemcpy(h.data(), clk, (threads / 64) * 8, hipMemcpyDeviceToHost));
    long long mx = 0;
    for (int i = 0; i < threads / 64; ++i) mx = h[i] > mx ? h[i] : mx;
    const double nops = iters * 32.0 * nchain;  // per wave
    printf("%-58s %7.2f ticks / op / wave   (%.2f ns / op / wave by the event clock)\n", name, mx / nops, ms * 1e6 / nops);
    return 0;
  };
  run("fma: 1 wave, 1 dependent chain", chain_kernel<1, 0>, 64, 1);
  run("fma: 1 wave, 2 chains", chain_kernel<2, 0>, 64, 2);
  run("fma: 1 wave, 4 chains", chain_kernel<4, 0>, 64, 4);
  run("fma: 4 waves (1 / SIMD), 1 chain", chain_kernel<1, 0>, 256, 1);
  run("fma: 8 waves (2 / SIMD), 1 chain each", chain_kernel<1, 0>, 512, 1);
  run("fma: 8 waves (2 / SIMD), 2 chains each", chain_kernel<2, 0>, 512, 2);
  run("fma: 16 waves (4 / SIMD), 1 chain each", chain_kernel<1, 0>, 1024, 1);
  run("rcp: 1 wave, 1 dependent chain", chain_kernel<1, 1>, 64, 1);
  run("rcp: 1 wave, 4 chains", chain_kernel<4, 1>, 64, 4);
  run("rcp: 8 waves (2 / SIMD), 1 chain each", chain_kernel<1, 1>, 512, 1);
  run("ldexp: 1 wave, 1 dependent chain", chain_kernel<1, 2>, 64, 1);
  run("ldexp: 1 wave, 4 chains", chain_kernel<4, 2>, 64, 4);
  return 0;
}
