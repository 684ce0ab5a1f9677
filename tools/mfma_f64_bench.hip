// v_mfma_f64_16x16x4_f64 on gfx950: operand / result layout (checked against a host product with an asymmetric B),
// issue interval and dependent-accumulator latency (s_memtime around unrolled chains; 1, 2 and 4 waves per SIMD).
// hipcc --offload-arch=gfx950 -O3 tools/mfma_f64_bench.hip -o /tmp/mfma_f64_bench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
typedef double v4d __attribute__((ext_vector_type(4)));

// D(16x16) = A(16x4) B(4x16): lane l supplies A[l & 15][l >> 4] and B[l >> 4][l & 15]; result reg r of lane l is
// D[(l >> 4) + 4 r][l & 15]
__global__ void layout_kernel(const double* A, const double* B, double* D) {
  const int l = threadIdx.x;
  v4d c = {0, 0, 0, 0};
  c = __builtin_amdgcn_mfma_f64_16x16x4f64(A[(l & 15) * 4 + (l >> 4)], B[(l >> 4) * 16 + (l & 15)], c, 0, 0, 0);
  for (int r = 0; r < 4; ++r) D[((l >> 4) + 4 * r) * 16 + (l & 15)] = c[r];
}

template <int NACC>
__global__ void __launch_bounds__(1024) rate_kernel(double* out, long long* clk, int iters) {
  v4d c[NACC];
  for (int i = 0; i < NACC; ++i) c[i] = {0, 0, 0, 0};
  double a = threadIdx.x * 1e-3, b = 1.0 + threadIdx.x * 1e-4;
  __syncthreads();
  const long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 8; ++u) {
#pragma unroll
      for (int i = 0; i < NACC; ++i) c[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c[i], 0, 0, 0);
    }
  }
  asm volatile("s_nop 7\n\ts_nop 7" ::: "memory");
  double s = 0;
  for (int i = 0; i < NACC; ++i) s += c[i][0] + c[i][1] + c[i][2] + c[i][3];
  const long long t1 = __builtin_amdgcn_s_memtime();
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if ((threadIdx.x & 63) == 0) clk[blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64] = t1 - t0;
}

// the same with fp64 VALU FMAs beside the MFMAs of a partner wave (do the two pipes run side by side?)
__global__ void __launch_bounds__(512) mixed_kernel(double* out, long long* clk, int iters) {
  const int w = threadIdx.x >> 6;
  double a = threadIdx.x * 1e-3, b = 1.0 + threadIdx.x * 1e-4;
  __syncthreads();
  const long long t0 = __builtin_amdgcn_s_memtime();
  double s = 0;
  if (w < 4) {
    v4d c0 = {0, 0, 0, 0}, c1 = {0, 0, 0, 0};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        c0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c0, 0, 0, 0);
        c1 = __builtin_amdgcn_mfma_f64_16x16x4f64(b, a, c1, 0, 0, 0);
      }
    }
    s = c0[0] + c0[1] + c1[2] + c1[3];
  } else {
    double x[8];
    for (int i = 0; i < 8; ++i) x[i] = i;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int u = 0; u < 32; ++u) {
#pragma unroll
        for (int i = 0; i < 8; ++i) x[i] = fma(x[i], a, b);
      }
    }
    for (int i = 0; i < 8; ++i) s += x[i];
  }
  const long long t1 = __builtin_amdgcn_s_memtime();
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if ((threadIdx.x & 63) == 0) clk[blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64] = t1 - t0;
}

int main() {
  double *dA, *dB, *dD, *out; long long* clk;
  CK(hipMalloc(&dA, 64 * 8)); CK(hipMalloc(&dB, 64 * 8)); CK(hipMalloc(&dD, 256 * 8));
  CK(hipMalloc(&out, 1 << 22)); CK(hipMalloc(&clk, 1 << 16));
  std::vector<double> A(64), B(64), D(256);
  for (int i = 0; i < 16; ++i) for (int k = 0; k < 4; ++k) A[i * 4 + k] = 1.0 + i + 0.01 * k * k;
  for (int k = 0; k < 4; ++k) for (int j = 0; j < 16; ++j) B[k * 16 + j] = 0.5 + 3.0 * k + 0.001 * j * (k + 1);
  CK(hipMemcpy(dA, A.data(), 64 * 8, hipMemcpyHostToDevice)); CK(hipMemcpy(dB, B.data(), 64 * 8, hipMemcpyHostToDevice));
  hipLaunchKernelGGL(layout_kernel, dim3(1), dim3(64), 0, 0, dA, dB, dD);
  CK(hipMemcpy(D.data(), dD, 256 * 8, hipMemcpyDeviceToHost));
  double worst = 0;
  for (int i = 0; i < 16; ++i) for (int j = 0; j < 16; ++j) {
    double r = 0;
    for (int k = 0; k < 4; ++k) r = fma(A[i * 4 + k], B[k * 16 + j], r);
    worst = fmax(worst, fabs(r - D[i * 16 + j]) / fabs(r));
  }
  printf("layout check: worst relative difference from the host's k-ordered fma chain %.3e (0 = same chain)\n", worst);
  const int iters = 200;
  std::vector<long long> h(1024);
  auto report = [&](const char* name, int waves, double mfma_per_wave) {
    CK(hipDeviceSynchronize());
    CK(hipMemcpy(h.data(), clk, waves * 8, hipMemcpyDeviceToHost));
    long long mx = 0;
    for (int i = 0; i < waves; ++i) mx = h[i] > mx ? h[i] : mx;
    printf("%-60s %8lld ticks  -> %.1f memtime ticks per MFMA per wave\n", name, mx, mx / mfma_per_wave);
    return 0;
  };
  // s_memtime ticks at a fixed 100 MHz; convert with the wall clock of a long run below
  hipLaunchKernelGGL(rate_kernel<1>, dim3(1), dim3(64), 0, 0, out, clk, iters); report("1 wave, 1 accumulator (dependent chain)", 1, iters * 8.0);
  hipLaunchKernelGGL(rate_kernel<2>, dim3(1), dim3(64), 0, 0, out, clk, iters); report("1 wave, 2 accumulators", 1, iters * 16.0);
  hipLaunchKernelGGL(rate_kernel<4>, dim3(1), dim3(64), 0, 0, out, clk, iters); report("1 wave, 4 accumulators", 1, iters * 32.0);
  hipLaunchKernelGGL(rate_kernel<1>, dim3(1), dim3(256), 0, 0, out, clk, iters); report("4 waves (1 / SIMD), 1 accumulator", 4, iters * 8.0);
  hipLaunchKernelGGL(rate_kernel<1>, dim3(1), dim3(512), 0, 0, out, clk, iters); report("8 waves (2 / SIMD), 1 accumulator each", 8, iters * 8.0);
  hipLaunchKernelGGL(rate_kernel<2>, dim3(1), dim3(512), 0, 0, out, clk, iters); report("8 waves (2 / SIMD), 2 accumulators each", 8, iters * 16.0);
  hipLaunchKernelGGL(rate_kernel<1>, dim3(1), dim3(1024), 0, 0, out, clk, iters); report("16 waves (4 / SIMD), 1 accumulator each", 16, iters * 8.0);
  hipLaunchKernelGGL(mixed_kernel, dim3(1), dim3(256), 0, 0, out, clk, iters); report("4 MFMA waves alone (2 acc)", 4, iters * 16.0);
  hipLaunchKernelGGL(mixed_kernel, dim3(1), dim3(512), 0, 0, out, clk, iters); report("4 MFMA waves + 4 fp64-FMA waves (256 v_fma_f64 per iter)", 8, iters * 16.0);
  // wall clock: whole chip, 4 waves per SIMD, 2 accumulators -> TFLOP/s
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const int big = 2000;
  hipLaunchKernelGGL(rate_kernel<2>, dim3(256), dim3(1024), 0, 0, out, clk, 10);
  CK(hipEventRecord(e0));
  hipLaunchKernelGGL(rate_kernel<2>, dim3(256), dim3(1024), 0, 0, out, clk, big);
  CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
  float ms; CK(hipEventElapsedTime(&ms, e0, e1));
  const double nm = 256.0 * 16 * big * 16;
  printf("whole chip: %.3f ms for %.3g MFMAs -> %.1f TFLOP/s fp64, %.1f ns per MFMA per SIMD\n", ms, nm, nm * 2048 / ms / 1e9,
         ms * 1e6 / (nm / 1024));
  CK(hipMemcpy(h.data(), clk, 8, hipMemcpyDeviceToHost));
  printf("  (block 0 wave 0: %lld memtime ticks for the same run -> %.2f ns per tick)\n", h[0], ms * 1e6 / h[0]);
  return 0;
}
