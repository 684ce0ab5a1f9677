// writebench2.hip -- store-policy / store-width micro-benchmark for the (col, lay, g) output planes
// (hipcc --offload-arch=gfx950 -O3 tools/writebench2.hip -o tools/writebench2).
// Question: hipMemsetAsync writes these planes at 6.1-6.4 TB/s while every plain 8-byte-per-lane store pattern
// tops out at 5.4-5.5 TB/s (tools/writebench.hip).  Which ingredient closes the gap: cache policy (nt / sc1 /
// sc0 sc1), 16 bytes per lane, or the visiting order?
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)

enum { PLAIN = 0, NT = 1, SC1 = 2, SC01 = 3, NTSC01 = 4 };

template <int POL>
__device__ __forceinline__ void st8(double* p, double v) {
  if (POL == PLAIN) *p = v;
  else if (POL == NT) __builtin_nontemporal_store(v, p);
  else if (POL == SC1) asm volatile("global_store_dwordx2 %0, %1, off sc1" ::"v"(p), "v"(v) : "memory");
  else if (POL == SC01) asm volatile("global_store_dwordx2 %0, %1, off sc0 sc1" ::"v"(p), "v"(v) : "memory");
  else asm volatile("global_store_dwordx2 %0, %1, off sc0 sc1 nt" ::"v"(p), "v"(v) : "memory");
}
typedef double d2v __attribute__((ext_vector_type(2)));
template <int POL>
__device__ __forceinline__ void st16(double2* p, double2 v) {
  d2v w = {v.x, v.y};
  if (POL == PLAIN) *(d2v*)p = w;
  else if (POL == NT) __builtin_nontemporal_store(w, (d2v*)p);
  else if (POL == SC1) asm volatile("global_store_dwordx4 %0, %1, off sc1" ::"v"(p), "v"(w) : "memory");
  else if (POL == SC01) asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1" ::"v"(p), "v"(w) : "memory");
  else asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1 nt" ::"v"(p), "v"(w) : "memory");
}

// A: block = (512 columns, one layer), loops over g planes (the tau kernel's pattern), 8 B per lane
template <int POL>
__global__ void __launch_bounds__(512) w_tile_layer(double* __restrict__ out, int ncol, int nlay, int ng) {
  const unsigned col = blockIdx.x * 512 + threadIdx.x, lay = blockIdx.y;
  if (col >= (unsigned)ncol) return;
  const size_t ncl = (size_t)ncol * nlay;
  double v = col * 1e-9 + lay;
  for (int g = 0; g < ng; ++g) st8<POL>(out + col + (size_t)ncol * lay + ncl * g, v + g);
}
// A2: same planes, 16 B per lane: a wave covers 128 consecutive columns, block = 256 threads = 512 columns
template <int POL>
__global__ void __launch_bounds__(256) w_tile_layer16(double* __restrict__ out, int ncol, int nlay, int ng) {
  const unsigned col = (blockIdx.x * 256 + threadIdx.x) * 2, lay = blockIdx.y;
  if (col + 1 >= (unsigned)ncol) return;
  const size_t ncl = (size_t)ncol * nlay;
  double v = col * 1e-9 + lay;
  for (int g = 0; g < ng; ++g) st16<POL>((double2*)(out + col + (size_t)ncol * lay + ncl * g), make_double2(v + g, v - g));
}
// B: block = (512 columns, 16 g planes), loops over layers (the Planck kernel's pattern)
template <int POL>
__global__ void __launch_bounds__(512) w_tile_band(double* __restrict__ out, int ncol, int nlay, int ng) {
  const unsigned col = blockIdx.x * 512 + threadIdx.x, g0 = blockIdx.y * 16;
  if (col >= (unsigned)ncol) return;
  const size_t ncl = (size_t)ncol * nlay;
  double v = col * 1e-9 + g0;
  for (int lay = 0; lay < nlay; ++lay)
#pragma unroll
    for (int j = 0; j < 16; ++j) st8<POL>(out + col + (size_t)ncol * lay + ncl * (g0 + j), v + lay + j);
}
// B2: 16 B per lane, block = 256 threads = 512 columns, 16 g planes, loops over layers
template <int POL>
__global__ void __launch_bounds__(256) w_tile_band16(double* __restrict__ out, int ncol, int nlay, int ng) {
  const unsigned col = (blockIdx.x * 256 + threadIdx.x) * 2, g0 = blockIdx.y * 16;
  if (col + 1 >= (unsigned)ncol) return;
  const size_t ncl = (size_t)ncol * nlay;
  double v = col * 1e-9 + g0;
  for (int lay = 0; lay < nlay; ++lay)
#pragma unroll
    for (int j = 0; j < 16; ++j)
      st16<POL>((double2*)(out + col + (size_t)ncol * lay + ncl * (g0 + j)), make_double2(v + lay + j, v - j));
}
// C / D: linear streaming, 8 / 16 B per lane
template <int POL>
__global__ void __launch_bounds__(256) w_linear(double* __restrict__ out, size_t n) {
  for (size_t i = blockIdx.x * 256ull + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) st8<POL>(out + i, (double)i);
}
template <int POL>
__global__ void __launch_bounds__(256) w_linear16(double2* __restrict__ out, size_t n2) {
  for (size_t i = blockIdx.x * 256ull + threadIdx.x; i < n2; i += (size_t)gridDim.x * 256)
    st16<POL>(out + i, make_double2((double)i, 1.0));
}
// E: linear, each block owns one contiguous chunk (what a memset kernel typically does), 16 B per lane
template <int POL>
__global__ void __launch_bounds__(256) w_chunk16(double2* __restrict__ out, size_t n2, size_t per_block) {
  const size_t b = (size_t)blockIdx.x * per_block, e = b + per_block < n2 ? b + per_block : n2;
  for (size_t i = b + threadIdx.x; i < e; i += 256) st16<POL>(out + i, make_double2((double)i, 1.0));
}

// reads: linear 16 B per lane, plain vs nt
template <int POL>
__global__ void __launch_bounds__(256) r_linear16(const double2* __restrict__ in, size_t n2, double* __restrict__ sink) {
  double acc = 0;
  for (size_t i = blockIdx.x * 256ull + threadIdx.x; i < n2; i += (size_t)gridDim.x * 256) {
    d2v v = POL == NT ? __builtin_nontemporal_load((const d2v*)(in + i)) : *(const d2v*)(in + i);
    acc += v.x + v.y;
  }
  if (acc == -1.2345) sink[0] = acc;
}
template <int POL>
__global__ void __launch_bounds__(256) r_linear8(const double* __restrict__ in, size_t n, double* __restrict__ sink) {
  double acc = 0;
  for (size_t i = blockIdx.x * 256ull + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
    double v = POL == NT ? __builtin_nontemporal_load(in + i) : in[i];
    acc += v;
  }
  if (acc == -1.2345) sink[0] = acc;
}

int main() {
  const int ncol = 100000, nlay = 60, ng = 256;
  const size_t n = (size_t)ncol * nlay * ng;
  double* out;
  CK(hipMalloc(&out, n * 8));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  auto timeit = [&](const char* name, auto launch) {
    launch(); CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    for (int r = 0; r < 3; ++r) launch();
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= 3;
    printf("%-64s %7.3f ms  %7.1f GB/s\n", name, ms, n * 8 / (ms * 1e-3) / 1e9);
  };
  const dim3 gA((ncol + 511) / 512, nlay), gB((ncol + 511) / 512, ng / 16);
#define POLS(M) M(PLAIN) M(NT) M(SC1) M(SC01) M(NTSC01)
#define RUN_A(P) timeit("A  (512 col,1 lay) loops g, 8B  " #P, [&] { hipLaunchKernelGGL(w_tile_layer<P>, gA, dim3(512), 0, 0, out, ncol, nlay, ng); });
#define RUN_A2(P) timeit("A2 (512 col,1 lay) loops g, 16B " #P, [&] { hipLaunchKernelGGL(w_tile_layer16<P>, gA, dim3(256), 0, 0, out, ncol, nlay, ng); });
#define RUN_B(P) timeit("B  (512 col,16 g) loops lay, 8B  " #P, [&] { hipLaunchKernelGGL(w_tile_band<P>, gB, dim3(512), 0, 0, out, ncol, nlay, ng); });
#define RUN_B2(P) timeit("B2 (512 col,16 g) loops lay, 16B " #P, [&] { hipLaunchKernelGGL(w_tile_band16<P>, gB, dim3(256), 0, 0, out, ncol, nlay, ng); });
#define RUN_C(P) timeit("C  linear grid-stride 8B  " #P, [&] { hipLaunchKernelGGL(w_linear<P>, dim3(4096), dim3(256), 0, 0, out, n); });
#define RUN_D(P) timeit("D  linear grid-stride 16B " #P, [&] { hipLaunchKernelGGL(w_linear16<P>, dim3(4096), dim3(256), 0, 0, (double2*)out, n / 2); });
#define RUN_E(P) timeit("E  linear chunk/block 16B " #P, [&] { hipLaunchKernelGGL(w_chunk16<P>, dim3(8192), dim3(256), 0, 0, (double2*)out, n / 2, (n / 2 + 8191) / 8192); });
  POLS(RUN_A) POLS(RUN_A2) POLS(RUN_B) POLS(RUN_B2) POLS(RUN_C) POLS(RUN_D) POLS(RUN_E)
  timeit("hipMemsetAsync", [&] { CK(hipMemsetAsync(out, 0, n * 8, 0)); });
  double* sink; CK(hipMalloc(&sink, 8));
  timeit("R8  linear read 8 B/lane plain", [&] { hipLaunchKernelGGL(r_linear8<PLAIN>, dim3(8192), dim3(256), 0, 0, out, n, sink); });
  timeit("R8  linear read 8 B/lane nt", [&] { hipLaunchKernelGGL(r_linear8<NT>, dim3(8192), dim3(256), 0, 0, out, n, sink); });
  timeit("R16 linear read 16 B/lane plain", [&] { hipLaunchKernelGGL(r_linear16<PLAIN>, dim3(8192), dim3(256), 0, 0, (double2*)out, n / 2, sink); });
  timeit("R16 linear read 16 B/lane nt", [&] { hipLaunchKernelGGL(r_linear16<NT>, dim3(8192), dim3(256), 0, 0, (double2*)out, n / 2, sink); });
  return 0;
}
