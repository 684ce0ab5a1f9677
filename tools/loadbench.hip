// Load-side counterpart of storebench2.hip: sequential streams vs the LW solver's pattern (a 64-column tile reads
// 512-byte pieces of 60 layer rows of three arrays per g-point) and wider tiles.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

__global__ void __launch_bounds__(512) k_chunk(const double* __restrict__ p, size_t chunk, double* __restrict__ out) {
  const double* q = p + blockIdx.x * chunk;
  double s = 0;
  for (size_t i = threadIdx.x; i < chunk; i += 512) s += q[i];
  if (s == 12345.678) out[0] = s;
}
__global__ void __launch_bounds__(512) k_range(const double* __restrict__ p, size_t per_block, double* __restrict__ out) {
  const double* q = p + blockIdx.x * per_block;
  double s = 0;
#pragma unroll 8
  for (size_t i = threadIdx.x; i < per_block; i += 512) s += q[i];
  if (s == 12345.678) out[0] = s;
}
// solver-like: block = (tile of W*64 columns, group of g-points); 8 waves; wave w reads layers [8w, 8w+8) of 3 arrays,
// one g-point ahead (the values are summed); lanes hold W columns each (W*8-byte loads)
template <int W>
__global__ void __launch_bounds__(512) k_solver(const double* __restrict__ base, int ncol, int nlay, int ngpt, int gpb, double* __restrict__ out) {
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const size_t plane = (size_t)ncol * nlay;
  const size_t c0 = ((size_t)blockIdx.x * 64 + lane) * W;
  const int g0 = blockIdx.y * gpb;
  double s = 0;
  for (int g = g0; g < g0 + gpb && g < ngpt; ++g) {
#pragma unroll
    for (int a = 0; a < 3; ++a)
#pragma unroll
      for (int l = 0; l < 8; ++l) {
        const int lay = min(w * 8 + l, nlay - 1);
        const double* q = base + ((size_t)a * ngpt + g) * plane + (size_t)lay * ncol + c0;
#pragma unroll
        for (int k = 0; k < W; ++k) s += q[k];
      }
  }
  if (s == 12345.678) out[0] = s;
}

int main() {
  const int ncol = 100352, nlay = 60, ngpt = 170;  // 3 arrays x 170 planes x 48 MB = 24.6 GB
  const size_t bytes = (size_t)3 * ngpt * ncol * nlay * 8;
  char* buf; CK(hipMalloc(&buf, bytes)); CK(hipMemset(buf, 0, bytes));
  double* out; CK(hipMalloc(&out, 8));
  hipStream_t st; CK(hipStreamCreate(&st));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  auto timeit = [&](const char* name, auto&& f, double gb) {
    f(); CK(hipStreamSynchronize(st));
    float best = 1e9;
    for (int r = 0; r < 3; ++r) {
      CK(hipEventRecord(e0, st)); f(); CK(hipEventRecord(e1, st)); CK(hipEventSynchronize(e1));
      float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (ms < best) best = ms;
    }
    printf("%-64s %7.3f ms  %6.0f GB/s\n", name, best, gb / (best * 1e-3));
  };
  const double GB = bytes / 1e9;
  char nm[128];
  for (size_t ch : {size_t(4096), size_t(32768)}) {
    snprintf(nm, 128, "chunk per block %zu KB", ch / 1024);
    timeit(nm, [&] { hipLaunchKernelGGL(k_chunk, dim3((unsigned)(bytes / ch)), dim3(512), 0, st, (const double*)buf, ch / 8, out); }, GB);
  }
  for (int g : {256, 512, 1024, 2048}) {
    snprintf(nm, 128, "persistent contiguous range per block, %d blocks", g);
    timeit(nm, [&] { hipLaunchKernelGGL(k_range, dim3(g), dim3(512), 0, st, (const double*)buf, bytes / 8 / g, out); }, GB);
  }
  for (int groups : {1, 2, 4, 8}) {
    const int gpb = (ngpt + groups - 1) / groups;
    snprintf(nm, 128, "solver-like, 64-column tiles (512 B pieces), %d g-groups", groups);
    timeit(nm, [&] { hipLaunchKernelGGL((k_solver<1>), dim3(ncol / 64, groups), dim3(512), 0, st, (const double*)buf, ncol, nlay, ngpt, gpb, out); }, GB);
    snprintf(nm, 128, "solver-like, 128-column tiles (1 KB pieces), %d g-groups", groups);
    timeit(nm, [&] { hipLaunchKernelGGL((k_solver<2>), dim3(ncol / 128, groups), dim3(512), 0, st, (const double*)buf, ncol, nlay, ngpt, gpb, out); }, GB);
  }
  return 0;
}
