// Which store recipe reaches what hipMemsetAsync reaches (6.1-6.4 TB/s on MI355X) and which stays at 5.4-5.5?
// hipcc --offload-arch=gfx950 -O3 tools/storebench.hip -o tools/_storebench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

// grid-stride, W bytes per thread and iteration
template <typename T>
__global__ void __launch_bounds__(256) k_stride(T* __restrict__ p, size_t n, T v) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] = v;
}
// one chunk of `chunk` elements per block
template <typename T>
__global__ void __launch_bounds__(256) k_chunk(T* __restrict__ p, size_t chunk, T v) {
  T* q = p + blockIdx.x * chunk;
  for (size_t i = threadIdx.x; i < chunk; i += blockDim.x) q[i] = v;
}
// Planck-like: block = (512-column tile, band of 16 g-points), walks nlay layers, writes NA arrays' planes
// order 0: id -> (tile, band) with band fastest (the 16 bands of a tile run together); 1: tile fastest (one band chip-wide)
template <typename T, int NA>
__global__ void __launch_bounds__(512) k_planes(double* __restrict__ base, int ncol, int nlay, int nbnd, int ntiles, int order, double v) {
  const int id = blockIdx.x;
  const int tile = order == 0 ? id / nbnd : id % ntiles, band = order == 0 ? id % nbnd : id / ntiles;
  const size_t plane = (size_t)ncol * nlay;
  constexpr int PER = sizeof(T) / 8;
  for (int l = 0; l < nlay; ++l)
    for (int a = 0; a < NA; ++a)
#pragma unroll 4
      for (int g = 0; g < 16; ++g) {
        double* row = base + ((size_t)a * nbnd * 16 + band * 16 + g) * plane + (size_t)l * ncol + (size_t)tile * 512;
        if (PER == 1) row[threadIdx.x] = v;
        else if (threadIdx.x < 256) { T t; double* tt = (double*)&t; tt[0] = v; tt[1] = v; reinterpret_cast<T*>(row)[threadIdx.x] = t; }
      }
}

int main() {
  const size_t bytes = size_t(24) << 30;  // 24 GiB: well beyond the 256 MB Infinity Cache
  char* buf; CK(hipMalloc(&buf, bytes));
  hipStream_t st; CK(hipStreamCreate(&st));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  auto timeit = [&](const char* name, auto&& f, double gb) {
    f(); CK(hipStreamSynchronize(st));
    float best = 1e9;
    for (int r = 0; r < 3; ++r) {
      CK(hipEventRecord(e0, st)); f(); CK(hipEventRecord(e1, st)); CK(hipEventSynchronize(e1));
      float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (ms < best) best = ms;
    }
    printf("%-58s %7.3f ms  %6.0f GB/s\n", name, best, gb / (best * 1e-3));
  };
  const double GB = bytes / 1e9;
  timeit("hipMemsetAsync", [&] { CK(hipMemsetAsync(buf, 0, bytes, st)); }, GB);
  for (int wg : {1024, 2048, 4096, 8192, 65536}) {
    char nm[96];
    snprintf(nm, 96, "grid-stride 16 B/thread, %d blocks", wg);
    timeit(nm, [&] { hipLaunchKernelGGL((k_stride<double2>), dim3(wg), dim3(256), 0, st, (double2*)buf, bytes / 16, double2{1, 1}); }, GB);
    snprintf(nm, 96, "grid-stride  8 B/thread, %d blocks", wg);
    timeit(nm, [&] { hipLaunchKernelGGL((k_stride<double>), dim3(wg), dim3(256), 0, st, (double*)buf, bytes / 8, 1.0); }, GB);
  }
  for (size_t ch : {size_t(4096), size_t(65536), size_t(1) << 20}) {
    char nm[96];
    snprintf(nm, 96, "chunk per block %zu KB, 8 B/thread", ch / 1024);
    timeit(nm, [&] { hipLaunchKernelGGL((k_chunk<double>), dim3((unsigned)(bytes / ch)), dim3(256), 0, st, (double*)buf, ch / 8, 1.0); }, GB);
    snprintf(nm, 96, "chunk per block %zu KB, 16 B/thread", ch / 1024);
    timeit(nm, [&] { hipLaunchKernelGGL((k_chunk<double2>), dim3((unsigned)(bytes / ch)), dim3(256), 0, st, (double2*)buf, ch / 16, double2{1, 1}); }, GB);
  }
  {
    const int ncol = 100352, nlay = 60, nbnd = 16, ntiles = ncol / 512;  // 2 arrays x 256 planes x 48 MB = 24.6 GB
    const double gb = 2.0 * 256 * (double)ncol * nlay * 8 / 1e9;
    if ((size_t)(gb * 1e9) <= bytes) {
      for (int order = 0; order < 2; ++order) {
        char nm[96];
        snprintf(nm, 96, "Planck-like planes, 8 B/thread, %s", order ? "one band chip-wide" : "bands of a tile together");
        timeit(nm, [&] { hipLaunchKernelGGL((k_planes<double, 2>), dim3(ntiles * nbnd), dim3(512), 0, st, (double*)buf, ncol, nlay, nbnd, ntiles, order, 1.0); }, gb);
        snprintf(nm, 96, "Planck-like planes, 16 B/thread, %s", order ? "one band chip-wide" : "bands of a tile together");
        timeit(nm, [&] { hipLaunchKernelGGL((k_planes<double2, 2>), dim3(ntiles * nbnd), dim3(512), 0, st, (double*)buf, ncol, nlay, nbnd, ntiles, order, 1.0); }, gb);
      }
    }
  }
  return 0;
}
