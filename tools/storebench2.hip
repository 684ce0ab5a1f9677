// Follow-up to storebench.hip: what makes "one 4 KB chunk per block" (7.0 TB/s) faster than every looping recipe (5.4-5.9)?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

template <int NT>
__global__ void __launch_bounds__(NT) k_chunk(double* __restrict__ p, size_t chunk, double v) {
  double* q = p + blockIdx.x * chunk;
  for (size_t i = threadIdx.x; i < chunk; i += NT) q[i] = v;
}
// persistent, block-cyclic: iteration i of block b writes chunk i * gridDim + b (4 KB = 512 threads x 8 B); WAIT: drain after each store
template <bool WAIT>
__global__ void __launch_bounds__(512) k_cyclic(double* __restrict__ p, size_t nchunks, double v) {
  for (size_t c = blockIdx.x; c < nchunks; c += gridDim.x) {
    p[c * 512 + threadIdx.x] = v;
    if (WAIT) __builtin_amdgcn_s_waitcnt(0x0F70);
  }
}
// persistent, each block owns a contiguous range and walks it 4 KB at a time
__global__ void __launch_bounds__(512) k_range(double* __restrict__ p, size_t per_block, double v) {
  double* q = p + blockIdx.x * per_block;
  for (size_t i = threadIdx.x; i < per_block; i += 512) q[i] = v;
}
// P planes: block (t, l) writes row l of every plane for its 512-column tile and exits (short-lived, P stores per thread)
__global__ void __launch_bounds__(512) k_rows(double* __restrict__ base, int ncol, int nlay, int P, double v) {
  const int ntiles = ncol / 512;
  const int t = blockIdx.x % ntiles, l = blockIdx.x / ntiles;
  const size_t plane = (size_t)ncol * nlay;
  for (int g = 0; g < P; ++g) base[(size_t)g * plane + (size_t)l * ncol + (size_t)t * 512 + threadIdx.x] = v;
}
// the same bytes, plane-major: block (t, l, g) writes one 4 KB row piece; dispatch order = memory order within a plane
__global__ void __launch_bounds__(512) k_rows1(double* __restrict__ base, int ncol, double v) {
  base[(size_t)blockIdx.x * 512 + threadIdx.x] = v;
}

int main() {
  const size_t bytes = size_t(24) << 30;
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
    printf("%-64s %7.3f ms  %6.0f GB/s\n", name, best, gb / (best * 1e-3));
  };
  const double GB = bytes / 1e9;
  char nm[128];
  for (size_t ch : {size_t(2048), size_t(4096), size_t(8192), size_t(16384), size_t(32768)}) {
    snprintf(nm, 128, "chunk per block %zu KB, 256 threads", ch / 1024);
    timeit(nm, [&] { hipLaunchKernelGGL((k_chunk<256>), dim3((unsigned)(bytes / ch)), dim3(256), 0, st, (double*)buf, ch / 8, 1.0); }, GB);
  }
  snprintf(nm, 128, "chunk per block 4 KB, 512 threads (1 store each)");
  timeit(nm, [&] { hipLaunchKernelGGL((k_chunk<512>), dim3((unsigned)(bytes / 4096)), dim3(512), 0, st, (double*)buf, 512, 1.0); }, GB);
  snprintf(nm, 128, "chunk per block 4 KB, 64 threads (8 stores each)");
  timeit(nm, [&] { hipLaunchKernelGGL((k_chunk<64>), dim3((unsigned)(bytes / 4096)), dim3(64), 0, st, (double*)buf, 512, 1.0); }, GB);
  snprintf(nm, 128, "chunk per block 32 KB, 512 threads (8 stores each)");
  timeit(nm, [&] { hipLaunchKernelGGL((k_chunk<512>), dim3((unsigned)(bytes / 32768)), dim3(512), 0, st, (double*)buf, 4096, 1.0); }, GB);
  for (int g : {256, 512, 1024, 2048}) {
    snprintf(nm, 128, "persistent block-cyclic 4 KB, %d blocks", g);
    timeit(nm, [&] { hipLaunchKernelGGL((k_cyclic<false>), dim3(g), dim3(512), 0, st, (double*)buf, bytes / 4096, 1.0); }, GB);
    snprintf(nm, 128, "persistent block-cyclic 4 KB, %d blocks, drain after each store", g);
    timeit(nm, [&] { hipLaunchKernelGGL((k_cyclic<true>), dim3(g), dim3(512), 0, st, (double*)buf, bytes / 4096, 1.0); }, GB);
    snprintf(nm, 128, "persistent contiguous range per block, %d blocks", g);
    timeit(nm, [&] { hipLaunchKernelGGL(k_range, dim3(g), dim3(512), 0, st, (double*)buf, bytes / 8 / g, 1.0); }, GB);
  }
  {
    const int ncol = 100352, nlay = 60;
    for (int P : {16, 48, 256}) {
      const int reps = 512 / P;  // same total bytes: 512 planes
      const double gb = 512.0 * ncol * nlay * 8 / 1e9;
      snprintf(nm, 128, "short blocks (tile, layer): %d planes x 4 KB each", P);
      timeit(nm, [&] { for (int r = 0; r < reps; ++r) hipLaunchKernelGGL(k_rows, dim3(ncol / 512 * nlay), dim3(512), 0, st, (double*)buf + (size_t)r * P * ncol * nlay, ncol, nlay, P, 1.0); }, gb);
    }
    const double gb = 512.0 * ncol * nlay * 8 / 1e9;
    timeit("short blocks (plane, layer, tile): 4 KB each, memory order", [&] { hipLaunchKernelGGL(k_rows1, dim3((unsigned)(512ull * ncol * nlay / 512)), dim3(512), 0, st, (double*)buf, ncol, 1.0); }, gb);
  }
  return 0;
}
