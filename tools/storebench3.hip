// Round 4: which decomposition of compute_Planck_source's 25 GB of stores (16 + 16 planes per band, planes (column, layer) of
// 48 MB) reaches which store rate?  Pure-store kernels with the real kernels' non-temporal stores, 512-thread blocks:
//   A  block = (512-column tile, band) walking the layers (today's kernel: 32 pieces of 4 KB per step, 0.8 MB apart per step)
//   B  block = (band, layer) walking the column tiles (32 pieces per step, each stream sequential)
//   C  block = (band, layer, chunk of the columns)   -- B with more, shorter blocks
//   D  block = (tile, layer) walking the bands (the tau kernel's shape with 32 planes per band)
//   E  block = (tile, band, chunk of L layers)
// hipcc --offload-arch=gfx950 -O3 tools/storebench3.hip -o tools/storebench3
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
constexpr int NCOL = 100352, NLAY = 60, NBND = 16, GPB = 16, NT = NCOL / 512;
__device__ __forceinline__ void st(double* p, double v) { __builtin_nontemporal_store(v, p); }
// lay planes: [g][lay][col]; lev planes behind them: [g][lev][col] (61 levels)
__device__ __forceinline__ void put(double* base, int tile, int band, int lay, double v) {
  const size_t plane = (size_t)NCOL * NLAY, planev = (size_t)NCOL * (NLAY + 1);
  double* lev = base + plane * NBND * GPB;
  const size_t o = (size_t)lay * NCOL + (size_t)tile * 512 + threadIdx.x;
#pragma unroll
  for (int g = 0; g < GPB; ++g) {
    st(base + plane * (band * GPB + g) + o, v);
    st(lev + planev * (band * GPB + g) + o, v);
  }
}
__global__ void __launch_bounds__(512) kA(double* b, int xcd_order) {
  int id = blockIdx.x, tile, band;
  if (xcd_order) { const int x = id & 7, r = id >> 3; band = r % NBND; tile = (r / NBND) * 8 + x; if (tile >= NT) return; }
  else { tile = id % NT; band = id / NT; }
  for (int l = 0; l < NLAY; ++l) put(b, tile, band, l, 1.0);
}
__global__ void __launch_bounds__(512) kB(double* b, int lay_fastest) {
  const int band = lay_fastest ? blockIdx.x / NLAY : blockIdx.x % NBND, lay = lay_fastest ? blockIdx.x % NLAY : blockIdx.x / NBND;
  for (int t = 0; t < NT; ++t) put(b, t, band, lay, 1.0);
}
__global__ void __launch_bounds__(512) kC(double* b, int chunk) {  // grid = NBND * NLAY * nchunks, chunk index fastest
  const int nch = (NT + chunk - 1) / chunk;
  const int c = blockIdx.x % nch, r = blockIdx.x / nch, lay = r % NLAY, band = r / NLAY;
  for (int t = c * chunk; t < min(NT, (c + 1) * chunk); ++t) put(b, t, band, lay, 1.0);
}
__global__ void __launch_bounds__(512) kD(double* b) {
  const int tile = blockIdx.x % NT, lay = blockIdx.x / NT;
  for (int band = 0; band < NBND; ++band) put(b, tile, band, lay, 1.0);
}
__global__ void __launch_bounds__(512) kE(double* b, int L) {  // grid: tile fastest, then band, then layer chunk
  const int nch = NLAY / L;
  const int tile = blockIdx.x % NT, r = blockIdx.x / NT, band = r % NBND, c = r / NBND;
  if (c >= nch) return;
  for (int l = c * L; l < (c + 1) * L; ++l) put(b, tile, band, l, 1.0);
}
// A with 16 bytes per lane: the same 512 threads, lane pairs store two columns of one plane (even lanes the even g-points, odd lanes
// the odd ones): half as many store instructions for the same bytes, pieces and planes
__global__ void __launch_bounds__(512) kA16(double* b) {
  const int id = blockIdx.x, tile = id % NT, band = id / NT;
  const size_t plane = (size_t)NCOL * NLAY, planev = (size_t)NCOL * (NLAY + 1);
  double* lev = b + plane * NBND * GPB;
  const int par = threadIdx.x & 1, c0 = threadIdx.x & ~1;
  for (int l = 0; l < NLAY; ++l) {
    const size_t o = (size_t)l * NCOL + (size_t)tile * 512 + c0;
#pragma unroll
    for (int g = 0; g < GPB; g += 2) {
      typedef double d2 __attribute__((ext_vector_type(2)));
      const d2 v = {1.0, 1.0};
      __builtin_nontemporal_store(v, reinterpret_cast<d2*>(b + plane * (band * GPB + g + par) + o));
      __builtin_nontemporal_store(v, reinterpret_cast<d2*>(lev + planev * (band * GPB + g + par) + o));
    }
  }
}
int main() {
  const size_t n = (size_t)NCOL * (2 * NLAY + 1) * NBND * GPB;
  double* buf; CK(hipMalloc(&buf, n * 8));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const double gb = n * 8 / 1e9 * (2.0 * NLAY) / (2 * NLAY + 1);
  auto timeit = [&](const char* name, auto&& f) {
    f(); CK(hipDeviceSynchronize());
    float best = 1e9;
    for (int r = 0; r < 3; ++r) { CK(hipEventRecord(e0)); f(); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (ms < best) best = ms; }
    printf("%-84s %7.3f ms  %6.0f GB/s\n", name, best, gb / (best * 1e-3));
  };
  timeit("A  (tile, band) walking layers, tile fastest", [&] { hipLaunchKernelGGL(kA, dim3(NT * NBND), dim3(512), 0, 0, buf, 0); });
  timeit("A16 as A with 16-byte stores of lane pairs", [&] { hipLaunchKernelGGL(kA16, dim3(NT * NBND), dim3(512), 0, 0, buf); });
  timeit("A' (tile, band) walking layers, XCD-aware order (bands of a tile on one XCD)", [&] { hipLaunchKernelGGL(kA, dim3(((NT + 7) / 8) * 8 * NBND), dim3(512), 0, 0, buf, 1); });
  timeit("B  (band, layer) walking the column tiles, band fastest", [&] { hipLaunchKernelGGL(kB, dim3(NBND * NLAY), dim3(512), 0, 0, buf, 0); });
  timeit("B' (band, layer) walking the column tiles, layer fastest", [&] { hipLaunchKernelGGL(kB, dim3(NBND * NLAY), dim3(512), 0, 0, buf, 1); });
  for (int ch : {49, 14, 7, 2, 1}) { char nm[128]; snprintf(nm, 128, "C  (band, layer, chunk of %d tiles), chunk fastest", ch);
    timeit(nm, [&] { hipLaunchKernelGGL(kC, dim3(NBND * NLAY * ((NT + ch - 1) / ch)), dim3(512), 0, 0, buf, ch); }); }
  timeit("D  (tile, layer) walking the bands", [&] { hipLaunchKernelGGL(kD, dim3(NT * NLAY), dim3(512), 0, 0, buf); });
  for (int L : {4, 10, 20}) { char nm[128]; snprintf(nm, 128, "E  (tile, band, %d layers), tile fastest", L);
    timeit(nm, [&] { hipLaunchKernelGGL(kE, dim3(NT * NBND * (NLAY / L)), dim3(512), 0, 0, buf, L); }); }
  return 0;
}
