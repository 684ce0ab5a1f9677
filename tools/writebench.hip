// writebench.hip -- MI355X store-pattern micro-benchmark (hipcc --offload-arch=gfx950 -O3 tools/writebench.hip).
// How fast can the (col, lay, g) output planes of the gas-optics kernels be written?  All variants write the
// same ncol*nlay*ng doubles; they differ in which block writes what, when.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)

// A: block = (TILE columns, one layer), loops over g planes (the tau kernel's pattern)
template <int TILE>
__global__ void __launch_bounds__(TILE) w_tile_layer(double* __restrict__ out, int ncol, int nlay, int ng) {
  const unsigned col = blockIdx.x * TILE + threadIdx.x, lay = blockIdx.y;
  if (col >= (unsigned)ncol) return;
  const size_t ncl = (size_t)ncol * nlay;
  double v = col * 1e-9 + lay;
  for (int g = 0; g < ng; ++g) out[col + (size_t)ncol * lay + ncl * g] = v + g;
}
// B: block = (TILE columns, 16 g planes), loops over layers (the Planck kernel's pattern)
template <int TILE>
__global__ void __launch_bounds__(TILE) w_tile_band(double* __restrict__ out, int ncol, int nlay, int ng) {
  const unsigned col = blockIdx.x * TILE + threadIdx.x, g0 = blockIdx.y * 16;
  if (col >= (unsigned)ncol) return;
  const size_t ncl = (size_t)ncol * nlay;
  double v = col * 1e-9 + g0;
  for (int lay = 0; lay < nlay; ++lay)
#pragma unroll
    for (int j = 0; j < 16; ++j) out[col + (size_t)ncol * lay + ncl * (g0 + j)] = v + lay + j;
}
// C: linear streaming (grid-stride, 8 B per lane)
__global__ void __launch_bounds__(256) w_linear(double* __restrict__ out, size_t n) {
  for (size_t i = blockIdx.x * 256ull + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) out[i] = (double)i;
}
// D: linear streaming, 16 B per lane
__global__ void __launch_bounds__(256) w_linear16(double2* __restrict__ out, size_t n2) {
  for (size_t i = blockIdx.x * 256ull + threadIdx.x; i < n2; i += (size_t)gridDim.x * 256) out[i] = make_double2((double)i, 1.0);
}

// R1: linear streaming read (grid-stride, 8 B per lane), sum kept in a register
__global__ void __launch_bounds__(256) r_linear(const double* __restrict__ in, size_t n, double* __restrict__ sink) {
  double acc = 0;
  for (size_t i = blockIdx.x * 256ull + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) acc += in[i];
  if (acc == -1.2345) sink[0] = acc;
}
// R2: the LW solver's pattern: block = (64 columns, group of g planes) x 8 waves, wave s reads layers [8s, 8s+8)
__global__ void __launch_bounds__(512) r_solver(const double* __restrict__ in, int ncol, int nlay, int ng, int gpb,
                                                double* __restrict__ sink) {
  const int lane = threadIdx.x & 63, s = threadIdx.x >> 6;
  const unsigned col = min(blockIdx.x * 64 + lane, (unsigned)ncol - 1);
  const size_t ncl = (size_t)ncol * nlay;
  double acc = 0;
  for (int g = blockIdx.y * gpb; g < min(ng, (int)(blockIdx.y + 1) * gpb); ++g) {
    const double* p = in + col + ncl * g;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int lay = min(8 * s + i, nlay - 1);
      acc += p[(size_t)ncol * lay];
    }
  }
  if (acc == -1.2345) sink[0] = acc;
}

// R3: three arrays in the solver's pattern: tau and lay (ncol, nlay, ng), lev (ncol, nlay+1, ng); wave s reads layers
// [8s, 8s+8) of tau and lay and levels [8s, 8s+9) of lev, PF g-points requested before the previous ones are consumed
template <int PF>
__global__ void __launch_bounds__(512) r_solver3(const double* __restrict__ tau, const double* __restrict__ lay,
                                                 const double* __restrict__ lev, int ncol, int nlay, int ng, int gpb,
                                                 double* __restrict__ sink) {
  const int lane = threadIdx.x & 63, s = threadIdx.x >> 6;
  const unsigned col = min(blockIdx.x * 64 + lane, (unsigned)ncol - 1);
  const size_t ncl = (size_t)ncol * nlay, nclv = (size_t)ncol * (nlay + 1);
  double acc = 0;
  const int gb = blockIdx.y * gpb, ge = min(ng, (int)(blockIdx.y + 1) * gpb);
  double buf[PF][25];
  auto load = [&](double (&b)[25], int g) {
    g = min(g, ge - 1);
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int l = min(8 * s + i, nlay - 1);
      b[i] = tau[col + (size_t)ncol * l + ncl * g];
      b[8 + i] = lay[col + (size_t)ncol * l + ncl * g];
    }
#pragma unroll
    for (int i = 0; i < 9; ++i) b[16 + i] = lev[col + (size_t)ncol * min(8 * s + i, nlay) + nclv * g];
  };
#pragma unroll
  for (int p = 0; p < PF; ++p) load(buf[p], gb + p);
  for (int g = gb; g < ge; g += PF) {
#pragma unroll
    for (int p = 0; p < PF; ++p) {
      double t = 0;
#pragma unroll
      for (int i = 0; i < 25; ++i) t += buf[p][i];
      acc += t;
      load(buf[p], g + PF + p);
    }
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
    printf("%-60s %7.3f ms  %7.1f GB/s\n", name, ms, n * 8 / (ms * 1e-3) / 1e9);
  };
  timeit("A256: block=(256 col, 1 lay) loops 256 g", [&] { hipLaunchKernelGGL(w_tile_layer<256>, dim3((ncol + 255) / 256, nlay), dim3(256), 0, 0, out, ncol, nlay, ng); });
  timeit("A512: block=(512 col, 1 lay) loops 256 g", [&] { hipLaunchKernelGGL(w_tile_layer<512>, dim3((ncol + 511) / 512, nlay), dim3(512), 0, 0, out, ncol, nlay, ng); });
  timeit("A1024: block=(1024 col, 1 lay) loops 256 g", [&] { hipLaunchKernelGGL(w_tile_layer<1024>, dim3((ncol + 1023) / 1024, nlay), dim3(1024), 0, 0, out, ncol, nlay, ng); });
  timeit("B256: block=(256 col, 16 g) loops 60 lay", [&] { hipLaunchKernelGGL(w_tile_band<256>, dim3((ncol + 255) / 256, ng / 16), dim3(256), 0, 0, out, ncol, nlay, ng); });
  timeit("B512: block=(512 col, 16 g) loops 60 lay", [&] { hipLaunchKernelGGL(w_tile_band<512>, dim3((ncol + 511) / 512, ng / 16), dim3(512), 0, 0, out, ncol, nlay, ng); });
  timeit("B1024: block=(1024 col, 16 g) loops 60 lay", [&] { hipLaunchKernelGGL(w_tile_band<1024>, dim3((ncol + 1023) / 1024, ng / 16), dim3(1024), 0, 0, out, ncol, nlay, ng); });
  timeit("C: linear 8 B/lane", [&] { hipLaunchKernelGGL(w_linear, dim3(4096), dim3(256), 0, 0, out, n); });
  timeit("D: linear 16 B/lane", [&] { hipLaunchKernelGGL(w_linear16, dim3(4096), dim3(256), 0, 0, (double2*)out, n / 2); });
  timeit("hipMemsetAsync", [&] { CK(hipMemsetAsync(out, 0, n * 8, 0)); });
  double* sink; CK(hipMalloc(&sink, 8));
  timeit("R1: linear read 8 B/lane", [&] { hipLaunchKernelGGL(r_linear, dim3(8192), dim3(256), 0, 0, out, n, sink); });
  timeit("R2: solver pattern read (64 col x 8 waves x 8 lay, 16 g/block)", [&] { hipLaunchKernelGGL(r_solver, dim3((ncol + 63) / 64, ng / 16), dim3(512), 0, 0, out, ncol, nlay, ng, 16, sink); });
  {
    double *lay, *lev;
    CK(hipMalloc(&lay, n * 8)); CK(hipMalloc(&lev, (size_t)ncol * (nlay + 1) * ng * 8));
    CK(hipMemset(lay, 0, n * 8)); CK(hipMemset(lev, 0, (size_t)ncol * (nlay + 1) * ng * 8));
    const double nn = 3.0 * n + (double)ncol * ng;  // elements read
    auto t3 = [&](const char* name, auto launch) {
      launch(); CK(hipDeviceSynchronize());
      CK(hipEventRecord(e0));
      for (int r = 0; r < 3; ++r) launch();
      CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
      float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= 3;
      printf("%-60s %7.3f ms  %7.1f GB/s\n", name, ms, nn * 8 / (ms * 1e-3) / 1e9);
    };
    t3("R3/1: tau+lay+lev solver pattern, 1 g-point ahead", [&] { hipLaunchKernelGGL(r_solver3<1>, dim3((ncol + 63) / 64, ng / 16), dim3(512), 0, 0, out, lay, lev, ncol, nlay, ng, 16, sink); });
    t3("R3/2: tau+lay+lev solver pattern, 2 g-points ahead", [&] { hipLaunchKernelGGL(r_solver3<2>, dim3((ncol + 63) / 64, ng / 16), dim3(512), 0, 0, out, lay, lev, ncol, nlay, ng, 16, sink); });
  }
  return 0;
}
