// membench.hip -- micro-benchmarks of the vector-L1 (TCP) and LDS read paths on gfx950 for the access
// patterns the gas-optics kernels can choose between.  Not part of the product; build+run:
//   hipcc --offload-arch=gfx950 -O3 tools/membench.hip -o /tmp/membench && /tmp/membench
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)

struct alignas(16) D2 { double x, y; };
constexpr int ITERS = 1024, ROWB = 128;
// cheap per-thread pseudo-random row sequence over 256 hot rows (one mad + and per step)
#define NEXT(st) ((st) = ((st) * 5u + 1u) & 255u)  // row = 128 bytes = 16 doubles

// mode 0: 8-B gather, every lane its own random row (col-lane mapping), element j of the row
// mode 1: 16-B loads, lane l reads piece (l&7) of row idx[l>>3] (8 lanes share a row)
// mode 2: 16-B loads, all lanes same row (broadcast)
// mode 3: 16-B loads, every lane its own random row, piece it%8
// mode 4: 16-B fully coalesced stream
template <int MODE>
__global__ void __launch_bounds__(256) gl_kernel(const double* __restrict__ tab, const int* __restrict__ idx, int nrows,
                                                 double* __restrict__ out) {
  const int lane = threadIdx.x & 63;
  const int gw = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;  // global wave id
  double acc = 0;
  unsigned sl = (gw * 64 + lane) * 7u & 255u, sg = (gw * 8 + (lane >> 3)) * 7u & 255u, sw = gw * 7u & 255u;
#pragma unroll 8
  for (int it = 0; it < ITERS; ++it) {
    const int base = (gw * ITERS + it) * 64;
    if (MODE == 0) {
      const int r = NEXT(sl) * 6;
      acc += tab[(size_t)r * 16 + (it & 15)];
    } else if (MODE == 1) {
      const int r = NEXT(sg) * 6;
      const D2 v = *reinterpret_cast<const D2*>(tab + (size_t)r * 16 + 2 * (lane & 7));
      acc += v.x + v.y;
    } else if (MODE == 2) {
      const int r = NEXT(sw) * 6;
      const D2 v = *reinterpret_cast<const D2*>(tab + (size_t)r * 16 + 2 * (it & 7));
      acc += v.x + v.y;
    } else if (MODE == 3) {
      const int r = NEXT(sl) * 6;
      const D2 v = *reinterpret_cast<const D2*>(tab + (size_t)r * 16 + 2 * (it & 7));
      acc += v.x + v.y;
    } else {
      const size_t o = ((size_t)(gw * ITERS + it) * 64 + lane) % ((size_t)nrows * 8);
      const D2 v = *reinterpret_cast<const D2*>(tab + o * 2);
      acc += v.x + v.y;
    }
  }
  if (acc == 123.456) out[0] = acc;
}

// LDS: slab of R rows (padded stride), staged once, then ITERS reads per lane
// mode 0: ds_read_b128, 8 lanes per row (lane&7 = piece), rows random per 8-lane group
// mode 1: ds_read_b128, every lane its own random row
// mode 2: ds_read_b64, every lane its own random row
// mode 3: ds_read_b128 broadcast (all lanes same row, same piece)
template <int MODE, int STRIDE>
__global__ void __launch_bounds__(256) lds_kernel(const double* __restrict__ tab, const int* __restrict__ idx, int R,
                                                  double* __restrict__ out) {
  extern __shared__ double slab[];
  for (int i = threadIdx.x; i < R * 16; i += 256) slab[(i >> 4) * STRIDE + (i & 15)] = tab[i];
  __syncthreads();
  const int lane = threadIdx.x & 63;
  const int gw = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  double acc = 0;
  unsigned sl = (gw * 64 + lane) * 7u & 255u, sg = (gw * 8 + (lane >> 3)) * 7u & 255u, sw = gw * 7u & 255u;
#pragma unroll 8
  for (int it = 0; it < ITERS; ++it) {
    if (MODE == 0) {
      const int r = NEXT(sg);
      const D2 v = *reinterpret_cast<const D2*>(slab + r * STRIDE + 2 * (lane & 7));
      acc += v.x + v.y;
    } else if (MODE == 1) {
      const int r = NEXT(sl);
      const D2 v = *reinterpret_cast<const D2*>(slab + r * STRIDE + 2 * (it & 7));
      acc += v.x + v.y;
    } else if (MODE == 2) {
      const int r = NEXT(sl);
      acc += slab[r * STRIDE + (it & 15)];
    } else {
      const int r = NEXT(sw);
      const D2 v = *reinterpret_cast<const D2*>(slab + r * STRIDE + 2 * (it & 7));
      acc += v.x + v.y;
    }
  }
  if (acc == 123.456) out[0] = acc;
}

template <class F> float timeit(F f) {
  hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  f(); CK(hipDeviceSynchronize());
  CK(hipEventRecord(a)); for (int i = 0; i < 5; ++i) f(); CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
  float ms; CK(hipEventElapsedTime(&ms, a, b)); return ms / 5;
}

int main() {
  const int nrows = 2000;  // 256 KB table: L2-resident, larger than L1 (32 KB)
  std::vector<double> h(nrows * 16, 1.0);
  std::vector<int> hi(1 << 20);
  srand(1);
  // indices clustered like the application: ~300 distinct hot rows
  for (auto& v : hi) v = (rand() % 300) * 6 % nrows;
  double *tab, *out; int* idx;
  CK(hipMalloc(&tab, h.size() * 8)); CK(hipMalloc(&out, 8)); CK(hipMalloc(&idx, hi.size() * 4));
  CK(hipMemcpy(tab, h.data(), h.size() * 8, hipMemcpyHostToDevice));
  CK(hipMemcpy(idx, hi.data(), hi.size() * 4, hipMemcpyHostToDevice));
  const int blocks = 256 * 16;  // 16 blocks of 4 waves per CU
  const double waves = blocks * 4.0, clk = 2.4e9, cus = 256;
  auto rep = [&](const char* name, float ms, double bytes_per_lane_iter) {
    const double bytes = waves * 64 * ITERS * bytes_per_lane_iter;
    printf("%-58s %8.3f ms  %7.1f B/clk/CU  (%5.2f wave-instr/clk/CU... %6.1f clk per wave-instr per CU)\n", name, ms,
           bytes / (ms * 1e-3) / clk / cus, waves * ITERS / (ms * 1e-3) / clk / cus, (ms * 1e-3) * clk * cus / (waves * ITERS));
  };
  rep("global 8B gather, lane-own hot row (native direct kernel)", timeit([&] { hipLaunchKernelGGL(gl_kernel<0>, dim3(blocks), dim3(256), 0, 0, tab, idx, nrows, out); }), 8);
  rep("global 16B, 8 lanes share a 128B row (v5)", timeit([&] { hipLaunchKernelGGL(gl_kernel<1>, dim3(blocks), dim3(256), 0, 0, tab, idx, nrows, out); }), 16);
  rep("global 16B, all lanes same address (broadcast)", timeit([&] { hipLaunchKernelGGL(gl_kernel<2>, dim3(blocks), dim3(256), 0, 0, tab, idx, nrows, out); }), 16);
  rep("global 16B, lane-own hot row (g-fast kernel)", timeit([&] { hipLaunchKernelGGL(gl_kernel<3>, dim3(blocks), dim3(256), 0, 0, tab, idx, nrows, out); }), 16);
  rep("global 16B fully coalesced (L2 stream)", timeit([&] { hipLaunchKernelGGL(gl_kernel<4>, dim3(blocks), dim3(256), 0, 0, tab, idx, nrows, out); }), 16);
  const int R = 256;
  rep("LDS b128, 8 lanes per row, stride 18", timeit([&] { hipLaunchKernelGGL((lds_kernel<0, 18>), dim3(blocks), dim3(256), R * 18 * 8, 0, tab, idx, R, out); }), 16);
  rep("LDS b128, 8 lanes per row, stride 16", timeit([&] { hipLaunchKernelGGL((lds_kernel<0, 16>), dim3(blocks), dim3(256), R * 16 * 8, 0, tab, idx, R, out); }), 16);
  rep("LDS b128, lane-own row, stride 18", timeit([&] { hipLaunchKernelGGL((lds_kernel<1, 18>), dim3(blocks), dim3(256), R * 18 * 8, 0, tab, idx, R, out); }), 16);
  rep("LDS b64,  lane-own row, stride 18", timeit([&] { hipLaunchKernelGGL((lds_kernel<2, 18>), dim3(blocks), dim3(256), R * 18 * 8, 0, tab, idx, R, out); }), 8);
  rep("LDS b64,  lane-own row, stride 17", timeit([&] { hipLaunchKernelGGL((lds_kernel<2, 17>), dim3(blocks), dim3(256), R * 17 * 8, 0, tab, idx, R, out); }), 8);
  rep("LDS b128 broadcast", timeit([&] { hipLaunchKernelGGL((lds_kernel<3, 18>), dim3(blocks), dim3(256), R * 18 * 8, 0, tab, idx, R, out); }), 16);
  return 0;
}
