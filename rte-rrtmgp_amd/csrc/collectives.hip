// collectives.hip -- the one exchange step of the column-sharded path, for host programs WITHOUT torch (an MPI host model,
// the reference's Fortran drivers): reduction / assembly of the broadband flux diagnostics over RCCL.
//
// The hot path has no cross-column dependence (every recurrence runs over the layers of one column:
// rte/kernels/mo_rte_solver_kernels.F90:697-706,1174-1202), so columns shard as contiguous ranges per rank with the
// k-distribution tables replicated and no data-path collective.  What ranks exchange is
//   * the domain-mean flux profile: per rank the column sums of flux_up / flux_dn (ncol_local, nlev), one ncclAllReduce(sum)
//     of 2 x nlev values, divided by the global column count;
//   * on request the assembled field: ncclAllGather of the (ncol_local, nlev) slabs into (ncol_local x nranks, nlev).
// The communicator is the CALLER's ncclComm_t (created from its MPI ranks: ncclGetUniqueId on rank 0, MPI_Bcast,
// ncclCommInitRank -- INTEGRATION.md section 5); the collective is enqueued on the context's stream behind the kernels that
// produced the fluxes.  RCCL is resolved at run time (the process's own librccl if it has one -- torch ships one -- else
// librccl.so): the library carries no link-time dependency on it, a single-GPU Fortran program never needs it installed.
#include <dlfcn.h>

#include <mutex>

#include "common.h"

namespace {

// the few declarations of rccl.h this file needs (ABI of NCCL 2.x / RCCL)
typedef void* NcclComm;
enum { kNcclSum = 0 };
enum { kNcclInt32 = 2, kNcclFloat32 = 7, kNcclFloat64 = 8 };
typedef int (*AllReduceFn)(const void*, void*, size_t, int, int, NcclComm, hipStream_t);
typedef int (*AllGatherFn)(const void*, void*, size_t, int, NcclComm, hipStream_t);
typedef int (*CommCountFn)(NcclComm, int*);
typedef const char* (*ErrStrFn)(int);

struct Rccl {
  AllReduceFn all_reduce = nullptr;
  AllGatherFn all_gather = nullptr;
  CommCountFn count = nullptr, user_rank = nullptr;
  ErrStrFn err = nullptr;
  bool ok = false;
};
Rccl& rccl() {
  static Rccl r;
  static std::once_flag once;
  std::call_once(once, [] {
    void* h = RTLD_DEFAULT;  // an RCCL the process has loaded already (the caller made the communicator with it)
    if (!dlsym(h, "ncclAllReduce")) {
      h = dlopen("librccl.so", RTLD_NOW | RTLD_GLOBAL);
      if (!h) h = dlopen("librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
      if (!h) return;
    }
    r.all_reduce = (AllReduceFn)dlsym(h, "ncclAllReduce");
    r.all_gather = (AllGatherFn)dlsym(h, "ncclAllGather");
    r.count = (CommCountFn)dlsym(h, "ncclCommCount");
    r.user_rank = (CommCountFn)dlsym(h, "ncclCommUserRank");
    r.err = (ErrStrFn)dlsym(h, "ncclGetErrorString");
    r.ok = r.all_reduce && r.all_gather && r.count;
  });
  return r;
}
void nccl_check(int rc, const char* what) {
  if (rc == 0) return;
  const char* msg = rccl().err ? rccl().err(rc) : "RCCL error";
  throw rte::Error{-1, std::string(what) + ": " + msg};
}

// column sums per level of two (ncol, nlev) fields: block = one level of one field, deterministic tree inside the block,
// blocks of a level added in a fixed order by the second stage
constexpr int kSumChunk = 16384;  // columns per block
__global__ void __launch_bounds__(256) column_sum_stage1(int ncol, int nlev, const Float* __restrict__ up, const Float* __restrict__ dn,
                                                          Float* __restrict__ part /*(2 * nlev, nchunk)*/, int nchunk) {
  __shared__ Float red[256];
  const int lev = blockIdx.y % nlev, which = blockIdx.y / nlev, chunk = blockIdx.x;
  const Float* f = (which ? dn : up) + (size_t)ncol * lev;
  const int c0 = chunk * kSumChunk, c1 = min(ncol, c0 + kSumChunk);
  Float s = 0;
  for (int c = c0 + threadIdx.x; c < c1; c += 256) s += f[c];
  red[threadIdx.x] = s;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if ((int)threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
    __syncthreads();
  }
  if (threadIdx.x == 0) part[(size_t)blockIdx.y * nchunk + chunk] = red[0];
}
__global__ void column_sum_stage2(int n, int nchunk, const Float* __restrict__ part, Float scale, Float* __restrict__ out) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  Float s = 0;
  for (int k = 0; k < nchunk; ++k) s += part[(size_t)i * nchunk + k];
  out[i] = s * scale;
}
// (nranks, ncol_local, nlev) -> (ncol_local * nranks, nlev): rank r's columns at r * ncol_local
__global__ void __launch_bounds__(256) interleave_slabs(int nranks, int ncol_local, int nlev, const Float* __restrict__ in,
                                                         Float* __restrict__ out) {
  const size_t n = (size_t)nranks * ncol_local * nlev;
  for (size_t i = blockIdx.x * (size_t)256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
    const size_t c = i % ncol_local, rest = i / ncol_local;
    const size_t lev = rest % nlev, r = rest / nlev;
    out[(r * ncol_local + c) + (size_t)nranks * ncol_local * lev] = in[i];
  }
}

// slabs of unequal width: (ncol_local, nlev) -> (ncol_slab, nlev), zero beyond the rank's own columns
__global__ void __launch_bounds__(256) pad_slab(int ncol_local, int ncol_slab, int nlev, const Float* __restrict__ in, Float* __restrict__ out) {
  const size_t n = (size_t)ncol_slab * nlev;
  for (size_t i = blockIdx.x * (size_t)256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
    const size_t c = i % ncol_slab, lev = i / ncol_slab;
    out[i] = c < (size_t)ncol_local ? in[c + (size_t)ncol_local * lev] : (Float)0;
  }
}
// (nranks, ncol_slab, nlev) with cnt[r] valid columns each -> (ncol_global, nlev): rank r's columns behind those of the ranks before it
__global__ void __launch_bounds__(256) compact_slabs(int nranks, int ncol_slab, int nlev, long long ncol_global, const int* __restrict__ cnt,
                                                      const Float* __restrict__ in, Float* __restrict__ out) {
  extern __shared__ long long off[];  // [nranks]: exclusive prefix sums of the counts
  if (threadIdx.x == 0) {
    long long o = 0;
    for (int r = 0; r < nranks; ++r) { off[r] = o; o += max(0, min(cnt[r], ncol_slab)); }
  }
  __syncthreads();
  const size_t n = (size_t)nranks * ncol_slab * nlev;
  for (size_t i = blockIdx.x * (size_t)256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
    const size_t c = i % ncol_slab, rest = i / ncol_slab;
    const size_t lev = rest % nlev, r = rest / nlev;
    const long long g = off[r] + (long long)c;
    if ((long long)c < (long long)cnt[r] && g < ncol_global) out[(size_t)g + (size_t)ncol_global * lev] = in[i];
  }
}

}  // namespace

extern "C" {

// 1 if an RCCL could be resolved in this process, else 0
int rte_hip_rccl_available(void) { return rccl().ok ? 1 : 0; }

// Domain-mean broadband flux profiles: mean_up(nlev), mean_dn(nlev) = sum over ALL ranks' columns / ncol_global, identical
// on every rank.  flux_up / flux_dn: this rank's (ncol_local, nlev) fields (device or host pointers, as everywhere);
// nccl_comm: the caller's ncclComm_t, or NULL for a single rank (then no RCCL is needed).  Returns 0, -1 on a HIP / RCCL
// error (error channel as for every entry point), -3 if nccl_comm is given but no RCCL can be resolved.
int rte_hip_allreduce_mean_profile(void* nccl_comm, int ncol_local, int nlev, const Float* flux_up, const Float* flux_dn,
                                   long long ncol_global, Float* mean_up, Float* mean_dn) {
  // (arguments every rank of the job shares decide the early return; a rank whose OWN column range is empty still joins the
  //  collective -- with zeros -- or the other ranks would wait for it forever)
  if (nlev <= 0 || ncol_global <= 0) return 0;
  if (ncol_local <= 0 && !nccl_comm) return 0;
  if (nccl_comm && !rccl().ok) return -3;
  RTE_TRY
  rte::Call c("rte_hip_allreduce_mean_profile");
  hipStream_t st = rte::stream();
  Float* prof = (Float*)rte::scratch(sizeof(Float) * (size_t)2 * nlev);
  if (ncol_local > 0) {
    const size_t n = (size_t)ncol_local * nlev;
    const Float* d_up = c.in(flux_up, n);
    const Float* d_dn = c.in(flux_dn, n);
    const int nchunk = (ncol_local + kSumChunk - 1) / kSumChunk;
    Float* part = (Float*)rte::scratch(sizeof(Float) * (size_t)2 * nlev * nchunk);
    rte::ProfScope p("mean_profile_sums");
    hipLaunchKernelGGL(column_sum_stage1, dim3(nchunk, 2 * nlev), dim3(256), 0, st, ncol_local, nlev, d_up, d_dn, part, nchunk);
    hipLaunchKernelGGL(column_sum_stage2, dim3(rte::cdiv(2 * nlev, 64)), dim3(64), 0, st, 2 * nlev, nchunk, (const Float*)part,
                       (Float)1 / (Float)ncol_global, prof);
  } else {
    HIP_CHECK(hipMemsetAsync(prof, 0, sizeof(Float) * (size_t)2 * nlev, st));
  }
  if (nccl_comm) {
    rte::ProfScope p("rccl_allreduce");
    nccl_check(rccl().all_reduce(prof, prof, (size_t)2 * nlev, sizeof(Float) == 8 ? kNcclFloat64 : kNcclFloat32, kNcclSum,
                                 (NcclComm)nccl_comm, st), "ncclAllReduce");
  }
  // (2 x nlev values: to wherever the caller wants them -- a synchronising copy for host pointers)
  const bool host_out = !rte::is_device_pointer(mean_up) || !rte::is_device_pointer(mean_dn);
  HIP_CHECK(hipMemcpyAsync(mean_up, prof, sizeof(Float) * nlev, hipMemcpyDefault, st));
  HIP_CHECK(hipMemcpyAsync(mean_dn, prof + nlev, sizeof(Float) * nlev, hipMemcpyDefault, st));
  if (host_out) HIP_CHECK(hipStreamSynchronize(st));
  return 0;
  RTE_CATCH("rte_hip_allreduce_mean_profile")
  return -1;
}

// The assembled field on every rank: global(ncol_local * nranks, nlev) from the ranks' local(ncol_local, nlev) slabs, rank r's
// columns at r * ncol_local (equal widths: pad the last rank's block to ncol_local columns).  `local` and `global` must be
// DEVICE pointers (RCCL moves device memory; 61 MB per field and rank at 125 000 columns x 61 levels).  ncol_local is the
// SAME on every rank (the slab width), so `ncol_local <= 0` returns on all ranks together: no rank waits for another.
int rte_hip_allgather_columns(void* nccl_comm, int ncol_local, int nlev, const Float* local, Float* global) {
  if (ncol_local <= 0 || nlev <= 0) return 0;
  if (!nccl_comm || !rccl().ok) return -3;
  if (!rte::is_device_memory(local) || !rte::is_device_memory(global)) return -2;
  RTE_TRY
  rte::Call c("rte_hip_allgather_columns");
  int nranks = 1;
  nccl_check(rccl().count((NcclComm)nccl_comm, &nranks), "ncclCommCount");
  hipStream_t st = rte::stream();
  const size_t n = (size_t)ncol_local * nlev;
  Float* staged = (Float*)rte::scratch(sizeof(Float) * n * nranks);
  {
    rte::ProfScope p("rccl_allgather");
    nccl_check(rccl().all_gather(local, staged, n, sizeof(Float) == 8 ? kNcclFloat64 : kNcclFloat32, (NcclComm)nccl_comm, st),
               "ncclAllGather");
  }
  rte::ProfScope p("interleave_slabs");
  hipLaunchKernelGGL(interleave_slabs, dim3(2048), dim3(256), 0, st, nranks, ncol_local, nlev, (const Float*)staged, global);
  return 0;
  RTE_CATCH("rte_hip_allgather_columns")
  return -1;
}

// The same for slabs of UNEQUAL width (shard boundaries on multiples of 64 columns: 1e6 columns on 8 ranks are 125 056 + 7 x
// 124 992): global(ncol_global, nlev) with rank r's ncol_local columns behind those of the ranks before
// it.  ncol_slab: a width every rank agrees on, >= every rank's ncol_local (the widest shard); ncol_global: the sum of the
// ranks' ncol_local.  The counts travel by a second, 4-byte-per-rank all-gather and stay on the device: no host synchronisation.
// ncol_slab, nlev and ncol_global are the same on every rank, so the early returns are taken by all ranks together; a rank
// whose own range is empty (ncol_local == 0) still takes part.
int rte_hip_allgatherv_columns(void* nccl_comm, int ncol_local, int nlev, const Float* local, int ncol_slab, long long ncol_global,
                               Float* global) {
  if (ncol_slab <= 0 || nlev <= 0 || ncol_global <= 0) return 0;
  if (ncol_local < 0 || ncol_local > ncol_slab) return -2;
  if (!nccl_comm || !rccl().ok) return -3;
  if ((ncol_local > 0 && !rte::is_device_memory(local)) || !rte::is_device_memory(global)) return -2;
  RTE_TRY
  rte::Call c("rte_hip_allgatherv_columns");
  int nranks = 1;
  nccl_check(rccl().count((NcclComm)nccl_comm, &nranks), "ncclCommCount");
  hipStream_t st = rte::stream();
  const size_t n = (size_t)ncol_slab * nlev;
  int* cnt = (int*)rte::scratch(sizeof(int) * (size_t)(nranks + 1));
  Float* staged = (Float*)rte::scratch(sizeof(Float) * n * nranks);
  const Float* mine = local;
  if (ncol_local != ncol_slab) {
    Float* padded = (Float*)rte::scratch(sizeof(Float) * n);
    hipLaunchKernelGGL(pad_slab, dim3(1024), dim3(256), 0, st, ncol_local, ncol_slab, nlev, local, padded);
    mine = padded;
  }
  HIP_CHECK(hipMemsetD32Async((hipDeviceptr_t)(cnt + nranks), ncol_local, 1, st));
  {
    rte::ProfScope p("rccl_allgather");
    nccl_check(rccl().all_gather(cnt + nranks, cnt, 1, kNcclInt32, (NcclComm)nccl_comm, st), "ncclAllGather (counts)");
    nccl_check(rccl().all_gather(mine, staged, n, sizeof(Float) == 8 ? kNcclFloat64 : kNcclFloat32, (NcclComm)nccl_comm, st),
               "ncclAllGather");
  }
  rte::ProfScope p("compact_slabs");
  hipLaunchKernelGGL(compact_slabs, dim3(2048), dim3(256), sizeof(long long) * nranks, st, nranks, ncol_slab, nlev, ncol_global,
                     (const int*)cnt, (const Float*)staged, global);
  return 0;
  RTE_CATCH("rte_hip_allgatherv_columns")
  return -1;
}

}  // extern "C"
