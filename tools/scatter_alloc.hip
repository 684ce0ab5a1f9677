// tools/scatter_alloc.hip -- experiment helper (not part of the library): a device array whose 2 MB pieces of physical memory are
// mapped into its virtual range in SHUFFLED order (HIP virtual-memory management: hipMemAddressReserve / hipMemCreate / hipMemMap).
// Why: compute_Planck_source is 14 % slower on physically contiguous output arrays than on arrays whose pages lie scattered over
// the device (docs/lab-notebook.md, round 6); which one an ordinary allocation gets is the driver's choice.
// build: hipcc -shared -fPIC -o tools/libscatter_alloc.so tools/scatter_alloc.hip
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

extern "C" void* scatter_alloc(size_t bytes, size_t chunk_hint, unsigned seed, int shuffle) {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) return nullptr;
  hipMemAllocationProp prop{};
  prop.type = hipMemAllocationTypePinned;
  prop.location.type = hipMemLocationTypeDevice;
  prop.location.id = dev;
  size_t gran = 0;
  if (hipMemGetAllocationGranularity(&gran, &prop, hipMemAllocationGranularityRecommended) != hipSuccess || gran == 0) return nullptr;
  size_t chunk = chunk_hint ? (chunk_hint + gran - 1) / gran * gran : gran;
  const size_t n = (bytes + chunk - 1) / chunk, total = n * chunk;
  void* va = nullptr;
  if (hipMemAddressReserve(&va, total, chunk, nullptr, 0) != hipSuccess) return nullptr;
  std::vector<hipMemGenericAllocationHandle_t> h(n);
  for (size_t i = 0; i < n; ++i)
    if (hipMemCreate(&h[i], chunk, &prop, 0) != hipSuccess) { fprintf(stderr, "scatter_alloc: hipMemCreate %zu of %zu failed\n", i, n); return nullptr; }
  std::vector<size_t> perm(n);
  for (size_t i = 0; i < n; ++i) perm[i] = i;
  if (shuffle) {
    unsigned long long st = seed * 2654435761ull + 88172645463325252ull;
    for (size_t i = n - 1; i > 0; --i) {
      st ^= st << 13; st ^= st >> 7; st ^= st << 17;
      const size_t j = (size_t)(st % (i + 1));
      const size_t t = perm[i]; perm[i] = perm[j]; perm[j] = t;
    }
  }
  for (size_t i = 0; i < n; ++i)
    if (hipMemMap((char*)va + i * chunk, chunk, 0, h[perm[i]], 0) != hipSuccess) { fprintf(stderr, "scatter_alloc: hipMemMap failed\n"); return nullptr; }
  hipMemAccessDesc acc{};
  acc.location.type = hipMemLocationTypeDevice;
  acc.location.id = dev;
  acc.flags = hipMemAccessFlagsProtReadWrite;
  if (hipMemSetAccess(va, total, &acc, 1) != hipSuccess) { fprintf(stderr, "scatter_alloc: hipMemSetAccess failed\n"); return nullptr; }
  for (size_t i = 0; i < n; ++i) (void)hipMemRelease(h[i]);  // (the mappings keep the memory alive)
  fprintf(stderr, "scatter_alloc: %zu chunks of %zu KB (granularity %zu KB), %s\n", n, chunk >> 10, gran >> 10, shuffle ? "shuffled" : "in order");
  return va;
}
