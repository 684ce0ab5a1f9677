/*
 * omp_mapped_check.c -- test infrastructure (our own file).  A host program with OpenMP target offload (as the reference
 * frontend is when built with -fopenmp --offload-arch=gfx950) keeps arrays on the device in a `target data` region and calls
 * kernel symbols of librte_rrtmgp_hip.so with the HOST addresses of the mapped arrays; the library must resolve them with
 * omp_get_mapped_ptr and work on the device copies in place (csrc/runtime.hip: omp_mapped), with no staging.
 *   rte_sum_broadband (reference interface rte/kernels/api/mo_fluxes_broadband_kernels.F90) on a mapped spectral array,
 *   then a device-side check of the result inside the same region (the host copy is only updated at the region's end),
 *   then the same call on unmapped (staged) arrays.  Prints "omp_mapped_check ok" and the library's staging counters.
 * Built by oracle/build_extern_offload.sh (amdclang -fopenmp --offload-arch=gfx950), run by tests/test_extern_frontend.py.
 */
#include <math.h>
#include <omp.h>
#include <stdio.h>
#include <stdlib.h>
#include "../include/rte_rrtmgp_kernels.h"
long long rte_hip_mirror_stat(int which);

int main(void) {
  const int ncol = 3000, nlev = 61, ngpt = 64;
  const size_t n2 = (size_t)ncol * nlev, n3 = n2 * ngpt;
  double* spec = (double*)malloc(n3 * sizeof(double));
  double* bb = (double*)malloc(n2 * sizeof(double));
  double* bb_staged = (double*)malloc(n2 * sizeof(double));
  for (size_t i = 0; i < n3; ++i) spec[i] = 1.0 + (double)(i % 977) * 1e-3;
  for (size_t i = 0; i < n2; ++i) bb[i] = bb_staged[i] = -1.0;
  if (omp_get_num_devices() < 1) { printf("omp_mapped_check: no offload device\n"); return 2; }
  int bad_dev = 0;
  rte_hip_mirror_stat(-1);
#pragma omp target data map(to : spec[0 : n3]) map(from : bb[0 : n2])
  {
    if (!omp_get_mapped_ptr(spec, omp_get_default_device())) { printf("omp_mapped_check: spec is not mapped\n"); exit(3); }
    rte_sum_broadband(&ncol, &nlev, &ngpt, spec, bb);   /* host addresses of mapped arrays */
    /* the result must be in the DEVICE copy now: checked by a target region, before anything is copied back */
#pragma omp target teams distribute parallel for reduction(+ : bad_dev) map(tofrom : bad_dev)
    for (size_t i = 0; i < n2; ++i) {
      double s = 0;
      for (int g = 0; g < ngpt; ++g) s += spec[i + n2 * (size_t)g];
      if (fabs(bb[i] - s) > 1e-12 * fabs(s)) bad_dev += 1;
    }
  }
  const long long h2d_mapped = rte_hip_mirror_stat(2), d2h_mapped = rte_hip_mirror_stat(3);
  rte_sum_broadband(&ncol, &nlev, &ngpt, spec, bb_staged);  /* unmapped: staged through the arena */
  const long long h2d_staged = rte_hip_mirror_stat(2) - h2d_mapped;
  int bad_host = 0;
  for (size_t i = 0; i < n2; ++i) bad_host += bb[i] != bb_staged[i];
  printf("mapped call: %lld bytes staged to the device, %lld back; staged call: %lld bytes to the device\n", h2d_mapped, d2h_mapped, h2d_staged);
  printf("device-side mismatches %d, mapped vs staged mismatches %d\n", bad_dev, bad_host);
  if (bad_dev == 0 && bad_host == 0 && h2d_mapped == 0 && d2h_mapped == 0 && h2d_staged >= (long long)(n3 * sizeof(double))) {
    printf("omp_mapped_check ok\n");
    return 0;
  }
  return 1;
}
