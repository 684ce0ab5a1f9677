/*
 * glue_recorder.c -- test infrastructure (our own file): a pass-through in front of three kernel symbols of the
 * reference's CPU build that RECORDS what the reference's Fortran FRONTEND computed on its way to them, then calls the
 * real kernel (dlsym RTLD_NEXT -> oracle/_ref/librefkernels.so), so the driver runs on normally:
 *   rrtmgp_interpolation           col_gas(ncol, nlay, 0:ngas)   -- the frontend's vmr x col_dry fill,
 *                                                                   rrtmgp/frontend/mo_gas_optics_rrtmgp.F90:594-609
 *   rrtmgp_compute_Planck_source   tlev(ncol, nlay+1)            -- interpolated by the frontend when the caller gives none, :893-912
 *   rte_lw_solver_noscat           Ds(ncol, ngpt, nmus)          -- secants: compute_optimal_angles :1536-1561 through rte_lw(lw_Ds=),
 *                                                                   or the Gauss tables, rte/frontend/mo_rte_lw.F90:346-365
 *                                  sfc_emis(ncol, ngpt)          -- expand_and_transpose of the by-band emissivity, mo_rte_lw.F90:478-501
 *                                  tau(ncol, nlay, ngpt)         -- (input of compute_optimal_angles; recorded so that the fixture is
 *                                                                   self-contained)
 * Linked into oracle/_ref/bin/ref_frontend_driver_glue (oracle/build_extern.sh); tests/golden/make_glue_golden.py turns the
 * records of one run into tests/golden/glue_frontend.npz, against which oracle/glue_oracle.c (CPU) and csrc/glue.hip (GPU)
 * are compared (tests/test_glue.py).  Record format as oracle/abi_recorder.c: tag char[32], kind int32 (1 float64), rank,
 * dims, payload (column-major); one set of records per call (= per block of columns).
 */
#define _GNU_SOURCE
#include <dlfcn.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "../include/rte_rrtmgp_kernels.h"

static FILE* out(void) {
  static FILE* f = NULL;
  if (!f) {
    const char* p = getenv("RTE_ABI_RECORD");
    f = fopen(p ? p : "glue_record.bin", "wb");
    if (!f) abort();
  }
  return f;
}
static void rec_r(const char* tag, int rank, const int* d, const Float* a) {
  char t[32];
  int kind = 1;
  size_t n = 1;
  memset(t, ' ', 32);
  memcpy(t, tag, strlen(tag) < 32 ? strlen(tag) : 32);
  fwrite(t, 1, 32, out()); fwrite(&kind, 4, 1, out()); fwrite(&rank, 4, 1, out()); fwrite(d, 4, rank, out());
  for (int i = 0; i < rank; ++i) n *= (size_t)d[i];
  for (size_t i = 0; i < n; ++i) { double v = (double)a[i]; fwrite(&v, 8, 1, out()); }
  fflush(out());
}
static void* next(const char* name) {
  void* p = dlsym(RTLD_NEXT, name);
  if (!p) { fprintf(stderr, "glue_recorder: no %s behind the recorder\n", name); abort(); }
  return p;
}

void rrtmgp_interpolation(const int* ncol, const int* nlay, const int* ngas, const int* nflav, const int* neta,
                          const int* npres, const int* ntemp, const int* flavor, const Float* press_ref_log,
                          const Float* temp_ref, const Float* press_ref_log_delta, const Float* temp_ref_min,
                          const Float* temp_ref_delta, const Float* press_ref_trop_log, const Float* vmr_ref,
                          const Float* play, const Float* tlay, const Float* col_gas, int* jtemp, Float* fmajor,
                          Float* fminor, Float* col_mix, Bool* tropo, int* jeta, int* jpress) {
  static __typeof__(rrtmgp_interpolation)* real = NULL;
  if (!real) real = (__typeof__(rrtmgp_interpolation)*)next("rrtmgp_interpolation");
  rec_r("col_gas", 3, (int[]){*ncol, *nlay, *ngas + 1}, col_gas);
  real(ncol, nlay, ngas, nflav, neta, npres, ntemp, flavor, press_ref_log, temp_ref, press_ref_log_delta, temp_ref_min,
       temp_ref_delta, press_ref_trop_log, vmr_ref, play, tlay, col_gas, jtemp, fmajor, fminor, col_mix, tropo, jeta, jpress);
}

void rrtmgp_compute_Planck_source(const int* ncol, const int* nlay, const int* nbnd, const int* ngpt, const int* nflav,
                                  const int* neta, const int* npres, const int* ntemp, const int* nPlanckTemp,
                                  const Float* tlay, const Float* tlev, const Float* tsfc, const int* sfc_lay,
                                  const Float* fmajor, const int* jeta, const Bool* tropo, const int* jtemp,
                                  const int* jpress, const int* gpoint_bands, const int* band_lims_gpt, const Float* pfracin,
                                  const Float* temp_ref_min, const Float* totplnk_delta, const Float* totplnk,
                                  const int* gpoint_flavor, Float* sfc_src, Float* lay_src, Float* lev_src,
                                  Float* sfc_source_Jac) {
  static __typeof__(rrtmgp_compute_Planck_source)* real = NULL;
  if (!real) real = (__typeof__(rrtmgp_compute_Planck_source)*)next("rrtmgp_compute_Planck_source");
  rec_r("tlev", 2, (int[]){*ncol, *nlay + 1}, tlev);
  real(ncol, nlay, nbnd, ngpt, nflav, neta, npres, ntemp, nPlanckTemp, tlay, tlev, tsfc, sfc_lay, fmajor, jeta, tropo, jtemp,
       jpress, gpoint_bands, band_lims_gpt, pfracin, temp_ref_min, totplnk_delta, totplnk, gpoint_flavor, sfc_src, lay_src,
       lev_src, sfc_source_Jac);
}

void rte_lw_solver_noscat(const int* ncol, const int* nlay, const int* ngpt, const Bool* top_at_1, const int* nmus,
                          const Float* Ds, const Float* weights, const Float* tau, const Float* lay_source,
                          const Float* lev_source, const Float* sfc_emis, const Float* sfc_src, const Float* inc_flux,
                          Float* flux_up, Float* flux_dn, const Bool* do_broadband, Float* broadband_up, Float* broadband_dn,
                          const Bool* do_Jacobians, const Float* sfc_srcJac, Float* flux_upJac, const Bool* do_rescaling,
                          const Float* ssa, const Float* g) {
  static __typeof__(rte_lw_solver_noscat)* real = NULL;
  if (!real) real = (__typeof__(rte_lw_solver_noscat)*)next("rte_lw_solver_noscat");
  rec_r("Ds", 3, (int[]){*ncol, *ngpt, *nmus}, Ds);
  rec_r("weights", 1, (int[]){*nmus}, weights);
  rec_r("sfc_emis_gpt", 2, (int[]){*ncol, *ngpt}, sfc_emis);
  rec_r("tau", 3, (int[]){*ncol, *nlay, *ngpt}, tau);
  real(ncol, nlay, ngpt, top_at_1, nmus, Ds, weights, tau, lay_source, lev_source, sfc_emis, sfc_src, inc_flux, flux_up, flux_dn,
       do_broadband, broadband_up, broadband_dn, do_Jacobians, sfc_srcJac, flux_upJac, do_rescaling, ssa, g);
}
