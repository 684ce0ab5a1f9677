/*
 * rte_rrtmgp_kernels.h -- C ABI of the MI355X-native RTE+RRTMGP kernel library
 * (librte_rrtmgp_hip.so).
 *
 * This is the drop-in boundary: every entry point below has the name, the
 * argument order and the argument meaning of the reference's `bind(C)`
 * kernel interface, i.e. exactly what the reference Fortran frontend links
 * against when it is configured with RTE_KERNEL_MODE=extern
 * (reference CMakeLists.txt:32-37, rte/kernels/CMakeLists.txt:3-13,
 * rrtmgp/kernels/CMakeLists.txt:3-9).  The reference generates an equivalent
 * header with cbind_generator.py; this one is hand-written so that it is
 * valid C (scalars are `const T*`, ABI-identical to Fortran by-reference
 * scalars and to the generator's C++ `const T&`).
 *
 * Conventions (reference rte/kernels/mo_rte_kind.F90:24-40,
 * rte/kernels/api/rte_types.h.in:23-26):
 *   - Float = double (default) or float when built with -DRTE_USE_SP.
 *   - Bool  = 1-byte C _Bool  (Fortran logical(c_bool)).
 *   - int   = 32-bit default Fortran integer.
 *   - ALL scalars are passed by address.
 *   - Arrays are dense, column-major (Fortran order), first element at the
 *     pointer; the dimension comment gives the Fortran shape, leftmost index
 *     fastest.  Index VALUES stored in integer arrays are 1-based.
 *   - Pointers may be device pointers (launched in place, asynchronously on
 *     the library stream) or host pointers (staged through a device arena and
 *     copied back, synchronously) -- see INTEGRATION.md.
 *   - There is no error channel in the reference interface (all `void`);
 *     a HIP failure prints to stderr and aborts.
 */
#ifndef RTE_RRTMGP_KERNELS_H
#define RTE_RRTMGP_KERNELS_H

#include <stdbool.h>
#include <stddef.h>

#ifdef RTE_USE_SP
typedef float Float;
#else
typedef double Float;
#endif
#ifdef __cplusplus
typedef bool Bool; /* 1 byte, same representation as C _Bool */
#else
typedef _Bool Bool;
#endif

#ifdef __cplusplus
extern "C" {
#endif

/* ------------------------------------------------------------------------
 * RRTMGP gas optics  (reference rrtmgp/kernels/api/mo_gas_optics_rrtmgp_kernels.F90)
 * ---------------------------------------------------------------------- */

/* replaces `interpolation`, api :9-67 (default impl
 * rrtmgp/kernels/mo_gas_optics_rrtmgp_kernels.F90:37-170) */
void rrtmgp_interpolation(
    const int* ncol, const int* nlay, const int* ngas, const int* nflav,
    const int* neta, const int* npres, const int* ntemp,
    const int* flavor,              /* (2,nflav) */
    const Float* press_ref_log,     /* (npres) */
    const Float* temp_ref,          /* (ntemp) */
    const Float* press_ref_log_delta, const Float* temp_ref_min,
    const Float* temp_ref_delta, const Float* press_ref_trop_log,
    const Float* vmr_ref,           /* (2,0:ngas,ntemp) */
    const Float* play,              /* (ncol,nlay) */
    const Float* tlay,              /* (ncol,nlay) */
    const Float* col_gas,           /* (ncol,nlay,0:ngas) */
    int* jtemp,                     /* (ncol,nlay) */
    Float* fmajor,                  /* (2,2,2,ncol,nlay,nflav) */
    Float* fminor,                  /* (2,2,ncol,nlay,nflav) */
    Float* col_mix,                 /* (2,ncol,nlay,nflav) */
    Bool* tropo,                    /* (ncol,nlay) */
    int* jeta,                      /* (2,ncol,nlay,nflav) */
    int* jpress);                   /* (ncol,nlay) */

/* replaces `compute_tau_absorption`, api :71-160 (default impl :176-501).
 * tau is intent(inout): the result is ACCUMULATED onto what is there. */
void rrtmgp_compute_tau_absorption(
    const int* ncol, const int* nlay, const int* nbnd, const int* ngpt,
    const int* ngas, const int* nflav, const int* neta, const int* npres,
    const int* ntemp,
    const int* nminorlower, const int* nminorklower,
    const int* nminorupper, const int* nminorkupper,
    const int* idx_h2o,
    const int* gpoint_flavor,       /* (2,ngpt) */
    const int* band_lims_gpt,       /* (2,nbnd) */
    const Float* kmajor,            /* (ntemp,neta,npres+1,ngpt) */
    const Float* kminor_lower,      /* (ntemp,neta,nminorklower) */
    const Float* kminor_upper,      /* (ntemp,neta,nminorkupper) */
    const int* minor_limits_gpt_lower,          /* (2,nminorlower) */
    const int* minor_limits_gpt_upper,          /* (2,nminorupper) */
    const Bool* minor_scales_with_density_lower,/* (nminorlower) */
    const Bool* minor_scales_with_density_upper,/* (nminorupper) */
    const Bool* scale_by_complement_lower,      /* (nminorlower) */
    const Bool* scale_by_complement_upper,      /* (nminorupper) */
    const int* idx_minor_lower,                 /* (nminorlower) */
    const int* idx_minor_upper,                 /* (nminorupper) */
    const int* idx_minor_scaling_lower,         /* (nminorlower) */
    const int* idx_minor_scaling_upper,         /* (nminorupper) */
    const int* kminor_start_lower,              /* (nminorlower) */
    const int* kminor_start_upper,              /* (nminorupper) */
    const Bool* tropo,              /* (ncol,nlay) */
    const Float* col_mix,           /* (2,ncol,nlay,nflav) */
    const Float* fmajor,            /* (2,2,2,ncol,nlay,nflav) */
    const Float* fminor,            /* (2,2,ncol,nlay,nflav) */
    const Float* play,              /* (ncol,nlay) */
    const Float* tlay,              /* (ncol,nlay) */
    const Float* col_gas,           /* (ncol,nlay,0:ngas) */
    const int* jeta,                /* (2,ncol,nlay,nflav) */
    const int* jtemp,               /* (ncol,nlay) */
    const int* jpress,              /* (ncol,nlay) */
    Float* tau);                    /* (ncol,nlay,ngpt) inout */

/* replaces `compute_tau_rayleigh`, api :163-199 (default impl :506-565) */
void rrtmgp_compute_tau_rayleigh(
    const int* ncol, const int* nlay, const int* nbnd, const int* ngpt,
    const int* ngas, const int* nflav, const int* neta, const int* npres,
    const int* ntemp,
    const int* gpoint_flavor,       /* (2,ngpt) */
    const int* band_lims_gpt,       /* (2,nbnd) */
    const Float* krayl,             /* (ntemp,neta,ngpt,2) */
    const int* idx_h2o,
    const Float* col_dry,           /* (ncol,nlay) */
    const Float* col_gas,           /* (ncol,nlay,0:ngas) */
    const Float* fminor,            /* (2,2,ncol,nlay,nflav) */
    const int* jeta,                /* (2,ncol,nlay,nflav) */
    const Bool* tropo,              /* (ncol,nlay) */
    const int* jtemp,               /* (ncol,nlay) */
    Float* tau_rayleigh);           /* (ncol,nlay,ngpt) out */

/* replaces `compute_Planck_source`, api :202-243 (default impl :568-710) */
void rrtmgp_compute_Planck_source(
    const int* ncol, const int* nlay, const int* nbnd, const int* ngpt,
    const int* nflav, const int* neta, const int* npres, const int* ntemp,
    const int* nPlanckTemp,
    const Float* tlay,              /* (ncol,nlay) */
    const Float* tlev,              /* (ncol,nlay+1) */
    const Float* tsfc,              /* (ncol) */
    const int* sfc_lay,
    const Float* fmajor,            /* (2,2,2,ncol,nlay,nflav) */
    const int* jeta,                /* (2,ncol,nlay,nflav) */
    const Bool* tropo,              /* (ncol,nlay) */
    const int* jtemp,               /* (ncol,nlay) */
    const int* jpress,              /* (ncol,nlay) */
    const int* gpoint_bands,        /* (ngpt) */
    const int* band_lims_gpt,       /* (2,nbnd) */
    const Float* pfracin,           /* (ntemp,neta,npres+1,ngpt) */
    const Float* temp_ref_min, const Float* totplnk_delta,
    const Float* totplnk,           /* (nPlanckTemp,nbnd) */
    const int* gpoint_flavor,       /* (2,ngpt) */
    Float* sfc_src,                 /* (ncol,ngpt) */
    Float* lay_src,                 /* (ncol,nlay,ngpt) */
    Float* lev_src,                 /* (ncol,nlay+1,ngpt) */
    Float* sfc_source_Jac);         /* (ncol,ngpt) */

/* ------------------------------------------------------------------------
 * RTE solvers  (reference rte/kernels/api/mo_rte_solver_kernels.F90)
 * ---------------------------------------------------------------------- */

/* replaces `lw_solver_noscat`, api :41-96 (default impl
 * rte/kernels/mo_rte_solver_kernels.F90:248-367, :51-240).
 * flux_up/flux_dn are written only when !do_broadband; broadband_* only when
 * do_broadband; sfc_srcJac/flux_upJac touched only when do_Jacobians; ssa/g
 * only when do_rescaling (the frontend passes decoys otherwise). */
void rte_lw_solver_noscat(
    const int* ncol, const int* nlay, const int* ngpt, const Bool* top_at_1,
    const int* nmus,
    const Float* Ds,                /* (ncol,ngpt,nmus) */
    const Float* weights,           /* (nmus) */
    const Float* tau,               /* (ncol,nlay,ngpt) */
    const Float* lay_source,        /* (ncol,nlay,ngpt) */
    const Float* lev_source,        /* (ncol,nlay+1,ngpt) */
    const Float* sfc_emis,          /* (ncol,ngpt) */
    const Float* sfc_src,           /* (ncol,ngpt) */
    const Float* inc_flux,          /* (ncol,ngpt) */
    Float* flux_up,                 /* (ncol,nlay+1,ngpt) */
    Float* flux_dn,                 /* (ncol,nlay+1,ngpt) */
    const Bool* do_broadband,
    Float* broadband_up,            /* (ncol,nlay+1) */
    Float* broadband_dn,            /* (ncol,nlay+1) */
    const Bool* do_Jacobians,
    const Float* sfc_srcJac,        /* (ncol,ngpt) */
    Float* flux_upJac,              /* (ncol,nlay+1) */
    const Bool* do_rescaling,
    const Float* ssa,               /* (ncol,nlay,ngpt) */
    const Float* g);                /* (ncol,nlay,ngpt) */

/* replaces `lw_solver_2stream`, api :107-133 (default impl :377-440) */
void rte_lw_solver_2stream(
    const int* ncol, const int* nlay, const int* ngpt, const Bool* top_at_1,
    const Float* tau, const Float* ssa, const Float* g, /* (ncol,nlay,ngpt) */
    const Float* lay_source,        /* (ncol,nlay,ngpt) */
    const Float* lev_source,        /* (ncol,nlay+1,ngpt) */
    const Float* sfc_emis,          /* (ncol,ngpt) */
    const Float* sfc_src,           /* (ncol,ngpt) */
    const Float* inc_flux,          /* (ncol,ngpt) */
    Float* flux_up,                 /* (ncol,nlay+1,ngpt) */
    Float* flux_dn);                /* (ncol,nlay+1,ngpt) */

/* replaces `sw_solver_noscat`, api :143-159 (default impl :450-494) */
void rte_sw_solver_noscat(
    const int* ncol, const int* nlay, const int* ngpt, const Bool* top_at_1,
    const Float* tau,               /* (ncol,nlay,ngpt) */
    const Float* mu0,               /* (ncol,nlay) */
    const Float* inc_flux_dir,      /* (ncol,ngpt) */
    Float* flux_dir);               /* (ncol,nlay+1,ngpt) */

/* replaces `sw_solver_2stream`, api :168-200 (default impl :503-609).
 * In broadband mode flux_up/flux_dn/flux_dir may all alias one decoy
 * (rte/frontend/mo_rte_sw.F90:204-207) and are never written. */
void rte_sw_solver_2stream(
    const int* ncol, const int* nlay, const int* ngpt, const Bool* top_at_1,
    const Float* tau, const Float* ssa, const Float* g, /* (ncol,nlay,ngpt) */
    const Float* mu0,               /* (ncol,nlay) */
    const Float* sfc_alb_dir,       /* (ncol,ngpt) */
    const Float* sfc_alb_dif,       /* (ncol,ngpt) */
    const Float* inc_flux_dir,      /* (ncol,ngpt) */
    Float* flux_up, Float* flux_dn, Float* flux_dir, /* (ncol,nlay+1,ngpt) */
    const Bool* has_dif_bc,
    const Float* inc_flux_dif,      /* (ncol,ngpt) */
    const Bool* do_broadband,
    Float* broadband_up, Float* broadband_dn, Float* broadband_dir); /* (ncol,nlay+1) */

/* ------------------------------------------------------------------------
 * Flux reductions (reference rte/kernels/api/mo_fluxes_broadband_kernels.F90;
 * default impl rte/kernels/mo_fluxes_broadband_kernels.F90:32-128)
 * ---------------------------------------------------------------------- */
void rte_sum_broadband(const int* ncol, const int* nlev, const int* ngpt,
                       const Float* spectral_flux,   /* (ncol,nlev,ngpt) */
                       Float* broadband_flux);       /* (ncol,nlev) */
void rte_net_broadband_full(const int* ncol, const int* nlev, const int* ngpt,
                            const Float* spectral_flux_dn, /* (ncol,nlev,ngpt) */
                            const Float* spectral_flux_up, /* (ncol,nlev,ngpt) */
                            Float* broadband_flux_net);    /* (ncol,nlev) */
void rte_net_broadband_precalc(const int* ncol, const int* nlev,
                               const Float* flux_dn, const Float* flux_up, /* (ncol,nlev) */
                               Float* broadband_flux_net);                 /* (ncol,nlev) */

/* ------------------------------------------------------------------------
 * Array utilities (reference rte/kernels/api/mo_rte_util_array.F90;
 * default impl rte/kernels/mo_rte_util_array.F90:32-132)
 * ---------------------------------------------------------------------- */
void zero_array_1D(const int* ni, Float* array);
void zero_array_2D(const int* ni, const int* nj, Float* array);
void zero_array_3D(const int* ni, const int* nj, const int* nk, Float* array);
void zero_array_4D(const int* ni, const int* nj, const int* nk, const int* nl, Float* array);
void set_to_scalar_1D(const int* ni, Float* array, const Float* value);
void set_to_scalar_2D(const int* ni, const int* nj, Float* array, const Float* value);
void set_to_scalar_3D(const int* ni, const int* nj, const int* nk, Float* array, const Float* value);
void set_to_scalar_4D(const int* ni, const int* nj, const int* nk, const int* nl, Float* array,
                      const Float* value);

/* ------------------------------------------------------------------------
 * Optical-properties arithmetic (reference rte/kernels/api/mo_optical_props_kernels.F90;
 * default impl rte/kernels/mo_optical_props_kernels.F90:44-778).  Operand 1 is modified in place;
 * operand 2 is on g-points, or on bands for the *_bybnd variants (gpt_lims = (2,nbnd), 1-based).
 * ---------------------------------------------------------------------- */
void rte_delta_scale_2str_f_k(const int* ncol, const int* nlay, const int* ngpt,
                              Float* tau, Float* ssa, Float* g,   /* (ncol,nlay,ngpt) inout */
                              const Float* f);                    /* (ncol,nlay,ngpt) */
void rte_delta_scale_2str_k(const int* ncol, const int* nlay, const int* ngpt,
                            Float* tau, Float* ssa, Float* g);    /* (ncol,nlay,ngpt) inout */
void rte_increment_1scalar_by_1scalar(const int* ncol, const int* nlay, const int* ngpt,
                                      Float* tau1, const Float* tau2);
void rte_increment_1scalar_by_2stream(const int* ncol, const int* nlay, const int* ngpt,
                                      Float* tau1, const Float* tau2, const Float* ssa2);
void rte_increment_1scalar_by_nstream(const int* ncol, const int* nlay, const int* ngpt,
                                      Float* tau1, const Float* tau2, const Float* ssa2);
void rte_increment_2stream_by_1scalar(const int* ncol, const int* nlay, const int* ngpt,
                                      Float* tau1, Float* ssa1, const Float* tau2);
void rte_increment_2stream_by_2stream(const int* ncol, const int* nlay, const int* ngpt,
                                      Float* tau1, Float* ssa1, Float* g1,
                                      const Float* tau2, const Float* ssa2, const Float* g2);
void rte_increment_2stream_by_nstream(const int* ncol, const int* nlay, const int* ngpt, const int* nmom2,
                                      Float* tau1, Float* ssa1, Float* g1,
                                      const Float* tau2, const Float* ssa2,
                                      const Float* p2);           /* (nmom2,ncol,nlay,ngpt) */
void rte_increment_nstream_by_1scalar(const int* ncol, const int* nlay, const int* ngpt,
                                      Float* tau1, Float* ssa1, const Float* tau2);
void rte_increment_nstream_by_2stream(const int* ncol, const int* nlay, const int* ngpt, const int* nmom1,
                                      Float* tau1, Float* ssa1,
                                      Float* p1,                  /* (nmom1,ncol,nlay,ngpt) */
                                      const Float* tau2, const Float* ssa2, const Float* g2);
void rte_increment_nstream_by_nstream(const int* ncol, const int* nlay, const int* ngpt,
                                      const int* nmom1, const int* nmom2,
                                      Float* tau1, Float* ssa1, Float* p1,
                                      const Float* tau2, const Float* ssa2, const Float* p2);
void rte_inc_1scalar_by_1scalar_bybnd(const int* ncol, const int* nlay, const int* ngpt,
                                      Float* tau1, const Float* tau2 /* (ncol,nlay,nbnd) */,
                                      const int* nbnd, const int* gpt_lims);
void rte_inc_1scalar_by_2stream_bybnd(const int* ncol, const int* nlay, const int* ngpt,
                                      Float* tau1, const Float* tau2, const Float* ssa2,
                                      const int* nbnd, const int* gpt_lims);
void rte_inc_1scalar_by_nstream_bybnd(const int* ncol, const int* nlay, const int* ngpt,
                                      Float* tau1, const Float* tau2, const Float* ssa2,
                                      const int* nbnd, const int* gpt_lims);
void rte_inc_2stream_by_1scalar_bybnd(const int* ncol, const int* nlay, const int* ngpt,
                                      Float* tau1, Float* ssa1, const Float* tau2,
                                      const int* nbnd, const int* gpt_lims);
void rte_inc_2stream_by_2stream_bybnd(const int* ncol, const int* nlay, const int* ngpt,
                                      Float* tau1, Float* ssa1, Float* g1,
                                      const Float* tau2, const Float* ssa2, const Float* g2,
                                      const int* nbnd, const int* gpt_lims);
void rte_inc_2stream_by_nstream_bybnd(const int* ncol, const int* nlay, const int* ngpt, const int* nmom2,
                                      Float* tau1, Float* ssa1, Float* g1,
                                      const Float* tau2, const Float* ssa2, const Float* p2,
                                      const int* nbnd, const int* gpt_lims);
void rte_inc_nstream_by_1scalar_bybnd(const int* ncol, const int* nlay, const int* ngpt,
                                      Float* tau1, Float* ssa1, const Float* tau2,
                                      const int* nbnd, const int* gpt_lims);
void rte_inc_nstream_by_2stream_bybnd(const int* ncol, const int* nlay, const int* ngpt, const int* nmom1,
                                      Float* tau1, Float* ssa1, Float* p1,
                                      const Float* tau2, const Float* ssa2, const Float* g2,
                                      const int* nbnd, const int* gpt_lims);
void rte_inc_nstream_by_nstream_bybnd(const int* ncol, const int* nlay, const int* ngpt,
                                      const int* nmom1, const int* nmom2,
                                      Float* tau1, Float* ssa1, Float* p1,
                                      const Float* tau2, const Float* ssa2, const Float* p2,
                                      const int* nbnd, const int* gpt_lims);
void rte_extract_subset_dim1_3d(const int* ncol, const int* nlay, const int* ngpt,
                                const Float* array_in,            /* (ncol,nlay,ngpt) */
                                const int* colS, const int* colE,
                                Float* array_out);                /* (colE-colS+1,nlay,ngpt) */
void rte_extract_subset_dim2_4d(const int* nmom, const int* ncol, const int* nlay, const int* ngpt,
                                const Float* array_in,            /* (nmom,ncol,nlay,ngpt) */
                                const int* colS, const int* colE,
                                Float* array_out);                /* (nmom,colE-colS+1,nlay,ngpt) */
void rte_extract_subset_absorption_tau(const int* ncol, const int* nlay, const int* ngpt,
                                       const Float* tau_in, const Float* ssa_in,
                                       const int* colS, const int* colE,
                                       Float* tau_out);           /* (colE-colS+1,nlay,ngpt) */

/* ------------------------------------------------------------------------
 * Cloud optics from look-up tables (reference rrtmgp/kernels/api/mo_cloud_optics_rrtmgp_kernels.F90;
 * default impl rrtmgp/kernels/mo_cloud_optics_rrtmgp_kernels.F90:24-65)
 * ---------------------------------------------------------------------- */
void rrtmgp_compute_cld_from_table(const int* ncol, const int* nlay, const int* ngpt,
                                   const Bool* mask,              /* (ncol,nlay) */
                                   const Float* lwp,              /* (ncol,nlay) */
                                   const Float* re,               /* (ncol,nlay) */
                                   const int* nsteps, const Float* step_size, const Float* offset,
                                   const Float* tau_table,        /* (nsteps,ngpt) */
                                   const Float* ssa_table,        /* (nsteps,ngpt) */
                                   const Float* asy_table,        /* (nsteps,ngpt) */
                                   Float* tau, Float* taussa, Float* taussag); /* (ncol,nlay,ngpt) */

/* ------------------------------------------------------------------------
 * Planck function on a wavenumber grid (reference rte/kernels/api/mo_gas_optics_utils.F90:6-34;
 * default impl rte/kernels/mo_gas_optics_utils.F90:36-95): source = B_nu(T, nus) * dnus
 * ---------------------------------------------------------------------- */
void rte_compute_Planck_source_2D(const int* ncol, const int* nlay, const int* nnu,
                                  const Float* nus, const Float* dnus, /* (nnu) */
                                  const Float* T,                      /* (ncol,nlay) */
                                  Float* source);                      /* (ncol,nlay,nnu) */
void rte_compute_Planck_source_1D(const int* ncol, const int* nnu,
                                  const Float* nus, const Float* dnus, /* (nnu) */
                                  const Float* T,                      /* (ncol) */
                                  Float* source);                      /* (ncol,nnu) */

/* ------------------------------------------------------------------------
 * By-band flux reductions (reference rte/extensions/mo_fluxes_byband.F90:156-209, bind(C) there);
 * band_lims = (2,nbnd) 1-based inclusive g-point limits
 * ---------------------------------------------------------------------- */
void rte_sum_byband(const int* ncol, const int* nlev, const int* ngpt, const int* nbnd,
                    const int* band_lims,
                    const Float* spectral_flux,  /* (ncol,nlev,ngpt) */
                    Float* byband_flux);         /* (ncol,nlev,nbnd) */
void rte_net_byband_full(const int* ncol, const int* nlev, const int* ngpt, const int* nbnd,
                         const int* band_lims,
                         const Float* spectral_flux_dn, const Float* spectral_flux_up, /* (ncol,nlev,ngpt) */
                         Float* byband_flux_net);                                     /* (ncol,nlev,nbnd) */
void net_byband_precalc(const int* ncol, const int* nlev, const int* nbnd,
                        const Float* byband_flux_dn, const Float* byband_flux_up,     /* (ncol,nlev,nbnd) */
                        Float* byband_flux_net);                                      /* (ncol,nlev,nbnd) */

#ifdef __cplusplus
}
#endif
#endif /* RTE_RRTMGP_KERNELS_H */
