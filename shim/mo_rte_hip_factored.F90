! mo_rte_hip_factored.F90 -- Fortran binding of the library's FACTORED LW sources (INTEGRATION.md section 4a): what a
! maintainer of the reference would add beside rrtmgp/kernels/api/mo_gas_optics_rrtmgp_kernels.F90 and
! rte/kernels/api/mo_rte_solver_kernels.F90 to use rte_hip_compute_Planck_source_factored /
! rte_hip_lw_solver_noscat_factored / rte_hip_expand_factored_sources (include/rte_hip_ext.h) in place of
! compute_Planck_source (rrtmgp/frontend/mo_gas_optics_rrtmgp.F90:893-928) and lw_solver_noscat
! (rte/frontend/mo_rte_lw.F90:342-365).  Extension symbols take scalars BY VALUE; arrays as in the kernel interface.
! Compiled and run by oracle/build_extern.sh + tests/test_extern_frontend.py (oracle/factored_binding_driver.F90).
module mo_rte_hip_factored
  use iso_c_binding, only: c_int, c_double, c_bool
  use mo_rte_kind,   only: wp, wl
  implicit none
  private
  public :: rte_hip_compute_Planck_source_factored, rte_hip_lw_solver_noscat_factored, rte_hip_expand_factored_sources
  interface
    integer(c_int) function rte_hip_compute_Planck_source_factored(ncol, nlay, nbnd, ngpt, nflav, neta, npres, ntemp, nPlanckTemp, &
        tlay, tlev, tsfc, sfc_lay, fmajor, jeta, tropo, jtemp, jpress, band_lims_gpt, pfracin, temp_ref_min, totplnk_delta, totplnk, &
        gpoint_flavor, sfc_src, pfrac, planck_lay, planck_lev, sfc_source_Jac) bind(C, name="rte_hip_compute_Planck_source_factored")
      import :: c_int, c_double, wp, wl
      integer(c_int), value :: ncol, nlay, nbnd, ngpt, nflav, neta, npres, ntemp, nPlanckTemp, sfc_lay
      real(c_double), value :: temp_ref_min, totplnk_delta          ! doubles in both precisions of the library
      real(wp), dimension(*), intent(in)  :: tlay, tlev, tsfc, fmajor, pfracin, totplnk
      integer(c_int), dimension(*), intent(in) :: jeta, jtemp, jpress, band_lims_gpt, gpoint_flavor
      logical(wl), dimension(*), intent(in) :: tropo
      real(wp), dimension(*), intent(out) :: sfc_src, pfrac, planck_lay, planck_lev, sfc_source_Jac
    end function
    ! 0: done; -2: shape not covered (rescaling, spectral output, > 80 layers) -> rte_hip_expand_factored_sources + the kernel interface
    integer(c_int) function rte_hip_lw_solver_noscat_factored(ncol, nlay, ngpt, nbnd, top_at_1, nmus, Ds, weights, band_lims_gpt, tau, &
        pfrac, planck_lay, planck_lev, sfc_emis, sfc_src, inc_flux, broadband_up, broadband_dn, do_jacobians, sfc_srcJac, flux_upJac) &
        bind(C, name="rte_hip_lw_solver_noscat_factored")
      import :: c_int, wp
      integer(c_int), value :: ncol, nlay, ngpt, nbnd, top_at_1, nmus, do_jacobians
      real(wp), dimension(*), intent(in)  :: Ds, weights, tau, pfrac, planck_lay, planck_lev, sfc_emis, sfc_src, inc_flux, sfc_srcJac
      integer(c_int), dimension(*), intent(in) :: band_lims_gpt
      real(wp), dimension(*), intent(out) :: broadband_up, broadband_dn, flux_upJac
    end function
    integer(c_int) function rte_hip_expand_factored_sources(ncol, nlay, nbnd, ngpt, band_lims_gpt, pfrac, planck_lay, planck_lev, &
        lay_source, lev_source) bind(C, name="rte_hip_expand_factored_sources")
      import :: c_int, wp
      integer(c_int), value :: ncol, nlay, nbnd, ngpt
      integer(c_int), dimension(*), intent(in) :: band_lims_gpt
      real(wp), dimension(*), intent(in)  :: pfrac, planck_lay, planck_lev
      real(wp), dimension(*), intent(out) :: lay_source, lev_source
    end function
  end interface
end module mo_rte_hip_factored
