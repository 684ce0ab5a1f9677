! rfmip_sw_glue_wrapper.F90 (ours; test infrastructure) -- a C-callable frame around the boundary-condition block of the
! reference's RFMIP shortwave driver, examples/rfmip-clear-sky/rrtmgp_rfmip_sw.F90:269-318 and :331-337.  That block is inline
! code of a main program that needs netCDF, so it cannot be linked; oracle/build_rfmip_sw_glue.sh cuts the statements out of
! the reference file WHERE IT LIES (sed by the block's own comment lines, into oracle/_ref/rfmip_build/*.inc, never into
! the repository) and this frame declares the variables the statements use, under the names the program gives them.
! The outputs of the compiled frame on seeded inputs are tests/golden/rfmip_sw_glue.npz (tests/golden/make_rfmip_sw_golden.py).
subroutine ref_rfmip_sw_boundary(block_size, ngpt, nbnd, nlay, nblocks, b, toa_flux, total_solar_irradiance, surface_albedo, &
                                 solar_zenith_angle, usecol_i, def_tsi, sfc_alb_spec, mu0, flux_up, flux_dn) &
    bind(C, name="ref_rfmip_sw_boundary")
  use iso_c_binding, only: c_int
  use mo_rte_kind,   only: wp
  implicit none
  integer(c_int), value :: block_size, ngpt, nbnd, nlay, nblocks, b
  real(wp), intent(inout) :: toa_flux(block_size, ngpt)
  real(wp), intent(in)    :: total_solar_irradiance(block_size, nblocks), surface_albedo(block_size, nblocks), &
                             solar_zenith_angle(block_size, nblocks)
  integer(c_int), intent(in) :: usecol_i(block_size, nblocks)
  real(wp), intent(out)   :: def_tsi(block_size), sfc_alb_spec(nbnd, block_size), mu0(block_size)
  real(wp), intent(inout) :: flux_up(block_size, nlay+1, nblocks), flux_dn(block_size, nlay+1, nblocks)
  logical :: usecol(block_size, nblocks)
  integer :: icol, igpt, ibnd
  include "rfmip_sw_param.inc"     ! the program's own  real(wp), parameter :: deg_to_rad = ...
  usecol = usecol_i /= 0
  include "rfmip_sw_block.inc"     ! def_tsi, toa_flux renormalised, sfc_alb_spec, mu0
  include "rfmip_sw_mask.inc"      ! fluxes of night columns zeroed
end subroutine ref_rfmip_sw_boundary
