! ref_wrappers.F90 -- C-callable wrappers around the two reference routines that have no C binding
! (rte/kernels/mo_gas_optics_utils.F90: get_layer_mass, get_layer_number), compiled INTO oracle/_ref/librefkernels.so
! next to the reference's own kernel files so that tests can pin the restatements against the reference itself.
! Test infrastructure only; our own file (nothing here is copied from the reference).
subroutine rte_ref_get_layer_number(ncol, nlay, vmr_h2o, plev, col_dry) bind(C, name="rte_ref_get_layer_number")
  use mo_rte_kind,         only: wp
  use mo_gas_optics_utils, only: get_layer_number
  implicit none
  integer, intent(in) :: ncol, nlay
  real(wp), intent(in)  :: vmr_h2o(ncol, nlay), plev(ncol, nlay+1)
  real(wp), intent(out) :: col_dry(ncol, nlay)
  col_dry = get_layer_number(ncol, nlay, vmr_h2o, plev)
end subroutine
subroutine rte_ref_get_layer_mass(ncol, nlay, ngas, vmr, plev, mol_weights, m_dry, layer_mass) bind(C, name="rte_ref_get_layer_mass")
  use mo_rte_kind,         only: wp
  use mo_gas_optics_utils, only: get_layer_mass
  implicit none
  integer, intent(in) :: ncol, nlay, ngas
  real(wp), intent(in)  :: vmr(ngas, ncol, nlay), plev(ncol, nlay+1), mol_weights(ngas), m_dry
  real(wp), intent(out) :: layer_mass(ngas, ncol, nlay)
  call get_layer_mass(ncol, nlay, ngas, vmr, plev, mol_weights, m_dry, layer_mass)
end subroutine
