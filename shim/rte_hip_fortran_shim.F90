! rte_hip_fortran_shim.F90 -- the two externals of the reference's extern-mode kernel interface that have NO C binding.
!
! rte/kernels/api/mo_gas_optics_utils.F90:38-66 declares get_layer_mass (a subroutine) and get_layer_number (an
! array-valued function) as plain Fortran externals, so an unchanged frontend built with RTE_KERNEL_MODE=extern
! references the Fortran-mangled symbols get_layer_mass_ / get_layer_number_ (rrtmgp/frontend/mo_gas_optics_rrtmgp.F90
! uses get_layer_number for col_dry).  A C library cannot export an array-valued Fortran function portably; this file
! is compiled with the frontend and forwards both to the HIP library's extension entry points (csrc/glue.hip).
! m_dry and grav are run-time settable in the reference (mo_gas_optics_constants: init_constants), so their current
! values are passed along.
subroutine get_layer_mass(ncol, nlay, ngas, vmr, plev, mol_weights, m_dry, layer_mass)
  use iso_c_binding,           only: c_int, c_double
  use mo_rte_kind,             only: wp
  use mo_gas_optics_constants, only: grav
  implicit none
  integer, intent(in)                                  :: ncol, nlay, ngas
  real(wp), dimension(ngas, ncol, nlay  ), intent(in ) :: vmr
  real(wp), dimension(      ncol, nlay+1), intent(in ) :: plev
  real(wp), dimension(ngas),               intent(in ) :: mol_weights
  real(wp),                                intent(in ) :: m_dry
  real(wp), dimension(ngas, ncol, nlay),   intent(out) :: layer_mass
  interface
    function rte_hip_get_layer_mass(ncol, nlay, ngas, vmr, plev, mol_weights, m_dry, grav, layer_mass) &
        bind(C, name="rte_hip_get_layer_mass") result(rc)
      use iso_c_binding, only: c_int, c_double
      use mo_rte_kind,   only: wp
      integer(c_int), value :: ncol, nlay, ngas
      real(wp), intent(in)  :: vmr(*), plev(*), mol_weights(*)
      real(c_double), value :: m_dry, grav
      real(wp), intent(out) :: layer_mass(*)
      integer(c_int)        :: rc
    end function
  end interface
  integer(c_int) :: rc
  rc = rte_hip_get_layer_mass(int(ncol, c_int), int(nlay, c_int), int(ngas, c_int), vmr, plev, mol_weights, &
                              real(m_dry, c_double), real(grav, c_double), layer_mass)
end subroutine get_layer_mass

function get_layer_number(ncol, nlay, vmr_h2o, plev) result(col_dry)
  use iso_c_binding,           only: c_int, c_double
  use mo_rte_kind,             only: wp
  use mo_gas_optics_constants, only: m_dry, grav
  implicit none
  integer, intent(in) :: ncol, nlay
  real(wp), dimension(ncol, nlay  ), intent(in) :: vmr_h2o
  real(wp), dimension(ncol, nlay+1), intent(in) :: plev
  real(wp), dimension(ncol, nlay) :: col_dry
  interface
    function rte_hip_get_layer_number(ncol, nlay, vmr_h2o, plev, m_dry, grav, col_dry) &
        bind(C, name="rte_hip_get_layer_number") result(rc)
      use iso_c_binding, only: c_int, c_double
      use mo_rte_kind,   only: wp
      integer(c_int), value :: ncol, nlay
      real(wp), intent(in)  :: vmr_h2o(*), plev(*)
      real(c_double), value :: m_dry, grav
      real(wp), intent(out) :: col_dry(*)
      integer(c_int)        :: rc
    end function
  end interface
  integer(c_int) :: rc
  rc = rte_hip_get_layer_number(int(ncol, c_int), int(nlay, c_int), vmr_h2o, plev, real(m_dry, c_double), &
                                real(grav, c_double), col_dry)
end function get_layer_number
