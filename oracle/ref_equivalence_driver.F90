! ref_equivalence_driver.F90 -- test infrastructure (our own file; SURVEY.md section 4, "Equivalence/regression").
! The reference's tests/check_equivalence.F90 asserts a list of invariances of the whole frontend (gas optics + solvers)
! but reads its atmosphere and k-distribution from netCDF files of the un-vendored rrtmgp-data repository.  This program
! states the same invariances, with the same tolerances (in units of spacing(), tests/check_equivalence.F90:261-352 LW,
! :386-475 SW), on the synthetic k-distribution / atmosphere streams of oracle/mo_raw_stream.F90 -- through the
! reference's UNCHANGED frontend classes, so every step below ends in the bind(C) kernels:
!   LW: net flux alone / in tandem (2 ulp); vertical flip (4); the problem in two column subsets through get_subset (2);
!       tau halved then incremented by itself (2); increments by transparent 1scl / 2str / nstr media (2); the surface
!       Jacobian leaves the fluxes alone (2) and predicts the fluxes for T_sfc + 1 K (30)
!   SW: vertical flip (4); total solar irradiance halved and the fluxes scaled back (up 10, dn / dir 8); halving / doubling and
!       the transparent increments (up 8, dn / dir 12)
! oracle/build_extern.sh links it twice (HIP library + shim / the reference's CPU kernels); tests/test_extern_frontend.py
! runs both.  Usage: ref_equivalence_driver <k-distribution stream> <atmosphere stream> <comma-separated gases>
! (atmosphere stream as for ref_frontend_driver; block size, repetitions and the use_* options are ignored: one block,
! col_dry from the library's get_layer_number, t_lev given).  Prints one line per check with the worst deviation in
! spacings and ends with "ref_equivalence_driver ok" or stops with code 1.
program ref_equivalence_driver
  use mo_rte_kind,           only: wp, wl
  use mo_rte_config,         only: rte_config_checks
  use mo_gas_concentrations, only: ty_gas_concs
  use mo_gas_optics_rrtmgp,  only: ty_gas_optics_rrtmgp
  use mo_optical_props,      only: ty_optical_props_arry, ty_optical_props_1scl, ty_optical_props_2str, ty_optical_props_nstr
  use mo_source_functions,   only: ty_source_func_lw
  use mo_fluxes,             only: ty_fluxes_broadband
  use mo_rte_lw,             only: rte_lw
  use mo_rte_sw,             only: rte_sw
  use mo_raw_stream,         only: split_names, load_kdist_stream, rd_i1, rd_r1, rd_r2, rd_r3
  implicit none
  character(len=512) :: fk, fatm, gases_arg
  character(len=32), allocatable :: gases(:)
  type(ty_gas_optics_rrtmgp) :: k
  logical :: is_lw, failed
  integer, allocatable :: opts(:)
  integer :: ncol, nlay, ngpt, nbnd, ngas, u, ig, ib
  real(wp), allocatable :: p_lay(:,:), p_lev(:,:), t_lay(:,:), t_lev(:,:), vmr(:,:,:), col_dry(:,:), t_sfc(:), sfc_emis(:), &
                           mu0(:), sfc_alb(:), bsfc(:,:), toa(:,:)
  real(wp), allocatable, target :: ref_up(:,:), ref_dn(:,:), ref_dir(:,:), tst_up(:,:), tst_dn(:,:), tst_dir(:,:), net(:,:), jac(:,:)
  type(ty_gas_concs) :: concs
  type(ty_optical_props_1scl) :: op1
  type(ty_optical_props_2str) :: op2
  type(ty_source_func_lw) :: src
  type(ty_fluxes_broadband) :: fluxes

  call get_command_argument(1, fk); call get_command_argument(2, fatm); call get_command_argument(3, gases_arg)
  call split_names(gases_arg, gases)
  ngas = size(gases)
  call load_kdist_stream(fk, gases, k, is_lw)
  ngpt = k%get_ngpt(); nbnd = k%get_nband()
  open(newunit=u, file=trim(fatm), access='stream', form='unformatted', status='old')
  call rd_i1(u, opts)
  ncol = opts(1); nlay = opts(2)
  call rd_r2(u, p_lay); call rd_r2(u, p_lev); call rd_r2(u, t_lay); call rd_r2(u, t_lev)
  call rd_r3(u, vmr); call rd_r2(u, col_dry)
  if (is_lw) then
    call rd_r1(u, t_sfc); call rd_r1(u, sfc_emis)
  else
    call rd_r1(u, mu0); call rd_r1(u, sfc_alb)
  end if
  close(u)
  if (mod(ncol, 2) /= 0) error stop 'ref_equivalence_driver: an even number of columns, please'
  call rte_config_checks(logical(opts(6) /= 0, wl))
  call stop_on_err(concs%init(gases))
  do ig = 1, ngas
    call stop_on_err(concs%set_vmr(trim(gases(ig)), vmr(:, :, ig)))
  end do
  allocate(bsfc(nbnd, ncol), ref_up(ncol, nlay+1), ref_dn(ncol, nlay+1), tst_up(ncol, nlay+1), tst_dn(ncol, nlay+1), &
           net(ncol, nlay+1))
  do ib = 1, nbnd
    if (is_lw) then
      bsfc(ib, :) = sfc_emis
    else
      bsfc(ib, :) = sfc_alb
    end if
  end do
  failed = .false.

  if (is_lw) then
    allocate(jac(ncol, nlay+1))
    ! optical properties made, released, made again (the reference starts the same way)
    call stop_on_err(op1%alloc_1scl(ncol, nlay, k)); call op1%finalize(); call stop_on_err(op1%alloc_1scl(ncol, nlay, k))
    call stop_on_err(src%alloc(ncol, nlay, k))
    fluxes%flux_up => ref_up; fluxes%flux_dn => ref_dn
    call lw_problem(p_lay, p_lev, t_lay, t_lev, t_sfc, concs, op1, src)
    call stop_on_err(rte_lw(op1, src, bsfc, fluxes))
    if (.not. (minval(ref_up) > 0._wp .and. maxval(ref_dn) > 0._wp)) call fail('default calculation: fluxes are not positive')
    ! ---- net flux alone, then in tandem with up / dn
    nullify(fluxes%flux_up, fluxes%flux_dn)
    fluxes%flux_net => net
    call stop_on_err(rte_lw(op1, src, bsfc, fluxes))
    call check('LW net flux alone', net, ref_dn - ref_up, 2._wp)
    fluxes%flux_up => tst_up; fluxes%flux_dn => tst_dn
    call stop_on_err(rte_lw(op1, src, bsfc, fluxes))
    call check('LW net flux in tandem', net, ref_dn - ref_up, 2._wp)
    call check('LW up in tandem', tst_up, ref_up, 2._wp)
    nullify(fluxes%flux_net)
    ! ---- vertical orientation
    call lw_flipped()
    call check('LW vertical flip, up', tst_up, ref_up, 4._wp); call check('LW vertical flip, dn', tst_dn, ref_dn, 4._wp)
    ! ---- the problem in two halves (optical properties and sources through get_subset)
    call lw_problem(p_lay, p_lev, t_lay, t_lev, t_sfc, concs, op1, src)
    call lw_halves()
    call check('LW column subsets, up', tst_up, ref_up, 2._wp); call check('LW column subsets, dn', tst_dn, ref_dn, 2._wp)
    ! ---- halve, then increment by itself
    op1%tau = 0.5_wp * op1%tau
    call stop_on_err(op1%increment(op1))
    call stop_on_err(rte_lw(op1, src, bsfc, fluxes))
    call check('LW halving / doubling, up', tst_up, ref_up, 2._wp); call check('LW halving / doubling, dn', tst_dn, ref_dn, 2._wp)
    ! ---- increments by transparent media of the three kinds
    call add_transparent(op1, 1); call stop_on_err(rte_lw(op1, src, bsfc, fluxes))
    call check('LW + transparent 1scl, up', tst_up, ref_up, 2._wp); call check('LW + transparent 1scl, dn', tst_dn, ref_dn, 2._wp)
    call add_transparent(op1, 2); call stop_on_err(rte_lw(op1, src, bsfc, fluxes))
    call check('LW + transparent 2str, up', tst_up, ref_up, 2._wp); call check('LW + transparent 2str, dn', tst_dn, ref_dn, 2._wp)
    call add_transparent(op1, 3); call stop_on_err(rte_lw(op1, src, bsfc, fluxes))
    call check('LW + transparent nstr, up', tst_up, ref_up, 2._wp); call check('LW + transparent nstr, dn', tst_dn, ref_dn, 2._wp)
    ! ---- Jacobian with respect to the surface temperature
    call stop_on_err(rte_lw(op1, src, bsfc, fluxes, flux_up_Jac=jac))
    call check('LW fluxes beside the Jacobian, up', tst_up, ref_up, 2._wp)
    call check('LW fluxes beside the Jacobian, dn', tst_dn, ref_dn, 2._wp)
    call lw_problem(p_lay, p_lev, t_lay, t_lev, t_sfc + 1._wp, concs, op1, src)
    call stop_on_err(rte_lw(op1, src, bsfc, fluxes))
    call check('LW Jacobian against T_sfc + 1 K', tst_up, ref_up + jac, 30._wp)
  else
    allocate(ref_dir(ncol, nlay+1), tst_dir(ncol, nlay+1), toa(ncol, ngpt))
    call stop_on_err(op2%alloc_2str(ncol, nlay, k)); call op2%finalize(); call stop_on_err(op2%alloc_2str(ncol, nlay, k))
    fluxes%flux_up => ref_up; fluxes%flux_dn => ref_dn; fluxes%flux_dn_dir => ref_dir
    call stop_on_err(k%gas_optics(p_lay, p_lev, t_lay, concs, op2, toa))
    call stop_on_err(rte_sw(op2, mu0, toa, bsfc, bsfc, fluxes))
    if (.not. (maxval(ref_dn) > 0._wp .and. minval(ref_dir) >= 0._wp)) call fail('default calculation: fluxes are not positive')
    fluxes%flux_up => tst_up; fluxes%flux_dn => tst_dn; fluxes%flux_dn_dir => tst_dir
    ! ---- vertical orientation
    call sw_flipped()
    call check('SW vertical flip, up', tst_up, ref_up, 4._wp); call check('SW vertical flip, dn', tst_dn, ref_dn, 4._wp)
    call check('SW vertical flip, dir', tst_dir, ref_dir, 4._wp)
    ! ---- total solar irradiance halved, fluxes scaled back
    call sw_half_tsi()
    call check('SW TSI scaling, up', tst_up, ref_up, 10._wp); call check('SW TSI scaling, dn', tst_dn, ref_dn, 8._wp)
    call check('SW TSI scaling, dir', tst_dir, ref_dir, 8._wp)
    ! ---- halve, then increment by itself
    call stop_on_err(k%gas_optics(p_lay, p_lev, t_lay, concs, op2, toa))
    op2%tau = 0.5_wp * op2%tau
    call stop_on_err(op2%increment(op2))
    call stop_on_err(rte_sw(op2, mu0, toa, bsfc, bsfc, fluxes))
    call check3('SW halving / doubling')
    ! ---- increments by transparent media
    do ig = 1, 3
      call stop_on_err(k%gas_optics(p_lay, p_lev, t_lay, concs, op2, toa))
      call add_transparent(op2, ig)
      call stop_on_err(rte_sw(op2, mu0, toa, bsfc, bsfc, fluxes))
      call check3('SW + transparent medium of kind ' // achar(48 + ig))
    end do
  end if
  if (failed) error stop 1
  print *, 'ref_equivalence_driver ok'
contains
  subroutine lw_problem(pl, pv, tl, tv, ts, gc, op, sr)
    real(wp), intent(in) :: pl(:,:), pv(:,:), tl(:,:), tv(:,:), ts(:)
    type(ty_gas_concs), intent(in) :: gc
    type(ty_optical_props_1scl), intent(inout) :: op
    type(ty_source_func_lw), intent(inout) :: sr
    call stop_on_err(k%gas_optics(pl, pv, tl, ts, gc, op, sr, tlev=tv))
  end subroutine
  ! the same columns upside down, fluxes turned back
  subroutine flipped_concs(gc)
    type(ty_gas_concs), intent(out) :: gc
    real(wp) :: w(ncol, nlay)
    integer :: i
    call stop_on_err(gc%init(gases))
    do i = 1, ngas
      call stop_on_err(concs%get_vmr(trim(gases(i)), w))   ! (through the class, as the reference's test does)
      call stop_on_err(gc%set_vmr(trim(gases(i)), w(:, nlay:1:-1)))
    end do
  end subroutine
  subroutine lw_flipped()
    type(ty_gas_concs) :: gc
    call flipped_concs(gc)
    call lw_problem(p_lay(:, nlay:1:-1), p_lev(:, nlay+1:1:-1), t_lay(:, nlay:1:-1), t_lev(:, nlay+1:1:-1), t_sfc, gc, op1, src)
    call stop_on_err(rte_lw(op1, src, bsfc, fluxes))
    tst_up = tst_up(:, nlay+1:1:-1); tst_dn = tst_dn(:, nlay+1:1:-1)
  end subroutine
  subroutine sw_flipped()
    type(ty_gas_concs) :: gc
    call flipped_concs(gc)
    call stop_on_err(k%gas_optics(p_lay(:, nlay:1:-1), p_lev(:, nlay+1:1:-1), t_lay(:, nlay:1:-1), gc, op2, toa))
    call stop_on_err(rte_sw(op2, mu0, toa, bsfc, bsfc, fluxes))
    tst_up = tst_up(:, nlay+1:1:-1); tst_dn = tst_dn(:, nlay+1:1:-1); tst_dir = tst_dir(:, nlay+1:1:-1)
  end subroutine
  subroutine lw_halves()
    type(ty_optical_props_1scl) :: part
    type(ty_source_func_lw) :: spart
    type(ty_fluxes_broadband) :: fl
    real(wp), target :: up(ncol/2, nlay+1), dn(ncol/2, nlay+1)
    integer :: i, c0, c1
    call stop_on_err(part%init(op1))
    fl%flux_up => up; fl%flux_dn => dn
    do i = 1, 2
      c0 = (i - 1) * (ncol / 2) + 1; c1 = i * (ncol / 2)
      call stop_on_err(op1%get_subset(c0, ncol / 2, part))
      call stop_on_err(src%get_subset(c0, ncol / 2, spart))
      call stop_on_err(rte_lw(part, spart, bsfc(:, c0:c1), fl))
      tst_up(c0:c1, :) = up; tst_dn(c0:c1, :) = dn
    end do
  end subroutine
  subroutine sw_half_tsi()
    real(wp) :: tsi
    tsi = sum(toa(1, :))
    call stop_on_err(k%set_tsi(0.5_wp * tsi))
    call stop_on_err(k%gas_optics(p_lay, p_lev, t_lay, concs, op2, toa))
    call stop_on_err(rte_sw(op2, mu0, toa, bsfc, bsfc, fluxes))
    tst_up = tst_up / 0.5_wp; tst_dn = tst_dn / 0.5_wp; tst_dir = tst_dir / 0.5_wp
    call stop_on_err(k%set_tsi(tsi))
  end subroutine
  ! increment `op` by a medium without optical depth of kind 1 (1scl), 2 (2str) or 3 (nstr)
  subroutine add_transparent(op, kind)
    class(ty_optical_props_arry), intent(inout) :: op
    integer, intent(in) :: kind
    type(ty_optical_props_1scl) :: a1
    type(ty_optical_props_2str) :: a2
    type(ty_optical_props_nstr) :: a3
    select case (kind)
    case (1)
      call stop_on_err(a1%alloc_1scl(ncol, nlay, op)); a1%tau = 0._wp
      call stop_on_err(a1%increment(op))
    case (2)
      call stop_on_err(a2%alloc_2str(ncol, nlay, op)); a2%tau = 0._wp; a2%ssa = 0._wp; a2%g = 0._wp
      call stop_on_err(a2%increment(op))
    case default
      call stop_on_err(a3%alloc_nstr(3, ncol, nlay, op)); a3%tau = 0._wp; a3%ssa = 0._wp; a3%p = 0._wp
      call stop_on_err(a3%increment(op))
    end select
  end subroutine
  subroutine check3(what)
    character(len=*), intent(in) :: what
    call check(what // ', up', tst_up, ref_up, 8._wp); call check(what // ', dn', tst_dn, ref_dn, 12._wp)
    call check(what // ', dir', tst_dir, ref_dir, 12._wp)
  end subroutine
  ! |a - b| <= tol * spacing(a) everywhere (the reference's allclose, tests/check_equivalence.F90:646-658)
  subroutine check(what, a, b, tol)
    character(len=*), intent(in) :: what
    real(wp), intent(in) :: a(:,:), b(:,:), tol
    real(wp) :: worst
    worst = maxval(abs(a - b) / spacing(a))
    if (worst <= tol) then
      print '(a,a,f8.2,a,f6.1,a)', 'check ', what // ':', worst, ' spacings (limit', tol, ') ok'
    else
      print '(a,a,f12.2,a,f6.1,a)', 'check ', what // ':', worst, ' spacings (limit', tol, ') FAIL'
      failed = .true.
    end if
  end subroutine
  subroutine fail(msg)
    character(len=*), intent(in) :: msg
    print *, 'check ', msg, ' FAIL'
    failed = .true.
  end subroutine
  subroutine stop_on_err(msg)
    character(len=*), intent(in) :: msg
    if (len_trim(msg) > 0) then
      print *, 'ref_equivalence_driver: ', trim(msg)
      error stop 3
    end if
  end subroutine
end program ref_equivalence_driver
