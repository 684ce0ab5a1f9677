! ref_load_driver.F90 -- test infrastructure (our own file): feeds a RAW k-distribution table, written by
! tests/test_kdist_load.py as a flat stream of records, to the REFERENCE's own ty_gas_optics_rrtmgp%load
! (rrtmgp/frontend/mo_gas_optics_rrtmgp.F90:938-1145) and dumps the arrays the loaded object holds, so that the
! numpy restatement of the load-time reductions (rte-rrtmgp_amd/kdist_load.py) is pinned against the reference itself.
! Built by oracle/build_load_check.sh into oracle/_ref/bin/ref_load_driver (needs /root/reference + flang).
!
! Input stream: see oracle/mo_raw_stream.F90.
! Usage: ref_load_driver <input file> <output file> <comma-separated available gases>
program ref_load_driver
  use mo_rte_kind,           only: wp, wl
  use mo_gas_concentrations, only: ty_gas_concs
  use mo_gas_optics_rrtmgp,  only: ty_gas_optics_rrtmgp
  use mo_raw_stream,         only: split_names, load_kdist_stream
  implicit none
  character(len=512) :: fin, fout, gases_arg
  character(len=32), allocatable :: avail(:)
  type(ty_gas_optics_rrtmgp) :: k
  logical :: is_lw

  call get_command_argument(1, fin); call get_command_argument(2, fout); call get_command_argument(3, gases_arg)
  call split_names(gases_arg, avail)
  call load_kdist_stream(fin, avail, k, is_lw)   ! oracle/mo_raw_stream.F90: the reference's own k%load on the raw table

  ! The loaded object's arrays are PRIVATE components; what matters is what reaches the kernels.  One call of
  ! gas_optics on a 3-column, 4-layer dummy atmosphere: the recorder library (oracle/abi_recorder.c, linked in front of
  ! the reference kernels) writes every table argument of the kernel calls to $RTE_ABI_RECORD.  The solar source comes
  ! back through toa_src and is written to <output file>.
  call record(fout)
  print *, 'ref_load_driver ok'
contains
  subroutine record(path)
    use mo_optical_props,    only: ty_optical_props_1scl, ty_optical_props_2str
    use mo_source_functions, only: ty_source_func_lw
    character(len=*), intent(in) :: path
    integer, parameter :: nc = 3, nl = 4
    type(ty_gas_concs) :: concs
    type(ty_optical_props_1scl) :: op1
    type(ty_optical_props_2str) :: op2
    type(ty_source_func_lw) :: src
    real(wp) :: p_lay(nc, nl), p_lev(nc, nl+1), t_lay(nc, nl), t_sfc(nc)
    real(wp), allocatable :: toa(:,:)
    integer :: ig, il, u
    character(len=128) :: e
    e = concs%init(avail)
    do ig = 1, size(avail)
      e = concs%set_vmr(trim(avail(ig)), 1.e-4_wp * ig)
      if (e /= '') error stop trim(e)
    end do
    do il = 1, nl + 1
      p_lev(:, il) = 90000._wp - 20000._wp * (il - 1)
    end do
    p_lay = 0.5_wp * (p_lev(:, 1:nl) + p_lev(:, 2:nl+1))
    t_lay = 250._wp; t_sfc = 260._wp
    if (is_lw) then
      e = op1%alloc_1scl(nc, nl, k)
      e = src%alloc(nc, nl, k)
      e = k%gas_optics(p_lay, p_lev, t_lay, t_sfc, concs, op1, src)
    else
      e = op2%alloc_2str(nc, nl, k)
      allocate(toa(nc, k%get_ngpt()))
      e = k%gas_optics(p_lay, p_lev, t_lay, concs, op2, toa)
      if (e == '') then
        open(newunit=u, file=trim(path), access='stream', form='unformatted', status='replace')
        write(u) toa(1, :)
        close(u)
      end if
    end if
    if (e /= '') then
      print *, 'gas_optics failed: ', trim(e)
      error stop 2
    end if
  end subroutine
end program ref_load_driver
