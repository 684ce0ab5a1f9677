! mo_raw_stream.F90 -- test infrastructure (our own file): reader of the flat record stream the Python tests write
! (tests/stream_io.py), and a loader that feeds a RAW k-distribution table read from such a stream to the REFERENCE's own
! ty_gas_optics_rrtmgp%load (rrtmgp/frontend/mo_gas_optics_rrtmgp.F90:938-1145).  Used by oracle/ref_load_driver.F90 and
! oracle/ref_frontend_driver.F90.
!
! Stream (unformatted stream, little endian): for every item a tag (character(32)), rank (int32), dims (rank x int32)
! then the payload: float64 / int32 / logical-as-int32, or for string tables n x character(32).
module mo_raw_stream
  use mo_rte_kind,           only: wp, wl
  use mo_gas_concentrations, only: ty_gas_concs
  use mo_gas_optics_rrtmgp,  only: ty_gas_optics_rrtmgp
  implicit none
  private
  public :: split_names, load_kdist_stream, rd_str, rd_i0, rd_r0, rd_i1, rd_l1, rd_i2, rd_i3, rd_r1, rd_r2, rd_r3, rd_r4
contains
  ! comma-separated list -> array of names
  subroutine split_names(arg, names)
    character(len=*), intent(in) :: arg
    character(len=32), allocatable, intent(out) :: names(:)
    integer :: i, n, p0
    n = 1
    do i = 1, len_trim(arg)
      if (arg(i:i) == ',') n = n + 1
    end do
    allocate(names(n))
    p0 = 1; n = 0
    do i = 1, len_trim(arg) + 1
      if (i > len_trim(arg)) then
        n = n + 1; names(n) = arg(p0:i-1)
      else if (arg(i:i) == ',') then
        n = n + 1; names(n) = arg(p0:i-1); p0 = i + 1
      end if
    end do
  end subroutine

  ! raw table from `path` -> k%load(...) with the gases in `avail`
  subroutine load_kdist_stream(path, avail, k, is_lw)
    character(len=*), intent(in) :: path
    character(len=32), intent(in) :: avail(:)
    type(ty_gas_optics_rrtmgp), intent(inout) :: k
    logical, intent(out) :: is_lw
    type(ty_gas_concs) :: available
    character(len=128) :: err
    integer :: uin, i
    character(len=32), allocatable :: gas_names(:), gas_minor(:), identifier_minor(:), minor_gases_lower(:), &
                                      minor_gases_upper(:), scaling_gas_lower(:), scaling_gas_upper(:)
    integer,  allocatable :: key_species(:,:,:), band2gpt(:,:), minor_limits_gpt_lower(:,:), minor_limits_gpt_upper(:,:), &
                             kminor_start_lower(:), kminor_start_upper(:)
    logical(wl), allocatable :: sd_lower(:), sd_upper(:), sc_lower(:), sc_upper(:)
    real(wp), allocatable :: band_lims(:,:), press_ref(:), temp_ref(:), vmr_ref(:,:,:), kmajor(:,:,:,:), kminor_lower(:,:,:), &
                             kminor_upper(:,:,:), totplnk(:,:), planck_frac(:,:,:,:), optimal_angle_fit(:,:), &
                             rayl_lower(:,:,:), rayl_upper(:,:,:), solar_quiet(:), solar_facular(:), solar_sunspot(:)
    real(wp) :: press_ref_trop, temp_ref_p, temp_ref_t, tsi_default, mg_default, sb_default

    err = available%init(avail)
    if (err /= '') error stop 'load_kdist_stream: gas list'
    open(newunit=uin, file=trim(path), access='stream', form='unformatted', status='old')
    call rd_str(uin, gas_names);  call rd_i3(uin, key_species); call rd_i2(uin, band2gpt); call rd_r2(uin, band_lims)
    call rd_r1(uin, press_ref); call rd_r1(uin, temp_ref)
    call rd_r0(uin, press_ref_trop); call rd_r0(uin, temp_ref_p); call rd_r0(uin, temp_ref_t)
    call rd_r3(uin, vmr_ref); call rd_r4(uin, kmajor); call rd_r3(uin, kminor_lower); call rd_r3(uin, kminor_upper)
    call rd_str(uin, gas_minor); call rd_str(uin, identifier_minor)
    call rd_str(uin, minor_gases_lower); call rd_str(uin, minor_gases_upper)
    call rd_i2(uin, minor_limits_gpt_lower); call rd_i2(uin, minor_limits_gpt_upper)
    call rd_l1(uin, sd_lower); call rd_l1(uin, sd_upper); call rd_str(uin, scaling_gas_lower); call rd_str(uin, scaling_gas_upper)
    call rd_l1(uin, sc_lower); call rd_l1(uin, sc_upper); call rd_i1(uin, kminor_start_lower); call rd_i1(uin, kminor_start_upper)
    call rd_i0(uin, i); is_lw = i /= 0
    if (is_lw) then
      call rd_r2(uin, totplnk); call rd_r4(uin, planck_frac); call rd_r2(uin, optimal_angle_fit)
      err = k%load(available, gas_names, key_species, band2gpt, band_lims, press_ref, press_ref_trop, temp_ref, temp_ref_p, &
                   temp_ref_t, vmr_ref, kmajor, kminor_lower, kminor_upper, gas_minor, identifier_minor, minor_gases_lower, &
                   minor_gases_upper, minor_limits_gpt_lower, minor_limits_gpt_upper, sd_lower, sd_upper, scaling_gas_lower, &
                   scaling_gas_upper, sc_lower, sc_upper, kminor_start_lower, kminor_start_upper, totplnk, planck_frac, &
                   rayl_lower, rayl_upper, optimal_angle_fit)
    else
      call rd_r3(uin, rayl_lower); call rd_r3(uin, rayl_upper)
      call rd_r1(uin, solar_quiet); call rd_r1(uin, solar_facular); call rd_r1(uin, solar_sunspot)
      call rd_r0(uin, tsi_default); call rd_r0(uin, mg_default); call rd_r0(uin, sb_default)
      err = k%load(available, gas_names, key_species, band2gpt, band_lims, press_ref, press_ref_trop, temp_ref, temp_ref_p, &
                   temp_ref_t, vmr_ref, kmajor, kminor_lower, kminor_upper, gas_minor, identifier_minor, minor_gases_lower, &
                   minor_gases_upper, minor_limits_gpt_lower, minor_limits_gpt_upper, sd_lower, sd_upper, scaling_gas_lower, &
                   scaling_gas_upper, sc_lower, sc_upper, kminor_start_lower, kminor_start_upper, solar_quiet, solar_facular, &
                   solar_sunspot, tsi_default, mg_default, sb_default, rayl_lower, rayl_upper)
    end if
    close(uin)
    if (err /= '') then
      print *, 'load failed: ', trim(err)
      error stop 1
    end if
  end subroutine

  subroutine hdr(uin, rank, dims)
    integer, intent(in) :: uin
    integer, intent(out) :: rank, dims(4)
    character(len=32) :: tag
    dims = 1
    read(uin) tag, rank
    if (rank > 0) read(uin) dims(1:rank)
  end subroutine
  subroutine rd_str(uin, a)
    integer, intent(in) :: uin
    character(len=32), allocatable, intent(out) :: a(:)
    integer :: r, d(4)
    call hdr(uin, r, d); allocate(a(d(1))); if (d(1) > 0) read(uin) a
  end subroutine
  subroutine rd_i0(uin, a)
    integer, intent(in) :: uin
    integer, intent(out) :: a
    integer :: r, d(4)
    call hdr(uin, r, d); read(uin) a
  end subroutine
  subroutine rd_r0(uin, a)
    integer, intent(in) :: uin
    real(wp), intent(out) :: a
    integer :: r, d(4)
    real(8) :: t   ! the streams hold float64 whatever the working precision (RTE_USE_SP builds convert here)
    call hdr(uin, r, d); read(uin) t; a = real(t, wp)
  end subroutine
  subroutine rd_i1(uin, a)
    integer, intent(in) :: uin
    integer, allocatable, intent(out) :: a(:)
    integer :: r, d(4)
    call hdr(uin, r, d); allocate(a(d(1))); if (size(a) > 0) read(uin) a
  end subroutine
  subroutine rd_l1(uin, a)
    integer, intent(in) :: uin
    logical(wl), allocatable, intent(out) :: a(:)
    integer, allocatable :: t(:)
    call rd_i1(uin, t); allocate(a(size(t))); a = t /= 0
  end subroutine
  subroutine rd_i2(uin, a)
    integer, intent(in) :: uin
    integer, allocatable, intent(out) :: a(:,:)
    integer :: r, d(4)
    call hdr(uin, r, d); allocate(a(d(1), d(2))); if (size(a) > 0) read(uin) a
  end subroutine
  subroutine rd_i3(uin, a)
    integer, intent(in) :: uin
    integer, allocatable, intent(out) :: a(:,:,:)
    integer :: r, d(4)
    call hdr(uin, r, d); allocate(a(d(1), d(2), d(3))); if (size(a) > 0) read(uin) a
  end subroutine
  subroutine rd_r1(uin, a)
    integer, intent(in) :: uin
    real(wp), allocatable, intent(out) :: a(:)
    real(8), allocatable :: t(:)
    integer :: r, d(4)
    call hdr(uin, r, d); allocate(a(d(1)), t(d(1)))
    if (size(a) > 0) then
      read(uin) t; a = real(t, wp)
    end if
  end subroutine
  subroutine rd_r2(uin, a)
    integer, intent(in) :: uin
    real(wp), allocatable, intent(out) :: a(:,:)
    real(8), allocatable :: t(:,:)
    integer :: r, d(4)
    call hdr(uin, r, d); allocate(a(d(1), d(2)), t(d(1), d(2)))
    if (size(a) > 0) then
      read(uin) t; a = real(t, wp)
    end if
  end subroutine
  subroutine rd_r3(uin, a)
    integer, intent(in) :: uin
    real(wp), allocatable, intent(out) :: a(:,:,:)
    real(8), allocatable :: t(:,:,:)
    integer :: r, d(4)
    call hdr(uin, r, d); allocate(a(d(1), d(2), d(3)), t(d(1), d(2), d(3)))
    if (size(a) > 0) then
      read(uin) t; a = real(t, wp)
    end if
  end subroutine
  subroutine rd_r4(uin, a)
    integer, intent(in) :: uin
    real(wp), allocatable, intent(out) :: a(:,:,:,:)
    real(8), allocatable :: t(:,:,:,:)
    integer :: r, d(4)
    call hdr(uin, r, d); allocate(a(d(1), d(2), d(3), d(4)), t(d(1), d(2), d(3), d(4)))
    if (size(a) > 0) then
      read(uin) t; a = real(t, wp)
    end if
  end subroutine
end module mo_raw_stream
