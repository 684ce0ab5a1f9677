! ref_frontend_driver.F90 -- test infrastructure (our own file; SURVEY.md section 8 row f4): drives the REFERENCE's
! UNCHANGED Fortran frontend end to end on host arrays, the way the RFMIP example drivers do
! (examples/rfmip-clear-sky/rrtmgp_rfmip_lw.F90:247-281, rrtmgp_rfmip_sw.F90:262-336) but without netCDF:
!
!   raw k-distribution stream -> ty_gas_optics_rrtmgp%load          (rrtmgp/frontend/mo_gas_optics_rrtmgp.F90:938-1145)
!   per block of columns:        k%gas_optics (LW :220-331, SW :337-414) -> rte_lw / rte_sw with ty_fluxes_broadband
!                                (rte/frontend/mo_rte_lw.F90:79-473, rte/frontend/mo_rte_sw.F90:56-394)
!
! oracle/build_extern.sh links this one object file twice: against librte_rrtmgp_hip.so + the Fortran shim (the HIP
! path behind the reference's bind(C) kernel interface) and against the reference's own CPU kernels
! (oracle/_ref/librefkernels.so).  tests/test_extern_frontend.py runs both on the same seeded atmosphere and compares
! the flux files.  The program is the same in both links: library options (host-mirror mode) are switched through
! environment variables of the HIP library, never through symbols the reference kernels lack.
!
! Usage: ref_frontend_driver <k-distribution stream> <atmosphere stream> <output file> <comma-separated gases> [cloud stream]
! With a cloud stream the block loop is the all-sky example's (examples/all-sky/rrtmgp_allsky.F90:362-404): the reference's
! ty_cloud_optics_rrtmgp%load on the given look-up tables, then per block cloud_optics%cloud_optics (by band) ->
! k%gas_optics -> [SW: clouds%delta_scale] -> clouds%increment(gas optical properties) -> rte_lw / rte_sw.
! Cloud stream: radliq_lwr, radliq_upr, diamice_lwr, diamice_upr (scalars); extliq, ssaliq, asyliq (nsize_liq, nbnd);
!   extice, ssaice, asyice (nsize_ice, nbnd, nrghice); ice roughness to select (int); lwp, iwp, rel, dei (ncol, nlay).
! Atmosphere stream (records as in oracle/mo_raw_stream.F90):
!   opts  int(8 or 9): ncol, nlay, block size, use_col_dry, use_tlev, checks on/off, repetitions of the block loop, n_gauss_angles
!         [, variant: 0 default; 1 LW optimal transport angles (k%compute_optimal_angles -> rte_lw(lw_Ds=)); 2 fluxes by band
!         (ty_fluxes_byband: the solvers' spectral output + rte_sum_byband; the bands are summed for the output file);
!         3 LW with two-stream clouds (needs the cloud stream): rte_lw's default for 2str properties, the Tang rescaling;
!         4 the same through lw_solver_2stream (use_2stream=.true.); 5 a diffuse flux incident at the top (LW inc_flux, SW
!         inc_flux_dif: 0.1 per g-point)]  -- the configurations of tests/check_variants.F90
!   p_lay, p_lev, t_lay, t_lev (ncol, nlay[+1]); vmr(ncol, nlay, ngases) in the order of <gases>; col_dry(ncol, nlay);
!   LW: t_sfc(ncol), sfc_emis(ncol);  SW: mu0(ncol), sfc_alb(ncol)
! Output (stream): flux_up, flux_dn (ncol, nlay+1) [, flux_dn_dir for SW], float64, column fastest.
! Built a second time with -fopenmp (ref_frontend_driver_omp[_cpuref]): the blocks are then spread over OMP_NUM_THREADS host
! threads; with RTE_HIP_THREAD_CONTEXTS=1 the HIP library gives every thread a context (stream, arena) of its own.
program ref_frontend_driver
  use mo_rte_kind,           only: wp, wl
  use mo_rte_config,         only: rte_config_checks
  use mo_gas_concentrations, only: ty_gas_concs
  use mo_gas_optics_rrtmgp,  only: ty_gas_optics_rrtmgp
  use mo_optical_props,      only: ty_optical_props_1scl, ty_optical_props_2str
  use mo_source_functions,   only: ty_source_func_lw
  use mo_fluxes,             only: ty_fluxes_broadband
  use mo_fluxes_byband,      only: ty_fluxes_byband
  use mo_rte_lw,             only: rte_lw
  use mo_rte_sw,             only: rte_sw
  use mo_cloud_optics_rrtmgp,only: ty_cloud_optics_rrtmgp
  use mo_raw_stream,         only: split_names, load_kdist_stream, rd_i0, rd_r0, rd_i1, rd_r1, rd_r2, rd_r3
  !$ use omp_lib
  implicit none
  character(len=512) :: fk, fatm, fout, gases_arg, fcld
  logical :: with_clouds
  type(ty_cloud_optics_rrtmgp) :: cloud_spec
  real(wp) :: radliq_lwr, radliq_upr, diamice_lwr, diamice_upr
  real(wp), allocatable :: extliq(:,:), ssaliq(:,:), asyliq(:,:), extice(:,:,:), ssaice(:,:,:), asyice(:,:,:)
  real(wp), allocatable :: lwp(:,:), iwp(:,:), rel(:,:), dei(:,:), blwp(:,:,:), biwp(:,:,:), brel(:,:,:), bdei(:,:,:)
  integer :: irgh, variant
  character(len=32), allocatable :: gases(:)
  type(ty_gas_optics_rrtmgp) :: k
  logical :: is_lw
  integer, allocatable :: opts(:)
  integer :: ncol, nlay, bs, nblocks, nrep, n_ang, ngpt, nbnd, ngas
  logical :: use_col_dry, use_tlev, checks, timing_lines, band_emis
  character(len=8) :: envv
  character(len=16) :: envpad
  integer :: pad_mb, ios
  real(wp), allocatable :: p_lay(:,:), p_lev(:,:), t_lay(:,:), t_lev(:,:), vmr(:,:,:), col_dry(:,:), t_sfc(:), sfc_emis(:), &
                           mu0(:), sfc_alb(:)
  real(wp), allocatable, target :: flux_up(:,:), flux_dn(:,:), flux_dir(:,:)
  ! blocked as in the RFMIP drivers (examples/rfmip-clear-sky/rrtmgp_rfmip_lw.F90:113-120: (block_size, nlay, nblocks)):
  ! a block is a contiguous slab that is passed to the frontend as it lies, no copies inside the block loop
  real(wp), allocatable :: bp_lay(:,:,:), bp_lev(:,:,:), bt_lay(:,:,:), bt_lev(:,:,:), bcol_dry(:,:,:), bsfc(:,:,:), bt_sfc(:,:), bmu0(:,:)
  real(wp), allocatable, target :: bup(:,:,:), bdn(:,:,:), bdir(:,:,:)
  type(ty_gas_concs), allocatable :: concs(:)
  integer :: u, b, c0, c1, ig, irep, nth
  integer(8) :: t0, t1, rate
  real(8) :: secs, best

  call get_command_argument(1, fk); call get_command_argument(2, fatm)
  call get_command_argument(3, fout); call get_command_argument(4, gases_arg)
  with_clouds = command_argument_count() >= 5
  if (with_clouds) call get_command_argument(5, fcld)
  call split_names(gases_arg, gases)
  ngas = size(gases)
  ! REF_DRIVER_DEEP_SETUP_MB (the OpenMP-offload build, oracle/build_extern_offload.sh): run the set-up calls that many
  ! megabytes further down the stack.  flang maps 40-byte descriptor temporaries of the classes' allocatable components with
  ! `target enter data` and never unmaps them (rrtmgp/frontend/mo_gas_optics_rrtmgp.F90:956,1171;
  ! gas-optics-template/mo_gas_concentrations.F90:89,269); left where the block loop's automatic arrays come to lie, the next
  ! `map` of such an array overlaps a stale entry and libomptarget stops ("explicit extension not allowed").
  pad_mb = 0
  call get_environment_variable('REF_DRIVER_DEEP_SETUP_MB', envpad)
  if (len_trim(envpad) > 0) read(envpad, *, iostat=ios) pad_mb
  call with_stack_pad(pad_mb, 1)
  ngpt = k%get_ngpt(); nbnd = k%get_nband()

  open(newunit=u, file=trim(fatm), access='stream', form='unformatted', status='old')
  call rd_i1(u, opts)
  ncol = opts(1); nlay = opts(2); bs = opts(3); use_col_dry = opts(4) /= 0; use_tlev = opts(5) /= 0
  checks = opts(6) /= 0; nrep = max(1, opts(7)); n_ang = max(1, opts(8))
  variant = 0
  if (size(opts) >= 9) variant = opts(9)
  call rd_r2(u, p_lay); call rd_r2(u, p_lev); call rd_r2(u, t_lay); call rd_r2(u, t_lev)
  call rd_r3(u, vmr); call rd_r2(u, col_dry)
  if (is_lw) then
    call rd_r1(u, t_sfc); call rd_r1(u, sfc_emis)
  else
    call rd_r1(u, mu0); call rd_r1(u, sfc_alb)
  end if
  close(u)
  if (with_clouds) then
    open(newunit=u, file=trim(fcld), access='stream', form='unformatted', status='old')
    call rd_r0(u, radliq_lwr); call rd_r0(u, radliq_upr); call rd_r0(u, diamice_lwr); call rd_r0(u, diamice_upr)
    call rd_r2(u, extliq); call rd_r2(u, ssaliq); call rd_r2(u, asyliq)
    call rd_r3(u, extice); call rd_r3(u, ssaice); call rd_r3(u, asyice)
    call rd_i0(u, irgh)
    call rd_r2(u, lwp); call rd_r2(u, iwp); call rd_r2(u, rel); call rd_r2(u, dei)
    close(u)
    ! by-band cloud optics (no band_lims_gpt: one "g-point" per band), examples/all-sky/mo_load_cloud_coefficients.F90
    call stop_on_err(cloud_spec%load(k%get_band_lims_wavenumber(), radliq_lwr, radliq_upr, diamice_lwr, diamice_upr, &
                                     extliq, ssaliq, asyliq, extice, ssaice, asyice))
    call stop_on_err(cloud_spec%set_ice_roughness(irgh))
  end if
  if (mod(ncol, bs) /= 0) error stop 'ref_frontend_driver: ncol is not a multiple of the block size'
  call get_environment_variable('REF_DRIVER_BAND_EMIS', envv)
  band_emis = len_trim(envv) > 0 .and. envv(1:1) /= '0'
  nblocks = ncol / bs
  ! rte/frontend/mo_rte_config.F90:25-49 (the all-sky example switches the checks off after its first pass,
  ! examples/all-sky/rrtmgp_allsky.F90:334)
  call rte_config_checks(logical(checks, wl))

  ! gas concentrations per block (examples/rfmip-clear-sky: read_and_block_gases_ty)
  allocate(concs(nblocks))
  call with_stack_pad(pad_mb, 2)

  allocate(flux_up(ncol, nlay+1), flux_dn(ncol, nlay+1))
  if (.not. is_lw) allocate(flux_dir(ncol, nlay+1))
  allocate(bp_lay(bs, nlay, nblocks), bp_lev(bs, nlay+1, nblocks), bt_lay(bs, nlay, nblocks), bt_lev(bs, nlay+1, nblocks), &
           bcol_dry(bs, nlay, nblocks), bsfc(nbnd, bs, nblocks), bup(bs, nlay+1, nblocks), bdn(bs, nlay+1, nblocks))
  if (is_lw) then
    allocate(bt_sfc(bs, nblocks))
  else
    allocate(bmu0(bs, nblocks), bdir(bs, nlay+1, nblocks))
  end if
  if (with_clouds) then
    allocate(blwp(bs, nlay, nblocks), biwp(bs, nlay, nblocks), brel(bs, nlay, nblocks), bdei(bs, nlay, nblocks))
    do b = 1, nblocks
      c0 = (b - 1) * bs + 1; c1 = b * bs
      blwp(:, :, b) = lwp(c0:c1, :); biwp(:, :, b) = iwp(c0:c1, :); brel(:, :, b) = rel(c0:c1, :); bdei(:, :, b) = dei(c0:c1, :)
    end do
  end if
  do b = 1, nblocks
    c0 = (b - 1) * bs + 1; c1 = b * bs
    bp_lay(:, :, b) = p_lay(c0:c1, :); bp_lev(:, :, b) = p_lev(c0:c1, :); bt_lay(:, :, b) = t_lay(c0:c1, :)
    bt_lev(:, :, b) = t_lev(c0:c1, :); bcol_dry(:, :, b) = col_dry(c0:c1, :)
    do ig = 1, nbnd
      if (is_lw) then
        bsfc(ig, :, b) = sfc_emis(c0:c1)
        ! (REF_DRIVER_BAND_EMIS: an emissivity that depends on the band, so that what rte_lw's expand_and_transpose hands to the
        !  solver is not a plain broadcast -- tests/golden/make_glue_golden.py)
        if (band_emis) bsfc(ig, :, b) = sfc_emis(c0:c1) * (1._wp - 0.00390625_wp * real(ig, wp))   ! (2**-8: the factor is exact in binary)
      else
        bsfc(ig, :, b) = sfc_alb(c0:c1)
      end if
    end do
    if (is_lw) then
      bt_sfc(:, b) = t_sfc(c0:c1)
    else
      bmu0(:, b) = mu0(c0:c1)
    end if
  end do

  ! The block loop of the RFMIP drivers; with OpenMP (the same source built with -fopenmp: ref_frontend_driver_omp) the
  ! blocks are dealt round-robin to the threads, each with its own optical-property / source / flux objects -- concurrent
  ! calls on distinct buffers, the use the reference intends (examples/all-sky/rrtmgp_allsky.F90:331).
  nth = 1
  !$ nth = omp_get_max_threads()
  best = huge(best)
  call system_clock(count_rate=rate)
  call get_environment_variable('REF_DRIVER_TIMING', envv)
  timing_lines = len_trim(envv) > 0 .and. envv(1:1) /= '0'
  do irep = 1, nrep
    call system_clock(t0)
    !$omp parallel default(shared)
    call worker()
    !$omp end parallel
    call system_clock(t1)
    secs = real(t1 - t0, 8) / real(rate, 8)
    best = min(best, secs)
    print '(a,i0,a,f10.4,a,f12.1,a,i0,a)', 'pass ', irep, ': ', secs, ' s, ', real(ncol, 8) / secs, ' columns/s (', nth, ' host threads)'
  end do
  print '(a,f12.1)', 'best columns/s: ', real(ncol, 8) / best

  do b = 1, nblocks
    c0 = (b - 1) * bs + 1; c1 = b * bs
    flux_up(c0:c1, :) = bup(:, :, b); flux_dn(c0:c1, :) = bdn(:, :, b)
    if (.not. is_lw) flux_dir(c0:c1, :) = bdir(:, :, b)
  end do
  open(newunit=u, file=trim(fout), access='stream', form='unformatted', status='replace')
  write(u) real(flux_up, 8); write(u) real(flux_dn, 8)   ! float64 whatever the working precision
  if (.not. is_lw) write(u) real(flux_dir, 8)
  close(u)
  print *, 'ref_frontend_driver ok'
contains
  ! one thread's share of the blocks: its own work arrays and frontend objects (all local, i.e. private)
  subroutine worker()
    real(wp), allocatable :: toa(:,:)
    type(ty_optical_props_1scl) :: op1
    type(ty_optical_props_2str) :: op2
    type(ty_source_func_lw) :: src
    type(ty_fluxes_broadband) :: fluxes
    type(ty_optical_props_1scl) :: cld1
    type(ty_optical_props_2str) :: cld2
    type(ty_fluxes_byband) :: bfl
    real(wp), allocatable, target :: bbu(:,:,:), bbd(:,:,:), bbdir(:,:,:)
    real(wp), allocatable :: ds(:,:), incf(:,:)
    integer :: b, tid, nthr
    integer(8) :: tb, tc, td, tick_go, tick_rte
    character(len=128) :: e
    tick_go = 0; tick_rte = 0
    tid = 0; nthr = 1
    !$ tid = omp_get_thread_num()
    !$ nthr = omp_get_num_threads()
    if (is_lw) then
      call stop_on_err(op1%alloc_1scl(bs, nlay, k))
      call stop_on_err(src%alloc(bs, nlay, k))
      if (with_clouds) call stop_on_err(cld1%alloc_1scl(bs, nlay, cloud_spec))
    else
      if (with_clouds) call stop_on_err(cld2%alloc_2str(bs, nlay, cloud_spec))
      allocate(toa(bs, ngpt))
      call stop_on_err(op2%alloc_2str(bs, nlay, k))
    end if
    if (variant == 1) allocate(ds(bs, ngpt))
    if (variant == 5) then
      allocate(incf(bs, ngpt)); incf = 0.1_wp
    end if
    if (variant == 2) then
      allocate(bbu(bs, nlay+1, nbnd), bbd(bs, nlay+1, nbnd))
      bfl%bnd_flux_up => bbu; bfl%bnd_flux_dn => bbd
      if (.not. is_lw) then
        allocate(bbdir(bs, nlay+1, nbnd)); bfl%bnd_flux_dn_dir => bbdir
      end if
    end if
    if (variant == 3 .or. variant == 4) then
      if (.not. (is_lw .and. with_clouds)) error stop 'ref_frontend_driver: variants 3 and 4 are longwave with clouds'
      call stop_on_err(op2%alloc_2str(bs, nlay, k)); call stop_on_err(cld2%alloc_2str(bs, nlay, cloud_spec))
    end if
    do b = 1 + tid, nblocks, nthr
      fluxes%flux_up => bup(:, :, b); fluxes%flux_dn => bdn(:, :, b)   ! (rrtmgp_rfmip_lw.F90:259-260)
      call system_clock(tb)
      if (is_lw) then
        if (use_col_dry .and. use_tlev) then
          e = k%gas_optics(bp_lay(:,:,b), bp_lev(:,:,b), bt_lay(:,:,b), bt_sfc(:,b), concs(b), op1, src, &
                           col_dry=bcol_dry(:,:,b), tlev=bt_lev(:,:,b))
        else if (use_col_dry) then
          e = k%gas_optics(bp_lay(:,:,b), bp_lev(:,:,b), bt_lay(:,:,b), bt_sfc(:,b), concs(b), op1, src, col_dry=bcol_dry(:,:,b))
        else if (use_tlev) then
          e = k%gas_optics(bp_lay(:,:,b), bp_lev(:,:,b), bt_lay(:,:,b), bt_sfc(:,b), concs(b), op1, src, tlev=bt_lev(:,:,b))
        else
          e = k%gas_optics(bp_lay(:,:,b), bp_lev(:,:,b), bt_lay(:,:,b), bt_sfc(:,b), concs(b), op1, src)
        end if
        call stop_on_err(e)
        if (with_clouds .and. variant < 3) then   ! rrtmgp_allsky.F90:362-375: clouds as absorbers, added band by band
          call stop_on_err(cloud_spec%cloud_optics(blwp(:,:,b), biwp(:,:,b), brel(:,:,b), bdei(:,:,b), cld1))
          call stop_on_err(cld1%increment(op1))
        end if
        call system_clock(tc)
        select case (variant)
        case (1)
          call stop_on_err(k%compute_optimal_angles(op1, ds))
          call stop_on_err(rte_lw(op1, src, bsfc(:,:,b), fluxes, lw_Ds=ds))
        case (2)
          call stop_on_err(rte_lw(op1, src, bsfc(:,:,b), bfl, n_gauss_angles=n_ang))
          bup(:, :, b) = sum(bbu, dim=3); bdn(:, :, b) = sum(bbd, dim=3)
        case (3, 4)
          ! scattering clouds in the longwave: gas optics into two-stream properties (ssa = 0), clouds added by band
          e = k%gas_optics(bp_lay(:,:,b), bp_lev(:,:,b), bt_lay(:,:,b), bt_sfc(:,b), concs(b), op2, src, tlev=bt_lev(:,:,b))
          call stop_on_err(e)
          call stop_on_err(cloud_spec%cloud_optics(blwp(:,:,b), biwp(:,:,b), brel(:,:,b), bdei(:,:,b), cld2))
          call stop_on_err(cld2%increment(op2))
          call stop_on_err(rte_lw(op2, src, bsfc(:,:,b), fluxes, n_gauss_angles=n_ang, use_2stream=(variant == 4)))
        case (5)
          call stop_on_err(rte_lw(op1, src, bsfc(:,:,b), fluxes, inc_flux=incf, n_gauss_angles=n_ang))
        case default
          call stop_on_err(rte_lw(op1, src, bsfc(:,:,b), fluxes, n_gauss_angles=n_ang))
        end select
        call system_clock(td)
      else
        fluxes%flux_dn_dir => bdir(:, :, b)
        if (use_col_dry) then
          e = k%gas_optics(bp_lay(:,:,b), bp_lev(:,:,b), bt_lay(:,:,b), concs(b), op2, toa, col_dry=bcol_dry(:,:,b))
        else
          e = k%gas_optics(bp_lay(:,:,b), bp_lev(:,:,b), bt_lay(:,:,b), concs(b), op2, toa)
        end if
        call stop_on_err(e)
        if (with_clouds) then   ! :382-396: two-stream clouds, delta-scaled, added band by band
          call stop_on_err(cloud_spec%cloud_optics(blwp(:,:,b), biwp(:,:,b), brel(:,:,b), bdei(:,:,b), cld2))
          call stop_on_err(cld2%delta_scale())
          call stop_on_err(cld2%increment(op2))
        end if
        call system_clock(tc)
        if (variant == 2) then
          call stop_on_err(rte_sw(op2, bmu0(:,b), toa, bsfc(:,:,b), bsfc(:,:,b), bfl))
          bup(:, :, b) = sum(bbu, dim=3); bdn(:, :, b) = sum(bbd, dim=3); bdir(:, :, b) = sum(bbdir, dim=3)
        else if (variant == 5) then
          call stop_on_err(rte_sw(op2, bmu0(:,b), toa, bsfc(:,:,b), bsfc(:,:,b), fluxes, inc_flux_dif=incf))
        else
          call stop_on_err(rte_sw(op2, bmu0(:,b), toa, bsfc(:,:,b), bsfc(:,:,b), fluxes))
        end if
        call system_clock(td)
      end if
      tick_go = tick_go + (tc - tb); tick_rte = tick_rte + (td - tc)
    end do
    ! where a thread's time went (wall clock of the frontend calls, the library's share of them is in its own report)
    if (tid == 0 .and. timing_lines) print '(a,f9.4,a,f9.4,a)', '  thread 0: gas_optics ', real(tick_go, 8) / real(rate, 8), &
      ' s, rte ', real(tick_rte, 8) / real(rate, 8), ' s'
  end subroutine
  subroutine setup_k()
    call load_kdist_stream(fk, gases, k, is_lw)
  end subroutine
  subroutine setup_concs()
    integer :: b, ig, c0, c1
    do b = 1, nblocks
      c0 = (b - 1) * bs + 1; c1 = b * bs
      call stop_on_err(concs(b)%init(gases))
      do ig = 1, ngas
        call stop_on_err(concs(b)%set_vmr(trim(gases(ig)), vmr(c0:c1, :, ig)))
      end do
    end do
  end subroutine
  ! runs `proc` with mb megabytes of stack pushed first (an automatic array that is touched at both ends)
  subroutine with_stack_pad(mb, which)
    integer, intent(in) :: mb, which
    character(len=1) :: pad(int(max(1, mb), 8) * 1048576_8)
    pad(1) = 'a'; pad(size(pad, kind=8)) = 'z'
    if (which == 1) call setup_k()
    if (which == 2) call setup_concs()
    if (pad(1) /= 'a' .or. pad(size(pad, kind=8)) /= 'z') error stop 'ref_frontend_driver: stack pad overwritten'
  end subroutine
  subroutine stop_on_err(msg)
    character(len=*), intent(in) :: msg
    if (len_trim(msg) > 0) then
      print *, 'ref_frontend_driver: ', trim(msg)
      error stop 3
    end if
  end subroutine
end program ref_frontend_driver
