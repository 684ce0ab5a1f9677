! factored_binding_driver.F90 (ours; test infrastructure) -- compiles and exercises the Fortran binding of the factored LW
! sources, shim/mo_rte_hip_factored.F90 (INTEGRATION.md section 4a), beside the reference's own kernel interface modules:
!   (1) compute_Planck_source -> lw_solver_noscat through the reference's interface (rrtmgp/kernels/api, rte/kernels/api);
!   (2) rte_hip_compute_Planck_source_factored -> rte_hip_lw_solver_noscat_factored through the binding;
!   (3) rte_hip_expand_factored_sources against the arrays of (1).
! Data-free: a small deterministic k-distribution-shaped input set (valid index ranges, weights that sum to one).
! Prints "factored binding: PASS" when the broadband fluxes of (1) and (2) and the sources of (1) and (3) are bit-identical.
program factored_binding_driver
  use iso_c_binding,                only: c_int, c_double
  use mo_rte_kind,                  only: wp, wl
  use mo_gas_optics_rrtmgp_kernels, only: compute_Planck_source
  use mo_rte_solver_kernels,        only: lw_solver_noscat
  use mo_rte_hip_factored
  implicit none
  integer, parameter :: ncol = 777, nlay = 60, nbnd = 4, gpb = 16, ngpt = nbnd*gpb, nflav = 3, neta = 9, npres = 59, &
                        ntemp = 14, nPlanckTemp = 196, nmus = 1
  real(wp), parameter :: temp_ref_min = 160._wp, totplnk_delta = 195._wp/real(nPlanckTemp-1, wp)
  real(wp), allocatable :: tlay(:,:), tlev(:,:), tsfc(:), fmajor(:,:,:,:,:,:), pfracin(:,:,:,:), totplnk(:,:)
  integer,  allocatable :: jeta(:,:,:,:), jtemp(:,:), jpress(:,:), gpoint_bands(:), band_lims_gpt(:,:), gpoint_flavor(:,:)
  logical(wl), allocatable :: tropo(:,:)
  real(wp), allocatable :: sfc_src(:,:), lay_src(:,:,:), lev_src(:,:,:), sfc_jac(:,:), tau(:,:,:), Ds(:,:,:), emis(:,:), inc(:,:)
  real(wp), allocatable :: sfc_src2(:,:), pfrac(:,:,:), plk_lay(:,:,:), plk_lev(:,:,:), sfc_jac2(:,:), lay3(:,:,:), lev3(:,:,:)
  real(wp), allocatable :: bb_up(:,:), bb_dn(:,:), bb_up2(:,:), bb_dn2(:,:), dum3(:,:,:), dum2(:,:)
  real(wp) :: weights(nmus), fe, fp, ft, r
  integer  :: icol, ilay, igpt, ibnd, iflav, i, j, k
  integer(c_int) :: rc
  integer  :: rc_env
  character(len=8) :: mirror_env
  logical  :: ok, fluxes_only

  allocate(tlay(ncol,nlay), tlev(ncol,nlay+1), tsfc(ncol), fmajor(2,2,2,ncol,nlay,nflav), pfracin(ntemp,neta,npres+1,ngpt), &
           totplnk(nPlanckTemp,nbnd), jeta(2,ncol,nlay,nflav), jtemp(ncol,nlay), jpress(ncol,nlay), gpoint_bands(ngpt), &
           band_lims_gpt(2,nbnd), gpoint_flavor(2,ngpt), tropo(ncol,nlay))
  do ibnd = 1, nbnd
    band_lims_gpt(1,ibnd) = (ibnd-1)*gpb + 1; band_lims_gpt(2,ibnd) = ibnd*gpb
    gpoint_bands(band_lims_gpt(1,ibnd):band_lims_gpt(2,ibnd)) = ibnd
    gpoint_flavor(1,band_lims_gpt(1,ibnd):band_lims_gpt(2,ibnd)) = 1 + mod(ibnd, nflav)
    gpoint_flavor(2,band_lims_gpt(1,ibnd):band_lims_gpt(2,ibnd)) = 1 + mod(ibnd+1, nflav)
    do i = 1, nPlanckTemp
      totplnk(i,ibnd) = (10._wp + real(ibnd,wp)) * (real(i,wp)/real(nPlanckTemp,wp))**3
    end do
  end do
  do igpt = 1, ngpt
    do k = 1, npres+1; do j = 1, neta; do i = 1, ntemp
      pfracin(i,j,k,igpt) = (1._wp + 0.3_wp*sin(0.37_wp*i + 0.91_wp*j + 0.13_wp*k + 0.7_wp*igpt)) / real(gpb,wp)
    end do; end do; end do
  end do
  do ilay = 1, nlay
    do icol = 1, ncol
      r = real(mod(icol*7 + ilay*13, 101), wp) / 101._wp
      tlay(icol,ilay) = 300._wp - 1.6_wp*real(ilay,wp) + 8._wp*r
      tropo(icol,ilay) = ilay <= 22 + mod(icol, 3)
      jtemp(icol,ilay) = min(ntemp-1, max(1, int((tlay(icol,ilay) - temp_ref_min)/15._wp) + 1))
      jpress(icol,ilay) = min(npres-1, max(1, ilay - merge(0, 1, tropo(icol,ilay)) + mod(icol, 2)))
      ft = r; fp = 0.25_wp + 0.5_wp*r
      do iflav = 1, nflav
        jeta(1,icol,ilay,iflav) = 1 + mod(icol + iflav + ilay, neta-1)
        jeta(2,icol,ilay,iflav) = 1 + mod(icol + 2*iflav + ilay, neta-1)
        fe = real(mod(icol*3 + iflav, 17), wp) / 17._wp
        fmajor(1,1,1,icol,ilay,iflav) = (1-fe)*(1-fp)*(1-ft); fmajor(2,1,1,icol,ilay,iflav) = fe*(1-fp)*(1-ft)
        fmajor(1,2,1,icol,ilay,iflav) = (1-fe)*fp*(1-ft);     fmajor(2,2,1,icol,ilay,iflav) = fe*fp*(1-ft)
        fmajor(1,1,2,icol,ilay,iflav) = (1-fe)*(1-fp)*ft;     fmajor(2,1,2,icol,ilay,iflav) = fe*(1-fp)*ft
        fmajor(1,2,2,icol,ilay,iflav) = (1-fe)*fp*ft;         fmajor(2,2,2,icol,ilay,iflav) = fe*fp*ft
      end do
    end do
  end do
  tlev(:,1) = tlay(:,1) + 0.8_wp
  do ilay = 2, nlay
    tlev(:,ilay) = 0.5_wp*(tlay(:,ilay-1) + tlay(:,ilay))
  end do
  tlev(:,nlay+1) = tlay(:,nlay) - 0.8_wp
  tsfc = tlev(:,1) + 1.5_wp

  allocate(sfc_src(ncol,ngpt), lay_src(ncol,nlay,ngpt), lev_src(ncol,nlay+1,ngpt), sfc_jac(ncol,ngpt), tau(ncol,nlay,ngpt), &
           Ds(ncol,ngpt,nmus), emis(ncol,ngpt), inc(ncol,ngpt), sfc_src2(ncol,ngpt), pfrac(ncol,nlay,ngpt), plk_lay(ncol,nlay,nbnd), &
           plk_lev(ncol,nlay+1,nbnd), sfc_jac2(ncol,ngpt), lay3(ncol,nlay,ngpt), lev3(ncol,nlay+1,ngpt), bb_up(ncol,nlay+1), &
           bb_dn(ncol,nlay+1), bb_up2(ncol,nlay+1), bb_dn2(ncol,nlay+1), dum3(1,1,1), dum2(ncol,ngpt))
  do igpt = 1, ngpt; do ilay = 1, nlay; do icol = 1, ncol
    tau(icol,ilay,igpt) = 0.002_wp * real(1 + mod(icol + 3*ilay + 5*igpt, 97), wp) * (1._wp + real(mod(igpt-1,gpb),wp))
  end do; end do; end do
  Ds = 1.66_wp; weights = 0.5_wp; emis = 0.97_wp; inc = 0._wp

  ! (1) the reference's kernel interface
  call compute_Planck_source(ncol, nlay, nbnd, ngpt, nflav, neta, npres, ntemp, nPlanckTemp, tlay, tlev, tsfc, 1, fmajor, jeta, &
                             tropo, jtemp, jpress, gpoint_bands, band_lims_gpt, pfracin, temp_ref_min, totplnk_delta, totplnk, &
                             gpoint_flavor, sfc_src, lay_src, lev_src, sfc_jac)
  call lw_solver_noscat(ncol, nlay, ngpt, logical(.false., wl), nmus, Ds, weights, tau, lay_src, lev_src, emis, sfc_src, inc, &
                        dum3, dum3, logical(.true., wl), bb_up, bb_dn, logical(.false., wl), dum2, dum3, logical(.false., wl), dum3, dum3)
  ! (2) the factored pair through the binding
  rc = rte_hip_compute_Planck_source_factored(ncol, nlay, nbnd, ngpt, nflav, neta, npres, ntemp, nPlanckTemp, tlay, tlev, tsfc, 1, fmajor, &
         jeta, tropo, jtemp, jpress, band_lims_gpt, pfracin, real(temp_ref_min, c_double), real(totplnk_delta, c_double), totplnk, &
         gpoint_flavor, sfc_src2, pfrac, plk_lay, plk_lev, sfc_jac2)
  if (rc /= 0) stop "rte_hip_compute_Planck_source_factored failed"
  rc = rte_hip_lw_solver_noscat_factored(ncol, nlay, ngpt, nbnd, 0, nmus, Ds, weights, band_lims_gpt, tau, pfrac, plk_lay, plk_lev, &
         emis, sfc_src2, inc, bb_up2, bb_dn2, 0, dum2, dum3)
  if (rc /= 0) stop "rte_hip_lw_solver_noscat_factored failed"
  ! (3) the reference's two arrays from the factors
  rc = rte_hip_expand_factored_sources(ncol, nlay, nbnd, ngpt, band_lims_gpt, pfrac, plk_lay, plk_lev, lay3, lev3)
  if (rc /= 0) stop "rte_hip_expand_factored_sources failed"

  ! host-mirror mode (RTE_HIP_HOST_MIRROR=1) keeps the arrays a frontend only hands from kernel to kernel -- the sources -- on the
  ! device and does NOT write the host copies: there only the fluxes can be compared on the host
  call get_environment_variable("RTE_HIP_HOST_MIRROR", mirror_env, status=rc_env)
  fluxes_only = rc_env == 0 .and. mirror_env(1:1) == "1"
  ok = all(bb_up == bb_up2) .and. all(bb_dn == bb_dn2)
  if (.not. fluxes_only) ok = ok .and. all(sfc_src == sfc_src2) .and. all(sfc_jac == sfc_jac2) &
       .and. all(lay3 == lay_src) .and. all(lev3 == lev_src)
  print '(a,es12.4,a,2es11.3)', "factored binding: max flux_up ", maxval(bb_up), "  max |difference| of flux_up, flux_dn ", &
        maxval(abs(bb_up - bb_up2)), maxval(abs(bb_dn - bb_dn2))
  if (.not. fluxes_only) print '(a,4es11.3)', "factored binding: max |difference| of sfc_src, sfc_jac, lay_source, lev_source ", &
        maxval(abs(sfc_src - sfc_src2)), maxval(abs(sfc_jac - sfc_jac2)), maxval(abs(lay3 - lay_src)), maxval(abs(lev3 - lev_src))
  if (ok .and. maxval(bb_up) > 0._wp .and. minval(bb_dn(:,1)) > 0._wp) then
    print '(a)', "factored binding: PASS"
  else
    print '(a)', "factored binding: FAIL"
    stop 1
  end if
end program factored_binding_driver
