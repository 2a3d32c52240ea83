!===============================================================================
! sb_oracle.f90 -- CPU ORACLE (test infrastructure, NOT product code)
!
! A from-scratch restatement of the sea-breeze trigger arithmetic of the
! reference (antarcticrainforest/seabreeze_param).  It exists so that tests/,
! __graft_entry__.smoke() and bench.py's cpu_baseline leg can check / time the
! reference's arithmetic on a box where /root/reference is absent.  Nothing in
! the product path (seabreeze_param_amd/, include/, fortran/) may call it.
!
! Parity status: PINNED.  tests/test_oracle_pin.py compares every routine in
! this file against the reference's own Fortran compiled unmodified from
! /root/reference into oracle/_ref/ (see oracle/Makefile), and
! tests/golden/*.npz holds inputs + outputs captured from that compiled
! reference.  The reference ships no test vectors of its own (SURVEY.md §4).
!
! Working precision follows the compile flag, exactly as the reference does:
!   amdflang -O2                   -> real = 4 bytes  (liboracle_r4.so)
!   amdflang -O2 -fdefault-real-8  -> real = 8 bytes  (liboracle_r8.so)
! so every literal below is converted at the working precision the same way
! the reference's literals are.
!
! Layout: Fortran order, longitude fastest: f(lon, lat[, lev]).
!
! Each routine cites the reference lines it follows ("ref:" = path under
! /root/reference).  Entry points are bind(C) with by-value scalars so ctypes
! and C can call them without a Fortran descriptor.
!===============================================================================
module sb_oracle
  use iso_c_binding, only : c_int
  implicit none
  private

  ! boundary rules for window / neighbour indexing
  integer, parameter :: BND_WRAPPER = 0  ! ref: python_wrapper/seabreezediag/seabreeze_diag_python.f90:201-202
  integer, parameter :: BND_GLOBAL  = 1  ! lat clamp + true periodic lon (single-domain reading of generic/:198-200)
  integer, parameter :: BND_HALO    = 2  ! raw reads into caller-provided ghost cells (ref: UM/vn10.7/sea_breeze_diag.F90:241-243)

  ! tuning constants, ref: generic/sea_breeze_diag.f90:127-138
  real, parameter :: c_rad2deg = 57.2957
  real, parameter :: c_gmma    = -0.0060956

contains

  !-----------------------------------------------------------------------------
  ! index maps
  !-----------------------------------------------------------------------------
  pure integer function map_lat(ii, nlats, mode) result(ki)
    integer, intent(in) :: ii, nlats, mode
    if (mode == BND_HALO) then
      ki = ii
    else
      ki = min(max(1, ii), nlats)        ! ref: seabreeze_diag_python.f90:201
    end if
  end function map_lat

  pure integer function map_lon(jj, nlons, mode) result(kj)
    integer, intent(in) :: jj, nlons, mode
    select case (mode)
    case (BND_WRAPPER)
      kj = max(1, modulo(jj, nlons))     ! ref: seabreeze_diag_python.f90:202 (0 and nlons both -> 1)
    case (BND_GLOBAL)
      kj = modulo(jj - 1, nlons) + 1     ! true periodic wrap
    case default
      kj = jj
    end select
  end function map_lon

  !-----------------------------------------------------------------------------
  ! sigmoid of the sub-grid orography standard deviation
  ! ref: generic/sea_breeze_diag.f90:457-481, seabreeze_diag_python.f90:287-311
  !-----------------------------------------------------------------------------
  subroutine sbo_sigmoid(ary, nlons, nlats, sm) bind(C, name='sbo_sigmoid')
    integer(c_int), value, intent(in) :: nlons, nlats
    real, intent(in)  :: ary(nlons, nlats)
    real, intent(out) :: sm(nlons, nlats)
    real :: mean, var, std, r
    integer :: i, j
    mean = sum(ary) / (nlons*nlats)
    var = 0
    do i = 1, nlats
      do j = 1, nlons
        var = var + (ary(j,i) - mean)**2
      end do
    end do
    std = 2 / sqrt(var / (nlons*nlats))
    r = (maxval(ary) - minval(ary)) / 4.
    sm = 1 / (1 + exp(-std*(ary - r)))
  end subroutine sbo_sigmoid

  !-----------------------------------------------------------------------------
  ! expanding-window land/sea mean contrast for one cell.
  ! cls is the signed distance field (>=0 land side), t0 the sea-level
  ! temperature; both dimensioned (1-h:nlons+h, 1-h:nlats+h) with h=0 unless
  ! mode==BND_HALO.  The sums restart from zero at each radius and run lat
  ! outer / lon inner, ref: generic/sea_breeze_diag.f90:191-216,
  ! seabreeze_diag_python.f90:191-221.  The reference loop has no upper bound on
  ! the radius (it never returns on a one-class grid); here the search stops
  ! at nn_cap and the 0/0 of the empty class yields NaN.
  !-----------------------------------------------------------------------------
  subroutine contrast(j, i, cls, t0, nlons, nlats, h, mode, nn_cap, n_thc, nn_used)
    integer, intent(in) :: j, i, nlons, nlats, h, mode, nn_cap
    real, intent(in) :: cls(1-h:nlons+h, 1-h:nlats+h), t0(1-h:nlons+h, 1-h:nlats+h)
    real, intent(out) :: n_thc
    integer, intent(out) :: nn_used
    real :: mul, T_l, T_s, n_l, n_s
    integer :: nn, ii, jj, ki, kj
    logical :: found

    if (cls(j,i) >= 0.0) then
      mul = 1
    else
      mul = -1
    end if
    found = .false.
    nn = 1
    do while (.not. found)
      n_l = 0
      n_s = 0
      T_l = 0
      T_s = 0
      do ii = i-nn, i+nn
        ki = map_lat(ii, nlats, mode)
        do jj = j-nn, j+nn
          kj = map_lon(jj, nlons, mode)
          if (cls(kj,ki) >= 0.0) then
            T_l = T_l + t0(kj,ki)
            n_l = n_l + 1
          else
            T_s = T_s + t0(kj,ki)
            n_s = n_s + 1
          end if
        end do
      end do
      if (n_s > 0 .and. n_l > 0) then
        found = .true.
      else if (nn >= nn_cap) then
        exit
      else
        nn = nn + 1
      end if
    end do
    nn_used = nn
    n_thc = mul * ((T_l/n_l) - (T_s/n_s))
  end subroutine contrast

  !-----------------------------------------------------------------------------
  ! threshold test + scaling, ref: generic/sea_breeze_diag.f90:242-259
  !-----------------------------------------------------------------------------
  pure real function trigger(n_thc, ws_old, wd_old, n_ws, n_wd, &
                             thr_wind, thr_dir, thr_ch, thr_thc) result(sb)
    real, intent(in) :: n_thc, ws_old, wd_old, n_ws, n_wd
    real, intent(in) :: thr_wind, thr_dir, thr_ch, thr_thc
    real :: thc_abs, mws, dws, dwd, scale_wind, scale_thc
    thc_abs = abs(n_thc)
    mws = (ws_old + n_ws) / 2.
    dws = abs(ws_old - n_ws)
    dwd = abs(modulo(((wd_old - n_wd) + 180.), 360.) - 180.)
    if (dwd < thr_dir .and. dws < thr_ch .and. mws < thr_wind .and. thc_abs > thr_thc) then
      scale_wind = (thr_wind - mws) / max(real(1), mws)
      scale_thc  = (thc_abs - thr_thc) / n_thc
      sb = scale_thc*scale_wind
    else
      sb = 0.0
    end if
  end function trigger

  !-----------------------------------------------------------------------------
  ! f2py-surface flavour: whole global grid, 1-D pressure, packed output.
  ! ref: python_wrapper/seabreezediag/seabreeze_diag_python.f90:49-285
  ! Units as at that surface: target_plev hPa, target_time h, timestep min.
  ! Unlike the reference the three unit scalars are NOT overwritten in place
  ! (they are by-value here).  windspeed/winddir/thc are updated in place like
  ! the reference's dummies; row nlats of output is left untouched (:165).
  !-----------------------------------------------------------------------------
  subroutine sbo_diag(tn, p, z, std, theta, v, u, cdist, windspeed, winddir, thc, &
                      target_plev, thresh_wind, thresh_winddir, thresh_windch, &
                      thresh_thc, target_time, maxdist, timestep, nps, nlons, nlats, &
                      output, nn_max) bind(C, name='sbo_diag')
    integer(c_int), value, intent(in) :: tn, nps, nlons, nlats
    real, value, intent(in) :: target_plev, thresh_wind, thresh_winddir, thresh_windch
    real, value, intent(in) :: thresh_thc, target_time, maxdist, timestep
    real, intent(in) :: p(nps), z(nlons,nlats), std(nlons,nlats), theta(nlons,nlats)
    real, intent(in) :: v(nlons,nlats,nps), u(nlons,nlats,nps), cdist(nlons,nlats)
    real, intent(inout) :: windspeed(nlons,nlats), winddir(nlons,nlats), thc(nlons,nlats)
    real, intent(inout) :: output(nlons,nlats,4)
    integer(c_int), intent(out) :: nn_max          ! largest search radius used (diagnostic)
    real, allocatable :: t0(:,:), smod(:,:)
    real :: dt_s, period_s, plev_pa, n_thc, n_ws, n_wd, sb
    integer :: i, j, lev, nn
    logical :: refresh

    dt_s     = timestep * 60.
    period_s = target_time * 60.**2
    plev_pa  = target_plev * 100.

    allocate(t0(nlons,nlats), smod(nlons,nlats))
    call sbo_sigmoid(std, nlons, nlats, smod)
    t0 = theta - (c_gmma * z * smod)

    lev = int(minloc(abs(p - plev_pa), 1))
    refresh = modulo(real(tn)*dt_s, period_s) < 0.0001
    nn_max = 0

    do i = 1, nlats-1
      do j = 1, nlons
        if (abs(cdist(j,i)) > maxdist) then
          sb = 2.0E20
        else
          call contrast(j, i, cdist, t0, nlons, nlats, 0, BND_WRAPPER, nlons+nlats, n_thc, nn)
          nn_max = max(nn_max, nn)
          n_ws = sqrt(u(j,i,lev)**2 + v(j,i,lev)**2)
          n_wd = atan2(-1*u(j,i,lev), -1*v(j,i,lev)) * c_rad2deg
          if (tn < 2) then
            thc(j,i) = n_thc
            winddir(j,i) = n_wd
            windspeed(j,i) = n_ws
          end if
          sb = trigger(n_thc, windspeed(j,i), winddir(j,i), n_ws, n_wd, &
                       thresh_wind, thresh_winddir, thresh_windch, thresh_thc)
          thc(j,i) = n_thc
          if (refresh) then                 ! ref :271-274 (both speed and direction)
            windspeed(j,i) = n_ws
            winddir(j,i) = n_wd
          end if
        end if
        output(j,i,1) = sb
        output(j,i,2) = t0(j,i)
        output(j,i,3) = windspeed(j,i)
        output(j,i,4) = winddir(j,i)
      end do
    end do
    deallocate(t0, smod)
  end subroutine sbo_diag

  !-----------------------------------------------------------------------------
  ! host-model flavour: 3-D pressure with per-column level search, all rows,
  ! zero fill outside the coastal band, wind speed refreshed every call.
  ! ref: generic/sea_breeze_diag.f90:55-271 with the search flag reset per cell
  ! as in seabreeze_diag_python.f90:167 (the generic file never initialises
  ! `found`, SURVEY.md App. C #1).  `h` ghost cells surround theta/mask/z/sigma
  ! when bnd==BND_HALO (UM-style bounds, interior at 1:n); h must be 0 otherwise.
  ! The sigmoid statistics run over the interior only.
  !-----------------------------------------------------------------------------
  subroutine sbo_seabreeze_diag(timestep, tn, p, u, v, theta, mask, z, sigma, &
                                windspeed, winddir, thc, sb_con, &
                                nlons, nlats, nz, h, bnd, nn_max) bind(C, name='sbo_seabreeze_diag')
    real, value, intent(in) :: timestep
    integer(c_int), value, intent(in) :: tn, nlons, nlats, nz, h, bnd
    real, intent(in) :: p(nlons,nlats,nz), u(nlons,nlats,nz), v(nlons,nlats,nz)
    real, intent(in) :: theta(1-h:nlons+h,1-h:nlats+h), mask(1-h:nlons+h,1-h:nlats+h)
    real, intent(in) :: z(1-h:nlons+h,1-h:nlats+h), sigma(1-h:nlons+h,1-h:nlats+h)
    real, intent(inout) :: windspeed(nlons,nlats), winddir(nlons,nlats)
    real, intent(inout) :: thc(nlons,nlats), sb_con(nlons,nlats)
    integer(c_int), intent(out) :: nn_max
    call sbo_seabreeze_diag_x(timestep, tn, p, u, v, theta, mask, z, sigma, windspeed, winddir, thc, sb_con, &
                              nlons, nlats, nz, h, bnd, 0, 0., 0., 0, nn_max)
  end subroutine sbo_seabreeze_diag

  ! scalars of the logistic (std, r) for a field, ref: generic/sea_breeze_diag.f90:466-479
  subroutine sbo_sigmoid_scalars(ary, nlons, nlats, std, r) bind(C, name='sbo_sigmoid_scalars')
    integer(c_int), value, intent(in) :: nlons, nlats
    real, intent(in)  :: ary(nlons, nlats)
    real, intent(out) :: std, r
    real :: mean, var
    integer :: i, j
    mean = sum(ary) / (nlons*nlats)
    var = 0
    do i = 1, nlats
      do j = 1, nlons
        var = var + (ary(j,i) - mean)**2
      end do
    end do
    std = 2 / sqrt(var / (nlons*nlats))
    r = (maxval(ary) - minval(ary)) / 4.
  end subroutine sbo_sigmoid_scalars

  ! As sbo_seabreeze_diag; with use_ext /= 0 the logistic scalars come from the caller
  ! (the GLOBAL field's std and r) instead of this sub-domain's sigma: what one latitude
  ! band of a decomposed grid has to use to reproduce the single-domain result.
  ! level_rule 0: first minimum of |p - target| over all levels (ref: generic/sea_breeze_diag.f90:223);
  ! level_rule 1: the UM copy's walk -- upwards from level 1 while the difference does not grow, starting from
  ! 1000000., stop at the first increase (ref: UM/vn10.7/sea_breeze_diag.F90:265-274; where even level 1 fails
  ! the UM copy leaves p_lev undefined, level 1 is used here).
  subroutine sbo_seabreeze_diag_x(timestep, tn, p, u, v, theta, mask, z, sigma, &
                                windspeed, winddir, thc, sb_con, &
                                nlons, nlats, nz, h, bnd, use_ext, ext_std, ext_r, level_rule, nn_max) &
      bind(C, name='sbo_seabreeze_diag_x')
    real, value, intent(in) :: timestep, ext_std, ext_r
    integer(c_int), value, intent(in) :: tn, nlons, nlats, nz, h, bnd, use_ext, level_rule
    real, intent(in) :: p(nlons,nlats,nz), u(nlons,nlats,nz), v(nlons,nlats,nz)
    real, intent(in) :: theta(1-h:nlons+h,1-h:nlats+h), mask(1-h:nlons+h,1-h:nlats+h)
    real, intent(in) :: z(1-h:nlons+h,1-h:nlats+h), sigma(1-h:nlons+h,1-h:nlats+h)
    real, intent(inout) :: windspeed(nlons,nlats), winddir(nlons,nlats)
    real, intent(inout) :: thc(nlons,nlats), sb_con(nlons,nlats)
    integer(c_int), intent(out) :: nn_max
    real, parameter :: target_plev = 100 * 700., thresh_wind = 11., thresh_winddir = 90.
    real, parameter :: thresh_windch = 5., thresh_thc = 0.75, target_time = 6.*60**2
    real, parameter :: maxdist = 180.
    real, allocatable :: t0(:,:), s_in(:,:), s_sm(:,:)
    real :: mean, var, std, r, n_thc, n_ws, n_wd, p_lev_diff
    integer :: i, j, lev, nn, k
    logical :: refresh

    allocate(t0(1-h:nlons+h,1-h:nlats+h))
    if (h == 0 .and. use_ext == 0) then
      allocate(s_sm(nlons,nlats))
      call sbo_sigmoid(sigma, nlons, nlats, s_sm)
      t0 = theta - (c_gmma * z * s_sm)
      deallocate(s_sm)
    else
      if (use_ext /= 0) then
        std = ext_std
        r = ext_r
      else
        ! statistics from the interior, logistic applied to ghost cells too
        allocate(s_in(nlons,nlats))
        s_in = sigma(1:nlons,1:nlats)
        call sbo_sigmoid_scalars(s_in, nlons, nlats, std, r)
        deallocate(s_in)
      end if
      t0 = theta - (c_gmma * z * (1 / (1 + exp(-std*(sigma - r)))))
    end if

    refresh = modulo(real(tn)*timestep, target_time) < 0.0001
    nn_max = 0
    do i = 1, nlats
      do j = 1, nlons
        if (abs(mask(j,i)) > maxdist) then
          sb_con(j,i) = 0.0                                   ! ref :174-176
        else
          call contrast(j, i, mask, t0, nlons, nlats, h, bnd, nlons+nlats, n_thc, nn)
          nn_max = max(nn_max, nn)
          if (level_rule == 0) then
            lev = int(minloc(abs(p(j,i,:) - target_plev), 1))   ! ref :223
          else
            p_lev_diff = 1000000.                               ! ref: UM/vn10.7/sea_breeze_diag.F90:265-274
            lev = 1
            do k = 1, nz
              if (abs(p(j,i,k) - target_plev) <= p_lev_diff) then
                lev = k
                p_lev_diff = abs(p(j,i,k) - target_plev)
              else
                exit
              end if
            end do
          end if
          n_ws = sqrt(u(j,i,lev)**2 + v(j,i,lev)**2)
          n_wd = atan2(-1*u(j,i,lev), -1*v(j,i,lev)) * c_rad2deg
          if (tn < 2) then
            thc(j,i) = n_thc
            winddir(j,i) = n_wd
            windspeed(j,i) = n_ws
          end if
          sb_con(j,i) = trigger(n_thc, windspeed(j,i), winddir(j,i), n_ws, n_wd, &
                                thresh_wind, thresh_winddir, thresh_windch, thresh_thc)
          windspeed(j,i) = n_ws                               ! ref :261 (every call)
          thc(j,i) = n_thc
          if (refresh) winddir(j,i) = n_wd                    ! ref :264-266
        end if
      end do
    end do
    deallocate(t0)
  end subroutine sbo_seabreeze_diag_x

  !-----------------------------------------------------------------------------
  ! coastline by binary 3x3 Sobel.
  ! rule 0: land <=> lsm+ci > 0.4            ref: sobel.f90:51,69-73
  ! rule 1: ice-aware two-branch mask         ref: generic/sea_breeze_diag.f90:325-337
  ! neighbour indexing: bnd as above (BND_WRAPPER reproduces sobel.f90:67-68).
  !-----------------------------------------------------------------------------
  subroutine sbo_get_edges(lsm, ci, nlons, nlats, rule, bnd, coast) bind(C, name='sbo_get_edges')
    integer(c_int), value, intent(in) :: nlons, nlats, rule, bnd
    real, intent(in)  :: lsm(nlons,nlats), ci(nlons,nlats)
    real, intent(out) :: coast(nlons,nlats)
    real, allocatable :: land(:,:)
    integer :: w(3,3), x, y, a, b, xx, yy
    real :: px, py, g

    allocate(land(nlons,nlats))
    do y = 1, nlats
      do x = 1, nlons
        if (rule == 0) then
          if (lsm(x,y) + ci(x,y) > 0.4) then
            land(x,y) = 1
          else
            land(x,y) = 0
          end if
        else
          if (ci(x,y) <= 0.2) then
            land(x,y) = merge(1., 0., lsm(x,y) >= 0.5)
          else
            land(x,y) = merge(1., 0., lsm(x,y) + ci(x,y) >= 0.5)
          end if
        end if
      end do
    end do

    w = reshape((/-1,-2,-1, 0,0,0, 1,2,1/), shape(w))
    do y = 1, nlats
      do x = 1, nlons
        px = 0.0
        py = 0.0
        do b = -1, 1          ! lon offset
          do a = -1, 1        ! lat offset
            yy = map_lat(y+a, nlats, bnd)
            xx = map_lon(x+b, nlons, bnd)
            px = px + w(a+2, b+2) * land(xx,yy)
            py = py + w(b+2, a+2) * land(xx,yy)
          end do
        end do
        g = sqrt(px**2 + py**2)
        if (g == 0) then
          coast(x,y) = 0.
        else
          coast(x,y) = 1
        end if
      end do
    end do
    deallocate(land)
  end subroutine sbo_get_edges

  !-----------------------------------------------------------------------------
  ! window half-width chosen from the grid spacing at 70 deg
  ! ref: sobel.f90:129-137
  !-----------------------------------------------------------------------------
  subroutine sbo_dist_window(lon, lat, nlons, nlats, maxdist, k) bind(C, name='sbo_dist_window')
    integer(c_int), value, intent(in) :: nlons, nlats
    real, value, intent(in) :: maxdist
    real, intent(in) :: lon(nlons), lat(nlats)
    integer(c_int), intent(out) :: k
    real, parameter :: R = 6370.9989, pi = 3.1415926, d2r = pi/180.0
    integer :: tlat
    real :: dphi, dlam, a, dx, p0, p1
    tlat = int(minloc(abs(70 - lat), 1))
    p0 = d2r*lat(tlat)
    p1 = d2r*lat(tlat+1)
    dphi = p1 - p0
    dlam = d2r*lon(2) - d2r*lon(1)
    a = sin(dphi/2)**2 + (cos(p1)*(cos(p0)*sin(dlam/2)**2))
    dx = R*2*atan2(sqrt(a), sqrt(1-a))
    k = int(maxdist / dx)
  end subroutine sbo_dist_window

  !-----------------------------------------------------------------------------
  ! signed great-circle distance to the nearest coast cell, scatter form in the
  ! reference's own sweep order (so the at-sweep-time reset of :188 is kept).
  ! ref: sobel.f90:91-193.  kwin < 0 -> window from sbo_dist_window (wrapper);
  ! kwin >= 0 -> fixed +-kwin window (generic/UM: the halo width,
  ! ref: generic/sea_breeze_diag.f90:422,425).
  !-----------------------------------------------------------------------------
  subroutine sbo_get_dist(coast, mask, lon, lat, nlons, nlats, maxdist, kwin, cdist) &
      bind(C, name='sbo_get_dist')
    integer(c_int), value, intent(in) :: nlons, nlats, kwin
    real, value, intent(in) :: maxdist
    real, intent(in)  :: coast(nlons,nlats), mask(nlons,nlats), lon(nlons), lat(nlats)
    real, intent(out) :: cdist(nlons,nlats)
    real, parameter :: R = 6370.9989, pi = 3.1415926, d2r = pi/180.0
    real, allocatable :: phi(:)
    integer :: k, i, j, ii, jj, xx, yy
    real :: l1, l2, dphi, dlam, a, c

    if (kwin < 0) then
      call sbo_dist_window(lon, lat, nlons, nlats, maxdist, k)
    else
      k = kwin
    end if
    allocate(phi(nlats))
    phi = d2r * lat
    cdist = 12000.

    do i = 1, nlats
      do j = 1, nlons
        if (coast(j,i) > 0.) then
          if (lon(j) > 180) then
            l1 = d2r * (lon(j) - 360.)
          else
            l1 = d2r * lon(j)
          end if
          do ii = -k, k
            yy = min(max(1, ii+i), nlats)
            dphi = phi(i) - phi(yy)
            do jj = -k, k
              xx = modulo(j+jj, nlons)
              if (xx == 0) xx = nlons
              if (lon(xx) > 180) then
                l2 = d2r * (lon(xx) - 360.)
              else
                l2 = d2r * lon(xx)
              end if
              dlam = l1 - l2
              a = sin(dphi/2)**2 + (cos(phi(i))*(cos(phi(yy))*sin(dlam/2)**2))
              c = R*2*atan2(sqrt(a), sqrt(1-a)) + 0.5
              if (c < abs(cdist(xx,yy))) then
                if (mask(xx,yy) > 0.0) then
                  cdist(xx,yy) = c
                else
                  cdist(xx,yy) = -c
                end if
              end if
            end do
          end do
        end if
        if (abs(cdist(j,i)) > 2*maxdist) cdist(j,i) = 12000.
      end do
    end do
    deallocate(phi)
  end subroutine sbo_get_dist

  !-----------------------------------------------------------------------------
  ! Race-free all-core variant of sbo_seabreeze_diag's point loop for the timed
  ! CPU baseline (BASELINE.md §3 (ii)).  Same arithmetic; rows are distributed
  ! over OpenMP threads, only per-thread scalars are private.  Built only when
  ! compiled with -fopenmp; without it this is the serial loop again.
  !-----------------------------------------------------------------------------
  subroutine sbo_seabreeze_diag_omp(timestep, tn, p, u, v, theta, mask, z, sigma, &
                                    windspeed, winddir, thc, sb_con, &
                                    nlons, nlats, nz) bind(C, name='sbo_seabreeze_diag_omp')
    real, value, intent(in) :: timestep
    integer(c_int), value, intent(in) :: tn, nlons, nlats, nz
    real, intent(in) :: p(nlons,nlats,nz), u(nlons,nlats,nz), v(nlons,nlats,nz)
    real, intent(in) :: theta(nlons,nlats), mask(nlons,nlats), z(nlons,nlats), sigma(nlons,nlats)
    real, intent(inout) :: windspeed(nlons,nlats), winddir(nlons,nlats)
    real, intent(inout) :: thc(nlons,nlats), sb_con(nlons,nlats)
    real, parameter :: target_plev = 100 * 700., thresh_wind = 11., thresh_winddir = 90.
    real, parameter :: thresh_windch = 5., thresh_thc = 0.75, target_time = 6.*60**2
    real, parameter :: maxdist = 180.
    real, allocatable :: t0(:,:), s_sm(:,:)
    real :: n_thc, n_ws, n_wd
    integer :: i, j, lev, nn
    logical :: refresh

    allocate(t0(nlons,nlats), s_sm(nlons,nlats))
    call sbo_sigmoid(sigma, nlons, nlats, s_sm)
    t0 = theta - (c_gmma * z * s_sm)
    deallocate(s_sm)
    refresh = modulo(real(tn)*timestep, target_time) < 0.0001
    !$omp parallel do schedule(dynamic,4) default(shared) private(i,j,lev,nn,n_thc,n_ws,n_wd)
    do i = 1, nlats
      do j = 1, nlons
        if (abs(mask(j,i)) > maxdist) then
          sb_con(j,i) = 0.0
        else
          call contrast(j, i, mask, t0, nlons, nlats, 0, BND_GLOBAL, nlons+nlats, n_thc, nn)
          lev = int(minloc(abs(p(j,i,:) - target_plev), 1))
          n_ws = sqrt(u(j,i,lev)**2 + v(j,i,lev)**2)
          n_wd = atan2(-1*u(j,i,lev), -1*v(j,i,lev)) * c_rad2deg
          if (tn < 2) then
            thc(j,i) = n_thc
            winddir(j,i) = n_wd
            windspeed(j,i) = n_ws
          end if
          sb_con(j,i) = trigger(n_thc, windspeed(j,i), winddir(j,i), n_ws, n_wd, &
                                thresh_wind, thresh_winddir, thresh_windch, thresh_thc)
          windspeed(j,i) = n_ws
          thc(j,i) = n_thc
          if (refresh) winddir(j,i) = n_wd
        end if
      end do
    end do
    !$omp end parallel do
    deallocate(t0)
  end subroutine sbo_seabreeze_diag_omp

  subroutine sbo_real_bytes(nbytes) bind(C, name='sbo_real_bytes')
    integer(c_int), intent(out) :: nbytes
    real :: x
    nbytes = storage_size(x) / 8
  end subroutine sbo_real_bytes

end module sb_oracle
