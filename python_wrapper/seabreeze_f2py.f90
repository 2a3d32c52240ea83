!===============================================================================
! seabreeze_f2py.f90 -- the f2py surface of the reference's extension module `seabreeze`
! (imported at ref: python_wrapper/seabreezediag/__init__.py:3), re-implemented as shims
! over the HIP library.  f2py turns this file into the same Python signatures the
! reference's two Fortran files produce (SURVEY.md App. B.3):
!
!   output = diag(timestep_number,p,z,std,theta,v,u,cdist,windspeed,winddir,thc,
!                 [target_plev,thresh_wind,thresh_winddir,thresh_windch,thresh_thc,
!                  target_time,maxdist,timestep,nps,nlons,nlats])
!   coast  = get_edges(lsm,ci,[nlons,nlats])
!   cdist  = get_dist(coast,mask,lon,lat,[nlons,nlats,maxdist])
!   sm     = sigmoid(ary,[nlons,nlats])
!   nt     = get_threads()
!
! Argument names, order, defaults and units are those of
! ref: python_wrapper/seabreezediag/seabreeze_diag_python.f90:49-52,129-142 and
! ref: python_wrapper/seabreezediag/sobel.f90:19-24,91-97,195-198.
! No arithmetic of the diagnostic lives here.
!===============================================================================

subroutine diag(timestep_number, p, z, std, theta, v, u, cdist, windspeed, winddir, thc, &
                target_plev, thresh_wind, thresh_winddir, thresh_windch, thresh_thc, &
                target_time, maxdist, timestep, nps, nlons, nlats, output)
  use iso_c_binding
  use sb_f2py_state
  implicit none
  integer :: timestep_number, nps, nlons, nlats
  real :: target_plev, thresh_wind, thresh_winddir, thresh_windch, thresh_thc
  real :: target_time, maxdist, timestep
  real, dimension(nps) :: p
  real, dimension(nlons,nlats,nps) :: v, u
  real, dimension(nlons,nlats) :: cdist, theta, z, std, windspeed, winddir, thc
  real, dimension(nlons,nlats,4) :: output
  integer(c_int) :: rc
  !f2py integer, intent(in) :: timestep_number
  !f2py integer, intent(in) :: nps, nlats, nlons
  !f2py real, intent(in) :: p, u, v, theta, z, std, cdist
  !f2py real, intent(in) :: thc, windspeed, winddir
  !f2py real optional, intent(in) :: target_plev = 700., thresh_wind = 11
  !f2py real optional, intent(in) :: thresh_winddir = 90., thresh_windch = 5.
  !f2py real optional, intent(in) :: thresh_thc = 0.75, target_time = 6
  !f2py real optional, intent(in) :: timestep = 24
  !f2py real optional, intent(in) :: maxdist = 180
  !f2py real, intent(out) :: output
  call sb_ok()
  call sb_ensure()
  if (.not. c_associated(ctx)) return
  rc = sb_diag_f32(ctx, timestep_number, p, z, std, theta, v, u, cdist, windspeed, winddir, thc, &
                   target_plev, thresh_wind, thresh_winddir, thresh_windch, thresh_thc, &
                   target_time, maxdist, timestep, nps, nlons, nlats, output)
  if (rc /= 0) call sb_fail('diag', rc)
end subroutine diag

subroutine sigmoid(ary, nlons, nlats, sm)
  use iso_c_binding
  use sb_f2py_state
  implicit none
  integer, intent(in) :: nlons, nlats
  real, dimension(nlons,nlats), intent(in) :: ary
  real, dimension(nlons,nlats), intent(out) :: sm
  integer(c_int) :: rc
  call sb_ok()
  call sb_ensure()
  if (.not. c_associated(ctx)) return
  rc = sb_sigmoid_f32(ctx, nlons, nlats, ary, sm)
  if (rc /= 0) call sb_fail('sigmoid', rc)
end subroutine sigmoid

subroutine get_edges(lsm, ci, nlons, nlats, coast)
  use iso_c_binding
  use sb_f2py_state
  implicit none
  integer :: nlats, nlons
  real, dimension(nlons,nlats) :: lsm, ci, coast
  integer(c_int) :: rc
  !f2py integer, intent(in) :: nlats
  !f2py integer, intent(in) :: nlons
  !f2py real, intent(in) :: lsm, ci
  !f2py real, intent(out) :: coast
  call sb_ok()
  call sb_ensure()
  if (.not. c_associated(ctx)) return
  ! rule 0 (lsm+ci > 0.4) and the wrapper's neighbour indexing, ref: sobel.f90:51,67-69
  rc = sb_get_edges_f32(ctx, nlons, nlats, lsm, ci, 0_c_int, 0_c_int, coast)
  if (rc /= 0) call sb_fail('get_edges', rc)
end subroutine get_edges

subroutine get_dist(coast, mask, lon, lat, nlons, nlats, maxdist, cdist)
  use iso_c_binding
  use sb_f2py_state
  implicit none
  integer :: nlons, nlats
  real, dimension(nlons) :: lon
  real, dimension(nlats) :: lat
  real, dimension(nlons,nlats) :: coast, mask, cdist
  real :: maxdist
  integer(c_int) :: rc
  !f2py integer, intent(in) :: nlons, nlats
  !f2py real, intent(in) :: coast, mask
  !f2py real, intent(in) :: lon, lat
  !f2py real optional, intent(in) :: maxdist = 180
  !f2py real, intent(out) :: cdist
  call sb_ok()
  call sb_ensure()
  if (.not. c_associated(ctx)) return
  ! window half-width from the 70-degree grid spacing (kwin < 0), ref: sobel.f90:129-137
  rc = sb_get_dist_f32(ctx, nlons, nlats, coast, mask, lon, lat, maxdist, -1_c_int, cdist)
  if (rc /= 0) call sb_fail('get_dist', rc)
end subroutine get_dist

subroutine get_threads(nt)
  use iso_c_binding
  use sb_f2py_state
  implicit none
  integer :: nt
  integer(c_int) :: rc, n
  !f2py integer, intent(out) :: nt
  ! the reference reports OpenMP threads (ref: sobel.f90:195-206); here: visible HIP devices
  rc = sb_get_threads(n)
  nt = n
end subroutine get_threads

!===============================================================================
! Additions to the reference's surface (its five routines above keep their signatures):
!
!   rc  = last_status()          status of the last call into this module: 0, or the library's
!   msg = last_message()         sb_status and message -- the Python layer raises on it
!   stream_begin / stream_step / stream_end / stream_stats
!        the timestep loop of the reference's driver (ref: python_wrapper/seabreezediag/__init__.py:222-245)
!        with z, std, cdist and the carried state resident on the device: a step uploads theta and the one
!        u, v plane p selects, and hands back the sb_con plane of the step BEFORE it (double precision, written
!        in place into the caller's array), so that staging step i+1 overlaps the device work of step i.
!        Results are those of diag called step by step.
!===============================================================================
subroutine last_status(rc)
  use sb_f2py_state
  implicit none
  integer :: rc
  !f2py integer, intent(out) :: rc
  rc = last_rc
end subroutine last_status

subroutine last_message(msg)
  use sb_f2py_state
  implicit none
  character(len=512) :: msg
  !f2py character(len=512), intent(out) :: msg
  msg = last_msg
end subroutine last_message

subroutine stream_begin(z, std, cdist, windspeed, winddir, thc, nlons, nlats)
  use iso_c_binding
  use sb_f2py_state
  implicit none
  integer :: nlons, nlats
  real, dimension(nlons,nlats) :: z, std, cdist, windspeed, winddir, thc
  integer(c_int) :: rc
  !f2py integer, intent(in) :: nlons, nlats
  !f2py real, intent(in) :: z, std, cdist, windspeed, winddir, thc
  call sb_ok()
  call sb_ensure()
  if (.not. c_associated(ctx)) return
  rc = sb_diag_stream_begin_f32(ctx, nlons, nlats, z, std, cdist, windspeed, winddir, thc)
  if (rc /= 0) call sb_fail('stream_begin', rc)
end subroutine stream_begin

subroutine stream_step(timestep_number, p, theta, v, u, sb_prev, have_prev, &
                       target_plev, thresh_wind, thresh_winddir, thresh_windch, thresh_thc, &
                       target_time, maxdist, timestep, nps, nlons, nlats)
  use iso_c_binding
  use sb_f2py_state
  implicit none
  integer :: timestep_number, nps, nlons, nlats, have_prev
  real :: target_plev, thresh_wind, thresh_winddir, thresh_windch, thresh_thc
  real :: target_time, maxdist, timestep
  real, dimension(nps) :: p
  real, dimension(nlons,nlats,nps) :: v, u
  real, dimension(nlons,nlats) :: theta
  real(8), dimension(nlons,nlats) :: sb_prev
  integer(c_int) :: rc, hp
  !f2py integer, intent(in) :: timestep_number
  !f2py integer, intent(in) :: nps, nlats, nlons
  !f2py real, intent(in) :: p, u, v, theta
  !f2py real(8), intent(inout) :: sb_prev
  !f2py integer, intent(out) :: have_prev
  !f2py real optional, intent(in) :: target_plev = 700., thresh_wind = 11
  !f2py real optional, intent(in) :: thresh_winddir = 90., thresh_windch = 5.
  !f2py real optional, intent(in) :: thresh_thc = 0.75, target_time = 6
  !f2py real optional, intent(in) :: timestep = 24
  !f2py real optional, intent(in) :: maxdist = 180
  have_prev = 0
  call sb_ok()
  call sb_ensure()
  if (.not. c_associated(ctx)) return
  rc = sb_diag_stream_step_f32(ctx, timestep_number, p, nps, theta, v, u, target_plev, thresh_wind, &
                               thresh_winddir, thresh_windch, thresh_thc, target_time, maxdist, timestep, &
                               sb_prev, hp)
  if (rc /= 0) then
    call sb_fail('stream_step', rc)
  else
    have_prev = hp
  end if
end subroutine stream_step

subroutine stream_end(sb_last, nlons, nlats, output, windspeed, winddir, thc)
  use iso_c_binding
  use sb_f2py_state
  implicit none
  integer :: nlons, nlats
  real(8), dimension(nlons,nlats) :: sb_last
  real, dimension(nlons,nlats,4) :: output
  real, dimension(nlons,nlats) :: windspeed, winddir, thc
  integer(c_int) :: rc
  !f2py integer, intent(in) :: nlons, nlats
  !f2py real(8), intent(inout) :: sb_last
  !f2py real, intent(out) :: output, windspeed, winddir, thc
  call sb_ok()
  call sb_ensure()
  if (.not. c_associated(ctx)) return
  rc = sb_diag_stream_end_f32(ctx, sb_last, output, windspeed, winddir, thc)
  if (rc /= 0) call sb_fail('stream_end', rc)
end subroutine stream_end

subroutine stream_stats(steps, seconds)
  use iso_c_binding
  use sb_f2py_state
  implicit none
  integer :: steps
  real(8) :: seconds(3)
  integer(c_long) :: n
  integer(c_int) :: rc
  !f2py integer, intent(out) :: steps
  !f2py real(8), intent(out) :: seconds
  steps = 0
  seconds = 0
  if (.not. c_associated(ctx)) return
  rc = sb_diag_stream_stats(ctx, n, seconds)
  steps = int(n)
end subroutine stream_stats
