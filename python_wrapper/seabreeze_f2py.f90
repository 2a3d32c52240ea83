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
  call sb_ensure()
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
  call sb_ensure()
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
  call sb_ensure()
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
  call sb_ensure()
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
