!===============================================================================
! dummy_model -- a RUNNABLE counterpart of the reference's outline program
! (ref: generic/dummy_model.f90:24-56, "not intended to work"): the same call order
!
!     atmos_step:  get_edges -> get_dist -> physics -> seabreeze_diag
!
! on the field layout of get_all_fields_mod (ref: generic/get_all_fields_mod.f90:9-19),
! linked against the drop-in modules of this directory, i.e. against libseabreeze_hip.so.
! BASELINE.json configs[0]: a 96x72 synthetic coastline.
!
! Usage:  dummy_model <input.bin> <output.bin> <nsteps>
!   input.bin : nx ny nz halo (4 x int32), then REAL arrays in Fortran order:
!               lon(nx) lat(ny) land_frac(nx,ny) ice_frac(nx,ny) z(nx,ny) sigma(nx,ny)
!               p(nx,ny,nz), then per step: theta(nx,ny) u(nx,ny,nz) v(nx,ny,nz)
!   output.bin: cdist(nx,ny), then per step: sb_con windspeed winddir thc (nx,ny each)
! The tests write input.bin with numpy and compare output.bin with the CPU oracle.
!
! Two deliberate differences from the outline, both documented in INTEGRATION.md:
!  * `physics` passes (theta, mask, z, sigma) in the order of seabreeze_diag's own dummies;
!    the outline permutes them (ref: generic/dummy_model.f90:52-54 vs
!    generic/sea_breeze_diag.f90:55-56, SURVEY.md App. C #4).
!  * the distance field, not the coast mask, is what seabreeze_diag receives as `mask`
!    (ref: generic/sea_breeze_diag.f90:76 "distance form the coast").
!===============================================================================
module model_fields
  implicit none
  integer :: nx, ny, nz, halo_size
  real :: timestep = 24*60.                  ! seconds
  integer :: timestep_number
  real, allocatable :: lon(:), lat(:)
  real, allocatable :: p(:,:,:), u(:,:,:), v(:,:,:)
  real, allocatable :: sb_con(:,:), land_frac(:,:), ice_frac(:,:), windspeed(:,:), winddir(:,:), thc(:,:)
  real, allocatable :: z(:,:), sigma(:,:), theta(:,:), cdist(:,:)
  real, allocatable :: mask(:,:)             ! coast mask with ghost cells, as get_edges writes it
end module model_fields

program dummy_model
  use model_fields
  implicit none
  character(len=512) :: fin, fout, arg
  integer :: nsteps, step, uin, uout
  integer(4) :: hdr(4)

  if (command_argument_count() < 3) then
    print *, 'usage: dummy_model <input.bin> <output.bin> <nsteps>'
    error stop 2
  end if
  call get_command_argument(1, fin)
  call get_command_argument(2, fout)
  call get_command_argument(3, arg)
  read (arg, *) nsteps

  open (newunit=uin, file=trim(fin), access='stream', form='unformatted', status='old')
  read (uin) hdr
  nx = hdr(1); ny = hdr(2); nz = hdr(3); halo_size = hdr(4)
  allocate(lon(nx), lat(ny), p(nx,ny,nz), u(nx,ny,nz), v(nx,ny,nz))
  allocate(sb_con(nx,ny), land_frac(nx,ny), ice_frac(nx,ny), windspeed(nx,ny), winddir(nx,ny), thc(nx,ny))
  allocate(z(nx,ny), sigma(nx,ny), theta(nx,ny), cdist(nx,ny))
  allocate(mask(nx+2*halo_size, ny+2*halo_size))
  read (uin) lon, lat, land_frac, ice_frac, z, sigma, p
  sb_con = 0.; windspeed = 0.; winddir = 0.; thc = 0.

  open (newunit=uout, file=trim(fout), access='stream', form='unformatted', status='replace')
  do step = 1, nsteps
    timestep_number = step
    read (uin) theta, u, v
    call atmos_step()
    if (step == 1) write (uout) cdist
    write (uout) sb_con, windspeed, winddir, thc
  end do
  close (uin); close (uout)
  print '(a,i0,a,i0,a,i0,a,es12.5)', 'dummy_model: ', nsteps, ' steps on ', nx, 'x', ny, &
        ', sum(sb_con) = ', sum(sb_con)

contains

  subroutine atmos_step()
    use sea_breeze_diag_mod, only : get_edges, get_dist
    call get_edges(mask, ice_frac, land_frac, halo_size)
    call get_dist(mask, land_frac, lon, lat, 180, cdist, halo_size)
    call physics(p, u, v, theta, z, sigma, cdist, windspeed, winddir, thc, sb_con, timestep_number, timestep)
  end subroutine atmos_step

  subroutine physics(p, u, v, theta, z, sigma, mask, windspeed, winddir, thc, sb_con, timestep_number, timestep)
    use sea_breeze_diag_mod, only : seabreeze_diag
    real, intent(in), dimension(:,:,:) :: p, u, v
    real, intent(in), dimension(:,:) :: mask, theta, z, sigma
    real, intent(inout), dimension(:,:) :: thc, windspeed, winddir, sb_con
    real, intent(in) :: timestep
    integer, intent(in) :: timestep_number
    call seabreeze_diag(timestep, timestep_number, p, u, v, theta, mask, z, sigma, windspeed, winddir, thc, sb_con)
  end subroutine physics

end program dummy_model
