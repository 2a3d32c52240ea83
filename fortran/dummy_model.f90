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
! Usage:  dummy_model <input.bin> <output.bin> <nsteps> [mode ...]
!   input.bin : nx ny nz halo (4 x int32), then REAL arrays in Fortran order:
!               lon(nx) lat(ny) land_frac(nx,ny) ice_frac(nx,ny) z(nx,ny) sigma(nx,ny)
!               p(nx,ny,nz), then per step: theta(nx,ny) u(nx,ny,nz) v(nx,ny,nz)
!   output.bin: cdist(nx,ny), then per step: sb_con windspeed winddir thc (nx,ny each)
!   mode (optional):
!     status                     the argument checks of seabreeze_diag_status / seabreeze_diag_um (error = 1 for
!                                inconsistent shapes, before any device work: runs without a GPU)
!     um                         the Unified Model hook's argument order and bounds (seabreeze_diag_um) on the
!                                sub-domain the rim of the input fields leaves
!     host                       the reference's call shape: host arrays in, host arrays out (default)
!     dev                        the fields live on the device (sb_dev_alloc); per step only theta, u, v go up
!                                and the four outputs come back; seabreeze_diag_dev
!     band <rank> <nranks> <idfile>
!                                one process per GPU, this one owns latitude band <rank> of <nranks>: the
!                                RCCL id travels through <idfile> (a real host model would MPI_Bcast it),
!                                static fields get their ghost rows once through swap_bounds, every step is one
!                                band_seabreeze_diag; output.bin then holds this band's rows only
!                                (cdist(nx,nyl), then per step the four (nx,nyl) fields)
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
  use iso_c_binding
  use model_fields
  implicit none
  character(len=512) :: fin, fout, arg, mode, idfile
  integer :: nsteps, step, uin, uout, rank, nranks
  integer(4) :: hdr(4)

  if (command_argument_count() < 3) then
    print *, 'usage: dummy_model <input.bin> <output.bin> <nsteps> [host | dev | um | status | band <rank> <nranks> <idfile>]'
    error stop 2
  end if
  call get_command_argument(1, fin)
  call get_command_argument(2, fout)
  call get_command_argument(3, arg)
  read (arg, *) nsteps
  mode = 'host'
  if (command_argument_count() >= 4) call get_command_argument(4, mode)
  rank = 0; nranks = 1
  if (trim(mode) == 'band') then
    if (command_argument_count() < 7) error stop 'band mode needs <rank> <nranks> <idfile>'
    call get_command_argument(5, arg); read (arg, *) rank
    call get_command_argument(6, arg); read (arg, *) nranks
    call get_command_argument(7, idfile)
  end if

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
  select case (trim(mode))
  case ('host')
    do step = 1, nsteps
      timestep_number = step
      read (uin) theta, u, v
      call atmos_step()
      if (step == 1) write (uout) cdist
      write (uout) sb_con, windspeed, winddir, thc
    end do
  case ('dev')
    call run_device_resident()
  case ('band')
    call run_band()
  case ('um')
    call run_um()
  case ('status')
    call check_status()
  case default
    error stop 'unknown mode'
  end select
  close (uin); close (uout)
  print '(a,a,a,i0,a,i0,a,i0,a,es12.5)', 'dummy_model (', trim(mode), '): ', nsteps, ' steps on ', nx, 'x', ny, &
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

  !---------------------------------------------------------------------------
  ! fields resident on the device: static ones go up once, per step theta, u, v
  !---------------------------------------------------------------------------
  subroutine run_device_resident()
    use sea_breeze_diag_mod, only : get_edges, get_dist, seabreeze_diag_dev
    use sb_context_mod, only : sb_dev_alloc, sb_dev_free, sb_dev_upload, sb_dev_download
    type(c_ptr) :: d_p, d_u, d_v, d_th, d_mask, d_z, d_sg, d_ws, d_wd, d_thc, d_sb
    integer(c_size_t) :: b2, b3
    real, target, allocatable :: buf2(:,:), buf3(:,:,:)
    b2 = int(nx, c_size_t) * ny * (storage_size(theta) / 8)
    b3 = b2 * nz
    call get_edges(mask, ice_frac, land_frac, halo_size)
    call get_dist(mask, land_frac, lon, lat, 180, cdist, halo_size)
    write (uout) cdist
    d_p = sb_dev_alloc(b3); d_u = sb_dev_alloc(b3); d_v = sb_dev_alloc(b3)
    d_th = sb_dev_alloc(b2); d_mask = sb_dev_alloc(b2); d_z = sb_dev_alloc(b2); d_sg = sb_dev_alloc(b2)
    d_ws = sb_dev_alloc(b2); d_wd = sb_dev_alloc(b2); d_thc = sb_dev_alloc(b2); d_sb = sb_dev_alloc(b2)
    allocate(buf2(nx,ny), buf3(nx,ny,nz))
    buf3 = p;      call sb_dev_upload(d_p, c_loc(buf3), b3)
    buf2 = cdist;  call sb_dev_upload(d_mask, c_loc(buf2), b2)
    buf2 = z;      call sb_dev_upload(d_z, c_loc(buf2), b2)
    buf2 = sigma;  call sb_dev_upload(d_sg, c_loc(buf2), b2)
    buf2 = 0.
    call sb_dev_upload(d_ws, c_loc(buf2), b2); call sb_dev_upload(d_wd, c_loc(buf2), b2)
    call sb_dev_upload(d_thc, c_loc(buf2), b2); call sb_dev_upload(d_sb, c_loc(buf2), b2)
    do step = 1, nsteps
      read (uin) theta, u, v
      buf2 = theta; call sb_dev_upload(d_th, c_loc(buf2), b2)
      buf3 = u;     call sb_dev_upload(d_u, c_loc(buf3), b3)
      buf3 = v;     call sb_dev_upload(d_v, c_loc(buf3), b3)
      call seabreeze_diag_dev(timestep, step, nx, ny, nz, 0, d_p, d_u, d_v, d_th, d_mask, d_z, d_sg, &
                              d_ws, d_wd, d_thc, d_sb)
      call sb_dev_download(c_loc(buf2), d_sb, b2);  sb_con = buf2
      call sb_dev_download(c_loc(buf2), d_ws, b2);  windspeed = buf2
      call sb_dev_download(c_loc(buf2), d_wd, b2);  winddir = buf2
      call sb_dev_download(c_loc(buf2), d_thc, b2); thc = buf2
      write (uout) sb_con, windspeed, winddir, thc
    end do
    call sb_dev_free(d_p); call sb_dev_free(d_u); call sb_dev_free(d_v); call sb_dev_free(d_th)
    call sb_dev_free(d_mask); call sb_dev_free(d_z); call sb_dev_free(d_sg)
    call sb_dev_free(d_ws); call sb_dev_free(d_wd); call sb_dev_free(d_thc); call sb_dev_free(d_sb)
  end subroutine run_device_resident

  !---------------------------------------------------------------------------
  ! The Unified Model hook's argument list and bounds (ref: UM/vn10.7/sea_breeze_diag.F90:55-56,66-117):
  ! the rim of the input fields serves as the ghost cells of a sub-domain -- theta, z, sigma with a small
  ! halo of halo_size + 1 cells (the widest window of a coastal-band cell), the coast distance with a large
  ! halo of halo_size + 3 -- and the diagnostic is called with (theta, z, sigma, mask) in the UM's order and
  ! its `error` status.  Output: the sub-domain's fields, per step.
  !---------------------------------------------------------------------------
  subroutine run_um()
    use sea_breeze_diag_mod, only : get_edges, get_dist, seabreeze_diag_um
    integer :: hs, hl, nxi, nyi, error
    real, allocatable :: p_i(:,:,:), u_i(:,:,:), v_i(:,:,:), th_s(:,:), z_s(:,:), sg_s(:,:)
    real, allocatable :: ws_i(:,:), wd_i(:,:), thc_i(:,:), sb_i(:,:)
    hs = halo_size + 1; hl = halo_size + 3
    nxi = nx - 2*hl; nyi = ny - 2*hl
    if (nxi < 1 .or. nyi < 1) error stop 'um mode: grid too small for the ghost frame'
    call get_edges(mask, ice_frac, land_frac, halo_size)
    call get_dist(mask, land_frac, lon, lat, 180, cdist, halo_size)
    write (uout) cdist
    allocate(p_i(nxi,nyi,nz), u_i(nxi,nyi,nz), v_i(nxi,nyi,nz))
    allocate(th_s(nxi+2*hs,nyi+2*hs), z_s(nxi+2*hs,nyi+2*hs), sg_s(nxi+2*hs,nyi+2*hs))
    allocate(ws_i(nxi,nyi), wd_i(nxi,nyi), thc_i(nxi,nyi), sb_i(nxi,nyi))
    ws_i = 0.; wd_i = 0.; thc_i = 0.; sb_i = 0.
    p_i = p(1+hl:nx-hl, 1+hl:ny-hl, :)
    z_s = z(1+hl-hs:nx-hl+hs, 1+hl-hs:ny-hl+hs)
    sg_s = sigma(1+hl-hs:nx-hl+hs, 1+hl-hs:ny-hl+hs)
    do step = 1, nsteps
      read (uin) theta, u, v
      u_i = u(1+hl:nx-hl, 1+hl:ny-hl, :)
      v_i = v(1+hl:nx-hl, 1+hl:ny-hl, :)
      th_s = theta(1+hl-hs:nx-hl+hs, 1+hl-hs:ny-hl+hs)
      call seabreeze_diag_um(timestep, step, p_i, u_i, v_i, th_s, z_s, sg_s, cdist, ws_i, wd_i, thc_i, sb_i, error)
      if (error /= 0) error stop 'um mode: seabreeze_diag_um reported an error'
      write (uout) sb_i, ws_i, wd_i, thc_i, th_s
    end do
    sb_con = 0.
    sb_con(1+hl:nx-hl, 1+hl:ny-hl) = sb_i
  end subroutine run_um

  !---------------------------------------------------------------------------
  ! Argument checks of the status-returning entry points: they answer error = 1 before any device work
  ! (ref: UM/vn10.7/sea_breeze_diag.F90:102,198-202), so this mode runs without a GPU as well.
  !---------------------------------------------------------------------------
  subroutine check_status()
    use sea_breeze_diag_mod, only : seabreeze_diag_status, seabreeze_diag_um
    real, allocatable :: th_big(:,:), mk_big(:,:), ws_small(:,:), mk_small(:,:), th_s(:,:), z_s(:,:), sg_s(:,:)
    integer :: e_mixed, e_state, e_um
    u = 0.; v = 0.; theta = 280.; cdist = 0.
    ! the outline's own declaration: mask, theta (nx+halo_size, ny+halo_size) beside z, sigma (nx, ny)
    ! (ref: generic/get_all_fields_mod.f90:17-19)
    allocate(th_big(nx+halo_size, ny+halo_size), mk_big(nx+halo_size, ny+halo_size))
    th_big = 280.; mk_big = 0.
    call seabreeze_diag_status(timestep, 1, p, u, v, th_big, mk_big, z, sigma, windspeed, winddir, thc, sb_con, e_mixed)
    ! a state field of another shape than p's horizontal extent
    allocate(ws_small(nx-1, ny)); ws_small = 0.
    call seabreeze_diag_status(timestep, 1, p, u, v, theta, cdist, z, sigma, ws_small, winddir, thc, sb_con, e_state)
    ! UM bounds: the large halo (mask) must not be narrower than the small one (theta, z, sigma)
    allocate(th_s(nx+4, ny+4), z_s(nx+4, ny+4), sg_s(nx+4, ny+4), mk_small(nx+2, ny+2))
    th_s = 280.; z_s = 0.; sg_s = 1.; mk_small = 0.
    call seabreeze_diag_um(timestep, 1, p, u, v, th_s, z_s, sg_s, mk_small, windspeed, winddir, thc, sb_con, e_um)
    print '(a,3(1x,i0))', 'status:', e_mixed, e_state, e_um
    write (uout) real(e_mixed), real(e_state), real(e_um)
  end subroutine check_status

  !---------------------------------------------------------------------------
  ! one latitude band of a multi-GPU run
  !---------------------------------------------------------------------------
  subroutine run_band()
    use sea_breeze_diag_mod, only : get_edges, get_dist, band_seabreeze_diag
    use halo_exchange_mod, only : swap_bounds
    use sb_context_mod, only : sb_comm_get_unique_id, sb_comm_init, sb_comm_finalize, sb_last_step_report
    integer(c_signed_char) :: id(128)
    integer :: n_launch, n_rccl, n_group, n_copy
    integer :: r0, r1, nyl, h, base, rem, uid, ios, tries
    real, allocatable :: th_b(:,:), mask_b(:,:), z_b(:,:), sg_b(:,:)
    real, allocatable :: p_b(:,:,:), u_b(:,:,:), v_b(:,:,:)
    real, allocatable :: ws_b(:,:), wd_b(:,:), thc_b(:,:), sb_b(:,:)
    logical :: there

    ! get_dist looks +-halo_size cells around every coast cell, so a band cell can be halo_size + 1 cells from
    ! the nearest cell of the other class: that is the ghost width the contrast window needs
    h = halo_size + 1
    ! the setup chain works on the global static fields and runs before the communicator exists
    ! (swap_bounds inside get_edges is then the single-domain fill)
    call get_edges(mask, ice_frac, land_frac, halo_size)
    call get_dist(mask, land_frac, lon, lat, 180, cdist, halo_size)
    ! the RCCL id: rank 0 makes it, the others wait for the file
    if (rank == 0) then
      call sb_comm_get_unique_id(id)
      open (newunit=uid, file=trim(idfile)//'.tmp', access='stream', form='unformatted', status='replace')
      write (uid) id
      close (uid)
      call rename(trim(idfile)//'.tmp', trim(idfile))
    else
      do tries = 1, 6000
        inquire (file=trim(idfile), exist=there)
        if (there) exit
        call sleep_ms(10)
      end do
      if (.not. there) error stop 'band mode: the id file never appeared'
      open (newunit=uid, file=trim(idfile), access='stream', form='unformatted', status='old', iostat=ios)
      read (uid) id
      close (uid)
    end if
    call sb_comm_init(id, rank, nranks)
    ! contiguous, near-equal bands
    base = ny / nranks; rem = mod(ny, nranks)
    r0 = rank * base + min(rank, rem) + 1
    nyl = base + merge(1, 0, rank < rem)
    r1 = r0 + nyl - 1
    allocate(th_b(nx+2*h, nyl+2*h), mask_b(nx+2*h, nyl+2*h), z_b(nx+2*h, nyl+2*h), sg_b(nx+2*h, nyl+2*h))
    allocate(p_b(nx,nyl,nz), u_b(nx,nyl,nz), v_b(nx,nyl,nz))
    allocate(ws_b(nx,nyl), wd_b(nx,nyl), thc_b(nx,nyl), sb_b(nx,nyl))
    ws_b = 0.; wd_b = 0.; thc_b = 0.; sb_b = 0.
    mask_b = 0.; z_b = 0.; sg_b = 0.; th_b = 0.
    mask_b(1+h:nx+h, 1+h:nyl+h) = cdist(:, r0:r1)
    z_b(1+h:nx+h, 1+h:nyl+h) = z(:, r0:r1)
    sg_b(1+h:nx+h, 1+h:nyl+h) = sigma(:, r0:r1)
    ! static fields: ghost cells once (RCCL send/recv with the band neighbours, poles and E-W locally)
    call swap_bounds(mask_b, h)
    call swap_bounds(z_b, h)
    call swap_bounds(sg_b, h)
    p_b = p(:, r0:r1, :)
    write (uout) cdist(:, r0:r1)
    do step = 1, nsteps
      read (uin) theta, u, v
      th_b(1+h:nx+h, 1+h:nyl+h) = theta(:, r0:r1)
      u_b = u(:, r0:r1, :)
      v_b = v(:, r0:r1, :)
      call band_seabreeze_diag(timestep, step, p_b, u_b, v_b, th_b, mask_b, z_b, sg_b, ws_b, wd_b, thc_b, sb_b)
      write (uout) sb_b, ws_b, wd_b, thc_b
    end do
    sb_con = 0.
    sb_con(:, r0:r1) = sb_b
    call sb_last_step_report(n_launch, n_rccl, n_group, n_copy)
    print '(a,4(1x,i0))', 'band step enqueued (launches, RCCL ops, RCCL groups, copies):', n_launch, n_rccl, n_group, n_copy
    call sb_comm_finalize()
  end subroutine run_band

  subroutine sleep_ms(ms)
    integer, intent(in) :: ms
    integer(8) :: c0, c1, rate
    call system_clock(c0, rate)
    do
      call system_clock(c1)
      if ((c1 - c0) * 1000 >= int(ms, 8) * rate) exit
    end do
  end subroutine sleep_ms

end program dummy_model
