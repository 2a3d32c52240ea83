!===============================================================================
! sea_breeze_diag_mod -- drop-in Fortran host module for the MI355X implementation.
!
! Same module name, same four public routines and the same dummy-argument order as the
! reference's generic module (ref: generic/sea_breeze_diag.f90:19, :55-56, :273, :375,
! :457), so a host model keeps its `use sea_breeze_diag_mod` / `call seabreeze_diag(...)`
! lines (ref: generic/dummy_model.f90:28,41,52).  Every routine is a thin ISO_C_BINDING
! shim over libseabreeze_hip.so (include/seabreeze_hip.h): descriptor in, contiguous
! buffers + extents out.  No arithmetic of the diagnostic lives here.
!
! Working precision follows the compile flag like the reference:
!   amdflang ...                              -> REAL = 4 bytes -> sb_*_f32
!   amdflang -fdefault-real-8 -DSB_REAL8 ...  -> REAL = 8 bytes -> sb_*_f64
!
! Beyond the reference's four routines (same names, same dummies) the module offers what a host
! model needs to use more than one GPU or to keep its fields on the device:
!   band_seabreeze_diag      one step of a latitude band (host arrays, ghost cells of width h): the
!                            sigma statistics are reduced over all bands and theta's ghost rows are
!                            exchanged over RCCL inside the call (sb_context_mod::sb_comm_init first)
!   seabreeze_diag_dev,      the same two calls with every array a device pointer (type(c_ptr) from
!   band_seabreeze_diag_dev  sb_context_mod::sb_dev_alloc): enqueue only, nothing crosses PCIe
!
! Errors: the reference's generic routines have no error channel; a non-zero status from
! the library ends the run with `error stop` and the library's message.  The UM hook's
! `error` out-argument (ref: UM/vn10.7/sea_breeze_diag.F90:102) is available through
! seabreeze_diag_status.
!
! Array conventions accepted by seabreeze_diag (all assumed-shape, as in the reference):
!   p, u, v                      (nx, ny, nz)
!   windspeed, winddir, thc, sb_con   (nx, ny)
!   theta, mask, z, sigma        (nx, ny)            -> single global domain: latitude is
!                                                       clamped, longitude periodic
!                                (nx+2h, ny+2h)      -> ghost cells of width h all round
!                                                       (what swap_bounds fills); raw reads
!===============================================================================
module sea_breeze_diag_mod
  use iso_c_binding
  use sb_context_mod, only : ctx => sb_ctx, ensure_ctx => sb_ensure_ctx, fail => sb_fail, sb_release_ctx
  implicit none
  private
  public :: seabreeze_diag, seabreeze_diag_status, get_edges, get_dist, sigmoid, sb_shutdown
  public :: band_seabreeze_diag, seabreeze_diag_dev, band_seabreeze_diag_dev
  public :: seabreeze_diag_um, SB_UM_THETA_TO_T0, SB_UM_LEVEL_WALK
  integer, parameter :: SB_UM_THETA_TO_T0 = 1, SB_UM_LEVEL_WALK = 2

#ifdef SB_REAL8
  integer, parameter :: rk = c_double
#define SB_SEABREEZE_DIAG "sb_seabreeze_diag_f64"
#define SB_SIGMOID        "sb_sigmoid_f64"
#define SB_GET_EDGES      "sb_get_edges_f64"
#define SB_GET_DIST       "sb_get_dist_f64"
#define SB_BAND_DIAG      "sb_band_seabreeze_diag_f64"
#define SB_DIAG_DEV       "sb_seabreeze_diag_f64_dev"
#define SB_BAND_DIAG_DEV  "sb_band_seabreeze_diag_f64_dev"
#define SB_DIAG_UM        "sb_seabreeze_diag_um_f64"
#else
  integer, parameter :: rk = c_float
#define SB_SEABREEZE_DIAG "sb_seabreeze_diag_f32"
#define SB_SIGMOID        "sb_sigmoid_f32"
#define SB_GET_EDGES      "sb_get_edges_f32"
#define SB_GET_DIST       "sb_get_dist_f32"
#define SB_BAND_DIAG      "sb_band_seabreeze_diag_f32"
#define SB_DIAG_DEV       "sb_seabreeze_diag_f32_dev"
#define SB_BAND_DIAG_DEV  "sb_band_seabreeze_diag_f32_dev"
#define SB_DIAG_UM        "sb_seabreeze_diag_um_f32"
#endif

  integer(c_int), parameter :: SB_BND_GLOBAL = 1, SB_BND_HALO = 2

  interface get_dist            ! the reference's caller passes the integer literal 180 for
    module procedure get_dist_r ! maxdist (ref: generic/dummy_model.f90:33); accept both
    module procedure get_dist_i
  end interface

  interface
    integer(c_int) function c_seabreeze_diag(ctx, timestep, tn, nx, ny, nz, halo, bnd, p, u, v, theta, mask, &
        z, sigma, ws, wd, thc, sb_con, tun) bind(C, name=SB_SEABREEZE_DIAG)
      import :: c_ptr, c_int, rk
      type(c_ptr), value :: ctx, tun
      real(rk), value :: timestep
      integer(c_int), value :: tn, nx, ny, nz, halo, bnd
      real(rk), intent(in) :: p(*), u(*), v(*), theta(*), mask(*), z(*), sigma(*)
      real(rk), intent(inout) :: ws(*), wd(*), thc(*), sb_con(*)
    end function
    integer(c_int) function c_diag_um(ctx, timestep, tn, nx, ny, nz, halo_s, halo_l, p, u, v, theta, z, sigma, mask, &
        ws, wd, thc, sb_con, flags, error) bind(C, name=SB_DIAG_UM)
      import :: c_ptr, c_int, rk
      type(c_ptr), value :: ctx
      real(rk), value :: timestep
      integer(c_int), value :: tn, nx, ny, nz, halo_s, halo_l, flags
      real(rk), intent(in) :: p(*), u(*), v(*), z(*), sigma(*), mask(*)
      real(rk), intent(inout) :: theta(*), ws(*), wd(*), thc(*), sb_con(*)
      integer(c_int), intent(out) :: error
    end function c_diag_um
    integer(c_int) function c_band_diag(ctx, timestep, tn, nx, ny, nz, halo, p, u, v, theta, mask, &
        z, sigma, ws, wd, thc, sb_con, tun) bind(C, name=SB_BAND_DIAG)
      import :: c_ptr, c_int, rk
      type(c_ptr), value :: ctx, tun
      real(rk), value :: timestep
      integer(c_int), value :: tn, nx, ny, nz, halo
      real(rk), intent(in) :: p(*), u(*), v(*), mask(*), z(*), sigma(*)
      real(rk), intent(inout) :: theta(*), ws(*), wd(*), thc(*), sb_con(*)
    end function
    integer(c_int) function c_diag_dev(ctx, timestep, tn, nx, ny, nz, halo, bnd, p, u, v, theta, mask, &
        z, sigma, ws, wd, thc, sb_con, tun, stream) bind(C, name=SB_DIAG_DEV)
      import :: c_ptr, c_int, rk
      type(c_ptr), value :: ctx, tun, stream, p, u, v, theta, mask, z, sigma, ws, wd, thc, sb_con
      real(rk), value :: timestep
      integer(c_int), value :: tn, nx, ny, nz, halo, bnd
    end function
    integer(c_int) function c_band_diag_dev(ctx, timestep, tn, nx, ny, nz, halo, p, u, v, theta, mask, &
        z, sigma, ws, wd, thc, sb_con, tun, stream) bind(C, name=SB_BAND_DIAG_DEV)
      import :: c_ptr, c_int, rk
      type(c_ptr), value :: ctx, tun, stream, p, u, v, theta, mask, z, sigma, ws, wd, thc, sb_con
      real(rk), value :: timestep
      integer(c_int), value :: tn, nx, ny, nz, halo
    end function
    integer(c_int) function c_sigmoid(ctx, nx, ny, ary, sm) bind(C, name=SB_SIGMOID)
      import :: c_ptr, c_int, rk
      type(c_ptr), value :: ctx
      integer(c_int), value :: nx, ny
      real(rk), intent(in) :: ary(*)
      real(rk), intent(out) :: sm(*)
    end function
    integer(c_int) function c_get_edges(ctx, nx, ny, lsm, ci, rule, bnd, coast) bind(C, name=SB_GET_EDGES)
      import :: c_ptr, c_int, rk
      type(c_ptr), value :: ctx
      integer(c_int), value :: nx, ny, rule, bnd
      real(rk), intent(in) :: lsm(*), ci(*)
      real(rk), intent(out) :: coast(*)
    end function
    integer(c_int) function c_get_dist(ctx, nx, ny, coast, mask, lon, lat, maxdist, kwin, cdist) &
        bind(C, name=SB_GET_DIST)
      import :: c_ptr, c_int, rk
      type(c_ptr), value :: ctx
      integer(c_int), value :: nx, ny, kwin
      real(rk), value :: maxdist
      real(rk), intent(in) :: coast(*), mask(*), lon(*), lat(*)
      real(rk), intent(out) :: cdist(*)
    end function
  end interface

contains

  !> Release the device context (optional; a host model may call it at shutdown).
  subroutine sb_shutdown()
    call sb_release_ctx()
  end subroutine sb_shutdown

  !---------------------------------------------------------------------------
  ! ref: generic/sea_breeze_diag.f90:55-56 -- same dummy order
  !---------------------------------------------------------------------------
  subroutine seabreeze_diag(timestep, timestep_number, &
      p, u, v, theta, mask, z, sigma, windspeed, winddir, thc, sb_con)
    integer, intent(in) :: timestep_number
    real, intent(in), contiguous :: p(:,:,:), u(:,:,:), v(:,:,:), theta(:,:)
    real, intent(in) :: timestep
    real, intent(in), contiguous :: mask(:,:), z(:,:), sigma(:,:)
    real, intent(inout), contiguous :: sb_con(:,:), windspeed(:,:), winddir(:,:), thc(:,:)
    integer :: error
    call seabreeze_diag_status(timestep, timestep_number, p, u, v, theta, mask, z, sigma, &
                               windspeed, winddir, thc, sb_con, error)
    if (error /= 0) call fail('seabreeze_diag', int(error, c_int))
  end subroutine seabreeze_diag

  !> Same call with the UM-style status out-argument instead of stopping
  !! (ref: UM/vn10.7/sea_breeze_diag.F90:55-56,102: error = 0 ok, 1 bad dimensions).
  subroutine seabreeze_diag_status(timestep, timestep_number, &
      p, u, v, theta, mask, z, sigma, windspeed, winddir, thc, sb_con, error)
    integer, intent(in) :: timestep_number
    real, intent(in), contiguous :: p(:,:,:), u(:,:,:), v(:,:,:), theta(:,:)
    real, intent(in) :: timestep
    real, intent(in), contiguous :: mask(:,:), z(:,:), sigma(:,:)
    real, intent(inout), contiguous :: sb_con(:,:), windspeed(:,:), winddir(:,:), thc(:,:)
    integer, intent(out) :: error
    integer(c_int) :: nx, ny, nz, h, bnd
    nx = size(p, 1); ny = size(p, 2); nz = size(p, 3)       ! ref :145-153: the loop extent is p's
    error = 1
    if (nx < 1 .or. ny < 1 .or. nz < 1) return               ! ref: UM copy :198-202
    if (any(shape(u) /= shape(p)) .or. any(shape(v) /= shape(p))) return
    if (any(shape(windspeed) /= [nx, ny]) .or. any(shape(winddir) /= [nx, ny]) .or. &
        any(shape(thc) /= [nx, ny]) .or. any(shape(sb_con) /= [nx, ny])) return
    if (any(shape(mask) /= shape(theta)) .or. any(shape(z) /= shape(theta)) .or. &
        any(shape(sigma) /= shape(theta))) return
    if (size(theta, 1) == nx .and. size(theta, 2) == ny) then
      h = 0; bnd = SB_BND_GLOBAL
    else
      h = (size(theta, 1) - nx) / 2; bnd = SB_BND_HALO
      if (h < 1 .or. size(theta, 1) /= nx + 2*h .or. size(theta, 2) /= ny + 2*h) return
    end if
    call ensure_ctx()
    error = c_seabreeze_diag(ctx, real(timestep, rk), int(timestep_number, c_int), nx, ny, nz, h, bnd, &
                             p, u, v, theta, mask, z, sigma, windspeed, winddir, thc, sb_con, c_null_ptr)
  end subroutine seabreeze_diag_status

  !---------------------------------------------------------------------------
  ! The Unified Model hook's argument list and field bounds
  ! (ref: UM/vn10.7/sea_breeze_diag.F90:55-56,66-117): (theta, z, sigma, mask) in the UM's order, `error` last;
  ! p, u, v and the four state fields on pdims/tdims (nx, ny[, nz]); theta, z, sigma on tdims_s (small halo) and
  ! mask on tdims_l (large halo) -- the halo widths are read off the array shapes.  theta is intent(inout) as in
  ! the UM copy: with SB_UM_THETA_TO_T0 in `flags` it comes back as t0 (:210-211); SB_UM_LEVEL_WALK selects the UM
  ! copy's level rule (:265-274).  flags absent: both, i.e. the UM copy's behaviour.
  !---------------------------------------------------------------------------
  subroutine seabreeze_diag_um(timestep, timestep_number, &
      p, u, v, theta, z, sigma, mask, windspeed, winddir, thc, sb_con, error, flags)
    integer, intent(in) :: timestep_number
    real, intent(in) :: timestep
    real, intent(in), contiguous :: p(:,:,:), u(:,:,:), v(:,:,:), z(:,:), sigma(:,:), mask(:,:)
    real, intent(inout), contiguous :: theta(:,:), sb_con(:,:), windspeed(:,:), winddir(:,:), thc(:,:)
    integer, intent(out) :: error
    integer, intent(in), optional :: flags
    integer(c_int) :: nx, ny, nz, hs, hl, fl, err, rc
    nx = size(p, 1); ny = size(p, 2); nz = size(p, 3)
    error = 1
    if (nx < 1 .or. ny < 1 .or. nz < 1) return               ! ref: UM copy :198-202
    if (any(shape(u) /= shape(p)) .or. any(shape(v) /= shape(p))) return
    if (any(shape(windspeed) /= [nx, ny]) .or. any(shape(winddir) /= [nx, ny]) .or. &
        any(shape(thc) /= [nx, ny]) .or. any(shape(sb_con) /= [nx, ny])) return
    hs = (size(theta, 1) - nx) / 2
    hl = (size(mask, 1) - nx) / 2
    if (hs < 0 .or. hl < hs) return
    if (any(shape(theta) /= [nx + 2*hs, ny + 2*hs]) .or. any(shape(z) /= shape(theta)) .or. &
        any(shape(sigma) /= shape(theta)) .or. any(shape(mask) /= [nx + 2*hl, ny + 2*hl])) return
    fl = SB_UM_THETA_TO_T0 + SB_UM_LEVEL_WALK
    if (present(flags)) fl = int(flags, c_int)
    call ensure_ctx()
    rc = c_diag_um(ctx, real(timestep, rk), int(timestep_number, c_int), nx, ny, nz, hs, hl, p, u, v, theta, z, sigma, &
                   mask, windspeed, winddir, thc, sb_con, fl, err)
    if (rc /= 0) call fail('seabreeze_diag_um', rc)
    error = int(err)
  end subroutine seabreeze_diag_um

  !---------------------------------------------------------------------------
  ! One step of a latitude band of a multi-GPU run (one process per GPU, sb_comm_init done).
  ! Arguments as seabreeze_diag for this band's rows; theta, mask, z, sigma carry h ghost cells
  ! all round ((nx+2h, ny+2h), h >= 1).  mask, z, sigma are static: fill their ghost cells once
  ! with halo_exchange_mod::swap_bounds.  theta's ghost cells are filled inside the call (and
  ! returned), overlapped with the kernels that do not need them; the sigma statistics are
  ! reduced over all bands, so every band reproduces the rows of the single-domain result.
  ! ref: generic/sea_breeze_diag.f90:55-271 + the exchange its UM twin makes,
  !      UM/vn10.7/sea_breeze_diag.F90:408-410
  !---------------------------------------------------------------------------
  subroutine band_seabreeze_diag(timestep, timestep_number, &
      p, u, v, theta, mask, z, sigma, windspeed, winddir, thc, sb_con)
    integer, intent(in) :: timestep_number
    real, intent(in), contiguous :: p(:,:,:), u(:,:,:), v(:,:,:)
    real, intent(in) :: timestep
    real, intent(inout), contiguous :: theta(:,:)
    real, intent(in), contiguous :: mask(:,:), z(:,:), sigma(:,:)
    real, intent(inout), contiguous :: sb_con(:,:), windspeed(:,:), winddir(:,:), thc(:,:)
    integer(c_int) :: nx, ny, nz, h, rc
    nx = size(p, 1); ny = size(p, 2); nz = size(p, 3)
    h = (size(theta, 1) - nx) / 2
    if (h < 1 .or. size(theta, 1) /= nx + 2*h .or. size(theta, 2) /= ny + 2*h .or. &
        any(shape(mask) /= shape(theta)) .or. any(shape(z) /= shape(theta)) .or. any(shape(sigma) /= shape(theta)) .or. &
        any(shape(u) /= shape(p)) .or. any(shape(v) /= shape(p)) .or. &
        any(shape(windspeed) /= [nx, ny]) .or. any(shape(winddir) /= [nx, ny]) .or. &
        any(shape(thc) /= [nx, ny]) .or. any(shape(sb_con) /= [nx, ny])) &
      call fail('band_seabreeze_diag: theta, mask, z, sigma must be (nx+2h, ny+2h) with h >= 1', 1_c_int)
    call ensure_ctx()
    rc = c_band_diag(ctx, real(timestep, rk), int(timestep_number, c_int), nx, ny, nz, h, &
                     p, u, v, theta, mask, z, sigma, windspeed, winddir, thc, sb_con, c_null_ptr)
    if (rc /= 0) call fail('band_seabreeze_diag', rc)
  end subroutine band_seabreeze_diag

  !---------------------------------------------------------------------------
  ! Device-resident forms: every array argument is a device pointer (sb_context_mod::sb_dev_alloc),
  ! Fortran order as above; the call only enqueues on the context's stream
  ! (sb_context_mod::sb_device_synchronize waits).  halo = 0: single global domain.
  !---------------------------------------------------------------------------
  subroutine seabreeze_diag_dev(timestep, timestep_number, nx, ny, nz, halo, &
      p, u, v, theta, mask, z, sigma, windspeed, winddir, thc, sb_con)
    integer, intent(in) :: timestep_number, nx, ny, nz, halo
    real, intent(in) :: timestep
    type(c_ptr), intent(in) :: p, u, v, theta, mask, z, sigma, windspeed, winddir, thc, sb_con
    integer(c_int) :: rc, bnd
    bnd = SB_BND_GLOBAL
    if (halo > 0) bnd = SB_BND_HALO
    call ensure_ctx()
    rc = c_diag_dev(ctx, real(timestep, rk), int(timestep_number, c_int), int(nx, c_int), int(ny, c_int), &
                    int(nz, c_int), int(halo, c_int), bnd, p, u, v, theta, mask, z, sigma, &
                    windspeed, winddir, thc, sb_con, c_null_ptr, c_null_ptr)
    if (rc /= 0) call fail('seabreeze_diag_dev', rc)
  end subroutine seabreeze_diag_dev

  subroutine band_seabreeze_diag_dev(timestep, timestep_number, nx, ny, nz, halo, &
      p, u, v, theta, mask, z, sigma, windspeed, winddir, thc, sb_con)
    integer, intent(in) :: timestep_number, nx, ny, nz, halo
    real, intent(in) :: timestep
    type(c_ptr), intent(in) :: p, u, v, theta, mask, z, sigma, windspeed, winddir, thc, sb_con
    integer(c_int) :: rc
    call ensure_ctx()
    rc = c_band_diag_dev(ctx, real(timestep, rk), int(timestep_number, c_int), int(nx, c_int), int(ny, c_int), &
                         int(nz, c_int), int(halo, c_int), p, u, v, theta, mask, z, sigma, &
                         windspeed, winddir, thc, sb_con, c_null_ptr, c_null_ptr)
    if (rc /= 0) call fail('band_seabreeze_diag_dev', rc)
  end subroutine band_seabreeze_diag_dev

  !---------------------------------------------------------------------------
  ! ref: generic/sea_breeze_diag.f90:273-373.  landfrac/icefrac are (nx, ny); the coast
  ! mask is written into mask(1+halo_size : nx+halo_size, 1+halo_size : ny+halo_size)
  ! exactly as :369 does, then the ghost cells are filled by swap_bounds (:371).
  !---------------------------------------------------------------------------
  subroutine get_edges(mask, icefrac, landfrac, halo_size)
    use halo_exchange_mod, only : swap_bounds
    integer, intent(in) :: halo_size
    real, intent(out) :: mask(:,:)
    real, intent(in), contiguous :: landfrac(:,:), icefrac(:,:)
    real, allocatable :: coast(:,:)
    integer(c_int) :: nx, ny, rc
    nx = size(landfrac, 1); ny = size(landfrac, 2)
    if (size(mask, 1) < nx + halo_size .or. size(mask, 2) < ny + halo_size) &
      call fail('get_edges: mask smaller than landfrac + halo_size', 1_c_int)
    allocate(coast(nx, ny))
    call ensure_ctx()
    rc = c_get_edges(ctx, nx, ny, landfrac, icefrac, 1_c_int, SB_BND_GLOBAL, coast)
    if (rc /= 0) call fail('get_edges', rc)
    mask = 0.
    mask(1+halo_size:nx+halo_size, 1+halo_size:ny+halo_size) = coast
    deallocate(coast)
    call swap_bounds(mask, halo_size)
  end subroutine get_edges

  !---------------------------------------------------------------------------
  ! ref: generic/sea_breeze_diag.f90:375-455.  coast is either (nlons, nlats) or the
  ! ghost-celled mask get_edges wrote (interior at offset halo_size, as the reference's
  ! caller passes it, ref: generic/dummy_model.f90:32-34); the window is +-halo_size cells
  ! (:422,:425) and the sign comes from landfrac > 0 (:442).
  !---------------------------------------------------------------------------
  subroutine get_dist_r(coast, landfrac, lon, lat, maxdist, cdist, halo_size)
    integer, intent(in) :: halo_size
    real, intent(in), contiguous :: lon(:), lat(:), landfrac(:,:)
    real, intent(in) :: coast(:,:)
    real, intent(in) :: maxdist
    real, intent(out), contiguous :: cdist(:,:)
    real, allocatable :: c2(:,:)
    integer(c_int) :: nx, ny, rc
    nx = size(lon, 1); ny = size(lat, 1)
    if (any(shape(landfrac) /= [nx, ny]) .or. any(shape(cdist) /= [nx, ny])) &
      call fail('get_dist: landfrac/cdist must be (size(lon), size(lat))', 1_c_int)
    allocate(c2(nx, ny))
    if (size(coast, 1) == nx .and. size(coast, 2) == ny) then
      c2 = coast
    else if (size(coast, 1) >= nx + halo_size .and. size(coast, 2) >= ny + halo_size) then
      c2 = coast(1+halo_size:nx+halo_size, 1+halo_size:ny+halo_size)
    else
      call fail('get_dist: coast has neither the grid shape nor the ghost-celled shape', 1_c_int)
    end if
    call ensure_ctx()
    rc = c_get_dist(ctx, nx, ny, c2, landfrac, lon, lat, real(maxdist, rk), int(halo_size, c_int), cdist)
    if (rc /= 0) call fail('get_dist', rc)
    deallocate(c2)
  end subroutine get_dist_r

  subroutine get_dist_i(coast, landfrac, lon, lat, maxdist, cdist, halo_size)
    integer, intent(in) :: halo_size, maxdist
    real, intent(in), contiguous :: lon(:), lat(:), landfrac(:,:)
    real, intent(in) :: coast(:,:)
    real, intent(out), contiguous :: cdist(:,:)
    call get_dist_r(coast, landfrac, lon, lat, real(maxdist), cdist, halo_size)
  end subroutine get_dist_i

  !---------------------------------------------------------------------------
  ! ref: generic/sea_breeze_diag.f90:457-481
  !---------------------------------------------------------------------------
  subroutine sigmoid(ary, sm)
    real, intent(in), contiguous :: ary(:,:)
    real, intent(out), contiguous :: sm(:,:)
    integer(c_int) :: rc
    if (any(shape(sm) /= shape(ary))) call fail('sigmoid: shape mismatch', 1_c_int)
    call ensure_ctx()
    rc = c_sigmoid(ctx, int(size(ary, 1), c_int), int(size(ary, 2), c_int), ary, sm)
    if (rc /= 0) call fail('sigmoid', rc)
  end subroutine sigmoid

end module sea_breeze_diag_mod
