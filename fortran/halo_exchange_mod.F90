!===============================================================================
! halo_exchange_mod -- drop-in for the reference's empty stub of the same name
! (ref: generic/halo_exchange_mod.f90:12-17: "In the present case it doesn't do anything").
!
! swap_bounds(field, halo_size) fills the ghost cells of a field whose interior sits at
! (1+halo_size : n-halo_size) in both dimensions; it is what get_edges calls on the coast
! mask (ref: generic/sea_breeze_diag.f90:342,371) and what a host model calls on theta.
!
! * One process owning the globe (no communicator): the exchange is local -- longitude is
!   periodic, the pole-side ghost rows replicate the edge row (the latitude clamp the
!   global-grid kernels use).  Pure data movement, done where the array lives.
! * One process per GPU owning a latitude band (after sb_context_mod::sb_comm_init): the
!   north / south ghost rows are exchanged with the band neighbours by ncclSend / ncclRecv
!   over RCCL (xGMI) through sb_swap_bounds_f32/f64 of include/seabreeze_hip.h; the two
!   bands at the poles replicate their pole-side edge row; east / west ghost columns are the
!   periodic wrap (a band holds full longitude circles).
!   swap_bounds_dev is the same for a field that already lives on the device (type(c_ptr)):
!   nothing crosses PCIe.
!
! Working precision follows the compile flag like the reference (see sea_breeze_diag_mod.F90).
!===============================================================================
module halo_exchange_mod
  use iso_c_binding
  use sb_context_mod, only : sb_ctx, sb_ensure_ctx, sb_fail, sb_comm_active
  implicit none
  private
  public :: swap_bounds, swap_bounds_dev

#ifdef SB_REAL8
  integer, parameter :: rk = c_double
#define SB_SWAP_BOUNDS     "sb_swap_bounds_f64"
#define SB_SWAP_BOUNDS_DEV "sb_swap_bounds_f64_dev"
#else
  integer, parameter :: rk = c_float
#define SB_SWAP_BOUNDS     "sb_swap_bounds_f32"
#define SB_SWAP_BOUNDS_DEV "sb_swap_bounds_f32_dev"
#endif

  interface
    integer(c_int) function c_swap_bounds(ctx, field, nx, ny, halo) bind(C, name=SB_SWAP_BOUNDS)
      import :: c_ptr, c_int, rk
      type(c_ptr), value :: ctx
      real(rk), intent(inout) :: field(*)
      integer(c_int), value :: nx, ny, halo
    end function
    integer(c_int) function c_swap_bounds_dev(ctx, field, nx, ny, halo, stream) bind(C, name=SB_SWAP_BOUNDS_DEV)
      import :: c_ptr, c_int
      type(c_ptr), value :: ctx, field, stream
      integer(c_int), value :: nx, ny, halo
    end function
  end interface

contains

  subroutine swap_bounds(field, halo_size)
    integer, intent(in) :: halo_size
    real, intent(inout), contiguous :: field(:,:)
    integer :: nxt, nyt, nx, ny, h, j
    integer(c_int) :: rc

    h = halo_size
    nxt = size(field, 1)
    nyt = size(field, 2)
    nx = nxt - 2*h
    ny = nyt - 2*h
    if (h < 1 .or. nx < h .or. ny < 1) return     ! no symmetric ghost frame: nothing to fill
    if (sb_comm_active()) then
      ! a latitude band of a multi-GPU run: ghost rows travel between the devices over RCCL
      rc = c_swap_bounds(sb_ctx, field, int(nx, c_int), int(ny, c_int), int(h, c_int))
      if (rc /= 0) call sb_fail('swap_bounds', rc)
      return
    end if
    ! north / south: replicate the edge rows (poles)
    do j = 1, h
      field(1+h:nx+h, j) = field(1+h:nx+h, 1+h)
      field(1+h:nx+h, ny+h+j) = field(1+h:nx+h, ny+h)
    end do
    ! east / west: periodic, corners included
    field(1:h, :) = field(nx+1:nx+h, :)
    field(nx+h+1:nx+2*h, :) = field(1+h:2*h, :)
  end subroutine swap_bounds

  !> The same for a (nx+2h, ny+2h) field in device memory: ghost cells filled in place on the
  !! context's stream (no synchronisation), with or without a communicator.
  subroutine swap_bounds_dev(field_dev, nx, ny, halo_size)
    type(c_ptr), intent(in) :: field_dev
    integer, intent(in) :: nx, ny, halo_size
    integer(c_int) :: rc
    call sb_ensure_ctx()
    rc = c_swap_bounds_dev(sb_ctx, field_dev, int(nx, c_int), int(ny, c_int), int(halo_size, c_int), c_null_ptr)
    if (rc /= 0) call sb_fail('swap_bounds_dev', rc)
  end subroutine swap_bounds_dev

end module halo_exchange_mod
