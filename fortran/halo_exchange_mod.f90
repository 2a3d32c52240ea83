!===============================================================================
! halo_exchange_mod -- drop-in for the reference's empty stub of the same name
! (ref: generic/halo_exchange_mod.f90:12-17: "In the present case it doesn't do anything").
!
! swap_bounds(field, halo_size) fills the ghost cells of a field whose interior sits at
! (1+halo_size : n-halo_size) in both dimensions.
!
! Single-process form (this file): the process owns the whole globe, so the exchange is
! local -- longitude is periodic, the pole-side ghost rows replicate the edge row (the
! latitude clamp the global-grid kernels use).  Pure data movement, no arithmetic.
!
! Multi-GPU form: one process per GPU owns a latitude band; the north/south ghost rows come
! from the band neighbours over RCCL.  That path keeps the fields on the device and is
! driven through seabreeze_param_amd/bands.py (torch.distributed "nccl" = RCCL) and the
! sb_sigma_moments / sb_use_gathered_moments entry points of include/seabreeze_hip.h.
!===============================================================================
module halo_exchange_mod
  implicit none
  private
  public :: swap_bounds

contains

  subroutine swap_bounds(field, halo_size)
    integer, intent(in) :: halo_size
    real, intent(inout) :: field(:,:)
    integer :: nxt, nyt, nx, ny, h, j

    h = halo_size
    nxt = size(field, 1)
    nyt = size(field, 2)
    nx = nxt - 2*h
    ny = nyt - 2*h
    if (h < 1 .or. nx < h .or. ny < 1) return     ! no symmetric ghost frame: nothing to fill
    ! north / south: replicate the edge rows (poles)
    do j = 1, h
      field(1+h:nx+h, j) = field(1+h:nx+h, 1+h)
      field(1+h:nx+h, ny+h+j) = field(1+h:nx+h, ny+h)
    end do
    ! east / west: periodic, corners included
    field(1:h, :) = field(nx+1:nx+h, :)
    field(nx+h+1:nx+2*h, :) = field(1+h:2*h, :)
  end subroutine swap_bounds

end module halo_exchange_mod
