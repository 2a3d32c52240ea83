!===============================================================================
! sb_context_mod -- the one device context the Fortran host modules share, and what a
! host model needs besides the reference's four routines to run the diagnostic across the
! GPUs of a node or to keep its fields on the device:
!
!   sb_comm_get_unique_id / sb_comm_init / sb_comm_finalize
!        the latitude-band communicator (RCCL over xGMI, one process per GPU).  Rank 0 makes
!        the 128-byte id, the host model hands it to every rank by its own means (MPI_Bcast in
!        a real model; a file in fortran/dummy_model.f90), every rank joins.  From then on
!        halo_exchange_mod::swap_bounds exchanges ghost rows with the band neighbours
!        (ref: generic/halo_exchange_mod.f90:12-17, call sites generic/sea_breeze_diag.f90:342,371)
!        and sea_breeze_diag_mod::band_seabreeze_diag runs one step of a band.
!   sb_dev_alloc / sb_dev_free / sb_dev_upload / sb_dev_download
!        device arrays as type(c_ptr), for the *_dev entry points of sea_breeze_diag_mod
!        (fields stay resident between calls: nothing crosses PCIe in a step).
!
! Thin ISO_C_BINDING shims over include/seabreeze_hip.h; no arithmetic lives here.
!===============================================================================
module sb_context_mod
  use iso_c_binding
  implicit none
  private
  public :: sb_ctx, sb_ensure_ctx, sb_fail, sb_release_ctx
  public :: sb_comm_get_unique_id, sb_comm_init, sb_comm_finalize, sb_comm_active, sb_comm_rank
  public :: sb_set_static_sigma, sb_last_step_report
  public :: sb_dev_alloc, sb_dev_free, sb_dev_upload, sb_dev_download, sb_device_synchronize

  type(c_ptr), save :: sb_ctx = c_null_ptr

  interface
    integer(c_int) function c_sb_create(ctx, device) bind(C, name="sb_create")
      import :: c_ptr, c_int
      type(c_ptr), intent(out) :: ctx
      integer(c_int), value :: device
    end function
    integer(c_int) function c_sb_destroy(ctx) bind(C, name="sb_destroy")
      import :: c_ptr, c_int
      type(c_ptr), value :: ctx
    end function
    type(c_ptr) function c_sb_last_error(ctx) bind(C, name="sb_last_error")
      import :: c_ptr
      type(c_ptr), value :: ctx
    end function
    integer(c_int) function c_sb_synchronize(ctx) bind(C, name="sb_synchronize")
      import :: c_ptr, c_int
      type(c_ptr), value :: ctx
    end function
    integer(c_int) function c_comm_get_unique_id(id) bind(C, name="sb_comm_get_unique_id")
      import :: c_int, c_signed_char
      integer(c_signed_char), intent(out) :: id(128)
    end function
    integer(c_int) function c_comm_init(ctx, id, rank, nranks) bind(C, name="sb_comm_init")
      import :: c_ptr, c_int, c_signed_char
      type(c_ptr), value :: ctx
      integer(c_signed_char), intent(in) :: id(128)
      integer(c_int), value :: rank, nranks
    end function
    integer(c_int) function c_comm_finalize(ctx) bind(C, name="sb_comm_finalize")
      import :: c_ptr, c_int
      type(c_ptr), value :: ctx
    end function
    integer(c_int) function c_set_static_sigma(ctx, on) bind(C, name="sb_set_static_sigma")
      import :: c_ptr, c_int
      type(c_ptr), value :: ctx
      integer(c_int), value :: on
    end function
    integer(c_int) function c_last_step_report(ctx, report) bind(C, name="sb_last_step_report")
      import :: c_ptr, c_int
      type(c_ptr), value :: ctx
      integer(c_int), intent(out) :: report(4)
    end function
    integer(c_int) function c_comm_rank(ctx, rank, nranks) bind(C, name="sb_comm_rank")
      import :: c_ptr, c_int
      type(c_ptr), value :: ctx
      integer(c_int), intent(out) :: rank, nranks
    end function
    integer(c_int) function c_dev_malloc(ctx, nbytes, dptr) bind(C, name="sb_device_malloc")
      import :: c_ptr, c_int, c_size_t
      type(c_ptr), value :: ctx
      integer(c_size_t), value :: nbytes
      type(c_ptr), intent(out) :: dptr
    end function
    integer(c_int) function c_dev_free(ctx, dptr) bind(C, name="sb_device_free")
      import :: c_ptr, c_int
      type(c_ptr), value :: ctx, dptr
    end function
    integer(c_int) function c_dev_upload(ctx, dst, src, nbytes) bind(C, name="sb_device_upload")
      import :: c_ptr, c_int, c_size_t
      type(c_ptr), value :: ctx, dst, src
      integer(c_size_t), value :: nbytes
    end function
    integer(c_int) function c_dev_download(ctx, dst, src, nbytes) bind(C, name="sb_device_download")
      import :: c_ptr, c_int, c_size_t
      type(c_ptr), value :: ctx, dst, src
      integer(c_size_t), value :: nbytes
    end function
  end interface

contains

  !> Create the context on first use (current HIP device).  Without a gfx950 device this ends the run:
  !! there is no CPU fallback.
  subroutine sb_ensure_ctx()
    integer(c_int) :: rc
    if (.not. c_associated(sb_ctx)) then
      rc = c_sb_create(sb_ctx, -1_c_int)
      if (rc /= 0) call sb_fail('sb_create', rc)
    end if
  end subroutine sb_ensure_ctx

  subroutine sb_release_ctx()
    integer(c_int) :: rc
    if (c_associated(sb_ctx)) rc = c_sb_destroy(sb_ctx)
    sb_ctx = c_null_ptr
  end subroutine sb_release_ctx

  subroutine sb_fail(what, rc)
    character(len=*), intent(in) :: what
    integer(c_int), intent(in) :: rc
    character(kind=c_char), pointer :: msg(:)
    character(len=512) :: text
    type(c_ptr) :: cp
    integer :: i
    text = ''
    cp = c_sb_last_error(sb_ctx)
    if (c_associated(cp)) then
      call c_f_pointer(cp, msg, [512])
      do i = 1, 512
        if (msg(i) == c_null_char) exit
        text(i:i) = msg(i)
      end do
    end if
    write (*, '(a,a,a,i0,a,a)') 'seabreeze (Fortran host modules): ', what, ' failed (', rc, '): ', trim(text)
    error stop 1
  end subroutine sb_fail

  subroutine sb_device_synchronize()
    integer(c_int) :: rc
    call sb_ensure_ctx()
    rc = c_sb_synchronize(sb_ctx)
    if (rc /= 0) call sb_fail('sb_synchronize', rc)
  end subroutine sb_device_synchronize

  !---------------------------------------------------------------------------
  ! latitude-band communicator
  !---------------------------------------------------------------------------
  subroutine sb_comm_get_unique_id(id)
    integer(c_signed_char), intent(out) :: id(128)
    integer(c_int) :: rc
    rc = c_comm_get_unique_id(id)
    if (rc /= 0) call sb_fail('sb_comm_get_unique_id', rc)
  end subroutine sb_comm_get_unique_id

  subroutine sb_comm_init(id, rank, nranks)
    integer(c_signed_char), intent(in) :: id(128)
    integer, intent(in) :: rank, nranks
    integer(c_int) :: rc
    call sb_ensure_ctx()
    rc = c_comm_init(sb_ctx, id, int(rank, c_int), int(nranks, c_int))
    if (rc /= 0) call sb_fail('sb_comm_init', rc)
  end subroutine sb_comm_init

  subroutine sb_comm_finalize()
    integer(c_int) :: rc
    if (.not. c_associated(sb_ctx)) return
    rc = c_comm_finalize(sb_ctx)
    if (rc /= 0) call sb_fail('sb_comm_finalize', rc)
  end subroutine sb_comm_finalize

  !> .true. once sb_comm_init has run on this process: swap_bounds then talks to the band neighbours.
  logical function sb_comm_active()
    integer(c_int) :: rc, r, n
    sb_comm_active = .false.
    if (.not. c_associated(sb_ctx)) return
    rc = c_comm_rank(sb_ctx, r, n)
    sb_comm_active = (rc == 0 .and. n > 0)
  end function sb_comm_active

  subroutine sb_comm_rank(rank, nranks)
    integer, intent(out) :: rank, nranks
    integer(c_int) :: rc, r, n
    call sb_ensure_ctx()
    rc = c_comm_rank(sb_ctx, r, n)
    if (rc /= 0) call sb_fail('sb_comm_rank', rc)
    rank = r; nranks = n
  end subroutine sb_comm_rank

  !> Opt-in: the caller states that sigma (an ancillary field) does not change between calls; its statistics are
  !! then formed (and, in a band run, gathered) in the first step only.  include/seabreeze_hip.h: sb_set_static_sigma.
  subroutine sb_set_static_sigma(on)
    logical, intent(in) :: on
    integer(c_int) :: rc
    call sb_ensure_ctx()
    rc = c_set_static_sigma(sb_ctx, merge(1_c_int, 0_c_int, on))
    if (rc /= 0) call sb_fail('sb_set_static_sigma', rc)
  end subroutine sb_set_static_sigma

  !> What the last diag call / band step enqueued: kernel launches, RCCL operations, RCCL groups, device copies.
  subroutine sb_last_step_report(launches, rccl_ops, rccl_groups, copies)
    integer, intent(out) :: launches, rccl_ops, rccl_groups, copies
    integer(c_int) :: rc, rep(4)
    call sb_ensure_ctx()
    rc = c_last_step_report(sb_ctx, rep)
    if (rc /= 0) call sb_fail('sb_last_step_report', rc)
    launches = rep(1); rccl_ops = rep(2); rccl_groups = rep(3); copies = rep(4)
  end subroutine sb_last_step_report

  !---------------------------------------------------------------------------
  ! device arrays
  !---------------------------------------------------------------------------
  function sb_dev_alloc(nbytes) result(dptr)
    integer(c_size_t), intent(in) :: nbytes
    type(c_ptr) :: dptr
    integer(c_int) :: rc
    call sb_ensure_ctx()
    rc = c_dev_malloc(sb_ctx, nbytes, dptr)
    if (rc /= 0) call sb_fail('sb_device_malloc', rc)
  end function sb_dev_alloc

  subroutine sb_dev_free(dptr)
    type(c_ptr), intent(inout) :: dptr
    integer(c_int) :: rc
    if (.not. c_associated(sb_ctx)) return
    rc = c_dev_free(sb_ctx, dptr)
    dptr = c_null_ptr
  end subroutine sb_dev_free

  !> host is c_loc of a contiguous host array
  subroutine sb_dev_upload(dptr, host, nbytes)
    type(c_ptr), intent(in) :: dptr, host
    integer(c_size_t), intent(in) :: nbytes
    integer(c_int) :: rc
    call sb_ensure_ctx()
    rc = c_dev_upload(sb_ctx, dptr, host, nbytes)
    if (rc /= 0) call sb_fail('sb_device_upload', rc)
  end subroutine sb_dev_upload

  subroutine sb_dev_download(host, dptr, nbytes)
    type(c_ptr), intent(in) :: dptr, host
    integer(c_size_t), intent(in) :: nbytes
    integer(c_int) :: rc
    call sb_ensure_ctx()
    rc = c_dev_download(sb_ctx, host, dptr, nbytes)
    if (rc /= 0) call sb_fail('sb_device_download', rc)
  end subroutine sb_dev_download

end module sb_context_mod
