!===============================================================================
! sb_f2py_state -- device context + C interfaces shared by the f2py-surface routines of
! seabreeze_f2py.f90.  Compiled and linked into the `seabreeze` extension but NOT scanned
! by f2py (it holds a type(c_ptr), which f2py cannot wrap).  f2py maps REAL to C float
! (the reference ships no .f2py_f2cmap, ref: python_wrapper/setup.py:12-16), so this
! surface binds the _f32 entry points of include/seabreeze_hip.h.
!===============================================================================
module sb_f2py_state
  use iso_c_binding
  implicit none
  type(c_ptr), save :: ctx = c_null_ptr
  ! status of the last routine of the surface: 0, or the library's sb_status with its message.  The reference
  ! kernels have no error channel here; the Python layer reads these after every call and raises.
  integer(c_int), save :: last_rc = 0
  character(len=512), save :: last_msg = ''

  interface
    integer(c_int) function sb_create(ctx, device) bind(C, name="sb_create")
      import :: c_ptr, c_int
      type(c_ptr), intent(out) :: ctx
      integer(c_int), value :: device
    end function
    type(c_ptr) function sb_last_error(ctx) bind(C, name="sb_last_error")
      import :: c_ptr
      type(c_ptr), value :: ctx
    end function
    integer(c_int) function sb_get_threads(nt) bind(C, name="sb_get_threads")
      import :: c_int
      integer(c_int), intent(out) :: nt
    end function
    integer(c_int) function sb_diag_f32(ctx, tn, p, z, std, theta, v, u, cdist, ws, wd, thc, target_plev, &
        thresh_wind, thresh_winddir, thresh_windch, thresh_thc, target_time, maxdist, timestep, nps, nlons, &
        nlats, output) bind(C, name="sb_diag_f32")
      import :: c_ptr, c_int, c_float
      type(c_ptr), value :: ctx
      integer(c_int), value :: tn, nps, nlons, nlats
      real(c_float), value :: target_plev, thresh_wind, thresh_winddir, thresh_windch, thresh_thc
      real(c_float), value :: target_time, maxdist, timestep
      real(c_float), intent(in) :: p(*), z(*), std(*), theta(*), v(*), u(*), cdist(*)
      real(c_float), intent(inout) :: ws(*), wd(*), thc(*), output(*)
    end function
    integer(c_int) function sb_diag_stream_begin_f32(ctx, nlons, nlats, z, std, cdist, ws, wd, thc) &
        bind(C, name="sb_diag_stream_begin_f32")
      import :: c_ptr, c_int, c_float
      type(c_ptr), value :: ctx
      integer(c_int), value :: nlons, nlats
      real(c_float), intent(in) :: z(*), std(*), cdist(*), ws(*), wd(*), thc(*)
    end function
    integer(c_int) function sb_diag_stream_step_f32(ctx, tn, p, nps, theta, v, u, target_plev, thresh_wind, &
        thresh_winddir, thresh_windch, thresh_thc, target_time, maxdist, timestep, sb_prev, have_prev) &
        bind(C, name="sb_diag_stream_step_f32")
      import :: c_ptr, c_int, c_float, c_double
      type(c_ptr), value :: ctx
      integer(c_int), value :: tn, nps
      real(c_float), value :: target_plev, thresh_wind, thresh_winddir, thresh_windch, thresh_thc
      real(c_float), value :: target_time, maxdist, timestep
      real(c_float), intent(in) :: p(*), theta(*), v(*), u(*)
      real(c_double), intent(inout) :: sb_prev(*)
      integer(c_int), intent(out) :: have_prev
    end function
    integer(c_int) function sb_diag_stream_end_f32(ctx, sb_last, output4, ws, wd, thc) &
        bind(C, name="sb_diag_stream_end_f32")
      import :: c_ptr, c_int, c_float, c_double
      type(c_ptr), value :: ctx
      real(c_double), intent(inout) :: sb_last(*)
      real(c_float), intent(out) :: output4(*), ws(*), wd(*), thc(*)
    end function
    integer(c_int) function sb_diag_stream_stats(ctx, steps, seconds) bind(C, name="sb_diag_stream_stats")
      import :: c_ptr, c_int, c_long, c_double
      type(c_ptr), value :: ctx
      integer(c_long), intent(out) :: steps
      real(c_double), intent(out) :: seconds(3)
    end function
    integer(c_int) function sb_sigmoid_f32(ctx, nx, ny, ary, sm) bind(C, name="sb_sigmoid_f32")
      import :: c_ptr, c_int, c_float
      type(c_ptr), value :: ctx
      integer(c_int), value :: nx, ny
      real(c_float), intent(in) :: ary(*)
      real(c_float), intent(out) :: sm(*)
    end function
    integer(c_int) function sb_get_edges_f32(ctx, nx, ny, lsm, ci, rule, bnd, coast) bind(C, name="sb_get_edges_f32")
      import :: c_ptr, c_int, c_float
      type(c_ptr), value :: ctx
      integer(c_int), value :: nx, ny, rule, bnd
      real(c_float), intent(in) :: lsm(*), ci(*)
      real(c_float), intent(out) :: coast(*)
    end function
    integer(c_int) function sb_get_dist_f32(ctx, nx, ny, coast, mask, lon, lat, maxdist, kwin, cdist) &
        bind(C, name="sb_get_dist_f32")
      import :: c_ptr, c_int, c_float
      type(c_ptr), value :: ctx
      integer(c_int), value :: nx, ny, kwin
      real(c_float), value :: maxdist
      real(c_float), intent(in) :: coast(*), mask(*), lon(*), lat(*)
      real(c_float), intent(out) :: cdist(*)
    end function
  end interface

contains

  subroutine sb_ensure()
    integer(c_int) :: rc
    if (.not. c_associated(ctx)) then
      rc = sb_create(ctx, -1_c_int)
      if (rc /= 0) then
        call sb_fail('sb_create', rc)
        ctx = c_null_ptr
      end if
    end if
  end subroutine sb_ensure

  ! The reference kernels have no error channel on this surface.  A failed device call leaves its status and the
  ! library's message here (and on standard error); the routine returns, and the Python layer -- which reads
  ! last_status() after every call -- raises.  The interpreter is never stopped.
  subroutine sb_fail(what, rc)
    character(len=*), intent(in) :: what
    integer(c_int), intent(in) :: rc
    character(kind=c_char), pointer :: msg(:)
    character(len=400) :: text
    type(c_ptr) :: cp
    integer :: i
    text = ''
    cp = sb_last_error(ctx)
    if (c_associated(cp)) then
      call c_f_pointer(cp, msg, [400])
      do i = 1, 400
        if (msg(i) == c_null_char) exit
        text(i:i) = msg(i)
      end do
    end if
    last_rc = rc
    last_msg = 'seabreeze.' // what // ': ' // trim(text)
    write (0, '(a,a,i0,a)') trim(last_msg), ' (status ', rc, ')'
  end subroutine sb_fail

  subroutine sb_ok()
    last_rc = 0
  end subroutine sb_ok

end module sb_f2py_state
