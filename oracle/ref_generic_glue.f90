! oracle/ref_generic_glue.f90 -- TEST INFRASTRUCTURE, not the product.
!
! A C-callable door to the reference's own host-model flavour, generic/sea_breeze_diag.f90 (module
! sea_breeze_diag_mod, assumed-shape dummies), compiled UNMODIFIED from where it lies under /root/reference
! by `make -C oracle ref_generic` together with generic/halo_exchange_mod.f90 and generic/get_all_fields_mod.f90.
! Only this file is ours: explicit-shape arguments in, a plain call of the reference's subroutine, nothing else.
! Used by tests/golden/make_golden_generic.py and tests/test_oracle_pin.py to pin oracle/sb_oracle.f90's
! host-model flavour (sbo_seabreeze_diag_x) against it.   ref: generic/sea_breeze_diag.f90:52-53 (interface)
subroutine sbref_generic_diag(nx, ny, nz, timestep, tn, p, u, v, theta, mask, z, sigma, ws, wd, thc, sb) &
    bind(C, name="sbref_generic_diag")
  use sea_breeze_diag_mod, only: seabreeze_diag
  implicit none
  integer, value :: nx, ny, nz, tn
  real, value :: timestep
  real, intent(in) :: p(nx, ny, nz), u(nx, ny, nz), v(nx, ny, nz)
  real, intent(in) :: theta(nx, ny), mask(nx, ny), z(nx, ny), sigma(nx, ny)
  real, intent(inout) :: ws(nx, ny), wd(nx, ny), thc(nx, ny), sb(nx, ny)
  call seabreeze_diag(timestep, tn, p, u, v, theta, mask, z, sigma, ws, wd, thc, sb)
end subroutine sbref_generic_diag
