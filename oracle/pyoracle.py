"""ctypes loaders for the CPU oracle -- TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import
this module.  It exposes two things behind the same numpy-level API:

* ``Oracle(prec)``    -- oracle/liboracle_r{4,8}.so, our Fortran restatement
                         (oracle/sb_oracle.f90), present wherever `make -C oracle
                         oracle` ran (build() does that).
* ``Reference(prec)`` -- oracle/_ref/libsb_ref_r{4,8}.so, the reference's own
                         two wrapper files compiled unmodified (built only where
                         /root/reference exists; the .so travels to the GPU box).

Arrays: C-contiguous numpy with reversed shape == Fortran (lon, lat[, lev]);
see seabreeze_param_amd/synth.py.
"""
from __future__ import annotations

import ctypes as C
import os
import threading

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
_DT = {4: np.float32, 8: np.float64}
_CT = {4: C.c_float, 8: C.c_double}


def run_big_stack(fn, *args, stack_bytes: int = 2 << 30, **kw):
    """Run fn in a thread with a large stack: the reference keeps ~10 whole-grid
    automatic arrays on the stack."""
    box = {}

    def tgt():
        try:
            box["r"] = fn(*args, **kw)
        except BaseException as e:  # pragma: no cover
            box["e"] = e

    old = threading.stack_size(stack_bytes)
    try:
        th = threading.Thread(target=tgt)
        th.start()
        th.join()
    finally:
        threading.stack_size(old)
    if "e" in box:
        raise box["e"]
    return box["r"]


def _ptr(a):
    return a.ctypes.data_as(C.c_void_p)


def _f(a, dt):
    a = np.ascontiguousarray(a, dtype=dt)
    return a


class Oracle:
    """Our restatement (oracle/sb_oracle.f90)."""

    def __init__(self, prec: int = 8, omp: bool = False):
        name = f"liboracle_r{prec}{'_omp' if omp else ''}.so"
        path = os.path.join(HERE, name)
        if not os.path.exists(path):
            raise FileNotFoundError(f"{path} missing: run `make -C oracle oracle`")
        self.lib = C.CDLL(path)
        self.prec = prec
        self.dt = _DT[prec]
        self.ct = _CT[prec]
        nb = C.c_int(0)
        self.lib.sbo_real_bytes(C.byref(nb))
        assert nb.value == prec, (nb.value, prec)

    # -- sigmoid -------------------------------------------------------------
    def sigmoid(self, ary):
        ary = _f(ary, self.dt)
        ny, nx = ary.shape
        sm = np.empty_like(ary)
        self.lib.sbo_sigmoid(_ptr(ary), C.c_int(nx), C.c_int(ny), _ptr(sm))
        return sm

    # -- wrapper-flavour diag (state arrays updated IN PLACE) ------------------
    def diag(self, tn, p, z, std, theta, v, u, cdist, ws, wd, thc, output=None,
             target_plev=700.0, thresh_wind=11.0, thresh_winddir=90.0, thresh_windch=5.0,
             thresh_thc=0.75, target_time=6.0, maxdist=180.0, timestep=24.0):
        dt, ct = self.dt, self.ct
        p = _f(p, dt); z = _f(z, dt); std = _f(std, dt); theta = _f(theta, dt)
        v = _f(v, dt); u = _f(u, dt); cdist = _f(cdist, dt)
        for s in (ws, wd, thc):
            assert s.dtype == dt and s.flags.c_contiguous
        nps = p.shape[0]
        ny, nx = z.shape
        assert u.shape == (nps, ny, nx) and v.shape == (nps, ny, nx)
        if output is None:
            output = np.zeros((4, ny, nx), dtype=dt)
        nn = C.c_int(0)
        run_big_stack(
            self.lib.sbo_diag, C.c_int(tn), _ptr(p), _ptr(z), _ptr(std), _ptr(theta), _ptr(v), _ptr(u),
            _ptr(cdist), _ptr(ws), _ptr(wd), _ptr(thc),
            ct(target_plev), ct(thresh_wind), ct(thresh_winddir), ct(thresh_windch),
            ct(thresh_thc), ct(target_time), ct(maxdist), ct(timestep),
            C.c_int(nps), C.c_int(nx), C.c_int(ny), _ptr(output), C.byref(nn))
        self.last_nn_max = nn.value
        return output

    # -- generic-flavour seabreeze_diag (state + sb_con updated IN PLACE) ------
    def sigmoid_scalars(self, ary):
        """(std, r) of the logistic for a field (generic/sea_breeze_diag.f90:466-479)."""
        ary = _f(ary, self.dt)
        ny, nx = ary.shape
        sd, r = self.ct(0), self.ct(0)
        self.lib.sbo_sigmoid_scalars(_ptr(ary), C.c_int(nx), C.c_int(ny), C.byref(sd), C.byref(r))
        return sd.value, r.value

    def seabreeze_diag(self, timestep, tn, p, u, v, theta, mask, z, sigma, ws, wd, thc, sb_con,
                       halo=0, bnd=1, omp=False, ext_stats=None, level_rule=0):
        dt, ct = self.dt, self.ct
        p = _f(p, dt); u = _f(u, dt); v = _f(v, dt)
        theta = _f(theta, dt); mask = _f(mask, dt); z = _f(z, dt); sigma = _f(sigma, dt)
        for s in (ws, wd, thc, sb_con):
            assert s.dtype == dt and s.flags.c_contiguous
        nz, ny, nx = p.shape
        assert theta.shape == (ny + 2 * halo, nx + 2 * halo), theta.shape
        nn = C.c_int(0)
        if omp:
            assert halo == 0 and bnd == 1
            self.lib.sbo_seabreeze_diag_omp(
                ct(timestep), C.c_int(tn), _ptr(p), _ptr(u), _ptr(v), _ptr(theta), _ptr(mask),
                _ptr(z), _ptr(sigma), _ptr(ws), _ptr(wd), _ptr(thc), _ptr(sb_con),
                C.c_int(nx), C.c_int(ny), C.c_int(nz))
            return sb_con
        use_ext = 0 if ext_stats is None else 1
        es, er = (0.0, 0.0) if ext_stats is None else ext_stats
        self.lib.sbo_seabreeze_diag_x(
            ct(timestep), C.c_int(tn), _ptr(p), _ptr(u), _ptr(v), _ptr(theta), _ptr(mask),
            _ptr(z), _ptr(sigma), _ptr(ws), _ptr(wd), _ptr(thc), _ptr(sb_con),
            C.c_int(nx), C.c_int(ny), C.c_int(nz), C.c_int(halo), C.c_int(bnd),
            C.c_int(use_ext), ct(es), ct(er), C.c_int(level_rule), C.byref(nn))
        self.last_nn_max = nn.value
        return sb_con

    # -- coast detection / distance -------------------------------------------
    def get_edges(self, lsm, ci, rule=0, bnd=0):
        lsm = _f(lsm, self.dt); ci = _f(ci, self.dt)
        ny, nx = lsm.shape
        coast = np.empty_like(lsm)
        self.lib.sbo_get_edges(_ptr(lsm), _ptr(ci), C.c_int(nx), C.c_int(ny),
                               C.c_int(rule), C.c_int(bnd), _ptr(coast))
        return coast

    def dist_window(self, lon, lat, maxdist=180.0):
        lon = _f(lon, self.dt); lat = _f(lat, self.dt)
        k = C.c_int(0)
        self.lib.sbo_dist_window(_ptr(lon), _ptr(lat), C.c_int(lon.size), C.c_int(lat.size),
                                 self.ct(maxdist), C.byref(k))
        return k.value

    def get_dist(self, coast, mask, lon, lat, maxdist=180.0, kwin=-1):
        dt = self.dt
        coast = _f(coast, dt); mask = _f(mask, dt); lon = _f(lon, dt); lat = _f(lat, dt)
        ny, nx = coast.shape
        cdist = np.empty_like(coast)
        self.lib.sbo_get_dist(_ptr(coast), _ptr(mask), _ptr(lon), _ptr(lat), C.c_int(nx), C.c_int(ny),
                              self.ct(maxdist), C.c_int(kwin), _ptr(cdist))
        return cdist


class Reference:
    """The reference's own wrapper Fortran, compiled unmodified (oracle/_ref)."""

    def __init__(self, prec: int = 8):
        path = os.path.join(HERE, "_ref", f"libsb_ref_r{prec}.so")
        if not os.path.exists(path):
            raise FileNotFoundError(f"{path} missing: run `make -C oracle ref` where /root/reference exists")
        self.lib = C.CDLL(path)
        self.prec = prec
        self.dt = _DT[prec]
        self.ct = _CT[prec]

    def sigmoid(self, ary):
        ary = _f(ary, self.dt)
        ny, nx = ary.shape
        sm = np.empty_like(ary)
        run_big_stack(self.lib.sigmoid_, _ptr(ary), C.byref(C.c_int(nx)), C.byref(C.c_int(ny)), _ptr(sm))
        return sm

    def diag(self, tn, p, z, std, theta, v, u, cdist, ws, wd, thc, output=None,
             target_plev=700.0, thresh_wind=11.0, thresh_winddir=90.0, thresh_windch=5.0,
             thresh_thc=0.75, target_time=6.0, maxdist=180.0, timestep=24.0):
        dt, ct = self.dt, self.ct
        p = _f(p, dt); z = _f(z, dt); std = _f(std, dt); theta = _f(theta, dt)
        v = _f(v, dt); u = _f(u, dt); cdist = _f(cdist, dt)
        for s in (ws, wd, thc):
            assert s.dtype == dt and s.flags.c_contiguous
        nps = p.shape[0]
        ny, nx = z.shape
        if output is None:
            output = np.zeros((4, ny, nx), dtype=dt)
        # diag_ overwrites the three unit scalars in place: fresh copies every call
        sc = [ct(x) for x in (target_plev, thresh_wind, thresh_winddir, thresh_windch,
                              thresh_thc, target_time, maxdist, timestep)]
        run_big_stack(
            self.lib.diag_, C.byref(C.c_int(tn)), _ptr(p), _ptr(z), _ptr(std), _ptr(theta), _ptr(v), _ptr(u),
            _ptr(cdist), _ptr(ws), _ptr(wd), _ptr(thc), *[C.byref(s) for s in sc],
            C.byref(C.c_int(nps)), C.byref(C.c_int(nx)), C.byref(C.c_int(ny)), _ptr(output))
        return output

    def get_edges(self, lsm, ci):
        lsm = _f(lsm, self.dt); ci = _f(ci, self.dt)
        ny, nx = lsm.shape
        coast = np.empty_like(lsm)
        run_big_stack(self.lib.get_edges_, _ptr(lsm), _ptr(ci), C.byref(C.c_int(nx)), C.byref(C.c_int(ny)),
                      _ptr(coast))
        return coast

    def get_dist(self, coast, mask, lon, lat, maxdist=180.0):
        dt = self.dt
        coast = _f(coast, dt); mask = _f(mask, dt); lon = _f(lon, dt); lat = _f(lat, dt)
        ny, nx = coast.shape
        cdist = np.empty_like(coast)
        run_big_stack(self.lib.get_dist_, _ptr(coast), _ptr(mask), _ptr(lon), _ptr(lat),
                      C.byref(C.c_int(nx)), C.byref(C.c_int(ny)), C.byref(self.ct(maxdist)), _ptr(cdist))
        return cdist


def reference_available(prec: int = 8) -> bool:
    return os.path.exists(os.path.join(HERE, "_ref", f"libsb_ref_r{prec}.so"))
