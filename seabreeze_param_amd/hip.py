"""ctypes binding of libseabreeze_hip.so (include/seabreeze_hip.h).

This is plumbing only: every numerical result comes from the HIP kernels behind
the C ABI.  There is no CPU fallback -- a missing library or a missing gfx950
device raises.

Arrays: C-contiguous numpy with reversed shape == Fortran (lon, lat[, lev]);
see seabreeze_param_amd/synth.py.
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("SEABREEZE_HIP_LIB", os.path.join(_HERE, "libseabreeze_hip.so"))

SB_BND_WRAPPER, SB_BND_GLOBAL, SB_BND_HALO = 0, 1, 2
SB_UM_THETA_TO_T0, SB_UM_LEVEL_WALK = 1, 2

_SFX = {np.dtype(np.float32): "f32", np.dtype(np.float64): "f64"}
_CT = {np.dtype(np.float32): C.c_float, np.dtype(np.float64): C.c_double}


class SeabreezeHipError(RuntimeError):
    pass


class Tunables(C.Structure):
    _fields_ = [(n, C.c_double) for n in (
        "target_plev_pa", "thresh_wind", "thresh_winddir", "thresh_windch",
        "thresh_thc", "target_time_s", "maxdist_km")]


_lib = None


def load_library() -> C.CDLL:
    """Load the product library; fail loudly when it has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise SeabreezeHipError(
                f"{LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                "(there is no CPU fallback)")
        _lib = C.CDLL(LIB_PATH)
        _lib.sb_last_error.restype = C.c_char_p
        _lib.sb_last_error.argtypes = [C.c_void_p]
        _lib.sb_version.restype = C.c_char_p
    return _lib


def hip_runtimes():
    """Paths of the HIP runtimes (libamdhip64) mapped into this process.  PyTorch bundles a runtime of its own and
    loads it by path; libseabreeze_hip.so asks for the soname.  So with `import torch` FIRST the library binds to
    torch's copy and the process has ONE runtime -- a torch stream handle is then a valid hipStream_t for the C ABI --
    while with the library loaded first there are TWO, whose streams and synchronisation know nothing of each other (and
    of which, on this pool, only the first to initialise finds the GPU)."""
    seen = []
    try:
        for ln in open("/proc/self/maps"):
            i = ln.find("/")
            path = ln[i:].strip() if i >= 0 else ""
            if "libamdhip64.so" in path and path not in seen:
                seen.append(path)
    except OSError:
        pass
    return seen


def torch_stream_handle(torch):
    """The hipStream_t to hand to the `_dev` entry points from a process that also runs PyTorch: torch's current stream.
    Valid only when torch and the library share ONE HIP runtime, i.e. when torch was imported before the library was
    loaded; with two runtimes in a process only the one that initialises first finds the GPU at all (measured on this
    pool: the other reports no device), so that order is refused here rather than timed wrongly."""
    rts = hip_runtimes()
    if len(rts) != 1:
        raise SeabreezeHipError("torch and libseabreeze_hip.so must share one HIP runtime: import torch BEFORE the library is "
                                f"loaded (mapped runtimes: {rts})")
    return torch.cuda.current_stream().cuda_stream


def _p(a):
    if a is None:
        return None
    if isinstance(a, np.ndarray):
        return C.c_void_p(a.ctypes.data)
    return C.c_void_p(int(a))       # raw device address


def _host(a, dt):
    a = np.ascontiguousarray(a, dtype=dt)
    return a


class Context:
    """One HIP device + stream + workspace (sb_ctx)."""

    def __init__(self, device: int = -1):
        self.lib = load_library()
        h = C.c_void_p()
        rc = self.lib.sb_create(C.byref(h), C.c_int(device))
        if rc != 0:
            raise SeabreezeHipError(f"sb_create failed ({rc}): {self.lib.sb_last_error(None).decode()}")
        self.h = h

    def close(self):
        if getattr(self, "h", None):
            self.lib.sb_destroy(self.h)
            self.h = None

    def __del__(self):  # pragma: no cover
        try:
            self.close()
        except Exception:
            pass

    def _chk(self, rc, what):
        if rc != 0:
            raise SeabreezeHipError(f"{what} failed ({rc}): {self.lib.sb_last_error(self.h).decode()}")

    def synchronize(self):
        self._chk(self.lib.sb_synchronize(self.h), "sb_synchronize")

    def set_search_radius_hint(self, r: int):
        self._chk(self.lib.sb_set_search_radius_hint(self.h, C.c_int(r)), "sb_set_search_radius_hint")

    def set_fold(self, on: bool):
        """k_prep's work inside the contrast kernel (default) or as a kernel of its own (measurement / test knob)."""
        self._chk(self.lib.sb_set_fold(self.h, C.c_int(1 if on else 0)), "sb_set_fold")

    def set_plan_cache(self, on: bool):
        """The strip kernel keeps its plan while the band plane stands (default) or plans every call (measurement / test knob)."""
        self._chk(self.lib.sb_set_plan_cache(self.h, C.c_int(1 if on else 0)), "sb_set_plan_cache")

    def set_wide_strip(self, on: bool):
        """Radii beyond 16 in single precision: the 96-column strip kernel (default) or the tile kernel (test knob)."""
        self._chk(self.lib.sb_set_wide_strip(self.h, C.c_int(1 if on else 0)), "sb_set_wide_strip")

    def set_workgroups(self, n: int):
        """Persistent workgroups of the one-per-CU kernels (0: the device's compute units); a test knob."""
        self._chk(self.lib.sb_set_workgroups(self.h, C.c_int(int(n))), "sb_set_workgroups")

    def set_band_order(self, contrast_first: bool):
        """A band step runs k_scan, k_wind | join | contrast (default) or k_scan | join | contrast, k_wind (measurement knob)."""
        self._chk(self.lib.sb_set_band_order(self.h, C.c_int(1 if contrast_first else 0)), "sb_set_band_order")

    def set_static_sigma(self, on: bool):
        """Opt-in: sigma does not change between calls; its statistics are formed once (include/seabreeze_hip.h)."""
        self._chk(self.lib.sb_set_static_sigma(self.h, C.c_int(1 if on else 0)), "sb_set_static_sigma")

    def last_step_report(self):
        """What the last diag call / band step enqueued on this rank."""
        arr = (C.c_int * 4)()
        self._chk(self.lib.sb_last_step_report(self.h, arr), "sb_last_step_report")
        return dict(kernel_launches=arr[0], rccl_ops=arr[1], rccl_groups=arr[2], d2d_copies=arr[3])

    def last_counters(self):
        arr = (C.c_longlong * 4)()
        self._chk(self.lib.sb_last_counters(self.h, arr), "sb_last_counters")
        return dict(band_cells=arr[0], global_path_cells=arr[1], one_class_cells=arr[2], max_radius=arr[3])

    def profile_begin(self, max_calls: int):
        self._chk(self.lib.sb_profile_begin(self.h, C.c_int(max_calls)), "sb_profile_begin")

    def profile_end(self):
        """-> ({'k_scan', 'k_wind', 'k_t0', 'k_thc', 'k_prep'} -> ms, ncalls): HIP-event averages of the
        launches of one diag call.  k_thc is k_thc3 (t0, tables, search); k_t0 exists for the f2py flavour
        only (0.0 where a call does not launch it); k_prep is the list / statistics kernel after k_scan."""
        ms = (C.c_double * 5)()
        n = C.c_int(0)
        self._chk(self.lib.sb_profile_end(self.h, ms, C.byref(n)), "sb_profile_end")
        return dict(k_scan=ms[0], k_wind=ms[1], k_t0=ms[2], k_thc=ms[3], k_prep=ms[4]), n.value

    # ------------------------------------------------------------------ host-pointer API
    def seabreeze_diag(self, timestep, tn, p, u, v, theta, mask, z, sigma, ws, wd, thc, sb_con,
                       halo=0, bnd=SB_BND_GLOBAL, tunables: Tunables | None = None):
        """generic/sea_breeze_diag.f90:55 semantics; ws, wd, thc, sb_con updated in place."""
        dt = np.dtype(ws.dtype)
        sfx, ct = _SFX[dt], _CT[dt]
        p = _host(p, dt); u = _host(u, dt); v = _host(v, dt)
        theta = _host(theta, dt); mask = _host(mask, dt); z = _host(z, dt); sigma = _host(sigma, dt)
        nz, ny, nx = p.shape
        for a in (theta, mask, z, sigma):
            if a.shape != (ny + 2 * halo, nx + 2 * halo):
                raise ValueError(f"2-D input shape {a.shape} != {(ny + 2 * halo, nx + 2 * halo)}")
        for a in (ws, wd, thc, sb_con):
            if a.shape != (ny, nx) or a.dtype != dt or not a.flags.c_contiguous:
                raise ValueError("state arrays must be C-contiguous (ny, nx) of the working dtype")
        fn = getattr(self.lib, f"sb_seabreeze_diag_{sfx}")
        rc = fn(self.h, ct(timestep), C.c_int(tn), C.c_int(nx), C.c_int(ny), C.c_int(nz), C.c_int(halo),
                C.c_int(bnd), _p(p), _p(u), _p(v), _p(theta), _p(mask), _p(z), _p(sigma),
                _p(ws), _p(wd), _p(thc), _p(sb_con), C.byref(tunables) if tunables is not None else None)
        self._chk(rc, "sb_seabreeze_diag")
        return sb_con

    def seabreeze_diag_um(self, timestep, tn, p, u, v, theta, z, sigma, mask, ws, wd, thc, sb_con,
                          halo_s, halo_l, flags=0):
        """UM vn10.7 layout and argument order (UM/vn10.7/sea_breeze_diag.F90:55-117): theta, z, sigma carry halo_s
        ghost cells, mask halo_l >= halo_s; theta is overwritten with t0 under SB_UM_THETA_TO_T0.  Returns `error`."""
        dt = np.dtype(ws.dtype)
        sfx, ct = _SFX[dt], _CT[dt]
        p = _host(p, dt); u = _host(u, dt); v = _host(v, dt)
        z = _host(z, dt); sigma = _host(sigma, dt); mask = _host(mask, dt)
        if theta.dtype != dt or not theta.flags.c_contiguous:
            raise ValueError("theta must be a C-contiguous array of the working dtype (it may be overwritten)")
        nz, ny, nx = p.shape if p.ndim == 3 else (0, 0, 0)
        err = C.c_int(0)
        fn = getattr(self.lib, f"sb_seabreeze_diag_um_{sfx}")
        rc = fn(self.h, ct(timestep), C.c_int(tn), C.c_int(nx), C.c_int(ny), C.c_int(nz), C.c_int(halo_s), C.c_int(halo_l),
                _p(p), _p(u), _p(v), _p(theta), _p(z), _p(sigma), _p(mask), _p(ws), _p(wd), _p(thc), _p(sb_con),
                C.c_int(flags), C.byref(err))
        self._chk(rc, "sb_seabreeze_diag_um")
        return err.value

    def diag(self, tn, p, z, std, theta, v, u, cdist, ws, wd, thc, output=None,
             target_plev=700.0, thresh_wind=11.0, thresh_winddir=90.0, thresh_windch=5.0,
             thresh_thc=0.75, target_time=6.0, maxdist=180.0, timestep=24.0):
        """seabreeze_diag_python.f90:49 semantics; ws, wd, thc updated in place; returns output(4, ny, nx)."""
        dt = np.dtype(ws.dtype)
        sfx, ct = _SFX[dt], _CT[dt]
        p = _host(p, dt); z = _host(z, dt); std = _host(std, dt); theta = _host(theta, dt)
        v = _host(v, dt); u = _host(u, dt); cdist = _host(cdist, dt)
        nps = p.shape[0]
        ny, nx = z.shape
        if u.shape != (nps, ny, nx) or v.shape != (nps, ny, nx):
            raise ValueError("u, v must be (nps, ny, nx)")
        if output is None:
            output = np.zeros((4, ny, nx), dtype=dt)
        fn = getattr(self.lib, f"sb_diag_{sfx}")
        rc = fn(self.h, C.c_int(tn), _p(p), _p(z), _p(std), _p(theta), _p(v), _p(u), _p(cdist),
                _p(ws), _p(wd), _p(thc), ct(target_plev), ct(thresh_wind), ct(thresh_winddir),
                ct(thresh_windch), ct(thresh_thc), ct(target_time), ct(maxdist), ct(timestep),
                C.c_int(nps), C.c_int(nx), C.c_int(ny), _p(output))
        self._chk(rc, "sb_diag")
        return output

    def sigmoid(self, ary):
        ary = np.ascontiguousarray(ary)
        dt = np.dtype(ary.dtype)
        ny, nx = ary.shape
        sm = np.empty_like(ary)
        rc = getattr(self.lib, f"sb_sigmoid_{_SFX[dt]}")(self.h, C.c_int(nx), C.c_int(ny), _p(ary), _p(sm))
        self._chk(rc, "sb_sigmoid")
        return sm

    def get_edges(self, lsm, ci, rule=0, bnd=SB_BND_WRAPPER):
        lsm = np.ascontiguousarray(lsm)
        dt = np.dtype(lsm.dtype)
        ci = _host(ci, dt)
        ny, nx = lsm.shape
        coast = np.empty_like(lsm)
        rc = getattr(self.lib, f"sb_get_edges_{_SFX[dt]}")(self.h, C.c_int(nx), C.c_int(ny), _p(lsm), _p(ci),
                                                          C.c_int(rule), C.c_int(bnd), _p(coast))
        self._chk(rc, "sb_get_edges")
        return coast

    def get_dist(self, coast, mask, lon, lat, maxdist=180.0, kwin=-1):
        coast = np.ascontiguousarray(coast)
        dt = np.dtype(coast.dtype)
        mask = _host(mask, dt); lon = _host(lon, dt); lat = _host(lat, dt)
        ny, nx = coast.shape
        cdist = np.empty_like(coast)
        rc = getattr(self.lib, f"sb_get_dist_{_SFX[dt]}")(self.h, C.c_int(nx), C.c_int(ny), _p(coast), _p(mask),
                                                         _p(lon), _p(lat), _CT[dt](maxdist), C.c_int(kwin),
                                                         _p(cdist))
        self._chk(rc, "sb_get_dist")
        return cdist

    # ------------------------------------------------------------------ device-pointer API
    def sigmoid_dev(self, dtype, nx, ny, ary, sm, stream=None):
        """sm = 1/(1+exp(-std*(ary-r))) on device arrays (raw addresses); enqueues without synchronising."""
        fn = getattr(self.lib, f"sb_sigmoid_{_SFX[np.dtype(dtype)]}_dev")
        self._chk(fn(self.h, C.c_int(nx), C.c_int(ny), _p(ary), _p(sm), C.c_void_p(stream) if stream else None), "sb_sigmoid_dev")

    def seabreeze_diag_dev(self, dtype, timestep, tn, nx, ny, nz, halo, bnd, p, u, v, theta, mask, z, sigma,
                           ws, wd, thc, sb_con, stream=None, tunables: Tunables | None = None):
        """All array arguments are raw device addresses (ints); enqueues without synchronising."""
        dt = np.dtype(dtype)
        fn = getattr(self.lib, f"sb_seabreeze_diag_{_SFX[dt]}_dev")
        rc = fn(self.h, _CT[dt](timestep), C.c_int(tn), C.c_int(nx), C.c_int(ny), C.c_int(nz), C.c_int(halo),
                C.c_int(bnd), _p(p), _p(u), _p(v), _p(theta), _p(mask), _p(z), _p(sigma), _p(ws), _p(wd),
                _p(thc), _p(sb_con), C.byref(tunables) if tunables is not None else None,
                C.c_void_p(stream) if stream else None)
        self._chk(rc, "sb_seabreeze_diag_dev")


    def get_edges_dev(self, dtype, nx, ny, lsm, ci, coast, rule=0, bnd=SB_BND_WRAPPER, stream=None):
        """get_edges on device arrays (raw addresses); enqueues without synchronising."""
        fn = getattr(self.lib, f"sb_get_edges_{_SFX[np.dtype(dtype)]}_dev")
        self._chk(fn(self.h, C.c_int(nx), C.c_int(ny), _p(lsm), _p(ci), C.c_int(rule), C.c_int(bnd), _p(coast),
                     C.c_void_p(stream) if stream else None), "sb_get_edges_dev")

    def get_dist_dev(self, dtype, nx, ny, coast, mask, lon, lat, cdist, maxdist=180.0, kwin=-1, stream=None):
        """get_dist on device arrays (raw addresses); lon, lat are host vectors."""
        dt = np.dtype(dtype)
        lon = _host(lon, dt); lat = _host(lat, dt)
        fn = getattr(self.lib, f"sb_get_dist_{_SFX[dt]}_dev")
        self._chk(fn(self.h, C.c_int(nx), C.c_int(ny), _p(coast), _p(mask), _p(lon), _p(lat), _CT[dt](maxdist),
                     C.c_int(kwin), _p(cdist), C.c_void_p(stream) if stream else None), "sb_get_dist_dev")

    def sigma_moments_dev(self, dtype, nx, ny, halo, sigma, moments5, stream=None):
        """Band-local sigma moments -> 5 doubles at device address moments5."""
        fn = getattr(self.lib, f"sb_sigma_moments_{_SFX[np.dtype(dtype)]}_dev")
        self._chk(fn(self.h, C.c_int(nx), C.c_int(ny), C.c_int(halo), _p(sigma), _p(moments5),
                     C.c_void_p(stream) if stream else None), "sb_sigma_moments_dev")

    def use_gathered_moments(self, gathered, nparts: int):
        """Following diag calls merge `nparts` gathered moments (device address) instead of scanning sigma."""
        self._chk(self.lib.sb_use_gathered_moments(self.h, _p(gathered) if gathered else None, C.c_int(nparts)),
                  "sb_use_gathered_moments")


    # ------------------------------------------------------------------ latitude-band communication (RCCL)
    def comm_init(self, unique_id: bytes, rank: int, nranks: int):
        assert len(unique_id) == 128
        buf = (C.c_ubyte * 128).from_buffer_copy(unique_id)
        self._chk(self.lib.sb_comm_init(self.h, buf, C.c_int(rank), C.c_int(nranks)), "sb_comm_init")

    def comm_finalize(self):
        self._chk(self.lib.sb_comm_finalize(self.h), "sb_comm_finalize")

    def swap_bounds_dev(self, dtype, field, nx, ny, halo, stream=None):
        """Ghost-cell fill of a device field (nx+2h, ny+2h): N-S rows over RCCL, poles and E-W locally."""
        fn = getattr(self.lib, f"sb_swap_bounds_{_SFX[np.dtype(dtype)]}_dev")
        self._chk(fn(self.h, _p(field), C.c_int(nx), C.c_int(ny), C.c_int(halo),
                     C.c_void_p(stream) if stream else None), "sb_swap_bounds_dev")

    def fill_ghosts_dev(self, dtype, field, nx, ny, halo, south, north, stream=None):
        """The local part of swap_bounds: E-W wrap of every row, pole replication where south / north is set."""
        fn = getattr(self.lib, f"sb_fill_ghosts_{_SFX[np.dtype(dtype)]}_dev")
        self._chk(fn(self.h, _p(field), C.c_int(nx), C.c_int(ny), C.c_int(halo), C.c_int(1 if south else 0),
                     C.c_int(1 if north else 0), C.c_void_p(stream) if stream else None), "sb_fill_ghosts_dev")

    def band_seabreeze_diag_dev(self, dtype, timestep, tn, nx, ny, nz, halo, p, u, v, theta, mask, z, sigma,
                                ws, wd, thc, sb_con, stream=None, tunables: Tunables | None = None):
        """One step of a latitude band incl. its communication (moments all-gather, theta ghost rows)."""
        dt = np.dtype(dtype)
        fn = getattr(self.lib, f"sb_band_seabreeze_diag_{_SFX[dt]}_dev")
        rc = fn(self.h, _CT[dt](timestep), C.c_int(tn), C.c_int(nx), C.c_int(ny), C.c_int(nz), C.c_int(halo),
                _p(p), _p(u), _p(v), _p(theta), _p(mask), _p(z), _p(sigma), _p(ws), _p(wd), _p(thc), _p(sb_con),
                C.byref(tunables) if tunables is not None else None, C.c_void_p(stream) if stream else None)
        self._chk(rc, "sb_band_seabreeze_diag_dev")

    def allgather_moments_dev(self, mine5, gathered, stream=None):
        self._chk(self.lib.sb_allgather_moments_dev(self.h, _p(mine5), _p(gathered),
                                                    C.c_void_p(stream) if stream else None),
                  "sb_allgather_moments_dev")


def comm_unique_id() -> bytes:
    """128-byte RCCL unique id (call on rank 0, broadcast to every rank, pass to Context.comm_init)."""
    lib = load_library()
    buf = (C.c_ubyte * 128)()
    rc = lib.sb_comm_get_unique_id(buf)
    if rc != 0:
        raise SeabreezeHipError(f"sb_comm_get_unique_id failed ({rc}): {lib.sb_last_error(None).decode()}")
    return bytes(buf)


def dist_window(lon, lat, maxdist=180.0) -> int:
    lib = load_library()
    lon = np.ascontiguousarray(lon)
    dt = np.dtype(lon.dtype)
    lat = _host(lat, dt)
    k = C.c_int(0)
    rc = getattr(lib, f"sb_dist_window_{_SFX[dt]}")(C.c_int(lon.size), C.c_int(lat.size), _p(lon), _p(lat),
                                                   _CT[dt](maxdist), C.byref(k))
    if rc != 0:
        raise SeabreezeHipError(f"sb_dist_window failed ({rc}): {lib.sb_last_error(None).decode()}")
    return k.value


def get_threads() -> int:
    lib = load_library()
    n = C.c_int(0)
    lib.sb_get_threads(C.byref(n))
    return n.value
