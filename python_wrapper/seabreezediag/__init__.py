"""Python driver layer over the `seabreeze` extension module (MI355X build).

Counterpart of the reference's ``seabreezediag`` package
(ref: python_wrapper/seabreezediag/__init__.py:13-40 ``c2f``, :91-263 ``diag``): same call
signature, same array conventions, same return tuple, so scripts written against the
reference keep working with ``import seabreezediag``.  The `seabreeze` module it imports
is the f2py surface built from python_wrapper/seabreeze_f2py.f90, whose routines run on
the GPU through libseabreeze_hip.so.

Conventions kept from the reference
  * 2-D fields are C-ordered ``(lat, lon)``, winds ``([time,] pres, lat, lon)``; ``c2f``
    flips them into the Fortran ``(lon, lat[, pres])`` layout the kernels use.
  * ``tt`` is the running timestep number (clamped to >= 1) and comes back advanced by the
    number of steps processed.
  * The second returned state array is the sea-level temperature t0 (output plane 2), which
    the reference hands back under the name ``thc`` (ref :244); the
    kernel never reads that state, so this is harmless and kept.
  * The last latitude row of every output plane is never written by the kernel
    (ref: seabreeze_diag_python.f90:165).

Differences (both are crashes in the reference)
  * ``ci=None`` works: no sea ice, the coast distance is computed once.
  * inputs without a time axis work.
"""
import warnings

import numpy as np

from seabreeze import diag as _diag_kernel, get_dist, get_edges

__all__ = ["diag", "c2f", "read_nc"]


def c2f(array):
    """Reinterpret a C-ordered array as the Fortran-ordered array with reversed shape.  Element for
    element that is the transpose whatever the memory layout of the input (ravel in C order, refill in
    Fortran order), so the transpose *view* is returned: no copy for C- or Fortran-contiguous input."""
    return np.asarray(array).T


def _pop(kwargs, key, default=None):
    return kwargs.pop(key) if key in kwargs else default


_dist_cache = {"key": None, "dist": None}


def _coast_distance(lsm, ice, lon, lat):
    """Signed coast distance for this land mask and sea-ice field.  The reference recomputes it at
    every timestep (ref :223-228); the result depends only on these four inputs, so the last one is
    kept and reused while they compare equal to the copies kept with it --
    sea ice in reanalysis files changes daily, not with every model step.  An exact comparison
    (a memcmp-speed pass) rather than a digest: cheaper, and no collision to argue about."""
    key = _dist_cache["key"]
    same = key is not None and all(
        a.shape == b.shape and a.dtype == b.dtype and np.array_equal(a, b)
        for a, b in zip((np.asarray(lsm), np.asarray(ice), np.asarray(lon), np.asarray(lat)), key))
    if not same:
        _dist_cache["dist"] = c2f(get_dist(get_edges(c2f(lsm), c2f(ice)), c2f(lsm), lon, lat))
        _dist_cache["key"] = tuple(np.array(a, copy=True) for a in (lsm, ice, lon, lat))
    return _dist_cache["dist"]


def diag(tt, lsm, z, std, lon, lat, pres, *args, **kwargs):
    """Potential sea-breeze convergence strength in the coastal band
    (Bergemann et al. 2017, doi:10.1002/2017MS001048).

    Positional: ``tt, lsm, z, std, lon, lat, pres, u, v, t, ci`` -- or pass ``meta=`` (an
    object with attributes ``u, v, theta`` and optionally ``ci``) instead of the last four.
    Keywords: state ``ws, wd, thc`` from the previous call (default zeros), plus the kernel
    tunables ``target_plev`` [hPa], ``thresh_wind``, ``thresh_winddir``, ``thresh_windch``,
    ``thresh_thc``, ``target_time`` [h], ``maxdist`` [km], ``timestep`` [min].

    Returns ``(tt, sb_con, thc, ws, wd)``; ``sb_con`` is float64 ``(ntime, lat, lon)``.
    """
    ws, wd, thc = (_pop(kwargs, k) for k in ("ws", "wd", "thc"))
    meta = _pop(kwargs, "meta")
    if meta is None:
        u, v, t, ci = args
    else:
        u, v, t = meta.u, meta.v, meta.theta
        ci = getattr(meta, "ci", None)
    tt = max(1, tt)
    names = {"ws": "Windspeed", "wd": "Wind direction", "thc": "Heating contrast"}
    state = {}
    for key, val in (("ws", ws), ("wd", wd), ("thc", thc)):
        if val is None:
            if tt > 1:
                warnings.warn(f"{names[key]} should be given from previous timestep")
            val = np.zeros_like(lsm)
        state[key] = val
    # The kernels update windspeed / winddir / thc in place (ref: seabreeze_diag_python.f90:237-239,
    # 268-273) and f2py hands Fortran-contiguous float32 arrays through without a copy, so the state is
    # kept in private arrays: the caller's arrays (e.g. what an earlier call returned) are never written.
    # thc is write-only in the kernel: one scratch plane serves every step.
    ws, wd = (np.array(state[k], dtype=np.float32, order="C") for k in ("ws", "wd"))
    thc = np.empty_like(ws)

    has_time = np.ndim(v) > 3
    nt = len(v) if has_time else 1
    nlat, nlon = np.shape(t)[-2:]
    sb_all = np.zeros([nt, nlat, nlon])           # float64 like the reference's (ref :214)
    dist = None if ci is not None else _coast_distance(lsm, np.zeros_like(lsm), lon, lat)
    t0 = thc
    for ts in range(nt):
        if ci is not None:
            ice = ci[ts] if has_time else ci
            ice = ice.filled(0) if hasattr(ice, "filled") else ice
            dist = _coast_distance(lsm, ice, lon, lat)       # the reference recomputes it every step (:223-228); here: when ice changes
        tk, vk, uk = (t[ts], v[ts], u[ts]) if has_time else (t[:], v[:], u[:])
        out = c2f(_diag_kernel(tt, c2f(pres), c2f(z), c2f(std), c2f(tk), c2f(vk), c2f(uk), c2f(dist),
                               c2f(ws), c2f(wd), c2f(thc), **kwargs))
        sb_all[ts] = out[0]
        t0, ws, wd = out[1], out[2], out[3]              # views: the next step reads them where they lie
        tt += 1
    # what the reference returns as "thc" is the t0 plane (ref :244); copies, so that the arrays handed out
    # do not alias the (4-plane) output buffer of the last step
    return tt, sb_all, np.array(t0), np.array(ws), np.array(wd)


def read_nc(fnv, fnu, fntheta, fnci, vv="v", vu="u", vtheta="t2m", vci="ci", vpres="pres", vtime="time"):
    """Open the four NetCDF inputs and return an object usable as ``diag(..., meta=...)``
    (ref: python_wrapper/seabreezediag/__init__.py:53-89).  Needs netCDF4, which this image
    does not ship; the import is deferred so the rest of the package works without it."""
    import os
    from types import SimpleNamespace

    from netCDF4 import Dataset, num2date

    files = {"v": fnv, "u": fnu, "theta": fntheta, "ci": fnci}
    varname = {"v": vv, "u": vu, "theta": vtheta, "ci": vci}
    meta = SimpleNamespace(nc={k: Dataset(os.path.expanduser(f)) for k, f in files.items()})
    for key, ds in meta.nc.items():
        setattr(meta, key, ds.variables[varname[key]])
    tvar = meta.nc["v"].variables[vtime]
    meta.time = num2date(tvar[:], tvar.units)
    meta.pres = meta.nc["v"].variables[vpres][:]
    meta.dt = (meta.time[1] - meta.time[0]).seconds / 60.0
    return meta
