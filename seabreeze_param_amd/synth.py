"""Deterministic synthetic inputs for the sea-breeze diagnostic.

The reference ships no data (its driver needs private ERA-Interim files,
/root/reference/python_wrapper/run.conf:2-10), so tests and bench.py feed both
the HIP path and the CPU oracle from this generator.

Memory layout: every array is a C-contiguous numpy array whose shape is the
REVERSE of the Fortran shape, so its buffer is exactly the Fortran array
``f(lon, lat[, lev])`` with longitude fastest:

    2-D field  -> shape (ny, nx)
    3-D field  -> shape (nz, ny, nx)

Noise comes from a counter-based integer hash mapped to dyadic rationals, so a
given (seed, grid, step) always regenerates the same noise whatever the libm.
"""
from __future__ import annotations

from dataclasses import dataclass, field

import numpy as np

SEED = 20170828  # the reference module's "Initial Version" date (generic/sea_breeze_diag.f90:15)

_M1 = np.uint64(0xBF58476D1CE4E5B9)
_M2 = np.uint64(0x94D049BB133111EB)
_G = np.uint64(0x9E3779B97F4A7C15)


def _mix(x: np.ndarray) -> np.ndarray:
    """splitmix64 finaliser on a uint64 array (wraps modulo 2**64)."""
    x = x.astype(np.uint64, copy=True)
    x ^= x >> np.uint64(30)
    x *= _M1
    x ^= x >> np.uint64(27)
    x *= _M2
    x ^= x >> np.uint64(31)
    return x


def hash_uniform(shape, stream: int, seed: int = SEED, rows=None) -> np.ndarray:
    """U[0,1) on a regular index grid: 24-bit dyadic rationals (exact in fp32).

    `rows=(r0, r1)` returns only rows r0:r1 of the second-to-last axis (a latitude band) of
    the field `shape` describes, with the same values the full field has there."""
    with np.errstate(over="ignore"):
        if rows is None:
            ctr = np.arange(int(np.prod(shape)), dtype=np.uint64)
            out_shape = tuple(shape)
        else:
            r0, r1 = rows
            ny, nx = shape[-2], shape[-1]
            lead = int(np.prod(shape[:-2])) if len(shape) > 2 else 1
            plane = (np.arange(r0, r1, dtype=np.uint64)[:, None] * np.uint64(nx)
                     + np.arange(nx, dtype=np.uint64)[None, :])
            ctr = (np.arange(lead, dtype=np.uint64)[:, None, None] * np.uint64(ny * nx) + plane[None]).ravel()
            out_shape = tuple(shape[:-2]) + (r1 - r0, nx)
        key = _mix(np.array([np.uint64(seed) * _G + np.uint64(stream)], dtype=np.uint64))[0]
        bits = _mix(ctr * _G + key)
    u = (bits >> np.uint64(40)).astype(np.float64) * (1.0 / (1 << 24))
    return u.reshape(out_shape)


def hash_normal(shape, stream: int, seed: int = SEED) -> np.ndarray:
    """Approximately N(0,1): Irwin-Hall sum of four hashed uniforms."""
    s = sum(hash_uniform(shape, stream * 4 + q, seed) for q in range(4))
    return (s - 2.0) * 1.7320508075688772


def grid(nx: int, ny: int):
    """Regular global lat-lon cell centres, degrees."""
    lon = 360.0 * np.arange(nx) / nx
    lat = -90.0 + (np.arange(ny) + 0.5) * 180.0 / ny
    return lon, lat


@dataclass
class StaticFields:
    nx: int
    ny: int
    lon: np.ndarray
    lat: np.ndarray
    landfrac: np.ndarray  # (ny, nx) 0/1
    icefrac: np.ndarray   # (ny, nx)
    z: np.ndarray         # surface height, m
    sigma: np.ndarray     # std of sub-grid orography
    f: np.ndarray = field(repr=False, default=None)


def static_fields(nx: int, ny: int, dtype=np.float64, seed: int = SEED,
                  fractional_coast: bool = False) -> StaticFields:
    lon, lat = grid(nx, ny)
    lam = np.deg2rad(lon)[None, :]
    phi = np.deg2rad(lat)[:, None]
    f = (np.sin(3 * lam) * np.cos(2 * phi)
         + 0.5 * np.sin(5 * lam + 1.0) * np.sin(4 * phi)
         + 0.3 * np.cos(7 * lam - 2 * phi))
    land = (f > 0.35).astype(np.float64)
    if fractional_coast:
        # fractional land-area values near the shoreline exercise the >0.4 / >=0.5 / >0 rules
        frac = np.clip((f - 0.25) / 0.2, 0.0, 1.0)
        land = np.round(frac * 8) / 8
    ice = np.where((np.abs(lat)[:, None] > 75.0) & (land < 0.5), 0.6, 0.0)
    z = np.maximum(0.0, f - 0.35) * 2000.0 * (land > 0)
    sigma = 0.1 * z + 5.0 * hash_uniform((ny, nx), 1, seed)
    cast = lambda a: np.ascontiguousarray(a, dtype=dtype)
    return StaticFields(nx, ny, cast(lon), cast(lat), cast(land), cast(ice), cast(z), cast(sigma), f)


def sigma_levels(nz: int) -> np.ndarray:
    """nz values decreasing geometrically 1.0 -> 0.02."""
    if nz == 1:
        return np.array([0.7])
    return 0.02 ** (np.arange(nz) / (nz - 1.0))


def pressure_1d(nz: int, dtype=np.float64) -> np.ndarray:
    return np.ascontiguousarray(101325.0 * sigma_levels(nz), dtype=dtype)


def pressure_3d(st: StaticFields, nz: int, dtype=np.float64, rows=None) -> np.ndarray:
    r0, r1 = rows if rows is not None else (0, st.ny)
    ps = 101325.0 - 11.5 * st.z[r0:r1].astype(np.float64)
    return np.ascontiguousarray(sigma_levels(nz)[:, None, None] * ps[None], dtype=dtype)


def theta_step(st: StaticFields, t: int, dtype=np.float64, seed: int = SEED) -> np.ndarray:
    phi = np.deg2rad(st.lat.astype(np.float64))[:, None]
    land = (st.landfrac.astype(np.float64) > 0)
    th = (288.0 + 15.0 * np.cos(phi) + 4.0 * land
          + hash_normal((st.ny, st.nx), 100 + t, seed)
          + 3.0 * land * np.sin(2 * np.pi * t / 60.0))
    return np.ascontiguousarray(th, dtype=dtype)


def wind_step(st: StaticFields, nz: int, t: int, dtype=np.float64, seed: int = SEED, rows=None):
    """u, v of shape (nz, ny, nx): speed straddles 11 m/s, direction drifts with t.
    `rows=(r0, r1)` generates only that latitude band, identical to the full field's rows."""
    r0, r1 = rows if rows is not None else (0, st.ny)
    lam = np.deg2rad(st.lon.astype(np.float64))[None, None, :]
    phi = np.deg2rad(st.lat[r0:r1].astype(np.float64))[None, :, None]
    lev = np.arange(nz, dtype=np.float64)[:, None, None]
    psi = 2.0 * lam + 3.0 * phi + 0.11 * lev + 0.02 * t
    shape = (nz, st.ny, st.nx)
    u = 7.0 * np.cos(psi) + 4.0 * (2.0 * hash_uniform(shape, 1000 + 2 * t, seed, rows=rows) - 1.0)
    v = 7.0 * np.sin(psi) + 4.0 * (2.0 * hash_uniform(shape, 1001 + 2 * t, seed, rows=rows) - 1.0)
    return np.ascontiguousarray(u, dtype=dtype), np.ascontiguousarray(v, dtype=dtype)
