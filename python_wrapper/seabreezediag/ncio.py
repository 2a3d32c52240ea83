"""NetCDF access for the file driver (the on-disk format either side of the path).

The reference reads and writes its files with the netCDF4 package (ref: python_wrapper/seabreezediag/configdir.py:13,
python_wrapper/test_run.py:1).  Where netCDF4 can be imported it is used here as well (any NetCDF flavour); where it
cannot -- this image -- `scipy.io.netcdf_file` takes over, which covers the classic (NetCDF-3) format.  Both give
objects with `.variables[name]` that slice like arrays and carry attributes, `createDimension`, `createVariable`
and `close`, which is all the driver uses.  CF time axes are converted here (no netCDF4.num2date needed).
"""
from __future__ import annotations

import os
import re
from datetime import datetime, timedelta

import numpy as np

try:                                    # pragma: no cover - not installed in the build image
    from netCDF4 import Dataset as _NC4
except ImportError:                     # the classic-format reader/writer that ships with scipy
    _NC4 = None
from scipy.io import netcdf_file as _NC3

BACKEND = "netCDF4" if _NC4 is not None else "scipy.io.netcdf_file (NetCDF-3 classic)"

_UNIT_SECONDS = {"second": 1.0, "sec": 1.0, "s": 1.0, "minute": 60.0, "min": 60.0, "hour": 3600.0, "hr": 3600.0,
                 "h": 3600.0, "day": 86400.0, "d": 86400.0}


def open_dataset(path, mode="r"):
    """Open `path` ('r' read, 'w' create, 'a' append)."""
    path = os.path.expanduser(path)
    if _NC4 is not None:                # pragma: no cover
        return _NC4(path, mode)
    # mmap=False: arrays stay valid after close() and files can be reopened for appending
    return _NC3(path, mode, mmap=False) if mode != "w" else _NC3(path, "w")


def _parse_units(units):
    m = re.match(r"\s*([A-Za-z]+?)s?\s+since\s+(\d{1,4})-(\d{1,2})-(\d{1,2})(?:[ T](\d{1,2}):(\d{1,2})(?::(\d{1,2})(?:\.\d*)?)?)?", str(units))
    if not m:
        raise ValueError(f"cannot read the time units {units!r}")
    unit = m.group(1).lower()
    if unit not in _UNIT_SECONDS:
        raise ValueError(f"unknown time unit in {units!r}")
    y, mo, d = (int(m.group(i)) for i in (2, 3, 4))
    hh, mm, ss = (int(m.group(i)) if m.group(i) else 0 for i in (5, 6, 7))
    return _UNIT_SECONDS[unit], datetime(y, mo, d, hh, mm, ss)


def num2date(values, units):
    """CF time numbers -> list of datetime (proleptic Gregorian; what reanalysis files use)."""
    scale, origin = _parse_units(_text(units))
    return [origin + timedelta(seconds=float(v) * scale) for v in np.asarray(values, dtype=np.float64).ravel()]


def date2num(times, units):
    scale, origin = _parse_units(_text(units))
    return np.array([(t - origin).total_seconds() / scale for t in times], dtype=np.float64)


def _text(x):
    return x.decode() if isinstance(x, (bytes, bytearray)) else str(x)
