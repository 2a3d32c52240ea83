"""Run configuration and static meta data of the file driver.

Counterparts of the reference's `Config` and `Meta` (ref: python_wrapper/seabreezediag/configdir.py:15-345), written
for this package: `Config` reads the `key = value  # comment` files the reference ships (ref: python_wrapper/run.conf)
into a dict with attribute access; `Meta` holds what is constant over a run -- the land fraction, surface height and
orography deviation on their grid, the period, and the list of input files found for it -- and writes the result
files.  File access goes through `ncio` (netCDF4 where available, scipy's NetCDF-3 reader/writer otherwise).
"""
from __future__ import annotations

import os
from datetime import datetime, timedelta

import numpy as np

from . import ncio

__all__ = ["Config", "Meta"]

_STRIP = "[]{}@\"'"


def _convert(text, maketuple=True):
    """'10' -> 10, '9.0' -> 9.0, 'true' -> True, 'none' -> None, '1,2,x' -> (1.0, 2.0, 'x'), else the string."""
    for conv in (int, float):
        try:
            return conv(text)
        except ValueError:
            pass
    low = text.lower()
    if low in ("true", "false"):
        return low == "true"
    if low == "none":
        return None
    if maketuple and "," in text:
        items = []
        for part in text.split(","):
            part = part.strip("()")
            try:
                items.append(float(part))
            except ValueError:
                items.append(part)
        return tuple(items)
    return text


class Config(dict):
    """`Config(filename)`: the entries of a configuration file as a dict whose keys are also attributes.

    One `key = value` per line; `#` starts a comment (whole line or trailing); blanks inside a value are dropped
    (`skipwhitespace`), quotes and brackets are stripped; numbers, `true/false/none` and comma-separated tuples are
    converted; a value starting with `$NAME/` has the environment variable substituted
    (ref: configdir.py:236-345 for the accepted syntax)."""

    def __init__(self, filename, maketuple=True, skipwhitespace=True, split="="):
        super().__init__()
        with open(os.path.expanduser(filename)) as f:
            lines = f.read().splitlines()
        for line in lines:
            body = line.split("#", 1)[0]
            if split not in body or not body.strip() or body.lstrip()[0] in _STRIP:
                continue
            key, value = body.split(split, 1)
            key = key.replace(" ", "").strip()
            value = value.replace(" ", "") if skipwhitespace else value.strip()
            value = value.strip().translate({ord(c): None for c in _STRIP})
            if not key:
                continue
            value = _convert(value, maketuple)
            if isinstance(value, str) and value.startswith("$"):
                name = value.split("/", 1)[0]
                try:
                    value = value.replace(name, os.environ[name[1:]], 1)
                except KeyError:
                    raise KeyError(f"environment variable {name} (used by '{key}') is not set") from None
            self[key] = value

    def __getattr__(self, key):
        try:
            return self[key]
        except KeyError:
            raise AttributeError(f"no entry '{key}'; entries are: {', '.join(sorted(self))}") from None

    def __setattr__(self, key, value):
        self[key] = value

    def __repr__(self):
        width = max([8] + [len(str(k)) for k in self]) + 1
        rows = [f"{'Keys':<{width}}| Values", "-" * (width + 9)]
        rows += [f"{str(k):<{width}}| {v}" for k, v in self.items()]
        return "\n".join(rows) + "\n"


def _plane(var):
    """The 2-D field of a static variable stored as (t, z, y, x), (t, y, x) or (y, x)."""
    a = np.asarray(var[:])
    while a.ndim > 2:
        a = a[0]
    return _native(a)


def _native(a):
    """A copy in native byte order (classic NetCDF data arrive big-endian)."""
    a = np.asarray(a)
    return np.array(a, dtype=a.dtype.newbyteorder("="))


_OUTPUT = {"thc": ("Thermal Heating Contrast Between Land and Ocean", "K"),
           "sb_con": ("Subgrid Sea-Breeze Convergence", " "),
           "windspeed": ("Coastal Windspeed", "m/s"),
           "winddir": ("Coastal Wind Direction", "deg"),
           "temp": ("2M Temperture", "degC")}


class Meta:
    """Static fields and the file list of one run.

    Attributes: `landfrac, z, std` (2-D, from `landfracfile/topofile/orofile`, variables `vlandfrac/vz/vstd`), `lon,
    lat`, `start, end` (datetimes from 'YYYY-MM-DD_HH:MM'), `datadir, prefix, vtheta, vu, vv, vpres`, and `dates`:
    the 'YYYY_MM_DD' (daily files) or 'YYYY_MM' (monthly files) stamps between start and end for which
    `<datadir>/<YYYY>/<prefix><var>_<stamp>.nc` exists for the wind and temperature variables
    (ref: configdir.py:41-119)."""

    def __init__(self, C):
        for fn, name in ((C.landfracfile, "landfrac"), (C.topofile, "z"), (C.orofile, "std")):
            f = ncio.open_dataset(fn)
            try:
                setattr(self, name, _plane(f.variables[C["v" + name]]))
                self.lon = _native(f.variables[C.vlon][:])
                self.lat = _native(f.variables[C.vlat][:])
            finally:
                f.close()
        self.start = datetime.strptime(C.start, "%Y-%m-%d_%H:%M")
        self.end = datetime.strptime(C.end, "%Y-%m-%d_%H:%M")
        for key in ("vtheta", "prefix", "vpres", "vu", "vv"):
            setattr(self, key, C[key])
        self.datadir = os.path.expanduser(C.datadir)
        self.dates = self._find_dates()

    def input_file(self, var, stamp):
        return os.path.join(self.datadir, stamp.split("_")[0], f"{self.prefix}{var}_{stamp}.nc")

    def _find_dates(self):
        def present(stamp):
            return all(os.path.isfile(self.input_file(v, stamp)) for v in (self.vv, self.vu, self.vtheta))

        day, stamps = self.start, []
        daily = monthly = False
        while day < self.end:
            for fmt in ("%Y_%m_%d", "%Y_%m"):
                stamp = day.strftime(fmt)
                if stamp not in stamps and present(stamp):
                    stamps.append(stamp)
                    daily |= fmt == "%Y_%m_%d"
                    monthly |= fmt == "%Y_%m"
            day += timedelta(days=1)
        if not stamps:
            raise ValueError(f"no daily (..._YYYY_MM_DD.nc) or monthly (..._YYYY_MM.nc) input files for "
                             f"{self.start:%Y-%m-%d} .. {self.end:%Y-%m-%d} under {self.datadir}")
        if daily and monthly:                       # the reference takes daily files when both exist (ref :86-95)
            stamps = [s for s in stamps if s.count("_") == 2]
        return stamps

    def read(self, f, varname, timestep=None):
        """A variable of an open file, whole or one time slice (ref: configdir.py:65-80)."""
        return f.variables[varname][:] if timestep is None else f.variables[varname][timestep]

    def create_nc(self, data, fname, varname, times, add=""):
        """Write `data (time, lat, lon)` as `varname` into `fname` (created, or extended when it exists) with the
        coordinate variables, CF attributes and the 2.0e20 missing value of the reference's files (ref :120-175)."""
        long_name, units = _OUTPUT[varname]
        f = ncio.open_dataset(fname, "a" if os.path.isfile(fname) else "w")
        try:
            # the record dimension first: the classic format (and scipy's writer) allow only the first one unlimited
            for dim, size, typ in (("time", None, "i"), ("lat", len(self.lat), "f"), ("lon", len(self.lon), "f")):
                if dim not in f.dimensions:
                    f.createDimension(dim, size)
                if dim not in f.variables:
                    f.createVariable(dim, typ, (dim,))
            for name, attrs in (("lon", dict(units="degrees_east", axis="X", long_name="Longitude")),
                                ("lat", dict(units="degrees_north", axis="Y", long_name="Latitude")),
                                ("time", dict(units="Seconds since 1970-01-01 00:00:00", long_name="Time", axis="T"))):
                for k, v in attrs.items():
                    setattr(f.variables[name], k, v)
            f.variables["lon"][:] = self.lon
            f.variables["lat"][:] = self.lat
            tnum = ncio.date2num(times, f.variables["time"].units)
            f.variables["time"][:len(tnum)] = np.asarray(np.rint(tnum), dtype=np.int32)
            if varname not in f.variables:
                f.createVariable(varname, "f", ("time", "lat", "lon"))
            var = f.variables[varname]
            var[:len(tnum)] = np.asarray(data, dtype=np.float32)
            var.long_name = long_name + add
            var.units = units
            var.grid = "lonlat"
            var.missing_value = np.float32(2.0e20)
        finally:
            f.close()
