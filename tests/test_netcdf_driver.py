"""The file driver either side of the path: configuration files, discovery of the daily /
monthly NetCDF inputs, reading them, carrying the state from file to file, writing the result files.

The reference's driver (python_wrapper/test_run.py, seabreezediag/configdir.py) needs netCDF4 and private reanalysis
files, neither of which exists here, and holds no fixtures: this part is **parity unpinned** against the reference's
own I/O.  What is checked: the syntax the reference's run.conf uses, the file-name scheme, and that the numbers that
reach the result file are exactly what `seabreezediag.diag` (oracle-checked elsewhere) returns for the same arrays.
Files are classic NetCDF written with scipy (`ncio` falls back to it where netCDF4 cannot be imported).
"""
import os
import sys
from datetime import datetime, timedelta

import numpy as np
import pytest
from scipy.io import netcdf_file

from conftest import ROOT
from seabreeze_param_amd import synth

PW = os.path.join(ROOT, "python_wrapper")


def _surface():
    if PW not in sys.path:
        sys.path.insert(0, PW)
    import seabreezediag
    from seabreezediag import configdir, ncio
    return seabreezediag, configdir, ncio


def _built():
    import glob
    return bool(glob.glob(os.path.join(PW, "seabreeze*.so")))


pytestmark = pytest.mark.skipif(not _built(), reason="python_wrapper extension not built (run __graft_entry__.build())")

NLON, NLAT, NLEV, NT = 96, 72, 4, 4


def _write(path, dims, variables):
    os.makedirs(os.path.dirname(path), exist_ok=True)
    f = netcdf_file(path, "w")
    for name, size in dims:
        f.createDimension(name, size)
    for name, (typ, vdims, data, attrs) in variables.items():
        v = f.createVariable(name, typ, vdims)
        v[:] = data
        for k, a in attrs.items():
            setattr(v, k, a)
    f.close()


def _dataset(root, stamps, monthly=False):
    """Static files + one u / v / t2m / ci file per stamp; returns the arrays per stamp."""
    st = synth.static_fields(NLON, NLAT, np.float32)
    lon, lat = st.lon.astype(np.float32), st.lat.astype(np.float32)
    axes = {"longitude": ("f", ("longitude",), lon, {}), "latitude": ("f", ("latitude",), lat, {})}
    hdims = [("longitude", NLON), ("latitude", NLAT)]
    _write(os.path.join(root, "static_lsm.nc"), [("time", 1)] + hdims,
           dict(axes, lsm=("f", ("time", "latitude", "longitude"), st.landfrac[None], {})))
    _write(os.path.join(root, "static_topo.nc"), hdims,
           dict(axes, z=("f", ("latitude", "longitude"), st.z, {}), sdfor=("f", ("latitude", "longitude"), st.sigma, {})))
    pres = (synth.pressure_1d(NLEV, np.float32) / 100.0).astype(np.float32)
    out, t0 = {}, 0
    for stamp in stamps:
        year = stamp.split("_")[0]
        first = datetime.strptime(stamp, "%Y_%m" if monthly else "%Y_%m_%d")
        hours = np.array([(first + timedelta(hours=6 * i) - datetime(1900, 1, 1)).total_seconds() / 3600 for i in range(NT)])
        u = np.stack([synth.wind_step(st, NLEV, t0 + i + 1, np.float32)[0] for i in range(NT)])
        v = np.stack([synth.wind_step(st, NLEV, t0 + i + 1, np.float32)[1] for i in range(NT)])
        th = np.stack([synth.theta_step(st, t0 + i + 1, np.float32) for i in range(NT)])
        ci = np.stack([np.clip(st.icefrac + 0.01 * ((t0 + i) // 2), 0, 1).astype(np.float32) for i in range(NT)])
        t0 += NT
        time = ("d", ("time",), hours, {"units": "hours since 1900-01-01 00:00:0.0"})
        d3 = [("time", None), ("latitude", NLAT), ("longitude", NLON)]
        d4 = [("time", None), ("level", NLEV), ("latitude", NLAT), ("longitude", NLON)]
        lev = ("f", ("level",), pres, {"units": "millibars"})
        for name, arr in (("u", u), ("v", v)):
            _write(os.path.join(root, year, f"Erai_{name}_{stamp}.nc"), d4,
                   dict(axes, time=time, level=lev, **{name: ("f", ("time", "level", "latitude", "longitude"), arr, {})}))
        for name, arr in (("t2m", th), ("ci", ci)):
            _write(os.path.join(root, year, f"Erai_{name}_{stamp}.nc"), d3,
                   dict(axes, time=time, **{name: ("f", ("time", "latitude", "longitude"), arr, {})}))
        out[stamp] = dict(u=u, v=v, th=th, ci=ci, hours=hours)
    return st, pres, out


def _config(root, start, end):
    path = os.path.join(root, "run.conf")
    with open(path, "w") as f:
        f.write(f"""# test configuration in the reference's syntax (ref: python_wrapper/run.conf)
datadir = {root}
landfracfile = {root}/static_lsm.nc   # the land-sea mask
topofile = '{root}/static_topo.nc'
orofile = {root}/static_topo.nc
prefix = Erai_
vtheta = t2m
vu = u
 vv = v
vci = ci
vlon = longitude
vlat = latitude
vpres = level
vlandfrac = lsm
vz = z
vstd = sdfor
vtime = time
plev = 700
start = {start} #format yyyy-mm-dd_HH:MM
end = {end}
""")
    return path


def test_config_syntax(tmp_path):
    _, configdir, _ = _surface()
    p = tmp_path / "c.conf"
    p.write_text("#Filename of the test data\nfilename = 'foo.nc' #\nvariable = bar # The variable\n x1 = 9.0 # First index\n"
                 "x2 =10  # Last index\nupdate = true\ntimes = 1,2,3 # Some time steps\nnothing = None\nhome = $HOME/Data\n"
                 "[section]\nnot a setting\n")
    c = configdir.Config(str(p))
    assert dict(c) == {"filename": "foo.nc", "variable": "bar", "x1": 9.0, "x2": 10, "update": True, "times": (1.0, 2.0, 3.0),
                       "nothing": None, "home": os.environ["HOME"] + "/Data"}
    assert c.x2 == 10 and c.variable == "bar" and isinstance(c.x2, int)
    with pytest.raises(AttributeError):
        c.missing
    c.extra = 5
    assert c["extra"] == 5 and "x1" in repr(c)


def test_time_axis_conversion():
    _, _, ncio = _surface()
    d = ncio.num2date([0, 6, 30.5], "hours since 1900-01-01 00:00:0.0")
    assert d == [datetime(1900, 1, 1), datetime(1900, 1, 1, 6), datetime(1900, 1, 2, 6, 30)]
    assert ncio.num2date([86400], b"seconds since 1970-01-01")[0] == datetime(1970, 1, 2)
    n = ncio.date2num([datetime(1987, 1, 1, 6)], "Seconds since 1970-01-01 00:00:00")
    assert n[0] == (datetime(1987, 1, 1, 6) - datetime(1970, 1, 1)).total_seconds()
    with pytest.raises(ValueError):
        ncio.num2date([0], "fortnights since 1900-01-01")


@pytest.mark.parametrize("monthly", [False, True])
def test_meta_finds_inputs_and_writes_results(tmp_path, monthly):
    _, configdir, ncio = _surface()
    root = str(tmp_path)
    stamps = ["1987_01", "1987_02"] if monthly else ["1987_01_01", "1987_01_02", "1987_01_04"]
    st, pres, _ = _dataset(root, stamps, monthly)
    cfg = configdir.Config(_config(root, "1987-01-01_00:00", "1987-02-10_18:00" if monthly else "1987-01-06_00:00"))
    m = configdir.Meta(cfg)
    assert m.dates == stamps                               # 1987_01_03 has no files and is left out
    assert m.landfrac.shape == (NLAT, NLON) and np.array_equal(m.landfrac, st.landfrac) and np.array_equal(m.std, st.sigma)
    assert np.allclose(m.lon, st.lon) and m.start == datetime(1987, 1, 1)
    assert m.input_file("u", stamps[0]).endswith(f"1987/Erai_u_{stamps[0]}.nc")
    # result file: created, then extended by a second variable
    times = [datetime(1987, 1, 1) + timedelta(hours=6 * i) for i in range(3)]
    data = np.arange(3 * NLAT * NLON, dtype=np.float64).reshape(3, NLAT, NLON) / 7.0
    out = os.path.join(root, "1987", "Erai_sb_test.nc")
    m.create_nc(data, out, "sb_con", times)
    m.create_nc(data * 2, out, "thc", times, add=" (test)")
    f = ncio.open_dataset(out)
    try:
        assert np.array_equal(f.variables["sb_con"][:], data.astype(np.float32))
        assert np.array_equal(f.variables["thc"][:], (data * 2).astype(np.float32))
        assert list(f.variables["time"][:]) == [int((t - datetime(1970, 1, 1)).total_seconds()) for t in times]
        assert ncio._text(f.variables["time"].units).startswith("Seconds since 1970")
        assert ncio._text(f.variables["thc"].long_name).endswith("(test)") and ncio._text(f.variables["thc"].units) == "K"
        assert np.float32(f.variables["sb_con"].missing_value) == np.float32(2.0e20)
        assert np.allclose(f.variables["lat"][:], st.lat)
    finally:
        f.close()
    with pytest.raises(ValueError):
        configdir.Meta(configdir.Config(_config(root, "1990-01-01_00:00", "1990-01-03_00:00")))


@pytest.mark.gpu
def test_file_driver_end_to_end(tmp_path, oracles):
    """run_seabreeze.main on two daily files: the result files hold what the CPU ORACLE computes stepping the files'
    arrays (wrapper flavour, fp32: coast from the land mask and every step's ice, coast distance, diag with the carried
    state -- the sequence of ref python_wrapper/seabreezediag/__init__.py:222-245), and, bit for bit, what
    seabreezediag.diag returns for the same arrays; the timestep counter and the state are threaded from the first
    file into the second."""
    sbd, configdir, ncio = _surface()
    orc = oracles[4]
    so = None
    otn = 1
    import run_seabreeze
    root = str(tmp_path)
    stamps = ["1987_01_01", "1987_01_02"]
    st, pres, data = _dataset(root, stamps)
    written = run_seabreeze.main(_config(root, "1987-01-01_00:00", "1987-01-03_00:00"), verbose=False)
    assert [os.path.basename(w) for w in written] == [f"Erai_sb_{s}.nc" for s in stamps]
    tt, ws, wd, thc = 1, None, None, None
    for stamp, path in zip(stamps, written):
        d = data[stamp]
        kw = {} if ws is None else dict(ws=ws, wd=wd, thc=thc)
        tt, sb, thc, ws, wd = sbd.diag(tt, st.landfrac, st.z, st.sigma, st.lon, st.lat, pres, d["u"], d["v"], d["th"], d["ci"], **kw)
        f = ncio.open_dataset(path)
        try:
            got = np.array(f.variables["sb_con"][:])
            assert got.shape == (NT, NLAT, NLON)
            assert np.array_equal(got, sb.astype(np.float32), equal_nan=True)
            # ... and the oracle, stepping the same arrays
            f32 = lambda a: np.ascontiguousarray(a, dtype=np.float32)
            if so is None:
                so = [np.zeros((NLAT, NLON), np.float32) for _ in range(3)]
            for k in range(NT):
                coast = orc.get_edges(f32(st.landfrac), f32(d["ci"][k]))
                cdist = orc.get_dist(coast, f32(st.landfrac), f32(st.lon), f32(st.lat))
                oo = orc.diag(otn, f32(pres), f32(st.z), f32(st.sigma), f32(d["th"][k]), f32(d["v"][k]), f32(d["u"][k]), cdist, *so)
                otn += 1
                band = oo[0, :-1] < 1e19
                near = np.abs(np.abs(so[2][:-1]) - 0.75) < 5e-3            # the 0.75 K knife edge of the trigger
                ok = band & ~near
                assert np.array_equal(got[k, :-1][~band] > 1e19, np.ones((~band).sum(), bool)), (stamp, k)
                assert np.max(np.abs(got[k, :-1][ok] - oo[0, :-1][ok]), initial=0.0) < 5e-3, (stamp, k)
            first = datetime.strptime(stamp, "%Y_%m_%d")
            assert list(f.variables["time"][:]) == [int((first + timedelta(hours=6 * i) - datetime(1970, 1, 1)).total_seconds())
                                                     for i in range(NT)]
        finally:
            f.close()
        assert np.count_nonzero((sb != 0) & (sb < 1e19)) > 0       # the sea breeze triggers somewhere: not vacuous
    assert tt == 1 + 2 * NT
