"""CPU-side checks of the product's boundary and host logic (no compute on a GPU):

* libseabreeze_hip.so loads and exports every entry point include/seabreeze_hip.h declares;
* without a device the library refuses to work (no CPU fallback);
* the synthetic generator and the band-decomposition helpers behave.
"""
import ctypes as C
import os
import re

import numpy as np
import pytest

from conftest import ROOT
from seabreeze_param_amd import bands, hip, synth


def _declared_symbols():
    text = open(os.path.join(ROOT, "include", "seabreeze_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(sb_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    lib = hip.load_library()
    names = _declared_symbols()
    assert len(names) >= 30
    missing = [n for n in names if not hasattr(lib, n)]
    assert not missing, f"declared in seabreeze_hip.h but not exported: {missing}"
    assert b"gfx950" in lib.sb_version()


def test_strip_kernels_wait_for_their_prefetched_blocks():
    """The march of k_strip keeps three blocks of loads in flight; where the compiler puts the s_waitcnt vmcnt(N) for
    them has gone wrong silently twice (N collapsing to 0..2; no wait at all): tools/check_waits.py reads the
    disassembly of the built library and wants three waits at full depth in every variant."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("check_waits", os.path.join(ROOT, "tools", "check_waits.py"))
    cw = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(cw)
    if not os.path.exists(cw.OBJDUMP):
        pytest.skip("no llvm-objdump in this image")
    waits = cw.strip_kernel_waits(hip.LIB_PATH if hasattr(hip, "LIB_PATH") else os.path.join(ROOT, "seabreeze_param_amd", "libseabreeze_hip.so"))
    # k_strip: float / double x t0 on the fly or from the plane; k_strip32 (single precision): on the fly or from the plane
    assert len(waits) == 6, sorted(waits)
    for name, hist in waits.items():
        need = cw.expected_depth(name)
        assert sum(v for k, v in hist.items() if k >= need) >= 3, (name, sorted(hist.items()))


def test_no_device_means_loud_failure():
    from conftest import gpu_visible
    if gpu_visible():
        pytest.skip("a GPU is visible here")
    lib = hip.load_library()
    h = C.c_void_p()
    rc = lib.sb_create(C.byref(h), C.c_int(-1))
    assert rc == 3 and not h.value                       # SB_ERR_NO_DEVICE, no context
    assert b"no CPU fallback" in lib.sb_last_error(None)
    with pytest.raises(hip.SeabreezeHipError):
        hip.Context()


def test_argument_errors_without_context():
    lib = hip.load_library()
    k = C.c_int(0)
    lon = np.array([0.0, 1.0]); lat = np.array([0.0, 1.0])
    assert lib.sb_dist_window_f64(C.c_int(1), C.c_int(2), lon.ctypes.data_as(C.c_void_p),
                                  lat.ctypes.data_as(C.c_void_p), C.c_double(180.0), C.byref(k)) == 1
    assert lib.sb_synchronize(None) == 1
    t = hip.Tunables()
    lib.sb_default_tunables(C.byref(t))
    assert (t.target_plev_pa, t.thresh_wind, t.thresh_winddir, t.thresh_windch, t.thresh_thc,
            t.target_time_s, t.maxdist_km) == (70000.0, 11.0, 90.0, 5.0, 0.75, 21600.0, 180.0)


@pytest.mark.parametrize("prec,dt", [(8, np.float64), (4, np.float32)])
@pytest.mark.parametrize("shape", [(96, 72), (256, 192), (1024, 768), (2560, 1920)])
def test_dist_window_matches_oracle(oracles, prec, dt, shape):
    """Host-side half of get_dist: the window half-width (sobel.f90:129-137): 0/1/6/15 at these grids."""
    lon, lat = synth.grid(*shape)
    lon, lat = lon.astype(dt), lat.astype(dt)
    assert hip.dist_window(lon, lat) == oracles[prec].dist_window(lon, lat)


def test_synth_is_deterministic_and_shaped():
    a = synth.static_fields(96, 72)
    b = synth.static_fields(96, 72)
    assert np.array_equal(a.sigma, b.sigma) and np.array_equal(a.landfrac, b.landfrac)
    assert a.sigma.shape == (72, 96) and a.lon.shape == (96,) and a.lat.shape == (72,)
    u, v = synth.wind_step(a, 3, 5)
    assert u.shape == (3, 72, 96) and u.flags.c_contiguous
    assert not np.array_equal(synth.theta_step(a, 1), synth.theta_step(a, 2))
    un = synth.hash_uniform((1000,), 7)
    assert 0.0 <= un.min() and un.max() < 1.0 and np.all(un * (1 << 24) == np.round(un * (1 << 24)))
    p = synth.pressure_3d(a, 5)
    assert np.all(np.diff(p, axis=0) < 0)                 # pressure falls with level


def test_split_rows_and_ghost_fill():
    for ny, w in ((1920, 8), (72, 3), (7, 7), (10, 4)):
        sp = bands.split_rows(ny, w)
        assert sp[0][0] == 0 and sp[-1][1] == ny
        assert all(a[1] == b[0] for a, b in zip(sp, sp[1:]))
        sizes = [b - a for a, b in sp]
        assert max(sizes) - min(sizes) <= 1
    # cuts by cost: contiguous, complete, no band thinner than the ghost frame, and better balanced than equal rows
    st = synth.static_fields(256, 192)
    band = (st.landfrac > 0) ^ np.roll(st.landfrac > 0, 3, axis=1)          # a clustered stand-in for the coastal band
    cost = bands.row_cost(band, 56)
    for w, hmin in ((2, 1), (4, 8), (8, 17)):
        sp = bands.split_rows(192, w, cost=cost, min_rows=hmin)
        assert sp[0][0] == 0 and sp[-1][1] == 192 and all(a[1] == b[0] for a, b in zip(sp, sp[1:]))
        assert min(b - a for a, b in sp) >= hmin
        worst = max(cost[a:b].sum() for a, b in sp)
        worst_equal = max(cost[a:b].sum() for a, b in bands.split_rows(192, w))
        assert worst <= worst_equal * 1.0001
    with pytest.raises(ValueError):
        bands.split_rows(10, 4, cost=np.ones(10), min_rows=3)
    nx, h = 10, 3
    core = np.arange(4 * nx, dtype=np.float64).reshape(4, nx)
    loc = np.zeros((4, nx + 2 * h))
    loc[:, h:h + nx] = core
    bands.fill_ew_ghosts(loc, nx, h)
    assert np.array_equal(loc[:, :h], core[:, -h:]) and np.array_equal(loc[:, -h:], core[:, :h])


def test_split_rows_properties():
    """Any cost vector, any rank count: the bands are contiguous, cover every row once, respect the minimum height,
    and every rank derives the same cuts (pure function of its arguments)."""
    from hypothesis import given, settings, strategies as hs

    @settings(max_examples=200, deadline=None)
    @given(hs.integers(1, 12), hs.integers(1, 6), hs.integers(0, 2 ** 32 - 1), hs.sampled_from(["flat", "spike", "ramp", "zeros"]))
    def check(world, min_rows, seed, kind):
        rng = np.random.default_rng(seed)
        ny = world * min_rows + int(rng.integers(0, 200))
        cost = {"flat": np.ones(ny), "zeros": np.zeros(ny), "ramp": np.arange(ny, dtype=float),
                "spike": np.where(rng.random(ny) > 0.95, 1000.0, 1.0)}[kind] * rng.random()
        sp = bands.split_rows(ny, world, cost=cost, min_rows=min_rows)
        assert len(sp) == world and sp[0][0] == 0 and sp[-1][1] == ny
        assert all(a[1] == b[0] for a, b in zip(sp, sp[1:]))
        assert all(b - a >= min_rows for a, b in sp)
        assert sp == bands.split_rows(ny, world, cost=cost.copy(), min_rows=min_rows)

    check()


def test_bench_watchdog_ends_a_stalled_rank():
    """bench.py, N > 1: a rank that passes no stage mark within the limit reports its stage and exits with code 5 (the
    multi-rank RCCL exchange has never run on hardware: a hang must not hold the launcher for ever)."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = "import bench, time; d = bench.Watchdog(1.0, 3); d.mark('ghost rows'); time.sleep(20)"
    r = subprocess.run([sys.executable, "-c", code], cwd=root, capture_output=True, text=True, timeout=120)
    assert r.returncode == 5, (r.returncode, r.stderr[-400:])
    assert "rank 3" in r.stderr and "ghost rows" in r.stderr
    code = "import bench, time; d = bench.Watchdog(0.0, 0); time.sleep(2.5); print('alive')"
    r = subprocess.run([sys.executable, "-c", code], cwd=root, capture_output=True, text=True, timeout=120)
    assert r.returncode == 0 and "alive" in r.stdout
