"""The Fortran host side: fortran/sea_breeze_diag_mod.F90 + halo_exchange_mod.F90 + sb_context_mod.F90 driven by
fortran/dummy_model.f90 in the reference's own call order (generic/dummy_model.f90:27-55).

BASELINE.json configs[0]: 96x72 synthetic coastline through the Fortran surface.
CPU: the executables exist and refuse to run without a device (no CPU fallback).
GPU: outputs match the CPU oracle.
"""
import os
import subprocess

import numpy as np
import pytest

from conftest import ROOT, relerr
from seabreeze_param_amd import synth

EXE = {4: os.path.join(ROOT, "fortran", "build", "r4", "dummy_model"),
       8: os.path.join(ROOT, "fortran", "build", "r8", "dummy_model")}


def _write_input(path, prec, nx, ny, nz, halo, nsteps):
    dt = np.float64 if prec == 8 else np.float32
    st = synth.static_fields(nx, ny, dt)
    p = synth.pressure_3d(st, nz, dt)
    steps = []
    with open(path, "wb") as f:
        np.array([nx, ny, nz, halo], dtype=np.int32).tofile(f)
        for a in (st.lon, st.lat, st.landfrac, st.icefrac, st.z, st.sigma, p):
            np.ascontiguousarray(a, dtype=dt).tofile(f)
        for t in range(1, nsteps + 1):
            th = synth.theta_step(st, t, dt)
            u, v = synth.wind_step(st, nz, t, dt)
            for a in (th, u, v):
                a.tofile(f)
            steps.append((th, u, v))
    return st, p, steps


def _built():
    return all(os.path.exists(p) for p in EXE.values())


@pytest.mark.skipif(not _built(), reason="fortran/build not made (run __graft_entry__.build())")
def test_driver_fails_loudly_without_device(tmp_path):
    from conftest import gpu_visible
    if gpu_visible():
        pytest.skip("a GPU is visible here")
    fin, fout = tmp_path / "in.bin", tmp_path / "out.bin"
    _write_input(fin, 8, 96, 72, 2, 4, 1)
    r = subprocess.run([EXE[8], str(fin), str(fout), "1"], capture_output=True, text=True)
    assert r.returncode != 0
    assert "no CPU fallback" in (r.stdout + r.stderr)


@pytest.mark.skipif(not _built(), reason="fortran/build not made (run __graft_entry__.build())")
@pytest.mark.parametrize("prec", [8, 4])
def test_status_entry_points_reject_inconsistent_shapes(tmp_path, prec):
    """seabreeze_diag_status and seabreeze_diag_um answer error = 1 -- the UM copy's status for bad dimensions
    (ref: UM/vn10.7/sea_breeze_diag.F90:102,198-202) -- for the outline's own mixed declaration (mask, theta with half
    a halo beside interior-sized z, sigma; ref: generic/get_all_fields_mod.f90:17-19), for a state field of the wrong
    shape and for a UM mask frame narrower than theta's, all before any device work (so this runs without a GPU)."""
    fin, fout = tmp_path / "in.bin", tmp_path / "out.bin"
    _write_input(fin, prec, 96, 72, 2, 4, 1)
    r = subprocess.run([EXE[prec], str(fin), str(fout), "1", "status"], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "status: 1 1 1" in r.stdout


@pytest.mark.gpu
@pytest.mark.parametrize("prec", [8, 4])
def test_dummy_model_matches_oracle(tmp_path, oracles, prec):
    assert _built(), "fortran/build missing: __graft_entry__.build() makes it"
    nx, ny, nz, halo, nsteps = 96, 72, 5, 4, 4
    dt = np.float64 if prec == 8 else np.float32
    fin, fout = tmp_path / "in.bin", tmp_path / "out.bin"
    st, p, steps = _write_input(fin, prec, nx, ny, nz, halo, nsteps)
    r = subprocess.run([EXE[prec], str(fin), str(fout), str(nsteps)], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr
    out = np.fromfile(fout, dtype=dt).reshape(1 + 4 * nsteps, ny, nx)

    orc = oracles[prec]
    coast = orc.get_edges(st.landfrac, st.icefrac, rule=1, bnd=1)
    cdist = orc.get_dist(coast, st.landfrac, st.lon, st.lat, maxdist=180.0, kwin=halo)
    tol = 1e-7 if prec == 8 else 2e-6
    assert relerr(out[0], cdist, floor=1e-2) < tol
    state = [np.zeros((ny, nx), dt) for _ in range(4)]      # ws wd thc sb_con
    for t, (th, u, v) in enumerate(steps, start=1):
        orc.seabreeze_diag(24 * 60.0, t, p, u, v, th, cdist, st.z, st.sigma, *state, halo=0, bnd=1)
        sb, ws, wd, thc = out[1 + 4 * (t - 1): 1 + 4 * t]
        if prec == 8:
            for a, b, nm in ((ws, state[0], "ws"), (wd, state[1], "wd"), (thc, state[2], "thc"), (sb, state[3], "sb")):
                assert relerr(a, b, floor=1e-2) < 1e-7, f"step {t} {nm}"
        else:
            assert relerr(ws, state[0], floor=1e-3) < 2e-6
            assert np.max(np.abs(thc - state[2])) < 2e-3
            near = np.abs(np.abs(state[2]) - 0.75) < 5e-3
            assert np.max(np.abs(sb - state[3])[~near]) < 5e-3


def _oracle_sequence(orc, st, p, steps, cdist, prec):
    dt = np.float64 if prec == 8 else np.float32
    ny, nx = st.ny, st.nx
    so = [np.zeros((ny, nx), dt) for _ in range(4)]
    out = []
    for t, (th, u, v) in enumerate(steps, start=1):
        orc.seabreeze_diag(1440.0, t, p, u, v, th, cdist, st.z, st.sigma, *so, halo=0, bnd=1)
        out.append([a.copy() for a in so])
    return out


def _check_against_oracle(raw, per, ref, prec, rows=slice(None)):
    """raw: the per-step part of output.bin (sb_con, ws, wd, thc per step); ref: oracle states (ws, wd, thc, sb_con)"""
    for t, so in enumerate(ref):
        blk = raw[t * 4 * per:(t + 1) * 4 * per]
        sb, ws, wd, thc = (blk[i * per:(i + 1) * per].reshape(so[0][rows].shape) for i in range(4))
        if prec == 8:
            for a, b, nm in ((ws, so[0], "ws"), (wd, so[1], "wd"), (thc, so[2], "thc"), (sb, so[3], "sb_con")):
                assert relerr(a, b[rows], floor=1e-2) < 1e-7, (t + 1, nm)
            assert np.array_equal(sb != 0, so[3][rows] != 0)
        else:
            assert relerr(ws, so[0][rows], floor=1e-3) < 2e-6 and relerr(wd, so[1][rows], floor=1e-1) < 2e-5
            assert np.max(np.abs(thc - so[2][rows])) < 2e-3
            near = np.abs(np.abs(so[2][rows]) - 0.75) < 5e-3
            assert np.max(np.abs(sb[~near] - so[3][rows][~near])) < 5e-3


@pytest.mark.gpu
@pytest.mark.parametrize("prec", [8, 4])
def test_dummy_model_device_resident_mode(tmp_path, oracles, prec):
    """`dummy_model ... dev`: the Fortran host keeps every field on the device (sb_dev_alloc / sb_dev_upload of
    sb_context_mod) and calls seabreeze_diag_dev -- the surface a GPU-resident host model uses."""
    assert _built()
    nx, ny, nz, halo, nsteps = 96, 72, 5, 4, 3
    dt = np.float64 if prec == 8 else np.float32
    fin, fout = tmp_path / "in.bin", tmp_path / "out.bin"
    st, p, steps = _write_input(fin, prec, nx, ny, nz, halo, nsteps)
    r = subprocess.run([EXE[prec], str(fin), str(fout), str(nsteps), "dev"], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    raw = np.fromfile(fout, dtype=dt)
    n2 = nx * ny
    orc = oracles[prec]
    coast = orc.get_edges(st.landfrac, st.icefrac, rule=1, bnd=1)
    cdist = orc.get_dist(coast, st.landfrac, st.lon, st.lat, maxdist=180.0, kwin=halo)
    assert relerr(raw[:n2].reshape(ny, nx), cdist) < (1e-9 if prec == 8 else 2e-6)
    _check_against_oracle(raw[n2:], n2, _oracle_sequence(orc, st, p, steps, raw[:n2].reshape(ny, nx), prec), prec)


@pytest.mark.gpu
@pytest.mark.parametrize("prec", [8, 4])
def test_dummy_model_band_mode_one_rank_communicator(tmp_path, oracles, prec):
    """`dummy_model ... band 0 1 <idfile>`: the Fortran host initialises the library's RCCL communicator
    (sb_comm_get_unique_id / sb_comm_init through sb_context_mod), fills the static fields' ghost cells through
    halo_exchange_mod::swap_bounds -> sb_swap_bounds_* and steps with band_seabreeze_diag.  One rank is what a
    one-GPU box can run: the band is the globe, its ghost rows replicate the pole rows, and the result must equal
    the single-domain oracle.  (Two and three bands: tests/test_bands_gloo.py, tests/test_bands_gpu.py.)"""
    assert _built()
    nx, ny, nz, halo, nsteps = 96, 72, 4, 5, 3
    dt = np.float64 if prec == 8 else np.float32
    fin, fout, idf = tmp_path / "in.bin", tmp_path / "out.bin", tmp_path / "rccl.id"
    st, p, steps = _write_input(fin, prec, nx, ny, nz, halo, nsteps)
    r = subprocess.run([EXE[prec], str(fin), str(fout), str(nsteps), "band", "0", "1", str(idf)],
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    assert idf.exists() and idf.stat().st_size == 128
    # one rank, last step: the ghost fill; k_scan (which publishes the band's moments -- for one rank the gathered set),
    # k_wind on the segment lists of the step before, the strip kernel (contrast, then the update behind its march)
    assert "band step enqueued (launches, RCCL ops, RCCL groups, copies): 4 0 0 0" in r.stdout, r.stdout
    raw = np.fromfile(fout, dtype=dt)
    n2 = nx * ny
    orc = oracles[prec]
    _check_against_oracle(raw[n2:], n2, _oracle_sequence(orc, st, p, steps, raw[:n2].reshape(ny, nx), prec), prec)


@pytest.mark.gpu
@pytest.mark.parametrize("prec", [8, 4])
def test_dummy_model_um_mode(tmp_path, oracles, prec):
    """`dummy_model ... um`: seabreeze_diag_um -- the UM vn10.7 hook's dummy order (theta, z, sigma, mask, ..., error)
    and bounds (small halo for theta/z/sigma, large halo for mask; ref: UM/vn10.7/sea_breeze_diag.F90:55-56,66-117),
    its level walk and the in-place theta <- t0 -- against the oracle's raw-index flavour with the UM level rule."""
    assert _built()
    nx, ny, nz, halo, nsteps = 96, 72, 5, 2, 3
    dt = np.float64 if prec == 8 else np.float32
    fin, fout = tmp_path / "in.bin", tmp_path / "out.bin"
    st, p, steps = _write_input(fin, prec, nx, ny, nz, halo, nsteps)
    r = subprocess.run([EXE[prec], str(fin), str(fout), str(nsteps), "um"], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    raw = np.fromfile(fout, dtype=dt)
    n2 = nx * ny
    cdist = raw[:n2].reshape(ny, nx)
    hs, hl = halo + 1, halo + 3
    nxi, nyi = nx - 2 * hl, ny - 2 * hl
    small = lambda a: np.ascontiguousarray(a[hl - hs:ny - hl + hs, hl - hs:nx - hl + hs])
    core = lambda a: np.ascontiguousarray(a[..., hl:ny - hl, hl:nx - hl])
    orc = oracles[prec]
    so = [np.zeros((nyi, nxi), dt) for _ in range(4)]
    per, ns = nxi * nyi, (nxi + 2 * hs) * (nyi + 2 * hs)
    off = n2
    for t, (th, u, v) in enumerate(steps, start=1):
        orc.seabreeze_diag(1440.0, t, core(p), core(u), core(v), small(th), small(cdist), small(st.z), small(st.sigma), *so,
                           halo=hs, bnd=2, level_rule=1)
        sb, ws, wd, thc = (raw[off + i * per: off + (i + 1) * per].reshape(nyi, nxi) for i in range(4))
        th_back = raw[off + 4 * per: off + 4 * per + ns].reshape(nyi + 2 * hs, nxi + 2 * hs)
        off += 4 * per + ns
        if prec == 8:
            for a, b, nm in ((ws, so[0], "ws"), (wd, so[1], "wd"), (thc, so[2], "thc"), (sb, so[3], "sb_con")):
                assert relerr(a, b, floor=1e-2) < 1e-7, (t, nm)
        else:
            assert relerr(ws, so[0], floor=1e-3) < 2e-6 and np.max(np.abs(thc - so[2])) < 2e-3
        # theta came back as t0: unchanged over the sea (z = 0), lowered over high ground
        zs, ths = small(st.z), small(th)
        assert np.array_equal(th_back[zs == 0], ths[zs == 0])
        assert np.all(th_back[zs > 0] >= ths[zs > 0])          # gmma < 0: t0 = theta - gmma z sigmoid >= theta
    assert off == raw.size
