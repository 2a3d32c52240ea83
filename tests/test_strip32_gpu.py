"""The 96-column marching-strip contrast kernel (k_strip32, sb_strip32_kernel.hip): search radii 17 .. 31 in single
precision -- what BASELINE configs[3] (5120 x 3840, distance window 30) runs.  ref: generic/sea_breeze_diag.f90:188-216.

Yardstick: the reference's arithmetic in DOUBLE precision on the same fp32-representable inputs (the oracle's fp64
build), held by the shared single-precision rule (oracle/fp32_criterion.py): thc within 2e-4 K, winds to fp32 rounding,
sb_con within what those input errors explain in each cell, no unexplained trigger flip.  The tile kernel
(sb_set_wide_strip(ctx, 0)) runs beside it on the same inputs: the two must agree far inside that tolerance."""
import numpy as np
import pytest

from conftest import relerr
from oracle import fp32_criterion as crit
from seabreeze_param_amd import hip, synth

pytestmark = pytest.mark.gpu

f8 = lambda a: np.ascontiguousarray(a, dtype=np.float64)


def _case(orc8, nx, ny, kwin, dt=np.float32):
    st = synth.static_fields(nx, ny, dt)
    coast = orc8.get_edges(f8(st.landfrac), f8(st.icefrac))
    cd = orc8.get_dist(coast, f8(st.landfrac), st.lon, st.lat, maxdist=20000.0, kwin=kwin)
    # every cell the window reached is "in the band" (the host-model flavour's maxdist is fixed at 180 km)
    cd = np.where(np.abs(cd) < 12000.0, np.sign(cd) * np.minimum(np.abs(cd), 179.0), cd).astype(dt)
    return st, cd


def _run_generic(hipctx, orc8, st, cd, nz, steps, bnd=hip.SB_BND_GLOBAL, obnd=1):
    ny, nx = cd.shape
    dt = np.float32
    p = synth.pressure_3d(st, nz, dt)
    sh = [np.zeros((ny, nx), dt) for _ in range(4)]
    so = [np.zeros((ny, nx), np.float64) for _ in range(4)]
    band = np.abs(f8(cd)) <= 180.0
    per = []
    for tn in steps:
        th = synth.theta_step(st, tn, dt)
        u, v = synth.wind_step(st, nz, tn, dt)
        gp, op = [a.copy() for a in sh], [a.copy() for a in so]
        orc8.seabreeze_diag(7200.0, tn, f8(p), f8(u), f8(v), f8(th), f8(cd), f8(st.z), f8(st.sigma), *so, halo=0, bnd=obnd)
        hipctx.seabreeze_diag(7200.0, tn, p, u, v, th, cd, st.z, st.sigma, *sh, halo=0, bnd=bnd)
        per.append(crit.check_step(tn, gp, sh, op, so, band, timestep=7200.0))
    return crit.merge(per), sh, so


@pytest.fixture
def wide(hipctx):
    hipctx.set_search_radius_hint(30)
    yield hipctx
    hipctx.set_search_radius_hint(16)
    hipctx.set_wide_strip(True)
    hipctx.set_plan_cache(True)
    hipctx.set_fold(True)
    hipctx.set_workgroups(0)


@pytest.mark.parametrize("shape", [(256, 192), (250, 190), (96, 80)], ids=["256x192", "ragged", "small"])
@pytest.mark.parametrize("keep,fold", [(True, True), (False, True), (True, False)], ids=["stored-plan", "replan", "kprep"])
def test_radii_up_to_31_from_lds(wide, oracles, shape, keep, fold):
    nx, ny = shape
    orc8 = oracles[8]
    st, cd = _case(orc8, nx, ny, kwin=27 if nx > 100 else 20)
    wide.set_plan_cache(keep)
    wide.set_fold(fold)
    res, sh, so = _run_generic(wide, orc8, st, cd, 3, (1, 2, 3, 15))
    c = wide.last_counters()
    assert res["ok"], res
    assert (nx < 100 or orc8.last_nn_max > 16) and orc8.last_nn_max <= 31, orc8.last_nn_max
    assert c["global_path_cells"] == 0 and c["max_radius"] == orc8.last_nn_max, (c, orc8.last_nn_max)
    # the tile kernel on the same inputs: the two kernels' window means differ by rounding only
    wide.set_wide_strip(False)
    res_t, sh_t, _ = _run_generic(wide, orc8, st, cd, 3, (1, 2, 3, 15))
    wide.set_wide_strip(True)
    assert res_t["ok"], res_t
    assert np.max(np.abs(sh[2].astype(np.float64) - sh_t[2])) < 2e-5 and relerr(sh[0], sh_t[0], floor=1e-3) < 1e-6


def test_f2py_boundary_rule_and_last_longitude(wide, oracles):
    """The f2py window rule (lon 0 and nlons both map to column 1; the centre of the window at the last longitude reads
    column 1) under the wide kernel: the host-model signature with SB_BND_WRAPPER against the oracle's rule 0."""
    orc8 = oracles[8]
    st, cd = _case(orc8, 256, 192, kwin=24)
    res, _, _ = _run_generic(wide, orc8, st, cd, 2, (1, 2), bnd=hip.SB_BND_WRAPPER, obnd=0)
    assert res["ok"], res


def test_wrapper_flavour_reads_the_t0_plane(wide, oracles):
    """The f2py flavour: t0 is an output plane (k_t0 writes it, the strip kernel stages from it), 1-D p."""
    orc8 = oracles[8]
    nx, ny, nz = 256, 192, 3
    dt = np.float32
    st, cd = _case(orc8, nx, ny, kwin=25)
    p = synth.pressure_1d(nz, dt)
    wh = [np.zeros((ny, nx), dt) for _ in range(3)]
    wo = [np.zeros((ny, nx), np.float64) for _ in range(3)]
    for tn in (1, 2):
        th = synth.theta_step(st, tn, dt)
        u, v = synth.wind_step(st, nz, tn, dt)
        oo = orc8.diag(tn, f8(p), f8(st.z), f8(st.sigma), f8(th), f8(v), f8(u), f8(cd), *wo)
        oh = wide.diag(tn, p, st.z, st.sigma, th, v, u, cd, *wh)
        band = (np.abs(f8(cd)) <= 180.0)[:-1]
        assert relerr(oh[1, :-1], oo[1, :-1], floor=1.0) < 2e-6                      # t0
        assert relerr(oh[2, :-1], oo[2, :-1], floor=1e-3) < 2e-6 and relerr(oh[3, :-1], oo[3, :-1], floor=1e-1) < 2e-5
        assert np.max(np.abs(wh[2][:-1][band].astype(np.float64) - wo[2][:-1][band])) < 2e-4, tn      # thc
        near = np.abs(np.abs(wo[2][:-1]) - 0.75) < 2e-3
        d = np.abs(oh[0, :-1].astype(np.float64) - oo[0, :-1])
        assert np.max(d[band & ~near]) < 2e-3, tn
    assert wide.last_counters()["global_path_cells"] == 0


def test_ghost_celled_band_frame(wide, oracles):
    """SB_BND_HALO (a latitude band of a multi-GPU run, the UM layout): windows stop at the frame's edge."""
    orc8 = oracles[8]
    nx, ny, nz, h = 192, 96, 2, 30
    dt = np.float32
    st = synth.static_fields(nx + 2 * h, ny + 2 * h, dt)
    coast = orc8.get_edges(f8(st.landfrac), f8(st.icefrac))
    cd = orc8.get_dist(coast, f8(st.landfrac), st.lon, st.lat, maxdist=20000.0, kwin=h - 1)
    cd = np.where(np.abs(cd) < 12000.0, np.sign(cd) * np.minimum(np.abs(cd), 179.0), cd).astype(dt)
    core = (slice(h, h + ny), slice(h, h + nx))
    p = synth.pressure_3d(st, nz, dt)[:, core[0], core[1]].copy()
    sh = [np.zeros((ny, nx), dt) for _ in range(4)]
    so = [np.zeros((ny, nx), np.float64) for _ in range(4)]
    band = np.abs(f8(cd[core])) <= 180.0
    per = []
    for tn in (1, 2, 3):
        th = synth.theta_step(st, tn, dt)
        u, v = (a[:, core[0], core[1]].copy() for a in synth.wind_step(st, nz, tn, dt))
        gp, op = [a.copy() for a in sh], [a.copy() for a in so]
        orc8.seabreeze_diag(7200.0, tn, f8(p), f8(u), f8(v), f8(th), f8(cd), f8(st.z), f8(st.sigma), *so, halo=h, bnd=2)
        wide.seabreeze_diag(7200.0, tn, p, u, v, th, cd, st.z, st.sigma, *sh, halo=h, bnd=hip.SB_BND_HALO)
        per.append(crit.check_step(tn, gp, sh, op, so, band, timestep=7200.0))
    res = crit.merge(per)
    assert res["ok"], res


def test_radii_beyond_31_take_the_global_path(wide, oracles):
    """Windows that outgrow the tables (a distance field with its signs flipped: beyond the band every cell counts as land
    side, so sea-side cells deep inside it search far): marked during the march, searched in global memory behind it --
    in the planning call and by the stored plan."""
    orc8 = oracles[8]
    nx, ny = 256, 192
    st = synth.static_fields(nx, ny, np.float32)
    coast = orc8.get_edges(f8(st.landfrac), f8(st.icefrac))
    base = orc8.get_dist(coast, f8(st.landfrac), st.lon, st.lat, maxdist=700.0)
    cd = np.where(np.abs(base) < 12000.0, -base, base)
    cd[np.abs(cd) > 180.0] = 12000.0
    res, _, _ = _run_generic(wide, orc8, st, cd.astype(np.float32), 2, (1, 2, 3))
    c = wide.last_counters()
    assert orc8.last_nn_max > 31 and c["global_path_cells"] > 0 and c["max_radius"] == orc8.last_nn_max, (c, orc8.last_nn_max)
    assert res["ok"], res


@pytest.mark.parametrize("nwg", [1, 3, 7])
def test_few_workgroups_march_in_rounds(wide, oracles, nwg):
    """Shares of several rounds (planned every call, not stored) and more query steps than a stored plan holds."""
    orc8 = oracles[8]
    st, cd = _case(orc8, 512, 384, kwin=27)
    wide.set_workgroups(nwg)
    res, _, _ = _run_generic(wide, orc8, st, cd, 2, (1, 2, 3))
    assert res["ok"], res
    assert wide.last_counters()["global_path_cells"] == 0


def test_double_precision_keeps_the_tile_kernel(wide, oracles):
    """Radii beyond 16 in fp64: the tile kernel (the packed tables hold 48-bit sums), to 1e-7 as ever."""
    orc8 = oracles[8]
    nx, ny, nz = 256, 192, 2
    st, cd = _case(orc8, nx, ny, kwin=27, dt=np.float64)
    p = synth.pressure_3d(st, nz, np.float64)
    sh = [np.zeros((ny, nx)) for _ in range(4)]
    so = [np.zeros((ny, nx)) for _ in range(4)]
    for tn in (1, 2):
        th = synth.theta_step(st, tn, np.float64)
        u, v = synth.wind_step(st, nz, tn, np.float64)
        orc8.seabreeze_diag(7200.0, tn, p, u, v, th, cd, st.z, st.sigma, *so, halo=0, bnd=1)
        wide.seabreeze_diag(7200.0, tn, p, u, v, th, cd, st.z, st.sigma, *sh, halo=0, bnd=hip.SB_BND_GLOBAL)
    for a, b, nm in zip(sh, so, ("ws", "wd", "thc", "sb_con")):
        assert relerr(a, b, floor=1e-2) < 1e-7, nm


@pytest.mark.parametrize("static_sigma", [False, True], ids=["default", "static-sigma"])
def test_band_step_owning_the_globe(oracles, static_sigma):
    """sb_band_seabreeze_diag_f32_dev with a ghost frame of 30 cells: the 96-column kernel merges the gathered moments,
    takes theta's east-west ghost columns and the rows beyond the poles by index arithmetic and applies the update behind
    its march; several steps against the single-domain oracle under the shared single-precision rule."""
    import torch
    orc8 = oracles[8]
    nx, ny, nz, h = 256, 192, 2, 30
    dt = np.float32
    st, cd = _case(orc8, nx, ny, kwin=27)
    p = synth.pressure_3d(st, nz, dt)
    ctx = hip.Context(0)
    try:
        ctx.set_search_radius_hint(h)
        ctx.set_static_sigma(static_sigma)
        stream = torch.cuda.current_stream().cuda_stream

        def frame(a):
            f = torch.zeros((ny + 2 * h, nx + 2 * h), dtype=torch.float32, device="cuda")
            f[h:h + ny, h:h + nx] = torch.from_numpy(np.ascontiguousarray(a, dtype=dt)).cuda()
            torch.cuda.synchronize()
            return f

        def static(a):
            f = frame(a)
            ctx.swap_bounds_dev(dt, f.data_ptr(), nx, ny, h, stream)
            ctx.synchronize()
            return f

        z, sg, mk = static(st.z), static(st.sigma), static(cd)
        pd = torch.from_numpy(p).cuda()
        state = [torch.zeros((ny, nx), dtype=torch.float32, device="cuda") for _ in range(4)]
        so = [np.zeros((ny, nx), np.float64) for _ in range(4)]
        band = np.abs(f8(cd)) <= 180.0
        per, launches = [], []
        for tn in (1, 2, 3, 4):
            th = synth.theta_step(st, tn, dt)
            u, v = synth.wind_step(st, nz, tn, dt)
            thf, ud, vd = frame(th), torch.from_numpy(u).cuda(), torch.from_numpy(v).cuda()
            gp, op = [s.cpu().numpy() for s in state], [a.copy() for a in so]
            torch.cuda.synchronize()
            ctx.band_seabreeze_diag_dev(dt, 7200.0, tn, nx, ny, nz, h, pd.data_ptr(), ud.data_ptr(), vd.data_ptr(),
                                        thf.data_ptr(), mk.data_ptr(), z.data_ptr(), sg.data_ptr(),
                                        *[s.data_ptr() for s in state], stream)
            ctx.synchronize()
            launches.append(ctx.last_step_report()["kernel_launches"])
            orc8.seabreeze_diag(7200.0, tn, f8(p), f8(u), f8(v), f8(th), f8(cd), f8(st.z), f8(st.sigma), *so, halo=0, bnd=1)
            per.append(crit.check_step(tn, gp, [s.cpu().numpy() for s in state], op, so, band, timestep=7200.0))
        res = crit.merge(per)
        assert res["ok"], res
        assert orc8.last_nn_max > 16 and ctx.last_counters()["global_path_cells"] == 0
        assert launches == [4, 3, 3, 3], launches            # k_prep in the first step only, no ghost-fill kernel
    finally:
        ctx.close()


def test_stored_plan_follows_the_planes(wide, oracles):
    """The 96-column kernel marches by the plan an earlier call stored while k_scan finds both planes unchanged: calls
    that repeat the coast distance, shift its band, flip its sign (same band, other classes, radii beyond 31 among
    them) and come back all match the oracle, call by call -- with the winds and the state carried through."""
    orc8 = oracles[8]
    nx, ny, nz = 256, 192, 2
    dt = np.float32
    st, base = _case(orc8, nx, ny, kwin=27)
    coast = orc8.get_edges(f8(st.landfrac), f8(st.icefrac))
    far = orc8.get_dist(coast, f8(st.landfrac), st.lon, st.lat, maxdist=700.0)
    flipped = np.where(np.abs(far) < 12000.0, -far, far)       # (as in test_radii_beyond_31_take_the_global_path)
    flipped[np.abs(flipped) > 180.0] = 12000.0
    flipped = flipped.astype(dt)
    rolled = np.roll(base, 7, axis=1)
    p = synth.pressure_3d(st, nz, dt)
    sh = [np.zeros((ny, nx), dt) for _ in range(4)]
    so = [np.zeros((ny, nx), np.float64) for _ in range(4)]
    marked = []
    for tn, cd in enumerate((base, base, rolled, rolled, flipped, flipped, base, base), start=1):
        th = synth.theta_step(st, tn, dt)
        u, v = synth.wind_step(st, nz, tn, dt)
        gp, op = [a.copy() for a in sh], [a.copy() for a in so]
        orc8.seabreeze_diag(7200.0, tn, f8(p), f8(u), f8(v), f8(th), f8(cd), f8(st.z), f8(st.sigma), *so, halo=0, bnd=1)
        wide.seabreeze_diag(7200.0, tn, p, u, v, th, cd, st.z, st.sigma, *sh, halo=0, bnd=hip.SB_BND_GLOBAL)
        res = crit.merge([crit.check_step(tn, gp, sh, op, so, np.abs(f8(cd)) <= 180.0, timestep=7200.0)])
        assert res["ok"], (tn, res)
        c = wide.last_counters()
        assert c["max_radius"] == orc8.last_nn_max, (tn, c, orc8.last_nn_max)
        marked.append(c["global_path_cells"])
    assert marked[0] == marked[1] == marked[2] == marked[3] == marked[6] == marked[7] == 0, marked
    assert marked[4] > 0 and marked[4] == marked[5], marked       # the stored plan names the same marked cells
