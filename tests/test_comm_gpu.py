"""The library's own RCCL path (sb_comm_*, sb_swap_bounds_*_dev, sb_allgather_moments_dev).

This pool gives one GPU per box and RCCL refuses two ranks on one device, so what can be
exercised here is a ONE-rank communicator: librccl.so is opened on demand, the communicator
is created, and swap_bounds degenerates to its local part (both poles replicate, longitude
wraps) -- checked against numpy.  The two-neighbour send/recv offsets are the same ones the
torch path uses, which tests/test_bands_gpu.py and tests/test_bands_gloo.py do check.
"""
import numpy as np
import pytest
import torch

from seabreeze_param_amd import hip

pytestmark = pytest.mark.gpu


def _expected(core, h):
    ny, nx = core.shape
    rows = np.clip(np.arange(-h, ny + h), 0, ny - 1)
    cols = np.arange(-h, nx + h) % nx
    return core[np.ix_(rows, cols)]


@pytest.mark.parametrize("dtype", [np.float64, np.float32])
@pytest.mark.parametrize("with_comm", [False, True])
def test_swap_bounds_single_rank(dtype, with_comm):
    ctx = hip.Context(0)
    try:
        if with_comm:
            ctx.comm_init(hip.comm_unique_id(), 0, 1)
        nx, ny, h = 150, 37, 5
        rng = np.random.default_rng(3)
        core = rng.normal(size=(ny, nx)).astype(dtype)
        frame = np.full((ny + 2 * h, nx + 2 * h), np.nan, dtype=dtype)
        frame[h:h + ny, h:h + nx] = core
        t = torch.from_numpy(frame).cuda()
        ctx.swap_bounds_dev(dtype, t.data_ptr(), nx, ny, h, torch.cuda.current_stream().cuda_stream)
        torch.cuda.synchronize()
        assert np.array_equal(t.cpu().numpy(), _expected(core, h))
        mine = torch.arange(5, dtype=torch.float64, device="cuda")
        gath = torch.zeros(5, dtype=torch.float64, device="cuda")
        ctx.allgather_moments_dev(mine.data_ptr(), gath.data_ptr(), torch.cuda.current_stream().cuda_stream)
        torch.cuda.synchronize()
        assert torch.equal(mine, gath)
        if with_comm:
            ctx.comm_finalize()
    finally:
        ctx.close()


@pytest.mark.parametrize("contrast_first", [False, True], ids=["wind-first", "contrast-first"])
@pytest.mark.parametrize("static_sigma", [False, True], ids=["default", "static-sigma"])
@pytest.mark.parametrize("with_comm", [False, True])
def test_band_step_single_rank_equals_global(oracles, with_comm, static_sigma, contrast_first):
    """sb_band_seabreeze_diag_*_dev with one band owning the globe: the ghost frame it fills itself
    (poles replicate, longitude wraps) must give the single-domain SB_BND_GLOBAL result, and the
    fork/join with the communication stream must order correctly over several steps."""
    from seabreeze_param_amd import synth
    nx, ny, nz, h = 200, 96, 3, 6
    dt, orc = np.float64, oracles[8]
    st = synth.static_fields(nx, ny, dt)
    coast = orc.get_edges(st.landfrac, st.icefrac)
    cdist = orc.get_dist(coast, st.landfrac, st.lon, st.lat, maxdist=900.0, kwin=h - 1)
    cdist[np.abs(cdist) > 180.0] = 12000.0
    p = synth.pressure_3d(st, nz, dt)
    ctx = hip.Context(0)
    try:
        if with_comm:
            ctx.comm_init(hip.comm_unique_id(), 0, 1)
        ctx.set_search_radius_hint(h)
        ctx.set_static_sigma(static_sigma)
        ctx.set_band_order(contrast_first)
        stream = torch.cuda.current_stream().cuda_stream
        reports = []

        def frame(a):
            f = torch.zeros((ny + 2 * h, nx + 2 * h), dtype=torch.float64, device="cuda")
            f[h:h + ny, h:h + nx] = torch.from_numpy(np.ascontiguousarray(a)).cuda()
            return f

        z, sg, mk = frame(st.z), frame(st.sigma), frame(cdist)
        for f in (z, sg, mk):
            ctx.swap_bounds_dev(dt, f.data_ptr(), nx, ny, h, stream)          # static fields: once
        pd = torch.from_numpy(p).cuda()
        state = [torch.zeros((ny, nx), dtype=torch.float64, device="cuda") for _ in range(4)]
        ref = [np.zeros((ny, nx)) for _ in range(4)]
        for tn in (1, 2, 3, 4):
            th = synth.theta_step(st, tn, dt)
            u, v = synth.wind_step(st, nz, tn, dt)
            thf, ud, vd = frame(th), torch.from_numpy(u).cuda(), torch.from_numpy(v).cuda()
            ctx.band_seabreeze_diag_dev(dt, 5400.0, tn, nx, ny, nz, h, pd.data_ptr(), ud.data_ptr(), vd.data_ptr(),
                                        thf.data_ptr(), mk.data_ptr(), z.data_ptr(), sg.data_ptr(),
                                        *[s.data_ptr() for s in state], stream)
            torch.cuda.synchronize()
            reports.append(ctx.last_step_report())
            orc.seabreeze_diag(5400.0, tn, p, u, v, th, cdist, st.z, st.sigma, *ref, halo=0, bnd=1)
            for nm, a, b in zip(("ws", "wd", "thc", "sb_con"), state, ref):
                err = np.max(np.abs(a.cpu().numpy() - b) / np.maximum(np.abs(b), 1e-2))
                assert err < 1e-7, f"step {tn} {nm}: {err}"
        # what a band step enqueues (one rank: its moments are the gathered set, no neighbours to send to):
        # nothing on the communication stream (theta's east-west ghost columns and the rows beyond the poles are index arithmetic in
        # the strip kernel since round 4: no fill kernel); on the caller's k_scan (whose last workgroup publishes the band's
        # moments), k_prep, k_wind, the strip kernel (which merges the gathered moments itself and applies the update); with
        # static sigma the statistics go.  With the contrast kernel ahead of k_wind (sb_set_band_order) k_prep goes too.
        n0 = 3 if contrast_first else 4
        assert reports[0] == dict(kernel_launches=n0, rccl_ops=0, rccl_groups=0, d2d_copies=0)
        n1 = n0 if contrast_first else n0 - 1            # (from the second step on the segment lists of the step before stand: no k_prep)
        later = dict(reports[0], kernel_launches=n1)     # (static sigma: k_scan skips the statistics -- the same launches)
        assert all(r == later for r in reports[1:]), reports
    finally:
        ctx.close()


def test_band_step_with_one_workgroup(oracles):
    """A band step whose strip kernel marches in several rounds (one persistent workgroup, sb_set_workgroups): no plan is
    stored for such a share, so the update behind the march walks the workgroup's blocks instead of its cell lists."""
    from seabreeze_param_amd import synth
    nx, ny, nz, h = 1024, 384, 2, 6
    dt, orc = np.float64, oracles[8]
    st = synth.static_fields(nx, ny, dt)
    coast = orc.get_edges(st.landfrac, st.icefrac)
    cdist = orc.get_dist(coast, st.landfrac, st.lon, st.lat, maxdist=900.0, kwin=h - 1)
    cdist[np.abs(cdist) > 180.0] = 12000.0
    p = synth.pressure_3d(st, nz, dt)
    ctx = hip.Context(0)
    try:
        ctx.set_search_radius_hint(h)
        ctx.set_workgroups(1)
        stream = torch.cuda.current_stream().cuda_stream

        def frame(a):
            f = torch.zeros((ny + 2 * h, nx + 2 * h), dtype=torch.float64, device="cuda")
            f[h:h + ny, h:h + nx] = torch.from_numpy(np.ascontiguousarray(a)).cuda()
            torch.cuda.synchronize()
            ctx.swap_bounds_dev(dt, f.data_ptr(), nx, ny, h, stream)
            ctx.synchronize()
            return f

        z, sg, mk = frame(st.z), frame(st.sigma), frame(cdist)
        pd = torch.from_numpy(p).cuda()
        state = [torch.zeros((ny, nx), dtype=torch.float64, device="cuda") for _ in range(4)]
        ref = [np.zeros((ny, nx)) for _ in range(4)]
        for tn in (1, 2, 3):
            th = synth.theta_step(st, tn, dt)
            u, v = synth.wind_step(st, nz, tn, dt)
            thf, ud, vd = frame(th), torch.from_numpy(u).cuda(), torch.from_numpy(v).cuda()
            torch.cuda.synchronize()
            ctx.band_seabreeze_diag_dev(dt, 5400.0, tn, nx, ny, nz, h, pd.data_ptr(), ud.data_ptr(), vd.data_ptr(),
                                        thf.data_ptr(), mk.data_ptr(), z.data_ptr(), sg.data_ptr(),
                                        *[s.data_ptr() for s in state], stream)
            ctx.synchronize()
            orc.seabreeze_diag(5400.0, tn, p, u, v, th, cdist, st.z, st.sigma, *ref, halo=0, bnd=1)
            for nm, a, b in zip(("ws", "wd", "thc", "sb_con"), state, ref):
                err = np.max(np.abs(a.cpu().numpy() - b) / np.maximum(np.abs(b), 1e-2))
                assert err < 1e-7, f"step {tn} {nm}: {err}"
    finally:
        ctx.close()


def test_band_step_follows_a_changing_coast(oracles):
    """From its second step on a band step runs no k_prep: k_wind takes the segment lists the strip kernel of the step
    before compacted, unless k_scan finds the planes changed in this very step -- then its waves walk the plane itself.
    Steps that repeat the coast distance, shift it and come back all match the single-domain oracle."""
    from seabreeze_param_amd import synth
    nx, ny, nz, h = 200, 96, 3, 6
    dt, orc = np.float64, oracles[8]
    st = synth.static_fields(nx, ny, dt)
    coast = orc.get_edges(st.landfrac, st.icefrac)
    base = orc.get_dist(coast, st.landfrac, st.lon, st.lat, maxdist=900.0, kwin=h - 1)
    base[np.abs(base) > 180.0] = 12000.0
    p = synth.pressure_3d(st, nz, dt)
    ctx = hip.Context(0)
    try:
        ctx.set_search_radius_hint(h)
        stream = torch.cuda.current_stream().cuda_stream

        def frame(a):
            # (torch's current stream is the null stream here, for which the library substitutes a stream of its own --
            # and torch brings its own copy of the HIP runtime: nothing orders the two, so every hand-over between
            # torch and the library is a synchronisation on the side that wrote last)
            f = torch.zeros((ny + 2 * h, nx + 2 * h), dtype=torch.float64, device="cuda")
            f[h:h + ny, h:h + nx] = torch.from_numpy(np.ascontiguousarray(a)).cuda()
            torch.cuda.synchronize()
            ctx.swap_bounds_dev(dt, f.data_ptr(), nx, ny, h, stream)
            ctx.synchronize()
            return f

        z, sg = frame(st.z), frame(st.sigma)
        mk = frame(base)                                   # one device array, rewritten in place: the planes' buffers stay
        pd = torch.from_numpy(p).cuda()
        state = [torch.zeros((ny, nx), dtype=torch.float64, device="cuda") for _ in range(4)]
        ref = [np.zeros((ny, nx)) for _ in range(4)]
        shifted = np.roll(base, 9, axis=1)
        launches = []
        for tn, cd in enumerate((base, base, shifted, shifted, base, base), start=1):
            mk.copy_(frame(cd))
            th = synth.theta_step(st, tn, dt)
            u, v = synth.wind_step(st, nz, tn, dt)
            thf, ud, vd = frame(th), torch.from_numpy(u).cuda(), torch.from_numpy(v).cuda()
            torch.cuda.synchronize()
            ctx.band_seabreeze_diag_dev(dt, 5400.0, tn, nx, ny, nz, h, pd.data_ptr(), ud.data_ptr(), vd.data_ptr(),
                                        thf.data_ptr(), mk.data_ptr(), z.data_ptr(), sg.data_ptr(),
                                        *[s.data_ptr() for s in state], stream)
            ctx.synchronize()
            launches.append(ctx.last_step_report()["kernel_launches"])
            orc.seabreeze_diag(5400.0, tn, p, u, v, th, cd, st.z, st.sigma, *ref, halo=0, bnd=1)
            for nm, a, b in zip(("ws", "wd", "thc", "sb_con"), state, ref):
                err = np.max(np.abs(a.cpu().numpy() - b) / np.maximum(np.abs(b), 1e-2))
                assert err < 1e-7, f"step {tn} {nm}: {err}"
        assert launches == [4, 3, 3, 3, 3, 3], launches     # k_prep in the first step only
    finally:
        ctx.close()


@pytest.mark.parametrize("prec", [8, 4])
def test_fill_ghosts_of_an_interior_band(prec):
    """The local ghost fill as an interior rank of a multi-rank run sees it (south = north = 0: the ghost rows came
    over the wire and must be kept, only their east-west ghost columns are wrapped) and as the two pole ranks see it,
    against the torch path of BandRunner (bands.fill_ew_ghosts / the pole rule of bands.exchange_ns).  Everything of
    a multi-rank exchange but the RCCL wire itself, which this one-GPU box cannot carry."""
    from seabreeze_param_amd import bands
    dt = np.float64 if prec == 8 else np.float32
    nx, nyl, h = 200, 40, 6
    rng = np.random.default_rng(11)
    ctx = hip.Context(0)
    try:
        for south, north in ((0, 0), (1, 0), (0, 1), (1, 1)):
            host = rng.standard_normal((nyl + 2 * h, nx + 2 * h)).astype(dt)     # ghost rows: "what the neighbours sent"
            want = host.copy()
            if south:
                want[0:h] = want[h:h + 1]
            if north:
                want[nyl + h:] = want[nyl + h - 1:nyl + h]
            bands.fill_ew_ghosts(want, nx, h)
            dev = torch.from_numpy(host).cuda()
            torch.cuda.synchronize()
            ctx.fill_ghosts_dev(dt, dev.data_ptr(), nx, nyl, h, south, north)
            ctx.synchronize()
            assert np.array_equal(dev.cpu().numpy(), want), (south, north)
    finally:
        ctx.close()


def test_static_sigma_does_not_outlive_other_statistics(oracles):
    """sb_set_static_sigma keeps the sigmoid scalars of the first call -- but they share a scratch buffer with the
    stand-alone sigmoid: diag, sb_sigmoid_*_dev on another field, diag again must form sigma's statistics anew
    (ADVICE round 2).  Every diag matches the oracle, which recomputes them every call (ref: generic/...f90:457-481)."""
    from seabreeze_param_amd import synth
    nx, ny, nz = 256, 192, 3
    dt, orc = np.float64, oracles[8]
    st = synth.static_fields(nx, ny, dt)
    coast = orc.get_edges(st.landfrac, st.icefrac, rule=1, bnd=1)
    cdist = orc.get_dist(coast, st.landfrac, st.lon, st.lat, maxdist=180.0, kwin=5)
    p = synth.pressure_3d(st, nz, dt)
    dev = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
    ctx = hip.Context(0)
    try:
        ctx.set_static_sigma(True)
        z, mk, pd, sg = dev(st.z), dev(cdist), dev(p), dev(st.sigma)
        other = dev(st.z * 3.0 + 11.0)
        sm = torch.empty_like(other)
        state = [torch.zeros((ny, nx), dtype=torch.float64, device="cuda") for _ in range(4)]
        ref = [np.zeros((ny, nx)) for _ in range(4)]
        for tn in (1, 2, 3, 4):
            th = synth.theta_step(st, tn, dt)
            u, v = synth.wind_step(st, nz, tn, dt)
            ud, vd, thd = dev(u), dev(v), dev(th)
            torch.cuda.synchronize()
            if tn == 3:                                  # another field's scalars land in the shared scratch
                ctx.sigmoid_dev(dt, nx, ny, other.data_ptr(), sm.data_ptr())
            ctx.seabreeze_diag_dev(dt, 7200.0, tn, nx, ny, nz, 0, hip.SB_BND_GLOBAL, pd.data_ptr(), ud.data_ptr(), vd.data_ptr(),
                                   thd.data_ptr(), mk.data_ptr(), z.data_ptr(), sg.data_ptr(), *[s.data_ptr() for s in state])
            ctx.synchronize()
            orc.seabreeze_diag(7200.0, tn, p, u, v, th, cdist, st.z, st.sigma, *ref, halo=0, bnd=1)
            for nm, a, b in zip(("ws", "wd", "thc", "sb_con"), state, ref):
                err = np.max(np.abs(a.cpu().numpy() - b) / np.maximum(np.abs(b), 1e-2))
                assert err < 1e-7, f"step {tn} {nm}: {err}"
    finally:
        ctx.close()


@pytest.mark.parametrize("fold", [True, False], ids=["fold", "kprep"])
def test_static_sigma_single_domain(oracles, fold):
    """sb_set_static_sigma on a single domain: the statistics of the first call stand while the same device
    array comes back, another array forms them anew, and results equal the default path's bit for bit."""
    from seabreeze_param_amd import synth
    nx, ny, nz = 256, 192, 3
    dt, orc = np.float64, oracles[8]
    st = synth.static_fields(nx, ny, dt)
    coast = orc.get_edges(st.landfrac, st.icefrac, rule=1, bnd=1)
    cdist = orc.get_dist(coast, st.landfrac, st.lon, st.lat, maxdist=180.0, kwin=5)
    p = synth.pressure_3d(st, nz, dt)
    dev = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
    ctx = hip.Context(0)
    try:
        stream = torch.cuda.current_stream().cuda_stream
        ctx.set_fold(fold)
        z, mk, pd = dev(st.z), dev(cdist), dev(p)
        sg_a, sg_b = dev(st.sigma), dev(st.sigma * 1.7 + 3.0)
        runs = {}
        for static in (False, True):
            ctx.set_static_sigma(static)
            state = [torch.zeros((ny, nx), dtype=torch.float64, device="cuda") for _ in range(4)]
            outs = []
            for tn, sg in ((1, sg_a), (2, sg_a), (3, sg_b), (4, sg_b), (5, sg_a)):
                th = synth.theta_step(st, tn, dt)
                u, v = synth.wind_step(st, nz, tn, dt)
                ud, vd, thd = dev(u), dev(v), dev(th)
                ctx.seabreeze_diag_dev(dt, 7200.0, tn, nx, ny, nz, 0, hip.SB_BND_GLOBAL, pd.data_ptr(), ud.data_ptr(),
                                       vd.data_ptr(), thd.data_ptr(), mk.data_ptr(), z.data_ptr(), sg.data_ptr(),
                                       *[s.data_ptr() for s in state], stream)
                torch.cuda.synchronize()
                assert ctx.last_step_report()["kernel_launches"] == (3 if fold else 4)     # k_scan, [k_prep,] k_thc3, k_wind
                outs.append([s.cpu().numpy().copy() for s in state])
            runs[static] = outs
        for a, b in zip(runs[False], runs[True]):
            for x, y in zip(a, b):
                assert np.array_equal(x, y, equal_nan=True)
        # and the default path is the oracle's
        ref = [np.zeros((ny, nx)) for _ in range(4)]
        for i, (tn, sg) in enumerate(((1, st.sigma), (2, st.sigma), (3, st.sigma * 1.7 + 3.0))):
            th = synth.theta_step(st, tn, dt)
            u, v = synth.wind_step(st, nz, tn, dt)
            orc.seabreeze_diag(7200.0, tn, p, u, v, th, cdist, st.z, sg, *ref, halo=0, bnd=1)
            for x, y in zip(runs[True][i], ref):
                assert np.max(np.abs(x - y) / np.maximum(np.abs(y), 1e-2)) < 1e-7
    finally:
        ctx.close()


def test_comm_argument_errors():
    ctx = hip.Context(0)
    try:
        with pytest.raises(hip.SeabreezeHipError):
            ctx.comm_init(b"\0" * 128, 3, 2)           # rank out of range
        t = torch.zeros((10, 10), dtype=torch.float64, device="cuda")
        with pytest.raises(hip.SeabreezeHipError):
            ctx.swap_bounds_dev(np.float64, t.data_ptr(), 0, 4, 1)
    finally:
        ctx.close()
