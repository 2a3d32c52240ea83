"""The multi-GPU path on ONE GPU: two (and three) processes share cuda:0 and exchange over gloo,
so everything except the RCCL transport itself is the code the 8-GPU bench runs -- the
BandRunner, the gathered-moments merge kernel, the ghost-cell (SB_BND_HALO) kernels.
Each band must reproduce the rows of the single-domain CPU oracle."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import ROOT

pytestmark = pytest.mark.gpu


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, nx, ny, nz, tmpdir, static_sigma=False):
    sys.path.insert(0, ROOT)
    from oracle.pyoracle import Oracle
    from seabreeze_param_amd import hip, synth
    from seabreeze_param_amd.bands import BandRunner, split_rows
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        orc = Oracle(8)
        st = synth.static_fields(nx, ny)
        coast = orc.get_edges(st.landfrac, st.icefrac)
        kwin = 5
        cdist = orc.get_dist(coast, st.landfrac, st.lon, st.lat, maxdist=900.0, kwin=kwin)
        cdist[np.abs(cdist) > 180.0] = 12000.0
        p = synth.pressure_3d(st, nz)
        ctx = hip.Context(0)
        runner = BandRunner(ctx, torch, dist, rank, world, nx, ny, nz, halo=kwin + 1, static_sigma=static_sigma)
        runner.upload_static(st.z, st.sigma, cdist)
        full = [np.zeros((ny, nx)) for _ in range(4)]
        r0, r1 = split_rows(ny, world)[rank]
        for tn in (1, 2, 3):
            th = synth.theta_step(st, tn)
            u, v = synth.wind_step(st, nz, tn)
            s = runner.upload_step_inputs(p, u, v, th)
            runner.step(7200.0, tn, s)
            ctx.synchronize()
            torch.cuda.synchronize()
            orc.seabreeze_diag(7200.0, tn, p, u, v, th, cdist, st.z, st.sigma, *full, halo=0, bnd=1)
            for nm, mine, ref in zip(("ws", "wd", "thc", "sb_con"),
                                     (runner.ws, runner.wd, runner.thc, runner.sb_con), full):
                a, b = mine.cpu().numpy(), ref[r0:r1]
                err = np.max(np.abs(a - b) / np.maximum(np.abs(b), 1e-2))
                assert err < 1e-7, f"rank {rank} step {tn} {nm}: {err}"
        ctx.close()
        open(os.path.join(tmpdir, f"ok{rank}"), "w").write("ok")
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("static_sigma", [False, True], ids=["default", "static-sigma"])
@pytest.mark.parametrize("world", [2, 3])
def test_band_runner_on_one_gpu(tmp_path, world, static_sigma):
    port = _free_port()
    mp.spawn(_worker, args=(world, port, 160, 96, 4, str(tmp_path), static_sigma), nprocs=world, join=True)
    assert all((tmp_path / f"ok{r}").exists() for r in range(world))


def _worker_wide(rank, world, port, nx, ny, nz, tmpdir):
    """Single precision, search radii up to 28: every band runs the 96-column strip kernel (k_strip32) inside a ghost
    frame of 28 cells; checked by the shared single-precision rule against the fp64 oracle of the whole grid."""
    sys.path.insert(0, ROOT)
    from oracle import fp32_criterion as crit
    from oracle.pyoracle import Oracle
    from seabreeze_param_amd import hip, synth
    from seabreeze_param_amd.bands import BandRunner, split_rows
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        dt = np.float32
        f8 = lambda a: np.ascontiguousarray(a, dtype=np.float64)
        orc = Oracle(8)
        st = synth.static_fields(nx, ny, dt)
        coast = orc.get_edges(f8(st.landfrac), f8(st.icefrac))
        kwin = 27
        cd = orc.get_dist(coast, f8(st.landfrac), st.lon, st.lat, maxdist=20000.0, kwin=kwin)
        cd = np.where(np.abs(cd) < 12000.0, np.sign(cd) * np.minimum(np.abs(cd), 179.0), cd).astype(dt)
        p = synth.pressure_3d(st, nz, dt)
        ctx = hip.Context(0)
        runner = BandRunner(ctx, torch, dist, rank, world, nx, ny, nz, halo=kwin + 1, dtype=dt)
        runner.upload_static(st.z, st.sigma, cd)
        full = [np.zeros((ny, nx)) for _ in range(4)]
        r0, r1 = split_rows(ny, world)[rank]
        band = (np.abs(f8(cd)) <= 180.0)[r0:r1]
        per = []
        for tn in (1, 2, 3):
            th = synth.theta_step(st, tn, dt)
            u, v = synth.wind_step(st, nz, tn, dt)
            s = runner.upload_step_inputs(p, u, v, th)
            gp = [t.cpu().numpy().copy() for t in (runner.ws, runner.wd, runner.thc, runner.sb_con)]
            op = [a[r0:r1].copy() for a in full]
            runner.step(7200.0, tn, s)
            ctx.synchronize()
            torch.cuda.synchronize()
            orc.seabreeze_diag(7200.0, tn, f8(p), f8(u), f8(v), f8(th), f8(cd), f8(st.z), f8(st.sigma), *full, halo=0, bnd=1)
            gn = [t.cpu().numpy().copy() for t in (runner.ws, runner.wd, runner.thc, runner.sb_con)]
            per.append(crit.check_step(tn, gp, gn, op, [a[r0:r1] for a in full], band, timestep=7200.0))
        res = crit.merge(per)
        assert res["ok"], (rank, res)
        assert orc.last_nn_max > 16 and ctx.last_counters()["global_path_cells"] == 0, (orc.last_nn_max, ctx.last_counters())
        ctx.close()
        open(os.path.join(tmpdir, f"ok{rank}"), "w").write("ok")
    finally:
        dist.destroy_process_group()


def test_band_runner_single_precision_wide_windows(tmp_path):
    port = _free_port()
    mp.spawn(_worker_wide, args=(2, port, 256, 192, 3, str(tmp_path)), nprocs=2, join=True)
    assert all((tmp_path / f"ok{r}").exists() for r in range(2))
