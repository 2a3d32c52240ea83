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
