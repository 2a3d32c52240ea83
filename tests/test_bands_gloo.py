"""world_size-2 (and 3) gloo test of the latitude-band path on CPU.

What runs here is the HOST side of the multi-GPU path -- band split, N-S halo exchange
over torch.distributed point-to-point, E-W ghost fill, pole replication -- with the CPU
oracle standing in for the kernels as the checker: every band, fed only its own rows
plus exchanged ghosts (and the global sigmoid scalars), must reproduce the rows the
single-domain run gives.  On the GPU box the same helpers run over RCCL.
"""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import ROOT


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, nx, ny, nz, h, tmpdir, balanced):
    sys.path.insert(0, ROOT)
    from oracle.pyoracle import Oracle
    from seabreeze_param_amd import bands, synth
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        orc = Oracle(8)
        st = synth.static_fields(nx, ny)
        coast = orc.get_edges(st.landfrac, st.icefrac)
        cdist = orc.get_dist(coast, st.landfrac, st.lon, st.lat, maxdist=900.0, kwin=h - 1)
        cdist[np.abs(cdist) > 180.0] = 12000.0
        # equal row counts, or bands cut by cost (unequal heights): every rank derives the same cuts
        cost = bands.row_cost(np.abs(cdist) <= 180.0, nz) if balanced else None
        r0, r1 = bands.split_rows(ny, world, cost=cost, min_rows=h)[rank]
        nyl = r1 - r0

        def halo_field(full):
            loc = torch.zeros((nyl + 2 * h, nx + 2 * h), dtype=torch.float64)
            loc[h:h + nyl, h:h + nx] = torch.from_numpy(np.ascontiguousarray(full[r0:r1]))
            bands.exchange_ns(loc, nyl, h, rank, world, dist, torch)
            bands.fill_ew_ghosts(loc, nx, h)
            return loc

        # what the exchange must produce: the global field, lat-clamped and lon-wrapped
        def expected(full):
            rows = np.clip(np.arange(r0 - h, r1 + h), 0, ny - 1)
            cols = np.arange(-h, nx + h) % nx
            return full[np.ix_(rows, cols)]

        z, sg, mk = (halo_field(a) for a in (st.z, st.sigma, cdist))
        for got, full in ((z, st.z), (sg, st.sigma), (mk, cdist)):
            assert np.array_equal(got.numpy(), expected(full)), "halo exchange mismatch"

        p3 = synth.pressure_3d(st, nz)
        ext = orc.sigmoid_scalars(st.sigma)                  # the all-gathered global statistics
        state = [np.zeros((nyl, nx)) for _ in range(4)]
        full_state = [np.zeros((ny, nx)) for _ in range(4)]
        for tn in (1, 2, 3):
            th = synth.theta_step(st, tn)
            u, v = synth.wind_step(st, nz, tn)
            th_loc = halo_field(th)                           # theta ghosts swapped every step
            orc.seabreeze_diag(7200.0, tn, p3[:, r0:r1], u[:, r0:r1], v[:, r0:r1], th_loc.numpy(), mk.numpy(),
                               z.numpy(), sg.numpy(), *state, halo=h, bnd=2, ext_stats=ext)
            orc.seabreeze_diag(7200.0, tn, p3, u, v, th, cdist, st.z, st.sigma, *full_state, halo=0, bnd=1)
            for a, b in zip(state, full_state):
                assert np.array_equal(a, b[r0:r1]), f"rank {rank} step {tn}: band != global rows"
        assert orc.last_nn_max <= h
        open(os.path.join(tmpdir, f"ok{rank}"), "w").write("ok")
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,balanced", [(2, False), (3, False), (2, True), (3, True)])
def test_band_decomposition_reproduces_global(tmp_path, world, balanced):
    port = _free_port()
    mp.spawn(_worker, args=(world, port, 96, 72, 3, 6, str(tmp_path), balanced), nprocs=world, join=True)
    assert all((tmp_path / f"ok{r}").exists() for r in range(world))
