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
