"""Probe: can two RCCL ranks share one GPU on this box?  (diagnostic, not a test)

    python -m torch.distributed.run --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29511 tools/probe_rccl_one_gpu.py
"""
import os
import time

import torch
import torch.distributed as dist

rank = int(os.environ["RANK"])
world = int(os.environ["WORLD_SIZE"])
torch.cuda.set_device(0)
try:
    dist.init_process_group("nccl", device_id=torch.device("cuda", 0))
    x = torch.full((5,), float(rank), device="cuda", dtype=torch.float64)
    g = torch.zeros(5 * world, device="cuda", dtype=torch.float64)
    dist.all_gather_into_tensor(g, x)
    torch.cuda.synchronize()
    print(rank, "all_gather ok", g.tolist(), flush=True)
    a = torch.zeros(1000, device="cuda")
    b = torch.ones(1000, device="cuda") * (rank + 1)
    ops = [dist.P2POp(dist.isend, b, 1 - rank), dist.P2POp(dist.irecv, a, 1 - rank)]
    for r in dist.batch_isend_irecv(ops):
        r.wait()
    torch.cuda.synchronize()
    print(rank, "p2p ok", float(a[0]), flush=True)
    t0 = time.perf_counter()
    for _ in range(200):
        dist.all_gather_into_tensor(g, x)
    torch.cuda.synchronize()
    print(rank, "all_gather us/call", (time.perf_counter() - t0) / 200 * 1e6, flush=True)
    t0 = time.perf_counter()
    for _ in range(200):
        for r in dist.batch_isend_irecv(ops):
            r.wait()
    torch.cuda.synchronize()
    print(rank, "p2p us/call", (time.perf_counter() - t0) / 200 * 1e6, flush=True)
    dist.destroy_process_group()
except Exception as e:  # noqa: BLE001
    print(rank, "FAILED:", type(e).__name__, str(e)[:500], flush=True)
