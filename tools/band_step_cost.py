"""Diagnostic: what a latitude-band step costs beyond its kernels, on ONE GPU.

    python tools/band_step_cost.py [nx ny_band nz halo steps [ny_full f32|f64]]

A band of ny_band rows of a 2560-wide grid (what one of N ranks owns; its ghost rows are the real
neighbouring rows, only theta's are re-filled locally by the band step) is run (a) as a plain
seabreeze_diag_dev call on its ghost-celled frame (SB_BND_HALO, ghosts already filled: kernels only) and
(b) as sb_band_seabreeze_diag_*_dev with a one-rank communicator (RCCL loaded, no neighbour: the all-gather
is a copy, the exchange is the local fill) -- the same launches, streams, events and joins a multi-rank step
issues, without the wire.  The difference is the fixed per-step overhead of the band machinery; real RCCL
latency comes on top of it.
"""
import sys
import time

import numpy as np
import torch

sys.path.insert(0, ".")
from seabreeze_param_amd import hip, synth  # noqa: E402

nx, nyb, nz, h, steps = (int(a) for a in sys.argv[1:6]) if len(sys.argv) >= 6 else (2560, 240, 56, 16, 200)
ny_full = int(sys.argv[6]) if len(sys.argv) >= 7 else 1920
dt = np.float32 if len(sys.argv) >= 8 and sys.argv[7] == "f32" else np.float64
tdt = torch.float32 if dt == np.float32 else torch.float64
st = synth.static_fields(nx, ny_full, dt)
ctx = hip.Context(0)
coast = ctx.get_edges(st.landfrac, st.icefrac)
cdist = ctx.get_dist(coast, st.landfrac, st.lon, st.lat)
r0 = (ny_full - nyb) // 2 + (300 * ny_full) // 1920    # a band with coast in it
rows = slice(r0, r0 + nyb)
stream = torch.cuda.current_stream().cuda_stream
ctx.comm_init(hip.comm_unique_id(), 0, 1)
ctx.set_search_radius_hint(h)


def frame(a):
    """The band with the ghost cells a neighbour would have sent: real rows north and south, the
    longitude wrap east and west."""
    g = np.concatenate([a[r0 - h:r0 + nyb + h, -h:], a[r0 - h:r0 + nyb + h], a[r0 - h:r0 + nyb + h, :h]], axis=1)
    return torch.from_numpy(np.ascontiguousarray(g)).cuda()


z, sg, mk = frame(st.z), frame(st.sigma), frame(cdist)
p = torch.from_numpy(synth.pressure_3d(st, nz, dt, rows=(r0, r0 + nyb))).cuda()
u_, v_ = synth.wind_step(st, nz, 1, dt, rows=(r0, r0 + nyb))
u, v = torch.from_numpy(u_).cuda(), torch.from_numpy(v_).cuda()
th = frame(synth.theta_step(st, 1, dt))
state = [torch.zeros((nyb, nx), dtype=tdt, device="cuda") for _ in range(4)]
args = (p.data_ptr(), u.data_ptr(), v.data_ptr(), th.data_ptr(), mk.data_ptr(), z.data_ptr(), sg.data_ptr(),
        *[s.data_ptr() for s in state])


def timed(fn):
    for tn in range(1, 6):
        fn(tn)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for tn in range(6, 6 + steps):
        fn(tn)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps * 1e6


plain = timed(lambda tn: ctx.seabreeze_diag_dev(dt, 1440.0, tn, nx, nyb, nz, h, hip.SB_BND_HALO, *args, stream))
ctx.profile_begin(20)
for tn in range(300, 320):
    ctx.seabreeze_diag_dev(dt, 1440.0, tn, nx, nyb, nz, h, hip.SB_BND_HALO, *args, stream)
km, _ = ctx.profile_end()
print("plain call kernels (us):", {k: round(v * 1e3, 1) for k, v in km.items()}, ctx.last_counters())
print("plain call enqueues:", ctx.last_step_report())
band = timed(lambda tn: ctx.band_seabreeze_diag_dev(dt, 1440.0, tn, nx, nyb, nz, h, *args, stream))
rep = ctx.last_step_report()
print(f"band of {nx}x{nyb}x{nz} ({'fp32' if dt == np.float32 else 'fp64'}, of a {nx}x{ny_full} grid), halo {h}: plain call {plain:.1f} us, band step (one-rank communicator) {band:.1f} us, "
      f"band machinery {band - plain:.1f} us per step; a band step enqueues {rep} (an interior rank of a multi-rank run: "
      f"4 RCCL sends/receives in one group + 1 all-gather on top)")
# opt-in: sigma's statistics formed once (sb_set_static_sigma): no moments pass, no all-gather, no merge after step 1
ctx.set_static_sigma(True)
band_s = timed(lambda tn: ctx.band_seabreeze_diag_dev(dt, 1440.0, tn, nx, nyb, nz, h, *args, stream))
print(f"  with static sigma (opt-in): band step {band_s:.1f} us, enqueues {ctx.last_step_report()}")
ctx.set_static_sigma(False)
ctx.comm_finalize()
ctx.close()
