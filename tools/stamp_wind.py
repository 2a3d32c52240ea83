"""Diagnostic: where does a wave of k_wind spend its life?  Needs `make -C seabreeze_param_amd/csrc stamps EXTRA=-DSB_STAMPS_WIND`.

    SEABREEZE_HIP_LIB=$PWD/seabreeze_param_amd/libseabreeze_hip_stamps.so python tools/stamp_wind.py [nx ny nz [f32]]

Every wave leaves the 100 MHz wall clock at the marks of k_wind (sb_diag_kernels.hip, diagnostic build only): start, the
sub-lists' sizes there, end, and for each of its first segments: entry there, column walked, u and v there, update done.
"""
import ctypes as C
import sys

import numpy as np

sys.path.insert(0, ".")
from seabreeze_param_amd import hip, synth  # noqa: E402

nx, ny, nz = (int(a) for a in sys.argv[1:4]) if len(sys.argv) >= 4 else (1024, 768, 56)
dt = np.float32 if len(sys.argv) > 4 and sys.argv[4] == "f32" else np.float64
st = synth.static_fields(nx, ny, dt)
ctx = hip.Context()
coast = ctx.get_edges(st.landfrac, st.icefrac)
cdist = ctx.get_dist(coast, st.landfrac, st.lon, st.lat)
ctx.set_search_radius_hint(hip.dist_window(st.lon, st.lat) + 1)
p = synth.pressure_3d(st, nz, dt)
u, v = synth.wind_step(st, nz, 1, dt)
th = synth.theta_step(st, 1, dt)
state = [np.zeros((ny, nx), dt) for _ in range(4)]
for tn in (1, 2, 3, 4):
    ctx.seabreeze_diag(1440.0, tn, p, u, v, th, cdist, st.z, st.sigma, *state)
NROW, NS = 4096, 32
buf = (C.c_longlong * (NROW * NS))()
rc = ctx.lib.sb_debug_stamps(ctx.h, buf, C.c_int(NROW))
assert rc == 0, rc
t = np.frombuffer(buf, dtype=np.int64).reshape(NROW, NS).astype(np.float64) / 100.0     # us
live = t[:, 0] > 0
t0 = t[live, 0].min()
worked = live & (t[:, 3] > 0)
print(f"{int(live.sum())} waves started within {t[live, 0].max() - t0:.2f} us; {int(worked.sum())} of them had a segment; "
      f"the last one ended {t[live, 2].max() - t0:.2f} us after the first started")
print(f"sub-lists' sizes there: {np.mean(t[live, 1] - t[live, 0]):.2f} us after a wave's start (max {np.max(t[live, 1] - t[live, 0]):.2f})")
for k in range(7):
    b = 3 + 4 * k
    ok = worked & (t[:, b] > 0) & (t[:, b + 3] > 0)
    if not ok.any():
        break
    prev = t[:, 1] if k == 0 else t[:, b - 1]
    print(f"  segment {k} ({int(ok.sum())} waves): entry there {np.mean((t[:, b] - prev)[ok]):.2f} us after the mark before; walk {np.mean((t[:, b + 1] - t[:, b])[ok]):.2f}"
          f" [max {np.max((t[:, b + 1] - t[:, b])[ok]):.2f}]; u, v {np.mean((t[:, b + 2] - t[:, b + 1])[ok]):.2f}; update + stores {np.mean((t[:, b + 3] - t[:, b + 2])[ok]):.2f};"
          f" begins {np.mean(t[ok, b] - t0):.2f} after the kernel's start, ends {np.mean(t[ok, b + 3] - t0):.2f} [max {np.max(t[ok, b + 3] - t0):.2f}]")
life = (t[:, 2] - t[:, 0])[worked]
print(f"life of a wave with work: mean {life.mean():.2f} max {life.max():.2f} us")
print(ctx.last_counters())
