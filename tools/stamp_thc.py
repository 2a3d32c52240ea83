"""Diagnostic: where does a k_thc tile spend its cycles?  Needs `make -C seabreeze_param_amd/csrc stamps`.

    SEABREEZE_HIP_LIB=$PWD/seabreeze_param_amd/libseabreeze_hip_stamps.so python tools/stamp_thc.py [nx ny nz]

Reads the shader-clock stamps thread 0 of every tile wrote (diagnostic build only) and
prints the mean cycles per phase over active tiles.  Shares, not absolute run time.
"""
import ctypes as C
import sys

import numpy as np

sys.path.insert(0, ".")
from seabreeze_param_amd import hip, synth  # noqa: E402

nx, ny, nz = (int(a) for a in sys.argv[1:4]) if len(sys.argv) >= 4 else (2560, 1920, 8)
dt = np.float64
st = synth.static_fields(nx, ny, dt)
ctx = hip.Context()
coast = ctx.get_edges(st.landfrac, st.icefrac)
cdist = ctx.get_dist(coast, st.landfrac, st.lon, st.lat)
p = synth.pressure_3d(st, nz, dt)
u, v = synth.wind_step(st, nz, 1, dt)
th = synth.theta_step(st, 1, dt)
state = [np.zeros((ny, nx), dt) for _ in range(4)]
for tn in (1, 2):
    ctx.seabreeze_diag(1440.0, tn, p, u, v, th, cdist, st.z, st.sigma, *state)
nt_max = 1 << 16
buf = (C.c_longlong * (nt_max * 8))()
n = C.c_int(0)
rc = ctx.lib.sb_debug_stamps(ctx.h, buf, C.c_int(nt_max), C.byref(n))
assert rc == 0, rc
s = np.frombuffer(buf, dtype=np.int64)[: n.value * 8].reshape(n.value, 8)
LAST = 5
active = s[:, LAST] != 0
print(f"tiles {n.value}, active {int(active.sum())}")
names = ["loads->LDS", "lon scan", "lat scan", "search+store"]
idx = [0, 1, 2, 3, 5]
a = s[active]
for i, nm in enumerate(names):
    d = a[:, idx[i + 1]] - a[:, idx[i]]
    print(f"  {nm:14s} mean {d.mean():9.0f} cyc   median {np.median(d):9.0f}   max {d.max():9.0f}")
tot = a[:, LAST] - a[:, 0]
print(f"  {'tile total':14s} mean {tot.mean():9.0f} cyc   median {np.median(tot):9.0f}   max {tot.max():9.0f}")
print(ctx.last_counters())
