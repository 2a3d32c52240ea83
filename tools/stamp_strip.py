"""Diagnostic: where does a k_strip workgroup spend its cycles?  Needs `make -C seabreeze_param_amd/csrc stamps`.

    SEABREEZE_HIP_LIB=$PWD/seabreeze_param_amd/libseabreeze_hip_stamps.so python tools/stamp_strip.py [nx ny nz]

Every wave leaves the 100 MHz wall clock at the marks of sb_strip_kernel.hip (diagnostic build only).
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
for tn in (1, 2, 3):
    ctx.seabreeze_diag(1440.0, tn, p, u, v, th, cdist, st.z, st.sigma, *state)
NWG, NS, NWV = 256, 32, 16
NROW = NWG * NWV
buf = (C.c_longlong * (NROW * NS))()
rc = ctx.lib.sb_debug_stamps(ctx.h, buf, C.c_int(NROW))
assert rc == 0, rc
t = np.frombuffer(buf, dtype=np.int64).reshape(NWG, NWV, NS).astype(np.float64) / 100.0     # us
t0 = t[:, :, 0].min()
k_end = t[:, :, 7].max()
print(f"kernel span (first wave's start to last wave's end) {k_end - t0:.2f} us; workgroup starts within {t[:, 0, 0].max() - t0:.2f} us")
names = ["start", "first barrier reached", "... passed", "planned", "march begins", "march done", "marked cells done", "end"]
print("marks, us after the kernel's first wave started; mean over workgroups [max]")
for w in (0, 5, 8, 15):
    row = []
    for i in (0, 1, 2, 4, 5, 6, 7):
        v = t[:, w, i] - t0
        row.append(f"{names[i]} {v.mean():.2f} [{v.max():.2f}]")
    print(f"  wave {w:2d}: " + "; ".join(row))
for w in (0, 15):
    print(f"  wave {w:2d}: barrier passed -> round begins {np.mean(t[:, w, 28] - t[:, w, 2]):.2f}, entries read {np.mean(t[:, w, 29] - t[:, w, 28]):.2f}, "
          f"first issue {np.mean(t[:, w, 30] - t[:, w, 29]):.2f}, two more {np.mean(t[:, w, 31] - t[:, w, 30]):.2f}, statistics {np.mean(t[:, w, 4] - t[:, w, 31]):.2f}")
# steps: time between the beginnings of consecutive steps, wave 0, by step number
print("step i begins -> step i + 1 begins, us (wave 0 | wave 15), mean over the workgroups that have the step")
for i in range(0, NS - 9):
    a, b = t[:, :, 8 + i], t[:, :, 9 + i]
    ok = (b > a) & (a >= t[:, :, 4]) & (b <= t[:, :, 5] + 1e-9)
    if ok[:, 0].sum() == 0:
        break
    print(f"   step {i:2d}: n {int(ok[:, 0].sum()):3d}   {np.mean((b - a)[:, 0][ok[:, 0]]):.2f} | {np.mean((b - a)[:, 15][ok[:, 15]]) if ok[:, 15].any() else float('nan'):.2f}")
life = t[:, :, 7].max(axis=1) - t[:, :, 0].min(axis=1)
print(f"workgroup life: mean {life.mean():.2f} max {life.max():.2f} min {life.min():.2f} us; march (wave 0) mean {np.mean(t[:, 0, 5] - t[:, 0, 4]):.2f} max {np.max(t[:, 0, 5] - t[:, 0, 4]):.2f}")
print(ctx.last_counters())
