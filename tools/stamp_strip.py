"""Diagnostic: where does a k_strip workgroup spend its cycles?  Needs `make -C seabreeze_param_amd/csrc stamps`.

    SEABREEZE_HIP_LIB=$PWD/seabreeze_param_amd/libseabreeze_hip_stamps.so python tools/stamp_strip.py [nx ny nz]

Every wave sums the shader clock over the phases of its steps (diagnostic build only).  Shares, not run time.
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
NROW = 1024 + 64 * NWV
buf = (C.c_longlong * (NROW * NS))()
rc = ctx.lib.sb_debug_stamps(ctx.h, buf, C.c_int(NROW))
assert rc == 0, rc
allrows = np.frombuffer(buf, dtype=np.int64).reshape(NROW, NS)
s = allrows[:NWG]
names = ["prologue", "schedule", "S1 stage", "barrier", "S2", "-", "-", "round tail", "seg lists"]
steps, drains = s[:, 6], s[:, 5]
print(f"workgroups {NWG}: steps per workgroup min {steps.min()} mean {steps.mean():.2f} max {steps.max()}; drain passes mean {drains.mean():.2f}")
tot = s[:, [0, 1, 2, 3, 4, 7, 8]].sum(axis=1)
for i in (0, 1, 2, 3, 4, 7, 8):
    print(f"  {names[i]:10s} thread 0: per workgroup mean {s[:, i].mean():9.0f} cyc ({100 * s[:, i].sum() / tot.sum():4.1f} %)"
          + (f"   per pass {s[:, i].sum() / max(1, (steps + drains).sum()):7.0f}" if i in (2, 3, 4) else ""))
print(f"  prologue split: flag loads returned {s[:, 13].mean():.0f}, plane + partial sums {s[:, 14].mean():.0f}, barrier {s[:, 11].mean():.0f}, "
      f"prefix + share {s[:, 12].mean():.0f}, picks + schedule {s[:, 15].mean():.0f}, barrier + statistics + first entries {s[:, 1].mean():.0f}")
print(f"  total cycles per workgroup mean {tot.mean():.0f} max {tot.max()}; wall (10 ns ticks): start spread "
      f"{s[:, 9].max() - s[:, 9].min()}, life mean {np.mean(s[:, 10] - s[:, 9]):.0f} max {np.max(s[:, 10] - s[:, 9])}, "
      f"kernel span {s[:, 10].max() - s[:, 9].min()}")
pw = allrows[1024:1024 + 64 * NWV].reshape(64, NWV, NS).astype(np.float64)
npass = np.maximum(pw[:, :, 5] + pw[:, :, 6], 1)
print("  S1 split, cycles per pass (wave: decode, restart+band words, load wait, t0, scans+writes, list, issue):")
for w in range(NWV):
    m = [(pw[:, w, i] / npass[:, w]).mean() for i in (16, 17, 18, 19, 20, 21, 2)]
    print(f"      {w:3d}  " + " ".join(f"{x:6.0f}" for x in m))
print("  per wave, cycles per pass:  wave     S1   barrier     S2")
for w in range(NWV):
    m = [(pw[:, w, i] / npass[:, w]).mean() for i in (2, 3, 4)]
    print(f"                              {w:3d}  " + " ".join(f"{x:7.0f}" for x in m))
print(ctx.last_counters())
