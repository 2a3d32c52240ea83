"""Diagnostic: where does a k_thc3 workgroup spend its cycles?  Needs `make -C seabreeze_param_amd/csrc stamps`.

    SEABREEZE_HIP_LIB=$PWD/seabreeze_param_amd/libseabreeze_hip_stamps.so python tools/stamp_thc.py [nx ny nz [threads [prefetch]]]

Thread 0 of every workgroup sums the shader clock over the phases of its tiles (diagnostic build only; the
prefetch wait is made explicit there, which the real kernel does not do).  Shares, not absolute run time.
"""
import ctypes as C
import sys

import numpy as np

sys.path.insert(0, ".")
from seabreeze_param_amd import hip, synth  # noqa: E402

nx, ny, nz = (int(a) for a in sys.argv[1:4]) if len(sys.argv) >= 4 else (2560, 1920, 8)
threads = int(sys.argv[4]) if len(sys.argv) >= 5 else 1024
prefetch = len(sys.argv) >= 6 and sys.argv[5] == "prefetch"
dt = np.float64
st = synth.static_fields(nx, ny, dt)
ctx = hip.Context()
ctx.set_thc_threads(threads)
ctx.set_thc_prefetch(prefetch)
coast = ctx.get_edges(st.landfrac, st.icefrac)
cdist = ctx.get_dist(coast, st.landfrac, st.lon, st.lat)
p = synth.pressure_3d(st, nz, dt)
u, v = synth.wind_step(st, nz, 1, dt)
th = synth.theta_step(st, 1, dt)
state = [np.zeros((ny, nx), dt) for _ in range(4)]
for tn in (1, 2, 3):
    ctx.seabreeze_diag(1440.0, tn, p, u, v, th, cdist, st.z, st.sigma, *state)
NWG, NS = 256, 16
NROW = 1024 + 64 * 16
buf = (C.c_longlong * (NROW * NS))()
rc = ctx.lib.sb_debug_stamps(ctx.h, buf, C.c_int(NROW))
assert rc == 0, rc
allrows = np.frombuffer(buf, dtype=np.int64).reshape(NROW, NS)
s = allrows[:NWG]
tiles = s[:, 8]
print(f"{threads} threads; workgroups {NWG}, tiles per workgroup min {tiles.min()} mean {tiles.mean():.2f} max {tiles.max()} (total {tiles.sum()})")
names = ["prologue", "prefetch wait", "A1 compute", "barrier 1", "A2 tables", "barrier 2", "A3 search"]
tot = s[:, :7].sum(axis=1)
for i, nm in enumerate(names):
    per_tile = s[:, i] / np.maximum(tiles, 1) if i else s[:, i]
    print(f"  {nm:14s} per workgroup mean {s[:, i].mean():9.0f} cyc ({100 * s[:, i].sum() / tot.sum():4.1f} %)   "
          f"{'per tile' if i else 'once   '} mean {per_tile.mean():8.0f}")
print(f"  total cycles per workgroup mean {tot.mean():.0f} max {tot.max()}; wall (10 ns ticks) start spread "
      f"{s[:, 9].max() - s[:, 9].min()}, life mean {np.mean(s[:, 10] - s[:, 9]):.0f} max {np.max(s[:, 10] - s[:, 9])}, "
      f"kernel span {s[:, 10].max() - s[:, 9].min()}")
print(f"  prologue split (cycles since kernel start, mean): borders zeroed {s[:, 11].mean():.0f}, list entries + scalars arrived {s[:, 12].mean():.0f}, first tile issued {s[:, 13].mean():.0f}, barrier passed {s[:, 0].mean():.0f}")
print(ctx.last_counters())
# per wave (mean over the first 64 workgroups, cycles per tile): which waves arrive late at the barriers
nwv = threads // 64
pw = allrows[1024:1024 + 64 * nwv].reshape(64, nwv, NS).astype(np.float64)
tl = np.maximum(s[:64, 8], 1)[:, None]
print("  per wave, cycles per tile:  wave  pf-wait   A1   bar1    A2   bar2    A3")
for w in range(nwv):
    m = [(pw[:, w, i] / tl[:, 0]).mean() for i in range(1, 7)]
    print(f"                              {w:3d}  " + " ".join(f"{x:6.0f}" for x in m))
