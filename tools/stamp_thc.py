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
NS = 32
buf = (C.c_longlong * (nt_max * NS))()
n = C.c_int(0)
rc = ctx.lib.sb_debug_stamps(ctx.h, buf, C.c_int(nt_max), C.byref(n))
assert rc == 0, rc
s = np.frombuffer(buf, dtype=np.int64)[: n.value * NS].reshape(n.value, NS)
LAST = 5
active = s[:, LAST] != 0
print(f"tiles {n.value}, active {int(active.sum())}")
names, idx = ["T0 compact", "T1 regs->LDS", "T2 band prefix", "T3 lon prefix", "T4 search"], [0, 1, 2, 3, 4, 5]
a = s[active]
for i, nm in enumerate(names):
    d = a[:, idx[i + 1]] - a[:, idx[i]]
    print(f"  {nm:14s} mean {d.mean():9.0f} cyc   median {np.median(d):9.0f}   max {d.max():9.0f}")
first = a[a[:, 9] != 0]
if len(first):
    w0 = first[:, 8].min()
    print(f"  prologue per workgroup ({len(first)}): list build mean {first[:, 10].mean():.0f} cyc, whole prologue mean "
          f"{first[:, 9].mean():.0f}, max {first[:, 9].max()}")
    print(f"  wall clock (10 ns ticks): workgroup starts spread over {first[:, 8].max() - w0}, last wave of a workgroup "
          f"starts {np.mean(first[:, 12] - first[:, 8]):.0f} after its first (max {np.max(first[:, 12] - first[:, 8])}); "
          f"workgroup life mean {np.mean(first[:, 11] - first[:, 8]):.0f}, kernel span {first[:, 11].max() - w0}")
    print("  prologue stamps (cycles since start): before list %.0f, list+first issue %.0f, zeroed %.0f, wave merges %.0f, "
          "after barrier %.0f" % tuple(first[:, 16 + i].mean() for i in range(5)))
    print("  T0 split: s_word %.0f, late issue %.0f, barrier %.0f, compaction+barrier %.0f, cell/state loads %.0f" % (
        (a[:, 24] - a[:, 0]).mean(), (a[:, 25] - a[:, 24]).mean(), (a[:, 26] - a[:, 25]).mean(),
        (a[:, 27] - a[:, 26]).mean(), (a[:, 1] - a[:, 27]).mean()))
    rest = a[a[:, 9] == 0]
    for nm, grp in (("first tile of a workgroup", first), ("later tiles", rest)):
        if len(grp):
            ph = [f"{(grp[:, idx[i + 1]] - grp[:, idx[i]]).mean():.0f}" for i in range(len(names))]
            print(f"  {nm}: phases {ph} total {(grp[:, LAST] - grp[:, 0]).mean():.0f}")
tot = a[:, LAST] - a[:, 0]
print(f"  {'tile total':14s} mean {tot.mean():9.0f} cyc   median {np.median(tot):9.0f}   max {tot.max():9.0f}")
print(ctx.last_counters())
