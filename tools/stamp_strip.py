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
dt = np.float32 if "f32" in sys.argv[4:] else np.float64      # f32 with a window beyond 16 cells: the marks of k_strip32
st = synth.static_fields(nx, ny, dt)
ctx = hip.Context()
if "replan" in sys.argv[4:]:     # the planning call's path: no stored plan
    ctx.set_plan_cache(False)
coast = ctx.get_edges(st.landfrac, st.icefrac)
cdist = ctx.get_dist(coast, st.landfrac, st.lon, st.lat)
ctx.set_search_radius_hint(hip.dist_window(st.lon, st.lat) + 1)
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
for i in range(0, NS - (14 if 'f32' in sys.argv[4:] else 9)):
    a, b = t[:, :, 8 + i], t[:, :, 9 + i]
    ok = (b > a) & (a >= t[:, :, 4]) & (b <= t[:, :, 5] + 1e-9)
    if ok[:, 0].sum() == 0:
        break
    print(f"   step {i:2d}: n {int(ok[:, 0].sum()):3d}   {np.mean((b - a)[:, 0][ok[:, 0]]):.2f} | {np.mean((b - a)[:, 15][ok[:, 15]]) if ok[:, 15].any() else float('nan'):.2f}")
if "f32" in sys.argv[4:]:
    # k_strip32: the inside of the march's tenth step (marks 27..31), by wave: queries on waves 0-7, sums along latitude on 8-10
    have = (t[:, 0, 31] > 0) & (t[:, 0, 27] > 0)
    print(f"inside step 12 ({int(have.sum())} workgroups): wave: step begin -> staged | -> barrier reached | barrier wait | queries or sums | second barrier wait")
    for w in (0, 3, 7, 8, 10, 12, 15):
        a = t[have][:, w, :]
        b0 = a[:, 5 + 12]
        print(f"   wave {w:2d}: {np.mean(a[:, 27] - b0):.2f} | {np.mean(a[:, 28] - a[:, 27]):.2f} | {np.mean(a[:, 29] - a[:, 28]):.2f} | {np.mean(a[:, 30] - a[:, 29]):.2f} | {np.mean(a[:, 31] - a[:, 30]):.2f}")
life = t[:, :, 7].max(axis=1) - t[:, :, 0].min(axis=1)
print(f"workgroup life: mean {life.mean():.2f} max {life.max():.2f} min {life.min():.2f} us; march (wave 0) mean {np.mean(t[:, 0, 5] - t[:, 0, 4]):.2f} max {np.max(t[:, 0, 5] - t[:, 0, 4]):.2f}")
# what a workgroup's life is made of: its steps by kind, from the stored plan; least squares
STRIDE, ENT_OFF = 3136 + 2048 * 64, 64
raw = (C.c_ubyte * (64 + NWG * STRIDE))()
if hasattr(ctx.lib, "sb_debug_plan") and ctx.lib.sb_debug_plan(ctx.h, raw, C.c_longlong(len(raw))) == 0:
    plan = np.frombuffer(raw, dtype=np.uint8)[64:].reshape(NWG, STRIDE)
    hdr = plan[:, :16].copy().view(np.int32)
    rows = []
    for b in range(NWG):
        n = int(hdr[b, 1])
        e = plan[b, ENT_OFF:ENT_OFF + 8 * n].copy().view(np.uint32).reshape(n, 2)[3:, 0]
        idle, drain = (e >> 20) & 1, (e >> 18) & 1
        q = np.where(drain == 1, (e >> 19) & 1, (e >> 16) & 1) * (1 - idle)
        rows.append([1.0, np.sum((1 - drain) * (1 - idle)), np.sum(q), np.sum(drain * (1 - idle)), np.sum((e >> 17) & 1 * (1 - idle)), np.sum(idle)])
    A = np.array(rows)
    coef, *_ = np.linalg.lstsq(A, life, rcond=None)
    print("life ~ %.2f + %.2f staged + %.2f queried + %.2f drain + %.2f run + %.2f padding   (us; residual rms %.2f)"
          % (*coef, float(np.sqrt(np.mean((A @ coef - life) ** 2)))))
    np.save("gpurun_out/plan_head.npy", plan[:, :3136].copy())
    order = np.argsort(life)
    print("  the five shortest and the five longest lives: workgroup, life, fitted, [staged queried drains runs padding], cells queried")
    ncell = []
    for b in range(NWG):
        lists = plan[b, 3136:3136 + 2048 * 64].copy().view(np.uint32).reshape(64, 512)
        ncell.append(int(np.sum(lists[:int(A[b, 2])] != 0xffffffff)))
    for b in list(order[:5]) + list(order[-5:]):
        print(f"     {b:4d} {life[b]:6.2f} {float(A[b] @ coef):6.2f} {A[b, 1:].astype(int).tolist()} {ncell[b]}")
    A2 = np.column_stack([A, np.array(ncell) / 64.0])
    c2, *_ = np.linalg.lstsq(A2, life, rcond=None)
    print("  with the number of 64-cell query passes as a sixth term: %s residual rms %.2f" % (np.round(c2, 2).tolist(), float(np.sqrt(np.mean((A2 @ c2 - life) ** 2)))))
    print("per workgroup: staged mean %.2f max %d; queried mean %.2f max %d; drains mean %.2f max %d; runs mean %.2f max %d"
          % (A[:, 1].mean(), A[:, 1].max(), A[:, 2].mean(), A[:, 2].max(), A[:, 3].mean(), A[:, 3].max(), A[:, 4].mean(), A[:, 4].max()))
print(ctx.last_counters())
