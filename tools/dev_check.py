"""Developer check: HIP path vs CPU oracle on a few grids (run on the GPU box).

    python tools/dev_check.py [nx ny nz steps]

Not a test -- tests/ holds the parity suite; this prints error summaries.
"""
import sys
import time

import numpy as np

sys.path.insert(0, ".")
from oracle.pyoracle import Oracle  # noqa: E402  (checker only)
from seabreeze_param_amd import hip, synth  # noqa: E402


def relerr(a, b):
    d = np.abs(a - b)
    s = np.maximum(np.abs(b), 1e-30)
    return float(np.nanmax(d / s)), float(np.nanmax(d))


def run(nx, ny, nz, steps, dt):
    prec = 8 if dt == np.float64 else 4
    O = Oracle(prec)
    ctx = hip.Context()
    st = synth.static_fields(nx, ny, dt)
    # setup chain: edges -> dist, both sides
    ce_o = O.get_edges(st.landfrac, st.icefrac)
    ce_h = ctx.get_edges(st.landfrac, st.icefrac)
    print(f"[{nx}x{ny} {dt.__name__}] edges equal: {np.array_equal(ce_o, ce_h)}  ncoast={int(ce_o.sum())}")
    t = time.time(); cd_o = O.get_dist(ce_o, st.landfrac, st.lon, st.lat); t_o = time.time() - t
    t = time.time(); cd_h = ctx.get_dist(ce_h, st.landfrac, st.lon, st.lat); t_h = time.time() - t
    k = O.dist_window(st.lon, st.lat)
    print(f"  dist: k={k} hipk={hip.dist_window(st.lon, st.lat)} rel/abs err {relerr(cd_h, cd_o)} sign-eq "
          f"{np.array_equal(np.sign(cd_h), np.sign(cd_o))}  cpu {t_o:.3f}s hip(host api) {t_h:.3f}s")
    sm_o = O.sigmoid(st.sigma); sm_h = ctx.sigmoid(st.sigma)
    print(f"  sigmoid rel/abs err {relerr(sm_h, sm_o)}")

    # wrapper flavour
    p1 = synth.pressure_1d(nz, dt)
    z2 = np.zeros((ny, nx), dt)
    so = [z2.copy() for _ in range(3)]
    sh = [z2.copy() for _ in range(3)]
    for tn in range(1, steps + 1):
        th = synth.theta_step(st, tn, dt)
        u, v = synth.wind_step(st, nz, tn, dt)
        oo = O.diag(tn, p1, st.z, st.sigma, th, v, u, cd_o, *so, timestep=90.0)
        oh = ctx.diag(tn, p1, st.z, st.sigma, th, v, u, cd_o, *sh, timestep=90.0)
        a, b = oh[:, :-1], oo[:, :-1]
        mism = int((a[0] != b[0]).sum())
        print(f"  wrapper tn={tn}: sb rel/abs {relerr(a[0], b[0])} t0 {relerr(a[1], b[1])} ws {relerr(a[2], b[2])} "
              f"wd {relerr(a[3], b[3])} thc {relerr(sh[2], so[2])} sb!=: {mism} nn_max={O.last_nn_max} "
              f"{ctx.last_counters()}")
    # generic flavour
    p3 = synth.pressure_3d(st, nz, dt)
    so = [z2.copy() for _ in range(4)]
    sh = [z2.copy() for _ in range(4)]
    for tn in range(1, steps + 1):
        th = synth.theta_step(st, tn, dt)
        u, v = synth.wind_step(st, nz, tn, dt)
        O.seabreeze_diag(5400.0, tn, p3, u, v, th, cd_o, st.z, st.sigma, *so, halo=0, bnd=1)
        ctx.seabreeze_diag(5400.0, tn, p3, u, v, th, cd_o, st.z, st.sigma, *sh, halo=0, bnd=1)
        print(f"  generic tn={tn}: sb {relerr(sh[3], so[3])} ws {relerr(sh[0], so[0])} wd {relerr(sh[1], so[1])} "
              f"thc {relerr(sh[2], so[2])} sb!=: {int((sh[3] != so[3]).sum())} {ctx.last_counters()}")
    ctx.close()


if __name__ == "__main__":
    if len(sys.argv) >= 5:
        cfgs = [tuple(int(x) for x in sys.argv[1:5])]
    else:
        cfgs = [(96, 72, 8, 5), (256, 192, 9, 5), (1024, 768, 17, 3)]
    for (nx, ny, nz, steps) in cfgs:
        for dt in (np.float64, np.float32):
            run(nx, ny, nz, steps, dt)
