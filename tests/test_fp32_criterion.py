"""The single-precision parity rule (oracle/fp32_criterion.py) on the CPU: the fp32 build of the oracle stands in for
the fp32 GPU path, the fp64 build is the yardstick.  The rule must accept what rounding explains -- without masking a
cell -- and must reject an sb_con that its own thc / wind errors do not explain."""
import numpy as np

from oracle import fp32_criterion as crit
from oracle.pyoracle import Oracle
from seabreeze_param_amd import synth


def _run(nx=256, ny=192, nz=5, steps=(1, 2, 15)):
    dt = np.float32
    o4, o8 = Oracle(4), Oracle(8)
    st = synth.static_fields(nx, ny, dt)
    coast = o8.get_edges(st.landfrac.astype(np.float64), st.icefrac.astype(np.float64), rule=1, bnd=1)
    cdist = o8.get_dist(coast, st.landfrac.astype(np.float64), st.lon, st.lat, maxdist=180.0, kwin=6).astype(dt)
    cdist[np.abs(cdist) > 180.0] = 12000.0
    f8 = lambda a: np.ascontiguousarray(a, dtype=np.float64)
    p = synth.pressure_3d(st, nz, dt)
    s4 = [np.zeros((ny, nx), dt) for _ in range(4)]
    s8 = [np.zeros((ny, nx), np.float64) for _ in range(4)]
    band = np.abs(f8(cdist)) <= 180.0
    out = []
    for tn in steps:
        th = synth.theta_step(st, tn, dt)
        u, v = synth.wind_step(st, nz, tn, dt)
        p4, p8 = [a.copy() for a in s4], [a.copy() for a in s8]
        o4.seabreeze_diag(1440.0, tn, p, u, v, th, cdist, st.z, st.sigma, *s4, halo=0, bnd=1)
        o8.seabreeze_diag(1440.0, tn, f8(p), f8(u), f8(v), f8(th), f8(cdist), f8(st.z), f8(st.sigma), *s8, halo=0, bnd=1)
        out.append((tn, p4, [a.copy() for a in s4], p8, [a.copy() for a in s8]))
    return band, out


def test_rounding_is_explained_cell_by_cell():
    band, seq = _run()
    res = crit.merge([crit.check_step(tn, gp, gn, op, on, band, timestep=1440.0) for tn, gp, gn, op, on in seq])
    assert res["cells_compared"] > 1000 and res["masked_cells"] == 0
    assert res["max_err"]["sb_con_bound_ratio"] <= 1.0, res
    assert res["unexplained_flips"] == 0, res
    # the relative error of sb_con alone is NOT small near the knife edges -- which is why the rule is a bound per cell
    assert res["worst_ratio_cell"]["bound"] > 0


def test_an_unexplained_sb_con_error_fails():
    band, seq = _run(steps=(1, 2))
    tn, gp, gn, op, on = seq[1]
    both = np.argwhere(band & (gn[3] != 0) & (on[3] != 0) & (np.abs(np.abs(on[2]) - 0.75) > 0.2))
    j, i = both[len(both) // 2]
    bad = [a.copy() for a in gn]
    bad[3][j, i] *= np.float32(1.001)                       # 1e-3 relative, far from every threshold
    r = crit.check_step(tn, gp, bad, op, on, band, timestep=1440.0)
    assert r["sb_con_bound_ratio"] > 1.0
    assert (r["worst_ratio_cell"]["lat_index"], r["worst_ratio_cell"]["lon_index"]) == (int(j), int(i))
    # and a flip away from every knife edge is not explained
    bad = [a.copy() for a in gn]
    bad[3][j, i] = 0
    r = crit.check_step(tn, gp, bad, op, on, band, timestep=1440.0)
    assert r["trigger_flips"] >= 1 and r["unexplained_flips"] >= 1
    assert not crit.merge([r])["ok"]
