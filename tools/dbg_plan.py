import sys, os
sys.path.insert(0, os.getcwd())
import numpy as np
from oracle.pyoracle import Oracle
from seabreeze_param_amd import hip, synth
nx, ny, nz = 256, 192, 2
dt, orc = np.float64, Oracle(8)
ctx = hip.Context(0)
st = synth.static_fields(nx, ny, dt)
coast = orc.get_edges(st.landfrac, st.icefrac)
base = orc.get_dist(coast, st.landfrac, st.lon, st.lat, maxdist=700.0)
ctx.set_search_radius_hint(16)
p = synth.pressure_3d(st, nz, dt)
so = [np.zeros((ny, nx), dt) for _ in range(4)]; sh = [np.zeros((ny, nx), dt) for _ in range(4)]
flipped = np.where(np.abs(base) < 12000.0, -base, base)
seq = [(base, 180.0), (base, 180.0), (np.roll(base, 7, axis=1), 180.0), (np.roll(base, 7, axis=1), 180.0),
       (flipped, 180.0), (flipped, 180.0), (base, 300.0), (base, 300.0), (base, 180.0), (base, 180.0)]
for tn, (cd, maxdist) in enumerate(seq, start=1):
    th = synth.theta_step(st, tn, dt); u, v = synth.wind_step(st, nz, tn, dt)
    cdm = cd.copy(); cdm[np.abs(cdm) > maxdist] = 12000.0
    orc.seabreeze_diag(7200.0, tn, p, u, v, th, cdm, st.z, st.sigma, *so, halo=0, bnd=1)
    ctx.seabreeze_diag(7200.0, tn, p, u, v, th, cdm, st.z, st.sigma, *sh, halo=0, bnd=hip.SB_BND_GLOBAL)
    for a, b, nm in zip(sh, so, ("ws", "wd", "thc", "sb_con")):
        bad = ~(np.isclose(a, b, rtol=1e-7, atol=1e-9) | (np.isnan(a) & np.isnan(b)))
        if bad.any():
            ij = np.argwhere(bad)
            band = np.abs(cdm) <= 180.0
            print(f"call {tn} {nm}: {bad.sum()} bad cells, rows {ij[:,0].min()}..{ij[:,0].max()} cols {ij[:,1].min()}..{ij[:,1].max()}; in band: {band[bad].sum()}; first: {[(int(i),int(j),float(a[i,j]),float(b[i,j])) for i,j in ij[:4]]}")
            sh[:] = [x.copy() for x in so]          # resync the state
    print("call", tn, "done", ctx.last_counters(), flush=True)
