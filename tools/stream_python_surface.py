"""BASELINE configs[4] rehearsal: stream timesteps through the Python surface (`seabreezediag.diag` on the
f2py extension `seabreeze`, both from python_wrapper/) on one GPU and report steps per second.

    python tools/stream_python_surface.py [nlon nlat nlev nsteps chunk]

The fields are synthetic (seabreeze_param_amd.synth), fp32 like the shipped f2py surface; `ci` changes once
per day of model time, so the coast distance is recomputed only then (the reference recomputes it every
step with the same result).  Steps are fed in chunks of `chunk` time slices per diag() call, state threaded
through like the reference's test_run.py does (ref: python_wrapper/test_run.py:23-57).
"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "python_wrapper"))

from seabreeze_param_amd import synth  # noqa: E402
import seabreezediag as sbd  # noqa: E402

nlon, nlat, nlev, nsteps, chunk = (int(a) for a in sys.argv[1:6]) if len(sys.argv) >= 6 else (1024, 768, 8, 200, 20)
dt = np.float32
st = synth.static_fields(nlon, nlat, dt)
pres = (synth.pressure_1d(nlev, dt) / 100.0).astype(dt)          # hPa at this surface
lsm, z, std = st.landfrac.astype(dt), st.z.astype(dt), st.sigma.astype(dt)
# one chunk of time-varying inputs, reused for every chunk (content does not matter for the rate)
u = np.stack([synth.wind_step(st, nlev, t, dt)[0] for t in range(1, chunk + 1)])
v = np.stack([synth.wind_step(st, nlev, t, dt)[1] for t in range(1, chunk + 1)])
th = np.stack([synth.theta_step(st, t, dt) for t in range(1, chunk + 1)])
ice0 = st.icefrac.astype(dt)
tt, ws, wd, thc = 1, None, None, None
t0 = time.perf_counter()
done = 0
while done < nsteps:
    day = done // 60                                             # 24-minute steps: 60 per day
    ci = np.broadcast_to(np.clip(ice0 + 0.01 * (day % 3), 0, 1).astype(dt), (chunk, nlat, nlon))
    kw = {} if ws is None else dict(ws=ws, wd=wd, thc=thc)
    tt, sb, thc, ws, wd = sbd.diag(tt, lsm, z, std, st.lon.astype(dt), st.lat.astype(dt), pres, u, v, th, ci, **kw)
    done += chunk
el = time.perf_counter() - t0
trig = int(np.count_nonzero((sb[-1] != 0) & (sb[-1] < 1e19)))
print(f"{done} steps of {nlon}x{nlat}x{nlev} fp32 through seabreezediag.diag: {el:.2f} s = {done / el:.1f} steps/s "
      f"({el / done * 1e3:.2f} ms per step, {nlon * nlat * done / el / 1e6:.1f} M grid-points/s incl. host copies); "
      f"triggered cells in the last step: {trig}")
