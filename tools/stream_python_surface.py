"""BASELINE configs[4] rehearsal: stream timesteps through the Python surface (`seabreezediag.diag` on the
f2py extension `seabreeze`, both from python_wrapper/) on one GPU and report steps per second.

    python tools/stream_python_surface.py [nlon nlat nlev nsteps chunk [out.json]]

With out.json the figures are also written there (the recorded 10^4-step run of profiles/ is this tool's output:
1024 768 8 10000 50).  The split of the wall time per step -- host copies into the pinned staging buffers,
enqueueing, waiting for the device -- is the extension's own account (seabreezediag.stream_stats).

The fields are synthetic (seabreeze_param_amd.synth), fp32 like the shipped f2py surface; `ci` changes once
per day of model time, so the coast distance is recomputed only then (the reference recomputes it every
step with the same result).  Steps are fed in chunks of `chunk` time slices per diag() call, state threaded
through like the reference's test_run.py does (ref: python_wrapper/test_run.py:23-57).
"""
import json
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
out_json = sys.argv[6] if len(sys.argv) >= 7 else None
tt, ws, wd, thc = 1, None, None, None
result = np.empty((chunk, nlat, nlon))                           # one result array for every chunk (diag's out=)
t0 = time.perf_counter()
done = 0
split = np.zeros(3)
streamed = 0
while done < nsteps:
    day = done // 60                                             # 24-minute steps: 60 per day
    ci = np.broadcast_to(np.clip(ice0 + 0.01 * (day % 3), 0, 1).astype(dt), (chunk, nlat, nlon))
    kw = {} if ws is None else dict(ws=ws, wd=wd, thc=thc)
    tt, sb, thc, ws, wd = sbd.diag(tt, lsm, z, std, st.lon.astype(dt), st.lat.astype(dt), pres, u, v, th, ci, out=result, **kw)
    done += chunk
    n, parts = sbd.stream_stats()
    streamed += int(n)
    split += np.asarray(parts, dtype=np.float64)
    if done % 1000 == 0:
        print(f"  {done} steps, {time.perf_counter() - t0:.1f} s", flush=True)
el = time.perf_counter() - t0
trig = int(np.count_nonzero((sb[-1] != 0) & (sb[-1] < 1e19)))
print(f"{done} steps of {nlon}x{nlat}x{nlev} fp32 through seabreezediag.diag: {el:.2f} s = {done / el:.1f} steps/s "
      f"({el / done * 1e3:.2f} ms per step, {nlon * nlat * done / el / 1e6:.1f} M grid-points/s incl. host copies); "
      f"triggered cells in the last step: {trig}")
if streamed:
    c, q, w = (split / streamed * 1e3).tolist()
    print(f"streamed steps {streamed}: per step {c:.3f} ms host copies into pinned staging, {q:.3f} ms enqueueing, "
          f"{w:.3f} ms waiting for the device; the rest of {el / done * 1e3:.3f} ms is Python (slicing, dtype checks, "
          f"the coast distance on ice changes)")
if out_json:
    json.dump({"what": "BASELINE.json configs[4] rehearsal: timesteps streamed through python_wrapper/seabreezediag.diag, one GPU",
               "grid": [nlon, nlat, nlev], "dtype": "f32", "steps": done, "chunk": chunk, "seconds": el,
               "steps_per_s": done / el, "ms_per_step": el / done * 1e3,
               "grid_points_per_s": nlon * nlat * done / el,
               "streamed_steps": streamed,
               "ms_per_step_split": dict(zip(("host_copies", "enqueue", "wait_device"), (split / max(streamed, 1) * 1e3).tolist())),
               "coast_distance_recomputed": int(done // 60) + 1,
               "triggered_cells_last_step": trig}, open(out_json, "w"), indent=1)
