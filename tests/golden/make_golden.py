"""Generate tests/golden/*.npz from the reference's own Fortran.

Runs ONLY where /root/reference exists (this build container): it calls the reference's
two wrapper files compiled UNMODIFIED into oracle/_ref/libsb_ref_r{4,8}.so by
`make -C oracle ref` (recipe: oracle/Makefile).  The fixtures hold
inputs and the reference's outputs only -- no reference source text.

    python tests/golden/make_golden.py

Inputs are stored as float32-representable values so one input set serves both the fp32
and the fp64 build of the reference.
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)

from oracle.pyoracle import Reference  # noqa: E402
from seabreeze_param_amd import synth  # noqa: E402


def f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def diag_case(name, nx, ny, nz, steps, maxdist, timestep_min):
    """get_edges -> get_dist -> diag for `steps` steps, both precisions."""
    st = synth.static_fields(nx, ny, np.float32, fractional_coast=True)
    inp = dict(lsm=f32(st.landfrac), ci=f32(st.icefrac), z=f32(st.z), std=f32(st.sigma),
               lon=f32(st.lon), lat=f32(st.lat), p=f32(synth.pressure_1d(nz)),
               maxdist=np.float32(maxdist), timestep=np.float32(timestep_min))
    theta = np.stack([f32(synth.theta_step(st, t)) for t in range(1, steps + 1)])
    uv = [synth.wind_step(st, nz, t) for t in range(1, steps + 1)]
    inp["theta"] = theta
    inp["u"] = np.stack([f32(a[0]) for a in uv])
    inp["v"] = np.stack([f32(a[1]) for a in uv])
    out = {}
    for prec in (8, 4):
        R = Reference(prec)
        dt = R.dt
        coast = R.get_edges(inp["lsm"], inp["ci"])
        cdist = R.get_dist(coast, inp["lsm"], inp["lon"], inp["lat"], maxdist=maxdist)
        sm = R.sigmoid(inp["std"])
        ws = np.zeros((ny, nx), dt); wd = ws.copy(); thc = ws.copy()
        outs, thcs = [], []
        for t in range(steps):
            o = np.full((4, ny, nx), -777.0, dt)          # last row must stay untouched
            R.diag(t + 1, inp["p"], inp["z"], inp["std"], inp["theta"][t], inp["v"][t], inp["u"][t], cdist,
                   ws, wd, thc, output=o, maxdist=maxdist, timestep=timestep_min)
            outs.append(o.copy()); thcs.append(thc.copy())
        s = f"r{prec}"
        out[f"coast_{s}"] = coast
        out[f"cdist_{s}"] = cdist
        out[f"sigmoid_{s}"] = sm
        out[f"output_{s}"] = np.stack(outs)
        out[f"thc_{s}"] = np.stack(thcs)
        out[f"ws_final_{s}"] = ws
        out[f"wd_final_{s}"] = wd
    path = os.path.join(HERE, f"{name}.npz")
    np.savez_compressed(path, **inp, **out)
    band = np.abs(out["cdist_r8"]) <= maxdist
    print(f"{name}: {os.path.getsize(path) / 1e6:.2f} MB, coast cells {int(out['coast_r8'].sum())}, "
          f"band fraction {band.mean():.3f}, triggers in last step "
          f"{int(((out['output_r8'][-1, 0, :-1] != 0) & (out['output_r8'][-1, 0, :-1] < 1e19)).sum())}")


def coast_case(name, nx, ny, maxdist):
    """get_edges + get_dist only, on a grid fine enough for a multi-cell search window."""
    st = synth.static_fields(nx, ny, np.float32, fractional_coast=True)
    inp = dict(lsm=f32(st.landfrac), ci=f32(st.icefrac), lon=f32(st.lon), lat=f32(st.lat),
               maxdist=np.float32(maxdist))
    out = {}
    for prec in (8, 4):
        R = Reference(prec)
        coast = R.get_edges(inp["lsm"], inp["ci"])
        out[f"coast_r{prec}"] = coast
        out[f"cdist_r{prec}"] = R.get_dist(coast, inp["lsm"], inp["lon"], inp["lat"], maxdist=maxdist)
    path = os.path.join(HERE, f"{name}.npz")
    np.savez_compressed(path, **inp, **out)
    print(f"{name}: {os.path.getsize(path) / 1e6:.2f} MB, coast cells {int(out['coast_r8'].sum())}")


if __name__ == "__main__":
    # 96x72 with a 1000 km band: window half-width k=3, search radii up to 5, refresh at step 4
    diag_case("diag_96x72", 96, 72, 2, 5, maxdist=1000.0, timestep_min=90.0)
    # the reference's own default tunables on the same grid (k=0: band = coast cells only)
    diag_case("diag_96x72_default", 96, 72, 2, 2, maxdist=180.0, timestep_min=24.0)
    coast_case("coast_256x192", 256, 192, maxdist=180.0)
