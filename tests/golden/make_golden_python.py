"""Capture a golden for the PYTHON surface from the reference itself.

Runs only in the build container: imports the reference's own ``seabreezediag`` package
from /root/reference/python_wrapper on top of the reference's genuine f2py extension
(`make -C oracle ref_f2py`), calls ``seabreezediag.diag`` on small 4-D inputs and stores
inputs + outputs in tests/golden/python_surface_96x72.npz.  Only data is stored.

    python tests/golden/make_golden_python.py

The last latitude row of every output is never written by the reference kernel and holds
whatever f2py allocated; it is stored as NaN.
"""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
PYREF = os.path.join(ROOT, "oracle", "_ref", "pyref")
REFPKG = "/root/reference/python_wrapper"

CHILD = r'''
import sys, warnings
import numpy as np
sys.path.insert(0, {root!r})
from seabreeze_param_amd import synth
sys.path.insert(0, {refpkg!r}); sys.path.insert(0, {pyref!r})
import seabreezediag as sbd                      # the reference's package
assert sbd.__file__.startswith({refpkg!r}), sbd.__file__

nx, ny, nps, nt = 96, 72, 2, 4
st = synth.static_fields(nx, ny, np.float32, fractional_coast=True)
f32 = lambda a: np.ascontiguousarray(a, dtype=np.float32)
lsm, z, std, lon, lat = f32(st.landfrac), f32(st.z), f32(st.sigma), f32(st.lon), f32(st.lat)
pres = f32(synth.pressure_1d(nps))
t = np.stack([f32(synth.theta_step(st, k)) for k in range(1, 2 * nt + 1)])
uv = [synth.wind_step(st, nps, k) for k in range(1, 2 * nt + 1)]
u = np.stack([f32(a[0]) for a in uv]); v = np.stack([f32(a[1]) for a in uv])
ci = np.stack([f32(st.icefrac)] * (2 * nt))
kw = dict(timestep=90.0, maxdist=1000.0)
with warnings.catch_warnings():
    warnings.simplefilter("ignore")
    tt1, sb1, thc1, ws1, wd1 = sbd.diag(1, lsm, z, std, lon, lat, pres, u[:nt], v[:nt], t[:nt], ci[:nt], **kw)
    tt2, sb2, thc2, ws2, wd2 = sbd.diag(tt1, lsm, z, std, lon, lat, pres, u[nt:], v[nt:], t[nt:], ci[nt:],
                                        ws=ws1, wd=wd1, thc=thc1, **kw)
out = dict(tt1=tt1, sb1=sb1, thc1=thc1, ws1=ws1, wd1=wd1, tt2=tt2, sb2=sb2, thc2=thc2, ws2=ws2, wd2=wd2)
for k, a in out.items():
    if isinstance(a, np.ndarray):
        a = np.array(a, copy=True)
        a[..., -1, :] = np.nan                   # never written by the kernel
        out[k] = a
assert sb1.shape == (nt, ny, nx) and sb1.dtype == np.float64 and ws1.shape == (ny, nx)
np.savez_compressed({path!r}, lsm=lsm, z=z, std=std, lon=lon, lat=lat, pres=pres, u=u, v=v, t=t, ci=ci,
                    timestep=np.float32(90.0), maxdist=np.float32(1000.0), **out)
print("python-surface golden:", {path!r}, "tt", tt1, tt2, "triggers", int((np.nan_to_num(sb2[-1]) != 0).sum()))
'''

if __name__ == "__main__":
    path = os.path.join(HERE, "python_surface_96x72.npz")
    code = CHILD.format(root=ROOT, refpkg=REFPKG, pyref=PYREF, path=path)
    env = dict(os.environ)
    env.pop("PYTHONPATH", None)
    subprocess.run([sys.executable, "-c", code], check=True, env=env, cwd="/tmp")
    print(f"{os.path.getsize(path) / 1e6:.2f} MB")
