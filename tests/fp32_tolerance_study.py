"""fp32 tolerance study (BASELINE.json configs[3]: "bandwidth-bound scaling + tolerance study").

For every output field of seabreeze_diag (generic flavour) the error of
    (a) the HIP path in single precision,
    (b) the reference's arithmetic in single precision (the CPU oracle, liboracle_r4: sequential fp32 window sums),
both against the reference's arithmetic in double precision on the same (fp32-representable) inputs, at N512, N1280
and -- with --big -- N2560 (5120x3840, windows of up to 31 cells).  The oracle is the checker here, never the product.

    python tests/fp32_tolerance_study.py [--big] [--out gpurun_out/fp32_tolerance.json]

What to expect, and why: the reference re-sums a (2nn+1)^2 window of temperatures near 290 K sequentially in the
working precision (generic/sea_breeze_diag.f90:192-208); in fp32 the running sum of a 33 x 33 window reaches 3e5, where
one bit is 0.03 K, so its window means carry 1e-4 .. 1e-2 K of rounding noise that grows with the window.  The HIP path
keeps its summed-area tables in fp64 whatever the working precision, so its fp32 error is what rounding t0 and the
result to fp32 costs (about 2e-5 K).  Fields that pass through no window sum (windspeed, winddir) agree to fp32
rounding on both sides.
"""
import argparse
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))   # the repository root
from oracle.pyoracle import Oracle  # noqa: E402  (checker)
from seabreeze_param_amd import hip, synth  # noqa: E402


def stats(a, truth, mask):
    d = np.abs(a.astype(np.float64) - truth)[mask]
    return {"max_abs": float(d.max()), "p99_abs": float(np.quantile(d, 0.99)), "rms": float(np.sqrt((d ** 2).mean()))}


def study(ctx, nx, ny, nz, steps=(1, 2, 15)):
    os.environ.setdefault("OMP_NUM_THREADS", str(min(16, len(os.sched_getaffinity(0)))))
    o4, o8 = Oracle(4, omp=True), Oracle(8, omp=True)
    f4 = np.float32
    st = synth.static_fields(nx, ny, f4)
    coast = ctx.get_edges(st.landfrac, st.icefrac)
    cdist = ctx.get_dist(coast, st.landfrac, st.lon, st.lat)
    p = synth.pressure_3d(st, nz, f4)
    f8 = lambda a: np.ascontiguousarray(a, dtype=np.float64)
    sh = [np.zeros((ny, nx), f4) for _ in range(4)]
    s4 = [np.zeros((ny, nx), f4) for _ in range(4)]
    s8 = [np.zeros((ny, nx), np.float64) for _ in range(4)]
    band = np.abs(cdist) <= 180.0
    res = {"grid": f"{nx}x{ny}x{nz}", "band_cells": int(band.sum()), "search_halo": int(hip.dist_window(st.lon, st.lat)) + 1, "steps": {}}
    for tn in steps:
        th = synth.theta_step(st, tn, f4)
        u, v = synth.wind_step(st, nz, tn, f4)
        ctx.seabreeze_diag(1440.0, tn, p, u, v, th, cdist, st.z, st.sigma, *sh, halo=0, bnd=hip.SB_BND_GLOBAL)
        o4.seabreeze_diag(1440.0, tn, p, u, v, th, cdist, st.z, st.sigma, *s4, halo=0, bnd=1, omp=True)
        o8.seabreeze_diag(1440.0, tn, f8(p), f8(u), f8(v), f8(th), f8(cdist), f8(st.z), f8(st.sigma), *s8, halo=0, bnd=1, omp=True)
        row = {}
        for k, nm in enumerate(("windspeed", "winddir", "thc", "sb_con")):
            row[nm] = {"hip_fp32": stats(sh[k], s8[k], band), "reference_fp32": stats(s4[k], s8[k], band)}
        # triggers that differ from the fp64 pattern (knife-edge decisions at 0.75 K, 5 m/s, 11 m/s, 90 deg)
        row["trigger_flips"] = {"hip_fp32": int(((sh[3] != 0) != (s8[3] != 0)).sum()),
                                "reference_fp32": int(((s4[3] != 0) != (s8[3] != 0)).sum()),
                                "triggered_cells_fp64": int((s8[3] != 0).sum())}
        res["steps"][str(tn)] = row
    return res


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--big", action="store_true", help="also the 5120x3840 grid of configs[3]")
    ap.add_argument("--out", default="gpurun_out/fp32_tolerance.json")
    args = ap.parse_args()
    ctx = hip.Context()
    grids = [(1024, 768, 8), (2560, 1920, 8)] + ([(5120, 3840, 4)] if args.big else [])
    out = {"what": "max / 99th percentile / rms absolute error over coastal-band cells against the fp64 reference arithmetic",
           "grids": [study(ctx, *g) for g in grids]}
    os.makedirs(os.path.dirname(args.out) or ".", exist_ok=True)
    json.dump(out, open(args.out, "w"), indent=1)
    for g in out["grids"]:
        print(g["grid"], "band cells", g["band_cells"], "halo", g["search_halo"])
        for tn, row in g["steps"].items():
            for nm in ("windspeed", "winddir", "thc", "sb_con"):
                h, r = row[nm]["hip_fp32"], row[nm]["reference_fp32"]
                print(f"  tn={tn:>2} {nm:9s} HIP fp32 max {h['max_abs']:.2e} p99 {h['p99_abs']:.2e} | reference fp32 max {r['max_abs']:.2e} p99 {r['p99_abs']:.2e}")
            print(f"  tn={tn:>2} trigger flips vs fp64: HIP {row['trigger_flips']['hip_fp32']}, reference fp32 {row['trigger_flips']['reference_fp32']} "
                  f"of {row['trigger_flips']['triggered_cells_fp64']} triggered cells")
    ctx.close()


if __name__ == "__main__":
    main()
