"""File driver: sea-breeze convergence for every input file of a configured period.

    python run_seabreeze.py --config=run.conf

The counterpart of the reference's `test_run.py` (ref: python_wrapper/test_run.py:8-57): read the configuration, the
static fields and the list of daily or monthly input files; for each file read the winds, temperature and sea ice,
run `seabreezediag.diag` with the state (timestep counter, wind speed, wind direction, heating contrast) carried from
file to file, and write `<prefix>sb_<stamp>.nc` beside the inputs.  `run.conf.example` shows the entries.
"""
from __future__ import annotations

import os
import sys

import numpy as np

import seabreezediag as sbd
from seabreezediag.configdir import Config, Meta


def main(config, verbose=True):
    cfg = Config(config)
    meta = Meta(cfg)
    shape = (len(meta.lat), len(meta.lon))
    thc, windspeed, winddir = (np.zeros(shape) for _ in range(3))
    tt = 1
    written = []
    result = None                   # one (ntime, lat, lon) result array for all files of equal length (diag's out=)
    for stamp in meta.dates:
        f_sb = meta.input_file("sb", stamp)
        if verbose:
            print(f"Creating sea-breeze data for {os.path.basename(f_sb)} ... ", end="", flush=True)
        files = {v: meta.input_file(cfg[v], stamp) for v in ("vv", "vu", "vtheta", "vci")}
        data = sbd.read_nc(files["vv"], files["vu"], files["vtheta"], files["vci"], vv=cfg.vv, vu=cfg.vu,
                           vtheta=cfg.vtheta, vci=cfg.get("vci", "ci"), vpres=cfg.vpres, vtime=cfg.vtime)
        try:
            nt = data.v.shape[0]
            if result is None or result.shape[0] != nt:
                result = np.empty((nt,) + shape)
            tt, sb_con, thc, windspeed, winddir = sbd.diag(tt, meta.landfrac, meta.z, meta.std, meta.lon, meta.lat,
                                                           data.pres, meta=data, ws=windspeed, wd=winddir, thc=thc, out=result)
            meta.create_nc(sb_con, f_sb, "sb_con", data.time)
        finally:
            for f in data.nc.values():
                f.close()
        written.append(f_sb)
        if verbose:
            print("ok", flush=True)
    return written


if __name__ == "__main__":
    conf = os.path.join(os.path.dirname(os.path.abspath(sys.argv[0])), "run.conf")
    for arg in sys.argv[1:]:
        key, _, value = arg.lstrip("-").partition("=")
        if key.lower() == "config" and value:
            conf = value
        else:
            sys.exit(__doc__)
    main(conf)
