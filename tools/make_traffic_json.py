"""Turn the FETCH_SIZE / WRITE_SIZE summaries of tools/collect_profiles.sh into
profiles/pmc_traffic.json, which bench.py reports as `roofline.traffic`.

    python tools/make_traffic_json.py profiles/<tag>_pmc_FETCH_SIZE.txt profiles/<tag>_pmc_WRITE_SIZE.txt nx ny nz

HBM bytes per launch = (2 * FETCH_SIZE + WRITE_SIZE) * 1024: on gfx950 FETCH_SIZE tallies the
128-byte requests of a coalesced read at 64 bytes (MI355X_MICROARCH.md, HBM section), so the
read side is doubled; WRITE_SIZE reads the bytes exactly.  Checked on k_scan, whose reads are
two plain streams: 2 * 38.5 MB = the 78.6 MB of sigma + mask.
"""
import ast
import json
import re
import sys

fetch_txt, write_txt, nx, ny, nz = sys.argv[1], sys.argv[2], int(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5])


def parse(path, key):
    out = {}
    for line in open(path):
        m = re.match(r"\s+(k_\w+)<[^{]*(\{.*\}) n=", line)
        if m:
            out[m.group(1)] = ast.literal_eval(m.group(2))[key]
    return out


f, w = parse(fetch_txt, "FETCH_SIZE"), parse(write_txt, "WRITE_SIZE")
res = {"config": {"nx": nx, "ny": ny, "nz": nz, "dtype": "f64"},
       "source": [fetch_txt, write_txt],
       "kernels": {k: {"FETCH_SIZE_KB": f[k], "WRITE_SIZE_KB": w.get(k, 0.0),
                       "hbm_bytes": int((2 * f[k] + w.get(k, 0.0)) * 1024)} for k in f}}
json.dump(res, open("profiles/pmc_traffic.json", "w"), indent=1)
print(json.dumps(res["kernels"], indent=1))
