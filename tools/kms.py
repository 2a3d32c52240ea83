"""Print step time and per-kernel event times of a bench.py JSON file (argv[1])."""
import json
import sys

d = json.loads([ln for ln in open(sys.argv[1]).read().splitlines() if ln.startswith("{")][-1])
r = d["roofline"]
print(f"step {d['ms_per_step'] * 1e3:.1f} us (median {d['median_ms_per_step'] * 1e3:.1f})  whole-call {r['whole_call']['frac']:.3f}",
      {k: round(v * 1e3, 1) for k, v in r["kernel_ms"].items()}, "parity", (d.get("parity") or {}).get("ok"))
