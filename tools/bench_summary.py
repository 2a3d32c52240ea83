"""Print a one-line summary of a bench.py JSON line read from stdin (optionally prefixed by argv[1])."""
import json
import sys

tag = sys.argv[1] if len(sys.argv) > 1 else ""
line = [ln for ln in sys.stdin.read().splitlines() if ln.startswith("{")][-1]
d = json.loads(line)
r = d["roofline"]
km = {k: round(v * 1e3, 1) for k, v in r["kernel_ms"].items()}
print(tag, f"step {d['ms_per_step'] * 1e3:.1f} us  {d['value'] / 1e9:.2f} Gpts/s  whole-call {r['whole_call']['frac'] * 100:.1f}% of 8 TB/s",
      "kernels(us)", km, f"dominant {r['kernel']} {r['achieved']:.0f} GB/s")
