#!/bin/bash
# Diagnostic: run bench.py once per environment setting given as arguments ("A=1 B=2" ...)
# and print step time and per-kernel event times.   tools/knob_sweep.sh "SB_SCAN_SPT=2" "SB_SCAN_SPT=3 SB_SCAN_WGS=1"
for setting in "$@"; do
  env $setting python bench.py --no-cpu-baseline --steps 40 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.readlines()[-1])
print('$setting', 'step_us', round(d['ms_per_step']*1e3,1), {k: round(v*1e3,1) for k,v in d['roofline']['kernel_ms'].items()})
"
done
