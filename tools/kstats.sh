#!/bin/bash
# Quick look on the GPU box: rocprofv3 kernel statistics of a short bench run, then the un-profiled bench line.
#   tools/kstats.sh <tag> [bench args]      -> gpurun_out/<tag>/kernel_stats.csv, gpurun_out/<tag>/bench.json
set -o pipefail
TAG=${1:-quick}; shift
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p $OUT
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- \
    python3 bench.py --steps 40 --warmup 5 --no-cpu-baseline "$@" > $OUT/rocprof.log 2>&1 || exit 2
cp $OUT/trace/*/*kernel_stats.csv $OUT/kernel_stats.csv && rm -rf $OUT/trace
timeout -k 10 300 python3 bench.py --steps 40 --warmup 5 --no-cpu-baseline "$@" 2>/dev/null | grep '^{' > $OUT/bench.json || exit 3
python3 -c "import csv; [print(r[\"Name\"][:36], r[\"Calls\"], r[\"AverageNs\"], r[\"MinNs\"], r[\"MaxNs\"]) for r in csv.DictReader(open(\"$OUT/kernel_stats.csv\"))]"
python3 -c "import json,sys; d=json.load(open('$OUT/bench.json')); print(d['ms_per_step'], d['median_ms_per_step'], d['roofline']['kernel_ms'])"
