#!/bin/bash
# The bench variants and band-step logs recorded per round (run through gpurun from the repo root):
#   bash tools/record_round.sh <tag>     -> gpurun_out/<tag>_*.json / *.log; copy what is judged into profiles/
set -o pipefail
TAG=${1:-r03}
O=gpurun_out
run() { name=$1; shift; python bench.py "$@" --cpu-budget 5 2>$O/${TAG}_$name.err | grep "^{" > $O/${TAG}_bench_$name.json && python tools/kms.py $O/${TAG}_bench_$name.json; }
run n512_f64 --nx 1024 --ny 768 --steps 50 --warmup 5 || exit 2
run f32 --dtype f32 --steps 50 --warmup 5 || exit 2
run static --static-sigma --steps 50 --warmup 5 || exit 2
timeout -k 10 200 python tools/band_step_cost.py > $O/${TAG}_band240.log 2>&1 || exit 3
timeout -k 10 200 python tools/band_step_cost.py 2560 480 56 16 200 > $O/${TAG}_band480.log 2>&1 || exit 3
timeout -k 10 200 python tools/band_step_cost.py 1024 768 56 16 200 > $O/${TAG}_band_n512.log 2>&1 || exit 3
tail -n 3 $O/${TAG}_band240.log
run n2560_f32 --nx 5120 --ny 3840 --dtype f32 --steps 20 --warmup 3 || exit 2
