#!/bin/bash
# The bench variants and band-step logs recorded per round (run through gpurun from the repo root):
#   bash tools/record_round.sh <tag> [variant ...]   -> gpurun_out/<tag>_*.json / *.log; copy what is judged into profiles/
# Variants: n512_f64 f32 static band n2560_f32 (default: all).  A variant whose bench line says "parity": {"ok": false}
# -- or carries no parity object -- ENDS THE SCRIPT with exit code 4 and a line on stderr: a red record is never left
# lying about as if it were a measurement.
set -o pipefail
TAG=${1:-r04}
shift
WANT=${*:-n512_f64 f32 static band n2560_f32}
O=gpurun_out
mkdir -p $O
run() {
    name=$1; shift
    python bench.py "$@" --cpu-budget 5 2>$O/${TAG}_$name.err | grep "^{" > $O/${TAG}_bench_$name.json
    rc=$?
    python tools/kms.py $O/${TAG}_bench_$name.json
    python - $O/${TAG}_bench_$name.json <<'PY' || { echo "[record_round] PARITY RED in variant $name (bench exit $rc): $O/${TAG}_bench_$name.json" >&2; exit 4; }
import json, sys
d = json.loads(open(sys.argv[1]).read().splitlines()[-1])
sys.exit(0 if (d.get("parity") or {}).get("ok") is True else 1)
PY
    [ $rc -eq 0 ] || { echo "[record_round] bench.py exit $rc in variant $name" >&2; exit 2; }
}
for v in $WANT; do
    case $v in
    n512_f64) run n512_f64 --nx 1024 --ny 768 --steps 50 --warmup 5 ;;
    f32) run f32 --dtype f32 --steps 50 --warmup 5 ;;
    static) run static --static-sigma --steps 50 --warmup 5 ;;
    band)
        timeout -k 10 200 python tools/band_step_cost.py > $O/${TAG}_band240.log 2>&1 || exit 3
        timeout -k 10 200 python tools/band_step_cost.py 2560 480 56 16 200 > $O/${TAG}_band480.log 2>&1 || exit 3
        timeout -k 10 200 python tools/band_step_cost.py 1024 768 56 16 200 > $O/${TAG}_band_n512.log 2>&1 || exit 3
        tail -n 3 $O/${TAG}_band240.log ;;
    n2560_f32) run n2560_f32 --nx 5120 --ny 3840 --dtype f32 --steps 20 --warmup 3 ;;
    *) echo "unknown variant $v" >&2; exit 1 ;;
    esac
done
