# instruction-cache and instruction-wait counters of the diag kernels: bash tools/pmc_icache.sh TAG   (on the GPU box)
set -o pipefail
TAG=$1
OUT=gpurun_out
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
B="python bench.py --steps 10 --warmup 2 --no-cpu-baseline --profile-passes 0"
for pass in "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE" \
            "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_IFETCH SQ_IFETCH_LEVEL"; do
    name=$(echo $pass | cut -d' ' -f1)
    timeout -k 10 250 rocprofv3 --pmc $pass --kernel-trace --output-format csv -d $OUT/${TAG}_pmc_$name -- $B > $OUT/${TAG}_pmc_$name.log 2>&1 || exit 3
    python tools/pmc_summary.py $OUT/${TAG}_pmc_$name/*/*counter_collection.csv > $OUT/${TAG}_pmc_$name.txt
    grep "k_strip\|k_scan\|k_wind" $OUT/${TAG}_pmc_$name.txt
done
