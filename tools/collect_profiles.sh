#!/bin/bash
# Collect the judged profile artefacts on the GPU box (run through gpurun from the repo root):
#   tools/collect_profiles.sh <tag> [bench.py arguments of the workload, e.g. --nx 5120 --ny 3840 --dtype f32]
# 1. rocprofv3 --kernel-trace --stats of the default bench command (--no-replan: the averages are the stored-plan state's;
#    the un-profiled line of record at the end carries roofline.replan)  -> gpurun_out/<tag>_kernel_stats.csv
# 2. PMC passes (FETCH_SIZE, WRITE_SIZE, two sets of SQ counters), each in its own run -> gpurun_out/<tag>_pmc_*.txt
#    (the files are named after the first counter of the pass: ..._pmc_SQ_WAVES.txt, ..._pmc_SQ_INSTS_SALU.txt)
# Copy the files from gpurun_out/ into profiles/ afterwards (gpurun_out is scratch).
set -o pipefail
TAG=${1:-r01}
shift
WL="$*"                      # the workload's bench.py arguments (none: the headline, configs[2])
OUT=gpurun_out
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p $OUT
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${TAG}_trace -- \
    python bench.py $WL --steps 50 --warmup 5 --cpu-budget 5 --no-replan > $OUT/${TAG}_bench.log 2>&1 || exit 2
cp $OUT/${TAG}_trace/*/*kernel_stats.csv $OUT/${TAG}_kernel_stats.csv
grep '^{' $OUT/${TAG}_bench.log > $OUT/${TAG}_bench_under_rocprof.json
B="python bench.py $WL --steps 10 --warmup 2 --no-cpu-baseline --profile-passes 0 --no-replan"
for pass in "FETCH_SIZE" "WRITE_SIZE" \
            "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU" \
            "SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS"; do
    name=$(echo $pass | cut -d' ' -f1)
    timeout -k 10 250 rocprofv3 --pmc $pass --kernel-trace --output-format csv -d $OUT/${TAG}_pmc_$name -- $B \
        > $OUT/${TAG}_pmc_$name.log 2>&1 || exit 3
    python tools/pmc_summary.py $OUT/${TAG}_pmc_$name/*/*counter_collection.csv > $OUT/${TAG}_pmc_$name.txt
done
# un-profiled bench line (the number of record)
timeout -k 10 400 python bench.py $WL --steps 50 --warmup 5 --cpu-budget 5 2>/dev/null | grep '^{' > $OUT/${TAG}_bench.json
python tools/bench_summary.py $TAG < $OUT/${TAG}_bench.json
cat $OUT/${TAG}_pmc_FETCH_SIZE.txt $OUT/${TAG}_pmc_WRITE_SIZE.txt | grep -E "k_scan|k_wind|k_prep|k_thc|k_strip"
