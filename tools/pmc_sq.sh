#!/bin/bash
# Diagnostic: SQ counters of one bench run (two PMC passes), summarised per kernel.  tools/pmc_sq.sh <tag>
set -o pipefail
TAG=${1:-sq}
OUT=gpurun_out
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
B="python bench.py --steps 10 --warmup 2 --no-cpu-baseline --profile-passes 0"
i=0
for pass in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU" \
            "SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS"; do
    i=$((i+1))
    timeout -k 10 250 rocprofv3 --pmc $pass --kernel-trace --output-format csv -d $OUT/${TAG}_pmc$i -- $B > $OUT/${TAG}_pmc$i.log 2>&1 || { tail -5 $OUT/${TAG}_pmc$i.log; continue; }
    python tools/pmc_summary.py $OUT/${TAG}_pmc$i/*/*counter_collection.csv > $OUT/${TAG}_pmc$i.txt
    grep -E "k_scan|k_wind|k_thc" $OUT/${TAG}_pmc$i.txt
done
