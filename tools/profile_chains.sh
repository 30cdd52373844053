#!/bin/bash
# PMC passes on the interleaved-chain recurrence (256 x 5 s in one call): gpurun -- 'bash tools/profile_chains.sh r04'
set -eo pipefail
TAG=${1:-r04}
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/prof_chains_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
i=0
for set in "TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_LATENCY_sum TA_BUSY_avr" \
           "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA" \
           "FETCH_SIZE" "WRITE_SIZE TCC_HIT_sum TCC_MISS_sum"; do
    i=$((i+1))
    rocprofv3 --pmc $set --kernel-trace -d $OUT/pmc$i -o pass -- python3 $REPO/tools/pmc_probe.py 430 256 > $OUT/pmc$i.log 2>&1
    echo "pmc pass $i done: $set"
done
python3 $REPO/tools/pmc_collect.py $OUT/${TAG}_pmc_per_launch_batch256.csv $(find $OUT/pmc* -name '*.db' | sort)
