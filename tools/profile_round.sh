#!/bin/bash
# Profiles of the round, run on the GPU box from the repository root (gpurun -- 'bash tools/profile_round.sh r03'):
#   kernel-trace statistics of the default bench.py workload (headline leg only: one batch at a time, no multi_stream leg), then separate --pmc passes (counters serialise the dispatches,
#   so they never share a run with the timing) on tools/pmc_probe.py = one 64 x 5 s encode + decode.
# Results land in gpurun_out/; copy the summaries you want judged into profiles/.
set -eo pipefail
TAG=${1:-r04}
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/prof_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o default -- python3 $REPO/bench.py --steps 10 --warmup 2 --multi-streams 0 --no-cpu-baseline --no-parity --no-extra > $OUT/bench_under_rocprof.json 2> $OUT/bench_under_rocprof.err
cp $(find $OUT/stats -name '*kernel_stats.csv' | head -1) $OUT/${TAG}_kernel_stats_default.csv
echo "stats done"
i=0
for set in "FETCH_SIZE" "WRITE_SIZE TCC_HIT_sum TCC_MISS_sum" \
           "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA" \
           "TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_LATENCY_sum TA_BUSY_avr" \
           "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_INSTS_SMEM"; do
    i=$((i+1))
    rocprofv3 --pmc $set --kernel-trace -d $OUT/pmc$i -o pass -- python3 $REPO/tools/pmc_probe.py 430 > $OUT/pmc$i.log 2>&1
    echo "pmc pass $i done: $set"
done
python3 $REPO/tools/pmc_collect.py $OUT/${TAG}_pmc_per_launch.csv $(find $OUT/pmc* -name '*.db' | sort)
