#!/bin/bash
# The bench lines of a round (run on the GPU box from the repository root): gpurun -- 'bash tools/bench_round.sh r03'
set -eo pipefail
TAG=${1:-r04}
OUT=gpurun_out/bench_$TAG
mkdir -p $OUT
python bench.py > $OUT/${TAG}_a_bench_default.json 2> $OUT/a.err
echo "default done"
python bench.py --seconds 10 --multi-streams 0 --no-cpu-baseline --no-extra > $OUT/${TAG}_b_bench_64x10s_target_workload.json 2> $OUT/b.err
echo "10 s done"
python bench.py --mode encode --multi-streams 0 --no-cpu-baseline --no-extra > $OUT/${TAG}_c_bench_configs2_encode_only.json 2> $OUT/c.err
echo "encode done"
for br in 1500 6000; do
  python bench.py --seconds 10 --bitrate $br --multi-streams 0 --no-cpu-baseline --no-extra > $OUT/${TAG}_d_bench_configs3_shard_$br.json 2> $OUT/d$br.err
done
python bench.py --batch 256 --multi-streams 0 --no-cpu-baseline --no-extra > $OUT/${TAG}_e_bench_batch256_one_call.json 2> $OUT/e.err
echo "batch 256 done"
cat $OUT/*.json | python tools/bench_brief.py
