set -e
export TMPDIR=/tmp
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $REPO/gpurun_out/prof_stream -o tick -- python3 $REPO/tools/bench_streaming.py > $REPO/gpurun_out/stream_under_prof.json 2>$REPO/gpurun_out/stream_under_prof.err
cp $(find $REPO/gpurun_out/prof_stream -name '*kernel_stats.csv' | head -1) $REPO/gpurun_out/stream_tick_kernel_stats.csv
