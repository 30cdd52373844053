#!/bin/bash
# A/B of the batched GEMM kernels in one box: per-dispatch durations from rocprofv3 --kernel-trace
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for v in lds nolds; do
  if [ $v = nolds ]; then export BVC_NO_LDS_GEMM=1; fi
  timeout -k 10 300 rocprofv3 --kernel-trace -f csv -d gpurun_out/bt_$v -o bt -- python3 tools/batched_trace.py > gpurun_out/bt_$v.log 2>&1 || exit 1
  echo "== $v"; grep -E "gemm_batched" gpurun_out/bt_$v/bt_kernel_trace.csv | python3 -c "
import sys,csv
for row in csv.reader(sys.stdin):
    nm=[c for c in row if 'bvc::' in c][0][10:40]
    nums=[int(c) for c in row if c.isdigit() and len(c)>12]
    print(nm, (nums[1]-nums[0])/1e3)
" | tail -6
  rm -rf gpurun_out/bt_$v
done
