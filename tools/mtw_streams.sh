#!/bin/bash
# throughput vs (GPU_MAX_HW_QUEUES, BVC_MTW, streams): python bench.py --no-cpu-baseline --no-roofline
for cfg in "8 1 3" "8 1 4" "8 1 6" "8 2 4" "8 2 6" "2 1 3" "16 1 5"; do set -- $cfg; GPU_MAX_HW_QUEUES=$1 BVC_MTW=$2 timeout -k 10 200 python bench.py --no-cpu-baseline --no-roofline --streams $3 --steps 12 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('hwq $1 mtw $2 streams $3', d['value'], d['ms_per_step'])" || exit 1; done
