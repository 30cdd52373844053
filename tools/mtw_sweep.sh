#!/bin/bash
# throughput of bench.py for BVC_MTW x streams (run on the GPU box)
for M in 1 2 4; do for S in 1 3 4 6; do
  BVC_MTW=$M timeout -k 10 200 python bench.py --streams $S --steps 12 --warmup 6 --no-cpu-baseline --no-roofline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('mtw $M streams', d['config']['streams'], d['value'], d['ms_per_step'])"
done; done
