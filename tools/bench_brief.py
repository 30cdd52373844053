#!/usr/bin/env python3
"""Prints the headline fields of bench.py JSON lines read from stdin."""
import json
import sys
for line in sys.stdin:
    line = line.strip()
    if not line.startswith("{"):
        continue
    d = json.loads(line)
    ms = d.get("multi_stream", {})
    rf = d.get("roofline", {})
    ls = ms.get("layers_schedule", {})
    print(f"value {d['value']} ({d['ms_per_step']} ms/step, streams {d['config']['streams']}) | multi_stream {ms.get('value')} / layers {ls.get('value')} "
          f"({ms.get('ms_per_step')} ms) | roofline frac {rf.get('frac')} rocprof {rf.get('frac_rocprof')} | "
          f"parity {d.get('parity', {}).get('code_bits_differing')} | {d['config']['workload'][:60]}")
