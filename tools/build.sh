#!/bin/bash
# rebuild the HIP library from any cwd; prints only problems
cd "$(dirname "$0")/.." && python -m bvcodec.build 2>&1 | grep -iE "error|warning" ; ls -la --time-style=+%T bernoulli-var-speech-codec_amd/csrc/libbvcodec_hip.so | awk '{print "lib built at", $6}'
