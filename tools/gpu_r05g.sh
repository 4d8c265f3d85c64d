#!/bin/bash
root=${GRAFT_REPO_ROOT:-$(pwd)}
cd $root
export ESVO_BENCH_STREAM_CACHE=/tmp/esvo_streams
for i in 1 2; do
timeout 300 python tools/sustained_probe.py 1600 base_$i 2>/dev/null | tail -1
ESVO_HIP_LIB=$root/tools/ab/libesvo_hip_bprio3.so timeout 300 python tools/sustained_probe.py 1600 backprio3_$i 2>/dev/null | tail -1
done
