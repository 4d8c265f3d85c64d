#!/bin/bash
root=${GRAFT_REPO_ROOT:-$(pwd)}
cd $root
export ESVO_BENCH_STREAM_CACHE=/tmp/esvo_streams
for i in 1 2; do
timeout 300 python tools/sustained_probe.py 1600 base_$i 2>/dev/null | tail -1
for v in prio3 prio1; do
ESVO_HIP_LIB=$root/tools/ab/libesvo_hip_$v.so timeout 300 python tools/sustained_probe.py 1600 ${v}_$i 2>/dev/null | tail -1
done
done
