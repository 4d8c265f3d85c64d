#!/bin/bash
root=${GRAFT_REPO_ROOT:-$(pwd)}
cd $root
export ESVO_BENCH_STREAM_CACHE=/tmp/esvo_streams
for i in 1 2; do
python tools/sustained_probe.py 1600 base_$i 2>/dev/null | tail -1
ESVO_HIP_LIB=$root/tools/ab/libesvo_hip_dl.so python tools/sustained_probe.py 1600 double_load_$i 2>/dev/null | tail -1
done
