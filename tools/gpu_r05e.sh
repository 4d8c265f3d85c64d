#!/bin/bash
root=${GRAFT_REPO_ROOT:-$(pwd)}
cd $root
export ESVO_BENCH_STREAM_CACHE=/tmp/esvo_streams
for i in 1 2; do
timeout 300 python tools/regime_probe.py 1000 free_$i 2>/dev/null | tail -1
ESVO_FRONT_THROTTLE=1 timeout 300 python tools/regime_probe.py 1000 throttle_$i 2>/dev/null | tail -1
done
timeout 300 python tools/sustained_probe.py 1600 free_s 2>/dev/null | tail -1
ESVO_FRONT_THROTTLE=1 timeout 300 python tools/sustained_probe.py 1600 throttle_s 2>/dev/null | tail -1
