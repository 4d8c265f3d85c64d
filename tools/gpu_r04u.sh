#!/bin/bash
root=${GRAFT_REPO_ROOT:-$(pwd)}
cd $root
export ESVO_BENCH_STREAM_CACHE=/tmp/esvo_streams
( timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py -m gpu -x -q -k "lm or tick or fullsize" ) 2>&1 | tail -3
for i in 1 2; do
python tools/sustained_probe.py 1600 cache_$i 2>/dev/null | tail -1
ESVO_HIP_LIB=$root/tools/ab/libesvo_hip_nocache.so python tools/sustained_probe.py 1600 nocache_$i 2>/dev/null | tail -1
done
