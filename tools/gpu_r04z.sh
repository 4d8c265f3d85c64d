#!/bin/bash
root=${GRAFT_REPO_ROOT:-$(pwd)}
cd $root
export ESVO_BENCH_STREAM_CACHE=/tmp/esvo_streams
( timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_edge.py tests/test_gpu_fullsize.py tests/test_gpu_shard.py -m gpu -x -q ) 2>&1 | tail -3
for i in 1 2; do
timeout 300 python tools/sustained_probe.py 1600 new_$i 2>/dev/null | tail -1
done
