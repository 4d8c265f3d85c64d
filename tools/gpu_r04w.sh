#!/bin/bash
root=${GRAFT_REPO_ROOT:-$(pwd)}
cd $root
export ESVO_BENCH_STREAM_CACHE=/tmp/esvo_streams
( timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_edge.py tests/test_gpu_ref.py -m gpu -x -q ) 2>&1 | tail -3
python tools/small_tick.py dsec640x480 10000 40 2>/dev/null | tail -1
python tools/small_tick.py dsec640x480 10000 40 pipelined 2>/dev/null | tail -1
python tools/small_tick.py upenn346x260 1000 40 2>/dev/null | tail -1
python tools/small_tick.py upenn346x260 1000 40 pipelined 2>/dev/null | tail -1
