#!/bin/bash
root=${GRAFT_REPO_ROOT:-$(pwd)}
cd $root
export ESVO_BENCH_STREAM_CACHE=/tmp/esvo_streams
bash tools/small_trace.sh dsec640x480 10000 > gpurun_out/r04v_small_dsec.txt 2>&1
bash tools/small_trace.sh upenn346x260 1000 > gpurun_out/r04v_small_upenn.txt 2>&1
tail -60 gpurun_out/r04v_small_dsec.txt
