#!/bin/bash
root=${GRAFT_REPO_ROOT:-$(pwd)}
cd $root
export ESVO_BENCH_STREAM_CACHE=/tmp/esvo_streams
python tools/sustained_probe.py 600 back_waves7 2>/dev/null | tail -1
ESVO_HIP_LIB=$root/tools/ab/libesvo_hip_bw6.so python tools/sustained_probe.py 600 back_waves6 2>/dev/null | tail -1
ESVO_LM_QUEUES=2 ESVO_LM_QUEUES_MAX_EVENTS=100000000 python tools/sustained_probe.py 600 bw7_two_lm 2>/dev/null | tail -1
python tools/sustained_probe.py 600 back_waves7_again 2>/dev/null | tail -1
