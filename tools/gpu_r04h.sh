#!/bin/bash
root=${GRAFT_REPO_ROOT:-$(pwd)}
cd $root
export ESVO_BENCH_STREAM_CACHE=/tmp/esvo_streams
ESVO_HIP_LIB=$root/tools/ab/libesvo_hip_fstats.so ESVO_FUSE_STATS=1 python tools/standalone_kernels.py dsec640x480 8 2>&1 | tail -3
ESVO_HIP_LIB=$root/tools/ab/libesvo_hip_fstats.so ESVO_FUSE_STATS=1 python tools/standalone_kernels.py upenn346x260 8 2>&1 | tail -3
