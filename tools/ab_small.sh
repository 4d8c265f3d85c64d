#!/bin/bash
# usage (GPU box): bash tools/ab_small.sh <variant> [<variant> ...]   -- reference-faithful ticks (synchronised / pipelined) per library variant
export ESVO_DEV_SWITCHES=1   # the library reads its A/B switches only with this set
root=${GRAFT_REPO_ROOT:-$(pwd)}
export ESVO_BENCH_STREAM_CACHE=/tmp/esvo_streams
for r in 1 2; do
for v in "$@"; do
  if [ "$v" = "cur" ]; then unset ESVO_HIP_LIB; else export ESVO_HIP_LIB=$root/tools/ab/libesvo_hip_$v.so; fi
  echo "== $v r$r"
  python $root/tools/small_tick.py dsec640x480 10000 40 2>/dev/null | tail -1
  python $root/tools/small_tick.py dsec640x480 10000 40 pipelined 2>/dev/null | tail -1
  python $root/tools/small_tick.py upenn346x260 1000 40 2>/dev/null | tail -1
done
done
