#!/bin/bash
# usage (GPU box): bash tools/ab_sustained.sh <rounds> <ticks> <variant> [<variant> ...]   variant = name of tools/ab/libesvo_hip_<name>.so, or "cur"
# interleaved SUSTAINED runs (tools/sustained_probe.py) on one box
export ESVO_DEV_SWITCHES=1   # the library reads its A/B switches only with this set
root=${GRAFT_REPO_ROOT:-$(pwd)}
rounds=$1; ticks=$2; shift; shift
export ESVO_BENCH_STREAM_CACHE=/tmp/esvo_streams
for r in $(seq 1 $rounds); do
  for v in "$@"; do
    if [ "$v" = "cur" ]; then unset ESVO_HIP_LIB; else export ESVO_HIP_LIB=$root/tools/ab/libesvo_hip_$v.so; fi
    python $root/tools/sustained_probe.py $ticks "$v r$r" 2>/dev/null | tail -1
  done
done
