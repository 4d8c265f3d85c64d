#!/bin/bash
# usage (GPU box): bash tools/ab_env_sustained.sh <rounds> <ticks> "<ENV=VAL ...>" ["<ENV=VAL ...>" ...]
# interleaved SUSTAINED runs (tools/sustained_probe.py) of the current library under different environments ("X=1" = no switch)
export ESVO_DEV_SWITCHES=1   # the library reads its A/B switches only with this set
root=${GRAFT_REPO_ROOT:-$(pwd)}
rounds=$1; ticks=$2; shift; shift
export ESVO_BENCH_STREAM_CACHE=/tmp/esvo_streams
for r in $(seq 1 $rounds); do
  for e in "$@"; do
    env $e python $root/tools/sustained_probe.py $ticks "[$e] r$r" 2>/dev/null | tail -1
  done
done
