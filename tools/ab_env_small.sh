#!/bin/bash
# usage (GPU box): bash tools/ab_env_small.sh <rounds> "<ENV=VAL ...>" ...  -- the reference-faithful ticks (synchronised / pipelined) under
# each environment ("X=1" = no switch), interleaved
export ESVO_DEV_SWITCHES=1
root=${GRAFT_REPO_ROOT:-$(pwd)}
export ESVO_BENCH_STREAM_CACHE=/tmp/esvo_streams
rounds=$1; shift
for r in $(seq 1 $rounds); do
  for e in "$@"; do
    a=$(env $e python $root/tools/small_tick.py dsec640x480 10000 60 2>/dev/null | tail -1 | sed 's/.*: \([0-9.]*\) ms per tick.*/\1/')
    b=$(env $e python $root/tools/small_tick.py dsec640x480 10000 60 pipelined 2>/dev/null | tail -1 | sed 's/.*: \([0-9.]*\) ms per tick.*/\1/')
    c=$(env $e python $root/tools/small_tick.py upenn346x260 1000 60 2>/dev/null | tail -1 | sed 's/.*: \([0-9.]*\) ms per tick.*/\1/')
    d=$(env $e python $root/tools/small_tick.py upenn346x260 1000 60 pipelined 2>/dev/null | tail -1 | sed 's/.*: \([0-9.]*\) ms per tick.*/\1/')
    echo "[$e] r$r  dsec 10000: $a sync / $b pipelined   upenn 1000: $c sync / $d pipelined"
  done
done
