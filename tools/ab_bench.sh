#!/bin/bash
# usage (GPU box): bash tools/ab_bench.sh <rounds> <variant> [<variant> ...]   variant = name of tools/ab/libesvo_hip_<name>.so, or "cur"
# interleaved bench runs on one box: ms/tick and per-stage HIP-event times of every variant, <rounds> times
export ESVO_DEV_SWITCHES=1   # the library reads its A/B switches only with this set
root=${GRAFT_REPO_ROOT:-$(pwd)}
rounds=$1; shift
for r in $(seq 1 $rounds); do
  for v in "$@"; do
    if [ "$v" = "cur" ]; then unset ESVO_HIP_LIB; else export ESVO_HIP_LIB=$root/tools/ab/libesvo_hip_$v.so; fi
    python $root/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extras 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        j = json.loads(l); print('$v', 'r$r', '%.1f Mev/s' % (j['value'] / 1e6), '%.4f ms' % j['ms_per_step'], {k: round(v, 3) for k, v in j['kernel_ms'].items()})
"
  done
done
