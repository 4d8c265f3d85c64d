#!/bin/bash
# usage (GPU box): bash tools/ab_env.sh <rounds> "<ENV=VAL ...>" ["<ENV=VAL ...>" ...]  -- bench of the current library under different environments
export ESVO_DEV_SWITCHES=1   # the library reads its A/B switches only with this set
root=${GRAFT_REPO_ROOT:-$(pwd)}
rounds=$1; shift
for r in $(seq 1 $rounds); do
  for e in "$@"; do
    env $e python $root/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extras 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        j = json.loads(l); print('[$e]', 'r$r', '%.1f Mev/s' % (j['value'] / 1e6), '%.4f ms' % j['ms_per_step'], {k: round(v, 3) for k, v in j['kernel_ms'].items()})
"
  done
done
