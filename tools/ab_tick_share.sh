#!/bin/bash
# usage (GPU box): bash tools/ab_tick_share.sh <rounds> "<ENV=VAL ...>" ...  -- interleaved tick_share probes under each environment
export ESVO_DEV_SWITCHES=1
root=${GRAFT_REPO_ROOT:-$(pwd)}
rounds=$1; shift
for rep in 1 2; do
  for e in "$@"; do
    r=$(env $e python $root/tools/tick_share_probe.py dsec640x480 8 $rounds 1.258 2>/dev/null | python -c "
import json,sys
d=json.load(sys.stdin)
k=d['kernel_ms_own_tick']
print('round %.3f sync %.3f  lm %.3f bm %.3f fuse %.3f reg %.3f  wait %.3f  x%.2f eq %s %s' % (d['round_ms_pipelined'], d['round_ms_synchronised']['mean'], k['lm_refine'], k['bm_match'], k['fuse'], k['regularize'], d['host_ms_per_round']['of_which_waiting_for_counts'], d['projected_speedup_at_world'], d['own_frames_equal_recorded'], d['map_equal_to_one_gpu']))")
    echo "[$e] $r"
  done
done
