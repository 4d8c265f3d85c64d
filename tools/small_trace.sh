#!/bin/bash
# usage (GPU box): bash tools/small_trace.sh <workload> <events>  -- one synchronised reference-faithful tick, op by op
export ESVO_DEV_SWITCHES=1   # the library reads its A/B switches only with this set
root=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
out=$root/gpurun_out/small_$1
rm -rf $out; mkdir -p $out
rocprofv3 --kernel-trace --memory-copy-trace -d $out -o t -- python $root/tools/small_tick.py $1 $2 40 2>/dev/null | tail -1
python $root/tools/tick_trace.py $out/t_results.db ${3:-30}
rm -rf $out
