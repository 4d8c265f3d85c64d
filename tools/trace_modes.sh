#!/bin/bash
# usage (GPU box): bash tools/trace_modes.sh "<ENV=VAL ...>" ...   -- kernel trace of the bench under each environment -> stream_trace
export ESVO_DEV_SWITCHES=1   # the library reads its A/B switches only with this set
root=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
i=0
for e in "$@"; do
  i=$((i+1))
  out=$root/gpurun_out/trace_$i
  rm -rf $out; mkdir -p $out
  env $e rocprofv3 --kernel-trace -d $out -o t -- python $root/bench.py --steps 16 --warmup 3 --no-cpu-baseline --no-extras > /dev/null 2>&1
  echo "=== [$e]"
  python $root/tools/stream_trace.py $out/t_results.db
  rm -rf $out
done
