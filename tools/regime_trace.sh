#!/bin/bash
# usage (GPU box): bash tools/regime_trace.sh  -- kernel trace of the sustained workload, a 3 ms stall of the back queue at tick 300:
# two steady-state ticks BEFORE (fast operating point) and AFTER (slow one), stream by stream
export ESVO_DEV_SWITCHES=1   # the library reads its A/B switches only with this set
root=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
out=$root/gpurun_out/regime_trace
rm -rf $out; mkdir -p $out
ESVO_BENCH_STREAM_CACHE=/tmp/esvo_streams ESVO_PROBE_DISTURB=back_stall:3000 $EXTRA_ENV rocprofv3 --kernel-trace -d $out -o t -- python $root/tools/regime_probe.py 520 trace 2>&1 | tail -1
echo "=== before the stall (ticks 200, 201)"
python $root/tools/stream_trace.py $out/t_results.db 208
echo "=== after the stall (ticks 450, 451)"
python $root/tools/stream_trace.py $out/t_results.db 458
rm -rf $out
