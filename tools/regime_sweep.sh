#!/bin/bash
# usage (GPU box): bash tools/regime_sweep.sh  -- does the pipeline recover from a stage that fell behind?  (tools/regime_probe.py:
# ms per tick over 100-tick windows, a stall of one queue at tick 300) with and without the resync of api_map.hip
export ESVO_DEV_SWITCHES=1   # the library reads its A/B switches only with this set
export ESVO_BENCH_STREAM_CACHE=/tmp/esvo_streams
for rs in 1 0; do
 for d in back_stall:3000 back_stall:20000 lm_stall:5000 second_handle; do
  ESVO_RESYNC=$rs ESVO_PROBE_DISTURB=$d python tools/regime_probe.py 1000 "resync$rs" 2>&1 | tail -1
 done
done
