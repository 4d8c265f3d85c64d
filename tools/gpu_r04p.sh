#!/bin/bash
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out
cd $root
export ESVO_BENCH_STREAM_CACHE=/tmp/esvo_streams
for i in 1 2; do
python bench.py --no-cpu-baseline --no-parity > $out/r04p_bench$i.json 2> $out/r04p_bench$i.err
python - <<P
import json
d = json.load(open("gpurun_out/r04p_bench$i.json"))
s = d["sustained"]
print("run $i value %.1f M (%.4f ms)  sustained %.1f M (%.4f ms) windows %s kernel %s" % (d["value"] / 1e6, d["ms_per_step"], s["events_per_s"] / 1e6, s["ms_per_tick"], s["ms_per_tick_100tick_windows"], s["kernel_ms"]))
o = d["other_operating_points"]
print({k: (round(v.get("events_per_s", 0) / 1e6, 1), v.get("ms_per_tick"), v.get("ms_per_tick_pipelined")) for k, v in o.items()})
P
done
