#!/bin/bash
# round 4, session D: the rewritten fusion front -- parity tests first, then the bound probe's base line and a bench line
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out
mkdir -p $out
cd $root
export ESVO_BENCH_STREAM_CACHE=/tmp/esvo_streams
( time timeout 1500 python -m pytest tests -m gpu -q -x > $out/r04d_pytest.log 2>&1 ) 2> $out/r04d_pytest.time
tail -12 $out/r04d_pytest.log
timeout 600 python tools/bound_probe.py dsec640x480 30 3 base,no_regulariser,fusion_2x2,base_again > $out/r04d_bound.json 2> $out/r04d_bound.txt
cat $out/r04d_bound.txt
python bench.py --no-cpu-baseline --no-parity > $out/r04d_bench.json 2> $out/r04d_bench.err
python - <<'P'
import json
d = json.load(open("gpurun_out/r04d_bench.json"))
print("value", d["value"], "ms", d["ms_per_step"], "kernel_ms", d["kernel_ms"])
s = d.get("sustained") or {}
print("sustained", s.get("events_per_s"), s.get("ms_per_tick"), s.get("kernel_ms"))
o = d["other_operating_points"]
print({k: (v.get("events_per_s"), v.get("ms_per_tick"), v.get("ms_per_tick_pipelined"), v.get("ms_tracking")) for k, v in o.items()})
P
