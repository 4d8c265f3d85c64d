#!/bin/bash
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out
cd $root
( time python bench.py > $out/r04m_bench.json 2> $out/r04m_bench.err ) 2> $out/r04m_bench.time
cat $out/r04m_bench.time
python - <<'P'
import json
d = json.load(open("gpurun_out/r04m_bench.json"))
print("value", d["value"], "ms", d["ms_per_step"], "sclk", d.get("sclk_mhz_timed_region"), d["kernel_ms"])
s = d.get("sustained") or {}
print("sustained", s.get("events_per_s"), s.get("ms_per_tick"), s.get("sclk_mhz"), s.get("ms_per_tick_100tick_windows"), s.get("error"))
print("parity", json.dumps(d.get("parity"))[:1800])
o = d["other_operating_points"]
for k, v in o.items():
    print(k, {kk: vv for kk, vv in v.items() if kk in ("events_per_s", "ms_per_tick", "ms_per_tick_pipelined", "ms_tracking", "ms_per_cycle", "final_position_error_mm", "check_oracle_equal", "error")})
print("cpu", d["cpu_baseline"]["value"], d["cpu_baseline"]["port"]["value"])
P
