#!/bin/bash
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out
cd $root
export ESVO_BENCH_STREAM_CACHE=/tmp/esvo_streams
timeout 600 python tools/bound_probe.py dsec640x480 30 3 base,no_regulariser,two_lm,base_again > $out/r04l_bound.json 2> $out/r04l_bound.txt
cat $out/r04l_bound.txt
( time timeout 1500 python -m pytest tests -m gpu -q > $out/r04l_pytest.log 2>&1 ) 2> $out/r04l_pytest.time
tail -5 $out/r04l_pytest.log
timeout 900 bash tools/profile_round.sh r04_v2 > /dev/null 2>&1
head -14 $out/r04_v2_kernel_stats.csv | cut -c1-110
python - <<'P'
import json
d = json.load(open("gpurun_out/r04_v2_bench.json"))
print("value", d["value"], "ms", d["ms_per_step"], d["kernel_ms"])
P
