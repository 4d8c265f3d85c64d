#!/bin/bash
# round 4, first GPU session: default bench line (sustained point, parity block, in-run clock), GPU tests, the bound probe,
# one profile round.  usage (gpurun): bash tools/gpu_r04a.sh
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out
mkdir -p $out
cd $root
export ESVO_BENCH_STREAM_CACHE=/tmp/esvo_streams
( time python bench.py > $out/r04a_bench.json 2> $out/r04a_bench.err ) 2> $out/r04a_bench.time
tail -c 600 $out/r04a_bench.err
python - <<'P'
import json
d = json.load(open("gpurun_out/r04a_bench.json"))
print("value", d["value"], "ms", d["ms_per_step"], "sclk", d.get("sclk_mhz_timed_region"))
print("sustained", json.dumps(d.get("sustained"))[:1500])
print("parity", json.dumps(d.get("parity"))[:1500])
P
( time timeout 900 python -m pytest tests -m gpu -x -q > $out/r04a_pytest.log 2>&1 ) 2> $out/r04a_pytest.time
tail -5 $out/r04a_pytest.log
timeout 600 python tools/bound_probe.py dsec640x480 30 3 > $out/r04a_bound.json 2> $out/r04a_bound.txt
cat $out/r04a_bound.txt
timeout 900 bash tools/profile_round.sh r04_v1 > /dev/null 2>&1
cat $out/r04_v1_timeline.txt | head -40
