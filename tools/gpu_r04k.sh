#!/bin/bash
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out
cd $root
export ESVO_BENCH_STREAM_CACHE=/tmp/esvo_streams
timeout 600 python tools/bound_probe.py dsec640x480 30 3 base,base_again > $out/r04k_bound.json 2> $out/r04k_bound.txt
cat $out/r04k_bound.txt
timeout 900 bash tools/profile_round.sh r04_v2 > /dev/null 2>&1
grep -i "tile_lists\|fuse_cells\|propagate\|ts_scatter\|reg_apply\|fuse_turn" $out/r04_v2_hbm_traffic.csv | cut -d, -f1-4
head -12 $out/r04_v2_kernel_stats.csv | cut -c1-110
python - <<'P'
import json
d = json.load(open("gpurun_out/r04_v2_bench.json"))
print("value", d["value"], "ms", d["ms_per_step"], d["kernel_ms"])
P
