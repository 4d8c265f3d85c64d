#!/bin/bash
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out
cd $root
export ESVO_BENCH_STREAM_CACHE=/tmp/esvo_streams
timeout 600 python tools/bound_probe.py dsec640x480 30 3 base,no_regulariser,base_again > $out/r04j_bound.json 2> $out/r04j_bound.txt
cat $out/r04j_bound.txt
bash tools/gpu_r04e.sh 2>&1 | grep -i "tile_lists\|fuse_cells\|propagate\|fuse_turn\|reg_apply\|reg_view" | cut -c1-120
