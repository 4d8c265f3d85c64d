#!/bin/bash
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out
mkdir -p $out
cd $root
export ESVO_BENCH_STREAM_CACHE=/tmp/esvo_streams
( time timeout 1500 python -m pytest tests -m gpu -q > $out/r04f_pytest.log 2>&1 ) 2> $out/r04f_pytest.time
tail -12 $out/r04f_pytest.log
timeout 600 python tools/bound_probe.py dsec640x480 30 3 base,no_regulariser,fusion_2x2,two_lm,base_again > $out/r04f_bound.json 2> $out/r04f_bound.txt
cat $out/r04f_bound.txt
bash tools/gpu_r04e.sh
