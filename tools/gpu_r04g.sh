#!/bin/bash
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out
mkdir -p $out
cd $root
export ESVO_BENCH_STREAM_CACHE=/tmp/esvo_streams
timeout 900 python -m pytest tests/test_gpu_edge.py tests/test_gpu_parity.py tests/test_gpu_ref.py tests/test_gpu_fullsize.py tests/test_gpu_shard.py -m gpu -q -x > $out/r04g_pytest.log 2>&1
tail -6 $out/r04g_pytest.log
timeout 600 python tools/bound_probe.py dsec640x480 30 3 base,no_regulariser,fusion_2x2,two_lm,base_again > $out/r04g_bound.json 2> $out/r04g_bound.txt
cat $out/r04g_bound.txt
bash tools/gpu_r04e.sh
