#!/bin/bash
# round 4, call r: general patch sizes (25x25, 10x4), median 5x5 / 7x7 on the GPU
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
( time timeout 900 python -m pytest tests/test_gpu_ref.py tests/test_gpu_parity.py tests/test_gpu_edge.py -m gpu -x -q -k "upenn25 or dsec10x4 or time_surface or back_to_back" ) > gpurun_out/r04r_pytest.log 2>&1
tail -25 gpurun_out/r04r_pytest.log
