#!/bin/bash
# round 4, call q: the all-gather band exchange + the ADVICE fixes on the GPU
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
( time timeout 900 python -m pytest tests/test_gpu_shard.py tests/test_gpu_comm.py tests/test_gpu_multiproc.py tests/test_gpu_edge.py -m gpu -x -q ) > gpurun_out/r04q_pytest.log 2>&1
tail -15 gpurun_out/r04q_pytest.log
