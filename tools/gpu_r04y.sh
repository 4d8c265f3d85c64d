#!/bin/bash
root=${GRAFT_REPO_ROOT:-$(pwd)}
cd $root
mkdir -p gpurun_out
export TMPDIR=/tmp
( time timeout 1800 python -m pytest tests -m gpu -q ) > gpurun_out/r04y_pytest.log 2>&1
tail -12 gpurun_out/r04y_pytest.log
