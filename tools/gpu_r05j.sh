#!/bin/bash
root=${GRAFT_REPO_ROOT:-$(pwd)}
cd $root
( timeout 150 python -m pytest tests/test_gpu_bench_parity.py -m gpu -x -q ) 2>&1 | tail -3
