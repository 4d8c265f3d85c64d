#!/bin/bash
root=${GRAFT_REPO_ROOT:-$(pwd)}
cd $root
( timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_edge.py -m gpu -x -q -k "time_surface or render_times or empty_and_tiny or ring_wraparound" ) 2>&1 | tail -15
