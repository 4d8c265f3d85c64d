#!/bin/bash
root=${GRAFT_REPO_ROOT:-$(pwd)}
cd $root
( timeout 600 python -m pytest tests/test_gpu_edge.py tests/test_gpu_threads.py -m gpu -x -q -k "event_queue_mode or thread" ) 2>&1 | tail -8
