#!/bin/bash
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
export ESVO_BENCH_STREAM_CACHE=/tmp/esvo_streams
rocprofv3 --kernel-trace --stats -d /tmp/prof_sa -o ks -- python $root/tools/standalone_kernels.py dsec640x480 12 > $out/r04e_sa.log 2>&1
python $root/tools/prof_summary.py /tmp/prof_sa/ks_results.db $out/r04e_standalone_kernel_stats.csv > /dev/null 2>&1
head -16 $out/r04e_standalone_kernel_stats.csv | cut -c1-160
