root=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
export ESVO_DEV_SWITCHES=1
out=$root/gpurun_out/trace_ts
rm -rf $out; mkdir -p $out
rocprofv3 --kernel-trace -d $out -o t -- python $root/tools/tick_share_probe.py dsec640x480 8 12 1.258 > $root/gpurun_out/ts_traced.json 2>/dev/null
python $root/tools/stream_trace.py $out/t_results.db 150 > $root/gpurun_out/ts_stream_trace.txt
rm -rf $out
for q in 4 8 16; do
  echo "=== GPU_MAX_HW_QUEUES=$q"
  GPU_MAX_HW_QUEUES=$q python $root/tools/tick_share_probe.py dsec640x480 8 20 1.258 2>/dev/null | grep -E "round_ms_pipelined|mean|lm_refine|fuse|regularize|projected_speedup"
done
