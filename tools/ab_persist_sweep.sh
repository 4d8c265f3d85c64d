export ESVO_DEV_SWITCHES=1
root=${GRAFT_REPO_ROOT:-$(pwd)}
export ESVO_BENCH_STREAM_CACHE=/tmp/esvo_streams
export ESVO_HIP_LIB=$root/tools/ab/libesvo_hip_p_ty4.so
for b in 2048 1792 1536 1280; do
  ESVO_LM_PERSIST_BLOCKS=$b python $root/tools/sustained_probe.py 600 "ty4 persist $b" 2>/dev/null | tail -1
done
ESVO_LM_PERSIST=0 python $root/tools/sustained_probe.py 600 "ty4 old kernel" 2>/dev/null | tail -1
