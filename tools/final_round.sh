#!/bin/bash
# usage (GPU box): bash tools/final_round.sh [tag]  -- the round's closing evidence on the current tree -> gpurun_out/<tag>_* (copy into profiles/)
tag=${1:-r06_final7}
python -m pytest tests -m gpu -x -q --durations=8 > gpurun_out/${tag}_pytest_gpu.txt 2>&1; echo "pytest exit $?" >> gpurun_out/${tag}_pytest_gpu.txt
python bench.py --steps 20 --warmup 5 > gpurun_out/${tag}_bench.json 2> gpurun_out/${tag}_bench.err
if [ -z "$FINAL_SHORT" ]; then
  bash tools/profile_round.sh ${tag/final/v} > /dev/null 2>&1
  bash tools/small_trace.sh dsec640x480 10000 > gpurun_out/r06_lowlat_trace_dsec.txt 2>&1
  bash tools/small_trace.sh upenn346x260 1000 > gpurun_out/r06_lowlat_trace_upenn.txt 2>&1
  bash tools/ab_env_small.sh 3 "ESVO_LOWLAT=0" "ESVO_LOWLAT=1" > gpurun_out/r06_lowlat_ab.txt 2>&1
  bash tools/ab_band_share.sh > gpurun_out/r06_lowlat_band.txt 2>&1
  python tools/closed_loop_ms.py > gpurun_out/r06_lowlat_closed_loop.txt 2>&1
  python bench.py --steps 20 --warmup 5 --extras --no-cpu-baseline > gpurun_out/${tag}_extras_bench.json 2> gpurun_out/${tag}_extras_bench.err
fi
tail -3 gpurun_out/${tag}_pytest_gpu.txt
