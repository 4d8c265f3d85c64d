#!/bin/bash
# usage (on the GPU box, from the repo root): bash tools/profile_round.sh <tag>
# usage: bash tools/profile_round.sh <tag> [workload]
# Writes gpurun_out/<tag>_{bench.json,meta.json,kernel_stats.csv,hbm_traffic.csv,sq_counters.csv}; copy them into profiles/.
# The counter passes run on their own (no trace domains besides --kernel-trace), one counter group per pass.
tag=$1
wl=${2:-dsec640x480}
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
export ESVO_BENCH_STREAM_CACHE=/tmp/esvo_streams   # the seeded stream is generated once for the five runs below
cmd="python $root/bench.py --steps 20 --warmup 3 --workload $wl --sustained-ticks 0 --no-parity"
$cmd --no-extras > $out/${tag}_bench.json 2> $out/${tag}_bench.err
echo "{\"workload\": \"$wl\", \"bench_cmd\": \"bench.py --steps 20 --warmup 3 (kernel stats), --steps 5 --warmup 2 (counter passes)\"}" > $out/${tag}_meta.json
rocprofv3 --kernel-trace --stats -d $out/prof_$tag -o ks -- $cmd --no-cpu-baseline --no-extras > /dev/null 2>&1
python $root/tools/prof_summary.py $out/prof_$tag/ks_results.db $out/${tag}_kernel_stats.csv
python $root/tools/stream_trace.py $out/prof_$tag/ks_results.db > $out/${tag}_stream_trace.txt 2>&1
python $root/tools/timeline.py $out/prof_$tag/ks_results.db > $out/${tag}_timeline.txt 2>&1
short="python $root/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-extras --no-parity --sustained-ticks 0 --workload $wl"
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $out/prof_$tag -o fetch -- $short > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $out/prof_$tag -o write -- $short > /dev/null 2>&1
python $root/tools/pmc_summary.py $out/${tag}_hbm_traffic.csv hbm $out/prof_$tag/fetch_results.db $out/prof_$tag/write_results.db
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES -d $out/prof_$tag -o sq -- $short > /dev/null 2>&1
python $root/tools/pmc_summary.py $out/${tag}_sq_counters.csv sq $out/prof_$tag/sq_results.db
rm -rf $out/prof_$tag
cat $out/${tag}_bench.json
