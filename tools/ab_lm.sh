#!/bin/bash
# usage (GPU box): bash tools/ab_lm.sh <tag> [lib path]   -- lm_refine: VALU instruction count per launch + kernel time, bench workload
export ESVO_DEV_SWITCHES=1   # the library reads its A/B switches only with this set
tag=$1
root=${GRAFT_REPO_ROOT:-$(pwd)}
if [ -n "$2" ]; then export ESVO_HIP_LIB=$root/$2; fi
out=$root/gpurun_out/ab_$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
cmd="python $root/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extras"
$cmd > $out/bench.json 2> $out/bench.err
rocprofv3 --kernel-trace --stats -d $out -o ks -- $cmd > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES -d $out -o pmc -- python $root/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-extras > /dev/null 2>&1
python - <<PY
import sqlite3, json
line=[l for l in open("$out/bench.json") if l.startswith("{")]
if line:
    j=json.loads(line[-1]); print("$tag", "events/s %.4g" % j["value"], "ms/tick %.4f" % j["ms_per_step"], j["kernel_ms"])
cur=sqlite3.connect("$out/pmc_results.db").cursor()
for k,c,v,n in cur.execute("select kernel_name, counter_name, avg(value), count(*) from counters_collection where kernel_name like '%lm_refine%' or kernel_name like '%bm_match%' group by kernel_name, counter_name"):
    print("$tag", k[:40], c, f"{v:.5g}", n)
cur=sqlite3.connect("$out/ks_results.db").cursor()
try:
    for r in cur.execute("select name, total_calls, average, percentage from top_kernels order by total_duration desc limit 8"): print("$tag", r[0][:60], r[1], "avg us %.1f" % r[2], "%.1f%%" % r[3])
except Exception as e:
    print("stats:", e)
PY
rm -rf $out/*.db
