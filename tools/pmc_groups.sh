#!/bin/bash
# usage (GPU box): bash tools/pmc_groups.sh <tag> "<ENV=VAL ...>" "<counters of pass 1>" ["<counters of pass 2>" ...]
# One rocprofv3 --pmc pass per counter group (a group must fit one pass) over a short bench run; per-kernel averages of the
# main kernels -> gpurun_out/<tag>_pmc.txt.  Counter passes only (--kernel-trace, no other trace domain).
export ESVO_DEV_SWITCHES=1   # the library reads its A/B switches only with this set
tag=$1; envs=$2; shift 2
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
export ESVO_BENCH_STREAM_CACHE=/tmp/esvo_streams
short="python $root/bench.py --steps 6 --warmup 3 --no-cpu-baseline --no-extras --no-parity --sustained-ticks 0"
: > $out/${tag}_pmc.txt
i=0
for grp in "$@"; do
  i=$((i+1))
  rm -rf /tmp/pmc_$tag_$i
  env $envs rocprofv3 --kernel-trace --pmc $grp -d /tmp/pmc_${tag}_$i -o p -- $short > /tmp/pmc_${tag}_$i.log 2>&1
  python - <<PY >> $out/${tag}_pmc.txt
import sqlite3, glob
dbs = glob.glob("/tmp/pmc_${tag}_$i/**/p_results.db", recursive=True)
if not dbs:
    print("pass $i FAILED: $grp")
    print(open("/tmp/pmc_${tag}_$i.log").read()[-600:])
else:
    cur = sqlite3.connect(dbs[0]).cursor()
    q = ("select kernel_name, counter_name, avg(value), count(*) from counters_collection group by kernel_name, counter_name")
    for k, c, v, n in cur.execute(q):
        if any(s in k for s in ("lm_refine", "bm_match", "fuse_cells", "reg_apply", "scatter_records", "propagate_kernel", "ts_scatter", "tile_lists", "fuse_turn")):
            print(f"{k.split('(')[0].replace('void ', '')[:60]:60s} {c:28s} {v:14.5g} n={n}")
    try:
        for r in cur.execute("select name, average from top_kernels where name like '%lm_refine%' or name like '%fuse_cells%' or name like '%bm_match%'"):
            print(f"  avg_us {r[0][:50]:50s} {r[1]:.1f}")
    except Exception as e:
        print("top_kernels:", e)
PY
done
cat $out/${tag}_pmc.txt
