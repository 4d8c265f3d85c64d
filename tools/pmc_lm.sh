export ESVO_DEV_SWITCHES=1   # the library reads its A/B switches only with this set
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_BUSY_CYCLES -d $GRAFT_REPO_ROOT/gpurun_out/$1 -o x -- python $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-extras > /dev/null 2>&1
python - <<PY
import sqlite3
cur=sqlite3.connect("$GRAFT_REPO_ROOT/gpurun_out/$1/x_results.db").cursor()
for k,c,v,n in cur.execute("select kernel_name, counter_name, avg(value), count(*) from counters_collection where kernel_name like '%lm_refine%' group by kernel_name, counter_name"):
    print("$1", c, f"{v:.4g}")
for r in cur.execute("select name, average from top_kernels where name like '%lm_refine%'"): print("$1 avg us", r[1])
PY
