#!/bin/bash
# usage (GPU box): bash tools/cl_trace.sh  -- one closed-loop cycle (tracker registration + mapper tick, 346x260), operation by operation
root=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
out=$root/gpurun_out/cl_trace
rm -rf $out; mkdir -p $out
rocprofv3 --kernel-trace --memory-copy-trace -d $out -o t -- python $root/tools/closed_loop_ms.py 2>/dev/null | head -1
python - $out/t_results.db <<'PY'
import sqlite3, sys
db = sqlite3.connect(sys.argv[1])
rows = [(n, s, e, f"q{q}/s{st}") for n, s, e, q, st in db.execute("select name, start, end, queue_id, stream_id from kernels").fetchall()]
try:
    rows += [("COPY " + str(n), s, e, "copy") for n, s, e in db.execute("select name, start, end from memory_copies").fetchall()]
except sqlite3.Error:
    pass
rows.sort(key=lambda r: r[1])
lm = [i for i, r in enumerate(rows) if "lm_refine" in r[0]]
i0, i1 = lm[-3], lm[-2]   # one cycle: from behind an LM launch to the next one's compaction
t0 = rows[i0 + 1][1]
prev = t0
for n, s, e, q in rows[i0 + 1:i1 + 3]:
    print(f"{(s - t0) / 1e3:8.1f} us  +{(e - s) / 1e3:7.1f}  gap {(s - prev) / 1e3:6.1f}  {q:8s} {n[:64]}")
    prev = max(prev, e)
PY
rm -rf $out
