#!/usr/bin/env python
"""One synchronised tick, operation by operation, from a rocprofv3 --kernel-trace --memory-copy-trace database.
usage: python tools/tick_trace.py results.db [which lm_refine launch, default 12]"""
import sqlite3
import sys


def main(path, which=12):
    db = sqlite3.connect(path)
    rows = [(n, s, e, f"q{q}/s{st}") for n, s, e, q, st in
            db.execute("select name, start, end, queue_id, stream_id from kernels").fetchall()]
    try:
        rows += [("COPY " + str(n), s, e, "copy") for n, s, e in db.execute("select name, start, end from memory_copies").fetchall()]
    except sqlite3.Error as ex:
        print("no memory copies:", ex)
    rows.sort(key=lambda r: r[1])
    lm = [i for i, r in enumerate(rows) if "lm_refine" in r[0]]
    i0, i1 = lm[which - 1], lm[which]
    # the tick = from the first Time-Surface scatter after the previous LM launch to the one after this LM launch
    first = next(i for i in range(i0, i1) if "ts_scatter" in rows[i][0])
    last = next((i for i in range(i1, len(rows)) if "ts_scatter" in rows[i][0]), len(rows))
    out = rows[first:last]
    end = max(r[2] for r in out)
    g = out[0][1] - max(r[2] for r in rows[:first]) if first else 0
    t0 = out[0][1]
    prev_end = t0
    busy = 0
    for n, s, e, q in out:
        print(f"{(s - t0) / 1e3:8.1f} us  +{(e - s) / 1e3:7.1f}  gap {(s - prev_end) / 1e3:6.1f}  {q:8s} {n[:70]}")
        prev_end = max(prev_end, e)
    print(f"ops {len(out)}  span {(end - t0) / 1e3:.1f} us  idle before tick {g / 1e3:.1f} us")


if __name__ == "__main__":
    main(sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 12)
