#!/usr/bin/env python
"""The ops of every stream over two steady-state ticks from a rocprofv3 --kernel-trace database: start (us from the first LM
launch of the window), duration, gap to the previous op of the same stream.
usage: python tools/stream_trace.py results.db [index of the window's first LM launch, default 10]"""
import sqlite3
import sys


def main(path, first=10):
    db = sqlite3.connect(path)
    rows = db.execute("select name, start, end, queue_id, stream_id from kernels order by start").fetchall()
    lm = [r for r in rows if "lm_refine" in r[0] and ", 2>" not in r[0]]   # single launch or first stage
    lm = [r for r in lm if ", 1>" in r[0]] or lm
    t0, t1 = lm[first][1], lm[first + 2][1]
    sel = [r for r in rows if t0 <= r[1] < t1]
    print("window %.1f us (2 ticks)" % ((t1 - t0) / 1e3))
    streams = {}
    for n, s, e, q, st in sel:
        streams.setdefault((q, st), []).append((n, s, e))
    for k, v in sorted(streams.items(), key=lambda kv: kv[1][0][1]):
        busy = sum(e - s for _, s, e in v) / 1e3
        print(f"--- stream {k}: {len(v)} ops, busy {busy:.1f} us")
        prev = None
        for n, s, e in v:
            gap = (s - prev) / 1e3 if prev is not None else 0.0
            short = n.replace("void esvo::", "").replace("esvo::", "")[:46]
            print(f"   {(s - t0) / 1e3:8.1f}  +{(e - s) / 1e3:7.1f}  gap {gap:6.1f}  {short}")
            prev = e


if __name__ == "__main__":
    main(sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 10)
