#!/usr/bin/env python
"""Per-stream occupancy of the tick from a rocprofv3 --kernel-trace database (rocpd sqlite).
usage: python tools/timeline.py results.db   -> busy time per queue/stream, union busy time, span (steady-state ticks)"""
import sqlite3
import sys


def main(path):
    db = sqlite3.connect(path)
    cols = [r[1] for r in db.execute("pragma table_info(kernels)")]
    print(cols)
    rows = db.execute("select name, start, end, queue_id, stream_id from kernels order by start").fetchall()
    # steady state: from the 8th to the 20th lm_refine launch
    lm = [r for r in rows if "lm_refine" in r[0]]
    t0, t1 = lm[7][1], lm[19][1]
    sel = [r for r in rows if t0 <= r[1] < t1]
    span = (t1 - t0) / 12e3
    per = {}
    for n, s, e, q, st in sel:
        per.setdefault((q, st), []).append((s, e))
    for k, v in per.items():
        busy = sum(e - s for s, e in v) / 12e3
        print("queue/stream", k, "busy us/tick %.1f" % busy, "kernels/tick %.1f" % (len(v) / 12))
    ev = sorted((s, e) for _, s, e, _, _ in sel)
    union = 0
    cs, ce = ev[0]
    for s, e in ev[1:]:
        if s > ce:
            union += ce - cs
            cs, ce = s, e
        else:
            ce = max(ce, e)
    union += ce - cs
    print("span us/tick %.1f  union busy us/tick %.1f" % (span, union / 12e3))
    # gaps on the front stream around the lm kernel
    names = {}
    for n, s, e, q, st in sel:
        names.setdefault(n[:40], [0, 0])
        names[n[:40]][0] += (e - s) / 12e3
        names[n[:40]][1] += 1
    for n, (t, c) in sorted(names.items(), key=lambda x: -x[1][0])[:14]:
        print("%8.1f us/tick  x%.1f  %s" % (t, c / 12, n))


if __name__ == "__main__":
    main(sys.argv[1])
