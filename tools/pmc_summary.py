#!/usr/bin/env python
"""Per-kernel averages of rocprofv3 --pmc passes (rocpd sqlite databases) as CSV.

usage: python tools/pmc_summary.py out.csv hbm|sq db1 [db2 ...]
  hbm: FETCH_SIZE / WRITE_SIZE passes -> kernel,counter,avg_value_per_dispatch_KB,dispatches,note
  sq : SQ_* passes                    -> kernel,counter,avg_value_per_dispatch,dispatches
"""
import csv
import re
import sqlite3
import sys

NOTE = ("rocprofv3 --pmc (separate pass); unit KB; FETCH_SIZE under-reports wide coalesced reads by 2x on gfx950 "
        "(MI355X_MICROARCH.md)")


def short(name):
    return re.sub(r"\(.*", "", name).replace("void ", "")


def main(out, kind, dbs):
    rows = {}
    for path in dbs:
        cur = sqlite3.connect(path).cursor()
        for k, c, v, n in cur.execute("select kernel_name, counter_name, avg(value), count(*) from counters_collection "
                                      "group by kernel_name, counter_name"):
            if k.startswith("esvo::") or "esvo::" in k:
                rows[(short(k), c)] = (v, n)
    with open(out, "w", newline="") as f:
        w = csv.writer(f)
        if kind == "hbm":
            w.writerow(["kernel", "counter", "avg_value_per_dispatch_KB", "dispatches", "note"])
            for (k, c), (v, n) in sorted(rows.items()):
                w.writerow([k, c, f"{v:.1f}", n, NOTE])
        else:
            w.writerow(["kernel", "counter", "avg_value_per_dispatch", "dispatches"])
            for (k, c), (v, n) in sorted(rows.items()):
                w.writerow([k, c, f"{v:.4g}", n])
    for (k, c), (v, n) in sorted(rows.items()):
        if "lm_refine" in k or "reg_chain" in k or "bm_match" in k:
            print(k, c, f"{v:.4g}", n)


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2], sys.argv[3:])
