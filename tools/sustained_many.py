#!/usr/bin/env python
"""N consecutive sustained runs of the headline workload in one process (bench.sustained_point: 1600 ticks = 2.1 s each): ms per tick
of every run, how many lie above a bound, how often the pipeline had to resynchronise (stats.pipeline_resyncs).
usage: python tools/sustained_many.py [runs] [ticks] [bound ms]"""
import os
import sys

sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench  # noqa: E402

runs = int(sys.argv[1]) if len(sys.argv) > 1 else 50
ticks = int(sys.argv[2]) if len(sys.argv) > 2 else 1600
bound = float(sys.argv[3]) if len(sys.argv) > 3 else 1.35
ms, rs = [], []
for k in range(runs):
    r = bench.sustained_point("dsec640x480", ticks, 0, None)
    w = r["ms_per_tick_100tick_windows"]
    ms.append(r["ms_per_tick"])
    rs.append(r["pipeline_resyncs"])
    print(f"run {k + 1:3d}  {r['ms_per_tick']:.4f} ms/tick  {r['events_per_s'] / 1e6:6.1f} M ev/s  100-tick windows min {w['min']:.3f} max {w['max']:.3f}  "
          f"lm {r['kernel_ms']['lm_refine']:.3f} reg {r['kernel_ms']['regularize']:.3f}  resyncs {r['pipeline_resyncs']}  sclk {r['sclk_mhz']:.0f}", flush=True)
above = sum(1 for x in ms if x > bound)
print(f"{runs} runs of {ticks} ticks: min {min(ms):.4f} median {sorted(ms)[len(ms) // 2]:.4f} max {max(ms):.4f} ms/tick; above {bound} ms: {above}; "
      f"runs with a resync: {sum(1 for x in rs if x)} (resyncs in total {sum(rs)})")
