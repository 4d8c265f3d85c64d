#!/usr/bin/env python
"""ms per tick of the SUSTAINED headline workload (bench.sustained_point: the looped stationary stream, all events resident) for
the library named by ESVO_HIP_LIB / the environment of this process: one line.  usage: python tools/sustained_probe.py [ticks] [tag]"""
import json
import os
import sys

sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 600
tag = sys.argv[2] if len(sys.argv) > 2 else "run"
r = bench.sustained_point("dsec640x480", n, 0, None)
w = r["ms_per_tick_100tick_windows"]
print(f"{tag:28s} {r['ms_per_tick']:.4f} ms/tick  {r['events_per_s'] / 1e6:.1f} M ev/s  windows first {w['first']:.3f} median {w['median']:.3f} last {w['last']:.3f}  "
      f"lm {r['kernel_ms']['lm_refine']:.3f} fuse {r['kernel_ms']['fuse']:.3f} reg {r['kernel_ms']['regularize']:.3f} bm {r['kernel_ms']['bm_match']:.3f}  sclk {r['sclk_mhz']:.0f}")
