#!/usr/bin/env python
"""The regulariser's floor on the headline workload (verdict r05 item 6): per launch, the longest dependent chain of Student-t
steps a workgroup of reg_apply_kernel walks (per block of staged rows the slowest wave's slowest lane, summed over the blocks), the
longest chain of a single element, and what lockstep costs (steps executed by waves vs close taps).  Needs a -DREG_STATS build:
    python tools/ab_build.py regstats -DREG_STATS && ESVO_HIP_LIB=tools/ab/libesvo_hip_regstats.so python tools/reg_floor.py [workload] [ticks]"""
import ctypes
import os
import sys

os.environ.setdefault("ESVO_DEV_SWITCHES", "1")
os.environ.setdefault("ESVO_LOWLAT_TIMED_EVERY", "1")   # every synchronised tick records its stage-timing events (s.ms_kernel below)
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench  # noqa: E402
from esvo_amd import lib  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "dsec640x480"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 10
rig, stream, p, ticks = bench.make_workload(name, n)
dev = lib.Esvo(p, rig)
dev.ts_push_events(0, stream.ev_left)
dev.ts_push_events(1, stream.ev_right)
bench.run_single(dev, stream, ticks, 0, n - 3, sync_each=True)
out = (ctypes.c_ulonglong * 8)()
dev.lib.esvo_debug_reg_stats.argtypes = [ctypes.c_void_p, ctypes.c_int]
dev.lib.esvo_debug_reg_stats(out, 1)
rows = []
for k in range(n - 3, n):
    bench.run_single(dev, stream, ticks, k, k + 1, sync_each=True)
    s = dev.stats()
    dev.lib.esvo_debug_reg_stats(out, 1)
    rows.append((int(out[0]), int(out[1]), int(out[2]), int(out[3]), float(s.ms_kernel[6]), int(s.last_map_size)))
for r in rows:
    print(f"{name}: workgroup chain {r[0]} steps, longest element {r[1]} steps, wave-steps {r[2]}, close taps {r[3]} "
          f"(lockstep x{r[2] * 64 / max(r[3], 1):.2f}), regulariser stage alone {r[4] * 1e3:.0f} us, map {r[5]} cells")
