#!/usr/bin/env python
"""Does the tick pipeline recover from a disturbance?  The sustained headline workload (bench.sustained_point) with a second
handle ticking the same workload for a few ticks in the middle of the run (its kernels compete for the chip, both chains of
the pipeline fall behind), then nothing: ms per tick over 100-tick windows before, during and after.  A pipeline with a
second stable operating point stays slow after the disturbance.  usage: python tools/regime_probe.py [ticks] [tag]"""
import os
import sys

sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench  # noqa: E402
from esvo_amd import lib, params, rostime  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1200
tag = sys.argv[2] if len(sys.argv) > 2 else "run"
name = "dsec640x480"
wl = bench.WORKLOADS[name]
rig, stream, p0, ticks0 = bench.make_workload(name, 40, 0)
p, _ = params.make_params(params.PRESETS[wl["preset"]], rig, throughput_events=p0.process_event_num)
noise = lib.Esvo(p, rig, device=0)
noise.ts_push_events(0, stream.ev_left)
noise.ts_push_events(1, stream.ev_right)
nt = []
for k in range(30):
    t = stream.t0_ns + int(round((bench.HIST_S + (k + 1) * bench.TICK_S) * 1e9))
    st, po = rostime.pose_table(stream.pose, t, p.bm_half_slice_thickness)
    nt.append((t, st, po, stream.pose(t)))
state = {"k": 0}


def hook(k):
    # ticks 300..317 of the measured run: one tick of the second handle after each tick of the first (un-synchronised)
    if 300 <= k < 318:
        t, st, po, T = nt[state["k"]]
        state["k"] += 1
        noise.tick_resident(t, T, st, po)


r = bench.sustained_point(name, n, 0, None, hook=hook)
noise.synchronize()
noise.close()
w = r["windows_ms"]
print(f"{tag:24s} windows(100 ticks) " + " ".join(f"{x:.3f}" for x in w) + f"   lm {r['kernel_ms']['lm_refine']:.3f} reg {r['kernel_ms']['regularize']:.3f}")
