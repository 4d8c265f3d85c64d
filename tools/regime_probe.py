#!/usr/bin/env python
"""Does the tick pipeline recover from a disturbance?  The sustained headline workload (bench.sustained_point) with a second
handle ticking the same workload for a few ticks in the middle of the run (its kernels compete for the chip, both chains of
the pipeline fall behind), then nothing: ms per tick over 100-tick windows before, during and after.  A pipeline with a
second stable operating point stays slow after the disturbance.  usage: python tools/regime_probe.py [ticks] [tag]"""
import os
os.environ.setdefault("ESVO_DEV_SWITCHES", "1")   # the library reads its A/B switches only with this set
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
mode = os.environ.get("ESVO_PROBE_DISTURB", "second_handle")   # second_handle | back_stall:<us> | lm_stall:<us> | front_stall:<us>


def hook(k, dev):
    if mode == "second_handle":
        # ticks 300..317 of the measured run: one tick of the second handle after each tick of the first (un-synchronised)
        if 300 <= k < 318:
            t, st, po, T = nt[state["k"]]
            state["k"] += 1
            noise.tick_resident(t, T, st, po)
    elif k == 300:   # ONE stage of the pipeline falls behind by <us> (a kernel that does nothing occupies its queue)
        which, us = mode.split(":")
        import ctypes
        dev.lib.esvo_debug_stall.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_uint]
        rc = dev.lib.esvo_debug_stall(dev.h, {"front_stall": 0, "lm_stall": 1, "back_stall": 2}[which], int(us))
        assert rc == 0, rc


def at_end(dev):
    """ESVO_TIMELINE=1: the stage times of ticks in the middle of the fast and of the slow stretch (esvo_debug_timeline)"""
    import ctypes
    import numpy as np
    rows = np.zeros((2000, 12), np.float32)
    nr = ctypes.c_int()
    dev.lib.esvo_debug_timeline.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.POINTER(ctypes.c_int)]
    if dev.lib.esvo_debug_timeline(dev.h, rows.ctypes.data, 2000, ctypes.byref(nr)) != 0 or nr.value < 100:
        return
    rows = rows[: nr.value]
    names = ["T0", "BM0", "BM1", "S1", "LM0", "LM1", "S2", "CNT", "FU0", "FU1", "CL1", "RG1"]
    for label, i0 in (("before", max(nr.value - n + 200, 3)), ("after", nr.value - 60)):
        t0 = rows[i0][0]
        for r in rows[i0: i0 + 3]:
            print(f"   {label:6s} " + "  ".join(f"{nm} {x - t0:7.3f}" for nm, x in zip(names, r)))


if os.environ.get("ESVO_TIMELINE"):
    hook.at_end = at_end
r = bench.sustained_point(name, n, 0, None, hook=hook)
noise.synchronize()
noise.close()
w = r["windows_ms"]
print(f"{tag:24s} {mode:18s} windows(100 ticks) " + " ".join(f"{x:.3f}" for x in w) + f"   lm {r['kernel_ms']['lm_refine']:.3f} reg {r['kernel_ms']['regularize']:.3f} resyncs {r.get('pipeline_resyncs')}")
