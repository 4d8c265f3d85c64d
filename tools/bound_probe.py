#!/usr/bin/env python
"""What bounds the tick?  One stream, one process, a list of VARIANTS of the headline workload that each remove a known share of
one stage's work (fewer LM iterations, no regulariser, 2x2 fusion, ...) or change only the scheduling (environment switches
read at esvo_create); per variant: ms per tick (median / min of R repeats of N ticks), the per-stage HIP-event times, matches /
points per tick and the shader clock measured inside the run (esvo_stats_t::clk_*).  The tick's sensitivity to each stage's
work says which stage(s) it waits for.  usage: python tools/bound_probe.py [workload] [ticks] [repeats] > gpurun_out/bound.json"""
import json
import os
os.environ.setdefault("ESVO_DEV_SWITCHES", "1")   # the library reads its A/B switches only with this set
import sys
import time

import numpy as np

sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench  # noqa: E402
from esvo_amd import lib  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "dsec640x480"
N = int(sys.argv[2]) if len(sys.argv) > 2 else 30
R = int(sys.argv[3]) if len(sys.argv) > 3 else 3
only = sys.argv[4].split(",") if len(sys.argv) > 4 else None

VARIANTS = [
    ("base", {}, {}),
    ("no_clk_probe", {}, {"ESVO_CLK_PROBE": "0"}),
    ("lm_iter7", {"lm_max_iteration": 7}, {}),
    ("lm_iter5", {"lm_max_iteration": 5}, {}),
    ("lm_iter3", {"lm_max_iteration": 3}, {}),
    ("no_regulariser", {"regularization": 0}, {}),
    ("fusion_2x2", {"fusion_radius": 0}, {}),
    ("no_reg_fusion_2x2", {"regularization": 0, "fusion_radius": 0}, {}),
    ("no_reg_lm_iter5", {"regularization": 0, "lm_max_iteration": 5}, {}),
    ("lm_stream_off", {}, {"ESVO_LM_STREAM": "0"}),
    ("one_stream", {}, {"ESVO_ONE_STREAM": "1"}),
    ("lm_split", {}, {"ESVO_LM_SPLIT": "1"}),
    ("base_again", {}, {}),
    # two LM launches in flight (the head of tick k+1's fills the tail of tick k's).  (Round 4 also ran it with the fusion stage's
    # stream confined to compute units of its own -- hipExtStreamCreateWithCUMask, 1.5-4 ms per tick: profiles/r04_bound_probe_*.txt;
    # that switch is gone from the library.)
    ("two_lm", {}, {"ESVO_LM_QUEUES": "2", "ESVO_LM_QUEUES_MAX_EVENTS": "100000000"}),
    ("base_third", {}, {}),
    # round 6: is the back chain what keeps the persistent LM layout from paying?
    ("persist", {}, {"ESVO_LM_PERSIST": "1"}),
    ("persist_no_reg", {"regularization": 0}, {"ESVO_LM_PERSIST": "1"}),
    ("persist_no_reg_fusion_2x2", {"regularization": 0, "fusion_radius": 0}, {"ESVO_LM_PERSIST": "1"}),
    ("persist_1536", {}, {"ESVO_LM_PERSIST": "1", "ESVO_LM_PERSIST_BLOCKS": "1536"}),
    ("base_fourth", {}, {}),
]

rig, stream, p, ticks = bench.make_workload(name, N + 6)
out = []
for tag, over, env in VARIANTS:
    if only and tag not in only:
        continue
    import copy
    pv = copy.copy(p)
    for k, v in over.items():
        setattr(pv, k, v)
    saved = {k: os.environ.get(k) for k in env}
    os.environ.update(env)
    try:
        dev = lib.Esvo(pv, rig, device=0)
    finally:
        for k, v in saved.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v
    times = []
    row = None
    for rep in range(R):
        dev.reset()
        dev.ts_push_events(0, stream.ev_left)
        dev.ts_push_events(1, stream.ev_right)
        bench.run_single(dev, stream, ticks, 0, 6)
        dev.synchronize()
        b = dev.stats()
        t0 = time.perf_counter()
        bench.run_single(dev, stream, ticks, 6, N + 6)
        dev.synchronize()
        dt = time.perf_counter() - t0
        s = dev.stats()
        times.append(dt / N * 1e3)
        ks = np.array(s.kernel_ms_mean(b))   # per sampled tick (stage timings are sampled: esvo_hip.h stage_timing_samples)
        sclk, _ = s.sclk_mhz(b)
        row = {"variant": tag, "params": over, "env": env, "events_per_tick": int(s.total_events_in - b.total_events_in) // N,
               "matches_per_tick": int(s.total_matches - b.total_matches) // N, "points_per_tick": int(s.total_points - b.total_points) // N,
               "kernel_ms": {"bm": round(float(ks[2]), 4), "lm": round(float(ks[3]), 4), "fuse": round(float(ks[4]), 4),
                             "clean": round(float(ks[5]), 4), "reg": round(float(ks[6]), 4)},
               "sclk_mhz": None if sclk is None else round(sclk, 1)}
    dev.close()
    row["ms_per_tick_median"] = float(np.median(times))
    row["ms_per_tick_min"] = float(np.min(times))
    row["ms_per_tick_all"] = [round(t, 4) for t in times]
    out.append(row)
    print(f"{tag:22s} {row['ms_per_tick_median']:.4f} ms (min {row['ms_per_tick_min']:.4f})  lm {row['kernel_ms']['lm']:.3f} fuse {row['kernel_ms']['fuse']:.3f} "
          f"reg {row['kernel_ms']['reg']:.3f} bm {row['kernel_ms']['bm']:.3f}  matches {row['matches_per_tick']}  sclk {row['sclk_mhz']}", file=sys.stderr)
print(json.dumps(out))
