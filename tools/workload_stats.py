"""Workload statistics of the bench stream on the GPU (records per DepthMap cell, map density)."""
import sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from esvo_amd import calib, lib, params, rostime, synth

rig = calib.dataset_rig("dsec")
K = 8
duration = 0.06 + (K + 1) * 0.01
st = synth.make_stream(rig, 180000, duration, 0.02, 0.25, seed=20250418 + 3, speed=2.0)
p, _ = params.make_params(params.PRESETS["mapping_dsec"], rig, throughput_events=300000,
                          event_ring_capacity=max(1 << 22, int(len(st.ev_left) * 1.1)))
dev = lib.Esvo(p, rig)
dev.ts_push_events(0, st.ev_left); dev.ts_push_events(1, st.ev_right)
frames = []
for k in range(K):
    t = st.t0_ns + int((0.06 + (k + 1) * 0.01) * 1e9)
    stamps, poses = rostime.pose_table(st.pose, t, p.bm_half_slice_thickness)
    dev.ts_render(0, t, download=False); dev.ts_render(1, t, download=False)
    dev.set_observation(t, None, None, st.pose(t)); dev.tick(t, stamps, poses)
    frames.append(dev.get_last_frame())
    s = dev.stats()
    print(k, "events", s.last_events_in, "matches", s.last_matches, "solved", s.last_solved, "points", s.last_points,
          "window", s.last_window_frames, s.last_window_points, "fusions", s.last_fusions,
          "ms", [round(x, 3) for x in s.ms_kernel[:7]])
mp = dev.get_map()
print("map size", len(mp), "of", rig.width * rig.height, "valid(inv>0)", int((mp["inv_depth"] > 0).sum()))
# records per cell (3x3 around each point of the last 5 frames; propagation shifts are sub-pixel here)
cnt = np.zeros((rig.height + 2, rig.width + 2), np.int64)
for f in frames[-5:]:
    for dy in (-1, 0, 1):
        for dx in (-1, 0, 1):
            np.add.at(cnt, (f["row"].astype(int) + 1 + dy, f["col"].astype(int) + 1 + dx), 1)
c = cnt[1:-1, 1:-1].ravel()
nz = c[c > 0]
print("touched cells", len(nz), "records", nz.sum(), "mean", nz.mean(), "p50/p90/p99/max", np.percentile(nz, [50, 90, 99]), nz.max())
w = c[: (len(c) // 64) * 64].reshape(-1, 64)
print("per-wave max mean", w.max(1).mean(), "sum of wave max", w.max(1).sum(), "vs sum", c.sum())
