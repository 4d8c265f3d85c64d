#!/usr/bin/env python
"""Compute-side projection of the sharded tick on ONE GPU: G logical shards (handles) of the bench workload run
the three phases one after the other; the sums between the phases are emulated on the device (as in
tests/test_gpu_shard.py).  Prints, per G, the max-over-shards wall time of each phase -- what one rank of a
G-GPU run computes per tick, without the two all-reduces (their sizes are printed).  Not a scaling measurement."""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from esvo_amd import calib, dist as edist, lib, params, rostime, synth  # noqa: E402
import bench  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--shards", type=int, nargs="+", default=[1, 2, 4, 8])
    ap.add_argument("--steps", type=int, default=6)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--workload", default="dsec640x480")
    args = ap.parse_args()
    import torch
    wl = bench.WORKLOADS[args.workload]
    rig = calib.dataset_rig(wl["rig"])
    K, Wm = args.steps, args.warmup
    tick_s, hist_s = 0.010, 0.060
    duration = hist_s + (K + Wm + 1) * tick_s
    stream = synth.make_stream(rig, wl["points"], duration, wl["rho"][0], wl["rho"][1], seed=20250418 + 3, speed=wl["speed"])
    cap = int(len(stream.ev_left) / duration * tick_s * 1.5) + 1024
    p, _ = params.make_params(params.PRESETS[wl["preset"]], rig, throughput_events=cap,
                              event_ring_capacity=max(1 << 22, int(len(stream.ev_left) * 1.1)))
    ticks = []
    for k in range(K + Wm):
        t = stream.t0_ns + int((hist_s + (k + 1) * tick_s) * 1e9)
        stamps, poses = rostime.pose_table(stream.pose, t, p.bm_half_slice_thickness)
        ticks.append((t, stamps, poses, stream.pose(t)))
    out = {}
    for G in args.shards:
        shards = [lib.Esvo(p, rig) for _ in range(G)]
        for g, d in enumerate(shards):
            y0, y1 = edist.band_of(g, G, rig.height)
            d.set_band(y0, y1, g, G)
            d.ts_push_events(0, stream.ev_left)
            d.ts_push_events(1, stream.ev_right)
        ph = np.zeros((3, G))
        xbytes = [0, 0]
        for k, (t, stamps, poses, T) in enumerate(ticks):
            for d in shards:
                d.ts_render(0, t, download=False); d.ts_render(1, t, download=False)
                d.set_observation(t, None, None, T)
                d.synchronize()
            for phase in range(3):
                for g, d in enumerate(shards):
                    t0 = time.perf_counter()
                    if G == 1:
                        if phase == 0:
                            d.tick(t, stamps, poses)
                    elif phase == 0:
                        d.shard_phase(0, t, stamps, poses)
                    else:
                        d.shard_phase(phase)
                    d.synchronize()
                    if k >= Wm:
                        ph[phase, g] += time.perf_counter() - t0
                if phase < 2 and G > 1:
                    bufs = []
                    for d in shards:
                        ptr, nb = d.shard_exchange()
                        bufs.append(edist.device_tensor(ptr, nb // 8, "<i8") if nb else None)
                    if bufs[0] is not None:
                        tot = torch.stack(bufs).sum(0)
                        for b in bufs:
                            b.copy_(tot)
                        torch.cuda.synchronize()
                        if k >= Wm:
                            xbytes[phase] += bufs[0].numel() * 8
        ph = ph / K * 1e3
        st = shards[0].stats()
        out[G] = dict(phase_ms_max=[round(float(x), 4) for x in ph.max(1)], phase_ms_mean=[round(float(x), 4) for x in ph.mean(1)],
                      tick_compute_ms=round(float(ph.max(1).sum()), 4), exchange_bytes=[int(x // K) for x in xbytes],
                      events=int(st.last_events_in), points=int(st.last_points))
        for d in shards:
            d.close() if hasattr(d, "close") else None
        del shards
    print(json.dumps(out))


if __name__ == "__main__":
    main()
