"""Operating points beside the headline: the sustained run, reference-faithful ticks, the other streams."""
import csv
import glob
import hashlib
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

from esvo_amd import calib, lib, params, rostime, synth  # noqa: E402,F401
from .roofline import attach_measured_clock, committed_profile, roofline_rows, whole_tick_valu  # noqa: E402
from .workload import HIST_S, KERNEL_NAMES, TICK_S, WORKLOADS, make_workload, run_single, shift_events  # noqa: E402


def looped_workload(name, n_ticks, events_cap=0, base_ticks=40):
    """n_ticks ticks of a bench workload without generating n_ticks x 10 ms of synthetic stream: 60 ms of history, then a
    segment of `base_ticks` ticks played again and again with its time stamps advanced by the segment's length; the trajectory
    continues (each pass starts from the segment's first pose shifted along the direction of travel).
    Returns (rig, stream, params, ticks, stage, passes); stage(dev) pushes history + passes into a handle."""
    wl = WORKLOADS[name]
    rig, stream, p0, ticks0 = make_workload(name, base_ticks, events_cap)
    T_b = int(round(base_ticks * TICK_S * 1e9))
    t_seg0 = stream.t0_ns + int(round(HIST_S * 1e9))       # the segment covers [t_seg0, t_seg0 + T_b)
    hist = (stream.slice(0, stream.t0_ns, t_seg0), stream.slice(1, stream.t0_ns, t_seg0))
    seg = (stream.slice(0, t_seg0, t_seg0 + T_b), stream.slice(1, t_seg0, t_seg0 + T_b))
    loops = (n_ticks + base_ticks - 1) // base_ticks
    total = max(len(hist[0]) + loops * len(seg[0]), len(hist[1]) + loops * len(seg[1]))
    p, _ = params.make_params(params.PRESETS[wl["preset"]], rig, throughput_events=p0.process_event_num,
                              event_ring_capacity=int(total * 1.01) + 4096)
    dx = wl["speed"] * T_b * 1e-9                          # what the rig travels during one pass

    def pose(t_ns):
        k = max((int(t_ns) - t_seg0) // T_b, 0) if t_ns >= t_seg0 else 0
        T = stream.pose(int(t_ns) - k * T_b).copy()
        T[0, 3] += k * dx
        return T

    def stage(dev):
        for cam in (0, 1):
            dev.ts_push_events(cam, hist[cam])
            for k in range(loops):
                dev.ts_push_events(cam, shift_events(seg[cam], k * T_b) if k else seg[cam])

    ticks = []
    for k in range(n_ticks):
        t = t_seg0 + (k + 1) * int(round(TICK_S * 1e9))
        stamps, poses = rostime.pose_table(pose, t, p.bm_half_slice_thickness)
        ticks.append((t, stamps, poses, pose(t)))
    return rig, stream, p, ticks, stage, loops


def sustained_point(name, n_ticks, device, prof, events_cap=0, base_ticks=40, hook=None):
    """The headline workload for >= 2 s of wall time instead of 20 ticks (28 ms): long enough for the chip's power management
    to settle, with the shader clock measured inside the run.  Generating 16 s of synthetic stream would take minutes of
    numpy, so the stationary stream is LOOPED: 60 ms of history, then a segment of `base_ticks` ticks played again and again
    with its time stamps advanced by the segment's length; the trajectory continues (the rig keeps moving at its speed, each
    pass starts from the segment's first pose shifted along the direction of travel), so windows, propagation and fusion see
    a continuous motion.  At each seam the scene jumps back to the segment's first arrangement: for ~6 ticks (the 60 ms the
    Time Surfaces remember) the surfaces mix two arrangements -- `seam_ticks` says how many ticks that concerns.  Everything is
    staged in HBM before the timed region, as for `value`."""
    n_warm = 8
    t_gen = time.perf_counter()
    rig, stream, p, ticks, stage, loops = looped_workload(name, n_ticks + n_warm, events_cap, base_ticks)
    dev = lib.Esvo(p, rig, device=device)
    stage(dev)
    t_gen = time.perf_counter() - t_gen
    run_single(dev, stream, ticks, 0, n_warm)
    dev.synchronize()
    b = dev.stats()
    marks = []
    t0 = time.perf_counter()
    for k in range(n_warm, n_warm + n_ticks):
        t, stamps, poses, T = ticks[k]
        dev.tick_resident(t, T, stamps, poses)
        if hook is not None:   # tools/regime_probe.py: a disturbance in the middle of the run
            hook(k - n_warm, dev)
        if (k - n_warm) % 100 == 99:   # host time stamps without a synchronisation: the lazy tick paces the host to the device
            marks.append(time.perf_counter())
    dev.synchronize()
    dt = time.perf_counter() - t0
    s = dev.stats()
    if hook is not None and hasattr(hook, "at_end"):
        hook.at_end(dev)
    dev.close()
    ev = int(s.total_events_in - b.total_events_in)
    sclk, per_xcd = s.sclk_mhz(b)
    ks = np.array(s.kernel_ms_mean(b))   # (per sampled tick: stage timings are sampled, esvo_hip.h stage_timing_samples)
    win = np.diff(np.array([t0] + marks)) / 100.0 * 1e3    # ms per tick over windows of 100 ticks
    res = {"windows_ms": [round(float(w), 4) for w in win], "events_per_s": ev / dt, "ms_per_tick": dt / n_ticks * 1e3, "ticks": n_ticks, "wall_s": dt,
           "events_per_tick": ev // n_ticks, "depth_points_per_s": int(s.total_points - b.total_points) / dt,
           "matches_per_tick": int(s.total_matches - b.total_matches) // n_ticks,
           "pipeline_resyncs": int(s.pipeline_resyncs - b.pipeline_resyncs),
           "sclk_mhz": sclk, "sclk_mhz_per_xcd": per_xcd, "sclk_samples": int(s.clk_samples - b.clk_samples),
           "ms_per_tick_100tick_windows": {"first": float(win[0]), "min": float(win.min()), "median": float(np.median(win)),
                                            "max": float(win.max()), "last": float(win[-1])} if len(win) else None,
           "kernel_ms": {"bm_match": round(float(ks[2]), 4), "lm_refine": round(float(ks[3]), 4), "fuse": round(float(ks[4]), 4),
                         "regularize": round(float(ks[6]), 4)},
           "loop": {"segment_ticks": base_ticks, "passes": loops, "seam_ticks": int(loops * round(HIST_S / TICK_S)),
                    "staging_s": round(t_gen, 2)},
           "note": "the headline workload looped (a 0.4 s segment of the stationary stream replayed with advancing stamps and a "
                   "continuing trajectory), all events resident in HBM before the timed region; sclk = shader clock of the LM "
                   "kernel's waves measured inside this run (s_memtime / s_memrealtime)"}
    whole = whole_tick_valu(prof, res["ms_per_tick"])
    if whole is not None and sclk:
        attach_measured_clock(whole, sclk)
        res["valu_whole_tick"] = whole
    return res


def _points(device, which):
    """Secondary figures of the same JSON line.  which = "reference_faithful" (default run): the two tick sizes the ROS node
    really runs (PROCESS_EVENT_NUM 10000 on DSEC, 1000 on upenn), each as the time of one tick completed on its own (latency:
    nothing is in flight beside it) and as the sustained time per tick with two ticks in flight.  which = "extras"
    (bench.py --extras): the 346x260 stream of north_star in throughput mode, the closed loop, the 1280x720 stress stream, the
    headline workload with PCIe staging inside the timed loop."""
    out = {}

    def throughput(name, n, check=False, timed_ingest=False, pinned=False):
        # (40 ticks of stream whatever n is: the noise events are drawn over the stream's whole duration, so streams of different
        #  lengths are different realisations, and this leg's figure must not depend on which leg of the process asked first --
        #  the 346x260 point read 0.36 or 0.89 ms per tick depending on that: profiles/r06_extras_note.txt)
        rig, stream, p, ticks = make_workload(name, max(n + 3, 40) if name != "hd1280x720" else n + 3)
        ticks = ticks[: n + 3]
        dev = lib.Esvo(p, rig, device=device)
        pins = []
        if timed_ingest:   # PCIe-inclusive: only the history is resident, every tick's events are staged inside the loop
            t_first = stream.t0_ns + int(HIST_S * 1e9)
            bounds = [t_first] + [tk[0] for tk in ticks]
            chunks = [(stream.slice(0, a, b), stream.slice(1, a, b)) for a, b in zip(bounds[:-1], bounds[1:])]
            if pinned:   # the node's message pool in pinned memory (esvo_host_alloc): filled before the timed region
                for k, pair in enumerate(chunks):
                    held = []
                    for ev in pair:
                        pe = lib.PinnedEvents(len(ev))
                        pe.array[:] = ev
                        pins.append(pe)
                        held.append(pe.array)
                    chunks[k] = tuple(held)
            dev.ts_push_events(0, stream.slice(0, stream.t0_ns, t_first))
            dev.ts_push_events(1, stream.slice(1, stream.t0_ns, t_first))
        else:
            dev.ts_push_events(0, stream.ev_left)
            dev.ts_push_events(1, stream.ev_right)

        def run(a, b):
            for k in range(a, b):
                if timed_ingest and pinned:   # enqueue the DMA and go on: the tick queues behind it on the device
                    dev.ts_push_events_async(0, chunks[k][0])
                    dev.ts_push_events_async(1, chunks[k][1])
                elif timed_ingest:
                    dev.ts_push_events(0, chunks[k][0])
                    dev.ts_push_events(1, chunks[k][1])
                t, stamps, poses, T = ticks[k]
                dev.tick_resident(t, T, stamps, poses)
        run(0, 3)
        dev.synchronize()
        b = dev.stats()
        t0 = time.perf_counter()
        run(3, n + 3)
        dev.synchronize()
        dt = time.perf_counter() - t0
        s = dev.stats()
        if pinned:
            dev.ts_push_wait(0)
            dev.ts_push_wait(1)
        dev.close()
        for pe in pins:
            pe.free()
        ev = int(s.total_events_in - b.total_events_in)
        res = {"events_per_s": ev / dt, "ms_per_tick": dt / n * 1e3, "events_per_tick": ev // n,
               "depth_points_per_s": int(s.total_points - b.total_points) / dt}
        if timed_ingest:
            res["note"] = ("host-to-device staging of each tick's events (2 x 16 B/event) inside the timed loop, " +
                           ("from pinned buffers through esvo_ts_push_events_async: the DMA overlaps the running tick" if pinned
                            else "from pageable memory through the synchronous esvo_ts_push_events"))
            return res
        ka = np.array(s.kernel_ms_mean(b))   # (per sampled tick / render: esvo_hip.h stage_timing_samples)
        scat = (int(s.events_scattered[0]) + int(s.events_scattered[1])) - (int(b.events_scattered[0]) + int(b.events_scattered[1]))
        res["kernel_ms"] = {KERNEL_NAMES[i]: round(float(ka[i]), 4) for i in range(7)}
        res["roofline_kernels"] = roofline_rows(s, ka, rig, p.bm_max_disparity - p.bm_min_disparity + 1, p, committed_profile(name),
                                                ev / n, int(s.total_matches - b.total_matches) / n, scat / n)
        if check:   # the first timed tick replayed on a fresh handle against the CPU oracle (as bench.py --check does)
            if (os.cpu_count() or 1) >= 64:
                from .baselines import check_against_oracle
                res["check_oracle_equal"] = bool(check_against_oracle(rig, stream, p, ticks, 3, device)["equal"])
            else:   # four 5e5-event ticks of the oracle take minutes on a small host: the default run must stay short
                res["check_oracle_equal"] = None
                res["check_note"] = "skipped on a host with fewer than 64 threads (bench.py --workload hd1280x720 --check runs it)"
        return res

    def latency(name, n_events, n):
        # (W warm-up ticks: a handle records its stage-timing events -- ~5 us of queue time each -- on its first 8 ticks that run
        #  alone and on one in 31 afterwards, esvo_hip.h stage_timing_samples; the timed ticks are those of a running node)
        W = 9
        rig, stream, p, ticks = make_workload(name, max(n + W, 40), events_cap=n_events)
        ticks = ticks[: n + W]
        dev = lib.Esvo(p, rig, device=device)
        dev.ts_push_events(0, stream.ev_left)
        dev.ts_push_events(1, stream.ev_right)
        run_single(dev, stream, ticks, 0, W, sync_each=True)
        t0 = time.perf_counter()
        run_single(dev, stream, ticks, W, n + W, sync_each=True)
        dt = time.perf_counter() - t0
        s = dev.stats()
        dev.close()
        # the same ticks with two in flight (no synchronisation inside the loop): the sustained rate of small ticks
        dev = lib.Esvo(p, rig, device=device)
        dev.ts_push_events(0, stream.ev_left)
        dev.ts_push_events(1, stream.ev_right)
        run_single(dev, stream, ticks, 0, 6)
        dev.synchronize()
        t0 = time.perf_counter()
        run_single(dev, stream, ticks, 6, n + 6)
        dev.synchronize()
        dp = time.perf_counter() - t0
        dev.close()
        return {"ms_per_tick": dt / n * 1e3, "ms_per_tick_pipelined": dp / n * 1e3, "events_per_tick": int(s.last_events_in),
                "points_per_tick": int(s.last_points)}

    def closed_loop():
        # BASELINE.json configs[2]: 346x260, the full mapping + tracking loop on one GPU -- SGM bootstrap, then per cycle both
        # Time Surfaces, the tracker's registration (residuals, Jacobian and their products J^T J / J^T f on the device in one
        # launch per iteration; the 6 x 6 Gauss-Newton update on the host in C++ inside the library -- the reference keeps its
        # optimiser on the host too) and the mapper tick fed with the TRACKED poses
        from esvo_amd import closed_loop as cl
        r = cl.run(n_ticks=15)
        med = lambda v: float(np.median(np.asarray(v[3:])))  # steady state: past the first cycles
        return {"ms_per_cycle": med(r["cycle_ms"]), "ms_tracking": med(r["track_ms"]), "ms_mapping": med(r["map_ms"]),
                "cycles": len(r["cycle_ms"]), "path_mm": r["gt_len"][-1] * 1e3, "final_position_error_mm": r["pos_err"][-1] * 1e3,
                "depth_points_per_cycle": int(np.median(r["points"])), "map_median_abs_inv_depth_error": r["map_median_abs_err"],
                "note": "synthetic 346x260 scene, poses from the tracker only (bootstrap pose given); tracker optimiser = "
                        "esvo_track_register (host C++ over esvo_track_normal_equations_batch: Levenberg-damped steps with Eigen's "
                        "accept test, the three trial dampings of an iteration in one launch, 12 iterations at most)"}

    def point(key, fn, *a, **kw):   # an extra never takes the headline down with it: its failure is reported in its place
        try:
            out[key] = fn(*a, **kw)
        except Exception as e:  # noqa: BLE001
            out[key] = {"error": f"{type(e).__name__}: {e}"}

    if which == "reference_faithful":
        point("dsec640x480_reference_faithful_10000", latency, "dsec640x480", 10000, 20)
        point("upenn346x260_reference_faithful_1000", latency, "upenn346x260", 1000, 20)
        return out
    point("upenn346x260_throughput", throughput, "upenn346x260", 20)
    point("upenn346x260_closed_loop", closed_loop)
    # SURVEY.md section 8 stress row: 1280x720, 145 disparity candidates, 100 Mev/s over both cameras, with the oracle equality flag
    point("hd1280x720_throughput", throughput, "hd1280x720", 6, check=True)
    # the headline workload with the PCIe transfer of every tick's events inside the timed loop (never `value`)
    point("dsec640x480_with_timed_ingest", throughput, "dsec640x480", 20, timed_ingest=True)
    point("dsec640x480_with_timed_ingest_pinned", throughput, "dsec640x480", 20, timed_ingest=True, pinned=True)
    return out


def reference_faithful_points(device):
    return _points(device, "reference_faithful")


def extra_operating_points(device):
    return _points(device, "extras")
