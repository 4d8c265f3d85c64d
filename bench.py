#!/usr/bin/env python
"""bench.py — mapped events/s of the ESVO hot path (Time-Surface raster + stereo mapper) on MI355X.

One "step" = one mapper tick over one 10 ms batch of a synthetic 640x480 stereo event stream
(BASELINE.json: "mapped events/sec + depth points/sec, 640x480 stereo TS"; DSEC calibration and
esvo_core/cfg/mapping/mapping_dsec.yaml parameters, throughput mode: every event of the 10 ms
slice is block-matched instead of the reference's PROCESS_EVENT_NUM = 10000):
    TS ingest (scatter of the new events of both cameras) -> TS render (both cameras) ->
    block matching -> LM refinement + culling -> window policy -> fusion -> clean -> regularisation.
All events are staged in HBM before the timed region starts.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

Prints ONE JSON line on rank 0 (see the task contract), including `roofline` for the dominant
kernel and `cpu_baseline` (the CPU oracle timed on the host cores, N=1 only).
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

from esvo_amd import calib, lib, params, rostime, synth  # noqa: E402

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md)

WORKLOADS = {
    # name: (rig, preset, rho_min, rho_max, points for ~target rate, speed)
    "dsec640x480": dict(rig="dsec", preset="mapping_dsec", rho=(0.02, 0.25), points=180000, speed=2.0),
    "upenn346x260": dict(rig="upenn", preset="mapping_upenn", rho=(0.16, 1.0), points=24000, speed=1.0),
    # SURVEY.md §8 stress row: 1280x720, 145 disparity candidates, ~100 Mev/s over both cameras
    "hd1280x720": dict(rig="hd", preset="mapping_hd", rho=(0.03, 0.45), points=185000, speed=1.5),
}

KERNEL_NAMES = ["ts_scatter", "ts_render", "bm_match", "lm_refine", "fuse", "clean", "regularize"]


def algorithmic_bytes(kernel, st, W, H, nd, fusion_radius=1, events=None, matches=None):
    """SURVEY.md §8(d) per-unit byte model x the units one launch processed (averages over the timed ticks)."""
    events = st.last_events_in if events is None else events
    matches = st.last_matches if matches is None else matches
    if kernel == "bm_match":  # 16 B event + 8 B LUT + 1 B mask + 105 B left + 7*(15+Nd-1) B strip + 48 B out
        return events * (16 + 8 + 1 + 105 + 7 * (15 + nd - 1) + 48)
    if kernel == "lm_refine":  # 48 B match + 2 x (16x8) B TS blocks + 64 B point
        return matches * (48 + 2 * 128 + 64)
    if kernel == "fuse":  # 64 B read + K cells x (52 B read + 52 B write) per window point
        k = 9 if fusion_radius else 4
        return st.last_window_points * (64 + k * 104)
    if kernel == "regularize":  # 52 B per valid pixel (+ cached taps)
        return st.last_map_size * 52
    if kernel == "ts_render":
        return W * H * 9
    if kernel == "ts_scatter":
        return 24 * 0  # per event; events per launch are reported separately
    return 0


def measured_traffic(kernel):
    """HBM bytes per launch of `kernel` from the newest committed rocprofv3 PMC summary
    (profiles/*_hbm_traffic.csv: separate FETCH_SIZE / WRITE_SIZE passes of this same command;
    FETCH_SIZE doubled per MI355X_MICROARCH.md's gfx950 correction).  None if no profile exists."""
    import csv
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "*_hbm_traffic.csv")))
    if not files:
        return None
    sym = {"lm_refine": "lm_refine_kernel", "bm_match": "bm_match_kernel", "fuse": "fuse_cells_kernel",
           "regularize": "reg_chain_kernel", "ts_render": "ts_decay_kernel", "ts_scatter": "ts_scatter_kernel"}.get(kernel)
    fetch = write = None
    with open(files[-1]) as f:
        for row in csv.DictReader(f):
            if sym and sym in row["kernel"]:
                v = float(row["avg_value_per_dispatch_KB"]) * 1024.0
                if row["counter"] == "FETCH_SIZE":
                    fetch = 2.0 * v
                elif row["counter"] == "WRITE_SIZE":
                    write = v
    if fetch is None or write is None:
        return None
    return fetch + write


def measured_valu_busy(kernel):
    """Fraction of the kernel's duration during which the vector ALUs issue, from the newest committed SQ counter pass
    (profiles/*_sq_counters.csv): SQ_ACTIVE_INST_VALU counts quad-cycles summed over the 1024 SIMDs, SQ_BUSY_CYCLES
    counts cycles summed over the 32 shader engines (MI355X_MICROARCH.md).  The hot kernels of this path are bound by
    f64 VALU issue, not by HBM or MFMA, so this is the utilisation figure that says how close they run to their limit."""
    import csv
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "*_sq_counters.csv")))
    if not files:
        return None
    sym = {"lm_refine": "lm_refine_kernel", "bm_match": "bm_match_kernel"}.get(kernel)
    v = {}
    with open(files[-1]) as f:
        for row in csv.DictReader(f):
            if sym and sym in row["kernel"]:
                v[row["counter"]] = float(row["avg_value_per_dispatch"])
    if "SQ_ACTIVE_INST_VALU" not in v or "SQ_BUSY_CYCLES" not in v:
        return None
    return (v["SQ_ACTIVE_INST_VALU"] * 4.0 / 1024.0) / (v["SQ_BUSY_CYCLES"] / 32.0)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", default="dsec640x480", choices=sorted(WORKLOADS))
    ap.add_argument("--events-per-tick", type=int, default=0, help="cap on block-matched events per tick (0 = all)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--timed-ingest", action="store_true",
                    help="stage each tick's events inside the timed loop (PCIe-inclusive rate; the default stages the whole stream first)")
    ap.add_argument("--check", action="store_true", help="also print a checksum of the final DepthMap (sharded == unsharded check)")
    args = ap.parse_args()

    import torch
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world and world > 1:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (torch.cuda.is_available() is False); there is no CPU fallback")
    if os.environ.get("ESVO_SHARED_GPU"):  # functional test of the N>1 path on a single GPU (with gloo)
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist
        backend = os.environ.get("ESVO_DIST_BACKEND", "nccl")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend)

    wl = WORKLOADS[args.workload]
    rig = calib.dataset_rig(wl["rig"])
    K, Wm = args.steps, args.warmup
    tick_s, hist_s = 0.010, 0.060
    duration = hist_s + (K + Wm + 1) * tick_s
    stream = synth.make_stream(rig, wl["points"], duration, wl["rho"][0], wl["rho"][1], seed=20250418 + 3, speed=wl["speed"])
    ev_per_tick = int(len(stream.ev_left) / duration * tick_s)
    cap = args.events_per_tick or int(ev_per_tick * 1.5) + 1024
    p, _ = params.make_params(params.PRESETS[wl["preset"]], rig, throughput_events=cap,
                              event_ring_capacity=max(1 << 22, int(len(stream.ev_left) * 1.1)))
    nd = p.bm_max_disparity - p.bm_min_disparity + 1

    shard_mode = os.environ.get("ESVO_SHARD_MODE", "tick")
    if world > 1:
        from esvo_amd import dist as edist
        # "tick": ticks dealt round-robin to the GPUs, one all-gather of frames per round (throughput scaling);
        # "band": every tick split over the GPUs by slot / image row band (latency of one tick)
        cls = edist.TickShardedEsvo if shard_mode == "tick" else edist.ShardedEsvo
        runner = cls(p, rig, rank, world, local_rank)
    else:
        runner = lib.Esvo(p, rig, device=local_rank)

    # ---- stage the whole stream in HBM (untimed); with --timed-ingest only the history before the first tick ----
    ticks = []
    for k in range(K + Wm):
        t = stream.t0_ns + int((hist_s + (k + 1) * tick_s) * 1e9)
        stamps, poses = rostime.pose_table(stream.pose, t, p.bm_half_slice_thickness)
        ticks.append((t, stamps, poses, stream.pose(t)))
    if args.timed_ingest:
        t_first = stream.t0_ns + int(hist_s * 1e9)
        bounds = [t_first] + [tk[0] for tk in ticks]
        chunks = [(stream.slice(0, a, b), stream.slice(1, a, b)) for a, b in zip(bounds[:-1], bounds[1:])]
        runner.ts_push_events(0, stream.slice(0, stream.t0_ns, t_first))
        runner.ts_push_events(1, stream.slice(1, stream.t0_ns, t_first))
    else:
        runner.ts_push_events(0, stream.ev_left)
        runner.ts_push_events(1, stream.ev_right)

    def step(k):
        t, stamps, poses, T = ticks[k]
        if args.timed_ingest:
            runner.ts_push_events(0, chunks[k][0])
            runner.ts_push_events(1, chunks[k][1])
        runner.ts_render(0, t, download=False)
        runner.ts_render(1, t, download=False)
        runner.set_observation(t, None, None, T)
        runner.tick(t, stamps, poses)

    for k in range(Wm):
        step(k)
    runner.synchronize()
    torch.cuda.synchronize()
    base = runner.stats()  # running totals so far (reading stats drains the handle: not done inside the timed loop)
    if dist:
        dist.barrier()
    t0 = time.perf_counter()
    for k in range(Wm, Wm + K):
        step(k)
    runner.synchronize()
    torch.cuda.synchronize()
    if dist:
        dist.barrier()
    dt = time.perf_counter() - t0
    st = runner.stats()
    n_events = int(st.total_events_in - base.total_events_in)
    n_points = int(st.total_points - base.total_points)
    n_matches = int(st.total_matches - base.total_matches)
    ksum = np.array(list(st.sum_ms_kernel)) - np.array(list(base.sum_ms_kernel))
    launches = max(int(st.ticks - base.ticks), 1)   # ticks THIS rank mapped (all of them unless ticks are interleaved)
    if dist:
        tt = torch.tensor([dt], device="cuda", dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
        if getattr(runner, "counts_are_local", False):  # tick-interleaved: every rank counted its own ticks
            cnt = torch.tensor([n_events, n_points], device="cuda", dtype=torch.float64)
            dist.all_reduce(cnt, op=dist.ReduceOp.SUM)
            ev_rank, mt_rank = n_events, n_matches
            n_events, n_points = int(cnt[0].item()), int(cnt[1].item())
        else:
            ev_rank, mt_rank = n_events / world, n_matches / world
    else:
        ev_rank, mt_rank = n_events, n_matches

    kavg = ksum / launches
    if ksum[7] > 0:  # TS kernels: per-render samples (some are skipped while their events are in flight), two renders per tick
        kavg[0], kavg[1] = 2 * ksum[0] / ksum[7], 2 * ksum[1] / ksum[7]
    # roofline of the dominant single kernel (slots 2 = bm_match_kernel, 3 = lm_refine_kernel; the fuse / regularize
    # slots are stages of several kernels and, like every slot, include the slowdown from the other stream's kernels)
    dom = 2 + int(np.argmax(kavg[2:4]))
    dom_name = KERNEL_NAMES[dom]
    dom_bytes = algorithmic_bytes(dom_name, st, rig.width, rig.height, nd, p.fusion_radius, events=ev_rank / launches,
                                  matches=mt_rank / launches)
    achieved = (dom_bytes / (kavg[dom] * 1e-3)) / 1e9 if kavg[dom] > 0 else 0.0
    out = {
        "metric": "mapped events/sec (stereo TS raster + block matching + LM depth refinement + fusion)",
        "value": n_events / dt,
        "unit": "events/s",
        "n_gpus": world,
        "steps": K,
        "warmup": Wm,
        "ms_per_step": dt / K * 1e3,
        "higher_is_better": True,
        "scaling": "strong",
        "vs_baseline": None,
        "dtype": "f64",
        "data": "synthetic" + (" (events staged tick by tick inside the timed loop)" if args.timed_ingest else ""),
        "depth_points_per_s": n_points / dt,
        "config": {
            "workload": f"{args.workload} synthetic stereo event stream, {len(stream.ev_left) / duration / 1e6:.1f} Mev/s/camera, "
                        f"DSEC calib + mapping_dsec.yaml params, throughput mode (all events of each 10 ms slice)"
                        if args.workload == "dsec640x480" else args.workload,
            "image": [rig.width, rig.height],
            "events_per_tick": n_events // max(K, 1),
            "disparity_range": [p.bm_min_disparity, p.bm_max_disparity],
            "parallelism": "1 GPU" if world == 1 else (f"{world} GPUs, ticks interleaved, all-gather of frames" if shard_mode == "tick"
                                                       else f"{world} GPUs, slots + image row bands"),
        },
        "kernel_ms": {KERNEL_NAMES[i]: round(float(kavg[i]), 4) for i in range(7)},
        "roofline": {
            "bound": "hbm",
            "kernel": dom_name,
            "achieved": achieved,
            "peak": HBM_PEAK_GBS,
            "unit": "GB/s",
            "frac": achieved / HBM_PEAK_GBS,
            "traffic": measured_traffic(dom_name) if args.workload == "dsec640x480" else None,  # the committed PMC passes ran this workload
            "algorithmic_bytes_per_launch": dom_bytes,
            # not an HBM- or MFMA-bound kernel: f64 vector-ALU issue is its limit
            "practical_bound": "f64 VALU issue",
            "valu_busy_frac": measured_valu_busy(dom_name) if args.workload == "dsec640x480" else None,
            "avg_launch_ms": float(kavg[dom]),
        },
    }

    # the same figures for both single-kernel slots (block matching is the one with a meaningful HBM fraction)
    out["roofline_kernels"] = []
    for slot in (2, 3):
        name = KERNEL_NAMES[slot]
        nbytes = algorithmic_bytes(name, st, rig.width, rig.height, nd, p.fusion_radius, events=ev_rank / launches, matches=mt_rank / launches)
        gbs = (nbytes / (kavg[slot] * 1e-3)) / 1e9 if kavg[slot] > 0 else 0.0
        out["roofline_kernels"].append({"kernel": name, "avg_launch_ms": float(kavg[slot]), "algorithmic_bytes_per_launch": nbytes,
                                        "achieved": gbs, "unit": "GB/s", "frac": gbs / HBM_PEAK_GBS,
                                        "valu_busy_frac": measured_valu_busy(name) if args.workload == "dsec640x480" else None})
    if args.check:
        mp_ = runner.get_map()
        if rank == 0:
            import hashlib
            key = np.ascontiguousarray(np.stack([mp_["row"].astype(np.float64), mp_["col"].astype(np.float64), mp_["inv_depth"],
                                                 mp_["variance"], mp_["age"].astype(np.float64)], axis=1))
            out["check"] = {"map_size": int(len(mp_)), "sha1": hashlib.sha1(key.tobytes()).hexdigest()}
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(rig, stream, p, ticks, Wm)
    if rank == 0:
        print(json.dumps(out))
    if dist:
        dist.destroy_process_group()


def cpu_baseline(rig, stream, p, ticks, first):
    """The CPU oracle ("port" of the reference mapper) on a bounded sample: one tick of the same
    workload, block matching + LM on all host cores, fusion single-threaded as in the reference."""
    from oracle import oracle
    try:
        oracle.build(fast=True, force=True)  # -O3 -march=native for THIS host
        fast = True
    except Exception:
        fast = False
    cores = os.cpu_count() or 1
    t, stamps, poses, T = ticks[first]
    ts = [oracle.OracleTS(rig.width, rig.height, fast=fast), oracle.OracleTS(rig.width, rig.height, fast=fast)]
    ts[0].push(stream.ev_left[stream.ns_left < t])
    ts[1].push(stream.ev_right[stream.ns_right < t])
    t0 = time.perf_counter()
    l = ts[0].render(t, map_x=rig.left.map_x, map_y=rig.left.map_y)
    r = ts[1].render(t, map_x=rig.right.map_x, map_y=rig.right.map_y)
    t_ts = time.perf_counter() - t0
    m = oracle.OracleMapper(p, rig, fast=fast)
    m.set_threads(cores)
    m.set_observation(t, l, r, T)
    m.set_poses(stamps, poses)
    idx = oracle.select_events(stream.ev_left, t, p.bm_half_slice_thickness, p.process_event_num, fast=fast)
    sample = stream.ev_left[idx]
    max_sample = 60000  # keeps the CPU leg around 10-30 s of core time
    if len(sample) > max_sample:
        sample = sample[:max_sample]
    t0 = time.perf_counter()
    m.tick(sample)
    t_map = time.perf_counter() - t0
    # the reference's own threading: NUM_THREAD_MAPPING = 4 (esvo_core/include/esvo_core/tools/utils.h:36), on a smaller sample
    ref_threads = 4
    small = sample[: max(len(sample) // 8, 1)]
    m4 = oracle.OracleMapper(p, rig, fast=fast)
    m4.set_threads(ref_threads)
    m4.set_observation(t, l, r, T)
    m4.set_poses(stamps, poses)
    t0 = time.perf_counter()
    m4.tick(small)
    t_map4 = time.perf_counter() - t0
    return {
        "value": len(sample) / (t_map + t_ts),
        "unit": "events/s",
        "cores": cores,
        "kind": "port",
        "sample": f"1 tick, {len(sample)} events block-matched (+ both TS renders {t_ts * 1e3:.0f} ms); "
                  f"BM+LM on {cores} threads, fusion/regularisation single-threaded as in the reference; "
                  f"mapper {t_map:.2f} s",
        "reference_threading": {"value": len(small) / t_map4, "unit": "events/s", "cores": ref_threads,
                                "sample": f"mapper only (no TS render): {len(small)} events, BM+LM on the reference's {ref_threads} "
                                          f"threads, {t_map4:.2f} s"},
    }


if __name__ == "__main__":
    main()
