#!/usr/bin/env python
"""bench.py — mapped events/s of the ESVO hot path (Time-Surface raster + stereo mapper) on MI355X.

One "step" = one mapper tick over one 10 ms batch of a synthetic 640x480 stereo event stream
(BASELINE.json: "mapped events/sec + depth points/sec, 640x480 stereo TS"; DSEC calibration and
esvo_core/cfg/mapping/mapping_dsec.yaml parameters, throughput mode: every event of the 10 ms
slice is block-matched instead of the reference's PROCESS_EVENT_NUM = 10000):
    TS ingest (scatter of the new events of both cameras) -> TS render (both cameras) ->
    block matching -> LM refinement + culling -> window policy -> fusion -> clean -> regularisation.
All events are staged in HBM before the timed region starts.  The stream is STATIONARY (SURVEY.md section 8(d):
20 Mev/s per camera = 200 k events per tick, whatever --steps is): a periodic scene seen from a rig in uniform motion.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

With N GPUs the job maps N x K ticks of one stream, K per GPU (weak scaling): ticks are dealt round-robin and the
ranks all-gather their frames (esvo_amd/dist.py), or, with ESVO_SHARD_MODE=band, every tick is split over the GPUs.

Prints ONE JSON line on rank 0 (see the task contract), including `roofline` for the dominant
kernel and `cpu_baseline` (the CPU oracle timed on the host cores, N=1 only).
"""
import argparse
import csv
import glob
import hashlib
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

from esvo_amd import calib, lib, params, rostime, synth  # noqa: E402

HBM_PEAK_GBS = 8000.0       # MI355X HBM3E spec peak (MI355X_MICROARCH.md)
VALU_PEAK_INST_S = 256 * 4 * 2.4e9 / 4.0  # wave64 VALU instructions/s: 1024 SIMDs, one f64/f32 instruction per 4 cycles, 2.4 GHz

TICK_S, HIST_S = 0.010, 0.060
WORKLOADS = {
    # name: rig, preset, rho range, scene points for the target event rate, rig speed (m/s)
    "dsec640x480": dict(rig="dsec", preset="mapping_dsec", rho=(0.02, 0.25), points=237500, speed=2.0,
                        note="DSEC calib + mapping_dsec.yaml params, 20 Mev/s/camera"),
    "upenn346x260": dict(rig="upenn", preset="mapping_upenn", rho=(0.16, 1.0), points=27800, speed=1.0,
                         note="upenn calib + mapping_upenn.yaml params, 2 Mev/s/camera"),
    # SURVEY.md section 8 stress row: 1280x720, 145 disparity candidates, 100 Mev/s over both cameras
    "hd1280x720": dict(rig="hd", preset="mapping_hd", rho=(0.03, 0.45), points=235000, speed=1.5,
                       note="synthetic HD rig (f b = 300 px m), mapping_dsec.yaml params with 145 disparities, 50 Mev/s/camera"),
}

KERNEL_NAMES = ["ts_scatter", "ts_render", "bm_match", "lm_refine", "fuse", "clean", "regularize"]
KERNEL_SYMBOLS = {"lm_refine": "lm_refine_kernel", "bm_match": "bm_match_kernel", "fuse": "fuse_cells_kernel",
                  "regularize": "reg_apply_kernel", "ts_render": "ts_render_fused_kernel", "ts_scatter": "ts_scatter_kernel"}


R01_POINTS = {"dsec640x480": 180000, "upenn346x260": 24000, "hd1280x720": 185000}


_STREAMS = {}   # (workload, r01_scene) -> (ticks it covers, SynthStream): the same seeded stream serves every operating point of a run


def make_workload(name, n_ticks, events_cap=0, r01_scene=False, share=None):
    """(rig, stream, params, ticks) of a bench workload: n_ticks ticks of 10 ms after 60 ms of history.
    r01_scene: round 1's thinning, swaying scene (only for like-for-like comparisons with round-1 figures).
    share = (rank, barrier): an N-rank job generates the (identical, seeded) stream ONCE -- rank 0 writes the two event
    arrays to a scratch file, the others read them after the barrier -- instead of N times in parallel on one host
    (an 8-GPU weak-scaling run maps 200 ticks = 2 s of stream = 40 M events per camera: ~100 s of numpy per rank).
    Within one process a stream generated for more ticks is reused for a request of fewer (it is seeded: the ticks are the same)."""
    wl = WORKLOADS[name]
    rig = calib.dataset_rig(wl["rig"])
    traj = None
    if not r01_scene:
        traj = synth.Trajectory(speed=wl["speed"], sway=0.002, yaw=0.0005, t0_s=10.0)

    def generate():
        duration = HIST_S + (n_ticks + 1) * TICK_S
        if r01_scene:
            return synth.make_stream(rig, R01_POINTS[name], duration, wl["rho"][0], wl["rho"][1], seed=20250418 + 3, speed=wl["speed"])
        return synth.make_stream(rig, wl["points"], duration, wl["rho"][0], wl["rho"][1], seed=20250418 + 3, speed=wl["speed"],
                                 stationary=True, traj=traj)
    if share is None:
        have = _STREAMS.get((name, r01_scene))
        if have is None or have[0] < n_ticks:
            # ESVO_BENCH_STREAM_CACHE=<dir> (profiling scripts that run bench.py several times on one box): the seeded stream on disk
            cdir = os.environ.get("ESVO_BENCH_STREAM_CACHE")
            cpath = os.path.join(cdir, f"esvo_stream_{name}_{n_ticks}_{int(r01_scene)}.npz") if cdir else None
            if cpath and os.path.exists(cpath):
                z = np.load(cpath)
                gen = synth.SynthStream(rig, z["l"], z["r"], traj or synth.Trajectory(speed=wl["speed"], t0_s=10.0), int(z["t"][0]), int(z["t"][1]), None)
            else:
                gen = generate()
                if cpath:
                    os.makedirs(cdir, exist_ok=True)
                    with open(cpath + ".tmp", "wb") as f:
                        np.savez(f, l=gen.ev_left, r=gen.ev_right, t=np.array([gen.t0_ns, gen.t1_ns], np.int64))
                    os.replace(cpath + ".tmp", cpath)
            have = (n_ticks, gen)
            _STREAMS[(name, r01_scene)] = have
        stream = have[1]
    else:
        import tempfile
        rank, barrier = share
        path = os.path.join(tempfile.gettempdir(), f"esvo_bench_stream_{name}_{n_ticks}_{int(r01_scene)}_{os.environ.get('MASTER_PORT', '0')}.npz")
        if rank == 0:
            stream = generate()
            with open(path + ".tmp", "wb") as f:
                np.savez(f, l=stream.ev_left, r=stream.ev_right, t=np.array([stream.t0_ns, stream.t1_ns], np.int64))
            os.replace(path + ".tmp", path)
        barrier()
        if rank != 0:
            z = np.load(path)
            stream = synth.SynthStream(rig, z["l"], z["r"], traj or synth.Trajectory(speed=wl["speed"], t0_s=10.0), int(z["t"][0]), int(z["t"][1]), None)
        barrier()
        if rank == 0:
            os.remove(path)
    duration = (stream.t1_ns - stream.t0_ns) * 1e-9
    ev_per_tick = int(len(stream.ev_left) / duration * TICK_S)
    cap = events_cap or int(ev_per_tick * 1.25) + 1024
    p, _ = params.make_params(params.PRESETS[wl["preset"]], rig, throughput_events=cap,
                              event_ring_capacity=max(1 << 22, int(len(stream.ev_left) * 1.05) + 4096))
    ticks = []
    for k in range(n_ticks):
        t = stream.t0_ns + int(round((HIST_S + (k + 1) * TICK_S) * 1e9))
        stamps, poses = rostime.pose_table(stream.pose, t, p.bm_half_slice_thickness)
        ticks.append((t, stamps, poses, stream.pose(t)))
    return rig, stream, p, ticks


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
    return 0


TRAFFIC_NOTE = ("traffic = 2 x FETCH_SIZE + WRITE_SIZE of separate rocprofv3 --pmc passes (committed CSV named in traffic_source). The x2 on "
                "FETCH_SIZE is MI355X_MICROARCH.md's gfx950 correction, calibrated for streaming 16-byte-per-lane loads only; for byte "
                "gathers and atomics (ts_scatter, scatter_records, reg_apply's LDS staging) it is an uncalibrated upper bound")


def roofline_rows(st, kavg, rig, nd, p, prof, events, matches, scattered):
    """per-kernel roofline entries: Time-Surface scatter and render (the HBM-bound stage), block matching, LM"""
    rows = []
    ts_bytes = {"ts_scatter": scattered * 24.0,                    # 16 B event read + 8 B SAE atomic
                "ts_render": 2.0 * rig.width * rig.height * 9.0}   # both cameras: 8 B SAE read + 1 B mono8 write per pixel
    for slot in (0, 1, 2, 3):
        name = KERNEL_NAMES[slot]
        if slot < 2:
            nbytes = ts_bytes[name]
        else:
            nbytes = algorithmic_bytes(name, st, rig.width, rig.height, nd, p.fusion_radius, events=events, matches=matches)
        gbs = (nbytes / (kavg[slot] * 1e-3)) / 1e9 if kavg[slot] > 0 else 0.0
        tr, vl = profile_figures(prof, name, float(kavg[slot]))
        rows.append({"kernel": name, "avg_launch_ms": float(kavg[slot]), "algorithmic_bytes_per_launch": nbytes,
                     "achieved": gbs, "unit": "GB/s", "frac": gbs / HBM_PEAK_GBS, "traffic": tr,
                     "valu_frac": None if vl is None else vl["frac"]})
    return rows


def committed_profile(workload):
    """The newest committed rocprofv3 round (profiles/<tag>_meta.json names the command and workload it ran): per-kernel
    HBM bytes (separate FETCH_SIZE / WRITE_SIZE passes; FETCH_SIZE doubled per MI355X_MICROARCH.md's gfx950 correction) and
    SQ counters per launch.  Counter values are properties of (build, workload): they are only attached to a bench line of
    the same workload, and the line says which files they came from."""
    metas = sorted(glob.glob(os.path.join(ROOT, "profiles", "*_meta.json")))
    for path in reversed(metas):
        try:
            meta = json.load(open(path))
        except Exception:
            continue
        if meta.get("workload") != workload:
            continue
        tag = os.path.basename(path)[:-len("_meta.json")]
        out = {"tag": tag, "hbm": {}, "sq": {}, "files": []}
        for kind, fn in (("hbm", f"{tag}_hbm_traffic.csv"), ("sq", f"{tag}_sq_counters.csv")):
            fp = os.path.join(ROOT, "profiles", fn)
            if not os.path.exists(fp):
                continue
            out["files"].append("profiles/" + fn)
            with open(fp) as f:
                for row in csv.DictReader(f):
                    k = row["kernel"]
                    if kind == "hbm":
                        out["hbm"].setdefault(k, {})[row["counter"]] = float(row["avg_value_per_dispatch_KB"]) * 1024.0
                    else:
                        out["sq"].setdefault(k, {})[row["counter"]] = float(row["avg_value_per_dispatch"])
                        out.setdefault("sq_dispatches", {})[k] = int(row.get("dispatches") or 0)
        return out
    return None


def profile_figures(prof, kernel, launch_ms):
    """(traffic bytes per launch, VALU block) of `kernel` from a committed profile, or (None, None)"""
    if prof is None:
        return None, None
    sym = KERNEL_SYMBOLS.get(kernel)
    traffic = valu = None
    for k, v in prof["hbm"].items():
        if sym and sym in k and "FETCH_SIZE" in v and "WRITE_SIZE" in v:
            traffic = 2.0 * v["FETCH_SIZE"] + v["WRITE_SIZE"]
    for k, v in prof["sq"].items():
        if sym and sym in k and "SQ_INSTS_VALU" in v:
            insts = v["SQ_INSTS_VALU"]
            rate = insts / (launch_ms * 1e-3) if launch_ms > 0 else 0.0
            valu = {"wave_insts_per_launch": insts, "achieved": rate, "peak": VALU_PEAK_INST_S, "unit": "wave64 VALU instructions/s",
                    "frac": rate / VALU_PEAK_INST_S,
                    "note": "instruction count per launch from the committed SQ_INSTS_VALU pass of this workload (deterministic for "
                            "a build), divided by this run's HIP-event launch time; peak = 1024 SIMDs x 2.4 GHz / 4 cycles per "
                            "f64 instruction"}
            if "SQ_ACTIVE_INST_VALU" in v and "SQ_BUSY_CYCLES" in v:
                valu["busy_frac_profiled"] = (v["SQ_ACTIVE_INST_VALU"] * 4.0 / 1024.0) / (v["SQ_BUSY_CYCLES"] / 32.0)
    return traffic, valu


def whole_tick_valu(prof, tick_ms):
    """VALU wave-instructions of EVERY kernel of a tick (committed SQ_INSTS_VALU pass: per-launch average x launches per tick,
    the LM kernel's dispatch count = the ticks of that pass) against what the chip can issue in this run's tick time.  The
    three stages overlap on three queues, so this -- not one kernel's rate -- is the figure that bounds the tick."""
    if prof is None or not prof.get("sq_dispatches"):
        return None
    ticks = max([n for k, n in prof["sq_dispatches"].items() if "lm_refine_kernel" in k] or [0])
    if ticks <= 0 or tick_ms <= 0:
        return None
    total = sum(v["SQ_INSTS_VALU"] * prof["sq_dispatches"].get(k, 0) for k, v in prof["sq"].items() if "SQ_INSTS_VALU" in v) / ticks
    rate = total / (tick_ms * 1e-3)
    return {"wave_insts_per_tick": total, "achieved": rate, "peak": VALU_PEAK_INST_S, "frac": rate / VALU_PEAK_INST_S,
            "note": "all kernels of a tick; peak as above (nominal 2.4 GHz); `frac_at_measured_clock` re-prices it at the shader clock "
                    "measured inside this run (DESIGN.md section 5: 2.3 GHz -- round 3's 1.75 GHz estimate was wrong)"}


def map_sha1(mp_):
    key = np.ascontiguousarray(np.stack([mp_["row"].astype(np.float64), mp_["col"].astype(np.float64), mp_["inv_depth"],
                                         mp_["variance"], mp_["age"].astype(np.float64)], axis=1))
    return hashlib.sha1(key.tobytes()).hexdigest()


def run_single(dev, stream, ticks, first, last, sync_each=False):
    for k in range(first, last):
        t, stamps, poses, T = ticks[k]
        dev.tick_resident(t, T, stamps, poses)   # = ts_render x2 + set_observation + tick
        if sync_each:
            dev.synchronize()


def self_launch(n):
    """`python bench.py --gpus N` without a launcher: re-execute this command as N ranks under torch.distributed.run (one
    process per GPU, rendezvous on 127.0.0.1, a free port).  Fails loudly when the node has fewer than N devices -- a
    single-rank number must never be reported as an N-GPU point.  (ESVO_SHARED_GPU=1: N ranks share device 0, functional
    tests only.)"""
    import socket
    import subprocess
    import torch
    have = torch.cuda.device_count() if torch.cuda.is_available() else 0
    if have < 1:
        print("bench.py needs an MI355X (torch.cuda.is_available() is False); there is no CPU fallback", file=sys.stderr)
        return 2
    if have < n and not os.environ.get("ESVO_SHARED_GPU"):
        print(f"bench.py --gpus {n}: only {have} GPU(s) visible on this node (hipGetDeviceCount); refusing to run fewer ranks "
              f"than asked for", file=sys.stderr)
        return 2
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # dmabuf IPC: RCCL across processes needs it on this driver
    env.setdefault("OMP_NUM_THREADS", str(max((os.cpu_count() or n) // n, 1)))
    env["ESVO_BENCH_SELF_LAUNCHED"] = "1"
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.call(cmd, env=env)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", default="dsec640x480", choices=sorted(WORKLOADS))
    ap.add_argument("--events-per-tick", type=int, default=0, help="cap on block-matched events per tick (0 = all)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="skip the secondary operating points (346x260, reference-faithful ticks)")
    ap.add_argument("--timed-ingest", action="store_true",
                    help="stage each tick's events inside the timed loop (PCIe-inclusive rate; the default stages the whole stream first)")
    ap.add_argument("--r01-scene", action="store_true", help="round 1's thinning scene (like-for-like comparisons only)")
    ap.add_argument("--strong", action="store_true", help="N GPUs share K ticks in total instead of mapping K ticks each")
    ap.add_argument("--check", action="store_true",
                    help="replay up to the first timed tick on a fresh handle and compare its DepthMap with the CPU oracle's (SHA-1)")
    ap.add_argument("--selftest", action="store_true",
                    help="N-GPU plumbing check in < 30 s instead of the benchmark: who is there (rank, device, bus id), the RCCL the library "
                         "resolved, one all-gather round trip, and the DepthMap SHA-1 of both N-GPU modes against the one-GPU run")
    ap.add_argument("--no-parity", action="store_true",
                    help="skip the `parity` block of the line (oracle equality of the first timed tick; IoU / RMSE against the reference node)")
    ap.add_argument("--sustained-ticks", type=int, default=1600,
                    help="ticks of the sustained operating point (the headline workload looped for >= 2 s of wall time); 0 skips it")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        raise SystemExit(self_launch(args.gpus))

    # stdout carries the ONE JSON line and nothing else: whatever libraries write to fd 1 on the way (RCCL prints a version
    # banner at communicator creation) is routed to stderr; the line itself goes to the saved descriptor.
    sys.stdout.flush()
    json_out = os.fdopen(os.dup(1), "w")
    os.dup2(2, 1)

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
    K, Wm = args.steps, args.warmup
    ranks_seen = None
    rccl = None
    if dist:  # who is really there: one line per rank (device ordinal, bus id) gathered onto rank 0's JSON line
        props = torch.cuda.get_device_properties(local_rank)
        mine = {"rank": rank, "local_rank": local_rank, "device": torch.cuda.current_device(), "name": props.name,
                "pci_bus_id": getattr(props, "pci_bus_id", None), "pid": os.getpid()}
        ranks_seen = [None] * world
        dist.all_gather_object(ranks_seen, mine)
        if os.environ.get("ESVO_DIST_BACKEND", "nccl") == "nccl":
            try:
                ver, path = lib.comm_rccl_info()   # the RCCL the C library resolved (dlopen), not torch's notion of it
                rccl = {"version_code": ver, "library": path}
            except Exception as e:  # noqa: BLE001
                rccl = {"error": str(e)}

    def measure(shard_mode, strong):
        """warm-up + the timed region for one way of putting the ranks on the stream; returns the raw figures"""
        # weak scaling: K timed (and Wm warm-up) ticks PER GPU in the tick-interleaved mode; the band mode splits every tick
        per_gpu = world if (world > 1 and shard_mode == "tick" and not strong) else 1
        n_ticks = (K + Wm) * per_gpu
        # (at least 40 ticks whatever N is: the seeded stream -- its noise events are drawn over the whole duration -- is then the
        #  same for an N-rank run and the one-rank run it is compared with, and serves the sustained point's 40-tick segment too)
        rig, stream, p, ticks = make_workload(args.workload, max(n_ticks, 40), args.events_per_tick,
                                              r01_scene=args.r01_scene, share=(rank, dist.barrier) if dist else None)
        duration = (stream.t1_ns - stream.t0_ns) * 1e-9

        native = (world > 1 and os.environ.get("ESVO_DIST_BACKEND", "nccl") == "nccl"
                  and os.environ.get("ESVO_NATIVE_COMM", "1") != "0")
        comm_note = None

        def make_runner(use_native):
            if world == 1:
                return lib.Esvo(p, rig, device=local_rank)
            from esvo_amd import dist as edist
            # "tick": ticks dealt round-robin to the GPUs, one all-gather of frames per round (throughput scaling);
            # "band": every tick split over the GPUs by slot / image row band (latency of one tick).
            # The exchange runs inside libesvo_hip.so (esvo_comm_*: RCCL called from C); the torch.distributed drivers remain
            # for other backends (gloo on a shared GPU: tests), with ESVO_NATIVE_COMM=0, and as the fallback below.
            if use_native:
                cls = edist.NativeTickSharded if shard_mode == "tick" else edist.NativeBandSharded
            else:
                cls = edist.TickShardedEsvo if shard_mode == "tick" else edist.ShardedEsvo
            return cls(p, rig, rank, world, local_rank)

        if args.timed_ingest:
            t_first = stream.t0_ns + int(HIST_S * 1e9)
            bounds = [t_first] + [tk[0] for tk in ticks]
            chunks = [(stream.slice(0, a, b), stream.slice(1, a, b)) for a, b in zip(bounds[:-1], bounds[1:])]

        def stage(r):
            # the whole stream goes to HBM before the timed region; with --timed-ingest only the history before the first tick
            if args.timed_ingest:
                r.ts_push_events(0, stream.slice(0, stream.t0_ns, t_first))
                r.ts_push_events(1, stream.slice(1, stream.t0_ns, t_first))
            else:
                r.ts_push_events(0, stream.ev_left)
                r.ts_push_events(1, stream.ev_right)

        def step(k):
            t, stamps, poses, T = ticks[k]
            if args.timed_ingest:
                runner.ts_push_events(0, chunks[k][0])
                runner.ts_push_events(1, chunks[k][1])
            if hasattr(runner, "tick_resident"):
                runner.tick_resident(t, T, stamps, poses)   # = ts_render x2 + set_observation + tick, one call
            else:                                            # the multi-GPU drivers see the four calls
                runner.ts_render(0, t, download=False)
                runner.ts_render(1, t, download=False)
                runner.set_observation(t, None, None, T)
                runner.tick(t, stamps, poses)

        n_warm, n_all = Wm * per_gpu, (Wm + K) * per_gpu
        runner = None
        for attempt_native in ([True, False] if native else [False]):
            failed = None
            try:
                runner = make_runner(attempt_native)
                stage(runner)
                for k in range(n_warm):
                    step(k)
                runner.synchronize()
            except Exception as e:  # an error code from the C library (a hang or a fault inside RCCL cannot be caught here)
                failed = f"{type(e).__name__}: {e}"
            if dist:  # every rank takes the same path
                flag = torch.tensor([1.0 if failed else 0.0], device="cuda")
                dist.all_reduce(flag, op=dist.ReduceOp.MAX)
                if flag.item() > 0 and not failed:
                    failed = "another rank failed"
            if not failed:
                native = attempt_native
                break
            if not attempt_native:
                raise SystemExit(f"multi-GPU warm-up failed: {failed}")
            comm_note = f"esvo_comm_* path failed in warm-up ({failed}); fell back to the torch.distributed driver"
            print(f"[bench rank {rank}] {comm_note}", file=sys.stderr)
            runner = None
        torch.cuda.synchronize()
        base = runner.stats()  # running totals so far (reading stats drains the handle: not done inside the timed loop)
        if dist:
            dist.barrier()
        t0 = time.perf_counter()
        for k in range(n_warm, n_all):
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
        return dict(per_gpu=per_gpu, rig=rig, stream=stream, p=p, ticks=ticks, duration=duration, native=native, comm_note=comm_note,
                    runner=runner, dt=dt, st=st, base=base, n_events=n_events, n_points=n_points, n_matches=n_matches, ksum=ksum,
                    launches=launches, ev_rank=ev_rank, mt_rank=mt_rank, shard_mode=shard_mode)

    if args.selftest:
        res = selftest(rank, world, local_rank, dist, ranks_seen, rccl)
        if rank == 0:
            print(json.dumps(res), file=json_out, flush=True)
        if dist:
            dist.destroy_process_group()
        raise SystemExit(0 if res.get("ok", True) else 1)

    shard_mode = os.environ.get("ESVO_SHARD_MODE", "tick")
    M = measure(shard_mode, args.strong)
    per_gpu, rig, stream, p, ticks, duration = M["per_gpu"], M["rig"], M["stream"], M["p"], M["ticks"], M["duration"]
    native, comm_note, runner, dt, st = M["native"], M["comm_note"], M["runner"], M["dt"], M["st"]
    n_events, n_points, n_matches, ksum, launches = M["n_events"], M["n_points"], M["n_matches"], M["ksum"], M["launches"]
    ev_rank, mt_rank = M["ev_rank"], M["mt_rank"]
    nd = p.bm_max_disparity - p.bm_min_disparity + 1

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
    workload_str = (f"{args.workload} synthetic stereo event stream, stationary, {len(stream.ev_left) / duration / 1e6:.1f} Mev/s/camera, "
                    f"{wl['note']}, throughput mode (all events of each 10 ms slice)")
    prof = committed_profile(args.workload) if world == 1 else None
    traffic, valu = profile_figures(prof, dom_name, float(kavg[dom]))
    total_ticks = K * per_gpu
    out = {
        "metric": "mapped events/sec (stereo TS raster + block matching + LM depth refinement + fusion), events resident in HBM "
                  "before the timed region" if not args.timed_ingest else
                  "mapped events/sec (stereo TS raster + block matching + LM depth refinement + fusion), host-to-device staging of every "
                  "tick's events inside the timed region",
        "value": n_events / dt,
        "unit": "events/s",
        "n_gpus": world,
        "steps": K,
        "warmup": Wm,
        "ms_per_step": dt / K * 1e3,
        "higher_is_better": True,
        "scaling": "weak" if per_gpu > 1 or world == 1 else "strong",
        "vs_baseline": None,
        "dtype": "f64",
        "data": "synthetic" + (" (events staged tick by tick inside the timed loop)" if args.timed_ingest else ""),
        "depth_points_per_s": n_points / dt,
        "config": {
            "workload": workload_str,
            "image": [rig.width, rig.height],
            "events_per_tick": n_events // max(total_ticks, 1),
            "matches_per_tick": n_matches // max(launches, 1),
            "depth_points_per_tick": n_points // max(total_ticks, 1),
            "ticks_timed": total_ticks,
            "disparity_range": [p.bm_min_disparity, p.bm_max_disparity],
            "parallelism": "1 GPU" if world == 1 else ((f"{world} GPUs, {K} ticks per GPU dealt round-robin, ncclAllGather of frames"
                                                        if shard_mode == "tick" else f"{world} GPUs, slots + image row bands")
                                                       + (" (esvo_comm_*: RCCL inside the C library)" if native else " (torch.distributed)")
                                                       + (f"; {comm_note}" if comm_note else "")),
        },
        "kernel_ms": {KERNEL_NAMES[i]: round(float(kavg[i]), 4) for i in range(7)},
        "roofline": {
            "bound": "hbm",
            "kernel": dom_name,
            "achieved": achieved,
            "peak": HBM_PEAK_GBS,
            "unit": "GB/s",
            "frac": achieved / HBM_PEAK_GBS,
            "traffic": traffic,
            "traffic_source": (prof["files"] if (prof and traffic is not None) else None),
            "algorithmic_bytes_per_launch": dom_bytes,
            "avg_launch_ms": float(kavg[dom]),
            # not an HBM- or MFMA-bound kernel: vector-ALU issue (f64) is its limit, so the figure
            # that says how close it runs to the chip is its VALU instruction rate against the issue peak
            "practical_bound": "f64 VALU issue",
            "valu": valu,
        },
    }

    whole = whole_tick_valu(prof, dt / K * 1e3 / per_gpu)
    if valu is not None and whole is not None:
        valu["whole_tick"] = whole
    out["roofline"]["traffic_note"] = TRAFFIC_NOTE
    # the same figures for the Time-Surface stage (the HBM-bound one: 24 B/event, 9 B/pixel) and both single-kernel slots
    n_scat = (int(st.events_scattered[0]) + int(st.events_scattered[1])) - (int(M["base"].events_scattered[0]) + int(M["base"].events_scattered[1]))
    out["roofline_kernels"] = roofline_rows(st, kavg, rig, nd, p, prof, ev_rank / launches, mt_rank / launches, n_scat / launches)
    if world > 1:
        out["ranks_seen"] = ranks_seen
        out["rccl"] = rccl
        out["launcher"] = "self (python bench.py --gpus N)" if os.environ.get("ESVO_BENCH_SELF_LAUNCHED") else "external (torch.distributed.run)"
        out["multi_gpu_note"] = ("`value` = tick-interleaved mode: rank r maps ticks k = r (mod N) whole, so the poses of N consecutive ticks "
                                 "must be known before the first of their maps exists -- true for esvo_MVStereo with given poses "
                                 "(BASELINE configs[1], [3]), NOT for the closed loop (configs[2]: tick k+1's poses come from tracking on "
                                 "tick k's map), where only `band_mode` (one tick split over the GPUs) applies")
    if args.check:
        mp_ = runner.get_map()  # collective at N > 1
        if rank == 0:
            out["check"] = {"final": {"map_size": int(len(mp_)), "sha1": map_sha1(mp_)}}
            if world == 1:
                out["check"]["oracle"] = check_against_oracle(rig, stream, p, ticks, Wm, local_rank)
    # the shader clock the dominant kernel really ran at inside the timed region (the handle's in-kernel probe: s_memtime
    # against s_memrealtime, include/esvo_hip.h ABI 3) -- the issue peak the VALU figures are priced against follows from it
    sclk, sclk_xcd = st.sclk_mhz(M["base"]) if world == 1 else (None, None)
    out["sclk_mhz_timed_region"] = sclk
    if sclk and valu is not None:
        attach_measured_clock(valu, sclk)
        if "whole_tick" in valu:
            attach_measured_clock(valu["whole_tick"], sclk)
    if rank == 0 and world == 1 and not args.timed_ingest and args.sustained_ticks > 0 and not args.no_extras:
        runner.close()
        runner = None
        try:
            out["sustained"] = sustained_point(args.workload, args.sustained_ticks, local_rank, prof, args.events_per_tick)
            out["sclk_mhz_sustained"] = out["sustained"].get("sclk_mhz")
            out["sustained"]["vs_headline"] = out["sustained"]["events_per_s"] / out["value"]
        except Exception as e:  # noqa: BLE001  (an extra: never takes the headline down with it)
            out["sustained"] = {"error": f"{type(e).__name__}: {e}"}
    if rank == 0 and world == 1 and not args.no_extras:
        try:
            out["other_operating_points"] = other_operating_points(local_rank)
        except Exception as e:  # noqa: BLE001  (extras: the headline line is printed whatever happens here)
            out["other_operating_points"] = {"error": f"{type(e).__name__}: {e}"}
    ref_maps = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        try:
            port = cpu_baseline(rig, stream, p, ticks)
        except Exception as e:  # noqa: BLE001  (the line is printed whatever happens to a baseline leg)
            port = {"error": f"{type(e).__name__}: {e}", "kind": "port"}
        try:
            out["cpu_baseline"], ref_maps = cpu_baseline_reference(args.workload, rig, stream, ticks)
            out["cpu_baseline"]["port"] = port     # the CPU oracle on every host thread, for the record
        except Exception as e:  # noqa: BLE001  (oracle/_ref is built in the build container and travels with the snapshot)
            port["reference_unavailable"] = f"{type(e).__name__}: {e}"
            out["cpu_baseline"] = port
    if rank == 0 and world == 1 and not args.no_parity:
        # parity evidence ON the line: (a) the DepthMap of the first timed tick of THIS workload against the CPU oracle in its
        # GPU-comparable arithmetic, element for element (what --check does); (b) the device against the REFERENCE's own node
        # object (oracle/_ref: esvo_Mapping.cpp compiled unmodified) on the reference-faithful ticks the CPU baseline just ran
        par = {}
        try:
            if "check" in out and "oracle" in out["check"]:
                chk = out["check"]["oracle"]
            else:
                chk = check_against_oracle(rig, stream, p, ticks, Wm, local_rank)
            par["oracle_equal"] = bool(chk["equal"])
            par["oracle"] = {"tick": chk["tick"], "map_size": chk["map_size"], "sha1": chk["sha1"], "oracle_sha1": chk["oracle_sha1"],
                             "what": "DepthMap after the first timed tick (row, col, inverse depth, variance, age of every element) "
                                     "replayed on a fresh handle vs the CPU oracle in canonical-reduction mode: SHA-1 of both"}
        except Exception as e:  # noqa: BLE001
            par["oracle_equal"] = None
            par["oracle_error"] = f"{type(e).__name__}: {e}"
        if ref_maps is not None:
            try:
                par["reference_node"] = parity_vs_reference_node(args.workload, rig, stream, ticks, ref_maps, local_rank)
            except Exception as e:  # noqa: BLE001
                par["reference_node"] = {"error": f"{type(e).__name__}: {e}"}
        else:
            par["reference_node"] = None
        out["parity"] = par
    if world > 1 and "ESVO_SHARD_MODE" not in os.environ and not args.strong and not args.check and not args.no_extras:
        # the OTHER way of using N GPUs, on the same line: every tick of ONE stream split over the ranks -- per-event work
        # by slot, per-cell work by image row band (north_star's image-tile partition), two ncclAllGather per tick (own-slot bytes; [count | kept points]) and the
        # all-gather of the DepthMap bands at read-out.  Strong scaling: K ticks in total, shorter ticks.
        if hasattr(runner, "dev"):
            runner.dev.close()
        B = measure("band", True)
        gm = B["runner"].get_map()   # collective: ncclAllGather of the bands
        if rank == 0:
            out["band_mode"] = {
                "value": B["n_events"] / B["dt"], "unit": "events/s", "ms_per_step": B["dt"] / K * 1e3, "scaling": "strong",
                "depth_points_per_s": B["n_points"] / B["dt"], "events_per_tick": B["n_events"] // max(K, 1),
                "map_size_after_gather": int(len(gm)),
                "parallelism": f"{world} GPUs: slots w % {world} for block matching + LM, {world} image row bands for fusion / clean / "
                               f"regularisation; 2 ncclAllGather per tick (one byte per own slot; [count | kept points] with the block sized by the largest kept count) + ncclAllGather of the map bands"
                               + (" (esvo_comm_*: RCCL inside the C library)" if B["native"] else " (torch.distributed)"),
            }
    if rank == 0:
        print(json.dumps(out), file=json_out, flush=True)
    if dist:
        dist.destroy_process_group()


def selftest(rank, world, local_rank, dist, ranks_seen, rccl, n_ticks=6):
    """`python bench.py --gpus N --selftest`: everything a failed scaling run would want to know, in well under 30 s.
    ranks_seen / rccl as on the benchmark line; one all-gather of 1 MiB per rank, verified and timed (10 repeats); then six
    ticks of the 346x260 workload through BOTH N-GPU modes (tick-interleaved: esvo_comm_tick; band: esvo_comm_shard_tick -- the
    native RCCL path unless ESVO_DIST_BACKEND / ESVO_NATIVE_COMM say otherwise) and, on rank 0, through one plain handle: the
    three DepthMap SHA-1 must be equal.  Prints ONE JSON line; the exit code is 0 only if every check passed."""
    import torch
    t_begin = time.perf_counter()
    backend = os.environ.get("ESVO_DIST_BACKEND", "nccl")
    res = {"selftest": True, "n_gpus": world, "backend": backend if dist else None, "ranks_seen": ranks_seen, "rccl": rccl, "ok": True}
    if dist:
        dev = "cuda" if backend == "nccl" else "cpu"
        n = 1 << 18   # 1 MiB of f32 per rank
        send = torch.full((n,), float(rank + 1), device=dev)
        recv = [torch.empty(n, device=dev) for _ in range(world)]
        times = []
        for i in range(13):
            if dev == "cuda":
                torch.cuda.synchronize()
            t0 = time.perf_counter()
            dist.all_gather(recv, send)
            if dev == "cuda":
                torch.cuda.synchronize()
            if i >= 3:
                times.append((time.perf_counter() - t0) * 1e6)
        good = all(bool((recv[r] == float(r + 1)).all().item()) for r in range(world))
        res["all_gather"] = {"bytes_per_rank": 4 * n, "us_min": min(times), "us_median": sorted(times)[len(times) // 2], "verified": good}
        res["ok"] = res["ok"] and good
    rig, stream, p, ticks = make_workload("upenn346x260", n_ticks, share=(rank, dist.barrier) if dist else None)

    def drive(runner):
        runner.ts_push_events(0, stream.ev_left)
        runner.ts_push_events(1, stream.ev_right)
        for t, stamps, poses, T in ticks:
            if hasattr(runner, "tick_resident"):
                runner.tick_resident(t, T, stamps, poses)
            else:
                runner.ts_render(0, t, download=False)
                runner.ts_render(1, t, download=False)
                runner.set_observation(t, None, None, T)
                runner.tick(t, stamps, poses)
        runner.synchronize()
        return runner.get_map()   # collective at N > 1

    shas = {}
    if dist:
        from esvo_amd import dist as edist
        native = backend == "nccl" and os.environ.get("ESVO_NATIVE_COMM", "1") != "0"
        res["exchange"] = "esvo_comm_* (RCCL inside libesvo_hip.so)" if native else f"torch.distributed ({backend})"
        # "band": events routed by image row, banded Time Surfaces; "band_broadcast": the A/B switch (every rank
        # stages everything, per-event work dealt by slot)
        for mode in ("tick", "band", "band_broadcast"):
            cls = ((edist.NativeTickSharded if mode == "tick" else edist.NativeBandSharded) if native
                   else (edist.TickShardedEsvo if mode == "tick" else edist.ShardedEsvo))
            kw = {} if mode == "tick" else {"routing": "y_rect" if mode == "band" else "broadcast"}
            t0 = time.perf_counter()
            try:
                runner = cls(p, rig, rank, world, local_rank, **kw)
                gm = drive(runner)
                shas[mode] = {"sha1": map_sha1(gm), "map_size": int(len(gm)), "seconds": round(time.perf_counter() - t0, 2)}
                if mode == "band":
                    st_ = runner.stats()
                    shas[mode]["rows"] = runner.dev.shard_rows()
                    shas[mode]["events_staged_rank0"] = [int(st_.events_staged[0]), int(st_.events_staged[1])]
                    shas[mode]["events_in_stream"] = [int(len(stream.ev_left)), int(len(stream.ev_right))]
                    shas[mode]["halo_violations"] = int(st_.halo_violations)
                runner.dev.close()
            except Exception as e:  # noqa: BLE001  (a hang inside RCCL cannot be caught: the 30 s budget is the caller's timeout)
                shas[mode] = {"error": f"{type(e).__name__}: {e}"}
                res["ok"] = False
    if rank == 0:
        single = lib.Esvo(p, rig, device=local_rank)
        gm = drive(single)
        single.close()
        shas["one_gpu"] = {"sha1": map_sha1(gm), "map_size": int(len(gm))}
        res["depth_map"] = shas
        same = all(v.get("sha1") == shas["one_gpu"]["sha1"] for v in shas.values())
        res["depth_map_equal_to_one_gpu"] = same
        res["ok"] = res["ok"] and same and shas["one_gpu"]["map_size"] > 100
    if dist:
        flag = torch.tensor([0.0 if res["ok"] else 1.0], device="cuda" if backend == "nccl" else "cpu")
        dist.all_reduce(flag, op=dist.ReduceOp.MAX)
        res["ok"] = flag.item() == 0.0
    res["seconds"] = round(time.perf_counter() - t_begin, 2)
    return res


def attach_measured_clock(v, sclk_mhz):
    """the same instruction rate priced against the issue peak at the clock measured inside this run"""
    peak = 256 * 4 * sclk_mhz * 1e6 / 4.0
    v["sclk_mhz_measured"] = sclk_mhz
    v["peak_at_measured_clock"] = peak
    v["frac_at_measured_clock"] = v["achieved"] / peak


def shift_events(ev, dt_ns):
    """the same events dt_ns later"""
    from esvo_amd.abi import event_ns
    ns = event_ns(ev) + np.uint64(dt_ns)
    out = ev.copy()
    out["sec"] = (ns // np.uint64(1_000_000_000)).astype(np.uint32)
    out["nsec"] = (ns % np.uint64(1_000_000_000)).astype(np.uint32)
    return out


def sustained_point(name, n_ticks, device, prof, events_cap=0, base_ticks=40, hook=None):
    """The headline workload for >= 2 s of wall time instead of 20 ticks (28 ms): long enough for the chip's power management
    to settle, with the shader clock measured inside the run.  Generating 16 s of synthetic stream would take minutes of
    numpy, so the stationary stream is LOOPED: 60 ms of history, then a segment of `base_ticks` ticks played again and again
    with its time stamps advanced by the segment's length; the trajectory continues (the rig keeps moving at its speed, each
    pass starts from the segment's first pose shifted along the direction of travel), so windows, propagation and fusion see
    a continuous motion.  At each seam the scene jumps back to the segment's first arrangement: for ~6 ticks (the 60 ms the
    Time Surfaces remember) the surfaces mix two arrangements -- `seam_ticks` says how many ticks that concerns.  Everything is
    staged in HBM before the timed region, as for `value`."""
    wl = WORKLOADS[name]
    rig, stream, p0, ticks0 = make_workload(name, base_ticks, events_cap)
    T_b = int(round(base_ticks * TICK_S * 1e9))
    t_seg0 = stream.t0_ns + int(round(HIST_S * 1e9))       # the segment covers [t_seg0, t_seg0 + T_b)
    hist = (stream.slice(0, stream.t0_ns, t_seg0), stream.slice(1, stream.t0_ns, t_seg0))
    seg = (stream.slice(0, t_seg0, t_seg0 + T_b), stream.slice(1, t_seg0, t_seg0 + T_b))
    n_warm = 8
    loops = (n_ticks + n_warm + base_ticks - 1) // base_ticks
    total = max(len(hist[0]) + loops * len(seg[0]), len(hist[1]) + loops * len(seg[1]))
    p, _ = params.make_params(params.PRESETS[wl["preset"]], rig, throughput_events=p0.process_event_num,
                              event_ring_capacity=int(total * 1.01) + 4096)
    dx = wl["speed"] * T_b * 1e-9                          # what the rig travels during one pass

    def pose(t_ns):
        k = max((int(t_ns) - t_seg0) // T_b, 0) if t_ns >= t_seg0 else 0
        T = stream.pose(int(t_ns) - k * T_b).copy()
        T[0, 3] += k * dx
        return T

    t_gen = time.perf_counter()
    dev = lib.Esvo(p, rig, device=device)
    for cam in (0, 1):
        dev.ts_push_events(cam, hist[cam])
        for k in range(loops):
            dev.ts_push_events(cam, shift_events(seg[cam], k * T_b) if k else seg[cam])
    ticks = []
    for k in range(n_ticks + n_warm):
        t = t_seg0 + (k + 1) * int(round(TICK_S * 1e9))
        stamps, poses = rostime.pose_table(pose, t, p.bm_half_slice_thickness)
        ticks.append((t, stamps, poses, pose(t)))
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
            hook(k - n_warm)
        if (k - n_warm) % 100 == 99:   # host time stamps without a synchronisation: the lazy tick paces the host to the device
            marks.append(time.perf_counter())
    dev.synchronize()
    dt = time.perf_counter() - t0
    s = dev.stats()
    dev.close()
    ev = int(s.total_events_in - b.total_events_in)
    sclk, per_xcd = s.sclk_mhz(b)
    ks = (np.array(list(s.sum_ms_kernel)) - np.array(list(b.sum_ms_kernel))) / n_ticks
    win = np.diff(np.array([t0] + marks)) / 100.0 * 1e3    # ms per tick over windows of 100 ticks
    res = {"windows_ms": [round(float(w), 4) for w in win], "events_per_s": ev / dt, "ms_per_tick": dt / n_ticks * 1e3, "ticks": n_ticks, "wall_s": dt,
           "events_per_tick": ev // n_ticks, "depth_points_per_s": int(s.total_points - b.total_points) / dt,
           "matches_per_tick": int(s.total_matches - b.total_matches) // n_ticks,
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


def _map_compare(gm, ref_map, W):
    """valid-set IoU over the believed cells and inverse-depth statistics on the intersection"""
    ka = gm["row"].astype(np.int64) * W + gm["col"]
    kb = ref_map["row"].astype(np.int64) * W + ref_map["col"]
    da = dict(zip(ka.tolist(), gm["inv_depth"].tolist()))
    db = dict(zip(kb.tolist(), ref_map["inv_depth"].tolist()))
    both = [k for k in da if k in db and da[k] > -1e-6 and db[k] > -1e-6]
    union = len(set(da) | set(db))
    d = np.array([da[k] - db[k] for k in both]) if both else np.zeros(0)
    return {"map_size": int(len(gm)), "reference_map_size": int(len(ref_map)),
            "iou": (len(set(da) & set(db)) / union) if union else 1.0,
            "rmse": float(np.sqrt(np.mean(d * d))) if len(d) else 0.0,
            "max_abs_diff": float(np.abs(d).max()) if len(d) else 0.0,
            "frac_within_1e-6": float((np.abs(d) <= 1e-6).mean()) if len(d) else 1.0,
            "frac_within_1e-4": float((np.abs(d) <= 1e-4).mean()) if len(d) else 1.0}


def _frame_compare(fr, ref_frame):
    out = {"frame_points": int(len(fr)), "reference_frame_points": int(len(ref_frame))}
    if len(fr) == len(ref_frame) and len(fr):
        rel = np.abs(fr["inv_depth"] - ref_frame["inv_depth"]) / np.maximum(np.abs(ref_frame["inv_depth"]), 1e-300)
        out["frame_same_points"] = bool(np.array_equal(fr["row"], ref_frame["row"]) and np.array_equal(fr["col"], ref_frame["col"]))
        out["frame_inv_depth_max_rel"] = float(rel.max())
        out["frame_inv_depth_median_rel"] = float(np.median(rel))
    return out


def parity_vs_reference_node(workload, rig, stream, ticks, ref_maps, device):
    """The device on exactly the ticks the reference's own code just mapped for `cpu_baseline`: raw events of both cameras in,
    esvo_map_tick_resident per tick, PROCESS_EVENT_NUM of the shipped yaml (10 000 on DSEC).  Compared after the last of those
    ticks -- the newest frame (same points; inverse depth to the LM tolerance: the reference's Eigen driver is third-party,
    DESIGN.md section 2) and the DepthMap (valid-set IoU, inverse-depth RMSE on the intersection; north_star: RMSE < 1e-4) --
    against TWO runs of the reference's sources (oracle/_ref, compiled unmodified):
      reference_classes  EventBM / DepthProblemSolver / DepthFusion / DepthRegularization driven in MappingAtTime's order on the
                         events the node selected (oracle/ref_harness.cpp), with the one behaviour the reference leaves undefined
                         DEFINED: a grid cell whose list element SmartGrid::clean erased reads empty
      reference_node     the esvo_Mapping node object itself, as is: its regulariser reads erased list elements through dangling
                         grid pointers (freed memory); with RegularizationRadius 20 every such cell is a stale neighbour of up to
                         41 x 41 cells, so inverse depths differ wherever the heap still holds the erased values -- reported, not
                         a parity target (no implementation can reproduce freed memory)."""
    pf, n_used, node_map, node_frame, cls_map, cls_frame, dangling = ref_maps
    dev = lib.Esvo(pf, rig, device=device)
    dev.ts_push_events(0, stream.ev_left)
    dev.ts_push_events(1, stream.ev_right)
    run_single(dev, stream, ticks, 0, n_used)
    gm, fr = dev.get_map(), dev.get_last_frame()
    dev.close()
    out = {"events": int(pf.process_event_num), "ticks": int(n_used)}
    out.update(_map_compare(gm, cls_map, rig.width))
    out.update(_frame_compare(fr, cls_frame))
    out["what"] = ("device (esvo_map_tick_resident, raw events in) vs the reference's mapper classes compiled from source and driven in "
                   "MappingAtTime's order on the same events (erased grid cells read empty)")
    node = {"erased_cells_still_referenced": int(dangling)}
    node.update(_map_compare(gm, node_map, rig.width))
    node.update(_frame_compare(fr, node_frame))
    node["what"] = ("the same device map vs the esvo_Mapping node object as is: its regulariser (radius 20) reads erased list elements "
                    "through dangling grid pointers -- undefined behaviour upstream (SURVEY Appendix A-7), reported for completeness")
    out["node_object_as_is"] = node
    return out


def check_against_oracle(rig, stream, p, ticks, n_first, device):
    """Replays ticks 0 .. n_first (the first timed tick) on a fresh handle and on the CPU oracle (GPU-comparable mode) and
    compares the two DepthMaps of that tick."""
    from oracle import oracle
    dev = lib.Esvo(p, rig, device=device)
    dev.ts_push_events(0, stream.ev_left)
    dev.ts_push_events(1, stream.ev_right)
    run_single(dev, stream, ticks, 0, n_first + 1)
    gm = dev.get_map()
    dev.close()
    m = oracle.OracleMapper(p, rig)
    m.set_mode(True, True)
    m.set_threads(os.cpu_count() or 1)
    ts = [oracle.OracleTS(rig.width, rig.height), oracle.OracleTS(rig.width, rig.height)]
    done = [0, 0]
    for t, stamps, poses, T in ticks[:n_first + 1]:
        for cam, (ev, ns) in enumerate(((stream.ev_left, stream.ns_left), (stream.ev_right, stream.ns_right))):
            hi = int(np.searchsorted(ns, t, side="left"))
            ts[cam].push(ev[done[cam]:hi])
            done[cam] = hi
        l = ts[0].render(t, map_x=rig.left.map_x, map_y=rig.left.map_y)
        r = ts[1].render(t, map_x=rig.right.map_x, map_y=rig.right.map_y)
        m.set_observation(t, l, r, T)
        m.set_poses(stamps, poses)
        idx = oracle.select_events(stream.ev_left, t, p.bm_half_slice_thickness, p.process_event_num)
        m.tick(stream.ev_left[idx])
    om = m.get_map()
    a, b = map_sha1(gm), map_sha1(om)
    return {"tick": n_first, "map_size": int(len(gm)), "sha1": a, "oracle_map_size": int(len(om)), "oracle_sha1": b, "equal": a == b}


def other_operating_points(device):
    """Secondary figures of the same JSON line: the 346x260 stream of north_star in throughput mode and the two
    reference-faithful tick sizes (the mode the ROS node runs: PROCESS_EVENT_NUM 10000 on DSEC, 1000 on upenn), each as
    the time of one tick completed on its own (latency: nothing is in flight beside it) and as the sustained time per tick
    with two ticks in flight."""
    out = {}

    def throughput(name, n, check=False, timed_ingest=False, pinned=False):
        rig, stream, p, ticks = make_workload(name, n + 3)
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
        ks = np.array(list(s.sum_ms_kernel)) - np.array(list(b.sum_ms_kernel))
        ka = ks / n
        if ks[7] > 0:
            ka[0], ka[1] = 2 * ks[0] / ks[7], 2 * ks[1] / ks[7]
        scat = (int(s.events_scattered[0]) + int(s.events_scattered[1])) - (int(b.events_scattered[0]) + int(b.events_scattered[1]))
        res["kernel_ms"] = {KERNEL_NAMES[i]: round(float(ka[i]), 4) for i in range(7)}
        res["roofline_kernels"] = roofline_rows(s, ka, rig, p.bm_max_disparity - p.bm_min_disparity + 1, p, committed_profile(name),
                                                ev / n, int(s.total_matches - b.total_matches) / n, scat / n)
        if check:   # the first timed tick replayed on a fresh handle against the CPU oracle (as bench.py --check does)
            if (os.cpu_count() or 1) >= 64:
                res["check_oracle_equal"] = bool(check_against_oracle(rig, stream, p, ticks, 3, device)["equal"])
            else:   # four 5e5-event ticks of the oracle take minutes on a small host: the default run must stay short
                res["check_oracle_equal"] = None
                res["check_note"] = "skipped on a host with fewer than 64 threads (bench.py --workload hd1280x720 --check runs it)"
        return res

    def latency(name, n_events, n):
        rig, stream, p, ticks = make_workload(name, n + 6, events_cap=n_events)
        dev = lib.Esvo(p, rig, device=device)
        dev.ts_push_events(0, stream.ev_left)
        dev.ts_push_events(1, stream.ev_right)
        run_single(dev, stream, ticks, 0, 6, sync_each=True)
        t0 = time.perf_counter()
        run_single(dev, stream, ticks, 6, n + 6, sync_each=True)
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
                        "esvo_track_register (host C++ over esvo_track_normal_equations: one launch and 224 B back per iteration, "
                        "12 Gauss-Newton iterations at most)"}

    def point(key, fn, *a, **kw):   # an extra never takes the headline down with it: its failure is reported in its place
        try:
            out[key] = fn(*a, **kw)
        except Exception as e:  # noqa: BLE001
            out[key] = {"error": f"{type(e).__name__}: {e}"}

    point("upenn346x260_throughput", throughput, "upenn346x260", 20)
    point("dsec640x480_reference_faithful_10000", latency, "dsec640x480", 10000, 20)
    point("upenn346x260_reference_faithful_1000", latency, "upenn346x260", 1000, 20)
    point("upenn346x260_closed_loop", closed_loop)
    # SURVEY.md section 8 stress row: 1280x720, 145 disparity candidates, 100 Mev/s over both cameras, with the oracle equality flag
    point("hd1280x720_throughput", throughput, "hd1280x720", 6, check=True)
    # the headline workload with the PCIe transfer of every tick's events inside the timed loop (never `value`)
    point("dsec640x480_with_timed_ingest", throughput, "dsec640x480", 20, timed_ingest=True)
    point("dsec640x480_with_timed_ingest_pinned", throughput, "dsec640x480", 20, timed_ingest=True, pinned=True)
    return out


def cpu_baseline(rig, stream, p, ticks):
    """The CPU oracle ("port" of the reference mapper) on the SAME stages as `value` (both Time-Surface renders + the
    mapper tick), in steady state: the fusion window (maxNumFusionFrames) is filled first, then the median of 5 ticks is
    taken.  Block matching + LM run on all host threads, fusion / regularisation single-threaded as in the reference."""
    from oracle import oracle
    try:
        oracle.build(fast=True, force=True)  # -O3 -march=native for THIS host
        fast = True
    except Exception:
        fast = False
    cores = os.cpu_count() or 1
    n_meas = 5
    n_fill = int(p.max_fusion_frames) if p.fusion_strategy == 0 else 5   # CONST_FRAMES: the window; CONST_POINTS: a few ticks
    n_fill = max(min(n_fill, len(ticks) - n_meas), 0)
    use = ticks[: n_fill + n_meas]
    cap = None if cores >= 32 else 60000  # a small host maps a bounded sample of every tick's events
    ts = [oracle.OracleTS(rig.width, rig.height, fast=fast), oracle.OracleTS(rig.width, rig.height, fast=fast)]
    m = oracle.OracleMapper(p, rig, fast=fast)
    m.set_threads(cores)
    done = [0, 0]
    per_tick, n_ev = [], []
    for k, (t, stamps, poses, T) in enumerate(use):
        t0 = time.perf_counter()
        for cam, (ev, ns) in enumerate(((stream.ev_left, stream.ns_left), (stream.ev_right, stream.ns_right))):
            hi = int(np.searchsorted(ns, t, side="left"))
            ts[cam].push(ev[done[cam]:hi])  # EventQueueMat::insertEvent of the tick's new events (TS ingest)
            done[cam] = hi
        l = ts[0].render(t, map_x=rig.left.map_x, map_y=rig.left.map_y)
        r = ts[1].render(t, map_x=rig.right.map_x, map_y=rig.right.map_y)
        t_ts = time.perf_counter() - t0
        m.set_observation(t, l, r, T)
        m.set_poses(stamps, poses)
        idx = oracle.select_events(stream.ev_left, t, p.bm_half_slice_thickness, p.process_event_num, fast=fast)
        sample = stream.ev_left[idx]
        if cap and len(sample) > cap:
            sample = sample[:cap]
        t0 = time.perf_counter()
        m.tick(sample)
        t_map = time.perf_counter() - t0
        if k >= n_fill:
            per_tick.append(t_ts + t_map)
            n_ev.append(len(sample))
    rates = sorted(n / s for n, s in zip(n_ev, per_tick))
    # the reference's own threading: NUM_THREAD_MAPPING = 4 (esvo_core/include/esvo_core/tools/utils.h:36), mapper only, one more tick
    m.set_threads(4)
    small = sample[: max(len(sample) // 8, 1)]
    t0 = time.perf_counter()
    pts = m.refine(m.match(small), cull=True)
    t_map4 = time.perf_counter() - t0
    return {
        "value": rates[len(rates) // 2],
        "unit": "events/s",
        "cores": cores,
        "kind": "port",
        "sample": f"median of {n_meas} steady-state ticks (fusion window of {n_fill} frames filled first), {int(np.mean(n_ev))} events "
                  f"block-matched per tick, same stages as `value` (TS ingest + both TS renders + mapper tick); BM + LM on {cores} "
                  f"threads, TS / fusion / regularisation single-threaded as in the reference; {np.median(per_tick):.2f} s per tick, "
                  f"min/max rate {rates[0]:.0f}/{rates[-1]:.0f} events/s",
        "reference_threading": {"value": len(small) / t_map4, "unit": "events/s", "cores": 4,
                                "sample": f"block matching + LM only ({len(small)} events, {len(pts)} points) on the reference's "
                                          f"NUM_THREAD_MAPPING = 4 threads"},
    }


def cpu_baseline_reference(workload, rig, stream, ticks):
    """The REFERENCE's own CPU path timed beside the GPU number (`kind: "reference"`): oracle/_ref = ESVO's sources compiled
    unmodified in the build container (oracle/Makefile; they cannot travel, so the library is -O2 for generic x86-64) --
    the Time-Surface node class (TimeSurface.cpp: eventsCallback + createTimeSurfaceAtTime, one thread per camera as in the
    ROS graph's two node processes) and the mapper NODE object (esvo_Mapping.cpp: dataTransferring + MappingAtTime with its
    own NUM_THREAD_MAPPING = 4 std::threads, esvo_core/include/esvo_core/tools/utils.h:36), driven through their own
    callbacks.  OpenCV is absent from the image: its three calls on the path (convertTo + medianBlur + remap of the raster,
    GaussianBlur of the observation) are done by the CPU oracle's restatement and timed with the stage they belong to.
    Steady state: the fusion window is filled first.  Two sizes: the tick the reference really runs (PROCESS_EVENT_NUM of
    the yaml) and a capped throughput tick (every event of the slice up to a bound, so that the default run stays short)."""
    from oracle import oracle, ref
    wl = WORKLOADS[workload]
    W, H = rig.width, rig.height

    def run(process_event_num, n_fill, n_meas):
        over = {} if process_event_num is None else dict(process_event_num=process_event_num)
        pf, _ = params.make_params(params.PRESETS[wl["preset"]], rig, **over)
        node = ref.RefNode(pf, rig, stream.pose)
        classes = ref.RefMapper(pf, rig) if process_event_num is None else None   # for bench.py's parity block (not timed)
        # (the node keeps the newest MAX_EVENT_QUEUE_LENGTH = 3 000 000 left events, esvo_Mapping.cpp:706-713: the stream is fed
        #  tick by tick as the events topic would, one 1 ms message ahead of the tick time)
        fed = 0
        ts = [ref.RefTS(W, H, pf.decay_ms, bool(pf.ignore_polarity)), ref.RefTS(W, H, pf.decay_ms, bool(pf.ignore_polarity))]
        done = [0, 0]
        rows = []
        for k, (t, stamps, poses, T) in enumerate(ticks[: n_fill + n_meas]):
            t_ts = []
            imgs = []
            for cam, (ev, ns, c) in enumerate(((stream.ev_left, stream.ns_left, rig.left), (stream.ev_right, stream.ns_right, rig.right))):
                hi = int(np.searchsorted(ns, t, side="left"))
                t0 = time.perf_counter()
                ts[cam].push(ev[done[cam]:hi])                       # TimeSurface::eventsCallback
                f64 = ts[cam].render(t)                              # createTimeSurfaceAtTime up to convertTo
                u8 = np.rint(f64).astype(np.uint8)                   # cv::Mat::convertTo(CV_8U): round half to even, values in [0, 255]
                if pf.median_blur_kernel_size:
                    u8 = oracle.median3(u8)                          # cv::medianBlur
                u8 = oracle.remap_bilinear(u8, c.map_x, c.map_y)     # cv::remap
                t_ts.append(time.perf_counter() - t0)
                done[cam] = hi
                imgs.append(u8)
            t0 = time.perf_counter()
            hi = int(np.searchsorted(stream.ns_left, t + 1_000_000, side="left"))
            node.push_events(stream.ev_left[fed:hi])                 # esvo_Mapping::eventsCallback (left camera)
            fed = hi
            obs = [oracle.gaussian5(i) for i in imgs] if pf.smooth_time_surface else imgs   # GaussianBlurTS(5), EventBM.cpp:68-72
            node.push_observation(t, obs[0], obs[1])                 # timeSurfaceCallback
            ok = node.data_transferring()                            # dataTransferring (event selection, 201 tf lookups)
            if ok:
                node.mapping_at_time()                               # MappingAtTime: BM + LM on 4 threads, fusion, clean, regularisation
            t_map = time.perf_counter() - t0
            if k >= n_fill and ok:
                rows.append((len(node.selected_events()), max(t_ts) + t_map, max(t_ts), t_map, len(node.newest_frame())))
            if classes is not None and ok:   # the reference's classes on what the node just handed to its matcher
                st_n, T_n = node.pose_table()
                classes.set_observation(t, obs[0], obs[1], T)
                classes.set_poses(st_n, T_n)
                classes.tick(stream.ev_left[node.matched_events()])
        if classes is None:
            return pf, rows, None
        return pf, rows, (pf, min(n_fill + n_meas, len(ticks)), node.get_map(), node.newest_frame(), classes.get_map(),
                          classes.get_last_frame(), classes.counters()["dangling_cells"])

    p0 = params.make_params(params.PRESETS[wl["preset"]], rig)[0]
    n_fill = int(p0.max_fusion_frames) if p0.fusion_strategy == 0 else 5   # CONST_FRAMES: the window; CONST_POINTS: a few ticks
    pf, rows, ref_maps = run(None, max(min(n_fill, len(ticks) - 3), 0), 3)
    if not rows:
        raise RuntimeError("the reference node mapped no tick (dataTransferring refused every observation)")
    rates = sorted(n / s for n, s, _, _, _ in rows)
    med = rows[len(rows) // 2]
    out = {
        "value": rates[len(rates) // 2], "unit": "events/s", "cores": 4, "kind": "reference",
        "sample": f"ESVO's own TimeSurface + esvo_Mapping node objects (oracle/_ref, -O2, stand-in Eigen / ROS headers): median of "
                  f"{len(rows)} steady-state ticks of the reference's own size (PROCESS_EVENT_NUM = {pf.process_event_num}: "
                  f"{med[0]} events selected, {med[4]} depth points), same stages as `value`; mapper on NUM_THREAD_MAPPING = 4 "
                  f"threads, one Time-Surface thread per camera (the slower camera counts); {med[1]:.2f} s per tick "
                  f"(Time Surface {med[2]:.3f} s, mapper {med[3]:.2f} s)",
    }
    cap = 30000
    _, rows2, _ = run(cap, 2, 1)
    if rows2:
        n, sec, tts, tmap, pts = rows2[0]
        out["throughput_tick_capped"] = {"value": n / sec, "unit": "events/s", "cores": 4,
                                         "sample": f"one tick with PROCESS_EVENT_NUM = {cap} ({n} events selected, {pts} depth points) "
                                                   f"after 2 ticks of window fill: {sec:.2f} s (Time Surface {tts:.3f} s, mapper {tmap:.2f} s)"}
    return out, ref_maps


if __name__ == "__main__":
    main()
