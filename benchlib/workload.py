"""The bench workloads: seeded synthetic stereo event streams of SURVEY.md section 8(d) and the ticks over them."""
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


def shift_events(ev, dt_ns):
    """the same events dt_ns later"""
    from esvo_amd.abi import event_ns
    ns = event_ns(ev) + np.uint64(dt_ns)
    out = ev.copy()
    out["sec"] = (ns // np.uint64(1_000_000_000)).astype(np.uint32)
    out["nsec"] = (ns % np.uint64(1_000_000_000)).astype(np.uint32)
    return out


