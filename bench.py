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
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

from esvo_amd import lib  # noqa: E402
# (re-exported: tools/ and tests/ address these through `bench.`)
from benchlib.baselines import check_against_oracle, cpu_baseline, cpu_baseline_reference, parity_vs_reference_node  # noqa: E402,F401
from benchlib.multigpu import band_share, selftest, self_launch, tick_share  # noqa: E402,F401
from benchlib.points import extra_operating_points, reference_faithful_points, sustained_point  # noqa: E402,F401
from benchlib.roofline import (TRAFFIC_NOTE, algorithmic_bytes, attach_measured_clock, committed_profile, profile_figures,  # noqa: E402,F401
                               roofline_rows, whole_tick_valu)
from benchlib.workload import (HBM_PEAK_GBS, HIST_S, KERNEL_NAMES, KERNEL_SYMBOLS, TICK_S, VALU_PEAK_INST_S, WORKLOADS, make_workload,  # noqa: E402,F401
                               map_sha1, run_single, shift_events)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", default="dsec640x480", choices=sorted(WORKLOADS))
    ap.add_argument("--events-per-tick", type=int, default=0, help="cap on block-matched events per tick (0 = all)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true",
                    help="the headline line only: no sustained point, no reference-faithful ticks, no band-share projection")
    ap.add_argument("--extras", action="store_true",
                    help="also the heavy secondary points (346x260 throughput, closed loop, 1280x720 stress stream, PCIe-inclusive "
                         "ingest): minutes; the default run keeps to what the driver's time budget allows")
    ap.add_argument("--no-band-share", action="store_true", help="skip the band-mode projection block (8 and 16 logical shards on this GPU)")
    ap.add_argument("--no-tick-share", action="store_true",
                    help="skip the tick-interleaved projection block (rank 0 of a world-8 run on this GPU, the other ranks' frames pre-recorded)")
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
            # "band": every tick split over the GPUs by image row band (latency of one tick).
            # The exchange runs inside libesvo_hip.so (esvo_comm_*: RCCL called from C); the torch.distributed drivers remain
            # for other backends (gloo on a shared GPU: tests), with ESVO_NATIVE_COMM=0, and as the fallback below.
            if use_native:
                cls = edist.NativeTickSharded if shard_mode == "tick" else edist.NativeBandSharded
            elif shard_mode == "tick" and os.environ.get("ESVO_DIST_BACKEND", "nccl") == "nccl":
                cls = edist.CallbackTickSharded   # the same C round logic (two rounds in flight), torch.distributed's all-gather as its transport
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
                try:
                    for k in range(n_warm):
                        step(k)
                    runner.synchronize()
                except Exception as e:  # noqa: BLE001
                    from esvo_amd import dist as edist
                    if not isinstance(e, edist.HaloViolation):
                        raise
                    # routed band mode refused a tick (a refinement left the rendered rows; every rank at the same tick):
                    # the exact-whatever-the-motion routing instead, stated on the line
                    comm_note = "routed band mode raised ESVO_ERR_HALO in warm-up; re-run with ESVO_ROUTE_BROADCAST"
                    runner.restart(routing="broadcast")
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
        # (ABI 8: how many of them recorded their stage-timing events -- all of them while ticks overlap; what sum_ms_kernel[2..6] sums)
        timed = int(st.stage_timing_samples - base.stage_timing_samples)
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
                    launches=launches, timed=timed if timed > 0 else launches, ev_rank=ev_rank, mt_rank=mt_rank, shard_mode=shard_mode)

    if args.selftest:
        res = selftest(rank, world, local_rank, dist, ranks_seen, rccl)
        if rank == 0:
            print(json.dumps(res), file=json_out, flush=True)
        if dist:
            dist.destroy_process_group()
        raise SystemExit(0 if res.get("ok", True) else 1)

    shard_mode = os.environ.get("ESVO_SHARD_MODE", "tick")
    legs = {}                      # wall seconds of every leg of this invocation (`wall_s` on the line)
    t_leg = [time.perf_counter()]

    def lap(name):
        now = time.perf_counter()
        legs[name] = round(legs.get(name, 0.0) + now - t_leg[0], 2)
        t_leg[0] = now

    M = measure(shard_mode, args.strong)
    lap("stream_generation_warmup_timed_region")
    per_gpu, rig, stream, p, ticks, duration = M["per_gpu"], M["rig"], M["stream"], M["p"], M["ticks"], M["duration"]
    native, comm_note, runner, dt, st = M["native"], M["comm_note"], M["runner"], M["dt"], M["st"]
    n_events, n_points, n_matches, ksum, launches = M["n_events"], M["n_points"], M["n_matches"], M["ksum"], M["launches"]
    ev_rank, mt_rank = M["ev_rank"], M["mt_rank"]
    nd = p.bm_max_disparity - p.bm_min_disparity + 1

    kavg = ksum / M["timed"]
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
                  "before the timed region; exactly K ticks between two synchronisations, i.e. including the fill and the drain of the "
                  "two-deep tick pipeline (`sustained`: the same workload over 1600 ticks)" if not args.timed_ingest else
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
        lap("sustained")
    if rank == 0 and world == 1 and not args.no_extras:
        try:
            out["other_operating_points"] = reference_faithful_points(local_rank)
            if args.extras:
                out["other_operating_points"].update(extra_operating_points(local_rank))
        except Exception as e:  # noqa: BLE001  (extras: the headline line is printed whatever happens here)
            out["other_operating_points"] = {"error": f"{type(e).__name__}: {e}"}
        lap("other_operating_points")
    if rank == 0 and world == 1 and not args.no_extras and not args.no_band_share and not args.timed_ingest:
        # SURVEY 8(e) on one GPU: what one rank of an 8-GPU band-mode run computes per tick (a projection, labelled as such)
        try:
            if runner is not None:
                runner.close()
                runner = None
            out["band_share"] = band_share(args.workload, local_rank, events_cap=args.events_per_tick)
        except Exception as e:  # noqa: BLE001
            out["band_share"] = {"error": f"{type(e).__name__}: {e}"}
        lap("band_share")
    if rank == 0 and world == 1 and not args.no_extras and not args.no_tick_share and not args.timed_ingest:
        # the mode whose number lands on the N-GPU line (`value` at --gpus N): what one rank of a world-8 run does per round
        try:
            if runner is not None:
                runner.close()
                runner = None
            sus = out.get("sustained") or {}
            out["tick_share"] = tick_share(args.workload, local_rank, events_cap=args.events_per_tick,
                                           one_gpu_ms_per_tick=sus.get("ms_per_tick") or out["ms_per_step"])
        except Exception as e:  # noqa: BLE001
            out["tick_share"] = {"error": f"{type(e).__name__}: {e}"}
        lap("tick_share")
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
        # both ratios ON the block (a GPU / CPU ratio says nothing about kernel quality -- the roofline does -- but if one is
        # quoted it must be the defensible one: against the allocation-free port on EVERY host thread, not against four threads)
        cb = out["cpu_baseline"]
        ratios = {}
        if cb.get("kind") == "reference" and cb.get("value"):
            ratios["gpu_over_reference_4_threads"] = out["value"] / cb["value"]
        pv = cb.get("port", cb if cb.get("kind") == "port" else {})
        if pv.get("value"):
            ratios[f"gpu_over_port_{pv.get('cores')}_threads"] = out["value"] / pv["value"]
        cb["ratios"] = ratios
        lap("cpu_baseline")
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
        lap("parity")
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
                "routing": getattr(B["runner"], "routing", None),
                "parallelism": f"{world} GPUs, {world} image row bands: a rank ingests the events of its rows, renders the Time Surfaces of its "
                               f"band + halo, block-matches and refines the events whose floor(y_rect) is in the band, fuses / cleans / "
                               f"regularises its rows; 2 ncclAllGather per tick (two bits per slot; [count | kept points] with the block sized by "
                               f"the largest kept count) + ncclAllGather of the map bands"
                               + (" (esvo_comm_*: RCCL inside the C library)" if B["native"] else " (torch.distributed)"),
            }
    if rank == 0:
        out["wall_s"] = legs
        print(json.dumps(out), file=json_out, flush=True)
    if dist:
        dist.destroy_process_group()



if __name__ == "__main__":
    main()
