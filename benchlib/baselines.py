"""CPU legs of the line (the reference's own code, the oracle port) and the parity block.  The ONLY place bench.py touches oracle/."""
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
from .workload import WORKLOADS, map_sha1, run_single  # noqa: E402


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
    unmodified in the build container (oracle/Makefile; they cannot travel, so the library is built there, for generic x86-64,
    at the -O3 the reference's own CMakeLists set) --
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
    # the objects at the reference's own optimisation level (-O3: esvo_core/CMakeLists.txt:7; bit-identical to the -O2 pin,
    # tests/test_ref_pin.py); a snapshot that only carries the -O2 pin is timed with that, and says so
    have_o3 = all(os.path.exists(os.path.join(ROOT, "oracle", "_ref", f)) for f in ("libesvo_ref_node_O3.so", "libesvo_ref_ts_O3.so"))
    opt = "-O3 as the reference's CMakeLists build it" if have_o3 else "-O2 (the -O3 objects are missing from this snapshot)"

    def run(process_event_num, n_fill, n_meas):
        over = {} if process_event_num is None else dict(process_event_num=process_event_num)
        pf, _ = params.make_params(params.PRESETS[wl["preset"]], rig, **over)
        node = ref.RefNode(pf, rig, stream.pose, o3=have_o3)
        classes = ref.RefMapper(pf, rig) if process_event_num is None else None   # for bench.py's parity block (not timed)
        # (the node keeps the newest MAX_EVENT_QUEUE_LENGTH = 3 000 000 left events, esvo_Mapping.cpp:706-713: the stream is fed
        #  tick by tick as the events topic would, one 1 ms message ahead of the tick time)
        fed = 0
        ts = [ref.RefTS(W, H, pf.decay_ms, bool(pf.ignore_polarity), o3=have_o3), ref.RefTS(W, H, pf.decay_ms, bool(pf.ignore_polarity), o3=have_o3)]
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
        "build": opt,
        "sample": f"ESVO's own TimeSurface + esvo_Mapping node objects (oracle/_ref, {opt}, generic x86-64, stand-in Eigen / ROS headers): median of "
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


