"""Randomised CALL SEQUENCES against the canonical oracle: one handle lives through 20-40 ticks during which every tick takes a
random path -- the four synchronous calls, the lazy esvo_map_tick_resident (two ticks in flight), or an image pair handed in from
the host -- events arrive through random ingest calls in packets of random size, and between ticks the caller does what callers
do: reads the map, the committed map, the last frame, the point cloud, the statistics, synchronises, CHANGES PARAMETERS
(esvo_set_params: thresholds, regulariser on / off and radius, fusion radius, LM iteration cap, ZNCC threshold, smoothing,
denoising, decay) and RESETS the handle and starts again from the stream's current time.  Every map, frame and cloud that is read must
equal what the oracle holds at that point, bit for bit.
usage: python tools/fuzz_api.py [cases] [first seed]      (GPU; exits 1 on any difference)"""
import os
import sys
import time

ROOT = os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import numpy as np  # noqa: E402

import fuzz_parity  # noqa: E402
from benchlib.workload import map_sha1  # noqa: E402
from esvo_amd import calib, lib, params, rostime, synth  # noqa: E402
from esvo_amd.abi import EVENT_DTYPE, serialize_event_array  # noqa: E402
from oracle import oracle  # noqa: E402


def run_case(seed):
    rig_name, cfg, over, sc = fuzz_parity.draw(seed)
    rng = np.random.default_rng(seed + 12345)
    pick = lambda xs: xs[int(rng.integers(len(xs)))]  # noqa: E731
    # (patch size, window capacities, queue length and tick size stay: esvo_set_params keeps the capacities of esvo_create)
    cfg["PROCESS_EVENT_NUM"] = pick([300, 1000, 3000])
    # the sequence is the subject here: the options that mostly produce empty maps stay at the preset's values
    preset = params.PRESETS[fuzz_parity.RIGS[rig_name][0]]
    for key in ("BM_ZNCC_Threshold", "BM_min_disparity", "BM_max_disparity", "Tdist_nu", "Tdist_scale", "stdVar_vis_threshold",
                "RegularizationMinNeighbours", "RegularizationMinCloseNeighbours"):
        cfg[key] = preset.get(key, params.CODE_DEFAULTS[key])
    cfg["BM_bUpDownConfiguration"] = False
    rig = calib.dataset_rig(rig_name)
    p, _ = params.make_params(cfg, rig, **over)
    n_ticks = int(rng.integers(20, 41))
    tick_s = sc["tick_ms"] * 1e-3
    st = synth.make_stream(rig, sc["points"], 0.06 + (n_ticks + 1) * tick_s, sc["rho"][0], sc["rho"][1], seed=sc["seed"], speed=sc["speed"])
    dev = lib.Esvo(p, rig)
    ql = p.max_event_queue_len

    def fresh_oracle(pp):
        m = oracle.OracleMapper(pp, rig)
        m.set_mode(True, True)
        m.set_threads(os.cpu_count() or 1)
        return m, [oracle.OracleTS(rig.width, rig.height, queue_len=ql or 20) for _ in range(2)]
    m, ots = fresh_oracle(p)
    done = [0, 0]
    first = [0, 0]            # index of the first event the handle has seen since its last reset
    expect = {}               # tick stamp -> (size, sha) of the oracle's map behind that tick
    bad, log = [], []
    t_last = None
    keep_alive = []

    def push(cam, ev_all, hi):
        a = done[cam]
        while a < hi:
            b = min(hi, a + int(rng.integers(200, 6000)))
            pk = ev_all[a:b]
            how = pick(["plain", "plain", "wire", "async"])
            if how == "wire":
                dev.ts_push_event_array(cam, serialize_event_array(pk, rig.width, rig.height))
            elif how == "async":
                buf = np.ascontiguousarray(pk.copy(), dtype=EVENT_DTYPE)
                keep_alive.append(buf)
                dev.ts_push_events_async(cam, buf)
            else:
                dev.ts_push_events(cam, pk)
            a = b
        ots[cam].push(ev_all[done[cam]:hi])
        done[cam] = hi

    for k in range(n_ticks):
        t = st.t0_ns + int((0.06 + (k + 1) * tick_s) * 1e9)
        # ---- between ticks
        r = rng.integers(12)
        if r == 0 and k > 2:                      # reset: the handle forgets everything; the caller starts again 60 ms before the next tick
            dev.reset()
            m, ots = fresh_oracle(p)
            t_from = t - 60_000_000
            for cam, ns in enumerate((st.ns_left, st.ns_right)):
                done[cam] = first[cam] = int(np.searchsorted(ns, t_from, side="left"))
            expect.clear()
            t_last = None
            log.append(f"reset@{k}")
        elif r == 1:                              # parameters change between two ticks
            cfg2 = dict(cfg)
            cfg2.update(residual_vis_threshold=float(rng.uniform(5, 40)), age_vis_threshold=pick([0, 1, 2]), age_max_range=pick([1, 3, 10]),
                        fusion_radius=pick([0, 1, 2]), Regularization=bool(rng.integers(2)), RegularizationRadius=pick([1, 3, 5, 12, 20]),
                        ITERATION_OPTIMIZATION=pick([1, 3, 10, 20]), BM_ZNCC_Threshold=pick([0.05, 0.1, 0.3]),
                        SmoothTimeSurface=bool(rng.integers(2)), Denoising=bool(rng.integers(3) == 0), decay_ms=pick([10.0, 30.0, 100.0]),
                        stdVar_vis_threshold=float(cfg["stdVar_vis_threshold"] * rng.uniform(0.5, 2.0)))
            p2, _ = params.make_params(cfg2, rig, **over)
            dev.set_params(p2)
            m.set_params(p2)
            cfg, p = cfg2, p2
            log.append(f"params@{k}")
        # ---- the tick
        stamps, poses = rostime.pose_table(st.pose, t, p.bm_half_slice_thickness)
        T = st.pose(t)
        for cam, (ev, ns) in enumerate(((st.ev_left, st.ns_left), (st.ev_right, st.ns_right))):
            push(cam, ev, int(np.searchsorted(ns, t, side="left")))
        kw = dict(decay_ms=p.decay_ms, ignore_polarity=bool(p.ignore_polarity), median_k=p.median_blur_kernel_size)
        l = ots[0].render(t, map_x=rig.left.map_x, map_y=rig.left.map_y, **kw)
        rr = ots[1].render(t, map_x=rig.right.map_x, map_y=rig.right.map_y, **kw)
        path = pick(["sync", "lazy", "lazy", "images"])
        if path == "sync":
            gl = dev.ts_render(0, t, download=bool(rng.integers(2)))
            dev.ts_render(1, t, download=False)
            if gl is not None and not np.array_equal(gl, l):
                bad.append((k, "time surface"))
            dev.set_observation(t, None, None, T)
            dev.tick(t, stamps, poses)
        elif path == "lazy":
            dev.tick_resident(t, T, stamps, poses)
        else:
            dev.set_observation(t, l, rr, T)
            dev.tick(t, stamps, poses)
        m.set_observation(t, l, rr, T)
        m.set_poses(stamps, poses)
        staged = st.ev_left[first[0]:done[0]]
        idx = oracle.select_events(staged, t, p.bm_half_slice_thickness, p.process_event_num)
        if p.denoising:
            idx = oracle.denoise_events(staged, idx, rig.width, rig.height, p.process_event_num)
        m.tick(staged[idx])
        om = m.get_map()
        expect[t] = (len(om), map_sha1(om))
        t_last = t
        # ---- reads
        for _ in range(int(rng.integers(0, 3))):
            what = pick(["map", "committed", "frame", "cloud", "stats", "sync"])
            if what == "map":
                g = dev.get_map()
                if (len(g), map_sha1(g)) != expect[t]:
                    bad.append((k, f"map {len(g)} vs {expect[t][0]}"))
            elif what == "committed":
                g, gt = dev.get_committed_map()
                if gt in expect:
                    if (len(g), map_sha1(g)) != expect[gt]:
                        bad.append((k, f"committed map of {gt}: {len(g)} vs {expect[gt][0]}"))
                elif gt != 0 or len(g):
                    bad.append((k, f"committed stamp {gt} is no tick of this run"))
            elif what == "frame":
                g, o = dev.get_last_frame(), m.get_last_frame()
                if len(g) != len(o) or not np.array_equal(g["inv_depth"], o["inv_depth"]):
                    bad.append((k, f"frame {len(g)} vs {len(o)}"))
            elif what == "cloud":
                g, o = dev.get_pointcloud(), m.get_pointcloud()
                if g.shape != o.shape or not np.array_equal(g, o):
                    bad.append((k, f"cloud {g.shape} vs {o.shape}"))
            elif what == "stats":
                dev.stats()
            else:
                dev.synchronize()
    g = dev.get_map()
    if t_last is not None and (len(g), map_sha1(g)) != expect[t_last]:
        bad.append((n_ticks - 1, f"final map {len(g)} vs {expect[t_last][0]}"))
    dev.close()
    sizes = [v[0] for v in expect.values()]
    brief = f"{rig_name} patch {cfg['patch_size_X']}x{cfg['patch_size_Y']} {cfg['LSnorm']} q{ql} {cfg['node']} ticks {n_ticks} [{' '.join(log)}] last maps {sizes[-3:]}"
    return bad, brief


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 20
    s0 = int(sys.argv[2]) if len(sys.argv) > 2 else 50000
    failed = 0
    t0 = time.time()
    for seed in range(s0, s0 + n):
        try:
            bad, brief = run_case(seed)
        except Exception as e:  # noqa: BLE001
            bad, brief = [("-", f"{type(e).__name__}: {e}")], str(fuzz_parity.draw(seed)[0])
        failed += bool(bad)
        print(f"seed {seed}: {'EQUAL' if not bad else 'DIFFERENT ' + str(bad[:4])}  {brief}", flush=True)
    print(f"{n} cases, {failed} with a difference, {time.time() - t0:.0f} s")
    sys.exit(1 if failed else 0)


if __name__ == "__main__":
    main()
