"""Randomised end-to-end parity: the device's DepthMap against the canonical CPU oracle's, tick by tick, bit for bit, over
configurations NO shipped yaml has -- random rig, patch size, norm, Student-t constants, LM iteration cap, fusion radius /
strategy / window, regulariser on / off and its radius and counts, culling and visibility thresholds, denoising, Time-Surface
smoothing / decay / median / queue length, block-matching step / threshold / disparity window, thread-stride count, node
type, tick size -- each on its own seeded synthetic stream.  A difference is a bug in one of the two.
usage: python tools/fuzz_parity.py [cases] [first seed]      (GPU; prints one line per case, exits 1 on any difference)
FUZZ_TICKS_X=k runs k times as many ticks per case (long windows: pops, ages, a ring that wraps many times)."""
import os
import sys
import time

sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np  # noqa: E402

from benchlib.workload import map_sha1  # noqa: E402
from esvo_amd import calib, lib, params, rostime, synth  # noqa: E402
from oracle import oracle  # noqa: E402

RIGS = {"upenn": ("mapping_upenn", (0.16, 1.0)), "rpg": ("mapping_rpg", (0.2, 2.0)), "hkust": ("mapping_hkust", (0.25, 2.0)),
        "dsec": ("mapping_dsec", (0.02, 0.25))}


def draw(seed):
    rng = np.random.default_rng(seed)
    pick = lambda xs: xs[int(rng.integers(len(xs)))]  # noqa: E731
    rig_name = pick(["upenn", "upenn", "rpg", "hkust", "dsec"])
    preset, rho = RIGS[rig_name]
    cfg = dict(params.PRESETS[preset])
    px, py = pick([(15, 7), (15, 7), (15, 7), (25, 25), (10, 4), (5, 5), (9, 3)])
    cfg.update(
        patch_size_X=px, patch_size_Y=py,
        LSnorm=pick(["Tdist", "Tdist", "Tdist", "l2"]),
        Tdist_nu=float(pick([rng.uniform(2.05, 6.0), rng.uniform(2.05, 6.0), 2.001, 30.0])),
        Tdist_scale=float(pick([rng.uniform(5.0, 40.0), rng.uniform(5.0, 40.0), 0.5, 400.0])),
        ITERATION_OPTIMIZATION=pick([1, 3, 10, 10, 20]),
        Regularization=bool(rng.integers(2)), RegularizationRadius=pick([1, 3, 5, 12, 20]),
        RegularizationMinNeighbours=pick([1, 4, 8, 32]), RegularizationMinCloseNeighbours=pick([1, 4, 8, 32]),
        SmoothTimeSurface=bool(rng.integers(2)),
        residual_vis_threshold=float(rng.uniform(5, 40)),
        stdVar_vis_threshold=float(cfg["stdVar_vis_threshold"] * rng.uniform(0.3, 3.0)),
        age_max_range=pick([1, 3, 10]), age_vis_threshold=pick([0, 1, 2]),
        fusion_radius=pick([0, 0, 1, 2]),
        FUSION_STRATEGY=pick(["CONST_POINTS", "CONST_FRAMES"]),
        maxNumFusionFrames=pick([1, 2, 5, 20]), maxNumFusionPoints=pick([200, 1000, 5000]),
        Denoising=bool(rng.integers(3) == 0),
        PROCESS_EVENT_NUM=pick([50, 300, 1000, 3000, 3000, 12000]),
        BM_step=pick([1, 1, 2, 3]), BM_ZNCC_Threshold=pick([0.05, 0.1, 0.3, -1.0, 0.9]),
        BM_bUpDownConfiguration=bool(rng.integers(6) == 0),
        BM_min_disparity=pick([0, 1, 3]), BM_max_disparity=pick([20, 40, 150]),
        decay_ms=pick([10.0, 30.0, 100.0]), median_blur_kernel_size=pick([1, 1, 3]),
        node=pick(["mapping", "mvstereo"]),
    )
    over = dict(num_threads=int(rng.integers(1, 9)), max_event_queue_len=pick([0, 0, 0, 20, 3]))
    small_ring = bool(rng.integers(4) == 0)   # a ring that wraps over the run (sized in run_case from the first tick's staging)
    scene = dict(points=pick([3000, 8000, 20000]) * (4 if rig_name == "dsec" else 1), speed=float(rng.uniform(0.3, 2.0)),
                 seed=int(rng.integers(1 << 30)), ticks=int(rng.integers(4, 8)) * int(os.environ.get("FUZZ_TICKS_X", "1")), tick_ms=pick([5, 10, 20]), rho=rho,
                 # sync: the four calls, every map read at once; lazy: esvo_map_tick_resident, two ticks in flight, maps read one tick late
                 path=pick(["sync", "sync", "lazy"]), small_ring=small_ring)
    return rig_name, cfg, over, scene


def run_case(seed, verbose=False):
    rig_name, cfg, over, sc = draw(seed)
    rig = calib.dataset_rig(rig_name)
    p, _ = params.make_params(cfg, rig, **over)
    dur = 0.06 + (sc["ticks"] + 1) * sc["tick_ms"] * 1e-3
    st = synth.make_stream(rig, sc["points"], dur, sc["rho"][0], sc["rho"][1], seed=sc["seed"], speed=sc["speed"])
    if sc["small_ring"]:   # holds what is staged before the first render (+ 10 %), so the later ticks wrap it
        t1 = st.t0_ns + int((0.06 + sc["tick_ms"] * 1e-3) * 1e9)
        first = max(int(np.searchsorted(st.ns_left, t1)), int(np.searchsorted(st.ns_right, t1)))
        p.event_ring_capacity = max(1 << 14, int(first * 1.1) + 8192)
    dev = lib.Esvo(p, rig)
    m = oracle.OracleMapper(p, rig)
    m.set_mode(True, True)
    m.set_threads(os.cpu_count() or 1)
    ql = p.max_event_queue_len
    ts = [oracle.OracleTS(rig.width, rig.height, queue_len=ql or 20) for _ in range(2)]   # (queue 0 on the device = one stamp per pixel: the same raster while render times rise)
    done = [0, 0]
    bad, sizes = [], []
    lazy = sc["path"] == "lazy"
    prev = None
    for k in range(sc["ticks"]):
        t = st.t0_ns + int((0.06 + (k + 1) * sc["tick_ms"] * 1e-3) * 1e9)
        stamps, poses = rostime.pose_table(st.pose, t, p.bm_half_slice_thickness)
        T = st.pose(t)
        for cam, (ev, ns) in enumerate(((st.ev_left, st.ns_left), (st.ev_right, st.ns_right))):
            hi = int(np.searchsorted(ns, t, side="left"))
            for a in range(done[cam], hi, 8192):   # (packets that fit the smallest ring)
                dev.ts_push_events(cam, ev[a:min(a + 8192, hi)])
            ts[cam].push(ev[done[cam]:hi])
            done[cam] = hi
        if lazy:
            dev.tick_resident(t, T, stamps, poses)
            gl = gr = None
        else:
            gl, gr = dev.ts_render(0, t), dev.ts_render(1, t)
            dev.set_observation(t, None, None, T)
            dev.tick(t, stamps, poses)
        kw = dict(decay_ms=p.decay_ms, ignore_polarity=bool(p.ignore_polarity), median_k=p.median_blur_kernel_size)
        l = ts[0].render(t, map_x=rig.left.map_x, map_y=rig.left.map_y, **kw)
        r = ts[1].render(t, map_x=rig.right.map_x, map_y=rig.right.map_y, **kw)
        if gl is not None and not (np.array_equal(gl, l) and np.array_equal(gr, r)):
            bad.append((k, "time surface"))
        m.set_observation(t, l, r, T)
        m.set_poses(stamps, poses)
        staged = st.ev_left[:done[0]]
        idx = oracle.select_events(staged, t, p.bm_half_slice_thickness, p.process_event_num)
        if p.denoising:   # createDenoisingMask + extractDenoisedEvents (esvo_Mapping.cpp:282-296, 1046-1072) on the selected events
            idx = oracle.denoise_events(staged, idx, rig.width, rig.height, p.process_event_num)
        m.tick(staged[idx])
        om = m.get_map()
        sizes.append(len(om))
        if lazy:
            if prev is not None:
                gm, gt = dev.get_committed_map()
                if gt != prev[0] or len(gm) != prev[2] or map_sha1(gm) != prev[1]:
                    bad.append((k - 1, f"committed map {len(gm)} vs {prev[2]}"))
            prev = (t, map_sha1(om), len(om))
            continue
        gm = dev.get_map()
        if len(om) != len(gm) or map_sha1(om) != map_sha1(gm):
            bad.append((k, f"map {len(gm)} vs {len(om)}"))
        of, gf = m.get_last_frame(), dev.get_last_frame()
        if len(of) != len(gf) or not np.array_equal(of["inv_depth"], gf["inv_depth"]):
            bad.append((k, f"frame {len(gf)} vs {len(of)}"))
    if lazy:
        gm = dev.get_map()
        if len(gm) != prev[2] or map_sha1(gm) != prev[1]:
            bad.append((sc["ticks"] - 1, f"final map {len(gm)} vs {prev[2]}"))
    dev.close()
    brief = (f"{rig_name} patch {cfg['patch_size_X']}x{cfg['patch_size_Y']} {cfg['LSnorm']} it{cfg['ITERATION_OPTIMIZATION']} ev{cfg['PROCESS_EVENT_NUM']} "
             f"fr{cfg['fusion_radius']} {cfg['FUSION_STRATEGY'][6:]} reg{int(cfg['Regularization'])}/{cfg['RegularizationRadius']} dn{int(cfg['Denoising'])} "
             f"sm{int(cfg['SmoothTimeSurface'])} q{ql} thr{p.num_threads} step{cfg['BM_step']} ud{int(cfg['BM_bUpDownConfiguration'])} {cfg['node']} {sc['path']} ring{p.event_ring_capacity} ticks {sc['ticks']}x{sc['tick_ms']}ms maps {sizes}")
    return bad, brief


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 20
    s0 = int(sys.argv[2]) if len(sys.argv) > 2 else 1000
    failed = 0
    t0 = time.time()
    for seed in range(s0, s0 + n):
        try:
            bad, brief = run_case(seed)
        except Exception as e:  # noqa: BLE001
            bad, brief = [("-", f"{type(e).__name__}: {e}")], str(draw(seed)[:3])
        failed += bool(bad)
        print(f"seed {seed}: {'EQUAL' if not bad else 'DIFFERENT ' + str(bad[:4])}  {brief}", flush=True)
    print(f"{n} cases, {failed} with a difference, {time.time() - t0:.0f} s")
    sys.exit(1 if failed else 0)


if __name__ == "__main__":
    main()
