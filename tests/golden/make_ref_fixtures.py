#!/usr/bin/env python
"""Generates tests/golden/ref_*.npz from oracle/_ref/libesvo_ref.so -- the REFERENCE's own mapper sources
(EventBM, DepthProblem, DepthProblemSolver, DepthFusion, DepthRegularization, DepthPoint, SmartGrid, CameraSystem)
compiled unmodified from /root/reference against the stand-in headers of oracle/ref_shim/ (see oracle/Makefile,
oracle/ref_harness.cpp).  Runs only where /root/reference exists (the build container); the fixtures travel.

What the fixtures pin, and what they cannot:
  * pinned to reference source: block matching (event rejection tests, ZNCC cost, argmin/tie rule, thread-stride
    order), the residual functor DepthProblem::operator() (warping, bilinear patches, Student-t scale loop), the
    solver's driver loop / DepthPoint initialisation / culling, DepthPoint::update_studentT, propagation, fusion
    (incl. the replace branch), SmartGrid::clean / getNeighbourhood, the regulariser;
  * still restated third-party arithmetic (not in /root/reference): Eigen's LevenbergMarquardt/NumericalDiff (the
    stand-in is a second, independent MINPACK restatement), Eigen's 4x4 inverse (cofactor form in the stand-in),
    OpenCV's calibration maps / remap / median / Gaussian (inputs here, computed by esvo_amd/calib.py and the oracle).

ref_node.npz comes from the mapper NODE object (esvo_Mapping.cpp compiled unmodified, oracle/ref_harness_node.cpp) fed
through its callbacks: what dataTransferring / MappingAtTime / InitializationAtTime do around the classes above.

    python tests/golden/make_ref_fixtures.py
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import scenarios as S  # noqa: E402
from oracle import ref as R  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))


def run_reference(sc, ticks):
    """-> (per-tick outputs, unit vectors) of the reference code on a scenario's inputs"""
    m = R.RefMapper(sc.params, sc.rig)
    out = []
    unit = {}
    rng = np.random.default_rng(sc.spec["seed"] + 7)
    for k, tk in enumerate(ticks):
        m.set_observation(tk["t"], tk["tsL"], tk["tsR"], tk["T"])
        m.set_poses(tk["stamps"], tk["poses"])
        mt = m.match(tk["ev"])
        pts = m.refine(mt, cull=True)
        if k == 0 and len(mt):
            # residual vectors of DepthProblem::operator() and single-problem solutions on a sample of matches
            pick = rng.choice(len(mt), size=min(40, len(mt)), replace=False)
            rho = mt["inv_depth"][pick] * (1 + 2e-3 * rng.standard_normal(len(pick)))
            rho[::4] = mt["inv_depth"][pick][::4]
            fv = np.stack([m.eval_residual(mt["x_left"][i], mt["pose_idx"][i], r)[0] for i, r in zip(pick, rho)])
            sol = [m.solve_single(mt["x_left"][i], mt["pose_idx"][i], mt["inv_depth"][i]) for i in pick]
            unit.update(res_pick=pick, res_rho=rho, res_fvec=fv, sol_result=np.stack([s[0] for s in sol]),
                        sol_ok=np.array([s[1] for s in sol]))
        m.push_frame(pts, tk["poses"])
        nf = m.fuse()
        out.append(dict(matches=mt, points=pts, nf=nf, map=m.get_map(), cells=m.get_map_cells()))
    # cam2World / world2Cam samples (CameraSystem.cpp:121-148)
    xs = rng.uniform([0, 0], [sc.rig.width, sc.rig.height], size=(64, 2))
    rhos = rng.uniform(sc.params.invdepth_min, sc.params.invdepth_max, 64)
    pc = np.stack([m.cam2world(x, r) for x, r in zip(xs, rhos)])
    unit.update(c2w_x=xs, c2w_rho=rhos, c2w_p=pc, w2c_left=np.stack([m.world2cam(p) for p in pc]),
                w2c_right=np.stack([m.world2cam(p, right=True) for p in pc]), baseline=m.baseline,
                dangling=m.counters()["dangling_cells"])
    return out, unit


MAP_SHA_FIELDS = ("row", "col", "age", "inv_depth", "scale2", "nu", "variance", "residual", "x")   # every field but p_cam (1e-12)


def ref_map_digest(mp, cells=None):
    """sha256 over the list order and every exactly reproducible field of a DepthMap dump (+ the elements' true cells)"""
    import hashlib
    h = hashlib.sha256()
    for f in MAP_SHA_FIELDS:
        h.update(np.ascontiguousarray(mp[f]).tobytes())
    if cells is not None:
        h.update(np.ascontiguousarray(cells, np.uint32).tobytes())
    return np.frombuffer(h.digest(), np.uint8).copy()


def make(name):
    sc = S.Scenario(name)
    ticks = sc.inputs()
    res, unit = run_reference(sc, ticks)
    out = dict(scenario=name, n_ticks=len(ticks), smooth=int(sc.params.smooth_time_surface))
    big = name in S.BIG   # shipped tick sizes: digests of the maps, the last one in full
    for k, (tk, r) in enumerate(zip(ticks, res)):
        # the un-smoothed pair is stored; consumers re-apply GaussianBlurTS(5) where the preset asks for it
        out.update({f"t{k}": tk["t"], f"tsL{k}": tk["raw"][0], f"tsR{k}": tk["raw"][1], f"T{k}": tk["T"],
                    f"stamps{k}": tk["stamps"], f"poses{k}": tk["poses"], f"ev{k}": tk["ev"],
                    f"matches{k}": r["matches"], f"points{k}": r["points"], f"nf{k}": r["nf"]})
        if not big or k == len(ticks) - 1:
            out.update({f"map{k}": r["map"], f"cells{k}": r["cells"]})
        if big:
            out.update({f"map_n{k}": len(r["map"]), f"map_sha{k}": ref_map_digest(r["map"]),
                        f"map_cells_sha{k}": ref_map_digest(r["map"], r["cells"])})
    out.update({"u_" + k: v for k, v in unit.items()})
    path = os.path.join(HERE, f"ref_{name}.npz")
    np.savez_compressed(path, **out)
    print(name, os.path.getsize(path) // 1024, "KiB", [(len(r["matches"]), len(r["points"]), r["nf"], len(r["map"])) for r in res],
          "dangling", unit["dangling"])


def make_units():
    """DepthPoint::update_studentT (DepthPoint.cpp:167-188) and EventBM::zncc_cost (EventBM.cpp:317-333, utils.h:74-92) on
    random inputs"""
    rng = np.random.default_rng(20250601)
    states, obs, new = [], [], []
    for _ in range(400):
        fresh = rng.random() < 0.2
        st = np.array([-1.0, 0, 0, 0, 0]) if fresh else np.array(
            [rng.uniform(0.05, 2), rng.uniform(1e-6, 1e-2), rng.uniform(2.1, 12), 0.0, rng.integers(0, 9)])
        if not fresh:
            st[3] = st[2] / (st[2] - 2) * st[1]
        nu = rng.uniform(2.1, 12)
        s2 = rng.uniform(1e-6, 1e-2)
        ob = np.array([rng.uniform(0.05, 2), s2, nu / (nu - 2) * s2, nu])
        states.append(st)
        obs.append(ob)
        new.append(R.update_student_t(st, *ob))
    patches_l = rng.integers(0, 256, size=(200, 7, 15)).astype(np.float64)
    patches_r = np.clip(patches_l + rng.normal(0, 25, size=patches_l.shape), 0, 255).round()
    patches_l[:5] = 0          # flat patches: sigma = 1e-6 (utils.h:80)
    patches_r[5:10] = 17
    cost = np.array([R.zncc_cost(l, r) for l, r in zip(patches_l, patches_r)])
    path = os.path.join(HERE, "ref_units.npz")
    np.savez_compressed(path, st_state=np.stack(states), st_obs=np.stack(obs), st_new=np.stack(new),
                        zncc_l=patches_l.astype(np.uint8), zncc_r=patches_r.astype(np.uint8), zncc_cost=cost)
    print("units", os.path.getsize(path) // 1024, "KiB")


def track_inputs():
    """a 346x260 Time Surface, a cloud of 900 scene points seen from it, reference pose 20 ms earlier"""
    from esvo_amd import calib, synth
    from oracle import oracle as O
    rig = calib.dataset_rig("upenn")
    st = synth.make_stream(rig, 8000, 0.2, 0.16, 1.0, seed=7, speed=1.0)
    t = st.t0_ns + int(0.1e9)
    ts = O.OracleTS(rig.width, rig.height)
    ts.push(st.ev_left[st.ns_left < t])
    L = ts.render(t, map_x=rig.left.map_x, map_y=rig.left.map_y)
    u, v, rho = st.true_inv_depth_image(t)
    ok = (u >= 10) & (u < rig.width - 10) & (v >= 10) & (v < rig.height - 10)
    Tw = st.pose(t)
    K = np.array(rig.left.P).reshape(3, 4)[:, :3]
    pc = (np.linalg.inv(K) @ np.stack([u[ok], v[ok], np.ones(ok.sum())])) / rho[ok]
    pw = ((Tw[:3, :3] @ pc).T + Tw[:3, 3])[:900].astype(np.float32)
    return rig, L, pw, st.pose(t - 20_000_000), st.pose(t)


def make_track():
    """RegProblemLM::setProblem / operator() / df (RegProblemLM.cpp:26-269) on the oracle's negated blurred Time Surface and
    its Sobel derivatives (OpenCV products, injected: oracle/ref_harness_track.cpp)"""
    from oracle import oracle as O
    rig, L, pw, T_ref, T_left = track_inputs()
    ot = O.OracleTracker(rig)
    ot.set_current(L, 5)
    neg, du, dv = ot.images()
    out = dict(ts_left=L, xyz_world=pw, T_world_ref=T_ref, T_world_left=T_left)
    for norm, huber in (("huber", True), ("l2", False)):
        rt = R.RefTracker(rig, huber=huber, huber_threshold=50.0, max_points=700)
        order, Rm, tv = rt.set_problem(neg, du, dv, pw, T_ref, T_left, seed=3)
        out["order"], out["n"], out["R"], out["t"] = order, rt.n, Rm, tv
        xs = [np.zeros(6), np.array([1e-3, -2e-3, 5e-4, 2e-3, -1e-3, 3e-3]), np.array([-4e-3, 1e-3, 2e-3, -5e-3, 4e-3, 1e-3])]
        for i, x in enumerate(xs):
            f, Tw = rt.residuals(100, 300, x)
            out[f"{norm}_x{i}"], out[f"{norm}_f{i}"], out[f"{norm}_T{i}"] = x, f, Tw
        out[f"{norm}_f_tail"], out[f"{norm}_T_tail"] = rt.residuals(600, 300, None)   # a batch cut short by the point count
        out[f"{norm}_J"] = rt.jacobian(100, 300)
    path = os.path.join(HERE, "ref_track.npz")
    np.savez_compressed(path, **out)
    print("track", os.path.getsize(path) // 1024, "KiB")


def track_solve_starts(T_left):
    """start poses of the registration cases: the true pose, the reference frame's pose (what the node starts from), and
    the true pose displaced by ~2, 5, 10 and 20 mm / a proportional rotation"""
    from esvo_amd import closed_loop as cl
    rng = np.random.default_rng(11)
    starts = [("truth", T_left.copy()), ("ref_pose", None)]
    for s in (0.002, 0.005, 0.01, 0.02):
        T0 = T_left.copy()
        T0[:3, :3] = T_left[:3, :3] @ cl.orth(cl.cayley2rot(rng.normal(0, s / 2, 3)))
        T0[:3, 3] += rng.normal(0, s, 3)
        starts.append((f"pert{int(s * 1000)}mm", T0))
    return starts


def make_track_solve():
    """The reference's tracker LOOP -- RegProblemSolverLM::solve_analytical (RegProblemSolverLM.cpp:148-178) around the
    reference's own functor, its LM class ref_shim's MINPACK restatement (oracle/ref_harness_track.cpp: ref_tracker_solve) --
    from six start poses, with the yaml's BATCH_SIZE 300 and with one batch: the poses it ends at.  tests/test_track_normal.py
    bounds the deviation of esvo_hip::gauss_newton_register (Levenberg damping, not lmpar) against them."""
    from oracle import oracle as O
    rig, L, pw, T_ref, T_left = track_inputs()
    ot = O.OracleTracker(rig)
    ot.set_current(L, 5)
    neg, du, dv = ot.images()
    out = dict(T_world_left_true=T_left)
    names = []
    for name, T0 in track_solve_starts(T_left):
        T0 = T_ref.copy() if T0 is None else T0
        names.append(name)
        out[f"{name}_T0"] = T0
        for B in (300, 0):
            rt = R.RefTracker(rig, huber=True, huber_threshold=50.0, max_points=2000)
            order, R0, t0 = rt.set_problem(neg, du, dv, pw, T_ref, T0, seed=3)
            Rr, tr, it, nfev, st = rt.solve(B or rt.n, 10)
            out["order"], out["n"] = order, rt.n
            out[f"{name}_R0"], out[f"{name}_t0"] = R0, t0
            out[f"{name}_b{B}"] = np.concatenate([Rr.reshape(9), tr, [it, nfev, st]])
    out["names"] = np.array(names)
    path = os.path.join(HERE, "ref_track_solve.npz")
    np.savez_compressed(path, **out)
    print("track_solve", os.path.getsize(path) // 1024, "KiB")


def sgm_inputs():
    """observation pair, SGM events and edgelet coordinates of the bootstrap case (tests/test_sgm.py uses the same stream)"""
    from esvo_amd import calib, params, synth
    from oracle import oracle as O
    rig = calib.dataset_rig("upenn")
    st = synth.make_stream(rig, 8000, 0.35, 0.16, 1.0, seed=20250419, speed=1.0)
    p, _ = params.make_params(params.PRESETS["mapping_upenn"], rig)
    t0 = st.t0_ns + int(0.08e9)
    ts = [O.OracleTS(rig.width, rig.height), O.OracleTS(rig.width, rig.height)]
    ts[0].push(st.ev_left[st.ns_left < t0])
    ts[1].push(st.ev_right[st.ns_right < t0])
    l0 = ts[0].render(t0, map_x=rig.left.map_x, map_y=rig.left.map_y)
    r0 = ts[1].render(t0, map_x=rig.right.map_x, map_y=rig.right.map_y)
    ev = st.ev_left[O.select_events_sgm(st.ev_left, t0, p.bm_half_slice_thickness, p.process_event_num)]
    lut = np.array(rig.left.rect_lut).reshape(rig.height, rig.width, 2)
    xy = []
    for e in ev:                                       # createEdgeMask, radius 0 (esvo_Mapping.cpp:1000-1044)
        xc, yc = int(np.floor(lut[e["y"], e["x"], 0])), int(np.floor(lut[e["y"], e["x"], 1]))
        if 0 <= xc < rig.width and 0 <= yc < rig.height:
            xy.append((xc, yc))
    return rig, p, t0, l0, r0, ev, np.array(xy, np.uint32), st.pose(t0)


def make_sgm():
    """InitializationAtTime behind the StereoSGBM call (esvo_Mapping.cpp:455-487) + DepthFusion::naive_propagation
    (DepthFusion.cpp:234-327) of the reference, fed with the ORACLE's disparity image (StereoSGBM is OpenCV: unpinned)"""
    from oracle import oracle as O
    rig, p, t0, l0, r0, ev, xy, T = sgm_inputs()
    m = O.OracleMapper(p, rig)
    m.set_observation(t0, l0, r0, T)
    n, disp = m.init_sgm(l0, r0, ev, min_points=100)
    rm = R.RefMapper(p, rig)
    rm.set_observation(t0, l0, r0, T)
    nr = rm.init_from_disparity(disp, xy, min_points=100)
    path = os.path.join(HERE, "ref_sgm.npz")
    np.savez_compressed(path, n_points=nr, ref_map=rm.get_map(), n_oracle=n)
    print("sgm", os.path.getsize(path) // 1024, "KiB", n, nr)


def ts_inputs():
    from esvo_amd import calib, synth
    rig = calib.dataset_rig("upenn")
    st = synth.make_stream(rig, 6000, 0.12, 0.16, 1.0, seed=11, speed=1.0)
    return rig, st


def make_ts():
    """TimeSurface::eventsCallback + createTimeSurfaceAtTime (TimeSurface.cpp:52-152, 403-425; EventQueueMat of
    TimeSurface.h:28-96) on 1 ms messages: per-pixel queues of 20 and of 3 events, renders at the newest stamp and at stamps
    5 and 11 ms EARLIER than events already inserted (getMostRecentEventBeforeT walks the queue back).  Stored: the f64
    images the node hands to convertTo(CV_8U), rounded as convertTo rounds, and their f64 sums."""
    rig, st = ts_inputs()
    out = {}
    for ql in (20, 3):
        ts = R.RefTS(rig.width, rig.height, 30.0, True, ql)
        done = 0
        for k in range(1, 9):
            tk = st.t0_ns + k * 12_000_000
            hi = int(np.searchsorted(st.ns_left, tk))
            ts.push(st.ev_left[done:hi])
            done = hi
            for j, back in enumerate((0, 5_000_000, 11_000_000)):
                img = ts.render(tk - back)
                # cv::Mat::convertTo(CV_8U) = saturate_cast<uchar>(cvRound(v)), round half to even (OpenCV: not in the build)
                out[f"q{ql}_k{k}_b{j}"] = np.clip(np.rint(img), 0, 255).astype(np.uint8)
                out[f"q{ql}_k{k}_b{j}_sum"] = float(img.sum())
    path = os.path.join(HERE, "ref_ts.npz")
    np.savez_compressed(path, **out)
    print("ts", os.path.getsize(path) // 1024, "KiB")


def jitter_arrival(ev, ns, seed=5, bundle_ns=250_000, jitter_ns=200_000):
    """Arrival order of a stream whose 250 us bundles are each delivered up to +-200 us off their time: bundles overtake each other
    (a late bundle's first event lies before stamps already delivered) -- what a driver that assembles packets from several USB
    transfers does.  Returns the permutation (stable inside a bundle)."""
    rng = np.random.default_rng(seed)
    bundle = (ns - ns[0]) // bundle_ns
    off = rng.integers(-jitter_ns, jitter_ns + 1, int(bundle.max()) + 1)
    key = ns.astype(np.int64) - (ns.astype(np.int64) - ns[0].astype(np.int64)) % bundle_ns + off[bundle]   # the bundle's delivery time
    return np.argsort(key, kind="stable")


def ts_jitter_cases():
    """(arrival-ordered left events, [(first, last) of each 1 ms delivery], render stamps) of the jitter fixture"""
    rig, st = ts_inputs()
    perm = jitter_arrival(st.ev_left, st.ns_left)
    ev = st.ev_left[perm]
    n = len(ev)
    cuts = list(range(0, n, max(n // 96, 1))) + [n]
    chunks = [(a, b) for a, b in zip(cuts[:-1], cuts[1:]) if b > a]
    renders = [st.t0_ns + k * 12_000_000 for k in range(2, 9)]
    return rig, st, ev, chunks, renders


def make_ts_jitter():
    """TimeSurface::eventsCallback on OUT-OF-ORDER deliveries (TimeSurface.cpp:403-425: insertion sort into events_, then
    insertEvent(events_.back()) -- SURVEY Appendix A-1): the reference's own class is fed the jittered stream delivery by
    delivery; before each delivery whose first stamp reaches a render stamp, the surface is rendered at that stamp (so render
    times never decrease).  Stored: the rounded images (queue length 20, the node's default, and 3) and their f64 sums."""
    rig, st, ev, chunks, renders = ts_jitter_cases()
    from esvo_amd.abi import event_ns
    ns = event_ns(ev)
    out = {"n_late": 0}
    run_max = np.maximum.accumulate(ns)
    out["n_late"] = int((ns[1:] < run_max[:-1]).sum())
    for ql in (20, 3):
        ts = R.RefTS(rig.width, rig.height, 30.0, True, ql)
        ri = 0
        for a, b in chunks:
            ts.push(ev[a:b])
            done_max = int(run_max[b - 1])
            while ri < len(renders) and done_max >= renders[ri] + 2_000_000:   # everything before the stamp (+ jitter margin) has arrived
                img = ts.render(renders[ri])
                out[f"q{ql}_r{ri}"] = np.clip(np.rint(img), 0, 255).astype(np.uint8)
                out[f"q{ql}_r{ri}_sum"] = float(img.sum())
                out[f"q{ql}_r{ri}_after"] = b          # the render happens after this many events have been delivered
                ri += 1
        out[f"q{ql}_renders"] = ri
    path = os.path.join(HERE, "ref_ts_jitter.npz")
    np.savez_compressed(path, **out)
    print("ts jitter", os.path.getsize(path) // 1024, "KiB", "late events", out["n_late"], "of", len(ev), "renders", out["q20_renders"])


def make_bm_step():
    """EventBM with BM_step = 2 and 3 (EventBM.cpp:113-138,169-225: coarse pass on the stride grid, the rule that both
    stride neighbours of the coarse minimum must have been evaluated on a valid patch, fine pass around it with the minimum
    carried over, size_t arithmetic at the lower end) on tick 1 of two scenarios.  No shipped configuration sets it; the
    device rejects it (ESVO_ERR_UNSUPPORTED), the oracle carries it for the day the kernel does."""
    import copy
    out = {}
    for name in ("upenn", "dsec"):
        sc = S.Scenario(name)
        tk = sc.inputs()[1]
        for step, updown in ((2, 0), (3, 0), (1, 1), (2, 1)):   # + BM_bUpDownConfiguration (:180-183,146-151): the search runs along y
            p = copy.copy(sc.params)
            p.bm_step = step
            p.bm_updown = updown
            r = R.RefMapper(p, sc.rig)
            r.set_observation(tk["t"], tk["tsL"], tk["tsR"], tk["T"])
            r.set_poses(tk["stamps"], tk["poses"])
            mt = r.match(tk["ev"])
            out[f"{name}_s{step}" + ("_ud" if updown else "")] = mt
            print("bm_step", name, step, updown, len(mt))
    path = os.path.join(HERE, "ref_bm_step.npz")
    np.savez_compressed(path, **out)
    print("ref_bm_step.npz", os.path.getsize(path) // 1024, "KiB")


L2_SCENARIOS = ("upenn", "hkust")
L2_MAP_FIELDS = ("row", "col", "age", "inv_depth", "variance", "residual", "x")   # nu / scale are never set on this path (A-8)


def l2_digest(mp):
    import hashlib
    h = hashlib.sha256()
    for f in L2_MAP_FIELDS:
        h.update(np.ascontiguousarray(mp[f]).tobytes())
    return np.frombuffer(h.digest(), np.uint8).copy()


def make_l2():
    """LSnorm: l2 -- the Gaussian model beside the Student-t one: the plain temporal residual (DepthProblem.cpp:43-45,67-75),
    the covariance from |f|^2 / (m - n) (DepthProblemSolver.cpp:200-206), DepthPoint::update (DepthPoint.cpp:146-164),
    variance propagation, chiSquareTest and Gaussian fusion (DepthFusion.cpp:49-53,130-131,150-152,164-165) and the
    inverse-variance mean of the regulariser (DepthRegularization.cpp:56-65), on two scenarios (hkust regularises).  The
    matches are those of ref_<name>.npz (block matching does not depend on the norm).  No shipped configuration sets it; the
    device rejects it, the oracle carries it."""
    import copy
    from esvo_amd.abi import LSNORM_L2
    out = {}
    for name in L2_SCENARIOS:
        sc = S.Scenario(name)
        ticks = sc.inputs()
        g = np.load(os.path.join(HERE, f"ref_{name}.npz"))
        p = copy.copy(sc.params)
        p.ls_norm = LSNORM_L2
        r = R.RefMapper(p, sc.rig)
        sizes = []
        for k, tk in enumerate(ticks):
            r.set_observation(tk["t"], tk["tsL"], tk["tsR"], tk["T"])
            r.set_poses(tk["stamps"], tk["poses"])
            mt = r.match(tk["ev"])
            assert mt.tobytes() == np.ascontiguousarray(g[f"matches{k}"]).tobytes()
            pts = r.refine(mt, cull=True)
            r.push_frame(pts, tk["poses"])
            nf = r.fuse()
            mp = r.get_map()
            out.update({f"{name}_points{k}": pts, f"{name}_nf{k}": nf, f"{name}_map_n{k}": len(mp), f"{name}_map_sha{k}": l2_digest(mp)})
            sizes.append((len(pts), nf, len(mp)))
        for f in L2_MAP_FIELDS:
            out[f"{name}_last_{f}"] = mp[f] if mp[f].dtype.kind == "f" else mp[f].astype(np.uint16)
        print("l2", name, sizes)
    path = os.path.join(HERE, "ref_l2.npz")
    np.savez_compressed(path, **out)
    print("ref_l2.npz", os.path.getsize(path) // 1024, "KiB")


def make_ts_forward():
    """The same node class in FORWARD mode (TimeSurface.cpp:85-116: bilinear splat at the rectified pixel positions with a
    clamp after every add), the rectified positions = the upenn rig's left rect_lut (cv::undistortPoints' output, an OpenCV
    product: injected).  Stored: the image handed to convertTo(CV_8U), rounded as convertTo rounds, and the sha-256 of its
    f64 bytes; queue of 20, with and without polarity, stamps at and 5 ms before the newest events."""
    import hashlib
    rig, st = ts_inputs()
    lut = np.array(rig.left.rect_lut, np.float32).reshape(-1, 2)
    out = {}
    for pol in (0, 1):
        ts = R.RefTS(rig.width, rig.height, 30.0, not pol, 20)
        ts.set_forward(lut)
        done = 0
        for k in range(1, 5):
            tk = st.t0_ns + k * 12_000_000
            hi = int(np.searchsorted(st.ns_left, tk))
            ts.push(st.ev_left[done:hi])
            done = hi
            for j, back in enumerate((0, 5_000_000)):
                img = ts.render(tk - back)
                out[f"p{pol}_k{k}_b{j}"] = np.clip(np.rint(img), 0, 255).astype(np.uint8)
                out[f"p{pol}_k{k}_b{j}_sha"] = np.frombuffer(hashlib.sha256(np.ascontiguousarray(img).tobytes()).digest(), np.uint8).copy()
    path = os.path.join(HERE, "ref_ts_forward.npz")
    np.savez_compressed(path, **out)
    print("ts forward", os.path.getsize(path) // 1024, "KiB")


NODE_SCENARIOS = ("dsec", "hkust")   # the two whose preset is the Mapping node's (upenn / rpg follow esvo_MVStereo)
MVSTEREO_SCENARIOS = ("upenn", "rpg")  # ... and the two that follow esvo_MVStereo
NODE_MAP_FIELDS = ("row", "col", "age", "inv_depth", "scale2", "nu", "variance", "residual", "x")


BM_ONLY_FIELDS = ("row", "col", "age", "residual")   # exact; inverse depth, variance, x: to 1e-12 (propagated p_cam)


def fields_digest(mp, fields):
    import hashlib
    h = hashlib.sha256()
    for f in fields:
        h.update(np.ascontiguousarray(mp[f]).tobytes())
    return np.frombuffer(h.digest(), np.uint8).copy()


def map_digest(mp):
    """sha256 over the list order and every field of a DepthMap dump that the mapper node defines"""
    import hashlib
    h = hashlib.sha256()
    for f in NODE_MAP_FIELDS:
        h.update(np.ascontiguousarray(mp[f]).tobytes())
    return np.frombuffer(h.digest(), np.uint8).copy()


# the same node with FUSION_STRATEGY = CONST_POINTS (esvo_Mapping.cpp:341-353): 1.5 x 1000 points keep two or three ~600-point
# frames, and "window full" (the clean rule, :385) is two frames
NODE_CONST_POINTS = dict(fusion_strategy=1, max_fusion_points=1000, max_fusion_frames=2)


NODE_VISUALIZE_RANGE = dict(dsec=12.0, hkust=1.5)   # visualize_range of the node (pc_near_), chosen to cut each scene's cloud


def cloud_digest(xyz):
    import hashlib
    return np.frombuffer(hashlib.sha256(np.ascontiguousarray(xyz, np.float32).tobytes()).digest(), np.uint8).copy()


def run_node(sc, ticks, st, regularization, mvstereo=False, **override):
    """the reference's esvo_Mapping node object on a scenario: the whole left stream through eventsCallback, the tick's
    Time-Surface pair through timeSurfaceCallback, poses through the tf stand-in, then dataTransferring + MappingAtTime"""
    import copy
    p = copy.copy(sc.params)
    p.regularization = int(regularization)
    for k, v in override.items():
        setattr(p, k, v)
    vr = NODE_VISUALIZE_RANGE.get(sc.name)
    node = R.RefNode(p, sc.rig, st.pose, mvstereo=mvstereo, extra={"visualize_range": vr} if vr and not mvstereo else None)
    node.push_events(st.ev_left)
    out = []
    for tk in ticks:
        node.push_observation(tk["t"], tk["tsL"], tk["tsR"])
        assert node.data_transferring()
        sel = node.selected_events()
        stamps, poses = node.pose_table()
        node.mapping_at_time()
        out.append(dict(obs_t=node.obs_time(), sel=sel, matched=node.matched_events(), stamps=stamps, poses=poses,
                        window=np.array(node.window(), np.uint32), frame=node.newest_frame(), map=node.get_map()))
        if not mvstereo:   # publishPointCloud's two clouds (esvo_Mapping.cpp:909-934)
            out[-1].update(pc=node.pointcloud(), pc_near=node.pointcloud(near=True))
    return out


def node_init():
    """the node while ESVO_System_Status_ is INITIALIZATION: dataTransferring (identity pose :508-513, the SGM events
    :538-552) and InitializationAtTime (:433-492) with the ORACLE's disparity image standing in for StereoSGBM::compute"""
    from esvo_amd import synth
    from oracle import oracle as O
    rig, p, t0, l0, r0, ev, xy, _ = sgm_inputs()
    st = synth.make_stream(rig, 8000, 0.35, 0.16, 1.0, seed=20250419, speed=1.0)   # the stream of sgm_inputs()
    m = O.OracleMapper(p, rig)
    m.set_observation(t0, l0, r0, np.eye(4))
    _, disp = m.init_sgm(l0, r0, ev, min_points=100)
    node = R.RefNode(p, rig, st.pose, extra={"INIT_SGM_DP_NUM_THRESHOLD": 100})
    node.set_status("INITIALIZATION")
    node.push_events(st.ev_left)
    node.push_observation(t0, l0, r0)
    assert node.data_transferring() and node.obs_time() == t0
    sel = node.sgm_events()
    assert st.ev_left[sel].tobytes() == np.ascontiguousarray(ev).tobytes()
    assert node.initialization_at_time(disp)
    mp = node.get_map()
    # oracle/ref_harness.cpp's restatement of the node's glue (:455-480), same disparity, identity pose: identical
    rm = R.RefMapper(p, rig)
    rm.set_observation(t0, l0, r0, np.eye(4))
    rm.init_from_disparity(disp, xy, min_points=100)
    ref = rm.get_map()
    assert len(mp) == len(ref) and all(np.array_equal(mp[f], ref[f]) for f in NODE_MAP_FIELDS + ("p_cam",))
    print("node init", len(sel), len(mp))
    return dict(init_sel=sel, init_row=mp["row"].astype(np.uint16), init_col=mp["col"].astype(np.uint16),
                init_inv_depth=mp["inv_depth"], init_variance=mp["variance"])


def make_node():
    """esvo_Mapping.cpp itself (eventsCallback :669-703, timeSurfaceCallback :718-760, dataTransferring :494-600, getPoseAt
    :602-667, MappingAtTime :261-431, createDenoisingMask / extractDenoisedEvents :975-998, 1046-1062) compiled unmodified
    (oracle/ref_harness_node.cpp, oracle/ref_shim_node/).  Recorded without the regulariser: with it the node reads erased
    list elements through the grid, which no restatement can reproduce; the regulariser is pinned on
    defined input by ref_<scenario>.npz."""
    out = {}
    for name in NODE_SCENARIOS:
        sc = S.Scenario(name)
        ticks, st = sc.inputs(), sc.stream()
        g = np.load(os.path.join(HERE, f"ref_{name}.npz"))
        res = run_node(sc, ticks, st, regularization=False)
        res_reg = run_node(sc, ticks, st, regularization=True)
        out[f"{name}_n_ticks"] = len(ticks)
        for k, (tk, r, rr) in enumerate(zip(ticks, res, res_reg)):
            # the node's virtual-view poses are the callback's answers at the stamps it chose: only the stamps are the node's
            assert np.array_equal(r["poses"], np.asarray(tk["poses"]).reshape(-1, 4, 4))
            # the frame the node's own BM + LM + culling kept == the frame ref_<name>.npz records from oracle/ref_harness.cpp
            fr, ref = r["frame"], g[f"points{k}"]
            assert len(fr) == len(ref) and all(np.array_equal(fr[f], ref[f]) for f in NODE_MAP_FIELDS + ("pose_idx", "p_cam"))
            # with the regulariser the node differs from the defined-input run only in inverse depths (Appendix A-7)
            assert all(np.array_equal(rr["map"][f], g[f"map{k}"][f]) for f in NODE_MAP_FIELDS if f != "inv_depth")
            pre = f"{name}_"
            out.update({pre + f"obs_t{k}": r["obs_t"], pre + f"sel{k}": r["sel"], pre + f"matched{k}": r["matched"],
                        pre + f"stamps{k}": r["stamps"], pre + f"window{k}": r["window"], pre + f"map_n{k}": len(r["map"]),
                        pre + f"map_sha{k}": map_digest(r["map"]),
                        pre + f"reg_same_inv_depth{k}": int(np.sum(rr["map"]["inv_depth"] == g[f"map{k}"]["inv_depth"])),
                        pre + f"pc_n{k}": len(r["pc"]), pre + f"pc_sha{k}": cloud_digest(r["pc"]),
                        pre + f"pc_near_n{k}": len(r["pc_near"]), pre + f"pc_near_sha{k}": cloud_digest(r["pc_near"])})
            assert 0 < len(r["pc_near"]) < len(r["pc"]) == len(r["map"])
        if name == "dsec":
            from esvo_amd.abi import FUSION_CONST_POINTS
            assert NODE_CONST_POINTS["fusion_strategy"] == FUSION_CONST_POINTS
            res_cp = run_node(sc, ticks, st, regularization=False, **NODE_CONST_POINTS)
            for k, r in enumerate(res_cp):
                assert all(np.array_equal(r["frame"][f], g[f"points{k}"][f]) for f in NODE_MAP_FIELDS)
                out.update({f"dsec_cp_window{k}": r["window"], f"dsec_cp_map_n{k}": len(r["map"]),
                            f"dsec_cp_map_sha{k}": map_digest(r["map"])})
            print("node dsec CONST_POINTS", [(r["window"].tolist(), len(r["map"])) for r in res_cp])
        last = res[-1]["map"]
        for f in NODE_MAP_FIELDS:
            out[f"{name}_last_{f}"] = last[f] if last[f].dtype.kind == "f" else last[f].astype(np.uint16)
        print("node", name, [(len(r["sel"]), len(r["matched"]), len(r["stamps"]), r["window"].tolist(), len(r["map"]))
                             for r in res])
    # the other node, esvo_MVStereo.cpp (BM_PLUS_ESTIMATION; cleans every tick, selects up to 10 000 events and cuts to
    # PROCESS_EVENT_NUM in MappingAtTime: :383-405, 496-497, 627-647), on the two scenarios with its presets
    for name in MVSTEREO_SCENARIOS:
        sc = S.Scenario(name)
        ticks, st = sc.inputs(), sc.stream()
        g = np.load(os.path.join(HERE, f"ref_{name}.npz"))
        res = run_node(sc, ticks, st, regularization=False, mvstereo=True)
        res_reg = run_node(sc, ticks, st, regularization=bool(sc.params.regularization), mvstereo=True)
        out[f"mvs_{name}_n_ticks"] = len(ticks)
        for k, (tk, r, rr) in enumerate(zip(ticks, res, res_reg)):
            assert np.array_equal(r["poses"], np.asarray(tk["poses"]).reshape(-1, 4, 4))
            fr, ref = r["frame"], g[f"points{k}"]
            assert len(fr) == len(ref) and all(np.array_equal(fr[f], ref[f]) for f in NODE_MAP_FIELDS + ("pose_idx", "p_cam"))
            assert all(np.array_equal(rr["map"][f], g[f"map{k}"][f]) for f in NODE_MAP_FIELDS if f != "inv_depth")
            pre = f"mvs_{name}_"
            out.update({pre + f"obs_t{k}": r["obs_t"], pre + f"sel{k}": r["sel"], pre + f"matched{k}": r["matched"],
                        pre + f"stamps{k}": r["stamps"], pre + f"window{k}": r["window"], pre + f"map_n{k}": len(r["map"]),
                        pre + f"map_sha{k}": map_digest(r["map"]),
                        pre + f"reg_same_inv_depth{k}": int(np.sum(rr["map"]["inv_depth"] == g[f"map{k}"]["inv_depth"]))})
        print("mvstereo node", name, [(len(r["sel"]), len(r["matched"]), len(r["stamps"]), r["window"].tolist(), len(r["map"]))
                                      for r in res])
    # esvo_MVStereo in PURE_BLOCK_MATCHING mode (MVStereoMode 1, esvo_MVStereo.cpp:383-432): BM, vEMP2vDP, naive_propagation of
    # the window.  Propagated points carry p_cam through T_frame_obs (the reference's cam2World inverts a 4x4, the oracle uses
    # the closed form: last-bit differences), so inverse depth, variance and x of older frames' points agree to 1e-12 only:
    # row / col / age / residual are digested, the inverse depths stored per tick, variance and x for the last map
    for name in MVSTEREO_SCENARIOS:
        sc = S.Scenario(name)
        ticks, st = sc.inputs(), sc.stream()
        node = R.RefNode(sc.params, sc.rig, st.pose, mvstereo=True, extra={"MVStereoMode": 1})
        node.push_events(st.ev_left)
        sizes = []
        for k, tk in enumerate(ticks):
            node.push_observation(tk["t"], tk["tsL"], tk["tsR"])
            assert node.data_transferring()
            node.mapping_at_time()
            mp = node.get_map()
            out.update({f"mvs1_{name}_window{k}": np.array(node.window(), np.uint32), f"mvs1_{name}_map_n{k}": len(mp),
                        f"mvs1_{name}_map_sha{k}": fields_digest(mp, BM_ONLY_FIELDS), f"mvs1_{name}_inv_depth{k}": mp["inv_depth"]})
            sizes.append((node.window(), len(mp)))
        out[f"mvs1_{name}_last_x"] = mp["x"]
        out[f"mvs1_{name}_last_variance"] = mp["variance"]
        print("mvstereo mode 1", name, sizes)
    out.update(node_init())
    path = os.path.join(HERE, "ref_node.npz")
    np.savez_compressed(path, **out)
    print("ref_node.npz", os.path.getsize(path) // 1024, "KiB")


def make_node_big():
    """the node objects on the shipped-size scenarios (esvo_Mapping on dsec10k, esvo_MVStereo on upenn1k), regulariser off as in
    make_node: selection, window and a digest of every tick's fused + cleaned DepthMap -> ref_node_big.npz"""
    out = {}
    for name, mvs in (("dsec10k", False), ("upenn1k", True)):
        sc = S.Scenario(name)
        ticks, st = sc.inputs(), sc.stream()
        g = np.load(os.path.join(HERE, f"ref_{name}.npz"))
        res = run_node(sc, ticks, st, regularization=False, mvstereo=mvs)
        pre = ("mvs_" if mvs else "") + f"{name}_"
        out[pre + "n_ticks"] = len(ticks)
        for k, (tk, r) in enumerate(zip(ticks, res)):
            fr, ref = r["frame"], g[f"points{k}"]
            assert len(fr) == len(ref) and all(np.array_equal(fr[f], ref[f]) for f in NODE_MAP_FIELDS + ("pose_idx", "p_cam"))
            assert st.ev_left[r["matched"]].tobytes() == np.ascontiguousarray(tk["ev"]).tobytes()
            out.update({pre + f"obs_t{k}": r["obs_t"], pre + f"window{k}": r["window"], pre + f"map_n{k}": len(r["map"]),
                        pre + f"map_sha{k}": map_digest(r["map"])})
        print("node (big)", name, [(len(r["sel"]), len(r["matched"]), r["window"].tolist(), len(r["map"])) for r in res])
    path = os.path.join(HERE, "ref_node_big.npz")
    np.savez_compressed(path, **out)
    print("ref_node_big.npz", os.path.getsize(path) // 1024, "KiB")


if __name__ == "__main__":
    assert R.available(), "needs /root/reference (build container only)"
    if "--big" in sys.argv:   # only the shipped-size fixtures (round 4): ref_upenn1k.npz, ref_dsec10k.npz, ref_node_big.npz
        for n in S.BIG:
            make(n)
        make_node_big()
        sys.exit(0)
    if "--track-solve" in sys.argv:   # only the tracker-loop fixture (round 6): ref_track_solve.npz
        make_track_solve()
        sys.exit(0)
    names = [a for a in sys.argv[1:] if not a.startswith("-")] or list(S.SCENARIOS)
    for n in names:
        make(n)
    if "--only" in sys.argv:   # just the named scenarios (e.g. `--only upenn25 dsec10x4`)
        sys.exit(0)
    make_node_big()
    make_units()
    make_track()
    make_track_solve()
    make_sgm()
    make_ts()
    make_ts_jitter()
    make_ts_forward()
    make_bm_step()
    make_l2()
    make_node()
