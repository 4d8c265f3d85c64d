"""Pins the CPU oracle to REFERENCE SOURCE.

tests/golden/ref_*.npz hold the outputs of oracle/_ref/libesvo_ref.so = the reference's own EventBM, DepthProblem,
DepthProblemSolver, DepthFusion, DepthRegularization, DepthPoint, SmartGrid and CameraSystem sources compiled unmodified
(tests/golden/make_ref_fixtures.py, oracle/ref_harness.cpp).  Here the oracle (literal mode) is checked against them
stage by stage -- each stage is fed the REFERENCE's output of the previous one, so a tolerance never cascades:

  block matching        matched set, order, disparity, pose: identical; ZNCC cost: bit-identical
  residual functor      DepthProblem::operator(): |df| <= 1e-9 grey levels (cam2World's inverse differs in form)
  LM refinement         same solved/culled set; inverse depth rel <= 1e-4 for all and <= 1e-6 for >= 99 % of every tick (measured
                        over 12 840 points: median 4e-10, 99.95 % below 1e-6, worst 1.3e-5; the LM stops at xtol = 1e-6);
                        variance rel <= 1e-3 for >= 98 % (measured 99.98 %; forward differences over the discontinuous
                        Student-t scale loop make single variances jump; the Eigen LM itself is third-party and is
                        restated twice, independently: n = 1 closed form in the oracle, general Householder/Givens
                        form in oracle/ref_shim)
  fusion/clean/regularise  every element of the DepthMap, list order, believed row/col, true cell, age, nu, inverse
                        depth, scale, variance, residual: bit-identical; p_cam rel <= 1e-12; fusion count identical
and end to end (the oracle's own chain): valid-set IoU >= 0.999, inverse-depth RMSE < 1e-6 on the intersection (measured: 1.0,
<= 1.7e-7; north_star asks for RMSE < 1e-4), also at the shipped tick sizes (ref_upenn1k.npz: 1000 events, ref_dsec10k.npz: 10 000).
Where the reference tree is present (build container) the library is also run live against the fixtures.
"""
import os

import numpy as np
import pytest

import scenarios as S
from oracle import oracle as O

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
NAMES = ["upenn", "dsec", "rpg", "hkust"]
# the same rigs at the SHIPPED tick sizes (PROCESS_EVENT_NUM 1000 / 10 000: mvstereo_upenn.yaml:18, mapping_dsec.yaml:18); their
# fixtures hold every tick's matches and points, digests of every DepthMap and the last map in full (tests/scenarios.py: BIG)
BIG_NAMES = list(S.BIG)
ALL_NAMES = NAMES + BIG_NAMES
# other patch sizes than the shipped 15 x 7 (the reference's code default 25 x 25; an even, non-square one): the general kernels
PATCH_NAMES = list(S.PATCH)
STAGE_NAMES = ALL_NAMES + PATCH_NAMES
EXACT = ["inv_depth", "scale2", "nu", "variance", "residual", "x"]
# End-to-end bars.  Measured: IoU 1.0 on every tick of every fixture, RMSE 0 ... 1.7e-7 on the small ones (the reference's Eigen
# LM driver is third-party; two restatements of it stop at xtol = 1e-6 a few 1e-7 apart), 2.1e-8 on upenn1k and 1.7e-6 on
# dsec10k (31 455 cells; the five LM outliers explained at BIG_POINT_BARS).  north_star's line is RMSE < 1e-4; the tests hold the
# chain to what it achieves, so that losing 0.1 % of the cells fails.
IOU_BAR, RMSE_BAR, RMSE_NORTH_STAR = 0.999, 1e-6, 1e-4
RMSE_BAR_BIG = 1e-5


def rmse_bar(name):
    return RMSE_BAR_BIG if name in S.BIG else RMSE_BAR
MAP_SHA_FIELDS = ("row", "col", "age", "inv_depth", "scale2", "nu", "variance", "residual", "x")


def map_sha(mp, cells=None):
    """the digest tests/golden/make_ref_fixtures.py stores for the maps of the shipped-size fixtures"""
    import hashlib
    h = hashlib.sha256()
    for f in MAP_SHA_FIELDS:
        h.update(np.ascontiguousarray(mp[f]).tobytes())
    if cells is not None:
        h.update(np.ascontiguousarray(cells, np.uint32).tobytes())
    return np.frombuffer(h.digest(), np.uint8)


def check_map(mp, g, k, cells=None, p_cam_rtol=1e-12):
    """every element of tick k's DepthMap equals the fixture's: field by field where the map is stored, else by digest"""
    if f"map{k}" in g.files:
        same_map(mp, g[f"map{k}"], cells, g[f"cells{k}"] if cells is not None else None, p_cam_rtol=p_cam_rtol)
        return
    assert len(mp) == int(g[f"map_n{k}"]), (k, len(mp), int(g[f"map_n{k}"]))
    assert np.array_equal(map_sha(mp), g[f"map_sha{k}"]), k
    if cells is not None:
        assert np.array_equal(map_sha(mp, cells), g[f"map_cells_sha{k}"]), k


def load_fixture(name):
    g = np.load(os.path.join(GOLDEN, f"ref_{name}.npz"))
    sc = S.Scenario(name)
    ticks = []
    for k in range(int(g["n_ticks"])):
        raw = (g[f"tsL{k}"], g[f"tsR{k}"])
        sm = (O.gaussian5(raw[0]), O.gaussian5(raw[1])) if int(g["smooth"]) else raw
        ticks.append(dict(t=int(g[f"t{k}"]), raw=raw, tsL=sm[0], tsR=sm[1], T=g[f"T{k}"], stamps=g[f"stamps{k}"],
                          poses=g[f"poses{k}"], ev=g[f"ev{k}"]))
    return g, sc, ticks


def same_map(mp, ref, cells=None, ref_cells=None, p_cam_rtol=1e-12):
    assert len(mp) == len(ref)
    for f in ("row", "col", "age"):
        assert np.array_equal(mp[f], ref[f]), f
    for f in EXACT:
        assert np.array_equal(mp[f], ref[f]), f
    assert np.allclose(mp["p_cam"], ref["p_cam"], rtol=p_cam_rtol, atol=1e-13)
    if cells is not None:
        assert np.array_equal(cells, ref_cells)


def check_matches(mt, ref, cost_exact=True, cost_atol=0.0):
    assert len(mt) == len(ref)
    for f in ("event_idx", "pose_idx", "disp", "x_left", "inv_depth"):
        assert np.array_equal(mt[f], ref[f]), f
    if cost_exact:
        assert np.array_equal(mt["cost"], ref["cost"])
    else:
        assert np.abs(mt["cost"] - ref["cost"]).max() <= cost_atol


# LM bars at the shipped tick sizes (5 800 solved points per tick on dsec10k).  The solved / culled SET is still identical and
# 99.85 % of 34 841 points agree to 1e-6, 99.986 % to 1e-4 -- but five points differ by 1e-4 ... 1.5e-2: the two functors agree to
# 1e-12 grey levels there, the OBJECTIVE is discontinuous (the uncapped Student-t scale loop of DepthProblem.cpp:96-124 jumps
# between its converged and its collapsed outcome: |F|^2 688 -> 5e-4 between two neighbouring inverse depths), and a last-bit
# difference in a forward-difference Jacobian decides on which side a trial step lands.  The reference's own result depends on
# its Eigen build in the same way; the bars are statistical by nature, not by sloppiness.
BIG_POINT_BARS = dict(rho_rtol=5e-2, rho_frac=0.995, mid_frac=0.999)


def check_points(pts, ref, rho_rtol=1e-4, rho_tight=1e-6, rho_frac=0.99, var_rtol=1e-3, var_frac=0.98, mid_frac=None):
    assert len(pts) == len(ref)
    if not len(ref):
        return
    for f in ("row", "col", "pose_idx", "age", "nu", "x"):
        assert np.array_equal(pts[f], ref[f]), f
    rel = np.abs(pts["inv_depth"] - ref["inv_depth"]) / ref["inv_depth"]
    assert rel.max() <= rho_rtol and (rel <= rho_tight).mean() >= rho_frac, (rel.max(), (rel <= rho_tight).mean())
    if mid_frac is not None:
        assert (rel <= 1e-4).mean() >= mid_frac, (rel <= 1e-4).mean()
        # outliers sit at jumps of the objective: compare the costs only where the solutions agree
        ok = rel <= 1e-4
        assert np.allclose(pts["residual"][ok], ref["residual"][ok], rtol=1e-3)
        rv = np.abs(pts["variance"] - ref["variance"]) / np.maximum(ref["variance"], 1e-300)
        assert (rv <= var_rtol).mean() >= var_frac, (rv <= var_rtol).mean()
        return
    rel = np.abs(pts["variance"] - ref["variance"]) / np.maximum(ref["variance"], 1e-300)
    assert (rel <= var_rtol).mean() >= var_frac, (rel <= var_rtol).mean()
    assert np.allclose(pts["residual"], ref["residual"], rtol=1e-3)


def map_stats(mp, ref, W):
    """valid-set IoU over believed cells and inverse-depth RMSE on the intersection"""
    ka = mp["row"].astype(np.int64) * W + mp["col"]
    kb = ref["row"].astype(np.int64) * W + ref["col"]
    da = dict(zip(ka.tolist(), mp["inv_depth"].tolist()))
    db = dict(zip(kb.tolist(), ref["inv_depth"].tolist()))
    both = [k for k in da if k in db and da[k] > -1e-6 and db[k] > -1e-6]
    union = len(set(da) | set(db))
    rmse = float(np.sqrt(np.mean([(da[k] - db[k]) ** 2 for k in both]))) if both else 0.0
    return (len(set(da) & set(db)) / union if union else 1.0), rmse


@pytest.mark.parametrize("name", STAGE_NAMES)
def test_oracle_stages_match_reference(name):
    g, sc, ticks = load_fixture(name)
    m = O.OracleMapper(sc.params, sc.rig)  # literal mode
    if name in BIG_NAMES:
        m.set_threads(os.cpu_count() or 1)
    for k, tk in enumerate(ticks):
        m.set_observation(tk["t"], tk["raw"][0], tk["raw"][1], tk["T"])
        m.set_poses(tk["stamps"], tk["poses"])
        check_matches(m.match(tk["ev"]), g[f"matches{k}"])
        check_points(m.refine(g[f"matches{k}"], cull=True), g[f"points{k}"], **(BIG_POINT_BARS if name in BIG_NAMES else {}))
        m.push_frame(g[f"points{k}"], tk["poses"])
        assert m.fuse() == int(g[f"nf{k}"])
        check_map(m.get_map(), g, k, m.get_map_cells())
    if name not in PATCH_NAMES:
        assert int(g["u_dangling"]) > 0 and m.counters()["replace_displaced"] > 0  # Appendix A-7 is exercised


@pytest.mark.parametrize("name", STAGE_NAMES)
def test_oracle_chain_matches_reference_end_to_end(name):
    g, sc, ticks = load_fixture(name)
    m = O.OracleMapper(sc.params, sc.rig)
    if name in BIG_NAMES:
        m.set_threads(os.cpu_count() or 1)
    res = S.run_stagewise(m, ticks, pre_smoothed=False)
    for k, r in enumerate(res):
        if f"map{k}" not in g.files:   # shipped-size fixtures keep the last map in full
            continue
        iou, rmse = map_stats(r["map"], g[f"map{k}"], sc.rig.width)
        assert iou >= IOU_BAR and rmse < rmse_bar(name) < RMSE_NORTH_STAR, (k, iou, rmse)


@pytest.mark.parametrize("name", NAMES)
def test_oracle_units_match_reference(name):
    g, sc, ticks = load_fixture(name)
    m = O.OracleMapper(sc.params, sc.rig)
    tk = ticks[0]
    m.set_observation(tk["t"], tk["raw"][0], tk["raw"][1], tk["T"])
    m.set_poses(tk["stamps"], tk["poses"])
    mt = g["matches0"]
    worst = 0.0
    for i, rho, f_ref in zip(g["u_res_pick"], g["u_res_rho"], g["u_res_fvec"]):
        f, _ = m.eval_residual(mt["x_left"][i], mt["pose_idx"][i], rho)
        worst = max(worst, float(np.abs(f - f_ref).max()))
    assert worst <= 1e-9, worst
    assert abs(m.baseline - float(g["u_baseline"])) <= 1e-15
    # cam2World / world2Cam through the oracle's camera: refine() on a zero-iteration problem is not exposed, so use
    # the DepthPoint initialisation of a fused map instead: p_cam of new cells == cam2World(cell centre, rho)
    # (DepthFusion.cpp:143-145) is covered by same_map(); here the closed form itself:
    P = sc.rig.left.P.reshape(3, 4)
    Kinv = np.linalg.inv(P[:, :3])
    for x, rho, p_ref in zip(g["u_c2w_x"], g["u_c2w_rho"], g["u_c2w_p"]):
        p = (1.0 / rho) * (Kinv @ np.array([x[0], x[1], 1.0])) - Kinv @ P[:, 3]
        assert np.allclose(p, p_ref, rtol=1e-12, atol=1e-13)


def test_student_t_and_zncc_units():
    g = np.load(os.path.join(GOLDEN, "ref_units.npz"))
    # DepthPoint::update_studentT through the oracle's fusion path is covered bit for bit by the map tests; the unit
    # vectors pin the closed form used by every implementation here
    for st, ob, new in zip(g["st_state"], g["st_obs"], g["st_new"]):
        d, s2, nu, var, age = st
        if d > -1e-6:
            nu_u = min(ob[3], nu)
            d_u = (ob[1] * d + s2 * ob[0]) / (s2 + ob[1])
            s2_u = (nu_u + (d - ob[0]) ** 2 / (s2 + ob[1])) / (nu_u + 1) * (s2 * ob[1]) / (s2 + ob[1])
            exp = np.array([d_u, s2_u, nu_u + 1, (nu_u + 1) / (nu_u + 1 - 2) * s2_u, age + 1])
        else:
            exp = np.array([ob[0], ob[1], ob[3], ob[2], 0.0])
        assert np.array_equal(exp, new)
    for l, r, c in zip(g["zncc_l"], g["zncc_r"], g["zncc_cost"]):
        lit = O.zncc_cost(l.astype(np.float64), r.astype(np.float64), exact_int=False)
        assert lit == c
        assert abs(O.zncc_cost(l.astype(np.float64), r.astype(np.float64), exact_int=True) - c) <= 1e-12


@pytest.mark.parametrize("name", STAGE_NAMES)
def test_live_reference_reproduces_fixture(name):
    from oracle import ref as R
    if not os.path.isdir(os.path.join(R.REFERENCE, "esvo_core", "src")):
        pytest.skip("reference tree not present (GPU box): the fixtures are the pin")
    g, sc, ticks = load_fixture(name)
    r = R.RefMapper(sc.params, sc.rig)
    res = S.run_stagewise(r, ticks, pre_smoothed=True)
    for k, x in enumerate(res):
        check_matches(x["matches"], g[f"matches{k}"])
        assert np.array_equal(x["points"]["inv_depth"], g[f"points{k}"]["inv_depth"])
        assert x["nf"] == int(g[f"nf{k}"])
        check_map(x["map"], g, k, p_cam_rtol=0.0)


def _tracker_on_fixture(make_tracker):
    """drive any tracker with the OracleTracker interface on the inputs of ref_track.npz; yields (key, value, reference)"""
    g = np.load(os.path.join(GOLDEN, "ref_track.npz"))
    n, order = int(g["n"]), g["order"]
    tr = make_tracker()
    tr.set_current(g["ts_left"], 5)
    tr.set_reference(g["xyz_world"][order][:n], g["T_world_ref"])   # the order the reference's shuffle produced
    for norm, huber in (("huber", True), ("l2", False)):
        for i in range(3):
            yield f"{norm}_f{i}", tr.residuals(g[f"{norm}_T{i}"], 100, 300, huber=huber, huber_threshold=50.0), g[f"{norm}_f{i}"]
        yield f"{norm}_f_tail", tr.residuals(g[f"{norm}_T_tail"], 600, 300, huber=huber, huber_threshold=50.0), g[f"{norm}_f_tail"]
        yield f"{norm}_J", tr.jacobian(g["R"], g["t"], 100, 300), g[f"{norm}_J"]


def test_tracker_functor_equals_reference_source():
    """RegProblemLM::operator() and df (RegProblemLM.cpp:91-269, compiled unmodified into oracle/_ref) against the oracle's
    restatement: residuals (Huber and l2, at x = 0 and at two motions, a batch cut short by the point count) and the
    Jacobian, bit for bit.  The negated blurred Time Surface and its Sobel derivatives are OpenCV products and are injected."""
    from esvo_amd import calib
    rig = calib.dataset_rig("upenn")
    for key, got, want in _tracker_on_fixture(lambda: O.OracleTracker(rig)):
        assert got.shape == want.shape, key
        assert np.array_equal(got, want), key
    g = np.load(os.path.join(GOLDEN, "ref_track.npz"))
    assert len(g["huber_f_tail"]) == 100 and np.abs(g["huber_J"]).max() > 1e3   # the cases are not degenerate


def test_live_reference_tracker_reproduces_fixture():
    from oracle import ref as R
    if not os.path.isdir(os.path.join(R.REFERENCE, "esvo_core", "src")):
        pytest.skip("reference tree not present (GPU box): the fixtures are the pin")
    sys_path = os.path.join(GOLDEN)
    import importlib.util
    spec = importlib.util.spec_from_file_location("make_ref_fixtures", os.path.join(sys_path, "make_ref_fixtures.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    g = np.load(os.path.join(GOLDEN, "ref_track.npz"))
    rig, L, pw, T_ref, T_left = m.track_inputs()
    assert np.array_equal(L, g["ts_left"]) and np.array_equal(pw, g["xyz_world"])
    ot = O.OracleTracker(rig)
    ot.set_current(L, 5)
    neg, du, dv = ot.images()
    rt = R.RefTracker(rig, huber=True, huber_threshold=50.0, max_points=700)
    order, Rm, tv = rt.set_problem(neg, du, dv, pw, T_ref, T_left, seed=3)
    assert np.array_equal(order, g["order"]) and np.array_equal(Rm, g["R"]) and np.array_equal(tv, g["t"])
    f, Tw = rt.residuals(100, 300, g["huber_x1"])
    assert np.array_equal(f, g["huber_f1"]) and np.array_equal(Tw, g["huber_T1"])
    assert np.array_equal(rt.jacobian(100, 300), g["huber_J"])


def test_sgm_bootstrap_behind_the_disparity_image_matches_reference_source():
    """esvo_Mapping.cpp:455-487 (glue) + the reference's DepthPoint::update / DepthFusion::naive_propagation, fed with the
    oracle's disparity image.  The node hands INTEGER pixel coordinates through cam2World -> T_frame_obs -> world2Cam and
    floors the result: the cell a point lands in depends on the last bit of that round trip (the reference inverts a 4x4
    matrix per cam2World call, CameraSystem.cpp:121-139; the oracle and the device use the closed form) -- upstream it
    depends on the Eigen build.  What is pinned: the same SGM points are accepted, and the propagated elements agree up
    to that one-cell ambiguity with inverse depths equal to 1e-12."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("make_ref_fixtures", os.path.join(GOLDEN, "make_ref_fixtures.py"))
    mk = importlib.util.module_from_spec(spec)
    sys_argv = list(__import__("sys").argv)
    spec.loader.exec_module(mk)
    g = np.load(os.path.join(GOLDEN, "ref_sgm.npz"))
    rig, p, t0, l0, r0, ev, xy, T = mk.sgm_inputs()
    m = O.OracleMapper(p, rig)
    m.set_observation(t0, l0, r0, T)
    n, _ = m.init_sgm(l0, r0, ev, min_points=100)
    assert n == int(g["n_points"]) > 500
    om, rmap = m.get_map(), g["ref_map"]
    _sgm_maps_agree(om["row"], om["col"], om["inv_depth"], rmap["row"], rmap["col"], rmap["inv_depth"])
    assert np.allclose(rmap["variance"], 1e-6, rtol=1e-9) and np.allclose(om["variance"], 1e-6, rtol=1e-9)   # boundVariance


def _sgm_maps_agree(o_row, o_col, o_rho, r_row, r_col, r_rho):
    """same propagated elements up to the one-cell ambiguity of the floor, inverse depths equal to 1e-12"""
    from collections import defaultdict
    assert abs(len(o_rho) - len(r_rho)) < 0.03 * len(r_rho)
    cells = defaultdict(list)
    for r, c, rho in zip(o_row.tolist(), o_col.tolist(), o_rho.tolist()):
        cells[(r, c)].append(rho)
    ok = 0
    for r, c, rho in zip(r_row.tolist(), r_col.tolist(), r_rho.tolist()):
        near = [v for dr in (-1, 0, 1) for dc in (-1, 0, 1) for v in cells.get((r + dr, c + dc), [])]
        ok += any(abs(v - rho) <= 1e-12 * rho for v in near)
    assert ok >= 0.95 * len(r_rho), (ok, len(r_rho))      # the rest lose a cell to a displaced neighbour


def test_initialization_glue_equals_the_reference_node():
    """The node object while its status is INITIALIZATION (ref_node.npz, init_*): dataTransferring's SGM event list is the
    oracle's, exactly; InitializationAtTime's map (identity pose, the oracle's disparity image in place of StereoSGBM)
    equals oracle/ref_harness.cpp's restatement of that glue exactly (asserted at generation) and the oracle's up to the
    floor's one-cell ambiguity."""
    import sys
    sys.path.insert(0, GOLDEN)
    import make_ref_fixtures as mk
    from esvo_amd import synth
    n = np.load(os.path.join(GOLDEN, "ref_node.npz"))
    rig, p, t0, l0, r0, ev, xy, T = mk.sgm_inputs()
    st = synth.make_stream(rig, 8000, 0.35, 0.16, 1.0, seed=20250419, speed=1.0)
    assert np.array_equal(O.select_events_sgm(st.ev_left, t0, p.bm_half_slice_thickness, p.process_event_num), n["init_sel"])
    m = O.OracleMapper(p, rig)
    m.set_observation(t0, l0, r0, np.eye(4))
    m.init_sgm(l0, r0, ev, min_points=100)
    om = m.get_map()
    _sgm_maps_agree(om["row"], om["col"], om["inv_depth"], n["init_row"], n["init_col"], n["init_inv_depth"])
    assert np.allclose(n["init_variance"], 1e-6, rtol=1e-9)


def _ts_cases():
    import importlib.util
    spec = importlib.util.spec_from_file_location("make_ref_fixtures", os.path.join(GOLDEN, "make_ref_fixtures.py"))
    mk = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mk)
    rig, st = mk.ts_inputs()
    for ql in (20, 3):
        done = 0
        for k in range(1, 9):
            tk = st.t0_ns + k * 12_000_000
            hi = int(np.searchsorted(st.ns_left, tk))
            yield ql, k, tk, st.ev_left[done:hi], rig
            done = hi


def test_time_surface_raster_equals_reference_source():
    """TimeSurface::eventsCallback / createTimeSurfaceAtTime and EventQueueMat (esvo_time_surface, compiled unmodified into
    oracle/_ref/libesvo_ref_ts.so) against the oracle's raster BEFORE the OpenCV stages (median filter, remap): the u8 image
    after convertTo's rounding, per-pixel queues of 20 and of 3 events, render stamps at and before the newest events."""
    g = np.load(os.path.join(GOLDEN, "ref_ts.npz"))
    ts, cur = None, None
    n = 0
    for ql, k, tk, chunk, rig in _ts_cases():
        if cur != ql:
            ts, cur = O.OracleTS(rig.width, rig.height, queue_len=ql), ql
        ts.push(chunk)
        for j, back in enumerate((0, 5_000_000, 11_000_000)):
            _, pre = ts.render(tk - back, decay_ms=30.0, ignore_polarity=True, median_k=0, want_prefilter=True)
            assert np.array_equal(pre, g[f"q{ql}_k{k}_b{j}"]), (ql, k, j)
            n += 1
    assert n == 48
    # the short queue really loses events that the long one still finds (the cases differ)
    assert not np.array_equal(g["q20_k8_b2"], g["q3_k8_b2"])


@pytest.mark.parametrize("name", ["upenn", "dsec"])
@pytest.mark.parametrize("step,updown", [(2, 0), (3, 0), (1, 1), (2, 1)])
def test_coarse_to_fine_block_matching_equals_reference_source(name, step, updown):
    """BM_step > 1 (EventBM.cpp:113-138,169-225) and BM_bUpDownConfiguration (:180-183,146-151: the search along y) in the
    oracle against the reference's EventBM: same matched events, order, disparities, virtual views and costs.  (The device
    side of both: tests/test_gpu_ref.py::test_gpu_coarse_to_fine_and_updown_block_matching.)"""
    import copy
    g = np.load(os.path.join(GOLDEN, "ref_bm_step.npz"))
    _, sc, ticks = load_fixture(name)
    tk = ticks[1]
    p = copy.copy(sc.params)
    p.bm_step = step
    p.bm_updown = updown
    m = O.OracleMapper(p, sc.rig)
    m.set_observation(tk["t"], tk["raw"][0], tk["raw"][1], tk["T"])
    m.set_poses(tk["stamps"], tk["poses"])
    ref = g[f"{name}_s{step}" + ("_ud" if updown else "")]
    check_matches(m.match(tk["ev"]), ref)
    p1 = copy.copy(sc.params)
    m1 = O.OracleMapper(p1, sc.rig)
    m1.set_observation(tk["t"], tk["raw"][0], tk["raw"][1], tk["T"])
    m1.set_poses(tk["stamps"], tk["poses"])
    assert 0 < len(ref) < len(m1.match(tk["ev"]))   # the coarse pass and its neighbour rule reject matches the dense search keeps


@pytest.mark.parametrize("name", ["upenn", "hkust"])
def test_gaussian_model_l2_equals_reference_source(name):
    """LSnorm: l2 in the oracle against the reference's classes configured with it (tests/golden/ref_l2.npz), stage by stage
    like the Student-t pin: LM result (plain residual, covariance from |f|^2 / (m - n)) to the LM tolerance; Gaussian
    DepthPoint::update, variance propagation, chiSquareTest fusion, clean and the inverse-variance regulariser: every map
    element bit-identical.  (The device rejects LSnorm != Tdist: no shipped configuration sets it.)"""
    import copy
    import sys
    sys.path.insert(0, GOLDEN)
    import make_ref_fixtures as mk
    from esvo_amd.abi import LSNORM_L2
    n = np.load(os.path.join(GOLDEN, "ref_l2.npz"))
    g, sc, ticks = load_fixture(name)
    p = copy.copy(sc.params)
    p.ls_norm = LSNORM_L2
    m = O.OracleMapper(p, sc.rig)
    for k, tk in enumerate(ticks):
        m.set_observation(tk["t"], tk["raw"][0], tk["raw"][1], tk["T"])
        m.set_poses(tk["stamps"], tk["poses"])
        ref_pts = n[f"{name}_points{k}"].copy()
        pts = m.refine(g[f"matches{k}"], cull=True)
        for f in ("nu", "scale2"):   # never set on the Gaussian path: the reference's DepthPoint() leaves them uninitialised (A-8)
            ref_pts[f] = pts[f] if len(pts) == len(ref_pts) else 0
        check_points(pts, ref_pts, rho_rtol=1e-5)
        m.push_frame(ref_pts, tk["poses"])
        assert m.fuse() == int(n[f"{name}_nf{k}"])
        mp = m.get_map()
        assert len(mp) == int(n[f"{name}_map_n{k}"])
        if k == len(ticks) - 1:
            for f in mk.L2_MAP_FIELDS:
                assert np.array_equal(mp[f], n[f"{name}_last_{f}"]), f
        assert np.array_equal(mk.l2_digest(mp), n[f"{name}_map_sha{k}"]), k
    # the Gaussian model really is another one: its maps differ from the Student-t fixture's
    assert int(n[f"{name}_map_n{len(ticks) - 1}"]) != len(g[f"map{len(ticks) - 1}"])


def test_forward_time_surface_equals_reference_source():
    """createTimeSurfaceAtTime in FORWARD mode (TimeSurface.cpp:85-116, the node class compiled unmodified): the oracle's
    splat -- raster order of the sources, clamp after every add -- gives the same f64 image bit for bit (sha-256) and the
    same u8 image, with and without polarity, at and before the newest stamps."""
    import hashlib
    import sys
    sys.path.insert(0, GOLDEN)
    import make_ref_fixtures as mk
    g = np.load(os.path.join(GOLDEN, "ref_ts_forward.npz"))
    rig, st = mk.ts_inputs()
    lut = np.array(rig.left.rect_lut, np.float32).reshape(-1, 2)
    n = 0
    for pol in (0, 1):
        ts = O.OracleTS(rig.width, rig.height, queue_len=20)
        done = 0
        for k in range(1, 5):
            tk = st.t0_ns + k * 12_000_000
            hi = int(np.searchsorted(st.ns_left, tk))
            ts.push(st.ev_left[done:hi])
            done = hi
            for j, back in enumerate((0, 5_000_000)):
                u8, f64 = ts.render_forward(tk - back, lut, ignore_polarity=not pol, median_k=0, want_f64=True)
                assert np.array_equal(u8, g[f"p{pol}_k{k}_b{j}"]), (pol, k, j)
                sha = np.frombuffer(hashlib.sha256(np.ascontiguousarray(f64).tobytes()).digest(), np.uint8)
                assert np.array_equal(sha, g[f"p{pol}_k{k}_b{j}_sha"]), (pol, k, j)
                n += 1
    assert n == 16 and int(g["p0_k4_b0"].max()) == 255 and not np.array_equal(g["p0_k4_b0"], g["p1_k4_b0"])


def test_live_reference_time_surface_reproduces_fixture():
    from oracle import ref as R
    if not os.path.isdir(os.path.join(R.REFERENCE, "esvo_core", "src")):
        pytest.skip("reference tree not present (GPU box): the fixtures are the pin")
    g = np.load(os.path.join(GOLDEN, "ref_ts.npz"))
    ts, cur = None, None
    for ql, k, tk, chunk, rig in _ts_cases():
        if cur != ql:
            ts, cur = R.RefTS(rig.width, rig.height, 30.0, True, ql), ql
        ts.push(chunk)
        if k in (3, 8):
            img = ts.render(tk - 5_000_000)
            assert np.array_equal(np.clip(np.rint(img), 0, 255).astype(np.uint8), g[f"q{ql}_k{k}_b1"])
            assert float(img.sum()) == float(g[f"q{ql}_k{k}_b1_sum"])


def _jitter_cases():
    import importlib.util
    spec = importlib.util.spec_from_file_location("make_ref_fixtures", os.path.join(GOLDEN, "make_ref_fixtures.py"))
    mk = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mk)
    return mk.ts_jitter_cases()


def jitter_replay(push, render, g, ql, ev, chunks, renders):
    """deliveries and renders in the fixture's order; yields (render index, image)"""
    ri = 0
    n_r = int(g[f"q{ql}_renders"])
    for a, b in chunks:
        push(ev[a:b])
        while ri < n_r and int(g[f"q{ql}_r{ri}_after"]) == b:
            yield ri, render(renders[ri])
            ri += 1
    assert ri == n_r


@pytest.mark.parametrize("ql", [20, 3])
def test_out_of_order_deliveries_equal_the_reference_time_surface_class(ql):
    """tests/golden/ref_ts_jitter.npz: the reference's TimeSurface class fed a stream whose 250 us bundles arrive up to +-200 us
    off their time (8 % of the events arrive late).  eventsCallback insertion-sorts the arriving event and then inserts
    events_.back() -- the NEWEST event -- into the per-pixel queues (TimeSurface.cpp:412-422, SURVEY A-1): a late event never
    reaches the surface, the newest one is queued again.  The oracle restates that literally."""
    g = np.load(os.path.join(GOLDEN, "ref_ts_jitter.npz"))
    rig, st, ev, chunks, renders = _jitter_cases()
    assert int(g["n_late"]) > 0.05 * len(ev)
    ts = O.OracleTS(rig.width, rig.height, queue_len=ql)
    n = 0
    for ri, pre in jitter_replay(ts.push, lambda t: ts.render(t, decay_ms=30.0, ignore_polarity=True, median_k=0, want_prefilter=True)[1],
                                 g, ql, ev, chunks, renders):
        assert np.array_equal(pre, g[f"q{ql}_r{ri}"]), (ql, ri, int(np.count_nonzero(pre != g[f"q{ql}_r{ri}"])))
        n += 1
    assert n == 7
    # ... and it matters: the same events delivered in time order give another surface
    ts2 = O.OracleTS(rig.width, rig.height, queue_len=ql)
    ts2.push(st.ev_left)
    _, pre2 = ts2.render(renders[-1], decay_ms=30.0, ignore_polarity=True, median_k=0, want_prefilter=True)
    assert not np.array_equal(pre2, g[f"q{ql}_r6"])


def test_live_reference_time_surface_reproduces_the_jitter_fixture():
    from oracle import ref as R
    if not os.path.isdir(os.path.join(R.REFERENCE, "esvo_core", "src")):
        pytest.skip("reference tree not present (GPU box): the fixtures are the pin")
    g = np.load(os.path.join(GOLDEN, "ref_ts_jitter.npz"))
    rig, st, ev, chunks, renders = _jitter_cases()
    ts = R.RefTS(rig.width, rig.height, 30.0, True, 20)
    for ri, img in jitter_replay(ts.push, ts.render, g, 20, ev, chunks, renders):
        assert np.array_equal(np.clip(np.rint(img), 0, 255).astype(np.uint8), g[f"q20_r{ri}"])
        assert float(img.sum()) == float(g[f"q20_r{ri}_sum"])


# ---- the mapper NODE (esvo_core/src/esvo_Mapping.cpp compiled unmodified: oracle/ref_harness_node.cpp) ----
NODE_NAMES = ["dsec", "hkust"]   # the scenarios whose preset is the Mapping node's
NODE_MAP_FIELDS = ("row", "col", "age", "inv_depth", "scale2", "nu", "variance", "residual", "x")


def _map_digest(mp):
    import hashlib
    h = hashlib.sha256()
    for f in NODE_MAP_FIELDS:
        h.update(np.ascontiguousarray(mp[f]).tobytes())
    return np.frombuffer(h.digest(), np.uint8)


@pytest.mark.parametrize("name", NODE_NAMES)
def test_data_transfer_and_tick_glue_equal_the_reference_node(name):
    """ref_node.npz = the reference's esvo_Mapping object fed through its own callbacks.  Pinned exactly:
      dataTransferring   the observation it picks, the events it selects (indices into the left queue, order), the stamps
                         of the virtual-view table (getPoseAt at 50 us steps over the slice)
      MappingAtTime      the events handed to the block matcher (denoising mask + extraction on hkust), the window
                         policy (frames kept, their sizes), publishPointCloud's clouds, and -- fed the frame the node's own
                         BM + LM kept, which equals
                         ref_<name>.npz's frame (asserted when the fixture was made) -- every element of the fused and
                         cleaned DepthMap.
    Recorded with the regulariser off (the node regularises over erased elements; with it on, the
    node's map differs from ref_<name>.npz in inverse depths only -- also asserted at generation)."""
    import copy
    from esvo_amd import rostime
    n = np.load(os.path.join(GOLDEN, "ref_node.npz"))
    g, sc, ticks = load_fixture(name)
    st = sc.stream()
    p = copy.copy(sc.params)
    p.regularization = 0
    m = O.OracleMapper(p, sc.rig)
    assert int(n[f"{name}_n_ticks"]) == len(ticks)
    for k, tk in enumerate(ticks):
        pre = f"{name}_"
        assert int(n[pre + f"obs_t{k}"]) == tk["t"]
        sel = O.select_events(st.ev_left, tk["t"], p.bm_half_slice_thickness, p.process_event_num)
        assert np.array_equal(sel, n[pre + f"sel{k}"])
        fed = O.denoise_events(st.ev_left, sel, sc.rig.width, sc.rig.height, p.process_event_num) if sc.denoise else sel
        assert np.array_equal(fed, n[pre + f"matched{k}"])
        assert st.ev_left[fed].tobytes() == np.ascontiguousarray(tk["ev"]).tobytes()
        stamps, _ = rostime.pose_table(st.pose, tk["t"], p.bm_half_slice_thickness)
        assert np.array_equal(np.asarray(stamps, np.uint64), n[pre + f"stamps{k}"])
        m.set_observation(tk["t"], tk["raw"][0], tk["raw"][1], tk["T"])
        m.set_poses(tk["stamps"], tk["poses"])
        m.push_frame(g[f"points{k}"], tk["poses"])
        m.fuse()
        c = m.counters()
        win = n[pre + f"window{k}"]
        assert c["window_frames"] == len(win) and c["window_points"] == int(win.sum())
        mp = m.get_map()
        assert len(mp) == int(n[pre + f"map_n{k}"])
        if k == len(ticks) - 1:  # the last map is stored in full, so that a mismatch names the field
            for f in NODE_MAP_FIELDS:
                assert np.array_equal(mp[f], n[pre + f"last_{f}"]), f
        assert np.array_equal(_map_digest(mp), n[pre + f"map_sha{k}"]), k
        # with the regulariser on, the node agreed with ref_<name>.npz on most inverse depths and on everything else
        assert int(n[pre + f"reg_same_inv_depth{k}"]) >= 0.9 * len(g[f"map{k}"])
        # publishPointCloud (:909-934): the tracker's cloud and the near cloud, float32 world coordinates, list order
        import hashlib
        import sys
        sys.path.insert(0, GOLDEN)
        from make_ref_fixtures import NODE_VISUALIZE_RANGE
        pc, near = m.get_pointcloud(), m.get_pointcloud_near(NODE_VISUALIZE_RANGE[name])
        assert len(pc) == int(n[pre + f"pc_n{k}"]) and len(near) == int(n[pre + f"pc_near_n{k}"])
        for cloud, key in ((pc, "pc_sha"), (near, "pc_near_sha")):
            sha = np.frombuffer(hashlib.sha256(np.ascontiguousarray(cloud, np.float32).tobytes()).digest(), np.uint8)
            assert np.array_equal(sha, n[pre + f"{key}{k}"]), key


@pytest.mark.parametrize("name", ["upenn", "rpg"])
def test_data_transfer_and_tick_glue_equal_the_reference_mvstereo_node(name):
    """the same for esvo_MVStereo.cpp (BM_PLUS_ESTIMATION), whose presets upenn and rpg follow: it selects up to 10 000 events
    of the slice and cuts to PROCESS_EVENT_NUM in MappingAtTime (:627-647, 383-405), and cleans every tick (:496-497)"""
    import copy
    from esvo_amd import rostime
    n = np.load(os.path.join(GOLDEN, "ref_node.npz"))
    g, sc, ticks = load_fixture(name)
    st = sc.stream()
    p = copy.copy(sc.params)
    p.regularization = 0
    assert not p.clean_requires_full_window
    m = O.OracleMapper(p, sc.rig)
    pre = f"mvs_{name}_"
    assert int(n[pre + "n_ticks"]) == len(ticks)
    for k, tk in enumerate(ticks):
        assert int(n[pre + f"obs_t{k}"]) == tk["t"]
        sel = O.select_events(st.ev_left, tk["t"], p.bm_half_slice_thickness, 10000)
        assert np.array_equal(sel, n[pre + f"sel{k}"])
        fed = (O.denoise_events(st.ev_left, sel, sc.rig.width, sc.rig.height, p.process_event_num) if sc.denoise
               else sel[:p.process_event_num])
        assert np.array_equal(fed, n[pre + f"matched{k}"])
        assert st.ev_left[fed].tobytes() == np.ascontiguousarray(tk["ev"]).tobytes()
        stamps, _ = rostime.pose_table(st.pose, tk["t"], p.bm_half_slice_thickness)
        assert np.array_equal(np.asarray(stamps, np.uint64), n[pre + f"stamps{k}"])
        m.set_observation(tk["t"], tk["raw"][0], tk["raw"][1], tk["T"])
        m.set_poses(tk["stamps"], tk["poses"])
        m.push_frame(g[f"points{k}"], tk["poses"])
        m.fuse()
        c, win = m.counters(), n[pre + f"window{k}"]
        assert c["window_frames"] == len(win) and c["window_points"] == int(win.sum())
        mp = m.get_map()
        assert len(mp) == int(n[pre + f"map_n{k}"])
        assert np.array_equal(_map_digest(mp), n[pre + f"map_sha{k}"]), k


@pytest.mark.parametrize("name", ["upenn", "rpg"])
def test_block_matching_only_mode_equals_the_reference_mvstereo_node(name):
    """esvo_MVStereo's PURE_BLOCK_MATCHING mode (MVStereoMode 1, esvo_MVStereo.cpp:383-432, vEMP2vDP :1072-1094) in the oracle
    against the node object: window, and every element of the naively propagated map (row / col / age / residual exact;
    inverse depth, variance, x to 1e-12: propagated points carry p_cam, where the reference inverts a 4x4 per cam2World call)."""
    import sys
    sys.path.insert(0, GOLDEN)
    import make_ref_fixtures as mk
    n = np.load(os.path.join(GOLDEN, "ref_node.npz"))
    g, sc, ticks = load_fixture(name)
    m = O.OracleMapper(sc.params, sc.rig)
    for k, tk in enumerate(ticks):
        m.set_observation(tk["t"], tk["raw"][0], tk["raw"][1], tk["T"])
        m.set_poses(tk["stamps"], tk["poses"])
        assert m.tick_bm_only(tk["ev"]) == len(g[f"matches{k}"]) == int(n[f"mvs1_{name}_window{k}"][-1])
        assert m.counters()["window_frames"] == len(n[f"mvs1_{name}_window{k}"])
        mp = m.get_map()
        assert len(mp) == int(n[f"mvs1_{name}_map_n{k}"])
        assert np.array_equal(mk.fields_digest(mp, mk.BM_ONLY_FIELDS), n[f"mvs1_{name}_map_sha{k}"]), k
        assert np.allclose(mp["inv_depth"], n[f"mvs1_{name}_inv_depth{k}"], rtol=1e-12, atol=0)
    assert np.allclose(mp["x"], n[f"mvs1_{name}_last_x"], rtol=1e-12, atol=1e-12)
    assert np.allclose(mp["variance"], n[f"mvs1_{name}_last_variance"], rtol=1e-12, atol=0)


def test_const_points_window_policy_equals_the_reference_node():
    """the node's FUSION_STRATEGY = CONST_POINTS branch (esvo_Mapping.cpp:341-353) on the dsec frames: frames kept, their
    sizes, and every element of the fused + cleaned map"""
    import copy
    import sys
    sys.path.insert(0, GOLDEN)
    import make_ref_fixtures as mk
    n = np.load(os.path.join(GOLDEN, "ref_node.npz"))
    g, sc, ticks = load_fixture("dsec")
    p = copy.copy(sc.params)
    p.regularization = 0
    for k, v in mk.NODE_CONST_POINTS.items():
        setattr(p, k, v)
    m = O.OracleMapper(p, sc.rig)
    popped = False
    for k, tk in enumerate(ticks):
        m.set_observation(tk["t"], tk["raw"][0], tk["raw"][1], tk["T"])
        m.set_poses(tk["stamps"], tk["poses"])
        m.push_frame(g[f"points{k}"], tk["poses"])
        m.fuse()
        c, win = m.counters(), n[f"dsec_cp_window{k}"]
        assert c["window_frames"] == len(win) and c["window_points"] == int(win.sum())
        popped |= len(win) < k + 1
        mp = m.get_map()
        assert len(mp) == int(n[f"dsec_cp_map_n{k}"])
        assert np.array_equal(_map_digest(mp), n[f"dsec_cp_map_sha{k}"]), k
    assert popped


@pytest.mark.parametrize("name,mvstereo", [("dsec", False), ("hkust", False), ("upenn", True), ("rpg", True)])
def test_live_reference_node_reproduces_fixture(name, mvstereo):
    from oracle import ref as R
    if not os.path.isdir(os.path.join(R.REFERENCE, "esvo_core", "src")):
        pytest.skip("reference tree not present (GPU box): the fixtures are the pin")
    import sys
    sys.path.insert(0, GOLDEN)
    import make_ref_fixtures as M
    n = np.load(os.path.join(GOLDEN, "ref_node.npz"))
    g, sc, ticks = load_fixture(name)
    res = M.run_node(sc, ticks, sc.stream(), regularization=False, mvstereo=mvstereo)
    for k, r in enumerate(res):
        pre = f"mvs_{name}_" if mvstereo else f"{name}_"
        assert r["obs_t"] == int(n[pre + f"obs_t{k}"])
        for f in ("sel", "matched", "stamps", "window"):
            assert np.array_equal(r[f], n[pre + f"{f}{k}"]), f
        assert np.array_equal(M.map_digest(r["map"]), n[pre + f"map_sha{k}"])
        assert all(np.array_equal(r["frame"][f], g[f"points{k}"][f]) for f in NODE_MAP_FIELDS + ("pose_idx", "p_cam"))


@pytest.mark.parametrize("name,mvstereo,n_ticks,override", [("hkust", False, 16, {}), ("dsec", False, 12, {}),
                                                             ("upenn", True, 14, dict(max_fusion_points=800))])
def test_live_reference_node_long_run_window_churn(name, mvstereo, n_ticks, override):
    """Build container only: the reference node objects over a longer run than the fixtures hold -- the window fills and
    rolls several times, ages grow, every tick cleans.  The oracle's fusion is fed the node's own newest frame each tick
    (so the LM's statistical tolerance stays out) and must reproduce every element of the node's map, tick after tick;
    its selection / denoising / pose stamps must equal the node's."""
    from oracle import ref as R
    if not os.path.isdir(os.path.join(R.REFERENCE, "esvo_core", "src")):
        pytest.skip("reference tree not present (GPU box): the fixtures are the pin")
    import copy
    from esvo_amd import rostime
    sc = S.Scenario(name, n_ticks=n_ticks)
    ticks, st = sc.inputs(), sc.stream()
    p = copy.copy(sc.params)
    p.regularization = 0
    for k, v in override.items():   # upenn: CONST_POINTS with a point budget that makes the window roll in 14 ticks
        setattr(p, k, v)
    node = R.RefNode(p, sc.rig, st.pose, mvstereo=mvstereo)
    node.push_events(st.ev_left)
    m = O.OracleMapper(p, sc.rig)
    sizes = []
    for tk in ticks:
        node.push_observation(tk["t"], tk["tsL"], tk["tsR"])
        assert node.data_transferring() and node.obs_time() == tk["t"]
        sel = O.select_events(st.ev_left, tk["t"], p.bm_half_slice_thickness, 10000 if mvstereo else p.process_event_num)
        assert np.array_equal(node.selected_events(), sel)
        stamps, poses = node.pose_table()
        assert np.array_equal(stamps, np.asarray(rostime.pose_table(st.pose, tk["t"], p.bm_half_slice_thickness)[0], np.uint64))
        node.mapping_at_time()
        fed = (O.denoise_events(st.ev_left, sel, sc.rig.width, sc.rig.height, p.process_event_num) if sc.denoise
               else sel[:p.process_event_num])
        assert np.array_equal(node.matched_events(), fed)
        m.set_observation(tk["t"], tk["raw"][0], tk["raw"][1], tk["T"])
        m.set_poses(stamps, poses)
        m.push_frame(node.newest_frame(), poses)
        m.fuse()
        win = node.window()
        c = m.counters()
        assert c["window_frames"] == len(win) and c["window_points"] == sum(win)
        a, b = m.get_map(), node.get_map()
        assert len(a) == len(b)
        for f in NODE_MAP_FIELDS:
            assert np.array_equal(a[f], b[f]), (f, len(sizes))
        sizes.append(len(win))
    assert max(sizes) > 1 and any(b <= a for a, b in zip(sizes[2:], sizes[3:]))   # the window rolled


@pytest.mark.parametrize("name,n_ticks", [("dsec", 8), ("hkust", 8)])
def test_reference_node_at_the_references_own_O3_maps_the_same_bits(name, n_ticks):
    """Build container only.  bench.py times oracle/_ref/libesvo_ref_node_O3.so + libesvo_ref_ts_O3.so -- the node objects at the
    optimisation level the reference's own CMakeLists set (-O3, nothing else; esvo_core/CMakeLists.txt:7) -- while the oracle
    is pinned to the -O2 -ffp-contract=off builds.  The two must agree bit for bit (frames, maps, Time Surfaces), so that the
    timed baseline is the pinned code."""
    from oracle import ref as R
    if not os.path.isdir(os.path.join(R.REFERENCE, "esvo_core", "src")):
        pytest.skip("reference tree not present (GPU box): the fixtures are the pin")
    import copy
    sc = S.Scenario(name, n_ticks=n_ticks)
    ticks, st = sc.inputs(), sc.stream()
    p = copy.copy(sc.params)
    p.regularization = 0   # (with it on the node object reads freed list elements, SURVEY A-7: not a function of the build alone)
    nodes = [R.RefNode(p, sc.rig, st.pose, o3=o3) for o3 in (False, True)]
    for nd in nodes:
        nd.push_events(st.ev_left)
    for tk in ticks:
        out = []
        for nd in nodes:
            nd.push_observation(tk["t"], tk["tsL"], tk["tsR"])
            assert nd.data_transferring()
            nd.mapping_at_time()
            out.append((nd.newest_frame(), nd.get_map()))
        for a, b in zip(out[0], out[1]):
            assert len(a) == len(b) and len(a) > 0
            for f in NODE_MAP_FIELDS:
                assert np.array_equal(a[f], b[f]), f
    ts = [R.RefTS(sc.rig.width, sc.rig.height, o3=o3) for o3 in (False, True)]
    for t_ in ts:
        t_.push(st.ev_left)
    for tk in ticks[:3]:
        a, b = ts[0].render(tk["t"]), ts[1].render(tk["t"])
        assert np.array_equal(a, b, equal_nan=True) and np.isfinite(a).any()


def test_hip_binding_compiles_against_the_reference_node_classes():
    """Build container only: include/esvo_hip_mapping_node.hpp (the reference-side binding of INTEGRATION.md) instantiated
    with the reference's real esvo_Mapping and esvo_MVStereo classes and linked with libesvo_hip.so; the GPU side of it is
    tests/test_gpu_node_dropin.py."""
    import ctypes
    from oracle import ref as R
    if not os.path.isdir(os.path.join(R.REFERENCE, "esvo_core", "src")):
        pytest.skip("reference tree not present (GPU box)")
    from esvo_amd import lib
    lib.build()
    R.build()
    for name in ("libesvo_ref_node_hip.so", "libesvo_ref_mvstereo_hip.so"):
        so = ctypes.CDLL(os.path.join(os.path.dirname(R._LIB), name))
        for sym in ("ref_node_hip_attach", "ref_node_hip_mapping_at_time", "ref_node_hip_newest_frame", "ref_node_hip_get_map"):
            getattr(so, sym)
