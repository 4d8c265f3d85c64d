"""CPU tests of the oracle (no GPU): self-consistency of the restated third-party arithmetic and
known-answer tests of the reference's quirks.  The reference ships
no tests or golden vectors ("parity unpinned"), so these are what pins the oracle."""
import numpy as np
import pytest
import scipy.ndimage as ndi
import scipy.optimize as opt

from esvo_amd import calib, params, rostime, synth
from esvo_amd.abi import DEPTH_POINT_DTYPE, EVENT_DTYPE, make_events
from oracle import oracle as O


# ---- OpenCV image primitives (Appendix B.2) -------------------------------------------------------
def test_median3_matches_scipy():
    rng = np.random.default_rng(0)
    img = rng.integers(0, 256, (37, 53), dtype=np.uint8)
    assert np.array_equal(O.median3(img), ndi.median_filter(img, size=3, mode="nearest"))


def test_gaussian5_close_to_float_conv():
    rng = np.random.default_rng(1)
    img = rng.integers(0, 256, (40, 61), dtype=np.uint8)
    k = np.array([1, 4, 6, 4, 1], np.float64) / 16
    ref = ndi.convolve1d(ndi.convolve1d(img.astype(np.float64), k, axis=1, mode="mirror"), k, axis=0, mode="mirror")
    out = O.gaussian5(img).astype(np.float64)
    assert np.abs(out - ref).max() <= 0.5 + 1e-9  # one rounding at the end


def test_remap_identity_and_bilinear():
    rng = np.random.default_rng(2)
    h, w = 31, 45
    img = rng.integers(0, 256, (h, w), dtype=np.uint8)
    yy, xx = np.meshgrid(np.arange(h, dtype=np.float32), np.arange(w, dtype=np.float32), indexing="ij")
    assert np.array_equal(O.remap_bilinear(img, xx, yy), img)  # integer maps -> identity
    mx = (xx + rng.uniform(-1.5, 1.5, xx.shape)).astype(np.float32)
    my = (yy + rng.uniform(-1.5, 1.5, yy.shape)).astype(np.float32)
    out = O.remap_bilinear(img, mx, my).astype(np.float64)
    # float bilinear on the 1/32-quantised coordinates, zero outside
    sx, sy = np.rint(mx * 32).astype(int), np.rint(my * 32).astype(int)
    ix, iy, fx, fy = sx >> 5, sy >> 5, (sx & 31) / 32.0, (sy & 31) / 32.0
    pad = np.zeros((h + 4, w + 4))
    pad[2:-2, 2:-2] = img
    tap = lambda x, y: pad[np.clip(y + 2, 0, h + 3), np.clip(x + 2, 0, w + 3)] * ((x >= -2) & (x < w + 2) & (y >= -2) & (y < h + 2))
    ref = ((1 - fx) * (1 - fy) * tap(ix, iy) + fx * (1 - fy) * tap(ix + 1, iy) + (1 - fx) * fy * tap(ix, iy + 1)
           + fx * fy * tap(ix + 1, iy + 1))
    assert np.abs(out - ref).max() <= 0.5 + 1e-9


# ---- calibration maths (Appendix B.3) ---------------------------------------------------------------
@pytest.mark.parametrize("name", ["upenn", "dsec"])
def test_undistort_distort_roundtrip(name):
    rig = calib.dataset_rig(name)
    for cam, intr in ((rig.left, rig.intr_left), (rig.right, rig.intr_right)):
        lut = cam.rect_lut.astype(np.float64)
        ok = lut[..., 0] > -1e5
        u, v = calib.rect_to_raw(lut[..., 0], lut[..., 1], intr["K"], intr["D"], intr["R"], intr["P"], intr["model"])
        yy, xx = np.meshgrid(np.arange(cam.height), np.arange(cam.width), indexing="ij")
        err = np.hypot(u - xx, v - yy)[ok]
        assert np.percentile(err, 99) < 2e-2 and err.max() < 0.3  # 5 fixed-point iterations + float32 LUT
    assert abs(rig.baseline - {"upenn": 0.09988, "dsec": 0.59903}[name]) < 1e-4


def test_ideal_rig_is_identity():
    rig = calib.ideal_rig(64, 48, 100.0, 0.1)
    yy, xx = np.meshgrid(np.arange(48, dtype=np.float32), np.arange(64, dtype=np.float32), indexing="ij")
    assert np.array_equal(rig.left.map_x, xx) and np.array_equal(rig.left.map_y, yy)
    assert np.allclose(rig.left.rect_lut[..., 0], xx) and np.all(rig.left.rect_mask == 255)


# ---- Time Surface --------------------------------------------------------------------------------------
def test_ts_strict_before_and_queue_quirk():
    W, H = 8, 6
    ts = O.OracleTS(W, H, queue_len=20)
    t0 = 2_000_000_000
    ev = make_events([1, 1, 2], [1, 1, 2], [t0, t0 + 5_000_000, t0 + 10_000_000])
    ts.push(ev)
    img = ts.render(t0 + 5_000_000, median_k=0)  # strict ts < T: the event AT T is ignored -> older one
    assert img[1, 1] == int(np.rint(255 * np.exp(-0.005 / 0.030)))
    assert img[2, 2] == 0  # newer than T
    # Appendix A-2: > queue_len events newer than T hide the older ones (queue trimmed to 20)
    many = make_events([3] * 25, [3] * 25, t0 + 20_000_000 + np.arange(25) * 1000)
    ts.push(make_events([3], [3], [t0]))
    ts.push(many)
    assert ts.render(t0 + 15_000_000, median_k=0)[3, 3] == 0


def test_ts_polarity_rounding_half_even():
    ts = O.OracleTS(4, 4)
    img = ts.render(1_000_000_000, ignore_polarity=False, median_k=0)
    assert np.all(img == 128)  # 255*(0+1)/2 = 127.5 -> cvRound (half to even) -> 128


# ---- block matching ------------------------------------------------------------------------------------
def test_zncc_properties():
    rng = np.random.default_rng(3)
    l = rng.integers(0, 256, (7, 15)).astype(np.float64)
    assert abs(O.zncc_cost(l, l)) < 1e-7                      # identical patches -> cost 0 (up to the 1e-6 sigma guard)
    r = np.clip(0.5 * l + 20, 0, 255)
    assert O.zncc_cost(l, np.rint(r)) < 2e-3                  # affine brightness change: nearly invariant
    assert abs(O.zncc_cost(l, 255 - l) - 1.0) < 1e-6          # anti-correlated -> 1
    z = np.zeros_like(l)
    assert O.zncc_cost(l, z) == 0.5                           # flat patch: normalised to 0 -> cost exactly 0.5
    for _ in range(50):
        r = rng.integers(0, 256, (7, 15)).astype(np.float64)
        assert abs(O.zncc_cost(l, r) - O.zncc_cost(l, r, exact_int=True)) < 1e-13


@pytest.fixture(scope="module")
def upenn_case(upenn_rig, upenn_stream):
    p, _ = params.make_params(params.PRESETS["mvstereo_upenn"], upenn_rig)
    t = upenn_stream.t0_ns + int(0.1e9)
    ts = [O.OracleTS(upenn_rig.width, upenn_rig.height), O.OracleTS(upenn_rig.width, upenn_rig.height)]
    ts[0].push(upenn_stream.ev_left)
    ts[1].push(upenn_stream.ev_right)
    l = ts[0].render(t, map_x=upenn_rig.left.map_x, map_y=upenn_rig.left.map_y)
    r = ts[1].render(t, map_x=upenn_rig.right.map_x, map_y=upenn_rig.right.map_y)
    stamps, poses = rostime.pose_table(upenn_stream.pose, t, p.bm_half_slice_thickness)
    idx = O.select_events(upenn_stream.ev_left, t, p.bm_half_slice_thickness, p.process_event_num)
    return dict(p=p, t=t, l=l, r=r, stamps=stamps, poses=poses, ev=upenn_stream.ev_left[idx], idx=idx)


def _mapper(rig, stream, c, mode=(False, False)):
    m = O.OracleMapper(c["p"], rig)
    m.set_mode(*mode)
    m.set_observation(c["t"], c["l"], c["r"], stream.pose(c["t"]))
    m.set_poses(c["stamps"], c["poses"])
    return m


def test_event_selection_off_by_one(upenn_stream, upenn_case):
    """esvo_Mapping.cpp:562-574: newest first, starts AT lower_bound(t_end), stops before lower_bound(t_begin)."""
    ns, t = upenn_stream.ns_left, upenn_case["t"]
    idx = upenn_case["idx"]
    hi = np.searchsorted(ns, t, side="left")
    assert idx[0] == hi and np.all(np.diff(idx.astype(np.int64)) == -1)
    assert len(idx) == upenn_case["p"].process_event_num


def test_thread_stride_order(upenn_rig, upenn_stream, upenn_case):
    m = _mapper(upenn_rig, upenn_stream, upenn_case)
    mt = m.match(upenn_case["ev"])
    e = mt["event_idx"].astype(int)
    # EventBM.cpp:289-308: thread t handles i = t, t+4, ...; results concatenated by thread
    key = (e % 4) * 10**6 + e
    assert np.all(np.diff(key) > 0) and len(mt) > 100


def test_bm_exact_int_vs_literal(upenn_rig, upenn_stream, upenn_case):
    a = _mapper(upenn_rig, upenn_stream, upenn_case, (False, False)).match(upenn_case["ev"])
    b = _mapper(upenn_rig, upenn_stream, upenn_case, (True, True)).match(upenn_case["ev"])
    assert len(a) == len(b) and np.array_equal(a["disp"], b["disp"])
    assert np.abs(a["cost"] - b["cost"]).max() < 1e-12
    m = _mapper(upenn_rig, upenn_stream, upenn_case)
    for e in upenn_case["ev"][:40]:
        c0, c1 = m.match_costs(e, False), m.match_costs(e, True)
        if c0 is not None:
            assert np.nanmax(np.abs(c0 - c1)) < 1e-12


def test_bm_recovers_true_disparity(upenn_rig, upenn_stream, upenn_case):
    m = _mapper(upenn_rig, upenn_stream, upenn_case)
    mt = m.match(upenn_case["ev"])
    u, v, rho = upenn_stream.true_inv_depth_image(upenn_case["t"])
    ok = (u >= 0) & (u < upenn_rig.width) & (v >= 0) & (v < upenn_rig.height)
    gt = np.full((upenn_rig.height, upenn_rig.width), np.nan)
    gt[v[ok].astype(int), u[ok].astype(int)] = rho[ok]
    err = []
    for q in mt:
        x, y = int(q["x_left"][0]), int(q["x_left"][1])
        g = gt[max(y - 1, 0):y + 2, max(x - 1, 0):x + 2]
        if np.isfinite(g).any():
            err.append(np.nanmin(np.abs(g - q["inv_depth"])))
    assert len(err) > 50 and np.median(err) < 0.06  # one disparity step = 1/(f b) = 0.05 1/m


# ---- LM (Appendix B.1) vs scipy's MINPACK wrapper -----------------------------------------------------
def test_lm_matches_minpack(upenn_rig, upenn_stream, upenn_case):
    m = _mapper(upenn_rig, upenn_stream, upenn_case)
    mt = m.match(upenn_case["ev"])[:40]
    pts, info = m.refine(mt, cull=False, want_info=True)
    solved = info[:, 3] > 0
    assert solved.sum() >= 30
    # map solver outputs back to matches (stride order of the solver)
    order = [i for t in range(4) for i in range(t, len(mt), 4) if solved[i]]
    n_close = 0
    for out, i in zip(pts, order):
        f = lambda x: m.eval_residual(mt[i]["x_left"], mt[i]["pose_idx"], float(x[0]))[0]
        ref = opt.least_squares(f, [mt[i]["inv_depth"]], method="lm", xtol=1e-10, ftol=1e-10, diff_step=1.5e-8)
        # both are local minimisers of a piecewise-smooth cost: compare cost reached
        c_ref, c_orc = 2 * ref.cost, out["residual"]
        if abs(out["inv_depth"] - ref.x[0]) < 2e-3 * max(abs(ref.x[0]), 1e-3):
            n_close += 1
        assert c_orc <= c_ref * 1.05 + 1e-6 or abs(out["inv_depth"] - ref.x[0]) < 5e-2
    assert n_close >= 0.7 * len(order)


def test_lm_failure_fill_and_canonical_mode(upenn_rig, upenn_stream, upenn_case):
    m = _mapper(upenn_rig, upenn_stream, upenn_case)
    p = upenn_case["p"]
    f, ok = m.eval_residual([5.0, 5.0], 0, 0.5)  # patch outside the image -> failure fill (Appendix A-13)
    w = (p.td_nu + 1) / (p.td_nu + (255 / p.td_scale) ** 2)
    assert ok == 0 and np.allclose(f, np.sqrt(w) * 255)
    mt = m.match(upenn_case["ev"])
    a = m.refine(mt, cull=False)
    m2 = _mapper(upenn_rig, upenn_stream, upenn_case, (True, True))
    b = m2.refine(mt, cull=False)
    assert len(a) == len(b)
    assert np.abs(a["inv_depth"] / b["inv_depth"] - 1).max() < 1e-5  # summation order only
    assert m.counters()["max_scale_iters"] > 1000  # the uncapped t-scale loop really collapses


# ---- fusion known answers --------------------------------------------------------------------------------
def _pt(rig, p, x, y, rho, var=1e-4, res=100.0, age=0):
    d = np.zeros(1, DEPTH_POINT_DTYPE)
    P = rig.left.P.reshape(3, 4)
    z = 1 / rho
    d["row"], d["col"] = int(y), int(x)
    d["x"] = [x, y]
    d["inv_depth"], d["variance"], d["nu"] = rho, var, p.td_nu
    d["scale2"] = var * (p.td_nu - 2) / p.td_nu
    d["residual"], d["age"] = res, age
    d["p_cam"] = [(x - P[0, 2]) / P[0, 0] * z, (y - P[1, 2]) / P[1, 1] * z, z]
    return d


def test_fusion_kat_new_compatible_occluded_replace():
    rig = calib.ideal_rig(64, 48, 100.0, 0.1)
    p, _ = params.make_params(params.PRESETS["mvstereo_upenn"], rig)
    m = O.OracleMapper(p, rig)
    I = np.eye(4)
    l = np.zeros((48, 64), np.uint8)
    m.set_observation(1_000_000_000, l, l, I)
    m.push_frame(_pt(rig, p, 20.3, 10.6, 0.5), I.reshape(1, 16))
    assert m.fuse() == 0
    mp = m.get_map()
    assert len(mp) == 0  # age 0 < age_vis_threshold 1 -> cleaned (Appendix A-6)
    p2, _ = params.make_params(params.PRESETS["mvstereo_upenn"], rig, age_vis_threshold=0.0)
    m = O.OracleMapper(p2, rig)
    m.set_observation(1_000_000_000, l, l, I)
    m.push_frame(_pt(rig, p2, 20.3, 10.6, 0.5), I.reshape(1, 16))
    m.fuse()
    mp = m.get_map()
    assert [(int(r), int(c)) for r, c in zip(mp["row"], mp["col"])] == [(10, 20), (10, 21), (11, 20), (11, 21)]
    assert np.allclose(mp["x"][0], [20.5, 10.5]) and np.all(mp["age"] == 0)
    # a compatible second observation: t-fusion, age += 2 (Appendix A-6), residual = min
    m.push_frame(_pt(rig, p2, 20.3, 10.6, 0.5001, res=50.0), I.reshape(1, 16))
    assert m.fuse() == 4
    mp = m.get_map()
    assert np.all(mp["age"] == 2) and np.all(mp["residual"] == 50.0) and np.all(mp["nu"] == p2.td_nu + 1)
    # an incompatible, farther observation is occluded (skipped)
    m.push_frame(_pt(rig, p2, 20.3, 10.6, 0.1), I.reshape(1, 16))
    m.fuse()
    assert np.all(m.get_map()["age"] == 2)
    c0 = m.counters()["replace"]
    # an incompatible, NEARER, better observation replaces the element (row/col/x travel, Appendix A-7).
    # Frames fuse newest first, so it creates the cells and the older ones then fuse/skip.
    m.push_frame(_pt(rig, p2, 20.3, 10.6, 0.9, var=1e-8, res=1.0), I.reshape(1, 16))
    m.fuse()
    assert m.counters()["replace"] >= c0


def test_regularisation_border_quirk():
    """SmartGrid::getNeighbourhood mixes int and size_t loop bounds: pixels within `radius` of the top or left
    border get NO neighbours and are invalidated (rho = -1)."""
    rig = calib.ideal_rig(64, 48, 100.0, 0.1)
    p, _ = params.make_params(params.PRESETS["mvstereo_rpg"], rig, age_vis_threshold=0.0, stdvar_vis_threshold=1.0,
                              invdepth_min=0.01, invdepth_max=5.0)
    assert p.regularization and p.reg_radius == 5
    m = O.OracleMapper(p, rig)
    I = np.eye(4)
    l = np.zeros((48, 64), np.uint8)
    m.set_observation(1_000_000_000, l, l, I)
    pts = np.concatenate([_pt(rig, p, x + 0.2, y + 0.2, 0.5) for y in range(1, 40, 2) for x in range(1, 50, 2)])
    m.push_frame(pts, I.reshape(1, 16))
    m.fuse()
    mp = m.get_map()
    border = (mp["row"] < 5) | (mp["col"] < 5)
    inner = (mp["row"] >= 12) & (mp["col"] >= 12) & (mp["row"] < 30) & (mp["col"] < 40)
    assert border.any() and np.all(mp["inv_depth"][border] == -1.0)
    assert inner.any() and np.all(np.abs(mp["inv_depth"][inner] - 0.5) < 1e-9)


def test_denoise_mask(upenn_rig, upenn_stream, upenn_case):
    idx = upenn_case["idx"]
    out = O.denoise_events(upenn_stream.ev_left, idx, upenn_rig.width, upenn_rig.height, 500)
    assert 0 < len(out) <= 500 and set(out.tolist()) <= set(idx.tolist())


def test_abi_struct_sizes():
    from esvo_amd import abi
    import ctypes
    s = O.abi_sizes()
    assert s[0] == abi.EVENT_DTYPE.itemsize == 16
    assert s[1] == ctypes.sizeof(abi.CalibStruct) and s[2] == ctypes.sizeof(abi.ParamsStruct)
    assert s[3] == abi.MATCH_DTYPE.itemsize and s[4] == abi.DEPTH_POINT_DTYPE.itemsize
    assert s[5] == ctypes.sizeof(abi.StatsStruct) and s[6] == 0
