"""Randomised SGM bootstrap (esvo_Mapping::InitializationAtTime, esvo_Mapping.cpp:433-492) against the oracle's
restatement of cv::StereoSGBM + the reference's glue behind the disparity image: a random rig, an image pair that is a rendered
Time-Surface pair of a seeded stream, or noise against its shifted / perturbed copy, or flat / saturated images; the SGM events selected
as the reference does.  Compared bit for bit: the whole fixed-point disparity image, the number of points, the DepthPoint frame and the
DepthMap the bootstrap leaves behind.  (The SGBM restatement itself is unpinned -- OpenCV is absent; this holds the HIP kernels to it.)
usage: python tools/fuzz_sgm.py [cases] [first seed]      (GPU; exits 1 on any difference)"""
import os
import sys
import time

ROOT = os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402

from esvo_amd import calib, lib, params, synth  # noqa: E402
from oracle import oracle  # noqa: E402

FIELDS = ("row", "col", "age", "inv_depth", "variance", "scale2", "nu", "residual", "x", "p_cam")


def run_case(seed):
    rng = np.random.default_rng(seed)
    pick = lambda xs: xs[int(rng.integers(len(xs)))]  # noqa: E731
    rig_name = pick(["upenn", "rpg", "hkust", "dsec", "upenn"])
    rig = calib.dataset_rig(rig_name)
    preset = {"dsec": "mapping_dsec", "rpg": "mapping_rpg", "hkust": "mapping_hkust"}.get(rig_name, "mapping_upenn")
    p, _ = params.make_params(params.PRESETS[preset], rig, throughput_events=int(pick([200, 1000, 5000])))
    W, H = rig.width, rig.height
    st = synth.make_stream(rig, pick([3000, 10000]), 0.09, 0.2, 1.0, seed=int(rng.integers(1 << 30)), speed=1.0)
    t = st.t0_ns + int(0.08e9)
    kind = pick(["ts", "ts", "ts", "noise_shift", "noise_pair", "flat", "saturated"])
    if kind == "ts":
        ts = [oracle.OracleTS(W, H), oracle.OracleTS(W, H)]
        ts[0].push(st.ev_left)
        ts[1].push(st.ev_right)
        l = ts[0].render(t, map_x=rig.left.map_x, map_y=rig.left.map_y)
        r = ts[1].render(t, map_x=rig.right.map_x, map_y=rig.right.map_y)
    elif kind == "noise_shift":
        l = rng.integers(0, 256, (H, W)).astype(np.uint8)
        r = np.roll(l, -int(rng.integers(0, 40)), axis=1)
    elif kind == "noise_pair":
        l = rng.integers(0, 256, (H, W)).astype(np.uint8)
        r = np.clip(np.roll(l, -int(rng.integers(0, 20)), axis=1).astype(np.int32) + rng.integers(-30, 31, (H, W)), 0, 255).astype(np.uint8)
    elif kind == "flat":
        l = np.full((H, W), int(rng.integers(0, 256)), np.uint8)
        r = np.full((H, W), int(rng.integers(0, 256)), np.uint8)
    else:
        l = np.where(rng.random((H, W)) < 0.5, 0, 255).astype(np.uint8)
        r = np.where(rng.random((H, W)) < 0.5, 0, 255).astype(np.uint8)
    T = st.pose(t)
    min_points = int(pick([1, 50, 500]))
    m = oracle.OracleMapper(p, rig)
    m.set_mode(True, True)
    m.set_observation(t, l, r, T)
    idx = oracle.select_events_sgm(st.ev_left, t, p.bm_half_slice_thickness, p.process_event_num)
    n_ref, d_ref = m.init_sgm(l, r, st.ev_left[idx], min_points=min_points)
    dev = lib.Esvo(p, rig)
    dev.ts_push_events(0, st.ev_left)
    dev.set_observation(t, l, r, T)
    n_dev, d_dev = dev.init_sgm(l, r, min_points=min_points)
    bad = []
    if not np.array_equal(d_dev, d_ref):
        bad.append(("disparity", int(np.count_nonzero(d_dev != d_ref))))
    if n_dev != n_ref:
        bad.append(("points", n_dev, n_ref))
    for name, a, b in (("frame", dev.get_last_frame(), m.get_last_frame()), ("map", dev.get_map(), m.get_map())):
        if len(a) != len(b):
            bad.append((name, len(a), len(b)))
        else:
            for f in FIELDS:
                if not np.array_equal(a[f], b[f]):
                    bad.append((name, f))
                    break
    dev.close()
    valid = float((d_ref >= 0).mean())
    return bad, f"{rig_name} {W}x{H} images {kind} events {len(idx)} min_points {min_points}: {n_ref} points, {100 * valid:.0f} % of the disparity image valid"


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 20
    s0 = int(sys.argv[2]) if len(sys.argv) > 2 else 11000
    failed = 0
    t0 = time.time()
    for seed in range(s0, s0 + n):
        try:
            bad, brief = run_case(seed)
        except Exception as e:  # noqa: BLE001
            bad, brief = [("-", f"{type(e).__name__}: {e}")], ""
        failed += bool(bad)
        print(f"seed {seed}: {'EQUAL' if not bad else 'DIFFERENT ' + str(bad[:4])}  {brief}", flush=True)
    print(f"{n} cases, {failed} with a difference, {time.time() - t0:.0f} s")
    sys.exit(1 if failed else 0)


if __name__ == "__main__":
    main()
