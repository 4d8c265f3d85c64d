"""Randomised tracker evaluation (RegProblemLM's setProblem / operator() / df on the device) against the CPU
oracle, bit for bit: a random rig, a Time Surface rendered from a seeded stream (or noise), the blur kernel 0 / 5, a cloud of
random world points -- in view, at the image border, behind the camera, far outside -- a random reference pose and trial poses from
a millimetre to decimetres / tens of degrees away, random batch windows (inside, across and beyond the cloud), Huber or l2 with a random
threshold.  Compared: the negated blurred image and its Sobel derivatives, the residual vector, the Jacobian, the normal equations
(H, b, |f|^2, n) alone and several poses per launch, and the library's registration loop against the same loop written in numpy
over the device's own residuals / Jacobian.
usage: python tools/fuzz_track.py [cases] [first seed]      (GPU; exits 1 on any difference)"""
import os
import sys
import time

ROOT = os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402

from esvo_amd import calib, closed_loop, lib, params, synth  # noqa: E402
from oracle import oracle  # noqa: E402


def run_case(seed):
    rng = np.random.default_rng(seed)
    pick = lambda xs: xs[int(rng.integers(len(xs)))]  # noqa: E731
    rig_name = pick(["upenn", "rpg", "hkust", "dsec", "ideal"])
    rig = calib.ideal_rig(pick([64, 346]), pick([48, 260]), 150.0, 0.1) if rig_name == "ideal" else calib.dataset_rig(rig_name)
    preset = {"dsec": "mapping_dsec", "rpg": "mapping_rpg", "hkust": "mapping_hkust"}.get(rig_name, "mapping_upenn")
    p, _ = params.make_params(params.PRESETS[preset], rig)
    W, H = rig.width, rig.height
    if rng.integers(4) == 0:
        ts = rng.integers(0, 256, (H, W)).astype(np.uint8)          # noise: every pixel a different gradient
    else:
        st = synth.make_stream(rig, pick([2000, 8000]), 0.08, 0.2, 1.0, seed=int(rng.integers(1 << 30)), speed=1.0)
        o = oracle.OracleTS(W, H)
        o.push(st.ev_left)
        ts = o.render(st.t0_ns + int(0.07e9), map_x=rig.left.map_x, map_y=rig.left.map_y)
    ks = pick([0, 5, 5])   # (the two sizes the library takes: 0 and the shipped 5)
    n = int(pick([1, 7, 300, 900, 2500]))
    # world points: around a plane 1-4 m ahead of the reference camera, some deliberately bad
    K = np.array(rig.left.P).reshape(3, 4)[:, :3]
    u = rng.uniform(-20, W + 20, n)
    v = rng.uniform(-20, H + 20, n)
    z = rng.uniform(0.8, 4.0, n)
    z[rng.random(n) < 0.03] *= -1.0                      # behind the camera
    pc = (np.linalg.inv(K) @ np.stack([u, v, np.ones(n)])) * z
    T_ref = np.eye(4)
    T_ref[:3, :3] = closed_loop.orth(closed_loop.cayley2rot(rng.normal(0, 0.3, 3)))
    T_ref[:3, 3] = rng.normal(0, 2.0, 3)
    xyz = ((T_ref[:3, :3] @ pc).T + T_ref[:3, 3]).astype(np.float32)
    dev = lib.Esvo(p, rig)
    trk = oracle.OracleTracker(rig)
    dev.track_set_current(ts, ks)
    trk.set_current(ts, ks)
    bad = []
    for a, b, name in zip(dev.track_images(), trk.images(), ("negative", "du", "dv")):
        if not np.array_equal(a, b):
            bad.append((name, int(np.count_nonzero(a != b))))
    dev.track_set_reference(xyz, T_ref)
    trk.set_reference(xyz, T_ref)
    n_eval = 0
    for _ in range(6):
        s = pick([1e-3, 1e-2, 1e-1, 0.5])
        R = closed_loop.orth(closed_loop.cayley2rot(rng.normal(0, s, 3)))
        t = rng.normal(0, s, 3)
        off = int(pick([0, 0, n // 3, max(n - 5, 0), n + 10]))
        cnt = int(pick([1, 50, 300, n, 2 * n + 3]))
        huber, thr = bool(rng.integers(2)), float(pick([5.0, 50.0, 500.0]))
        Tlr = np.eye(4)
        Tlr[:3, :3] = R.T
        Tlr[:3, 3] = -R.T @ t
        fd, fo = dev.track_residuals(Tlr, off, cnt, huber, thr), trk.residuals(Tlr, off, cnt, huber=huber, huber_threshold=thr)
        Jd, Jo = dev.track_jacobian(R, t, off, cnt), trk.jacobian(R, t, off, cnt)
        Hd, bd, cd, nd = dev.track_normal_equations(R, t, off, cnt, huber, thr)
        Ho, bo, co, no = trk.normal_equations(R, t, off, cnt, huber=huber, huber_threshold=thr)
        n_eval += 1
        if fd.shape != fo.shape or fd.tobytes() != fo.tobytes():
            bad.append(("residuals", off, cnt))
        if Jd.shape != Jo.shape or Jd.tobytes() != Jo.tobytes():
            bad.append(("jacobian", off, cnt))
        if nd != no or Hd.tobytes() != Ho.tobytes() or bd.tobytes() != bo.tobytes() or cd != co:
            bad.append(("normal equations", off, cnt, nd, no))
    # several poses in one launch == each alone
    k = int(rng.integers(1, 5))
    Rs = np.stack([closed_loop.orth(closed_loop.cayley2rot(rng.normal(0, 0.02, 3))) for _ in range(k)])
    tt = rng.normal(0, 0.03, (k, 3))
    Hb, bb, cb, nb = dev.track_normal_equations_batch(Rs, tt, 0, n)
    for q in range(k):
        Ho, bo, co, no = trk.normal_equations(Rs[q], tt[q], 0, n)
        if nb != no or Hb[q].tobytes() != Ho.tobytes() or bb[q].tobytes() != bo.tobytes() or cb[q] != co:
            bad.append(("batch", q, k))
    # the registration loop inside the library == the same loop in numpy over the device's residuals and Jacobian
    if n >= 300:
        R0 = closed_loop.orth(closed_loop.cayley2rot(rng.normal(0, 0.01, 3)))
        t0 = rng.normal(0, 0.01, 3)
        R1, t1, rms1, it1 = dev.track_register(n, R0, t0)
        R2, t2, rms2 = closed_loop.register_python(dev, n, R0, t0)
        if np.abs(R1 - R2).max() > 1e-9 or np.abs(t1 - t2).max() > 1e-9 or abs(rms1 - rms2) > 1e-6 * abs(rms2) + 1e-9:
            bad.append(("register", float(np.abs(R1 - R2).max()), float(np.abs(t1 - t2).max())))
    dev.close()
    return bad, f"{rig_name} {W}x{H} kernel {ks} points {n} evaluations {n_eval} + batch of {k}"


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 20
    s0 = int(sys.argv[2]) if len(sys.argv) > 2 else 7000
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
