"""Closed-loop harness: SGM bootstrap -> tracker -> mapper, all three on the device.

The tracker's optimiser is host-side C++ inside the library (esvo_track_register: a damped Gauss-Newton over
esvo_track_normal_equations -- the same residual and Jacobian RegProblemLM hands to Eigen's LM,
esvo_core/src/core/RegProblemLM.cpp:93-236, reduced to J^T J / J^T f on the device);
the poses the mapper fuses with are the tracker's estimates, never ground truth after the bootstrap pose.

Property of the formulation worth knowing when reading the bars: the map sits on the LEADING edge of the
Time Surface (the newest events), the blurred negative surface has its minimum about 0.4 trail lengths
BEHIND it, so the registered pose lags the true one by a constant ~1.2-1.7 ticks of motion.  With a fixed
reference map the lag stays bounded while the path grows; every re-reference to a map fused with lagged
poses adds one more lag.  The bars below are measured with that in mind.
"""
import time

import numpy as np

from esvo_amd import calib, lib, params, rostime, synth

TICK_NS = 10_000_000


def cayley2rot(c):
    s = 1 + c @ c
    R = np.array([[1 + c[0]**2 - c[1]**2 - c[2]**2, 2*(c[0]*c[1]-c[2]), 2*(c[0]*c[2]+c[1])],
                  [2*(c[0]*c[1]+c[2]), 1 - c[0]**2 + c[1]**2 - c[2]**2, 2*(c[1]*c[2]-c[0])],
                  [2*(c[0]*c[2]-c[1]), 2*(c[1]*c[2]+c[0]), 1 - c[0]**2 - c[1]**2 + c[2]**2]])
    return R / s


def orth(R):
    U, _, Vt = np.linalg.svd(R)
    return U @ Vt


def register(dev, n_points, R_, t_, iters=12):
    """The registration of one frame: the library's driver (esvo_track_register: host C++ over esvo_track_normal_equations,
    the three trial dampings of an iteration in one launch, 224 bytes back per trial)."""
    R, t, rms, _ = dev.track_register(n_points, R_, t_, huber=True, huber_threshold=50.0, max_iterations=iters, damping=1e-3)
    return R, t, rms


def lm_gn_loop(evaluate, R_, t_, iters=12, damping=1e-3):
    """include/esvo_hip.hpp's gauss_newton_register in numpy (same batch every iteration): Levenberg-damped Gauss-Newton steps
    linearised at x = 0, each accepted only if actual / predicted reduction >= 1e-4 (cost evaluated at the trial pose), the
    damping raised tenfold otherwise (6 attempts), lowered tenfold (not below `damping`) after an accepted step and carried
    over.  The C++ driver evaluates three trial dampings per call; taking the first acceptable one in rising order is what this
    sequential loop does.  evaluate(R, t) -> (H, b, cost, n).  Returns (R, t, rms, iterations)."""
    H, b, cost, n = evaluate(R_, t_)
    it, lam = 0, damping
    for it in range(iters):
        accepted = False
        for _ in range(6):
            dx = np.linalg.solve(H + lam * np.diag(np.diag(H)) + 1e-9 * np.eye(6), -b)
            dR = cayley2rot(dx[:3])
            Rn, tn = orth(dR @ R_), dx[3:] + dR @ t_
            Ht, bt, cost_t, nt = evaluate(Rn, tn)
            pred = -(2.0 * b @ dx + dx @ H @ dx)
            if pred > 0 and (cost - cost_t) >= 1e-4 * pred:
                accepted = True
                break
            lam *= 10.0
        if not accepted:
            break
        R_, t_, H, b, cost, n = Rn, tn, Ht, bt, cost_t, nt
        lam = max(lam / 10.0, damping)
        if np.linalg.norm(dx) < 1e-6:
            break
    return R_, t_, float(np.sqrt(cost / n)) if n else 0.0, it + 1


def register_python(dev, n_points, R_, t_, iters=12):
    """The library's driver (esvo_track_register) restated in Python over esvo_track_residuals / esvo_track_jacobian (two
    synchronous calls and a count x 7 download per evaluation): the cross-check of the C++ one."""
    def evaluate(R, t):
        Tlr = np.eye(4)
        Tlr[:3, :3] = R.T
        Tlr[:3, 3] = -R.T @ t
        r = dev.track_residuals(Tlr, 0, n_points, huber=True, huber_threshold=50.0)
        J = dev.track_jacobian(R, t, 0, n_points)
        return J.T @ J, J.T @ r, float(r @ r), len(r)
    R, t, rms, _ = lm_gn_loop(evaluate, R_, t_, iters)
    return R, t, rms


def run(n_ticks=15, reref=10**9, speed=1.0, seed=20250419, verbose=False):
    rig = calib.dataset_rig("upenn")
    st = synth.make_stream(rig, 8000, 0.35, 0.16, 1.0, seed=seed, speed=speed)
    p, _ = params.make_params(params.PRESETS["mapping_upenn"], rig)
    dev = lib.Esvo(p, rig)
    dev.ts_push_events(0, st.ev_left)
    dev.ts_push_events(1, st.ev_right)
    t0 = st.t0_ns + int(0.08e9)
    dev.ts_render(0, t0, download=False)
    dev.ts_render(1, t0, download=False)
    T_est = {t0: st.pose(t0)}             # bootstrap pose given, as the reference's identity at start-up
    dev.set_observation(t0, None, None, T_est[t0])
    n_sgm, _ = dev.init_sgm(None, None, min_points=100)
    out = {"sgm_points": n_sgm, "pos_err": [], "rot_err_deg": [], "cos": [], "est_len": [], "gt_len": [], "points": [],
           "cycle_ms": [], "track_ms": [], "map_ms": []}
    rng = np.random.default_rng(0)
    t_ref, xyz, sel, R_, t_ = t0, None, None, np.eye(3), np.zeros(3)
    for k in range(1, n_ticks + 1):
        t = t0 + k * TICK_NS
        c0 = time.perf_counter()
        dev.ts_render(0, t, download=False)
        dev.ts_render(1, t, download=False)
        if xyz is None or (k - 1) % reref == 0:
            xyz = dev.get_pointcloud()
            sel = rng.permutation(len(xyz))[:2000]
            t_ref = t - TICK_NS
            R_, t_ = np.eye(3), np.zeros(3)
        dev.track_set_current(None, 5)
        dev.track_set_reference(xyz[sel], T_est[t_ref])
        R_, t_, rms = register(dev, len(sel), R_, t_)
        Tw = np.eye(4)
        Tw[:3, :3] = T_est[t_ref][:3, :3] @ R_
        Tw[:3, 3] = T_est[t_ref][:3, :3] @ t_ + T_est[t_ref][:3, 3]
        T_est[t] = Tw
        c1 = time.perf_counter()
        gt = st.pose(t)
        d_est = Tw[:3, 3] - T_est[t0][:3, 3]
        d_gt = gt[:3, 3] - st.pose(t0)[:3, 3]
        out["pos_err"].append(float(np.linalg.norm(Tw[:3, 3] - gt[:3, 3])))
        out["rot_err_deg"].append(float(np.degrees(np.arccos(np.clip((np.trace(Tw[:3, :3].T @ gt[:3, :3]) - 1) / 2, -1, 1)))))
        out["cos"].append(float(d_est @ d_gt / (np.linalg.norm(d_est) * np.linalg.norm(d_gt) + 1e-12)))
        out["est_len"].append(float(np.linalg.norm(d_est)))
        out["gt_len"].append(float(np.linalg.norm(d_gt)))

        def pose_at(tq, a=T_est[t - TICK_NS], b=Tw, ta=t - TICK_NS):
            w = (tq - ta) / TICK_NS                       # tf-style interpolation between tracked poses
            T = np.eye(4)
            T[:3, 3] = (1 - w) * a[:3, 3] + w * b[:3, 3]
            T[:3, :3] = orth((1 - w) * a[:3, :3] + w * b[:3, :3])
            return T
        stamps, poses = rostime.pose_table(pose_at, t, p.bm_half_slice_thickness)
        c2 = time.perf_counter()
        dev.set_observation(t, None, None, Tw)
        dev.tick(t, stamps, poses)
        out["points"].append(int(dev.stats().last_points))   # (reading the statistics completes the tick)
        c3 = time.perf_counter()
        out["track_ms"].append((c1 - c0) * 1e3)               # both renders + reference upload + the registration
        out["map_ms"].append((c3 - c2) * 1e3)
        out["cycle_ms"].append((c1 - c0 + c3 - c2) * 1e3)
        if verbose:
            print(f"tick {k}: |r| {rms:.1f} pos err {out['pos_err'][-1]*1e3:.2f} mm of {out['gt_len'][-1]*1e3:.1f} mm, "
                  f"cos {out['cos'][-1]:.3f}, rot {out['rot_err_deg'][-1]:.3f} deg, points {out['points'][-1]}")
    mp = dev.get_map()
    u, v, rho = st.true_inv_depth_image(t)
    ok = (u >= 0) & (u < rig.width) & (v >= 0) & (v < rig.height)
    gtimg = {(int(b), int(a)): c for a, b, c in zip(u[ok], v[ok], rho[ok])}
    err = np.array([m["inv_depth"] - gtimg[(int(m["row"]), int(m["col"]))] for m in mp
                    if (int(m["row"]), int(m["col"])) in gtimg and m["inv_depth"] > 0])
    out["map_cells"] = len(mp)
    out["map_on_gt"] = len(err)
    out["map_median_abs_err"] = float(np.median(np.abs(err))) if len(err) else float("nan")
    dev.close()
    return out
