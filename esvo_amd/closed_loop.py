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


class _OneGpu:
    """the calls of the loop on one plain handle"""

    def __init__(self, dev):
        self.dev = dev

    def full_left_ts(self):
        pass

    def pointcloud(self):
        return self.dev.get_pointcloud()

    def tick(self, t, stamps, poses):
        self.dev.tick(t, stamps, poses)

    def final_map(self):
        return self.dev.get_map()


class _Band:
    """... on one rank of a band-sharded run (every call here is collective: the ranks make them in the same order)"""

    def __init__(self, dev):
        self.dev = dev

    def full_left_ts(self):
        self.dev.comm_gather_ts(0)         # the tracker reads the WHOLE left Time Surface; a routed rank rendered its rows

    def pointcloud(self):
        return self.dev.comm_gather_pointcloud()   # the whole map's cloud on every rank, merged on the device

    def tick(self, t, stamps, poses):
        self.dev.comm_shard_tick(t, stamps, poses)

    def final_map(self):
        return self.dev.comm_gather_map()


def _loop(rig, st, p, ops, t0, T0, xyz0, n_ticks, reref, verbose=False):
    """tracker -> mapper for n_ticks ticks after the bootstrap at t0 (pose T0, reference cloud xyz0); `ops`: _OneGpu / _Band"""
    dev = ops.dev
    T_est = {t0: T0}
    out = {"pos_err": [], "rot_err_deg": [], "cos": [], "est_len": [], "gt_len": [], "points": [],
           "cycle_ms": [], "track_ms": [], "map_ms": [], "poses": []}
    rng = np.random.default_rng(0)
    t_ref, xyz, sel, R_, t_ = t0, None, None, np.eye(3), np.zeros(3)
    for k in range(1, n_ticks + 1):
        t = t0 + k * TICK_NS
        c0 = time.perf_counter()
        dev.ts_render(0, t, download=False)
        dev.ts_render(1, t, download=False)
        ops.full_left_ts()
        if xyz is None or (k - 1) % reref == 0:
            xyz = xyz0 if (xyz is None and xyz0 is not None) else ops.pointcloud()
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
        out["poses"].append(Tw.copy())

        def pose_at(tq, a=T_est[t - TICK_NS], b=Tw, ta=t - TICK_NS):
            w = (tq - ta) / TICK_NS                       # tf-style interpolation between tracked poses
            T = np.eye(4)
            T[:3, 3] = (1 - w) * a[:3, 3] + w * b[:3, 3]
            T[:3, :3] = orth((1 - w) * a[:3, :3] + w * b[:3, :3])
            return T
        stamps, poses = rostime.pose_table(pose_at, t, p.bm_half_slice_thickness)
        c2 = time.perf_counter()
        dev.set_observation(t, None, None, Tw)
        ops.tick(t, stamps, poses)
        out["points"].append(int(dev.stats().last_points))   # (reading the statistics completes the tick)
        c3 = time.perf_counter()
        out["track_ms"].append((c1 - c0) * 1e3)               # both renders + reference upload + the registration
        out["map_ms"].append((c3 - c2) * 1e3)
        out["cycle_ms"].append((c1 - c0 + c3 - c2) * 1e3)
        if verbose:
            print(f"tick {k}: |r| {rms:.1f} pos err {out['pos_err'][-1]*1e3:.2f} mm of {out['gt_len'][-1]*1e3:.1f} mm, "
                  f"cos {out['cos'][-1]:.3f}, rot {out['rot_err_deg'][-1]:.3f} deg, points {out['points'][-1]}")
    mp = ops.final_map()
    u, v, rho = st.true_inv_depth_image(t)
    ok = (u >= 0) & (u < rig.width) & (v >= 0) & (v < rig.height)
    gtimg = {(int(b), int(a)): c for a, b, c in zip(u[ok], v[ok], rho[ok])}
    err = np.array([m["inv_depth"] - gtimg[(int(m["row"]), int(m["col"]))] for m in mp
                    if (int(m["row"]), int(m["col"])) in gtimg and m["inv_depth"] > 0])
    out["map"] = mp
    out["map_cells"] = len(mp)
    out["map_on_gt"] = len(err)
    out["map_median_abs_err"] = float(np.median(np.abs(err))) if len(err) else float("nan")
    return out


def _scene(seed, speed):
    rig = calib.dataset_rig("upenn")
    st = synth.make_stream(rig, 8000, 0.35, 0.16, 1.0, seed=seed, speed=speed)
    p, _ = params.make_params(params.PRESETS["mapping_upenn"], rig)
    return rig, st, p, st.t0_ns + int(0.08e9)


def run(n_ticks=15, reref=10**9, speed=1.0, seed=20250419, verbose=False):
    rig, st, p, t0 = _scene(seed, speed)
    dev = lib.Esvo(p, rig)
    dev.ts_push_events(0, st.ev_left)
    dev.ts_push_events(1, st.ev_right)
    dev.ts_render(0, t0, download=False)
    dev.ts_render(1, t0, download=False)
    T0 = st.pose(t0)                      # bootstrap pose given, as the reference's identity at start-up
    dev.set_observation(t0, None, None, T0)
    n_sgm, _ = dev.init_sgm(None, None, min_points=100)
    out = _loop(rig, st, p, _OneGpu(dev), t0, T0, None, n_ticks, reref, verbose)
    out["sgm_points"] = n_sgm
    dev.close()
    return out


def run_bands(G, routing="y_rect", n_ticks=15, reref=10**9, speed=1.0, seed=20250419, transport=None):
    """The same loop with the mapper split over G row bands (BASELINE configs[2] on configs[3]'s partition): G ranks -- handles on
    ONE GPU here, driven from G threads through the library's collective calls with an in-process all-gather (`transport`:
    tests/test_gpu_comm.LocalTransport's interface) -- each ingesting its rows, rendering its band of the Time Surfaces, matching
    and refining its events, fusing and regularising its rows.  What the loop adds to the band mode: the tracker reads the
    WHOLE left Time Surface (esvo_comm_gather_ts) and registers against the WHOLE map's cloud (esvo_comm_gather_pointcloud_xyz,
    merged on the device); it runs replicated -- same inputs, same bits, same pose on every rank, nothing to exchange.
    The SGM bootstrap runs on one plain handle that sees the whole stream (initialisation precedes the sharded operation); its
    frame opens every rank's fusion window.  Returns rank 0's figures (+ "ranks_agree")."""
    import threading
    from esvo_amd import dist as edist
    transport = transport or edist.LocalTransport(G)
    rig, st, p, t0 = _scene(seed, speed)
    boot = lib.Esvo(p, rig)
    boot.ts_push_events(0, st.ev_left)
    boot.ts_push_events(1, st.ev_right)
    boot.ts_render(0, t0, download=False)
    boot.ts_render(1, t0, download=False)
    T0 = st.pose(t0)
    boot.set_observation(t0, None, None, T0)
    n_sgm, _ = boot.init_sgm(None, None, min_points=100)
    frame0, xyz0 = boot.get_last_frame(), boot.get_pointcloud()
    boot.close()
    outs, errs = [None] * G, []

    def rank_main(r):
        try:
            dev = lib.Esvo(p, rig)
            y0, y1 = edist.band_of(r, G, rig.height)
            dev.set_band(y0, y1, r, G, routing=routing)
            dev.comm_init_callbacks(r, G, lambda s, d, n, stream: transport.all_gather(r, s, d, n))
            dev.ts_push_events(0, st.ev_left)   # every rank is handed the whole stream; a routed handle keeps its rows
            dev.ts_push_events(1, st.ev_right)
            dev.push_frame(frame0, T0.reshape(1, 16))
            outs[r] = _loop(rig, st, p, _Band(dev), t0, T0, xyz0, n_ticks, reref)
            outs[r]["halo_violations"] = int(dev.stats().halo_violations)
            outs[r]["events_staged"] = [int(x) for x in dev.stats().events_staged]
            dev.close()
        except BaseException as e:  # noqa: BLE001
            errs.append((r, e))
            transport.abort()

    th = [threading.Thread(target=rank_main, args=(r,)) for r in range(G)]
    for t in th:
        t.start()
    for t in th:
        t.join(timeout=600)
    if errs:
        raise errs[0][1]
    out = outs[0]
    out["sgm_points"] = n_sgm
    out["ranks_agree"] = all(np.array_equal(np.array(o["poses"]), np.array(out["poses"])) and np.array_equal(o["map"], out["map"]) for o in outs)
    out["events_staged_max_frac"] = max(o["events_staged"][0] for o in outs) / max(len(st.ev_left), 1)
    return out
