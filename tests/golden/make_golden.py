#!/usr/bin/env python
"""Generates tests/golden/*.npz — seeded inputs and the CPU oracle's outputs for them.

The reference (ESVO) ships no golden vectors and cannot be built or imported in this
environment, so these fixtures pin the ORACLE (regression gate for oracle/ and the bar for the
GPU path), not the reference itself: "parity unpinned" (see oracle/esvo_oracle.h).
Oracle mode: exact-integer ZNCC moments + canonical reduction order, i.e. the mode that is
bit-comparable with the GPU.

    python tests/golden/make_golden.py        # rewrites the .npz files
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

from esvo_amd import calib, params, rostime, synth  # noqa: E402
from oracle import oracle as O  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))


def make(name, rig_name, preset, n_points, rho, seed, n_ticks, n_events, speed=1.0):
    rig = calib.dataset_rig(rig_name)
    st = synth.make_stream(rig, n_points, 0.07 + 0.01 * n_ticks, rho[0], rho[1], seed=seed, speed=speed)
    p, _ = params.make_params(params.PRESETS[preset], rig, process_event_num=n_events)
    m = O.OracleMapper(p, rig)
    m.set_mode(True, True)
    ts = [O.OracleTS(rig.width, rig.height), O.OracleTS(rig.width, rig.height)]
    ts[0].push(st.ev_left)
    ts[1].push(st.ev_right)
    out = dict(rig=rig_name, preset=preset, n_events=n_events, n_ticks=n_ticks)
    for k in range(n_ticks):
        t = st.t0_ns + int((0.06 + 0.01 * k) * 1e9)
        l = ts[0].render(t, map_x=rig.left.map_x, map_y=rig.left.map_y)
        r = ts[1].render(t, map_x=rig.right.map_x, map_y=rig.right.map_y)
        stamps, poses = rostime.pose_table(st.pose, t, p.bm_half_slice_thickness)
        idx = O.select_events(st.ev_left, t, p.bm_half_slice_thickness, p.process_event_num)
        ev = st.ev_left[idx]
        m.set_observation(t, l, r, st.pose(t))
        m.set_poses(stamps, poses)
        mt = m.match(ev)
        pts = m.refine(mt, cull=True)
        m.push_frame(pts)
        nf = m.fuse()
        out.update({f"t{k}": t, f"tsL{k}": l, f"tsR{k}": r, f"T{k}": st.pose(t), f"stamps{k}": stamps, f"poses{k}": poses,
                    f"ev{k}": ev, f"matches{k}": mt, f"points{k}": pts, f"nf{k}": nf, f"map{k}": m.get_map()})
    path = os.path.join(HERE, name + ".npz")
    np.savez_compressed(path, **out)
    print(name, os.path.getsize(path) // 1024, "KiB", [len(out[f"points{k}"]) for k in range(n_ticks)],
          [len(out[f"map{k}"]) for k in range(n_ticks)])


def make_track(name, base):
    """tracker residual / Jacobian vectors (RegProblemLM.cpp) on the last tick of an existing mapper fixture: the local map's
    point cloud as reference, that tick's left Time Surface as current frame, a small warp"""
    g = np.load(os.path.join(HERE, base + ".npz"))
    rig = calib.dataset_rig(str(g["rig"]))
    k = int(g["n_ticks"]) - 1
    mp, T = g[f"map{k}"], np.asarray(g[f"T{k}"], np.float64).reshape(4, 4)
    xyz = (mp["p_cam"] @ T[:3, :3].T + T[:3, 3]).astype(np.float32)     # publishPointCloud, esvo_Mapping.cpp:925-932
    rng = np.random.default_rng(20250503)
    xyz = xyz[rng.permutation(len(xyz))][:2000]
    c = np.array([0.004, -0.003, 0.002])
    kk = 1 + c @ c
    R_ = np.array([[1 + c[0]**2 - c[1]**2 - c[2]**2, 2 * (c[0] * c[1] - c[2]), 2 * (c[0] * c[2] + c[1])],
                   [2 * (c[0] * c[1] + c[2]), 1 - c[0]**2 + c[1]**2 - c[2]**2, 2 * (c[1] * c[2] - c[0])],
                   [2 * (c[0] * c[2] - c[1]), 2 * (c[1] * c[2] + c[0]), 1 - c[0]**2 - c[1]**2 + c[2]**2]]) / kk
    t_ = np.array([0.01, -0.005, 0.008])
    Tw = np.eye(4)
    Tw[:3, :3] = R_.T
    Tw[:3, 3] = -R_.T @ t_
    trk = O.OracleTracker(rig)
    trk.set_current(g[f"tsL{k}"], 5)
    trk.set_reference(xyz, T)
    neg, du, dv = trk.images()
    out = dict(rig=str(g["rig"]), tsL=g[f"tsL{k}"], xyz=xyz, T_world_ref=T, T_left_ref=Tw, R=R_, t=t_, neg=neg, du=du, dv=dv,
               fvec_huber=trk.residuals(Tw, 0, 300, huber=True, huber_threshold=50.0),
               fvec_l2=trk.residuals(Tw, 300, 300, huber=False), fjac=trk.jacobian(R_, t_, 0, 300))
    path = os.path.join(HERE, name + ".npz")
    np.savez_compressed(path, **out)
    print(name, os.path.getsize(path) // 1024, "KiB", len(xyz), "points,", int((out["fvec_l2"] < 255).sum()), "of 300 reproject")


if __name__ == "__main__":
    if "--track-only" not in sys.argv:
        make("upenn_small", "upenn", "mvstereo_upenn", 5000, (0.16, 1.0), 20250501, 3, 600)
        make("dsec_small", "dsec", "mapping_dsec", 8000, (0.02, 0.25), 20250502, 2, 600, speed=2.0)
    make_track("track_small", "upenn_small")
