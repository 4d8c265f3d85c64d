"""Deterministic synthetic stereo event streams.

A cloud of 3-D edge points (dense samples of random line segments with inverse depth in the
config's range) is observed by a stereo rig moving on a smooth trajectory.  Whenever a
projected point enters a new raw pixel an event is emitted there (left and right cameras
independently), plus a fraction of uniform noise events.  The Time Surfaces built from these
events therefore carry real stereo structure: block matching finds the true disparity, the LM
refinement converges and fusion accumulates — a random TS would be rejected at EventBM's
low-texture test and measure nothing.

No dataset is needed; `numpy.random.default_rng(seed)` makes every stream reproducible.
"""
import math

import numpy as np

from .abi import EVENT_DTYPE, event_ns, make_events
from .calib import rect_to_raw


def _rot_y(a):
    c, s = math.cos(a), math.sin(a)
    return np.array([[c, 0, s], [0, 1, 0], [-s, 0, c]], np.float64)


class Trajectory:
    """Smooth camera motion: constant velocity along +x plus small sinusoidal sway and yaw."""

    def __init__(self, speed=1.0, sway=0.02, yaw=0.01, period=0.8, t0_s=0.0):
        self.speed, self.sway, self.yaw, self.period, self.t0_s = speed, sway, yaw, period, t0_s

    def T_world_cam(self, t_s):
        t_s = t_s - self.t0_s
        w = 2 * math.pi / self.period
        R = _rot_y(self.yaw * math.sin(w * t_s))
        T = np.eye(4)
        T[:3, :3] = R
        T[:3, 3] = [self.speed * t_s, self.sway * math.sin(w * t_s), 0.3 * self.sway * math.cos(w * t_s)]
        return T


class SynthStream:
    def __init__(self, rig, ev_left, ev_right, traj, t0_ns, t1_ns, points_world):
        self.rig, self.ev_left, self.ev_right = rig, ev_left, ev_right
        self.traj, self.t0_ns, self.t1_ns = traj, t0_ns, t1_ns
        self.points_world = points_world
        self.ns_left = event_ns(ev_left)
        self.ns_right = event_ns(ev_right)

    def pose(self, t_ns):
        return self.traj.T_world_cam(t_ns * 1e-9)

    def slice(self, cam, t_lo_ns, t_hi_ns):
        """events with t_lo <= ts < t_hi"""
        ns = self.ns_left if cam == 0 else self.ns_right
        ev = self.ev_left if cam == 0 else self.ev_right
        a, b = np.searchsorted(ns, [t_lo_ns, t_hi_ns], side="left")
        return ev[a:b]

    def true_inv_depth_image(self, t_ns):
        """sparse GT: (u, v, rho) of the scene points in the rectified left view at t."""
        u, v, z, _ = _project(self.rig, self.points_world, self.pose(t_ns))
        return u, v, 1.0 / z


def _project(rig, pw, T_world_cam, wrap=None):
    R, c = T_world_cam[:3, :3], T_world_cam[:3, 3]
    d = pw - c
    if wrap is not None:  # stationary scene: a point that leaves the (wider than the view) strip re-enters on the other side
        d = d.copy()
        d[:, 0] = (d[:, 0] + 0.5 * wrap) % wrap - 0.5 * wrap
    pc = d @ R  # R^T (p - c)
    P = rig.left.P.reshape(3, 4)
    f, cx, cy = P[0, 0], P[0, 2], P[1, 2]
    z = pc[:, 2]
    zs = np.where(z > 1e-6, z, 1.0)
    u = f * pc[:, 0] / zs + cx
    v = f * pc[:, 1] / zs + cy
    ur = u - f * rig.baseline / zs
    return u, v, np.where(z > 1e-6, z, -1.0), ur


def _raw_coords(rig, cam, u, v):
    intr = rig.intr_left if cam == 0 else rig.intr_right
    if intr is None or (not np.any(intr["D"]) and np.allclose(intr["R"], np.eye(3)) and np.allclose(intr["K"], intr["P"][:, :3])):
        return u, v
    return rect_to_raw(u, v, intr["K"], intr["D"], intr["R"], intr["P"], intr["model"])


def make_scene(rig, n_points, rho_min, rho_max, rng, margin=1.3):
    """Random 3-D line segments, sampled at about one rectified pixel spacing."""
    P = rig.left.P.reshape(3, 4)
    f, cx, cy = P[0, 0], P[0, 2], P[1, 2]
    W, H = rig.width, rig.height
    pts = []
    total = 0
    while total < n_points:
        rho = rng.uniform(rho_min * 1.15, rho_max * 0.85)
        z = 1.0 / rho
        u0 = rng.uniform(-0.1 * W, 1.1 * W) * margin - (margin - 1) * W / 2
        v0 = rng.uniform(0.02 * H, 0.98 * H)
        length_px = rng.uniform(15, 90)
        ang = rng.uniform(-0.45 * math.pi, 0.45 * math.pi) + math.pi / 2  # mostly vertical edges
        n = max(int(length_px), 2)
        s = np.linspace(-0.5, 0.5, n) * length_px
        uu = u0 + s * math.cos(ang)
        vv = v0 + s * math.sin(ang)
        dz = rng.uniform(-0.08, 0.08) * z * np.linspace(-1, 1, n)  # slanted in depth
        zz = z + dz
        X = (uu - cx) / f * zz
        Y = (vv - cy) / f * zz
        pts.append(np.stack([X, Y, zz], axis=1))
        total += n
    return np.concatenate(pts)[:n_points]


def make_stream(rig, n_points, duration_s, rho_min, rho_max, seed, speed=1.0, t0_s=10.0,
                chunk_s=2e-3, noise_frac=0.05, traj=None, stationary=False):
    """stationary=True: the scene is periodic along the direction of travel (per point, with the width of the strip
    make_scene fills at that depth), so the event rate does not depend on the duration -- the benchmark stream."""
    rng = np.random.default_rng(seed)
    traj = traj or Trajectory(speed=speed, t0_s=t0_s)
    pw = make_scene(rig, n_points, rho_min, rho_max, rng)
    wrap = None
    if stationary:
        P = rig.left.P.reshape(3, 4)
        wrap = 1.56 * rig.width / P[0, 0] * pw[:, 2]  # make_scene's strip: u0 in [-0.28 W, 1.28 W] at margin 1.3
        pw[:, 0] -= (P[0, 2] - 0.5 * rig.width) / P[0, 0] * pw[:, 2]  # centre the strip on the optical axis
    else:
        # spread the scene along x so that points keep entering the view while the rig moves
        pw[:, 0] += rng.uniform(0, speed * duration_s, size=pw.shape[0]) * 0.5
    W, H = rig.width, rig.height
    n_chunks = max(int(math.ceil(duration_s / chunk_s)), 1)
    out = {0: [], 1: []}
    t_prev = t0_s
    u, v, z, ur = _project(rig, pw, traj.T_world_cam(t_prev), wrap)
    prev = {0: _raw_coords(rig, 0, u, v), 1: _raw_coords(rig, 1, ur, v)}
    z_prev = z
    for ci in range(n_chunks):
        t_next = t0_s + (ci + 1) * duration_s / n_chunks
        u, v, z, ur = _project(rig, pw, traj.T_world_cam(t_next), wrap)
        cur = {0: _raw_coords(rig, 0, u, v), 1: _raw_coords(rig, 1, ur, v)}
        for cam in (0, 1):
            ax, ay = prev[cam]
            bx, by = cur[cam]
            ok = (z_prev > 0) & (z > 0) & np.isfinite(ax) & np.isfinite(bx)
            if wrap is not None:
                ok &= np.abs(bx - ax) < 0.5 * W  # a point that wrapped around moved outside the view: no event trail
            disp = np.maximum(np.abs(bx - ax), np.abs(by - ay))
            disp = np.where(ok, disp, 0.0)
            n_sub = int(min(max(math.ceil(float(disp.max(initial=0.0)) * 2.0), 1), 64))
            px0 = np.floor(ax + 0.5)
            py0 = np.floor(ay + 0.5)
            for k in range(1, n_sub + 1):
                a = k / n_sub
                px1 = np.floor(ax + (bx - ax) * a + 0.5)
                py1 = np.floor(ay + (by - ay) * a + 0.5)
                moved = ok & ((px1 != px0) | (py1 != py0)) & (px1 >= 0) & (px1 < W) & (py1 >= 0) & (py1 < H)
                idx = np.nonzero(moved)[0]
                if idx.size:
                    ts = t_prev + (t_next - t_prev) * ((k - 1 + rng.random(idx.size)) / n_sub)
                    pol = (bx[idx] >= ax[idx]).astype(np.uint8)
                    out[cam].append((px1[idx].astype(np.int64), py1[idx].astype(np.int64), ts, pol))
                px0, py0 = px1, py1
        prev, z_prev, t_prev = cur, z, t_next
    streams = []
    for cam in (0, 1):
        if out[cam]:
            x = np.concatenate([o[0] for o in out[cam]])
            y = np.concatenate([o[1] for o in out[cam]])
            t = np.concatenate([o[2] for o in out[cam]])
            p = np.concatenate([o[3] for o in out[cam]])
        else:
            x = y = np.zeros(0, np.int64); t = np.zeros(0); p = np.zeros(0, np.uint8)
        n_noise = int(noise_frac * x.size)
        if n_noise:
            x = np.concatenate([x, rng.integers(0, W, n_noise)])
            y = np.concatenate([y, rng.integers(0, H, n_noise)])
            t = np.concatenate([t, rng.uniform(t0_s, t0_s + duration_s, n_noise)])
            p = np.concatenate([p, rng.integers(0, 2, n_noise).astype(np.uint8)])
        t_ns = np.round(t * 1e9).astype(np.int64)
        order = np.argsort(t_ns, kind="stable")
        streams.append(make_events(x[order], y[order], t_ns[order].astype(np.uint64), p[order]))
    t0_ns = int(round(t0_s * 1e9))
    t1_ns = int(round((t0_s + duration_s) * 1e9))
    return SynthStream(rig, streams[0], streams[1], traj, t0_ns, t1_ns, pw)


def random_events(width, height, n, t0_ns, t1_ns, seed):
    """Uniform random events (for ingest / raster unit tests)."""
    rng = np.random.default_rng(seed)
    t = np.sort(rng.integers(t0_ns, t1_ns, n)).astype(np.uint64)
    return make_events(rng.integers(0, width, n), rng.integers(0, height, n), t, rng.integers(0, 2, n).astype(np.uint8))
