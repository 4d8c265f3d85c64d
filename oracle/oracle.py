"""ctypes bindings of the CPU oracle (TEST INFRASTRUCTURE ONLY).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg import this module.
"""
import ctypes as C
import os
import subprocess

import numpy as np

from esvo_amd.abi import (DEPTH_POINT_DTYPE, EVENT_DTYPE, MATCH_DTYPE, CalibStruct, ParamsStruct)

_HERE = os.path.dirname(os.path.abspath(__file__))


def build(fast=False, force=False):
    """Compile the oracle with g++ (oracle/Makefile)."""
    target = "libesvo_oracle_fast.so" if fast else "libesvo_oracle.so"
    path = os.path.join(_HERE, target)
    if force and os.path.exists(path):
        os.remove(path)
    subprocess.check_call(["make", "-C", _HERE, target], stdout=subprocess.DEVNULL)  # make tracks the header deps
    return path


_libs = {}


def load(fast=False):
    key = bool(fast)
    if key in _libs:
        return _libs[key]
    target = "libesvo_oracle_fast.so" if fast else "libesvo_oracle.so"
    path = os.path.join(_HERE, target)
    if os.environ.get("ESVO_ORACLE_ASAN"):  # the AddressSanitizer / UBSan build (make asan), for tests/test_oracle_asan.py
        path = os.path.join(_HERE, "libesvo_oracle_asan.so")
        if not os.path.exists(path) or os.path.getmtime(path) < os.path.getmtime(os.path.join(_HERE, "esvo_oracle.cpp")):
            subprocess.check_call(["make", "-C", _HERE, "asan"], stdout=subprocess.DEVNULL)
    elif not os.path.exists(path):
        path = build(fast=fast)
    lib = C.CDLL(path)
    vp, u64, sz, dbl, i32 = C.c_void_p, C.c_uint64, C.c_size_t, C.c_double, C.c_int
    lib.orc_ts_create.restype = vp
    lib.orc_ts_create.argtypes = [i32, i32, i32]
    lib.orc_ts_destroy.argtypes = [vp]
    lib.orc_ts_clear.argtypes = [vp]
    lib.orc_ts_push.argtypes = [vp, vp, sz]
    lib.orc_ts_render.argtypes = [vp, u64, dbl, i32, i32, vp, vp, vp, vp]
    lib.orc_ts_render_forward.argtypes = [vp, u64, dbl, i32, i32, vp, vp, vp]
    lib.orc_median3_u8.argtypes = [vp, vp, i32, i32]
    lib.orc_remap_bilinear_u8.argtypes = [vp, vp, i32, i32, vp, vp]
    lib.orc_gaussian5_u8.argtypes = [vp, vp, i32, i32]
    lib.orc_mapper_create.restype = vp
    lib.orc_mapper_create.argtypes = [vp, vp, vp]
    lib.orc_mapper_destroy.argtypes = [vp]
    lib.orc_mapper_reset.argtypes = [vp]
    lib.orc_mapper_set_params.argtypes = [vp, vp]
    lib.orc_mapper_set_threads.argtypes = [vp, i32]
    lib.orc_mapper_set_mode.argtypes = [vp, i32, i32]
    lib.orc_mapper_baseline.restype = dbl
    lib.orc_mapper_baseline.argtypes = [vp]
    lib.orc_mapper_set_observation.argtypes = [vp, u64, vp, vp, vp]
    lib.orc_mapper_set_poses.argtypes = [vp, vp, vp, sz]
    lib.orc_select_events.restype = sz
    lib.orc_select_events.argtypes = [vp, sz, u64, dbl, sz, vp, sz]
    lib.orc_denoise_events.restype = sz
    lib.orc_denoise_events.argtypes = [vp, vp, sz, i32, i32, sz, vp]
    lib.orc_mapper_match.restype = sz
    lib.orc_mapper_match.argtypes = [vp, vp, sz, vp, sz]
    lib.orc_mapper_match_costs.restype = i32
    lib.orc_mapper_match_costs.argtypes = [vp, vp, vp, i32]
    lib.orc_mapper_refine.restype = sz
    lib.orc_mapper_refine.argtypes = [vp, vp, sz, i32, vp, sz, vp]
    lib.orc_mapper_push_frame.argtypes = [vp, vp, sz, vp, sz]
    lib.orc_mapper_fuse.restype = sz
    lib.orc_mapper_fuse.argtypes = [vp]
    lib.orc_mapper_tick.restype = sz
    lib.orc_mapper_tick.argtypes = [vp, vp, sz]
    lib.orc_mapper_tick_bm_only.restype = sz
    lib.orc_mapper_tick_bm_only.argtypes = [vp, vp, sz]
    lib.orc_mapper_map_size.restype = sz
    lib.orc_mapper_map_size.argtypes = [vp]
    lib.orc_mapper_get_map.restype = sz
    lib.orc_mapper_get_map.argtypes = [vp, vp, sz]
    lib.orc_mapper_get_map_cells.restype = sz
    lib.orc_mapper_get_map_cells.argtypes = [vp, vp, sz]
    lib.orc_mapper_get_last_frame.restype = sz
    lib.orc_mapper_get_last_frame.argtypes = [vp, vp, sz]
    lib.orc_mapper_get_pointcloud_xyz.restype = sz
    lib.orc_mapper_get_pointcloud_xyz.argtypes = [vp, vp, sz]
    lib.orc_mapper_counters.argtypes = [vp, vp]
    lib.orc_sgbm_compute.argtypes = [vp, vp, i32, i32, i32, i32, i32, i32, i32, vp]
    lib.orc_decode_event_array.restype = C.c_long
    lib.orc_decode_event_array.argtypes = [vp, sz, vp, sz, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32)]
    lib.orc_select_events_sgm.restype = sz
    lib.orc_select_events_sgm.argtypes = [vp, sz, u64, dbl, sz, vp, sz]
    lib.orc_mapper_init_sgm.restype = sz
    lib.orc_mapper_init_sgm.argtypes = [vp, vp, vp, vp, sz, sz, vp]
    lib.orc_jet_bgr.argtypes = [vp]
    lib.orc_mapper_debug_image.argtypes = [vp, i32, dbl, vp]
    lib.orc_mapper_get_pointcloud_near_xyz.restype = sz
    lib.orc_mapper_get_pointcloud_near_xyz.argtypes = [vp, dbl, vp, sz]
    lib.orc_voxel_filter.restype = sz
    lib.orc_voxel_filter.argtypes = [vp, sz, C.c_float, vp]
    lib.orc_mapper_eval_residual.restype = i32
    lib.orc_mapper_eval_residual.argtypes = [vp, vp, C.c_uint32, dbl, vp]
    lib.orc_zncc_cost.restype = dbl
    lib.orc_zncc_cost.argtypes = [vp, vp, i32, i32, i32]
    lib.orc_tracker_create.restype = vp
    lib.orc_tracker_create.argtypes = [vp]
    lib.orc_tracker_destroy.argtypes = [vp]
    lib.orc_tracker_set_current.restype = i32
    lib.orc_tracker_set_current.argtypes = [vp, vp, i32]
    lib.orc_tracker_get_images.argtypes = [vp, vp, vp, vp]
    lib.orc_tracker_set_reference.argtypes = [vp, vp, sz, vp]
    lib.orc_tracker_residuals.restype = sz
    lib.orc_tracker_residuals.argtypes = [vp, vp, sz, sz, i32, dbl, vp]
    lib.orc_tracker_jacobian.restype = sz
    lib.orc_tracker_jacobian.argtypes = [vp, vp, vp, sz, sz, vp]
    lib.orc_tracker_normal_equations.restype = sz
    lib.orc_tracker_normal_equations.argtypes = [vp, vp, vp, sz, sz, i32, dbl, vp]
    lib.orc_abi_sizes.argtypes = [vp]
    _libs[key] = lib
    return lib


def _p(a):
    return None if a is None else a.ctypes.data


class OracleTS:
    """esvo_time_surface node state: per-pixel event queues + raster."""

    def __init__(self, width, height, queue_len=20, fast=False):
        self.lib = load(fast)
        self.W, self.H = width, height
        self.h = self.lib.orc_ts_create(width, height, queue_len)

    def __del__(self):
        if getattr(self, "h", None):
            self.lib.orc_ts_destroy(self.h)
            self.h = None

    def push(self, ev):
        ev = np.ascontiguousarray(ev, dtype=EVENT_DTYPE)
        self.lib.orc_ts_push(self.h, ev.ctypes.data, ev.shape[0])

    def render_forward(self, t_ns, rect_lut, decay_ms=30.0, ignore_polarity=True, median_k=1, want_f64=False):
        """createTimeSurfaceAtTime in FORWARD mode (TimeSurface.cpp:85-116): bilinear splat at the rectified positions"""
        lut = np.ascontiguousarray(rect_lut, np.float32).reshape(self.H * self.W, 2)
        out = np.empty((self.H, self.W), np.uint8)
        f64 = np.empty((self.H, self.W), np.float64) if want_f64 else None
        self.lib.orc_ts_render_forward(self.h, int(t_ns), float(decay_ms), int(bool(ignore_polarity)), int(median_k),
                                       lut.ctypes.data, out.ctypes.data, f64.ctypes.data if want_f64 else None)
        return (out, f64) if want_f64 else out

    def render(self, t_ns, decay_ms=30.0, ignore_polarity=True, median_k=1, map_x=None, map_y=None,
               want_prefilter=False):
        out = np.empty((self.H, self.W), np.uint8)
        pre = np.empty((self.H, self.W), np.uint8) if want_prefilter else None
        self.lib.orc_ts_render(self.h, int(t_ns), float(decay_ms), int(ignore_polarity), int(median_k),
                               _p(map_x), _p(map_y), out.ctypes.data, _p(pre))
        return (out, pre) if want_prefilter else out


def median3(img):
    lib = load()
    img = np.ascontiguousarray(img, np.uint8)
    out = np.empty_like(img)
    lib.orc_median3_u8(img.ctypes.data, out.ctypes.data, img.shape[1], img.shape[0])
    return out


def remap_bilinear(img, map_x, map_y):
    lib = load()
    img = np.ascontiguousarray(img, np.uint8)
    out = np.empty_like(img)
    lib.orc_remap_bilinear_u8(img.ctypes.data, out.ctypes.data, img.shape[1], img.shape[0],
                              map_x.ctypes.data, map_y.ctypes.data)
    return out


def gaussian5(img):
    lib = load()
    img = np.ascontiguousarray(img, np.uint8)
    out = np.empty_like(img)
    lib.orc_gaussian5_u8(img.ctypes.data, out.ctypes.data, img.shape[1], img.shape[0])
    return out


def select_events(ev, t_ns, half_slice, max_num, fast=False):
    lib = load(fast)
    ev = np.ascontiguousarray(ev, dtype=EVENT_DTYPE)
    cap = min(max_num, ev.shape[0]) + 1
    idx = np.empty(cap, np.uint32)
    n = lib.orc_select_events(ev.ctypes.data, ev.shape[0], int(t_ns), float(half_slice), int(max_num),
                              idx.ctypes.data, cap)
    return idx[:n]


def decode_event_array(msg):
    """ros::serialization of one dvs_msgs/EventArray (what rosbag's instantiate<> does, EventMessageEditor.cpp:111): returns
    (events as esvo_event_t array, height, width); raises ValueError on a malformed buffer"""
    lib = load()
    buf = np.frombuffer(bytes(msg), np.uint8)
    h, w = C.c_uint32(), C.c_uint32()
    n = lib.orc_decode_event_array(buf.ctypes.data, buf.size, None, 0, C.byref(h), C.byref(w))
    if n < 0:
        raise ValueError("not a complete dvs_msgs/EventArray")
    out = np.zeros(max(n, 1), EVENT_DTYPE)
    lib.orc_decode_event_array(buf.ctypes.data, buf.size, out.ctypes.data, n, C.byref(h), C.byref(w))
    return out[:n], int(h.value), int(w.value)


def denoise_events(ev, idx, width, height, max_num):
    lib = load()
    ev = np.ascontiguousarray(ev, dtype=EVENT_DTYPE)
    idx = np.ascontiguousarray(idx, np.uint32)
    out = np.empty(max(idx.shape[0], 1), np.uint32)
    n = lib.orc_denoise_events(ev.ctypes.data, idx.ctypes.data, idx.shape[0], width, height, int(max_num),
                               out.ctypes.data)
    return out[:n]


def unpack_normal_equations(v):
    """28 sums (upper triangle of J^T J row-major, J^T f, |f|^2) -> (H 6x6 symmetric, b, cost)"""
    H = np.zeros((6, 6))
    n = 0
    for i in range(6):
        for j in range(i, 6):
            H[i, j] = H[j, i] = v[n]
            n += 1
    return H, np.array(v[21:27]), float(v[27])


class OracleTracker:
    """RegProblemLM residual / Jacobian evaluation (esvo_core/src/core/RegProblemLM.cpp), SURVEY.md section 8(f).1."""

    def __init__(self, rig, fast=False):
        self.lib = load(fast)
        self.W, self.H = rig.width, rig.height
        self._cl = rig.left.as_struct()
        self.h = self.lib.orc_tracker_create(C.addressof(self._cl))

    def __del__(self):
        if getattr(self, "h", None):
            self.lib.orc_tracker_destroy(self.h)
            self.h = None

    def set_current(self, ts_left, kernel_size=5):
        img = np.ascontiguousarray(ts_left, np.uint8)
        if self.lib.orc_tracker_set_current(self.h, img.ctypes.data, int(kernel_size)) != 0:
            raise ValueError("unsupported kernel size")

    def images(self):
        neg = np.empty((self.H, self.W), np.uint8)
        du = np.empty((self.H, self.W), np.int16)
        dv = np.empty((self.H, self.W), np.int16)
        self.lib.orc_tracker_get_images(self.h, neg.ctypes.data, du.ctypes.data, dv.ctypes.data)
        return neg, du, dv

    def set_reference(self, xyz_world, T_world_ref):
        xyz = np.ascontiguousarray(xyz_world, np.float32).reshape(-1, 3)
        T = np.ascontiguousarray(T_world_ref, np.float64).reshape(16)
        self.n = xyz.shape[0]
        self.lib.orc_tracker_set_reference(self.h, xyz.ctypes.data, self.n, T.ctypes.data)

    def residuals(self, T_left_ref, offset, count, huber=True, huber_threshold=50.0):
        T = np.ascontiguousarray(T_left_ref, np.float64).reshape(16)
        out = np.empty(count, np.float64)
        n = self.lib.orc_tracker_residuals(self.h, T.ctypes.data, int(offset), int(count), int(bool(huber)), float(huber_threshold),
                                           out.ctypes.data)
        return out[:n]

    def jacobian(self, R, t, offset, count):
        R = np.ascontiguousarray(R, np.float64).reshape(9)
        t = np.ascontiguousarray(t, np.float64).reshape(3)
        out = np.empty(6 * count, np.float64)
        n = self.lib.orc_tracker_jacobian(self.h, R.ctypes.data, t.ctypes.data, int(offset), int(count), out.ctypes.data)
        return out[:6 * n].reshape(6, n).T  # (n, 6); the C layout is column-major like Eigen's

    def normal_equations(self, R, t, offset, count, huber=True, huber_threshold=50.0):
        """(H 6x6, b 6, |f|^2, n) of one tracker iteration at (R, t), summed in the device's order"""
        R = np.ascontiguousarray(R, np.float64).reshape(9)
        t = np.ascontiguousarray(t, np.float64).reshape(3)
        out = np.zeros(28, np.float64)
        n = self.lib.orc_tracker_normal_equations(self.h, R.ctypes.data, t.ctypes.data, int(offset), int(count), int(bool(huber)),
                                                  float(huber_threshold), out.ctypes.data)
        return unpack_normal_equations(out) + (int(n),)


class OracleMapper:
    """esvo_Mapping / esvo_MVStereo mapper state (EventBM + DepthProblemSolver + DepthFusion + ...)."""

    def __init__(self, params: ParamsStruct, rig, fast=False):
        self.lib = load(fast)
        self.rig, self.params = rig, params
        self._cl, self._cr = rig.left.as_struct(), rig.right.as_struct()
        self.h = self.lib.orc_mapper_create(C.addressof(params), C.addressof(self._cl), C.addressof(self._cr))
        self.W, self.H = rig.width, rig.height
        self._poses = None

    def __del__(self):
        if getattr(self, "h", None):
            self.lib.orc_mapper_destroy(self.h)
            self.h = None

    def reset(self):
        self.lib.orc_mapper_reset(self.h)

    def set_params(self, params):
        self.params = params
        self.lib.orc_mapper_set_params(self.h, C.addressof(params))

    def set_threads(self, n):
        self.lib.orc_mapper_set_threads(self.h, int(n))

    def set_mode(self, bm_exact_int=False, lm_canonical=False):
        self.lib.orc_mapper_set_mode(self.h, int(bm_exact_int), int(lm_canonical))

    @property
    def baseline(self):
        return self.lib.orc_mapper_baseline(self.h)

    def set_observation(self, t_ns, ts_left, ts_right, T_world_cam):
        l = np.ascontiguousarray(ts_left, np.uint8)
        r = np.ascontiguousarray(ts_right, np.uint8)
        T = np.ascontiguousarray(T_world_cam, np.float64).reshape(16)
        self.lib.orc_mapper_set_observation(self.h, int(t_ns), l.ctypes.data, r.ctypes.data, T.ctypes.data)

    def set_poses(self, stamps, poses):
        st = np.ascontiguousarray(stamps, np.uint64)
        T = np.ascontiguousarray(poses, np.float64).reshape(-1, 16)
        self._poses = T
        self.lib.orc_mapper_set_poses(self.h, st.ctypes.data, T.ctypes.data, st.shape[0])

    def match(self, ev):
        ev = np.ascontiguousarray(ev, dtype=EVENT_DTYPE)
        out = np.zeros(max(ev.shape[0], 1), MATCH_DTYPE)
        n = self.lib.orc_mapper_match(self.h, ev.ctypes.data, ev.shape[0], out.ctypes.data, out.shape[0])
        return out[:n]

    def match_costs(self, ev_one, exact_int=False):
        ev = np.ascontiguousarray(ev_one, dtype=EVENT_DTYPE).reshape(1)
        nd = self.params.bm_max_disparity - self.params.bm_min_disparity + 1
        costs = np.empty(nd, np.float64)
        ok = self.lib.orc_mapper_match_costs(self.h, ev.ctypes.data, costs.ctypes.data, int(exact_int))
        return costs if ok else None

    def refine(self, matches, cull=True, want_info=False):
        m = np.ascontiguousarray(matches, dtype=MATCH_DTYPE)
        out = np.zeros(max(m.shape[0], 1), DEPTH_POINT_DTYPE)
        info = np.zeros((max(m.shape[0], 1), 4), np.float64) if want_info else None
        n = self.lib.orc_mapper_refine(self.h, m.ctypes.data, m.shape[0], int(cull), out.ctypes.data,
                                       out.shape[0], _p(info))
        return (out[:n], info[: m.shape[0]]) if want_info else out[:n]

    def push_frame(self, pts, poses=None):
        pts = np.ascontiguousarray(pts, dtype=DEPTH_POINT_DTYPE)
        T = self._poses if poses is None else np.ascontiguousarray(poses, np.float64).reshape(-1, 16)
        self.lib.orc_mapper_push_frame(self.h, pts.ctypes.data, pts.shape[0], T.ctypes.data, T.shape[0])

    def fuse(self):
        return self.lib.orc_mapper_fuse(self.h)

    def tick(self, ev):
        ev = np.ascontiguousarray(ev, dtype=EVENT_DTYPE)
        return self.lib.orc_mapper_tick(self.h, ev.ctypes.data, ev.shape[0])

    def tick_bm_only(self, ev):
        """esvo_MVStereo::MappingAtTime in PURE_BLOCK_MATCHING mode (esvo_MVStereo.cpp:383-432); returns the match count"""
        ev = np.ascontiguousarray(ev, dtype=EVENT_DTYPE)
        return self.lib.orc_mapper_tick_bm_only(self.h, ev.ctypes.data, ev.shape[0])

    def get_map(self):
        n = self.lib.orc_mapper_map_size(self.h)
        out = np.zeros(max(n, 1), DEPTH_POINT_DTYPE)
        n = self.lib.orc_mapper_get_map(self.h, out.ctypes.data, out.shape[0])
        return out[:n]

    def get_map_cells(self):
        n = self.lib.orc_mapper_map_size(self.h)
        out = np.zeros(max(n, 1), np.int32)
        n = self.lib.orc_mapper_get_map_cells(self.h, out.ctypes.data, out.shape[0])
        return out[:n]

    def get_last_frame(self):
        cap = max(int(self.params.max_events_per_tick), int(self.params.process_event_num) + 1, 1)
        out = np.zeros(cap, DEPTH_POINT_DTYPE)
        n = self.lib.orc_mapper_get_last_frame(self.h, out.ctypes.data, cap)
        if n > cap:   # (the call reports the frame's size whatever the capacity: fetch again rather than hand back a truncated frame)
            out = np.zeros(n, DEPTH_POINT_DTYPE)
            n = self.lib.orc_mapper_get_last_frame(self.h, out.ctypes.data, n)
        return out[:n]

    def get_pointcloud(self):
        n = self.lib.orc_mapper_map_size(self.h)
        out = np.zeros((max(n, 1), 3), np.float32)
        n = self.lib.orc_mapper_get_pointcloud_xyz(self.h, out.ctypes.data, out.shape[0])
        return out[:n]

    def init_sgm(self, ts_left, ts_right, ev, min_points=500):
        """InitializationAtTime (esvo_Mapping.cpp:433-492) on the observation set last; returns (#points, disparity*16 image)"""
        l = np.ascontiguousarray(ts_left, np.uint8)
        r = np.ascontiguousarray(ts_right, np.uint8)
        ev = np.ascontiguousarray(ev, dtype=EVENT_DTYPE)
        disp = np.empty((self.H, self.W), np.int16)
        n = self.lib.orc_mapper_init_sgm(self.h, l.ctypes.data, r.ctypes.data, ev.ctypes.data, ev.shape[0], int(min_points), disp.ctypes.data)
        return int(n), disp

    def debug_image(self, kind, age_max_range=10.0):
        """Visualization::plot_map as publishMappingResults calls it; kind: 0 InvDepth, 1 StdVar, 2 Cost, 3 Age -> BGR image"""
        out = np.empty((self.H, self.W, 3), np.uint8)
        self.lib.orc_mapper_debug_image(self.h, int(kind), float(age_max_range), out.ctypes.data)
        return out

    def get_pointcloud_near(self, visualize_range):
        n = self.lib.orc_mapper_map_size(self.h)
        out = np.zeros((max(n, 1), 3), np.float32)
        n = self.lib.orc_mapper_get_pointcloud_near_xyz(self.h, float(visualize_range), out.ctypes.data, out.shape[0])
        return out[:n]

    def eval_residual(self, x_left, pose_idx, rho):
        x = np.ascontiguousarray(x_left, np.float64)
        f = np.empty(self.params.patch_size_x * self.params.patch_size_y, np.float64)
        ok = self.lib.orc_mapper_eval_residual(self.h, x.ctypes.data, int(pose_idx), float(rho), f.ctypes.data)
        return f, ok

    def counters(self):
        c = np.zeros(8, np.uint64)
        self.lib.orc_mapper_counters(self.h, c.ctypes.data)
        return dict(window_frames=int(c[0]), window_points=int(c[1]), replace=int(c[2]),
                    replace_displaced=int(c[3]), max_scale_iters=int(c[4]), lm_evals=int(c[5]))


def zncc_cost(l, r, exact_int=False):
    lib = load()
    l = np.ascontiguousarray(l, np.float64)
    r = np.ascontiguousarray(r, np.float64)
    return lib.orc_zncc_cost(l.ctypes.data, r.ctypes.data, l.shape[1], l.shape[0], int(exact_int))


def sgbm(left, right, num_disp=48, block=11, uniqueness=11):
    """cv::StereoSGBM::compute restated (MODE_SGBM, P1 = 8 b^2, P2 = 32 b^2): int16 disparity * 16"""
    l = np.ascontiguousarray(left, np.uint8)
    r = np.ascontiguousarray(right, np.uint8)
    out = np.empty(l.shape, np.int16)
    load().orc_sgbm_compute(l.ctypes.data, r.ctypes.data, l.shape[1], l.shape[0], int(num_disp), int(block), 8 * block * block,
                            32 * block * block, int(uniqueness), out.ctypes.data)
    return out


def select_events_sgm(ev, t_ns, half_slice, max_num):
    lib = load()
    ev = np.ascontiguousarray(ev, dtype=EVENT_DTYPE)
    cap = min(max_num + 1, ev.shape[0]) + 1
    idx = np.empty(cap, np.uint32)
    n = lib.orc_select_events_sgm(ev.ctypes.data, ev.shape[0], int(t_ns), float(half_slice), int(max_num), idx.ctypes.data, cap)
    return idx[:n]


def jet_bgr():
    out = np.empty((256, 3), np.uint8)
    load().orc_jet_bgr(out.ctypes.data)
    return out


def voxel_filter(xyz, leaf):
    xyz = np.ascontiguousarray(xyz, np.float32).reshape(-1, 3)
    out = np.empty_like(xyz) if len(xyz) else np.empty((1, 3), np.float32)
    n = load().orc_voxel_filter(xyz.ctypes.data, xyz.shape[0], float(leaf), out.ctypes.data)
    return out[:n].copy()


def abi_sizes():
    lib = load()
    out = (C.c_size_t * 8)()
    lib.orc_abi_sizes(out)
    return list(out)
