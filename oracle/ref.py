"""ctypes bindings of oracle/_ref/libesvo_ref.so (TEST INFRASTRUCTURE ONLY).

libesvo_ref.so is the REFERENCE's own mapper code -- esvo_core/src/{container/DepthPoint, container/CameraSystem,
core/EventBM, core/DepthProblem, core/DepthProblemSolver, core/DepthFusion, core/DepthRegularization,
core/RegProblemLM, tools/cayley, container/ResidualItem}.cpp and SmartGrid.h, compiled unmodified where they lie under /root/reference against the stand-in headers of
oracle/ref_shim/ (oracle/Makefile, target `ref`).  It exists only in the build container (no /root/reference on the
GPU box): tests/golden/make_ref_fixtures.py records its outputs, tests/test_ref_pin.py compares the oracle with them.

RefMapper has the interface of oracle.OracleMapper, so one script drives both.  RefTracker (RegProblemLM), RefTS
(libesvo_ref_ts.so: the Time-Surface node class) and RefNode (libesvo_ref_node.so: the esvo_Mapping node object, driven
through its own callbacks) are built by the same make target.
"""
import ctypes as C
import os
import subprocess
import tempfile

import numpy as np

from esvo_amd.abi import DEPTH_POINT_DTYPE, EVENT_DTYPE, MATCH_DTYPE, ParamsStruct

_HERE = os.path.dirname(os.path.abspath(__file__))
REFERENCE = os.environ.get("ESVO_REFERENCE", "/root/reference")
_LIB = os.path.join(_HERE, "_ref", "libesvo_ref.so")


def available():
    """True when the library is built or can be built here (the reference tree is present)."""
    return os.path.exists(_LIB) or os.path.isdir(os.path.join(REFERENCE, "esvo_core", "src"))


def build():
    subprocess.check_call(["make", "-C", _HERE, "ref", "ref_o3", "REF=" + REFERENCE], stdout=subprocess.DEVNULL)
    # the node objects with the product's binding attached link against the HIP library: only once that is built
    if os.path.exists(os.path.join(_HERE, "..", "esvo_amd", "csrc", "libesvo_hip.so")):
        subprocess.check_call(["make", "-C", _HERE, "ref_hip", "REF=" + REFERENCE], stdout=subprocess.DEVNULL)
    return _LIB


_lib = None


def load():
    global _lib
    if _lib is not None:
        return _lib
    if os.path.isdir(os.path.join(REFERENCE, "esvo_core", "src")):
        build()  # make tracks the dependencies
    lib = C.CDLL(_LIB)
    vp, u64, sz, dbl, i32, u32 = C.c_void_p, C.c_uint64, C.c_size_t, C.c_double, C.c_int, C.c_uint32
    if hasattr(lib, "ref_tracker_create"):
        lib.ref_tracker_create.restype = vp
        lib.ref_tracker_create.argtypes = [C.c_char_p, vp, vp, i32, dbl, sz]
        lib.ref_tracker_destroy.argtypes = [vp]
        lib.ref_tracker_set_problem.argtypes = [vp, vp, vp, vp, vp, sz, vp, vp, C.c_uint, vp]
        lib.ref_tracker_num_points.restype = sz
        lib.ref_tracker_num_points.argtypes = [vp]
        lib.ref_tracker_relative_pose.argtypes = [vp, vp, vp]
        lib.ref_tracker_residuals.restype = sz
        lib.ref_tracker_residuals.argtypes = [vp, sz, sz, vp, vp, vp]
        lib.ref_tracker_jacobian.restype = sz
        lib.ref_tracker_jacobian.argtypes = [vp, sz, sz, vp]
        if hasattr(lib, "ref_tracker_solve"):
            lib.ref_tracker_solve.restype = i32
            lib.ref_tracker_solve.argtypes = [vp, sz, sz, vp, vp, vp, vp, vp]
    lib.ref_mapper_create.restype = vp
    lib.ref_mapper_create.argtypes = [C.c_char_p, vp, vp, vp]
    lib.ref_mapper_destroy.argtypes = [vp]
    lib.ref_mapper_reset.argtypes = [vp]
    lib.ref_mapper_set_params.argtypes = [vp, vp]
    lib.ref_mapper_baseline.restype = dbl
    lib.ref_mapper_baseline.argtypes = [vp]
    lib.ref_cam2world.argtypes = [vp, vp, dbl, vp]
    lib.ref_world2cam.argtypes = [vp, i32, vp, vp]
    lib.ref_get_lut_mask.argtypes = [vp, vp, vp]
    lib.ref_mapper_set_observation.argtypes = [vp, u64, vp, vp, vp]
    lib.ref_mapper_set_poses.argtypes = [vp, vp, vp, sz]
    lib.ref_mapper_match.restype = sz
    lib.ref_mapper_match.argtypes = [vp, vp, sz, vp, sz]
    lib.ref_mapper_refine.restype = sz
    lib.ref_mapper_refine.argtypes = [vp, vp, sz, i32, vp, sz]
    lib.ref_mapper_eval_residual.restype = i32
    lib.ref_mapper_eval_residual.argtypes = [vp, vp, u32, dbl, vp]
    lib.ref_mapper_solve_single.restype = i32
    lib.ref_mapper_solve_single.argtypes = [vp, vp, u32, dbl, vp]
    lib.ref_zncc_cost.restype = dbl
    lib.ref_zncc_cost.argtypes = [vp, vp, i32, i32]
    lib.ref_mapper_push_frame.argtypes = [vp, vp, sz, vp, sz]
    lib.ref_mapper_init_from_disparity.restype = sz
    lib.ref_mapper_init_from_disparity.argtypes = [vp, vp, vp, sz, sz]
    lib.ref_mapper_fuse.restype = sz
    lib.ref_mapper_fuse.argtypes = [vp]
    lib.ref_mapper_tick.restype = sz
    lib.ref_mapper_tick.argtypes = [vp, vp, sz]
    lib.ref_mapper_map_size.restype = sz
    lib.ref_mapper_map_size.argtypes = [vp]
    lib.ref_mapper_get_map.restype = sz
    lib.ref_mapper_get_map.argtypes = [vp, vp, sz]
    lib.ref_mapper_get_map_cells.restype = sz
    lib.ref_mapper_get_map_cells.argtypes = [vp, vp, sz]
    lib.ref_mapper_get_last_frame.restype = sz
    lib.ref_mapper_get_last_frame.argtypes = [vp, vp, sz]
    lib.ref_mapper_counters.argtypes = [vp, vp]
    lib.ref_update_student_t.argtypes = [vp, dbl, dbl, dbl, dbl]
    _lib = lib
    return lib


def _yaml_camera(path, name, intr, T_right_left):
    """One camera file in the format CameraSystem::loadCalibInfo reads (CameraSystem.cpp:168-212)."""
    fmt = lambda a: "[" + ", ".join(repr(float(v)) for v in np.asarray(a, np.float64).reshape(-1)) + "]"
    with open(path, "w") as f:
        f.write(f"image_width: {int(intr['width'])}\nimage_height: {int(intr['height'])}\ncamera_name: {name}\n")
        f.write(f"camera_matrix:\n  rows: 3\n  cols: 3\n  data: {fmt(intr['K'])}\n")
        f.write(f"distortion_model: {intr['model']}\n")
        f.write(f"distortion_coefficients:\n  rows: 1\n  cols: 4\n  data: {fmt(np.asarray(intr['D'])[:4])}\n")
        f.write(f"rectification_matrix:\n  rows: 3\n  cols: 3\n  data: {fmt(intr['R'])}\n")
        f.write(f"projection_matrix:\n  rows: 3\n  cols: 4\n  data: {fmt(intr['P'])}\n")
        f.write(f"T_right_left:\n  rows: 3\n  cols: 4\n  data: {fmt(T_right_left)}\n")


def write_calib_dir(rig, path):
    T = np.hstack([np.eye(3), np.array([[-rig.baseline], [0.0], [0.0]])])  # read, never used on the mapper path
    _yaml_camera(os.path.join(path, "left.yaml"), rig.name + "_left", rig.intr_left, T)
    _yaml_camera(os.path.join(path, "right.yaml"), rig.name + "_right", rig.intr_right, T)


def _p(a):
    return None if a is None else a.ctypes.data


class RefMapper:
    """The reference's EventBM + DepthProblemSolver + DepthFusion + DepthRegularization + DepthMap, driven in
    MappingAtTime's order (esvo_Mapping.cpp:261-431)."""

    def __init__(self, params: ParamsStruct, rig):
        self.lib = load()
        self.rig, self.params = rig, params
        self._cl, self._cr = rig.left.as_struct(), rig.right.as_struct()
        with tempfile.TemporaryDirectory() as d:
            write_calib_dir(rig, d)
            self.h = self.lib.ref_mapper_create(d.encode(), C.addressof(params), C.addressof(self._cl), C.addressof(self._cr))
        self.W, self.H = rig.width, rig.height
        self._poses = None

    def __del__(self):
        if getattr(self, "h", None):
            self.lib.ref_mapper_destroy(self.h)
            self.h = None

    def reset(self):
        self.lib.ref_mapper_reset(self.h)

    def init_from_disparity(self, disp16, xy, min_points=500):
        """InitializationAtTime behind the StereoSGBM call (esvo_Mapping.cpp:455-487) + naive_propagation"""
        d = np.ascontiguousarray(disp16, np.int16)
        q = np.ascontiguousarray(xy, np.uint32).reshape(-1, 2)
        return int(self.lib.ref_mapper_init_from_disparity(self.h, _p(d), _p(q), len(q), int(min_points)))

    def set_params(self, params):
        self.params = params
        self.lib.ref_mapper_set_params(self.h, C.addressof(params))

    @property
    def baseline(self):
        return self.lib.ref_mapper_baseline(self.h)

    def cam2world(self, x, inv_depth):
        x = np.ascontiguousarray(x, np.float64)
        p = np.empty(3, np.float64)
        self.lib.ref_cam2world(self.h, x.ctypes.data, float(inv_depth), p.ctypes.data)
        return p

    def world2cam(self, p, right=False):
        p = np.ascontiguousarray(p, np.float64)
        x = np.empty(2, np.float64)
        self.lib.ref_world2cam(self.h, int(right), p.ctypes.data, x.ctypes.data)
        return x

    def lut_mask(self):
        lut = np.empty((self.H, self.W, 2), np.float64)
        mask = np.empty((self.H, self.W), np.int32)
        self.lib.ref_get_lut_mask(self.h, lut.ctypes.data, mask.ctypes.data)
        return lut, mask

    def set_observation(self, t_ns, ts_left, ts_right, T_world_cam):
        l = np.ascontiguousarray(ts_left, np.uint8)
        r = np.ascontiguousarray(ts_right, np.uint8)
        T = np.ascontiguousarray(T_world_cam, np.float64).reshape(16)
        self.lib.ref_mapper_set_observation(self.h, int(t_ns), l.ctypes.data, r.ctypes.data, T.ctypes.data)

    def set_poses(self, stamps, poses):
        st = np.ascontiguousarray(stamps, np.uint64)
        T = np.ascontiguousarray(poses, np.float64).reshape(-1, 16)
        self._poses = T
        self.lib.ref_mapper_set_poses(self.h, st.ctypes.data, T.ctypes.data, st.shape[0])

    def match(self, ev):
        ev = np.ascontiguousarray(ev, dtype=EVENT_DTYPE)
        out = np.zeros(max(ev.shape[0], 1), MATCH_DTYPE)
        n = self.lib.ref_mapper_match(self.h, ev.ctypes.data, ev.shape[0], out.ctypes.data, out.shape[0])
        return out[:n]

    def refine(self, matches, cull=True):
        m = np.ascontiguousarray(matches, dtype=MATCH_DTYPE)
        out = np.zeros(max(m.shape[0], 1), DEPTH_POINT_DTYPE)
        n = self.lib.ref_mapper_refine(self.h, m.ctypes.data, m.shape[0], int(cull), out.ctypes.data, out.shape[0])
        return out[:n]

    def eval_residual(self, x_left, pose_idx, rho):
        x = np.ascontiguousarray(x_left, np.float64)
        f = np.empty(self.params.patch_size_x * self.params.patch_size_y, np.float64)
        ok = self.lib.ref_mapper_eval_residual(self.h, x.ctypes.data, int(pose_idx), float(rho), f.ctypes.data)
        return f, ok

    def solve_single(self, x_left, pose_idx, d_init):
        x = np.ascontiguousarray(x_left, np.float64)
        res = np.zeros(3, np.float64)
        ok = self.lib.ref_mapper_solve_single(self.h, x.ctypes.data, int(pose_idx), float(d_init), res.ctypes.data)
        return res, bool(ok)

    def push_frame(self, pts, poses=None):
        pts = np.ascontiguousarray(pts, dtype=DEPTH_POINT_DTYPE)
        T = self._poses if poses is None else np.ascontiguousarray(poses, np.float64).reshape(-1, 16)
        self.lib.ref_mapper_push_frame(self.h, pts.ctypes.data, pts.shape[0], T.ctypes.data, T.shape[0])

    def fuse(self):
        return self.lib.ref_mapper_fuse(self.h)

    def tick(self, ev):
        ev = np.ascontiguousarray(ev, dtype=EVENT_DTYPE)
        return self.lib.ref_mapper_tick(self.h, ev.ctypes.data, ev.shape[0])

    def get_map(self):
        n = self.lib.ref_mapper_map_size(self.h)
        out = np.zeros(max(n, 1), DEPTH_POINT_DTYPE)
        n = self.lib.ref_mapper_get_map(self.h, out.ctypes.data, out.shape[0])
        return out[:n]

    def get_map_cells(self):
        n = self.lib.ref_mapper_map_size(self.h)
        out = np.zeros(max(n, 1), np.int32)
        n = self.lib.ref_mapper_get_map_cells(self.h, out.ctypes.data, out.shape[0])
        return out[:n]

    def get_last_frame(self):
        cap = max(int(self.params.max_events_per_tick), 1)
        out = np.zeros(cap, DEPTH_POINT_DTYPE)
        n = self.lib.ref_mapper_get_last_frame(self.h, out.ctypes.data, cap)
        return out[:n]

    def counters(self):
        c = np.zeros(8, np.uint64)
        self.lib.ref_mapper_counters(self.h, c.ctypes.data)
        return dict(window_frames=int(c[0]), window_points=int(c[1]), dangling_cells=int(c[2]))


def zncc_cost(l, r):
    lib = load()
    l = np.ascontiguousarray(l, np.float64)
    r = np.ascontiguousarray(r, np.float64)
    return lib.ref_zncc_cost(l.ctypes.data, r.ctypes.data, l.shape[1], l.shape[0])


def update_student_t(state, inv_depth, scale2, variance, nu):
    """DepthPoint::update_studentT on state = (invDepth, scale2, nu, variance, age); returns the new state."""
    lib = load()
    s = np.array(state, np.float64)
    lib.ref_update_student_t(s.ctypes.data, float(inv_depth), float(scale2), float(variance), float(nu))
    return s


class RefTracker:
    """The reference's RegProblemLM (esvo_core/src/core/RegProblemLM.cpp) on injected images: setProblem, operator(), df."""

    def __init__(self, rig, huber=True, huber_threshold=50.0, max_points=2000):
        self.lib = load()
        self.rig = rig
        self._cl, self._cr = rig.left.as_struct(), rig.right.as_struct()
        with tempfile.TemporaryDirectory() as d:
            write_calib_dir(rig, d)
            self.h = self.lib.ref_tracker_create(d.encode(), C.addressof(self._cl), C.addressof(self._cr), int(bool(huber)),
                                                 float(huber_threshold), int(max_points))
        self.W, self.H = rig.width, rig.height

    def __del__(self):
        if getattr(self, "h", None):
            self.lib.ref_tracker_destroy(self.h)
            self.h = None

    def set_problem(self, neg, du, dv, xyz_world, T_world_ref, T_world_left, seed=1):
        """returns (order, R, t): order[i] = index of the input point the reference's shuffle put at position i (only
        the first num_points positions are used), (R, t) = T_ref_left as setProblem derives it"""
        neg = np.ascontiguousarray(neg, np.uint8)
        du, dv = np.ascontiguousarray(du, np.int16), np.ascontiguousarray(dv, np.int16)
        xyz = np.ascontiguousarray(xyz_world, np.float32).reshape(-1, 3)
        Tr = np.ascontiguousarray(T_world_ref, np.float64).reshape(16)
        Tl = np.ascontiguousarray(T_world_left, np.float64).reshape(16)
        order = np.empty(len(xyz), np.uint32)
        self._keep = (neg, du, dv, xyz)
        self.lib.ref_tracker_set_problem(self.h, _p(neg), _p(du), _p(dv), _p(xyz), len(xyz), _p(Tr), _p(Tl), int(seed), _p(order))
        self.n = int(self.lib.ref_tracker_num_points(self.h))
        R, t = np.empty(9), np.empty(3)
        self.lib.ref_tracker_relative_pose(self.h, _p(R), _p(t))
        return order, R.reshape(3, 3), t

    def residuals(self, offset, count, x=None):
        """(fvec, T_warping) of operator()(x) on the batch [offset, offset + count)"""
        x = np.zeros(6) if x is None else np.ascontiguousarray(x, np.float64)
        out = np.empty(count, np.float64)
        T = np.empty(16)
        n = self.lib.ref_tracker_residuals(self.h, int(offset), int(count), _p(x), _p(out), _p(T))
        return out[:n], T.reshape(4, 4)

    def jacobian(self, offset, count):
        out = np.empty(6 * count, np.float64)
        n = self.lib.ref_tracker_jacobian(self.h, int(offset), int(count), _p(out))
        return out[:6 * n].reshape(6, n).T

    def solve(self, batch_size=300, max_iteration=10):
        """RegProblemSolverLM::solve_analytical's loop (RegProblemSolverLM.cpp:148-178) from the pose of the last set_problem:
        (R, t, outer iterations, functor evaluations, last LM status) -- R, t = T_ref_left after the loop"""
        R, t = np.empty(9), np.empty(3)
        it, nfev, st = C.c_size_t(0), C.c_size_t(0), C.c_int(0)
        rc = self.lib.ref_tracker_solve(self.h, int(batch_size), int(max_iteration), _p(R), _p(t), C.byref(it), C.byref(nfev), C.byref(st))
        if rc != 0:
            raise RuntimeError("ref_tracker_solve: ImproperInputParameters")
        return R.reshape(3, 3), t, it.value, nfev.value, st.value


_LIB_TS = os.path.join(_HERE, "_ref", "libesvo_ref_ts.so")
_lib_ts = {}


def load_ts(o3=False):
    """o3: the build at the reference's own -O3 (oracle/Makefile, ref_o3): what bench.py times"""
    if _lib_ts.get(o3) is None:
        if os.path.isdir(os.path.join(REFERENCE, "esvo_core", "src")):
            build()
        lib = C.CDLL(_LIB_TS.replace(".so", "_O3.so") if o3 else _LIB_TS)
        lib.ref_ts_create.restype = C.c_void_p
        lib.ref_ts_create.argtypes = [C.c_int, C.c_int, C.c_double, C.c_int, C.c_int]
        lib.ref_ts_destroy.argtypes = [C.c_void_p]
        lib.ref_ts_push.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t]
        lib.ref_ts_render.argtypes = [C.c_void_p, C.c_uint64, C.c_void_p]
        lib.ref_ts_set_forward.argtypes = [C.c_void_p, C.c_void_p]
        _lib_ts[o3] = lib
    return _lib_ts[o3]


class RefTS:
    """The reference's Time-Surface node class (esvo_time_surface/src/TimeSurface.cpp): eventsCallback + createTimeSurfaceAtTime
    (BACKWARD mode).  render() returns the f64 image the node hands to cv::Mat::convertTo(CV_8U); OpenCV's rounding, median
    filter and remap are not part of the build."""

    def __init__(self, width, height, decay_ms=30.0, ignore_polarity=True, queue_len=20, o3=False):
        self.lib = load_ts(o3)
        self.W, self.H = width, height
        self.h = self.lib.ref_ts_create(width, height, float(decay_ms), int(bool(ignore_polarity)), int(queue_len))

    def __del__(self):
        if getattr(self, "h", None):
            self.lib.ref_ts_destroy(self.h)
            self.h = None

    def push(self, ev):
        ev = np.ascontiguousarray(ev, dtype=EVENT_DTYPE)
        self.lib.ref_ts_push(self.h, _p(ev), len(ev))

    def set_forward(self, rect_lut):
        """FORWARD mode (TimeSurface.cpp:85-116) with the given rectified pixel positions (cv::undistortPoints' output)"""
        lut = np.ascontiguousarray(rect_lut, np.float32).reshape(self.H * self.W, 2)
        self._lut = lut
        self.lib.ref_ts_set_forward(self.h, _p(lut))

    def render(self, t_ns):
        out = np.empty((self.H, self.W), np.float64)
        self.lib.ref_ts_render(self.h, int(t_ns), _p(out))
        return out


_lib_node = {}
_POSE_FN = C.CFUNCTYPE(C.c_int, C.c_ulonglong, C.POINTER(C.c_double))


def load_node(mvstereo=False, hip=False, o3=False):
    """oracle/_ref/libesvo_ref_node.so: esvo_core/src/esvo_Mapping.cpp (the mapper NODE) + the mapper sources, against the
    inert ROS / tf / cv_bridge / pcl stand-ins of oracle/ref_shim_node/ (oracle/ref_harness_node.cpp);
    libesvo_ref_mvstereo.so: the same entry points around esvo_core/src/esvo_MVStereo.cpp."""
    key = (mvstereo, hip, o3)
    if _lib_node.get(key) is None:
        if os.path.isdir(os.path.join(REFERENCE, "esvo_core", "src")):
            build()
        # hip: the same objects + include/esvo_hip_mapping_node.hpp, linked with esvo_amd/csrc/libesvo_hip.so
        name = ("libesvo_ref_mvstereo" if mvstereo else "libesvo_ref_node") + ("_hip.so" if hip else ".so")
        if o3:  # esvo_Mapping only, at the reference's own -O3 (bench.py's timed baseline)
            assert not mvstereo and not hip
            name = "libesvo_ref_node_O3.so"
        lib = C.CDLL(os.path.join(_HERE, "_ref", name))
        vp, u64, sz = C.c_void_p, C.c_uint64, C.c_size_t
        lib.ref_node_create.restype = vp
        lib.ref_node_create.argtypes = [C.c_char_p, vp, vp, vp]
        lib.ref_node_destroy.argtypes = [vp]
        lib.ref_node_set_pose_fn.argtypes = [_POSE_FN]
        lib.ref_node_push_events.argtypes = [vp, vp, sz]
        lib.ref_node_push_time_surfaces.argtypes = [vp, u64, vp, vp]
        lib.ref_node_data_transferring.restype = C.c_int
        lib.ref_node_data_transferring.argtypes = [vp]
        lib.ref_node_obs_time.restype = u64
        lib.ref_node_obs_time.argtypes = [vp]
        for f in ("ref_node_selected_events", "ref_node_matched_events", "ref_node_sgm_events", "ref_node_window", "ref_node_get_map",
                  "ref_node_newest_frame"):
            getattr(lib, f).restype = sz
            getattr(lib, f).argtypes = [vp, vp, sz]
        lib.ref_node_pose_table.restype = sz
        lib.ref_node_pose_table.argtypes = [vp, vp, vp, sz]
        lib.ref_node_mapping_at_time.argtypes = [vp]
        if not mvstereo:
            lib.ref_node_pointcloud.restype = sz
            lib.ref_node_pointcloud.argtypes = [vp, C.c_int, vp, sz]
        lib.ref_node_set_status.argtypes = [vp, C.c_char_p]
        lib.ref_node_preset_param.argtypes = [C.c_char_p, C.c_char_p]
        lib.ref_node_initialization_at_time.restype = C.c_int
        lib.ref_node_initialization_at_time.argtypes = [vp, vp]
        if hip:
            lib.ref_node_hip_error.restype = C.c_char_p
            lib.ref_node_hip_attach.restype = C.c_int
            lib.ref_node_hip_attach.argtypes = [vp, vp, vp, vp, C.c_int]
            lib.ref_node_hip_mapping_at_time.restype = C.c_int
            lib.ref_node_hip_mapping_at_time.argtypes = [vp]
            for f in ("ref_node_hip_matched_events", "ref_node_hip_newest_frame", "ref_node_hip_get_map"):
                getattr(lib, f).restype = sz
                getattr(lib, f).argtypes = [vp, vp, sz]
        _lib_node[key] = lib
    return _lib_node[key]


class RefNode:
    """The reference's mapper node object (esvo_Mapping): events enter through eventsCallback, Time-Surface pairs through
    timeSurfaceCallback, poses through the tf stand-in (pose(t_ns) -> 4x4 T_world_cam or None); tick() = dataTransferring +
    MappingAtTime as MappingLoop calls them (esvo_Mapping.cpp:146-259) without the threads, the rate and the publishers."""

    def __init__(self, params, rig, pose, extra=None, mvstereo=False, hip=False, o3=False):
        self.lib = load_node(mvstereo, hip, o3)   # mvstereo: the esvo_MVStereo node object (BM_PLUS_ESTIMATION) instead of esvo_Mapping
        self.rig = rig
        for k, v in (extra or {}).items():   # node parameters the POD has no field for, e.g. INIT_SGM_DP_NUM_THRESHOLD
            self.lib.ref_node_preset_param(k.encode(), str(v).encode())

        def cb(t_ns, out):
            T = pose(int(t_ns))
            if T is None:
                return 0
            T = np.asarray(T, np.float64).reshape(16)
            for i in range(16):
                out[i] = T[i]
            return 1

        self._cb = _POSE_FN(cb)  # kept alive with the object
        self.lib.ref_node_set_pose_fn(self._cb)
        cl, cr = rig.left.as_struct(), rig.right.as_struct()
        with tempfile.TemporaryDirectory() as d:
            write_calib_dir(rig, d)
            self.h = self.lib.ref_node_create(d.encode(), C.addressof(params), C.addressof(cl), C.addressof(cr))
        self._keep = (cl, cr)
        self._last_ts = 0
        if hip:
            # the product's binding takes over MappingAtTime of THIS node object; the observations the tests hand to the node
            # are smoothed already (the stand-in OpenCV has no GaussianBlur), so the device must not smooth again
            import copy
            ph = copy.copy(params)
            ph.smooth_time_surface = 0
            rc = self.lib.ref_node_hip_attach(self.h, C.addressof(ph), C.addressof(cl), C.addressof(cr), 0)
            if rc:
                raise RuntimeError(f"esvo_hip binding: {rc}: {self.lib.ref_node_hip_error().decode(errors='replace')}")

    def __del__(self):
        if getattr(self, "h", None):
            self.lib.ref_node_destroy(self.h)
            self.h = None

    def push_events(self, ev):
        ev = np.ascontiguousarray(ev, dtype=EVENT_DTYPE)
        self.lib.ref_node_push_events(self.h, _p(ev), len(ev))

    def push_time_surfaces(self, t_ns, left, right):
        left, right = np.ascontiguousarray(left, np.uint8), np.ascontiguousarray(right, np.uint8)
        self.lib.ref_node_push_time_surfaces(self.h, int(t_ns), _p(left), _p(right))
        self._last_ts = int(t_ns)

    def push_observation(self, t_ns, left, right, step_ns=100_000):
        """Time-Surface history so that the pair at t_ns is the SECOND newest (dataTransferring observes that one,
        esvo_Mapping.cpp:503-506) behind ten older ones (the node waits for a history of > 10, :497)."""
        for tt in [t_ns - (11 - i) * step_ns for i in range(1, 11)] + [t_ns, t_ns + step_ns]:
            if tt > self._last_ts:
                self.push_time_surfaces(tt, left, right)

    def data_transferring(self):
        return bool(self.lib.ref_node_data_transferring(self.h))

    def obs_time(self):
        return int(self.lib.ref_node_obs_time(self.h))

    def _indices(self, fn, cap=1 << 20):
        idx = np.empty(cap, np.uint32)
        n = fn(self.h, _p(idx), cap)
        return idx[:n].copy()

    def selected_events(self):
        return self._indices(self.lib.ref_node_selected_events)

    def matched_events(self):
        return self._indices(self.lib.ref_node_matched_events)

    def pose_table(self, cap=4096):
        stamps, poses = np.empty(cap, np.uint64), np.empty((cap, 4, 4), np.float64)
        n = self.lib.ref_node_pose_table(self.h, _p(stamps), _p(poses), cap)
        return stamps[:n].copy(), poses[:n].copy()

    def mapping_at_time(self):
        self.lib.ref_node_mapping_at_time(self.h)

    def set_status(self, status):
        """ESVO_System_Status_ and the /ESVO_SYSTEM_STATUS parameter: "INITIALIZATION" or "WORKING" """
        self.lib.ref_node_set_status(self.h, status.encode())

    def sgm_events(self):
        return self._indices(self.lib.ref_node_sgm_events)

    def initialization_at_time(self, disp16):
        """InitializationAtTime (esvo_Mapping.cpp:433-492) with disp16 standing in for StereoSGBM::compute's output"""
        d = np.ascontiguousarray(disp16, np.int16)
        return bool(self.lib.ref_node_initialization_at_time(self.h, _p(d)))

    def hip_mapping_at_time(self):
        """MappingAtTime through include/esvo_hip_mapping_node.hpp (device) on what dataTransferring loaded"""
        rc = self.lib.ref_node_hip_mapping_at_time(self.h)
        if rc:
            raise RuntimeError(f"esvo_hip binding: {rc}: {self.lib.ref_node_hip_error().decode(errors='replace')}")

    def hip_matched_events(self):
        return self._indices(self.lib.ref_node_hip_matched_events)

    def hip_newest_frame(self):
        return self._points(self.lib.ref_node_hip_newest_frame)

    def hip_get_map(self):
        return self._points(self.lib.ref_node_hip_get_map)

    def window(self):
        return self._indices(self.lib.ref_node_window, 256).tolist()

    def _points(self, fn):
        out = np.zeros(self.rig.width * self.rig.height, DEPTH_POINT_DTYPE)
        n = fn(self.h, _p(out), len(out))
        return out[:n].copy()

    def pointcloud(self, near=False):
        """publishPointCloud (esvo_Mapping.cpp:909-978) on the current DepthFrame: pc_ (the tracker's reference cloud) or
        pc_near_ (|p_cam| < visualize_range), float32 xyz, world frame"""
        out = np.zeros((self.rig.width * self.rig.height, 3), np.float32)
        n = self.lib.ref_node_pointcloud(self.h, int(bool(near)), _p(out), len(out))
        return out[:n].copy()

    def newest_frame(self):
        return self._points(self.lib.ref_node_newest_frame)

    def get_map(self):
        return self._points(self.lib.ref_node_get_map)
