"""ctypes bindings of libesvo_hip.so (the C-ABI of include/esvo_hip.h) and a thin Python host
mirror of the reference's call sequence (TimeSurface node + mapper node) for tests/bench.

There is NO fallback: if the HIP extension is missing or no MI355X is visible, loading /
creating a handle raises.  Nothing in this module touches oracle/.
"""
import ctypes as C
import os
import subprocess

import numpy as np

from .abi import (DEPTH_POINT_DTYPE, EVENT_DTYPE, MATCH_DTYPE, CalibStruct, ParamsStruct,
                  StatsStruct)

_CSRC = os.path.join(os.path.dirname(os.path.abspath(__file__)), "csrc")
# ESVO_HIP_LIB: another build of the same library (A/B measurements of kernel variants, tools/ab_build.py); never a fallback
_LIB_PATH = os.environ.get("ESVO_HIP_LIB") or os.path.join(_CSRC, "libesvo_hip.so")
_SOURCES = ["api_core.hip", "api_ts.hip", "api_map.hip", "api_comm.hip", "api_bag.hip", "api_track.hip", "scan.hip", "kernels_ts.hip", "kernels_bm.hip", "kernels_lm.hip", "kernels_lm_any.hip", "kernels_fuse.hip", "kernels_shard.hip", "kernels_track.hip", "kernels_viz.hip", "kernels_sgm.hip"]
HIPCC_FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fPIC", "-shared",
               "-Wno-unused-value", "-Wno-unused-result", "-ldl"]

SYMBOLS = [
    "esvo_default_params", "esvo_create", "esvo_destroy", "esvo_reset", "esvo_set_params", "esvo_last_error",
    "esvo_set_stream", "esvo_synchronize", "esvo_ts_push_events", "esvo_ts_push_events_async", "esvo_ts_push_wait", "esvo_host_alloc", "esvo_host_free", "esvo_ts_push_event_array", "esvo_ts_render", "esvo_ts_render_forward", "esvo_map_set_observation",
    "esvo_map_match", "esvo_map_set_poses", "esvo_map_refine", "esvo_map_push_frame", "esvo_map_fuse",
    "esvo_map_tick", "esvo_map_tick_bm_only", "esvo_map_fuse_matches_naive", "esvo_map_tick_resident", "esvo_map_get_depth_points", "esvo_map_get_committed", "esvo_map_get_pointcloud_xyz", "esvo_map_get_last_frame",
    "esvo_get_stats", "esvo_shard_set_band", "esvo_shard_set_routing", "esvo_shard_get_rows", "esvo_shard_exchange", "esvo_shard_tick_phase", "esvo_abi_sizes",
    "esvo_map_front", "esvo_map_front_frame", "esvo_map_push_frame_device", "esvo_map_fuse_async",
    "esvo_track_set_current", "esvo_track_get_images", "esvo_track_set_reference", "esvo_track_residuals", "esvo_track_jacobian",
    "esvo_track_normal_equations", "esvo_track_normal_equations_batch", "esvo_track_register",
    "esvo_map_init_sgm",
    "esvo_bag_open", "esvo_bag_close", "esvo_bag_last_error", "esvo_bag_next_event_array", "esvo_ts_push_bag",
    "esvo_map_get_debug_images", "esvo_map_get_pointcloud_near_xyz", "esvo_voxel_filter_xyz", "esvo_map_save_depth_map",
    "esvo_comm_unique_id", "esvo_comm_rccl_info", "esvo_comm_init", "esvo_comm_init_callbacks", "esvo_comm_destroy", "esvo_comm_owns_next_tick",
    "esvo_comm_tick", "esvo_comm_tick_resident", "esvo_comm_get_stats", "esvo_comm_flush", "esvo_comm_newest_map", "esvo_comm_shard_tick", "esvo_comm_gather_map", "esvo_comm_gather_pointcloud_xyz", "esvo_comm_gather_ts",
]

ALL_GATHER_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p)


class EsvoError(RuntimeError):
    """a failed C call; `code` is its esvo_status_t (None for errors raised on the Python side)"""

    def __init__(self, msg, code=None):
        super().__init__(msg)
        self.code = code


ERR_HALO = -7   # ESVO_ERR_HALO (include/esvo_hip.h)
ESVO_AGAIN = 1  # esvo_shard_tick_phase: exchange, then the same phase once more


_PERTURBED_PATH = os.path.join(_CSRC, "libesvo_hip_perturbed.so")


def build(force=False, verbose=False, perturbed=False):
    """hipcc cross-compiles the extension for gfx950 in-tree (works without a GPU): one object per source file (in parallel,
    only the files that changed), then one link.
    perturbed=True additionally links libesvo_hip_perturbed.so: the same library with -DESVO_PERTURB_ONE_ULP, i.e. the depth
    of every eighth solver slot's point off by one unit in the last place (kernels_lm.hip) -- never loaded by the product; tests/test_gpu_bench_parity.py
    points ESVO_HIP_LIB at it to show that bench.py's `parity.oracle_equal` notices a single flipped bit."""
    from concurrent.futures import ThreadPoolExecutor
    inc = os.path.join(_CSRC, "..", "..", "include")
    headers = [os.path.join(_CSRC, f) for f in sorted(os.listdir(_CSRC)) if f.endswith(".hpp")] + [
        os.path.join(inc, "esvo_hip.h"), os.path.join(inc, "esvo_hip.hpp")]
    hdr_time = max(os.path.getmtime(h) for h in headers)
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    extra = os.environ.get("ESVO_EXTRA_HIPCC_FLAGS", "").split()  # A/B experiments only
    lib_path = os.path.join(_CSRC, "libesvo_hip.so")  # always: ESVO_HIP_LIB names a library to LOAD (an A/B build), never one to write
    objdir = os.path.join(_CSRC, "build" + ("_" + str(abs(hash(" ".join(extra))) % 100000) if extra else ""))
    os.makedirs(objdir, exist_ok=True)
    cflags = [f for f in HIPCC_FLAGS if f not in ("-shared", "-ldl")] + extra + ["-I", inc, "-c"]

    def compile_one(job):
        src, obj, defs = job
        if not force and os.path.exists(obj) and os.path.getmtime(obj) >= max(os.path.getmtime(src), hdr_time):
            return False
        cmd = [hipcc] + cflags + defs + ["-o", obj, src]
        if verbose:
            print(" ".join(cmd))
        subprocess.check_call(cmd)
        return True

    jobs = [(os.path.join(_CSRC, s), os.path.join(objdir, s[:-4] + ".o"), []) for s in _SOURCES]
    if perturbed:
        jobs.append((os.path.join(_CSRC, "kernels_lm.hip"), os.path.join(objdir, "kernels_lm_perturbed.o"), ["-DESVO_PERTURB_ONE_ULP"]))
    with ThreadPoolExecutor(max_workers=min(len(jobs), os.cpu_count() or 4)) as ex:
        rebuilt = list(ex.map(compile_one, jobs))
    objs = [j[1] for j in jobs[:len(_SOURCES)]]

    def link(out, obj_list, changed):
        if not changed and os.path.exists(out) and all(os.path.getmtime(out) >= os.path.getmtime(o) for o in obj_list):
            return
        cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", out] + obj_list + ["-ldl"]
        if verbose:
            print(" ".join(cmd))
        subprocess.check_call(cmd)

    link(lib_path, objs, any(rebuilt[:len(_SOURCES)]))
    if perturbed:
        pobjs = [jobs[-1][1] if o.endswith(os.sep + "kernels_lm.o") else o for o in objs]
        link(_PERTURBED_PATH, pobjs, any(rebuilt))
    return lib_path


_lib = None


def load():
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(_LIB_PATH):
        raise EsvoError(f"{_LIB_PATH} is missing: run `python -c 'import __graft_entry__ as g; g.build()'` "
                        "(the ESVO hot path has no CPU fallback)")
    lib = C.CDLL(_LIB_PATH)
    if os.environ.get("ESVO_HIP_LIB"):  # an A/B build of an older revision may lack the newest entry points: stub them
        class _Missing:
            argtypes = restype = None

            def __call__(self, *a):
                raise EsvoError("entry point missing in the ESVO_HIP_LIB build")
        for s in SYMBOLS:
            if not hasattr(lib, s):
                setattr(lib, s, _Missing())
    vp, u64, sz, i32 = C.c_void_p, C.c_uint64, C.c_size_t, C.c_int
    psz = C.POINTER(C.c_size_t)
    lib.esvo_default_params.argtypes = [vp]
    lib.esvo_default_params.restype = None
    lib.esvo_create.argtypes = [vp, vp, vp, i32, C.POINTER(vp)]
    lib.esvo_destroy.argtypes = [vp]
    lib.esvo_reset.argtypes = [vp]
    lib.esvo_set_params.argtypes = [vp, vp]
    lib.esvo_last_error.argtypes = [vp]
    lib.esvo_last_error.restype = C.c_char_p
    lib.esvo_set_stream.argtypes = [vp, vp]
    lib.esvo_synchronize.argtypes = [vp]
    lib.esvo_ts_push_events.argtypes = [vp, i32, vp, sz]
    lib.esvo_ts_push_events_async.argtypes = [vp, i32, vp, sz]
    lib.esvo_ts_push_wait.argtypes = [vp, i32]
    lib.esvo_host_alloc.argtypes = [sz, C.POINTER(vp)]
    lib.esvo_host_free.argtypes = [vp]
    lib.esvo_ts_push_event_array.argtypes = [vp, i32, vp, sz, psz]
    lib.esvo_ts_render.argtypes = [vp, i32, u64, vp]
    lib.esvo_ts_render_forward.argtypes = [vp, i32, u64, vp]
    lib.esvo_map_set_observation.argtypes = [vp, u64, vp, vp, vp]
    lib.esvo_map_match.argtypes = [vp, vp, sz, vp, vp, sz, vp, sz, psz]
    lib.esvo_map_set_poses.argtypes = [vp, vp, vp, sz]
    lib.esvo_map_refine.argtypes = [vp, vp, sz, i32, vp, sz, psz]
    lib.esvo_map_push_frame.argtypes = [vp, vp, sz, vp, sz]
    lib.esvo_map_fuse.argtypes = [vp, psz]
    lib.esvo_map_tick.argtypes = [vp, u64, vp, vp, sz]
    lib.esvo_map_tick_resident.argtypes = [vp, u64, vp, vp, vp, sz]
    lib.esvo_map_tick_bm_only.argtypes = [vp, u64, vp, vp, sz]
    lib.esvo_map_fuse_matches_naive.argtypes = [vp, vp, sz, vp, sz]
    lib.esvo_map_get_depth_points.argtypes = [vp, vp, sz, psz]
    lib.esvo_map_get_committed.argtypes = [vp, vp, sz, psz, C.POINTER(C.c_uint64)]
    lib.esvo_map_get_pointcloud_xyz.argtypes = [vp, vp, sz, psz]
    lib.esvo_map_get_last_frame.argtypes = [vp, vp, sz, psz]
    lib.esvo_get_stats.argtypes = [vp, vp]
    lib.esvo_shard_set_band.argtypes = [vp, i32, i32, i32, i32]
    lib.esvo_shard_set_routing.argtypes = [vp, i32, i32]
    lib.esvo_shard_get_rows.argtypes = [vp, vp, vp, vp, vp]
    lib.esvo_shard_exchange.argtypes = [vp, vp, vp, vp]
    lib.esvo_shard_tick_phase.argtypes = [vp, i32, u64, vp, vp, sz]
    lib.esvo_map_front.argtypes = [vp, u64, vp, vp, sz, psz]
    lib.esvo_map_front_frame.argtypes = [vp, vp]
    lib.esvo_map_push_frame_device.argtypes = [vp, vp, sz, vp, sz]
    lib.esvo_map_fuse_async.argtypes = [vp]
    lib.esvo_track_set_current.argtypes = [vp, vp, i32]
    lib.esvo_track_get_images.argtypes = [vp, vp, vp, vp]
    lib.esvo_track_set_reference.argtypes = [vp, vp, sz, vp]
    lib.esvo_track_residuals.argtypes = [vp, vp, sz, sz, i32, C.c_double, vp, psz]
    lib.esvo_track_jacobian.argtypes = [vp, vp, vp, sz, sz, vp, psz]
    lib.esvo_track_normal_equations.argtypes = [vp, vp, vp, sz, sz, i32, C.c_double, vp, vp, C.POINTER(C.c_double), psz]
    lib.esvo_track_normal_equations_batch.argtypes = [vp, i32, vp, vp, sz, sz, i32, C.c_double, vp, vp, vp, psz]
    lib.esvo_track_register.argtypes = [vp, sz, vp, vp, i32, C.c_double, i32, C.c_double, C.POINTER(C.c_double), C.POINTER(C.c_int)]
    lib.esvo_map_init_sgm.argtypes = [vp, vp, vp, sz, psz, vp]
    lib.esvo_bag_open.argtypes = [C.c_char_p, C.POINTER(vp)]
    lib.esvo_bag_close.argtypes = [vp]
    lib.esvo_bag_last_error.argtypes = [vp]
    lib.esvo_bag_next_event_array.argtypes = [vp, C.c_char_p, C.POINTER(vp), psz, C.POINTER(C.c_uint64), C.POINTER(C.c_char_p)]
    lib.esvo_ts_push_bag.argtypes = [vp, i32, vp, C.c_char_p, u64, psz]
    lib.esvo_map_get_debug_images.argtypes = [vp, C.c_double, vp, vp, vp, vp]
    lib.esvo_map_get_pointcloud_near_xyz.argtypes = [vp, C.c_double, vp, sz, psz]
    lib.esvo_voxel_filter_xyz.argtypes = [vp, sz, C.c_float, vp, sz, psz]
    lib.esvo_map_save_depth_map.argtypes = [vp, C.c_char_p, u64, psz]
    lib.esvo_comm_unique_id.argtypes = [vp]
    lib.esvo_comm_init.argtypes = [vp, vp, i32, i32]
    lib.esvo_comm_rccl_info.argtypes = [C.POINTER(C.c_int), C.c_char_p, sz]
    lib.esvo_comm_init_callbacks.argtypes = [vp, i32, i32, ALL_GATHER_FN, vp]
    lib.esvo_comm_destroy.argtypes = [vp]
    lib.esvo_comm_owns_next_tick.argtypes = [vp]
    lib.esvo_comm_tick.argtypes = [vp, u64, vp, vp, vp, sz]
    lib.esvo_comm_tick_resident.argtypes = [vp, u64, vp, vp, vp, sz]
    lib.esvo_comm_get_stats.argtypes = [vp, vp]
    lib.esvo_comm_flush.argtypes = [vp]
    lib.esvo_comm_newest_map.argtypes = [vp, vp, sz, psz, C.POINTER(C.c_longlong)]
    lib.esvo_comm_shard_tick.argtypes = [vp, u64, vp, vp, sz]
    lib.esvo_comm_gather_map.argtypes = [vp, vp, sz, psz]
    lib.esvo_comm_gather_pointcloud_xyz.argtypes = [vp, vp, sz, psz]
    lib.esvo_comm_gather_ts.argtypes = [vp, i32]
    for s in SYMBOLS:
        if s not in ("esvo_default_params", "esvo_last_error", "esvo_abi_sizes", "esvo_bag_last_error"):
            getattr(lib, s).restype = C.c_int
    lib.esvo_abi_sizes.argtypes = [vp]
    lib.esvo_abi_sizes.restype = None
    lib.esvo_bag_last_error.restype = C.c_char_p
    _lib = lib
    return lib


def selftest_division(n=1 << 28, seed=1):
    """device self-test of fdiv.hpp: returns the number of (a, b) pairs where the shared-divisor
    quotient differs from a / b (must be 0)"""
    lib = load()
    bad = C.c_ulonglong(0)
    lib.esvo_selftest_division.argtypes = [C.c_ulonglong, C.c_ulonglong, C.POINTER(C.c_ulonglong)]
    lib.esvo_selftest_division.restype = C.c_int
    rc = lib.esvo_selftest_division(int(n), int(seed), C.byref(bad))
    if rc != 0:
        raise EsvoError(f"selftest failed ({rc}): {lib.esvo_last_error(None).decode(errors='replace')}")
    return bad.value


class BagReader:
    """rosbag format 2.0 reader of the C-ABI (esvo_bag_*): iterates (topic, bag stamp ns, serialised EventArray bytes)"""

    def __init__(self, path):
        self.lib = load()
        b = C.c_void_p()
        rc = self.lib.esvo_bag_open(path.encode(), C.byref(b))
        if rc != 0:
            raise EsvoError(f"esvo_bag_open failed ({rc}): {self.lib.esvo_last_error(None).decode(errors='replace')}")
        self.b = b

    def close(self):
        if getattr(self, "b", None):
            self.lib.esvo_bag_close(self.b)
            self.b = None

    def __del__(self):
        self.close()

    def messages(self, topic=None):
        while True:
            msg, nb, st, tp = C.c_void_p(), C.c_size_t(), C.c_uint64(), C.c_char_p()
            rc = self.lib.esvo_bag_next_event_array(self.b, topic.encode() if topic else None, C.byref(msg), C.byref(nb), C.byref(st), C.byref(tp))
            if rc == 1:
                return
            if rc != 0:
                raise EsvoError(f"bag read failed ({rc}): {self.lib.esvo_bag_last_error(self.b).decode(errors='replace')}")
            yield tp.value.decode(errors='replace'), int(st.value), C.string_at(msg.value, nb.value)


def voxel_filter(xyz, leaf):
    """pcl::VoxelGrid with a cubic leaf (host helper of the C-ABI)"""
    xyz = np.ascontiguousarray(xyz, np.float32).reshape(-1, 3)
    out = np.empty((max(len(xyz), 1), 3), np.float32)
    n = C.c_size_t()
    rc = load().esvo_voxel_filter_xyz(xyz.ctypes.data, xyz.shape[0], float(leaf), out.ctypes.data, out.shape[0], C.byref(n))
    if rc != 0:
        raise EsvoError(f"esvo_voxel_filter_xyz failed ({rc})")
    return out[: n.value].copy()


def comm_unique_id():
    """ncclGetUniqueId through the C-ABI (128 bytes; create on one rank, hand to the others)"""
    buf = (C.c_uint8 * 128)()
    rc = load().esvo_comm_unique_id(buf)
    if rc != 0:
        raise EsvoError(f"esvo_comm_unique_id failed ({rc}): {load().esvo_last_error(None).decode(errors='replace')}")
    return bytes(buf)


def comm_rccl_info():
    """(ncclGetVersion code, path of the RCCL shared object the C library resolved)"""
    v, buf = C.c_int(), C.create_string_buffer(512)
    rc = load().esvo_comm_rccl_info(C.byref(v), buf, 512)
    if rc != 0:
        raise EsvoError(f"esvo_comm_rccl_info failed ({rc}): {load().esvo_last_error(None).decode(errors='replace')}")
    return int(v.value), buf.value.decode(errors="replace")


class PinnedEvents:
    """an esvo_event_t array in pinned host memory (esvo_host_alloc): what a node's message pool would be"""

    def __init__(self, n):
        self.lib = load()
        self.ptr = C.c_void_p()
        rc = self.lib.esvo_host_alloc(max(int(n), 1) * EVENT_DTYPE.itemsize, C.byref(self.ptr))
        if rc:
            raise EsvoError(f"esvo_host_alloc: {rc}")
        buf = (C.c_char * (max(int(n), 1) * EVENT_DTYPE.itemsize)).from_address(self.ptr.value)
        self.array = np.frombuffer(buf, dtype=EVENT_DTYPE, count=int(n))

    def free(self):
        if self.ptr:
            self.array = None
            self.lib.esvo_host_free(self.ptr)
            self.ptr = None


def abi_sizes():
    out = (C.c_size_t * 8)()
    load().esvo_abi_sizes(out)
    return list(out)


def _p(a):
    return None if a is None else a.ctypes.data


class Esvo:
    """One handle = one GPU: TS-left, TS-right and the mapper behind the same device state."""

    def __init__(self, params: ParamsStruct, rig, device=0):
        self.lib = load()
        self.rig, self.params = rig, params
        self.W, self.H = rig.width, rig.height
        self._cl, self._cr = rig.left.as_struct(), rig.right.as_struct()
        h = C.c_void_p()
        rc = self.lib.esvo_create(C.addressof(params), C.addressof(self._cl), C.addressof(self._cr), int(device), C.byref(h))
        if rc != 0:
            raise EsvoError(f"esvo_create failed ({rc}): {self.lib.esvo_last_error(None).decode(errors='replace')}")
        self.h = h
        self._poses = None

    def close(self):
        if getattr(self, "h", None):
            self.lib.esvo_destroy(self.h)
            self.h = None

    def __del__(self):
        self.close()

    def _ck(self, rc):
        if rc != 0:
            raise EsvoError(f"esvo call failed ({rc}): {self.lib.esvo_last_error(self.h).decode(errors='replace')}", code=rc)

    # ---- lifecycle
    def reset(self):
        self._ck(self.lib.esvo_reset(self.h))

    def set_params(self, params):
        self._ck(self.lib.esvo_set_params(self.h, C.addressof(params)))
        self.params = params

    def set_stream(self, stream_ptr):
        self._ck(self.lib.esvo_set_stream(self.h, C.c_void_p(stream_ptr)))

    def synchronize(self):
        self._ck(self.lib.esvo_synchronize(self.h))

    # ---- Time Surface (esvo_time_surface node: eventsCallback / createTimeSurfaceAtTime)
    def ts_push_events(self, cam, ev):
        ev = np.ascontiguousarray(ev, dtype=EVENT_DTYPE)
        self._ck(self.lib.esvo_ts_push_events(self.h, int(cam), ev.ctypes.data, ev.shape[0]))

    def ts_push_events_async(self, cam, ev):
        """enqueue the copy and return; `ev` (ideally pinned: pinned_events) is kept alive here until ts_push_wait(cam)"""
        assert ev.dtype == EVENT_DTYPE and ev.flags["C_CONTIGUOUS"]
        self._async_keep = getattr(self, "_async_keep", {0: [], 1: []})
        self._async_keep[int(cam)].append(ev)
        self._ck(self.lib.esvo_ts_push_events_async(self.h, int(cam), ev.ctypes.data, ev.shape[0]))

    def ts_push_wait(self, cam):
        self._ck(self.lib.esvo_ts_push_wait(self.h, int(cam)))
        if hasattr(self, "_async_keep"):
            self._async_keep[int(cam)].clear()

    def ts_push_event_array(self, cam, msg):
        """stage one serialised dvs_msgs/EventArray (bytes / uint8 array, ROS1 wire format); returns its event count"""
        buf = np.frombuffer(msg, np.uint8) if isinstance(msg, (bytes, bytearray, memoryview)) else np.ascontiguousarray(msg, np.uint8)
        n = C.c_size_t()
        self._ck(self.lib.esvo_ts_push_event_array(self.h, int(cam), buf.ctypes.data, buf.size, C.byref(n)))
        return int(n.value)

    def ts_push_bag(self, cam, bag, topic, until_ns=0):
        """stage the dvs_msgs/EventArray messages of `topic` (bag time < until_ns; 0: all) from a BagReader; returns the event count"""
        n = C.c_size_t()
        self._ck(self.lib.esvo_ts_push_bag(self.h, int(cam), bag.b, topic.encode() if topic else None, int(until_ns), C.byref(n)))
        return int(n.value)

    def ts_render(self, cam, t_ns, download=True):
        out = np.empty((self.H, self.W), np.uint8) if download else None
        self._ck(self.lib.esvo_ts_render(self.h, int(cam), int(t_ns), _p(out)))
        return out

    # ---- mapper, stage-wise (EventBM / DepthProblemSolver / DepthFusion seams)
    def ts_render_forward(self, cam, t_ns, download=True):
        """esvo_ts_render_forward: the camera's Time Surface in the reference's FORWARD mode"""
        out = np.empty((self.H, self.W), np.uint8) if download else None
        self._ck(self.lib.esvo_ts_render_forward(self.h, int(cam), int(t_ns), _p(out)))
        return out

    def set_observation(self, t_ns, ts_left, ts_right, T_world_cam):
        l = None if ts_left is None else np.ascontiguousarray(ts_left, np.uint8)
        r = None if ts_right is None else np.ascontiguousarray(ts_right, np.uint8)
        T = np.ascontiguousarray(T_world_cam, np.float64).reshape(16)
        self._ck(self.lib.esvo_map_set_observation(self.h, int(t_ns), _p(l), _p(r), T.ctypes.data))

    def set_poses(self, stamps, poses):
        st = np.ascontiguousarray(stamps, np.uint64)
        T = np.ascontiguousarray(poses, np.float64).reshape(-1, 16)
        self._poses = T
        self._ck(self.lib.esvo_map_set_poses(self.h, st.ctypes.data, T.ctypes.data, st.shape[0]))

    def match(self, ev, stamps=None, poses=None):
        ev = np.ascontiguousarray(ev, dtype=EVENT_DTYPE)
        if stamps is not None:
            self.set_poses(stamps, poses)
        out = np.zeros(max(ev.shape[0], 1), MATCH_DTYPE)
        n = C.c_size_t(0)
        self._ck(self.lib.esvo_map_match(self.h, ev.ctypes.data, ev.shape[0], None, None, 0, out.ctypes.data,
                                         out.shape[0], C.byref(n)))
        return out[: n.value]

    def refine(self, matches, cull=True):
        m = np.ascontiguousarray(matches, dtype=MATCH_DTYPE)
        out = np.zeros(max(m.shape[0], 1), DEPTH_POINT_DTYPE)
        n = C.c_size_t(0)
        self._ck(self.lib.esvo_map_refine(self.h, m.ctypes.data, m.shape[0], int(cull), out.ctypes.data, out.shape[0],
                                          C.byref(n)))
        return out[: n.value]

    def push_frame(self, pts, poses=None):
        pts = np.ascontiguousarray(pts, dtype=DEPTH_POINT_DTYPE)
        T = self._poses if poses is None else np.ascontiguousarray(poses, np.float64).reshape(-1, 16)
        self._ck(self.lib.esvo_map_push_frame(self.h, pts.ctypes.data, pts.shape[0], T.ctypes.data, T.shape[0]))

    def fuse(self):
        n = C.c_size_t(0)
        self._ck(self.lib.esvo_map_fuse(self.h, C.byref(n)))
        return n.value

    def init_sgm(self, ts_left=None, ts_right=None, min_points=500, want_disp=True):
        """InitializationAtTime (SGM bootstrap) on the observation set last; returns (#points, disparity*16 image or None)"""
        l = None if ts_left is None else np.ascontiguousarray(ts_left, np.uint8)
        r = None if ts_right is None else np.ascontiguousarray(ts_right, np.uint8)
        disp = np.empty((self.H, self.W), np.int16) if want_disp else None
        n = C.c_size_t()
        self._ck(self.lib.esvo_map_init_sgm(self.h, _p(l), _p(r), int(min_points), C.byref(n), _p(disp)))
        return int(n.value), disp

    # ---- mapper, fused tick (MappingAtTime on device-resident data)
    def tick(self, t_ns, stamps, poses):
        st = np.ascontiguousarray(stamps, np.uint64)
        T = np.ascontiguousarray(poses, np.float64).reshape(-1, 16)
        self._ck(self.lib.esvo_map_tick(self.h, int(t_ns), st.ctypes.data, T.ctypes.data, st.shape[0]))

    def tick_bm_only(self, t_ns, stamps, poses):
        """esvo_MVStereo's PURE_BLOCK_MATCHING mode: block matching, vEMP2vDP, naive propagation of the window"""
        st = np.ascontiguousarray(stamps, np.uint64)
        T = np.ascontiguousarray(poses, np.float64).reshape(-1, 16)
        self._ck(self.lib.esvo_map_tick_bm_only(self.h, int(t_ns), st.ctypes.data, T.ctypes.data, st.shape[0]))

    def fuse_matches_naive(self, matches, poses=None):
        """vEMP2vDP + window of maxNumFusionFrames + naive propagation on the given matches (stage-wise PURE_BLOCK_MATCHING)"""
        mt = np.ascontiguousarray(matches, dtype=MATCH_DTYPE)
        T = self._poses if poses is None else np.ascontiguousarray(poses, np.float64).reshape(-1, 16)
        self._ck(self.lib.esvo_map_fuse_matches_naive(self.h, mt.ctypes.data, mt.shape[0], T.ctypes.data, T.shape[0]))

    def tick_resident(self, t_ns, T_world_cam, stamps, poses):
        """render both Time Surfaces at t_ns, take them as the observation, tick: one call"""
        st = np.ascontiguousarray(stamps, np.uint64)
        T = np.ascontiguousarray(poses, np.float64).reshape(-1, 16)
        Tw = np.ascontiguousarray(T_world_cam, np.float64).reshape(16)
        self._ck(self.lib.esvo_map_tick_resident(self.h, int(t_ns), Tw.ctypes.data, st.ctypes.data, T.ctypes.data, st.shape[0]))

    # ---- outputs
    def get_map(self):
        n = C.c_size_t(0)
        out = np.zeros(self.W * self.H, DEPTH_POINT_DTYPE)
        self._ck(self.lib.esvo_map_get_depth_points(self.h, out.ctypes.data, out.shape[0], C.byref(n)))
        return out[: n.value].copy()

    def get_committed_map(self):
        """(DepthMap of the newest committed tick, its stamp) without completing a pending tick"""
        out = np.zeros(self.W * self.H, DEPTH_POINT_DTYPE)
        n, t = C.c_size_t(), C.c_uint64()
        self._ck(self.lib.esvo_map_get_committed(self.h, out.ctypes.data, out.shape[0], C.byref(n), C.byref(t)))
        return out[:n.value].copy(), int(t.value)

    def get_pointcloud(self):
        n = C.c_size_t(0)
        out = np.zeros((self.W * self.H, 3), np.float32)
        self._ck(self.lib.esvo_map_get_pointcloud_xyz(self.h, out.ctypes.data, out.shape[0], C.byref(n)))
        return out[: n.value].copy()

    def get_debug_images(self, age_max_range=10.0):
        """(inverse depth, standard deviation, age, cost) images of publishMappingResults, BGR8"""
        imgs = [np.empty((self.H, self.W, 3), np.uint8) for _ in range(4)]
        self._ck(self.lib.esvo_map_get_debug_images(self.h, float(age_max_range), *[i.ctypes.data for i in imgs]))
        return imgs

    def get_pointcloud_near(self, visualize_range):
        n = C.c_size_t(0)
        out = np.zeros((self.W * self.H, 3), np.float32)
        self._ck(self.lib.esvo_map_get_pointcloud_near_xyz(self.h, float(visualize_range), out.ctypes.data, out.shape[0], C.byref(n)))
        return out[: n.value].copy()

    def save_depth_map(self, save_dir, t_ns):
        """esvo_MVStereo::saveDepthMap: writes <save_dir><t_ns>.txt ("x y depth" per valid element); returns the line count"""
        n = C.c_size_t()
        self._ck(self.lib.esvo_map_save_depth_map(self.h, str(save_dir).encode(), int(t_ns), C.byref(n)))
        return int(n.value)

    def get_last_frame(self):
        n = C.c_size_t(0)
        cap = max(int(self.params.max_events_per_tick), int(self.params.process_event_num), 1)
        out = np.zeros(cap, DEPTH_POINT_DTYPE)
        self._ck(self.lib.esvo_map_get_last_frame(self.h, out.ctypes.data, cap, C.byref(n)))
        return out[: n.value].copy()

    def stats(self):
        s = StatsStruct()
        self._ck(self.lib.esvo_get_stats(self.h, C.addressof(s)))
        return s

    # ---- device-resident stage calls (tick-interleaved multi-GPU operation, dist.TickShardedEsvo) ----
    def front(self, t_ns, stamps, poses):
        """front stage of a tick on the staged events; returns (device pointer of the frame, number of points)"""
        st = np.ascontiguousarray(stamps, np.uint64)
        T = np.ascontiguousarray(poses, np.float64).reshape(-1, 16)
        n = C.c_size_t()
        self._ck(self.lib.esvo_map_front(self.h, int(t_ns), st.ctypes.data, T.ctypes.data, st.shape[0], C.byref(n)))
        ptr = C.c_void_p()
        self._ck(self.lib.esvo_map_front_frame(self.h, C.byref(ptr)))
        return (ptr.value or 0), int(n.value)

    def push_frame_device(self, d_ptr, n, poses):
        T = np.ascontiguousarray(poses, np.float64).reshape(-1, 16)
        self._ck(self.lib.esvo_map_push_frame_device(self.h, C.c_void_p(int(d_ptr)), int(n), T.ctypes.data, T.shape[0]))

    def fuse_async(self):
        self._ck(self.lib.esvo_map_fuse_async(self.h))

    # ---- tracker residual / Jacobian evaluation (RegProblemLM.cpp), SURVEY.md section 8(f).1 ----
    def track_set_current(self, ts_left=None, kernel_size=5):
        img = None if ts_left is None else np.ascontiguousarray(ts_left, np.uint8)
        self._ck(self.lib.esvo_track_set_current(self.h, None if img is None else img.ctypes.data, int(kernel_size)))

    def track_images(self):
        neg = np.empty((self.H, self.W), np.uint8)
        du = np.empty((self.H, self.W), np.int16)
        dv = np.empty((self.H, self.W), np.int16)
        self._ck(self.lib.esvo_track_get_images(self.h, neg.ctypes.data, du.ctypes.data, dv.ctypes.data))
        return neg, du, dv

    def track_set_reference(self, xyz_world, T_world_ref):
        xyz = np.ascontiguousarray(xyz_world, np.float32).reshape(-1, 3)
        T = np.ascontiguousarray(T_world_ref, np.float64).reshape(16)
        self._ck(self.lib.esvo_track_set_reference(self.h, xyz.ctypes.data, xyz.shape[0], T.ctypes.data))

    def track_residuals(self, T_left_ref, offset, count, huber=True, huber_threshold=50.0):
        T = np.ascontiguousarray(T_left_ref, np.float64).reshape(16)
        out = np.empty(max(count, 1), np.float64)
        n = C.c_size_t()
        self._ck(self.lib.esvo_track_residuals(self.h, T.ctypes.data, int(offset), int(count), 1 if huber else 0,
                                               float(huber_threshold), out.ctypes.data, C.byref(n)))
        return out[:n.value]

    def track_jacobian(self, R, t, offset, count):
        R = np.ascontiguousarray(R, np.float64).reshape(9)
        t = np.ascontiguousarray(t, np.float64).reshape(3)
        out = np.empty(6 * max(count, 1), np.float64)
        n = C.c_size_t()
        self._ck(self.lib.esvo_track_jacobian(self.h, R.ctypes.data, t.ctypes.data, int(offset), int(count), out.ctypes.data, C.byref(n)))
        return out[:6 * n.value].reshape(6, n.value).T  # (n, 6); column-major like Eigen's fjac

    def track_normal_equations(self, R, t, offset, count, huber=True, huber_threshold=50.0):
        """(H = J^T J 6x6, b = J^T f, |f|^2, n) at (R, t): residuals, Jacobian and their products in one device call"""
        R = np.ascontiguousarray(R, np.float64).reshape(9)
        t = np.ascontiguousarray(t, np.float64).reshape(3)
        H, b = np.zeros((6, 6), np.float64), np.zeros(6, np.float64)
        cost, n = C.c_double(), C.c_size_t()
        self._ck(self.lib.esvo_track_normal_equations(self.h, R.ctypes.data, t.ctypes.data, int(offset), int(count), 1 if huber else 0,
                                                      float(huber_threshold), H.ctypes.data, b.ctypes.data, C.byref(cost), C.byref(n)))
        return H, b, cost.value, n.value

    def track_normal_equations_batch(self, Rs, ts, offset, count, huber=True, huber_threshold=50.0):
        """the same at k poses (Rs: k x 3 x 3, ts: k x 3) in one launch -> (H k x 6 x 6, b k x 6, cost k, n)"""
        Rs = np.ascontiguousarray(Rs, np.float64).reshape(-1, 9)
        ts = np.ascontiguousarray(ts, np.float64).reshape(-1, 3)
        k = len(Rs)
        H, b, cost = np.zeros((k, 6, 6), np.float64), np.zeros((k, 6), np.float64), np.zeros(k, np.float64)
        n = C.c_size_t()
        self._ck(self.lib.esvo_track_normal_equations_batch(self.h, k, Rs.ctypes.data, ts.ctypes.data, int(offset), int(count),
                                                            1 if huber else 0, float(huber_threshold), H.ctypes.data, b.ctypes.data,
                                                            cost.ctypes.data, C.byref(n)))
        return H, b, cost, n.value

    def track_register(self, n_points, R, t, huber=True, huber_threshold=50.0, max_iterations=12, damping=1e-3):
        """the registration loop inside the library (esvo_hip::gauss_newton_register over the normal equations):
        -> (R 3x3, t, rms, iterations)"""
        R = np.ascontiguousarray(R, np.float64).reshape(9).copy()
        t = np.ascontiguousarray(t, np.float64).reshape(3).copy()
        rms, it = C.c_double(), C.c_int()
        self._ck(self.lib.esvo_track_register(self.h, int(n_points), R.ctypes.data, t.ctypes.data, 1 if huber else 0, float(huber_threshold),
                                              int(max_iterations), float(damping), C.byref(rms), C.byref(it)))
        return R.reshape(3, 3), t, rms.value, it.value

    # ---- multi-GPU exchange behind the C-ABI (api_comm.hip): RCCL, or the two collectives as callbacks ----
    def comm_init(self, unique_id, rank, world):
        buf = (C.c_uint8 * 128).from_buffer_copy(bytes(unique_id))
        self._ck(self.lib.esvo_comm_init(self.h, buf, int(rank), int(world)))

    def comm_init_callbacks(self, rank, world, all_gather):
        """all_gather(d_send, d_recv, bytes_per_rank, stream) -> 0 on success: the one collective the library issues"""
        def trampoline(user, s, r, n, st):   # an exception must not unwind through the C frames: report it, fail the collective
            try:
                return int(all_gather(s, r, n, st) or 0)
            except BaseException:  # noqa: BLE001
                import traceback
                traceback.print_exc()
                return 1
        self._cb = ALL_GATHER_FN(trampoline)  # kept alive with the handle
        self._ck(self.lib.esvo_comm_init_callbacks(self.h, int(rank), int(world), self._cb, None))

    def comm_destroy(self):
        self._ck(self.lib.esvo_comm_destroy(self.h))

    def comm_owns_next_tick(self):
        return bool(self.lib.esvo_comm_owns_next_tick(self.h))

    def comm_tick(self, t_ns, T_world_cam, stamps, poses):
        st = np.ascontiguousarray(stamps, np.uint64)
        P = np.ascontiguousarray(poses, np.float64).reshape(-1, 16)
        T = np.ascontiguousarray(T_world_cam, np.float64).reshape(16)
        self._ck(self.lib.esvo_comm_tick(self.h, int(t_ns), T.ctypes.data, st.ctypes.data, P.ctypes.data, st.shape[0]))

    def comm_tick_resident(self, t_ns, T_world_cam, stamps, poses):
        """esvo_comm_tick with the owner's Time Surfaces rendered inside the call"""
        st = np.ascontiguousarray(stamps, np.uint64)
        P = np.ascontiguousarray(poses, np.float64).reshape(-1, 16)
        T = np.ascontiguousarray(T_world_cam, np.float64).reshape(16)
        self._ck(self.lib.esvo_comm_tick_resident(self.h, int(t_ns), T.ctypes.data, st.ctypes.data, P.ctypes.data, st.shape[0]))

    def comm_stats(self):
        from .abi import CommStatsStruct
        st = CommStatsStruct()
        self._ck(self.lib.esvo_comm_get_stats(self.h, C.byref(st)))
        return st

    def comm_flush(self):
        self._ck(self.lib.esvo_comm_flush(self.h))

    def comm_newest_map(self):
        out = np.zeros(self.W * self.H, DEPTH_POINT_DTYPE)
        n, k = C.c_size_t(), C.c_longlong()
        self._ck(self.lib.esvo_comm_newest_map(self.h, out.ctypes.data, out.shape[0], C.byref(n), C.byref(k)))
        return out[: n.value].copy(), int(k.value)

    def comm_shard_tick(self, t_ns, stamps, poses):
        st = np.ascontiguousarray(stamps, np.uint64)
        P = np.ascontiguousarray(poses, np.float64).reshape(-1, 16)
        self._ck(self.lib.esvo_comm_shard_tick(self.h, int(t_ns), st.ctypes.data, P.ctypes.data, st.shape[0]))

    def comm_gather_pointcloud(self):
        """the whole map's publishPointCloud cloud on every rank (collective): (n, 3) float32, world frame"""
        out = np.zeros((self.W * self.H, 3), np.float32)
        n = C.c_size_t()
        self._ck(self.lib.esvo_comm_gather_pointcloud_xyz(self.h, out.ctypes.data, out.shape[0], C.byref(n)))
        return out[: n.value].copy()

    def comm_gather_ts(self, cam):
        """routed band mode: the other ranks' rows of camera `cam`'s resident Time Surface (collective)"""
        self._ck(self.lib.esvo_comm_gather_ts(self.h, int(cam)))

    def comm_gather_map(self):
        out = np.zeros(self.W * self.H, DEPTH_POINT_DTYPE)
        n = C.c_size_t()
        self._ck(self.lib.esvo_comm_gather_map(self.h, out.ctypes.data, out.shape[0], C.byref(n)))
        return out[: n.value].copy()

    def set_band(self, y0, y1, shard=0, n_shards=1, routing=None, ts_halo_rows=-1):
        """row band + shard of this handle; routing: None / "broadcast" (every rank stages all events and renders the full Time
        Surfaces, per-event work dealt by slot) or "y_rect" (events routed by image row, banded raster)"""
        self._ck(self.lib.esvo_shard_set_band(self.h, int(y0), int(y1), int(shard), int(n_shards)))
        if routing not in (None, "broadcast"):
            assert routing == "y_rect", routing
            self._ck(self.lib.esvo_shard_set_routing(self.h, 1, int(ts_halo_rows)))

    def shard_rows(self):
        """dict of (begin, end) rows: render, observation, source_left, source_right (esvo_shard_get_rows)"""
        r = [(C.c_int * 2)() for _ in range(4)]
        self._ck(self.lib.esvo_shard_get_rows(self.h, *r))
        return dict(zip(("render", "observation", "source_left", "source_right"), [(int(a[0]), int(a[1])) for a in r]))

    def shard_exchange(self):
        """(send pointer, receive pointer, block bytes) of the all-gather due before the next phase (device pointers; block
        bytes == 0: nothing to do; one shard: receive aliases send)"""
        snd, rcv, nb = C.c_void_p(), C.c_void_p(), C.c_size_t()
        self._ck(self.lib.esvo_shard_exchange(self.h, C.byref(snd), C.byref(rcv), C.byref(nb)))
        return (snd.value or 0), (rcv.value or 0), int(nb.value)

    def shard_phase(self, phase, t_ns=0, stamps=None, poses=None):
        """one phase of esvo_shard_tick_phase; returns True when the phase must be called AGAIN behind the exchange that is now due
        (ESVO_AGAIN: phase 0 of a routed handle with Denoising -- the second call takes no arguments)"""
        if phase == 0 and stamps is not None:
            st = np.ascontiguousarray(stamps, np.uint64)
            T = np.ascontiguousarray(poses, np.float64).reshape(-1, 16)
            rc = self.lib.esvo_shard_tick_phase(self.h, 0, int(t_ns), st.ctypes.data, T.ctypes.data, st.shape[0])
        else:
            rc = self.lib.esvo_shard_tick_phase(self.h, int(phase), 0, None, None, 0)
        if rc == ESVO_AGAIN:
            return True
        self._ck(rc)
        return False
