"""ctypes / numpy mirrors of the POD types declared in include/esvo_hip.h.

Shared by the product bindings (esvo_amd.lib) and by the test-only oracle bindings
(oracle/oracle.py).  Layouts are asserted against sizeof() exported by the libraries.
"""
import ctypes as C

import numpy as np

EVENT_DTYPE = np.dtype(
    [("x", "<u2"), ("y", "<u2"), ("sec", "<u4"), ("nsec", "<u4"), ("polarity", "u1"), ("_pad", "u1", (3,))]
)
assert EVENT_DTYPE.itemsize == 16

MATCH_DTYPE = np.dtype(
    [("x_left", "<f8", (2,)), ("inv_depth", "<f8"), ("cost", "<f8"), ("disp", "<f8"),
     ("event_idx", "<u4"), ("pose_idx", "<u4")]
)
assert MATCH_DTYPE.itemsize == 48

DEPTH_POINT_DTYPE = np.dtype(
    [("row", "<u4"), ("col", "<u4"), ("x", "<f8", (2,)), ("inv_depth", "<f8"), ("scale2", "<f8"),
     ("nu", "<f8"), ("variance", "<f8"), ("residual", "<f8"), ("age", "<u8"), ("p_cam", "<f8", (3,)),
     ("pose_idx", "<u4"), ("seq", "<u4")]
)
assert DEPTH_POINT_DTYPE.itemsize == 104

CAM_LEFT, CAM_RIGHT = 0, 1
FUSION_CONST_FRAMES, FUSION_CONST_POINTS = 0, 1
LSNORM_TDIST, LSNORM_L2 = 0, 1


class CalibStruct(C.Structure):
    _fields_ = [
        ("width", C.c_int32), ("height", C.c_int32),
        ("P", C.c_double * 12),
        ("rect_lut", C.c_void_p), ("rect_mask", C.c_void_p),
        ("map_x", C.c_void_p), ("map_y", C.c_void_p),
    ]


class ParamsStruct(C.Structure):
    _fields_ = [
        ("decay_ms", C.c_double),
        ("median_blur_kernel_size", C.c_int32),
        ("ignore_polarity", C.c_int32),
        ("patch_size_x", C.c_int32),
        ("patch_size_y", C.c_int32),
        ("ls_norm", C.c_int32),
        ("td_nu", C.c_double),
        ("td_scale", C.c_double),
        ("lm_max_iteration", C.c_int32),
        ("reg_radius", C.c_int32),
        ("reg_min_neighbours", C.c_int32),
        ("reg_min_close_neighbours", C.c_int32),
        ("bm_min_disparity", C.c_int32),
        ("bm_max_disparity", C.c_int32),
        ("bm_step", C.c_int32),
        ("bm_zncc_threshold", C.c_double),
        ("bm_updown", C.c_int32),
        ("smooth_time_surface", C.c_int32),
        ("invdepth_min", C.c_double),
        ("invdepth_max", C.c_double),
        ("stdvar_vis_threshold", C.c_double),
        ("residual_vis_threshold", C.c_double),
        ("age_vis_threshold", C.c_double),
        ("fusion_radius", C.c_int32),
        ("fusion_strategy", C.c_int32),
        ("max_fusion_frames", C.c_int32),
        ("max_fusion_points", C.c_int32),
        ("clean_requires_full_window", C.c_int32),
        ("regularization", C.c_int32),
        ("denoising", C.c_int32),
        ("process_event_num", C.c_int32),
        ("bm_half_slice_thickness", C.c_double),
        ("num_threads", C.c_int32),
        ("max_events_per_tick", C.c_int32),
        ("max_window_points", C.c_int32),
        ("max_poses_per_tick", C.c_int32),
        ("event_ring_capacity", C.c_int64),
        ("max_event_queue_len", C.c_int32),   # ABI 5: 0 = one stamp per pixel (fast path), 1..32 = EventQueueMat semantics
        ("pad_params_", C.c_int32),
    ]


class StatsStruct(C.Structure):
    _fields_ = [
        ("ticks", C.c_uint64),
        ("events_staged", C.c_uint64 * 2),
        ("events_scattered", C.c_uint64 * 2),
        ("ts_frames", C.c_uint64 * 2),
        ("last_events_in", C.c_uint32),
        ("last_matches", C.c_uint32),
        ("last_solved", C.c_uint32),
        ("last_points", C.c_uint32),
        ("last_window_frames", C.c_uint32),
        ("last_window_points", C.c_uint32),
        ("last_fusions", C.c_uint32),
        ("last_map_size", C.c_uint32),
        ("ms_ts_scatter", C.c_float), ("ms_ts_render", C.c_float),
        ("ms_bm", C.c_float), ("ms_refine", C.c_float), ("ms_fusion", C.c_float),
        ("ms_regularization", C.c_float), ("ms_tick_total", C.c_float),
        ("ms_kernel", C.c_float * 8),
        ("pad_", C.c_float),
        ("total_events_in", C.c_uint64), ("total_matches", C.c_uint64), ("total_points", C.c_uint64),
        ("sum_ms_kernel", C.c_double * 8),
        ("last_bm_info_noise_low", C.c_uint32), ("last_bm_coarse_fail", C.c_uint32), ("last_bm_fine_fail", C.c_uint32), ("pad2_", C.c_uint32),
        ("total_bm_info_noise_low", C.c_uint64), ("total_bm_coarse_fail", C.c_uint64), ("total_bm_fine_fail", C.c_uint64),
        # ABI 3: in-run shader-clock probe of the refinement kernel (include/esvo_hip.h)
        ("clk_cycles", C.c_uint64 * 8), ("clk_ref_ticks", C.c_uint64 * 8), ("clk_samples", C.c_uint64),
        ("clk_ref_khz", C.c_uint32), ("stage_timing_samples", C.c_uint32),
        # ABI 6: routed band mode
        ("halo_violations", C.c_uint64),
        ("late_events", C.c_uint64 * 2),
        ("pipeline_resyncs", C.c_uint64),
    ]

    def kernel_ms_mean(self, base=None):
        """mean HIP-event time per launch of the eight ms_kernel slots since `base` (an earlier StatsStruct) or since esvo_create:
        sum_ms_kernel[2..6] over the ticks that recorded their stage events (stage_timing_samples, ABI 8: stage timings are sampled),
        [0..1] over the sampled Time-Surface renders (two per pair sample, counted in [7])"""
        ks = [float(self.sum_ms_kernel[i]) - (float(base.sum_ms_kernel[i]) if base is not None else 0.0) for i in range(8)]
        n = int(self.stage_timing_samples) - (int(base.stage_timing_samples) if base is not None else 0)
        out = [k / n if n > 0 else 0.0 for k in ks]
        if ks[7] > 0:
            out[0], out[1] = 2 * ks[0] / ks[7], 2 * ks[1] / ks[7]
        out[7] = ks[7]
        return out

    def sclk_mhz(self, base=None):
        """(all XCDs, [per XCD]) shader clock in MHz the LM kernel ran at since `base` (an earlier StatsStruct) or since
        esvo_create: sum of s_memtime differences / sum of s_memrealtime differences x the reference rate; None without samples"""
        cyc = [int(self.clk_cycles[i]) - (int(base.clk_cycles[i]) if base is not None else 0) for i in range(8)]
        ref = [int(self.clk_ref_ticks[i]) - (int(base.clk_ref_ticks[i]) if base is not None else 0) for i in range(8)]
        khz = float(self.clk_ref_khz)
        per = [c / r * khz / 1e3 if r > 0 else None for c, r in zip(cyc, ref)]
        tot = sum(cyc) / sum(ref) * khz / 1e3 if sum(ref) > 0 else None
        return tot, per


class CommStatsStruct(C.Structure):
    """esvo_comm_stats_t (ABI 7): the tick-interleaved frame exchange of one rank"""
    _fields_ = [
        ("rounds", C.c_uint64), ("gathers", C.c_uint64), ("regrows", C.c_uint64), ("bytes_sent", C.c_uint64),
        ("points_gathered", C.c_uint64), ("host_wait_us", C.c_uint64), ("last_stride_points", C.c_uint32), ("stride_cap_points", C.c_uint32),
    ]


def make_events(x, y, t_ns, polarity=None):
    """Build an esvo_event_t array from coordinate / nanosecond-timestamp arrays."""
    t_ns = np.asarray(t_ns, dtype=np.uint64)
    ev = np.zeros(t_ns.shape[0], dtype=EVENT_DTYPE)
    ev["x"] = x
    ev["y"] = y
    ev["sec"] = (t_ns // np.uint64(1_000_000_000)).astype(np.uint32)
    ev["nsec"] = (t_ns % np.uint64(1_000_000_000)).astype(np.uint32)
    ev["polarity"] = 1 if polarity is None else polarity
    return ev


def event_ns(ev):
    return ev["sec"].astype(np.uint64) * np.uint64(1_000_000_000) + ev["nsec"].astype(np.uint64)


class Calib:
    """Host-side calibration products for one camera, kept alive for the C struct."""

    def __init__(self, width, height, P, rect_lut, rect_mask, map_x, map_y):
        self.width, self.height = int(width), int(height)
        self.P = np.ascontiguousarray(P, dtype=np.float64).reshape(12)
        self.rect_lut = np.ascontiguousarray(rect_lut, dtype=np.float32).reshape(height, width, 2)
        self.rect_mask = None if rect_mask is None else np.ascontiguousarray(rect_mask, dtype=np.uint8).reshape(height, width)
        self.map_x = np.ascontiguousarray(map_x, dtype=np.float32).reshape(height, width)
        self.map_y = np.ascontiguousarray(map_y, dtype=np.float32).reshape(height, width)

    def as_struct(self):
        s = CalibStruct()
        s.width, s.height = self.width, self.height
        for i in range(12):
            s.P[i] = float(self.P[i])
        s.rect_lut = self.rect_lut.ctypes.data
        s.rect_mask = 0 if self.rect_mask is None else self.rect_mask.ctypes.data
        s.map_x = self.map_x.ctypes.data
        s.map_y = self.map_y.ctypes.data
        return s


def serialize_event_array(ev, width, height, seq=0, stamp_ns=0, frame_id=""):
    """ROS1 wire format of a dvs_msgs/EventArray carrying `ev` (esvo_event_t array): std_msgs/Header (u32 seq, u32 sec,
    u32 nsec, u32 len + frame_id), u32 height, u32 width, u32 count, count x {u16 x, u16 y, u32 sec, u32 nsec, u8 polarity}."""
    import struct
    fid = frame_id.encode()
    head = struct.pack("<III", seq, stamp_ns // 1_000_000_000, stamp_ns % 1_000_000_000) + struct.pack("<I", len(fid)) + fid
    head += struct.pack("<III", height, width, len(ev))
    wire = np.zeros(len(ev), dtype=np.dtype([("x", "<u2"), ("y", "<u2"), ("sec", "<u4"), ("nsec", "<u4"), ("polarity", "u1")]))
    for f in ("x", "y", "sec", "nsec", "polarity"):
        wire[f] = ev[f]
    assert wire.dtype.itemsize == 13
    return head + wire.tobytes()
