"""Host-side logic and the C-ABI library (no compute calls without a GPU)."""
import ctypes
import os

import numpy as np
import pytest

from esvo_amd import abi, calib, params, rostime


def test_library_loads_and_exports_every_symbol():
    from esvo_amd import lib
    so = ctypes.CDLL(lib._LIB_PATH)
    hdr = open(os.path.join(os.path.dirname(lib._CSRC), "..", "include", "esvo_hip.h")).read()
    import re
    declared = set(re.findall(r"^(?:int|void|const char\*) (esvo_[a-z_]+)\(", hdr, re.M))
    assert declared == set(lib.SYMBOLS), declared ^ set(lib.SYMBOLS)
    for s in declared:
        getattr(so, s)


def test_abi_struct_sizes_match_bindings():
    from esvo_amd import lib
    s = lib.abi_sizes()
    assert s[0] == abi.EVENT_DTYPE.itemsize == 16
    assert s[1] == ctypes.sizeof(abi.CalibStruct) and s[2] == ctypes.sizeof(abi.ParamsStruct)
    assert s[3] == abi.MATCH_DTYPE.itemsize == 48 and s[4] == abi.DEPTH_POINT_DTYPE.itemsize == 104
    assert s[5] == ctypes.sizeof(abi.StatsStruct) and s[6] == 0
    assert s[7] == 8


def test_kernel_time_means_divide_by_the_sampled_ticks():
    """ABI 8: stage timings are sampled; sum_ms_kernel[2..6] sums over esvo_stats_t.stage_timing_samples ticks (not over `ticks`),
    the Time-Surface slots over the sampled renders counted in [7] (two renders per pair sample)"""
    base, s = abi.StatsStruct(), abi.StatsStruct()
    base.ticks, base.stage_timing_samples = 10, 8
    base.sum_ms_kernel[3], base.sum_ms_kernel[0], base.sum_ms_kernel[1], base.sum_ms_kernel[7] = 8.0, 0.8, 1.6, 16
    s.ticks, s.stage_timing_samples = 110, 12           # 100 more ticks, 4 of them sampled
    s.sum_ms_kernel[3], s.sum_ms_kernel[0], s.sum_ms_kernel[1], s.sum_ms_kernel[7] = 8.0 + 4 * 1.25, 0.8 + 0.4, 1.6 + 0.8, 16 + 8
    m = s.kernel_ms_mean(base)
    assert abs(m[3] - 1.25) < 1e-12
    assert abs(m[0] - 2 * 0.4 / 8) < 1e-12 and abs(m[1] - 2 * 0.8 / 8) < 1e-12 and m[7] == 8
    assert abs(s.kernel_ms_mean()[3] - (8.0 + 5.0) / 12) < 1e-12
    assert abs(abi.StatsStruct().kernel_ms_mean()[3]) == 0.0   # no sample yet: zeros, not a division by zero


def test_create_fails_loudly_without_gpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is visible here")
    from esvo_amd import lib
    rig = calib.ideal_rig(64, 48, 100.0, 0.1)
    p, _ = params.make_params(params.PRESETS["mvstereo_upenn"], rig)
    with pytest.raises(lib.EsvoError, match="no HIP device|no CPU fallback|failed"):
        lib.Esvo(p, rig)


def test_unsupported_params_are_rejected_not_approximated():
    from esvo_amd import lib
    so = lib.load()
    rig = calib.ideal_rig(64, 48, 100.0, 0.1)
    # (patch 25 x 25 and median_blur_kernel_size 2 are supported since round 4: general kernels)
    for kw in (dict(patch_size_x=65), dict(patch_size_y=41), dict(patch_size_x=0), dict(ls_norm=7), dict(median_blur_kernel_size=4)):
        p, _ = params.make_params(params.PRESETS["mvstereo_upenn"], rig, **kw)
        h = ctypes.c_void_p()
        cl, cr = rig.left.as_struct(), rig.right.as_struct()
        rc = so.esvo_create(ctypes.addressof(p), ctypes.addressof(cl), ctypes.addressof(cr), 0, ctypes.byref(h))
        assert rc == -5, (kw, rc, so.esvo_last_error(None))


def test_disparity_ranges_of_shipped_configs():
    """effective BM ranges after the clamp of esvo_Mapping.cpp:110-116."""
    cases = [("mvstereo_rpg", 156.925, 0.14805, (4, 40)), ("mvstereo_upenn", 199.653, 0.09988, (3, 20)),
             ("mapping_hkust", 189.705, 0.07308, (3, 28)), ("mapping_dsec", 534.094, 0.59903, (0, 80))]
    for preset, f, b, expect in cases:
        rig = calib.ideal_rig(64, 48, f, b)
        p, _ = params.make_params(params.PRESETS[preset], rig)
        assert (p.bm_min_disparity, p.bm_max_disparity) == expect, preset
    p, den = params.make_params(params.PRESETS["mvstereo_rpg"], calib.ideal_rig(64, 48, 156.925, 0.14805))
    assert den and p.regularization == 1 and p.clean_requires_full_window == 0


def test_ros_time_helpers():
    assert rostime.ros_time_from_sec(1.5) == 1_500_000_000
    assert rostime.ns_to_sec(1_600_000_000_123_456_789) == 1_600_000_000.0 + 1e-9 * 123_456_789
    st = rostime.pose_stamps(10_100_000_000, 0.001)
    assert len(st) == 201 and st[0] == 10_090_000_000 and st[-1] <= 10_100_000_000  # 10 ms / 50 us + 1
    assert np.all(np.diff(st.astype(np.int64)) > 0)


def test_synthetic_stream_is_deterministic_and_sorted():
    from esvo_amd import synth
    rig = calib.ideal_rig(96, 64, 120.0, 0.1)
    a = synth.make_stream(rig, 500, 0.03, 0.2, 1.0, seed=11)
    b = synth.make_stream(rig, 500, 0.03, 0.2, 1.0, seed=11)
    assert np.array_equal(a.ev_left, b.ev_left) and np.array_equal(a.ev_right, b.ev_right)
    assert np.all(np.diff(a.ns_left.astype(np.int64)) >= 0) and len(a.ev_left) > 100
    assert a.ev_left["x"].max() < 96 and a.ev_left["y"].max() < 64
