"""esvo_params_t from the reference's yaml keys.

Mirrors the parameter reads of the esvo_Mapping / esvo_MVStereo constructors
(esvo_core/src/esvo_Mapping.cpp:37-99) and of the TimeSurface node
(esvo_time_surface/src/TimeSurface.cpp:23-30), including the disparity-range clamp of
esvo_Mapping.cpp:110-116.  `PRESETS` embeds the values of the shipped configs as data.
"""
import math

import yaml

from .abi import FUSION_CONST_FRAMES, FUSION_CONST_POINTS, LSNORM_L2, LSNORM_TDIST, ParamsStruct

# code defaults: tools::param(pnh_, key, default) in esvo_Mapping.cpp:37-99
CODE_DEFAULTS = dict(
    patch_size_X=25, patch_size_Y=25, LSnorm="Tdist", Tdist_nu=0.0, Tdist_scale=0.0,
    ITERATION_OPTIMIZATION=10, RegularizationRadius=5, RegularizationMinNeighbours=8,
    RegularizationMinCloseNeighbours=8, SmoothTimeSurface=False,
    invDepth_min_range=0.16, invDepth_max_range=2.0, residual_vis_threshold=15,
    stdVar_vis_threshold=0.005, age_max_range=5, age_vis_threshold=0, fusion_radius=0,
    maxNumFusionFrames=10, FUSION_STRATEGY="CONST_FRAMES", maxNumFusionPoints=2000,
    Denoising=False, Regularization=False, PROCESS_EVENT_NUM=500, TS_HISTORY_LENGTH=100,
    mapping_rate_hz=20, BM_half_slice_thickness=0.001, BM_min_disparity=3, BM_max_disparity=40,
    BM_step=1, BM_ZNCC_Threshold=0.1, BM_bUpDownConfiguration=False,
    # esvo_time_surface (ts_parameters.yaml / TimeSurface.cpp:23-30)
    decay_ms=30.0, median_blur_kernel_size=1, ignore_polarity=True, time_surface_mode=0,
    max_event_queue_len=20,
)

# values of the shipped yaml files (esvo_core/cfg/{mapping,mvstereo}/*.yaml), Appendix C
PRESETS = {
    "mvstereo_upenn": dict(
        patch_size_X=15, patch_size_Y=7, LSnorm="Tdist", Tdist_nu=2.182, Tdist_scale=17.277,
        invDepth_min_range=0.16, invDepth_max_range=1.0, residual_vis_threshold=20, stdVar_vis_threshold=0.15,
        age_max_range=10, age_vis_threshold=1, fusion_radius=0, FUSION_STRATEGY="CONST_POINTS",
        maxNumFusionFrames=40, maxNumFusionPoints=3000, Denoising=False, SmoothTimeSurface=False,
        Regularization=False, PROCESS_EVENT_NUM=1000, BM_min_disparity=1, BM_max_disparity=40, BM_step=1,
        BM_ZNCC_Threshold=0.1, node="mvstereo"),
    "mapping_upenn": dict(
        patch_size_X=15, patch_size_Y=7, LSnorm="Tdist", Tdist_nu=2.182, Tdist_scale=17.277,
        invDepth_min_range=0.16, invDepth_max_range=1.0, residual_vis_threshold=20, stdVar_vis_threshold=0.15,
        age_max_range=10, age_vis_threshold=1, fusion_radius=0, FUSION_STRATEGY="CONST_POINTS",
        maxNumFusionFrames=40, maxNumFusionPoints=3000, Denoising=False, SmoothTimeSurface=False,
        Regularization=False, PROCESS_EVENT_NUM=1000, BM_min_disparity=1, BM_max_disparity=40, BM_step=1,
        BM_ZNCC_Threshold=0.1, node="mapping"),
    "mvstereo_rpg": dict(
        patch_size_X=15, patch_size_Y=7, LSnorm="Tdist", Tdist_nu=2.1897, Tdist_scale=16.6397,
        invDepth_min_range=0.2, invDepth_max_range=2.0, residual_vis_threshold=20, stdVar_vis_threshold=0.015,
        age_max_range=10, age_vis_threshold=1, fusion_radius=0, FUSION_STRATEGY="CONST_FRAMES",
        maxNumFusionFrames=40, maxNumFusionPoints=5000, Denoising=True, SmoothTimeSurface=False,
        Regularization=True, RegularizationRadius=5, RegularizationMinNeighbours=8,
        RegularizationMinCloseNeighbours=8, PROCESS_EVENT_NUM=1000, BM_min_disparity=1, BM_max_disparity=40,
        BM_step=1, BM_ZNCC_Threshold=0.1, node="mvstereo"),
    "mapping_rpg": dict(  # cfg/mapping/mapping_rpg.yaml (`Lnorm` is misspelt there -> code default Tdist)
        patch_size_X=15, patch_size_Y=7, LSnorm="Tdist", Tdist_nu=2.1897, Tdist_scale=16.6397,
        invDepth_min_range=0.2, invDepth_max_range=2.0, residual_vis_threshold=20, stdVar_vis_threshold=0.015,
        age_max_range=10, age_vis_threshold=1, fusion_radius=0, FUSION_STRATEGY="CONST_POINTS",
        maxNumFusionFrames=40, maxNumFusionPoints=5000, Denoising=True, SmoothTimeSurface=False,
        Regularization=True, RegularizationRadius=5, RegularizationMinNeighbours=8,
        RegularizationMinCloseNeighbours=8, PROCESS_EVENT_NUM=1000, BM_min_disparity=1, BM_max_disparity=40,
        BM_step=1, BM_ZNCC_Threshold=0.1, node="mapping"),
    "mapping_hkust": dict(
        patch_size_X=15, patch_size_Y=7, LSnorm="Tdist", Tdist_nu=2.1897, Tdist_scale=16.6397,
        invDepth_min_range=0.25, invDepth_max_range=2.0, residual_vis_threshold=20, stdVar_vis_threshold=0.15,
        age_max_range=10, age_vis_threshold=1, fusion_radius=0, FUSION_STRATEGY="CONST_FRAMES",
        maxNumFusionFrames=20, maxNumFusionPoints=4000, Denoising=True, SmoothTimeSurface=False,
        Regularization=True, RegularizationRadius=5, RegularizationMinNeighbours=8,
        RegularizationMinCloseNeighbours=8, PROCESS_EVENT_NUM=1000, BM_min_disparity=1, BM_max_disparity=40,
        BM_step=1, BM_ZNCC_Threshold=0.1, node="mapping"),
    "mapping_dsec": dict(
        patch_size_X=15, patch_size_Y=7, LSnorm="Tdist", Tdist_nu=2.182, Tdist_scale=17.277,
        invDepth_min_range=0.001, invDepth_max_range=0.25, residual_vis_threshold=30, stdVar_vis_threshold=1.0,
        age_max_range=10, age_vis_threshold=1, fusion_radius=1, FUSION_STRATEGY="CONST_FRAMES",
        maxNumFusionFrames=5, maxNumFusionPoints=20000, Denoising=False, SmoothTimeSurface=True,
        Regularization=True, RegularizationRadius=20, RegularizationMinNeighbours=32,
        RegularizationMinCloseNeighbours=32, PROCESS_EVENT_NUM=10000, BM_min_disparity=0, BM_max_disparity=150,
        BM_step=1, BM_ZNCC_Threshold=0.1, node="mapping"),
    # SURVEY.md §8 "synthetic HD" stress row: mapping_dsec.yaml with the inverse-depth range 0.02-0.5 and the
    # disparity cap 150 (6...150 at f*b = 300 -> 145 candidates)
    "mapping_hd": dict(
        patch_size_X=15, patch_size_Y=7, LSnorm="Tdist", Tdist_nu=2.182, Tdist_scale=17.277,
        invDepth_min_range=0.02, invDepth_max_range=0.5, residual_vis_threshold=30, stdVar_vis_threshold=1.0,
        age_max_range=10, age_vis_threshold=1, fusion_radius=1, FUSION_STRATEGY="CONST_FRAMES",
        maxNumFusionFrames=5, maxNumFusionPoints=20000, Denoising=False, SmoothTimeSurface=True,
        Regularization=True, RegularizationRadius=20, RegularizationMinNeighbours=32,
        RegularizationMinCloseNeighbours=32, PROCESS_EVENT_NUM=100000, BM_min_disparity=0, BM_max_disparity=150,
        BM_step=1, BM_ZNCC_Threshold=0.1, node="mapping"),
}


def disparity_range(cfg, focal, baseline):
    """esvo_Mapping.cpp:110-116."""
    min_d = max(int(math.floor(focal * baseline * cfg["invDepth_min_range"])), 0)
    max_d = int(math.ceil(focal * baseline * cfg["invDepth_max_range"]))
    return max(min_d, int(cfg["BM_min_disparity"])), min(max_d, int(cfg["BM_max_disparity"]))


def make_params(cfg, rig, node=None, throughput_events=None, **overrides):
    """Build a ParamsStruct from a yaml-key dict + the rig (focal, baseline).

    node: 'mapping' (clean only when the window is full, esvo_Mapping.cpp:385) or 'mvstereo'
    (always clean, esvo_MVStereo.cpp:496).  throughput_events: override PROCESS_EVENT_NUM for
    the throughput-mode benchmark."""
    c = dict(CODE_DEFAULTS)
    c.update(cfg)
    node = node or c.get("node", "mapping")
    p = ParamsStruct()
    p.decay_ms = float(c["decay_ms"])
    p.median_blur_kernel_size = int(c["median_blur_kernel_size"])
    p.ignore_polarity = int(bool(c["ignore_polarity"]))
    p.patch_size_x, p.patch_size_y = int(c["patch_size_X"]), int(c["patch_size_Y"])
    p.ls_norm = LSNORM_TDIST if str(c["LSnorm"]) == "Tdist" else LSNORM_L2
    p.td_nu, p.td_scale = float(c["Tdist_nu"]), float(c["Tdist_scale"])
    p.lm_max_iteration = int(c["ITERATION_OPTIMIZATION"])
    p.reg_radius = int(c["RegularizationRadius"])
    p.reg_min_neighbours = int(c["RegularizationMinNeighbours"])
    p.reg_min_close_neighbours = int(c["RegularizationMinCloseNeighbours"])
    dmin, dmax = disparity_range(c, rig.focal, rig.baseline)
    p.bm_min_disparity, p.bm_max_disparity = dmin, dmax
    p.bm_step = int(c["BM_step"])
    p.bm_zncc_threshold = float(c["BM_ZNCC_Threshold"])
    p.bm_updown = int(bool(c["BM_bUpDownConfiguration"]))
    p.smooth_time_surface = int(bool(c["SmoothTimeSurface"]))
    p.invdepth_min, p.invdepth_max = float(c["invDepth_min_range"]), float(c["invDepth_max_range"])
    p.stdvar_vis_threshold = float(c["stdVar_vis_threshold"])
    p.residual_vis_threshold = float(c["residual_vis_threshold"])
    p.age_vis_threshold = float(c["age_vis_threshold"])
    p.fusion_radius = int(c["fusion_radius"])
    p.fusion_strategy = FUSION_CONST_POINTS if str(c["FUSION_STRATEGY"]) == "CONST_POINTS" else FUSION_CONST_FRAMES
    p.max_fusion_frames = int(c["maxNumFusionFrames"])
    p.max_fusion_points = int(c["maxNumFusionPoints"])
    p.clean_requires_full_window = 1 if node == "mapping" else 0
    p.regularization = int(bool(c["Regularization"]))
    p.denoising = int(bool(c["Denoising"]))
    p.process_event_num = int(throughput_events if throughput_events else c["PROCESS_EVENT_NUM"])
    p.bm_half_slice_thickness = float(c["BM_half_slice_thickness"])
    p.num_threads = 4  # NUM_THREAD_MAPPING, esvo_core/include/esvo_core/tools/utils.h:36
    p.max_events_per_tick = max(p.process_event_num, 1024)
    if p.fusion_strategy == FUSION_CONST_POINTS:
        p.max_window_points = int(1.5 * p.max_fusion_points) + p.max_events_per_tick + 1
    else:
        p.max_window_points = p.max_fusion_frames * p.max_events_per_tick
    p.max_poses_per_tick = 256
    p.event_ring_capacity = 1 << 24
    # The reference's max_event_queue_len (20) is NOT taken over by default: 0 selects the one-stamp-per-pixel fast path
    # (include/esvo_hip.h); pass max_event_queue_len=20 as an override for EventQueueMat's exact semantics.
    p.max_event_queue_len = 0
    for k, v in overrides.items():
        if not hasattr(p, k):
            raise KeyError(k)
        setattr(p, k, v)
    return p, bool(c["Denoising"])


def load_yaml_cfg(path):
    with open(path) as f:
        return yaml.safe_load(f)
