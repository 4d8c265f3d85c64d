"""Seeded mapper scenarios shared by the fixture generators and the parity tests.

A scenario = a shipped rig + a shipped parameter preset + a synthetic stereo stream + a tick schedule.  `inputs()`
produces, per tick, exactly what esvo_Mapping::dataTransferring hands to MappingAtTime (esvo_Mapping.cpp:494-605):
the Time-Surface pair, its pose, the 50 us pose table and the selected (optionally denoised) left events.  Those
inputs come from the CPU oracle's TS raster / event selection (the node code and the TS node are not part of
oracle/_ref); everything downstream of them is what the scenarios compare between mappers.
"""
import numpy as np

from esvo_amd import calib, params, rostime, synth

# name -> (rig, preset, n_points, rho range, seed, n_ticks, PROCESS_EVENT_NUM, speed, first tick offset s, tick period s)
SCENARIOS = {
    # 346x260 equidistant, 2x2 fusion, CONST_POINTS, no regulariser (cfg/mvstereo/mvstereo_upenn.yaml)
    "upenn": dict(rig="upenn", preset="mvstereo_upenn", n_points=5000, rho=(0.16, 1.0), seed=20250501, n_ticks=4,
                  n_events=600, speed=1.0, t_first=0.06, dt=0.01),
    # 640x480 plumb_bob, smoothed TS, 3x3 fusion, CONST_FRAMES 5 (window fills -> clean), regulariser r=20
    # (cfg/mapping/mapping_dsec.yaml)
    "dsec": dict(rig="dsec", preset="mapping_dsec", n_points=9000, rho=(0.02, 0.25), seed=20250502, n_ticks=6,
                 n_events=900, speed=2.0, t_first=0.05, dt=0.01),
    # 240x180 DAVIS240C, Denoising, regulariser r=5, always-clean MVStereo node (cfg/mvstereo/mvstereo_rpg.yaml).
    # PROCESS_EVENT_NUM is raised from 1000 on a densely firing scene: the 3x3 median of the event map (Denoising)
    # keeps next to nothing of 1000 synthetic events
    "rpg": dict(rig="rpg", preset="mvstereo_rpg", n_points=3000, rho=(0.3, 1.8), seed=20250503, n_ticks=5,
                n_events=6000, speed=1.5, t_first=0.05, dt=0.01),
    # 346x260 plumb_bob, Denoising, regulariser r=5, CONST_FRAMES 20 (cfg/mapping/mapping_hkust.yaml); the window is
    # shortened to 4 frames so that the Mapping node's "clean only when the window is full" rule fires in a short run
    "hkust": dict(rig="hkust", preset="mapping_hkust", n_points=5000, rho=(0.35, 1.8), seed=20250504, n_ticks=6,
                  n_events=8000, speed=1.5, t_first=0.05, dt=0.01, overrides=dict(max_fusion_frames=4)),
    # ---- the same rigs at the SHIPPED tick sizes (PROCESS_EVENT_NUM of cfg/mvstereo/mvstereo_upenn.yaml:18 and
    # cfg/mapping/mapping_dsec.yaml:18).  Their fixtures keep every tick's matches and points but only digests of the
    # DepthMaps (the last map in full): BIG below.
    "upenn1k": dict(rig="upenn", preset="mvstereo_upenn", n_points=5000, rho=(0.16, 1.0), seed=20250501, n_ticks=4,
                    n_events=1000, speed=1.0, t_first=0.06, dt=0.01),
    # 15 000 scene points at 2 m/s fire ~1.2 M events/s per camera: the 10 ms slice holds > 10 000, the selection is cut
    "dsec10k": dict(rig="dsec", preset="mapping_dsec", n_points=15000, rho=(0.02, 0.25), seed=20250512, n_ticks=6,
                    n_events=10000, speed=2.0, t_first=0.05, dt=0.01),
    # ---- the reference's CODE-DEFAULT patch (patch_size_X = patch_size_Y = 25, esvo_Mapping.cpp:38-39 / esvo_MVStereo.cpp:39-40; every
    # shipped yaml sets 15 x 7): the general block-matching and refinement kernels
    "upenn25": dict(rig="upenn", preset="mvstereo_upenn", n_points=5000, rho=(0.16, 1.0), seed=20250521, n_ticks=3,
                    n_events=600, speed=1.0, t_first=0.06, dt=0.01, overrides=dict(patch_size_x=25, patch_size_y=25)),
    # an even, non-square patch on the smoothed DSEC surfaces with the regulariser on (left-top = centre - (w - 1) / 2: the
    # block is not centred, EventBM.cpp:255-258)
    "dsec10x4": dict(rig="dsec", preset="mapping_dsec", n_points=9000, rho=(0.02, 0.25), seed=20250522, n_ticks=3,
                     n_events=900, speed=2.0, t_first=0.05, dt=0.01, overrides=dict(patch_size_x=10, patch_size_y=4)),
}
BIG = ("upenn1k", "dsec10k")
PATCH = ("upenn25", "dsec10x4")   # other patch sizes than the shipped 15 x 7


class Scenario:
    def __init__(self, name, n_ticks=None):
        s = dict(SCENARIOS[name])
        if n_ticks is not None:   # a longer run of the same scene (the stream's first n_ticks are the fixture's)
            s["n_ticks"] = n_ticks
        self.name, self.spec = name, s
        self.rig = calib.dataset_rig(s["rig"])
        self.params, self.denoise = params.make_params(params.PRESETS[s["preset"]], self.rig,
                                                       process_event_num=s["n_events"], **s.get("overrides", {}))
        self.n_ticks = s["n_ticks"]

    def stream(self):
        s = self.spec
        dur = s["t_first"] + s["dt"] * s["n_ticks"] + 0.01
        return synth.make_stream(self.rig, s["n_points"], dur, s["rho"][0], s["rho"][1], seed=s["seed"], speed=s["speed"])

    def inputs(self):
        """per tick: dict(t, tsL, tsR, T, stamps, poses, ev).  tsL/tsR are what the mapper's TS_obs_ holds, i.e. after
        GaussianBlurTS(5) when SmoothTimeSurface is set; `raw` keeps the un-smoothed pair for the C-ABI, which smooths
        on the device."""
        from oracle import oracle as O
        s, rig, p = self.spec, self.rig, self.params
        st = self.stream()
        ts = [O.OracleTS(rig.width, rig.height), O.OracleTS(rig.width, rig.height)]
        done = [0, 0]
        out = []
        for k in range(s["n_ticks"]):
            t = st.t0_ns + int(round((s["t_first"] + s["dt"] * k) * 1e9))
            for cam, (ev, ns) in enumerate(((st.ev_left, st.ns_left), (st.ev_right, st.ns_right))):
                hi = int(np.searchsorted(ns, t, side="left"))
                ts[cam].push(ev[done[cam]:hi])
                done[cam] = hi
            l = ts[0].render(t, map_x=rig.left.map_x, map_y=rig.left.map_y)
            r = ts[1].render(t, map_x=rig.right.map_x, map_y=rig.right.map_y)
            stamps, poses = rostime.pose_table(st.pose, t, p.bm_half_slice_thickness)
            if self.denoise:  # esvo_Mapping.cpp:282-296
                idx_all = O.select_events(st.ev_left, t, p.bm_half_slice_thickness, p.process_event_num)
                idx = O.denoise_events(st.ev_left, idx_all, rig.width, rig.height, p.process_event_num)
            else:
                idx = O.select_events(st.ev_left, t, p.bm_half_slice_thickness, p.process_event_num)
            out.append(dict(t=t, raw=(l, r), tsL=O.gaussian5(l) if p.smooth_time_surface else l,
                            tsR=O.gaussian5(r) if p.smooth_time_surface else r, T=st.pose(t), stamps=stamps, poses=poses,
                            ev=st.ev_left[idx]))
        return out


def run_stagewise(mapper, ticks, pre_smoothed):
    """MappingAtTime stage by stage on any mapper with the OracleMapper interface; returns per-tick outputs.
    pre_smoothed: the mapper receives TS_obs_ as it is after GaussianBlurTS (oracle/_ref has no OpenCV)."""
    res = []
    for tk in ticks:
        l, r = (tk["tsL"], tk["tsR"]) if pre_smoothed else tk["raw"]
        mapper.set_observation(tk["t"], l, r, tk["T"])
        mapper.set_poses(tk["stamps"], tk["poses"])
        mt = mapper.match(tk["ev"])
        pts = mapper.refine(mt, cull=True)
        mapper.push_frame(pts, tk["poses"])
        nf = mapper.fuse()
        res.append(dict(matches=mt, points=pts, nf=nf, map=mapper.get_map()))
    return res
