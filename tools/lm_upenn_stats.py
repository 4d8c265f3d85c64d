import ctypes, sys, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np
from esvo_amd import calib, lib, params, rostime, synth
rig = calib.dataset_rig("upenn")
st = synth.make_stream(rig, 24000, 0.06 + 4 * 0.01, 0.16, 1.0, seed=20250418 + 3, speed=1.0)
p, _ = params.make_params(params.PRESETS["mapping_upenn"], rig, throughput_events=40000, event_ring_capacity=max(1 << 22, int(len(st.ev_left) * 1.1)))
dev = lib.Esvo(p, rig); dev.ts_push_events(0, st.ev_left); dev.ts_push_events(1, st.ev_right)
L = lib.load()
buf = np.zeros((3, 1 << 18), np.uint32)
for k in range(3):
    t = st.t0_ns + int((0.06 + (k + 1) * 0.01) * 1e9)
    stamps, poses = rostime.pose_table(st.pose, t, p.bm_half_slice_thickness)
    dev.ts_render(0, t, download=False); dev.ts_render(1, t, download=False)
    dev.set_observation(t, None, None, st.pose(t)); dev.tick(t, stamps, poses)
    M = dev.stats().last_matches
    L.esvo_debug_lm_slots(buf.ctypes.data_as(ctypes.c_void_p), 1)
    ev, it = buf[0, :M].astype(np.int64), buf[1, :M].astype(np.int64)
    knz = buf[2, :M].astype(np.int64) // 1000
    print("tick", k, "M", M, "evals mean/max", ev.mean(), ev.max(), "iters mean/max", it.mean(), it.max(), "p99", np.percentile(it, 99), "knz mean", knz.mean())
    top = np.argsort(-it)[:5]; print("   top iters", it[top], "their evals", ev[top], "knz", knz[top])
