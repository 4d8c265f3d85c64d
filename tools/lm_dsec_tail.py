"""Distribution of the t-scale iteration counts per match on the bench workload (needs -DLM_STATS)."""
import ctypes, sys, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np
from esvo_amd import calib, lib, params, rostime, synth
rig = calib.dataset_rig("dsec")
st = synth.make_stream(rig, 180000, 0.06 + 3 * 0.01, 0.02, 0.25, seed=20250418 + 3, speed=2.0)
p, _ = params.make_params(params.PRESETS["mapping_dsec"], rig, throughput_events=300000, event_ring_capacity=max(1 << 22, int(len(st.ev_left) * 1.1)))
dev = lib.Esvo(p, rig); dev.ts_push_events(0, st.ev_left); dev.ts_push_events(1, st.ev_right)
L = lib.load()
buf = np.zeros((3, 1 << 18), np.uint32)
for k in range(2):
    t = st.t0_ns + int((0.06 + (k + 1) * 0.01) * 1e9)
    stamps, poses = rostime.pose_table(st.pose, t, p.bm_half_slice_thickness)
    dev.ts_render(0, t, download=False); dev.ts_render(1, t, download=False)
    dev.set_observation(t, None, None, st.pose(t)); dev.tick(t, stamps, poses)
    M = dev.stats().last_matches
    L.esvo_debug_lm_slots(buf.ctypes.data_as(ctypes.c_void_p), 1)
    ev, it = buf[0, :M].astype(np.int64), buf[1, :M].astype(np.int64)
    knz = buf[2, :M].astype(np.int64) // 1000
    print("tick", k, "M", M, "iters/match mean", it.mean(), "p50/p99/p99.9/max", np.percentile(it, [50, 99, 99.9]), it.max(), "evals max", ev.max())
    cost = ev * 900 + it * 141          # rough wave-instructions if the match ran alone
    w = cost[: M // 4 * 4].reshape(-1, 4).max(1)
    print("   wave cost mean %.0f p99 %.0f max %.0f (instr); matches with > 1000 iterations: %d, knz of the top 5: %s" % (w.mean(), np.percentile(w, 99), w.max(), (it > 1000).sum(), knz[np.argsort(-it)[:5]]))
