import ctypes, sys, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np
from esvo_amd import calib, lib, params, rostime, synth
rig = calib.dataset_rig("dsec")
st = synth.make_stream(rig, 180000, 0.06 + 6 * 0.01, 0.02, 0.25, seed=20250418 + 3, speed=2.0)
p, _ = params.make_params(params.PRESETS["mapping_dsec"], rig, throughput_events=300000, event_ring_capacity=max(1 << 22, int(len(st.ev_left) * 1.1)))
dev = lib.Esvo(p, rig); dev.ts_push_events(0, st.ev_left); dev.ts_push_events(1, st.ev_right)
L = lib.load(); out = (ctypes.c_ulonglong * 8)()
tot_m = 0
for k in range(5):
    t = st.t0_ns + int((0.06 + (k + 1) * 0.01) * 1e9)
    stamps, poses = rostime.pose_table(st.pose, t, p.bm_half_slice_thickness)
    dev.ts_render(0, t, download=False); dev.ts_render(1, t, download=False)
    dev.set_observation(t, None, None, st.pose(t)); dev.tick(t, stamps, poses)
    tot_m += dev.stats().last_matches
L.esvo_debug_lm_counters(out)
o = list(out)
print("matches", tot_m, "evals/group", o[0], "evals/wave", o[1], "loop iters/group", o[2], "loop iters/wave", o[3], "shortcuts", o[4])
print("evals per match %.1f; wave evals per (match/4) %.1f; loop iters per eval (group) %.2f; per wave-eval %.2f" % (o[0]/tot_m, o[1]/(tot_m/4), o[2]/o[0], o[3]/o[1]))
print("Jacobian evaluations", o[5], "of which at the same x as the previous one (same x):", o[6], "= %.1f %% of all evaluations" % (100.0*o[6]/o[0]))
