"""Is there a cheap predictor of a match's LM cost (t-scale iterations, evaluations) among what block matching already
knows?  Needs -DLM_STATS.  Prints rank correlations and the wave-divergence each ordering would leave."""
import ctypes, sys, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np
from scipy import stats
from esvo_amd import calib, lib, params, rostime, synth
rig = calib.dataset_rig("dsec")
st = synth.make_stream(rig, 180000, 0.06 + 2 * 0.01, 0.02, 0.25, seed=20250418 + 3, speed=2.0)
p, _ = params.make_params(params.PRESETS["mapping_dsec"], rig, throughput_events=300000, event_ring_capacity=max(1 << 22, int(len(st.ev_left) * 1.1)))
dev = lib.Esvo(p, rig); dev.ts_push_events(0, st.ev_left); dev.ts_push_events(1, st.ev_right)
L = lib.load()
t = st.t0_ns + int(0.07e9)
stamps, poses = rostime.pose_table(st.pose, t, p.bm_half_slice_thickness)
dev.ts_render(0, t, download=False); dev.ts_render(1, t, download=False)
dev.set_observation(t, None, None, st.pose(t))
ns = st.ns_left
sel = st.ev_left[(ns >= t - 10_000_000) & (ns < t)][::-1][:150000]
m = dev.match(sel, stamps, poses)
buf = np.zeros((3, 1 << 18), np.uint32)
L.esvo_debug_lm_slots(buf.ctypes.data_as(ctypes.c_void_p), 1)
pts = dev.refine(m, cull=False)
L.esvo_debug_lm_slots(buf.ctypes.data_as(ctypes.c_void_p), 1)
M = len(m)
# slot s solves match j = stride_item(s, M, 4): thread q gets items q, q+4, ...
order = np.concatenate([np.arange(q, M, 4) for q in range(4)])
ev, it = buf[0, :M].astype(np.int64), buf[1, :M].astype(np.int64)
cost_by_slot = m["cost"][order]; disp_by_slot = m["disp"][order]; y_by_slot = m["x_left"][order, 1]
print("matches", M, "iters mean", it.mean())
for name, key in (("zncc cost", cost_by_slot), ("disparity", disp_by_slot), ("row", y_by_slot)):
    print("  spearman(iters, %s) = %.3f   spearman(evals, %s) = %.3f" % (name, stats.spearmanr(it, key).correlation, name, stats.spearmanr(ev, key).correlation))
def wave_cost(o):
    pad = (-len(o)) % 4
    o = np.concatenate([o, np.full(pad, -1)])
    i = np.where(o >= 0, it[o], 0).reshape(-1, 4); e = np.where(o >= 0, ev[o], 0).reshape(-1, 4)
    return e.max(1).sum(), i.max(1).sum()
print("  slot order", wave_cost(np.arange(M)), " by cost", wave_cost(np.argsort(cost_by_slot)), " by disparity", wave_cost(np.argsort(disp_by_slot)),
      " ideal", (ev.sum() // 4, it.sum() // 4))
