"""How much LM work does lockstep execution of 4 matches per wave waste, and what would grouping matches by the
t-scale iteration count of their FIRST evaluation recover?  Needs a library built with -DLM_STATS
(ESVO_EXTRA_HIPCC_FLAGS=-DLM_STATS python -c "from esvo_amd import lib; lib.build(force=True)")."""
import ctypes, sys, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np
from esvo_amd import calib, lib, params, rostime, synth
rig = calib.dataset_rig("dsec")
st = synth.make_stream(rig, 180000, 0.06 + 4 * 0.01, 0.02, 0.25, seed=20250418 + 3, speed=2.0)
p, _ = params.make_params(params.PRESETS["mapping_dsec"], rig, throughput_events=300000, event_ring_capacity=max(1 << 22, int(len(st.ev_left) * 1.1)))
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
    ev, it, it0 = buf[0, :M].astype(np.int64), buf[1, :M].astype(np.int64), buf[2, :M].astype(np.int64) % 1000
    knz = buf[2, :M].astype(np.int64) // 1000
    print('   nonzero residuals at x0: mean %.1f, percentiles 10/50/90: %s' % (knz.mean(), np.percentile(knz, [10, 50, 90])))
    def cost(order):
        pad = (-len(order)) % 4
        o = np.concatenate([order, np.full(pad, -1)])
        e = np.where(o >= 0, ev[o], 0).reshape(-1, 4); i = np.where(o >= 0, it[o], 0).reshape(-1, 4)
        return e.max(1).sum(), i.max(1).sum()   # lower bounds of the wave-level counts (perfect alignment inside a wave)
    base = cost(np.arange(M))
    by0 = cost(np.argsort(it0, kind="stable"))
    byit = cost(np.argsort(it, kind="stable"))
    byboth = cost(np.lexsort((it, ev)))
    print(f"tick {k}: M={M} evals/match {ev.mean():.1f} iters/match {it.mean():.1f} first-eval iters {it0.mean():.2f}")
    print(f"   ideal (no divergence): evals {ev.sum()/4:.0f} iters {it.sum()/4:.0f}")
    print(f"   slot order          : evals {base[0]} iters {base[1]}")
    print(f"   sorted by first-eval iters: evals {by0[0]} iters {by0[1]}")
    print(f"   sorted by total iters (oracle): evals {byit[0]} iters {byit[1]};  by (evals, iters): {byboth[0]} {byboth[1]}")
