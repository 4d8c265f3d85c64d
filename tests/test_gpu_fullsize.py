"""Full-size (bench workload: 640x480, ~180 k events per tick, 81 disparity candidates) checks.

* element-for-element parity with the CPU oracle AT THE BENCHMARKED SIZE: three throughput ticks of the dsec640x480
  bench workload (window of three frames, long per-cell record lists, > 64-candidate block-matching passes, the
  SORT_CAP fallback) and one tick of the hd1280x720 stress workload; the oracle runs BM + LM on all host threads in its
  GPU-comparable mode, so the bar is bit equality of every DepthMap element;
* properties that need no oracle: determinism, idempotence of the Time-Surface render, the invariants
  SmartGrid::clean / DepthRegularization guarantee for every element, agreement of the lazy and the eager tick, and
  tick-interleaved == single handle."""
import numpy as np
import pytest

import bench
from esvo_amd import calib, params, rostime, synth

pytestmark = pytest.mark.gpu
F64 = ["inv_depth", "scale2", "nu", "variance", "residual", "x", "p_cam"]


@pytest.fixture(scope="module")
def full():
    return bench.make_workload("dsec640x480", 6)   # the benchmarked configuration itself (stream, parameters, ticks)


def _run(full, eager):
    from esvo_amd import lib
    rig, stream, p, ticks = full
    dev = lib.Esvo(p, rig)
    dev.ts_push_events(0, stream.ev_left)
    dev.ts_push_events(1, stream.ev_right)
    maps = []
    for t, stamps, poses, T in ticks:
        dev.ts_render(0, t, download=False); dev.ts_render(1, t, download=False)
        dev.set_observation(t, None, None, T)
        dev.tick(t, stamps, poses)
        if eager:
            maps.append(dev.get_map())     # completes the tick at once
    if not eager:
        maps.append(dev.get_map())
    return dev, maps


def _same(a, b):
    assert len(a) == len(b)
    for f in ("row", "col", "age"):
        assert np.array_equal(a[f], b[f]), f
    for f in F64:
        assert np.array_equal(a[f], b[f]), f


def test_full_size_determinism_lazy_vs_eager_and_invariants(full):
    rig, stream, p, ticks = full
    dev_e, maps_e = _run(full, eager=True)
    dev_l, maps_l = _run(full, eager=False)
    _same(maps_e[-1], maps_l[-1])                      # lazily completed ticks == ticks completed one by one
    s = dev_l.stats()
    assert s.ticks == len(ticks) and s.total_events_in > 150000 * len(ticks) and s.total_points > 8000 * len(ticks)
    m = maps_l[-1]
    assert len(m) > 30000
    # SmartGrid::clean (window full from tick 5 on): every element passed valid(); the regulariser then only changes rho
    reg_invalid = m["inv_depth"] == -1.0             # DepthRegularization.cpp:99-101
    ok = m[~reg_invalid]
    assert reg_invalid.sum() < len(m) and len(ok) > 10000
    assert (ok["inv_depth"] > 0).all() and (m["variance"] <= p.stdvar_vis_threshold ** 2).all() and (m["age"] >= p.age_vis_threshold).all()
    assert (m["scale2"] > 0).all() and (m["nu"] >= 2.0).all() and np.isfinite(m["p_cam"]).all()
    assert (m["row"] < rig.height).all() and (m["col"] < rig.width).all()
    keys = m["row"].astype(np.int64) * rig.width + m["col"]
    assert len(np.unique(keys)) >= 0.98 * len(keys)    # displaced elements (Appendix A-7) may share a believed cell, rarely
    # idempotence of the render: same T, same image
    t = ticks[-1][0]
    assert np.array_equal(dev_l.ts_render(0, t), dev_l.ts_render(0, t))


def test_full_size_tick_interleaved_equals_single(full):
    import torch
    from esvo_amd import dist as edist
    from esvo_amd import lib
    rig, stream, p, ticks = full
    _, maps = _run(full, eager=True)
    G = 2
    ranks = [lib.Esvo(p, rig) for _ in range(G)]
    for d in ranks:
        d.ts_push_events(0, stream.ev_left)
        d.ts_push_events(1, stream.ev_right)
    for k0 in range(0, len(ticks), G):
        rnd = list(range(k0, min(k0 + G, len(ticks))))
        fronts = {}
        for kk in rnd:
            d = ranks[kk % G]
            t, stamps, poses, T = ticks[kk]
            d.ts_render(0, t, download=False); d.ts_render(1, t, download=False)
            d.set_observation(t, None, None, T)
            fronts[kk] = d.front(t, stamps, poses)
        frames = {kk: edist.device_tensor(ptr, max(n, 1) * 13, "<i8")[: n * 13].clone() for kk, (ptr, n) in fronts.items()}
        torch.cuda.synchronize()
        for g, d in enumerate(ranks):
            for kk in rnd:
                d.push_frame_device(frames[kk].data_ptr(), fronts[kk][1], ticks[kk][2])
                if kk % G == g:
                    d.fuse_async()
        for kk in rnd:
            _same(ranks[kk % G].get_map(), maps[kk])


def _oracle_ticks(rig, stream, p, ticks, n):
    """the canonical-mode oracle over the first n ticks (TS raster, event selection, MappingAtTime); maps per tick"""
    import os
    from oracle import oracle as O
    fast = False   # the portable -O2 build: the -march=native one may have been compiled on another host
    m = O.OracleMapper(p, rig, fast=fast)
    m.set_mode(True, True)
    m.set_threads(os.cpu_count() or 1)
    ts = [O.OracleTS(rig.width, rig.height, fast=fast), O.OracleTS(rig.width, rig.height, fast=fast)]
    done = [0, 0]
    maps = []
    for t, stamps, poses, T in ticks[:n]:
        for cam, (ev, ns) in enumerate(((stream.ev_left, stream.ns_left), (stream.ev_right, stream.ns_right))):
            hi = int(np.searchsorted(ns, t, side="left"))
            ts[cam].push(ev[done[cam]:hi])
            done[cam] = hi
        l = ts[0].render(t, map_x=rig.left.map_x, map_y=rig.left.map_y)
        r = ts[1].render(t, map_x=rig.right.map_x, map_y=rig.right.map_y)
        m.set_observation(t, l, r, T)
        m.set_poses(stamps, poses)
        idx = O.select_events(stream.ev_left, t, p.bm_half_slice_thickness, p.process_event_num, fast=fast)
        m.tick(stream.ev_left[idx])
        maps.append((len(idx), m.get_map(), m.counters()))
    return maps


def test_full_size_ticks_equal_oracle(full):
    """the benchmarked configuration itself, compared with the oracle element for element"""
    rig, stream, p, ticks = full
    n = 3
    dev, maps = _run((rig, stream, p, ticks[:n]), eager=True)
    ora = _oracle_ticks(rig, stream, p, ticks, n)
    for k in range(n):
        n_ev, om, cnt = ora[k]
        assert n_ev > 120000 and len(om) > 20000
        _same(maps[k], om)
    assert ora[-1][2]["replace_displaced"] > 100      # Appendix A-7 at scale
    s = dev.stats()
    assert s.last_events_in == ora[-1][0] and s.last_window_frames == n


def test_hd_tick_equals_oracle():
    """one tick of the 1280x720 / 145-candidate stress workload against the oracle"""
    from esvo_amd import lib
    rig, stream, p, ticks = bench.make_workload("hd1280x720", 1)
    dev, maps = _run((rig, stream, p, ticks), eager=True)
    n_ev, om, _ = _oracle_ticks(rig, stream, p, ticks, 1)[0]
    assert n_ev > 300000 and len(om) > 20000
    _same(maps[0], om)


_SWITCH_SETS = [
    {"ESVO_LM_PERSIST": "1"},                                                       # persistent narrow LM layout
    {"ESVO_LM_PERSIST": "1", "ESVO_LM_PERSIST_BLOCKS": "300"},                      # ... with fewer groups than matches per pass
    {"ESVO_COLLECT_ASIDE": "0", "ESVO_RESYNC": "0", "ESVO_FRONT_THROTTLE": "1"},    # round 4's queue discipline
    {"ESVO_LM_QUEUES": "2"},                                                        # two LM queues whatever the launch size
    {"ESVO_ONE_STREAM": "1"},                                                       # every stage in one queue
    {"ESVO_REG_SPARSE": "1", "ESVO_BACK_PROLOGUE": "1", "ESVO_PIPE_BIG_TIMED_EVERY": "4"},   # round 6: the regulariser's sparse-map layout on a
                                                                                    # dense map, the one-launch back prologue, sampled stage events
]


@pytest.mark.parametrize("env_set", _SWITCH_SETS, ids=lambda e: "+".join(f"{k[5:]}={v}" for k, v in e.items()))
def test_scheduling_switches_change_no_bit(full, env_set):
    """`Scheduling is the library's business and never changes a result` (include/esvo_hip.h): six lazily completed (pipelined)
    ticks of the benchmarked configuration under the experiment switches that move work between queues or pick another LM layout
    -- read at esvo_create, only with ESVO_DEV_SWITCHES=1 -- give the DepthMap of the default handle, SHA-1 for SHA-1."""
    import os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    _, maps = _run(full, eager=False)
    want = bench.map_sha1(maps[-1])
    code = ("import bench, sys; sys.path.insert(0, 'tests'); import test_gpu_fullsize as T; "
            "full = bench.make_workload('dsec640x480', 6); d, m = T._run(full, eager=False); "
            "print('SHA1', bench.map_sha1(m[-1]), len(m[-1]))")
    env = dict(os.environ, ESVO_DEV_SWITCHES="1", ESVO_BENCH_STREAM_CACHE="/tmp/esvo_streams_test", **env_set)  # (one generation for the five)
    r = subprocess.run([sys.executable, "-c", code], cwd=root, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    got = [l.split() for l in r.stdout.splitlines() if l.startswith("SHA1")][-1]
    assert got[1] == want and int(got[2]) == len(maps[-1]), (env_set, got, want)
