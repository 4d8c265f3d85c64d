"""The multi-GPU exchange behind the C-ABI (esvo_comm_*, esvo_amd/csrc/api_comm.hip) on ONE GPU.

* real RCCL with a one-rank communicator: ncclGetUniqueId / ncclCommInitRank / ncclAllGather run from libesvo_hip.so;
* several ranks = several handles driven from threads, their collectives supplied as callbacks that rendezvous in-process
  (esvo_comm_init_callbacks): the round logic, the in-band counts, block growth, partial rounds, buffer reuse, the band
  mode's two all-gathers per tick and the all-gather of the DepthMap bands are the code a node with one process per GPU runs.
Every tick's DepthMap must equal the single-handle one bit for bit."""
import threading

import numpy as np
import pytest

from esvo_amd import dist as edist
from esvo_amd import params, rostime

pytestmark = pytest.mark.gpu
F64 = ["inv_depth", "scale2", "nu", "variance", "residual", "x", "p_cam"]


def _same(a, b):
    assert len(a) == len(b), (len(a), len(b))
    for f in ("row", "col", "age"):
        assert np.array_equal(a[f], b[f]), f
    for f in F64:
        assert np.array_equal(a[f], b[f]), f


def _ticks(stream, p, n, t_first=0.06, dt=0.01):
    out = []
    for k in range(n):
        t = stream.t0_ns + int((t_first + k * dt) * 1e9)
        stamps, poses = rostime.pose_table(stream.pose, t, p.bm_half_slice_thickness)
        out.append((t, stamps, poses, stream.pose(t)))
    return out


def _single(p, rig, stream, ticks):
    from esvo_amd import lib
    dev = lib.Esvo(p, rig)
    dev.ts_push_events(0, stream.ev_left)
    dev.ts_push_events(1, stream.ev_right)
    maps = []
    for t, stamps, poses, T in ticks:
        dev.ts_render(0, t, download=False)
        dev.ts_render(1, t, download=False)
        dev.set_observation(t, None, None, T)
        dev.tick(t, stamps, poses)
        maps.append(dev.get_map())
    return maps


LocalTransport = edist.LocalTransport   # the in-process all-gather between handles of one process (esvo_amd/dist.py)


def _run_ranks(world, body):
    errs, outs = [], [None] * world

    def main(r):
        try:
            outs[r] = body(r)
        except BaseException as e:  # noqa: BLE001
            errs.append((r, e))
            try:
                tr_abort()
            except Exception:
                pass

    def tr_abort():
        pass

    th = [threading.Thread(target=main, args=(r,)) for r in range(world)]
    for t in th:
        t.start()
    for t in th:
        t.join(timeout=300)
    assert not errs, errs
    return outs


def test_rccl_one_rank_communicator(upenn_rig, upenn_stream):
    from esvo_amd import lib
    p, _ = params.make_params(params.PRESETS["mapping_upenn"], upenn_rig)
    ticks = _ticks(upenn_stream, p, 5)
    ref = _single(p, upenn_rig, upenn_stream, ticks)
    dev = lib.Esvo(p, upenn_rig)
    dev.comm_init(lib.comm_unique_id(), 0, 1)
    dev.ts_push_events(0, upenn_stream.ev_left)
    dev.ts_push_events(1, upenn_stream.ev_right)
    for k, (t, stamps, poses, T) in enumerate(ticks):
        assert dev.comm_owns_next_tick()
        dev.ts_render(0, t, download=False)
        dev.ts_render(1, t, download=False)
        dev.comm_tick(t, T, stamps, poses)
        mp, idx = dev.comm_newest_map()
        assert idx == k
        _same(mp, ref[k])
    assert dev.stats().ticks == len(ticks)
    dev.comm_destroy()


@pytest.mark.parametrize("world,preset,rig_fix,stream_fix,n_ticks", [
    (2, "mapping_upenn", "upenn_rig", "upenn_stream", 7),      # CONST_POINTS window, a partial last round
    (3, "mapping_dsec", "dsec_rig", "dsec_stream", 7),         # CONST_FRAMES 5, 3x3 fusion, regulariser
])
def test_tick_interleaved_ranks_through_the_c_calls(request, world, preset, rig_fix, stream_fix, n_ticks):
    from esvo_amd import lib
    rig, stream = request.getfixturevalue(rig_fix), request.getfixturevalue(stream_fix)
    p, _ = params.make_params(params.PRESETS[preset], rig, process_event_num=3000)
    ticks = _ticks(stream, p, n_ticks)
    ref = _single(p, rig, stream, ticks)
    tr = LocalTransport(world)

    def body(r):
        dev = lib.Esvo(p, rig)
        dev.comm_init_callbacks(r, world, lambda s, d, n, st: tr.all_gather(r, s, d, n))
        dev.ts_push_events(0, stream.ev_left)
        dev.ts_push_events(1, stream.ev_right)
        got = {}
        for k, (t, stamps, poses, T) in enumerate(ticks):
            if dev.comm_owns_next_tick():
                dev.ts_render(0, t, download=False)
                dev.ts_render(1, t, download=False)
            dev.comm_tick(t, T, stamps, poses)
            if k % world == world - 1 or k == 3 or k == len(ticks) - 1:   # round ends, one flush inside a round, the tail
                mp, idx = dev.comm_newest_map()
                got[idx] = mp
        s = dev.stats()
        return got, int(s.ticks)

    outs = _run_ranks(world, body)
    assert sum(o[1] for o in outs) == n_ticks
    for got, _ in outs:
        assert got and (n_ticks - 1) in got
        for idx, mp in got.items():
            _same(mp, ref[idx])


@pytest.mark.parametrize("world,stride0,resident", [(2, None, True), (3, 16, True), (8, 64, False)])
def test_tick_interleaved_two_rounds_in_flight(request, monkeypatch, world, stride0, resident):
    """The pipelined round logic (ABI 7): the exchange of round j is enqueued on its own stream and collected a round later;
    blocks are sized from the counts of earlier rounds.  No flush inside the run (the rounds really overlap); a tiny initial
    block capacity (ESVO_COMM_STRIDE0) forces the regrow path on every rank alike, for the round being collected AND the one
    enqueued behind it; esvo_comm_tick_resident renders the owner's Time Surfaces inside the call.  Every rank's view of the
    newest map -- and every own tick's frame along the way -- equals the single-handle run bit for bit."""
    from esvo_amd import lib
    rig, stream = request.getfixturevalue("dsec_rig"), request.getfixturevalue("dsec_stream")
    if stride0 is not None:
        monkeypatch.setenv("ESVO_COMM_STRIDE0", str(stride0))
    p, _ = params.make_params(params.PRESETS["mapping_dsec"], rig, process_event_num=3000)
    n_ticks = 3 * world + 1                                  # three full rounds and a one-tick tail
    ticks = _ticks(stream, p, n_ticks, t_first=0.04, dt=0.075 / (n_ticks - 1))   # (the fixture stream lasts 0.12 s)
    ref = _single(p, rig, stream, ticks)
    tr = LocalTransport(world)

    def body(r):
        dev = lib.Esvo(p, rig)
        dev.comm_init_callbacks(r, world, lambda s, d, n, st: tr.all_gather(r, s, d, n))
        dev.ts_push_events(0, stream.ev_left)
        dev.ts_push_events(1, stream.ev_right)
        for k, (t, stamps, poses, T) in enumerate(ticks):
            if resident:
                dev.comm_tick_resident(t, T, stamps, poses)
            else:
                if dev.comm_owns_next_tick():
                    dev.ts_render(0, t, download=False)
                    dev.ts_render(1, t, download=False)
                dev.comm_tick(t, T, stamps, poses)
        mp, idx = dev.comm_newest_map()
        cs = dev.comm_stats()
        s = dev.stats()
        return mp, idx, int(s.ticks), (int(cs.rounds), int(cs.gathers), int(cs.regrows), int(cs.bytes_sent), int(cs.points_gathered),
                                       int(cs.last_stride_points), int(cs.stride_cap_points)), int(s.events_scattered[0]), int(s.events_scattered[1])

    outs = _run_ranks(world, body)
    assert sum(o[2] for o in outs) == n_ticks
    for mp, idx, _, cs, sc0, sc1 in outs:
        assert idx == n_ticks - 1
        _same(mp, ref[-1])
        rounds, gathers, regrows, sent, pts, last_stride, cap = cs
        assert rounds == 4 and cs == outs[0][3]             # every rank took the same path
        if stride0 is not None:
            assert regrows >= 1 and gathers > rounds and cap > stride0
        else:
            assert regrows == 0 and gathers == rounds
        assert last_stride <= cap
    # every rank scattered every event below its last own render time, whoever's tick they belong to
    assert min(o[4] for o in outs) > 0.5 * max(o[4] for o in outs)


@pytest.mark.parametrize("world,routing,rig_fix,stream_fix,preset", [
    (2, "broadcast", "dsec_rig", "dsec_stream", "mapping_dsec"),
    (2, "y_rect", "dsec_rig", "dsec_stream", "mapping_dsec"),
    (8, "y_rect", "dsec_rig", "dsec_stream", "mapping_dsec"),       # 60-row bands, regulariser halo of 20 rows on both sides
    (8, "broadcast", "dsec_rig", "dsec_stream", "mapping_dsec"),
    (8, "y_rect", "upenn_rig", "upenn_stream", "mapping_upenn"),    # 260 rows / 8: ragged bands (33 x 7 + 29), CONST_POINTS window
])
def test_band_sharded_ranks_through_the_c_calls(request, world, routing, rig_fix, stream_fix, preset):
    """world ranks (threads, one handle each, ONE GPU) through esvo_comm_shard_tick / esvo_comm_gather_map with an in-process
    all-gather: every rank ends up with the unsharded DepthMap, bit for bit -- at the world size BASELINE.json's headline names"""
    from esvo_amd import lib
    rig, stream = request.getfixturevalue(rig_fix), request.getfixturevalue(stream_fix)
    p, _ = params.make_params(params.PRESETS[preset], rig, process_event_num=3000)
    ticks = _ticks(stream, p, 3)
    ref = _single(p, rig, stream, ticks)
    tr = LocalTransport(world)

    def body(r):
        dev = lib.Esvo(p, rig)
        y0, y1 = edist.band_of(r, world, rig.height)
        dev.set_band(y0, y1, r, world, routing=routing)
        dev.comm_init_callbacks(r, world, lambda s, d, n, st: tr.all_gather(r, s, d, n))
        dev.ts_push_events(0, stream.ev_left)   # every rank is handed the whole stream; a routed handle keeps its rows
        dev.ts_push_events(1, stream.ev_right)
        maps = []
        for t, stamps, poses, T in ticks:
            dev.ts_render(0, t, download=False)
            dev.ts_render(1, t, download=False)
            dev.set_observation(t, None, None, T)
            dev.comm_shard_tick(t, stamps, poses)
            maps.append(dev.comm_gather_map())
        st = dev.stats()
        assert st.halo_violations == 0
        return maps, int(st.events_staged[0]), int(st.events_staged[1])

    outs = _run_ranks(world, body)
    for maps, _, _ in outs:
        for k, mp in enumerate(maps):
            _same(mp, ref[k])
    if routing == "y_rect" and world == 8:  # band-local ingest: a rank stages its rows (+ halo), not the stream
        n_l, n_r = len(stream.ev_left), len(stream.ev_right)
        assert max(o[1] for o in outs) < 0.6 * n_l and max(o[2] for o in outs) < 0.6 * n_r, [(o[1], o[2]) for o in outs]
        assert sum(o[1] for o in outs) >= n_l * 0.9   # (every event is somebody's)


def test_denoising_rig_band_sharded_through_the_c_calls():
    """esvo_comm_shard_tick on a routed handle with Denoising (hkust): the C driver loop handles ESVO_AGAIN -- the all-gather of the
    mask bits, then phase 0 proper -- and every rank's gathered map equals the one-GPU map."""
    from esvo_amd import lib
    from tests import scenarios
    sc = scenarios.Scenario("hkust")
    rig, p, stream, spec = sc.rig, sc.params, sc.stream(), sc.spec
    world = 4
    ticks = []
    for k in range(4):
        t = stream.t0_ns + int(round((spec["t_first"] + spec["dt"] * k) * 1e9))
        stamps, poses = rostime.pose_table(stream.pose, t, p.bm_half_slice_thickness)
        ticks.append((t, stamps, poses, stream.pose(t)))
    ref = _single(p, rig, stream, ticks)
    tr = LocalTransport(world)

    def body(r):
        dev = lib.Esvo(p, rig)
        y0, y1 = edist.band_of(r, world, rig.height)
        dev.set_band(y0, y1, r, world, routing="y_rect")
        dev.comm_init_callbacks(r, world, lambda s, d, n, st: tr.all_gather(r, s, d, n))
        dev.ts_push_events(0, stream.ev_left)
        dev.ts_push_events(1, stream.ev_right)
        maps = []
        for t, stamps, poses, T in ticks:
            dev.ts_render(0, t, download=False)
            dev.ts_render(1, t, download=False)
            dev.set_observation(t, None, None, T)
            dev.comm_shard_tick(t, stamps, poses)
            maps.append(dev.comm_gather_map())
        assert dev.stats().halo_violations == 0
        return maps

    for maps in _run_ranks(world, body):
        for k, mp in enumerate(maps):
            _same(mp, ref[k])
    assert len(ref[-1]) > 50
