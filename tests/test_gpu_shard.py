"""Sharded == unsharded on ONE GPU: G handles with disjoint event shards / row bands run the three-phase
tick, the two all-gathers between the phases are emulated with torch copies on the same device buffers the real
run hands to RCCL (esvo_shard_exchange).  The merged DepthMap must equal the unsharded one bit for bit (gpurun exposes a
single GPU; the multi-process path itself is covered by tests/test_dist.py with gloo)."""
import numpy as np
import pytest

from esvo_amd import calib, params, rostime, synth

pytestmark = pytest.mark.gpu

F64 = ["inv_depth", "scale2", "nu", "variance", "residual", "x", "p_cam"]


def _emulated_gather(shards):
    """what dist.ShardedEsvo._exchange does with an all-gather: every rank's block into every rank's receive buffer,
    rank-major.  Returns the list of blocks (int64 tensors), or None when nothing is due."""
    import torch
    from esvo_amd import dist as edist
    for d in shards:
        d.synchronize()
    ex = [d.shard_exchange() for d in shards]
    sizes = {e[2] for e in ex}
    assert len(sizes) == 1, sizes  # every rank derives the same block length
    nb = ex[0][2]
    if nb == 0:
        return None
    assert nb % 8 == 0
    blocks = [edist.device_tensor(snd, nb // 8, "<i8").clone() for snd, _, _ in ex]
    for _, rcv, _ in ex:
        for r, blk in enumerate(blocks):
            edist.device_tensor(rcv + r * nb, nb // 8, "<i8").copy_(blk)
    torch.cuda.synchronize()
    return blocks


@pytest.mark.parametrize("preset,rig_fix,stream_fix,n_ev,G,routing", [
    ("mvstereo_upenn", "upenn_rig", "upenn_stream", None, 2, "broadcast"),
    ("mapping_dsec", "dsec_rig", "dsec_stream", 4000, 4, "broadcast"),
    ("mapping_dsec", "dsec_rig", "dsec_stream", 4000, 3, "broadcast"),   # ragged bands
    ("mapping_dsec", "dsec_rig", "dsec_stream", 4000, 8, "broadcast"),
    ("mvstereo_upenn", "upenn_rig", "upenn_stream", None, 2, "y_rect"),
    ("mapping_dsec", "dsec_rig", "dsec_stream", 4000, 3, "y_rect"),
    ("mapping_dsec", "dsec_rig", "dsec_stream", 4000, 8, "y_rect"),      # BASELINE.json's world size: 60-row bands, r = 20 halo
    ("mvstereo_upenn", "upenn_rig", "upenn_stream", None, 8, "y_rect"),  # 260 / 8: ragged bands of 33 and 29 rows
    ("mapping_upenn", "upenn_rig", "upenn_stream", None, 8, "broadcast"),
])
def test_logical_shards_equal_unsharded(request, preset, rig_fix, stream_fix, n_ev, G, routing):
    import torch
    from esvo_amd import dist as edist
    from esvo_amd import lib
    rig, stream = request.getfixturevalue(rig_fix), request.getfixturevalue(stream_fix)
    over = dict(process_event_num=n_ev) if n_ev else {}
    p, _ = params.make_params(params.PRESETS[preset], rig, **over)
    ref = lib.Esvo(p, rig)
    shards = [lib.Esvo(p, rig) for _ in range(G)]
    bands = [edist.band_of(g, G, rig.height) for g in range(G)]
    for g, (d, (y0, y1)) in enumerate(zip(shards, bands)):
        d.set_band(y0, y1, g, G, routing=routing)
    routed = routing == "y_rect"
    if routed:  # the rows a rank renders and ingests: its band + halo, not the image
        rows = [d.shard_rows() for d in shards]
        for (y0, y1), r in zip(bands, rows):
            assert r["observation"][0] <= max(0, y0 - 4) and r["observation"][1] >= min(rig.height, y1 + 4)
            assert r["render"][0] <= r["observation"][0] and r["render"][1] >= r["observation"][1]
        if G == 8:
            assert max(r["render"][1] - r["render"][0] for r in rows) <= rig.height // 8 + 2 * 48, rows
    t_prev = stream.t0_ns
    for k in range(5):
        t = stream.t0_ns + int((0.07 + 0.01 * k) * 1e9)
        stamps, poses = rostime.pose_table(stream.pose, t, p.bm_half_slice_thickness)
        for d in [ref] + shards:
            for cam in (0, 1):
                d.ts_push_events(cam, stream.slice(cam, t_prev, t + 5_000_000))
            d.ts_render(0, t, download=False); d.ts_render(1, t, download=False)
            d.set_observation(t, None, None, stream.pose(t))
        t_prev = t + 5_000_000
        ref.tick(t, stamps, poses)
        for d in shards:  # BM + LM of the own events -> all-gather of their (matched, kept) bits
            d.shard_phase(0, t, stamps, poses)
        codes = _emulated_gather(shards)
        for d in shards:  # frame order from all bits, own kept points packed -> all-gather of [count | points]
            d.shard_phase(1)
        n_in = ref.stats().last_events_in
        assert shards[0].stats().last_events_in == n_in
        by = [c.view(torch.uint8).cpu().numpy() for c in codes]
        if routed:
            assert codes[0].numel() * 8 == (-(-n_in // 16) * 4 + 7) // 8 * 8  # two bits per slot of the whole tick, whole words
            bits = [np.stack([(b >> (2 * q)) & 3 for q in range(4)], 1).reshape(-1)[:n_in] for b in by]
            assert all(int(((bits[a] != 0) & (bits[b] != 0)).sum()) == 0 for a in range(G) for b in range(a))  # one owner per slot
            n_matched = sum(int((c & 1).sum()) for c in bits)
            kept = [int(((c >> 1) & 1).sum()) for c in bits]
        else:
            assert codes[0].numel() * 8 == (-(-n_in // G) + 7) // 8 * 8      # ceil(n / G) bytes, whole words
            n_matched = sum(int((b[: len(range(g, n_in, G))] & 1).sum()) for g, b in enumerate(by))
            kept = [int(((b[: len(range(g, n_in, G))] >> 1) & 1).sum()) for g, b in enumerate(by)]
        assert n_matched == ref.stats().last_matches
        assert shards[0].stats().last_matches == ref.stats().last_matches
        assert sum(d.stats().last_solved for d in shards) == ref.stats().last_solved
        pts = _emulated_gather(shards)
        if ref.stats().last_points:
            assert pts[0].numel() == 1 + 13 * max(kept), (pts[0].numel(), kept)   # block = the largest kept count, no more
            assert [int(b[0]) for b in pts] == kept                              # in-band counts (high half: halo violations, none)
            assert sum(kept) == ref.stats().last_points
        for d in shards:
            d.shard_phase(2)
        assert shards[0].stats().last_points == ref.stats().last_points
        merged = edist.merge_band_maps([d.get_map() for d in shards])
        full = ref.get_map()
        assert len(merged) == len(full), (k, len(merged), len(full))
        for f in ("row", "col", "age"):
            assert np.array_equal(merged[f], full[f]), (k, f)
        for f in F64:
            assert np.array_equal(merged[f], full[f]), (k, f)
        # every element is exported by the rank that owns its TRUE cell; believed rows may differ by one
    assert len(full) > 50
    if routed:
        assert all(d.stats().halo_violations == 0 for d in shards)
        staged = [int(d.stats().events_staged[0]) for d in shards]
        assert sum(staged) >= int(ref.stats().events_staged[0])       # every event is kept by the owner of its row (+ halos)
        if G == 8:
            assert max(staged) < 0.6 * int(ref.stats().events_staged[0]), staged


@pytest.mark.parametrize("name,G", [("rpg", 8), ("hkust", 8), ("hkust", 3)])
def test_denoising_rigs_routed_by_rows(name, G):
    """The small DAVIS configurations switch Denoising on (esvo_Mapping.cpp:1046-1072: 3 x 3 median of the selected events' map,
    events off the mask dropped BEFORE the thread-stride deal).  Routed by rows since round 6: a rank decides the mask's verdict
    for the events whose raw row lies in its band (its ring holds one more row on either side), phase 0 returns ESVO_AGAIN with
    one bit per selected event to all-gather, and its second part matches the kept sequence -- which events, which order, how
    many -- of the unsharded tick.  rpg 240 x 180 and hkust 346 x 260 (ragged bands) at BASELINE.json's world size."""
    from esvo_amd import dist as edist
    from esvo_amd import lib
    from tests import scenarios
    sc = scenarios.Scenario(name)
    rig, p, stream, spec = sc.rig, sc.params, sc.stream(), sc.spec
    assert p.denoising
    ref = lib.Esvo(p, rig)
    shards = [lib.Esvo(p, rig) for _ in range(G)]
    for g, d in enumerate(shards):
        y0, y1 = edist.band_of(g, G, rig.height)
        d.set_band(y0, y1, g, G, routing="y_rect")
    t_prev = stream.t0_ns
    kept_seen = []
    for k in range(spec["n_ticks"]):
        t = stream.t0_ns + int(round((spec["t_first"] + spec["dt"] * k) * 1e9))
        stamps, poses = rostime.pose_table(stream.pose, t, p.bm_half_slice_thickness)
        for d in [ref] + shards:
            for cam in (0, 1):
                d.ts_push_events(cam, stream.slice(cam, t_prev, t + 2_000_000))
            d.ts_render(0, t, download=False); d.ts_render(1, t, download=False)
            d.set_observation(t, None, None, stream.pose(t))
        t_prev = t + 2_000_000
        ref.tick(t, stamps, poses)
        again = [d.shard_phase(0, t, stamps, poses) for d in shards]
        assert all(again)                               # every rank asks for the exchange of the mask bits
        bits = _emulated_gather(shards)
        assert bits is not None
        assert not any(d.shard_phase(0) for d in shards)
        _emulated_gather(shards)
        for d in shards:
            d.shard_phase(1)
        _emulated_gather(shards)
        for d in shards:
            d.shard_phase(2)
        rs = ref.stats()
        for d in shards:
            ds = d.stats()
            assert (ds.last_events_in, ds.last_matches, ds.last_points) == (rs.last_events_in, rs.last_matches, rs.last_points), k
        kept_seen.append(int(rs.last_events_in))
        merged = edist.merge_band_maps([d.get_map() for d in shards])
        full = ref.get_map()
        assert len(merged) == len(full), (k, len(merged), len(full))
        for f in ("row", "col", "age"):
            assert np.array_equal(merged[f], full[f]), (k, f)
        for f in F64:
            assert np.array_equal(merged[f], full[f]), (k, f)
    assert len(full) > 50 and max(kept_seen) < p.process_event_num and min(kept_seen) > 100   # the mask really dropped events
    assert all(d.stats().halo_violations == 0 for d in shards)


@pytest.mark.parametrize("cull_all", [False, True])
def test_routed_band_detects_a_refinement_that_leaves_its_rows(dsec_rig, dsec_stream, cull_all):
    """The guard of the routed mode: with the smallest halo and a camera that moves VERTICALLY between an event and the
    observation, refinements warp their patches out of the rows the rank renders.  That must be counted (by every rank: the
    count travels with the second exchange) and the next tick refused with ESVO_ERR_HALO -- never a silently different map.
    cull_all: an inverse-depth range nothing passes, so NO point is kept anywhere -- the second exchange then carries the count
    words alone and the violations still reach every rank (round 5 skipped the exchange in that case)."""
    import torch  # noqa: F401
    from esvo_amd import dist as edist
    from esvo_amd import lib
    rig, stream = dsec_rig, dsec_stream
    over = dict(invdepth_min=100.0, invdepth_max=101.0) if cull_all else {}   # (pointCulling's range test: no refined point passes)
    p, _ = params.make_params(params.PRESETS["mapping_dsec"], rig, process_event_num=4000, **over)
    G = 4
    shards = [lib.Esvo(p, rig) for _ in range(G)]
    for g, d in enumerate(shards):
        y0, y1 = edist.band_of(g, G, rig.height)
        d.set_band(y0, y1, g, G, routing="y_rect", ts_halo_rows=5)

    def pose(t_ns):  # the stream's pose + a fast vertical translation: ~40 rows over the 10 ms a tick looks back at these depths
        T = np.array(stream.pose(t_ns), np.float64).reshape(4, 4).copy()
        T[1, 3] += 60.0 * (t_ns - stream.t0_ns) * 1e-9
        return T

    refused = False
    t_prev = stream.t0_ns
    for k in range(5):
        t = stream.t0_ns + int((0.07 + 0.01 * k) * 1e9)
        stamps, poses = rostime.pose_table(pose, t, p.bm_half_slice_thickness)
        for d in shards:
            for cam in (0, 1):
                d.ts_push_events(cam, stream.slice(cam, t_prev, t + 5_000_000))
            d.ts_render(0, t, download=False); d.ts_render(1, t, download=False)
            d.set_observation(t, None, None, pose(t))
        t_prev = t + 5_000_000
        try:
            for d in shards:
                d.shard_phase(0, t, stamps, poses)
        except lib.EsvoError as e:
            assert "(-7)" in str(e) and e.code == lib.ERR_HALO, e   # ESVO_ERR_HALO
            refused = True
            break
        _emulated_gather(shards)
        for d in shards:
            d.shard_phase(1)
        _emulated_gather(shards)
        for d in shards:
            d.shard_phase(2)
    viol = [int(d.stats().halo_violations) for d in shards]
    assert refused and min(viol) > 0 and len(set(viol)) == 1, (refused, viol)   # every rank holds the same total
    if cull_all:
        assert all(int(d.stats().total_points) == 0 for d in shards)
