"""Sharded == unsharded on ONE GPU: G handles with disjoint event shards / row bands run the three-phase
tick, the two sums between the phases are emulated with torch ops on the same device buffers the real
run hands to RCCL (esvo_shard_exchange).  The merged DepthMap must equal the unsharded one bit for bit (gpurun exposes a
single GPU; the multi-process path itself is covered by tests/test_dist.py with gloo)."""
import numpy as np
import pytest

from esvo_amd import calib, params, rostime, synth

pytestmark = pytest.mark.gpu

F64 = ["inv_depth", "scale2", "nu", "variance", "residual", "x", "p_cam"]


def _emulated_sum(shards, whole_words=False):
    """what dist.ShardedEsvo._exchange does with an all-reduce: 64-bit integer SUM of the ranks' buffers"""
    import torch
    from esvo_amd import dist as edist
    for d in shards:
        d.synchronize()
    bufs = []
    for d in shards:
        ptr, nbytes = d.shard_exchange()
        bufs.append(edist.device_tensor(ptr, nbytes // 8, "<i8") if nbytes else None)
    sizes = {0 if b is None else b.numel() for b in bufs}
    assert len(sizes) == 1, sizes  # every rank derives the same size
    if bufs[0] is None:
        return None
    stack = torch.stack(bufs)
    by = stack.view(torch.uint8).view(len(bufs), -1)
    if not whole_words:
        assert int(((by != 0).sum(0) > 1).sum()) == 0  # byte-wise disjoint: the word sums cannot carry
    else:
        assert int(((stack.view(len(bufs), -1, 13) != 0).any(2).sum(0) > 1).sum()) == 0  # every point from ONE shard
    tot = stack.sum(0)
    for b in bufs:
        b.copy_(tot)
    torch.cuda.synchronize()
    return tot


@pytest.mark.parametrize("preset,rig_fix,stream_fix,n_ev,G", [
    ("mvstereo_upenn", "upenn_rig", "upenn_stream", None, 2),
    ("mapping_dsec", "dsec_rig", "dsec_stream", 4000, 4),
    ("mapping_dsec", "dsec_rig", "dsec_stream", 4000, 3),   # ragged bands
])
def test_logical_shards_equal_unsharded(request, preset, rig_fix, stream_fix, n_ev, G):
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
        d.set_band(y0, y1, g, G)
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
        for d in shards:  # BM + LM of the own slots -> sum of the (matched, kept) bytes
            d.shard_phase(0, t, stamps, poses)
        codes = _emulated_sum(shards)
        for d in shards:  # frame order, own points placed -> sum of the frame
            d.shard_phase(1)
        assert shards[0].stats().last_events_in == ref.stats().last_events_in
        assert codes.numel() * 8 == (ref.stats().last_events_in + 7) // 8 * 8
        assert shards[0].stats().last_matches == ref.stats().last_matches
        assert sum(d.stats().last_solved for d in shards) == ref.stats().last_solved
        _emulated_sum(shards, whole_words=True)
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
