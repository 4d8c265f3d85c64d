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
        for d in shards:  # BM + LM of the own slots -> all-gather of the (matched, kept) bytes of the own slots
            d.shard_phase(0, t, stamps, poses)
        codes = _emulated_gather(shards)
        for d in shards:  # frame order from all bytes, own kept points packed -> all-gather of [count | points]
            d.shard_phase(1)
        n_in = ref.stats().last_events_in
        assert shards[0].stats().last_events_in == n_in
        assert codes[0].numel() * 8 == (-(-n_in // G) + 7) // 8 * 8      # ceil(n / G) bytes, whole words
        by = [c.view(torch.uint8).cpu().numpy() for c in codes]
        n_matched = sum(int((b[: len(range(g, n_in, G))] & 1).sum()) for g, b in enumerate(by))
        kept = [int(((b[: len(range(g, n_in, G))] >> 1) & 1).sum()) for g, b in enumerate(by)]
        assert n_matched == ref.stats().last_matches
        assert shards[0].stats().last_matches == ref.stats().last_matches
        assert sum(d.stats().last_solved for d in shards) == ref.stats().last_solved
        pts = _emulated_gather(shards)
        if ref.stats().last_points:
            assert pts[0].numel() == 1 + 13 * max(kept), (pts[0].numel(), kept)   # block = the largest kept count, no more
            assert [int(b[0]) for b in pts] == kept                              # in-band counts
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
