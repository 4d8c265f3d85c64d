"""Sharded == unsharded on ONE GPU: G handles with disjoint row bands run the four-phase tick, the
collectives between the phases are emulated with torch ops on the same device buffers the real
run hands to RCCL.  The merged DepthMap must equal the unsharded one bit for bit (gpurun exposes a
single GPU; the multi-process path itself is covered by tests/test_dist.py with gloo)."""
import numpy as np
import pytest

from esvo_amd import calib, params, rostime, synth

pytestmark = pytest.mark.gpu

F64 = ["inv_depth", "scale2", "nu", "variance", "residual", "x", "p_cam"]


def _views(dev):
    from esvo_amd import dist as edist
    b = dev.shard_buffers()
    return dict(mflags=edist.device_tensor(b.d_match_flags, b.max_events, "<i4"),
                pflags=edist.device_tensor(b.d_point_flags, b.max_events, "<i4"),
                pslots=edist.device_tensor(b.d_point_slots, b.max_events * edist.POINT_WORDS, "<i8"),
                valid=edist.device_tensor(b.d_reg_valid, b.n_cells, "|u1"),
                ab=edist.device_tensor(b.d_reg_ab, b.n_cells * 2, "<f8"),
                cd=edist.device_tensor(b.d_reg_cd, b.n_cells * 2, "<f8"))


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
    views = [_views(d) for d in shards]
    W = rig.width
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
        # phase 0 + "all-reduce" of the match flags
        for d in shards:
            d.shard_phase(0, t, stamps, poses)
        n = shards[0].stats().last_events_in
        assert n == ref.stats().last_events_in
        for d in shards:
            d.synchronize()
        tot = sum(v["mflags"][:n] for v in views)
        assert int(tot.max()) <= 1  # the event shards are disjoint
        for v in views:
            v["mflags"][:n] = tot
        torch.cuda.synchronize()
        # phase 1 + "all-reduce" of point flags / slots
        for d in shards:
            d.shard_phase(1)
        m = shards[0].stats().last_matches
        assert m == ref.stats().last_matches
        tf = sum(v["pflags"][:m] for v in views)
        ts = sum(v["pslots"][: m * edist.POINT_WORDS] for v in views)
        for v in views:
            v["pflags"][:m] = tf
            v["pslots"][: m * edist.POINT_WORDS] = ts
        torch.cuda.synchronize()
        # phase 2 + "all-gather" of the regulariser view bands
        for d in shards:
            d.shard_phase(2)
            d.synchronize()
        if p.regularization:
            for key, per in (("valid", 1), ("ab", 2), ("cd", 2)):
                for g, (y0, y1) in enumerate(bands):
                    src = views[g][key][y0 * W * per:y1 * W * per].clone()
                    for v in views:
                        v[key][y0 * W * per:y1 * W * per] = src
            torch.cuda.synchronize()
        for d in shards:
            d.shard_phase(3)
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
