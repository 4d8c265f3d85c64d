"""Tick-interleaved multi-GPU operation on ONE GPU: G handles play the ranks of dist.TickShardedEsvo through the same
device-resident stage calls (esvo_map_front / push_frame_device / fuse_async), the all-gather of the round's frames is a
device copy.  Every tick's DepthMap, on whichever "rank" owns it, must equal the single-handle run bit for bit."""
import numpy as np
import pytest

from esvo_amd import params, rostime

pytestmark = pytest.mark.gpu
F64 = ["inv_depth", "scale2", "nu", "variance", "residual", "x", "p_cam"]


@pytest.mark.parametrize("preset,rig_fix,stream_fix,over,G", [
    ("mapping_upenn", "upenn_rig", "upenn_stream", dict(max_fusion_points=1200), 3),   # CONST_POINTS window
    ("mapping_dsec", "dsec_rig", "dsec_stream", dict(process_event_num=4000), 4),      # 5-frame window < a round + history
])
def test_tick_interleaved_equals_single(request, preset, rig_fix, stream_fix, over, G):
    import torch
    from esvo_amd import dist as edist
    from esvo_amd import lib
    rig, stream = request.getfixturevalue(rig_fix), request.getfixturevalue(stream_fix)
    p, _ = params.make_params(params.PRESETS[preset], rig, **over)
    ref = lib.Esvo(p, rig)
    ranks = [lib.Esvo(p, rig) for _ in range(G)]
    words = 13
    n_ticks = 2 * G + 1                                   # two full rounds and a partial one
    times = [stream.t0_ns + int((0.05 + 0.007 * k) * 1e9) for k in range(n_ticks)]
    for d in [ref] + ranks:
        for cam in (0, 1):
            d.ts_push_events(cam, stream.slice(cam, stream.t0_ns, times[-1] + 1))
    ref_maps = []
    for t in times:
        stamps, poses = rostime.pose_table(stream.pose, t, p.bm_half_slice_thickness)
        ref.ts_render(0, t, download=False); ref.ts_render(1, t, download=False)
        ref.set_observation(t, None, None, stream.pose(t))
        ref.tick(t, stamps, poses)
        ref_maps.append(ref.get_map())
    k = 0
    while k < n_ticks:
        rnd = list(range(k, min(k + G, n_ticks)))
        fronts, tables = {}, {}
        for kk in rnd:                                     # every rank: front stage of its own tick of the round
            d, t = ranks[kk % G], times[kk]
            stamps, poses = rostime.pose_table(stream.pose, t, p.bm_half_slice_thickness)
            tables[kk] = poses
            d.ts_render(0, t, download=False); d.ts_render(1, t, download=False)
            d.set_observation(t, None, None, stream.pose(t))
            fronts[kk] = d.front(t, stamps, poses)
        # "all-gather": every rank gets a copy of every frame of the round
        frames = {kk: edist.device_tensor(ptr, max(n, 1) * words, "<i8")[: n * words].clone() for kk, (ptr, n) in fronts.items()}
        torch.cuda.synchronize()
        for g, d in enumerate(ranks):
            for kk in rnd:
                d.push_frame_device(frames[kk].data_ptr(), fronts[kk][1], tables[kk])
                if kk % G == g:
                    d.fuse_async()
        for kk in rnd:                                     # the owner's DepthMap of tick kk
            got, want = ranks[kk % G].get_map(), ref_maps[kk]
            assert len(got) == len(want), (kk, len(got), len(want))
            for f in ("row", "col", "age"):
                assert np.array_equal(got[f], want[f]), (kk, f)
            for f in F64:
                assert np.array_equal(got[f], want[f]), (kk, f)
        k += G
    assert len(ref_maps[-1]) > 50
