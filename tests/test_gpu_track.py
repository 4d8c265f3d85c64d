"""Tracker residual / Jacobian evaluation on the GPU against the CPU oracle: the images are
integers and every per-point f64 expression is evaluated in the oracle's order, so everything must be bit-exact."""
import numpy as np
import pytest

from esvo_amd import params, rostime

pytestmark = pytest.mark.gpu


def _cayley2rot(c):
    c1, c2, c3 = c
    k = 1 + c1 * c1 + c2 * c2 + c3 * c3
    return np.array([[1 + c1 * c1 - c2 * c2 - c3 * c3, 2 * (c1 * c2 - c3), 2 * (c1 * c3 + c2)],
                     [2 * (c1 * c2 + c3), 1 - c1 * c1 + c2 * c2 - c3 * c3, 2 * (c2 * c3 - c1)],
                     [2 * (c1 * c3 - c2), 2 * (c2 * c3 + c1), 1 - c1 * c1 - c2 * c2 + c3 * c3]]) / k


@pytest.mark.parametrize("preset,rig_fix,stream_fix,n_ev", [("mapping_upenn", "upenn_rig", "upenn_stream", None),
                                                          ("mapping_dsec", "dsec_rig", "dsec_stream", 4000)])
def test_tracker_evaluation_bit_exact(request, preset, rig_fix, stream_fix, n_ev):
    from esvo_amd import lib
    from oracle import oracle as O
    rig, stream = request.getfixturevalue(rig_fix), request.getfixturevalue(stream_fix)
    over = dict(process_event_num=n_ev) if n_ev else {}
    p, _ = params.make_params(params.PRESETS[preset], rig, **over)
    dev = lib.Esvo(p, rig)
    # a few mapper ticks give the local map the tracker registers against (/esvo_mapping/pointcloud_local)
    t_prev = stream.t0_ns
    for k in range(4):
        t = stream.t0_ns + int((0.06 + 0.01 * k) * 1e9)
        for cam in (0, 1):
            dev.ts_push_events(cam, stream.slice(cam, t_prev, t))
        t_prev = t
        ts_left = dev.ts_render(0, t)
        dev.ts_render(1, t, download=False)
        stamps, poses = rostime.pose_table(stream.pose, t, p.bm_half_slice_thickness)
        dev.set_observation(t, None, None, stream.pose(t))
        dev.tick(t, stamps, poses)
    cloud = dev.get_pointcloud()
    assert len(cloud) > 300
    rng = np.random.default_rng(11)
    cloud = cloud[rng.permutation(len(cloud))][:2000]          # the stochastic swaps + MAX_REGISTRATION_POINTS
    T_world_ref = stream.pose(t)
    trk = O.OracleTracker(rig)
    for ksize, from_device in ((5, True), (0, False)):
        trk.set_current(ts_left, ksize)
        dev.track_set_current(None if from_device else ts_left, ksize)   # device-resident left TS / host image
        for a, b in zip(dev.track_images(), trk.images()):
            assert np.array_equal(a, b)
        trk.set_reference(cloud, T_world_ref)
        dev.track_set_reference(cloud, T_world_ref)
        # the current frame a little later: T_ref_left = T_world_ref^-1 T_world_left, then an LM increment x
        T_ref_left = np.linalg.inv(T_world_ref) @ stream.pose(t + 8_000_000)
        R_, t_ = T_ref_left[:3, :3].copy(), T_ref_left[:3, 3].copy()
        hit = 0
        for x in (np.zeros(6), np.array([0.002, -0.001, 0.003, 0.004, -0.002, 0.001])):
            dR = _cayley2rot(x[:3])
            U, _, Vt = np.linalg.svd(R_.T @ dR.T)
            Tw = np.eye(4)
            Tw[:3, :3] = U @ Vt
            Tw[:3, 3] = -Tw[:3, :3] @ (x[3:] + dR @ t_)
            for off, cnt in ((0, 300), (300, 300), (len(cloud) - 100, 300), (len(cloud), 5)):
                for huber in (True, False):
                    g = dev.track_residuals(Tw, off, cnt, huber=huber, huber_threshold=50.0)
                    o = trk.residuals(Tw, off, cnt, huber=huber, huber_threshold=50.0)
                    assert len(g) == len(o) == max(0, min(cnt, len(cloud) - off)) and np.array_equal(g, o)
                    hit += int((o < 255).sum())
        assert hit > 500                                        # most points reproject into the image
        for off, cnt in ((0, 300), (600, 300), (len(cloud) - 7, 300)):
            g, o = dev.track_jacobian(R_, t_, off, cnt), trk.jacobian(R_, t_, off, cnt)
            assert g.shape == o.shape and np.array_equal(g, o)
            assert np.abs(o).max() > 0


def test_tracker_errors(upenn_rig):
    from esvo_amd import lib
    p, _ = params.make_params(params.PRESETS["mapping_upenn"], upenn_rig)
    dev = lib.Esvo(p, upenn_rig)
    with pytest.raises(lib.EsvoError, match="esvo_track_set_current"):
        dev.track_residuals(np.eye(4), 0, 10)
    with pytest.raises(lib.EsvoError, match="esvo_ts_render"):
        dev.track_set_current(None, 5)                          # no device-resident Time Surface yet
    with pytest.raises(lib.EsvoError, match="kernelSize"):
        dev.track_set_current(np.zeros((upenn_rig.height, upenn_rig.width), np.uint8), 15)
    dev.track_set_current(np.zeros((upenn_rig.height, upenn_rig.width), np.uint8), 5)
    assert len(dev.track_residuals(np.eye(4), 0, 10)) == 0      # no reference points yet
