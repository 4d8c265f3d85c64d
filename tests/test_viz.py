"""Debug image publishers and global-cloud helpers: Visualization::plot_map / DrawPoint
(esvo_core/src/tools/Visualization.cpp:13-94) and the voxel filter of publishPointCloud (esvo_Mapping.cpp:956-977)."""
import os

import numpy as np
import pytest

from esvo_amd import params, rostime

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def test_jet_colours_are_the_reference_tables():
    """tests/golden/jet256.npy = the bytes the reference's own r/g/b tables produce (tests/golden/make_jet_fixture.py)"""
    from oracle import oracle as O
    assert np.array_equal(O.jet_bgr(), np.load(os.path.join(GOLDEN, "jet256.npy")))


def test_voxel_filter_host_helper_matches_the_oracle():
    from esvo_amd import lib
    from oracle import oracle as O
    rng = np.random.default_rng(5)
    pts = rng.normal(0, 3.0, size=(20000, 3)).astype(np.float32)
    pts[::1000] = np.nan                                        # non-finite points are skipped (pcl: is_dense == false path)
    for leaf in (0.3, 0.03, 1.7):
        a, b = lib.voxel_filter(pts, leaf), O.voxel_filter(pts, leaf)
        assert len(a) == len(b) and 0 < len(a) < len(pts) and np.array_equal(a, b)
    # known answer: two points of one voxel average, voxels come out in ascending x-fastest index
    kat = lib.voxel_filter(np.array([[1, 1, 1], [0.1, 0.1, 0.1], [0.2, 0.2, 0.25], [-0.5, 0, 0]], np.float32), 0.3)
    assert np.allclose(kat, [[-0.5, 0, 0], [0.15, 0.15, 0.175], [1, 1, 1]])
    assert len(lib.voxel_filter(np.zeros((0, 3), np.float32), 0.3)) == 0


@pytest.mark.gpu
@pytest.mark.parametrize("preset,rig_fix,stream_fix,n_ev", [("mapping_dsec", "dsec_rig", "dsec_stream", 4000),
                                                            ("mvstereo_upenn", "upenn_rig", "upenn_stream", None)])
def test_debug_images_and_near_cloud_equal_the_oracle(request, preset, rig_fix, stream_fix, n_ev):
    from esvo_amd import lib
    from oracle import oracle as O
    rig, stream = request.getfixturevalue(rig_fix), request.getfixturevalue(stream_fix)
    over = dict(process_event_num=n_ev) if n_ev else {}
    p, _ = params.make_params(params.PRESETS[preset], rig, **over)
    dev = lib.Esvo(p, rig)
    m = O.OracleMapper(p, rig)
    m.set_mode(True, True)
    ots = [O.OracleTS(rig.width, rig.height), O.OracleTS(rig.width, rig.height)]
    dev.ts_push_events(0, stream.ev_left)
    dev.ts_push_events(1, stream.ev_right)
    ots[0].push(stream.ev_left)
    ots[1].push(stream.ev_right)
    for k in range(6):
        t = stream.t0_ns + int((0.06 + 0.01 * k) * 1e9)
        l = ots[0].render(t, map_x=rig.left.map_x, map_y=rig.left.map_y)
        r = ots[1].render(t, map_x=rig.right.map_x, map_y=rig.right.map_y)
        stamps, poses = rostime.pose_table(stream.pose, t, p.bm_half_slice_thickness)
        m.set_observation(t, l, r, stream.pose(t))
        m.set_poses(stamps, poses)
        m.tick(stream.ev_left[O.select_events(stream.ev_left, t, p.bm_half_slice_thickness, p.process_event_num)])
        dev.ts_render(0, t, download=False)
        dev.ts_render(1, t, download=False)
        dev.set_observation(t, None, None, stream.pose(t))
        dev.tick(t, stamps, poses)
    imgs = dev.get_debug_images(age_max_range=10.0)
    for img, kind in zip(imgs, (0, 1, 3, 2)):                   # inverse depth, standard deviation, age, cost
        ref = m.debug_image(kind, 10.0)
        assert ref.any(), kind
        assert np.array_equal(img, ref), kind
    for rng_ in (2.5, 50.0):
        a, b = dev.get_pointcloud_near(rng_), m.get_pointcloud_near(rng_)
        assert len(a) == len(b) and np.array_equal(a, b)
    assert len(dev.get_pointcloud_near(1e9)) == len(dev.get_pointcloud())
    # esvo_MVStereo::saveDepthMap (esvo_MVStereo.cpp:982-1000): "<x> <y> <depth>" per valid element, list order; the 1 x 2 row vector
    # in Eigen's default format (6 significant digits, both coefficients right-aligned to the longer one), the depth as the stream prints it
    import tempfile
    with tempfile.TemporaryDirectory() as d:
        n = dev.save_depth_map(d + os.sep, 1234567890123)
        lines = open(os.path.join(d, "1234567890123.txt")).read().split("\n")
    mp = m.get_map()
    valid = mp[mp["inv_depth"] > -1e-6]
    assert n == len(valid) > 50 and lines[-1] == "" and len(lines) == n + 1
    for ln, e in zip(lines, valid):
        a, b = "%g" % e["x"][0], "%g" % e["x"][1]
        w = max(len(a), len(b))
        assert ln == f"{a:>{w}} {b:>{w}} {'%g' % e['p_cam'][2]}", (ln, e)
