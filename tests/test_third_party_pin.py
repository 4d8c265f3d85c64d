"""Two-source agreement for the OpenCV arithmetic the hot path depends on ("parity unpinned").

oracle/esvo_oracle.cpp restates cv::remap / GaussianBlur / medianBlur in closed form; tests/opencv_restated.py restates them a
second time from the structure of OpenCV's implementation (weight table in shorts, 8.8 fixed-point two-pass blur).  They must agree
bit for bit -- on random images, on the shipped rigs' rectification maps, at the image border and at the table's one saturated
entry.  (The device is compared with the oracle on the GPU: tests/test_gpu_parity.py.)"""
import numpy as np
import pytest

import opencv_restated as cvr
from esvo_amd import calib
from oracle import oracle as O


def _images(W, H, seed):
    rng = np.random.default_rng(seed)
    yield rng.integers(0, 256, (H, W)).astype(np.uint8)
    img = np.zeros((H, W), np.uint8)
    img[rng.integers(0, H, 4000), rng.integers(0, W, 4000)] = rng.integers(1, 256, 4000)   # sparse, like a Time Surface
    yield img
    yield np.full((H, W), 255, np.uint8)
    yield (np.add.outer(np.arange(H), np.arange(W)) % 256).astype(np.uint8)


def test_weight_table_is_the_closed_form_except_its_one_saturated_entry():
    tab = cvr.bilinear_tab_i()
    fy, fx = np.meshgrid(np.arange(32), np.arange(32), indexing="ij")
    closed = np.stack([np.stack([(32 - fx) * (32 - fy) * 32, fx * (32 - fy) * 32], -1),
                       np.stack([(32 - fx) * fy * 32, fx * fy * 32], -1)], -2)       # [fy, fx, k1, k2]
    diff = tab != closed
    assert diff.sum() == 1 and tab[0, 0, 0, 0] == 32767 and closed[0, 0, 0, 0] == 32768
    p = np.arange(256)
    assert np.array_equal((32767 * p + 16384) >> 15, p)          # ... which cannot change an 8-bit result
    assert (tab.sum((2, 3)) == 32768).sum() == 1023


@pytest.mark.parametrize("rig_name", ["upenn", "dsec", "rpg", "hkust"])
def test_remap_two_sources_agree_on_the_shipped_rigs(rig_name):
    rig = calib.dataset_rig(rig_name)
    for cam in (rig.left, rig.right):
        for k, img in enumerate(_images(rig.width, rig.height, 3)):
            a = O.remap_bilinear(img, cam.map_x, cam.map_y)
            b = cvr.cv_remap_linear_u8(img, cam.map_x, cam.map_y)
            assert np.array_equal(a, b), (rig_name, k, int(np.count_nonzero(a != b)))


def test_remap_two_sources_agree_on_random_maps_and_at_the_border():
    W, H = 173, 131
    rng = np.random.default_rng(9)
    xs, ys = np.meshgrid(np.arange(W, dtype=np.float32), np.arange(H, dtype=np.float32))
    cases = [(xs + rng.uniform(-3, 3, (H, W)).astype(np.float32), ys + rng.uniform(-3, 3, (H, W)).astype(np.float32)),
             (xs * np.float32(1.3) - np.float32(20.25), ys * np.float32(1.2) - np.float32(15.5)),     # leaves the image on every side
             (xs + np.float32(0.5), ys + np.float32(0.5)),                                           # the half-pixel ties of cvRound
             (xs + np.float32(1 / 64), ys - np.float32(1 / 64)),                                     # exactly between two table entries
             (xs, ys)]                                                                               # identity: the saturated entry everywhere
    for k, (mx, my) in enumerate(cases):
        for img in _images(W, H, 20 + k):
            a = O.remap_bilinear(img, np.ascontiguousarray(mx), np.ascontiguousarray(my))
            b = cvr.cv_remap_linear_u8(img, mx, my)
            assert np.array_equal(a, b), (k, int(np.count_nonzero(a != b)))
    img = next(_images(W, H, 1))
    assert np.array_equal(cvr.cv_remap_linear_u8(img, xs, ys), img)


@pytest.mark.parametrize("shape", [(346, 260), (640, 480), (17, 9), (5, 3)])
def test_gaussian_and_median_two_sources_agree(shape):
    W, H = shape
    for img in _images(W, H, 40 + W):
        g1, g2 = O.gaussian5(img), cvr.cv_gaussian5_u8(img)
        assert np.array_equal(g1, g2), int(np.count_nonzero(g1 != g2))
        m1, m2 = O.median3(img), cvr.cv_median3_u8(img)
        assert np.array_equal(m1, m2), int(np.count_nonzero(m1 != m2))


def test_time_surface_raster_two_sources_agree():
    """the whole OpenCV tail of createTimeSurfaceAtTime (TimeSurface.cpp:123-149): convertTo(CV_8U), medianBlur(3), remap -- the
    oracle's render against the reference class's f64 image pushed through the second restatement"""
    from esvo_amd import synth
    rig = calib.dataset_rig("upenn")
    st = synth.make_stream(rig, 6000, 0.1, 0.16, 1.0, seed=4)
    ts = O.OracleTS(rig.width, rig.height)
    ts.push(st.ev_left)
    t = st.t0_ns + 90_000_000
    out, pre = ts.render(t, decay_ms=30.0, ignore_polarity=True, median_k=1, map_x=rig.left.map_x, map_y=rig.left.map_y, want_prefilter=True)
    again = cvr.cv_remap_linear_u8(cvr.cv_median3_u8(pre), rig.left.map_x, rig.left.map_y)
    assert np.array_equal(out, again) and int(np.count_nonzero(out)) > 1000
