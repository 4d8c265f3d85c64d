"""CPU self-consistency tests of the tracker restatement (oracle, RegProblemLM.cpp / SURVEY.md section 8(f).1): the reference
ships no golden vectors, so the restatement is pinned against independent formulations (scipy's Sobel, a direct bilinear
fetch, central differences of the residual through the reference's own warping parametrisation)."""
import numpy as np
import pytest
from scipy import ndimage

from esvo_amd import calib


def _oracle():
    from oracle import oracle as O
    O.build()
    return O


def cayley2rot(c):
    """esvo_core/src/tools/cayley.cpp (Cayley parameters -> rotation)"""
    c1, c2, c3 = c
    k = 1 + c1 * c1 + c2 * c2 + c3 * c3
    R = np.array([[1 + c1 * c1 - c2 * c2 - c3 * c3, 2 * (c1 * c2 - c3), 2 * (c1 * c3 + c2)],
                  [2 * (c1 * c2 + c3), 1 - c1 * c1 + c2 * c2 - c3 * c3, 2 * (c2 * c3 - c1)],
                  [2 * (c1 * c3 - c2), 2 * (c2 * c3 + c1), 1 - c1 * c1 - c2 * c2 + c3 * c3]])
    return R / k


def warping_transformation(R_, t_, x):
    """RegProblemLM::getWarpingTransformation (RegProblemLM.cpp:328-351)"""
    dR = cayley2rot(x[:3])
    U, _, Vt = np.linalg.svd(R_.T @ dR.T)
    Rc = U @ Vt
    T = np.eye(4)
    T[:3, :3] = Rc
    T[:3, 3] = -Rc @ (x[3:] + dR @ t_)
    return T


def test_negative_and_sobel_images():
    O = _oracle()
    rig = calib.ideal_rig(64, 48, 80.0, 0.1)
    rng = np.random.default_rng(3)
    ts = rng.integers(0, 256, (48, 64)).astype(np.uint8)
    trk = O.OracleTracker(rig)
    for k in (0, 5):
        trk.set_current(ts, k)
        neg, du, dv = trk.images()
        blur = ts if k == 0 else O.gaussian5(ts)
        assert np.array_equal(neg, 255 - blur)
        f = neg.astype(np.float64)
        assert np.array_equal(du, ndimage.sobel(f, axis=1, mode="mirror"))   # cv::Sobel dx=1, BORDER_REFLECT_101
        assert np.array_equal(dv, ndimage.sobel(f, axis=0, mode="mirror"))
    with pytest.raises(ValueError):
        trk.set_current(ts, 15)


def test_residuals_are_bilinear_fetches_with_huber():
    O = _oracle()
    rig = calib.ideal_rig(120, 100, 90.0, 0.1)
    rng = np.random.default_rng(4)
    ts = rng.integers(0, 256, (100, 120)).astype(np.uint8)
    trk = O.OracleTracker(rig)
    trk.set_current(ts, 5)
    neg = trk.images()[0].astype(np.float64)
    # reference frame = world; points at depth z that project to chosen (sub)pixel positions
    uv = np.array([[10.0, 20.0], [57.25, 33.5], [118.4, 98.3], [119.3, 50.0], [-3.0, 10.0], [30.0, 99.5], [60.0, 40.0]])
    z = np.array([2.0, 3.0, 1.5, 2.0, 2.0, 2.0, -1.0])
    P = np.asarray(rig.left.P, np.float64).reshape(3, 4)
    xyz = np.stack([(uv[:, 0] - P[0, 2]) / P[0, 0] * z, (uv[:, 1] - P[1, 2]) / P[1, 1] * z, z], axis=1)
    trk.set_reference(xyz, np.eye(4))
    r = trk.residuals(np.eye(4), 0, len(xyz), huber=False)
    # independent evaluation: project the float32 points (pcl::PointXYZ) and fetch bilinearly
    q = xyz.astype(np.float32).astype(np.float64)
    u = (P[0, 0] * q[:, 0] + P[0, 2] * q[:, 2]) / q[:, 2]
    v = (P[1, 1] * q[:, 1] + P[1, 2] * q[:, 2]) / q[:, 2]

    def bil(x, y):
        x0, y0 = int(np.floor(x)), int(np.floor(y))
        a, b = x - x0, y - y0
        return (1 - b) * ((1 - a) * neg[y0, x0] + a * neg[y0, x0 + 1]) + b * ((1 - a) * neg[y0 + 1, x0] + a * neg[y0 + 1, x0 + 1])
    for i in (0, 1, 2):
        assert abs(r[i] - bil(u[i], v[i])) < 1e-9, i
    assert r[3] == 255.0                    # x > W-1: fails isValidPatch
    assert r[4] == 255.0 and r[5] == 255.0  # outside the image
    # (point 6 is behind the camera: its projection mirrors through the centre; a plain fetch or 255 like any other)
    rh = trk.residuals(np.eye(4), 0, len(xyz), huber=True, huber_threshold=50.0)
    big = r > 50.0
    assert big.any() and np.allclose(rh[big], np.sqrt(50.0 / r[big]) * r[big], rtol=0, atol=1e-12) and np.array_equal(rh[~big], r[~big])
    # batches: offset/count are clamped like setStochasticSampling
    assert len(trk.residuals(np.eye(4), 5, 300)) == 2 and len(trk.residuals(np.eye(4), 7, 10)) == 0


def test_analytical_jacobian_matches_central_differences():
    """df() linearises r(x) = TS_negative(pi(T_warp(x) p)) at x = 0 (Cayley rotation + translation of the reference
    frame).  On a linear image (Sobel/8 = exact slope, bilinear = exact) the analytical Jacobian must equal central
    differences of the residual taken through getWarpingTransformation."""
    O = _oracle()
    rig = calib.ideal_rig(120, 100, 90.0, 0.1)
    yy, xx = np.mgrid[0:100, 0:120]
    neg = (xx + yy).astype(np.uint8)                    # <= 218, slopes (1, 1)
    trk = O.OracleTracker(rig)
    trk.set_current((255 - neg).astype(np.uint8), 0)
    rng = np.random.default_rng(5)
    n = 40
    uv = np.stack([rng.uniform(25, 95, n), rng.uniform(25, 75, n)], axis=1)
    z = rng.uniform(1.0, 4.0, n)
    P = np.asarray(rig.left.P, np.float64).reshape(3, 4)
    p_ref = np.stack([(uv[:, 0] - P[0, 2]) / P[0, 0] * z, (uv[:, 1] - P[1, 2]) / P[1, 1] * z, z], axis=1)
    trk.set_reference(p_ref, np.eye(4))
    # The reference's df() evaluates dPi_dT at the point in the REFERENCE frame (ri.p_, RegProblemLM.cpp:224-229), so
    # it is the exact derivative for T_ref_left = identity and a few-percent approximation for a small motion.
    for R_, t_, tol in ((np.eye(3), np.zeros(3), 1e-5),
                        (cayley2rot(np.array([0.01, -0.02, 0.015])), np.array([0.03, -0.02, 0.05]), 0.08)):
        J = trk.jacobian(R_, t_, 0, n)
        assert J.shape == (n, 6)
        h = 1e-6
        Jn = np.zeros_like(J)
        for j in range(6):
            e = np.zeros(6); e[j] = h
            rp = trk.residuals(warping_transformation(R_, t_, e), 0, n, huber=False)
            rm = trk.residuals(warping_transformation(R_, t_, -e), 0, n, huber=False)
            Jn[:, j] = (rp - rm) / (2 * h)
        assert np.all(trk.residuals(warping_transformation(R_, t_, np.zeros(6)), 0, n, huber=False) < 255)
        scale = np.abs(Jn).max()
        assert np.abs(J - Jn).max() <= tol * scale, (tol, np.abs(J - Jn).max(), scale)
        assert scale > 10  # not trivially zero
