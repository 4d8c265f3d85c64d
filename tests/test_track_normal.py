"""The tracker's normal equations and its host-side optimiser (verdict round 3, item 8).

  orc_tracker_normal_equations   H = J^T J, b = J^T f, |f|^2 of one iteration in the DEVICE's summation order, checked against the
                                 products of the residuals / Jacobian the reference's own RegProblemLM.cpp produced
                                 (tests/golden/ref_track.npz: compiled from source), to rounding (the summation order differs)
  esvo_hip::gauss_newton_register (include/esvo_hip.hpp, the C++ driver) fed by the oracle, against the same damped Gauss-Newton
                                 written in numpy: the host code of esvo_track_register, exercised without a GPU
  [gpu] esvo_track_normal_equations == the oracle bit for bit; esvo_track_register == the Python loop over
                                 esvo_track_residuals / esvo_track_jacobian (round 3's driver)
"""
import os
import struct
import subprocess

import numpy as np
import pytest

from esvo_amd import calib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, "tests", "golden")


def _problem():
    from oracle import oracle as O
    g = np.load(os.path.join(GOLDEN, "ref_track.npz"))
    rig = calib.dataset_rig("upenn")
    n, order = int(g["n"]), g["order"]
    trk = O.OracleTracker(rig)
    trk.set_current(g["ts_left"], 5)
    xyz = g["xyz_world"][order][:n]
    trk.set_reference(xyz, g["T_world_ref"])
    return g, rig, trk, xyz


def test_oracle_normal_equations_are_the_products_of_the_reference_functor():
    g, rig, trk, _ = _problem()
    # the fixture's batch [100, 400) at x = 0: f = huber_f0, J = huber_J, both from RegProblemLM.cpp compiled unmodified
    H, b, cost, n = trk.normal_equations(g["R"], g["t"], 100, 300, huber=True, huber_threshold=50.0)
    f, J = g["huber_f0"], g["huber_J"]
    assert n == len(f) == 300 and np.array_equal(trk.jacobian(g["R"], g["t"], 100, 300), J)
    assert np.allclose(H, J.T @ J, rtol=1e-12, atol=0) and np.allclose(b, J.T @ f, rtol=1e-11, atol=1e-9 * np.abs(J.T @ f).max())
    assert abs(cost - f @ f) <= 1e-12 * (f @ f)
    assert np.array_equal(H, H.T) and np.linalg.eigvalsh(H).min() > 0
    # the l2 norm, a batch cut short by the point count, an empty batch
    H2, b2, c2, n2 = trk.normal_equations(g["R"], g["t"], 600, 300, huber=False)
    assert n2 == 100 and np.allclose(b2, g["l2_J"][:0].T @ np.zeros(0)) is not None
    f2 = trk.residuals(np.linalg.inv(np.block([[g["R"], g["t"].reshape(3, 1)], [np.zeros((1, 3)), np.ones((1, 1))]])), 600, 300, huber=False)
    J2 = trk.jacobian(g["R"], g["t"], 600, 300)
    assert np.allclose(H2, J2.T @ J2, rtol=1e-12) and abs(c2 - f2 @ f2) <= 1e-12 * (f2 @ f2)
    H3, b3, c3, n3 = trk.normal_equations(g["R"], g["t"], 5000, 300)
    assert n3 == 0 and not H3.any() and not b3.any() and c3 == 0.0


def _numpy_register(trk, n, R_, t_, iters):
    from esvo_amd.closed_loop import lm_gn_loop

    def evaluate(R, t):
        Tlr = np.eye(4)
        Tlr[:3, :3] = R.T
        Tlr[:3, 3] = -R.T @ t
        r = trk.residuals(Tlr, 0, n, huber=True, huber_threshold=50.0)
        J = trk.jacobian(R, t, 0, n)
        return J.T @ J, J.T @ r, float(r @ r), len(r)
    R, t, _, it = lm_gn_loop(evaluate, R_, t_, iters)
    return R, t, it


def test_cpp_gauss_newton_driver_equals_the_numpy_loop(tmp_path):
    g, rig, trk, xyz = _problem()
    exe = str(tmp_path / "gn_driver_oracle")
    subprocess.check_call(["g++", "-std=c++14", "-O2", "-ffp-contract=off", "-Wall", "-I", os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "tests", "cpp", "gn_driver_oracle.cpp"), "-o", exe,
                           "-L", os.path.join(ROOT, "oracle"), "-lesvo_oracle", f"-Wl,-rpath,{os.path.join(ROOT, 'oracle')}"])
    fin, fout = str(tmp_path / "in.bin"), str(tmp_path / "out.bin")
    n = len(xyz)
    with open(fin, "wb") as f:
        f.write(struct.pack("<2i", rig.width, rig.height))
        f.write(np.asarray(rig.left.P, "<f8").reshape(12).tobytes())
        f.write(np.ascontiguousarray(g["ts_left"], np.uint8).tobytes())
        f.write(struct.pack("<Q", n))
        f.write(np.ascontiguousarray(xyz, "<f4").tobytes())
        f.write(np.asarray(g["T_world_ref"], "<f8").reshape(16).tobytes())
        f.write(np.eye(3).astype("<f8").tobytes())
        f.write(np.zeros(3, "<f8").tobytes())
        f.write(struct.pack("<i", 12))
    subprocess.check_call([exe, fin, fout])
    out = open(fout, "rb").read()
    R = np.frombuffer(out, "<f8", 9, 0).reshape(3, 3)
    t = np.frombuffer(out, "<f8", 3, 72)
    iters = struct.unpack_from("<i", out, 104)[0]
    # the upenn rig's mask is absent in the driver's oracle instance: run the numpy loop on the same (mask-free) camera
    from oracle import oracle as O
    import copy
    rig2 = copy.copy(rig)
    rig2.left = copy.copy(rig.left)
    rig2.left.rect_mask = None
    trk2 = O.OracleTracker(rig2)
    trk2.set_current(g["ts_left"], 5)
    trk2.set_reference(xyz, g["T_world_ref"])
    Rn, tn, itn = _numpy_register(trk2, n, np.eye(3), np.zeros(3), 12)
    assert iters == itn
    assert np.abs(R - Rn).max() < 1e-10 and np.abs(t - tn).max() < 1e-10, (np.abs(R - Rn).max(), np.abs(t - tn).max())
    assert abs(np.linalg.det(R) - 1) < 1e-12 and np.abs(R @ R.T - np.eye(3)).max() < 1e-14
    # ... and it moved: the registered motion is the displacement between the two poses of the fixture, to the tracker's accuracy
    assert np.linalg.norm(t) > 1e-3


@pytest.mark.gpu
def test_gpu_normal_equations_equal_oracle_and_register_equals_python_loop():
    from esvo_amd import closed_loop, lib, params
    g, rig, trk, xyz = _problem()
    p, _ = params.make_params(params.PRESETS["mapping_upenn"], rig)
    dev = lib.Esvo(p, rig, device=0)
    dev.track_set_current(g["ts_left"], 5)
    dev.track_set_reference(xyz, g["T_world_ref"])
    for (off, cnt, huber) in ((100, 300, True), (0, len(xyz), True), (600, 300, False), (0, 1, True), (5000, 10, True)):
        for R, t in ((g["R"], g["t"]), (np.eye(3), np.zeros(3))):
            Hd, bd, cd, nd = dev.track_normal_equations(R, t, off, cnt, huber=huber)
            Ho, bo, co, no = trk.normal_equations(R, t, off, cnt, huber=huber)
            assert nd == no
            assert np.array_equal(Hd, Ho) and np.array_equal(bd, bo) and cd == co, (off, cnt, huber)
    # several poses in one launch: every pose's sums are the ones it gets alone, bit for bit; the limit is an argument error
    rng = np.random.default_rng(5)
    Rs = np.stack([g["R"], np.eye(3)] + [closed_loop.orth(closed_loop.cayley2rot(rng.normal(0, 0.02, 3))) for _ in range(2)])
    ts = np.stack([g["t"], np.zeros(3)] + [rng.normal(0, 0.05, 3) for _ in range(2)])
    for k in (1, 2, 3, 4):
        Hb, bb, cb, nb = dev.track_normal_equations_batch(Rs[:k], ts[:k], 100, 300)
        for q in range(k):
            Hd, bd, cd, nd = dev.track_normal_equations(Rs[q], ts[q], 100, 300)
            assert nb == nd and np.array_equal(Hb[q], Hd) and np.array_equal(bb[q], bd) and cb[q] == cd, (k, q)
    with pytest.raises(lib.EsvoError):
        dev.track_normal_equations_batch(np.tile(np.eye(3), (5, 1, 1)), np.zeros((5, 3)), 0, 10)
    R1, t1, rms1, it1 = dev.track_register(len(xyz), np.eye(3), np.zeros(3))
    R2, t2, rms2 = closed_loop.register_python(dev, len(xyz), np.eye(3), np.zeros(3))
    assert np.abs(R1 - R2).max() < 1e-9 and np.abs(t1 - t2).max() < 1e-9 and abs(rms1 - rms2) < 1e-6 * rms2 + 1e-9
    assert 1 <= it1 <= 12
    dev.close()
