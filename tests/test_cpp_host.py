"""The C++ host layer (include/esvo_hip.hpp: TimeSurface / EventBM / DepthProblemSolver / DepthFusion with the
reference's method names) driving libesvo_hip.so with MappingAtTime's call sequence, checked against the
golden fixtures."""
import os
import struct
import subprocess

import numpy as np
import pytest

from esvo_amd import calib, params
from esvo_amd.abi import DEPTH_POINT_DTYPE, MATCH_DTYPE

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "tests", "cpp", "replay_golden.cpp")
F64 = ["inv_depth", "scale2", "nu", "variance", "residual", "x", "p_cam"]


def _compile(tmp_path):
    from esvo_amd import lib
    exe = str(tmp_path / "replay_golden")
    libdir = os.path.dirname(lib._LIB_PATH)
    subprocess.check_call(["g++", "-std=c++14", "-O2", "-Wall", "-I", os.path.join(ROOT, "include"), SRC, "-o", exe,
                           "-L", libdir, "-lesvo_hip", f"-Wl,-rpath,{libdir}"])
    return exe


def test_cpp_host_layer_compiles_and_links(tmp_path):
    _compile(tmp_path)


def _dump(path, g, rig, p):
    import ctypes
    with open(path, "wb") as f:
        f.write(struct.pack("<3i", rig.width, rig.height, int(g["n_ticks"])))
        f.write(bytes(memoryview(p).cast("B")) if False else ctypes.string_at(ctypes.addressof(p), ctypes.sizeof(p)))
        for c in (rig.left, rig.right):
            f.write(c.P.astype("<f8").tobytes())
            f.write(c.rect_lut.astype("<f4").tobytes())
            f.write((c.rect_mask if c.rect_mask is not None else np.full((rig.height, rig.width), 255, np.uint8)).tobytes())
            f.write(c.map_x.astype("<f4").tobytes())
            f.write(c.map_y.astype("<f4").tobytes())
        for k in range(int(g["n_ticks"])):
            f.write(struct.pack("<Q", int(g[f"t{k}"])))
            f.write(np.asarray(g[f"T{k}"], "<f8").reshape(16).tobytes())
            st, T = g[f"stamps{k}"], g[f"poses{k}"]
            f.write(struct.pack("<Q", len(st)))
            f.write(st.astype("<u8").tobytes())
            f.write(np.ascontiguousarray(T, "<f8").tobytes())
            ev = g[f"ev{k}"]
            f.write(struct.pack("<Q", len(ev)))
            f.write(ev.tobytes())
            f.write(g[f"tsL{k}"].tobytes())
            f.write(g[f"tsR{k}"].tobytes())


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["upenn_small", "dsec_small"])
def test_cpp_replay_matches_golden(tmp_path, name):
    g = np.load(os.path.join(ROOT, "tests", "golden", name + ".npz"))
    rig = calib.dataset_rig(str(g["rig"]))
    p, _ = params.make_params(params.PRESETS[str(g["preset"])], rig, process_event_num=int(g["n_events"]))
    exe = _compile(tmp_path)
    fin, fout = str(tmp_path / "fixture.bin"), str(tmp_path / "out.bin")
    _dump(fin, g, rig, p)
    subprocess.check_call([exe, fin, fout])
    buf = open(fout, "rb").read()
    off = 0

    def take(dtype):
        nonlocal off
        n = struct.unpack_from("<Q", buf, off)[0]
        off += 8
        a = np.frombuffer(buf, dtype, n, off)
        off += n * dtype.itemsize
        return a

    for k in range(int(g["n_ticks"])):
        mt = take(MATCH_DTYPE)
        pts = take(DEPTH_POINT_DTYPE)
        nf = struct.unpack_from("<Q", buf, off)[0]
        off += 8
        mp = take(DEPTH_POINT_DTYPE)
        assert np.array_equal(mt["event_idx"], g[f"matches{k}"]["event_idx"]) and np.array_equal(mt["cost"], g[f"matches{k}"]["cost"])
        assert len(pts) == len(g[f"points{k}"]) and nf == int(g[f"nf{k}"]) and len(mp) == len(g[f"map{k}"])
        for f in F64:
            assert np.array_equal(pts[f], g[f"points{k}"][f]), f
            assert np.array_equal(mp[f], g[f"map{k}"][f]), f
    # tracker evaluation through esvo_hip::RegProblemLM, against the oracle on the very same inputs
    from oracle import oracle as O
    xyz = take(np.dtype("<f4")).reshape(-1, 3)
    fvec = take(np.dtype("<f8"))
    fjac = take(np.dtype("<f8"))
    assert off == len(buf)
    k = int(g["n_ticks"]) - 1
    trk = O.OracleTracker(rig)
    trk.set_current(g[f"tsL{k}"], 5)
    trk.set_reference(xyz[:2000], np.asarray(g[f"T{k}"], np.float64).reshape(4, 4))
    o = trk.residuals(np.eye(4), 0, 300, huber=True, huber_threshold=50.0)
    assert len(fvec) == len(o) > 0 and np.array_equal(fvec, o) and (o < 255).any()
    oj = trk.jacobian(np.eye(3), np.zeros(3), 0, 300)
    assert np.array_equal(fjac.reshape(6, -1).T, oj)
