import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
# the library reads its A/B and test switches (ESVO_FUSE_TILE_CAP, ESVO_LM_PAIR, ESVO_TS_STAGE_CAP ...) only when this is set:
# several tests force rare code paths with them, also in child processes
os.environ.setdefault("ESVO_DEV_SWITCHES", "1")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_sessionstart(session):
    """Tests that emulate collectives view library buffers as torch tensors.  torch's lazy CUDA initialisation has been
    seen to report "No HIP GPUs are available" when it first runs after libesvo_hip.so has created and destroyed several
    large contexts in the same process, so it is done once, up front, where a GPU exists."""
    try:
        import torch
        if torch.cuda.is_available():
            torch.cuda.init()
    except Exception:
        pass


@pytest.fixture(scope="session")
def upenn_rig():
    from esvo_amd import calib
    return calib.dataset_rig("upenn")


@pytest.fixture(scope="session")
def upenn_stream(upenn_rig):
    from esvo_amd import synth
    return synth.make_stream(upenn_rig, 6000, 0.2, 0.16, 1.0, seed=20250419)


@pytest.fixture(scope="session")
def dsec_rig():
    from esvo_amd import calib
    return calib.dataset_rig("dsec")


@pytest.fixture(scope="session")
def dsec_stream(dsec_rig):
    from esvo_amd import synth
    # DSEC geometry: f*b = 320 px*m, rho in [0.001, 0.25] -> disparity 0..80
    return synth.make_stream(dsec_rig, 20000, 0.12, 0.02, 0.25, seed=20250421, speed=2.0)


@pytest.fixture(scope="session")
def hd_rig():
    from esvo_amd import calib
    return calib.dataset_rig("hd")  # synthetic 1280x720, f*b = 300 px*m


@pytest.fixture(scope="session")
def hd_stream(hd_rig):
    from esvo_amd import synth
    # rho in [0.03, 0.45] -> disparity 9..135 inside the 6..150 search range (145 candidates)
    return synth.make_stream(hd_rig, 30000, 0.12, 0.03, 0.45, seed=20250423, speed=1.5)
