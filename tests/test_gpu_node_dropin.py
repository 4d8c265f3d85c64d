"""The drop-in, literally: the REFERENCE's node objects (esvo_Mapping.cpp / esvo_MVStereo.cpp compiled unmodified against
stand-in ROS / OpenCV headers, oracle/ref_harness_node.cpp) run twice on the same callbacks' input -- once with their own
MappingAtTime on the host cores, once with include/esvo_hip_mapping_node.hpp (the reference-side binding INTEGRATION.md
describes, compiled against the reference's real class definition) calling libesvo_hip.so in its place.  The node's own
dataTransferring and denoising feed both.  oracle/_ref/libesvo_ref_{node,mvstereo}_hip.so are built in the build container
(the reference tree is absent on the GPU box) and travel with the snapshot; the test skips where they are missing.

  events handed to the matcher   identical (the node's own code on both sides)
  newest frame                   same points; inverse depth to the LM tolerance (the reference's LM is Eigen's driver on
                                 the host, the device runs its canonical restatement: tests/test_ref_pin.py)
  DepthMap after the tick        valid-set IoU >= 0.97, inverse-depth RMSE < 1e-4 (BASELINE.json north_star) with the
                                 regulariser off; with it on, the reference node itself regularises through erased list
                                 elements (undefined behaviour, a handful of inverse depths per tick
                                 depend on the heap), so there the bar is IoU >= 0.97 and >= 97 % of the common cells
                                 within 1e-4
"""
import os

import numpy as np
import pytest

import scenarios as S
from test_ref_pin import check_points, map_stats

pytestmark = pytest.mark.gpu
_REF = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle", "_ref")


def _agreement(a, b, W):
    """fraction of the cells both maps hold whose inverse depths agree within 1e-4"""
    da = dict(zip((a["row"].astype(np.int64) * W + a["col"]).tolist(), a["inv_depth"].tolist()))
    db = dict(zip((b["row"].astype(np.int64) * W + b["col"]).tolist(), b["inv_depth"].tolist()))
    both = [k for k in da if k in db]
    return sum(abs(da[k] - db[k]) < 1e-4 for k in both) / max(len(both), 1)


@pytest.mark.parametrize("regularise", [False, True])
@pytest.mark.parametrize("name,mvstereo", [("dsec", False), ("hkust", False), ("upenn", True), ("rpg", True)])
def test_reference_node_with_the_hip_binding_equals_the_reference_node(name, mvstereo, regularise):
    lib = os.path.join(_REF, ("libesvo_ref_mvstereo" if mvstereo else "libesvo_ref_node") + "_hip.so")
    if not os.path.exists(lib):
        pytest.skip("oracle/_ref/*_hip.so not built (needs the reference tree: build container only)")
    from oracle import ref as R
    import copy
    sc = S.Scenario(name)
    ticks, st = sc.inputs(), sc.stream()
    p = copy.copy(sc.params)
    if regularise and not p.regularization:
        pytest.skip("the preset does not regularise")
    p.regularization = int(regularise)
    cpu = R.RefNode(p, sc.rig, st.pose, mvstereo=mvstereo)             # the reference node as it is
    dev = R.RefNode(p, sc.rig, st.pose, mvstereo=mvstereo, hip=True)
    for node in (cpu, dev):
        node.push_events(st.ev_left)
    for k, tk in enumerate(ticks):
        for node in (cpu, dev):
            node.push_observation(tk["t"], tk["tsL"], tk["tsR"])
            assert node.data_transferring() and node.obs_time() == tk["t"]
        cpu.mapping_at_time()          # esvo_Mapping::MappingAtTime / esvo_MVStereo::MappingAtTime, host
        dev.hip_mapping_at_time()      # the binding, device
        assert np.array_equal(cpu.matched_events(), dev.hip_matched_events())
        fr = dev.hip_newest_frame()
        ref = cpu.newest_frame()
        fr["pose_idx"] = ref["pose_idx"]   # the same virtual views by construction (one table); the field is the window's on the device
        check_points(fr, ref)
        a, b = dev.hip_get_map(), cpu.get_map()
        iou, rmse = map_stats(a, b, sc.rig.width)
        if regularise:
            assert iou >= 0.97 and _agreement(a, b, sc.rig.width) >= 0.97, (k, iou, _agreement(a, b, sc.rig.width))
        else:
            assert iou >= 0.97 and rmse < 1e-4, (k, iou, rmse)


@pytest.mark.parametrize("name", ["upenn", "rpg"])
def test_mvstereo_node_in_block_matching_only_mode_with_the_hip_binding(name):
    """esvo_MVStereo in MVStereoMode 1, PURE_BLOCK_MATCHING (esvo_MVStereo.cpp:383-432): the same node object once with its own
    MappingAtTime (block matching, vEMP2vDP, naive_propagation of the window on the host) and once with the binding, whose
    branch for that mode calls esvo_map_match + esvo_map_fuse_matches_naive.  Matched events identical; maps element by
    element -- row / col / age exact and the ZNCC residual to 1e-12 on all but the handful of cells where that difference
    flips naive_propagation's `residual <` between two near-equal costs (see tests/test_gpu_ref.py), inverse depth to 1e-12."""
    lib = os.path.join(_REF, "libesvo_ref_mvstereo_hip.so")
    if not os.path.exists(lib):
        pytest.skip("oracle/_ref/*_hip.so not built (needs the reference tree: build container only)")
    from oracle import ref as R
    sc = S.Scenario(name)
    ticks, st = sc.inputs(), sc.stream()
    p = sc.params
    cpu = R.RefNode(p, sc.rig, st.pose, mvstereo=True, extra={"MVStereoMode": 1})
    dev = R.RefNode(p, sc.rig, st.pose, mvstereo=True, hip=True, extra={"MVStereoMode": 1})
    for node in (cpu, dev):
        node.push_events(st.ev_left)
    for k, tk in enumerate(ticks):
        for node in (cpu, dev):
            node.push_observation(tk["t"], tk["tsL"], tk["tsR"])
            assert node.data_transferring() and node.obs_time() == tk["t"]
        cpu.mapping_at_time()
        dev.hip_mapping_at_time()
        assert np.array_equal(cpu.matched_events(), dev.hip_matched_events())
        assert len(dev.hip_newest_frame()) == 0          # no refinement in this mode: the frame is the match list
        a, b = dev.hip_get_map(), cpu.get_map()
        assert len(a) == len(b) and len(a) > 0, (k, len(a), len(b))
        same = (a["row"] == b["row"]) & (a["col"] == b["col"]) & (a["age"] == b["age"]) & (np.abs(a["residual"] - b["residual"]) <= 1e-12)
        assert same.mean() >= 0.995, (k, same.mean())
        assert np.allclose(a["inv_depth"][same], b["inv_depth"][same], rtol=1e-12, atol=0)
        assert np.allclose(a["variance"][same], b["variance"][same], rtol=1e-12, atol=0)
