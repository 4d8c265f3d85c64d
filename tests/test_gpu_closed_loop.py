"""Closed loop on the device: SGM bootstrap (f3) -> tracker evaluation (f1) -> mapper (a*) with the tracked
poses (esvo_core/src/esvo_Mapping.cpp:309-353 InitializationAtTime, esvo_Tracking.cpp:150-260).
Bars are measured on MI355X (see esvo_amd/closed_loop.py for the lag the formulation carries by design)."""
import numpy as np
import pytest

from tests import closed_loop

pytestmark = pytest.mark.gpu


def test_bootstrap_map_tracks_a_15_tick_path():
    r = closed_loop.run(n_ticks=15)                      # fixed reference: the SGM-bootstrapped map
    assert r["sgm_points"] >= 500
    path = r["gt_len"][-1]
    assert path > 0.14
    # lag stays bounded while the path grows: measured 10.5 mm after 150 mm, worst 21 mm on the way
    assert r["pos_err"][-1] < 0.12 * path
    assert max(r["pos_err"]) < 0.2 * path
    assert min(r["cos"][4:]) > 0.97                      # direction of travel, once past the first steps
    assert 0.85 < r["est_len"][-1] / path < 1.05
    assert max(r["rot_err_deg"]) < 1.0
    # the mapper keeps producing depth points from the tracked poses and the map stays on the true surfaces
    assert min(r["points"]) > 150
    assert r["map_on_gt"] > 1000
    assert r["map_median_abs_err"] < 0.03                # inverse depth; measured 0.0155


def test_re_referencing_to_the_fused_map_keeps_tracking():
    r = closed_loop.run(n_ticks=15, reref=5)             # reference = the map fused with TRACKED poses, every 5 ticks
    path = r["gt_len"][-1]
    assert r["pos_err"][-1] < 0.4 * path                 # measured 45 mm of 150 mm: one lag per re-reference
    assert min(r["cos"][4:]) > 0.95
    assert r["est_len"][-1] / path > 0.6
    assert r["map_median_abs_err"] < 0.035


@pytest.mark.parametrize("G,routing", [(2, "y_rect"), (8, "y_rect"), (8, "broadcast")])
def test_closed_loop_on_row_bands_reproduces_one_gpu(G, routing):
    """BASELINE configs[2] on configs[3]'s partition: the mapper split over G row bands (logical ranks on the one GPU, the
    library's collective calls with an in-process all-gather), the tracker consuming the gathered left Time Surface
    (esvo_comm_gather_ts) and the gathered map's cloud (esvo_comm_gather_pointcloud_xyz: band maps merged on the device) at
    every re-reference.  Same cloud, same order, same bits -> the same registered pose at every tick and the same final map as
    the one-GPU loop: not "similar end error", equal."""
    one = closed_loop.run(n_ticks=8, reref=3)
    bands = closed_loop.run_bands(G, routing=routing, n_ticks=8, reref=3)
    assert bands["ranks_agree"]
    assert bands["halo_violations"] == 0
    assert np.array_equal(np.array(bands["poses"]), np.array(one["poses"]))
    assert bands["pos_err"] == one["pos_err"] and bands["points"] == one["points"]
    a, b = bands["map"], one["map"]
    assert len(a) == len(b) > 500
    for f in ("row", "col", "age", "seq", "inv_depth", "scale2", "nu", "variance", "residual", "x", "p_cam"):
        assert np.array_equal(a[f], b[f]), f
    if routing == "y_rect" and G == 8:
        assert bands["events_staged_max_frac"] < 0.6   # band-local ingest
