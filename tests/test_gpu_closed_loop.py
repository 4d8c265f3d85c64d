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
