"""Randomised configurations against the canonical oracle, bit for bit (tools/fuzz_parity.py holds the generator: random rig,
patch size, norm, Student-t constants, iteration cap, fusion radius / strategy / window, regulariser, thresholds, denoising,
Time-Surface options, block-matching window / step / direction, thread-stride count, node type, tick size, call path, ring
size -- none of them a shipped yaml).  1100 seeds ran equal on an MI355X in round 6 (profiles/r06_fuzz_parity.txt), and 215
more split over 2-8 ranks against the one-GPU run (tools/fuzz_dist.py); the suite keeps a dozen + seven that cover every
option with non-trivial maps."""
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))

# seed: what it exercises (from the generator's one-line summary)
SEEDS = {
    5021: "upenn 25x25 l2, 12000 events, regulariser r=12, a ring that wraps",
    5025: "dsec 9x3, fusion radius 2, regulariser r=3, queue length 20, step 3",
    5035: "hkust 10x4, queue length 3, mvstereo node, lazy path",
    5042: "upenn Denoising, 12000 events, CONST_POINTS, wrapping ring",
    5064: "rpg 5x5, up-down search, regulariser r=12, smoothed, lazy path",
    5080: "hkust l2 + Denoising + smoothing, fusion radius 1, lazy path, wrapping ring",
    5082: "dsec 5x5 l2, regulariser r=20, mvstereo, lazy path",
    5088: "hkust l2, fusion radius 2, regulariser r=1, wrapping ring",
    5116: "dsec 10x4 up-down, queue 20, regulariser r=5, mvstereo",
    5132: "dsec 15x7, 12000 events, fusion radius 1, CONST_POINTS",
    5156: "upenn 25x25 l2 + Denoising, CONST_POINTS, regulariser r=3",
    5171: "hkust 50-event ticks, fusion radius 2, queue 20, lazy path, wrapping ring",
}


@pytest.mark.gpu
@pytest.mark.parametrize("seed", sorted(SEEDS))
def test_random_configuration_equals_the_oracle(seed):
    import fuzz_parity
    bad, brief = fuzz_parity.run_case(seed)
    assert not bad, (bad[:4], brief)
    sizes = [int(x) for x in brief.split("maps [")[1].rstrip("]").split(",")]
    assert max(sizes) > 100, brief   # the case still maps something (the generator has not drifted)


# the same generator, the run split over several ranks on the one GPU (tools/fuzz_dist.py): seed -> what it exercises
DIST_SEEDS = {
    9145: "band mode, 8 ranks, y_rect routing: upenn 25x25, regulariser r=20, smoothed",
    9177: "band mode, 2 ranks, y_rect: hkust 5x5 with Denoising routed by rows, fusion radius 2, 12000 events",
    9186: "tick-interleaved, 4 ranks, resident renders, a read-out inside a round: hkust Denoising, regulariser r=5",
    9228: "tick-interleaved, 8 ranks, four calls: hkust 25x25, fusion radius 1",
    9296: "band mode, 8 ranks, broadcast: dsec 5x5 l2 up-down, queue length 3, fusion radius 2",
    9456: "tick-interleaved, 8 ranks, resident: dsec, queue length 3, smoothed, regulariser",
    9457: "band mode, 3 ranks (ragged bands), y_rect: rpg, CONST_POINTS, 12000 events",
}


@pytest.mark.gpu
@pytest.mark.parametrize("seed", sorted(DIST_SEEDS))
def test_random_configuration_split_over_ranks_equals_one_gpu(seed):
    import fuzz_dist
    bad, brief = fuzz_dist.run_case(seed)
    assert bad is not None, brief     # (None = the configuration maps nothing: the generator has drifted)
    assert not bad, (bad[:4], brief)
    assert ("band world" if DIST_SEEDS[seed].startswith("band") else "tick world") in brief   # the mode recorded above


# Time-Surface ingest under out-of-order delivery (tools/fuzz_ts.py): seed -> what it exercises
TS_SEEDS = {
    3014: "upenn, queues of 20, 8 % late events, 12 renders",
    3019: "dsec, one stamp per pixel, 40 % late events, 3x3 median, 16 renders",
    3067: "upenn, queues of 3, FORWARD raster",
    3081: "dsec, queues of 20, 40 % late events, 45 renders",
    3110: "identity remap 346x180, FORWARD raster, 35 % late events",
    3131: "dsec, queues of 20, FORWARD raster, 3x3 median, 45 renders",
}


@pytest.mark.gpu
@pytest.mark.parametrize("seed", sorted(TS_SEEDS))
def test_random_out_of_order_ingest_equals_the_oracle_raster(seed):
    import fuzz_ts
    bad, brief = fuzz_ts.run_case(seed)
    assert not bad, (bad[:4], brief)
    assert " 0 late" not in brief and " 0 renders" not in brief, brief


# the tracker's evaluation side on the device (tools/fuzz_track.py): images, residuals, Jacobian, normal equations, batch, register
TRACK_SEEDS = [7000, 7003, 7019, 7020, 7021]


@pytest.mark.gpu
@pytest.mark.parametrize("seed", TRACK_SEEDS)
def test_random_tracker_evaluations_equal_the_oracle(seed):
    import fuzz_track
    bad, brief = fuzz_track.run_case(seed)
    assert not bad, (bad[:4], brief)


# the SGM bootstrap (tools/fuzz_sgm.py): disparity image, point count, frame, map
SGM_SEEDS = {11012: "dsec Time-Surface pair, 5001 events", 11048: "upenn noise vs its perturbed shift", 11110: "dsec saturated images",
             11124: "hkust Time-Surface pair, threshold 500", 11213: "dsec noise vs its shift, 4701 points"}


@pytest.mark.gpu
@pytest.mark.parametrize("seed", sorted(SGM_SEEDS))
def test_random_sgm_bootstrap_equals_the_oracle(seed):
    import fuzz_sgm
    bad, brief = fuzz_sgm.run_case(seed)
    assert not bad, (bad[:4], brief)
    assert ": 0 points" not in brief, brief


# random CALL SEQUENCES on one handle (tools/fuzz_api.py): tick paths mixed, ingest calls mixed, reads in between, esvo_set_params and
# esvo_reset mid-run -- every read against the oracle
API_SEEDS = [50000, 50007, 50092, 50163, 50225]


@pytest.mark.gpu
@pytest.mark.parametrize("seed", API_SEEDS)
def test_random_call_sequence_equals_the_oracle(seed):
    import fuzz_api
    bad, brief = fuzz_api.run_case(seed)
    assert not bad, (bad[:4], brief)
    assert "reset@" in brief and "params@" in brief, brief
