"""Host-side pieces of bench.py that run without a GPU: the committed-profile reader behind `roofline.traffic` / `roofline.valu`,
the algorithmic byte counts of the roofline rows, the synthetic workloads' shape."""
import json
import os

import numpy as np

import bench

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_committed_profile_is_the_newest_round_of_the_workload():
    for wl in ("dsec640x480", "upenn346x260"):
        prof = bench.committed_profile(wl)
        assert prof is not None, wl
        meta = json.load(open(os.path.join(ROOT, "profiles", prof["tag"] + "_meta.json")))
        assert meta["workload"] == wl
        assert all(os.path.exists(os.path.join(ROOT, f)) for f in prof["files"])
        lm = [k for k in prof["sq"] if "lm_refine_kernel" in k]
        assert lm and prof["sq"][lm[0]]["SQ_INSTS_VALU"] > 1e6
        # FETCH / WRITE come from separate counter passes: both present for the dominant kernel
        hb = [v for k, v in prof["hbm"].items() if "lm_refine_kernel" in k]
        assert hb and {"FETCH_SIZE", "WRITE_SIZE"} <= set(hb[0])


def test_profile_figures_and_the_whole_tick_rate():
    prof = bench.committed_profile("dsec640x480")
    traffic, valu = bench.profile_figures(prof, "lm_refine", 1.27)
    assert traffic > 1e6 and 0.2 < valu["frac"] < 1.0
    assert valu["peak"] == bench.VALU_PEAK_INST_S
    whole = bench.whole_tick_valu(prof, 1.37)
    # every kernel of a tick: more instructions than the LM kernel alone, and below the chip's issue peak
    assert whole["wave_insts_per_tick"] > valu["wave_insts_per_launch"]
    assert valu["frac"] < whole["frac"] < 1.0
    assert bench.whole_tick_valu(None, 1.0) is None and bench.profile_figures(None, "lm_refine", 1.0) == (None, None)


def test_workloads_have_the_stated_shape():
    rig, stream, p, ticks = bench.make_workload("upenn346x260", 3)
    assert (rig.width, rig.height) == (346, 260) and len(ticks) == 3
    t, stamps, poses, T = ticks[0]
    assert len(stamps) == len(poses) and np.asarray(T).shape == (4, 4)
    assert np.all(np.diff(stream.ns_left.astype(np.int64)) >= 0)
    # the reference-faithful variant cuts the per-tick selection to the yaml's PROCESS_EVENT_NUM
    _, _, p_small, _ = bench.make_workload("upenn346x260", 3, events_cap=1000)
    assert p_small.process_event_num == 1000 < p.process_event_num


def test_looped_stream_of_the_sustained_point_is_sorted_and_continuous():
    from esvo_amd.abi import event_ns
    rig, stream, p, ticks = bench.make_workload("upenn346x260", 3)
    a = stream.slice(0, stream.t0_ns, stream.t0_ns + 20_000_000)
    b = bench.shift_events(a, 3_000_000_123)
    assert np.array_equal(event_ns(b), event_ns(a) + np.uint64(3_000_000_123))
    assert np.array_equal(b["x"], a["x"]) and np.array_equal(b["y"], a["y"]) and np.array_equal(b["polarity"], a["polarity"])
    assert np.all(b["nsec"] < 1_000_000_000)
