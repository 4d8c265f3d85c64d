"""The latency path (ABI 8; include/esvo_hip.h "Scheduling"): a tick that runs alone -- handed in after its
predecessor was completed and read, the ROS node's pattern (esvo_Mapping.cpp:261-431 once per MappingLoop turn) -- is enqueued
differently from ticks that overlap (one queue for front stage + LM launch, stage-timing events sampled, matches by index, the frame
compacted by the back stage's first launch, counters written to pinned memory, polled waits).  None of it may change a bit, whatever
the order in which the two kinds of tick follow each other; and the statistics must say what they are: sampled."""
import hashlib
import json
import os
import subprocess
import sys

import numpy as np
import pytest

from esvo_amd import params, rostime

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
F = ["row", "col", "age", "inv_depth", "scale2", "nu", "variance", "residual", "x", "p_cam"]


def _sha(m):
    h = hashlib.sha1()
    for f in F:
        h.update(np.ascontiguousarray(m[f]).tobytes())
    return h.hexdigest()


def _ticks(stream, p, n, t_first=0.06):
    out = []
    for k in range(n):
        t = stream.t0_ns + int((t_first + 0.01 * k) * 1e9)
        stamps, poses = rostime.pose_table(stream.pose, t, p.bm_half_slice_thickness)
        out.append((t, stamps, poses, stream.pose(t)))
    return out


def _mixed_run(rig, stream, events):
    """12 ticks each waited for (reads in between), 5 handed in back to back, 3 waited for again, 4 back to back: SHA-1 of every map read"""
    from esvo_amd import lib
    p, _ = params.make_params(params.PRESETS["mapping_dsec"], rig, process_event_num=events)
    dev = lib.Esvo(p, rig)
    dev.ts_push_events(0, stream.ev_left)
    dev.ts_push_events(1, stream.ev_right)
    shas = []
    ticks = _ticks(stream, p, 24)
    plan = [True] * 12 + [False] * 5 + [True] * 3 + [False] * 4   # True: read the map right after the tick
    for (t, stamps, poses, T), wait in zip(ticks, plan):
        dev.tick_resident(t, T, stamps, poses)
        if wait:
            shas.append(_sha(dev.get_map()))
    shas.append(_sha(dev.get_map()))
    s = dev.stats()
    dev.close()
    return shas, int(s.ticks), int(s.stage_timing_samples)


def _rig_and_stream(duration_s=0.32):
    """the suite's DSEC scene (tests/conftest.py dsec_stream) over enough seconds for 24 ticks"""
    from esvo_amd import calib, synth
    rig = calib.dataset_rig("dsec")
    return rig, synth.make_stream(rig, 20000, duration_s, 0.02, 0.25, seed=20250421, speed=2.0)


_CODE = ("import sys, json; sys.path.insert(0, 'tests'); import test_gpu_latency_mode as T; "
         "rig, stream = T._rig_and_stream(); "
         "print('RESULT', json.dumps(T._mixed_run(rig, stream, int(sys.argv[1]))))")


def _sub(env_extra, events):
    env = dict(os.environ, ESVO_DEV_SWITCHES="1", **env_extra)
    r = subprocess.run([sys.executable, "-c", _CODE, str(events)], cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    return json.loads([l for l in r.stdout.splitlines() if l.startswith("RESULT")][-1][7:])


def test_latency_path_changes_no_bit():
    """the same call sequence -- ticks waited for and ticks that overlap, mixed -- with the latency path off, on (default), with every
    tick recording its stage events, with the path's event bound below the tick size (lone ticks then keep the three queues and only
    sample their events) and with the throughput path's A/B switches on top of that: every map read is the same, SHA-1 for SHA-1"""
    events = 3000
    want = _sub({"ESVO_LOWLAT": "0"}, events)
    assert len(set(want[0])) > 4 and want[1] == 24    # the maps do differ from tick to tick; 24 ticks were mapped
    on = _sub({}, events)
    assert on[0] == want[0] and on[1] == want[1]
    for env in ({"ESVO_LOWLAT_TIMED_EVERY": "1", "ESVO_PIPE_TIMED_EVERY": "1", "ESVO_REG_SPARSE": "0"}, {"ESVO_LOWLAT_MAX_EVENTS": "100"},
                {"ESVO_LOWLAT_MAX_EVENTS": "100", "ESVO_PIPE_BIG_TIMED_EVERY": "4", "ESVO_BACK_PROLOGUE": "1"}):
        got = _sub(env, events)
        assert got[0] == want[0] and got[1] == want[1], env
    # with the path off every tick that mapped something is a sample; with it on they are sampled
    assert want[2] >= 20
    assert 8 <= on[2] < want[2]


def test_stage_timings_are_sampled_and_say_so():
    """esvo_stats_t.stage_timing_samples (ABI 8): ticks that run alone record their stage events on the handle's first 8 and on one in 31
    afterwards; ms_* keep the latest sample; ticks / totals count every tick; the mean is sum / samples"""
    from esvo_amd import lib
    rig, stream = _rig_and_stream(0.22)
    p, _ = params.make_params(params.PRESETS["mapping_dsec"], rig, process_event_num=3000)
    dev = lib.Esvo(p, rig)
    dev.ts_push_events(0, stream.ev_left)
    dev.ts_push_events(1, stream.ev_right)
    seen = []
    for k, (t, stamps, poses, T) in enumerate(_ticks(stream, p, 14)):
        dev.tick_resident(t, T, stamps, poses)
        s = dev.stats()                      # completes the tick: the next one runs alone again
        assert int(s.ticks) == k + 1 and int(s.last_events_in) > 0
        seen.append(int(s.stage_timing_samples))
        assert s.ms_kernel[3] > 0.0 and s.ms_bm > 0.0    # the latest sample stays readable on ticks that recorded nothing
    assert seen[:8] == list(range(1, 9)), seen
    assert seen[8:] == [8] * 6, seen                      # ticks 9..14 of the handle: not sampled (the next sample is its 32nd lone tick)
    mean = s.kernel_ms_mean()
    assert abs(mean[3] - float(s.sum_ms_kernel[3]) / 8) < 1e-9 and mean[3] > 0.0
    dev.close()
