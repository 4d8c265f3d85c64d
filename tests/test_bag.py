"""rosbag (format 2.0) ingest, SURVEY.md section 8(f).2: esvo_bag_* walks the chunks of a bag and hands out the serialised
dvs_msgs/EventArray messages (events_repacking_helper/src/EventMessageEditor.cpp:66-119 reads its input this way through
rosbag::View); esvo_ts_push_bag stages them."""
import os

import numpy as np
import pytest

import bagfile
from esvo_amd import abi, params, synth


def _messages(upenn_rig, stream, n_msgs=40):
    msgs = []
    t0 = stream.t0_ns
    for k in range(n_msgs):
        for cam, topic in ((0, "/davis/left/events"), (1, "/davis/right/events")):
            ev = stream.slice(cam, t0 + k * 1_000_000, t0 + (k + 1) * 1_000_000)   # 1 ms messages (EventMessageEditor.cpp:95)
            msgs.append((topic, t0 + (k + 1) * 1_000_000, abi.serialize_event_array(ev, upenn_rig.width, upenn_rig.height, seq=k)))
        if k % 5 == 0:
            msgs.append(("/imu", t0 + k * 1_000_000, b"\x01\x02\x03" * 20))
    return msgs


TYPES = {"/davis/left/events": "dvs_msgs/EventArray", "/davis/right/events": "dvs_msgs/EventArray", "/imu": "sensor_msgs/Imu"}


@pytest.mark.parametrize("compression", ["none", "bz2", "lz4"])
def test_bag_reader_returns_the_event_messages(tmp_path, upenn_rig, upenn_stream, compression):
    from esvo_amd import lib
    msgs = _messages(upenn_rig, upenn_stream)
    path = str(tmp_path / f"events_{compression}.bag")
    bagfile.write_bag(path, msgs, TYPES, compression=compression)
    want = [m for m in msgs if m[0] != "/imu"]
    got = list(lib.BagReader(path).messages())
    assert [(t, s) for t, s, _ in got] == [(t, s) for t, s, _ in want]
    assert all(a[2] == b[2] for a, b in zip(got, want))
    left = list(lib.BagReader(path).messages("/davis/left/events"))
    assert [m[2] for m in left] == [m[2] for m in want if m[0] == "/davis/left/events"]
    assert list(lib.BagReader(path).messages("/no/such/topic")) == []


def test_bag_reader_rejects_what_is_not_a_bag(tmp_path):
    from esvo_amd import lib
    p = tmp_path / "x.bag"
    p.write_bytes(b"#ROSBAG V1.2\n" + b"\0" * 100)
    with pytest.raises(lib.EsvoError, match="format 2.0"):
        lib.BagReader(str(p))
    with pytest.raises(lib.EsvoError, match="cannot open"):
        lib.BagReader(str(tmp_path / "missing.bag"))
    q = tmp_path / "trunc.bag"
    bagfile.write_bag(str(q), [("/e", 5, b"abc")], {"/e": "dvs_msgs/EventArray"})
    q.write_bytes(q.read_bytes()[:4096 + 13 + 30])                  # cut inside the first chunk record
    with pytest.raises(lib.EsvoError, match="truncated"):
        list(lib.BagReader(str(q)).messages())


@pytest.mark.gpu
def test_push_bag_equals_push_events(tmp_path, upenn_rig, upenn_stream):
    from esvo_amd import lib
    msgs = _messages(upenn_rig, upenn_stream, n_msgs=60)
    path = str(tmp_path / "events.bag")
    bagfile.write_bag(path, msgs, TYPES, compression="bz2")
    p, _ = params.make_params(params.PRESETS["mvstereo_upenn"], upenn_rig)
    a, b = lib.Esvo(p, upenn_rig), lib.Esvo(p, upenn_rig)
    t0 = upenn_stream.t0_ns
    t_mid, t_end = t0 + 25_000_000, t0 + 60_000_000
    bags = [lib.BagReader(path), lib.BagReader(path)]
    n1 = [b.ts_push_bag(cam, bags[cam], TYPES_BY_CAM[cam], until_ns=t_mid + 1) for cam in (0, 1)]   # messages stamped <= t_mid
    for cam in (0, 1):
        a.ts_push_events(cam, upenn_stream.slice(cam, t0, t_mid))
        assert n1[cam] == len(upenn_stream.slice(cam, t0, t_mid))
        assert np.array_equal(a.ts_render(cam, t_mid), b.ts_render(cam, t_mid))
    n2 = [b.ts_push_bag(cam, bags[cam], TYPES_BY_CAM[cam]) for cam in (0, 1)]                        # the rest of the bag
    for cam in (0, 1):
        a.ts_push_events(cam, upenn_stream.slice(cam, t_mid, t_end))
        assert n1[cam] + n2[cam] == len(upenn_stream.slice(cam, t0, t_end))
        assert np.array_equal(a.ts_render(cam, t_end), b.ts_render(cam, t_end))


TYPES_BY_CAM = {0: "/davis/left/events", 1: "/davis/right/events"}


@pytest.mark.parametrize("compression", ["none", "bz2", "lz4"])
def test_bag_reader_survives_corruption(tmp_path, compression):
    """Byte flips, truncation and absurd length fields: every outcome is either messages or an error code, never a fault
    (a 6000-mutation run of the same generator found none; this keeps a short version of it in the suite)."""
    import random
    from esvo_amd import lib
    ev = np.zeros(50, abi.EVENT_DTYPE)
    ev["x"], ev["y"], ev["sec"], ev["nsec"] = np.arange(50) % 30, 3, 1, np.arange(50) * 1000
    msgs = []
    for k in range(6):
        msgs.append(("/l", 1000 + k, abi.serialize_event_array(ev, 64, 48, seq=k)))
        msgs.append(("/imu", 1000 + k, b"xyz" * 10))
    path = str(tmp_path / "ok.bag")
    bagfile.write_bag(path, msgs, {"/l": "dvs_msgs/EventArray", "/imu": "sensor_msgs/Imu"}, compression=compression)
    orig = open(path, "rb").read()
    rng = random.Random(20250925)
    outcomes = {"read": 0, "error": 0}
    for _ in range(200):
        b = bytearray(orig)
        mode = rng.randrange(4)
        if mode == 0:
            for _ in range(rng.randrange(1, 6)):
                b[rng.randrange(len(b))] = rng.randrange(256)
        elif mode == 1:
            b = b[: rng.randrange(13, len(b))]
        elif mode == 2:
            i = rng.randrange(13, len(b) - 4)
            b[i:i + 4] = rng.choice([0xFFFFFFFF, 0x7FFFFFFF, 0x80000000, len(b) * 2, 0]).to_bytes(4, "little")
        else:
            i = rng.randrange(13, len(b) - 8)
            b[i:i + 8] = bytes(rng.randrange(256) for _ in range(8))
        p = str(tmp_path / "mut.bag")
        open(p, "wb").write(b)
        try:
            r = lib.BagReader(p)
            for n, _ in enumerate(r.messages()):
                if n > 100:
                    break
            r.close()
            outcomes["read"] += 1
        except lib.EsvoError:
            outcomes["error"] += 1
    assert outcomes["read"] + outcomes["error"] == 200 and outcomes["error"] > 0
