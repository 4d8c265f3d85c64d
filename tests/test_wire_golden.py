"""Ingest against fixtures laid out by hand from the public format definitions
(tests/golden/make_wire_fixtures.py: ROS 1 serialisation of dvs_msgs/EventArray, bag format 2.0) -- independent of the
product's own serialiser / bag writer -- with an ORACLE leg: the CPU oracle decodes the bytes the way
rosbag::MessageInstance::instantiate<dvs_msgs::EventArray>() does (EventMessageEditor.cpp:104-119) and feeds OracleTS; the
device stages the same bytes (esvo_ts_push_event_array / esvo_ts_push_bag) and must render the same Time Surface."""
import os
import struct
import sys

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
GOLDEN = os.path.join(HERE, "golden")
sys.path.insert(0, GOLDEN)
import make_wire_fixtures as W  # noqa: E402


def _expected(events):
    from esvo_amd.abi import EVENT_DTYPE
    ev = np.zeros(len(events), EVENT_DTYPE)
    for i, (x, y, s, ns, pol) in enumerate(events):
        ev[i]["x"], ev[i]["y"], ev[i]["sec"], ev[i]["nsec"], ev[i]["polarity"] = x, y, s, ns, pol
    return ev


def _walk_bag(blob):
    """the records of a format-2.0 bag, straight from the format description (no product code): yields (op, header dict, data)"""
    assert blob[:13] == b"#ROSBAG V2.0\n"
    o = 13

    def rec(buf, o):
        hl = struct.unpack_from("<I", buf, o)[0]
        h = buf[o + 4:o + 4 + hl]
        dl = struct.unpack_from("<I", buf, o + 4 + hl)[0]
        d = buf[o + 8 + hl:o + 8 + hl + dl]
        fields, q = {}, 0
        while q < len(h):
            fl = struct.unpack_from("<I", h, q)[0]
            name, _, val = h[q + 4:q + 4 + fl].partition(b"=")
            fields[name.decode()] = val
            q += 4 + fl
        return fields, d, o + 8 + hl + dl
    while o < len(blob):
        f, d, o = rec(blob, o)
        yield f["op"][0], f, d
        if f["op"][0] == 0x05:
            q = 0
            while q < len(d):
                g, e, q = rec(d, q)
                yield g["op"][0], g, e


def test_fixture_files_are_what_the_generator_lays_out():
    a = open(os.path.join(GOLDEN, "wire_event_array.bin"), "rb").read()
    assert a == W.event_array(41, (1_600_000_001, 1_000_000), b"davis_left", 260, 346, W.EVENTS_A, ff_index=W.POLARITY_BYTE_FF_INDEX)
    # the first bytes, literally: seq 41, stamp 1600000001 s + 1 ms, "davis_left", height 260, width 346, 7 events, event 0
    assert a[:12].hex() == "29000000" + "01105e5f" + "40420f00"
    assert a[12:16] == struct.pack("<I", 10) and a[16:26] == b"davis_left"
    assert a[26:38] == struct.pack("<III", 260, 346, 7)
    assert a[38:51].hex() == "0000" + "0000" + "00105e5f" + "64000000" + "01"
    assert len(a) == 38 + 7 * 13
    bag = open(os.path.join(GOLDEN, "wire_mini.bag"), "rb").read()
    assert bag[:13] == b"#ROSBAG V2.0\n" and struct.unpack_from("<I", bag, 13)[0] + 8 + struct.unpack_from("<I", bag, 13 + 4 + struct.unpack_from("<I", bag, 13)[0])[0] == 4096


def test_oracle_decodes_the_hand_laid_event_array():
    from oracle import oracle as O
    a = open(os.path.join(GOLDEN, "wire_event_array.bin"), "rb").read()
    ev, h, w = O.decode_event_array(a)
    assert (h, w) == (260, 346)
    want = _expected(W.EVENTS_A)
    for f in ("x", "y", "sec", "nsec", "polarity"):
        assert np.array_equal(ev[f], want[f]), f
    for cut in (0, 15, 37, len(a) - 1):          # truncated buffers are refused
        with pytest.raises(ValueError):
            O.decode_event_array(a[:cut])
    with pytest.raises(ValueError):
        O.decode_event_array(a + b"\0")            # trailing bytes: the count does not match the length


def test_bag_reader_on_the_hand_laid_bag():
    """the product's host-side reader (esvo_bag_*, no GPU) and the format walker above find the same two messages; the oracle
    decodes them to the listed events"""
    from esvo_amd import lib
    from oracle import oracle as O
    path = os.path.join(GOLDEN, "wire_mini.bag")
    blob = open(path, "rb").read()
    msgs = [(f, d) for op, f, d in _walk_bag(blob) if op == 0x02]
    conns = [(f, d) for op, f, d in _walk_bag(blob) if op == 0x07]
    assert len(msgs) == 2 and len(conns) == 2 and conns[0][0]["topic"] == b"/davis/left/events"
    got = list(lib.BagReader(path).messages())
    assert [m[0] for m in got] == ["/davis/left/events"] * 2
    assert [m[1] for m in got] == [1_600_000_001 * 10**9 + 1_500_000, 1_600_000_001 * 10**9 + 4_500_000]
    assert [m[2] for m in got] == [d for _, d in msgs]
    for (_, _, payload), events in zip(got, (W.EVENTS_A, W.EVENTS_B)):
        ev, h, w = O.decode_event_array(payload)
        want = _expected(events)
        assert (h, w) == (260, 346) and all(np.array_equal(ev[f], want[f]) for f in ("x", "y", "sec", "nsec", "polarity"))
    assert list(lib.BagReader(path).messages("/davis/right/events")) == []


@pytest.mark.gpu
def test_device_ingest_of_the_hand_laid_bytes_equals_the_oracle(upenn_rig):
    """device: wire bytes / bag -> ring -> SAE -> Time Surface;  oracle: decode -> OracleTS -> Time Surface.  Rendered after
    each message (with and without polarity) and compared bit for bit."""
    from esvo_amd import lib, params
    from oracle import oracle as O
    a = open(os.path.join(GOLDEN, "wire_event_array.bin"), "rb").read()
    path = os.path.join(GOLDEN, "wire_mini.bag")
    rig = upenn_rig
    for ignore_polarity in (1, 0):
        # (no median filter: the fixture's handful of isolated events would not survive a 3x3 median -- every image would be 0)
        p, _ = params.make_params(params.PRESETS["mvstereo_upenn"], rig, ignore_polarity=ignore_polarity, median_blur_kernel_size=0)
        # (1) the EventArray bytes through esvo_ts_push_event_array
        dev = lib.Esvo(p, rig)
        ots = O.OracleTS(rig.width, rig.height)
        assert dev.ts_push_event_array(0, a) == len(W.EVENTS_A)
        ev, _, _ = O.decode_event_array(a)
        ots.push(ev)
        for t in (1_600_000_000 * 10**9 + 3_000, 1_600_000_001 * 10**9 + 10, 1_600_000_001 * 10**9 + 5_000_000):
            g = dev.ts_render(0, t)
            o = ots.render(t, ignore_polarity=bool(ignore_polarity), median_k=0, map_x=rig.left.map_x, map_y=rig.left.map_y)
            assert np.array_equal(g, o)
            assert o.any() or t < 1_600_000_001 * 10**9   # (the three early events fall outside the rectified image)
        dev.close()
        # (2) the bag through esvo_ts_push_bag, message by message (until_ns)
        dev = lib.Esvo(p, rig)
        ots = O.OracleTS(rig.width, rig.height)
        bag = lib.BagReader(path)
        payloads = [m[2] for m in lib.BagReader(path).messages()]
        n1 = dev.ts_push_bag(0, bag, "/davis/left/events", until_ns=1_600_000_001 * 10**9 + 2_000_000)
        assert n1 == len(W.EVENTS_A)
        ots.push(O.decode_event_array(payloads[0])[0])
        t1 = 1_600_000_001 * 10**9 + 1_900_000
        assert np.array_equal(dev.ts_render(0, t1), ots.render(t1, ignore_polarity=bool(ignore_polarity), median_k=0, map_x=rig.left.map_x, map_y=rig.left.map_y))
        n2 = dev.ts_push_bag(0, bag, "/davis/left/events")
        assert n2 == len(W.EVENTS_B)
        ots.push(O.decode_event_array(payloads[1])[0])
        t2 = 1_600_000_001 * 10**9 + 4_200_000
        g, o = dev.ts_render(0, t2), ots.render(t2, ignore_polarity=bool(ignore_polarity), median_k=0, map_x=rig.left.map_x, map_y=rig.left.map_y)
        assert np.array_equal(g, o) and o.any()
        dev.close()
