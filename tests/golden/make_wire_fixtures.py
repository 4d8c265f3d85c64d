#!/usr/bin/env python
"""Hand-laid wire fixtures for the ingest row: bytes written field by field from the PUBLIC FORMAT
DEFINITIONS -- ROS 1 message serialisation (wiki.ros.org/msg, wiki.ros.org/roscpp/Overview/MessagesSerializationAndAdaptingTypes),
the message definitions of rpg_dvs_ros (dvs_msgs/EventArray.msg, dvs_msgs/Event.msg) and the bag format 2.0
(wiki.ros.org/Bags/Format/2.0) -- NOT through tests/bagfile.py or esvo_amd.abi.serialize_event_array, which the product's
other tests use.  Every field below names the line of the definition it implements.

    python tests/golden/make_wire_fixtures.py      -> tests/golden/wire_event_array.bin, tests/golden/wire_mini.bag

The events are listed literally in EVENTS_A / EVENTS_B (tests/test_wire_golden.py imports them as the known answer).
"""
import os
import struct

HERE = os.path.dirname(os.path.abspath(__file__))

# (x, y, secs, nsecs, polarity) -- 346 x 260 sensor; stamps sorted; a pixel hit twice; both polarities; polarity byte 0xFF once
EVENTS_A = [
    (0, 0, 1_600_000_000, 100, 1), (345, 259, 1_600_000_000, 2_000, 0), (17, 40, 1_600_000_000, 2_000, 1),
    (200, 131, 1_600_000_000, 999_999_999, 1), (200, 131, 1_600_000_001, 0, 0), (7, 255, 1_600_000_001, 15, 1),
    (300, 3, 1_600_000_001, 1_000_000, 1),
]
EVENTS_B = [
    (12, 12, 1_600_000_001, 2_000_000, 0), (13, 12, 1_600_000_001, 2_000_500, 1), (345, 0, 1_600_000_001, 3_000_000, 1),
    (0, 259, 1_600_000_001, 3_999_999, 0), (100, 100, 1_600_000_001, 4_000_000, 1),
]
POLARITY_BYTE_FF_INDEX = 2   # EVENTS_A[2] is written with the byte 0xFF: any non-zero byte is `true` for a ROS bool

EVENT_ARRAY_DEFINITION = """# This message contains an array of events
# (0, 0) is at top-left corner of image
#

Header header

uint32 height         # image height, that is, number of rows
uint32 width          # image width, that is, number of columns

# an array of events
Event[] events

================================================================================
MSG: std_msgs/Header
# Standard metadata for higher-level stamped data types.
# This is generally used to communicate timestamped data 
# in a particular coordinate frame.
# 
# sequence ID: consecutively increasing ID 
uint32 seq
#Two-integer timestamp that is expressed as:
# * stamp.sec: seconds (stamp_secs) since epoch (in Python the variable is called 'secs')
# * stamp.nsec: nanoseconds since stamp_secs (in Python the variable is called 'nsecs')
# time-handling sugar is provided by the client library
time stamp
#Frame this data is associated with
string frame_id

================================================================================
MSG: dvs_msgs/Event
# A DVS event
uint16 x
uint16 y
time ts
bool polarity
"""


def event_array(seq, stamp, frame_id, height, width, events, ff_index=None):
    b = b""
    b += struct.pack("<I", seq)                       # std_msgs/Header: uint32 seq
    b += struct.pack("<II", stamp[0], stamp[1])       # time stamp: uint32 secs, uint32 nsecs
    b += struct.pack("<I", len(frame_id)) + frame_id  # string frame_id: uint32 length + bytes, no terminator
    b += struct.pack("<I", height)                    # EventArray.msg: uint32 height
    b += struct.pack("<I", width)                     # uint32 width
    b += struct.pack("<I", len(events))               # Event[] events: uint32 element count ...
    for i, (x, y, s, ns, pol) in enumerate(events):   # ... then every element in turn, fields in definition order, unpadded
        b += struct.pack("<H", x)                     # Event.msg: uint16 x
        b += struct.pack("<H", y)                     # uint16 y
        b += struct.pack("<II", s, ns)                # time ts
        b += bytes([0xFF if (ff_index is not None and i == ff_index) else (1 if pol else 0)])  # bool polarity: one byte
    return b


def field(name, value):
    """bag format 2.0, "Headers": <uint32 field_len><name>=<value>, field_len counts name, '=' and value"""
    f = name + b"=" + value
    return struct.pack("<I", len(f)) + f


def record(header_fields, data):
    """bag format 2.0, "Records": <uint32 header_len><header><uint32 data_len><data>"""
    h = b"".join(header_fields)
    return struct.pack("<I", len(h)) + h + struct.pack("<I", len(data)) + data


def mini_bag(msg_a, msg_b):
    topic = b"/davis/left/events"
    conn = struct.pack("<I", 0)
    # 3.5 Connection record (op 0x07): header conn, topic; data = the connection header (topic, type, md5sum, message_definition)
    conn_data = (field(b"topic", topic) + field(b"type", b"dvs_msgs/EventArray") +
                 field(b"md5sum", b"5e8beee5a6c107e504c2e78903c224b8") +      # md5 of dvs_msgs/EventArray as rosbag info prints it
                 field(b"message_definition", EVENT_ARRAY_DEFINITION.encode()))
    conn_rec = record([field(b"conn", conn), field(b"op", b"\x07"), field(b"topic", topic)], conn_data)
    # 3.3 Message data records (op 0x02): header conn, time (the time the message was RECEIVED); data = the serialised message
    t_a, t_b = (1_600_000_001, 1_500_000), (1_600_000_001, 4_500_000)
    msg_rec_a = record([field(b"conn", conn), field(b"op", b"\x02"), field(b"time", struct.pack("<II", *t_a))], msg_a)
    msg_rec_b = record([field(b"conn", conn), field(b"op", b"\x02"), field(b"time", struct.pack("<II", *t_b))], msg_b)
    chunk_data = conn_rec + msg_rec_a + msg_rec_b
    # 3.2 Chunk record (op 0x05): header compression, size (uncompressed); data = connection and message records
    chunk_rec = record([field(b"compression", b"none"), field(b"op", b"\x05"), field(b"size", struct.pack("<I", len(chunk_data)))], chunk_data)
    # 3.4 Index data record (op 0x04), version 1: header ver, conn, count; data = count x (time, uint32 offset into the chunk)
    off_a = len(conn_rec)
    off_b = off_a + len(msg_rec_a)
    index_rec = record([field(b"conn", conn), field(b"count", struct.pack("<I", 2)), field(b"op", b"\x04"), field(b"ver", struct.pack("<I", 1))],
                       struct.pack("<III", t_a[0], t_a[1], off_a) + struct.pack("<III", t_b[0], t_b[1], off_b))
    chunk_pos = 13 + 4096
    # 3.6 Chunk info record (op 0x06), version 1: header ver, chunk_pos, start_time, end_time, count; data = count x (conn, count)
    info_rec = record([field(b"chunk_pos", struct.pack("<Q", chunk_pos)), field(b"count", struct.pack("<I", 1)),
                       field(b"end_time", struct.pack("<II", *t_b)), field(b"op", b"\x06"),
                       field(b"start_time", struct.pack("<II", *t_a)), field(b"ver", struct.pack("<I", 1))],
                      struct.pack("<II", 0, 2))
    index_pos = chunk_pos + len(chunk_rec) + len(index_rec)
    # 3.1 Bag header record (op 0x03): index_pos, conn_count, chunk_count; padded with ASCII spaces to 4096 bytes in total
    hdr_fields = [field(b"chunk_count", struct.pack("<I", 1)), field(b"conn_count", struct.pack("<I", 1)),
                  field(b"index_pos", struct.pack("<Q", index_pos)), field(b"op", b"\x03")]
    hlen = sum(len(f) for f in hdr_fields)
    bag_header = record(hdr_fields, b" " * (4096 - 4 - hlen - 4))
    assert len(bag_header) == 4096
    # 2. "#ROSBAG V2.0\n", the bag header, chunks each followed by its index records, then connection and chunk info records
    return b"#ROSBAG V2.0\n" + bag_header + chunk_rec + index_rec + conn_rec + info_rec


def main():
    a = event_array(41, (1_600_000_001, 1_000_000), b"davis_left", 260, 346, EVENTS_A, ff_index=POLARITY_BYTE_FF_INDEX)
    b = event_array(42, (1_600_000_001, 4_000_000), b"davis_left", 260, 346, EVENTS_B)
    open(os.path.join(HERE, "wire_event_array.bin"), "wb").write(a)
    open(os.path.join(HERE, "wire_mini.bag"), "wb").write(mini_bag(a, b))
    print("wire_event_array.bin", len(a), "bytes; wire_mini.bag", len(mini_bag(a, b)), "bytes")


if __name__ == "__main__":
    main()
