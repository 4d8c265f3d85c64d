"""Minimal writer of ROS bag files (format 2.0, wiki.ros.org/Bags/Format/2.0) for the ingest tests: connections, chunks
(none / bz2 / lz4 frames), message records, index and chunk-info records in the places rosbag::Bag::write puts them."""
import bz2
import ctypes
import struct


def _fields(**kw):
    out = b""
    for k, v in kw.items():
        f = k.encode() + b"=" + v
        out += struct.pack("<I", len(f)) + f
    return out


def _record(header, data):
    return struct.pack("<I", len(header)) + header + struct.pack("<I", len(data)) + data


def lz4_frame(raw, block=65536):
    """an LZ4 frame (v1, independent blocks, no checksums) built with the system's liblz4; a block that does not shrink is
    stored uncompressed, as the format allows"""
    lib = ctypes.CDLL("liblz4.so.1")
    lib.LZ4_compress_default.argtypes = [ctypes.c_char_p, ctypes.c_char_p, ctypes.c_int, ctypes.c_int]
    out = struct.pack("<I", 0x184D2204) + bytes([0x60, 0x40, 0x82])  # FLG: version 01, independent; BD: 64 KB; header checksum byte
    for o in range(0, len(raw), block):
        piece = raw[o:o + block]
        dst = ctypes.create_string_buffer(len(piece) + len(piece) // 255 + 32)
        n = lib.LZ4_compress_default(piece, dst, len(piece), len(dst))
        if 0 < n < len(piece):
            out += struct.pack("<I", n) + dst.raw[:n]
        else:
            out += struct.pack("<I", 0x80000000 | len(piece)) + piece
    return out + struct.pack("<I", 0)


def write_bag(path, messages, types, compression="none", msgs_per_chunk=7):
    """messages: list of (topic, stamp_ns, bytes) in file order; types: topic -> message type"""
    conn_id = {t: i for i, t in enumerate(sorted(types))}

    def conn_record(topic):
        hdr = _fields(op=b"\x07", conn=struct.pack("<I", conn_id[topic]), topic=topic.encode())
        data = _fields(topic=topic.encode(), type=types[topic].encode(), md5sum=b"0" * 32, message_definition=b"# test")
        return _record(hdr, data)

    body = b""
    n_chunks = 0
    for c0 in range(0, len(messages), msgs_per_chunk):
        part = messages[c0:c0 + msgs_per_chunk]
        raw, seen = b"", set()
        for topic, st, payload in part:
            if topic not in seen:  # rosbag writes the connection record into the chunk before its first message there
                raw += conn_record(topic)
                seen.add(topic)
            hdr = _fields(op=b"\x02", conn=struct.pack("<I", conn_id[topic]), time=struct.pack("<II", st // 10**9, st % 10**9))
            raw += _record(hdr, payload)
        comp = {"none": lambda b: b, "bz2": bz2.compress, "lz4": lz4_frame}[compression](raw)
        body += _record(_fields(op=b"\x05", compression=compression.encode(), size=struct.pack("<I", len(raw))), comp)
        for topic in sorted(seen):  # index data records follow their chunk
            cnt = sum(1 for t, _, _ in part if t == topic)
            body += _record(_fields(op=b"\x04", ver=struct.pack("<I", 1), conn=struct.pack("<I", conn_id[topic]), count=struct.pack("<I", cnt)),
                            b"\x00" * (12 * cnt))
        n_chunks += 1
    tail = b"".join(conn_record(t) for t in sorted(types))
    tail += _record(_fields(op=b"\x06", ver=struct.pack("<I", 1), chunk_pos=struct.pack("<Q", 4096 + 13), start_time=b"\0" * 8,
                            end_time=b"\0" * 8, count=struct.pack("<I", 0)), b"")
    hdr = _fields(op=b"\x03", index_pos=struct.pack("<Q", 13 + 4096 + len(body)), conn_count=struct.pack("<I", len(types)),
                  chunk_count=struct.pack("<I", n_chunks))
    pad = 4096 - 8 - len(hdr)
    with open(path, "wb") as f:
        f.write(b"#ROSBAG V2.0\n")
        f.write(_record(hdr, b" " * pad))
        f.write(body)
        f.write(tail)
