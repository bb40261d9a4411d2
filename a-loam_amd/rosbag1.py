"""ROS 1 bag files (format "#ROSBAG V2.0") and sensor_msgs/PointCloud2 payloads without ROS: what `rosbag play` feeds the
reference's scanRegistration node (`/velodyne_points`, README.md:36-47 of the reference; src/scanRegistration.cpp:114-133 reads the
message by field name into x, y, z only) and what src/kittiHelper.cpp:74-76,164-176 writes with `to_bag`.

Reader: sequential scan of the records (bag header, chunks with `none` or `bz2` compression, connection and message records inside
them); the index records a bag carries are not needed for a front-to-back read and are skipped, so unindexed bags read too.
Writer: one connection per topic, chunked, fully indexed (index-data records behind every chunk, connection + chunk-info records at
`index_pos`), so that the standard tools can open what it writes.

This image has no ROS installation: the record layout follows the published bag 2.0 specification and ROS 1 message serialisation
(little-endian, strings and arrays length-prefixed with uint32); tests pin hand-built records and the writer -> reader round trip,
not interoperability with the rosbag tools — say "unpinned against rosbag" wherever that matters.  `lz4` chunks use ROS' own lz4
framing and need an lz4 decoder this image does not have: they are reported, not guessed at.
"""
from __future__ import annotations

import bz2
import struct
from typing import Dict, Iterator, List, Optional, Tuple

import numpy as np

MAGIC = b"#ROSBAG V2.0\n"
OP_MSG, OP_BAG_HEADER, OP_INDEX, OP_CHUNK, OP_CHUNK_INFO, OP_CONNECTION = 0x02, 0x03, 0x04, 0x05, 0x06, 0x07
BAG_HEADER_RECORD_BYTES = 4096          # the bag header record is padded to this size so that it can be rewritten in place

POINTCLOUD2_TYPE = "sensor_msgs/PointCloud2"
POINTCLOUD2_MD5 = "1158d486dd51d683ce2f1be655c3c181"
POINTCLOUD2_DEFINITION = """Header header
uint32 height
uint32 width
PointField[] fields
bool    is_bigendian
uint32  point_step
uint32  row_step
uint8[] data
bool is_dense

================================================================================
MSG: std_msgs/Header
uint32 seq
time stamp
string frame_id

================================================================================
MSG: sensor_msgs/PointField
uint8 INT8    = 1
uint8 UINT8   = 2
uint8 INT16   = 3
uint8 UINT16  = 4
uint8 INT32   = 5
uint8 UINT32  = 6
uint8 FLOAT32 = 7
uint8 FLOAT64 = 8

string name
uint32 offset
uint8  datatype
uint32 count
"""
# sensor_msgs/PointField datatypes -> numpy
FIELD_DTYPES = {1: "i1", 2: "u1", 3: "<i2", 4: "<u2", 5: "<i4", 6: "<u4", 7: "<f4", 8: "<f8"}


class BagError(ValueError):
    pass


# ---- record level ------------------------------------------------------------------------------------------------------------
def _pack_header(fields: Dict[str, bytes]) -> bytes:
    out = b"".join(struct.pack("<I", len(k) + 1 + len(v)) + k.encode() + b"=" + v for k, v in fields.items())
    return struct.pack("<I", len(out)) + out


def _parse_header(buf: bytes) -> Dict[str, bytes]:
    fields, p = {}, 0
    while p < len(buf):
        if p + 4 > len(buf):
            raise BagError("truncated record header")
        (n,) = struct.unpack_from("<I", buf, p)
        p += 4
        item = buf[p:p + n]
        if len(item) != n or b"=" not in item:
            raise BagError("malformed header field")
        k, v = item.split(b"=", 1)
        fields[k.decode()] = v
        p += n
    return fields


def _records(buf: bytes, what: str) -> Iterator[Tuple[Dict[str, bytes], bytes, int]]:
    """(header fields, data, offset of the record) for every record in `buf`."""
    p = 0
    while p < len(buf):
        start = p
        if p + 4 > len(buf):
            raise BagError(f"truncated {what}: record length cut off at byte {p}")
        (hl,) = struct.unpack_from("<I", buf, p)
        p += 4
        header = buf[p:p + hl]
        p += hl
        if len(header) != hl or p + 4 > len(buf):
            raise BagError(f"truncated {what}: record header cut off at byte {start}")
        (dl,) = struct.unpack_from("<I", buf, p)
        p += 4
        data = buf[p:p + dl]
        if len(data) != dl:
            raise BagError(f"truncated {what}: record data cut off at byte {start}")
        p += dl
        yield _parse_header(header), data, start


def _file_records(f, what: str) -> Iterator[Tuple[Dict[str, bytes], bytes, int]]:
    """The same over a file object, one record in memory at a time (a KITTI sequence written as a bag is 10 - 20 GB)."""
    import os
    try:
        size = os.fstat(f.fileno()).st_size
    except (AttributeError, OSError, ValueError):       # not a real file (BytesIO ...): fall back to the length check after the read
        size = None

    def need(n, start, part):
        # a corrupt length word must not be handed to read() (a garbage 32-bit length is a 4 GiB allocation): compare it with what is left
        if size is not None and n > size - f.tell():
            raise BagError(f"truncated {what}: record {part} of {n} bytes at byte {start} runs past the end of the file")
        b = f.read(n)
        if len(b) != n:
            raise BagError(f"truncated {what}: record {part} cut off at byte {start}")
        return b
    while True:
        start = f.tell()
        first = f.read(4)
        if not first:
            return
        if len(first) != 4:
            raise BagError(f"truncated {what}: record length cut off at byte {start}")
        header = need(struct.unpack("<I", first)[0], start, "header")
        (dl,) = struct.unpack("<I", need(4, start, "header"))
        yield _parse_header(header), need(dl, start, "data"), start


def _time_ns(v: bytes) -> int:
    secs, nsecs = struct.unpack("<II", v)
    return secs * 1_000_000_000 + nsecs


def read_messages(path: str, topics: Optional[List[str]] = None) -> Iterator[Tuple[str, str, int, bytes]]:
    """Yields (topic, message type, bag time in ns, serialised message) in file order (= recording order).  The file stays open while
    the generator lives: a caller that stops early should close() the generator (or wrap it in contextlib.closing) rather than wait
    for the garbage collector."""
    f = open(path, "rb")
    if f.read(len(MAGIC)) != MAGIC:
        f.close()
        raise BagError("not a ROS bag (format 2.0) file")
    conns: Dict[int, Tuple[str, str]] = {}
    want = set(topics) if topics else None

    def handle(h, data):
        op = h["op"][0]
        if op == OP_CONNECTION:
            ch = _parse_header(data)
            conns[struct.unpack("<I", h["conn"])[0]] = (h["topic"].decode(), ch.get("type", b"").decode())
        elif op == OP_MSG:
            topic, typ = conns.get(struct.unpack("<I", h["conn"])[0], (None, None))
            if topic is None:
                raise BagError("message on a connection that was never declared")
            if want is None or topic in want:
                return topic, typ, _time_ns(h["time"]), data
        return None

    try:
        yield from _walk(f, handle)
    finally:
        f.close()


def _walk(f, handle):
    """Records of the file one at a time: a chunk (and, for bz2, its decompressed form) is the most that is held in memory."""
    for h, data, _ in _file_records(f, "bag"):
        if "op" not in h:
            raise BagError("record without op field")
        op = h["op"][0]
        if op == OP_CHUNK:
            comp = h.get("compression", b"none").decode()
            if comp == "bz2":
                data = bz2.decompress(data)
            elif comp != "none":
                raise BagError(f"chunk compression '{comp}' is not supported here (rewrite the bag with `rosbag decompress`)")
            if "size" in h and struct.unpack("<I", h["size"])[0] != len(data):
                raise BagError("chunk size field does not match its data")
            for ih, idata, _ in _records(data, "chunk"):
                m = handle(ih, idata)
                if m:
                    yield m
        elif op in (OP_MSG, OP_CONNECTION):          # bags without chunks (format allows top-level message records)
            m = handle(h, data)
            if m:
                yield m
        # bag header, index data and chunk info are not needed for a sequential read


# ---- sensor_msgs/PointCloud2 ---------------------------------------------------------------------------------------------------
def _take_string(buf: bytes, p: int) -> Tuple[str, int]:
    (n,) = struct.unpack_from("<I", buf, p)
    if p + 4 + n > len(buf):
        raise BagError("string runs past the end of the message")
    return buf[p + 4:p + 4 + n].decode(errors="replace"), p + 4 + n


def decode_pointcloud2(raw: bytes) -> dict:
    """-> {seq, stamp_ns, frame_id, height, width, fields: {name: (offset, datatype, count)}, is_bigendian, point_step, row_step,
    data (bytes), is_dense}."""
    try:
        seq, secs, nsecs = struct.unpack_from("<III", raw, 0)
        frame_id, p = _take_string(raw, 12)
        height, width, nf = struct.unpack_from("<III", raw, p)
        p += 12
        fields = {}
        for _ in range(nf):
            name, p = _take_string(raw, p)
            off, dt, cnt = struct.unpack_from("<IBI", raw, p)
            p += 9
            fields[name] = (off, dt, cnt)
        big, point_step, row_step, dl = struct.unpack_from("<BIII", raw, p)
        p += 13
        data = raw[p:p + dl]
        if len(data) != dl or p + dl + 1 > len(raw):
            raise BagError("PointCloud2 data runs past the end of the message")
        is_dense = raw[p + dl]
    except struct.error as e:
        raise BagError(f"truncated PointCloud2 message: {e}") from None
    return {"seq": seq, "stamp_ns": secs * 1_000_000_000 + nsecs, "frame_id": frame_id, "height": height, "width": width, "fields": fields,
            "is_bigendian": bool(big), "point_step": point_step, "row_step": row_step, "data": data, "is_dense": bool(is_dense)}


def pointcloud2_xyz(msg: dict) -> np.ndarray:
    """(N, 3) float32 x, y, z read BY FIELD NAME, like pcl::fromROSMsg into pcl::PointXYZ (reference src/scanRegistration.cpp:132-133):
    intensity, ring, time and whatever else the driver adds are ignored; points are taken row by row (row_step) for organised clouds."""
    if msg["is_bigendian"]:
        raise BagError("big-endian PointCloud2 payloads are not supported")
    n, step = msg["height"] * msg["width"], msg["point_step"]
    cols = []
    for name in ("x", "y", "z"):
        if name not in msg["fields"]:
            raise BagError(f"PointCloud2 without field '{name}'")
        off, dt, cnt = msg["fields"][name]
        if dt not in (7, 8) or cnt != 1 or off + (4 if dt == 7 else 8) > step:
            raise BagError(f"field '{name}' is not a scalar FLOAT32 / FLOAT64 inside the point")
        cols.append((off, FIELD_DTYPES[dt]))
    if n == 0:
        return np.zeros((0, 3), np.float32)
    if msg["row_step"] < msg["width"] * step or len(msg["data"]) < (msg["height"] - 1) * msg["row_step"] + msg["width"] * step:
        raise BagError("PointCloud2 data shorter than height x width points")
    rows = [np.frombuffer(msg["data"], np.uint8, msg["width"] * step, r * msg["row_step"]).reshape(msg["width"], step) for r in range(msg["height"])]
    pts = rows[0] if len(rows) == 1 else np.concatenate(rows)
    out = np.empty((n, 3), np.float32)
    for k, (off, dt) in enumerate(cols):
        w = 4 if dt == "<f4" else 8
        out[:, k] = np.ascontiguousarray(pts[:, off:off + w]).view(dt)[:, 0]           # FLOAT64 fields are narrowed like PCL's field mapping does
    return out


def encode_pointcloud2_xyzi(points: np.ndarray, stamp_ns: int, frame_id: str = "/camera_init", seq: int = 0) -> bytes:
    """What pcl::toROSMsg<pcl::PointXYZI> produces (reference src/kittiHelper.cpp:152-156, src/scanRegistration.cpp:413-441): fields
    x@0, y@4, z@8, intensity@16 FLOAT32, point_step 32, height 1, little-endian, is_dense true.  `points`: (N, 4) float32."""
    pts = np.ascontiguousarray(points, np.float32).reshape(-1, 4)
    n = len(pts)
    body = np.zeros((n, 8), np.float32)
    body[:, 0:3] = pts[:, 0:3]
    body[:, 3] = 1.0                                                          # PCL_ADD_POINT4D: the padding float of the xyz block holds 1
    body[:, 4] = pts[:, 3]
    def s(x): return struct.pack("<I", len(x)) + x.encode()
    out = struct.pack("<III", seq, stamp_ns // 1_000_000_000, stamp_ns % 1_000_000_000) + s(frame_id) + struct.pack("<III", 1, n, 4)
    for name, off in (("x", 0), ("y", 4), ("z", 8), ("intensity", 16)):
        out += s(name) + struct.pack("<IBI", off, 7, 1)
    data = body.tobytes()
    return out + struct.pack("<BIII", 0, 32, 32 * n, len(data)) + data + b"\x01"


# ---- writer --------------------------------------------------------------------------------------------------------------------
class BagWriter:
    """with BagWriter(path, compression="none" | "bz2", chunk_bytes=768 KiB) as w: w.write(topic, type, md5, definition, t_ns, raw)"""

    def __init__(self, path: str, compression: str = "none", chunk_bytes: int = 768 * 1024):
        if compression not in ("none", "bz2"):
            raise BagError("compression must be 'none' or 'bz2'")
        self.f = open(path, "wb")
        self.compression, self.chunk_bytes = compression, chunk_bytes
        self.conns: Dict[str, Tuple[int, bytes]] = {}          # topic -> (id, connection record)
        self.chunk = bytearray()
        self.chunk_index: Dict[int, List[Tuple[int, int]]] = {}   # conn -> [(t_ns, offset in the uncompressed chunk)]
        self.chunk_infos: List[Tuple[int, int, int, Dict[int, int]]] = []
        self.f.write(MAGIC)
        self._write_bag_header(0, 0, 0)

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    @staticmethod
    def _time(t_ns: int) -> bytes:
        return struct.pack("<II", t_ns // 1_000_000_000, t_ns % 1_000_000_000)

    def _write_bag_header(self, index_pos: int, conn_count: int, chunk_count: int):
        h = _pack_header({"index_pos": struct.pack("<Q", index_pos), "conn_count": struct.pack("<I", conn_count),
                          "chunk_count": struct.pack("<I", chunk_count), "op": bytes([OP_BAG_HEADER])})
        pad = BAG_HEADER_RECORD_BYTES - len(h) - 4
        self.f.write(h + struct.pack("<I", pad) + b" " * pad)

    def write(self, topic: str, msg_type: str, md5sum: str, definition: str, t_ns: int, raw: bytes):
        if self.f is None:
            raise BagError("bag already closed")
        if topic not in self.conns:
            cid = len(self.conns)
            ch = _pack_header({"topic": topic.encode(), "type": msg_type.encode(), "md5sum": md5sum.encode(), "message_definition": definition.encode()})
            rec = _pack_header({"conn": struct.pack("<I", cid), "topic": topic.encode(), "op": bytes([OP_CONNECTION])}) + struct.pack("<I", len(ch) - 4) + ch[4:]
            self.conns[topic] = (cid, rec)
            self.chunk += rec                                    # a connection record precedes its first message inside the chunk
        cid = self.conns[topic][0]
        self.chunk_index.setdefault(cid, []).append((t_ns, len(self.chunk)))
        self.chunk += _pack_header({"conn": struct.pack("<I", cid), "time": self._time(t_ns), "op": bytes([OP_MSG])}) + struct.pack("<I", len(raw)) + raw
        if len(self.chunk) >= self.chunk_bytes:
            self._flush_chunk()

    def _flush_chunk(self):
        if not self.chunk_index:
            return
        pos = self.f.tell()
        raw = bytes(self.chunk)
        data = bz2.compress(raw) if self.compression == "bz2" else raw
        self.f.write(_pack_header({"compression": self.compression.encode(), "size": struct.pack("<I", len(raw)), "op": bytes([OP_CHUNK])}) + struct.pack("<I", len(data)) + data)
        times = [t for entries in self.chunk_index.values() for t, _ in entries]
        for cid, entries in self.chunk_index.items():
            body = b"".join(self._time(t) + struct.pack("<I", off) for t, off in entries)
            self.f.write(_pack_header({"ver": struct.pack("<I", 1), "conn": struct.pack("<I", cid), "count": struct.pack("<I", len(entries)), "op": bytes([OP_INDEX])}) + struct.pack("<I", len(body)) + body)
        self.chunk_infos.append((pos, min(times), max(times), {cid: len(e) for cid, e in self.chunk_index.items()}))
        self.chunk, self.chunk_index = bytearray(), {}

    def close(self):
        if self.f is None:
            return
        self._flush_chunk()
        index_pos = self.f.tell()
        for _, rec in self.conns.values():
            self.f.write(rec)
        for pos, t0, t1, counts in self.chunk_infos:
            body = b"".join(struct.pack("<II", cid, n) for cid, n in counts.items())
            self.f.write(_pack_header({"ver": struct.pack("<I", 1), "chunk_pos": struct.pack("<Q", pos), "start_time": self._time(t0), "end_time": self._time(t1),
                                       "count": struct.pack("<I", len(counts)), "op": bytes([OP_CHUNK_INFO])}) + struct.pack("<I", len(body)) + body)
        self.f.seek(len(MAGIC))
        self._write_bag_header(index_pos, len(self.conns), len(self.chunk_infos))
        self.f.close()
        self.f = None
