"""ROS 1 bag input ("KITTI / bag input formats"): a-loam_amd/rosbag1.py and tools/run_bag.py.

No ROS in this image, so nothing here proves interoperability with the rosbag tools ("unpinned against rosbag"); what is pinned: a
bag assembled byte by byte in this file from the format description (independent of the writer's helpers) reads back, the writer's
output has the structure the format prescribes (4096-byte header record, index entries that point at the message records, chunk
infos, connection records at index_pos), writer -> reader round trips, and PointCloud2 payloads decode BY FIELD NAME like
pcl::fromROSMsg does for the reference (src/scanRegistration.cpp:132-133), whatever else the driver packs into a point."""
import bz2
import importlib
import importlib.util
import os
import struct
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
rb = importlib.import_module("a-loam_amd.rosbag1")


def _field(name, value):
    body = name.encode() + b"=" + value
    return struct.pack("<I", len(body)) + body


def _record(fields, data):
    h = b"".join(_field(k, v) for k, v in fields)
    return struct.pack("<I", len(h)) + h + struct.pack("<I", len(data)) + data


def _hand_built_bag(compression):
    conn_hdr = b"".join(_field(k, v) for k, v in (("topic", b"/velodyne_points"), ("type", b"sensor_msgs/PointCloud2"), ("md5sum", b"1158d486dd51d683ce2f1be655c3c181"),
                                                  ("message_definition", b"Header header\n")))
    inner = _record((("op", b"\x07"), ("conn", struct.pack("<I", 3)), ("topic", b"/velodyne_points")), conn_hdr)
    inner += _record((("op", b"\x02"), ("conn", struct.pack("<I", 3)), ("time", struct.pack("<II", 12, 500))), b"payload-one")
    inner += _record((("conn", struct.pack("<I", 3)), ("time", struct.pack("<II", 13, 0)), ("op", b"\x02")), b"payload-two")    # field order is free
    data = bz2.compress(inner) if compression == "bz2" else inner
    chunk = _record((("op", b"\x05"), ("compression", compression.encode()), ("size", struct.pack("<I", len(inner)))), data)
    index = _record((("op", b"\x04"), ("ver", struct.pack("<I", 1)), ("conn", struct.pack("<I", 3)), ("count", struct.pack("<I", 2))), b"\0" * 24)
    head = _record((("op", b"\x03"), ("index_pos", struct.pack("<Q", 0)), ("conn_count", struct.pack("<I", 1)), ("chunk_count", struct.pack("<I", 1))), b" " * 100)
    return b"#ROSBAG V2.0\n" + head + chunk + index


@pytest.mark.parametrize("compression", ["none", "bz2"])
def test_reader_on_a_bag_assembled_by_hand(tmp_path, compression):
    path = tmp_path / "hand.bag"
    path.write_bytes(_hand_built_bag(compression))
    msgs = list(rb.read_messages(str(path)))
    assert msgs == [("/velodyne_points", "sensor_msgs/PointCloud2", 12_000_000_500, b"payload-one"),
                    ("/velodyne_points", "sensor_msgs/PointCloud2", 13_000_000_000, b"payload-two")]
    assert list(rb.read_messages(str(path), ["/other"])) == []


def test_reader_rejects_what_it_cannot_read(tmp_path):
    p = tmp_path / "x.bag"
    p.write_bytes(b"#ROSBAG V1.2\n")
    with pytest.raises(rb.BagError, match="not a ROS bag"):
        list(rb.read_messages(str(p)))
    good = _hand_built_bag("none")
    p.write_bytes(good[:-40])                                               # cut inside the trailing index record
    with pytest.raises(rb.BagError, match="truncated"):
        list(rb.read_messages(str(p)))
    p.write_bytes(b"#ROSBAG V2.0\n" + struct.pack("<I", 0xfffffff0) + b"garbage")   # a corrupt length word: refused before read() is asked for 4 GiB
    with pytest.raises(rb.BagError, match="runs past the end"):
        list(rb.read_messages(str(p)))
    lz4 = b"#ROSBAG V2.0\n" + _record((("op", b"\x05"), ("compression", b"lz4"), ("size", struct.pack("<I", 10))), b"0123456789")
    p.write_bytes(lz4)
    with pytest.raises(rb.BagError, match="lz4"):
        list(rb.read_messages(str(p)))
    orphan = b"#ROSBAG V2.0\n" + _record((("op", b"\x05"), ("compression", b"none"), ("size", struct.pack("<I", 0))), b"")
    inner = _record((("op", b"\x02"), ("conn", struct.pack("<I", 9)), ("time", struct.pack("<II", 1, 0))), b"x")
    p.write_bytes(b"#ROSBAG V2.0\n" + _record((("op", b"\x05"), ("compression", b"none"), ("size", struct.pack("<I", len(inner)))), inner))
    with pytest.raises(rb.BagError, match="never declared"):
        list(rb.read_messages(str(p)))
    (tmp_path / "e.bag").write_bytes(orphan)                                 # an empty chunk is legal
    assert list(rb.read_messages(str(tmp_path / "e.bag"))) == []


@pytest.mark.parametrize("compression", ["none", "bz2"])
def test_writer_structure_and_round_trip(tmp_path, compression):
    rng = np.random.default_rng(3)
    path = str(tmp_path / "w.bag")
    sent = []
    with rb.BagWriter(path, compression=compression, chunk_bytes=4096) as w:
        for k in range(40):
            topic = "/velodyne_points" if k % 3 else "/other"
            raw = rng.integers(0, 256, int(rng.integers(0, 700)), dtype=np.uint8).tobytes()
            t = 1_600_000_000_000_000_000 + k * 100_000_000 + 7
            w.write(topic, "pkg/Type" + topic[1], "0" * 32, "definition of " + topic, t, raw)
            sent.append((topic, "pkg/Type" + topic[1], t, raw))
    assert list(rb.read_messages(path)) == sent
    assert list(rb.read_messages(path, ["/other"])) == [m for m in sent if m[0] == "/other"]
    blob = open(path, "rb").read()
    assert blob.startswith(rb.MAGIC)
    recs = list(rb._records(blob[len(rb.MAGIC):], "bag"))
    head, hdata, _ = recs[0]
    assert head["op"] == b"\x03" and 4 + sum(4 + len(k) + 1 + len(v) for k, v in head.items()) + 4 + len(hdata) == 4096    # padded record
    index_pos, = struct.unpack("<Q", head["index_pos"])
    n_conn, = struct.unpack("<I", head["conn_count"])
    n_chunk, = struct.unpack("<I", head["chunk_count"])
    chunks = [(h, d, off + len(rb.MAGIC)) for h, d, off in recs if h["op"] == b"\x05"]
    assert n_conn == 2 and n_chunk == len(chunks) > 3
    tail = list(rb._records(blob[index_pos:], "index section"))
    assert [h["op"] for h, _, _ in tail] == [b"\x07"] * n_conn + [b"\x06"] * n_chunk
    assert [struct.unpack("<Q", h["chunk_pos"])[0] for h, _, _ in tail[n_conn:]] == [off for _, _, off in chunks]
    # every index entry of a chunk points at a message record of its connection with the same time
    k = 1
    seen = 0
    for h, d, _ in chunks:
        raw = bz2.decompress(d) if compression == "bz2" else d
        assert struct.unpack("<I", h["size"])[0] == len(raw)
        k = recs.index((h, d, _ - len(rb.MAGIC))) + 1
        while k < len(recs) and recs[k][0]["op"] == b"\x04":
            ih, idata, _o = recs[k]
            conn, = struct.unpack("<I", ih["conn"])
            cnt, = struct.unpack("<I", ih["count"])
            assert len(idata) == 12 * cnt and struct.unpack("<I", ih["ver"])[0] == 1
            for e in range(cnt):
                secs, nsecs, off = struct.unpack_from("<III", idata, 12 * e)
                mh, _md, _ = next(rb._records(raw[off:], "chunk"))
                assert mh["op"] == b"\x02" and struct.unpack("<I", mh["conn"])[0] == conn and struct.unpack("<II", mh["time"]) == (secs, nsecs)
                seen += 1
            k += 1
    assert seen == len(sent)


def _pc2(fields, point_step, width, data, height=1, row_step=None, big=0, frame="velodyne", stamp=(100, 250)):
    s = lambda x: struct.pack("<I", len(x)) + x.encode()
    out = struct.pack("<III", 7, *stamp) + s(frame) + struct.pack("<III", height, width, len(fields))
    for name, off, dt, cnt in fields:
        out += s(name) + struct.pack("<IBI", off, dt, cnt)
    return out + struct.pack("<BIII", big, point_step, row_step if row_step is not None else point_step * width, len(data)) + data + b"\x01"


def test_pointcloud2_is_read_by_field_name():
    rng = np.random.default_rng(1)
    # (a) what pcl::toROSMsg<PointXYZI> writes (kittiHelper, and every cloud the reference's own nodes publish)
    pts = rng.normal(size=(50, 4)).astype(np.float32)
    m = rb.decode_pointcloud2(rb.encode_pointcloud2_xyzi(pts, 12_345_678_900, "/camera_init", 4))
    assert (m["seq"], m["stamp_ns"], m["frame_id"], m["height"], m["width"], m["point_step"], m["is_dense"]) == (4, 12_345_678_900, "/camera_init", 1, 50, 32, True)
    assert m["fields"] == {"x": (0, 7, 1), "y": (4, 7, 1), "z": (8, 7, 1), "intensity": (16, 7, 1)}
    assert np.array_equal(rb.pointcloud2_xyz(m), pts[:, :3])
    body = np.frombuffer(m["data"], np.float32).reshape(50, 8)
    assert np.array_equal(body[:, 4], pts[:, 3]) and np.all(body[:, 3] == 1.0)
    # (b) a velodyne_pointcloud-style point: x y z intensity f32, ring u16, time f32 = 22 bytes, fields in another order
    dt = np.dtype([("x", "<f4"), ("y", "<f4"), ("z", "<f4"), ("intensity", "<f4"), ("ring", "<u2"), ("time", "<f4")])
    assert dt.itemsize == 22
    a = np.zeros(9, dt)
    for f in ("x", "y", "z"):
        a[f] = rng.normal(size=9)
    a["ring"] = np.arange(9)
    raw = _pc2([("ring", 16, 4, 1), ("z", 8, 7, 1), ("x", 0, 7, 1), ("time", 18, 7, 1), ("y", 4, 7, 1), ("intensity", 12, 7, 1)], 22, 9, a.tobytes())
    xyz = rb.pointcloud2_xyz(rb.decode_pointcloud2(raw))
    assert np.array_equal(xyz, np.stack([a["x"], a["y"], a["z"]], 1))
    # (c) an organised cloud with padded rows and FLOAT64 coordinates
    d = rng.normal(size=(2, 3, 3))
    rows = b"".join(d[r].astype("<f8").tobytes() + b"\xee" * 5 for r in range(2))
    raw = _pc2([("x", 0, 8, 1), ("y", 8, 8, 1), ("z", 16, 8, 1)], 24, 3, rows, height=2, row_step=77)
    assert np.array_equal(rb.pointcloud2_xyz(rb.decode_pointcloud2(raw)), d.reshape(6, 3).astype(np.float32))
    # (d) empty cloud, and the refusals
    assert rb.pointcloud2_xyz(rb.decode_pointcloud2(_pc2([("x", 0, 7, 1), ("y", 4, 7, 1), ("z", 8, 7, 1)], 12, 0, b""))).shape == (0, 3)
    with pytest.raises(rb.BagError, match="big-endian"):
        rb.pointcloud2_xyz(rb.decode_pointcloud2(_pc2([("x", 0, 7, 1), ("y", 4, 7, 1), ("z", 8, 7, 1)], 12, 1, b"\0" * 12, big=1)))
    with pytest.raises(rb.BagError, match="without field 'z'"):
        rb.pointcloud2_xyz(rb.decode_pointcloud2(_pc2([("x", 0, 7, 1), ("y", 4, 7, 1)], 8, 1, b"\0" * 8)))
    with pytest.raises(rb.BagError, match="shorter"):
        rb.pointcloud2_xyz(rb.decode_pointcloud2(_pc2([("x", 0, 7, 1), ("y", 4, 7, 1), ("z", 8, 7, 1)], 12, 3, b"\0" * 24)))
    with pytest.raises(rb.BagError, match="not a scalar"):
        rb.pointcloud2_xyz(rb.decode_pointcloud2(_pc2([("x", 0, 5, 1), ("y", 4, 7, 1), ("z", 8, 7, 1)], 12, 1, b"\0" * 12)))
    with pytest.raises(rb.BagError):
        rb.decode_pointcloud2(raw[:40])


def test_kitti_folder_to_bag_like_kitti_helper(tmp_path):
    """`to_bag` of reference src/kittiHelper.cpp (lidar topic): stamps from times.txt through stof, clouds in PointXYZI layout."""
    ds = tmp_path / "ds"
    (ds / "sequences" / "00").mkdir(parents=True)
    (ds / "velodyne" / "sequences" / "00" / "velodyne").mkdir(parents=True)
    rng = np.random.default_rng(9)
    clouds = [rng.normal(size=(n, 4)).astype(np.float32) for n in (17, 5, 30)]
    for k, c in enumerate(clouds):
        c.tofile(ds / "velodyne" / "sequences" / "00" / "velodyne" / f"{k:06d}.bin")
    (ds / "sequences" / "00" / "times.txt").write_text("0.000000e+00\n1.036224e-01\n2.072449e-01\n")
    bag = tmp_path / "k.bag"
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "run_bag.py"), "--from-kitti", str(ds), "--seq", "00", "--write-bag", str(bag), "--out", str(tmp_path / "o")],
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    spec = importlib.util.spec_from_file_location("run_bag", os.path.join(ROOT, "tools", "run_bag.py"))
    tool = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(tool)
    got = list(tool.sweeps(str(bag), "/velodyne_points"))
    assert len(got) == 3
    for (stamp, xyz), c, t in zip(got, clouds, (0.0, 1.036224e-01, 2.072449e-01)):
        assert abs(stamp - float(np.float32(t))) < 1e-9 and np.array_equal(xyz, c[:, :3])
    m = rb.decode_pointcloud2(next(rb.read_messages(str(bag)))[3])
    assert m["frame_id"] == "/camera_init" and m["point_step"] == 32


@pytest.mark.gpu
def test_bag_sweeps_reach_the_gpu_like_arrays_do(tmp_path, syn, O, binding):
    """bag -> tools/run_bag.py reader -> 12-byte records -> C ABI, against the oracle on the same points: features bit for bit, poses
    to 1e-4 (tests/bag_gpu_check.py; run stand-alone on the GPU box as well: it needs neither torch nor pytest)."""
    spec = importlib.util.spec_from_file_location("run_bag", os.path.join(ROOT, "tools", "run_bag.py"))
    tool = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(tool)
    scans, _, _, model = syn.make_sequence("VLP-16", 3, seed=5)
    bag = str(tmp_path / "vlp16.bag")
    tool.write_bag(bag, [s.numpy() for s in scans], [0.1 * k for k in range(len(scans))], compression="bz2")
    spec = importlib.util.spec_from_file_location("bag_gpu_check", os.path.join(ROOT, "tests", "bag_gpu_check.py"))
    chk = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(chk)
    assert chk.check(bag, model.n_scans, model.min_range) == 3
