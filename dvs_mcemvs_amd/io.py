"""On-disk formats at the edges of the DSI path, so that the
reference's own Python tools can consume this engine's outputs and its recorded
trajectories can be replayed without ROS:

  write_grid_npy        Grid3D::writeGridNpy           cartesian3dgrid_IO.cpp:30-36   (.npy, shape {Z,Y,X} f32)
  save_depth_points     saveDepthMaps (txt part)        utils.cpp:31-46                ("col row depth" lines)
  read_pose_bag         parse of geometry_msgs/PoseStamped bags   data_loading.cpp:221-302 (ROSBAG v2.0, uncompressed)
  write_pose_bag        test helper: minimal writer of the same subset of the format

Host-side file I/O only; nothing here touches voxels.
"""
import struct

import numpy as np


def write_grid_npy(path, grid):
    """grid: engine Grid3D (downloaded here) or a numpy array [Z][Y][X]; float32 C order, like
    cnpy::npy_save(filename, &data_array_[0], {size_[2], size_[1], size_[0]}, "w")."""
    vol = grid.download() if hasattr(grid, "download") else np.asarray(grid)
    vol = np.ascontiguousarray(vol, np.float32)
    assert vol.ndim == 3
    with open(path, "wb") as f:
        np.save(f, vol, allow_pickle=False)
    return vol.shape


def save_depth_points(path, depth_map, mask):
    """utils.cpp:31-46: one line "c r depth" per pixel with mask > 0, row-major scan; depth is
    streamed with operator<<(float), i.e. up to 6 significant digits (%g)."""
    depth_map = np.asarray(depth_map, np.float32)
    mask = np.asarray(mask)
    rows, cols = np.nonzero(mask > 0)
    with open(path, "w") as f:
        for r, c in zip(rows, cols):
            f.write("%d %d %g\n" % (c, r, depth_map[r, c]))
    return rows.shape[0]


# ------------------------------------------------------------------ ROSBAG v2.0 (subset)
_MAGIC = b"#ROSBAG V2.0\n"
_OP_MSG, _OP_BAG_HEADER, _OP_INDEX, _OP_CHUNK, _OP_CHUNK_INFO, _OP_CONN = 2, 3, 4, 5, 6, 7


def _read_fields(buf):
    fields, i = {}, 0
    while i < len(buf):
        (n,) = struct.unpack_from("<I", buf, i)
        i += 4
        name, _, val = buf[i:i + n].partition(b"=")
        fields[name.decode()] = val
        i += n
    return fields


def _records(buf, start=0, end=None):
    i = start
    end = len(buf) if end is None else end
    while i + 8 <= end:
        (hl,) = struct.unpack_from("<I", buf, i)
        header = _read_fields(buf[i + 4:i + 4 + hl])
        i += 4 + hl
        (dl,) = struct.unpack_from("<I", buf, i)
        yield header, buf[i + 4:i + 4 + dl]
        i += 4 + dl


def _parse_pose_stamped(data):
    """std_msgs/Header (seq, stamp.sec, stamp.nsec, frame_id) + geometry_msgs/Pose."""
    seq, sec, nsec, n = struct.unpack_from("<IIII", data, 0)
    off = 16 + n
    px, py, pz, qx, qy, qz, qw = struct.unpack_from("<7d", data, off)
    return sec + 1e-9 * nsec, (px, py, pz, qw, qx, qy, qz)


def read_pose_bag(path, topic=None):
    """Reads geometry_msgs/PoseStamped messages of an uncompressed ROSBAG v2.0 file.
    Returns (times float64[n] by header stamp, poses float64[n][7] = tx,ty,tz,qw,qx,qy,qz),
    sorted by time like the reference's std::map<ros::Time, Transformation>."""
    buf = open(path, "rb").read()
    if not buf.startswith(_MAGIC):
        raise ValueError("%s is not a ROSBAG V2.0 file" % path)
    conns, out = {}, []

    def handle(header, data):
        op = header["op"][0]
        if op == _OP_CONN:
            conn = struct.unpack("<I", header["conn"])[0]
            info = _read_fields(data)
            conns[conn] = (header["topic"].decode(), info.get("type", b"").decode())
        elif op == _OP_MSG:
            conn = struct.unpack("<I", header["conn"])[0]
            tpc, typ = conns.get(conn, ("", ""))
            if (topic is None or tpc == topic) and typ == "geometry_msgs/PoseStamped":
                out.append(_parse_pose_stamped(data))

    for header, data in _records(buf, len(_MAGIC)):
        op = header["op"][0]
        if op == _OP_CHUNK:
            if header.get("compression", b"none") != b"none":
                raise ValueError("compressed chunks (%r) are not supported" % header["compression"])
            for h2, d2 in _records(data):
                handle(h2, d2)
        else:
            handle(header, data)
    out.sort(key=lambda tp: tp[0])
    times = np.array([t for t, _ in out], np.float64)
    poses = np.array([p for _, p in out], np.float64).reshape(-1, 7)
    return times, poses


def _field(name, val):
    body = name.encode() + b"=" + val
    return struct.pack("<I", len(body)) + body


def _record(fields, data):
    header = b"".join(_field(k, v) for k, v in fields)
    return struct.pack("<I", len(header)) + header + struct.pack("<I", len(data)) + data


def write_pose_bag(path, times, poses, topic="/pose", frame_id="world"):
    """Minimal single-chunk, uncompressed ROSBAG v2.0 with PoseStamped messages (no index
    records: readers that scan chunks, like read_pose_bag, accept it).  Test helper."""
    conn = struct.pack("<I", 0)
    conn_data = b"".join(_field(k, v) for k, v in (
        ("topic", topic.encode()), ("type", b"geometry_msgs/PoseStamped"),
        ("md5sum", b"d3812c3cbc69362b77dc0b19b345f8f5"), ("message_definition", b"")))
    recs = [_record((("op", bytes([_OP_CONN])), ("conn", conn), ("topic", topic.encode())), conn_data)]
    for seq, (t, p) in enumerate(zip(times, poses)):
        sec = int(np.floor(t))
        nsec = int(round((t - sec) * 1e9))
        if nsec >= 1000000000:
            sec, nsec = sec + 1, nsec - 1000000000
        fid = frame_id.encode()
        msg = struct.pack("<IIII", seq, sec, nsec, len(fid)) + fid + struct.pack(
            "<7d", p[0], p[1], p[2], p[4], p[5], p[6], p[3])
        recs.append(_record((("op", bytes([_OP_MSG])), ("conn", conn),
                             ("time", struct.pack("<II", sec, nsec))), msg))
    chunk = b"".join(recs)
    bag_header = _record((("op", bytes([_OP_BAG_HEADER])), ("index_pos", struct.pack("<Q", 0)),
                          ("conn_count", struct.pack("<I", 1)), ("chunk_count", struct.pack("<I", 1))), b"")
    pad = 4096 - len(_MAGIC) - len(bag_header)
    if pad > 0:  # rosbag pads the bag header record to 4096 bytes
        bag_header = bag_header[:-4] + struct.pack("<I", pad) + b" " * pad
    with open(path, "wb") as f:
        f.write(_MAGIC + bag_header)
        f.write(_record((("op", bytes([_OP_CHUNK])), ("compression", b"none"),
                         ("size", struct.pack("<I", len(chunk)))), chunk))
