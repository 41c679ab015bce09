"""On-disk formats at the edges of the DSI path, so that the
reference's own Python tools can consume this engine's outputs and its recorded
trajectories can be replayed without ROS:

  write_grid_npy        Grid3D::writeGridNpy           cartesian3dgrid_IO.cpp:30-36   (.npy, shape {Z,Y,X} f32)
  save_depth_points     saveDepthMaps (txt part)        utils.cpp:31-46                ("col row depth" lines)
  read_pose_bag         parse of geometry_msgs/PoseStamped bags   data_loading.cpp:221-302 (ROSBAG v2.0; none / bz2 chunks)
  read_event_bag        parse of dvs_msgs/EventArray bags         data_loading.cpp:31-107, 211-216
  write_pose_bag, write_event_bag   test helpers: minimal writers of the same subset of the format

Host-side file I/O only; nothing here touches voxels.
"""
import bz2
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


def _scan_bag(path, on_message):
    """Walks every message record of a ROSBAG v2.0 file (chunks with compression none or bz2) in
    file order and calls on_message(topic, type, data)."""
    buf = open(path, "rb").read()
    if not buf.startswith(_MAGIC):
        raise ValueError("%s is not a ROSBAG V2.0 file" % path)
    conns = {}

    def handle(header, data):
        op = header["op"][0]
        if op == _OP_CONN:
            conn = struct.unpack("<I", header["conn"])[0]
            info = _read_fields(data)
            conns[conn] = (header["topic"].decode(), info.get("type", b"").decode())
        elif op == _OP_MSG:
            conn = struct.unpack("<I", header["conn"])[0]
            tpc, typ = conns.get(conn, ("", ""))
            return on_message(tpc, typ, data)
        return True

    for header, data in _records(buf, len(_MAGIC)):
        op = header["op"][0]
        if op == _OP_CHUNK:
            comp = header.get("compression", b"none")
            if comp == b"bz2":
                data = bz2.decompress(data)
            elif comp != b"none":
                raise ValueError("chunk compression %r is not supported (none, bz2)" % comp)
            for h2, d2 in _records(data):
                if handle(h2, d2) is False:
                    return
        elif handle(header, data) is False:
            return


def _parse_pose_stamped(data):
    """std_msgs/Header (seq, stamp.sec, stamp.nsec, frame_id) + geometry_msgs/Pose."""
    seq, sec, nsec, n = struct.unpack_from("<IIII", data, 0)
    off = 16 + n
    px, py, pz, qx, qy, qz, qw = struct.unpack_from("<7d", data, off)
    return sec + 1e-9 * nsec, (px, py, pz, qw, qx, qy, qz)


def read_pose_bag(path, topic=None):
    """Reads geometry_msgs/PoseStamped messages of a ROSBAG v2.0 file.
    Returns (times float64[n] by header stamp, poses float64[n][7] = tx,ty,tz,qw,qx,qy,qz),
    sorted by time like the reference's std::map<ros::Time, Transformation>."""
    out = []

    def on_message(tpc, typ, data):
        if (topic is None or tpc == topic) and typ == "geometry_msgs/PoseStamped":
            out.append(_parse_pose_stamped(data))
        return True

    _scan_bag(path, on_message)
    out.sort(key=lambda tp: tp[0])
    times = np.array([t for t, _ in out], np.float64)
    poses = np.array([p for _, p in out], np.float64).reshape(-1, 7)
    return times, poses


def parse_rosbag_gt(path, topic=None, tmin=0.0, tmax=float("inf")):
    """data_loading::parse_rosbag_gt (data_loading.cpp:303-420) for PoseStamped bags: the stamps of the
    returned control poses are RELATIVE to the first pose message of the topic (`initial_timestamp`,
    :339-343 -- the same convention parse_rosbag applies to the events, :262-268, so that events and
    poses share a time axis starting at ~0); poses with relative stamp < tmin are skipped, the first
    one beyond tmax is still taken and ends the scan (:346-353, the flag is tested at the next
    message).  read_pose_bag() above keeps the absolute stamps instead.
    Returns (times float64[n], poses float64[n][7] = tx,ty,tz,qw,qx,qy,qz), in bag order."""
    out = []
    state = {"t0": None, "go": True}

    def on_message(tpc, typ, data):
        if not state["go"]:
            return False
        if (topic is None or tpc == topic) and typ == "geometry_msgs/PoseStamped":
            t, p = _parse_pose_stamped(data)
            if state["t0"] is None:
                state["t0"] = t
            rel = t - state["t0"]
            if rel < tmin:
                return True
            if rel > tmax:
                state["go"] = False
            out.append((rel, p))
        return True

    _scan_bag(path, on_message)
    # the reference keeps them in a std::map<ros::Time, Transformation> filled with insert(): ascending by stamp,
    # and of several poses with the SAME stamp only the first one is kept (insert does not overwrite)
    out.sort(key=lambda tp: tp[0])          # stable: equal stamps stay in bag order
    out = [tp for i, tp in enumerate(out) if i == 0 or tp[0] != out[i - 1][0]]
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


# dvs_msgs/Event as serialised by ROS: uint16 x, uint16 y, time ts (u32 sec, u32 nsec), bool polarity
_EVENT_DTYPE = np.dtype([("x", "<u2"), ("y", "<u2"), ("sec", "<u4"), ("nsec", "<u4"), ("p", "u1")])


def _parse_event_array(data):
    """std_msgs/Header, uint32 height, uint32 width, dvs_msgs/Event[] events."""
    (n,) = struct.unpack_from("<I", data, 12)
    off = 16 + n
    height, width, count = struct.unpack_from("<III", data, off)
    ev = np.frombuffer(data, _EVENT_DTYPE, count, off + 12)
    return height, width, ev


def read_event_bag(path, topic, tmin=0.0, tmax=float("inf"), events_offset=0.0):
    """parse_rosbag's event branch (data_loading.cpp:66-104) + the final sort (:211-216).
    The first event of the first non-empty message defines the initial stamp; an event is kept
    iff its stamp relative to that is >= tmin; the message in which a relative stamp exceeds tmax
    is still taken whole (the reference only stops reading AFTER it); timestamps become
    ts - initial - events_offset (seconds, double); events are sorted by timestamp.
    Returns dict(x u16[n], y u16[n], ts f64[n], polarity u8[n], height, width, initial_stamp)."""
    xs, ys, tss, ps = [], [], [], []
    state = {"t0": None, "h": 0, "w": 0}

    def on_message(tpc, typ, data):
        if tpc != topic or typ != "dvs_msgs/EventArray":
            return True
        h, w, ev = _parse_event_array(data)
        if ev.shape[0] == 0:
            return True
        state["h"], state["w"] = h, w
        if state["t0"] is None:
            state["t0"] = (int(ev["sec"][0]), int(ev["nsec"][0]))
        s0, n0 = state["t0"]
        # (ts - initial).toSec() on ros::Duration: exact integer nanoseconds -> double
        rel_ns = (ev["sec"].astype(np.int64) - s0) * 1000000000 + (ev["nsec"].astype(np.int64) - n0)
        rel = rel_ns.astype(np.float64) * 1e-9
        keep = rel >= tmin
        # ev.ts.toSec() - initial.toSec() - offset, each toSec() = sec + 1e-9 * nsec in double
        t_abs = ev["sec"].astype(np.float64) + 1e-9 * ev["nsec"].astype(np.float64)
        t_new = t_abs - (float(s0) + 1e-9 * float(n0)) - events_offset
        xs.append(ev["x"][keep])
        ys.append(ev["y"][keep])
        tss.append(t_new[keep])
        ps.append(ev["p"][keep])
        return not bool(np.any(rel > tmax))   # stop after this message

    _scan_bag(path, on_message)
    if not xs:
        z = np.empty(0)
        return {"x": z.astype(np.uint16), "y": z.astype(np.uint16), "ts": z.astype(np.float64),
                "polarity": z.astype(np.uint8), "height": 0, "width": 0, "initial_stamp": None}
    x = np.concatenate(xs)
    y = np.concatenate(ys)
    ts = np.concatenate(tss)
    p = np.concatenate(ps)
    order = np.argsort(ts, kind="stable")
    t0 = state["t0"]
    return {"x": x[order], "y": y[order], "ts": ts[order], "polarity": p[order],
            "height": state["h"], "width": state["w"], "initial_stamp": t0[0] + 1e-9 * t0[1]}


def write_event_bag(path, x, y, ts, polarity=None, topic="/dvs/events", height=260, width=346,
                    events_per_message=5000, compression="none"):
    """Minimal ROSBAG v2.0 with dvs_msgs/EventArray messages (one chunk per message, optional
    bz2).  ts: absolute seconds (float64), written as (sec, nsec).  Test helper."""
    x = np.asarray(x, np.uint16)
    y = np.asarray(y, np.uint16)
    ts = np.asarray(ts, np.float64)
    pol = np.ones(x.shape[0], np.uint8) if polarity is None else np.asarray(polarity, np.uint8)
    sec = np.floor(ts).astype(np.int64)
    nsec = np.rint((ts - sec) * 1e9).astype(np.int64)
    carry = nsec >= 1000000000
    sec, nsec = sec + carry, nsec - carry * 1000000000
    conn = struct.pack("<I", 0)
    conn_data = b"".join(_field(k, v) for k, v in (
        ("topic", topic.encode()), ("type", b"dvs_msgs/EventArray"),
        ("md5sum", b"5e8beee5a6c107e504c2e78903c224b8"), ("message_definition", b"")))
    conn_rec = _record((("op", bytes([_OP_CONN])), ("conn", conn), ("topic", topic.encode())), conn_data)
    chunks = []
    n = x.shape[0]
    for seq, a in enumerate(range(0, max(n, 1), events_per_message)):
        b = min(n, a + events_per_message)
        ev = np.empty(b - a, _EVENT_DTYPE)
        ev["x"], ev["y"], ev["sec"], ev["nsec"], ev["p"] = x[a:b], y[a:b], sec[a:b], nsec[a:b], pol[a:b]
        hs, hn = (int(sec[a]), int(nsec[a])) if b > a else (0, 0)
        msg = struct.pack("<IIII", seq, hs, hn, 0) + struct.pack("<III", height, width, b - a) + ev.tobytes()
        rec = _record((("op", bytes([_OP_MSG])), ("conn", conn), ("time", struct.pack("<II", hs, hn))), msg)
        body = (conn_rec if seq == 0 else b"") + rec
        raw_len = len(body)
        if compression == "bz2":
            body = bz2.compress(body)
        chunks.append(_record((("op", bytes([_OP_CHUNK])), ("compression", compression.encode()),
                               ("size", struct.pack("<I", raw_len))), body))
    bag_header = _record((("op", bytes([_OP_BAG_HEADER])), ("index_pos", struct.pack("<Q", 0)),
                          ("conn_count", struct.pack("<I", 1)),
                          ("chunk_count", struct.pack("<I", len(chunks)))), b"")
    pad = 4096 - len(_MAGIC) - len(bag_header)
    if pad > 0:
        bag_header = bag_header[:-4] + struct.pack("<I", pad) + b" " * pad
    with open(path, "wb") as f:
        f.write(_MAGIC + bag_header)
        for c in chunks:
            f.write(c)
