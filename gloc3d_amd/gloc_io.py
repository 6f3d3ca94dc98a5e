"""File formats of the reference's command lines (Python mirror of csrc/host/gloc_io.hpp).

valset  registration/global_localization.cpp:64-122; writers dataset/kitti_i2i.py:82-104
poses   registration/global_localization.cpp:124-156; writer dataset/kitti_i2i.py:108-120
scans   KITTI .bin float32 x y z i (global_localization.cpp:160-182);
        NCLT raw u16 x y z, u8 intensity, label (global_registration.cpp:181-209)
"""
import struct

import numpy as np


def write_valset(path, db_files, q_files, positives):
    with open(path, "w") as f:
        f.write(f"{len(db_files)} {len(q_files)}\n")
        for p in list(db_files) + list(q_files):
            f.write(p + "\n")
        for qi, pos in enumerate(positives):
            f.write(f"{qi}:" + " ".join(str(int(p)) for p in pos) + "\n")


def read_valset(path):
    """Same quirks as the reference: the index before ':' is ignored (positional assignment,
    :104,116) and the positives section ends at EOF or the first empty line (:95-98)."""
    with open(path) as f:
        lines = f.read().split("\n")
    n_db, n_q = (int(t) for t in lines[0].split()[:2])
    db = lines[1:1 + n_db]
    q = lines[1 + n_db:1 + n_db + n_q]
    pos = []
    for line in lines[1 + n_db + n_q:1 + n_db + 2 * n_q]:
        if line == "":
            break
        parts = [p for p in line.split(":") if p != ""]
        pos.append([int(t) for t in parts[1].split()] if len(parts) > 1 else [])
    return db, q, pos


def quat_from_R(R):
    """(qx, qy, qz, qw) of a rotation matrix."""
    t = np.trace(R)
    if t > 0:
        s = np.sqrt(t + 1.0) * 2
        return np.array([(R[2, 1] - R[1, 2]) / s, (R[0, 2] - R[2, 0]) / s, (R[1, 0] - R[0, 1]) / s, 0.25 * s])
    i = int(np.argmax(np.diag(R)))
    j, k = (i + 1) % 3, (i + 2) % 3
    s = np.sqrt(1.0 + R[i, i] - R[j, j] - R[k, k]) * 2
    q = np.zeros(4)
    q[i] = 0.25 * s
    q[j] = (R[j, i] + R[i, j]) / s
    q[k] = (R[k, i] + R[i, k]) / s
    q[3] = (R[k, j] - R[j, k]) / s
    return q


def write_poses(path, poses):
    """One line per pose, db poses first then query poses: qx qy qz qw x y z."""
    with open(path, "w") as f:
        for T in poses:
            q = quat_from_R(np.asarray(T)[:3, :3])
            t = np.asarray(T)[:3, 3]
            f.write(" ".join(f"{v:.9g}" for v in list(q) + list(t)) + "\n")


def read_poses(path):
    out = []
    for line in open(path):
        t = line.split()
        if not t:
            continue
        assert len(t) == 7, "pose lines have 7 tokens (global_localization.cpp:138)"
        qx, qy, qz, qw, x, y, z = (np.float32(v) for v in t)
        R = np.array([[1 - 2 * (qy * qy + qz * qz), 2 * (qx * qy - qz * qw), 2 * (qx * qz + qy * qw)],
                      [2 * (qx * qy + qz * qw), 1 - 2 * (qx * qx + qz * qz), 2 * (qy * qz - qx * qw)],
                      [2 * (qx * qz - qy * qw), 2 * (qy * qz + qx * qw), 1 - 2 * (qx * qx + qy * qy)]], np.float32)
        T = np.eye(4, dtype=np.float32)
        T[:3, :3] = R
        T[:3, 3] = (x, y, z)
        out.append(T)
    return out


def write_descriptors(path, desc):
    """Descriptor file standing in for the TorchScript model argument of global_localization."""
    d = np.ascontiguousarray(desc, np.float32)
    with open(path, "wb") as f:
        f.write(b"GLOCDESC" + struct.pack("<II", d.shape[0], d.shape[1]))
        d.tofile(f)


def read_descriptors(path):
    with open(path, "rb") as f:
        assert f.read(8) == b"GLOCDESC"
        n, dim = struct.unpack("<II", f.read(8))
        return np.fromfile(f, np.float32, n * dim).reshape(n, dim)


# ---- scans ------------------------------------------------------------------------------------

def write_lidar_nclt(path, xyz, intensity=None, label=None):
    """NCLT velodyne_sync records: u16 x, y, z = round((metres + 100) / 0.005), u8 intensity, u8 label."""
    p = np.asarray(xyz, np.float64)[:, :3]
    rec = np.zeros(p.shape[0], dtype=[("x", "<u2"), ("y", "<u2"), ("z", "<u2"), ("i", "u1"), ("l", "u1")])
    q = np.clip(np.rint((p + 100.0) / 0.005), 0, 65535).astype(np.uint16)
    rec["x"], rec["y"], rec["z"] = q[:, 0], q[:, 1], q[:, 2]
    if intensity is not None:
        rec["i"] = np.asarray(intensity, np.uint8)
    if label is not None:
        rec["l"] = np.asarray(label, np.uint8)
    rec.tofile(path)


def read_lidar_nclt(path):
    """float32 [n + 1, 4] (x, y, z, intensity) of an n-record NCLT file -- n + 1: the reference's reader tests
    eof() before it reads (global_registration.cpp:191-192), so the last record is pushed twice; the bytes of a
    truncated trailing record overwrite the leading fields of that extra point.  Empty file: empty cloud."""
    raw = np.fromfile(path, np.uint8)
    if raw.size == 0:
        return np.zeros((0, 4), np.float32)
    n = raw.size // 8
    recs = raw[:8 * n].reshape(n, 8)
    last = recs[-1].copy() if n else np.zeros(8, np.uint8)
    last[:raw.size - 8 * n] = raw[8 * n:]
    recs = np.concatenate([recs, last[None]])
    v = np.ascontiguousarray(recs[:, :6]).view("<u2").astype(np.float32)
    out = np.empty((recs.shape[0], 4), np.float32)
    out[:, :3] = v * np.float32(0.005) + np.float32(-100.0)
    out[:, 3] = recs[:, 6].astype(np.float32)
    return out


def read_lidar_kitti(path):
    raw = np.fromfile(path, np.float32)
    return raw[:raw.size - raw.size % 4].reshape(-1, 4)


def looks_like_kitti(path):
    """Content sniff (never the file size: an NCLT file with an even record count is a multiple of 16 bytes
    too): the first 256 float quadruples are finite, within a kilometre and not denormal -- NCLT records read as
    floats are ~5e8 (x | y << 16) or denormals (z | intensity << 16)."""
    import os
    size = os.path.getsize(path)
    if size == 0 or size % 16:
        return size == 0
    v = np.fromfile(path, np.float32, 4 * min(size // 16, 256)).reshape(-1, 4)[:, :3]
    with np.errstate(invalid="ignore"):
        bad = ~np.isfinite(v) | (np.abs(v) > 1000.0) | ((v != 0) & (np.abs(v) < 1e-20))
    return not bool(bad.any())


def read_lidar_any(path, fmt="auto"):
    """fmt: 'kitti', 'nclt' or 'auto' (by content)."""
    if fmt == "auto":
        fmt = "kitti" if looks_like_kitti(path) else "nclt"
    return read_lidar_kitti(path) if fmt == "kitti" else read_lidar_nclt(path)
