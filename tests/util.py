"""Shared helpers for the tests: golden loading and input regeneration."""
import importlib.util
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
GOLDEN = os.path.join(HERE, "golden")

_spec = importlib.util.spec_from_file_location("make_goldens", os.path.join(GOLDEN, "make_goldens.py"))
make_goldens = importlib.util.module_from_spec(_spec)
_spec.loader.exec_module(make_goldens)

KNN_CASES = list(make_goldens.KNN_CASES)


def load_knn_case(case):
    """Regenerate the inputs from their seeds and load the committed reference outputs."""
    db, q, k = make_goldens.knn_inputs(case)
    g = np.load(os.path.join(GOLDEN, case + ".npz"))
    # the inputs are regenerated, not stored: make sure they are the ones the goldens were made on
    assert int(q.view(np.uint32).sum(dtype=np.uint64)) == int(g["q_crc"])
    assert int(db.view(np.uint32).sum(dtype=np.uint64)) == int(g["db_crc"])
    return db, q, k, g["idx"], g["d2_bits"]


def load_nn3_case():
    g = np.load(os.path.join(GOLDEN, "nn3_scanpair.npz"))
    return g["src"], g["tgt"], g["idx"], g["d2_bits"]


def load_nn3_fullsize():
    """Full-size scan pair + initial guess (regenerated, crc-checked) and the reference kd-tree's output."""
    src, tgt, T = make_goldens.nn3_fullsize_inputs()
    g = np.load(os.path.join(GOLDEN, "nn3_fullsize.npz"))
    assert int(make_goldens.crc(src)) == int(g["src_crc"]) and int(make_goldens.crc(tgt)) == int(g["tgt_crc"])
    assert (T == g["T"]).all()
    return src, tgt, T, g["idx"], g["d2_bits"]


def bits(a):
    return np.ascontiguousarray(a, np.float32).view(np.uint32)


def random_se3(rng, max_deg=10.0, max_t=2.0):
    from gloc3d_amd import synth
    return synth.se3(rng.uniform(-max_deg, max_deg), (rng.uniform(-max_t, max_t), rng.uniform(-max_t, max_t),
                                                      rng.uniform(-0.2, 0.2)))


# ---- BEV occupancy projection: test clouds and an independent numpy statement ---------------------

def bev_cases():
    """name -> float32 [n, 3|4] clouds covering the edge cases of the projection."""
    from gloc3d_amd import synth
    rng = np.random.default_rng(41)
    out = {}
    w = synth.make_world(7)
    out["lidar"] = synth.lidar_scan(w, synth.se3(0.3, (5, 2, 0)), 7)             # x y z i, ~120k points
    out["gauss_dense"] = (rng.standard_normal((60000, 3)) * (6, 6, 1.5)).astype(np.float32)
    wide = rng.uniform(-130, 130, (50000, 3)).astype(np.float32)                 # most of it beyond 100 m
    wide[:, 2] = rng.uniform(-3, 3, 50000)
    out["wide"] = wide
    # range boundary: exactly 100 m, one ulp either side, and the two summation orders disagreeing
    edge = [(100, 0, 0), (60, 80, 0), (0, 60, 80), (np.nextafter(np.float32(100), np.float32(200)), 0, 0),
            (np.nextafter(np.float32(100), np.float32(0)), 0, 0), (57.735027, 57.735027, 57.735027),
            (57.73503, 57.735027, 57.735023), (99.99999, 0.3, 0.2), (70.71068, 70.71068, 0.01)]
    edge += [(v[0], v[1], v[2] + 0.2) for v in edge] + [(-v[0], -v[1], v[2]) for v in edge]
    near = rng.uniform(99.9995, 100.0005, 4000)[:, None] * _unit(rng, 4000)
    out["range_edge"] = np.concatenate([np.array(edge, np.float32), near.astype(np.float32),
                                        (near * 1.0).astype(np.float32) + np.float32(0.2) * np.array([0, 0, 1], np.float32)])
    # rounding boundary: coordinates at (k + 0.5) * 0.2 and neighbours, both signs
    k = np.arange(-40, 40)
    half = ((k + 0.5) * 0.2).astype(np.float32)
    pts = []
    for d in (-1, 0, 1):
        h = half.view(np.int32) + d
        pts.append(h.view(np.float32))
    h = np.concatenate(pts)
    grid = np.stack([np.tile(h, 3), np.repeat(h[:3 * 80:80].tolist() + [0.1, -0.1, 0.3], h.size // 2)[:h.size * 3],
                     np.resize(np.array([0.0, 0.1, 0.3, -0.1, 0.5], np.float32), h.size * 3)], 1).astype(np.float32)
    out["round_edge"] = grid
    out["one_point"] = np.array([[1.0, 2.0, 0.5]], np.float32)
    out["one_column"] = np.array([[1.0, 2.0, 0.5], [1.01, 2.01, 0.9], [1.0, 2.0, 0.5]], np.float32)
    bad = (rng.standard_normal((500, 3)) * 10).astype(np.float32)
    bad[::7, 0] = np.nan; bad[3::11, 1] = np.inf; bad[5::13, 2] = -np.inf; bad[1::17] = 1e30
    out["non_finite"] = bad
    out["tiny_image"] = (rng.standard_normal((300, 3)) * (0.6, 0.9, 0.5)).astype(np.float32)
    return out


def _unit(rng, n):
    v = rng.standard_normal((n, 3))
    return v / np.linalg.norm(v, axis=1, keepdims=True)


def bev_numpy(points, resolution=0.2, max_range=100.0):
    """The projection stated directly: a pixel is 0 iff its column holds >= 2 distinct hit voxels."""
    p = np.ascontiguousarray(points, np.float32)[:, :3]
    x, y, z = p[:, 0], p[:, 1], p[:, 2]
    with np.errstate(all="ignore"):
        s1 = (x * x + y * y) + z * z
        s2 = x * x + (y * y + z * z)
        keep = ~(np.sqrt(s1) > np.float32(max_range)) & (np.sqrt(s2) <= np.float32(int(max_range)))
    p = p[keep]
    q = (p / np.float32(resolution)).astype(np.float32).astype(np.float64)   # exact widening
    idx = (np.sign(q) * np.floor(np.abs(q) + 0.5)).astype(np.int64)          # lround in fp64 is exact here
    if idx.shape[0] == 0:
        return None, dict(n_returns=0)
    mn, mx = idx.min(0), idx.max(0)
    w, h = int(mx[0] - mn[0] + 1), int(mx[1] - mn[1] + 1)
    vox = np.unique(idx, axis=0)
    cnt = np.zeros((h, w), np.int64)
    np.add.at(cnt, (vox[:, 1] - mn[1], vox[:, 0] - mn[0]), 1)
    img = np.where(cnt >= 2, 0, 255).astype(np.uint8)
    return img, dict(n_returns=int(keep.sum()), min_ix=int(mn[0]), min_iy=int(mn[1]), max_ix=int(mx[0]),
                     max_iy=int(mx[1]), width=w, height=h)


def bev_crop_pad_numpy(img, out_w=768, out_h=768, pad=(255, 0, 0)):
    dst = np.empty((out_h, out_w, 3), np.uint8)
    dst[:] = np.array(pad, np.uint8)
    if img is None:
        return dst
    h, w = img.shape
    cw, ch = min(w, out_w), min(h, out_h)
    sx, sy, dx, dy = (w - cw) // 2, (h - ch) // 2, (out_w - cw) // 2, (out_h - ch) // 2
    dst[dy:dy + ch, dx:dx + cw, :] = img[sy:sy + ch, sx:sx + cw, None]
    return dst


# ---- ground pre-alignment: a tilted scan with a dominant ground plane ------------------------------

def ground_scene(roll_deg=3.0, pitch_deg=-2.0, n_az=400):
    """A synthetic spinning-lidar scan (ground 1.73 m below the sensor) seen from a sensor tilted by
    roll / pitch: returns ([n, 4] float32 x y z i, sensor height)."""
    from gloc3d_amd import synth
    w = synth.make_world(7)
    pts = synth.lidar_scan(w, synth.se3(0.3, (5, 2, 0)), 7, n_az=n_az)
    r, p = np.deg2rad(roll_deg), np.deg2rad(pitch_deg)
    Rx = np.array([[1, 0, 0], [0, np.cos(r), -np.sin(r)], [0, np.sin(r), np.cos(r)]])
    Ry = np.array([[np.cos(p), 0, np.sin(p)], [0, 1, 0], [-np.sin(p), 0, np.cos(p)]])
    out = pts.copy()
    out[:, :3] = (pts[:, :3].astype(np.float64) @ (Ry @ Rx).T).astype(np.float32)
    return out, 1.73
