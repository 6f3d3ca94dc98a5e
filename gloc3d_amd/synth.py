"""Deterministic synthetic inputs: place descriptors and lidar scans.

Everything is derived from a random-access counter RNG (splitmix64 finaliser) whose "gaussian" is an
Irwin-Hall sum of the draw's four 16-bit fields.  Only integer arithmetic and single IEEE fp32
operations are involved, so numpy (here), plain C (the test oracle) and the HIP generator
(csrc/synth.hip, used for databases too large to upload) produce the same bits on every machine.
"""
import numpy as np

_G = np.uint64(0x9E3779B97F4A7C15)
_M1 = np.uint64(0xBF58476D1CE4E5B9)
_M2 = np.uint64(0x94D049BB133111EB)
_KS = np.uint64(0xD1B54A32D192ED03)
_GAUSS_SCALE = np.float32(1.0) / np.float32(37837.227)

# stream tags (xor-ed into the seed) so anchors / noise / queries never share a stream
TAG_NOISE = 0x6E6F697365
TAG_QUERY = 0x7175657279


def _u64(x):
    return np.asarray(x, dtype=np.uint64)


def mix64(z):
    with np.errstate(over="ignore"):
        z = _u64(z)
        z = (z ^ (z >> np.uint64(30))) * _M1
        z = (z ^ (z >> np.uint64(27))) * _M2
        return z ^ (z >> np.uint64(31))


def rng_key(seed, stream):
    with np.errstate(over="ignore"):
        return mix64(mix64(_u64(seed) + _G) ^ (_u64(stream) * _KS + np.uint64(1)))


def rng_draw(key, ctr):
    with np.errstate(over="ignore"):
        return mix64(_u64(key) + (_u64(ctr) + np.uint64(1)) * _G)


def rng_gauss(key, ctr):
    u = rng_draw(key, ctr)
    m = np.uint64(0xFFFF)
    s = (u & m) + ((u >> np.uint64(16)) & m) + ((u >> np.uint64(32)) & m) + ((u >> np.uint64(48)) & m)
    return (s.astype(np.int64) - 131070).astype(np.float32) * _GAUSS_SCALE


def rng_uniform(key, ctr):
    """Uniform in [0,1) with 24 bits (exact in fp32)."""
    return (rng_draw(key, ctr) >> np.uint64(40)).astype(np.float32) * np.float32(1.0 / (1 << 24))


def _gauss_rows(seed, rows, dim):
    keys = rng_key(seed, _u64(rows))[:, None]
    cols = np.arange(dim, dtype=np.uint64)[None, :]
    return rng_gauss(keys, cols)


def _scale(dim):
    return np.float32(1.0 / np.sqrt(np.float64(dim)))


def descriptors_iid(seed, first_row, n_rows, dim, chunk=2048):
    """Rows of N(0,1)/sqrt(dim): norms ~1, nearest neighbours far and closely spaced (hard case)."""
    out = np.empty((n_rows, dim), np.float32)
    for a in range(0, n_rows, chunk):
        b = min(n_rows, a + chunk)
        rows = np.arange(first_row + a, first_row + b, dtype=np.uint64)
        out[a:b] = _gauss_rows(seed, rows, dim) * _scale(dim)
    return out


def descriptors_traj(seed, first_row, n_rows, dim, stride=16, noise=0.05, chunk=2048):
    """'Trajectory' database: row i interpolates between random anchors every `stride` rows plus
    small per-row noise, so consecutive places are near neighbours (realistic retrieval)."""
    out = np.empty((n_rows, dim), np.float32)
    sc = _scale(dim)
    nz = np.float32(noise)
    for a in range(0, n_rows, chunk):
        b = min(n_rows, a + chunk)
        rows = np.arange(first_row + a, first_row + b, dtype=np.uint64)
        k = rows // np.uint64(stride)
        t = ((rows % np.uint64(stride)).astype(np.float32) / np.float32(stride))[:, None]
        a0 = _gauss_rows(seed, k, dim) * sc
        a1 = _gauss_rows(seed, k + np.uint64(1), dim) * sc
        ns = _gauss_rows(seed ^ TAG_NOISE, rows, dim) * sc
        w0 = np.float32(1.0) - t
        out[a:b] = (w0 * a0 + t * a1) + nz * ns
    return out


def queries_near(seed, db_rows, dim, qnoise=0.05, gen=None, **kw):
    """Queries = chosen database rows + qnoise * gaussian (db rows regenerated, not read back)."""
    db_rows = np.asarray(db_rows, dtype=np.uint64)
    gen = gen or descriptors_traj
    base = np.concatenate([gen(seed, int(r), 1, dim, **kw) for r in db_rows], axis=0)
    ns = _gauss_rows(seed ^ TAG_QUERY, np.arange(len(db_rows), dtype=np.uint64), dim) * _scale(dim)
    return base + np.float32(qnoise) * ns


# ---- lidar scans ---------------------------------------------------------------------------

def make_world(seed, n_boxes=70, extent=90.0):
    """Procedural scene: ground plane z = -1.73 (sensor height, cf. the reference's
    global_registration.cpp:1226 comment) plus axis-aligned boxes (buildings, vehicles, poles)."""
    key = rng_key(seed, 0)
    u = lambda c: rng_uniform(key, np.arange(c, c + n_boxes, dtype=np.uint64)).astype(np.float64)
    cx = (u(0) * 2 - 1) * extent
    cy = (u(1000) * 2 - 1) * extent
    kind = u(2000)
    sx = np.where(kind < 0.5, 6 + 14 * u(3000), np.where(kind < 0.85, 1.8 + 2.6 * u(3000), 0.3))
    sy = np.where(kind < 0.5, 6 + 14 * u(4000), np.where(kind < 0.85, 1.6 + 0.4 * u(4000), 0.3))
    h = np.where(kind < 0.5, 4 + 10 * u(5000), np.where(kind < 0.85, 1.4 + 0.5 * u(5000), 5.0))
    # keep a clear corridor around the origin so the sensor is never inside a box
    far = np.hypot(cx, cy) > 8.0 + 0.75 * np.maximum(sx, sy)
    lo = np.stack([cx - sx / 2, cy - sy / 2, np.full(n_boxes, -1.73)], 1)[far]
    hi = np.stack([cx + sx / 2, cy + sy / 2, -1.73 + h], 1)[far]
    return dict(lo=lo, hi=hi, ground=-1.73)


# ---- SURVEY 8d cfg D: a loop trajectory through one procedural world --------------------------------------

def loop_trajectory(n_poses=4541, length_m=3724.0, samples=400000):
    """A closed, gently waving loop about the origin of the given arc length (KITTI odometry 00: 4541 poses over
    3724 m), sampled at n_poses EQUAL arc-length steps.  Returns (poses [n, 4, 4] world <- sensor with the heading
    along the tangent, z = 0, no roll / pitch; xy [n, 2])."""
    a = np.linspace(0.0, 2.0 * np.pi, samples + 1)
    shape = 1.0 + 0.10 * np.sin(3.0 * a) + 0.04 * np.sin(7.0 * a + 1.0)
    x, y = shape * np.cos(a), shape * np.sin(a)
    seg = np.hypot(np.diff(x), np.diff(y))
    scale = length_m / seg.sum()
    x, y = x * scale, y * scale
    cum = np.concatenate([[0.0], np.cumsum(seg * scale)])
    s = np.arange(n_poses) * (length_m / n_poses)
    px, py = np.interp(s, cum, x), np.interp(s, cum, y)
    ds = 0.05
    hx = np.interp((s + ds) % length_m, cum, x) - np.interp((s - ds) % length_m, cum, x)
    hy = np.interp((s + ds) % length_m, cum, y) - np.interp((s - ds) % length_m, cum, y)
    yaw = np.arctan2(hy, hx)
    T = np.tile(np.eye(4), (n_poses, 1, 1))
    T[:, 0, 0], T[:, 0, 1], T[:, 1, 0], T[:, 1, 1] = np.cos(yaw), -np.sin(yaw), np.sin(yaw), np.cos(yaw)
    T[:, 0, 3], T[:, 1, 3] = px, py
    return T, np.stack([px, py], 1)


def make_road_world(seed, road_xy, density=70.0 / (180.0 * 180.0), margin=100.0, corridor=2.5):
    """make_world's scene (ground plane + buildings / vehicles / poles as axis-aligned boxes, the same size classes and
    the same density: 70 per 180 m x 180 m) over the bounding square of a trajectory, with a ROAD: every box whose footprint comes
    within `corridor` metres of a trajectory point is dropped, so the sensor is never inside or against a box."""
    road_xy = np.asarray(road_xy, np.float64)
    lo_xy, hi_xy = road_xy.min(0) - margin, road_xy.max(0) + margin
    size = hi_xy - lo_xy
    n_boxes = int(round(density * size[0] * size[1]))
    key = rng_key(seed, 0)
    u = lambda c: rng_uniform(key, (np.uint64(c) << np.uint64(32)) + np.arange(n_boxes, dtype=np.uint64)).astype(np.float64)
    cx = lo_xy[0] + u(0) * size[0]
    cy = lo_xy[1] + u(1) * size[1]
    kind = u(2)
    sx = np.where(kind < 0.5, 6 + 14 * u(3), np.where(kind < 0.85, 1.8 + 2.6 * u(3), 0.3))
    sy = np.where(kind < 0.5, 6 + 14 * u(4), np.where(kind < 0.85, 1.6 + 0.4 * u(4), 0.3))
    h = np.where(kind < 0.5, 4 + 10 * u(5), np.where(kind < 0.85, 1.4 + 0.5 * u(5), 5.0))
    lo = np.stack([cx - sx / 2, cy - sy / 2, np.full(n_boxes, -1.73)], 1)
    hi = np.stack([cx + sx / 2, cy + sy / 2, -1.73 + h], 1)
    keep = np.ones(n_boxes, bool)
    for a in range(0, len(road_xy), 512):              # distance of every road point to every footprint, a block at a time
        p = road_xy[a:a + 512]
        gap = np.maximum(np.maximum(lo[None, :, :2] - p[:, None, :], p[:, None, :] - hi[None, :, :2]), 0.0)
        keep &= ~(np.hypot(gap[..., 0], gap[..., 1]) < corridor).any(axis=0)
    return dict(lo=np.ascontiguousarray(lo[keep]), hi=np.ascontiguousarray(hi[keep]), ground=-1.73)


def boxes_near(world, T_world_sensor, reach=81.0):
    """The sub-world a sensor at this pose can see (boxes whose footprint comes within `reach`): what lidar_scan needs."""
    o = np.asarray(T_world_sensor, np.float64)[:2, 3]
    gap = np.maximum(np.maximum(world["lo"][:, :2] - o, o - world["hi"][:, :2]), 0.0)
    near = np.hypot(gap[:, 0], gap[:, 1]) <= reach
    return dict(lo=world["lo"][near], hi=world["hi"][near], ground=world["ground"])


def se3(yaw_deg=0.0, t=(0.0, 0.0, 0.0), pitch_deg=0.0, roll_deg=0.0):
    """RollPitchYaw = Rz(yaw) Ry(pitch) Rx(roll), as registration/3d/rigid_transform.cpp:29-35."""
    y, p, r = np.deg2rad([yaw_deg, pitch_deg, roll_deg])
    Rz = np.array([[np.cos(y), -np.sin(y), 0], [np.sin(y), np.cos(y), 0], [0, 0, 1]])
    Ry = np.array([[np.cos(p), 0, np.sin(p)], [0, 1, 0], [-np.sin(p), 0, np.cos(p)]])
    Rx = np.array([[1, 0, 0], [0, np.cos(r), -np.sin(r)], [0, np.sin(r), np.cos(r)]])
    T = np.eye(4)
    T[:3, :3] = Rz @ Ry @ Rx
    T[:3, 3] = t
    return T


def lidar_scan(world, T_world_sensor=None, seed=0, n_beams=64, n_az=2000, max_range=80.0,
               noise=0.02, fov=(-24.8, 2.0)):
    """Ray-cast a 64-beam spinning lidar (KITTI HDL-64E-like: 64 x 2000 rays, ~120k returns).
    Returns float32 [n,4] (x,y,z,intensity) in the SENSOR frame, KITTI .bin layout
    (registration/global_localization.cpp:160-182)."""
    T = np.eye(4) if T_world_sensor is None else np.asarray(T_world_sensor, np.float64)
    el = np.deg2rad(np.linspace(fov[0], fov[1], n_beams))
    az = np.linspace(0.0, 2 * np.pi, n_az, endpoint=False)
    ce, se_ = np.cos(el)[:, None], np.sin(el)[:, None]
    d = np.stack([ce * np.cos(az)[None, :], ce * np.sin(az)[None, :], se_ * np.ones_like(az)[None, :]], -1)
    d = d.reshape(-1, 3)
    dw = d @ T[:3, :3].T
    o = T[:3, 3]
    tbest = np.full(dw.shape[0], np.inf)
    # ground
    with np.errstate(divide="ignore", invalid="ignore"):
        tg = (world["ground"] - o[2]) / dw[:, 2]
    tg = np.where((dw[:, 2] < 0) & (tg > 0), tg, np.inf)
    tbest = np.minimum(tbest, tg)
    # boxes (slab test), a few at a time to bound memory
    lo, hi = world["lo"], world["hi"]
    with np.errstate(divide="ignore", invalid="ignore"):
        inv = 1.0 / dw
    for b in range(lo.shape[0]):
        t1 = (lo[b] - o) * inv
        t2 = (hi[b] - o) * inv
        tn = np.nanmax(np.minimum(t1, t2), axis=1)
        tf = np.nanmin(np.maximum(t1, t2), axis=1)
        hit = (tn <= tf) & (tn > 0.5)
        tbest = np.where(hit & (tn < tbest), tn, tbest)
    ok = tbest < max_range
    key = rng_key(seed, 7)
    nz = rng_gauss(key, np.arange(dw.shape[0], dtype=np.uint64)).astype(np.float64) * noise
    r = (tbest + nz)[ok]
    pts = d[ok] * r[:, None]
    inten = rng_uniform(key, np.arange(dw.shape[0], dtype=np.uint64) + np.uint64(1 << 32))[ok]
    return np.concatenate([pts.astype(np.float32), inten[:, None].astype(np.float32)], 1)


def scan_variant(xyz, T=None, noise_sigma=0.0, seed=0):
    """numpy twin of gloc_scan_store_add_variant (csrc/scan_store.hip::scan_variant_kernel): point i =
    T p_i (fixed un-fused fp32 order ((r0 x + r1 y) + r2 z) + t) + sigma * gauss(key, 3 i + a).
    Same bits as the device kernel."""
    p = np.ascontiguousarray(xyz, np.float32)[:, :3]
    T = np.eye(4, dtype=np.float32) if T is None else np.asarray(T, np.float32)
    x, y, z = p[:, 0], p[:, 1], p[:, 2]
    out = np.empty((p.shape[0], 3), np.float32)
    key = rng_key(seed, 11)
    sg = np.float32(noise_sigma)
    for a in range(3):
        v = ((T[a, 0] * x + T[a, 1] * y) + T[a, 2] * z) + T[a, 3]
        nz = sg * rng_gauss(key, np.arange(p.shape[0], dtype=np.uint64) * np.uint64(3) + np.uint64(a))
        out[:, a] = v + nz
    return out


def write_kitti_bin(path, scan_xyzi):
    np.ascontiguousarray(scan_xyzi, np.float32).tofile(path)


def read_kitti_bin(path):
    return np.fromfile(path, np.float32).reshape(-1, 4)
