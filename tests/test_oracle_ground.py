"""CPU: the ground pre-alignment oracle (oracle/ground_oracle.c) against independent numpy statements
and known-answer scenes."""
import numpy as np
import pytest

from util import ground_scene


def _knn_numpy(p, k):
    d = p[:, None, :].astype(np.float32) - p[None, :, :].astype(np.float32)
    d2 = (d[..., 0] * d[..., 0] + d[..., 1] * d[..., 1]) + d[..., 2] * d[..., 2]   # fp32, same order
    order = np.lexsort((np.broadcast_to(np.arange(p.shape[0]), d2.shape), d2), axis=1)[:, :k]
    return order.astype(np.uint32), np.take_along_axis(d2, order, 1)


def test_knn_exact_with_ties(oracle_mod):
    rng = np.random.default_rng(3)
    p = rng.uniform(-5, 5, (700, 3)).astype(np.float32)
    p[100:140] = p[0:40]                      # duplicates: equal distances, smaller index first
    p[200:230] = np.round(p[200:230])         # lattice points: more exact ties
    idx, d2 = oracle_mod.ground_knn(p, 10)
    ri, rd = _knn_numpy(p, 10)
    assert (idx == ri).all() and (d2.view(np.uint32) == rd.view(np.uint32)).all()
    assert (idx[:100, 0] == np.arange(100)).all() and (d2[:, 0] == 0).all()   # itself first
    idx, d2 = oracle_mod.ground_knn(p[:6], 10)                                 # fewer points than k
    assert (idx[:, 6:] == 0xFFFFFFFF).all() and (idx[:, :6] != 0xFFFFFFFF).all()


def test_normals_of_planes(oracle_mod):
    rng = np.random.default_rng(4)
    xy = rng.uniform(-8, 8, (1500, 2))
    floor = np.c_[xy, np.full(1500, -1.7)].astype(np.float32)                  # below the sensor
    wall = np.c_[np.full(800, 6.0), rng.uniform(-4, 4, 800), rng.uniform(-1.5, 2, 800)].astype(np.float32)
    ceil = np.c_[rng.uniform(-3, 3, (600, 2)), np.full(600, 2.5)].astype(np.float32)
    for cloud, want_n, want_bin in ((floor, (0, 0, 1), 17), (wall, (-1, 0, 0), 9), (ceil, (0, 0, -1), 0)):
        idx, _ = oracle_mod.ground_knn(cloud, 10)
        nrm, bins = oracle_mod.ground_normals(cloud, idx)
        assert np.allclose(nrm, want_n, atol=1e-5)                             # flipped towards the origin
        assert (bins == want_bin).all()
    # every bin edge: a normal at elevation e lands in floor((e + 90) / 10)
    for e in np.arange(-89.0, 90.0, 2.0):
        n = np.array([np.cos(np.deg2rad(e)), 0.0, np.sin(np.deg2rad(e))])
        u, v = np.array([0.0, 1.0, 0.0]), np.cross(n, [0.0, 1.0, 0.0])
        patch = (-12 * n + rng.uniform(-0.5, 0.5, (40, 1)) * u + rng.uniform(-0.5, 0.5, (40, 1)) * v).astype(np.float32)
        idx, _ = oracle_mod.ground_knn(patch, 10)
        _, bins = oracle_mod.ground_normals(patch, idx)
        assert (bins == int(np.floor((e + 90.0) / 10.0))).all(), e


def test_transform_from_plane_properties(oracle_mod):
    rng = np.random.default_rng(5)
    for _ in range(200):
        n = rng.standard_normal(3)
        n[2] = abs(n[2]) + 0.3 * rng.random() + 0.05                           # mostly upward, any tilt
        if rng.random() < 0.5:
            n = -n                                                             # the fit may come out downward
        scale = rng.uniform(0.5, 2.0)
        d = rng.uniform(-3, 3)
        T = oracle_mod.ground_transform_from_plane(np.r_[n * scale, d * scale].astype(np.float32))
        R = T[:3, :3].astype(np.float64)
        assert np.allclose(R @ R.T, np.eye(3), atol=1e-6) and abs(np.linalg.det(R) - 1) < 1e-6
        up = n / np.linalg.norm(n) * (1 if n[2] > 0 else -1)
        assert np.allclose(R @ up, [0, 0, 1], atol=1e-5)                       # the ground normal becomes z
        assert abs(T[2, 3] - abs(d) / np.linalg.norm(n)) < 1e-5 and T[0, 3] == 0 and T[1, 3] == 0
        assert abs(R[1, 0]) < 1e-6                                             # RollPitchYaw(r, p, 0): no yaw term
    T = oracle_mod.ground_transform_from_plane(np.array([0, 0, 1, 1.73], np.float32))
    assert np.allclose(np.abs(np.diag(T[:3, :3])), 1, atol=1e-6) and abs(T[2, 3] - 1.73) < 1e-6


@pytest.mark.parametrize("roll,pitch", [(0.0, 0.0), (3.0, -2.0), (-6.0, 4.0)])
def test_estimate_recovers_the_ground(oracle_mod, roll, pitch):
    cloud, height = ground_scene(roll, pitch)
    T, info = oracle_mod.ground_estimate(cloud)
    assert info["found"] == 1 and info["ground_bin"] == 17 and info["n_ground"] > 0.5 * info["n_near"]
    assert info["hist"].sum() == info["n_near"] and info["inliers"] > 0.9 * info["n_ground"]
    near = cloud[np.einsum("ij,ij->i", cloud[:, :3], cloud[:, :3]) < 400][:, :3]
    g = near @ T[:3, :3].T + T[:3, 3]
    floor = g[np.abs(g[:, 2]) < 0.3]
    assert floor.shape[0] > 0.5 * near.shape[0] and abs(np.median(floor[:, 2])) < 0.03   # ground at z = 0
    assert abs(T[2, 3] - height) < 0.05
    tilt = np.degrees(np.arccos(np.clip(abs(T[2, 2]), -1, 1)))
    assert abs(tilt - np.degrees(np.arccos(np.cos(np.radians(roll)) * np.cos(np.radians(pitch))))) < 0.5


def test_estimate_without_ground(oracle_mod):
    rng = np.random.default_rng(6)
    wall = np.c_[np.full(500, 5.0), rng.uniform(-4, 4, 500), rng.uniform(-1, 2, 500)].astype(np.float32)
    T, info = oracle_mod.ground_estimate(wall)            # vertical normals only: bins 5..12 excluded
    assert info["found"] == 0 and (T == np.eye(4)).all() and info["ground_bin"] == -1
    T, info = oracle_mod.ground_estimate(np.full((10, 3), 50.0, np.float32))   # nothing within 20 m
    assert info["found"] == 0 and info["n_near"] == 0
    T, info = oracle_mod.ground_estimate(np.zeros((0, 3), np.float32))
    assert info["found"] == 0
