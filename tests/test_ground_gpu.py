"""GPU: the HIP ground pre-alignment through the C ABI against the oracle (oracle/ground_oracle.c)."""
import numpy as np
import pytest

from util import bits, ground_scene

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module", params=["culled", "exhaustive"])
def est(capi, request):
    """Every test runs on both forms of the 10-NN search: the lists must not differ."""
    g = capi.GroundEstimator()
    g.set_option(capi.GROUND_OPT_KNN_EXHAUSTIVE, 1 if request.param == "exhaustive" else 0)
    yield g
    g.close()


@pytest.mark.parametrize("n,k", [(700, 10), (1, 3), (5, 10), (129, 16), (4000, 10), (257, 3), (30000, 10)])
def test_knn_bit_exact(est, oracle_mod, n, k):
    rng = np.random.default_rng(n * 31 + k)
    p = rng.uniform(-5, 5, (n, 3)).astype(np.float32)
    if n > 300:
        p[100:140] = p[0:40]                  # duplicates: equal distances keep the smaller index first
        p[200:230] = np.round(p[200:230])     # lattice points: more exact ties
    gi, gd = est.knn(p, k)
    oi, od = oracle_mod.ground_knn(p, k)
    assert (gi == oi).all() and (bits(gd) == bits(od)).all()


def test_normals_and_bins_bit_exact(est, oracle_mod):
    cloud, _ = ground_scene(3.0, -2.0, n_az=300)
    near = np.ascontiguousarray(cloud[np.einsum("ij,ij->i", cloud[:, :3], cloud[:, :3]) < 400][:, :3])
    gn, gb = est.normals(near, 10)
    oi, _ = oracle_mod.ground_knn(near, 10)
    on, ob = oracle_mod.ground_normals(near, oi)
    assert (bits(gn) == bits(on)).all() and (gb == ob).all()
    assert len(set(gb.tolist())) > 4          # the scene has walls and boxes, not only ground


@pytest.mark.parametrize("roll,pitch,stride", [(0.0, 0.0, 4), (3.0, -2.0, 4), (-6.0, 4.0, 3)])
def test_estimate_matches_oracle(est, capi, oracle_mod, roll, pitch, stride):
    cloud, height = ground_scene(roll, pitch)
    cloud = np.ascontiguousarray(cloud[:, :stride])
    T, info, moved = est.estimate(cloud, want_cloud=True)
    oT, oinfo = oracle_mod.ground_estimate(cloud)
    for key in ("n_near", "ground_bin", "n_ground", "best_hyp", "inliers", "iters_used", "found"):
        assert info[key] == oinfo[key], key
    assert (info["hist"] == oinfo["hist"]).all() and (bits(info["plane"]) == bits(oinfo["plane"])).all()
    assert np.abs(T - oT).max() < 1e-6        # a few libm calls on four numbers, on the host in both
    assert abs(T[2, 3] - height) < 0.05
    want = (cloud[:, :3].astype(np.float64) @ T[:3, :3].astype(np.float64).T + T[:3, 3]).astype(np.float32)
    assert np.abs(moved[:, :3] - want).max() < 1e-4
    if stride == 4:
        assert (moved[:, 3] == cloud[:, 3]).all()                      # the extra channel is carried along


def test_other_parameters(est, capi, oracle_mod):
    cloud, _ = ground_scene(2.0, 1.0, n_az=300)
    for kw in (dict(knn=6), dict(plane_thresh=0.03, seed=77), dict(near_range2=100.0), dict(ransac_conf=0.0, ransac_iters=64)):
        T, info = est.estimate(cloud, capi.default_ground_params(**kw))
        oT, oinfo = oracle_mod.ground_estimate(cloud, **kw)
        assert info["best_hyp"] == oinfo["best_hyp"] and info["inliers"] == oinfo["inliers"], kw
        assert info["iters_used"] == oinfo["iters_used"] and (info["hist"] == oinfo["hist"]).all(), kw
        assert np.abs(T - oT).max() < 1e-6


def test_no_ground_and_empty(est, capi):
    rng = np.random.default_rng(6)
    wall = np.c_[np.full(500, 5.0), rng.uniform(-4, 4, 500), rng.uniform(-1, 2, 500)].astype(np.float32)
    T, info, moved = est.estimate(wall, want_cloud=True)
    assert info["found"] == 0 and (T == np.eye(4)).all() and (moved == wall).all()
    T, info = est.estimate(np.full((10, 3), 50.0, np.float32))
    assert info["found"] == 0 and info["n_near"] == 0 and (T == np.eye(4)).all()
    T, info = est.estimate(np.zeros((0, 3), np.float32))
    assert info["found"] == 0
    with pytest.raises(capi.GlocError):
        est.estimate(wall, capi.default_ground_params(knn=40))


def test_transform_from_plane(capi, oracle_mod):
    rng = np.random.default_rng(8)
    for _ in range(50):
        pl = np.r_[rng.standard_normal(3), rng.uniform(-3, 3)].astype(np.float32)
        assert np.abs(capi.ground_transform_from_plane(pl) - oracle_mod.ground_transform_from_plane(pl)).max() < 1e-6


def test_device_entry_point(est, capi, oracle_mod):
    import torch
    cloud, _ = ground_scene(3.0, -2.0, n_az=300)
    d_in = torch.from_numpy(cloud).cuda()
    d_out = torch.empty_like(d_in)
    T, info = est.estimate_device(d_in.data_ptr(), cloud.shape[0], 4, d_out.data_ptr())
    oT, oinfo = oracle_mod.ground_estimate(cloud)
    assert info["inliers"] == oinfo["inliers"] and np.abs(T - oT).max() < 1e-6
    want = (cloud[:, :3].astype(np.float64) @ T[:3, :3].astype(np.float64).T + T[:3, 3]).astype(np.float32)
    assert np.abs(d_out.cpu().numpy()[:, :3] - want).max() < 1e-4


def test_loop_detector_ground_alignment(capi, oracle_mod):
    from gloc3d_amd.loop_detector import RpyPCLoopDetector
    det = RpyPCLoopDetector(k_dim=16)
    try:
        cloud, height = ground_scene(3.0, -2.0, n_az=300)
        T, moved = det.align_to_ground(cloud)
        oT, _ = oracle_mod.ground_estimate(cloud)
        assert np.abs(T - oT).max() < 1e-6 and moved.shape == cloud.shape
        near = moved[np.einsum("ij,ij->i", cloud[:, :3], cloud[:, :3]) < 400]
        floor = near[np.abs(near[:, 2]) < 0.3]
        assert floor.shape[0] > 0.5 * near.shape[0] and abs(np.median(floor[:, 2])) < 0.03
    finally:
        det.close()
