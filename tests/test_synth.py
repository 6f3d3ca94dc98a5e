"""CPU: the numpy generators and the C oracle's generator produce identical bits."""
import numpy as np

from util import bits


def test_rng_and_iid_match_c(oracle_mod):
    from gloc3d_amd import synth
    L = oracle_mod.lib()
    k = int(synth.rng_key(2001, 5))
    assert k == L.oracle_rng_key(2001, 5)
    assert int(synth.rng_draw(k, 77)) == L.oracle_rng_draw(k, 77)
    a = synth.descriptors_iid(2001, 3, 5, 512)
    b = np.empty((5, 512), np.float32)
    L.oracle_synth_iid(2001, 3, 5, 512, b)
    assert (bits(a) == bits(b)).all()


def test_generators_are_row_addressable():
    from gloc3d_amd import synth
    full = synth.descriptors_traj(9, 0, 100, 64)
    part = synth.descriptors_traj(9, 40, 10, 64)
    assert (bits(full[40:50]) == bits(part)).all()
    assert 0.5 < np.linalg.norm(full, axis=1).mean() < 1.2


def test_lidar_scan_shape():
    from gloc3d_amd import synth
    w = synth.make_world(1001)
    s = synth.lidar_scan(w, None, seed=1, n_az=500)
    assert s.shape[1] == 4 and 20000 < s.shape[0] < 32000
    assert np.abs(s[:, :3]).max() < 81


def test_loop_trajectory_and_road_world():
    """Round 6: bench.py's data -- poses at EQUAL arc-length steps along a closed loop of the given length, headings along
    the tangent; a world whose boxes keep clear of the road; the sub-world within a sensor's reach."""
    from gloc3d_amd import synth
    n, length = 500, 410.0
    T, xy = synth.loop_trajectory(n, length)
    assert T.shape == (n, 4, 4) and xy.shape == (n, 2)
    step = np.hypot(*(np.roll(xy, -1, 0) - xy).T)
    assert abs(step.mean() - length / n) < 1e-3 and step.max() - step.min() < 1e-3          # equal steps, the loop closes
    assert np.allclose(T[:, :2, 3], xy) and np.allclose(T[:, 2, 3], 0.0)
    head = T[:, :2, 0]                                                                       # the sensor's x axis ...
    tang = (np.roll(xy, -1, 0) - np.roll(xy, 1, 0))
    tang /= np.linalg.norm(tang, axis=1, keepdims=True)
    assert (np.sum(head * tang, axis=1) > 0.999).all()                                       # ... is the tangent
    assert np.allclose(np.einsum("nij,nkj->nik", T[:, :3, :3], T[:, :3, :3]), np.eye(3), atol=1e-12)
    w = synth.make_road_world(1001, xy)
    assert len(w["lo"]) > 50 and (w["hi"] > w["lo"]).all() and w["ground"] == -1.73
    gap = np.maximum(np.maximum(w["lo"][None, :, :2] - xy[:, None, :], xy[:, None, :] - w["hi"][None, :, :2]), 0.0)
    assert np.hypot(gap[..., 0], gap[..., 1]).min() >= 2.5                                   # no box within the corridor
    w2 = synth.make_road_world(1001, xy)
    assert (w2["lo"] == w["lo"]).all() and (w2["hi"] == w["hi"]).all()                       # deterministic
    assert len(synth.make_road_world(2002, xy)["lo"]) != len(w["lo"]) or not (synth.make_road_world(2002, xy)["lo"] == w["lo"]).all()
    sub = synth.boxes_near(w, T[7], reach=81.0)
    d = np.hypot(*np.maximum(np.maximum(sub["lo"][:, :2] - xy[7], xy[7] - sub["hi"][:, :2]), 0.0).T)
    assert (d <= 81.0).all() and 0 < len(sub["lo"]) < len(w["lo"])
    # a cast from a pose on the road sees ground and boxes, and nothing nearer than the corridor allows above the ground
    sc = synth.lidar_scan(sub, T[7], seed=3, n_az=200)
    assert sc.shape[0] > 0.85 * 64 * 200
    above = sc[sc[:, 2] > -1.5]
    assert np.hypot(above[:, 0], above[:, 1]).min() > 2.0
