"""GPU: the device ray-caster that makes bench.py's 4541 distinct place scans (gloc_scan_store_add_raycast_batch) against
its numpy twin gloc3d_amd/synth.py::lidar_scan -- the same rays, slab test and counter-RNG range noise in fp64, one
rounding to fp32.  Bench / test support, not the hot path; what matters is that the data the headline runs on is what
the generator's description says."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def road():
    from gloc3d_amd import synth
    traj, xy = synth.loop_trajectory(400, 328.0)
    return traj, synth.make_road_world(1001, xy)


def test_device_cast_equals_the_numpy_twin(capi, road):
    from gloc3d_amd import synth
    traj, world = road
    store = capi.ScanStore()
    poses = [traj[7], traj[200] @ synth.se3(3.0, (0.3, -0.4, 0.02)), synth.se3(0.0, tuple(traj[90][:3, 3]), pitch_deg=2.0, roll_deg=-1.5)]
    ids = store.add_raycast(world, poses, np.array([11, 12, 13], np.uint64), n_az=500)
    for sid, T, seed in zip(ids, poses, (11, 12, 13)):
        dev = store.download(sid)
        ref = synth.lidar_scan(synth.boxes_near(world, T), T, seed=seed, n_az=500)[:, :3]
        assert abs(len(dev) - len(ref)) <= 2                 # (a range within 1e-9 of max_range may fall either way)
        if len(dev) == len(ref):
            assert np.abs(dev - ref).max() < 1e-4
        assert len(dev) > 0.9 * 64 * 500                      # an HDL-64-like return rate on this road
    # one by one == in a batch, bit for bit; the same call twice gives the same bits
    one = [store.download(store.add_raycast(world, [T], np.array([s], np.uint64), n_az=500)[0]) for T, s in zip(poses, (11, 12, 13))]
    for sid, o in zip(ids, one):
        assert (store.download(sid).view(np.uint32) == o.view(np.uint32)).all()
    store.close()


def test_edge_cases(capi):
    from gloc3d_amd import synth
    store = capi.ScanStore()
    empty_world = dict(lo=np.zeros((0, 3)), hi=np.zeros((0, 3)), ground=-1.73)
    sid = store.add_raycast(empty_world, [np.eye(4)], np.array([5], np.uint64), n_beams=8, n_az=16)[0]      # ground returns only
    dev = store.download(sid)
    ref = synth.lidar_scan(empty_world, np.eye(4), seed=5, n_beams=8, n_az=16)[:, :3]
    assert len(dev) == len(ref) and np.abs(dev - ref).max() < 1e-4
    up = synth.se3(0.0, (0, 0, 0), pitch_deg=-89.0)                                                        # looking at the sky: no return at all
    sid = store.add_raycast(empty_world, [up], np.array([5], np.uint64), n_beams=4, n_az=8, fov=(0.0, 0.5))[0]
    assert store.points(sid) == len(synth.lidar_scan(empty_world, up, seed=5, n_beams=4, n_az=8, fov=(0.0, 0.5)))
    with pytest.raises(capi.GlocError):
        store.add_raycast(empty_world, [np.eye(4)], np.array([5], np.uint64), n_beams=4096, n_az=4096)     # too many rays
    store.close()
