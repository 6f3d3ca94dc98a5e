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
