"""CPU: the coarse (x, y, yaw) match restatement -- known-answer poses on synthetic scans (parity unpinned
upstream: the reference's SURF / FLANN / RANSAC-affine live in OpenCV, absent here)."""
import numpy as np
import pytest

CASES = [(0.0, (0.0, 0.0)), (30.0, (6.0, -4.0)), (90.0, (8.0, 3.0)), (170.0, (-7.0, 6.0)), (-120.0, (2.0, 9.0))]


@pytest.fixture(scope="module")
def grids(oracle_mod):
    from gloc3d_amd import synth
    w = synth.make_world(1001)
    out = {}
    for key, T in [("A", None)] + [(c, synth.se3(c[0], (c[1][0], c[1][1], 0.0))) for c in CASES]:
        img, info = oracle_mod.bev_project(synth.lidar_scan(w, T, seed=1 if key == "A" else 2, n_az=1000))
        out[key] = oracle_mod.CoarseGrid(img, info["ox"], info["oy"], info["resolution"])
    return out


@pytest.mark.parametrize("case", CASES)
def test_known_pose_is_recovered(oracle_mod, grids, case):
    yaw, (tx, ty) = case
    r = oracle_mod.coarse_match(grids[case], grids["A"])      # the query's pose in A's frame
    assert r["ok"] and r["ratio"] > 0.6
    assert abs(r["xy_yaw"][0] - tx) <= 0.8 and abs(r["xy_yaw"][1] - ty) <= 0.8
    d = (np.degrees(r["xy_yaw"][2]) - yaw + 180.0) % 360.0 - 180.0
    assert abs(d) <= 2.0


def test_grid_and_rejection(oracle_mod, grids):
    from gloc3d_amd import synth
    cells = grids["A"].cells()
    assert len(cells) > 300 and len(np.unique(cells)) == len(cells)
    assert (np.diff(cells.astype(np.int64)) > 0).all()                    # row-major order
    # an empty image and a tiny one: no match, no crash
    empty = oracle_mod.CoarseGrid(np.full((10, 10), 255, np.uint8), -1.0, -1.0, 0.2)
    r = oracle_mod.coarse_match(empty, grids["A"])
    assert not r["ok"] and r["overlap"] == 0
    # a different world overlaps far less than the same one
    img, info = oracle_mod.bev_project(synth.lidar_scan(synth.make_world(2002), None, seed=3, n_az=1000))
    other = oracle_mod.CoarseGrid(img, info["ox"], info["oy"], info["resolution"])
    assert oracle_mod.coarse_match(other, grids["A"])["ratio"] < 0.6 < oracle_mod.coarse_match(grids[CASES[1]], grids["A"])["ratio"]
