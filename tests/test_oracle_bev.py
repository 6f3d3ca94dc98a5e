"""CPU: the BEV projection oracle (oracle/bev_oracle.c, the reference's long form: voxel grid, hit
and miss tables, x-ray projection) against an independent numpy statement of what it reduces to."""
import numpy as np
import pytest

from util import bev_cases, bev_crop_pad_numpy, bev_numpy

CASES = bev_cases()


@pytest.mark.parametrize("name", sorted(CASES))
def test_long_form_equals_column_count(oracle_mod, name):
    pts = CASES[name]
    img, info = oracle_mod.bev_project(pts)
    ref, rinfo = bev_numpy(pts)
    assert info["n_returns"] == rinfo["n_returns"]
    if ref is None:
        assert img is None
        return
    for k in ("min_ix", "min_iy", "max_ix", "max_iy", "width", "height"):
        assert info[k] == rinfo[k], k
    assert np.array_equal(img, ref)
    assert info["ox"] == info["min_ix"] * float(np.float32(0.2))
    assert info["oy"] == info["min_iy"] * float(np.float32(0.2))


def test_cases_exercise_both_outcomes(oracle_mod):
    img, info = oracle_mod.bev_project(CASES["lidar"])
    assert (img == 0).sum() > 500 and (img == 255).sum() > 500
    assert info["n_cells_known"] > info["n_cells_obstructed"] > 0      # misses exist and are ignored
    img1, _ = oracle_mod.bev_project(CASES["one_point"])
    assert img1.shape == (1, 1) and img1[0, 0] == 255                  # a single hit voxel: 0.55 < 0.9
    img2, _ = oracle_mod.bev_project(CASES["one_column"])
    assert img2.shape == (1, 1) and img2[0, 0] == 0                    # two voxels in one column
    # range boundary cases really straddle the limit
    _, i3 = oracle_mod.bev_project(CASES["range_edge"])
    assert 0 < i3["n_returns"] < CASES["range_edge"].shape[0]


@pytest.mark.parametrize("shape,out", [((703, 743), (768, 768)), ((1001, 990), (768, 768)),
                                       ((1001, 400), (768, 768)), ((5, 7), (8, 6)), ((9, 4), (4, 9)),
                                       ((768, 768), (768, 768)), ((3, 3), (101, 77))])
def test_crop_pad(oracle_mod, shape, out):
    rng = np.random.default_rng(shape[0] * 1000 + shape[1])
    img = np.where(rng.random(shape) < 0.3, 0, 255).astype(np.uint8)
    got = oracle_mod.bev_crop_pad(img, out_w=out[0], out_h=out[1])
    assert np.array_equal(got, bev_crop_pad_numpy(img, out[0], out[1]))
    chw = oracle_mod.bev_to_chw_f32(got)
    assert chw.shape == (3, out[1], out[0])
    assert np.array_equal(chw, np.transpose(got, (2, 0, 1)).astype(np.float32) / np.float32(255))
    assert set(np.unique(chw)) <= {0.0, 1.0}


def test_empty_cloud(oracle_mod):
    img, info = oracle_mod.bev_project(np.zeros((0, 3), np.float32))
    assert img is None and info["n_returns"] == 0
    far = np.full((10, 3), 90.0, np.float32)
    img, info = oracle_mod.bev_project(far)
    assert img is None and info["n_returns"] == 0
