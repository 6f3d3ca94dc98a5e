"""GPU: the HIP BEV occupancy projection through the C ABI against the oracle's long form
(oracle/bev_oracle.c), byte for byte."""
import numpy as np
import pytest

from util import bev_cases, bev_crop_pad_numpy

pytestmark = pytest.mark.gpu
CASES = bev_cases()


@pytest.fixture(scope="module")
def proj(capi):
    p = capi.BevProjector()
    yield p
    p.close()


def _expect(oracle_mod, pts, out_w=768, out_h=768, **kw):
    img, info = oracle_mod.bev_project(pts, **kw)
    if img is None:
        return None, info, bev_crop_pad_numpy(None, out_w, out_h)
    return img, info, oracle_mod.bev_crop_pad(img, out_w=out_w, out_h=out_h)


@pytest.mark.parametrize("name", sorted(CASES))
def test_project_matches_oracle(capi, oracle_mod, proj, name):
    pts = CASES[name]
    raw, oinfo, want = _expect(oracle_mod, pts)
    got, info = proj.project(pts)
    assert info["n_returns"] == oinfo["n_returns"]
    assert np.array_equal(got, want)
    if raw is None:
        assert info["empty"] == 1 and info["width"] == 0
        return
    for k in ("min_ix", "min_iy", "max_ix", "max_iy", "width", "height", "ox", "oy", "resolution"):
        assert info[k] == oinfo[k], k
    assert np.array_equal(proj.raw_image(info), raw)


def test_tensor_format(capi, oracle_mod, proj):
    pts = CASES["lidar"]
    _, _, want = _expect(oracle_mod, pts)
    got, _ = proj.project(pts, capi.default_bev_params(format=capi.BEV_F32_CHW))
    assert got.dtype == np.float32 and got.shape == (3, 768, 768)
    assert np.array_equal(got, oracle_mod.bev_to_chw_f32(want))


@pytest.mark.parametrize("out_w,out_h", [(768, 768), (1024, 512), (101, 77), (64, 2048), (7, 5)])
@pytest.mark.parametrize("fmt", [0, 1])
def test_output_sizes(capi, oracle_mod, proj, out_w, out_h, fmt):
    for name in ("wide", "tiny_image"):
        pts = CASES[name]
        _, _, want = _expect(oracle_mod, pts, out_w, out_h)
        got, _ = proj.project(pts, capi.default_bev_params(out_width=out_w, out_height=out_h, format=fmt))
        if fmt == 1:
            want = oracle_mod.bev_to_chw_f32(want)
        assert np.array_equal(got, want)


@pytest.mark.parametrize("res,rng_", [(0.2, 100.0), (0.5, 100.0), (0.1, 40.0), (0.25, 64.0)])
def test_other_resolutions(capi, oracle_mod, proj, res, rng_):
    pts = CASES["wide"]
    raw, oinfo, want = _expect(oracle_mod, pts, resolution=res, max_range=rng_)
    got, info = proj.project(pts, capi.default_bev_params(resolution=res, max_range=rng_))
    assert info["n_returns"] == oinfo["n_returns"] and info["width"] == oinfo["width"]
    assert np.array_equal(got, want)
    assert np.array_equal(proj.raw_image(info), raw)


def test_strides_and_pad_colour(capi, oracle_mod, proj):
    pts4 = CASES["lidar"]
    a, _ = proj.project(pts4)
    b, _ = proj.project(np.ascontiguousarray(pts4[:, :3]))
    assert np.array_equal(a, b)
    c, _ = proj.project(CASES["tiny_image"], capi.default_bev_params(pad_bgr=(7, 8, 9)))
    raw, _, _ = _expect(oracle_mod, CASES["tiny_image"])
    assert np.array_equal(c, bev_crop_pad_numpy(raw, pad=(7, 8, 9)))


def test_empty_inputs(capi, proj):
    got, info = proj.project(np.zeros((0, 3), np.float32))
    assert info["empty"] == 1 and info["n_returns"] == 0
    assert np.array_equal(got, bev_crop_pad_numpy(None))
    got, info = proj.project(np.full((10, 3), 90.0, np.float32))
    assert info["empty"] == 1 and np.array_equal(got, bev_crop_pad_numpy(None))
    with pytest.raises(capi.GlocError):
        proj.raw_image(dict(width=1, height=1))


def test_repeat_is_clean(capi, oracle_mod, proj):
    """Column words are cleared between projections: a big scan followed by a small one."""
    proj.project(CASES["gauss_dense"])
    got, _ = proj.project(CASES["one_point"])
    _, _, want = _expect(oracle_mod, CASES["one_point"])
    assert np.array_equal(got, want)


def test_batch_device(capi, oracle_mod, proj):
    import torch
    names = ["lidar", "one_point", "wide", "non_finite", "tiny_image", "gauss_dense"]
    clouds = [np.ascontiguousarray(CASES[n][:, :3]) for n in names]
    clouds.insert(2, np.zeros((0, 3), np.float32))                    # an empty scan inside the batch
    off = np.concatenate([[0], np.cumsum([c.shape[0] for c in clouds])]).astype(np.uint64)
    d_xyz = torch.from_numpy(np.concatenate(clouds)).cuda()
    for fmt, dt in ((capi.BEV_U8_HWC3, torch.uint8), (capi.BEV_F32_CHW, torch.float32)):
        p = capi.default_bev_params(format=fmt, out_width=512, out_height=384)
        shape = (len(clouds), 384, 512, 3) if fmt == 0 else (len(clouds), 3, 384, 512)
        d_out = torch.empty(shape, dtype=dt, device="cuda")
        infos = proj.project_batch_device(d_xyz.data_ptr(), off, 3, d_out.data_ptr(), p)
        out = d_out.cpu().numpy()
        for i, c in enumerate(clouds):
            raw, oinfo, want = _expect(oracle_mod, c, 512, 384)
            if fmt == 1:
                want = oracle_mod.bev_to_chw_f32(want)
            assert np.array_equal(out[i], want), i
            assert infos[i]["n_returns"] == oinfo["n_returns"]
            if raw is not None:
                assert np.array_equal(proj.raw_image(infos[i], scan=i), raw)
    # asynchronous form: no infos, caller synchronises
    proj.project_batch_device(d_xyz.data_ptr(), off, 3, d_out.data_ptr(), p, want_info=False)
    proj.synchronize()
    assert np.array_equal(d_out.cpu().numpy(), out)


def test_invalid_params(capi, proj):
    pts = CASES["one_point"]
    for kw in (dict(resolution=0.0), dict(max_range=-1.0), dict(resolution=0.01, max_range=100.0),
               dict(out_width=0), dict(format=9)):
        with pytest.raises(capi.GlocError):
            proj.project(pts, capi.default_bev_params(**kw))


def test_loop_detector_projection(capi, oracle_mod):
    """The host mirror's get_projected_grid / get_place_input (loop_detector.cpp:122-151)."""
    from gloc3d_amd.loop_detector import RpyPCLoopDetector
    det = RpyPCLoopDetector(k_dim=16)
    try:
        pts = CASES["lidar"]
        raw, oinfo, want = _expect(oracle_mod, pts)
        img, (ox, oy, res) = det.get_projected_grid(pts)
        assert np.array_equal(img, raw)
        assert (ox, oy, res) == (oinfo["ox"], oinfo["oy"], oinfo["resolution"])
        x, _ = det.get_place_input(pts)
        assert x.shape == (1, 3, 768, 768) and np.array_equal(x[0], oracle_mod.bev_to_chw_f32(want))
        with pytest.raises(ValueError):
            det.get_projected_grid(np.full((4, 3), 99.0, np.float32))
    finally:
        det.close()
