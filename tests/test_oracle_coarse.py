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


def test_scale_estimate_and_its_acceptance(oracle_mod, grids):
    """The reference's match returns a scale and accepts only |1 - scale| < 0.1 (loop_detector.cpp:262-272).  Here the
    factor by which the query, scaled about the sensor, overlaps the database grid most: ~1 for true revisits; a query
    whose coordinates are stretched by 6 % comes back as 1 / 1.06, one stretched by 30 % is not accepted."""
    from gloc3d_amd import synth
    for case in CASES:
        r = oracle_mod.coarse_match(grids[case], grids["A"])
        assert r["ok"] and abs(r["scale"] - 1.0) < 0.02 and r["matched"] == r["overlap"]
    w = synth.make_world(1001)
    base = synth.lidar_scan(w, synth.se3(30.0, (6.0, -4.0, 0.0)), seed=2, n_az=1000)
    scales = {}
    for f in (1.0, 1.06, 1.3):
        sc = base.copy()
        sc[:, :2] *= np.float32(f)
        img, info = oracle_mod.bev_project(sc)
        r = oracle_mod.coarse_match(oracle_mod.CoarseGrid(img, info["ox"], info["oy"], info["resolution"]), grids["A"])
        scales[f] = r
    assert scales[1.0]["ok"] and abs(scales[1.0]["scale"] - 1.0) < 0.02
    assert abs(scales[1.06]["scale"] - 1.0 / 1.06) < 0.02, scales[1.06]  # the database is SMALLER than the stretched query
    assert not scales[1.3]["ok"]
    e = oracle_mod.CoarseGrid(np.full((10, 10), 255, np.uint8), -1.0, -1.0, 0.2)
    assert oracle_mod.coarse_match(e, grids["A"])["scale"] == 0.0       # no overlap at all


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


def _exhaustive_optimum(q_cells, d_cells, n_yaw=360, max_shift=64, cell_px=2):
    """An independent search for the same objective: EVERY yaw step x EVERY shift within +-max_shift cells, the
    overlap of the rotated query cells with the 3x3-dilated database map as one FFT cross-correlation per yaw step
    (numpy only) -- no projections, no shortlist, no refinement window.  Returns (best overlap, k, tx, ty)."""
    G, H = 512, 256
    du, dv = (d_cells & 0xFFFF).astype(np.int64), (d_cells >> 16).astype(np.int64)
    D = np.zeros((G, G), np.float64)
    for a in (-1, 0, 1):
        for b in (-1, 0, 1):
            uu, vv = du + a, dv + b
            m = (uu >= 0) & (uu < G) & (vv >= 0) & (vv < G)
            D[vv[m], uu[m]] = 1.0
    FD = np.fft.rfft2(D, (2 * G, 2 * G))
    x = (((q_cells & 0xFFFF).astype(np.int64) - H) * cell_px).astype(np.float32) + np.float32(0.5 * (cell_px - 1))
    y = (((q_cells >> 16).astype(np.int64) - H) * cell_px).astype(np.float32) + np.float32(0.5 * (cell_px - 1))
    best = (-1, 0, 0, 0)
    for k in range(n_yaw):
        a = 2.0 * np.pi * k / n_yaw
        c, s = np.float32(np.cos(a)), np.float32(np.sin(a))
        rx, ry = c * x - s * y, s * x + c * y                               # fp32, un-fused, as the specification says
        rnd = lambda v: (np.sign(v) * np.floor(np.abs(v).astype(np.float64) + 0.5)).astype(np.int64)   # halves away from zero
        u, v = np.floor_divide(rnd(rx), cell_px) + H, np.floor_divide(rnd(ry), cell_px) + H
        m = (u >= 0) & (u < G) & (v >= 0) & (v < G)
        Q = np.zeros((G, G), np.float64)
        np.add.at(Q, (v[m], u[m]), 1.0)
        C = np.fft.irfft2(np.conj(np.fft.rfft2(Q, (2 * G, 2 * G))) * FD, (2 * G, 2 * G))   # C[ty, tx] = sum Q[v, u] D[v + ty, u + tx]
        T = max_shift
        win = np.rint(np.block([[C[-T:, -T:], C[-T:, :T + 1]], [C[:T + 1, -T:], C[:T + 1, :T + 1]]])).astype(np.int64)
        iy, ix = np.unravel_index(np.argmax(win), win.shape)
        if win[iy, ix] > best[0]:
            best = (int(win[iy, ix]), k, int(ix) - T, int(iy) - T)
    return best


@pytest.mark.parametrize("case", [CASES[1], CASES[3]])
def test_the_shortlisted_search_finds_the_exhaustive_optimum(oracle_mod, grids, case):
    """The restatement prunes: projections pick 12 yaw steps, a 9 x 9 window is verified around each.  An independent
    exhaustive search of the same objective (360 yaw steps x 129 x 129 shifts, FFT correlations in numpy) must not
    find anything better, and must agree on where the optimum is."""
    r = oracle_mod.coarse_match(grids[case], grids["A"])
    over, k, tx, ty = _exhaustive_optimum(grids[case].cells(), grids["A"].cells())
    assert r["overlap"] == over and r["k"] == k
    assert abs(r["xy_yaw"][0] - tx * 0.4) < 1e-5 and abs(r["xy_yaw"][1] - ty * 0.4) < 1e-5
