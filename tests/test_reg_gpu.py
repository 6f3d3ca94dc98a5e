"""GPU: candidate registration through the C ABI against the oracle (NN pinned by the reference's
nanoflann golden; RANSAC/ICP against the repo's CPU restatement -- parity unpinned upstream)."""
import numpy as np
import pytest

from util import bits, load_nn3_case, load_nn3_fullsize

pytestmark = pytest.mark.gpu
POSE_TOL_M, POSE_TOL_RAD = 1e-4, 1e-4  # north_star: final 4x4 pose within 1e-4 m / 1e-4 rad


def _rot_angle(Ra, Rb):
    """Angle of Ra^T Rb, well conditioned near zero (atan2 of the skew part, not acos of the trace)."""
    E = Ra.astype(np.float64).T @ Rb.astype(np.float64)
    v = 0.5 * np.array([E[2, 1] - E[1, 2], E[0, 2] - E[2, 0], E[1, 0] - E[0, 1]])
    return float(np.arctan2(np.linalg.norm(v), (np.trace(E) - 1) / 2))


@pytest.fixture(scope="module")
def scans():
    from gloc3d_amd import synth
    w = synth.make_world(1001)
    A = synth.lidar_scan(w, None, seed=1001)[:, :3]
    T = synth.se3(5.0, (0.5, -0.3, 0.1))
    B = synth.lidar_scan(w, T, seed=1002)[:, :3]
    C_ = synth.lidar_scan(synth.make_world(77), None, seed=5)[:, :3]  # a different place
    return dict(A=np.ascontiguousarray(A), B=np.ascontiguousarray(B), C=np.ascontiguousarray(C_), T=T)


@pytest.fixture(scope="module", params=["culled", "culled-kd", "exhaustive"])
def reg(capi, request):
    """Every registration test runs on both 1-NN search modes, the culled one on both orders of the target
    index (curve order, and the kd order of gloc_scan_store_build_target_index): the results must not differ."""
    r = capi.Registrar()
    r.set_option(capi.REG_OPT_NN_MODE, capi.REG_NN_EXHAUSTIVE if request.param == "exhaustive" else capi.REG_NN_CULLED)
    r.set_option(capi.REG_OPT_TEMP_TARGET_INDEX, 1 if request.param == "culled-kd" else 0)
    yield r
    r.close()


def test_nn_matches_reference_golden(reg):
    src, tgt, g_idx, g_bits = load_nn3_case()
    idx, d2 = reg.nn(src, tgt)
    assert (idx == g_idx).all() and (bits(d2) == g_bits).all()


def test_nn_full_size_matches_reference_golden(reg):
    """124k x 124k, source moved by an initial guess: indices and d2 bits of the reference's kd-tree
    (tests/golden/nn3_fullsize.npz, made by oracle/_ref), on both search modes."""
    src, tgt, T, g_idx, g_bits = load_nn3_fullsize()
    idx, d2 = reg.nn(src, tgt, T)
    assert (idx == g_idx).all() and (bits(d2) == g_bits).all()


@pytest.fixture(scope="module")
def fullsize_oracle(oracle_mod, scans):
    """RANSAC 3000 + ICP 20 at full size through the CPU checker: one positive and one different-scene
    candidate (~16 s)."""
    kw = dict(ransac_iters=3000, icp_iters=20, max_rmse=1.0)
    return [oracle_mod.reg_one(scans["B"], scans[c], cand_id=i, **kw) for i, c in enumerate("AC")]


def test_full_size_registration_matches_oracle(reg, capi, scans, fullsize_oracle):
    prm = capi.default_reg_params(ransac_iters=3000, icp_iters=20, max_rmse=1.0)
    g = reg.batch(scans["B"], [scans["A"], scans["C"]], params=prm)
    for c, o in enumerate(fullsize_oracle):
        assert np.abs(g["T"][c][:3, 3] - o["T"][:3, 3]).max() < POSE_TOL_M
        assert _rot_angle(g["T"][c][:3, :3], o["T"][:3, :3]) < POSE_TOL_RAD
        assert g["inliers"][c] == o["inliers"] and bool(g["ok"][c]) == o["ok"]
        assert abs(g["rmse"][c] - o["rmse"]) < 1e-5
    assert bool(g["ok"][0]) and not bool(g["ok"][1])
    er, ep = capi_pose_error(scans["T"], g["T"][0])
    assert ep < 1.0 and er < 5.0        # the reference's success criterion (global_localization.cpp:307)


def test_full_size_matches_the_independent_statement(reg, capi, scans):
    """The HIP path against the scipy / numpy statement of SURVEY Appendix B at BASELINE configs[2]'s shape: the
    committed result of tests/golden/make_crosscheck.py::full_size_case (cKDTree + numpy.linalg.svd, ~124 k x 124 k
    points, RANSAC 3000 adaptive + ICP 20), pose within 1e-4 m / 1e-4 rad, on every search mode."""
    import importlib.util, os
    here = os.path.dirname(os.path.abspath(__file__))
    spec = importlib.util.spec_from_file_location("make_crosscheck", os.path.join(here, "golden", "make_crosscheck.py"))
    mc = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mc)
    g = np.load(os.path.join(here, "golden", "reg_crosscheck.npz"))
    assert (g["full_size_crc"] == np.array([mc.crc(scans["B"]), mc.crc(scans["A"])], np.uint64)).all()   # the fixture's inputs
    prm = capi.default_reg_params(ransac_iters=3000, icp_iters=20, max_rmse=1.0, max_final_step=0.03)   # (as the fixture was made)
    out = reg.batch(scans["B"], [scans["A"]], params=prm, stream_ids=[0])
    T, (rmse, inl, hyp, ok, fstep) = g["full_size_T"], g["full_size_meta"]
    assert abs(float(reg.final_steps(1)[0]) - fstep) < 2e-4
    assert np.abs(out["T"][0][:3, 3] - T[:3, 3]).max() < POSE_TOL_M
    assert _rot_angle(out["T"][0][:3, :3], T[:3, :3]) < POSE_TOL_RAD
    assert abs(out["rmse"][0] - rmse) < 1e-4 and bool(out["ok"][0]) == bool(ok)
    assert abs(int(out["inliers"][0]) - int(inl)) <= max(2, int(1e-4 * inl))


def capi_pose_error(T_gt, T):
    from gloc3d_amd import loop_detector as ld
    return ld.pose_error(np.asarray(T_gt, np.float32), T)


@pytest.mark.parametrize("ns,nt", [(1, 1), (5, 255), (300, 256), (257, 257), (4000, 9000), (1025, 3)])
def test_nn_ragged_with_transform(reg, oracle_mod, ns, nt):
    from gloc3d_amd import synth
    rng = np.random.default_rng(ns * 7 + nt)
    src = rng.uniform(-50, 50, (ns, 3)).astype(np.float32)
    tgt = rng.uniform(-50, 50, (nt, 3)).astype(np.float32)
    if nt > 10:
        tgt[nt // 2] = tgt[1]  # duplicate target: the smaller index must win
        src[0] = tgt[1]
    T = synth.se3(12.0, (1.0, -2.0, 0.3)).astype(np.float32)
    idx, d2 = reg.nn(src, tgt, T)
    oi, od = oracle_mod.nn3(oracle_mod.transform_points(T, src), tgt)
    assert (idx == oi).all() and (bits(d2) == bits(od)).all()


def test_nn_lattice_ties(reg, oracle_mod):
    """Exact ties everywhere: lattice targets in shuffled order, sources on cell centres, edge and
    face midpoints (2, 4 or 8 equidistant targets, usually in different sub-blocks and chunks):
    the smallest ORIGINAL index must win, through the tie flag and the slow rescan."""
    rng = np.random.default_rng(5)
    g = np.stack(np.meshgrid(np.arange(24), np.arange(24), np.arange(12), indexing="ij"), -1).reshape(-1, 3)
    tgt = (g[rng.permutation(len(g))] * 0.5).astype(np.float32)            # 6912 points, spacing 0.5 (exact)
    cells = g[(g[:, 0] < 23) & (g[:, 1] < 23) & (g[:, 2] < 11)].astype(np.float32) * 0.5
    src = np.concatenate([cells + np.float32(0.25),                          # cell centres: 8-way ties
                          cells + np.array([0.25, 0, 0], np.float32),        # edge midpoints: 2-way
                          cells + np.array([0.25, 0.25, 0], np.float32),     # face centres: 4-way
                          tgt[::7]]).astype(np.float32)                      # exact hits
    src = src[rng.permutation(len(src))]
    idx, d2 = reg.nn(src, tgt)
    oi, od = oracle_mod.nn3(src, tgt)
    assert (idx == oi).all() and (bits(d2) == bits(od)).all()
    # the same clouds through a warm-started ICP run, against the exhaustive search
    # (reg is one of the modes; the batch outputs must not depend on it)
    from gloc3d_amd import capi as c
    r = reg.batch(src[:4000], [tgt], params=c.default_reg_params(ransac_iters=0, icp_iters=3))
    assert np.isfinite(r["T"]).all()
    first = _LATTICE.setdefault("T", r["T"].copy())
    assert np.abs(first - r["T"]).max() < 2e-6


_LATTICE = {}


def test_nn_contested_minima_are_decided_unfused(reg, oracle_mod):
    """The culled search compares FUSED distances (within 2e-7 of the reference's un-fused ones) and re-decides, with
    dist2() itself, every minimum that another target comes within 1e-6 of.  Adversarial input: every source has a
    ring of 6 targets whose distances differ by a few ulps -- in different directions, so fused and un-fused
    arithmetic order them differently -- scattered over different sub-blocks and chunks.  The returned neighbour and
    distance must be the oracle's (un-fused, smallest index among equals), bit for bit, in every search mode."""
    rng = np.random.default_rng(11)
    g = np.arange(-30, 31, dtype=np.float64)
    src = np.stack(np.meshgrid(g, g, [0.0], indexing="ij"), -1).reshape(-1, 3)            # 3721 sources, 1 m apart (exact)
    r, e = 0.3125, 2.0 ** -13                                                             # exact offsets at |x| <= 31
    ring = []
    for s_ in src:
        k = rng.permutation(4)                     # squared distances r^2 + (k e)^2: relative gaps 1.5e-7, 6e-7, 1.4e-6
        ring += [s_ + [r, k[0] * e, 0], s_ + [-r, k[1] * e, 0], s_ + [k[2] * e, r, 0], s_ + [k[3] * e, -r, 0]]
    far = rng.uniform(-45, 45, (20000, 3)) * np.array([1, 1, 0.05]) + np.array([90.0, 0, 0])    # clutter, elsewhere
    tgt = np.concatenate([np.array(ring), far]).astype(np.float32)
    tgt = tgt[rng.permutation(len(tgt))]
    src = src.astype(np.float32)
    idx, d2 = reg.nn(src, tgt)
    oi, od = oracle_mod.nn3(src, tgt)
    assert (idx == oi).all() and (bits(d2) == bits(od)).all()
    # the construction does what it says: the runner-up is within 1e-6 of the minimum for every source
    dd = np.sort(((src[:300, None, :].astype(np.float64) - tgt[None, :, :].astype(np.float64)) ** 2).sum(-1), axis=1)
    assert (dd[:, 1] / dd[:, 0] - 1.0 < 1e-6).all() and (dd[:, 1] > dd[:, 0]).all()


def test_nn_with_nan_points_in_source_and_target(reg, oracle_mod, scans):
    """A scan file with a few NaN records (the reference's readers do not filter them): the search must neither
    fault nor let them disturb the finite points -- every finite source gets the oracle's neighbour among the
    finite targets, bit for bit, in every search mode; a NaN source gets none."""
    src = np.ascontiguousarray(scans["B"][::6]).copy()
    tgt = np.ascontiguousarray(scans["A"][::3]).copy()
    bad_s, bad_t = np.array([0, 77, 4097, len(src) - 1]), np.array([5, 1000, 1001, len(tgt) - 2])
    src[bad_s, 1] = np.nan
    tgt[bad_t] = np.nan
    idx, d2 = reg.nn(src, tgt)
    fin_s = np.ones(len(src), bool)
    fin_s[bad_s] = False
    fin_t = np.ones(len(tgt), bool)
    fin_t[bad_t] = False
    oi, od = oracle_mod.nn3(src[fin_s], tgt[fin_t])
    back = np.flatnonzero(fin_t)          # positions in tgt of the finite targets
    assert (idx[fin_s] == back[oi]).all() and (bits(d2[fin_s]) == bits(od)).all()
    # and through a registration: finite pose, same inlier count in every mode
    from gloc3d_amd import capi as c
    r = reg.batch(src, [tgt], params=c.default_reg_params(ransac_iters=64, icp_iters=3))
    first = _LATTICE.setdefault("nan", (r["inliers"].copy(), r["T"].copy()))
    assert (first[0] == r["inliers"]).all() and np.isfinite(r["T"]).all() and r["inliers"][0] > 1000


def test_nn_full_size_properties(reg, scans):
    A = scans["A"]
    idx, d2 = reg.nn(A, A)                      # idempotence: every point is its own neighbour
    assert (d2 == 0).all()
    assert (A[idx] == A).all()
    idx, d2 = reg.nn(scans["B"], A, scans["T"].astype(np.float32))
    d = np.linalg.norm((scans["B"].astype(np.float64) @ scans["T"][:3, :3].T + scans["T"][:3, 3]) - A[idx], axis=1)
    assert np.allclose(d ** 2, d2, rtol=1e-3, atol=1e-5)


def test_ransac_hypotheses_bit_exact(reg, oracle_mod, scans):
    s = np.ascontiguousarray(scans["B"][::16])
    t = np.ascontiguousarray(scans["A"][::4])
    corr, _ = oracle_mod.nn3(s, t)
    Rt, valid, inl = reg.ransac_hypotheses(s, t, corr, 99, 7, 400, 0.6)
    L = oracle_mod.lib()
    for h in range(400):
        R = np.zeros(9, np.float32)
        tt = np.zeros(3, np.float32)
        v = L.oracle_ransac_hypothesis(s, t, corr, s.shape[0], 99, 7, h, R, tt)
        assert v == valid[h]
        if v:
            assert (bits(np.concatenate([R, tt])) == bits(Rt[h])).all(), h
            assert L.oracle_count_inliers(s, t, corr, s.shape[0], R, tt, 0.6) == inl[h], h


@pytest.mark.parametrize("conf", [0.99, 0.0, 0.999999])
def test_batch_matches_oracle(reg, capi, oracle_mod, scans, conf):
    """conf 0.99: adaptive stop inside the first phase (64 hypotheses); 0: all 500 scored; 0.999999:
    the stop lands in the second phase."""
    q = np.ascontiguousarray(scans["B"][::16])
    cands = [np.ascontiguousarray(scans["A"][::4]), np.ascontiguousarray(scans["A"][1::5]),
             np.ascontiguousarray(scans["C"][::4])]
    prm = capi.default_reg_params(ransac_iters=500, icp_iters=10, ransac_confidence=conf)
    g = reg.batch(q, cands, params=prm)
    for c, cd in enumerate(cands):
        o = oracle_mod.reg_one(q, cd, cand_id=c, ransac_iters=500, icp_iters=10, ransac_confidence=conf)
        assert np.abs(g["T"][c][:3, 3] - o["T"][:3, 3]).max() < POSE_TOL_M
        assert _rot_angle(g["T"][c][:3, :3], o["T"][:3, :3]) < POSE_TOL_RAD
        assert g["inliers"][c] == o["inliers"] and bool(g["ok"][c]) == o["ok"]
        assert abs(g["rmse"][c] - o["rmse"]) < 1e-5


def test_all_3000_hypotheses_scored_without_scoring_every_pair(reg, capi, oracle_mod, scans):
    """ransac_confidence = 0 (SURVEY App. B's wording of S2: every one of the 3000 hypotheses counts, the best wins, ties
    to the smallest h): the scorer takes the pairs a quarter at a time and drops, in between, the hypotheses that can no
    longer beat the winner of the first 256 (ransac_alive_kernel).  The selected hypothesis, its inlier count and the
    pose are the every-pair oracle's -- on a same-place candidate, a thinner copy and a different scene, whose best
    hypotheses are beaten late; and the same as with a cap of 700 hypotheses' worth of the old every-pair path where
    the winner is among them."""
    q = np.ascontiguousarray(scans["B"][::4])                    # 31 k pairs: eight chunks, two per quarter
    cands = [np.ascontiguousarray(scans["A"][::2]), np.ascontiguousarray(scans["A"][1::5]),
             np.ascontiguousarray(scans["C"][::2])]
    prm = capi.default_reg_params(ransac_iters=3000, icp_iters=2, ransac_confidence=0.0, max_final_step=0.0)
    g = reg.batch(q, cands, params=prm)
    for c, cd in enumerate(cands):
        o = oracle_mod.reg_one(q, cd, cand_id=c, ransac_iters=3000, icp_iters=2, ransac_confidence=0.0, max_final_step=0.0)
        assert g["inliers"][c] == o["inliers"] and bool(g["ok"][c]) == o["ok"], c
        assert np.abs(g["T"][c][:3, 3] - o["T"][:3, 3]).max() < POSE_TOL_M
        assert _rot_angle(g["T"][c][:3, :3], o["T"][:3, :3]) < POSE_TOL_RAD


@pytest.mark.parametrize("ransac,icp,init", [(0, 5, False), (200, 0, False), (0, 0, True), (100, 3, True)])
def test_batch_modes_and_init_guess(reg, capi, oracle_mod, scans, ransac, icp, init):
    from gloc3d_amd import synth
    q = np.ascontiguousarray(scans["B"][::40])
    cd = np.ascontiguousarray(scans["A"][::10])
    T0 = synth.se3(4.0, (0.4, -0.2, 0.0)).astype(np.float32) if init else None
    prm = capi.default_reg_params(ransac_iters=ransac, icp_iters=icp, max_corr_dist=3.0 if icp else 0.0)
    g = reg.batch(q, [cd], init_T=None if T0 is None else T0[None], params=prm)
    o = oracle_mod.reg_one(q, cd, init_T=T0, ransac_iters=ransac, icp_iters=icp,
                           max_corr_dist=3.0 if icp else 0.0)
    assert np.abs(g["T"][0] - o["T"]).max() < POSE_TOL_M
    assert abs(g["rmse"][0] - o["rmse"]) < 1e-5 and g["inliers"][0] == o["inliers"]


def test_known_answer_pose_recovery_and_selection(reg, capi, oracle_mod):
    """Constructed SE(3): target = moved source + noise; the lowest-rank successful candidate wins."""
    from gloc3d_amd import synth
    rng = np.random.default_rng(11)
    P = rng.uniform(-25, 25, (6000, 3)).astype(np.float32)
    T = synth.se3(-2.0, (0.3, 0.2, -0.05))
    good = (P.astype(np.float64) @ T[:3, :3].T + T[:3, 3] + rng.normal(0, 0.01, P.shape)).astype(np.float32)
    junk = rng.uniform(-25, 25, (5000, 3)).astype(np.float32)
    prm = capi.default_reg_params(ransac_iters=600, icp_iters=15, min_inlier_ratio=0.8)
    g = reg.batch(P, [junk, good[rng.permutation(len(good))], junk[:3000]], params=prm)
    assert list(g["ok"]) == [False, True, False]
    assert capi.reg_select_first_ok(g["ok"].astype(np.int32)) == 1
    er, ep = oracle_mod.pose_error(T, g["T"][1])
    assert ep < 5e-3 and er < 0.1 and g["rmse"][1] < 0.03


def test_max_rmse_acceptance(reg, capi, oracle_mod, scans):
    """The optional plausibility check on the final pose: a different scene keeps a high RANSAC inlier
    ratio at 0.6 m (shared ground plane) but ends far from it; GPU and oracle agree on ok."""
    q = np.ascontiguousarray(scans["B"][::16])
    cands = [np.ascontiguousarray(scans["A"][::4]), np.ascontiguousarray(scans["C"][::4])]
    for max_rmse, expect in ((0.0, None), (1.0, [True, False])):
        prm = capi.default_reg_params(ransac_iters=300, icp_iters=8, max_rmse=max_rmse, max_final_step=0.0)   # (this gate alone)
        g = reg.batch(q, cands, params=prm)
        o = [oracle_mod.reg_one(q, c, cand_id=i, ransac_iters=300, icp_iters=8, max_rmse=max_rmse, max_final_step=0.0)
             for i, c in enumerate(cands)]
        assert [bool(x) for x in g["ok"]] == [x["ok"] for x in o]
        if expect is not None:
            assert [bool(x) for x in g["ok"]] == expect


def test_convergence_gate_matches_oracle(reg, capi, oracle_mod, scans):
    """gloc_reg_params.max_final_step (off by default; bench.py passes 0.03 m): ok additionally requires that the last ICP update moved the
    matched points by no more than that, RMS -- the ICP has converged.  The measure itself equals the oracle's
    (fp64 moments on both sides), so does ok at thresholds on either side of it; a different scene -- whose ICP keeps
    creeping -- is rejected where the same place, given passes enough, is accepted; without ICP passes the gate is off."""
    q = np.ascontiguousarray(scans["B"][::16])
    cands = [np.ascontiguousarray(scans["A"][::4]), np.ascontiguousarray(scans["C"][::4])]
    for iters in (3, 12, 40):
        o0 = [oracle_mod.reg_one(q, c, cand_id=i, ransac_iters=300, icp_iters=iters, max_final_step=0.0) for i, c in enumerate(cands)]
        g0 = reg.batch(q, cands, params=capi.default_reg_params(ransac_iters=300, icp_iters=iters, max_final_step=0.0))
        steps = reg.final_steps(2)
        for c in range(2):
            assert abs(steps[c] - o0[c]["final_step"]) < 1e-5 + 1e-3 * o0[c]["final_step"], (iters, c, steps[c], o0[c]["final_step"])
        for thr in (0.5 * float(steps[0]), 2.0 * float(steps[0]) + 1e-6, 0.03):
            if min(abs(thr - float(s_)) for s_ in steps) < 1e-4:
                continue                                                   # (too close to call in fp32)
            g = reg.batch(q, cands, params=capi.default_reg_params(ransac_iters=300, icp_iters=iters, max_final_step=thr))
            o = [oracle_mod.reg_one(q, c, cand_id=i, ransac_iters=300, icp_iters=iters, max_final_step=thr) for i, c in enumerate(cands)]
            assert [bool(x) for x in g["ok"]] == [x["ok"] for x in o], (iters, thr)
            assert [bool(x) for x in g["ok"]] == [bool(g0["ok"][c]) and steps[c] <= thr for c in range(2)], (iters, thr)
            assert (bits(g["T"]) == bits(g0["T"])).all()                 # a gate changes ok, never a pose
    assert steps[0] < 0.01 < steps[1]              # 40 passes: the same place has converged, the other scene never does
    g = reg.batch(q, cands, params=capi.default_reg_params(ransac_iters=300, icp_iters=0))      # no ICP: nothing to converge
    assert bool(g["ok"][0]) and (reg.final_steps(2) == 0).all()
    # An ICP that stops for want of correspondences (nothing within a micrometre) has NOT converged: its step reads +inf
    # and the check, when asked for, rejects it -- in the kernel and in the checker (it read 0 and passed)
    kw = dict(ransac_iters=300, icp_iters=3, max_corr_dist=1e-6)
    g = reg.batch(q, cands[:1], params=capi.default_reg_params(**kw))
    assert bool(g["ok"][0]) and np.isinf(reg.final_steps(1)[0])
    g = reg.batch(q, cands[:1], params=capi.default_reg_params(max_final_step=0.03, **kw))
    o = oracle_mod.reg_one(q, cands[0], max_final_step=0.03, **kw)
    assert not bool(g["ok"][0]) and not o["ok"] and np.isinf(o["final_step"])


def test_scan_store_ids_equal_host_buffers(reg, capi, scans):
    q = np.ascontiguousarray(scans["B"][::30])
    cands = [np.ascontiguousarray(scans["A"][::9]), np.ascontiguousarray(scans["C"][::9])]
    reg.scan_clear()
    kitti = np.concatenate([q, np.ones((len(q), 1), np.float32)], 1)  # x,y,z,i layout (stride 4)
    qid = reg.scan_upload(kitti)
    ids = [reg.scan_upload(c) for c in cands]
    assert reg.scan_count() == 3
    prm = capi.default_reg_params(ransac_iters=200, icp_iters=4)
    a = reg.batch_ids(qid, ids, params=prm)
    b = reg.batch(q, cands, params=prm)
    assert (bits(a["T"]) == bits(b["T"])).all() and (a["inliers"] == b["inliers"]).all()
    with pytest.raises(capi.GlocError):
        reg.batch_ids(99, ids, params=prm)
    reg.scan_clear()


def test_culled_equals_exhaustive_full_size(capi, scans):
    """Full-size scans, warm-started ICP passes included: all search modes (and every sources-per-lane
    setting of the culled ones), bit for bit."""
    outs = []
    configs = [(capi.REG_NN_EXHAUSTIVE, 2)] + [(capi.REG_NN_CULLED, cs) for cs in (1, 2, 4)]
    for mode, cs in configs:
        r = capi.Registrar()
        r.set_option(capi.REG_OPT_NN_MODE, mode)
        r.set_option(capi.REG_OPT_NN_SRC_PER_LANE, cs)
        prm = capi.default_reg_params(ransac_iters=300, icp_iters=14)  # late passes carry most points
        outs.append((r.batch(scans["B"], [scans["A"], scans["C"]], params=prm),
                     r.nn(scans["B"], scans["A"], scans["T"].astype(np.float32))))
        r.close()
    ref_b, ref_nn = outs[0]
    for (b, nn), cfg in zip(outs[1:], configs[1:]):
        # the 1-NN results are identical bit for bit; the moments are summed per wave about the wave's centre in fp32
        # and recombined in fp64 (culled) or per 2048-slot block in fp64 (exhaustive): 1e-10 apart per ICP step, which
        # the discrete correspondences of 14 steps turn into micrometres (both within 1e-4 of the oracle)
        assert (nn[0] == ref_nn[0]).all() and (bits(nn[1]) == bits(ref_nn[1])).all(), cfg
        assert (b["inliers"] == ref_b["inliers"]).all() and (b["ok"] == ref_b["ok"]).all(), cfg
        assert np.abs(b["T"] - ref_b["T"]).max() < 1e-5, cfg
        assert np.abs(b["rmse"] - ref_b["rmse"]).max() < 1e-6, cfg


def test_store_shared_by_handles_release_and_variants(capi, scans):
    """One scan store, three registration handles: identical results; release keeps HBM flat; the
    device-made variant scans carry the bits of the numpy twin."""
    from gloc3d_amd import synth
    store = capi.ScanStore()
    q = np.ascontiguousarray(scans["B"][::30])
    cands = [np.ascontiguousarray(scans["A"][::9]), np.ascontiguousarray(scans["C"][::9])]
    qid = store.add(np.concatenate([q, np.ones((len(q), 1), np.float32)], 1))   # x y z i
    ids = [store.add(c) for c in cands]
    assert len(store) == 3 and store.points(qid) == len(q)
    assert (store.download(qid) == q).all()
    prm = capi.default_reg_params(ransac_iters=200, icp_iters=4)
    regs = [capi.Registrar(store=store) for _ in range(3)]
    outs = [r.batch_ids(qid, ids, params=prm) for r in regs]
    own = capi.Registrar()
    ref = own.batch(q, cands, params=prm)
    for o in outs:
        assert (bits(o["T"]) == bits(ref["T"])).all() and (o["inliers"] == ref["inliers"]).all()
    with pytest.raises(capi.GlocError):
        store.close()                      # handles still attached
    # transient query scans: ids and memory are recycled
    live0, _ = store.bytes()
    for it in range(6):
        t = store.add(q[: 3000 + 17 * it])
        assert len(store) == 4
        regs[it % 3].batch_ids(t, ids, params=prm)
        regs[it % 3].scan_release(t)
    live1, cached = store.bytes()
    assert len(store) == 3 and live1 == live0 and cached > 0
    with pytest.raises(capi.GlocError):
        regs[0].batch_ids(t, ids, params=prm)  # released id
    # variants made on the device = the numpy twin, bit for bit
    T = synth.se3(3.0, (0.4, -0.1, 0.02)).astype(np.float32)
    v = store.add_variant(ids[0], T, 0.02, 77)
    assert (bits(store.download(v)) == bits(synth.scan_variant(cands[0], T, 0.02, 77))).all()
    v0 = store.add_variant(ids[0])
    assert (store.download(v0) == cands[0]).all()
    for r in regs:
        r.close()
    own.close()
    store.close()


def test_batch_multi_equals_one_query_at_a_time(capi, scans):
    """Several queries in one batch (one launch covers all their candidates): every row equals the
    single-query call, bit for bit -- different query sizes, a missing candidate, both search modes."""
    store = capi.ScanStore()
    A, B, Cc = scans["A"], scans["B"], scans["C"]
    qs = [store.add(np.ascontiguousarray(B[::20])), store.add(np.ascontiguousarray(A[5::33])),
          store.add(np.ascontiguousarray(B[3::45]))]
    cs = [store.add(np.ascontiguousarray(x)) for x in (A[::6], A[1::7], Cc[::6], B[::8])]
    cand = np.array([[cs[0], cs[1], cs[2]], [cs[3], capi.NO_SCAN, cs[0]], [cs[2], cs[1], cs[3]]], np.uint32)
    prm = capi.default_reg_params(ransac_iters=300, icp_iters=6)
    for mode in (capi.REG_NN_CULLED, capi.REG_NN_EXHAUSTIVE):
        r = capi.Registrar(store=store)
        r.set_option(capi.REG_OPT_NN_MODE, mode)
        m = r.batch_multi(qs, cand, params=prm)
        for qi in range(3):
            keep = [c for c in range(3) if cand[qi, c] != capi.NO_SCAN]
            one = r.batch_ids(qs[qi], cand[qi, keep], params=prm, stream_ids=np.array(keep, np.uint32))
            assert (bits(m["T"][qi, keep]) == bits(one["T"])).all()
            assert (m["inliers"][qi, keep] == one["inliers"]).all() and (m["ok"][qi, keep] == one["ok"]).all()
            assert (bits(m["rmse"][qi, keep]) == bits(one["rmse"])).all()
        assert not m["ok"][1, 1] and (m["T"][1, 1] == np.eye(4)).all()
        r.close()
    store.close()


def test_scan_as_query_and_candidate_with_other_sources_per_lane(capi, scans):
    """One store scan is the QUERY of one row and a CANDIDATE of another in a single batch_multi call, at
    sources-per-lane 4 and 1 (the launch order of a scan is one array per setting, built once and never
    rewritten: round 2 rebuilt it in place and a job could read an order that no longer matched its
    group count), and two handles with different settings share the store: every row equals the
    single-query call and the default setting's result."""
    store = capi.ScanStore()
    A, B, Cc = scans["A"], scans["B"], scans["C"]
    sa, sb, sc = (store.add(np.ascontiguousarray(x)) for x in (A[::3], B[::3], Cc[::4]))
    qs = [sa, sb]
    cand = np.array([[sb, sc], [sa, sc]], np.uint32)      # sa / sb: query of one row, candidate of the other
    prm = capi.default_reg_params(ransac_iters=200, icp_iters=5)
    ref = None
    regs = []
    for cs in (2, 4, 1, 4):
        r = capi.Registrar(store=store)
        r.set_option(capi.REG_OPT_NN_SRC_PER_LANE, cs)
        regs.append(r)
        m = r.batch_multi(qs, cand, params=prm)
        for qi in range(2):
            one = r.batch_ids(qs[qi], cand[qi], params=prm)
            assert (bits(m["T"][qi]) == bits(one["T"])).all() and (m["inliers"][qi] == one["inliers"]).all()
        if ref is None:
            ref = m
        # the 1-NN results do not depend on the setting; the fp64 moments are summed per wave of 64 * cs sources
        assert (m["inliers"] == ref["inliers"]).all() and (m["ok"] == ref["ok"]).all()
        assert np.abs(m["T"] - ref["T"]).max() < 2e-6
    again = regs[0].batch_multi(qs, cand, params=prm)     # the first handle's order (cs = 2) is still intact
    assert (bits(again["T"]) == bits(ref["T"])).all()
    for r in regs:
        r.close()
    store.close()


def test_target_index_changes_no_bit_of_any_result(capi, scans):
    """gloc_scan_store_build_target_index re-sorts a scan's index into kd order: correspondences, distances,
    poses, inlier counts -- every output bit stays what it was (the search is exact, ties go to the smallest
    original index, moments are summed in the SOURCE's order); a scan can still be the source afterwards; ragged
    sizes (one sub-block, one point more than a power of two, an empty scan) re-sort without harm."""
    store = capi.ScanStore()
    A, B, Cc = scans["A"], scans["B"], scans["C"]
    rng = np.random.default_rng(3)
    clouds = [A, np.ascontiguousarray(Cc[::3]), np.ascontiguousarray(A[:16]), np.ascontiguousarray(A[:2049]),
              np.ascontiguousarray(B[5::7]), rng.uniform(-30, 30, (4097, 3)).astype(np.float32)]
    q = [store.add(B), store.add(np.ascontiguousarray(A[3::11]))]
    ids = [store.add(c) for c in clouds]
    cand = np.array([ids, ids[::-1]], np.uint32)
    prm = capi.default_reg_params(ransac_iters=300, icp_iters=6)
    r = capi.Registrar(store=store)
    before = r.batch_multi(q, cand, params=prm)
    corr_before = [r.debug_corr(j, len(B)) for j in range(len(ids))]
    for i in ids:
        store.build_target_index(i)
    store.build_target_index(ids[0])                      # idempotent
    after = r.batch_multi(q, cand, params=prm)
    for k in ("T", "rmse"):
        assert (bits(after[k]) == bits(before[k])).all(), k
    assert (after["inliers"] == before["inliers"]).all() and (after["ok"] == before["ok"]).all()
    for j, (ci, cd) in enumerate(corr_before):            # the last pass's correspondences, caller's index space
        ai, ad = r.debug_corr(j, len(B))
        assert (ai == ci).all() and (bits(ad) == bits(cd)).all(), j
    assert (store.download(ids[3]) == clouds[3]).all()    # the caller's copy of the points is untouched
    # a re-sorted scan as the SOURCE of a registration (its launch orders are rebuilt), and as both at once
    s2 = r.batch_multi([ids[0], q[0]], np.array([[q[0], ids[1]], [ids[0], ids[1]]], np.uint32), params=prm)
    fresh = capi.ScanStore()
    f_ids = [fresh.add(x) for x in (A, B, clouds[1])]
    r2 = capi.Registrar(store=fresh)
    s1 = r2.batch_multi([f_ids[0], f_ids[1]], np.array([[f_ids[1], f_ids[2]], [f_ids[0], f_ids[2]]], np.uint32), params=prm)
    # (the source's own order changed with its index: its moments are summed per wave of that order)
    assert (s2["inliers"] == s1["inliers"]).all() and (s2["ok"] == s1["ok"]).all() and np.abs(s2["T"] - s1["T"]).max() < 2e-6
    e = store.add(np.zeros((0, 3), np.float32))
    store.build_target_index(e)
    with pytest.raises(capi.GlocError):
        store.build_target_index(12345)
    r.close()
    r2.close()
    store.close()
    fresh.close()


def test_every_pass_bit_identical_to_the_brute_force_kernel(capi, scans):
    """After EVERY number of ICP passes the correspondences and distances of the last (warm-started) pass of the
    culled search equal those of the exhaustive kernel AT THE SAME POSE, bit for bit -- full-size scans, a positive
    and a different-scene candidate, with and without the RANSAC stage in front.  The pose the k-th pass searches at
    is the result of k - 1 passes (the run is deterministic); the exhaustive kernel searches there from cold.  (Until
    the culled path's moments became fp32-about-the-wave's-centre both modes were simply run side by side: their
    poses then agreed to the last bit of the fp32 result, now they are 1e-9 apart after the first step.)"""
    q, cands = scans["B"], [scans["A"], scans["C"]]
    store = capi.ScanStore()
    qid = store.add(q)
    cids = [store.add(c) for c in cands]

    def run(mode, iters, ransac, init_T=None):
        r = capi.Registrar(store=store)
        r.set_option(capi.REG_OPT_NN_MODE, mode)
        out = r.batch_ids(qid, cids, init_T=init_T, params=capi.default_reg_params(ransac_iters=ransac, icp_iters=iters))
        corr = [r.debug_corr(j, len(q)) for j in range(2)]
        r.close()
        return out, corr

    for iters, ransac in ((1, 0), (2, 0), (3, 0), (6, 0), (12, 0), (4, 300), (9, 300)):
        out, corr = run(capi.REG_NN_CULLED, iters, ransac)
        if iters == 1 and ransac == 0:
            T_prev = np.tile(np.eye(4, dtype=np.float32), (2, 1, 1))
        else:
            T_prev = run(capi.REG_NN_CULLED, iters - 1, ransac)[0]["T"]       # where the k-th pass searched
        ref_out, ref_corr = run(capi.REG_NN_EXHAUSTIVE, 1, 0, init_T=T_prev)
        for j in range(2):
            assert (corr[j][0] == ref_corr[j][0]).all(), (iters, ransac, j)
            assert (bits(corr[j][1]) == bits(ref_corr[j][1])).all(), (iters, ransac, j)
        # and the whole exhaustive run ends where the culled one does, to micrometres
        full_out, _ = run(capi.REG_NN_EXHAUSTIVE, iters, ransac)
        assert (out["inliers"] == full_out["inliers"]).all() and np.abs(out["T"] - full_out["T"]).max() < 1e-5
    store.close()


def test_split_groups_change_no_bit_of_any_result(capi, scans):
    """Heavy source groups searched by several waves (nn_compact.hpp, NnSplit: parts take disjoint candidate chunks,
    fold (distance, original index) keys in global memory, the last part to arrive writes the outputs): whatever the
    plan -- off, the default, every group split as far as the helper slots go, too few slots, other sources per
    lane -- poses, rmse, inlier counts and the last pass's correspondences and distances keep their bits.  Full-size
    scans, a positive and a different-scene candidate (the one whose groups are heavy), RANSAC + ICP."""
    store = capi.ScanStore()
    qid = store.add(scans["B"])
    cids = [store.add(scans["A"]), store.add(scans["C"])]
    store.build_target_index_batch(cids[:1])                    # one kd-ordered target, one in curve order
    prm = capi.default_reg_params(ransac_iters=300, icp_iters=5)

    def run(helpers, thresh, cs=2):
        r = capi.Registrar(store=store)
        r.set_option(capi.REG_OPT_NN_SRC_PER_LANE, cs)
        r.set_option(capi.REG_OPT_NN_SPLIT_HELPERS, helpers)
        r.set_option(capi.REG_OPT_NN_SPLIT_THRESH, thresh)
        out = r.batch_ids(qid, cids, params=prm)
        corr = [r.debug_corr(j, len(scans["B"])) for j in range(2)]
        r.close()
        return out, corr

    ref, ref_corr = run(0, 60000)
    for helpers, thresh, cs in ((-1, 60000, 2), (700, 1, 2), (37, 20000, 2), (8, 1, 2), (300, 1, 1), (300, 1000, 4)):
        out, corr = run(helpers, thresh, cs)
        what = (helpers, thresh, cs)
        if cs == 2:
            assert (bits(out["T"]) == bits(ref["T"])).all() and (bits(out["rmse"]) == bits(ref["rmse"])).all(), what
        else:                                                   # (other wave partials: the moments' rounding differs)
            assert np.abs(out["T"] - ref["T"]).max() < 1e-5, what
        assert (out["inliers"] == ref["inliers"]).all() and (out["ok"] == ref["ok"]).all(), what
        if cs == 2:
            for j in range(2):
                assert (corr[j][0] == ref_corr[j][0]).all() and (bits(corr[j][1]) == bits(ref_corr[j][1])).all(), what
    store.close()


def test_cold_waves_that_give_up_change_no_bit(capi, scans):
    """Round 6 (NnHeavy): a wave of a batch's first (cold) pass that has processed GLOC_REG_OPT_NN_HEAVY_THRESH target chunks
    appends its source group to a list and leaves; a second launch searches every listed group with 8 waves (candidate
    chunks dealt c mod 8, keys folded through device-scope atomics, the last part writes the outputs).  Whatever the
    threshold -- off, the default, 1 (every wave that processes anything gives up: the list overflows its 16 entries per job
    and the surplus goes on alone), 3 -- the pairs pass (RANSAC) and the moments pass (ICP only) and everything behind them
    keep their bits; gloc_reg_nn (one cold pass) equals the brute-force kernel.  Full-size scans, both target orders."""
    store = capi.ScanStore()
    qid = store.add(scans["B"])
    cids = [store.add(scans["A"]), store.add(scans["C"]), store.add(np.ascontiguousarray(scans["A"][::3]))]
    store.build_target_index_batch(cids[:1])                    # one kd-ordered target, two in curve order

    def run(thresh, ransac, icp, helpers=-1):
        r = capi.Registrar(store=store)
        r.set_option(capi.REG_OPT_NN_HEAVY_THRESH, thresh)
        r.set_option(capi.REG_OPT_NN_SPLIT_HELPERS, helpers)
        out = r.batch_ids(qid, cids, params=capi.default_reg_params(ransac_iters=ransac, icp_iters=icp))
        corr = [r.debug_corr(j, len(scans["B"])) for j in range(3)]
        r.close()
        return out, corr

    for ransac, icp in ((300, 0), (300, 3), (0, 1), (0, 4)):
        ref, ref_corr = run(0, ransac, icp)
        for thresh, helpers in ((32, -1), (1, -1), (3, -1), (1, 0), (5, 0)):
            out, corr = run(thresh, ransac, icp, helpers)
            what = (ransac, icp, thresh, helpers)
            assert (bits(out["T"]) == bits(ref["T"])).all() and (bits(out["rmse"]) == bits(ref["rmse"])).all(), what
            assert (out["inliers"] == ref["inliers"]).all() and (out["ok"] == ref["ok"]).all(), what
            for j in range(3):
                assert (corr[j][0] == ref_corr[j][0]).all() and (bits(corr[j][1]) == bits(ref_corr[j][1])).all(), what
    # one cold pass through the C ABI's 1-NN entry point: culled (giving up at once) == brute force, both target orders
    sub = np.ascontiguousarray(scans["B"][::2])
    ex = capi.Registrar()
    ex.set_option(capi.REG_OPT_NN_MODE, capi.REG_NN_EXHAUSTIVE)
    want = ex.nn(sub, scans["A"])
    ex.close()
    for kd in (0, 1):
        for thresh in (1, 2, 32):
            r = capi.Registrar()
            r.set_option(capi.REG_OPT_TEMP_TARGET_INDEX, kd)
            r.set_option(capi.REG_OPT_NN_HEAVY_THRESH, thresh)
            idx, d2 = r.nn(sub, scans["A"])
            assert (idx == want[0]).all() and (bits(d2) == bits(want[1])).all(), (kd, thresh)
            r.close()
    store.close()


def test_sub_batches_on_their_own_streams_change_no_bit(capi, scans):
    """GLOC_REG_OPT_SUB_BATCHES (round 6): a small batch cut into runs of jobs, each enqueued on its own internal stream
    (forked from and joined into the handle's): the same poses, rmse, inliers, ok and correspondences as one stream, bit
    for bit -- with and without RANSAC, with the split plan and the cold pass's second launch in play, on a caller's stream,
    twice in a row (the streams are reused), and through begin / end."""
    import torch
    store = capi.ScanStore()
    A, B, Cc = scans["A"], scans["B"], scans["C"]
    qid = store.add(np.ascontiguousarray(B[::4]))
    cids = [store.add(np.ascontiguousarray(x)) for x in (A[::4], A[1::5], Cc[::4], A[2::6], Cc[1::5], A[3::7], A[::9], Cc[::7], A[5::8])]
    store.build_target_index_batch(cids[:5])
    n = len(B[::4])

    def run(G, ransac, icp, stream=None, begin_end=False):
        r = capi.Registrar(store=store)
        r.set_option(capi.REG_OPT_SUB_BATCHES, G)
        r.set_option(capi.REG_OPT_NN_HEAVY_THRESH, 3)
        if stream is not None:
            r.set_stream(stream.cuda_stream)
        prm = capi.default_reg_params(ransac_iters=ransac, icp_iters=icp)
        outs = []
        for _ in range(2):
            if begin_end:
                r.batch_multi_begin([qid], np.array([cids], np.uint32), params=prm)
                o = r.batch_multi_end()
                outs.append({k: v[0] for k, v in o.items()})
            else:
                outs.append(r.batch_ids(qid, cids, params=prm))
        corr = [r.debug_corr(j, n) for j in range(len(cids))]
        r.close()
        return outs, corr

    for ransac, icp in ((200, 4), (0, 3), (200, 0)):
        (ref, ref2), ref_corr = run(1, ransac, icp)
        for G, stream, be in ((2, None, False), (4, None, False), (8, None, True), (-1, torch.cuda.Stream(), False), (3, torch.cuda.Stream(), True)):
            outs, corr = run(G, ransac, icp, stream, be)
            for out in outs:
                what = (ransac, icp, G, be)
                assert (bits(out["T"]) == bits(ref["T"])).all() and (bits(out["rmse"]) == bits(ref["rmse"])).all(), what
                assert (out["inliers"] == ref["inliers"]).all() and (out["ok"] == ref["ok"]).all(), what
            for j in range(len(cids)):
                assert (corr[j][0] == ref_corr[j][0]).all() and (bits(corr[j][1]) == bits(ref_corr[j][1])).all(), (ransac, icp, G, j)
    store.close()


def test_adaptive_stop_count_on_the_device_equals_the_oracles_loop(capi, oracle_mod):
    """The adaptive RANSAC stop's iteration count (OpenCV's rule: the smallest k with (1 - w^3)^k <= 1 - confidence, by
    repeated multiplication in the oracle) as the device computes it -- from the two logarithms where no rounding can
    change the answer, by the loop otherwise: equal for 200 000 random (inliers, points) pairs, for every inlier count of
    a few scan sizes, and for the pairs whose exact quotient lies closest to an integer; confidences 0.5 .. 0.9999, caps
    3000 and 2^20."""
    import ctypes
    L = oracle_mod.lib()
    L.oracle_ransac_needed_iters.restype = ctypes.c_uint32
    L.oracle_ransac_needed_iters.argtypes = [ctypes.c_uint32, ctypes.c_uint32, ctypes.c_float, ctypes.c_uint32]
    rng = np.random.default_rng(77)
    n = rng.integers(3, 200000, 200000).astype(np.uint32)
    inl = (rng.random(200000) ** 2 * (n + 1)).astype(np.uint32)
    inl = np.minimum(inl, n)
    # every inlier count of small scans, and of a full-size one the counts nearest an integer quotient
    extra_n, extra_i = [], []
    for m in (3, 10, 97, 1000, 4096):
        extra_n += [m] * (m + 1)
        extra_i += list(range(m + 1))
    m = 123675
    w = np.arange(1, m, dtype=np.float64) / m
    with np.errstate(divide="ignore"):
        ke = np.log(0.01) / np.log1p(-w ** 3)
    near = (np.argsort(np.abs(ke - np.rint(ke)))[:3000] + 1)[::-1]  # (the closest last)
    extra_n += [m] * len(near)
    extra_i += list(near)
    n = np.concatenate([n, np.array(extra_n, np.uint32)])
    inl = np.concatenate([inl, np.array(extra_i, np.uint32)])
    r = capi.Registrar()
    for conf, cap in ((0.99, 3000), (0.999, 3000), (0.5, 3000), (0.9999, 1 << 20), (0.99, 1 << 20), (0.99, 7)):
        got = r.debug_needed_iters(inl, n, conf, cap)
        sel = np.arange(len(n)) if cap <= 3000 else np.concatenate([np.arange(0, 200000, 130), np.arange(len(n) - 1000, len(n))])  # (the loop on the host: up to a million steps each)
        ref = np.array([L.oracle_ransac_needed_iters(int(inl[i]), int(n[i]), conf, cap) for i in sel], np.uint32)
        bad = np.nonzero(got[sel] != ref)[0]
        assert bad.size == 0, (conf, cap, [(int(inl[sel[i]]), int(n[sel[i]]), int(got[sel[i]]), int(ref[i])) for i in bad[:5]])
    r.close()


def test_passes_chained_in_one_launch_change_no_bit(capi, scans):
    """GLOC_REG_OPT_NN_CHAIN (round 6): the warm ICP passes of a small batch as ONE launch -- searches, reductions, solves
    and plans of every pass end to end, a wave waiting on the device for its own job's previous solve -- give the poses,
    rmse, inliers, ok, final steps and correspondences of the launch-by-launch pipeline, bit for bit: with and without
    RANSAC (the first ICP pass is then a cold launch of its own), with many groups split over helper waves (a low
    threshold), a job with a one-point target, on a caller's stream, twice in a row, through begin / end."""
    import torch
    store = capi.ScanStore()
    A, B, Cc = scans["A"], scans["B"], scans["C"]
    qid = store.add(np.ascontiguousarray(B[::4]))
    cids = [store.add(np.ascontiguousarray(x)) for x in (A[::4], A[1::5], Cc[::4], A[2::6], Cc[1::5], A[3::7], A[::9], Cc[::7], A[5::8], A[:1])]
    store.build_target_index_batch(cids[:5])
    n = len(B[::4])

    def run(chain, ransac, icp, thresh, stream=None, begin_end=False):
        r = capi.Registrar(store=store)
        r.set_option(capi.REG_OPT_NN_CHAIN, chain)
        r.set_option(capi.REG_OPT_NN_SPLIT_THRESH, thresh)
        if stream is not None:
            r.set_stream(stream.cuda_stream)
        prm = capi.default_reg_params(ransac_iters=ransac, icp_iters=icp)
        outs = []
        for _ in range(2):
            if begin_end:
                r.batch_multi_begin([qid], np.array([cids], np.uint32), params=prm)
                o = r.batch_multi_end()
                outs.append({k: v[0] for k, v in o.items()})
            else:
                outs.append(r.batch_ids(qid, cids, params=prm))
            outs[-1]["steps"] = np.array(r.final_steps(len(cids)))
        corr = [r.debug_corr(j, n) for j in range(len(cids))]
        launches, timeouts = r.debug_chain()
        r.close()
        return outs, corr, launches, timeouts

    for ransac, icp in ((200, 6), (0, 5), (200, 2), (0, 21), (200, 1)):
        for thresh in (60000, 25000):
            (ref, _), ref_corr, l0, _ = run(0, ransac, icp, thresh)
            assert l0 == 0
            for stream, be in ((None, False), (torch.cuda.Stream(), True)):
                outs, corr, launches, timeouts = run(1, ransac, icp, thresh, stream, be)
                what = (ransac, icp, thresh, be)
                assert timeouts == 0, what
                # (a chain needs two warm passes: with RANSAC every ICP pass is warm, without it the first is a cold launch)
                assert launches == (2 if icp - (0 if ransac else 1) >= 2 else 0), (what, launches)
                for out in outs:
                    assert (bits(out["T"]) == bits(ref["T"])).all() and (bits(out["rmse"]) == bits(ref["rmse"])).all(), what
                    assert (out["inliers"] == ref["inliers"]).all() and (out["ok"] == ref["ok"]).all(), what
                    assert (bits(out["steps"].astype(np.float32)) == bits(ref["steps"].astype(np.float32))).all(), what
                for j in range(len(cids)):
                    assert (corr[j][0] == ref_corr[j][0]).all() and (bits(corr[j][1]) == bits(ref_corr[j][1])).all(), (what, j)
    store.close()


def test_chained_batches_of_unequal_jobs_and_two_chains_at_once(capi, scans):
    """The chained launch with (1) jobs of DIFFERENT sizes in one batch -- three queries of 31 k, 12 k and 700 points (a job's
    waves beyond its own groups only count), a NO_SCAN cell, 47 jobs (the largest batch that chains) -- and (2) two handles on
    two streams driven by two host threads, each chaining batch after batch at the same time: the poses, rmse, inliers and
    ok of the launch-by-launch pipeline, no wait runs out."""
    import threading
    import torch
    store = capi.ScanStore()
    A, B, Cc = scans["A"], scans["B"], scans["C"]
    qids = [store.add(np.ascontiguousarray(x)) for x in (B[::4], B[1::10], B[:700])]
    cids = [store.add(np.ascontiguousarray(x)) for x in (A[::4], A[1::5], Cc[::4], A[2::6], Cc[1::5], A[3::7], A[::9], Cc[::7], A[5::8], A[4::11],
                                                         A[::13], Cc[::6], A[6::7], A[1::9], Cc[2::9], A[7::10])]
    store.build_target_index_batch(cids[:6])
    prm = capi.default_reg_params(ransac_iters=150, icp_iters=5)
    grid = np.array([cids[:15] + [capi.NO_SCAN], cids[1:16] + [cids[0]], cids[::-1][:15] + [cids[3]]], np.uint32)  # 47 jobs + an empty cell

    def run(chain, q, ids, stream=None, reps=1, shares=0):
        r = capi.Registrar(store=store)
        r.set_option(capi.REG_OPT_NN_CHAIN, chain)
        r.set_option(capi.REG_OPT_NN_SUB_JOBS, shares)
        if stream is not None:
            r.set_stream(stream.cuda_stream)
        outs = [r.batch_multi(q, ids, params=prm) for _ in range(reps)]
        st = r.debug_chain()
        r.close()
        return outs, st

    (ref,), _ = run(0, qids, grid)
    for shares in (0, 1, 2, 4):  # (GLOC_REG_OPT_NN_SUB_JOBS: interleaved shares of a job in the launch order; 0 = 8 in a small batch)
        (out,), (launches, timeouts) = run(1, qids, grid, shares=shares)
        assert (launches, timeouts) == (1, 0), shares
        assert (bits(out["T"]) == bits(ref["T"])).all() and (bits(out["rmse"]) == bits(ref["rmse"])).all(), shares
        assert (out["inliers"] == ref["inliers"]).all() and (out["ok"] == ref["ok"]).all(), shares

    # two chains at once
    (ref_a,), _ = run(0, qids[:1], grid[:1])
    (ref_b,), _ = run(0, qids[1:2], grid[1:2])
    res = {}

    def worker(name, q, ids):
        try:
            res[name] = run(1, q, ids, torch.cuda.Stream(), reps=12)
        except Exception as e:  # (reported below: an exception in a thread would otherwise pass silently)
            res[name] = e
    ta = threading.Thread(target=worker, args=("a", qids[:1], grid[:1]))
    tb = threading.Thread(target=worker, args=("b", qids[1:2], grid[1:2]))
    ta.start(); tb.start(); ta.join(); tb.join()
    for name, refx in (("a", ref_a), ("b", ref_b)):
        assert not isinstance(res[name], Exception), res[name]
        outs, (launches, timeouts) = res[name]
        assert (launches, timeouts) == (12, 0), (name, launches, timeouts)
        for o in outs:
            assert (bits(o["T"]) == bits(refx["T"])).all() and (bits(o["rmse"]) == bits(refx["rmse"])).all(), name
            assert (o["inliers"] == refx["inliers"]).all() and (o["ok"] == refx["ok"]).all(), name
    store.close()


def test_a_chained_launch_that_stalls_ends_by_itself(capi, scans):
    """Every wait inside the chained launch is bounded: with solvers made to wait for a wave that never comes
    (gloc_reg_debug_chain_stall) the launch ends by itself within its time limit, no pose of it is used -- the library runs the
    SAME batch again launch by launch before it returns (the caller gets the unchained bits, late) -- and the handle goes on
    launch by launch until the option is set again.  Through batch_ids, through begin / end, and with initial poses."""
    import time
    store = capi.ScanStore()
    A, B = scans["A"], scans["B"]
    qid = store.add(np.ascontiguousarray(B[::4]))
    cids = [store.add(np.ascontiguousarray(x)) for x in (A[::4], A[1::5], A[2::6])]
    prm = capi.default_reg_params(ransac_iters=100, icp_iters=4)
    from gloc3d_amd import synth
    init = np.stack([synth.se3(1.5 * i, (0.2 * i, -0.1, 0.0)) for i in range(3)]).astype(np.float32)
    r0 = capi.Registrar(store=store)
    r0.set_option(capi.REG_OPT_NN_CHAIN, 0)
    ref = r0.batch_ids(qid, cids, params=prm)
    ref_init = r0.batch_ids(qid, cids, init_T=init, params=prm)
    r0.close()

    def same(out, want):
        assert (bits(out["T"]) == bits(want["T"])).all() and (bits(out["rmse"]) == bits(want["rmse"])).all()
        assert (out["inliers"] == want["inliers"]).all() and (out["ok"] == want["ok"]).all()

    r = capi.Registrar(store=store)
    r.debug_chain_stall(True)
    t0 = time.time()
    same(r.batch_ids(qid, cids, init_T=init, params=prm), ref_init)  # (stalls, times out, is run again launch by launch)
    assert 1.0 < time.time() - t0 < 15.0
    assert r.debug_chain() == (1, 1)
    same(r.batch_ids(qid, cids, params=prm), ref)  # (the handle has stopped chaining)
    assert r.debug_chain() == (1, 1)
    r.set_option(capi.REG_OPT_NN_CHAIN, 1)  # (switched on again by hand, still stalling: through begin / end this time)
    r.batch_multi_begin([qid], np.array([cids], np.uint32), params=prm)
    o = r.batch_multi_end()
    same({k: v[0] for k, v in o.items()}, ref)
    assert r.debug_chain() == (2, 2)
    r.debug_chain_stall(False)
    r.set_option(capi.REG_OPT_NN_CHAIN, 1)
    same(r.batch_ids(qid, cids, params=prm), ref)
    assert r.debug_chain() == (3, 2)
    r.close()
    store.close()


def test_split_groups_decide_ties_by_the_original_index(capi, oracle_mod):
    """Equidistant targets in DIFFERENT parts of a split group: the parts' keys are (distance, original index), so the
    smallest original index wins as in the single wave.  Lattice targets in shuffled order, sources on cell centres /
    edges / faces; max_corr_dist rejects every pair, so the pose stays the identity and the second (warm, split) pass
    searches the same configuration the oracle does."""
    rng = np.random.default_rng(5)
    g = np.stack(np.meshgrid(np.arange(24), np.arange(24), np.arange(12), indexing="ij"), -1).reshape(-1, 3)
    tgt = (g[rng.permutation(len(g))] * 0.5).astype(np.float32)
    cells = g[(g[:, 0] < 23) & (g[:, 1] < 23) & (g[:, 2] < 11)].astype(np.float32) * 0.5
    src = np.concatenate([cells + np.float32(0.25), cells + np.array([0.25, 0, 0], np.float32),
                          cells + np.array([0.25, 0.25, 0], np.float32)]).astype(np.float32)   # (no exact hits: no pair survives the gate)
    src = src[rng.permutation(len(src))]
    oi, od = oracle_mod.nn3(src, tgt)
    for kd in (0, 1):
        r = capi.Registrar()
        r.set_option(capi.REG_OPT_TEMP_TARGET_INDEX, kd)
        r.set_option(capi.REG_OPT_NN_SPLIT_HELPERS, 600)
        r.set_option(capi.REG_OPT_NN_SPLIT_THRESH, 1)
        out = r.batch(src, [tgt], params=capi.default_reg_params(ransac_iters=0, icp_iters=3, max_corr_dist=1e-4))
        assert (out["T"][0] == np.eye(4, dtype=np.float32)).all()
        idx, d2 = r.debug_corr(0, len(src))
        assert (idx == oi).all() and (bits(d2) == bits(od)).all()
        assert np.isinf(r.final_steps(1)[0])     # an ICP that stops for want of correspondences has not converged
        r.close()


def test_begin_end_pipeline_on_a_shared_stream(capi, scans):
    """gloc_reg_batch_multi_begin / _end: two handles with their own workspaces on ONE stream, batch i + 1 enqueued
    before batch i's results are waited for (bench.py's registration pipeline).  Every batch equals the blocking call
    bit for bit; one batch per handle in flight; end without begin is an error."""
    import torch
    store = capi.ScanStore()
    A, B, Cc = scans["A"], scans["B"], scans["C"]
    qs = [store.add(np.ascontiguousarray(x)) for x in (B[::20], A[5::33], B[3::45], Cc[1::50])]
    cs = [store.add(np.ascontiguousarray(x)) for x in (A[::6], A[1::7], Cc[::6])]
    store.build_target_index_batch(cs)
    cand = np.array([cs, cs[::-1]], np.uint32)
    prm = capi.default_reg_params(ransac_iters=300, icp_iters=6)
    ref = capi.Registrar(store=store)
    want = [ref.batch_multi(qs[2 * b:2 * b + 2], cand, params=prm) for b in range(2)]
    stream = torch.cuda.Stream()
    h = [capi.Registrar(store=store), capi.Registrar(store=store)]
    for r in h:
        r.set_stream(stream.cuda_stream)
    for rep in range(3):                                   # a stream of batches: 0 1 0 1 0 1, always one ahead
        h[0].batch_multi_begin(qs[0:2], cand, params=prm)
        with pytest.raises(capi.GlocError):
            h[0].batch_multi_begin(qs[0:2], cand, params=prm)          # already one in flight on this handle
        h[1].batch_multi_begin(qs[2:4], cand, params=prm)              # queued behind it, before it is waited for
        got = [h[0].batch_multi_end(), h[1].batch_multi_end()]
        for g, w in zip(got, want):
            assert (bits(g["T"]) == bits(w["T"])).all() and (bits(g["rmse"]) == bits(w["rmse"])).all()
            assert (g["inliers"] == w["inliers"]).all() and (g["ok"] == w["ok"]).all()
    with pytest.raises(capi.GlocError):
        h[0].batch_multi_end()
    # every other entry point that runs jobs on the handle's workspaces is refused (GLOC_ERR_STATE) while a batch
    # is in flight, and the batch in flight is not disturbed by the refused calls
    h[0].batch_multi_begin(qs[0:2], cand, params=prm)
    small = np.ascontiguousarray(A[::50])
    for call in (lambda: h[0].batch(small, [small], params=prm),
                 lambda: h[0].batch_ids(qs[0], cs, params=prm),
                 lambda: h[0].first_success_multi(qs[0:2], cand, params=prm),
                 lambda: h[0].nn(small, small),
                 lambda: h[0].ransac_hypotheses(small, small, np.arange(len(small), dtype=np.uint32), 1, 0, 16, 0.6),
                 lambda: h[0].debug_corr(0, 16)):
        with pytest.raises(capi.GlocError) as ei:
            call()
        assert ei.value.code == 5, ei.value                # GLOC_ERR_STATE
    with pytest.raises(capi.GlocError) as ei:              # nor may a scan THE BATCH READS be re-sorted in place under it
        store.build_target_index_batch(qs[:1])             # (a query scan of the batch, still in curve order)
    assert ei.value.code == 5
    with pytest.raises(capi.GlocError) as ei:              # ... nor released: its memory would be handed to the next upload
        store.release(qs[0])
    assert ei.value.code == 5
    store.build_target_index_batch(cs[:1])                 # already in kd order: nothing to re-sort, no error (round 5)
    fresh = store.add(np.ascontiguousarray(A[2::40]))
    store.build_target_index_batch([fresh])                # a scan no batch in flight reads: re-sorted (add_keyframe's path)
    g = h[0].batch_multi_end()
    assert (bits(g["T"]) == bits(want[0]["T"])).all() and (g["inliers"] == want[0]["inliers"]).all()
    store.build_target_index_batch(qs[:1])                 # (fine again)
    for r in h + [ref]:
        r.close()
    store.close()


def test_views_and_pins_are_taken_in_one_step(capi, scans):
    """gloc_reg_batch_multi_begin took its by-value scan views first and pinned the scans afterwards, so a
    re-sort (add_keyframe's gloc_scan_store_build_target_index) or a release from another thread in between left the
    batch with a stale view.  Views and pins are now one step under the store's mutex: (1) a begin that fails on an
    unknown id pins nothing; (2) with another thread re-sorting the batch's candidates into kd order all the while, every
    batch equals the quiet run bit for bit (a re-sort either happens before the views are taken or is refused)."""
    import threading
    import time
    store = capi.ScanStore()
    A, B, Cc = scans["A"], scans["B"], scans["C"]
    q = store.add(np.ascontiguousarray(B[::20]))
    c0 = store.add(np.ascontiguousarray(A[::6]))
    prm = capi.default_reg_params(ransac_iters=200, icp_iters=4)
    r = capi.Registrar(store=store)
    with pytest.raises(capi.GlocError):
        r.batch_multi_begin([q], np.array([[c0, 123456]], np.uint32), params=prm)      # an id that does not exist
    store.release(c0)                                        # nothing was left pinned by the failed begin ...
    with pytest.raises(capi.GlocError):
        r.batch_multi_end()                                  # ... and no batch is in flight
    thin = [np.ascontiguousarray(x) for x in (A[::6], A[1::7], Cc[::6], A[2::8])]
    quiet_ids = [store.add(x) for x in thin]
    store.build_target_index_batch(quiet_ids)
    want = r.batch_multi([q], np.array([quiet_ids], np.uint32), params=prm)
    stop, refused, errors = threading.Event(), [0], []
    shared = {"ids": None}

    def resorter():
        while not stop.is_set():
            ids = shared["ids"]
            if ids is None:
                time.sleep(0.0002)
                continue
            for i in ids:
                try:
                    store.build_target_index_batch([i])
                except capi.GlocError as e:
                    if e.code == 5:
                        refused[0] += 1                      # pinned by the batch in flight: refused, as documented
                    elif e.code != 1:                        # (GLOC_ERR_INVALID: the main thread has released the id meanwhile)
                        errors.append(e)
    th = threading.Thread(target=resorter)
    th.start()
    try:
        for rep in range(12):
            ids = [store.add(x) for x in thin]               # fresh candidates in curve order ...
            shared["ids"] = ids                              # ... which the other thread starts re-sorting at once
            g = r.batch_multi([q], np.array([ids], np.uint32), params=prm)
            shared["ids"] = None
            # kd order or curve order, the registration's result is the same (test_target_index_*): bit for bit
            assert (bits(g["T"]) == bits(want["T"])).all() and (g["inliers"] == want["inliers"]).all(), rep
            for i in ids:
                while True:
                    try:
                        store.release(i)
                        break
                    except capi.GlocError as e:
                        assert e.code == 5
    finally:
        stop.set()
        th.join()
    assert not errors, errors
    r.close()
    store.close()


def test_first_success_equals_select_over_the_full_batch(capi, scans):
    """The reference's stop-at-first-success loop (global_localization.cpp:519-572) for several queries at once:
    the same rank and pose, bit for bit, as registering all candidates and selecting afterwards -- with fewer
    registrations run."""
    store = capi.ScanStore()
    A, B, Cc = scans["A"], scans["B"], scans["C"]
    qs = [store.add(np.ascontiguousarray(B[::20])), store.add(np.ascontiguousarray(A[5::33])),
          store.add(np.ascontiguousarray(Cc[3::25]))]
    a0, a1, c0 = (store.add(np.ascontiguousarray(x)) for x in (A[::6], A[1::7], Cc[::6]))
    cand = np.array([[c0, a0, a1], [c0, capi.NO_SCAN, a1], [a0, a1, capi.NO_SCAN]], np.uint32)
    prm = capi.default_reg_params(ransac_iters=300, icp_iters=6, max_rmse=1.0, max_final_step=0.0)   # (6 passes on thinned clouds: not converged)
    r = capi.Registrar(store=store)
    full = r.batch_multi(qs, cand, params=prm)
    fs = r.first_success_multi(qs, cand, params=prm)
    for qi in range(3):
        want = capi.reg_select_first_ok(full["ok"][qi].astype(np.int32))
        assert fs["rank"][qi] == want
        if want >= 0:
            assert (bits(fs["T"][qi]) == bits(full["T"][qi, want])).all()
            assert fs["inliers"][qi] == full["inliers"][qi, want] and bits(fs["rmse"][qi]) == bits(full["rmse"][qi, want])
        else:
            assert (fs["T"][qi] == np.eye(4)).all()
    assert list(fs["rank"]) == [1, 2, -1]            # a different scene first; a missing slot; a lost query
    assert fs["jobs_run"] == 2 + 2 + 2 < 7            # of the 7 real (query, candidate) pairs
    r.close()
    store.close()


def test_degenerate_inputs(reg, capi):
    prm = capi.default_reg_params(ransac_iters=50, icp_iters=2)
    two = np.array([[0, 0, 0], [1, 0, 0]], np.float32)
    g = reg.batch(two, [two], params=prm)             # < 3 points: identity, not ok
    assert np.allclose(g["T"][0], np.eye(4)) and not g["ok"][0]
    line = np.stack([np.linspace(0, 10, 200), np.zeros(200), np.zeros(200)], 1).astype(np.float32)
    g = reg.batch(line, [line], params=prm)           # collinear: every hypothesis degenerate
    assert g["inliers"][0] == 0 and not g["ok"][0] and np.isfinite(g["T"]).all()


def test_empty_scans(reg, capi):
    prm = capi.default_reg_params(ransac_iters=50, icp_iters=2)
    pts = np.random.default_rng(0).uniform(-5, 5, (300, 3)).astype(np.float32)
    empty = np.zeros((0, 3), np.float32)
    g = reg.batch(pts, [empty, pts], params=prm)          # an empty candidate next to a real one
    assert np.allclose(g["T"][0], np.eye(4)) and not g["ok"][0] and g["inliers"][0] == 0
    assert g["ok"][1] and np.abs(g["T"][1] - np.eye(4)).max() < 1e-5
    g = reg.batch(empty, [pts], params=prm)               # an empty query
    assert np.allclose(g["T"][0], np.eye(4)) and not g["ok"][0]
    idx, d2 = reg.nn(pts[:5], empty)
    assert (idx == np.iinfo(np.uint32).max).all()
