"""CPU: the registration oracle (oracle/reg_oracle.c) against an independent numpy / scipy statement of the same
specification (tests/golden/make_crosscheck.py: cKDTree + numpy.linalg.svd, written from SURVEY.md Appendix B).
The reference delegates this arithmetic to PCL / OpenCV, so nothing upstream can pin it; two implementations on
different libraries agreeing on hypotheses, inlier counts and poses is what can be had against a misreading of the
specification.  (Not bit-exact by nature: LAPACK's SVD and the oracle's Jacobi sweep round differently; a pose
differs by ~1e-6, an inlier count by a point at the threshold.)"""
import importlib.util
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
_spec = importlib.util.spec_from_file_location("make_crosscheck", os.path.join(HERE, "golden", "make_crosscheck.py"))
mc = importlib.util.module_from_spec(_spec)
_spec.loader.exec_module(mc)


def _rot_angle(Ra, Rb):
    E = Ra.astype(np.float64).T @ Rb.astype(np.float64)
    v = 0.5 * np.array([E[2, 1] - E[1, 2], E[0, 2] - E[2, 0], E[1, 0] - E[0, 1]])
    return float(np.arctan2(np.linalg.norm(v), (np.trace(E) - 1) / 2))


def test_oracle_agrees_with_the_independent_statement(oracle_mod):
    g = np.load(os.path.join(HERE, "golden", "reg_crosscheck.npz"))
    for name, s, t, kw in mc.cases():
        assert (g[name + "_crc"] == np.array([mc.crc(s), mc.crc(t)], np.uint64)).all(), "inputs are regenerated, not stored"
        o = oracle_mod.reg_one(s, t, init_T=kw.get("init_T"), cand_id=kw["cand_id"], ransac_iters=kw["ransac_iters"],
                               icp_iters=kw["icp_iters"], ransac_confidence=kw.get("confidence", 0.99),
                               max_rmse=kw.get("max_rmse", 0.0), max_final_step=0.03)   # (the fixture was made with the check at 0.03)
        T, (rmse, inl, hyp, ok, fstep) = g[name + "_T"], g[name + "_meta"]
        assert abs(o["final_step"] - fstep) < 2e-4 * max(1.0, fstep / 0.03), name   # the convergence measure max_final_step is compared with
        assert np.abs(o["T"][:3, 3] - T[:3, 3]).max() < 1e-4, name        # north_star's tolerance: 1e-4 m / 1e-4 rad
        assert _rot_angle(o["T"][:3, :3], T[:3, :3]) < 1e-4, name
        assert abs(o["rmse"] - rmse) < 1e-4, name
        if abs(fstep - 0.03) > 1e-3:                                       # (at the threshold itself the two may round apart)
            assert o["ok"] == bool(ok), name
        assert abs(int(o["inliers"]) - int(inl)) <= 2, name                # (a point exactly at the 0.6 m threshold)
        assert (o["best_hyp"] if o["best_hyp"] != 0xFFFFFFFF else -1) == int(hyp), name


def test_the_fixture_is_what_the_generator_makes():
    """The committed vectors come from the committed script (one case re-run: ~0.5 s)."""
    g = np.load(os.path.join(HERE, "golden", "reg_crosscheck.npz"))
    name, s, t, kw = mc.cases()[3]
    r = mc.register(s, t, **kw)
    assert np.abs(r["T"] - g[name + "_T"]).max() < 1e-6 and r["inliers"] == int(g[name + "_meta"][1])


def _check(o, g, name):
    T, (rmse, inl, hyp, ok, fstep) = g[name + "_T"], g[name + "_meta"]
    assert np.abs(np.asarray(o["T"])[:3, 3] - T[:3, 3]).max() < 1e-4, name
    assert _rot_angle(np.asarray(o["T"])[:3, :3], T[:3, :3]) < 1e-4, name
    assert abs(float(o["rmse"]) - rmse) < 1e-4 and bool(o["ok"]) == bool(ok), name
    assert abs(int(o["inliers"]) - int(inl)) <= max(2, int(1e-4 * inl)), name   # (points exactly at the 0.6 m threshold)


def test_oracle_agrees_at_full_size(oracle_mod):
    """BASELINE configs[2]'s shape (124 k x 124 k points, RANSAC 3000 adaptive + ICP 20): the C restatement against
    the scipy / numpy statement's committed result (tests/golden/reg_crosscheck.npz: full_size).
    The same fixture checks the HIP path in tests/test_reg_gpu.py::test_full_size_matches_the_independent_statement."""
    g = np.load(os.path.join(HERE, "golden", "reg_crosscheck.npz"))
    name, s, t, kw = mc.full_size_case()
    assert (g[name + "_crc"] == np.array([mc.crc(s), mc.crc(t)], np.uint64)).all(), "inputs are regenerated, not stored"
    o = oracle_mod.reg_one(s, t, cand_id=kw["cand_id"], ransac_iters=kw["ransac_iters"], icp_iters=kw["icp_iters"],
                           max_rmse=kw["max_rmse"], max_final_step=0.03)
    _check(o, g, name)
    assert (o["best_hyp"] if o["best_hyp"] != 0xFFFFFFFF else -1) == int(g[name + "_meta"][2])
