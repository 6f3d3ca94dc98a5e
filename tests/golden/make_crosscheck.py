"""An INDEPENDENT statement of the registration semantics (S1 exact 1-NN, S2 RANSAC 3-point
Kabsch + inlier count + adaptive stop + refit, S3 point-to-point ICP) in numpy / scipy -- scipy.spatial.cKDTree for
the nearest neighbours, numpy.linalg.svd for Kabsch -- written from the specification, not from oracle/reg_oracle.c,
and the fixtures it produces (tests/golden/reg_crosscheck.npz).

Why: the reference delegates this arithmetic to PCL / OpenCV (absent, no fixtures upstream), so the C restatement
(the oracle) and the HIP kernels are one author's reading of Appendix B.  This does not PIN the oracle by the rules
(it is not the reference), but a second implementation on different libraries that lands on the same poses removes
the single-author risk.  tests/test_oracle_crosscheck.py compares the oracle with the fixture.

Shared with the oracle by construction (they are part of the specification, not of the algorithm): the counter RNG
of gloc3d_amd/synth.py that draws the three sample ids of hypothesis h of candidate c, and the inputs.

    python tests/golden/make_crosscheck.py          # rewrites tests/golden/reg_crosscheck.npz (six small cases in
                                                    # seconds, the full-size one in about half a minute)
"""
import os
import sys

import numpy as np
from scipy.spatial import cKDTree

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from gloc3d_amd import synth  # noqa: E402

F = np.float32


def sample_ids(seed, cand, hyp, n):
    """Three distinct ids in [0, n): draws of the counter RNG keyed by (seed, candidate, hypothesis), each mapped to
    floor(u * n / 2^64); the second and third are redrawn (up to 16 times) while they collide."""
    key = synth.rng_key(seed, (np.uint64(cand) << np.uint64(32)) | np.uint64(hyp))
    ctr = 0

    def draw():
        nonlocal ctr
        u = int(synth.rng_draw(key, np.uint64(ctr)))
        ctr += 1
        return (u * n) >> 64
    s0 = draw()
    s1 = s2 = s0
    tries = 0
    while tries < 16 and s1 == s0:
        s1 = draw()
        tries += 1
    tries = 0
    while tries < 16 and (s2 == s0 or s2 == s1):
        s2 = draw()
        tries += 1
    return s0, s1, s2


def kabsch(P, Q):
    """Least-squares rigid (R, t) with Q ~ R P + t (no scale), reflection fixed: R = V diag(1, 1, det) U^T."""
    pb, qb = P.mean(0), Q.mean(0)
    M = (P - pb).T @ (Q - qb)
    U, _, Vt = np.linalg.svd(M)
    V = Vt.T
    d = np.sign(np.linalg.det(V @ U.T))
    R = V @ np.diag([1.0, 1.0, d]) @ U.T
    return R, qb - R @ pb


def move_f32(R, t, X):
    """((r0 x + r1 y) + r2 z) + t in fp32, one rounding per operation (numpy does not fuse)."""
    R, t = R.astype(F), t.astype(F)
    x, y, z = X[:, 0], X[:, 1], X[:, 2]
    return np.stack([((R[a, 0] * x + R[a, 1] * y) + R[a, 2] * z) + t[a] for a in range(3)], 1)


def d2_f32(A, B):
    d = A - B
    return (d[:, 0] * d[:, 0] + d[:, 1] * d[:, 1]) + d[:, 2] * d[:, 2]


def needed_iters(inl, n, conf, cap):
    """Smallest k with (1 - w^3)^k <= 1 - conf, w = inl / n, capped."""
    w = inl / n
    q, target, pw, k = 1.0 - w * w * w, 1.0 - float(F(conf)), 1.0, 0
    while pw > target and k < cap:
        pw *= q
        k += 1
    return k


def register(src, tgt, cand_id=0, ransac_iters=3000, inlier_thresh=0.6, min_inlier_ratio=0.3, icp_iters=30, seed=1234,
             confidence=0.99, max_rmse=0.0, init_T=None, max_final_step=0.03):
    src, tgt = np.ascontiguousarray(src, F), np.ascontiguousarray(tgt, F)
    n = len(src)
    tree = cKDTree(tgt.astype(np.float64))           # float32 coordinates are exact in float64
    T = np.eye(4) if init_T is None else np.asarray(init_T, np.float64)
    thr2 = F(inlier_thresh) * F(inlier_thresh)

    def match(T):
        moved = move_f32(T[:3, :3], T[:3, 3], src)
        _, j = tree.query(moved.astype(np.float64))
        return moved, j, d2_f32(moved, tgt[j])       # the distance as the fp32 pipeline sees it

    best_inl, best_h, best = 0, None, None
    ok = False
    moved, j, d2 = match(T)
    if ransac_iters and n >= 3:
        q = tgt[j]
        niters = ransac_iters
        h = 0
        adaptive = 0.0 < confidence < 1.0
        while h < niters:
            s = sample_ids(seed, cand_id, h, n)
            h += 1
            if len(set(s)) < 3:
                continue
            P, Q = moved[list(s)].astype(np.float64), q[list(s)].astype(np.float64)
            a, b = P[1] - P[0], P[2] - P[0]
            c = np.cross(a, b)
            aa, bb, cc = a @ a, b @ b, c @ c
            if not (aa > 1e-12) or not (bb > 1e-12) or not (cc > 1e-6 * (aa * bb)):
                continue                              # near-collinear sample
            R, t = kabsch(P, Q)
            inl = int(np.count_nonzero(d2_f32(move_f32(R, t, moved), q) < thr2))
            if inl > best_inl:                        # first strictly better hypothesis wins: ties -> smallest h
                best_inl, best_h, best = inl, h - 1, (R.astype(F), t.astype(F))
                if adaptive:
                    niters = min(niters, needed_iters(inl, n, confidence, ransac_iters))
        ok = best_inl >= max(int(F(min_inlier_ratio) * F(n)), 3)
        if best is not None:
            R, t = best
            keep = d2_f32(move_f32(R, t, moved), q) < thr2
            if keep.sum() >= 3:                       # refit on the winner's inliers, moved -> target
                R, t = kabsch(moved[keep].astype(np.float64), q[keep].astype(np.float64))
            Tr = np.eye(4)
            Tr[:3, :3], Tr[:3, 3] = R, t
            T = Tr @ T
    final_step = 0.0
    for _ in range(icp_iters):
        moved, j, d2 = match(T)
        P = moved.astype(np.float64)
        R, t = kabsch(P, tgt[j].astype(np.float64))
        # convergence measure of the specification: RMS displacement of the matched points by this update, as
        # |R c + t - c|^2 + |R - I|_F^2 / 2 * tr cov(P)
        c = P.mean(0)
        final_step = float(np.sqrt(np.sum((R @ c + t - c) ** 2) + 0.5 * np.sum((R - np.eye(3)) ** 2) * ((P - c) ** 2).sum(1).mean()))
        Td = np.eye(4)
        Td[:3, :3], Td[:3, 3] = R, t
        T = Td @ T
    rmse = float(np.sqrt(d2.astype(np.float64).sum() / n)) if n else 0.0
    if max_rmse > 0 and not rmse <= max_rmse:
        ok = False
    if max_final_step > 0 and icp_iters > 0 and not final_step <= max_final_step:
        ok = False                                    # the ICP has not converged
    return dict(T=T.astype(F), rmse=rmse, inliers=best_inl, best_hyp=-1 if best_h is None else best_h, ok=bool(ok),
                final_step=final_step)


def cases():
    """(name, source, target, kwargs): a same-place pair, a different place, an initial guess, RANSAC only, ICP only,
    all 500 hypotheses scored."""
    w = synth.make_world(1001)
    A = synth.lidar_scan(w, None, seed=1001, n_az=500)[:, :3]
    B = synth.lidar_scan(w, synth.se3(4.0, (0.5, -0.3, 0.1)), seed=1002, n_az=500)[:, :3]
    C = synth.lidar_scan(synth.make_world(77), None, seed=5, n_az=500)[:, :3]
    q, a, c = np.ascontiguousarray(B[::3]), np.ascontiguousarray(A[::2]), np.ascontiguousarray(C[::2])
    T0 = synth.se3(3.0, (0.4, -0.2, 0.0)).astype(F)
    return [("same_place", q, a, dict(cand_id=0, ransac_iters=500, icp_iters=12, max_rmse=1.0)),
            ("other_place", q, c, dict(cand_id=1, ransac_iters=500, icp_iters=12, max_rmse=1.0)),
            ("init_guess", q, a, dict(cand_id=2, ransac_iters=300, icp_iters=6, init_T=T0)),
            ("ransac_only", q, a, dict(cand_id=3, ransac_iters=400, icp_iters=0)),
            ("icp_only", q, a, dict(cand_id=4, ransac_iters=0, icp_iters=15)),
            ("all_hypotheses", q, a, dict(cand_id=5, ransac_iters=500, icp_iters=4, confidence=0.0))]


def full_size_case():
    """BASELINE configs[2]'s shape: ~124 k x ~124 k points, RANSAC (3000 cap, adaptive) + ICP 20 -- the two full scans of
    tests/test_reg_gpu.py's fixture (a same-place pair 5 degrees / 0.6 m apart).  ~25 s of cKDTree queries here, so it
    is a case of its own: the oracle (CPU suite) and the HIP path (-m gpu) are compared with the committed fixture."""
    w = synth.make_world(1001)
    A = synth.lidar_scan(w, None, seed=1001)[:, :3]
    B = synth.lidar_scan(w, synth.se3(5.0, (0.5, -0.3, 0.1)), seed=1002)[:, :3]
    return ("full_size", np.ascontiguousarray(B), np.ascontiguousarray(A),
            dict(cand_id=0, ransac_iters=3000, icp_iters=20, max_rmse=1.0))


def crc(a):
    return np.uint64(np.ascontiguousarray(a, F).view(np.uint32).sum(dtype=np.uint64))


if __name__ == "__main__":
    out = {}
    for name, s, t, kw in cases() + [full_size_case()]:
        r = register(s, t, **kw)
        print(name, "inliers", r["inliers"], "hyp", r["best_hyp"], "rmse", round(r["rmse"], 4), "ok", r["ok"], "final step", round(r["final_step"], 5))
        out[name + "_T"] = r["T"]
        out[name + "_meta"] = np.array([r["rmse"], r["inliers"], r["best_hyp"], float(r["ok"]), r["final_step"]], np.float64)
        out[name + "_crc"] = np.array([crc(s), crc(t)], np.uint64)
    np.savez(os.path.join(HERE, "reg_crosscheck.npz"), **out)
