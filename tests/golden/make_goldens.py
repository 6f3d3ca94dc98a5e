"""Generate the committed golden fixtures from the REFERENCE's own vendored nanoflann.

Run in the build container (needs /root/reference):  python tests/golden/make_goldens.py
The reference headers are compiled where they lie by oracle/Makefile (`make ref`) into
oracle/_ref/libgloc_ref.so; this script only calls that library and stores DATA (inputs are
regenerated from gloc3d_amd.synth seeds; outputs = indices + distance bit patterns).
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import oracle  # noqa: E402
from gloc3d_amd import synth  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))

# name -> (N, D, Q, k, generator, db seed, query seed)
KNN_CASES = {
    "knn_kitti00_d512_q1_traj": (4541, 512, 1, 20, "traj", 4001, 4001),
    "knn_kitti00_d512_q64_traj": (4541, 512, 64, 20, "traj", 4001, 4001),
    "knn_kitti00_d512_q64_iid": (4541, 512, 64, 20, "iid", 2001, 2002),
    "knn_cfgB_d4096_q64_iid": (10000, 4096, 64, 20, "iid", 2001, 2002),
    "knn_cfgB_d4096_q64_traj": (10000, 4096, 64, 20, "traj", 2003, 2003),
    "knn_small_d64_q5": (300, 64, 5, 7, "iid", 11, 12),
    # BASELINE.json configs[3]'s own shape (KITTI-00-sized database x 4096-D), one query and a batch
    "knn_kitti00_d4096_q1_traj": (4541, 4096, 1, 20, "traj", 4001, 4001),
    "knn_kitti00_d4096_q64_traj": (4541, 4096, 64, 20, "traj", 4001, 4001),
}


def knn_inputs(case):
    N, D, Q, k, gen, s_db, s_q = KNN_CASES[case]
    if gen == "traj":
        db = synth.descriptors_traj(s_db, 0, N, D)
        q = synth.queries_near(s_q, (np.arange(Q, dtype=np.uint64) * 67 + 5) % N, D)
    else:
        db = synth.descriptors_iid(s_db, 0, N, D)
        q = synth.descriptors_iid(s_q, 0, Q, D)
    return db, q, k


def nn3_inputs():
    w = synth.make_world(1001)
    A = synth.lidar_scan(w, None, seed=1001)[:, :3]
    B = synth.lidar_scan(w, synth.se3(5.0, (0.5, -0.3, 0.1)), seed=1002)[:, :3]
    return np.ascontiguousarray(B[::60]), np.ascontiguousarray(A[::20])


def nn3_fullsize_inputs():
    """Two full-size scans (~124k points each) and an initial guess.  Coordinates are snapped to
    multiples of 1/1024 m so that the regenerated inputs are bit-identical on every machine (the
    ray-caster uses libm sin/cos, whose last bits may differ between CPUs)."""
    w = synth.make_world(1001)
    snap = lambda a: np.ascontiguousarray(np.round(a.astype(np.float64) * 1024.0) / 1024.0, np.float32)
    A = snap(synth.lidar_scan(w, None, seed=1001)[:, :3])
    B = snap(synth.lidar_scan(w, synth.se3(5.0, (0.5, -0.3, 0.1)), seed=1002)[:, :3])
    T = synth.se3(4.5, (0.4, -0.25, 0.08)).astype(np.float32)   # near the true pose: a realistic ICP state
    return B, A, T


def crc(a):
    return np.uint64(int(np.ascontiguousarray(a, np.float32).view(np.uint32).sum(dtype=np.uint64)))


def main():
    oracle.build(ref=True)
    for case in KNN_CASES:
        db, q, k = knn_inputs(case)
        idx, d2 = oracle.ref_knn_search(db, q, k)
        # the fixture must be free of exact ties inside the top-k (tie order is tree dependent)
        assert all(len(np.unique(r)) == len(r) for r in d2), f"{case}: tie in top-k"
        np.savez_compressed(os.path.join(HERE, case + ".npz"), idx=idx, d2_bits=d2.view(np.uint32),
                            q_crc=np.uint64(int(q.view(np.uint32).sum(dtype=np.uint64))),
                            db_crc=np.uint64(int(db.view(np.uint32).sum(dtype=np.uint64))))
        print(case, idx.shape, "d2[0,:3] =", d2[0, :3])
    src, tgt = nn3_inputs()
    idx, d2 = oracle.ref_nn3(src, tgt)
    np.savez_compressed(os.path.join(HERE, "nn3_scanpair.npz"), src=src, tgt=tgt, idx=idx,
                        d2_bits=d2.view(np.uint32))
    print("nn3_scanpair", src.shape, tgt.shape)
    # full size: the moved source (fixed-order fp32 transform) against the reference's kd-tree
    src, tgt, T = nn3_fullsize_inputs()
    moved = oracle.transform_points(T, src)
    idx, d2 = oracle.ref_nn3(moved, tgt)
    gi, gd = oracle.nn3(moved, tgt, grid=True)   # smallest-index tie rule: the fixture must be tie-free
    assert (gi == idx).all() and (gd.view(np.uint32) == d2.view(np.uint32)).all(), "tie in the full-size fixture"
    np.savez_compressed(os.path.join(HERE, "nn3_fullsize.npz"), idx=idx, d2_bits=d2.view(np.uint32),
                        src_crc=crc(src), tgt_crc=crc(tgt), T=T)
    print("nn3_fullsize", src.shape, tgt.shape, os.path.getsize(os.path.join(HERE, "nn3_fullsize.npz")))


if __name__ == "__main__":
    main()
