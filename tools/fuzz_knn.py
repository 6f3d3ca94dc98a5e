"""dev: random shapes, windows and data scales through the matrix-core kNN paths (algo 2: split-bf16 coarse pass, algo 3:
fp32 MFMA) against the exact path (algo 1, itself pinned to the oracle and the reference's goldens by tests/test_knn_gpu.py).
Every index and every distance bit must agree.  usage: fuzz_knn.py [cases] [seed]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from gloc3d_amd import capi, synth

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 100
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
bad = 0
t0 = time.time()
for c in range(cases):
    D = int(rng.choice([8, 16, 24, 64, 72, 128, 200, 256, 512, 1000, 1024, 2048, 4096]))
    N = int(rng.choice([70, 300, 1000, 5000, 16384, 16385, 20000, 33000, 70000]))
    if N * D > 2.5e8:
        N = int(2.5e8 // D)
    Q = int(rng.choice([9, 17, 33, 64, 65, 100, 200]))
    k = int(rng.choice([1, 5, 20, 33, 52]))
    kind = int(rng.integers(0, 4))
    db = synth.descriptors_iid(1000 + c, 0, N, D)
    q = synth.descriptors_iid(5000 + c, 0, Q, D)
    if kind == 1:      # clustered: rows near a few centres, queries near rows
        cen = synth.descriptors_iid(9000 + c, 0, 7, D)
        db = (cen[rng.integers(0, 7, N)] + np.float32(0.05) * db).astype(np.float32)
        q = (db[rng.integers(0, N, Q)] + np.float32(0.02) * q).astype(np.float32)
    elif kind == 2:    # large offset: norms far above distances
        db = (db + np.float32(30.0)).astype(np.float32)
        q = (q + np.float32(30.0)).astype(np.float32)
    elif kind == 3:    # tiny scale
        db = (db * np.float32(1e-6)).astype(np.float32)
        q = (q * np.float32(1e-6)).astype(np.float32)
    dup = rng.integers(0, N, 6)
    db[dup[:3]] = db[dup[3:]]                                  # duplicated rows: ties by row index
    first = int(rng.integers(0, max(1, N // 3))) if rng.random() < 0.5 else 0
    last = N - int(rng.integers(0, max(1, N // 5))) if rng.random() < 0.5 else N
    res = {}
    for algo in (1, 2, 3):
        ix = capi.KnnIndex(D)
        ix.set_option(capi.KNN_OPT_ALGO, algo)
        ix.add(db)
        res[algo] = ix.search(q, k, first, last) + (ix.stats()["queries_fallback"],)
        ix.close()
    for algo in (2, 3):
        same = (res[algo][0] == res[1][0]).all() and (res[algo][1].view(np.uint32) == res[1][1].view(np.uint32)).all()
        if not same:
            bad += 1
            print(f"MISMATCH case {c}: algo {algo} N {N} D {D} Q {Q} k {k} kind {kind} window [{first}, {last})", flush=True)
    if c % 10 == 9:
        print(f"{c + 1} cases, {bad} mismatches, {time.time() - t0:.0f} s; last: N {N} D {D} Q {Q} k {k} kind {kind} fallbacks {res[2][2]}/{res[3][2]}", flush=True)
print("mismatches:", bad)
sys.exit(1 if bad else 0)
