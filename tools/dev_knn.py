"""Developer probe: kNN parity + timing on the GPU box (not part of the test suite)."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import oracle
from gloc3d_amd import capi, synth

def run(N, D, Q, k, gen, algo, reps=5):
    db = (synth.descriptors_traj if gen == "traj" else synth.descriptors_iid)(2001, 0, N, D)
    if gen == "traj":
        q = synth.queries_near(2001, (np.arange(Q) * 67) % N, D)
    else:
        q = synth.descriptors_iid(2002, 0, Q, D)
    oi, od = oracle.knn_search(db, q, k, threads=8)
    ix = capi.KnnIndex(D)
    ix.set_option(capi.KNN_OPT_ALGO, algo)
    ix.set_option(capi.KNN_OPT_PROFILE, 1)
    ix.add(db)
    gi, gd = ix.search(q, k)
    ok_i = (gi == oi).all(); ok_d = (gd.view(np.uint32) == od.view(np.uint32)).all()
    ix.profile_reset()
    t = time.time()
    for _ in range(reps): ix.search(q, k)
    dt = (time.time() - t) / reps
    prof = {n: ix.profile(n) for n in ["dist_exact", "dist_mfma", "select", "rerank", "finalize", "norms"]}
    ps = " ".join(f"{n}={ms/max(c,1)*1e3:.1f}us x{c//reps}" for n, (ms, c) in prof.items() if c)
    print(f"N={N} D={D} Q={Q} k={k} {gen} algo={algo}: idx_eq={ok_i} d2_biteq={ok_d} host_call={dt*1e6:.0f}us | {ps} | {ix.stats()}", flush=True)
    if not ok_i:
        bad = np.argwhere(gi != oi)
        print("  first mismatches:", bad[:5], gi[bad[0][0]][:8], oi[bad[0][0]][:8], gd[bad[0][0]][:4], od[bad[0][0]][:4])
    ix.close()

if __name__ == "__main__":
    for algo in (1, 2):
        run(300, 64, 3, 5, "iid", algo)
        run(4541, 512, 1, 20, "traj", algo)
        run(4541, 512, 64, 20, "traj", algo)
        run(4541, 512, 64, 20, "iid", algo)
        run(10000, 4096, 64, 20, "iid", algo)
        run(10000, 4096, 64, 20, "traj", algo)
    run(4541, 4096, 1, 20, "traj", 0)
    run(10000, 510, 5, 20, "iid", 0)
    run(5000, 1024, 20, 20, "iid", 0)
    run(5000, 1024, 40, 20, "iid", 0)
