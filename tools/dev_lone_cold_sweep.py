"""dev: one query alone (registration of 20 jobs, chained) under the cold pass's give-up threshold, same box; 5 queries.
usage: dev_lone_cold_sweep.py"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import bench
from gloc3d_amd import capi
traj, world_a, world_b = bench.headline_world(bench.N_PLACES_1GPU)
store = capi.ScanStore()
rows, qids = [], []
for q, g in enumerate((300, 1188, 2076, 2964, 3852)):
    qids.append(store.add_raycast(world_a, [traj[g] @ bench.query_offset(q)], np.array([bench.QUERY_SEED + q], np.uint64))[0])
    places = [g + d for d in (0, 1, -1, 2, -2, 3, -3, 4, -4, 5, -5, 6, -6, 7, -7, 8, -8, 9, -9, 10)]
    row = [store.add_raycast(world_b if pl % bench.NEG_EVERY == 1 else world_a, [traj[pl]], np.array([bench.PLACE_SEED + pl], np.uint64))[0] for pl in places]
    store.build_target_index_batch(row)
    rows.append(row)

def run(heavy, icp=20, reps=6, opts=()):
    reg = capi.Registrar(store=store)
    reg.set_option(capi.REG_OPT_NN_HEAVY_THRESH, heavy)
    for o, v in opts:
        reg.set_option(o, v)
    prm = capi.default_reg_params(ransac_iters=3000, icp_iters=icp, max_rmse=1.0)
    ts = [[] for _ in qids]
    for _ in range(reps):
        for i, (qid, row) in enumerate(zip(qids, rows)):
            t0 = time.time()
            reg.batch_multi([qid], [row], params=prm)
            ts[i].append(time.time() - t0)
    reg.close()
    return np.array([np.median(t[2:]) for t in ts]) * 1e3

for heavy in (32, 32, 32):
    a = run(heavy)
    b = run(heavy, icp=0)
    print("give-up at %2d chunks: per query %s  mean %.3f ms;  without ICP (cold pass + RANSAC) %s mean %.3f" % (heavy, np.round(a, 2), a.mean(), np.round(b, 2), b.mean()))
