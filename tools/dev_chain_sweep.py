"""dev: one query alone (registration of 20 jobs) with the chained launch under split thresholds / helper slots, same box.
usage: dev_chain_sweep.py"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import bench
from gloc3d_amd import capi
traj, world_a, world_b = bench.headline_world(bench.N_PLACES_1GPU)
store = capi.ScanStore()
rows, qids = [], []
for q, g in enumerate((300, 1500, 2900)):
    qids.append(store.add_raycast(world_a, [traj[g] @ bench.query_offset(q)], np.array([bench.QUERY_SEED + q], np.uint64))[0])
    places = [g + d for d in (0, 1, -1, 2, -2, 3, -3, 4, -4, 5, -5, 6, -6, 7, -7, 8, -8, 9, -9, 10)]
    row = [store.add_raycast(world_b if pl % bench.NEG_EVERY == 1 else world_a, [traj[pl]], np.array([bench.PLACE_SEED + pl], np.uint64))[0] for pl in places]
    store.build_target_index_batch(row)
    rows.append(row)
prm = capi.default_reg_params(ransac_iters=3000, icp_iters=20, max_rmse=1.0)

def run(chain, thresh, helpers, sub=0, reps=7, jg=0):
    reg = capi.Registrar(store=store)
    reg.set_option(capi.REG_OPT_NN_CHAIN, chain)
    reg.set_option(capi.REG_OPT_NN_SPLIT_THRESH, thresh)
    reg.set_option(capi.REG_OPT_NN_SPLIT_HELPERS, helpers)
    if sub:
        reg.set_option(capi.REG_OPT_NN_SUB_JOBS, sub)
    if jg:
        reg.set_option(capi.REG_OPT_NN_JOB_GROUP, jg)
    ts = []
    for _ in range(reps):
        for qid, row in zip(qids, rows):
            t0 = time.time()
            reg.batch_multi([qid], [row], params=prm)
            ts.append(time.time() - t0)
    n, t = reg.debug_chain()
    reg.close()
    return float(np.median(ts[3:])) * 1e3, n

# 16 jobs (the first 16 candidates): whole jobs per XCD against halves, both balanced over the 8 XCDs
rows16 = [r[:16] for r in rows]
def run16(sub, jg, reps=9):
    reg = capi.Registrar(store=store)
    reg.set_option(capi.REG_OPT_NN_SPLIT_THRESH, 85000)
    reg.set_option(capi.REG_OPT_NN_SUB_JOBS, sub)
    reg.set_option(capi.REG_OPT_NN_JOB_GROUP, jg)
    ts = []
    for _ in range(reps):
        for qid, row in zip(qids, rows16):
            t0 = time.time()
            reg.batch_multi([qid], [row], params=prm)
            ts.append(time.time() - t0)
    reg.close()
    return float(np.median(ts[3:])) * 1e3
for rep in range(2):
    for jg, sub in ((8, 1), (8, 2), (8, 4), (16, 2), (24, 8)):
        print("chain, 16 jobs: slots per group %2d, shares of a job %d: %.3f ms" % (jg, sub, run16(sub, jg)))
