"""dev: host-side time of each call of one query alone (scan add, retrieval, registration, release), 30 queries."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import bench
from gloc3d_amd import capi, synth
traj, world_a, world_b = bench.headline_world(bench.N_PLACES_1GPU)
store = capi.ScanStore()
g = 2964
places = [g + d for d in (0, 1, -1, 2, -2, 3, -3, 4, -4, 5, -5, 6, -6, 7, -7, 8, -8, 9, -9, 10)]
row = [store.add_raycast(world_b if pl % bench.NEG_EVERY == 1 else world_a, [traj[pl]], np.array([bench.PLACE_SEED + pl], np.uint64))[0] for pl in places]
store.build_target_index_batch(row)
qid0 = store.add_raycast(world_a, [traj[g] @ bench.query_offset(0)], np.array([bench.QUERY_SEED], np.uint64))[0]
q_host = store.download(qid0)
index = capi.KnnIndex(bench.DIM, device=0)
index.add_synthetic(1, bench.DB_SEED, 0, 4541, row_stride=1)
qd = synth.queries_near(bench.DB_SEED, np.array([g]), bench.DIM)
reg = capi.Registrar(store=store)
prm = capi.default_reg_params(ransac_iters=3000, icp_iters=20, max_rmse=1.0)
rows = np.array([row], np.uint32)
ts = []
for i in range(30):
    t0 = time.perf_counter()
    sid = store.add(q_host)
    t1 = time.perf_counter()
    ci, _ = index.search(qd, 20)
    t2 = time.perf_counter()
    reg.batch_multi([sid], rows, params=prm)
    t3 = time.perf_counter()
    reg.scan_release(sid)
    t4 = time.perf_counter()
    ts.append((t1 - t0, t2 - t1, t3 - t2, t4 - t3, t4 - t0))
a = np.median(np.array(ts[5:]), 0) * 1e3
print("ms: scan add %.3f  retrieval %.3f  registration %.3f  release %.3f  total %.3f" % tuple(a))
