"""dev: one query alone as bench.py's sub-record runs it (scan upload + index, registration, release), 10 times -- run under
`rocprofv3 --kernel-trace` by tools/dev_lone_prep_timeline.sh, which prints the last repetition's operations before the cold pass."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import bench
from gloc3d_amd import capi
traj, world_a, world_b = bench.headline_world(bench.N_PLACES_1GPU)
store = capi.ScanStore()
g = 2964
places = [g + d for d in (0, 1, -1, 2, -2, 3, -3, 4, -4, 5, -5, 6, -6, 7, -7, 8, -8, 9, -9, 10)]
row = [store.add_raycast(world_b if pl % bench.NEG_EVERY == 1 else world_a, [traj[pl]], np.array([bench.PLACE_SEED + pl], np.uint64))[0] for pl in places]
store.build_target_index_batch(row)
qid0 = store.add_raycast(world_a, [traj[g] @ bench.query_offset(0)], np.array([bench.QUERY_SEED], np.uint64))[0]
q_host = store.download(qid0)
reg = capi.Registrar(store=store)
prm = capi.default_reg_params(ransac_iters=3000, icp_iters=20, max_rmse=1.0)
ts = []
for i in range(10):
    t0 = time.time()
    sid = store.add(q_host)
    t1 = time.time()
    reg.batch_multi([sid], [row], params=prm)
    reg.scan_release(sid)
    ts.append((time.time() - t0, t1 - t0))
print("query alone (no retrieval), ms: total / scan add", np.round(np.array(ts) * 1e3, 3).tolist())
