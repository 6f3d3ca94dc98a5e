"""dev: one query alone, 12 times -- run under `rocprofv3 --kernel-trace` (tools/dev_lone_timeline.sh prints the last
repetition's kernels: start offset, duration, gap to the previous kernel's end)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import bench
from gloc3d_amd import capi
traj, world_a, world_b = bench.headline_world(bench.N_PLACES_1GPU)
store = capi.ScanStore()
g = int(sys.argv[1]) if len(sys.argv) > 1 else 300
places = [g + d for d in (0, 1, -1, 2, -2, 3, -3, 4, -4, 5, -5, 6, -6, 7, -7, 8, -8, 9, -9, 10)]
row = [store.add_raycast(world_b if pl % bench.NEG_EVERY == 1 else world_a, [traj[pl]], np.array([bench.PLACE_SEED + pl], np.uint64))[0] for pl in places]
store.build_target_index_batch(row)
qid0 = store.add_raycast(world_a, [traj[g] @ bench.query_offset(0)], np.array([bench.QUERY_SEED], np.uint64))[0]
q_host = store.download(qid0) if hasattr(store, "download") else None
reg = capi.Registrar(store=store)
prm = capi.default_reg_params(ransac_iters=3000, icp_iters=20, max_rmse=1.0)
ts = []
for i in range(12):
    t0 = time.time()
    reg.batch_multi([qid0], [row], params=prm)
    ts.append(time.time() - t0)
print("registration of 20 jobs, ms:", np.round(np.array(ts) * 1e3, 3))
