"""dev: host-side overhead of one registration batch (500 jobs) -- the same call with no RANSAC / ICP work."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import bench
from gloc3d_amd import capi, synth
w = synth.make_world(1001)
store = capi.ScanStore()
base = [store.add(np.ascontiguousarray(synth.lidar_scan(w, bench.pool_pose(s), seed=3000 + s)[:, :3])) for s in range(4)]
cands = [store.add_variant(base[g % 4], bench.place_perturbation(g), 0.01, 7000 + g) for g in range(40)]
qs = [store.add_variant(base[g % 4], bench.place_perturbation(100 + g), 0.01, 8000 + g) for g in range(25)]
reg = capi.Registrar(store=store)
reg.set_option(capi.REG_OPT_PROFILE, 1)
cl = [[cands[(q + c) % 40] for c in range(20)] for q in range(25)]
for name, prm in (("no work (ransac 0, icp 0)", capi.default_reg_params(ransac_iters=0, icp_iters=0)),
                  ("1 ICP pass", capi.default_reg_params(ransac_iters=0, icp_iters=1)),
                  ("ransac 3000 only", capi.default_reg_params(ransac_iters=3000, icp_iters=0)),
                  ("full", capi.default_reg_params(ransac_iters=3000, icp_iters=20, max_rmse=1.0))):
    reg.batch_multi(qs, cl, params=prm)
    reg.profile_reset()
    t = time.time()
    for _ in range(5):
        r = reg.batch_multi(qs, cl, params=prm)
    dt = (time.time() - t) / 5
    g = sum(reg.profile(k)[0] for k in ("nn", "ransac_score", "ransac_hyp", "accum", "solve")) / 5
    print(f"{name}: {dt*1e3:.2f} ms per batch of 500 jobs, GPU stage sum {g:.2f} ms, difference {dt*1e3-g:.2f} ms")
