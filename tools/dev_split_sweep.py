"""dev: one query alone (20 full-size candidates, RANSAC 3000 adaptive + ICP 20) under different split plans
(REG_OPT_NN_SPLIT_HELPERS / _THRESH): device time of the 1-NN launches and wall clock of the query."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import bench
from gloc3d_amd import capi, synth

va, vb, vq, vf, far_poses = bench.build_views("/tmp/views.npz")
store = capi.ScanStore()
base_a = [store.add(v) for v in va]
base_b = [store.add(v) for v in vb]
cands = [store.add_variant(base_b[(g // 4) % len(base_b)] if g % 4 == 1 else base_a[g % len(base_a)], bench.place_perturbation(g), 0.01, 7000 + g)
         for g in range(20)]
store.build_target_index_batch(cands)
prm = capi.default_reg_params(ransac_iters=bench.RANSAC_ITERS, icp_iters=bench.ICP_ITERS, min_inlier_ratio=bench.MIN_INLIER_RATIO,
                              max_rmse=bench.MAX_RMSE)
cid = np.array([cands], np.uint32)
plans = [(0, 60000)]
if len(sys.argv) > 1:
    plans = [tuple(int(x) for x in a.split(",")) for a in sys.argv[1:]]
ref = None
for plan in plans:
    helpers, thresh = plan[:2]
    reg = capi.Registrar(store=store)
    if len(plan) > 2:
        reg.set_option(capi.REG_OPT_NN_JOB_GROUP, plan[2])
    if len(plan) > 3:
        reg.set_option(capi.REG_OPT_NN_SUB_JOBS, plan[3])
    if len(plan) > 4:
        reg.set_option(capi.REG_OPT_NN_SRC_PER_LANE, plan[4])
    reg.set_option(capi.REG_OPT_NN_SPLIT_HELPERS, helpers)
    reg.set_option(capi.REG_OPT_NN_SPLIT_THRESH, thresh)
    res = {}
    for prof in (1, 0):
        reg.set_option(capi.REG_OPT_PROFILE, prof)
        ts = []
        for j in range(12):
            if j == 4:
                reg.profile_reset()
            sid = store.add(vq[j % len(vq)])
            t1 = time.perf_counter()
            out = reg.batch_multi([sid], cid, params=prm)
            ts.append(time.perf_counter() - t1)
            store.release(sid)
            if j == 0:
                if ref is None:
                    ref = out
                same = (out["T"].view(np.uint32) == ref["T"].view(np.uint32)).all() and (out["inliers"] == ref["inliers"]).all()
        res[prof] = np.median(ts[4:]) * 1e3
        if prof:
            ms, cnt = reg.profile("nn")
            nn_us = ms / max(cnt, 1) * 1e3
            sms, scnt = reg.profile("solve")
    print(f"helpers {helpers:4d} thresh {thresh:6d} job group {plan[2] if len(plan) > 2 else 24} subs {plan[3] if len(plan) > 3 else 0}: nn {nn_us:6.1f} us per launch, solve {sms / max(scnt, 1) * 1e3:5.1f} us; batch_multi {res[0]:.3f} ms (profiling on: {res[1]:.3f}); same bits as the first plan: {bool(same)}", flush=True)
    reg.close()
