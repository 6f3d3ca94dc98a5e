"""dev: where one query alone (20 full-size candidates, RANSAC 3000 adaptive + ICP 20) spends its time: wall clock of
the three host calls, device time per stage (HIP events around every kernel), and the same with profiling off."""
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
reg = capi.Registrar(store=store)
prm = capi.default_reg_params(ransac_iters=bench.RANSAC_ITERS, icp_iters=bench.ICP_ITERS, min_inlier_ratio=bench.MIN_INLIER_RATIO,
                              max_rmse=bench.MAX_RMSE)
cid = np.array([cands], np.uint32)
for prof, cs in ((1, 2), (0, 2), (0, 1), (0, 4)):
    reg.set_option(capi.REG_OPT_PROFILE, prof)
    reg.set_option(capi.REG_OPT_NN_SRC_PER_LANE, cs)
    ts = []
    for j in range(12):
        if j == 4:
            reg.profile_reset()
        t0 = time.perf_counter()
        sid = store.add(vq[j % len(vq)])
        t1 = time.perf_counter()
        reg.batch_multi([sid], cid, params=prm)
        t2 = time.perf_counter()
        store.release(sid)
        ts.append((t1 - t0, t2 - t1, time.perf_counter() - t2))
    a = np.median(np.array(ts[4:]), axis=0) * 1e3
    print(f"profiling {prof}, sources per lane {cs}: add {a[0]:.3f} ms, batch_multi {a[1]:.3f} ms, release {a[2]:.3f} ms")
    if prof:
        for n in ("nn", "ransac_score", "ransac_hyp", "accum", "solve"):
            ms, cnt = reg.profile(n)
            print(f"   {n}: {ms / 8:.3f} ms per query in {cnt / 8:.0f} launches ({ms / max(cnt, 1) * 1e3:.1f} us each)")
