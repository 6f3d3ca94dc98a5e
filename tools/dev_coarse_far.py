"""dev: candidates 5-20 m from the query -- the coarse 2-D match's pose against the ground truth, and the 3-D
registration from the identity prior and from that seed (what bench.py's coarse_seeded_far_5_20m leg shows)."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import bench
from gloc3d_amd import capi, synth, loop_detector as ld

va, vb, vq, vf, far_poses = bench.build_views("/tmp/views.npz")
store = capi.ScanStore()
reg = capi.Registrar(store=store)
cm = capi.CoarseMatcher(0)
prm = capi.default_reg_params(ransac_iters=bench.RANSAC_ITERS, icp_iters=bench.ICP_ITERS, min_inlier_ratio=bench.MIN_INLIER_RATIO,
                              max_rmse=bench.MAX_RMSE)
for v in (0, 5):
    qs = store.add(vq[v]); qg = cm.add_store_scan(store, qs)
    Tq = bench.query_view_pose(v)
    ids = [store.add(f) for f in vf]
    gs = [cm.add_store_scan(store, i) for i in ids]
    xyyaw, ratio, ok = cm.match(qg, gs)
    init = np.tile(np.eye(4, dtype=np.float32), (len(ids), 1, 1))
    for k in range(len(ids)):
        c, s = np.cos(xyyaw[k, 2]), np.sin(xyyaw[k, 2])
        if ok[k]:
            init[k, :2, :2] = [[c, -s], [s, c]]; init[k, :2, 3] = xyyaw[k, :2]
    r0 = reg.batch_multi([qs], np.array([ids], np.uint32), params=prm)
    r1 = reg.batch_multi([qs], np.array([ids], np.uint32), params=prm, init_T=init[None])
    for k, Tf in enumerate(far_poses):
        gt = np.linalg.inv(Tf) @ Tq
        d = np.linalg.norm(Tf[:3, 3] - Tq[:3, 3])
        gyaw = np.degrees(np.arctan2(gt[1, 0], gt[0, 0]))
        e0 = ld.pose_error(gt, r0["T"][0, k]); e1 = ld.pose_error(gt, r1["T"][0, k]); ec = ld.pose_error(gt, init[k])
        print(f"q{v} far{k} dist {d:5.1f} m gt yaw {gyaw:6.1f} | coarse ok {int(ok[k])} ratio {ratio[k]:.2f} scale {cm.last_scale[k]:.3f} "
              f"err {ec[1]:5.2f} m {ec[0]:5.1f} deg | identity: ok {int(r0['ok'][0,k])} inl {r0['inliers'][0,k]/len(vq[v]):.2f} rmse {r0['rmse'][0,k]:.2f} "
              f"err {e0[1]:5.2f} m {e0[0]:4.1f} | seeded: ok {int(r1['ok'][0,k])} inl {r1['inliers'][0,k]/len(vq[v]):.2f} rmse {r1['rmse'][0,k]:.2f} err {e1[1]:5.2f} m {e1[0]:4.1f}")
