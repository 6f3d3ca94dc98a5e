"""dev: what separates a right final pose from a wrong one among the registrations the pipeline ACCEPTS (2-D match ok,
RANSAC inlier ratio ok): statistics of the nearest-neighbour residuals at the final pose, for same-place candidates,
candidates 5-20 m away and different-world candidates, seeded by the coarse match as bench.py's coarse legs do."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import bench
from scipy.spatial import cKDTree
from gloc3d_amd import capi, synth, loop_detector as ld

va, vb, vq, vf, far_poses = bench.build_views("/tmp/views.npz")
store = capi.ScanStore()
reg = capi.Registrar(store=store)
cm = capi.CoarseMatcher(0)
prm = capi.default_reg_params(ransac_iters=bench.RANSAC_ITERS, icp_iters=bench.ICP_ITERS, min_inlier_ratio=bench.MIN_INLIER_RATIO, max_rmse=0.0)
cands, kinds, poses, clouds = [], [], [], []
for i, f in enumerate(vf):
    cands.append(store.add(f)); kinds.append("far"); poses.append(far_poses[i]); clouds.append(f)
for s in range(0, len(va), 3):
    P = bench.place_perturbation(1000 + s)
    sid = store.add_variant(store.add(va[s]), P, 0.01, 777 + s)
    cands.append(sid); kinds.append("near"); poses.append(bench.pool_pose(s) @ np.linalg.inv(P)); clouds.append(store.download(sid))
for s in range(len(vb)):
    cands.append(store.add(vb[s])); kinds.append("other-world"); poses.append(None); clouds.append(vb[s])
store.build_target_index_batch(cands)
grids = [cm.add_store_scan(store, c) for c in cands]
trees = [cKDTree(c[:, :3].astype(np.float64)) for c in clouds]
rows = []
extra = []
for v in range(len(vq)):
    qs = store.add(vq[v]); qg = cm.add_store_scan(store, qs)
    Tq = bench.query_view_pose(v)
    xyyaw, ratio, ok2 = cm.match(qg, grids)
    init = np.tile(np.eye(4, dtype=np.float32), (len(cands), 1, 1))
    for k in range(len(cands)):
        c, s = np.cos(xyyaw[k, 2]), np.sin(xyyaw[k, 2])
        if ok2[k]:
            init[k, :2, :2] = [[c, -s], [s, c]]; init[k, :2, 3] = xyyaw[k, :2]
    mode = os.environ.get("GATE_PRIOR", "coarse")
    if mode == "identity":
        init[:] = np.eye(4, dtype=np.float32)
        ok2 = np.ones(len(cands), bool)
    r = reg.batch_multi([qs], np.array([cands], np.uint32), params=prm, init_T=init[None])
    prm19 = capi.default_reg_params(ransac_iters=bench.RANSAC_ITERS, icp_iters=bench.ICP_ITERS - 1, min_inlier_ratio=bench.MIN_INLIER_RATIO, max_rmse=0.0)
    r19 = reg.batch_multi([qs], np.array([cands], np.uint32), params=prm19, init_T=init[None])
    qtree = cKDTree(vq[v][:, :3].astype(np.float64))
    for k in range(len(cands)):
        if not (ok2[k] and r["ok"][0, k]):
            continue
        T = r["T"][0, k].astype(np.float64)
        moved = vq[v][:, :3] @ T[:3, :3].T + T[:3, 3]
        d, _ = trees[k].query(moved)
        Ti = np.linalg.inv(T)
        back = clouds[k][:, :3] @ Ti[:3, :3].T + Ti[:3, 3]
        db, _ = qtree.query(back)
        if poses[k] is None:
            right, ep, er = False, 99.0, 99.0
        else:
            er, ep = ld.pose_error(np.linalg.inv(poses[k]) @ Tq, r["T"][0, k])
            right = ep < 1.0 and er < 5.0
        inl = d < 0.6
        dT = r["T"][0, k].astype(np.float64) @ np.linalg.inv(r19["T"][0, k].astype(np.float64))   # the last ICP step
        ang = np.degrees(np.arccos(np.clip((np.trace(dT[:3, :3]) - 1) / 2, -1, 1)))
        step = np.linalg.norm((vq[v][::97, :3] @ r19["T"][0, k][:3, :3].T.astype(np.float64) + r19["T"][0, k][:3, 3]) @ dT[:3, :3].T + dT[:3, 3]
                              - (vq[v][::97, :3] @ r19["T"][0, k][:3, :3].T.astype(np.float64) + r19["T"][0, k][:3, 3]), axis=1)
        extra.append((np.linalg.norm(dT[:3, 3]), ang, np.sqrt((step ** 2).mean()), step.max()))
        rows.append((kinds[k], right, ep, er, np.sqrt((d * d).mean()), inl.mean(), np.sqrt((d[inl] ** 2).mean()), np.median(d),
                     (d < 0.2).mean(), (db < 0.6).mean(), np.sqrt((db[db < 0.6] ** 2).mean()), np.median(db),
                     r["inliers"][0, k] / len(vq[v]), np.percentile(d, 75), np.sqrt((np.minimum(d, 0.6) ** 2).mean())))
    store.release(qs)
rows = [r_ + e_ for r_, e_ in zip(rows, extra)]
hdr = "kind right err_m err_deg rmse_all in06 rmse_in med in02 sym_in06 sym_rmse_in sym_med ransac_ratio p75 rmse_clip06 last_dt last_deg last_step_rms last_step_max"
print(hdr)
for r_ in sorted(rows, key=lambda x: (x[0], not x[1], x[4])):
    print(f"{r_[0]:11s} {int(r_[1])} " + " ".join(f"{x:7.3f}" for x in r_[2:]))
a = np.array([[float(x) for x in r_[1:]] for r_ in rows])
right = a[:, 0] > 0
for j, name in enumerate(hdr.split()[4:]):
    col = a[:, 3 + j]
    if right.any() and (~right).any():
        print(f"{name:12s}: right [{col[right].min():.3f}, {col[right].max():.3f}]  wrong [{col[~right].min():.3f}, {col[~right].max():.3f}]")
