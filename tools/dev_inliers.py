"""dev: inlier ratios / ok of positive and negative candidates for the bench's worlds."""
import sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from gloc3d_amd import synth
n_boxes = int(sys.argv[1]) if len(sys.argv) > 1 else 70
wa, wb = synth.make_world(1001, n_boxes=n_boxes), synth.make_world(2002, n_boxes=n_boxes)
q = np.ascontiguousarray(synth.lidar_scan(wa, bench.pool_pose(4) @ synth.se3(1.5, (0.3, -0.2, 0.02)), seed=9)[:, :3])
pos = [np.ascontiguousarray(synth.lidar_scan(wa, bench.pool_pose(s), seed=3000 + s)[:, :3]) for s in (4, 9, 14, 23)]
neg = [np.ascontiguousarray(synth.lidar_scan(wb, synth.se3(7.0 * s, (1.5 * s, -0.7 * s, 0.0)), seed=5000 + s)[:, :3]) for s in (0, 3)]
from gloc3d_amd import capi
r = capi.Registrar()
for thr in (0.6, 0.3):
    prm = capi.default_reg_params(ransac_iters=3000, icp_iters=20, min_inlier_ratio=0.8, inlier_thresh=thr)
    g = r.batch(q, pos + neg, params=prm)
    print("boxes", n_boxes, "thr", thr, "n", len(q), "inlier ratio", np.round(g["inliers"] / len(q), 3), "ok", g["ok"], "rmse", np.round(g["rmse"], 3))
