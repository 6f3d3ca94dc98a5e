"""Ground pre-alignment stage benchmark ("next" row N3): full-size synthetic scans, device resident."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from gloc3d_amd import capi, synth
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from util import ground_scene

cloud, _ = ground_scene(3.0, -2.0, n_az=2000)
d_in = torch.from_numpy(cloud).cuda(); d_out = torch.empty_like(d_in)
est = capi.GroundEstimator()
T, info = est.estimate_device(d_in.data_ptr(), cloud.shape[0], 4, d_out.data_ptr())
print(f"scan {cloud.shape[0]} points, {info['n_near']} within 20 m, ground bin {info['ground_bin']} with {info['n_ground']} points, "
      f"{info['inliers']} plane inliers after {info['iters_used']} iterations")
est.set_profile(True)
n = 20
t = time.time()
for _ in range(n): est.estimate_device(d_in.data_ptr(), cloud.shape[0], 4, d_out.data_ptr())
dt = (time.time() - t) / n
print(f"{dt*1e3:.2f} ms per scan ({1/dt:.0f} scans/s)")
m = info["n_near"]
for k in ("ground_knn", "ground_normals", "ground_plane", "ground_transform"):
    ms, c = est.profile(k); print(f"  {k}: {ms/c*1e3:.1f} us/launch")
ms, c = est.profile("ground_knn")
print(f"  10-NN (index build + culled search): {ms/c*1e3:.1f} us for {m} points")
est.set_option(capi.GROUND_OPT_KNN_EXHAUSTIVE, 1); est.set_profile(True)
for _ in range(3): est.estimate_device(d_in.data_ptr(), cloud.shape[0], 4, d_out.data_ptr())
est2 = capi.GroundEstimator(); est2.set_option(capi.GROUND_OPT_KNN_EXHAUSTIVE, 1); est2.set_profile(True)
for _ in range(5): est2.estimate_device(d_in.data_ptr(), cloud.shape[0], 4, d_out.data_ptr())
ms, c = est2.profile("ground_knn"); print(f"  exhaustive 10-NN for comparison: {ms/c*1e3:.1f} us/launch")
if len(sys.argv) > 1:
    import oracle
    t = time.time(); oracle.ground_estimate(cloud); print(f"CPU oracle (exhaustive 10-NN, 1 core): {time.time()-t:.1f} s")
