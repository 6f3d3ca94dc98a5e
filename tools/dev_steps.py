"""dev: how far do points move per ICP pass? (decides whether pass-to-pass coherence can be exploited)"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import bench
from gloc3d_amd import capi, synth
wa, wb = synth.make_world(1001), synth.make_world(2002)
store = capi.ScanStore()
q = store.add(np.ascontiguousarray(synth.lidar_scan(wa, bench.pool_pose(10) @ synth.se3(1.5, (0.3, -0.2, 0.02)), seed=9000)[:, :3]))
cands = [store.add(np.ascontiguousarray(synth.lidar_scan(wa, bench.pool_pose(s), seed=3000 + s)[:, :3])) for s in (10, 12, 16, 20)]
cands.append(store.add(np.ascontiguousarray(synth.lidar_scan(wb, None, seed=5000)[:, :3])))
reg = capi.Registrar(store=store)
prev = None
for it in range(0, 21):
    r = reg.batch_ids(q, cands, params=capi.default_reg_params(ransac_iters=3000, icp_iters=it))
    T = r["T"].astype(np.float64)
    if prev is not None:
        out = []
        for c in range(len(cands)):
            D = np.linalg.inv(prev[c]) @ T[c]
            ang = np.degrees(np.arccos(np.clip((np.trace(D[:3, :3]) - 1) / 2, -1, 1)))
            out.append(f"{np.linalg.norm(D[:3, 3]) * 100:.2f}cm/{ang * np.pi / 180 * 30 * 100:.2f}cm@30m")
        print(f"pass {it}: " + "  ".join(out), flush=True)
    prev = T
