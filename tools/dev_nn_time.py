"""Developer probe: average culled 1-NN pass time per mode / sources-per-lane (20 candidates, warm ICP passes)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from gloc3d_amd import capi, synth
w = synth.make_world(1001)
A = synth.lidar_scan(w, None, seed=1001)[:, :3]
B = synth.lidar_scan(w, synth.se3(5.0, (0.5, -0.3, 0.1)), seed=1002)[:, :3]
for mode, name in ((0, "culled"), (1, "exhaustive")):
    for cs in (1, 2, 4):
        reg = capi.Registrar(); reg.set_option(capi.REG_OPT_NN_SRC_PER_LANE, cs); reg.set_option(capi.REG_OPT_NN_MODE, mode)
        ids = [reg.scan_upload(A), reg.scan_upload(B)]
        prm = capi.default_reg_params(ransac_iters=3000, icp_iters=20)
        reg.batch_ids(ids[1], [ids[0]] * 20, params=prm)
        reg.set_option(capi.REG_OPT_PROFILE, 1); reg.profile_reset()
        for _ in range(3): reg.batch_ids(ids[1], [ids[0]] * 20, params=prm)
        ms, n = reg.profile("nn"); pairs, launches = reg.nn_stats()
        print(f"{name} CS={cs}: {ms/n*1e3:.1f} us/pass over {n} passes; pairs evaluated/pass {pairs/launches:.3e}")
        reg.close()
