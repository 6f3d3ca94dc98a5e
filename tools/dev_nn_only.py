"""Developer probe: run only ICP passes (culled 1-NN dominated) for counter collection."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from gloc3d_amd import capi, synth
cs = int(sys.argv[1]) if len(sys.argv) > 1 else 1
mode = int(sys.argv[2]) if len(sys.argv) > 2 else 0
w = synth.make_world(1001)
A = synth.lidar_scan(w, None, seed=1001)[:, :3]
B = synth.lidar_scan(w, synth.se3(5.0, (0.5, -0.3, 0.1)), seed=1002)[:, :3]
reg = capi.Registrar(); reg.set_option(capi.REG_OPT_NN_SRC_PER_LANE, cs); reg.set_option(capi.REG_OPT_NN_MODE, mode)
ids = [reg.scan_upload(A), reg.scan_upload(B)]
reg.batch_ids(ids[1], [ids[0]] * 20, params=capi.default_reg_params(ransac_iters=0, icp_iters=6))
