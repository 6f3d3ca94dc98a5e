"""Developer probe: culled 1-NN pass time vs number of candidates in the launch."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from gloc3d_amd import capi, synth
if os.environ.get('GLOC3D_DEV_LIB'): capi.LIB_PATH = os.environ['GLOC3D_DEV_LIB']
mode = int(sys.argv[1]) if len(sys.argv) > 1 else 0
heavy = int(sys.argv[2]) if len(sys.argv) > 2 else 100
w = synth.make_world(1001)
A = synth.lidar_scan(w, None, seed=1001)[:, :3]
B = synth.lidar_scan(w, synth.se3(5.0, (0.5, -0.3, 0.1)), seed=1002)[:, :3]
reg = capi.Registrar(); reg.set_option(capi.REG_OPT_NN_SRC_PER_LANE, 2); reg.set_option(capi.REG_OPT_NN_MODE, mode); reg.set_option(capi.REG_OPT_NN_HEAVY_PERMILLE, heavy)
ids = [reg.scan_upload(A), reg.scan_upload(B)]
prm = capi.default_reg_params(ransac_iters=0, icp_iters=12)
for nc in (20, 40):
    reg.batch_ids(ids[1], [ids[0]] * nc, params=prm)
    reg.set_option(capi.REG_OPT_PROFILE, 1); reg.profile_reset()
    for _ in range(3): reg.batch_ids(ids[1], [ids[0]] * nc, params=prm)
    ms, n = reg.profile("nn")
    print(f"mode {mode} heavy {heavy} candidates {nc}: {ms/n*1e3:.1f} us/pass -> {ms/n*1e3/nc:.1f} us per candidate")
    reg.set_option(capi.REG_OPT_PROFILE, 0)
