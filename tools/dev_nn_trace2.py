"""Developer probe: per-wave trace of the culled 1-NN kernel for several candidate counts."""
import sys, os, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from gloc3d_amd import capi, synth
cs = int(sys.argv[1]) if len(sys.argv) > 1 else 2
mode = int(sys.argv[2]) if len(sys.argv) > 2 else 0
w = synth.make_world(1001)
A = synth.lidar_scan(w, None, seed=1001)[:, :3]
B = synth.lidar_scan(w, synth.se3(5.0, (0.5, -0.3, 0.1)), seed=1002)[:, :3]
reg = capi.Registrar(); reg.set_option(capi.REG_OPT_NN_SRC_PER_LANE, cs); reg.set_option(capi.REG_OPT_NN_MODE, mode)
L = capi.lib(); f = L.gloc_reg_debug_trace; f.restype = C.c_int
f.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_size_t, C.POINTER(C.c_size_t)]
ids = [reg.scan_upload(A), reg.scan_upload(B)]
f(reg._h, 1, None, 0, None)
for nc in (1, 4, 20):
    reg.batch_ids(ids[1], [ids[0]] * nc, params=capi.default_reg_params(ransac_iters=0, icp_iters=3))
    n = C.c_size_t(); f(reg._h, 1, None, 0, C.byref(n))
    tr = np.zeros((n.value, 4), np.uint32); f(reg._h, 1, tr.ctypes.data_as(C.c_void_p), n.value, C.byref(n))
    tr = tr[tr[:, 0] > 0]
    cyc = tr[:, 0].astype(np.float64)
    print(f"nc={nc}: waves {len(tr)} cycles mean {cyc.mean():.0f} p50 {np.percentile(cyc,50):.0f} p90 {np.percentile(cyc,90):.0f} p99 {np.percentile(cyc,99):.0f} p99.9 {np.percentile(cyc,99.9):.0f} max {cyc.max():.0f}")
    order = np.argsort(-cyc)[:8]
    print("   heaviest waves (idx, cycles, cand chunks, chunks, rounds):", [(int(i), int(tr[i,0]), int(tr[i,1]), int(tr[i,2]), int(tr[i,3])) for i in order])
