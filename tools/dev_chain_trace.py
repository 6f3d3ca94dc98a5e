"""dev: the time line of the chained launch (GLOC_REG_OPT_NN_CHAIN) for one query alone -- per (pass, job) stamps of the
100 MHz clock through gloc_reg_debug_chain_trace.  usage: dev_chain_trace.py [icp_iters]"""
import sys, os, time, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import bench
from gloc3d_amd import capi
icp = int(sys.argv[1]) if len(sys.argv) > 1 else 20
traj, world_a, world_b = bench.headline_world(bench.N_PLACES_1GPU)
store = capi.ScanStore()
g = 300
qid = store.add_raycast(world_a, [traj[g] @ bench.query_offset(0)], np.array([bench.QUERY_SEED], np.uint64))[0]
places = [g + d for d in (0, 1, -1, 2, -2, 3, -3, 4, -4, 5, -5, 6, -6, 7, -7, 8, -8, 9, -9, 10)]
row = [store.add_raycast(world_b if pl % bench.NEG_EVERY == 1 else world_a, [traj[pl]], np.array([bench.PLACE_SEED + pl], np.uint64))[0] for pl in places]
store.build_target_index_batch(row)
reg = capi.Registrar(store=store)
f = capi.lib().gloc_reg_debug_chain_trace
f.restype = C.c_int
f.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_size_t, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32)]
prm = capi.default_reg_params(ransac_iters=3000, icp_iters=icp, max_rmse=1.0)
for chain in (0, 1):
    reg.set_option(capi.REG_OPT_NN_CHAIN, chain)
    ts = []
    for _ in range(6):
        t0 = time.time()
        reg.batch_multi([qid], [row], params=prm)
        ts.append(time.time() - t0)
    print("chain", chain, "registration of 20 jobs, ms:", np.round(np.array(ts) * 1e3, 3))
f(reg._h, 1, None, 0, None, None)
reg.batch_multi([qid], [row], params=prm)
npass, njobs = C.c_uint32(), C.c_uint32()
f(reg._h, 1, None, 0, C.byref(npass), C.byref(njobs))
tr = np.zeros((npass.value, njobs.value, 16), np.uint32)
f(reg._h, 1, tr.ctypes.data_as(C.c_void_p), tr.size, C.byref(npass), C.byref(njobs))
t0 = tr[0, :, 0].min()
us = (tr.astype(np.int64) - int(t0)) / 100.0
names = ["first wave here", "first past wait", "last wave leaves", "reducer0 waits", "sees done", "sub stored", "last reducer", "sums loaded", "solved"]
print("launch spans", us[:, :, 8].max(), "us;  passes", npass.value, "jobs", njobs.value)
for j in (0, 1, 10, 19):
    print("job", j)
    for p in range(min(npass.value, 6)):
        print("  pass", p, " ".join(f"{names[k]} {us[p, j, k]:8.1f}" for k in range(9)))
print("search waves past their wait at the ordinary look / the first look past the caches / after polling, per pass (all jobs):")
for p in range(npass.value):
    print("  pass", p, tr[p, :, 9].sum(), tr[p, :, 10].sum(), tr[p, :, 11].sum())
print("per pass, over jobs (us):")
for p in range(npass.value):
    print(f"  pass {p:2d}: first wave {us[p,:,0].min():8.1f}  last leaves {us[p,:,2].max():8.1f}  solved {us[p,:,8].min():8.1f} .. {us[p,:,8].max():8.1f}   "
          f"search span per job {np.mean(us[p,:,2]-us[p,:,1]):6.1f}  wait at head {np.mean(us[p,:,1]-us[p,:,0]):6.1f}  done->solved {np.mean(us[p,:,8]-us[p,:,2]):6.1f} (seen {np.mean(us[p,:,4]-us[p,:,2]):5.1f}, reduce {np.mean(us[p,:,6]-us[p,:,4]):5.1f}, load {np.mean(us[p,:,7]-us[p,:,6]):5.1f}, solve {np.mean(us[p,:,8]-us[p,:,7]):5.1f})")
