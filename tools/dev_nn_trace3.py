"""dev: per-wave trace of the culled 1-NN kernel on a bench-like query (15 positives + 5 negatives), last ICP pass."""
import sys, os, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import bench
from gloc3d_amd import capi, synth
wa, wb = synth.make_world(1001), synth.make_world(2002)
store = capi.ScanStore()
base_a = [store.add(np.ascontiguousarray(synth.lidar_scan(wa, bench.pool_pose(s), seed=3000 + s)[:, :3])) for s in range(0, 20)]
base_b = [store.add(np.ascontiguousarray(synth.lidar_scan(wb, synth.se3(7.0 * s, (1.5 * s, -0.7 * s, 0.0)), seed=5000 + s)[:, :3])) for s in range(2)]
qv = np.ascontiguousarray(synth.lidar_scan(wa, bench.pool_pose(10) @ synth.se3(1.5, (0.3, -0.2, 0.02)), seed=9000)[:, :3])
qid = store.add(qv)
cands = []
for c in range(20):
    g = c
    cands.append(store.add_variant(base_b[(g // 4) % 2] if g % 4 == 1 else base_a[g], bench.place_perturbation(g), 0.01, 7000 + g))
reg = capi.Registrar(store=store)
L = capi.lib(); f = L.gloc_reg_debug_trace; f.restype = C.c_int
f.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_size_t, C.POINTER(C.c_size_t)]
f(reg._h, 1, None, 0, None)
prm = capi.default_reg_params(ransac_iters=3000, icp_iters=int(sys.argv[1]) if len(sys.argv) > 1 else 20, max_rmse=1.0)
r = reg.batch_ids(qid, cands, params=prm)
n = C.c_size_t(); f(reg._h, 1, None, 0, C.byref(n))
tr = np.zeros((n.value, 8), np.uint32); f(reg._h, 1, tr.ctypes.data_as(C.c_void_p), n.value, C.byref(n))
job = tr[:, 6]
ok = tr[:, 0] > 0
print("ok", r["ok"], "rmse", np.round(r["rmse"], 2))
for name, sel in (("positives", np.isin(job, [c for c in range(20) if c % 4 != 1])), ("negatives", np.isin(job, [c for c in range(20) if c % 4 == 1]))):
    t = tr[ok & sel].astype(np.float64)
    tot, chunks, pro = t[:, 0], t[:, 1], t[:, 5]
    w7 = tr[ok & sel][:, 7]
    live_sb, live_pairs, steps = (w7 >> 20), (w7 >> 10) & 1023, w7 & 1023
    epi = 0 * tot
    sweep = tot - chunks - pro
    print(f"{name}: waves {len(t)} cycles mean {tot.mean():.0f} p50 {np.percentile(tot,50):.0f} p99 {np.percentile(tot,99):.0f} max {tot.max():.0f}")
    print(f"   prologue {pro.mean():.0f}  sweep outside chunks {sweep.mean():.0f}  chunk processing {chunks.mean():.0f} ({t[:,2].mean():.1f} chunks, {t[:,3].mean():.1f} rounds, {t[:,4].mean():.0f} items)  ")
    print(f"   per processed chunk: listed {t[:,5].sum() and 0 or 0} live sub-blocks {live_sb.sum()/t[:,2].sum():.2f} of 8, live pairs {live_pairs.sum()/t[:,2].sum():.2f} of 4, test steps {steps.sum()/t[:,2].sum():.2f}, items {t[:,4].sum()/t[:,2].sum():.1f}")
