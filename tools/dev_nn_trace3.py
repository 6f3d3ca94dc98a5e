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
n_cand = tr[:, 3] >> 16
n_ties = tr[:, 5] >> 24
tr[:, 5] &= 0xFFFFFF
rh = tr[:, 1].copy()
tr[:, 1] = 0
tr[:, 3] &= 0xFFFF
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
# cycles against the wave's rank in the launch order (rank 0 = the widest source group of its job): what a split of
# the widest groups would have to cover
n_wg = int(tr.shape[0] // 20)
per = n_wg * 20     # one job group (20 < job_group): blockIdx = wg * 20 + job
wid = np.arange(tr.shape[0])
rank = (wid % per) // 20
t_all = tr[:, 0].astype(np.float64)
print(f"all waves: mean {t_all[ok].mean():.0f} p50 {np.percentile(t_all[ok], 50):.0f} p90 {np.percentile(t_all[ok], 90):.0f} "
      f"p99 {np.percentile(t_all[ok], 99):.0f} p99.9 {np.percentile(t_all[ok], 99.9):.0f} max {t_all[ok].max():.0f}")
for lo, hi in ((0, 4), (4, 16), (16, 64), (64, 256), (256, n_wg)):
    s = ok & (rank >= lo) & (rank < hi)
    if s.any():
        print(f"   ranks [{lo}, {hi}): waves {s.sum()} cycles mean {t_all[s].mean():.0f} max {t_all[s].max():.0f} chunks {tr[s, 2].mean():.1f} items {tr[s, 4].mean():.0f}")
top = np.argsort(-t_all)[:12]
print("slowest waves (cycles, rank, job, candidate chunks, chunks, items):", [(int(t_all[i]), int(rank[i]), int(job[i]), int(n_cand[i]), int(tr[i, 2]), int(tr[i, 4])) for i in top])
print("candidate chunks per wave: mean", n_cand[ok].mean(), "p99", np.percentile(n_cand[ok], 99), "max", n_cand[ok].max(), "; share of waves with > 32:", (n_cand[ok] > 32).mean())
A = np.stack([n_cand[ok], tr[ok, 2], tr[ok, 4], np.ones(ok.sum())], 1).astype(np.float64)
coef = np.linalg.lstsq(A, t_all[ok], rcond=None)[0]
print("cycles ~ %.0f x candidate + %.0f x processed + %.1f x item + %.0f" % tuple(coef))
# what the widest groups look like: nearest-neighbour distances (at the final pose) of their 128 sources
from scipy.spatial import cKDTree
ix = store.debug_index(qid)
qs = qv[ix["perm"]]
for jb in (0, 11, 5):
    T = r["T"][jb].astype(np.float64)
    moved = qs @ T[:3, :3].T + T[:3, 3]
    d, _ = cKDTree(store.download(cands[jb])).query(moved)
    for rk in (0, 5, 100, 506):
        g = int(ix["order2"][rk])
        dd = np.sort(d[g * 128:(g + 1) * 128])[::-1]
        pts = moved[g * 128:(g + 1) * 128]
        print(f"job {jb} rank {rk} group {g}: box {np.round(pts.max(0) - pts.min(0), 1)} nn dist top {np.round(dd[:6], 2)} median {np.median(dd):.2f}; above 1 m: {(dd > 1).sum()}, above 4x median: {(dd > 4 * np.median(dd)).sum()}")

h = np.stack([(rh >> (8 * q)) & 255 for q in range(4)], 1)[ok]
print("rounds by occupancy (<=16, <=32, <=48, <=64 items):", h.sum(0), "per wave", np.round(h.mean(0), 2))
print("contested sources (tie path) per wave:", n_ties[ok].mean(), "; waves with any:", (n_ties[ok] > 0).mean(), "; per source:", n_ties[ok].sum() / (128.0 * ok.sum()))
