"""dev: per-wave trace of the culled 1-NN kernel on bench-like queries (15 positives + 5 negatives each), last ICP pass.
usage: dev_nn_trace3.py [icp_iters [n_queries]]   (n_queries x 20 jobs in ONE launch; 8 fill the chip like the bench)"""
import sys, os, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import bench
from gloc3d_amd import capi, synth
W = 32  # NN_TRACE_WORDS
icp = int(sys.argv[1]) if len(sys.argv) > 1 else 20
nq = int(sys.argv[2]) if len(sys.argv) > 2 else 1
# round 6: the headline's data -- distinct device ray-casts along the loop (bench.headline_world); query q revisits place
# 300 + 170 q, its 20 candidates are the places around it in index order (what the descriptor retrieval returns)
traj, world_a, world_b = bench.headline_world(bench.N_PLACES_1GPU)
store = capi.ScanStore()
g0 = [300 + 170 * q for q in range(nq)]
qids = store.add_raycast(world_a, [traj[g] @ bench.query_offset(q) for q, g in enumerate(g0)], np.array([bench.QUERY_SEED + q for q in range(nq)], np.uint64))
cands_all, cand_dist = [], []
for q, g in enumerate(g0):
    places = [g + d for d in (0, 1, -1, 2, -2, 3, -3, 4, -4, 5, -5, 6, -6, 7, -7, 8, -8, 9, -9, 10)]
    row = []
    for pl in places:
        neg = pl % bench.NEG_EVERY == 1
        row.append(store.add_raycast(world_b if neg else world_a, [traj[pl]], np.array([bench.PLACE_SEED + pl], np.uint64))[0])
        cand_dist.append(-1.0 if neg else float(np.linalg.norm(traj[pl][:2, 3] - traj[g][:2, 3])))
    cands_all.append(row)
    store.build_target_index_batch(row)
cands = cands_all[0]
cand_dist = np.array(cand_dist)
reg = capi.Registrar(store=store)
L = capi.lib(); f = L.gloc_reg_debug_trace; f.restype = C.c_int
f.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_size_t, C.POINTER(C.c_size_t)]
f(reg._h, 1, None, 0, None)
prm = capi.default_reg_params(ransac_iters=3000, icp_iters=icp, max_rmse=1.0)
r = reg.batch_multi(qids, cands_all, params=prm)
n = C.c_size_t(); f(reg._h, 1, None, 0, C.byref(n))
tr = np.zeros((n.value, W), np.uint32); f(reg._h, 1, tr.ctypes.data_as(C.c_void_p), n.value, C.byref(n))
job = tr[:, 6]
n_cand = tr[:, 3] >> 16
n_ties = tr[:, 5] >> 24
tr[:, 5] &= 0xFFFFFF
rh = tr[:, 1].copy()
tr[:, 3] &= 0xFFFF
ok = tr[:, 0] > 0
print("queries", nq, "jobs", nq * 20, "waves traced", int(ok.sum()))
print("ok", np.asarray(r["ok"]).reshape(nq, 20)[0], "rmse", np.round(np.asarray(r["rmse"]).reshape(nq, 20)[0], 2))
cand_of = job % 20
names = ["load+xform+box", "upper bounds", "chunk-box batches", "thinning", "candidate lane tests", "staging+listing",
         "sub-block tests", "evaluation rounds", "bound refresh", "(whole sweep)", "index recovery", "contested minima",
         "outputs+moments", "moment reduction"]
dist_of_job = cand_dist[np.minimum(job, len(cand_dist) - 1)]
for name, sel in (("all", np.ones_like(ok)), ("same world <= 2.5 m", (dist_of_job >= 0) & (dist_of_job <= 2.5)),
                  ("same world > 2.5 m", dist_of_job > 2.5), ("other world", dist_of_job < 0)):
    t = tr[ok & sel].astype(np.float64)
    tot = t[:, 0]
    w7 = tr[ok & sel][:, 7]
    live_sb, live_pairs, steps = (w7 >> 20), (w7 >> 10) & 1023, w7 & 1023
    print(f"{name}: waves {len(t)} cycles mean {tot.mean():.0f} p50 {np.percentile(tot,50):.0f} p99 {np.percentile(tot,99):.0f} max {tot.max():.0f}")
    print(f"   per wave: candidate chunks {n_cand[ok & sel].mean():.1f}, processed {t[:,2].mean():.2f}, rounds {t[:,3].mean():.2f}, items {t[:,4].mean():.0f}, test steps {steps.mean():.2f}")
    print(f"   per processed chunk: live sub-blocks {live_sb.sum()/t[:,2].sum():.2f} of 8, live pairs {live_pairs.sum()/t[:,2].sum():.2f} of 4, test steps {steps.sum()/t[:,2].sum():.2f}, items {t[:,4].sum()/t[:,2].sum():.1f}")
    reg_c = t[:, 8:22]
    inner = reg_c[:, 2:9].sum(1)
    other_sweep = reg_c[:, 9] - inner
    acc = reg_c[:, [0, 1]].sum(1) + reg_c[:, 9] + reg_c[:, 10:14].sum(1)
    print("   share of the wave's cycles per region (sum over waves / sum of totals):")
    for k, nm in enumerate(names):
        if k == 9:
            print(f"      {'sweep: loops, ballots, waits':28s} {100 * other_sweep.sum() / tot.sum():5.1f} %   mean {other_sweep.mean():7.0f} cycles")
            continue
        print(f"      {nm:28s} {100 * reg_c[:, k].sum() / tot.sum():5.1f} %   mean {reg_c[:, k].mean():7.0f} cycles")
    print(f"      {'(stamps cover)':28s} {100 * acc.sum() / tot.sum():5.1f} %")
# cycles against the wave's rank in the launch order (rank 0 = the widest source group of its job): what a split of
# the widest groups would have to cover
t_all = tr[:, 0].astype(np.float64)
rank_of_group = {}
print(f"all waves: mean {t_all[ok].mean():.0f} p50 {np.percentile(t_all[ok], 50):.0f} p90 {np.percentile(t_all[ok], 90):.0f} "
      f"p99 {np.percentile(t_all[ok], 99):.0f} p99.9 {np.percentile(t_all[ok], 99.9):.0f} max {t_all[ok].max():.0f}")
top = np.argsort(-t_all)[:12]
print("slowest waves (cycles, group, job, candidate chunks, chunks, items):", [(int(t_all[i]), int(tr[i, 22]), int(job[i]), int(n_cand[i]), int(tr[i, 2]), int(tr[i, 4])) for i in top])
# how concentrated the work is: share of all cycles in the heaviest x % of the waves; what a split by work would cover
srt = np.sort(t_all[ok])[::-1]
cs = np.cumsum(srt) / srt.sum()
for pc in (0.1, 0.5, 1, 2, 5, 10):
    k = max(1, int(len(srt) * pc / 100))
    print(f"   heaviest {pc:4.1f} % of waves ({k}): {100 * cs[k - 1]:.1f} % of the cycles, lightest of them {srt[k - 1]:.0f} cycles")
# items as a predictor of cycles (what a split decision could use): items of the same (job, group) -> cycles
A = np.stack([n_cand[ok], tr[ok, 2], tr[ok, 4], np.ones(ok.sum())], 1).astype(np.float64)
coef = np.linalg.lstsq(A, t_all[ok], rcond=None)[0]
print("cycles ~ %.0f x candidate + %.0f x processed + %.1f x item + %.0f" % tuple(coef))
print("candidate chunks per wave: mean", n_cand[ok].mean(), "p99", np.percentile(n_cand[ok], 99), "max", n_cand[ok].max(), "; share of waves with > 32:", (n_cand[ok] > 32).mean())
h = np.stack([(rh >> (8 * q)) & 255 for q in range(4)], 1)[ok]
print("rounds by occupancy (<=16, <=32, <=48, <=64 items):", h.sum(0), "per wave", np.round(h.mean(0), 2))
print("contested sources (tie path) per wave:", n_ties[ok].mean(), "; waves with any:", (n_ties[ok] > 0).mean(), "; per source:", n_ties[ok].sum() / (128.0 * ok.sum()))
# per XCD: when its waves started / ended (100 MHz clock shared by the XCDs) -- is the launch waiting for one of them?
t0 = tr[ok, 23].astype(np.int64); t1 = tr[ok, 24].astype(np.int64)
base = t0.min()
st = ((t0 - base) & 0xFFFFFFFF) / 100.0
en = ((t1 - base) & 0xFFFFFFFF) / 100.0
xcc = tr[ok, 25] & 15
print(f"launch span {en.max():.1f} us (trace build); per XCD:")
for x in range(8):
    m = xcc == x
    if m.any():
        jj = job[ok][m]
        print(f"  XCD {x}: waves {m.sum():6d}  last start {st[m].max():7.1f}  last end {en[m].max():7.1f}  wave-time {(en[m] - st[m]).sum() / 1e3:7.1f} ms  jobs {len(set(jj.tolist()))}  other-world among them {len(set(j for j in jj.tolist() if cand_dist[j] < 0))}")
