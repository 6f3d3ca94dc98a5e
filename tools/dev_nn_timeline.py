"""dev: when the waves of the LAST culled 1-NN launch of one query alone (20 jobs) started and ended, per XCD: what the
launch waits for.  usage: dev_nn_timeline.py [helpers thresh [job_group]]"""
import sys, os, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import bench
from gloc3d_amd import capi, synth
W = 32
helpers = int(sys.argv[1]) if len(sys.argv) > 1 else 0
thresh = int(sys.argv[2]) if len(sys.argv) > 2 else 60000
va, vb, vq, vf, far_poses = bench.build_views("/tmp/views.npz")
store = capi.ScanStore()
base_a = [store.add(v) for v in va]
base_b = [store.add(v) for v in vb]
cands = [store.add_variant(base_b[(g // 4) % len(base_b)] if g % 4 == 1 else base_a[g % len(base_a)], bench.place_perturbation(g), 0.01, 7000 + g)
         for g in range(20)]
store.build_target_index_batch(cands)
prm = capi.default_reg_params(ransac_iters=bench.RANSAC_ITERS, icp_iters=bench.ICP_ITERS, min_inlier_ratio=bench.MIN_INLIER_RATIO, max_rmse=bench.MAX_RMSE)
reg = capi.Registrar(store=store)
reg.set_option(capi.REG_OPT_NN_SPLIT_HELPERS, helpers)
reg.set_option(capi.REG_OPT_NN_SPLIT_THRESH, thresh)
if len(sys.argv) > 3:
    reg.set_option(capi.REG_OPT_NN_JOB_GROUP, int(sys.argv[3]))
if len(sys.argv) > 4:
    reg.set_option(capi.REG_OPT_NN_SUB_JOBS, int(sys.argv[4]))
L = capi.lib(); f = L.gloc_reg_debug_trace; f.restype = C.c_int
f.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_size_t, C.POINTER(C.c_size_t)]
sid = store.add(vq[0])
reg.batch_multi([sid], [cands], params=prm)          # warm
f(reg._h, 1, None, 0, None)
reg.batch_multi([sid], [cands], params=prm)
n = C.c_size_t(); f(reg._h, 1, None, 0, C.byref(n))
tr = np.zeros((n.value, W), np.uint32); f(reg._h, 1, tr.ctypes.data_as(C.c_void_p), n.value, C.byref(n))
ok = tr[:, 0] > 0
t = tr[ok]
t0 = t[:, 23].astype(np.int64); t1 = t[:, 24].astype(np.int64)
base = t0.min()
st = ((t0 - base) & 0xFFFFFFFF) / 100.0   # us (s_memrealtime: 100 MHz)
en = ((t1 - base) & 0xFFFFFFFF) / 100.0
xcc = t[:, 25] & 15
print(f"plan: helpers {helpers} thresh {thresh}; traced waves {ok.sum()} of {len(tr)} slots; span {en.max():.1f} us (trace build: slower than production)")
print("waves by parts:", {int(p): int(((t[:, 22] >> 24) == p).sum()) for p in np.unique(t[:, 22] >> 24)})
for x in range(8):
    m = xcc == x
    if m.any():
        print(f"  XCD {x}: waves {m.sum():5d}  first start {st[m].min():6.1f}  last start {st[m].max():6.1f}  last end {en[m].max():6.1f}  sum of durations {(en[m] - st[m]).sum() / 1e3:7.1f} ms  jobs {sorted(set((t[m, 6]).tolist()))}")
# occupancy over time
edges = np.linspace(0, en.max(), 21)
occ = [((st < b) & (en > a)).sum() for a, b in zip(edges[:-1], edges[1:])]
print("waves alive per 5 % slice of the launch:", occ)
last = np.argsort(-en)[:15]
print("last waves to end (end us, start us, duration us, job, rank, part/parts, cand chunks, chunks, items):")
for i in last:
    print(f"   {en[i]:6.1f} {st[i]:6.1f} {en[i] - st[i]:6.1f}  job {t[i, 6]:3d} rank {t[i, 22] & 0xFFFFF:4d} part {(t[i, 22] >> 20) & 15}/{t[i, 22] >> 24}  {t[i, 3] >> 16:4d} {t[i, 2]:3d} {t[i, 4]:5d}")
dur = en - st
print(f"durations us: mean {dur.mean():.1f} p50 {np.percentile(dur, 50):.1f} p99 {np.percentile(dur, 99):.1f} max {dur.max():.1f}; started after 50 % of the span: {(st > 0.5 * en.max()).mean():.3f}")
