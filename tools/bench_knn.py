"""kNN-only benchmark (BASELINE.json configs[1]: 64 queries x 10k DB x 4096-D fp32), device resident.
Not the bench.py line: a stage benchmark for DESIGN.md and rocprof."""
import argparse, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from gloc3d_amd import capi, synth

ap = argparse.ArgumentParser()
ap.add_argument("--n", type=int, default=10000); ap.add_argument("--d", type=int, default=4096)
ap.add_argument("--q", type=int, default=64); ap.add_argument("--k", type=int, default=20)
ap.add_argument("--algo", type=int, default=0); ap.add_argument("--reps", type=int, default=200)
ap.add_argument("--kind", type=int, default=0)
a = ap.parse_args()
ix = capi.KnnIndex(a.d); ix.set_option(capi.KNN_OPT_ALGO, a.algo)
ix.add_synthetic(a.kind, 2001, 0, a.n)
q = torch.from_numpy(synth.descriptors_iid(2002, 0, a.q, a.d) if a.kind == 0 else
                     synth.queries_near(2001, (np.arange(a.q) * 67 + 5) % a.n, a.d)).cuda()
idx = torch.empty((a.q, a.k), dtype=torch.int64, device="cuda"); d2 = torch.empty((a.q, a.k), dtype=torch.float32, device="cuda")
for _ in range(10): ix.search_device(q.data_ptr(), a.q, a.k, idx.data_ptr(), d2.data_ptr())
ix.synchronize(); t = time.time()
for _ in range(a.reps): ix.search_device(q.data_ptr(), a.q, a.k, idx.data_ptr(), d2.data_ptr())
ix.synchronize(); dt = (time.time() - t) / a.reps
flop = 2.0 * a.q * a.n * a.d; byt = 4.0 * (a.n * a.d + a.q * a.d)
print(f"kNN {a.q}x{a.n}x{a.d} k={a.k} algo={a.algo}: {dt*1e6:.1f} us/search (wall, device resident) -> {a.q/dt:.0f} queries/s; "
      f"whole-search {flop/dt/1e12:.1f} TFLOP/s, {byt/dt/1e12:.2f} TB/s; stats {ix.stats()}")
if os.environ.get("GLOC3D_KNN_PROF"):  # dev: per-stage device time (HIP events around the stages, below the C ABI)
    ix.set_option(capi.KNN_OPT_PROFILE, 1); ix.profile_reset()
    for _ in range(10): ix.search_device(q.data_ptr(), a.q, a.k, idx.data_ptr(), d2.data_ptr())
    print("stage us:", {n: round(ix.profile(n)[0] / 10 * 1e3, 1) for n in ("norms", "dist_mfma", "dist_exact", "select", "select_rerank", "rerank", "finalize")})
if os.environ.get("GLOC3D_KNN_TRACE"):  # dev: phase stamps of the fused select + re-rank kernel (s_memtime ticks)
    import ctypes as C
    L = capi.lib(); f = L.gloc_knn_debug_trace; f.restype = C.c_int
    f.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_size_t]
    f(ix._h, a.q, None, 0)
    ix.search_device(q.data_ptr(), a.q, a.k, idx.data_ptr(), d2.data_ptr()); ix.synchronize()
    tr = np.zeros((a.q, 16), np.uint64); f(ix._h, 0, tr.ctypes.data_as(C.c_void_p), a.q * 16)
    t = tr[:, :6].astype(np.int64); d = np.diff(t, axis=1)
    s4 = tr[:, 8:12].astype(np.int64) - t[:, :1]
    print("inside select (ticks from kernel start, mean): keys ready %d, minima sorted %d, tournament done %d, collected %d, end %d" % (*s4.mean(0), d[:, 0].mean()))
    print("phase ticks (mean over queries): select %d, prefix+stage %d, R1 group sums %d, R2 chains %d, rank %d | m mean %.1f | first start -> last end %d"
          % (*d.mean(0), tr[:, 6].mean(), t[:, 5].max() - t[:, 0].min()))
