"""Developer probe: registration throughput with K queries in flight (K Registrar handles, K host threads)."""
import sys, os, time, threading
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from gloc3d_amd import capi, synth
import bench
pool, qscans = bench.build_scans(bench.SCAN_POOL, bench.QUERY_POOL)
params = capi.default_reg_params(ransac_iters=bench.RANSAC_ITERS, icp_iters=bench.ICP_ITERS)
for K in (1, 2, 3):
    regs = [capi.Registrar() for _ in range(K)]
    ids = [[r.scan_upload(p) for p in pool] for r in regs]
    qids = [[r.scan_upload(q) for q in qscans] for r in regs]
    def work(k, i):
        cands = [ids[k][(i * 7 + c) % bench.SCAN_POOL] for c in range(20)]
        regs[k].batch_ids(qids[k][i % bench.QUERY_POOL], cands, params=params)
    for i in range(2):
        for k in range(K): work(k, i)
    n = 12
    t = time.time()
    for i in range(n):
        th = [threading.Thread(target=work, args=(k, i * K + k)) for k in range(K)]
        for x in th: x.start()
        for x in th: x.join()
    dt = time.time() - t
    print(f"in flight {K}: {dt / (n * K) * 1e3:.2f} ms per registration ({n * K / dt:.1f} /s)")
    for r in regs: r.close()
