"""dev: random SMALL batches through the chained launch (GLOC_REG_OPT_NN_CHAIN, the default) against the launch-by-launch
pipeline: every pose, rmse, inlier count, ok flag, final step and correspondence bit must agree, no device-side wait may run
out.  Random numbers of queries and candidates (1 .. 47 jobs), sizes from 3 points to 60 000, clouds with exact ties, NaN
points, empty cells, RANSAC on / off, 2 .. 9 ICP passes, correspondence gates, split thresholds / helper slots / shares of a
job / job groups.  usage: fuzz_chain.py [cases] [seed]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from gloc3d_amd import capi, synth

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 40
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 11)


def cloud(n, kind):
    if kind == 0:
        p = rng.uniform(-40, 40, (n, 3)) * np.array([1, 1, 0.1])
    elif kind == 1:   # ground plane + walls + clutter
        p = np.concatenate([np.c_[rng.uniform(-50, 50, (n // 2, 2)), rng.normal(0, 0.02, n // 2)],
                            np.c_[rng.uniform(-50, 50, n // 4), np.full(n // 4, 12.0), rng.uniform(0, 4, n // 4)],
                            rng.normal(0, 6, (n - n // 2 - n // 4, 3))])
    elif kind == 2:   # two far clusters
        p = np.concatenate([rng.normal(0, 1.5, (n // 2, 3)) + [60, 0, 0], rng.normal(0, 1.5, (n - n // 2, 3)) - [60, 10, 0]])
    else:             # lattice: exact ties everywhere
        g = int(np.ceil(n ** (1 / 3)))
        p = np.stack(np.meshgrid(*[np.arange(g)] * 3, indexing="ij"), -1).reshape(-1, 3)[:n] * 0.5
    return np.ascontiguousarray(p, np.float32)


bits = lambda a: np.ascontiguousarray(a, np.float32).view(np.uint32)
bad = n_chained = 0
t0 = time.time()
for c in range(cases):
    store = capi.ScanStore()
    n_q = int(rng.choice([1, 1, 2, 3]))
    n_c = int(rng.integers(1, 47 // n_q + 1))
    kind = int(rng.integers(0, 4))
    base = cloud(int(rng.choice([300, 2000, 9000, 30000, 60000])), kind)
    tg = []
    for _ in range(min(n_c, 6)):   # a few distinct targets, reused
        sel = rng.random(len(base)) < rng.uniform(0.3, 1.0)
        t = base[sel] if sel.sum() >= 3 else base[:3]
        if rng.random() < 0.2:
            t = cloud(int(rng.choice([3, 40, 3000])), int(rng.integers(0, 4)))   # an unrelated scene
        tg.append(store.add(np.ascontiguousarray(t)))
    if rng.random() < 0.6:
        store.build_target_index_batch(tg[: max(1, len(tg) // 2)])
    qs = []
    for _ in range(n_q):
        n_s = int(rng.choice([3, 64, 129, 1000, 5000, 20000, 60000]))
        T = synth.se3(float(rng.uniform(-6, 6)), (float(rng.uniform(-1, 1)), float(rng.uniform(-1, 1)), float(rng.uniform(-0.1, 0.1))))
        sel = rng.choice(len(base), min(n_s, len(base)), replace=False)
        s = (base[sel] @ T[:3, :3].T + T[:3, 3] + (rng.normal(0, 0.01, (len(sel), 3)) if kind != 3 else 0)).astype(np.float32)
        if rng.random() < 0.3 and len(s) > 10:
            s[rng.integers(0, len(s), 3)] = np.nan
        qs.append(store.add(np.ascontiguousarray(s)))
    grid = np.array([[tg[int(rng.integers(0, len(tg)))] for _ in range(n_c)] for _ in range(n_q)], np.uint32)
    if rng.random() < 0.3:
        grid[int(rng.integers(0, n_q)), int(rng.integers(0, n_c))] = capi.NO_SCAN
    prm = capi.default_reg_params(ransac_iters=int(rng.choice([0, 64, 300])), icp_iters=int(rng.integers(2, 10)),
                                  max_corr_dist=float(rng.choice([0.0, 0.0, 1.5])), max_final_step=0.0)
    opts = [(capi.REG_OPT_NN_SPLIT_THRESH, int(rng.choice([15000, 40000, 60000, 85000]))),
            (capi.REG_OPT_NN_SPLIT_HELPERS, int(rng.choice([-1, 32, 64, 256]))),
            (capi.REG_OPT_NN_SUB_JOBS, int(rng.choice([0, 0, 1, 2, 4]))),
            (capi.REG_OPT_NN_JOB_GROUP, int(rng.choice([24, 24, 8, 16, 48])))]
    res = {}
    for chain in (0, 1):
        r = capi.Registrar(store=store)
        r.set_option(capi.REG_OPT_NN_CHAIN, chain)
        for o, v in opts:
            r.set_option(o, v)
        out = r.batch_multi(qs, grid, params=prm)
        n_real = int((grid != capi.NO_SCAN).sum())
        out["steps"] = r.final_steps(n_real)
        corr = [r.debug_corr(j, 3) for j in range(n_real)]   # (the first three sources of every job: indices + distances)
        res[chain] = (out, corr, r.debug_chain())
        r.close()
    (a, ca, _), (b, cb, (launches, timeouts)) = res[0], res[1]
    n_chained += launches
    same = timeouts == 0 and all((bits(a[k]) == bits(b[k])).all() for k in ("T", "rmse", "steps")) and (a["inliers"] == b["inliers"]).all() \
        and (a["ok"] == b["ok"]).all() and all((x[0] == y[0]).all() and (bits(x[1]) == bits(y[1])).all() for x, y in zip(ca, cb))
    if not same:
        bad += 1
        print(f"MISMATCH case {c}: queries {n_q} x candidates {n_c} kind {kind} ransac {prm.ransac_iters} icp {prm.icp_iters} opts {opts} "
              f"chained launches {launches} timed out {timeouts}", flush=True)
    store.close()
    if c % 10 == 9:
        print(f"{c + 1} cases, {bad} mismatches, {n_chained} chained launches, {time.time() - t0:.0f} s; last: {n_q} x {n_c} jobs, icp {prm.icp_iters}", flush=True)
print("chained launches:", n_chained)
print("mismatches:", bad)
sys.exit(1 if bad or not n_chained else 0)
