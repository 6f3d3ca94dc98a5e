"""dev: random scan pairs through the culled exact 1-NN search (the default: nn_compact.hpp) against the exhaustive kernel
at the same pose, after a random number of ICP passes: every correspondence and every distance bit must agree.  Random sizes
(down to a handful of points), random clouds (uniform boxes, planes + clutter, two far clusters, lattices with exact ties),
NaN points, duplicated points, kd- and curve-ordered targets, random initial transforms.  usage: fuzz_reg.py [cases] [seed]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from gloc3d_amd import capi, synth

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 60
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 3)


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


bad = 0
t0 = time.time()
for c in range(cases):
    n_t = int(rng.choice([3, 17, 130, 1000, 5000, 20000, 60000]))
    n_s = int(rng.choice([3, 64, 129, 1000, 5000, 20000, 60000]))
    kind = int(rng.integers(0, 4))
    tgt = cloud(n_t, kind)
    T = synth.se3(float(rng.uniform(-8, 8)), (float(rng.uniform(-1.5, 1.5)), float(rng.uniform(-1.5, 1.5)), float(rng.uniform(-0.1, 0.1))))
    if rng.random() < 0.6 and n_s <= n_t:      # the source: a moved, noisy part of the target
        sel = rng.choice(n_t, n_s, replace=False)
        src = (tgt[sel] @ T[:3, :3].T + T[:3, 3] + (rng.normal(0, 0.01, (n_s, 3)) if kind != 3 else 0)).astype(np.float32)
    else:
        src = cloud(n_s, int(rng.integers(0, 4)))
    if rng.random() < 0.3 and n_s > 10:
        src[rng.integers(0, n_s, 3)] = np.nan
    if rng.random() < 0.3 and n_t > 10:
        tgt[rng.integers(0, n_t, 3)] = tgt[rng.integers(0, n_t, 3)]
    iters = int(rng.integers(1, 7))
    ransac = int(rng.choice([0, 0, 64]))   # with the RANSAC stage in front: the first pass is the kernel that also writes the pairs
    store = capi.ScanStore()
    qid, cid = store.add(src), store.add(tgt)
    if rng.random() < 0.5:
        store.build_target_index_batch([cid])
    res = {}
    cs = int(rng.choice([1, 2, 2, 2, 4]))   # (one choice per case: the wave grouping of the fp32 moments is part of the pose's bits)
    for mode in (capi.REG_NN_CULLED, capi.REG_NN_EXHAUSTIVE):
        def run(it, init_T=None, m=mode):
            r = capi.Registrar(store=store)
            r.set_option(capi.REG_OPT_NN_MODE, m)
            if m == capi.REG_NN_CULLED:
                r.set_option(capi.REG_OPT_NN_SRC_PER_LANE, cs)
            out = r.batch_ids(qid, [cid], init_T=init_T,
                              params=capi.default_reg_params(ransac_iters=ransac if m == capi.REG_NN_CULLED else 0, icp_iters=it, max_final_step=0.0))
            corr = r.debug_corr(0, n_s)
            r.close()
            return out, corr
        if mode == capi.REG_NN_CULLED:
            T_prev = np.eye(4, dtype=np.float32)[None] if (iters == 1 and not ransac) else run(iters - 1)[0]["T"]
            res[mode] = run(iters)[1]
        else:
            res[mode] = run(1, init_T=T_prev)[1]
    a, b = res[capi.REG_NN_CULLED], res[capi.REG_NN_EXHAUSTIVE]
    same = (a[0] == b[0]).all() and (a[1].view(np.uint32) == b[1].view(np.uint32)).all()
    if not same:
        bad += 1
        print(f"MISMATCH case {c}: n_s {n_s} n_t {n_t} kind {kind} passes {iters} ransac {ransac}: {int((a[0] != b[0]).sum())} indices, "
              f"{int((a[1].view(np.uint32) != b[1].view(np.uint32)).sum())} distances differ", flush=True)
    store.close()
    if c % 10 == 9:
        print(f"{c + 1} cases, {bad} mismatches, {time.time() - t0:.0f} s; last: n_s {n_s} n_t {n_t} kind {kind} passes {iters} ransac {ransac}", flush=True)
print("mismatches:", bad)
sys.exit(1 if bad else 0)
