"""Held-out check of the convergence check gloc_reg_params.max_final_step.

The suggested value (GLOC_REG_FINAL_STEP_SUGGESTED = 0.03 m, what bench.py passes) was chosen in round 4 by sweeping
thresholds over bench.py's own legs (world seeds 1001 / 2002, the views of bench.build_views).  This module builds data
that sweep never saw -- other worlds (seeds 3003 and 4004), other sensor poses, other perturbation streams -- and runs the
reference's order on it: the coarse 2-D match of every (query, candidate) pair decides which candidates are registered and
seeds them (loop_detector.cpp:192-288 then icp_match_3d), RANSAC 3000 adaptive + ICP 20, the check OFF; the final steps
are read back and every threshold is then applied on the host.  Candidates of a query, in the order a retrieval would
rank them (by distance, the other world's interleaved): its own place re-cast 0.7 m away, the same place perturbed as
SURVEY cfg C, places 3 ... 21 m along the drive, four views of the other world.

Used by tools/dev_gate_holdout.py, bench.py (legs.gate_holdout) and tests/test_gate_holdout_gpu.py.
"""
import os

import numpy as np

WORLD_A, WORLD_B = 3003, 4004          # never used by bench.py, tools/dev_gate_stats.py or any test
N_PLACES, N_OTHER = 8, 4
THRESHOLDS = (0.025, 0.03, 0.034, 0.04, 0.05)


def place_pose(k):
    from gloc3d_amd import synth
    return synth.se3(9.0 * k - 20.0, (3.0 * k - 10.0, 0.45 * k - 1.0, 0.0))      # 3 m apart, turning 9 deg a place


def query_pose(k):
    from gloc3d_amd import synth
    return place_pose(k) @ synth.se3(4.0, (0.6, -0.4, 0.03))


def other_pose(k):
    from gloc3d_amd import synth
    return synth.se3(31.0 * k, (2.2 * k - 3.0, 1.1 * k, 0.0))


def _cast(job):
    from gloc3d_amd import synth
    world_seed, T, seed = job
    return np.ascontiguousarray(synth.lidar_scan(synth.make_world(world_seed), T, seed=seed)[:, :3])


def build_views(cache=None, workers=0):
    """Ray-cast the 20 views on the host (numpy).  workers > 0: a process pool -- only BEFORE anything touches the GPU."""
    n_all = 2 * N_PLACES + N_OTHER
    if cache and os.path.exists(cache):
        z = np.load(cache)
        if len(z.files) == n_all:
            v = [z[f"v{i}"] for i in range(n_all)]
            return v[:N_PLACES], v[N_PLACES:2 * N_PLACES], v[2 * N_PLACES:]
    jobs = [(WORLD_A, place_pose(k), 61000 + k) for k in range(N_PLACES)]
    jobs += [(WORLD_A, query_pose(k), 62000 + k) for k in range(N_PLACES)]
    jobs += [(WORLD_B, other_pose(k), 63000 + k) for k in range(N_OTHER)]
    if workers > 0:
        from concurrent.futures import ProcessPoolExecutor
        with ProcessPoolExecutor(max_workers=workers) as ex:
            v = list(ex.map(_cast, jobs))
    else:
        v = [_cast(j) for j in jobs]
    if cache:
        tmp = f"{cache}.{os.getpid()}.tmp"
        with open(tmp, "wb") as f:
            np.savez(f, **{f"v{i}": x for i, x in enumerate(v)})
        os.replace(tmp, cache)
    return v[:N_PLACES], v[N_PLACES:2 * N_PLACES], v[2 * N_PLACES:]


def _perturbation(i):
    """SURVEY 8d cfg C's candidate perturbation (yaw U(-10, 10) deg, t U(-2, 2)^2 x U(-0.2, 0.2) m), its own stream."""
    from gloc3d_amd import synth
    u = synth.rng_uniform(synth.rng_key(0x6A7E, np.uint64(i)), np.arange(4, dtype=np.uint64)).astype(np.float64) * 2 - 1
    return synth.se3(10.0 * u[0], (2.0 * u[1], 2.0 * u[2], 0.2 * u[3]))


def run(views, device=0, ransac_iters=3000, icp_iters=20, min_inlier_ratio=0.3, thresholds=THRESHOLDS, coarse=True):
    """-> dict: per threshold the outcome of the first-success rule over each query's ranked candidates, and the final
    steps of right / wrong poses among the registrations the inlier test accepts."""
    from gloc3d_amd import capi, loop_detector as ld
    places, queries, others = views
    store = capi.ScanStore(device=device)
    reg = capi.Registrar(device=device, store=store)
    cm = capi.CoarseMatcher(device) if coarse else None
    prm = capi.default_reg_params(ransac_iters=ransac_iters, icp_iters=icp_iters, min_inlier_ratio=min_inlier_ratio,
                                  max_rmse=0.0, max_final_step=0.0)
    # the candidate pool: every place as cast, every place perturbed (distinct points), the other world
    cand, pose, kind = [], [], []
    for k in range(N_PLACES):
        cand.append(store.add(places[k])); pose.append(place_pose(k)); kind.append(("place", k))
    for k in range(N_PLACES):
        P = _perturbation(k)
        cand.append(store.add_variant(cand[k], P, 0.01, 64000 + k)); pose.append(place_pose(k) @ np.linalg.inv(P)); kind.append(("perturbed", k))
    for k in range(N_OTHER):
        cand.append(store.add(others[k])); pose.append(None); kind.append(("other-world", k))
    store.build_target_index_batch(cand)
    grids = [cm.add_store_scan(store, c) for c in cand] if coarse else None
    rows = []          # (query, rank, kind, dist_m, coarse_ok, inlier_ok, right, err_m, err_deg, final_step)
    per_query = []
    for v in range(N_PLACES):
        qs = store.add(queries[v])
        Tq = query_pose(v)
        # retrieval order: same-world candidates by distance from the query, an other-world view after every third
        d = [np.linalg.norm(p[:3, 3] - Tq[:3, 3]) if p is not None else np.inf for p in pose]
        same = sorted([i for i in range(len(cand)) if pose[i] is not None], key=lambda i: d[i])
        oth = [i for i in range(len(cand)) if pose[i] is None]
        order = []
        for j, i in enumerate(same):
            order.append(i)
            if j % 3 == 0 and oth:
                order.append(oth.pop(0))
        order += oth
        ids = [cand[i] for i in order]
        init = np.tile(np.eye(4, dtype=np.float32), (len(ids), 1, 1))
        ok2 = np.ones(len(ids), bool)
        if coarse:
            qg = cm.add_store_scan(store, qs)
            xyyaw, _, ok2 = cm.match(qg, [grids[i] for i in order])
            for r in range(len(ids)):
                if ok2[r]:
                    c, s = np.cos(xyyaw[r, 2]), np.sin(xyyaw[r, 2])
                    init[r, :2, :2] = [[c, -s], [s, c]]
                    init[r, :2, 3] = xyyaw[r, :2]
        out = reg.batch_multi([qs], np.array([ids], np.uint32), params=prm, init_T=init[None])
        steps = reg.final_steps(len(ids))
        q_rows = []
        for r, i in enumerate(order):
            if pose[i] is None:
                right, ep, er = False, float("inf"), float("inf")
            else:
                er, ep = ld.pose_error(np.linalg.inv(pose[i]) @ Tq, out["T"][0, r])
                right = ep < 1.0 and er < 5.0
            row = (v, r, kind[i][0], float(d[i]), bool(ok2[r]), bool(out["ok"][0, r]), bool(right), float(ep), float(er), float(steps[r]))
            rows.append(row)
            q_rows.append(row)
        per_query.append(q_rows)
        store.release(qs)
    res = {"queries": N_PLACES, "candidates_per_query": len(cand), "worlds": [WORLD_A, WORLD_B], "coarse_seeded": bool(coarse),
           "thresholds": {}}
    acc = [r for r in rows if r[4] and r[5]]                      # registered (2-D match ok) and accepted by the inlier test
    right_steps = [r[9] for r in acc if r[6]]
    wrong_steps = [r[9] for r in acc if not r[6]]
    res["accepted_by_inlier_test"] = len(acc)
    res["right_pose_final_step_max"] = max(right_steps) if right_steps else None
    res["right_pose_final_step_p95"] = float(np.percentile(right_steps, 95)) if right_steps else None
    res["wrong_pose_final_step_min"] = min(wrong_steps) if wrong_steps else None
    res["right_poses"] = len(right_steps)
    res["wrong_poses"] = len(wrong_steps)
    for thr in (0.0,) + tuple(thresholds):
        succ = wrong = none = 0
        for q_rows in per_query:
            sel = next((r for r in q_rows if r[4] and r[5] and (thr <= 0 or r[9] <= thr)), None)   # first success in rank order
            if sel is None:
                none += 1
            elif sel[6]:
                succ += 1
            else:
                wrong += 1
        res["thresholds"]["off" if thr <= 0 else f"{thr:g}"] = {
            "success": succ, "located_but_wrong": wrong, "not_located": none,
            "right_poses_rejected": sum(1 for s in right_steps if thr > 0 and s > thr),
            "wrong_poses_accepted": sum(1 for s in wrong_steps if thr <= 0 or s <= thr)}
    res["rows"] = rows
    if cm is not None:
        cm.close()
    reg.close()
    store.close()
    return res
