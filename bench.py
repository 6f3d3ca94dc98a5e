#!/usr/bin/env python3
"""bench.py -- localization queries/sec (kNN + top-20 registration) on MI355X.

    python bench.py --gpus N --steps K --warmup W        (N > 1: launches its own N ranks, see below)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

One STEP = one batch of `--batch` (default 25) localization queries per GPU through the hot path, as
SURVEY.md 8d config D states it -- nothing about a query is on the device before its step starts:
  fresh query scan H2D + Hilbert index (gloc_scan_store_add)            } inside the timed region,
  query descriptor H2D -> descriptor top-20 (gloc_knn_search_device)    } prefetched one step ahead
  -> the 20 retrieved places' scans from the RESIDENT store of 4541 distinct scans (20 GB of HBM)
  -> batched RANSAC(3000, adaptive) + ICP(20) registration of all B x 20 (query, candidate) jobs in one
     launch sequence (gloc_reg_batch_multi) -> lowest-rank successful candidate -> query scan released.
5 of every 20 consecutive places carry a scan of a DIFFERENT world (true negatives, SURVEY cfg C).
The K timed steps are repeated `--reps` (5) times; `value` is the median repetition.

N = 1  BASELINE.json configs[3]: KITTI-00-sized database, 4541 places x 4096-D, ~123k-point scans;
       20 steps x 25 queries = the 500-query stream.
N > 1  the same database, interleave-sharded over the N ranks; a step handles N x B queries: their
       descriptors are searched on every shard, the per-shard top-k lists all-gathered over RCCL/xGMI
       and merged on every rank; rank r then registers its B queries against its replica of the scan
       store, and the result tables are all-gathered.  Per-GPU work is fixed as N grows ("weak").
       --places 1000000 gives BASELINE.json configs[4]'s sharded database; --mode latency shards ONE
       query's candidates over the ranks instead.  Without torchrun's environment `--gpus N` starts the N
       ranks itself (a child `python -m torch.distributed.run`, before anything here touches the GPU).

Prints ONE JSON line (rank 0):
  value / ms_per_step   the metric;
  accuracy              recall@1/5/10/20, registration success rate, position / rotation error of the stream
                        against its constructed ground truth, exactly as the reference's evaluator defines
                        them (registration/global_localization.cpp:221-268, 270-335); the run FAILS (exit 3)
                        when the success rate falls below --min-success;
  roofline              the dominant kernel (K4 point-NN), HIP-event timed inside the timed region;
  legs                  (N = 1) the same pipeline with each work-reducing choice switched off, and on harder
                        data: all 3000 RANSAC hypotheses scored; the brute-force 1-NN kernel; candidate poses
                        drawn as SURVEY cfg C writes them; candidates 5-20 m away; 96 distinct ray-cast poses
                        along a loop (data_loop_views); the convergence check on worlds it was not chosen on
                        (gate_holdout);
  top-level scalars     copies of the nested figures (knn_cfgB_us, lone_query_ms, knn_shard125k_q64_us,
                        success_rate, loop_views_qps ...): the driver's record keeps scalar keys only;
  sub_records           (N = 1) BASELINE configs[1] (kNN 64 x 10k x 4096), configs[2] (one query alone incl.
                        its preparation), one shard of configs[4] (kNN over 125k x 4096);
  cpu_baseline          the CPU checker on this box's host cores: ONE WHOLE query (kNN + its 20 candidates),
                        not an extrapolation; its 20 results also check the timed 500-job launch's rows.
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import threading
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

DIM = 4096
TOP_K = 20
N_PLACES_1GPU = 4541          # KITTI odometry 00 (dataset/kitti_i2i.py:46 of the reference)
TRAJ_LENGTH_M = 3724.0        # KITTI odometry 00: 4541 poses over 3724 m -> 0.82 m between consecutive places
WORLD_A_SEED = 1001           # the world the trajectory runs through (gloc3d_amd/synth.py::make_road_world) ...
WORLD_B_SEED = 2002           # ... and a different world along the same road: the scans of the negatives
PLACE_SEED, QUERY_SEED = 7000, 880000     # range-noise streams of the place / query casts (+ place id / stream id)
POOL_A = 24                   # legs.data_rigid_copies (rounds 1-5's data): ray-cast views of a small world along a 4.6 m drive,
POOL_B = 6                    # ... views of a different world (its negatives),
QUERY_VIEWS = 8               # ... query views, each next to pool view 3 * v + 1
NEG_EVERY = 4                 # place g carries a world-B scan iff g % 4 == 1 -> 5 of 20 consecutive places
RANSAC_ITERS = 3000           # registration/loop_detector.cpp:257 (cap; adaptive stop at the
                              # reference's OpenCV default confidence 0.99, see gloc_reg_params)
ICP_ITERS = 20                # BASELINE.json configs[2]
MIN_INLIER_RATIO = 0.3        # the library default (ok iff RANSAC inliers >= ratio x n) ...
MAX_RMSE = 0.0                # ... NO rmse gate (round 3 needed a hand-tuned 0.5 m), and ...
MAX_FINAL_STEP = 0.03         # ... the convergence check, passed EXPLICITLY (gloc_reg_params.max_final_step; the library's
                              # default is off since round 5, as the reference's 3-D stage has no such check): ok also
                              # requires that the last ICP update moved the matched points by no more than this, RMS.
                              # Why a check at all: both worlds share a ground plane, so a different-world candidate
                              # still has ~0.83 inliers at 0.6 m, and a same-world place ~4 m away that 20 ICP passes
                              # leave half-way was accepted 13 times in 500 -- ICPs that have not converged.
                              # WHERE THE VALUE COMES FROM: a sweep of 0.025 / 0.03 / 0.04 / 0.05 m over THIS bench's own
                              # legs in round 4 (LAB_NOTES.md) -- it is tuned on this data; = GLOC_REG_FINAL_STEP_SUGGESTED
                              # of include/gloc3d.h.  legs.gate_holdout runs it on worlds / views / poses the sweep never saw.
POSITIVE_RADIUS_M = 5.0       # SURVEY 8d cfg D: ground-truth positives = places within 5 m (dataset/kitti_i2i.py:94-95)
DB_SEED = 4001
PEAK_HBM_GBS = 8000.0         # MI355X_MICROARCH.md: HBM3E 8 TB/s spec (6.29 TB/s measured copy)
PEAK_FP32_TFLOPS = 157.3
PEAK_BF16_TFLOPS = 2500.0      # dense bf16 MFMA


def knn_roofline_s(n_rows, nq, dim):
    """kNN search, device resident: the rows stream from HBM once; the coarse pass takes three bf16 MFMAs per product
    (operands split in two bf16 values, round 4).  Also returned: the fp32-MFMA time rounds 1 - 3 priced against."""
    byts = 4.0 * (n_rows + nq) * dim
    flop = 2.0 * nq * n_rows * dim
    return max(byts / (PEAK_HBM_GBS * 1e9), 3.0 * flop / (PEAK_BF16_TFLOPS * 1e12)), flop / (PEAK_FP32_TFLOPS * 1e12)
FLOP_PER_PAIR = 8             # 3 sub + 3 mul + 2 add per (source, target) pair
BYTES_PER_POINT_INDEXED = 16  # sorted float4 (x, y, z, original index)
KNN_CFGB = (64, 10000)        # BASELINE.json configs[1]
KNN_SHARD_ROWS = 125000       # one of the 8 shards of configs[4]
PMC_FILES = ("r06_pmc_traffic_nn_compact.json",)     # counter passes over THIS round's kernel on THIS round's data only (earlier rounds ran rigid copies)


def log(msg):
    if int(os.environ.get("RANK", "0")) == 0:
        print(f"[bench] {msg}", file=sys.stderr, flush=True)


# ---- the synthetic world: poses are known, so every query has a ground truth ----------------------------

def pool_pose(s):
    from gloc3d_amd import synth
    return synth.se3(0.5 * (s - POOL_A / 2), (0.2 * s, 0.04 * s, 0.0))


def query_view_pose(v):
    from gloc3d_amd import synth
    return pool_pose(3 * v + 1) @ synth.se3(1.5, (0.3, -0.2, 0.02))


def place_perturbation(g):
    """Per-place rigid perturbation (yaw +-2 deg, t +-0.3 m) from the counter RNG: every place's scan
    is a distinct cloud (distinct bits, distinct Hilbert order), not an alias of a pool scan."""
    from gloc3d_amd import synth
    key = synth.rng_key(DB_SEED ^ 0x5CA4, np.uint64(g))
    u = synth.rng_uniform(key, np.arange(4, dtype=np.uint64)).astype(np.float64) * 2 - 1
    return synth.se3(2.0 * u[0], (0.3 * u[1], 0.3 * u[2], 0.03 * u[3]))


def query_perturbation(j):
    from gloc3d_amd import synth
    key = synth.rng_key(DB_SEED ^ 0x9E77, np.uint64(j))
    u = synth.rng_uniform(key, np.arange(4, dtype=np.uint64)).astype(np.float64) * 2 - 1
    return synth.se3(1.0 * u[0], (0.2 * u[1], 0.2 * u[2], 0.02 * u[3]))


def query_offset(j):
    """The headline's query j relative to its place's sensor pose: a revisit that is not on the database pose -- up to
    0.4 m along the road, 0.5 m across it, 3 deg of heading (counter RNG).  The query scan is its own ray-cast from there."""
    from gloc3d_amd import synth
    key = synth.rng_key(DB_SEED ^ 0x0FF5, np.uint64(j))
    u = synth.rng_uniform(key, np.arange(4, dtype=np.uint64)).astype(np.float64) * 2 - 1
    return synth.se3(3.0 * u[0], (0.4 * u[1], 0.5 * u[2], 0.02 * u[3]))


def headline_world(n_store):
    """SURVEY 8d cfg D's scans: n_store poses, equally spaced along a closed loop of TRAJ_LENGTH_M through ONE procedural
    world (world A); world B is a different scene along the same road (the negatives' scans).  -> (poses [n, 4, 4],
    world A, world B)."""
    from gloc3d_amd import synth
    traj, xy = synth.loop_trajectory(n_store, TRAJ_LENGTH_M * n_store / N_PLACES_1GPU)
    return traj, synth.make_road_world(WORLD_A_SEED, xy), synth.make_road_world(WORLD_B_SEED, xy)


FAR_AWAY = None


def far_away_pose():
    """The 'pose' of a different-world place in world A's frame: nowhere near -- selecting it is a failure."""
    global FAR_AWAY
    if FAR_AWAY is None:
        from gloc3d_amd import synth
        FAR_AWAY = synth.se3(0.0, (1.0e4, 1.0e4, 0.0))
    return FAR_AWAY


def accuracy_of(cands, sels, tables, q_ids, place_pose, query_pose, is_positive_place, recall_defined=True):
    """The reference evaluator's report for a stream (host mirror gloc3d_amd/loop_detector.py of
    GlocEvaluator::recognition_recalls / registration_recalls, global_localization.cpp:221-268, 270-335):
    cands [Q, k] retrieved place ids, sels [Q] selected retrieval rank (-1: none), tables [Q, k, 19]."""
    from gloc3d_amd import loop_detector as ld
    Q = len(q_ids)
    qpos = [query_pose(j)[:3, 3] for j in q_ids]
    gt_pos = []
    for qi in range(Q):     # ground-truth positives: the retrieved or not, every same-world place within 5 m
        gt_pos.append([int(g) for g in cands[qi] if g >= 0 and is_positive_place(int(g)) and
                       np.linalg.norm(place_pose(int(g))[:3, 3] - qpos[qi]) < POSITIVE_RADIUS_M])
    # (positives outside the retrieved list cannot change a first-hit recall: a hit needs a retrieved place)
    # a query whose retrieved list holds no positive is still a VALID query (the database has ~3400 places within
    # 5 m of it): it counts as a miss, not as "no ground truth" (which the reference skips, :226)
    rec, failed_detect = ld.recognition_recalls(cands, [p if p else [-2] for p in gt_pos])
    if not recall_defined:
        rec, failed_detect = [None] * 4, []
    er_all, ep_all, located = [], [], 0
    ok_rot, ok_pos, wrong, ok_rmse = [], [], [], []
    n_wrong = 0
    for qi in range(Q):
        r = int(sels[qi])
        if r < 0:
            continue
        located += 1
        g = int(cands[qi][r])
        T = np.asarray(tables[qi][r][:16], np.float32).reshape(4, 4)
        q2db = np.linalg.inv(place_pose(g)) @ query_pose(q_ids[qi])
        er, ep = ld.pose_error(q2db, T)
        er_all.append(er)
        ep_all.append(ep)
        if ep < 1.0 and er < 5.0:
            ok_rot.append(er)
            ok_pos.append(ep)
            ok_rmse.append(round(float(tables[qi][r][16]), 3))
        else:
            n_wrong += 1
        if not (ep < 1.0 and er < 5.0) and len(wrong) < 16:
            d_place = float(np.linalg.norm(place_pose(g)[:3, 3] - qpos[qi]))
            wrong.append({"query": int(q_ids[qi]), "rank": r, "place": g, "place_to_query_m": round(d_place, 2),
                          "err_pos_m": round(ep, 3), "err_rot_deg": round(er, 3), "rmse_m": round(float(tables[qi][r][16]), 3),
                          "inliers": int(tables[qi][r][17])})

    def mean_std(v):    # caculate_mean_std: n - 1 in the denominator (global_localization.cpp:185-196)
        if len(v) < 2:
            return (float(v[0]) if v else 0.0), 0.0
        return float(np.mean(v)), float(np.std(v, ddof=1))
    pm, ps = mean_std(ok_pos)
    rm, rs = mean_std(ok_rot)
    f_ = lambda v: None if v is None else float(v)
    return {"queries": Q, "recall_at_1": f_(rec[0]), "recall_at_5": f_(rec[1]), "recall_at_10": f_(rec[2]),
            "recall_at_20": f_(rec[3]), "failed_detect": len(failed_detect),
            "success_rate": len(ok_pos) / Q if Q else 0.0, "succeeded": len(ok_pos), "located": located,
            "not_located": Q - located, "pos_err_mean_m": pm, "pos_err_std_m": ps, "rot_err_mean_deg": rm,
            "rot_err_std_deg": rs, "pos_err_max_m_located": float(max(ep_all)) if ep_all else 0.0,
            "rmse_max_m_of_successes": max(ok_rmse) if ok_rmse else None,
            "located_but_wrong": wrong, "located_but_wrong_count": n_wrong,
            "definition": "recall@N: first hit among the top N (global_localization.cpp:221-268), positives = same-world "
                          f"places within {POSITIVE_RADIUS_M:g} m; success: err_pos < 1 m and err_rot < 5 deg against "
                          "pose_db^-1 pose_q, mean / std (n - 1) over the successes (:270-335)"}


# ---- the CPU checker: one whole query ---------------------------------------------------------------------

def cpu_baseline(q_scan, cand_scans, n_places, min_inlier_ratio, gpu_rows=None):
    """The CPU checker on this host, ONE WHOLE QUERY of the stream (no extrapolation): top-20 over the
    descriptor database + full registration of its 20 retrieved candidates (15 same-world, 5 different-world),
    on one thread (the reference's kNN and registration are single-threaded) and on all cores (candidates over
    threads).  With oracle/_ref present both searches are the reference's own vendored nanoflann kd-trees.
    gpu_rows: the 20 result rows the TIMED 500-job launch produced for this query -- checked against it."""
    import oracle
    from gloc3d_amd import synth
    oracle.build(ref=False)
    use_ref = oracle.have_ref()
    db = synth.descriptors_traj(DB_SEED, 0, n_places, DIM)
    q = synth.queries_near(DB_SEED, [1234], DIM)
    t_build = 0.0
    if use_ref:
        R = oracle.ref()
        t0 = time.time()
        tree = R.ref_knn_build(db, db.shape[0], db.shape[1])          # once per database, not per query
        t_build = time.time() - t0
        idx, d2 = np.empty((1, TOP_K), np.uint64), np.empty((1, TOP_K), np.float32)
        t0 = time.time()
        R.ref_knn_query(tree, q, 1, TOP_K, idx, d2)
        t_knn = time.time() - t0
        R.ref_knn_free(tree)
    else:
        t0 = time.time()
        oracle.knn_search(db, q, TOP_K)
        t_knn = time.time() - t0
    kw = dict(ransac_iters=RANSAC_ITERS, icp_iters=ICP_ITERS, min_inlier_ratio=min_inlier_ratio, max_rmse=MAX_RMSE, max_final_step=MAX_FINAL_STEP)
    t0 = time.time()
    res = [oracle.reg_one(q_scan, c, cand_id=i, ref_nn=use_ref, **kw) for i, c in enumerate(cand_scans)]
    t_reg = time.time() - t0
    per_query = t_knn + t_reg
    extra = {}
    if gpu_rows is not None:
        T = np.stack([r["T"] for r in res])
        g_T = np.asarray(gpu_rows[:, :16], np.float32).reshape(-1, 4, 4)
        extra["timed_launch_parity"] = {
            "what": f"rows of query 0 in the LAST timed repetition's first {len(cand_scans)}-candidate batch "
                    "(one gloc_reg_batch_multi of all queries in flight) against the CPU checker, candidate by candidate",
            "candidates": len(cand_scans),
            "pose_max_abs_diff": float(np.abs(g_T - T).max()),
            "inliers_equal": bool(all(int(gpu_rows[c, 17]) == int(res[c]["inliers"]) for c in range(len(res)))),
            "ok_equal": bool(all(bool(gpu_rows[c, 18] > 0.5) == bool(res[c]["ok"]) for c in range(len(res)))),
            "rmse_max_abs_diff": float(max(abs(float(gpu_rows[c, 16]) - res[c]["rmse"]) for c in range(len(res)))),
            "ok": [bool(r["ok"]) for r in res],
            "inlier_ratio": [round(float(r["inliers"]) / len(q_scan), 4) for r in res]}
    cores = os.cpu_count() or 1
    nthr = min(cores, len(cand_scans))
    t0 = time.time()
    oracle.reg_many_mt(q_scan, cand_scans, nthr, ref_nn=use_ref, cand_ids=np.arange(len(cand_scans), dtype=np.uint32), **kw)
    t_mt = time.time() - t0
    per_query_mt = t_knn + t_mt
    cpu_model = ""
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                cpu_model = line.split(":", 1)[1].strip()
                break
    except OSError:
        pass
    return {**extra, "value": 1.0 / per_query, "unit": "queries/s", "cores": 1,
            "kind": "reference" if use_ref else "port", "extrapolated": False,
            "nn_search": ("the reference's vendored nanoflann kd-trees (oracle/_ref): descriptors -- InvKeyTree, built once "
                          f"({t_build*1e3:.0f} ms, not counted), queried per query; 3-D points -- built once per candidate, "
                          "queried every pass, as PCL's ICP does") if use_ref else "the port's brute force / uniform grid",
            "sample": f"1 whole query: kNN over {n_places}x{DIM} ({t_knn*1e3:.1f} ms) + RANSAC{RANSAC_ITERS}+ICP{ICP_ITERS} "
                      f"registration of its {len(cand_scans)} retrieved candidates ({t_reg:.1f} s), ~123k-pt scans, measured, "
                      "not extrapolated",
            "all_cores": {"value": 1.0 / per_query_mt, "threads": nthr,
                          "sample": f"the same query, its {len(cand_scans)} candidates on {nthr} threads ({t_mt:.1f} s)"},
            "host_cpus": cores, "cpu_model": cpu_model,
            "compiler_flags": "gcc -O2 -ffp-contract=off (port), g++ -O3 -DNDEBUG -std=c++14 -ffp-contract=off (reference nanoflann)"}


# ---- the line the driver parses ---------------------------------------------------------------------------
FINAL_LINE_MAX = 6144        # bytes; BENCH_r05.parsed was null for a 23.6 KB line (the driver keeps an ~8 KB tail of stdout)
REQUIRED_KEYS = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                 "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline")


DETAIL_ONLY_KEYS = ("legs", "sub_records", "first_success_mode", "selected_candidate_rank_histogram")


def _scalars(d, keep_str=(), cut=160):
    """The int / float / bool / None entries of a dict, plus the named string entries (cut to `cut` characters)."""
    o = {}
    for k, v in (d or {}).items():
        if v is None or isinstance(v, (bool, int, float)):
            o[k] = v
        elif isinstance(v, str) and k in keep_str:
            o[k] = v if len(v) <= cut else v[:cut - 3] + "..."
    return o


def _clean(v):
    """Strict-JSON form of a record: numpy scalars become Python's, NaN / +-Infinity become null."""
    if isinstance(v, (np.floating, np.integer, np.bool_)):
        v = v.item()
    if isinstance(v, float) and (v != v or v in (float("inf"), float("-inf"))):
        return None
    if isinstance(v, dict):
        return {str(k): _clean(x) for k, x in v.items()}
    if isinstance(v, (list, tuple)):
        return [_clean(x) for x in v]
    if isinstance(v, np.ndarray):
        return _clean(v.tolist())
    return v


def compact_line(out, limit=FINAL_LINE_MAX, detail_file="bench_detail.json"):
    """The FINAL stdout line: the contract's keys, a short `config`, `roofline` and `cpu_baseline` without their prose and
    lists, the stage times and the top-level scalars -- nothing nested deeper than two levels, nothing that grows with the
    number of legs.  Everything else (legs, sub_records, accuracy, per-failure lists) is in the `[bench-detail] ` line printed
    BEFORE it and in `detail_file`.  Strict JSON (no NaN / Infinity), ASCII, at most `limit` bytes: groups are dropped, least
    important first, until it fits (tests/test_bench_line.py holds the shape)."""
    out = _clean(out)
    line = {k: out.get(k) for k in REQUIRED_KEYS[:12]}
    cfg = out.get("config") or {}
    line["config"] = _scalars(cfg, keep_str=("workload", "nn_mode", "collectives", "collectives_requested", "parallelism",
                                             "database_scans_target_index", "value_is"))
    if isinstance(cfg.get("workload"), str):
        line["config"]["workload"] = cfg["workload"] if len(cfg["workload"]) <= 300 else cfg["workload"][:297] + "..."
    rf = out.get("roofline")
    if rf:
        r = _scalars(rf, keep_str=("kernel", "bound", "unit"))
        if rf.get("issue_model"):
            r["issue_model"] = _scalars(rf["issue_model"])
        line["roofline"] = r
    else:
        line["roofline"] = None
    cb = out.get("cpu_baseline")
    if cb:
        c = _scalars(cb, keep_str=("unit", "kind", "sample", "cpu_model"), cut=200)
        if cb.get("all_cores"):
            c["all_cores_value"], c["all_cores_threads"] = cb["all_cores"].get("value"), cb["all_cores"].get("threads")
        if cb.get("timed_launch_parity"):
            c.update({"parity_" + k: v for k, v in _scalars(cb["timed_launch_parity"]).items()})
        line["cpu_baseline"] = c
    else:
        line["cpu_baseline"] = None
    line["stage_ms_per_step_rank0"] = _scalars(out.get("stage_ms_per_step_rank0"))
    acc = out.get("accuracy") or {}
    line["accuracy"] = {**_scalars(acc), "located_but_wrong": acc.get("located_but_wrong_count", len(acc.get("located_but_wrong") or []))}
    for k, v in out.items():          # the top-level scalars (copies of the nested figures, per_gpu_value, rccl_ranks_seen ...)
        if k not in line and k not in DETAIL_ONLY_KEYS and (v is None or isinstance(v, (bool, int, float)) or (isinstance(v, str) and len(v) <= 80)):
            line[k] = v
    line["detail"] = detail_file
    for drop in (None, "accuracy", "stage_ms_per_step_rank0", ("roofline", "issue_model"), ("config", "parallelism")):
        if isinstance(drop, tuple):
            if isinstance(line.get(drop[0]), dict):
                line[drop[0]].pop(drop[1], None)
        elif drop:
            line.pop(drop, None)
        text = json.dumps(line, allow_nan=False, ensure_ascii=True, separators=(",", ":"))
        if len(text) <= limit:
            return text
    raise ValueError(f"the final bench line is {len(text)} bytes (limit {limit}) even without its optional groups")


def emit(out, stream=None, detail_path=None):
    """Print the two stdout lines of rank 0 -- `[bench-detail] {everything}` first, the compact line LAST -- and write the
    detail to `detail_path` (default: bench_detail.json beside this file)."""
    stream = stream or sys.stdout
    detail_path = detail_path or os.path.join(ROOT, "bench_detail.json")
    final = compact_line(out, detail_file=os.path.basename(detail_path))
    full = json.dumps(_clean(out), allow_nan=False, ensure_ascii=True)
    try:
        with open(detail_path, "w") as f:
            f.write(full + "\n")
    except OSError as e:
        print(f"[bench] could not write {detail_path}: {e}", file=sys.stderr, flush=True)
    print("[bench-detail] " + full, file=stream, flush=True)
    print(final, file=stream, flush=True)
    return final


def self_launch(args):
    """`python bench.py --gpus N` without torchrun's environment: start the N ranks as a CHILD process tree
    (nothing here has touched the GPU, torch is not imported: no exec of a GPU-initialised process), pass its
    output through and leave with its return code."""
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    print(f"[bench] --gpus {args.gpus} without WORLD_SIZE: launching {' '.join(cmd[1:8])} ...", file=sys.stderr, flush=True)
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    return subprocess.run(cmd, env=env).returncode


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--reps", type=int, default=5, help="timed repetitions of the K steps; value = median")
    ap.add_argument("--batch", type=int, default=25, help="queries per step per GPU (registered in one batch)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--places", type=int, default=N_PLACES_1GPU,
                    help="database size (1000000 = BASELINE.json configs[4], sharded over the ranks)")
    ap.add_argument("--mode", choices=["throughput", "latency"], default="throughput",
                    help="N > 1: B queries per rank per step (weak scaling) or one query per step "
                         "with its candidates sharded over the ranks (strong scaling)")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend (nccl = RCCL)")
    ap.add_argument("--collectives", choices=["capi", "torch"], default="capi",
                    help="N > 1: the all-gathers through the C ABI's own RCCL communicator "
                         "(gloc_knn_search_sharded, gloc_comm_all_gather_device) or through torch.distributed")
    ap.add_argument("--same-device", action="store_true",
                    help="rehearsal: all ranks on GPU 0 (use with --backend gloo)")
    ap.add_argument("--no-prefetch", action="store_true", help="prepare each step's queries inline")
    ap.add_argument("--no-pipeline", action="store_true",
                    help="wait for a step's registration before the next step's is enqueued (round 2's loop)")
    ap.add_argument("--reg-streams", type=int, choices=[1, 2], default=1,
                    help="streams of the two pipelined registration handles: 1 = one stream (batch after batch), 2 = a stream "
                         "each (two batches run side by side on the device)")
    ap.add_argument("--no-negatives", action="store_true", help="every place carries a world-A scan")
    ap.add_argument("--coarse", action="store_true",
                    help="also run the reference's 2-D step: the coarse (x, y, yaw) match of every (query, candidate) "
                         "pair on their BEV grids seeds the 3-D registration (gloc_coarse_*); not part of the metric's "
                         "default configuration")
    ap.add_argument("--scan-store", type=int, default=0,
                    help="distinct resident scans (0 = one per place up to 4541; places beyond alias modulo)")
    ap.add_argument("--no-legs", "--no-lone-query", dest="no_legs", action="store_true",
                    help="skip everything after the timed region but the CPU baseline: legs and sub-records (profiling runs)")
    ap.add_argument("--only-lone", action="store_true", help="of the legs and sub-records only the one-query-alone measurement (development)")
    ap.add_argument("--nn-heavy-thresh", type=int, default=None,
                    help="culled 1-NN tuning: processed chunks at which a wave of a batch's first pass hands its group to the second launch (0: off)")
    ap.add_argument("--leg-steps", type=int, default=4, help="steps of each leg (x --batch queries)")
    ap.add_argument("--cfge-rows", type=int, default=1_000_000,
                    help="N > 1: rows of the sharded descriptor database of sub_records.knn_cfgE_sharded (BASELINE configs[4]: 1M; 0: skip)")
    ap.add_argument("--views-cache", default=None, help="npz cache of the ray-cast base views (profiling runs)")
    ap.add_argument("--nn-src-per-lane", type=int, default=0, help="culled 1-NN tuning (1, 2, 4)")
    ap.add_argument("--nn-job-group", type=int, default=0, help="culled 1-NN tuning: jobs interleaved in the launch order")
    ap.add_argument("--nn-split-helpers", type=int, default=None, help="culled 1-NN tuning: wave slots per job for planned (split / early) source groups (0: off)")
    ap.add_argument("--nn-split-thresh", type=int, default=None, help="culled 1-NN tuning: work estimate (cycles) above which a source group is split")
    ap.add_argument("--nn-sub-jobs", type=int, default=None, help="culled 1-NN tuning: shares of a job's work-groups with their own slot in the launch order")
    ap.add_argument("--nn-mode", choices=["culled", "exhaustive"], default="culled",
                    help="1-NN search of the registration: identical correspondences, distances and selections bit for bit; poses agree to "
                         "1e-5 (the culled search sums its moments in fp32 about each wave's centre, the exhaustive one block-wise in fp64)")
    ap.add_argument("--ransac-confidence", type=float, default=None,
                    help="override gloc_reg_params.ransac_confidence (0: score all 3000 hypotheses)")
    ap.add_argument("--no-target-index", action="store_true",
                    help="leave the database scans in curve order (A/B of gloc_scan_store_build_target_index)")
    ap.add_argument("--min-success", type=float, default=0.9,
                    help="the run fails (exit code 3) when the stream's registration success rate is below this (0.95 in rounds 1-5, "
                         "whose different-world places were rigid copies of views at unrelated poses; a different world cast from the "
                         "SAME pose on the same road shares ground and corridor with the query, and the unseeded 3-D stage accepts "
                         "about one in nine of those at rank 0 -- `located_but_wrong`; the reference's 2-D step rejects them: "
                         "legs.coarse_seeded)")
    args = ap.parse_args()

    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        sys.exit(self_launch(args))
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        log(f"note: WORLD_SIZE = {world} but --gpus {args.gpus}: running with the {world} ranks that exist")
    t_setup = time.time()
    gate_views = None
    if world == 1 and args.mode == "throughput" and not args.no_legs:
        sys.path.insert(0, os.path.join(ROOT, "tools"))
        import gate_holdout
        # (the only host ray-casts left: numpy, a process per view -- forks, so BEFORE the GPU is initialised; every scan of
        # the stream itself is cast on the device, gloc_scan_store_add_raycast_batch)
        gate_views = gate_holdout.build_views(args.views_cache + ".gate.npz" if args.views_cache else None, workers=min(16, os.cpu_count() or 1))
    import torch
    import torch.distributed as dist
    from gloc3d_amd import capi, sharded, synth

    if args.same_device:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    comm_dev = None
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group(args.backend, rank=rank, world_size=world)
            comm_dev = torch.device("cpu")

    n_places = args.places
    B = max(1, args.batch) if args.mode == "throughput" else 1
    per_step = world * B if args.mode == "throughput" else 1   # queries per step, whole job
    n_steps, n_warm, n_reps = args.steps, args.warmup, max(1, args.reps)

    # ---- resident state: descriptor database -------------------------------------------------
    index = capi.KnnIndex(DIM, device=local_rank)
    n_local = len(sharded.shard_rows(n_places, rank, world))
    index.reserve(n_local)
    index.add_synthetic(1, DB_SEED, rank, n_local, row_stride=world)   # this rank's interleaved shard
    index.synchronize()
    log(f"database: {n_places} x {DIM} ({n_local} rows on rank 0), generated on device")

    # ---- resident state: the scan store (every rank holds a replica) ---------------------------
    store = capi.ScanStore(device=local_rank)
    n_store = min(n_places, args.scan_store or N_PLACES_1GPU)
    neg_on = not args.no_negatives
    traj, world_a, world_b = headline_world(n_store)       # n_store poses along a closed loop through ONE world (+ the negatives' world)

    def is_negative(g):
        return neg_on and (g % NEG_EVERY == 1)

    # every place's scan is its OWN ray-cast from its own pose (device: 64 casts a launch sequence); place g % 4 == 1 is
    # cast from the same pose in the other world
    place_scan = np.empty(n_store, np.uint32)
    g_all = np.arange(n_store)
    neg_mask = np.array([is_negative(int(g)) for g in g_all], bool)
    for wrld, sel in ((world_a, g_all[~neg_mask]), (world_b, g_all[neg_mask])):
        if len(sel):
            place_scan[sel] = store.add_raycast(wrld, traj[sel], (PLACE_SEED + sel).astype(np.uint64))
    if not args.no_target_index:
        store.build_target_index_batch(place_scan)   # the database places: kd-ordered target index, once (batches of 64 scans)
    live_b, _ = store.bytes()
    mean_pts = float(np.mean([store.points(int(sid)) for sid in place_scan]))
    log(f"scan store: {n_store} distinct ray-casts along a {TRAJ_LENGTH_M * n_store / N_PLACES_1GPU:.0f} m loop "
        f"({len(world_a['lo'])} boxes in world A), {live_b / 2**30:.1f} GiB, ~{mean_pts:.0f} pts each; "
        + (f"negatives: places g % {NEG_EVERY} == 1 (cast in world B)" if neg_on else "no negatives"))

    # ground truth: the sensor pose of a place's scan
    scan_override = {}          # place -> (scan id, pose) while a data leg runs
    pose_cache = {}

    def place_pose(g):
        g = int(g) % n_store
        if g in scan_override:
            return scan_override[g][1]
        return far_away_pose() if is_negative(g) else traj[g]

    query_pose_override = {}    # stream id -> pose while a data leg runs (its queries are other scans)

    def query_pose(j):
        if int(j) in query_pose_override:
            return query_pose_override[int(j)]
        return traj[int(q_place[int(j)])] @ query_offset(int(j))

    # ---- the query stream: distinct host-side scans + descriptors -------------------------------
    n_stream = n_steps * per_step                       # distinct queries of one repetition
    total = (n_steps + n_warm) * per_step
    # query j revisits place g_j (any place: a quarter of them carry a different-world scan, so that place cannot be found)
    q_place = (np.arange(total, dtype=np.int64) * 977 + 211) % n_store
    q_desc_host = torch.from_numpy(synth.queries_near(DB_SEED, q_place, DIM)).pin_memory()
    # the queries' scans: each its OWN ray-cast of world A from query_pose(j) (device), read back into pinned host memory --
    # from then on they exist only on the host, like scans arriving from a sensor
    # (a rank makes only the scans of the queries it will prepare: its B of every step's world x B)
    q_scan_host = {}
    mine_all = []
    for i in range(n_steps + n_warm):
        q0 = i * per_step
        mine_all.extend(range(q0 + rank * B, q0 + rank * B + B) if args.mode == "throughput" else range(q0, q0 + 1))
    for a in range(0, len(mine_all), 64):
        js = mine_all[a:a + 64]
        sids = store.add_raycast(world_a, [query_pose(j) for j in js], np.array([QUERY_SEED + j for j in js], np.uint64))
        for j, sid in zip(js, sids):
            q_scan_host[j] = torch.from_numpy(store.download(sid)).pin_memory()
            store.release(sid)
    store_scans_resident = len(store)

    reg = capi.Registrar(device=local_rank, store=store)
    reg.set_option(capi.REG_OPT_PROFILE, 1)
    reg.set_option(capi.REG_OPT_NN_MODE, capi.REG_NN_CULLED if args.nn_mode == "culled" else capi.REG_NN_EXHAUSTIVE)
    def tune(r):
        if args.nn_src_per_lane:
            r.set_option(capi.REG_OPT_NN_SRC_PER_LANE, args.nn_src_per_lane)
        if args.nn_job_group:
            r.set_option(capi.REG_OPT_NN_JOB_GROUP, args.nn_job_group)
        for opt, v in ((capi.REG_OPT_NN_SPLIT_HELPERS, args.nn_split_helpers), (capi.REG_OPT_NN_SPLIT_THRESH, args.nn_split_thresh),
                       (capi.REG_OPT_NN_SUB_JOBS, args.nn_sub_jobs), (capi.REG_OPT_NN_HEAVY_THRESH, args.nn_heavy_thresh)):
            if v is not None:
                r.set_option(opt, v)
    tune(reg)
    params = capi.default_reg_params(ransac_iters=RANSAC_ITERS, icp_iters=ICP_ITERS,
                                     min_inlier_ratio=MIN_INLIER_RATIO, max_rmse=MAX_RMSE, max_final_step=MAX_FINAL_STEP)
    if args.ransac_confidence is not None:
        params.ransac_confidence = args.ransac_confidence
    cur = {"params": params}       # (legs swap the parameters / the mode)
    # Registration pipeline: two handles with their own workspaces on ONE stream; batch i + 1 is enqueued
    # (gloc_reg_batch_multi_begin) before batch i's results are waited for (gloc_reg_batch_multi_end), so the device runs
    # batch after batch while the host unpacks, selects, releases and looks the next scans up.  One host thread: at
    # N > 1 every rank issues its collectives in the same order (search i + 1, tables i, search i + 2, tables i + 1 ...)
    pipeline = args.mode == "throughput" and not args.no_pipeline
    regs = [reg]
    if pipeline:
        reg_b = capi.Registrar(device=local_rank, store=store)
        reg_b.set_option(capi.REG_OPT_PROFILE, 1)
        reg_b.set_option(capi.REG_OPT_NN_MODE, capi.REG_NN_CULLED if args.nn_mode == "culled" else capi.REG_NN_EXHAUSTIVE)
        tune(reg_b)
        reg_stream = torch.cuda.Stream(device=dev)
        reg.set_stream(reg_stream.cuda_stream)
        reg_stream_b = torch.cuda.Stream(device=dev) if args.reg_streams == 2 else reg_stream
        reg_b.set_stream(reg_stream_b.cuda_stream)
        regs.append(reg_b)

    def prof(name):
        ms, n = 0.0, 0
        for r_ in regs:
            a_, b_ = r_.profile(name)
            ms, n = ms + a_, n + b_
        return ms, n

    def prof_reset():
        for r_ in regs:
            r_.profile_reset()

    def nn_stats_all():
        a_ = [r_.nn_stats() for r_ in regs]
        return sum(x[0] for x in a_), sum(x[1] for x in a_)

    capi_knn, collectives, rccl_ranks_seen, comm = None, "none", None, None
    # the transport the caller asked for: the C-ABI RCCL path unless --collectives torch, a gloo backend or ranks that
    # share a device (rehearsals) say otherwise; if it was asked for and is not what ran, the run FAILS (after its line)
    want_capi = world > 1 and args.collectives == "capi" and args.backend == "nccl" and not args.same_device
    if want_capi:
        try:
            comm = capi.Comm(local_rank, rank, world, sharded.torch_exchange(dev))
            capi_knn = sharded.CapiShardedKnn(index, comm)
            probe = torch.full((1, 1, sharded.RESULT_COLS), float(rank), dtype=torch.float32, device=dev)
            got = capi_knn.all_gather_tables(probe)     # self-test: every rank's row must arrive in rank order
            torch.cuda.synchronize()
            if not (got[:, 0, 0].cpu() == torch.arange(world, dtype=torch.float32)).all():
                raise RuntimeError("all-gather self-test returned the wrong rows")
            rccl_ranks_seen = comm.rank_world()[1]      # gloc_comm_rank: what the communicator itself reports
            collectives = "capi (gloc_knn_search_sharded + gloc_comm_all_gather_device, RCCL)"
        except Exception as e:   # loud, and recorded in the JSON line: never a silent change of path
            capi_knn = None
            print(f"[bench] rank {rank}: C-ABI RCCL communicator unavailable ({e}); using torch.distributed", file=sys.stderr, flush=True)
    if world > 1:
        ok_all = torch.tensor([1 if capi_knn is not None else 0], device=comm_dev or dev)
        dist.all_reduce(ok_all, op=dist.ReduceOp.MIN)     # all ranks take the same path
        if int(ok_all.item()) == 0:
            capi_knn = None
            rccl_ranks_seen = None
    if capi_knn is not None:
        # second self-test: the sharded search below the C ABI (local top-k -> grouped RCCL all-gather -> merge)
        # against the same search over torch.distributed collectives, bit for bit, on a few queries
        same = False
        try:
            from gloc3d_amd import synth as _synth
            ref_knn = sharded.ShardedKnn(rank, world, sharded.hip_local_search(index), sharded.hip_merge(local_rank),
                                         comm_device=comm_dev)
            probe_q = torch.from_numpy(_synth.queries_near(DB_SEED, np.arange(7, 7 + 8 * 1000, 1000) % max(n_places, 1), DIM)).to(dev)
            ia, da = capi_knn.search(probe_q, TOP_K)
            ib, db_ = ref_knn.search(probe_q, TOP_K)
            torch.cuda.synchronize()
            same = (bool((ia.cpu().to(torch.int64) == ib.cpu().to(torch.int64)).all())
                    and bool((da.cpu().view(torch.int32) == db_.cpu().view(torch.int32)).all()))
        except Exception as e:
            print(f"[bench] rank {rank}: sharded-search self-test raised {e!r}", file=sys.stderr, flush=True)
        flag = torch.tensor([1 if same else 0], device=comm_dev or dev)
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        if int(flag.item()) == 0:
            print(f"[bench] rank {rank}: gloc_knn_search_sharded disagrees with the torch.distributed path; using torch.distributed",
                  file=sys.stderr, flush=True)
            capi_knn, rccl_ranks_seen = None, None
    if capi_knn is not None:
        knn = capi_knn
    else:
        knn = sharded.ShardedKnn(rank, world, sharded.hip_local_search(index), sharded.hip_merge(local_rank),
                                 comm_device=comm_dev)
        if world > 1:
            collectives = f"torch.distributed ({args.backend})"

    # optional: the coarse 2-D match (one grid per place, made on the device from the resident scans)
    cm, place_grid, cm_lock, cur_qgrids = None, None, threading.Lock(), {}
    NO_GRID = np.uint32(0xFFFFFFFF)
    coarse_stat = {"pairs": 0, "accepted": 0}
    if args.coarse:
        cm = capi.CoarseMatcher(local_rank)
        place_grid = np.concatenate([cm.add_store_scans(store, place_scan[i:i + 256]) for i in range(0, len(place_scan), 256)])
        log(f"coarse grids: {n_store} places")

    def coarse_init(q_ids, places):
        """init_T [B, n, 4, 4] from the 2-D match of every (query, candidate) pair; identity where it fails."""
        p = np.asarray(places, np.int64)
        Bq, n = p.shape
        qg = np.repeat(np.array([cur_qgrids[int(q)] for q in q_ids], np.uint32), n)
        dg = place_grid[np.clip(p, 0, None).reshape(-1) % n_store]
        have = dg != NO_GRID                      # (a leg makes grids only for the places its queries retrieve)
        dg = np.where(have, dg, dg[have][0] if have.any() else 0).astype(np.uint32)
        with cm_lock:
            xy_yaw, _, ok2 = cm.match_pairs(qg, dg)
        T = np.tile(np.eye(4, dtype=np.float32), (Bq * n, 1, 1))
        c, s_ = np.cos(xy_yaw[:, 2]), np.sin(xy_yaw[:, 2])
        use = ok2 & have & (p.reshape(-1) >= 0)
        coarse_stat["pairs"] += int(np.count_nonzero(have & (p.reshape(-1) >= 0)))
        coarse_stat["accepted"] += int(np.count_nonzero(use))
        T[use, 0, 0], T[use, 0, 1], T[use, 1, 0], T[use, 1, 1] = c[use], -s_[use], s_[use], c[use]
        T[use, 0, 3], T[use, 1, 3] = xy_yaw[use, 0], xy_yaw[use, 1]
        # the reference registers only what its 2-D match accepts (global_localization.cpp:519-526: `if (matched)`)
        return T.reshape(Bq, n, 4, 4), np.where(use.reshape(Bq, n), p, -1)

    def scans_of(places):
        """global place ids [.., n] (-1 = none) -> resident scan ids (every rank holds all scans)."""
        p = np.asarray(places, np.int64)
        out = place_scan[np.clip(p, 0, None) % n_store].astype(np.uint32)
        if scan_override:
            flat, pf = out.reshape(-1), (np.clip(p, 0, None) % n_store).reshape(-1)
            for i_, g_ in enumerate(pf):
                o = scan_override.get(int(g_))
                if o is not None:
                    flat[i_] = o[0]
        out[p < 0] = capi.NO_SCAN
        return out

    fs_state = {"on": False, "jobs": 0, "queries": 0}

    def register_multi(q_ids, places):
        init = None
        if cm is not None:
            init, places = coarse_init(q_ids, places)
        if fs_state["on"]:
            # the reference's loop as written: stop at the first success (gloc_reg_first_success_multi)
            f = reg.first_success_multi(q_ids, scans_of(places), params=cur["params"], init_T=init)
            shape = np.asarray(places).shape
            out = np.zeros(shape + (sharded.RESULT_COLS,), np.float32)
            for k_, rk in enumerate(f["rank"]):
                if rk >= 0:
                    out[k_, rk, :16] = f["T"][k_].reshape(16)
                    out[k_, rk, 16], out[k_, rk, 17], out[k_, rk, 18] = f["rmse"][k_], f["inliers"][k_], 1.0
            fs_state["jobs"] += f["jobs_run"]
            fs_state["queries"] += len(q_ids)
            return out
        r = reg.batch_multi(q_ids, scans_of(places), params=cur["params"], init_T=init)
        return sharded.pack_results(r, np.asarray(places).shape)

    def local_register(q_id, local_rows, ranks):   # latency mode: this rank's share of one query's candidates
        g = np.asarray(local_rows, np.int64) * world + rank
        r = reg.batch_ids(q_id, scans_of(g), params=cur["params"], stream_ids=ranks)
        return sharded.pack_results(r, (len(g),))

    sreg = sharded.ShardedRegistrar(rank, world, local_register, comm_device=comm_dev,
                                    gather=capi_knn.all_gather_tables if capi_knn is not None else None)
    qreg = sharded.QueryParallelRegistrar(rank, world, None, comm_device=comm_dev)
    stage = {"prep_wait": 0.0, "h2d_index": 0.0, "knn": 0.0, "register": 0.0}
    work_pairs = []

    # ---- one step -----------------------------------------------------------------------------
    def my_slice(i, Bq=None):
        Bq = B if Bq is None else Bq
        q0 = i * per_step
        return (q0 + rank * Bq, q0 + rank * Bq + Bq) if args.mode == "throughput" else (q0, q0 + 1)

    def prepare(i, Bq=None):
        """Query preparation of step i: this rank's fresh query scans go H2D and are indexed
        (gloc_scan_store_add, on the store's stream); the step's descriptors go H2D."""
        t0 = time.time()
        a, b = my_slice(i, Bq)
        ids = store.add_batch([q_scan_host[j].numpy() for j in range(a, b)])   # one launch sequence for all of them
        if cm is not None:
            with cm_lock:
                for sid, gid_ in zip(ids, cm.add_store_scans(store, ids)):   # the step's query grids in one launch sequence
                    cur_qgrids[sid] = int(gid_)
        q0 = i * per_step
        n_q = per_step if Bq is None else Bq * (world if args.mode == "throughput" else 1)
        qd = q_desc_host[q0:q0 + n_q].to(dev, non_blocking=True)
        return ids, qd, time.time() - t0

    class Prefetcher:
        def __init__(self):
            self.slot, self.th = None, None

        def start(self, i, Bq=None):
            def run():
                torch.cuda.set_device(local_rank)
                self.slot = prepare(i, Bq)
            self.th = threading.Thread(target=run)
            self.th.start()

        def take(self):
            self.th.join()
            s, self.slot, self.th = self.slot, None, None
            return s

    pre = Prefetcher()

    def step(i, last, record, Bq=None, prefetch=True):
        """One step.  Bq: queries of this rank (legs use smaller batches); returns (cand, sel, tables)."""
        t0 = time.time()
        if args.no_prefetch or not prefetch:
            ids, qd, t_prep = prepare(i, Bq)
        else:
            ids, qd, t_prep = pre.take()
            if not last:
                pre.start(i + 1, Bq)      # the next step's uploads + indexing overlap this step's registration
        t1 = time.time()
        torch.cuda.current_stream().synchronize()   # the descriptors' H2D
        idx, d2 = knn.search(qd, TOP_K)
        cand = idx.cpu().numpy()                          # [queries of the step, 20] global place ids, retrieval order
        t2 = time.time()
        a, b = my_slice(i, Bq)
        if args.mode == "throughput":
            tables = qreg.register_many(ids, cand, dev, register_multi, capi_knn=capi_knn)   # [world*B, 20, 19]
        else:
            tables = sreg.register(ids[0], cand[0], dev)[None]
        if not isinstance(tables, np.ndarray):
            tables = tables.detach().cpu().numpy()     # one D2H of the gathered tables; the selection rule runs on the host
        sel = [sharded.ShardedRegistrar.select_first_ok(t) for t in tables]
        for sid in ids:
            reg.scan_release(sid)
            if cm is not None:
                with cm_lock:
                    cm.release(cur_qgrids.pop(sid))
        t3 = time.time()
        if record:
            stage["prep_wait"] += t1 - t0
            stage["h2d_index"] += t_prep
            stage["knn"] += t2 - t1
            stage["register"] += t3 - t2
            nb = b - a
            mine = cand[rank * nb:(rank + 1) * nb] if args.mode == "throughput" else cand[:1]
            for k in range(mine.shape[0]):
                nq = q_scan_host[a + k].shape[0] if args.mode == "throughput" else q_scan_host[a].shape[0]
                work_pairs.append((nq, int(np.count_nonzero(mine[k] >= 0))))
        return cand, sel, tables

    def fence():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    def run_stream_pipelined(first, count, record, Bq, keep):
        fence()
        t0 = time.time()
        pre.start(first, Bq)
        sels_, cands_, tabs_, pend = [], [], [], None

        def finish(p):
            i_, ids_, cand_, h_ = p
            t_a = time.time()
            r = h_.batch_multi_end()                       # waits for THAT batch; the next one is already queued behind it
            t_b = time.time()
            nb_ = len(ids_)
            tables = qreg.gather_tables(sharded.pack_results(r, (nb_, cand_.shape[1])), dev, capi_knn)   # all ranks' rows
            if not isinstance(tables, np.ndarray):
                tables = tables.detach().cpu().numpy()
            sels_.extend(sharded.ShardedRegistrar.select_first_ok(t) for t in tables)
            for sid in ids_:
                store.release(sid)                         # (its batch has completed; no stream to wait for)
                if cm is not None:
                    with cm_lock:
                        cm.release(cur_qgrids.pop(sid))
            if keep:
                cands_.append(cand_)
                tabs_.append(tables)
            if record:
                stage["register"] += t_b - t_a
                a_, _ = my_slice(i_, Bq)
                for k in range(nb_):
                    work_pairs.append((q_scan_host[a_ + k].shape[0], int(np.count_nonzero(cand_[rank * nb_ + k] >= 0))))

        for i in range(first, first + count):
            t_a = time.time()
            ids, qd, t_prep = pre.take()
            if i + 1 < first + count:
                pre.start(i + 1, Bq)
            t_b = time.time()
            torch.cuda.current_stream().synchronize()       # the descriptors' H2D
            idx, d2 = knn.search(qd, TOP_K)
            cand = idx.cpu().numpy()
            t_c = time.time()
            mine = cand[rank * len(ids):(rank + 1) * len(ids)]          # this rank's queries of the step
            init, go = (None, mine) if cm is None else coarse_init(ids, mine)
            h = regs[i % 2]
            h.batch_multi_begin(ids, scans_of(go), params=cur["params"], init_T=init)
            if record:
                stage["prep_wait"] += t_b - t_a
                stage["h2d_index"] += t_prep
                stage["knn"] += t_c - t_b
                stage["enqueue"] = stage.get("enqueue", 0.0) + time.time() - t_c
            if pend is not None:
                finish(pend)
            pend = (i, ids, cand, h)
        finish(pend)
        fence()
        return time.time() - t0, sels_, cands_, tabs_

    def run_stream(first, count, record=False, Bq=None, keep=False):
        """`count` steps starting at `first`, fenced on both sides.  Returns (seconds, selections, [cand], [tables])."""
        if pipeline and not fs_state["on"] and not args.no_prefetch:
            return run_stream_pipelined(first, count, record, Bq, keep)
        fence()
        t0 = time.time()
        if not args.no_prefetch:
            pre.start(first, Bq)       # the first step's preparation has no earlier step to hide behind: paid in full
        sels_, cands_, tabs_ = [], [], []
        for i in range(first, first + count):
            cand, sel, tables = step(i, i == first + count - 1, record, Bq)
            sels_.extend(sel)
            if keep:
                cands_.append(cand)
                tabs_.append(tables)
        fence()
        return time.time() - t0, sels_, cands_, tabs_

    log(f"setup {time.time() - t_setup:.1f} s; warmup {n_warm}, timing {n_reps} x {n_steps} steps of {per_step} queries")
    if n_warm and not args.no_prefetch:
        pre.start(n_steps)
    for i in range(n_steps, n_steps + n_warm):     # warm-up queries: the tail of the stream
        cand, sel, _ = step(i, i == n_steps + n_warm - 1, False)
        assert (cand[:, 0] == q_place[i * per_step:(i + 1) * per_step]).all(), "retrieval sanity: top-1 != query place"
    rep_s, sels, cands_last, tabs_last = [], [], [], []
    for rep in range(n_reps):
        if rep == n_reps - 1:
            fence()
            prof_reset()
        elapsed, rsel, rc_, rt_ = run_stream(0, n_steps, record=rep == n_reps - 1, keep=rep == n_reps - 1)
        if world > 1:
            t = torch.tensor([elapsed], dtype=torch.float64, device=comm_dev or dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            elapsed = float(t.item())
        rep_s.append(elapsed)
        sels, cands_last, tabs_last = rsel, rc_, rt_
    elapsed = float(np.median(rep_s))

    # ---- accuracy of the timed stream (the last repetition's results; every repetition computes the same) ----
    all_cand = np.concatenate(cands_last) if cands_last else np.zeros((0, TOP_K), np.int64)
    all_tabs = np.concatenate(tabs_last) if tabs_last else np.zeros((0, TOP_K, sharded.RESULT_COLS), np.float32)
    stream_q = list(range(all_cand.shape[0]))
    accuracy = accuracy_of(all_cand, sels, all_tabs, stream_q, place_pose, query_pose, lambda g: not is_negative(g % n_store))

    # ---- roofline of the dominant kernel (K4 point-NN), from the HIP events of the last repetition --
    nn_all_ms, nn_all_launches = prof("nn")
    cold_ms, cold_launches = prof("nn_cold")      # a batch's first pass (no previous correspondence, writes the pairs): another instantiation
    # the dominant kernel is the WARM instantiation (20 of a query's 21 passes): its launches alone
    nn_ms, nn_launches = (nn_all_ms - cold_ms, nn_all_launches - cold_launches) if nn_all_launches > cold_launches else (nn_all_ms, nn_all_launches)
    stage_ms = {n: prof(n)[0] for n in ("nn", "ransac_score", "ransac_hyp", "accum", "solve")}
    passes = 1 + ICP_ITERS
    pairs_eval, _ = nn_stats_all()
    avg_launch_s = (nn_ms / max(nn_launches, 1)) * 1e-3
    jobs_per_launch = float(np.mean([c for _, c in work_pairs])) * (B if args.mode == "throughput" else 1) \
        if work_pairs else 0.0
    # algorithmic bytes of one launch: every job reads its query scan and its candidate scan once
    # (sorted float4 = 16 B/point) and writes corr + d2 (8 B/source point)
    pts_q = float(np.mean([n for n, _ in work_pairs])) if work_pairs else 0.0
    alg_bytes = jobs_per_launch * (BYTES_PER_POINT_INDEXED * (pts_q + mean_pts) + 8 * pts_q)
    all_pairs = jobs_per_launch * pts_q * mean_pts
    eval_pairs = all_pairs if args.nn_mode == "exhaustive" else float(pairs_eval) / max(nn_all_launches, 1)
    kname = "gloc::reg::nn_kernel" if args.nn_mode == "exhaustive" else "gloc::reg::nn_compact_kernel<2,false,false,false,true> (warm passes)"
    traffic = None  # HBM bytes per launch from the committed PMC passes (profiles/), same workload
    pmc = next((os.path.join(ROOT, "profiles", f) for f in PMC_FILES if os.path.exists(os.path.join(ROOT, "profiles", f))), None)
    issue_model = None
    if args.nn_mode == "culled" and world == 1 and pmc:
        try:
            pj = json.load(open(pmc))
        except ValueError:      # an empty or damaged file: the line is still printed, without the PMC-derived fields
            pj = {}
        traffic = pj.get("hbm_bytes_per_launch")
        if traffic and pj.get("jobs_per_launch") and abs(pj["jobs_per_launch"] - jobs_per_launch) > 1e-6:
            traffic = traffic * jobs_per_launch / pj["jobs_per_launch"]      # PMC passes ran at another batch size
        # instruction-issue model: the committed per-wave instruction counts (PMC) x the waves of a launch over the
        # live launch time, against the vector pipes' issue rate (39.3 T lane-instructions/s)
        pw = pj.get("per_wave", {})
        if pw.get("valu") and avg_launch_s > 0:
            waves = jobs_per_launch * np.ceil(pts_q / 128.0)
            lane_ops = pw["valu"] * 64.0 * waves / avg_launch_s
            issue_model = {"valu_per_wave": pw["valu"], "salu_per_wave": pw.get("salu"), "lds_per_wave": pw.get("lds"),
                           "vmem_per_wave": pw.get("vmem_rd"), "waves_per_launch": waves,
                           "vector_lane_ops_per_s": lane_ops,
                           # 157.3 TFLOP/s counts an FMA as two flops and a packed instruction as two lanes' worth: one
                           # instruction per lane and cycle is a quarter of it (256 CUs x 4 SIMDs x 16 lanes x 2.4 GHz)
                           "frac_of_vector_issue_peak": lane_ops / (PEAK_FP32_TFLOPS * 1e12 / 4),
                           "valu_busy_frac_under_pmc": pj.get("valu_busy_frac"),
                           "source": f"profiles/{os.path.basename(pmc)} (SQ_INSTS_* / SQ_WAVES; busy = SQ_ACTIVE_INST_VALU "
                                     "over SQ_BUSY_CYCLES): the vector pipes are busy for that fraction of the launch, a wave64 "
                                     "instruction holding its SIMD for 4 cycles (8 for fp64 and the packed-fp32 forms)"}
    if args.nn_mode == "exhaustive":
        ach = FLOP_PER_PAIR * all_pairs / avg_launch_s / 1e12 if avg_launch_s > 0 else 0.0
        roofline = {"kernel": kname, "bound": "mfma", "achieved": ach, "peak": PEAK_FP32_TFLOPS, "unit": "TFLOP/s",
                    "frac": ach / PEAK_FP32_TFLOPS, "traffic": traffic,
                    "note": f"{FLOP_PER_PAIR} flop/pair x {all_pairs:.3e} pairs per launch / {avg_launch_s*1e3:.3f} ms; "
                            f"fp32 vector peak = fp32 MFMA peak {PEAK_FP32_TFLOPS} TF"}
    else:
        ach = alg_bytes / avg_launch_s / 1e9 if avg_launch_s > 0 else 0.0
        roofline = {"kernel": kname, "bound": "hbm", "achieved": ach, "peak": PEAK_HBM_GBS, "unit": "GB/s",
                    "frac": ach / PEAK_HBM_GBS, "traffic": traffic,
                    "algorithmic_bytes_per_launch": alg_bytes, "jobs_per_launch": jobs_per_launch,
                    "launch_ms": avg_launch_s * 1e3, "launches": nn_launches,
                    "cold_launch_ms": cold_ms / cold_launches if cold_launches else None, "cold_launches": cold_launches,
                    "pairs_evaluated_per_launch": eval_pairs, "pairs_exhaustive_per_launch": all_pairs,
                    "pairs_evaluated_per_source": eval_pairs / max(jobs_per_launch * pts_q, 1.0),
                    "issue_model": issue_model,
                    "note": "the culled search evaluates ~1e-3 of the pairs SURVEY 8d's flop count assumes, so its "
                            "floor is reading each scan once: (16 B x (query + candidate points) + 8 B x query "
                            "points) x jobs per launch over the HIP-event duration of the launch (one stream, "
                            "launches do not overlap); the kernel itself is vector-issue bound, see DESIGN.md; the "
                            "brute-force nn_kernel north_star names: legs.nn_exhaustive_sample"}

    run_legs = world == 1 and args.mode == "throughput" and not args.no_legs
    L = max(1, min(args.leg_steps, n_steps))

    def leg_run(n_leg_steps, Bq=None, acc=True, recall_defined=True):
        """A leg = the same pipeline steps 0 .. n-1 under changed settings: q/s, stage times, pairs per source, accuracy."""
        Bq_ = B if Bq is None else Bq
        run_stream(n_steps if n_warm else 0, 1, Bq=Bq)      # one untimed step under the new settings
        prof_reset()
        t, s_, c_, tb_ = run_stream(0, n_leg_steps, Bq=Bq, keep=True)
        ms, nl = prof("nn")
        pe, _ = nn_stats_all()
        st = {n: prof(n)[0] / n_leg_steps for n in ("nn", "ransac_score", "ransac_hyp", "accum", "solve")}
        nq = n_leg_steps * Bq_
        cc = np.concatenate(c_)
        # queries of step i are stream ids i * per_step .. + Bq
        qids = [i * per_step + k for i in range(n_leg_steps) for k in range(Bq_)]
        out = {"value": nq / t, "unit": "queries/s", "queries": nq, "queries_per_batch": Bq_, "ms_per_step": t / n_leg_steps * 1e3,
               "stage_ms_per_step": st, "nn_launch_ms": ms / max(nl, 1),
               "pairs_evaluated_per_source": float(pe) / max(nl, 1) / max(Bq_ * TOP_K * pts_q, 1.0)}
        if acc:
            out["accuracy"] = accuracy_of(cc, s_, np.concatenate(tb_), qids, place_pose, query_pose,
                                          lambda g: not is_negative(g % n_store), recall_defined=recall_defined)
            out["accuracy"].pop("definition", None)
        return out, s_

    legs = {}
    first_success = None
    lone = None
    if run_legs:
        # one query alone (20 jobs per launch), its preparation included and NOT hidden: BASELINE configs[2]
        q_desc_np = q_desc_host.numpy()

        def lone_pass(n_):
            out_ = []
            for j in range(n_):
                t0 = time.time()
                sid = store.add(q_scan_host[j].numpy())
                t_prep = time.time() - t0
                # (retrieval through the C ABI's host call -- descriptor H2D, top-20, indices back in ONE call: gloc_knn_search;
                # until round 6 through the torch-side sharded wrapper: ~0.1 ms of tensor bookkeeping per query)
                ci, _ = index.search(q_desc_np[j:j + 1], TOP_K)
                reg.batch_multi([sid], scans_of(ci.astype(np.int64)), params=params)
                reg.scan_release(sid)
                out_.append((time.time() - t0, t_prep))
            return out_
        # wall clock with the per-kernel HIP events OFF (two events around each of a query's ~70 launches cost 0.4 ms of
        # its 3), then the stage times with them on
        reg.set_option(capi.REG_OPT_PROFILE, 0)
        t_lone = lone_pass(10)
        # ... and the same ten queries launch by launch (GLOC_REG_OPT_NN_CHAIN 0: round 5's pipeline, 2 launches per ICP pass)
        reg.set_option(capi.REG_OPT_NN_CHAIN, 0)
        t_unchained = lone_pass(10)
        reg.set_option(capi.REG_OPT_NN_CHAIN, 1)
        n_chain, n_chain_timeouts = reg.debug_chain()
        reg.set_option(capi.REG_OPT_PROFILE, 1)
        prof_reset()
        t_prof = lone_pass(8)
        ms1, n1 = reg.profile("nn")
        msc, nc = reg.profile("nn_cold")
        mss, ns = reg.profile("solve")
        roofline["launch_ms_one_query_20_jobs"] = (ms1 - msc) / max(n1 - nc, 1)
        t_med = float(np.median([a for a, _ in t_lone[2:]]))
        lone = {"ms_per_query": t_med * 1e3, "queries_per_s": 1.0 / t_med,
                "ms_per_query_launch_by_launch": float(np.median([a for a, _ in t_unchained[2:]])) * 1e3,
                "chained_launches": int(n_chain), "chained_launches_timed_out": int(n_chain_timeouts),
                "icp_passes": "one chained launch for the 20 warm ICP passes of the query's 20 jobs (GLOC_REG_OPT_NN_CHAIN; the stage times below "
                              "are of the launch-by-launch pipeline: the per-kernel events switch the chain off)",
                "prep_ms": float(np.median([b for _, b in t_lone[2:]])) * 1e3,
                "ms_per_query_with_stage_events": float(np.median([a for a, _ in t_prof[2:]])) * 1e3,
                "nn_launch_ms": (ms1 - msc) / max(n1 - nc, 1), "nn_cold_launch_ms": msc / max(nc, 1), "nn_ms_per_query": ms1 / 8,
                "solve_launch_ms": mss / max(ns, 1),
                "heavy_group_plan": "default (gloc_reg_set_option NN_SPLIT_HELPERS -1: 256 wave slots per job at 20 jobs, threshold 60000 cycles, 8 slots per job in the launch order)",
                "what": "BASELINE configs[2]: 1 query x 20 full-size candidates, RANSAC 3000 adaptive + ICP 20, batch of ONE: "
                        "scan H2D + index, descriptor H2D, top-20, registration, release -- wall clock, nothing overlapped "
                        "(retrieval enqueued beside the scan's indexing was tried in round 5: 3.09 -> 3.16 ms)"}

        more_legs = not args.only_lone
        if more_legs:
            # the same stream with the reference's early exit (registration stops at a query's first successful candidate)
            fs_state["on"] = True
            t_fs, fsel, _, _ = run_stream(0, n_steps)
            fs_state["on"] = False
            first_success = {"value": n_steps * per_step / t_fs, "unit": "queries/s",
                             "registrations_per_query": fs_state["jobs"] / max(fs_state["queries"], 1),
                             "same_selection_as_full_batch": bool(fsel == sels),
                             "note": "gloc_reg_first_success_multi: rank by rank, only queries still without a success go on "
                                     "(registration/global_localization.cpp:519-572 stops at the first match()==true); not the "
                                     "metric's configuration, which registers all 20 candidates"}

            # leg 1: no adaptive RANSAC stop -- all 3000 hypotheses generated and scored (SURVEY App. B's wording of S2)
            log("leg: RANSAC without the adaptive stop (3000 hypotheses scored)")
            cur["params"] = capi.default_reg_params(ransac_iters=RANSAC_ITERS, icp_iters=ICP_ITERS, min_inlier_ratio=MIN_INLIER_RATIO,
                                                    max_rmse=MAX_RMSE, max_final_step=MAX_FINAL_STEP, ransac_confidence=0.0)
            legs["ransac_all_3000"], s3k = leg_run(L)
            legs["ransac_all_3000"]["same_selection_as_adaptive"] = bool(s3k == sels[:len(s3k)])
            legs["ransac_all_3000"]["what"] = ("ransac_confidence = 0: every one of the 3000 hypotheses is generated and scored, best = max "
                                               "inliers, tie -> smallest h (SURVEY App. B); the headline follows the reference's call, "
                                               "cv::estimateAffinePartial2D at its default confidence 0.99 (loop_detector.cpp:256-257), "
                                               "which stops after the adaptive count")
            cur["params"] = params

            # leg 2: the brute-force 1-NN kernel north_star names (every (source, target) pair), a few whole queries
            log("leg: exhaustive 1-NN kernel")
            for r_ in regs:
                r_.set_option(capi.REG_OPT_NN_MODE, capi.REG_NN_EXHAUSTIVE)
            n_ex = min(3, n_steps)
            legs["nn_exhaustive_sample"], sx = leg_run(n_ex, Bq=1, acc=False)
            lx = legs["nn_exhaustive_sample"]
            pairs_x = TOP_K * pts_q * mean_pts
            tf = FLOP_PER_PAIR * pairs_x / (lx["nn_launch_ms"] * 1e-3) / 1e12
            lx.update({"roofline": {"kernel": "gloc::reg::nn_kernel", "bound": "mfma", "achieved": tf, "peak": PEAK_FP32_TFLOPS,
                                    "unit": "TFLOP/s", "frac": tf / PEAK_FP32_TFLOPS,
                                    "note": f"{FLOP_PER_PAIR} flop x {pairs_x:.3e} pairs per launch (20 jobs) / {lx['nn_launch_ms']:.2f} ms; "
                                            "the un-fused reference arithmetic cannot use FMA, which the peak counts as 2 flop"},
                       "same_selection_as_culled": bool(sx == [sels[i * per_step] for i in range(n_ex)]),
                       "what": "GLOC_REG_NN_EXHAUSTIVE, one query (20 candidates) per batch; correspondences, distances and selections identical to the "
                               "culled search bit for bit, poses to 1e-5 (tests/test_reg_gpu.py::test_every_pass_bit_identical_to_the_brute_force_kernel)"})
            lx.pop("pairs_evaluated_per_source", None)
            for r_ in regs:
                r_.set_option(capi.REG_OPT_NN_MODE, capi.REG_NN_CULLED if args.nn_mode == "culled" else capi.REG_NN_EXHAUSTIVE)

            # the places the leg's queries retrieve (a data leg gives them other scans for its duration)
            leg_q = q_desc_host[:L * per_step].to(dev)
            leg_c, _ = knn.search(leg_q, TOP_K)
            leg_places = sorted(set(int(g) % n_store for g in leg_c.cpu().numpy().reshape(-1) if g >= 0))

            def with_override(make):
                for g in leg_places:
                    scan_override[g] = make(g)
                if not args.no_target_index:
                    store.build_target_index_batch([scan_override[g][0] for g in leg_places])

            def drop_override():
                for g, (sid, _) in list(scan_override.items()):
                    store.release(sid)
                scan_override.clear()

            def coarse_leg(recall_defined):
                """The reference's own order (loop_detector.cpp:192-288 then icp_match_3d): the coarse (x, y, yaw, scale) match
                of every (query, candidate) pair on their BEV grids seeds the 3-D registration.  Grids are made for the
                places the leg's queries retrieve, from the scans the leg registers."""
                nonlocal cm, place_grid
                cm = capi.CoarseMatcher(local_rank)
                place_grid = np.full(n_store, NO_GRID, np.uint32)
                sids = [int(scan_override[g][0]) if g in scan_override else int(place_scan[g]) for g in leg_places]
                for i in range(0, len(sids), 256):
                    place_grid[leg_places[i:i + 256]] = cm.add_store_scans(store, sids[i:i + 256])
                coarse_stat["pairs"] = coarse_stat["accepted"] = 0
                # acceptance as in the reference: the 2-D match decides which candidates are registered at all; the 3-D
                # step keeps its inlier-ratio test and the convergence check
                cur["params"] = capi.default_reg_params(ransac_iters=RANSAC_ITERS, icp_iters=ICP_ITERS, min_inlier_ratio=MIN_INLIER_RATIO,
                                                        max_rmse=0.0, max_final_step=MAX_FINAL_STEP)
                out, _ = leg_run(L, recall_defined=recall_defined)
                cur["params"] = params
                out["coarse_pairs_accepted"] = coarse_stat["accepted"] / max(coarse_stat["pairs"], 1)
                cm.close()
                cm, place_grid = None, None
                cur_qgrids.clear()
                return out

            # leg 3: the reference NEVER registers unseeded -- its 2-D match (loop_detector.cpp:192-288) decides which candidates
            # are registered and seeds them.  The headline's data with that step in front.
            if cm is None:
                log("leg: coarse 2-D match seeds the registration (the headline's data)")
                legs["coarse_seeded"] = coarse_leg(True)
                legs["coarse_seeded"]["what"] = ("the headline's data with the reference's 2-D step in front (bench.py --coarse): per-query grid "
                                                 "construction + 500 pair matches per step + seeded registration of the accepted pairs")

            # leg 4: rounds 1-5's data, kept for continuity -- the retrieved places carry RIGID COPIES
            # (+-2 deg / +-0.3 m, 1 cm noise) of 24 + 6 ray-cast views of a small world instead of their own casts, the queries
            # rigid copies (+-1 deg / +-0.2 m) of 8 views beside them: 17 % easier than distinct casts (round 5 measured it)
            log("leg: rigid copies of 24 + 6 views (the data of rounds 1-5)")
            old_a, old_b = synth.make_world(1001), synth.make_world(2002)
            base_a = store.add_raycast(old_a, [pool_pose(s_) for s_ in range(POOL_A)], np.arange(3000, 3000 + POOL_A, dtype=np.uint64))
            base_b = store.add_raycast(old_b, [synth.se3(7.0 * s_, (1.5 * s_, -0.7 * s_, 0.0)) for s_ in range(POOL_B)],
                                       np.arange(5000, 5000 + POOL_B, dtype=np.uint64))
            qbase = store.add_raycast(old_a, [query_view_pose(v) for v in range(QUERY_VIEWS)], np.arange(9000, 9000 + QUERY_VIEWS, dtype=np.uint64))
            with_override(lambda g: (store.add_variant(base_b[(g // NEG_EVERY) % POOL_B] if is_negative(g) else base_a[g % POOL_A],
                                                       place_perturbation(g), 0.01, seed=PLACE_SEED + g),
                                     far_away_pose() if is_negative(g) else pool_pose(g % POOL_A) @ np.linalg.inv(place_perturbation(g))))
            saved_q = {}
            for j in range(L * per_step):
                v = ((int(q_place[j]) % POOL_A) // 3) % QUERY_VIEWS          # the query view beside the place's pool view (+-2 views)
                sid = store.add_variant(qbase[v], query_perturbation(j), 0.01, seed=QUERY_SEED + j)
                saved_q[j] = q_scan_host[j]
                q_scan_host[j] = torch.from_numpy(store.download(sid)).pin_memory()
                store.release(sid)
                query_pose_override[j] = query_view_pose(v) @ np.linalg.inv(query_perturbation(j))
            legs["data_rigid_copies"], _ = leg_run(L)
            legs["data_rigid_copies"]["what"] = (f"rounds 1-5's headline data: every retrieved place carries a rigid variant (+-2 deg / +-0.3 m, 1 cm noise) "
                                                 f"of one of {POOL_A} + {POOL_B} ray-cast views ({POOL_A} along a 4.6 m drive through a small world, {POOL_B} "
                                                 "of a different one), a query is a rigid variant of a view 0.3-0.6 m from its place's; identity prior, "
                                                 "the headline's parameters.  NOT what SURVEY cfg D describes; round 5: 626 q/s against 513 on distinct casts")
            for j, t_ in saved_q.items():
                q_scan_host[j] = t_
            query_pose_override.clear()
            drop_override()
            for sid in base_a + base_b + qbase:
                store.release(sid)

            # leg: the convergence check on data its value was not chosen on (tools/gate_holdout.py)
            if gate_views:
                log("leg: held-out check of max_final_step (worlds 3003 / 4004)")
                gh = gate_holdout.run(gate_views, device=local_rank, ransac_iters=RANSAC_ITERS, icp_iters=ICP_ITERS, min_inlier_ratio=MIN_INLIER_RATIO)
                gh.pop("rows", None)
                at = gh["thresholds"][f"{MAX_FINAL_STEP:g}"]
                legs["gate_holdout"] = {**gh, "max_final_step": MAX_FINAL_STEP, "success_rate": at["success"] / gh["queries"],
                                        "located_but_wrong": at["located_but_wrong"],
                                        "what": "gloc_reg_params.max_final_step on data its value was NOT chosen on: worlds 3003 / 4004, 8 query views, "
                                                "20 ranked candidates each (the place re-cast 0.7 m away, places 3-21 m along the drive, cfg-C-perturbed "
                                                "copies, 4 other-world views), coarse 2-D match in front as the reference, check off during the run and "
                                                "every threshold applied to the final steps afterwards (first success in rank order); right_pose_* / "
                                                "wrong_pose_*: final steps of the registrations the inlier test accepts, by whether the pose is within "
                                                "1 m / 5 deg"}

    sub_records = None
    if run_legs and not args.only_lone:
        log("sub-records: kNN cfg B, one shard of cfg E")
        sub_records = {"cfgC_lone_query": lone}

        def knn_record(n_rows, nq, reps_, pipelined=False):
            ix = capi.KnnIndex(DIM, device=local_rank)
            ix.reserve(n_rows)
            ix.add_synthetic(1, 2003, 0, n_rows)
            qh = synth.queries_near(2003, (np.arange(nq) * 131 + 7) % n_rows, DIM)
            qd_ = torch.from_numpy(qh).to(dev)
            oi = torch.empty((nq, TOP_K), dtype=torch.int64, device=dev)
            od = torch.empty((nq, TOP_K), dtype=torch.float32, device=dev)
            torch.cuda.synchronize()
            for _ in range(3):
                ix.search_device(qd_.data_ptr(), nq, TOP_K, oi.data_ptr(), od.data_ptr())
            ix.synchronize()
            t0 = time.time()
            for _ in range(reps_):
                ix.search_device(qd_.data_ptr(), nq, TOP_K, oi.data_ptr(), od.data_ptr())
            ix.synchronize()
            us = (time.time() - t0) / reps_ * 1e6
            # two search handles over the same rows (gloc_knn_create_view), each on its own stream, searches alternating: the
            # selection + re-rank of one (a work-group per query: 64 of 256 CUs) runs under the other's distance kernel
            us_pipe = None
            if pipelined:
                vw = ix.view()
                oi2, od2 = torch.empty_like(oi), torch.empty_like(od)
                qd2 = torch.from_numpy(synth.queries_near(2003, (np.arange(nq) * 97 + 29) % n_rows, DIM)).to(dev)   # (another batch of queries)
                for _ in range(3):
                    ix.search_device(qd_.data_ptr(), nq, TOP_K, oi.data_ptr(), od.data_ptr())
                    vw.search_device(qd2.data_ptr(), nq, TOP_K, oi2.data_ptr(), od2.data_ptr())
                ix.synchronize()
                vw.synchronize()
                t0 = time.time()
                for _ in range(reps_ // 2):
                    ix.search_device(qd_.data_ptr(), nq, TOP_K, oi.data_ptr(), od.data_ptr())
                    vw.search_device(qd2.data_ptr(), nq, TOP_K, oi2.data_ptr(), od2.data_ptr())
                ix.synchronize()
                vw.synchronize()
                us_pipe = (time.time() - t0) / (2 * (reps_ // 2)) * 1e6
                vw.close()
            ix.set_option(capi.KNN_OPT_PROFILE, 1)
            ix.profile_reset()
            for _ in range(10):
                ix.search_device(qd_.data_ptr(), nq, TOP_K, oi.data_ptr(), od.data_ptr())
            kern = {n: ix.profile(n)[0] / 10 * 1e3 for n in ("split_queries", "dist_mfma", "dist_exact", "select", "select_rerank", "rerank")}
            st_ = ix.stats()
            ix.close()
            return us, kern, st_, us_pipe

        us, kern, st_, us_pipe = knn_record(KNN_CFGB[1], KNN_CFGB[0], 50, pipelined=True)
        flop = 2.0 * KNN_CFGB[0] * KNN_CFGB[1] * DIM
        roof_s, fp32_s = knn_roofline_s(KNN_CFGB[1], KNN_CFGB[0], DIM)
        sub_records["knn_cfgB"] = {
            "us_per_search": us, "us_per_search_pipelined": us_pipe,
            "pipelined_what": "two search handles over the same rows (gloc_knn_create_view), a stream each, 2 x 25 searches of two query "
                              "batches issued alternately: THROUGHPUT of back-to-back searches (one's select + re-rank under the other's "
                              "distance kernel), not the latency of one -- us_per_search is that",
            "queries": KNN_CFGB[0], "rows": KNN_CFGB[1], "dim": DIM, "kernel_us": kern,
            "roofline_us": roof_s * 1e6, "frac_of_roofline": roof_s * 1e6 / us,
            "fp32_mfma_roofline_us": fp32_s * 1e6, "frac_of_fp32_mfma_roofline": fp32_s * 1e6 / us,
            "frac_of_round3_roofline": fp32_s * 1e6 / us,   # (the key of the shard record; VERDICT r3 item 2 asked >= 0.55 of THIS roofline)
            "coarse_kernel_tflops": flop / (kern["dist_mfma"] * 1e-6) / 1e12 if kern["dist_mfma"] > 0 else None,
            "coarse_kernel_hbm_gbs": 4.0 * KNN_CFGB[1] * DIM / (kern["dist_mfma"] * 1e-6) / 1e9 if kern["dist_mfma"] > 0 else None,
            # the coarse kernel against three peaks: the fp32 MFMA peak it no longer runs on (by the 2 Q N D flop of the product --
            # round 3's mfma_kernel_frac_of_peak, 0.52 then), the bf16 MFMA peak by the 3 x 2 Q N D it issues, and HBM
            "coarse_kernel_frac_of_fp32_mfma_peak": flop / (kern["dist_mfma"] * 1e-6) / 1e12 / PEAK_FP32_TFLOPS if kern["dist_mfma"] > 0 else None,
            "coarse_kernel_frac_of_bf16_mfma_peak": 3.0 * flop / (kern["dist_mfma"] * 1e-6) / 1e12 / PEAK_BF16_TFLOPS if kern["dist_mfma"] > 0 else None,
            "coarse_kernel_frac_of_hbm": 4.0 * KNN_CFGB[1] * DIM / (kern["dist_mfma"] * 1e-6) / 1e9 / PEAK_HBM_GBS if kern["dist_mfma"] > 0 else None,
            "queries_fallback": st_["queries_fallback"],
            # SURVEY 8d asked that cache-resident runs be labelled: 164 MB of rows < the 256 MiB Infinity Cache and the searches run
            # back to back, so after the first one the rows most likely come from it, not from HBM; FETCH_SIZE (185 MB per coarse
            # launch after the guide's doubling, profiles/r05_knn_cfgB_pmc.json) counts Infinity-Cache hits too and cannot tell
            "infinity_cache_resident": True,
            "infinity_cache_note": "the 164 MB database fits the 256 MiB Infinity Cache and the timed searches run back to back; the L2-miss "
                                   "counter (FETCH_SIZE, 185 MB per coarse launch) includes Infinity-Cache hits, so residency is inferred, not counted",
            "what": "BASELINE configs[1]: 64 queries x 10 000 x 4096 fp32, device resident, wall clock over 50 back-to-back searches. "
                    "Round 4: the coarse pass runs on the bf16 matrix cores (operands split in two bf16 values, three MFMAs per "
                    "product, a proven bound on what that drops; results bit-identical), so the search's roofline is the rows' "
                    "stream from HBM (roofline_us), no longer the fp32 MFMA time rounds 1 - 3 priced against "
                    "(fp32_mfma_roofline_us, kept beside it)"}
        sh = {}
        for nq in (1, 64):
            us, kern, st_, _ = knn_record(KNN_SHARD_ROWS, nq, 50)
            byts = 4.0 * KNN_SHARD_ROWS * DIM
            flop = 2.0 * nq * KNN_SHARD_ROWS * DIM
            roof_s, fp32_s = knn_roofline_s(KNN_SHARD_ROWS, nq, DIM)
            sh[f"q{nq}"] = {"us_per_search": us, "kernel_us": kern,
                            "hbm_gbs": byts / (us * 1e-6) / 1e9, "tflops": flop / (us * 1e-6) / 1e12,
                            "roofline_us": roof_s * 1e6, "frac_of_roofline": roof_s / (us * 1e-6),
                            "frac_of_round3_roofline": max(byts / (PEAK_HBM_GBS * 1e9), fp32_s) / (us * 1e-6)}
        sub_records["knn_shard_125k"] = {**sh, "rows": KNN_SHARD_ROWS, "dim": DIM,
                                         "what": "one of the 8 shards of BASELINE configs[4] (1M x 4096): the per-rank search before the "
                                                 "all-gather of the top-k lists; roofline = the shard's stream from HBM at 8 TB/s (three bf16 "
                                                 "MFMAs per product are far below it); frac_of_round3_roofline = against max(HBM, fp32 MFMA time), "
                                                 "the round-3 definition"}

    # ---- N > 1: BASELINE configs[4] in the line the driver runs -- the 1 M x 4096 database row-interleaved over the
    # ranks (1 M / N rows generated on each device), searched through gloc_knn_search_sharded (local top-k with global
    # indices -> ONE grouped RCCL all-gather of the lists over xGMI -> K3 merge, all below the C ABI, one stream) ----
    if world > 1 and args.cfge_rows > 0 and not args.no_legs:
        log(f"sub-record: configs[4], {args.cfge_rows} x {DIM} over {world} ranks")
        rows_total, per_rank = args.cfge_rows, args.cfge_rows // world
        CFGE_SEED = 5001
        ixe = capi.KnnIndex(DIM, device=local_rank)
        ixe.reserve(per_rank)
        ixe.add_synthetic(1, CFGE_SEED, rank, per_rank, row_stride=world)       # global rows rank, rank + N, ...
        q_rows = (np.arange(64, dtype=np.int64) * 15485 + 977) % (per_rank * world)
        qe = torch.from_numpy(synth.queries_near(CFGE_SEED, q_rows, DIM)).to(dev)
        ref_e = sharded.ShardedKnn(rank, world, sharded.hip_local_search(ixe), sharded.hip_merge(local_rank), comm_device=comm_dev)
        ri, rd = ref_e.search(qe[:8], TOP_K)                                    # the same search over torch.distributed
        torch.cuda.synchronize()
        rec = {"rows": per_rank * world, "rows_per_rank": per_rank, "dim": DIM, "ranks": world, "top_k": TOP_K,
               "what": "BASELINE configs[4]: synthetic 1M-place x 4096-D descriptor DB row-interleaved over the ranks, "
                       "queries replicated, per-shard top-k all-gathered over xGMI and merged on every rank"}
        own_row_first = bool((ri[:, 0].cpu().numpy() == q_rows[:8]).all())       # a query sits next to its own row
        if capi_knn is not None:
            cke = sharded.CapiShardedKnn(ixe, comm)
            ci, cd = cke.search(qe[:8], TOP_K)
            torch.cuda.synchronize()
            same = (bool((ci.cpu() == ri.cpu()).all()) and bool((cd.cpu().view(torch.int32) == rd.cpu().view(torch.int32)).all()))
            flag = torch.tensor([1 if (same and own_row_first) else 0], device=comm_dev or dev)
            dist.all_reduce(flag, op=dist.ReduceOp.MIN)
            rec["equal_to_torch_gathered_path_on_8_queries"] = bool(int(flag.item()))
            search_e = cke.search
            rec["transport"] = collectives
        else:
            rec["equal_to_torch_gathered_path_on_8_queries"] = None
            search_e = ref_e.search
            rec["transport"] = collectives
        rec["query_finds_its_own_row_first"] = own_row_first
        for nq_e, reps_e in ((1, 30), (64, 20)):
            qq = qe[:nq_e].contiguous()
            for _ in range(3):
                search_e(qq, TOP_K)
            fence()
            t0 = time.time()
            for _ in range(reps_e):
                search_e(qq, TOP_K)
            torch.cuda.synchronize()
            te = torch.tensor([(time.time() - t0) / reps_e], dtype=torch.float64, device=comm_dev or dev)
            dist.all_reduce(te, op=dist.ReduceOp.MAX)
            us_e = float(te.item()) * 1e6
            stage_e = None
            if capi_knn is not None:                                            # per-stage HIP events inside the C-ABI call
                ixe.set_option(capi.KNN_OPT_PROFILE, 1)
                ixe.profile_reset()
                for _ in range(10):
                    search_e(qq, TOP_K)
                torch.cuda.synchronize()
                stage_e = {n_: ixe.profile(n_)[0] / 10 * 1e3 for n_ in ("shard_local", "shard_gather", "shard_merge")}
                ixe.set_option(capi.KNN_OPT_PROFILE, 0)
            byts = 4.0 * per_rank * DIM
            flop = 2.0 * nq_e * per_rank * DIM
            roof = knn_roofline_s(per_rank, nq_e, DIM)[0] * 1e6
            rec[f"q{nq_e}"] = {"us_per_search": us_e, "queries_per_s": nq_e / (us_e * 1e-6), "stage_us_rank0": stage_e,
                               "per_rank_roofline_us": roof, "frac_of_roofline": roof / us_e,
                               "timing": f"wall clock over {reps_e} back-to-back searches, barrier + synchronize on both sides, max over ranks"}
        ixe.close()
        del qe
        torch.cuda.empty_cache()
        if rank == 0:
            sub_records = dict(sub_records or {}, knn_cfgE_sharded=rec)

    q_per_rep = n_steps * per_step
    out = {
        "metric": "localization queries/sec (kNN+top-20 reg), KITTI-00-sized DB",
        "value": q_per_rep / elapsed, "unit": "queries/s", "n_gpus": world, "steps": n_steps,
        "warmup": n_warm, "ms_per_step": elapsed / n_steps * 1e3, "higher_is_better": True,
        "scaling": "weak" if args.mode == "throughput" else "strong", "vs_baseline": None, "dtype": "f32",
        "data": "synthetic",
        "config": {"workload": f"cfgD: KITTI-00-sized DB {n_places}x{DIM} fp32, {q_per_rep} queries ({n_steps} steps x {per_step}): scan H2D+index, "
                               f"descriptor H2D -> top-{TOP_K} -> {TOP_K} candidate scans (~{mean_pts:.0f} pts) x (RANSAC {RANSAC_ITERS} adaptive + ICP "
                               f"{ICP_ITERS}); {n_store} places = DISTINCT device ray-casts along a {TRAJ_LENGTH_M * n_store / N_PLACES_1GPU:.0f} m loop "
                               "through one world, 1 in 4 cast in another world; a query = its own cast <= 0.65 m / 3 deg off its place",
                   "queries_per_step": per_step, "queries_per_batch_per_gpu": B, "queries_per_repetition": q_per_rep,
                   "repetitions": n_reps, "repetition_seconds": rep_s, "value_is": "median repetition",
                   "places": n_places, "dim": DIM, "top_k": TOP_K, "points_per_scan": int(mean_pts),
                   "scan_store_scans": int(store_scans_resident), "scan_store_distinct_places": int(n_store),
                   "scan_store_gib": live_b / 2**30, "database_scans_target_index": "kd order" if not args.no_target_index else "curve order",
                   "negatives_per_query": (TOP_K // NEG_EVERY) if neg_on else 0,
                   "query_prep_in_timed_region": True,
                   "query_prep": "the step's query scans: H2D from pinned host memory + device indexing in one launch sequence "
                                 "(gloc_scan_store_add_batch) + descriptor H2D, "
                                 + ("inline" if args.no_prefetch else "prefetched one step ahead on a second host thread + stream"),
                   "ransac_iters_cap": RANSAC_ITERS, "ransac_confidence": float(params.ransac_confidence),
                   "min_inlier_ratio": MIN_INLIER_RATIO, "max_rmse": MAX_RMSE, "max_final_step": MAX_FINAL_STEP,
                   "max_final_step_origin": "chosen by a sweep over this bench's own synthetic legs in round 4 (tuned on this data); "
                                            "the library default is off; legs.gate_holdout is its check on unseen worlds / views / poses",
                   "icp_iters": ICP_ITERS, "nn_passes_per_query": passes,
                   "nn_mode": args.nn_mode, "collectives": collectives, "rccl_ranks_seen": rccl_ranks_seen,
                   "coarse_2d_match": bool(args.coarse),
                   "registration_pipeline": ("two handles on one stream: batch i + 1 enqueued before batch i's results are waited "
                                             "for (gloc_reg_batch_multi_begin / _end)") if pipeline else "none",
                   "parallelism": (f"1 gpu, {B} queries registered per batch on one stream") if world == 1 else (
                       f"{B} queries per gpu per step; "
                       f"db rows interleave-sharded over {world} ranks (all-gather of per-shard top-k, merge); "
                       + ("each rank registers its queries locally, result tables all-gathered"
                          if args.mode == "throughput" else
                          "one query, candidates sharded over the ranks, result rows gathered"))},
        "accuracy": accuracy,
        "roofline": roofline,
        "stage_ms_per_step_rank0": {**{k_: v / n_steps for k_, v in stage_ms.items()},
                                    **{"host_" + k_: v / n_steps * 1e3 for k_, v in stage.items()}},
        "selected_candidate_rank_histogram": {str(k_): int(v) for k_, v in
                                              zip(*np.unique(np.asarray(sels), return_counts=True))},
        "first_success_mode": first_success,
        "legs": legs or None,
        "sub_records": sub_records,
    }
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        log("timing the CPU checker (one whole query) ...")
        # query 0 of the stream and the 20 places it retrieved in the timed repetition (rank order)
        c0 = all_cand[0]
        cand_scans = [store.download(int(place_scan[int(g) % n_store])) for g in c0 if g >= 0]
        rows0 = all_tabs[0][:len(cand_scans)] if args.mode == "throughput" else None
        out["cpu_baseline"] = cpu_baseline(q_scan_host[0].numpy(), cand_scans, n_places, MIN_INLIER_RATIO, gpu_rows=rows0)
    elif rank == 0:
        out["cpu_baseline"] = None
    if world > 1:
        out["per_gpu_value"] = out["value"] / world      # (to set beside the N = 1 line)
        out["config"]["collectives_requested"] = "capi (RCCL below the C ABI)" if want_capi else f"torch.distributed ({args.backend})"
    # Top-level SCALAR copies of what the nested records hold (the driver's record keeps scalar keys only: three of the five
    # BASELINE configs lived in sub_records and never reached it -- VERDICT r4 item 6)
    sr = sub_records or ({"cfgC_lone_query": lone} if lone else {})
    flat = {"success_rate": accuracy["success_rate"], "nn_ms_per_step": (stage_ms.get("nn", 0.0) / n_steps) or None,
            "nn_launch_ms": roofline.get("launch_ms") if roofline else None, "roofline_frac": roofline.get("frac") if roofline else None}
    if "knn_cfgB" in sr:
        flat.update(knn_cfgB_us=sr["knn_cfgB"]["us_per_search"], knn_cfgB_frac=sr["knn_cfgB"]["frac_of_roofline"],
                    knn_cfgB_pipelined_us=sr["knn_cfgB"]["us_per_search_pipelined"],
                    knn_cfgB_fallbacks=sr["knn_cfgB"]["queries_fallback"], knn_cfgB_infinity_cache_resident=sr["knn_cfgB"]["infinity_cache_resident"])
    if "knn_shard_125k" in sr:
        flat.update(knn_shard125k_q64_us=sr["knn_shard_125k"]["q64"]["us_per_search"], knn_shard125k_q64_frac=sr["knn_shard_125k"]["q64"]["frac_of_roofline"],
                    knn_shard125k_q1_us=sr["knn_shard_125k"]["q1"]["us_per_search"])
    if sr.get("cfgC_lone_query"):
        flat.update(lone_query_ms=sr["cfgC_lone_query"]["ms_per_query"], lone_query_ms_launch_by_launch=sr["cfgC_lone_query"]["ms_per_query_launch_by_launch"],
                    lone_query_nn_launch_ms=sr["cfgC_lone_query"]["nn_launch_ms"],
                    lone_query_nn_cold_launch_ms=sr["cfgC_lone_query"]["nn_cold_launch_ms"])
    if "knn_cfgE_sharded" in sr:
        flat.update(knn_cfgE_q64_us=sr["knn_cfgE_sharded"]["q64"]["us_per_search"], knn_cfgE_q1_us=sr["knn_cfgE_sharded"]["q1"]["us_per_search"],
                    knn_cfgE_equal_to_torch_path=sr["knn_cfgE_sharded"]["equal_to_torch_gathered_path_on_8_queries"])
    if world > 1:
        flat.update(rccl_ranks_seen=rccl_ranks_seen)
    for leg, key in (("coarse_seeded", "coarse_seeded_qps"), ("data_rigid_copies", "rigid_copies_qps"), ("ransac_all_3000", "ransac_all_3000_qps")):
        if legs and leg in legs and "value" in legs[leg]:
            flat[key] = legs[leg]["value"]
    flat["located_but_wrong"] = accuracy["located_but_wrong_count"]
    flat["pairs_per_source"] = roofline.get("pairs_evaluated_per_source") if roofline else None
    if legs and "coarse_seeded" in legs:
        flat["coarse_seeded_success_rate"] = legs["coarse_seeded"]["accuracy"]["success_rate"]
        flat["coarse_seeded_located_but_wrong"] = legs["coarse_seeded"]["accuracy"]["located_but_wrong_count"]
    if legs and "data_rigid_copies" in legs:
        flat["rigid_copies_success_rate"] = legs["data_rigid_copies"]["accuracy"]["success_rate"]
    if legs and "gate_holdout" in legs:
        flat.update(gate_holdout_success=legs["gate_holdout"].get("success_rate"), gate_holdout_wrong=legs["gate_holdout"].get("located_but_wrong"))
    from gloc3d_amd import build as _build
    bf = _build.build_flags()
    flat["build_mfma_vgpr_form"] = None if bf is None else bool(bf.get("mfma_vgpr_form"))
    # the switches the library was REALLY compiled with (side-cars of its objects): a developer build that returns early from
    # the culled search (-DGLOC_NN_RET=..) or duplicates work (-DGLOC_NN_DUP_..) computes wrong results fast -- no headline from it
    flat["build_extra_flags"] = None if bf is None else " ".join(bf.get("extra_flags") or [])[:80]
    flat["build_defines"] = None if bf is None else " ".join(bf.get("defines") or [])[:80]
    dev_build = [d for d in ((bf or {}).get("defines") or []) if d.startswith(("-DGLOC_NN_RET", "-DGLOC_NN_DUP", "-DGLOC_SOLVE_RET"))]
    if dev_build and not os.environ.get("GLOC3D_LIB_PATH"):
        log(f"REFUSED: libgloc3d.so was compiled with {dev_build}: a developer build with wrong results; rebuild without GLOC3D_EXTRA_FLAGS")
        sys.exit(5)
    out.update({k_: v for k_, v in flat.items() if k_ not in out})
    if rank == 0:
        emit(out)
    wrong_transport = want_capi and capi_knn is None
    if world > 1:
        dist.destroy_process_group()
    if wrong_transport:
        log("FAILED: the C-ABI RCCL transport was requested (--collectives capi) but the run fell back to torch.distributed "
            "(its communicator or self-tests failed, see above); pass --collectives torch to accept that path")
        sys.exit(4)
    if accuracy["success_rate"] < args.min_success:
        log(f"FAILED: registration success rate {accuracy['success_rate']:.3f} < {args.min_success} on the timed stream")
        sys.exit(3)


if __name__ == "__main__":
    main()
