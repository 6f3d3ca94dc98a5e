#!/usr/bin/env python3
"""bench.py -- localization queries/sec (kNN + top-20 registration) on MI355X.

    python bench.py --gpus 1 --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

One STEP = one localization query through the hot path, everything resident in HBM when the timed
region starts:  descriptor top-20 (gloc_knn_search_device) -> the 20 retrieved candidate scans ->
batched RANSAC(3000) + ICP(20) registration (gloc_reg_batch_ids) -> lowest-rank successful candidate.

N = 1  (BASELINE.json configs[3], the configuration the metric is quoted on):
        KITTI-00-sized database, 4541 places x 4096-D, ~124k-point scans, one query per step.
N > 1  the same database, interleave-sharded over the N ranks; a step handles N queries (one per
        rank): their descriptors are searched on every shard, the per-shard top-k lists all-gathered
        over RCCL/xGMI and merged on every rank; rank r then registers query r's 20 candidates against
        its replica of the scan store, and the N result tables are all-gathered.  Per-GPU work is
        fixed as N grows ("weak" scaling).  --places 1000000 gives BASELINE.json configs[4]'s sharded
        database; --mode latency shards ONE query's candidates over the ranks instead.

Prints ONE JSON line (rank 0) with the `roofline` object of the dominant kernel (K4 point-NN,
HIP-event timed inside the timed region on the stream it runs on) and the `cpu_baseline` object
(the CPU oracle timed on this box's host cores, rank 0, N = 1 only, bounded sample).
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

DIM = 4096
TOP_K = 20
N_PLACES_1GPU = 4541          # KITTI odometry 00 (dataset/kitti_i2i.py:46 of the reference)
SCAN_POOL = 24                # distinct synthetic scans; place g carries scan g % SCAN_POOL
QUERY_POOL = 4
RANSAC_ITERS = 3000           # registration/loop_detector.cpp:257 (cap; adaptive stop at the
                              # reference's OpenCV default confidence 0.99, see gloc_reg_params)
ICP_ITERS = 20                # BASELINE.json configs[2]
DB_SEED = 4001
PEAK_FP32_TFLOPS = 157.3      # MI355X_MICROARCH.md: fp32 MFMA = fp32 vector peak
FLOP_PER_PAIR = 8             # 3 sub + 3 mul + 2 add per (source, target) pair


def log(msg):
    if int(os.environ.get("RANK", "0")) == 0:
        print(f"[bench] {msg}", file=sys.stderr, flush=True)


def build_scans(n_pool, n_query):
    """Procedural world, a short drive: pool scan s at pose P_s, query scans between poses."""
    from gloc3d_amd import synth
    w = synth.make_world(1001)
    pool, qs = [], []
    for s in range(n_pool):
        T = synth.se3(0.8 * (s - n_pool / 2), (0.5 * s, 0.1 * s, 0.0))
        pool.append(np.ascontiguousarray(synth.lidar_scan(w, T, seed=3000 + s)[:, :3]))
    for s in range(n_query):
        j = (s * 5 + 3) % n_pool
        T = synth.se3(0.8 * (j - n_pool / 2) + 1.5, (0.5 * j + 0.3, 0.1 * j - 0.2, 0.02))
        qs.append(np.ascontiguousarray(synth.lidar_scan(w, T, seed=9000 + s)[:, :3]))
    return pool, qs


def cpu_baseline(pool, qscans, n_places, gpu_check=None):
    """The CPU oracle on this host, 1 thread (the reference's kNN and registration are
    single-threaded): kNN of one query + full registration of ONE of its 20 candidates,
    extrapolated to 20 candidates."""
    import oracle
    from gloc3d_amd import synth
    oracle.build(ref=False)
    db = synth.descriptors_traj(DB_SEED, 0, n_places, DIM)
    q = synth.queries_near(DB_SEED, [1234], DIM)
    t0 = time.time()
    oracle.knn_search(db, q, TOP_K)
    t_knn = time.time() - t0
    t0 = time.time()
    o = oracle.reg_one(qscans[0], pool[3], cand_id=0, ransac_iters=RANSAC_ITERS, icp_iters=ICP_ITERS)
    t_cand = time.time() - t0
    per_query = t_knn + TOP_K * t_cand
    extra = {}
    if gpu_check is not None:
        # the oracle as the checker: the same full-size registration through the HIP path
        g = gpu_check(qscans[0], pool[3])
        extra["full_size_parity"] = {"pose_max_abs_diff": float(np.abs(g["T"][0] - o["T"]).max()),
                                     "inliers_equal": bool(g["inliers"][0] == o["inliers"]),
                                     "ok_equal": bool(g["ok"][0] == o["ok"]),
                                     "rmse_abs_diff": float(abs(g["rmse"][0] - o["rmse"]))}
    if oracle.have_ref():
        # the same nearest-neighbour pass on the REFERENCE's vendored nanoflann kd-tree (oracle/_ref),
        # and what the query would cost with it in place of the oracle's grid search
        t0 = time.time()
        oracle.ref_nn3(qscans[0], pool[3])
        t_ref = time.time() - t0
        t0 = time.time()
        oracle.nn3(qscans[0], pool[3], grid=True)
        t_port = time.time() - t0
        passes = 1 + ICP_ITERS
        extra.update({"nn_pass_s_port_grid": t_port, "nn_pass_s_reference_kdtree": t_ref,
                      "value_with_reference_nn": 1.0 / (t_knn + TOP_K * max(t_cand - passes * (t_port - t_ref), 0.0))})
    return {**extra, "value": 1.0 / per_query, "unit": "queries/s", "cores": 1, "kind": "port",
            "sample": f"1 query: kNN over {n_places}x{DIM} ({t_knn*1e3:.0f} ms) + RANSAC{RANSAC_ITERS}"
                      f"+ICP{ICP_ITERS} registration of 1 of its {TOP_K} candidates "
                      f"({t_cand:.1f} s, ~124k-pt scans), extrapolated x{TOP_K} candidates",
            "host_cpus": os.cpu_count()}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--places", type=int, default=N_PLACES_1GPU,
                    help="database size (1000000 = BASELINE.json configs[4], sharded over the ranks)")
    ap.add_argument("--mode", choices=["throughput", "latency"], default="throughput",
                    help="N > 1: one query per rank per step (weak scaling) or one query per step "
                         "with its candidates sharded over the ranks (strong scaling)")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend (nccl = RCCL)")
    ap.add_argument("--same-device", action="store_true",
                    help="rehearsal: all ranks on GPU 0 (use with --backend gloo)")
    ap.add_argument("--inflight", type=int, default=3,
                    help="throughput mode: queries in flight per GPU (one registration handle, HIP stream and "
                         "host thread each); a step handles gpus x inflight queries")
    ap.add_argument("--nn-src-per-lane", type=int, default=0, help="culled 1-NN tuning (1, 2, 4)")
    ap.add_argument("--nn-mode", choices=["culled", "exhaustive"], default="culled",
                    help="1-NN search of the registration (identical results)")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch with torch.distributed.run --nproc-per-node N for --gpus N")
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
    inflight = max(1, args.inflight) if args.mode == "throughput" else 1
    per_step = world * inflight if args.mode == "throughput" else 1   # queries per step
    n_steps, n_warm = args.steps, args.warmup
    t_setup = time.time()

    # ---- resident state -----------------------------------------------------------------------
    index = capi.KnnIndex(DIM, device=local_rank)
    n_local = len(sharded.shard_rows(n_places, rank, world))
    index.reserve(n_local)
    index.add_synthetic(1, DB_SEED, rank, n_local, row_stride=world)   # this rank's interleaved shard
    index.synchronize()
    log(f"database: {n_places} x {DIM} ({n_local} rows on rank 0), generated on device")

    pool, qscans = build_scans(SCAN_POOL, QUERY_POOL)
    # one registration handle (own HIP stream, own scan store) per query in flight
    regs, pool_ids_k, q_ids_k = [], [], []
    for _ in range(inflight):
        r_ = capi.Registrar(device=local_rank)
        r_.set_option(capi.REG_OPT_PROFILE, 1)
        r_.set_option(capi.REG_OPT_NN_MODE,
                      capi.REG_NN_CULLED if args.nn_mode == "culled" else capi.REG_NN_EXHAUSTIVE)
        if args.nn_src_per_lane:
            r_.set_option(capi.REG_OPT_NN_SRC_PER_LANE, args.nn_src_per_lane)
        pool_ids_k.append([r_.scan_upload(p) for p in pool])
        q_ids_k.append([r_.scan_upload(q) for q in qscans])
        regs.append(r_)
    reg, pool_ids, q_ids = regs[0], pool_ids_k[0], q_ids_k[0]
    params = capi.default_reg_params(ransac_iters=RANSAC_ITERS, icp_iters=ICP_ITERS)
    mean_pts = float(np.mean([p.shape[0] for p in pool]))
    log(f"scans: {SCAN_POOL} pool + {QUERY_POOL} query scans, ~{mean_pts:.0f} pts each, resident")

    # queries: noisy copies of database places (replicated on every rank)
    total = (n_steps + n_warm) * per_step
    q_rows = (np.arange(total, dtype=np.int64) * 977 + 211) % n_places
    queries = torch.from_numpy(synth.queries_near(DB_SEED, q_rows, DIM)).to(dev)

    knn = sharded.ShardedKnn(rank, world, sharded.hip_local_search(index), sharded.hip_merge(local_rank),
                             comm_device=comm_dev)
    base_register = sharded.hip_local_register(reg, params)

    def local_register(q_id, local_rows, ranks):
        # place g = local_row * world + rank carries pool scan g % SCAN_POOL
        g = np.asarray(local_rows, np.int64) * world + rank
        return base_register(q_id, [pool_ids[int(x) % SCAN_POOL] for x in g], ranks)

    def make_register_all(k):
        reg_k = sharded.hip_local_register(regs[k], params)

        def fn(q_id, global_places, ranks):
            # every rank holds the whole scan pool: place g carries pool scan g % SCAN_POOL
            return reg_k(q_id, [pool_ids_k[k][int(g) % SCAN_POOL] for g in global_places], ranks)
        return fn

    register_all_k = [make_register_all(k) for k in range(inflight)]
    from concurrent.futures import ThreadPoolExecutor
    executor = ThreadPoolExecutor(max_workers=inflight)

    sreg = sharded.ShardedRegistrar(rank, world, local_register, comm_device=comm_dev)
    qreg = sharded.QueryParallelRegistrar(rank, world, register_all_k[0], comm_device=comm_dev)
    pairs_per_launch = []

    def step(i):
        q0 = i * per_step
        idx, d2 = knn.search(queries[q0:q0 + per_step], TOP_K)
        cand = idx.cpu().numpy()                          # [per_step, 20] global place ids, retrieval order
        if args.mode == "throughput":
            mys = [q0 + rank * inflight + k for k in range(inflight)]
            tables = qreg.register_many([q_ids_k[k][mys[k] % QUERY_POOL] for k in range(inflight)], cand, dev,
                                        register_all_k, executor)              # [world*inflight, 20, 19]
            sel = [sharded.ShardedRegistrar.select_first_ok(t) for t in tables]
            if i >= n_warm:
                for k in range(inflight):
                    mine = cand[rank * inflight + k]
                    nq = qscans[mys[k] % QUERY_POOL].shape[0]
                    pairs_per_launch.append(float(nq) * float(sum(pool[int(g) % SCAN_POOL].shape[0]
                                                                    for g in mine[mine >= 0])))
            return cand, sel
        else:
            table = sreg.register(q_ids[q0 % QUERY_POOL], cand[0], dev)
            sel = [sreg.select_first_ok(table)]
            mine = cand[0][(cand[0] >= 0) & (cand[0] % world == rank)]
            nq = qscans[q0 % QUERY_POOL].shape[0]
        if i >= n_warm:
            pairs_per_launch.append(float(nq) * float(sum(pool[int(g) % SCAN_POOL].shape[0] for g in mine)))
        return cand, sel

    def fence():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    log(f"setup {time.time() - t_setup:.1f} s; warmup {n_warm}, timing {n_steps} steps")
    for i in range(n_warm):
        cand, sel = step(i)
        assert (cand[:, 0] == q_rows[i * per_step:(i + 1) * per_step]).all(), "retrieval sanity: top-1 != query place"
    fence()
    for r_ in regs:
        r_.profile_reset()
    t0 = time.time()
    sels = []
    for i in range(n_warm, n_warm + n_steps):
        cand, sel = step(i)
        sels.extend(sel)
    fence()
    elapsed = time.time() - t0
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=comm_dev or dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    # ---- roofline of the dominant kernel (K4 point-NN), from the HIP events of the timed region --
    nn_ms = sum(r_.profile("nn")[0] for r_ in regs)
    nn_launches = sum(r_.profile("nn")[1] for r_ in regs)
    stage_ms = {n: sum(r_.profile(n)[0] for r_ in regs)
                for n in ("nn", "ransac_score", "ransac_hyp", "accum", "solve", "transform")}
    passes = 1 + ICP_ITERS
    pairs_eval = sum(r_.nn_stats()[0] for r_ in regs)
    # the same kernel with nothing else on the GPU (outside the timed region): three queries, one at a time
    serial_launch_ms = None
    if inflight > 1 and args.mode == "throughput":
        regs[0].profile_reset()
        for j in range(3):
            cands_j = [pool_ids_k[0][(j * 5 + c) % SCAN_POOL] for c in range(TOP_K)]
            regs[0].batch_ids(q_ids_k[0][j % QUERY_POOL], np.asarray(cands_j, np.uint32), params=params)
        ms_, n_ = regs[0].profile("nn")
        serial_launch_ms = ms_ / max(n_, 1)
    all_pairs = float(np.mean(pairs_per_launch)) if pairs_per_launch else 0.0   # exhaustive pair count
    avg_launch_s = (nn_ms / max(nn_launches, 1)) * 1e-3
    if args.nn_mode == "exhaustive":
        eval_pairs = all_pairs
    else:  # culled: pairs the kernel actually evaluated (its own counter)
        eval_pairs = float(pairs_eval) / max(nn_launches, 1)
    achieved = FLOP_PER_PAIR * eval_pairs / avg_launch_s / 1e12 if avg_launch_s > 0 else 0.0
    kname = "gloc::reg::nn_kernel" if args.nn_mode == "exhaustive" else "gloc::reg::nn_compact_kernel"
    traffic = None  # HBM bytes per launch from the committed PMC passes (profiles/), same workload
    pmc = os.path.join(ROOT, "profiles", "r01_pmc_traffic_nn_compact.json")
    if args.nn_mode == "culled" and world == 1 and os.path.exists(pmc):
        traffic = json.load(open(pmc))["hbm_bytes_per_launch"]
    roofline = {"kernel": kname, "bound": "mfma", "achieved": achieved,
                "peak": PEAK_FP32_TFLOPS, "unit": "TFLOP/s", "frac": achieved / PEAK_FP32_TFLOPS,
                "traffic": traffic,
                "pairs_evaluated_per_launch": eval_pairs, "pairs_exhaustive_per_launch": all_pairs,
                "exhaustive_equivalent_tflops": FLOP_PER_PAIR * all_pairs / avg_launch_s / 1e12
                if avg_launch_s > 0 else 0.0,
                "launch_ms_timed_region": avg_launch_s * 1e3, "launch_ms_alone": serial_launch_ms,
                "note": f"{FLOP_PER_PAIR} flop/pair x {eval_pairs:.3e} pairs EVALUATED per launch (rank 0; "
                        f"the exhaustive count is {all_pairs:.3e}) / {avg_launch_s*1e3:.3f} ms avg over "
                        f"{nn_launches} launches; fp32 peak (vector = MFMA) {PEAK_FP32_TFLOPS} TF"
                        + (f"; durations are HIP-event spans with {inflight} queries in flight: launches of "
                           f"different queries overlap on the GPU, so a span is longer than the kernel run alone"
                           if inflight > 1 else "")}

    out = {
        "metric": "localization queries/sec (kNN+top-20 reg), KITTI-00-sized DB",
        "value": n_steps * per_step / elapsed, "unit": "queries/s", "n_gpus": world, "steps": n_steps,
        "warmup": n_warm, "ms_per_step": elapsed / n_steps * 1e3, "higher_is_better": True,
        "scaling": "weak" if args.mode == "throughput" else "strong", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": f"cfgD: KITTI-00-sized DB {n_places}x{DIM} fp32, {per_step} quer"
                               f"{'y' if per_step == 1 else 'ies'}/step -> top-{TOP_K} -> {TOP_K} candidate "
                               f"scans (~{mean_pts:.0f} pts) x (RANSAC {RANSAC_ITERS} + ICP {ICP_ITERS})",
                   "queries_per_step": per_step, "queries_in_flight_per_gpu": inflight,
                   "places": n_places, "dim": DIM, "top_k": TOP_K, "points_per_scan": int(mean_pts),
                   "ransac_iters": RANSAC_ITERS, "icp_iters": ICP_ITERS, "nn_passes_per_query": passes,
                   "nn_mode": args.nn_mode, "ransac_confidence": float(params.ransac_confidence),
                   "parallelism": (f"1 gpu, {inflight} queries in flight (one registration handle + HIP stream + "
                                   f"host thread each)") if world == 1 else (
                       f"{inflight} queries in flight per gpu; "
                       f"db rows interleave-sharded over {world} ranks (all-gather of per-shard top-k, merge); "
                       + ("one query per rank registered locally, result tables all-gathered"
                          if args.mode == "throughput" else
                          "one query, candidates sharded over the ranks, all-reduce of poses"))},
        "roofline": roofline,
        "stage_ms_per_step_rank0": {k_: v / n_steps for k_, v in stage_ms.items()},
        "selected_candidate_rank": sels,
    }
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        log("timing the CPU oracle (bounded sample, ~30 s) ...")
        out["cpu_baseline"] = cpu_baseline(
            pool, qscans, n_places,
            gpu_check=lambda q_, c_: regs[0].batch(q_, [c_], params=params))
    elif rank == 0:
        out["cpu_baseline"] = None
    if rank == 0:
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
