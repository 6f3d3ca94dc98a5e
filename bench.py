#!/usr/bin/env python3
"""bench.py -- localization queries/sec (kNN + top-20 registration) on MI355X.

    python bench.py --gpus 1 --steps K --warmup W
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
       query's candidates over the ranks instead.

Prints ONE JSON line (rank 0) with the `roofline` object of the dominant kernel (K4 point-NN, HIP-event
timed inside the timed region on the one stream it runs on) and the `cpu_baseline` object (the CPU
checker timed on this box's host cores, rank 0, N = 1 only, bounded sample).
"""
import argparse
import json
import os
import sys
import threading
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

DIM = 4096
TOP_K = 20
N_PLACES_1GPU = 4541          # KITTI odometry 00 (dataset/kitti_i2i.py:46 of the reference)
POOL_A = 24                   # ray-cast views of world A along a short drive (0.2 m / 0.5 deg apart)
POOL_B = 6                    # ray-cast views of a different world (the negatives)
QUERY_VIEWS = 8               # ray-cast query views of world A, each next to pool view 3 * v + 1
NEG_EVERY = 4                 # place g carries a world-B scan iff g % 4 == 1 -> 5 of 20 consecutive places
RANSAC_ITERS = 3000           # registration/loop_detector.cpp:257 (cap; adaptive stop at the
                              # reference's OpenCV default confidence 0.99, see gloc_reg_params)
ICP_ITERS = 20                # BASELINE.json configs[2]
MIN_INLIER_RATIO = 0.3        # the library default (ok iff RANSAC inliers >= ratio x n) ...
MAX_RMSE = 1.0                # ... and the final RMS nearest-neighbour distance <= 1 m: both worlds share a
                              # ground plane, so a different-world candidate still has ~0.83 inliers at 0.6 m
                              # (positives 0.86-0.92) but ends at an rmse of 2.1-2.5 m (positives 0.17-0.7)
DB_SEED = 4001
PEAK_HBM_GBS = 8000.0         # MI355X_MICROARCH.md: HBM3E 8 TB/s spec (6.29 TB/s measured copy)
PEAK_FP32_TFLOPS = 157.3
FLOP_PER_PAIR = 8             # 3 sub + 3 mul + 2 add per (source, target) pair
BYTES_PER_POINT_INDEXED = 16  # sorted float4 (x, y, z, original index)


def log(msg):
    if int(os.environ.get("RANK", "0")) == 0:
        print(f"[bench] {msg}", file=sys.stderr, flush=True)


def pool_pose(s):
    from gloc3d_amd import synth
    return synth.se3(0.5 * (s - POOL_A / 2), (0.2 * s, 0.04 * s, 0.0))


def _cast_view(job):
    from gloc3d_amd import synth
    world_seed, T, seed = job
    return np.ascontiguousarray(synth.lidar_scan(synth.make_world(world_seed), T, seed=seed)[:, :3])


def build_views(cache=None):
    """Ray-cast the few base views on the host (numpy, a process per view; called BEFORE anything
    touches the GPU): world A pool, world B pool, query views.  `cache`: an .npz to load them from /
    save them to (profiling runs: rocprofv3 and forked workers do not mix)."""
    from concurrent.futures import ProcessPoolExecutor
    from gloc3d_amd import synth
    if cache and os.path.exists(cache):
        z = np.load(cache)
        views = [z[f"v{i}"] for i in range(POOL_A + POOL_B + QUERY_VIEWS)]
        return views[:POOL_A], views[POOL_A:POOL_A + POOL_B], views[POOL_A + POOL_B:]
    jobs = [(1001, pool_pose(s), 3000 + s) for s in range(POOL_A)]
    jobs += [(2002, synth.se3(7.0 * s, (1.5 * s, -0.7 * s, 0.0)), 5000 + s) for s in range(POOL_B)]
    jobs += [(1001, pool_pose(3 * v + 1) @ synth.se3(1.5, (0.3, -0.2, 0.02)), 9000 + v) for v in range(QUERY_VIEWS)]
    with ProcessPoolExecutor(max_workers=min(16, os.cpu_count() or 1)) as ex:
        views = list(ex.map(_cast_view, jobs))
    if cache:
        np.savez(cache, **{f"v{i}": v for i, v in enumerate(views)})
    return views[:POOL_A], views[POOL_A:POOL_A + POOL_B], views[POOL_A + POOL_B:]


def place_perturbation(g):
    """Per-place rigid perturbation (yaw +-2 deg, t +-0.3 m) from the counter RNG: every place's scan
    is a distinct cloud (distinct bits, distinct Hilbert order), not an alias of a pool scan."""
    from gloc3d_amd import synth
    key = synth.rng_key(DB_SEED ^ 0x5CA4, np.uint64(g))
    u = synth.rng_uniform(key, np.arange(4, dtype=np.uint64)).astype(np.float64) * 2 - 1
    return synth.se3(2.0 * u[0], (0.3 * u[1], 0.3 * u[2], 0.03 * u[3]))


def cpu_baseline(sample, n_places, min_inlier_ratio, gpu_check=None):
    """The CPU checker on this host: kNN of one query + full registration of candidates of that query
    (one positive, one negative), extrapolated to the 15 + 5 of a query; 1 thread (the reference's kNN
    and registration are single-threaded) and all cores (candidates over threads)."""
    import oracle
    from gloc3d_amd import synth
    oracle.build(ref=False)
    db = synth.descriptors_traj(DB_SEED, 0, n_places, DIM)
    q = synth.queries_near(DB_SEED, [1234], DIM)
    t0 = time.time()
    oracle.knn_search(db, q, TOP_K)
    t_knn = time.time() - t0
    qscan, pos, neg = sample["query"], sample["positive"], sample["negative"]
    kw = dict(ransac_iters=RANSAC_ITERS, icp_iters=ICP_ITERS, min_inlier_ratio=min_inlier_ratio, max_rmse=MAX_RMSE)
    use_ref = oracle.have_ref()
    t0 = time.time()
    o_pos = oracle.reg_one(qscan, pos, cand_id=0, ref_nn=use_ref, **kw)
    t_pos = time.time() - t0
    t0 = time.time()
    o_neg = oracle.reg_one(qscan, neg, cand_id=1, ref_nn=use_ref, **kw)
    t_neg = time.time() - t0
    n_neg = TOP_K // NEG_EVERY
    per_query = t_knn + (TOP_K - n_neg) * t_pos + n_neg * t_neg
    extra = {}
    if gpu_check is not None:
        # the oracle as the checker: the same two full-size registrations through the HIP path
        g = gpu_check(qscan, [pos, neg])
        extra["full_size_parity"] = {
            "pose_max_abs_diff": float(max(np.abs(g["T"][0] - o_pos["T"]).max(), np.abs(g["T"][1] - o_neg["T"]).max())),
            "inliers_equal": bool(g["inliers"][0] == o_pos["inliers"] and g["inliers"][1] == o_neg["inliers"]),
            "ok_equal": bool(g["ok"][0] == o_pos["ok"] and g["ok"][1] == o_neg["ok"]),
            "ok_positive_negative": [bool(o_pos["ok"]), bool(o_neg["ok"])],
            "inlier_ratio_positive_negative": [float(o_pos["inliers"]) / len(qscan), float(o_neg["inliers"]) / len(qscan)]}
    # all cores: the candidates of a query over threads (the per-candidate work is independent)
    cores = os.cpu_count() or 1
    nthr = min(cores, TOP_K)
    t0 = time.time()
    oracle.reg_many_mt(qscan, [pos] * (nthr - nthr // NEG_EVERY) + [neg] * (nthr // NEG_EVERY), nthr,
                       ref_nn=use_ref, **kw)
    t_mt = time.time() - t0           # nthr candidates in parallel
    per_query_mt = t_knn + t_mt * (TOP_K / nthr)
    cpu_model = ""
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                cpu_model = line.split(":", 1)[1].strip()
                break
    except OSError:
        pass
    return {**extra, "value": 1.0 / per_query, "unit": "queries/s", "cores": 1,
            "kind": "reference" if use_ref else "port",
            "nn_search": ("the reference's vendored nanoflann kd-tree (oracle/_ref), built once per candidate, "
                          "queried every pass -- as PCL's ICP does") if use_ref else "the port's uniform grid",
            "sample": f"1 query: kNN over {n_places}x{DIM} ({t_knn*1e3:.0f} ms) + RANSAC{RANSAC_ITERS}+ICP{ICP_ITERS} "
                      f"registration of 1 positive ({t_pos:.1f} s) and 1 negative ({t_neg:.1f} s) candidate, "
                      f"~123k-pt scans, extrapolated to {TOP_K - n_neg} + {n_neg} candidates",
            "all_cores": {"value": 1.0 / per_query_mt, "threads": nthr,
                          "sample": f"{nthr} candidates registered concurrently ({t_mt:.1f} s), scaled to {TOP_K}"},
            "host_cpus": cores, "cpu_model": cpu_model,
            "compiler_flags": "gcc -O2 -ffp-contract=off (port), g++ -O3 -DNDEBUG -std=c++14 -ffp-contract=off (reference nanoflann)"}


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
                    help="N > 1: the two all-gathers through the C ABI's own RCCL communicator "
                         "(gloc_knn_search_sharded, gloc_comm_all_gather_device) or through torch.distributed")
    ap.add_argument("--same-device", action="store_true",
                    help="rehearsal: all ranks on GPU 0 (use with --backend gloo)")
    ap.add_argument("--no-prefetch", action="store_true", help="prepare each step's queries inline")
    ap.add_argument("--no-negatives", action="store_true", help="every place carries a world-A scan")
    ap.add_argument("--coarse", action="store_true",
                    help="also run the reference's 2-D step: the coarse (x, y, yaw) match of every (query, candidate) "
                         "pair on their BEV grids seeds the 3-D registration (gloc_coarse_*); not part of the metric's "
                         "default configuration")
    ap.add_argument("--scan-store", type=int, default=0,
                    help="distinct resident scans (0 = one per place up to 4541; places beyond alias modulo)")
    ap.add_argument("--no-lone-query", action="store_true", help="skip the one-query-alone launches after the timed region (profiling runs)")
    ap.add_argument("--views-cache", default=None, help="npz cache of the ray-cast base views (profiling runs)")
    ap.add_argument("--nn-src-per-lane", type=int, default=0, help="culled 1-NN tuning (1, 2, 4)")
    ap.add_argument("--nn-job-group", type=int, default=0, help="culled 1-NN tuning: jobs interleaved in the launch order")
    ap.add_argument("--nn-mode", choices=["culled", "exhaustive"], default="culled",
                    help="1-NN search of the registration (identical results)")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch with torch.distributed.run --nproc-per-node N for --gpus N")
    t_setup = time.time()
    pool_a, pool_b, qviews = build_views(args.views_cache)   # forks worker processes: before the GPU is initialised
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
    base_a = [store.add(p) for p in pool_a]
    base_b = [store.add(p) for p in pool_b]
    n_store = min(n_places, args.scan_store or N_PLACES_1GPU)
    neg_on = not args.no_negatives

    def is_negative(g):
        return neg_on and (g % NEG_EVERY == 1)

    place_scan = np.empty(n_store, np.uint32)
    for g in range(n_store):
        base = base_b[(g // NEG_EVERY) % POOL_B] if is_negative(g) else base_a[g % POOL_A]
        place_scan[g] = store.add_variant(base, place_perturbation(g), 0.01, seed=7000 + g)
    live_b, _ = store.bytes()
    mean_pts = float(np.mean([p.shape[0] for p in pool_a]))
    log(f"scan store: {n_store} distinct resident scans (+{POOL_A + POOL_B} base views), "
        f"{live_b / 2**30:.1f} GiB, ~{mean_pts:.0f} pts each; negatives: places g % {NEG_EVERY} == 1"
        if neg_on else f"scan store: {n_store} distinct resident scans, {live_b / 2**30:.1f} GiB, no negatives")

    # ---- the query stream: distinct host-side scans + descriptors -------------------------------
    n_stream = n_steps * per_step                       # distinct queries of one repetition
    total = (n_steps + n_warm) * per_step
    # query j is taken next to place g_j whose pool view has a query view beside it
    rng_rows = (np.arange(total, dtype=np.int64) * 977 + 211) % max(n_store - POOL_A, 1)
    q_view = np.arange(total) % QUERY_VIEWS
    q_place = rng_rows - (rng_rows % POOL_A) + (3 * q_view + 1)            # g with g % POOL_A == 3 v + 1
    q_place = np.clip(q_place, 0, n_store - 1)
    q_desc_host = torch.from_numpy(synth.queries_near(DB_SEED, q_place, DIM)).pin_memory()
    # the queries' scans: made on the device from the query views, read back into pinned host memory --
    # from then on they exist only on the host, like scans arriving from a sensor
    qbase = [store.add(v) for v in qviews]
    q_scan_host = []
    for j in range(total):
        key = synth.rng_key(DB_SEED ^ 0x9E77, np.uint64(j))
        u = synth.rng_uniform(key, np.arange(4, dtype=np.uint64)).astype(np.float64) * 2 - 1
        sid = store.add_variant(qbase[int(q_view[j])], synth.se3(1.0 * u[0], (0.2 * u[1], 0.2 * u[2], 0.02 * u[3])),
                                0.01, seed=880000 + j)
        h = torch.from_numpy(store.download(sid)).pin_memory()
        store.release(sid)
        q_scan_host.append(h)
    for sid in qbase:
        store.release(sid)
    store_scans_resident = len(store)

    reg = capi.Registrar(device=local_rank, store=store)
    reg.set_option(capi.REG_OPT_PROFILE, 1)
    reg.set_option(capi.REG_OPT_NN_MODE, capi.REG_NN_CULLED if args.nn_mode == "culled" else capi.REG_NN_EXHAUSTIVE)
    if args.nn_src_per_lane:
        reg.set_option(capi.REG_OPT_NN_SRC_PER_LANE, args.nn_src_per_lane)
    if args.nn_job_group:
        reg.set_option(capi.REG_OPT_NN_JOB_GROUP, args.nn_job_group)
    params = capi.default_reg_params(ransac_iters=RANSAC_ITERS, icp_iters=ICP_ITERS,
                                     min_inlier_ratio=MIN_INLIER_RATIO, max_rmse=MAX_RMSE)

    capi_knn, collectives = None, "none"
    if world > 1 and args.collectives == "capi" and args.backend == "nccl" and not args.same_device:
        try:
            comm = capi.Comm(local_rank, rank, world, sharded.torch_exchange(dev))
            capi_knn = sharded.CapiShardedKnn(index, comm)
            probe = torch.full((1, 1, sharded.RESULT_COLS), float(rank), dtype=torch.float32, device=dev)
            got = capi_knn.all_gather_tables(probe)     # self-test: every rank's row must arrive in rank order
            torch.cuda.synchronize()
            if not (got[:, 0, 0].cpu() == torch.arange(world, dtype=torch.float32)).all():
                raise RuntimeError("all-gather self-test returned the wrong rows")
            collectives = "capi (gloc_knn_search_sharded + gloc_comm_all_gather_device, RCCL)"
        except Exception as e:   # loud, and recorded in the JSON line: never a silent change of path
            capi_knn = None
            print(f"[bench] rank {rank}: C-ABI RCCL communicator unavailable ({e}); using torch.distributed", file=sys.stderr, flush=True)
    if world > 1:
        ok_all = torch.tensor([1 if capi_knn is not None else 0], device=comm_dev or dev)
        dist.all_reduce(ok_all, op=dist.ReduceOp.MIN)     # all ranks take the same path
        if int(ok_all.item()) == 0:
            capi_knn = None
    if capi_knn is not None:
        knn = capi_knn
    else:
        knn = sharded.ShardedKnn(rank, world, sharded.hip_local_search(index), sharded.hip_merge(local_rank),
                                 comm_device=comm_dev)
        if world > 1:
            collectives = f"torch.distributed ({args.backend})"

    # optional: the coarse 2-D match (one grid per place, made on the device from the resident scans)
    cm, place_grid, cm_lock, cur_qgrids = None, None, threading.Lock(), {}
    if args.coarse:
        cm = capi.CoarseMatcher(local_rank)
        place_grid = np.array([cm.add_store_scan(store, int(sid)) for sid in place_scan], np.uint32)
        log(f"coarse grids: {n_store} places")

    def coarse_init(q_ids, places):
        """init_T [B, n, 4, 4] from the 2-D match of every (query, candidate) pair; identity where it fails."""
        p = np.asarray(places, np.int64)
        Bq, n = p.shape
        qg = np.repeat(np.array([cur_qgrids[int(q)] for q in q_ids], np.uint32), n)
        dg = place_grid[np.clip(p, 0, None).reshape(-1) % n_store]
        with cm_lock:
            xy_yaw, _, ok2 = cm.match_pairs(qg, dg)
        T = np.tile(np.eye(4, dtype=np.float32), (Bq * n, 1, 1))
        c, s_ = np.cos(xy_yaw[:, 2]), np.sin(xy_yaw[:, 2])
        use = ok2 & (p.reshape(-1) >= 0)
        T[use, 0, 0], T[use, 0, 1], T[use, 1, 0], T[use, 1, 1] = c[use], -s_[use], s_[use], c[use]
        T[use, 0, 3], T[use, 1, 3] = xy_yaw[use, 0], xy_yaw[use, 1]
        return T.reshape(Bq, n, 4, 4)

    def scans_of(places):
        """global place ids [.., n] (-1 = none) -> resident scan ids (every rank holds all scans)."""
        p = np.asarray(places, np.int64)
        out = place_scan[np.clip(p, 0, None) % n_store].astype(np.uint32)
        out[p < 0] = capi.NO_SCAN
        return out

    fs_state = {"on": False, "jobs": 0, "queries": 0}

    def register_multi(q_ids, places):
        init = coarse_init(q_ids, places) if cm is not None else None
        if fs_state["on"]:
            # the reference's loop as written: stop at the first success (gloc_reg_first_success_multi)
            f = reg.first_success_multi(q_ids, scans_of(places), params=params, init_T=init)
            shape = np.asarray(places).shape
            out = np.zeros(shape + (sharded.RESULT_COLS,), np.float32)
            for k_, rk in enumerate(f["rank"]):
                if rk >= 0:
                    out[k_, rk, :16] = f["T"][k_].reshape(16)
                    out[k_, rk, 16], out[k_, rk, 17], out[k_, rk, 18] = f["rmse"][k_], f["inliers"][k_], 1.0
            fs_state["jobs"] += f["jobs_run"]
            fs_state["queries"] += len(q_ids)
            return out
        r = reg.batch_multi(q_ids, scans_of(places), params=params, init_T=init)
        return sharded.pack_results(r, np.asarray(places).shape)

    def local_register(q_id, local_rows, ranks):   # latency mode: this rank's share of one query's candidates
        g = np.asarray(local_rows, np.int64) * world + rank
        r = reg.batch_ids(q_id, scans_of(g), params=params, stream_ids=ranks)
        return sharded.pack_results(r, (len(g),))

    sreg = sharded.ShardedRegistrar(rank, world, local_register, comm_device=comm_dev)
    qreg = sharded.QueryParallelRegistrar(rank, world, None, comm_device=comm_dev)
    stage = {"prep_wait": 0.0, "h2d_index": 0.0, "knn": 0.0, "register": 0.0}
    work_pairs = []

    # ---- one step -----------------------------------------------------------------------------
    def my_slice(i):
        q0 = i * per_step
        return (q0 + rank * B, q0 + rank * B + B) if args.mode == "throughput" else (q0, q0 + 1)

    def prepare(i):
        """Query preparation of step i: this rank's fresh query scans go H2D and are indexed
        (gloc_scan_store_add, on the store's stream); the step's descriptors go H2D."""
        t0 = time.time()
        a, b = my_slice(i)
        ids = [store.add(q_scan_host[j].numpy()) for j in range(a, b)]
        if cm is not None:
            with cm_lock:
                for sid in ids:
                    cur_qgrids[sid] = cm.add_store_scan(store, sid)
        q0 = i * per_step
        qd = q_desc_host[q0:q0 + per_step].to(dev, non_blocking=True)
        return ids, qd, time.time() - t0

    class Prefetcher:
        def __init__(self):
            self.slot, self.th = None, None

        def start(self, i):
            def run():
                torch.cuda.set_device(local_rank)
                self.slot = prepare(i)
            self.th = threading.Thread(target=run)
            self.th.start()

        def take(self):
            self.th.join()
            s, self.slot, self.th = self.slot, None, None
            return s

    pre = Prefetcher()

    def step(i, last, record):
        t0 = time.time()
        if args.no_prefetch:
            ids, qd, t_prep = prepare(i)
        else:
            ids, qd, t_prep = pre.take()
            if not last:
                pre.start(i + 1)          # the next step's uploads + indexing overlap this step's registration
        t1 = time.time()
        torch.cuda.current_stream().synchronize()   # the descriptors' H2D
        idx, d2 = knn.search(qd, TOP_K)
        cand = idx.cpu().numpy()                          # [per_step, 20] global place ids, retrieval order
        t2 = time.time()
        a, b = my_slice(i)
        if args.mode == "throughput":
            tables = qreg.register_many(ids, cand, dev, register_multi, capi_knn=capi_knn)   # [world*B, 20, 19]
            sel = [sharded.ShardedRegistrar.select_first_ok(t) for t in tables]
        else:
            table = sreg.register(ids[0], cand[0], dev)
            sel = [sreg.select_first_ok(table)]
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
            mine = cand[rank * B:(rank + 1) * B] if args.mode == "throughput" else cand[:1]
            for k in range(mine.shape[0]):
                nq = q_scan_host[a + k].shape[0] if args.mode == "throughput" else q_scan_host[a].shape[0]
                work_pairs.append((nq, int(np.count_nonzero(mine[k] >= 0))))
        return cand, sel

    def fence():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    log(f"setup {time.time() - t_setup:.1f} s; warmup {n_warm}, timing {n_reps} x {n_steps} steps of {per_step} queries")
    if n_warm and not args.no_prefetch:
        pre.start(n_steps)
    for i in range(n_steps, n_steps + n_warm):     # warm-up queries: the tail of the stream
        cand, sel = step(i, i == n_steps + n_warm - 1, False)
        assert (cand[:, 0] == q_place[i * per_step:(i + 1) * per_step]).all(), "retrieval sanity: top-1 != query place"
    rep_s, sels = [], []
    for rep in range(n_reps):
        fence()
        if rep == n_reps - 1:
            reg.profile_reset()
        t0 = time.time()
        if not args.no_prefetch:
            pre.start(0)            # step 0's preparation has no earlier step to hide behind: it is paid in full
        rsel = []
        for i in range(n_steps):
            cand, sel = step(i, i == n_steps - 1, rep == n_reps - 1)
            rsel.extend(sel)
        fence()
        elapsed = time.time() - t0
        if world > 1:
            t = torch.tensor([elapsed], dtype=torch.float64, device=comm_dev or dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            elapsed = float(t.item())
        rep_s.append(elapsed)
        sels = rsel
    elapsed = float(np.median(rep_s))

    # ---- roofline of the dominant kernel (K4 point-NN), from the HIP events of the last repetition --
    nn_ms, nn_launches = reg.profile("nn")
    stage_ms = {n: reg.profile(n)[0] for n in ("nn", "ransac_score", "ransac_hyp", "accum", "solve")}
    passes = 1 + ICP_ITERS
    pairs_eval, _ = reg.nn_stats()
    avg_launch_s = (nn_ms / max(nn_launches, 1)) * 1e-3
    jobs_per_launch = float(np.mean([c for _, c in work_pairs])) * (B if args.mode == "throughput" else 1) \
        if work_pairs else 0.0
    # algorithmic bytes of one launch: every job reads its query scan and its candidate scan once
    # (sorted float4 = 16 B/point) and writes corr + d2 (8 B/source point)
    pts_q = float(np.mean([n for n, _ in work_pairs])) if work_pairs else 0.0
    alg_bytes = jobs_per_launch * (BYTES_PER_POINT_INDEXED * (pts_q + mean_pts) + 8 * pts_q)
    all_pairs = jobs_per_launch * pts_q * mean_pts
    eval_pairs = all_pairs if args.nn_mode == "exhaustive" else float(pairs_eval) / max(nn_launches, 1)
    kname = "gloc::reg::nn_kernel" if args.nn_mode == "exhaustive" else "gloc::reg::nn_compact_kernel"
    traffic = None  # HBM bytes per launch from the committed PMC passes (profiles/), same workload
    pmc = os.path.join(ROOT, "profiles", "r02_pmc_traffic_nn_compact.json")
    issue_model = None
    if args.nn_mode == "culled" and world == 1 and os.path.exists(pmc):
        try:
            pj = json.load(open(pmc))
        except ValueError:      # an empty or damaged file: the line is still printed, without the PMC-derived fields
            pj = {}
        traffic = pj.get("hbm_bytes_per_launch")
        if pj.get("jobs_per_launch") and abs(pj["jobs_per_launch"] - jobs_per_launch) > 1e-6:
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
                           "source": "profiles/r02_pmc_traffic_nn_compact.json (SQ_INSTS_* / SQ_WAVES; busy = SQ_ACTIVE_INST_VALU "
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
                    "pairs_evaluated_per_launch": eval_pairs, "pairs_exhaustive_per_launch": all_pairs,
                    "issue_model": issue_model,
                    "note": "the culled search evaluates ~1e-3 of the pairs SURVEY 8d's flop count assumes, so its "
                            "floor is reading each scan once: (16 B x (query + candidate points) + 8 B x query "
                            "points) x jobs per launch over the HIP-event duration of the launch (one stream, "
                            "launches do not overlap); the kernel itself is vector-issue bound, see DESIGN.md; the "
                            "brute-force nn_kernel north_star names runs at 35.7 % of the fp32 peak "
                            "(--nn-mode exhaustive)"}

    # one query alone (20 jobs per launch): the latency-bound end of the same kernel
    if world == 1 and args.mode == "throughput" and not args.no_lone_query:
        reg.profile_reset()
        for j in range(3):
            sid = store.add(q_scan_host[j].numpy())
            ci, _ = knn.search(q_desc_host[j:j + 1].to(dev), TOP_K)
            reg.batch_multi([sid], scans_of(ci.cpu().numpy()), params=params)
            reg.scan_release(sid)
        ms1, n1 = reg.profile("nn")
        roofline["launch_ms_one_query_20_jobs"] = ms1 / max(n1, 1)

    # the same stream with the reference's early exit (registration stops at a query's first successful candidate)
    first_success = None
    if world == 1 and args.mode == "throughput" and not args.no_lone_query:
        fs_state["on"] = True
        fence()
        t0 = time.time()
        if not args.no_prefetch:
            pre.start(0)
        fsel = []
        for i in range(n_steps):
            cand, sel = step(i, i == n_steps - 1, False)
            fsel.extend(sel)
        fence()
        t_fs = time.time() - t0
        fs_state["on"] = False
        first_success = {"value": n_steps * per_step / t_fs, "unit": "queries/s",
                         "registrations_per_query": fs_state["jobs"] / max(fs_state["queries"], 1),
                         "same_selection_as_full_batch": bool(fsel == sels),
                         "note": "gloc_reg_first_success_multi: rank by rank, only queries still without a success go on "
                                 "(registration/global_localization.cpp:519-572 stops at the first match()==true); not the "
                                 "metric's configuration, which registers all 20 candidates"}

    q_per_rep = n_steps * per_step
    out = {
        "metric": "localization queries/sec (kNN+top-20 reg), KITTI-00-sized DB",
        "value": q_per_rep / elapsed, "unit": "queries/s", "n_gpus": world, "steps": n_steps,
        "warmup": n_warm, "ms_per_step": elapsed / n_steps * 1e3, "higher_is_better": True,
        "scaling": "weak" if args.mode == "throughput" else "strong", "vs_baseline": None, "dtype": "f32",
        "data": "synthetic",
        "config": {"workload": f"cfgD: KITTI-00-sized DB {n_places}x{DIM} fp32, stream of {q_per_rep} queries "
                               f"({n_steps} steps x {per_step}), each: fresh scan H2D+index, descriptor H2D -> top-{TOP_K} "
                               f"-> {TOP_K} candidate scans (~{mean_pts:.0f} pts) x (RANSAC {RANSAC_ITERS} adaptive + ICP {ICP_ITERS})",
                   "queries_per_step": per_step, "queries_per_batch_per_gpu": B, "queries_per_repetition": q_per_rep,
                   "repetitions": n_reps, "repetition_seconds": rep_s, "value_is": "median repetition",
                   "places": n_places, "dim": DIM, "top_k": TOP_K, "points_per_scan": int(mean_pts),
                   "scan_store_scans": int(store_scans_resident), "scan_store_distinct_places": int(n_store),
                   "scan_store_gib": live_b / 2**30,
                   "negatives_per_query": (TOP_K // NEG_EVERY) if neg_on else 0,
                   "query_prep_in_timed_region": True,
                   "query_prep": "per query: scan H2D from pinned host memory + device indexing + descriptor H2D, "
                                 + ("inline" if args.no_prefetch else "prefetched one step ahead on a second host thread + stream"),
                   "ransac_iters_cap": RANSAC_ITERS, "ransac_confidence": float(params.ransac_confidence),
                   "min_inlier_ratio": MIN_INLIER_RATIO, "max_rmse": MAX_RMSE, "icp_iters": ICP_ITERS, "nn_passes_per_query": passes,
                   "nn_mode": args.nn_mode, "collectives": collectives, "coarse_2d_match": bool(args.coarse),
                   "parallelism": (f"1 gpu, {B} queries registered per batch on one stream") if world == 1 else (
                       f"{B} queries per gpu per step; "
                       f"db rows interleave-sharded over {world} ranks (all-gather of per-shard top-k, merge); "
                       + ("each rank registers its queries locally, result tables all-gathered"
                          if args.mode == "throughput" else
                          "one query, candidates sharded over the ranks, all-reduce of poses"))},
        "roofline": roofline,
        "stage_ms_per_step_rank0": {**{k_: v / n_steps for k_, v in stage_ms.items()},
                                    **{"host_" + k_: v / n_steps * 1e3 for k_, v in stage.items()}},
        "selected_candidate_rank_histogram": {str(k_): int(v) for k_, v in
                                              zip(*np.unique(np.asarray(sels), return_counts=True))},
        "first_success_mode": first_success,
    }
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        log("timing the CPU checker (bounded sample) ...")
        g_pos = int(q_place[0]) + (1 if is_negative(int(q_place[0])) else 0)     # a same-world neighbour of query 0
        g_neg = int(q_place[0]) - (int(q_place[0]) % NEG_EVERY) + 1               # the different-world place next to it
        sample = {"query": q_scan_host[0].numpy(), "positive": store.download(int(place_scan[g_pos])),
                  "negative": store.download(int(place_scan[g_neg]))}
        out["cpu_baseline"] = cpu_baseline(sample, n_places, MIN_INLIER_RATIO,
                                           gpu_check=lambda q_, c_: reg.batch(q_, c_, params=params))
    elif rank == 0:
        out["cpu_baseline"] = None
    if rank == 0:
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
