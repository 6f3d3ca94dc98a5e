/*
 * gloc3d.h -- C ABI of the MI355X-native place-retrieval + global-registration hot path.
 *
 * This is the drop-in boundary: plain C, opaque handles, `int` status codes, no exceptions and no
 * torch/HIP types in any signature (streams travel as `void*`).  One handle = one device + one HIP
 * stream + one caller thread (thread-compatible, not thread-safe -- the same contract as the
 * reference's RpyPCLoopDetector, registration/loop_detector.h:41-119).  The caller owns every host
 * buffer; handles own their device memory.  There is NO CPU fallback: every entry point fails with
 * GLOC_ERR_NODEVICE / GLOC_ERR_HIP when no gfx950 device is usable.
 *
 * Each entry point cites the reference interface it replaces (paths relative to the reference
 * repository root).  INTEGRATION.md shows the reference-side bindings.
 */
#ifndef GLOC3D_H
#define GLOC3D_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define GLOC3D_ABI_VERSION 6

enum {
  GLOC_OK = 0,
  GLOC_ERR_INVALID = 1,  /* bad argument (null pointer, k = 0, k too large, dim mismatch ...) */
  GLOC_ERR_HIP = 2,      /* a HIP runtime call failed; see gloc_last_error() */
  GLOC_ERR_NOMEM = 3,    /* device or host allocation failed */
  GLOC_ERR_NODEVICE = 4, /* no usable gfx950 device */
  GLOC_ERR_STATE = 5     /* call not valid in the handle's current state */
};

/* Thread-local description of the last failure on the calling thread ("" if none). */
const char* gloc_last_error(void);
int gloc_abi_version(void);
/* Number of visible HIP devices (0 if none; never fails). */
int gloc_device_count(void);

/* ============================ descriptor kNN ============================================= *
 * Replaces the KD-tree the reference builds over its descriptor database,
 *   InvKeyTree(k_dim_, db_features_, 10)            registration/loop_detector.cpp:36,70
 *   kdtree_->query(&feat[0], top_k_, idx, d2)        registration/loop_detector.cpp:45,79
 *   (KDTreeVectorOfVectorsAdaptor<KeyMat,float>::query,
 *    registration/KDTreeVectorOfVectorsAdaptor.h:95-102)
 * and its Python twin faiss.IndexFlatL2(pool).add / .search(qFeat, 20), main.py:317-324.
 * Results are the exact squared-L2 top-k, ascending, with the d2 bit patterns of
 * nanoflann's L2_Adaptor::evalMetric (registration/nanoflann.hpp:453-487); rows at exactly equal
 * distance come out in ascending row index.
 */
typedef struct gloc_knn gloc_knn;

/* Algorithm selection for gloc_knn_set_option(GLOC_KNN_OPT_ALGO, ...). */
enum {
  GLOC_KNN_ALGO_AUTO = 0,  /* exact streaming kernel for few queries, MFMA path otherwise */
  GLOC_KNN_ALGO_EXACT = 1, /* reference-order fp32 differences on the vector ALUs, one pass */
  GLOC_KNN_ALGO_MFMA = 2,  /* matrix-core coarse pass -2Q.D^T + norms (operands split in two bf16 values, three bf16
                              MFMAs per product -- a proven bound on what that drops; fp32 MFMA when dim % 8 != 0),
                              top-k' selection, exact re-rank in the reference's order, completeness proven per query
                              (an unproven query is redone on the exact path): the same bits as GLOC_KNN_ALGO_EXACT.  A handle
                              whose searches keep failing that proof (rows clustered tightly relative to their norms) runs
                              its next searches with the fp32 coarse pass, whose bound is eight times tighter */
  GLOC_KNN_ALGO_MFMA_FP32 = 3 /* the same with the coarse pass on the fp32 MFMA (the rounds 1 - 3 form) */
};
enum {
  GLOC_KNN_OPT_ALGO = 1,
  GLOC_KNN_OPT_CANDIDATES = 2, /* k' kept by the MFMA path before the exact re-rank (<= 64) */
  GLOC_KNN_OPT_PROFILE = 3     /* 1: bracket every kernel with HIP events (gloc_knn_profile) */
};

/* device: HIP ordinal.  dim: descriptor length (reference: k_dim_ = 512, loop_detector.h:97). */
int gloc_knn_create(int device, size_t dim, gloc_knn** out);
/* A second SEARCH handle over the same resident database (round 6): its own HIP stream and workspace, the parent's rows
 * (no copy; rows the parent has added -- and finished adding: synchronize it -- are seen by the view's next search).
 * Searches on the parent and on a view run side by side on the device: one's selection + re-rank (a work-group per query:
 * 64 of 256 CUs at 64 queries) under the other's distance kernel -- back-to-back searches of DIFFERENT query batches then
 * cost max(stage) instead of the sum (bench.py: knn_cfgB_pipelined_us).  nanoflann's query is const for the same reason
 * (KDTreeVectorOfVectorsAdaptor.h:95-102).  A view cannot add / reserve / clear / load (GLOC_ERR_STATE); the parent cannot be
 * destroyed while views live. */
int gloc_knn_create_view(gloc_knn* parent, gloc_knn** out);
int gloc_knn_destroy(gloc_knn* h);

/* Use `hip_stream` (a hipStream_t created by the caller on the same device) for all work of this
 * handle instead of the handle's own stream.  NULL restores the own stream. */
int gloc_knn_set_stream(gloc_knn* h, void* hip_stream);
int gloc_knn_synchronize(gloc_knn* h);
int gloc_knn_set_option(gloc_knn* h, int option, int64_t value);

/* Append n rows (row-major n x dim, host memory).  Replaces db_features_.push_back(feat)
 * (registration/loop_detector.cpp:14) and faiss_index.add(dbFeat) (main.py:320).  Unlike the
 * reference's tree, which silently goes stale after the first detect (loop_detector.cpp:34-37
 * builds once over a const-ref), rows are searchable immediately. */
int gloc_knn_add(gloc_knn* h, const float* rows, size_t n);
/* Same with rows already in device memory on the handle's device. */
int gloc_knn_add_device(gloc_knn* h, const float* d_rows, size_t n);
int gloc_knn_reserve(gloc_knn* h, size_t n_rows);
int gloc_knn_clear(gloc_knn* h);
int gloc_knn_size(const gloc_knn* h, size_t* n_rows);
int gloc_knn_dim(const gloc_knn* h, size_t* dim);
/* Device pointer of the resident row-major database (valid until the next add/reserve/clear). */
int gloc_knn_device_rows(const gloc_knn* h, const float** d_rows);

/* Persist / restore the database ("next" row N4; the reference keeps its descriptor database in
 * memory only and rebuilds it from the scans on every start, global_localization.cpp:419-449).
 * File: "GLOCDESC", u32 rows, u32 dim, then rows x dim little-endian fp32 -- the same format the
 * drop-in global_localization command line reads in place of the TorchScript model.
 * load APPENDS the file's rows (dim must match). */
int gloc_knn_save(gloc_knn* h, const char* path);
int gloc_knn_load(gloc_knn* h, const char* path);

/* Top-k over rows [first_row, last_row) for nq queries (host buffers).  last_row is clamped to the
 * database size; pass SIZE_MAX for "all".  The row window expresses the SLAM-mode exclusion of the
 * newest frames (db_features_.begin() .. end()-num_exclude_recent_, loop_detector.cpp:66-72).
 * out_idx: nq x k row indices, out_d2: nq x k squared distances, ascending.  If the window holds
 * fewer than k rows the tail is idx = UINT64_MAX, d2 = FLT_MAX. */
int gloc_knn_search(gloc_knn* h, const float* queries, size_t nq, size_t k, size_t first_row,
                    size_t last_row, uint64_t* out_idx, float* out_d2);

/* Same with device buffers, enqueued on the handle's stream; `index_offset` is added to every
 * returned index (a shard's first global row, for row-sharded databases).  Returns after the work
 * is enqueued: for windows of up to 16384 rows there is no host synchronisation at all (queries whose
 * candidate set the MFMA path cannot prove complete are redone on the exact path by kernels that are
 * always enqueued and leave at once otherwise); larger windows read the completeness flags back. */
int gloc_knn_search_device(gloc_knn* h, const float* d_queries, size_t nq, size_t k,
                           size_t first_row, size_t last_row, uint64_t index_offset,
                           uint64_t* d_out_idx, float* d_out_d2);

/* Merge `n_lists` sorted top-k lists per query (gathered from the shards of a row-sharded
 * database: layout [n_lists][nq][k]) into one top-k, ordered by (d2, idx).  Device buffers;
 * enqueued on `hip_stream` (may be NULL = default stream). */
int gloc_topk_merge_device(int device, void* hip_stream, const uint64_t* d_idx, const float* d_d2,
                           size_t n_lists, size_t nq, size_t k, uint64_t* d_out_idx,
                           float* d_out_d2);

/* ---- multi-GPU through the C ABI (one process per GPU, RCCL over xGMI) ------------ *
 * The descriptor database is row-sharded over the ranks of a communicator; queries are replicated.
 * librccl is bound at run time (the copy already in the process -- PyTorch's -- else GLOC3D_RCCL, else
 * /opt/rocm/lib).  Rank 0 makes the 128-byte id and hands it to the others by any channel (a file,
 * MPI, torch.distributed); every rank then calls gloc_comm_create (ncclCommInitRank: collective). */
typedef struct gloc_comm gloc_comm;
int gloc_comm_unique_id(uint8_t* id128);
int gloc_comm_create(int device, int rank, int world, const uint8_t* id128, gloc_comm** out);
int gloc_comm_destroy(gloc_comm* c);
int gloc_comm_rank(const gloc_comm* c, int* rank, int* world);
/* d_recv[r * bytes_per_rank ..] = rank r's d_send; enqueued on hip_stream (result tables of a step). */
int gloc_comm_all_gather_device(gloc_comm* c, const void* d_send, void* d_recv, size_t bytes_per_rank,
                                void* hip_stream);
/* Top-k over the WHOLE row-sharded database, replicated on every rank and bit-equal to the one-GPU
 * search: this rank's shard is searched (local row l is global row l * index_stride + index_offset:
 * stride = world, offset = rank for the interleaved layout; stride 1, offset = first row for contiguous
 * shards), the per-shard (d2, idx) lists are all-gathered in ONE fused RCCL launch on the handle's
 * stream, and merged by (d2, idx) on the device (K3).  No host hop, no host synchronisation beyond the
 * search's own.  Collective: every rank calls it with the same nq and k. */
int gloc_knn_search_sharded(gloc_knn* h, gloc_comm* comm, const float* d_queries, size_t nq, size_t k,
                            uint64_t index_stride, uint64_t index_offset, uint64_t* d_out_idx,
                            float* d_out_d2);

/* Host-buffer forms (the C++ command line's sharded mode): staged through the device, synchronous. */
int gloc_knn_search_sharded_host(gloc_knn* h, gloc_comm* comm, const float* queries, size_t nq, size_t k,
                                 uint64_t index_stride, uint64_t index_offset, uint64_t* out_idx,
                                 float* out_d2);
int gloc_comm_all_gather_host(gloc_comm* c, const void* send, void* recv, size_t bytes_per_rank);

/* Counters since creation: searches on each path, queries that needed the exact fallback. */
typedef struct gloc_knn_stats {
  uint64_t searches_exact, searches_mfma, queries_total, queries_fallback;
  uint32_t last_n_tile, last_k_split, last_candidates;
} gloc_knn_stats;
int gloc_knn_get_stats(const gloc_knn* h, gloc_knn_stats* out);

/* With GLOC_KNN_OPT_PROFILE = 1: accumulated HIP-event time of one kernel family since the last
 * gloc_knn_profile_reset.  Names: "dist_exact", "dist_mfma", "select", "rerank", "norms",
 * "finalize".  Synchronizes the stream. */
int gloc_knn_profile(gloc_knn* h, const char* kernel, double* total_ms, uint64_t* launches);
int gloc_knn_profile_reset(gloc_knn* h);

/* ============================ 3-D registration =========================================== *
 * Replaces the reference's per-candidate registration seam,
 *   icp_match_3d(src, tgt, guess, pose)   registration/global_registration.cpp:237-248
 *   (pcl::IterativeClosestPoint, 30 iterations), and the RANSAC transform estimate the reference
 *   runs per candidate (cv::estimateAffinePartial2D(..., RANSAC, 3*res, 3000),
 *   registration/loop_detector.cpp:256-257), as the batched loop over the top-20 candidates of
 *   GlocEvaluator::global_registraion (registration/global_localization.cpp:511-574).
 * Semantics: SURVEY.md Appendix B (S1 exact 1-NN, S2 RANSAC 3-point Kabsch/SVD + inlier count +
 * refit, S3 point-to-point ICP).
 */
typedef struct gloc_reg gloc_reg;

typedef struct gloc_reg_params {
  uint32_t ransac_iters;  /* 3000 (loop_detector.cpp:257); 0 disables RANSAC */
  float inlier_thresh;    /* 0.6 m = 3 x 0.2 m (loop_detector.cpp:257, loop_detector.h:116) */
  float min_inlier_ratio; /* ok iff best inliers >= ratio x n_src (and >= 3) */
  uint32_t icp_iters;     /* 30 (global_registration.cpp:242) */
  float max_corr_dist;    /* <= 0: no correspondence rejection (PCL default) */
  uint64_t seed;          /* RANSAC sampling seed */
  float ransac_confidence; /* 0.99: adaptive stop as in OpenCV's RANSAC, which the reference calls with
                             its default confidence (loop_detector.cpp:256-257): ransac_iters is the
                             cap, hypotheses beyond the iteration count that reaches this confidence
                             for the best inlier ratio so far are not considered.  <= 0 or >= 1: off */
  float max_rmse;          /* > 0: a candidate is ok only if, in addition, the RMS nearest-neighbour distance of
                             its final pose is <= this (metres).  Plausibility check on the estimated
                             transform, the analogue of the reference's |1 - scale| < 0.1
                             (loop_detector.cpp:268-272): the RANSAC inlier ratio at 0.6 m alone cannot
                             tell two scenes apart that share a ground plane.  <= 0: off (default) */
  float max_final_step;    /* > 0 (needs icp_iters > 0): a candidate is ok only if, in addition, the ICP has CONVERGED:
                             the RMS displacement its last update gives the matched points --
                             sqrt(|R c + t - c|^2 + |R - I|_F^2 / 2 * tr cov), c and cov the centroid and covariance
                             of those points -- is <= this; an ICP that stopped for want of correspondences
                             (max_corr_dist) counts as not converged.  A plausibility check of the 3-D stage in the
                             role the reference gives |1 - scale| < 0.1 on its 2-D fit (loop_detector.cpp:268-272);
                             pcl::IterativeClosestPoint::hasConverged() is the nearest thing upstream, and the
                             reference does not consult it.  <= 0: OFF -- THE DEFAULT (round 5; 0.03 in round 4): the
                             reference's 3-D stage accepts whatever its ICP returns, and so does this library unless
                             the caller asks (the loop_detector mirrors and the two command lines do not: their
                             plausibility check is the reference's, on the 2-D match).  bench.py passes
                             GLOC_REG_FINAL_STEP_SUGGESTED below and says so in its line. */
} gloc_reg_params;

/* A value for gloc_reg_params.max_final_step.  NOT reference-derived (the reference has no such quantity): it was
 * chosen by sweeping 0.025 / 0.03 / 0.04 / 0.05 m over bench.py's own synthetic legs (LAB_NOTES.md, round 4) and
 * checked in round 5 on worlds, views and poses that sweep never saw (bench.py legs.gate_holdout,
 * tests/test_gate_holdout_gpu.py; DESIGN.md section 4 has the numbers, including the right poses it rejects). */
#define GLOC_REG_FINAL_STEP_SUGGESTED 0.03f

/* Fills the defaults above: the reference's constants where it has them (3000 hypotheses, 0.6 m, 30 ICP passes,
 * confidence 0.99), this library's own otherwise (min_inlier_ratio 0.3, seed 1234); both plausibility checks off. */
void gloc_reg_default_params(gloc_reg_params* p);

int gloc_reg_create(int device, gloc_reg** out);
int gloc_reg_destroy(gloc_reg* h);
/* Use `hip_stream` for all work of this handle (NULL: back to its own).  Handles may SHARE a stream: a batch call
 * returns when its own results have arrived (an event behind its last copy), not when the stream is idle, so two
 * handles driven by two host threads queue their batches back to back on one stream -- the device never waits for the
 * host between batches (bench.py's registration pipeline). */
int gloc_reg_set_stream(gloc_reg* h, void* hip_stream);
int gloc_reg_synchronize(gloc_reg* h);
int gloc_reg_set_option(gloc_reg* h, int option, int64_t value);
enum {
  GLOC_REG_OPT_PROFILE = 1, /* 1: bracket every kernel with HIP events (gloc_reg_profile) */
  GLOC_REG_OPT_NN_MODE = 2, /* how S1 (exact 1-NN) is searched; the result is identical */
  GLOC_REG_OPT_NN_SRC_PER_LANE = 3, /* culled search tuning: source points per lane (1, 2 or 4) */
  GLOC_REG_OPT_NN_JOB_GROUP = 4,    /* culled search tuning: jobs whose work-groups are interleaved in the
                                       launch order (their scans share the caches); default 24 -- 8 in a batch under
                                       48 jobs, with 2 / 4 / 8 shares of a job per slot (NN_SUB_JOBS below), unless either
                                       option is set.  A multiple of 8 keeps each slot's work-groups on one XCD, i.e. its
                                       scans in one L2 */
  GLOC_REG_OPT_TEMP_TARGET_INDEX = 5, /* 1: the host-buffer calls (gloc_reg_batch, gloc_reg_nn) build the kd-ordered
                                       target index for their temporary candidate scans too (default 0: a
                                       millisecond per candidate is more than one registration saves) */
  GLOC_REG_OPT_NN_SPLIT_HELPERS = 6, /* culled search: wave slots per job at the head of the launch order for the
                                       heaviest source groups (a group whose work estimate of the previous pass
                                       exceeds the threshold is searched by 2, 4 or 8 waves that start first;
                                       identical results).  -1 (default): by batch size (256 up to
                                       64 jobs, 64 up to 256, else off); 0: off */
  GLOC_REG_OPT_NN_SPLIT_THRESH = 7,  /* the estimate (cycles of one wave) above which a group is split; default
                                       60000 (85000 for the batches whose passes are chained: GLOC_REG_OPT_NN_CHAIN); 0: off */
  GLOC_REG_OPT_NN_SUB_JOBS = 8,      /* culled search tuning: interleaved shares of a job's work-groups that take a
                                       slot of the launch order each (a slot stays on one XCD); 0 (default): in a
                                       batch under 48 jobs, which 8 XCDs cannot balance job by job, the fewest of 2 / 4 /
                                       8 that make the slots a multiple of 8 (20 jobs: 2), else 1 */
  GLOC_REG_OPT_NN_HEAVY_THRESH = 9,  /* culled search, first (cold) pass of a batch: a wave that has processed this many
                                       target chunks hands its source group to a second launch, which searches it with
                                       8 waves (identical results); default 32; 0: off */
  GLOC_REG_OPT_SUB_BATCHES = 10,     /* a small batch is cut into this many runs of jobs, each enqueued on its own internal
                                       stream (forked from and joined back into the handle's stream), so that one
                                       run's solve and ramp-up run under the others' searches: 2 .. 8; -1 (default), 0, 1: off --
                                       measured SLOWER for one query alone (20 jobs: 3.2 ms on one stream, 3.9 with 4): a
                                       search launch of 5 jobs already has more waves than the chip has slots.  Identical
                                       results */
  GLOC_REG_OPT_NN_CHAIN = 11         /* 1 (default): the warm ICP passes of a SMALL batch (fewer than 48 jobs: one query's
                                       20 candidates) run as ONE launch -- every pass's searches, reductions, solves and
                                       plans laid end to end, a wave of pass p + 1 waiting on the device for its own
                                       job's solve of pass p -- instead of 2 launches per pass with the chip draining
                                       in between; 0: launch by launch.  Identical results.  Every device-side wait is
                                       bounded (3 s): a batch whose chain runs out is run again launch by launch before
                                       the call returns (said once on stderr) and the handle stops chaining until this
                                       option is set again */
};
enum {
  GLOC_REG_NN_CULLED = 0,    /* default: Hilbert-sorted scans, box hierarchy, skip what cannot win */
  GLOC_REG_NN_EXHAUSTIVE = 1 /* every (source, target) pair: the brute-force kernel */
};

/* ---- scan store ---------------------------------------------------------------------------- *
 * The database scans stay resident in HBM with their search index (38 B/point: all 4541 scans of
 * KITTI-00 = 20 GB of the 288 GB) -- the reference re-reads every candidate scan from disk for every
 * query (registration/global_localization.cpp:521-525).  A store is shared by any number of
 * registration handles (gloc_reg_attach_store); query scans are added, used and released.
 * add/release/count are thread-safe; a scan must not be released while a registration that uses it
 * is in flight.  Indexing (bounding box, Hilbert keys, radix sort, boxes) runs on the device; add
 * returns when the scan is ready.  `stride_floats` is 3 for packed xyz or 4 for KITTI x,y,z,i
 * (registration/global_localization.cpp:160-182). */
typedef struct gloc_scan_store gloc_scan_store;
int gloc_scan_store_create(int device, gloc_scan_store** out);
/* GLOC_ERR_STATE while registration handles are still attached. */
int gloc_scan_store_destroy(gloc_scan_store* st);
int gloc_scan_store_add(gloc_scan_store* st, const float* pts, size_t n, size_t stride_floats,
                        uint32_t* scan_id);
/* Same with the points already in device memory on the store's device. */
int gloc_scan_store_add_device(gloc_scan_store* st, const float* d_pts, size_t n,
                               size_t stride_floats, uint32_t* scan_id);
/* `count` scans (host buffers pts[i] of n[i] points) uploaded and indexed in ONE launch sequence: the launch count
 * does not depend on `count` (every indexing kernel covers all the scans, the sorts are segmented).  The query scans
 * of a batch of localizations in flight go in together.  Each scan ends up exactly as gloc_scan_store_add leaves it. */
int gloc_scan_store_add_batch(gloc_scan_store* st, const float* const* pts, const size_t* n, size_t count,
                              size_t stride_floats, uint32_t* scan_ids);
/* Re-sort a resident scan's search index into kd order (the TARGET index): chunks and sub-blocks become
 * disjoint kd cells fitted to the point density instead of runs of a space-filling curve, and the culled 1-NN
 * search of every registration AGAINST this scan tests ~35 % fewer boxes and evaluates ~30 % fewer pairs.
 * Costs about a millisecond of device time per 120k-point scan, once: meant for the database places
 * (db_files_ of the reference's GlocEvaluator, read at registration/global_localization.cpp:521-525), not for
 * query scans, which are added, used as the source once and released.  Results never depend on it (the search is
 * exact either way).  The scan must not be in use by a registration in flight. */
int gloc_scan_store_build_target_index(gloc_scan_store* st, uint32_t scan_id);
/* The same for many scans, in batches of up to 8 M points per launch sequence (a database build). */
int gloc_scan_store_build_target_index_batch(gloc_scan_store* st, const uint32_t* scan_ids, size_t count);
/* Frees the scan's id and memory for reuse by later adds (ids of other scans do not change). */
int gloc_scan_store_release(gloc_scan_store* st, uint32_t scan_id);
int gloc_scan_store_clear(gloc_scan_store* st);
int gloc_scan_store_count(gloc_scan_store* st, size_t* n_scans);
/* HBM held by live scans, and by released allocations kept for reuse (either may be NULL). */
int gloc_scan_store_bytes(gloc_scan_store* st, size_t* live_bytes, size_t* cached_bytes);
int gloc_scan_store_points(gloc_scan_store* st, uint32_t scan_id, size_t* n_points);
/* The scan's points, original order, packed xyz (tests). */
int gloc_scan_store_download(gloc_scan_store* st, uint32_t scan_id, float* out_xyz,
                             size_t capacity_points);

/* Use `store` for every scan id of this handle (NULL: back to the handle's private store, which
 * gloc_reg_scan_upload creates on first use). */
int gloc_reg_attach_store(gloc_reg* h, gloc_scan_store* store);

/* Shims over the handle's current store (private unless one is attached). */
int gloc_reg_scan_upload(gloc_reg* h, const float* pts, size_t n, size_t stride_floats,
                         uint32_t* scan_id);
int gloc_reg_scan_build_target_index(gloc_reg* h, uint32_t scan_id); /* gloc_scan_store_build_target_index */
int gloc_reg_scan_release(gloc_reg* h, uint32_t scan_id);
int gloc_reg_scan_count(const gloc_reg* h, size_t* n_scans);
int gloc_reg_scan_clear(gloc_reg* h);

/* Register one query scan against n_cand candidate scans (host buffers, packed xyz).
 * init_T: n_cand x 16 row-major 4x4 (query -> candidate frame) or NULL for identity.
 * cand_stream_ids: RANSAC sampling stream of each candidate, or NULL for 0..n_cand-1 (its rank in
 * the retrieval list).  A rank that registers only a subset of a query's candidates passes their
 * ranks here so that the result is independent of how candidates are sharded over GPUs.
 * Outputs per candidate: out_T (n_cand x 16, query -> db), out_rmse, out_inliers, out_ok
 * (any may be NULL except out_T). */
int gloc_reg_batch(gloc_reg* h, const float* q_xyz, size_t nq_pts, const float* const* cand_xyz,
                   const size_t* cand_npts, size_t n_cand, const uint32_t* cand_stream_ids,
                   const float* init_T, const gloc_reg_params* params, float* out_T,
                   float* out_rmse, uint32_t* out_inliers, int* out_ok);

/* Same with the query scan and the candidates taken from the scan store. */
int gloc_reg_batch_ids(gloc_reg* h, uint32_t q_scan_id, const uint32_t* cand_scan_ids,
                       size_t n_cand, const uint32_t* cand_stream_ids, const float* init_T,
                       const gloc_reg_params* params, float* out_T, float* out_rmse,
                       uint32_t* out_inliers, int* out_ok);

/* Several queries in flight: query q is registered against candidates cand_scan_ids[q*n_cand ..]
 * (UINT32_MAX = no candidate: that row keeps its initial guess, ok = 0).  Every kernel launch covers
 * the candidates of ALL the queries, on the handle's one stream.  cand_stream_ids: n_queries x n_cand,
 * or NULL for 0..n_cand-1 per query.  init_T and the outputs are n_queries x n_cand rows.  Each row
 * equals what gloc_reg_batch_ids returns for that query alone, bit for bit. */
int gloc_reg_batch_multi(gloc_reg* h, size_t n_queries, const uint32_t* q_scan_ids,
                         const uint32_t* cand_scan_ids, size_t n_cand,
                         const uint32_t* cand_stream_ids, const float* init_T,
                         const gloc_reg_params* params, float* out_T, float* out_rmse,
                         uint32_t* out_inliers, int* out_ok);

/* The same in two halves.  begin: looks the scans up and ENQUEUES the whole launch sequence of the batch plus the copy of
 * its results on the handle's stream, then returns without waiting (nothing it needs from the caller is read later).
 * end: waits for THAT batch's results (its own event, not the stream) and writes the outputs, rows as in
 * gloc_reg_batch_multi.  One batch per handle may be in flight.  Two handles that share a stream (gloc_reg_set_stream)
 * pipeline a stream of batches from one host thread: begin(A, batch i + 1) goes in before end(B, batch i), so the
 * device runs batch after batch while the host post-processes -- bench.py's registration pipeline.  The scans of a batch
 * must stay in the store until its end. */
int gloc_reg_batch_multi_begin(gloc_reg* h, size_t n_queries, const uint32_t* q_scan_ids,
                               const uint32_t* cand_scan_ids, size_t n_cand, const uint32_t* cand_stream_ids,
                               const float* init_T, const gloc_reg_params* params);
int gloc_reg_batch_multi_end(gloc_reg* h, float* out_T, float* out_rmse, uint32_t* out_inliers, int* out_ok);

/* The reference's candidate loop as it is written -- stop at the first match()==true
 * (registration/global_localization.cpp:519-572) -- for several queries at once: rank by rank, the
 * rank-r candidates of all queries still without a success are registered in one batch.  Every job is the
 * one gloc_reg_batch_multi runs for that (query, rank), so out_rank / out_T equal
 * gloc_reg_select_first_ok over its result, bit for bit; only the candidates behind a success are never
 * touched.  out_rank: -1 where no candidate succeeded (out_T = identity); out_jobs_run (may be NULL):
 * registrations actually run. */
int gloc_reg_first_success_multi(gloc_reg* h, size_t n_queries, const uint32_t* q_scan_ids,
                                 const uint32_t* cand_scan_ids, size_t n_cand, const float* init_T,
                                 const gloc_reg_params* params, int* out_rank, float* out_T,
                                 float* out_rmse, uint32_t* out_inliers, uint64_t* out_jobs_run);

/* The reference's selection rule: lowest-rank candidate whose registration succeeded
 * (registration/global_localization.cpp:519-572 stops at the first match()==true).
 * Returns the rank or -1. */
int gloc_reg_select_first_ok(const int* ok, size_t n_cand);
/* The convergence measure of the last batch the handle returned: per job, in job order, the RMS displacement of the
 * last ICP update (what gloc_reg_params.max_final_step is compared with; 0 for a job without an ICP pass). */
int gloc_reg_final_steps(gloc_reg* h, float* out, size_t n_jobs);

/* Building blocks, exposed for tests and for callers that drive ICP themselves (host buffers). */
int gloc_reg_nn(gloc_reg* h, const float* src_xyz, size_t n_src, const float* tgt_xyz,
                size_t n_tgt, const float* T16 /* may be NULL */, uint32_t* out_idx, float* out_d2);
int gloc_reg_ransac_hypotheses(gloc_reg* h, const float* src_xyz, const float* tgt_xyz,
                               const uint32_t* corr, size_t n, uint64_t seed, uint32_t cand,
                               uint32_t n_hyp, float* out_Rt /* n_hyp x 12 */,
                               uint32_t* out_valid, uint32_t* out_inliers, float inlier_thresh);

/* HIP-event time per kernel family ("nn", "ransac_hyp", "ransac_score", "accum", "solve",
 * "transform"), as gloc_knn_profile. */
int gloc_reg_profile(gloc_reg* h, const char* kernel, double* total_ms, uint64_t* launches);
int gloc_reg_profile_reset(gloc_reg* h);
/* With profiling on: (source, target) pairs evaluated by the culled 1-NN kernel and the number of
 * 1-NN launches since the last gloc_reg_profile_reset. */
int gloc_reg_nn_stats(gloc_reg* h, uint64_t* pairs_evaluated, uint64_t* launches);

/* ============================ NetVLAD-FC pooling head ("next" row N2) ===================== *
 * Replaces NetVLAD.forward of the reference (model/netvlad_fc.py:73-109, built without gating at
 * main.py:594) -- the tail of the TorchScript module RpyPCLoopDetector::get_place_feature runs
 * (registration/loop_detector.cpp:152-163): per-position L2 normalisation, 1x1-conv soft assignment,
 * residual aggregation to `clusters` x `dim`, intra-normalisation, L2, FC to `out_dim`.
 * conv_w [clusters][dim], conv_b [clusters] or NULL (vladv2), centroids [clusters][dim],
 * fc_w [clusters*dim][out_dim] (hidden1_weights), all row-major fp32, copied to the device.
 * feat: n feature maps in NCHW order, [n][dim][hw]; out: [n][out_dim].  fp32 throughout. */
typedef struct gloc_vlad gloc_vlad;
int gloc_vlad_create(int device, size_t dim, size_t clusters, size_t out_dim, const float* conv_w,
                     const float* conv_b, const float* centroids, const float* fc_w,
                     int normalize_input, gloc_vlad** out);
/* Optional GatingContext after the FC (model/netvlad_fc.py:106-107,120-146; off in the reference's constructor
 * call, main.py:594): out = y * sigmoid((y W) * scale + shift), W [out_dim][out_dim] = gating_weights.  With
 * add_batch_norm (eval mode) scale = bn.weight / sqrt(bn.running_var + eps), shift = bn.bias -
 * bn.running_mean * scale; without, scale = 1, shift = gating_biases.  gating_w = NULL switches it off. */
int gloc_vlad_set_gating(gloc_vlad* h, const float* gating_w, const float* scale, const float* shift);
int gloc_vlad_destroy(gloc_vlad* h);
int gloc_vlad_set_stream(gloc_vlad* h, void* hip_stream);
int gloc_vlad_forward(gloc_vlad* h, const float* feat, size_t n, size_t hw, float* out);
int gloc_vlad_forward_device(gloc_vlad* h, const float* d_feat, size_t n, size_t hw, float* d_out);
int gloc_vlad_set_profile(gloc_vlad* h, int enable);
/* kernel families: "vlad_tile", "vlad_cluster", "vlad_fc", "vlad_gate" */
int gloc_vlad_profile(gloc_vlad* h, const char* kernel, double* total_ms, uint64_t* launches);

/* ============================ BEV occupancy projection ("next" row N1) ==================== *
 * Replaces RpyPCLoopDetector::get_projected_grid + crop_pad_occupancy + the tensor packing of
 * get_place_feature (registration/loop_detector.cpp:83-106,122-151): one scan inserted into a fresh
 * Submap3D (3d/submap_3d.cpp:162-177, 3d/range_data_inserter_3d.cpp:27-78) and x-ray projected
 * (ProjectToCvMat, 3d/submap_3d.cpp:238-326), then centred-cropped / padded to the network's input
 * size.  Output is byte-identical to the reference's: a pixel is 0 where the column holds two or
 * more distinct hit voxels, else 255; the padding is (255, 0, 0) (cv::Mat::ones sets channel 0
 * only).  Scans are independent: a batch is n_scans scans back to back with host offsets. */
typedef struct gloc_bev gloc_bev;

enum {
  GLOC_BEV_U8_HWC3 = 0, /* [out_height][out_width][3] u8: crop_pad_occupancy's cv::Mat (CV_8UC3) */
  GLOC_BEV_F32_CHW = 1  /* [3][out_height][out_width] f32 in {0,1}: the module's input tensor */
};

typedef struct gloc_bev_params {
  float resolution;    /* 0.2 m: high_resolution_, loop_detector.h:116 */
  float max_range;     /* 100 m: loop_detector.cpp:113 and high_resolution_max_range_, loop_detector.h:115 */
  uint32_t out_width;  /* 768: loop_detector.cpp:142 */
  uint32_t out_height; /* 768: loop_detector.cpp:143 */
  uint32_t format;     /* GLOC_BEV_* */
  uint8_t pad_bgr[3];  /* 255, 0, 0: loop_detector.cpp:84 */
  uint8_t reserved_;
} gloc_bev_params;

typedef struct gloc_bev_info {
  int32_t min_ix, min_iy, max_ix, max_iy; /* voxel-index box of the occupied columns */
  uint32_t width, height;                 /* size of the uncropped image (occupancy_grid) */
  uint32_t n_returns;                     /* points kept by both range tests */
  uint32_t empty;                         /* 1: no point kept (the reference aborts); image = padding */
  double ox, oy, resolution;              /* xy_res: min index * resolution, loop_detector.cpp:133 */
} gloc_bev_info;

int gloc_bev_default_params(gloc_bev_params* p);
int gloc_bev_create(int device, gloc_bev** out);
int gloc_bev_destroy(gloc_bev* h);
int gloc_bev_set_stream(gloc_bev* h, void* hip_stream);
int gloc_bev_synchronize(gloc_bev* h);
/* One scan, host buffers: xyz = n points, stride_floats apart (3 packed, 4 for x y z i). */
int gloc_bev_project(gloc_bev* h, const float* xyz, size_t n, size_t stride_floats,
                     const gloc_bev_params* p, void* out_image, gloc_bev_info* info);
/* n_scans scans, device buffers: scan i = points [offsets[i], offsets[i+1]) of d_xyz (offsets on
 * the host, in points); d_out_images holds n_scans images back to back.  infos (host, n_scans) may
 * be NULL, in which case the call does not synchronise. */
int gloc_bev_project_batch_device(gloc_bev* h, const float* d_xyz, const uint64_t* offsets,
                                  size_t n_scans, size_t stride_floats, const gloc_bev_params* p,
                                  void* d_out_images, gloc_bev_info* infos);
/* The uncropped single-channel image of scan `scan` of the last projection (the occupancy_grid
 * get_place_feature hands back, loop_detector.cpp:139-140): [height][width] u8 into `out`. */
int gloc_bev_raw_image(gloc_bev* h, size_t scan, uint8_t* out, size_t capacity);
/* Device pointer of the column flags of scan `scan` of the last projection: flags[(iy + R) * S + (ix + R)]
 * != 0 iff the BEV pixel of voxel column (ix, iy) is occupied (value 0 in the image).  Valid until the
 * next projection on this handle; used by the coarse matcher to stay on the device. */
int gloc_bev_device_flags(gloc_bev* h, size_t scan, const uint8_t** d_flags, int* R, int* S);
int gloc_bev_set_profile(gloc_bev* h, int enable);
/* kernel families: "bev_clear", "bev_mark", "bev_flag", "bev_image" */
int gloc_bev_profile(gloc_bev* h, const char* kernel, double* total_ms, uint64_t* launches);

/* ============================ coarse global (x, y, yaw) match (row a-12) =================== *
 * Replaces RpyPCLoopDetector::match(q_grid, db_idx, xy_yaw, scale) (registration/loop_detector.cpp:186-288):
 * the coarse pose of the query in a database place's frame from their two BEV occupancy images,
 * p_db = R(yaw) p_q + (x, y).  The reference finds it with SURF keypoints + FLANN matching + a RANSAC
 * partial-affine fit (OpenCV + contrib, absent here); this finds it with an exhaustive integer search on
 * the GPU -- every yaw step x every shift, scored by how many occupied query cells land on occupied
 * database cells (coarse_kernels.hpp) -- so a reverse-direction revisit is handled as well as a small
 * offset, and a scale estimate with the reference's |1 - scale| < 0.1 acceptance on top (round 3).
 * The result seeds the 3-D registration (init_T of gloc_reg_batch*), as the reference seeds its pose
 * composition (global_localization.cpp:526-570).  Parity: unpinned upstream (third-party arithmetic, no
 * fixtures); oracle/coarse_oracle.c states the search step by step and the GPU equals it exactly. */
typedef struct gloc_coarse gloc_coarse;

typedef struct gloc_coarse_params {
  float resolution;    /* 0.2 m: the BEV pixel (loop_detector.h:116) */
  uint32_t cell_px;    /* 2: a search cell is cell_px x cell_px pixels (0.4 m) */
  uint32_t n_yaw;      /* 360 yaw steps */
  uint32_t max_shift;  /* 64 cells: |x|, |y| <= 25.6 m */
  uint32_t top_yaw;    /* 12: yaw steps verified in 2-D (the identity is always verified too) */
  uint32_t refine;     /* 4 cells: 2-D window around the best x / y lags */
  float min_overlap;   /* 0.25: ok iff overlapping cells >= this x occupied query cells (and >= 16 cells) */
  uint32_t reserved_;
} gloc_coarse_params;

int gloc_coarse_default_params(gloc_coarse_params* p);
int gloc_coarse_create(int device, gloc_coarse** out);
int gloc_coarse_destroy(gloc_coarse* h);
/* A place's grid from its occupancy image as get_projected_grid returns it (OccupancyGrid of
 * loop_detector.h:36-39: [height][width] u8, below 100 = occupied as the reference's threshold,
 * ox_oy_res) -- what add_keyframe keeps in db_grids_ (loop_detector.cpp:16-19). */
int gloc_coarse_add_image(gloc_coarse* h, const uint8_t* occupancy, uint32_t width, uint32_t height, float ox,
                          float oy, float resolution, const gloc_coarse_params* params, uint32_t* grid_id);
/* The same straight from the scan (BEV projection at 0.2 m / 100 m on the device, no image round trip). */
int gloc_coarse_add_scan(gloc_coarse* h, const float* xyz, size_t n, size_t stride_floats,
                         const gloc_coarse_params* params, uint32_t* grid_id);
/* ... or from a scan resident in a scan store on the same device (no host copy of the points needed). */
int gloc_coarse_add_store_scan(gloc_coarse* h, gloc_scan_store* store, uint32_t scan_id,
                               const gloc_coarse_params* params, uint32_t* grid_id);
/* n scans of the store in one launch sequence (the query scans of a step, a database being loaded): two
 * synchronisations for the whole batch instead of three per grid. */
int gloc_coarse_add_store_scans(gloc_coarse* h, gloc_scan_store* store, const uint32_t* scan_ids, size_t n,
                                const gloc_coarse_params* params, uint32_t* grid_ids);
int gloc_coarse_release(gloc_coarse* h, uint32_t grid_id);
/* Occupied cells of a grid ((v << 16) | u, u / v in [0, 512): cell u spans the pixels
 * (u - 256) cell_px .. + cell_px - 1); out_cells may be NULL to get the count only. */
int gloc_coarse_cells(gloc_coarse* h, uint32_t grid_id, uint32_t* n_cells, uint32_t* out_cells,
                      size_t capacity);
/* One query grid against n_db database grids: out_xy_yaw [n_db][3] = (x, y, yaw in (-pi, pi]),
 * out_ratio = overlapping / occupied query cells, out_scale = the `scale` of the reference's match (the similarity
 * factor cv::estimateAffinePartial2D fits, loop_detector.cpp:262-266): here the factor in 0.88 .. 1.12 by which the
 * query, scaled about the sensor under the chosen rotation, overlaps the database grid most (parabola-refined; an end of
 * the range means "10 % or more off"; 0: no overlap at all); out_ok = enough overlap
 * AND |1 - scale| < 0.1, the reference's acceptance (loop_detector.cpp:268-272).  Any of the three may be NULL. */
int gloc_coarse_match(gloc_coarse* h, uint32_t q_grid, const uint32_t* db_grids, size_t n_db,
                      const gloc_coarse_params* params, float* out_xy_yaw, float* out_ratio, int* out_ok,
                      float* out_scale);
/* n independent (query grid, database grid) pairs in one launch sequence (several queries in flight). */
int gloc_coarse_match_pairs(gloc_coarse* h, const uint32_t* q_grids, const uint32_t* db_grids, size_t n_pairs,
                            const gloc_coarse_params* params, float* out_xy_yaw, float* out_ratio,
                            int* out_ok, float* out_scale);

/* ============================ ground pre-alignment ("next" row N3) ========================= *
 * Replaces GroundEstimator::EsitmateGroundAndTransform (registration/ground_estimator.cpp:196-228),
 * the optional 4th-argument mode of global_localization (registration/global_localization.cpp:431-436,
 * 495-499): points within 20 m -> a normal per point from its 10 nearest neighbours -> the fullest
 * 10-degree elevation bin outside 5..12 is the ground -> plane RANSAC (0.1 m) -> T_l2g (roll, pitch and
 * height; yaw removed).  The reference leaves the numerics to PCL and Eigen; the exact arithmetic of
 * this implementation is stated step by step in oracle/ground_oracle.c (parity unpinned). */
typedef struct gloc_ground gloc_ground;

typedef struct gloc_ground_params {
  float near_range2;     /* 400 = (20 m)^2: ground_estimator.cpp:203 */
  uint32_t knn;          /* 10: ground_estimator.cpp:79 (3..16) */
  float plane_thresh;    /* 0.1 m: ground_estimator.cpp:27 */
  uint32_t ransac_iters; /* 1000: pcl::SampleConsensus max_iterations_ default */
  float ransac_conf;     /* 0.99: pcl::SampleConsensus probability_ default; <= 0 or >= 1: no early stop */
  uint32_t reserved_;
  uint64_t seed;
} gloc_ground_params;

typedef struct gloc_ground_info {
  uint32_t n_near;      /* points within the range filter */
  uint32_t hist[18];    /* elevation bins of the normals, 0 = pointing down .. 17 = pointing up */
  int32_t ground_bin;   /* -1: none */
  uint32_t n_ground;
  uint32_t best_hyp, inliers, iters_used;
  float plane[4];       /* a x + b y + c z + d = 0 with unit normal, as fitted */
  int32_t found;        /* 0: no ground, T = identity (ground_estimator.cpp:218-220) */
} gloc_ground_info;

int gloc_ground_default_params(gloc_ground_params* p);
int gloc_ground_create(int device, gloc_ground** out);
int gloc_ground_destroy(gloc_ground* h);
int gloc_ground_set_stream(gloc_ground* h, void* hip_stream);
enum { GLOC_GROUND_OPT_KNN_EXHAUSTIVE = 1 /* 1: every pair instead of the chunk-culled search; same lists */ };
int gloc_ground_set_option(gloc_ground* h, int option, int64_t value);
/* T16: T_l2g, row-major 4x4 f32 (host).  out_xyz (may be NULL): the cloud transformed by T_l2g, same
 * layout as the input (extra channels copied) -- cloud_out of the reference. */
int gloc_ground_estimate(gloc_ground* h, const float* xyz, size_t n, size_t stride_floats,
                         const gloc_ground_params* p, float* T16, gloc_ground_info* info, float* out_xyz);
int gloc_ground_estimate_device(gloc_ground* h, const float* d_xyz, size_t n, size_t stride_floats,
                                const gloc_ground_params* p, float* T16, gloc_ground_info* info,
                                float* d_out_xyz);
/* Building blocks, exposed for tests (host buffers, packed xyz): the exact k nearest neighbours of every
 * point within the cloud itself ([n][k], ascending (d2, index), the point itself first); the normals
 * and their elevation bins. */
int gloc_ground_knn(gloc_ground* h, const float* xyz, size_t n, uint32_t k, uint32_t* out_idx, float* out_d2);
int gloc_ground_normals(gloc_ground* h, const float* xyz, size_t n, uint32_t k, float* out_normals,
                        uint8_t* out_bins);
/* T_l2g from plane coefficients (TransformPointsToGround, ground_estimator.cpp:163-194); host only. */
int gloc_ground_transform_from_plane(const float* plane4, float* T16);
int gloc_ground_set_profile(gloc_ground* h, int enable);
/* kernel families: "ground_knn", "ground_normals", "ground_plane", "ground_transform" */
int gloc_ground_profile(gloc_ground* h, const char* kernel, double* total_ms, uint64_t* launches);

/* ============================ synthetic inputs (bench / tests) ============================ *
 * On-device twin of gloc3d_amd/synth.py for databases too large to upload (SURVEY.md 8d cfg E).
 * kind 0: iid N(0,1)/sqrt(dim); kind 1: anchored trajectory (stride 16, noise 0.05).
 * Appends n rows of the global synthetic database to the handle: global rows first_row,
 * first_row + row_stride, ... (row_stride = G gives rank first_row's interleaved shard). */
int gloc_knn_add_synthetic(gloc_knn* h, int kind, uint64_t seed, uint64_t first_row, size_t n,
                           uint64_t row_stride);
int gloc_synth_fill_device(int device, void* hip_stream, int kind, uint64_t seed,
                           uint64_t first_row, size_t n, size_t dim, uint64_t row_stride,
                           float* d_out);
/* A new resident scan made on the device from scan `base_id`: point i = T p_i + sigma * gauss(seed, i)
 * per coordinate (T16 row-major 4x4 or NULL = identity; gloc3d_amd/synth.py::scan_variant gives the
 * same bits).  Fills a KITTI-00-sized store (4541 distinct scans) in seconds. */
int gloc_scan_store_add_variant(gloc_scan_store* st, uint32_t base_id, const float* T16,
                                float noise_sigma, uint64_t seed, uint32_t* scan_id);

/* `count` new resident scans RAY-CAST on the device (bench / tests; round 6): a spinning lidar of n_beams x n_az rays
 * (elevations linspace(fov_lo_deg, fov_hi_deg, n_beams), azimuths [0, 2 pi) -- ray r = beam * n_az + az) at the sensor
 * pose T16[i] (world <- sensor, row-major 4x4 DOUBLES) over a world of axis-aligned boxes [box_lo, box_hi] (doubles,
 * [n][3]; scan i sees boxes box_first[i] .. box_first[i + 1] - 1 of the arrays -- the caller passes the boxes within
 * reach of each pose) and a ground plane z = ground_z; a ray returns at the nearest hit below max_range (a box face
 * nearer than 0.5 m is ignored), its range gets noise_sigma * gauss(seeds[i], r), the point is kept in the SENSOR frame,
 * returns stay in ray order.  The twin of gloc3d_amd/synth.py::lidar_scan (fp64 throughout, one rounding to fp32): equal
 * to it to ~1e-6 m.  This is how bench.py makes SURVEY.md 8d cfg D's "4541-pose loop trajectory through one procedural
 * world" -- 4541 DISTINCT casts, the KITTI .bin point layout of registration/global_localization.cpp:160-182 minus the
 * intensity -- in seconds instead of rigid copies of a few host-cast views. */
typedef struct gloc_raycast_params {
  uint32_t n_beams, n_az;          /* 64 x 2000 (HDL-64E-like) */
  double max_range, noise_sigma;   /* 80 m, 0.02 m */
  double fov_lo_deg, fov_hi_deg;   /* -24.8, 2.0 */
} gloc_raycast_params;
int gloc_scan_store_add_raycast_batch(gloc_scan_store* st, size_t count, const double* box_lo, const double* box_hi,
                                      const uint32_t* box_first /* [count + 1] */, double ground_z,
                                      const double* T16 /* [count][16] */, const uint64_t* seeds /* [count] */,
                                      const gloc_raycast_params* params, uint32_t* scan_ids /* [count] */);

#ifdef __cplusplus
}
#endif
#endif /* GLOC3D_H */
