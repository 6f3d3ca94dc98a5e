// knn.hip -- C ABI of the descriptor kNN (include/gloc3d.h) over the kernels in knn_kernels.hpp.
// Replaces registration/loop_detector.cpp:34-45,66-79 (KD-tree build + query) and
// main.py:317-324 (faiss.IndexFlatL2 add/search) of the reference.
#include <algorithm>
#include <atomic>
#include <cfloat>
#include <cstdlib>
#include <vector>
#include <new>

#include "comm.hpp"
#include "common.hpp"
#include "knn_kernels.hpp"
#include "synth_kernels.hpp"

using namespace gloc;
using namespace gloc::knn;

struct gloc_knn {
  int device = 0;
  size_t dim = 0;
  size_t n = 0;
  hipStream_t own_stream = nullptr;
  hipStream_t stream = nullptr;
  DevBuf rows;      // n x dim fp32, row-major, dense
  DevBuf norms;     // n fp32 (coarse form only)
  DevBuf mirror;    // the rows again, split into two bf16 values and tiled by 64 rows (knn_kernels.hpp: mirror_rows_kernel) --
                    // what the split-bf16 coarse pass streams; dim % 8 == 0 only; kept current by every add
  DevBuf dn_max;    // 1 x uint32 (bits of the largest row norm)
  DevBuf dist;      // exact: [nq][ld]; mfma: [splits][Qpad][ld]
  DevBuf keys;      // select output [nq][K]
  DevBuf n_incomplete;  // 1 x u64: queries the MFMA path sent to the exact fallback (device counter)
  DevBuf shard_ws;  // gloc_knn_search_sharded: local + gathered lists
  DevBuf klists, klists2;  // per-chunk K-lists during selection
  DevBuf exact;     // rerank: reference-order distances [nq][KC]
  DevBuf keys2;     // rerank output [nq][k]
  DevBuf qnorm;     // [nq]
  DevBuf qsplit;    // [nq][dim / 8][8 bf16 h, 8 bf16 m]: the queries of the split-bf16 coarse pass
  DevBuf dev_trace; // developer aid: phase stamps of the fused select + re-rank kernel (null unless enabled)
  DevBuf flags;     // [nq] int
  DevBuf redo_tickets;  // [nq] u32: flagged_redo_kernel's tickets (0 between searches)
  DevBuf bmin;          // [nq][blocks of 32 rows]: the coarse kernel's block minima (large windows, one K-split)
  DevBuf stage_q;   // host-API staging: queries
  DevBuf stage_idx, stage_d2;
  int* h_flags = nullptr;  // pinned
  size_t h_flags_cap = 0;
  // The split-bf16 coarse pass proves less on data whose neighbours lie close together relative to the norms (its bound is
  // eight times the fp32 form's): a search whose queries mostly went to the exact pass costs far more than the fp32 coarse pass
  // would have.  The fallback counter is copied to the host behind every tracked search (no wait: it is looked at when the
  // next search finds the copy done); above a quarter of the queries the handle's next 64 searches take the fp32 form, then
  // the split form is tried again.  Results do not depend on it.
  unsigned long long* h_inc = nullptr;  // pinned: the counter as of the tracked search
  hipEvent_t inc_ev = nullptr;
  bool inc_pending = false;
  unsigned long long inc_seen = 0;      // the counter at the last look
  size_t inc_queries = 0;               // queries of the tracked search
  int coarse_fp32_left = 0;             // searches still to run on the fp32 coarse pass
  int algo = GLOC_KNN_ALGO_AUTO;
  int candidates = 32;
  Profiler prof;
  gloc_knn_stats stats{};
  // gloc_knn_create_view: a view searches its parent's rows / norms with its own stream and workspace (rows, norms, dn_max
  // below are then ALIASES of the parent's buffers, refreshed at every search and never grown or freed through the view)
  gloc_knn* parent = nullptr;
  int views = 0;  // live views of this handle
};

namespace {

#define GLOC_NOT_VIEW(h) \
  GLOC_REQUIRE(!(h)->parent, GLOC_ERR_STATE, "a view (gloc_knn_create_view) searches its parent's rows: add / reserve / clear / load on the parent")

// A view's rows are its parent's as of now (the parent may have grown since the view's last search).
void view_sync(gloc_knn* h) {
  if (!h->parent) return;
  h->rows.p = h->parent->rows.p;
  h->rows.cap = h->parent->rows.cap;
  h->norms.p = h->parent->norms.p;
  h->norms.cap = h->parent->norms.cap;
  h->mirror.p = h->parent->mirror.p;
  h->mirror.cap = h->parent->mirror.cap;
  h->dn_max.p = h->parent->dn_max.p;
  h->dn_max.cap = h->parent->dn_max.cap;
  h->n = h->parent->n;
}

int ensure_rows(gloc_knn* h, size_t n_rows) {
  GLOC_NOT_VIEW(h);
  GLOC_TRY(h->rows.ensure(n_rows * h->dim * sizeof(float), h->stream, true,
                          h->n * h->dim * sizeof(float)));
  GLOC_TRY(h->norms.ensure(n_rows * sizeof(float), h->stream, true, h->n * sizeof(float)));
  if (h->dim % 8 == 0) {  // (whole tiles, and two more: the coarse kernel's last work-group may own a tile past the end)
    const size_t tile_bytes = mirror_tile_u32x4((int)h->dim) * 16;
    // (+ 8 KB: a step's four planes of the last tile when dim / 8 is no multiple of four -- dist_bf16x3_tiled_kernel)
    GLOC_TRY(h->mirror.ensure(((n_rows + MIR_ROWS - 1) / MIR_ROWS + 2) * tile_bytes + 8192, h->stream, true,
                              (h->n + MIR_ROWS - 1) / MIR_ROWS * tile_bytes));
  }
  if (!h->dn_max.p) {
    GLOC_TRY(h->dn_max.ensure(sizeof(uint32_t), h->stream));
    GLOC_HIP(hipMemsetAsync(h->dn_max.p, 0, sizeof(uint32_t), h->stream));
  }
  return GLOC_OK;
}

int update_norms(gloc_knn* h, size_t first, size_t count) {
  if (!count) return GLOC_OK;
  ProfScope ps(h->prof, "norms", h->stream);
  const unsigned blocks = (unsigned)((count + 3) / 4);
  hipLaunchKernelGGL(row_norms_kernel, dim3(blocks), dim3(256), 0, h->stream,
                     h->rows.as<float>() + first * h->dim, count, (int)h->dim,
                     h->norms.as<float>() + first, h->dn_max.as<uint32_t>());
  GLOC_HIP(hipGetLastError());
  if (h->dim % 8 == 0) {  // the coarse pass's mirror of the same rows
    ProfScope pm(h->prof, "mirror", h->stream);
    hipLaunchKernelGGL(mirror_rows_kernel, dim3((unsigned)((count + 63) / 64), (unsigned)((h->dim / 8 + 3) / 4)), dim3(256), 0, h->stream,
                       h->rows.as<float>(), first, count, (int)h->dim, h->mirror.as<u32x4>());
    GLOC_HIP(hipGetLastError());
  }
  return GLOC_OK;
}

// ---- exact path ------------------------------------------------------------------------------
int launch_dist_exact(gloc_knn* h, const float* d_q, int nq, size_t first, int n_range,
                      size_t ld, const int* only_flagged = nullptr) {
  ProfScope ps(h->prof, "dist_exact", h->stream);
  const int QT = only_flagged ? 1 : (nq >= 8 ? 8 : (nq >= 4 ? 4 : (nq >= 2 ? 2 : 1)));
  const int qgroups = (nq + QT - 1) / QT;
  if (nq <= 2 && !getenv("GLOC3D_KNN_NO_SMALL")) {
    // one or two queries: the streaming form (one wave per work-group, all group sums before the chains);
    // measured against the general kernel at 4541 x 4096: Q = 1 18.6 vs 33 us, Q = 2 equal, Q = 4 / 8 slower
    const int G = (int)h->dim >> 2, Gs = std::min(G, EXS_G);
    float* dist = h->dist.as<float>();
    const float* db = h->rows.as<float>();
#define LAUNCH_SMALL(QT_, RW_)                                                                             \
  hipLaunchKernelGGL((dist_exact_small_kernel<QT_, RW_>), dim3((unsigned)((n_range + RW_ - 1) / RW_), (unsigned)qgroups), \
                     dim3(64), sizeof(float) * (QT_ * RW_) * (Gs + 4), h->stream, db, d_q, dist, (int)h->dim, \
                     first, n_range, nq, ld, only_flagged)
    if (QT == 2) LAUNCH_SMALL(2, 4);
    else LAUNCH_SMALL(1, 4);
#undef LAUNCH_SMALL
    GLOC_HIP(hipGetLastError());
    return GLOC_OK;
  }
  // rows per wave: fill the chip with >= ~2048 waves when the window is small, up to 64/QT
  int RW = 64 / QT;
  while (RW > 4 && (long long)((n_range + RW - 1) / RW) * qgroups < 2048) RW >>= 1;
  if (RW < 4) RW = 4;
  const unsigned gx = (unsigned)((n_range + 4 * RW - 1) / (4 * RW));
  // (the flagged pass over a large window: y = 1, every work-group walks the flags)
  dim3 grid(gx, (only_flagged && (long long)gx * qgroups > 4096) ? 1u : (unsigned)qgroups), block(256);
  float* dist = h->dist.as<float>();
  const float* db = h->rows.as<float>();
#define LAUNCH_EXACT(QT_)                                                                      \
  hipLaunchKernelGGL(dist_exact_kernel<QT_>, grid, block, 0, h->stream, db, d_q, dist,         \
                     (int)h->dim, first, n_range, nq, RW, ld, only_flagged)
  switch (QT) {
    case 8: LAUNCH_EXACT(8); break;
    case 4: LAUNCH_EXACT(4); break;
    case 2: LAUNCH_EXACT(2); break;
    default: LAUNCH_EXACT(1); break;
  }
#undef LAUNCH_EXACT
  GLOC_HIP(hipGetLastError());
  return GLOC_OK;
}

// Per-query top-K of h->dist: chunked threshold selection + merge(s).  MODE as select_chunk_kernel.
constexpr int SELECT_ONE_BLOCK_MAX = 16384;  // rows one work-group per query selects from in one launch

// Windows above 16 384 rows: S slices of L rows (a multiple of 64, <= 16 384), one work-group per (slice, query);
// more slices than the window needs while the launch would leave CUs idle.
struct SlicePlan {
  int S, L;
};
bool plan_slices(int n_range, int nq, int K, SlicePlan* out) {
  static const bool off = getenv("GLOC3D_KNN_NO_SLICES") != nullptr;  // developer switch: the chunked form
  if (off) return false;
  long long s = ((long long)n_range + SELQ_MAX_ROWS - 1) / SELQ_MAX_ROWS;
  while (s * nq < 512 && n_range / (s * 2) >= 4096 && s * 2 * K <= SELQ_MAX_ROWS) s *= 2;
  const int L = (int)((((long long)n_range + s - 1) / s + 63) & ~63ll);
  const int S = (n_range + L - 1) / L;  // (no empty slice)
  if ((long long)S * K > SELQ_MAX_ROWS || S > 65535) return false;  // the lists no longer fit one work-group's registers
  *out = SlicePlan{S, L};
  return true;
}
template <int MODE>
int launch_slices(gloc_knn* h, const float* d_q, int nq, int K, size_t first, int n_range, size_t ld, size_t strideP,
                  int n_splits, const SlicePlan& sl, const int* only_flagged) {
  GLOC_TRY(h->klists.ensure((size_t)nq * sl.S * K * sizeof(uint64_t), h->stream));
  // (the flagged pass: y = 1, every work-group walks the flags)
#define SLICES_ARGS                                                                                                 \
  dim3(SELQ_THREADS), 0, h->stream, h->dist.as<float>(), ld, strideP, n_splits, h->qnorm.as<float>(), d_q, (int)h->dim, \
      h->norms.as<float>(), first, n_range, sl.L, K, h->klists.as<uint64_t>(), only_flagged, nq
  if constexpr (MODE == 0) {  // (only the exact pass is ever launched for flagged queries)
    if (only_flagged)
      hipLaunchKernelGGL((select_slices_kernel<0, true>), dim3(sl.S, 1), SLICES_ARGS);
    else
      hipLaunchKernelGGL((select_slices_kernel<0, false>), dim3(sl.S, nq), SLICES_ARGS);
  } else {
    GLOC_REQUIRE(!only_flagged, GLOC_ERR_STATE, "internal: flagged selection of coarse distances");
    hipLaunchKernelGGL((select_slices_kernel<MODE, false>), dim3(sl.S, nq), SLICES_ARGS);
  }
#undef SLICES_ARGS
  GLOC_HIP(hipGetLastError());
  return GLOC_OK;
}

template <int MODE>
int run_select(gloc_knn* h, const float* d_q, int nq, int K, size_t first, int n_range, size_t ld, size_t strideP,
               int n_splits, uint64_t* d_keys_out, const int* only_flagged = nullptr,
               const FinalOut& fo = FinalOut{nullptr, nullptr, 0, 1}, bool* finalized = nullptr) {
  if (finalized) *finalized = false;
  ProfScope ps(h->prof, "select", h->stream);
  static const bool no_selq = getenv("GLOC3D_KNN_NO_SELECT_QUERY") != nullptr;  // developer switch: the chunked form
  if (n_range <= SELQ_MAX_ROWS && K <= 64 && !no_selq) {  // one launch, one work-group per query
    hipLaunchKernelGGL(select_query_kernel<MODE>, dim3(nq), dim3(SELQ_THREADS), 0, h->stream, h->dist.as<float>(), ld,
                       strideP, n_splits, h->qnorm.as<float>(), d_q, (int)h->dim, h->norms.as<float>(), first, n_range,
                       K, d_keys_out, only_flagged, fo);
    GLOC_HIP(hipGetLastError());
    if (finalized) *finalized = fo.idx != nullptr;
    return GLOC_OK;
  }
  SlicePlan sl;
  if (K <= 64 && !no_selq && plan_slices(n_range, nq, K, &sl)) {  // two launches: slices, then their lists
    GLOC_TRY(launch_slices<MODE>(h, d_q, nq, K, first, n_range, ld, strideP, n_splits, sl, only_flagged));
    hipLaunchKernelGGL(select_query_kernel<2>, dim3(nq), dim3(SELQ_THREADS), 0, h->stream,
                       reinterpret_cast<const float*>(h->klists.as<uint64_t>()), (size_t)2 * sl.S * K, (size_t)0, 1,
                       (float*)nullptr, (const float*)nullptr, (int)h->dim, (const float*)nullptr, (size_t)0, sl.S * K, K,
                       d_keys_out, only_flagged, fo);
    GLOC_HIP(hipGetLastError());
    if (finalized) *finalized = fo.idx != nullptr;
    return GLOC_OK;
  }
  const int per_group = SEL_LIST / K;  // lists one merge can take
  int E = (n_range + 256 * per_group - 1) / (256 * per_group);
  E = std::max(E, 8);
  E = std::min(E, std::max(1, SEL_LIST / K));
  // the device-side fallback (only_flagged) selects with ONE work-group per query, so that no merge
  // launch is needed (measured for the main path: one work-group per query over 10 000 rows takes 47 us,
  // five chunks + one merge 18 + 10 us -- the main path keeps its chunks)
  const int e_one = (n_range + 255) / 256;
  if (only_flagged && n_range <= SELECT_ONE_BLOCK_MAX && e_one <= std::max(1, SEL_LIST / K)) E = std::max(e_one, 1);
  int nlists = (n_range + 256 * E - 1) / (256 * E);
  GLOC_TRY(h->klists.ensure((size_t)nq * nlists * K * sizeof(uint64_t), h->stream));
  uint64_t* cur = nlists == 1 ? d_keys_out : h->klists.as<uint64_t>();
  hipLaunchKernelGGL(select_chunk_kernel<MODE>, dim3(nlists, nq), dim3(256), 0, h->stream,
                     h->dist.as<float>(), ld, strideP, n_splits, h->qnorm.as<float>(), d_q, (int)h->dim,
                     h->norms.as<float>(), first, n_range, K, E, cur, only_flagged);
  GLOC_HIP(hipGetLastError());
  bool flip = false;
  while (nlists > 1) {
    const int ngroups = (nlists + per_group - 1) / per_group;
    uint64_t* nxt;
    if (ngroups == 1) {
      nxt = d_keys_out;
    } else {
      DevBuf& b = flip ? h->klists : h->klists2;
      GLOC_TRY(b.ensure((size_t)nq * ngroups * K * sizeof(uint64_t), h->stream));
      nxt = b.as<uint64_t>();
    }
    hipLaunchKernelGGL(select_merge_kernel, dim3(ngroups, nq), dim3(256), 0, h->stream, cur, nlists,
                       per_group, K, nxt);
    GLOC_HIP(hipGetLastError());
    cur = nxt;
    nlists = ngroups;
    flip = !flip;
  }
  return GLOC_OK;
}

int run_exact(gloc_knn* h, const float* d_q, int nq, int k, size_t first, int n_range,
              uint64_t* d_keys_out, const FinalOut& fo = FinalOut{nullptr, nullptr, 0, 1}, bool* finalized = nullptr) {
  const size_t ld = ((size_t)n_range + 63) & ~(size_t)63;
  GLOC_TRY(h->dist.ensure((size_t)nq * ld * sizeof(float), h->stream));
  GLOC_TRY(launch_dist_exact(h, d_q, nq, first, n_range, ld));
  return run_select<0>(h, d_q, nq, k, first, n_range, ld, 0, 1, d_keys_out, nullptr, fo, finalized);
}

// ---- MFMA path -------------------------------------------------------------------------------
struct MfmaPlan {
  int WQ, NT, KS, BQ, BN;
  int t32 = 0;  // 1: the 32 x 32 x 2 tiles (dist_mfma32_kernel: BQ = 64, BN = 64 * NT)
  int b3 = 0;   // 1: the split-bf16 form (dist_bf16x3_kernel: BQ = 64, BN = 64 * NT)
};

MfmaPlan plan_mfma(int nq, int n_range, int dim, bool fp32_only) {
  // The split-bf16 coarse pass (round 4): the matrix cores stop being the bound, the rows' stream from HBM is.  Tiles of
  // 64 queries x 128 rows when those alone fill the CUs twice, 64 rows otherwise; K split until ~512 work-groups.
  static const bool no_b3 = getenv("GLOC3D_KNN_NO_BF16X3") != nullptr;  // developer switch
  if (!fp32_only && !no_b3 && dim % 8 == 0 && dim >= 8) {
    const int qblocks = (nq + 63) / 64;
    int nt = ((long long)((n_range + 127) / 128) * qblocks >= 512) ? 2 : 1;
    int ks = 1;
    if (const char* e = getenv("GLOC3D_KNN_B3")) {  // developer override: "NT,KS"
      int a = 0, b = 0;
      if (sscanf(e, "%d,%d", &a, &b) == 2 && (a == 1 || a == 2) && b >= 1 && b <= 16) nt = a, ks = -b;
    }
    const long long tiles = (long long)((n_range + 64 * nt - 1) / (64 * nt)) * qblocks;
    if (ks < 0) {
      ks = -ks;
      while (ks > 1 && !((dim % (64 * ks)) == 0 && dim / ks >= 128)) ks /= 2;
    } else {
      while (tiles * ks < 512 && ks < 16 && (dim % (64 * ks * 2)) == 0 && dim / (ks * 2) >= 128) ks *= 2;
    }
    MfmaPlan b{4, nt, ks, 64, 64 * nt};
    b.b3 = 1;
    return b;
  }
  MfmaPlan best{};
  double best_cost = 1e300;
  const int WQ = nq <= 16 ? 1 : (nq <= 32 ? 2 : 4);
  static const int NT4[] = {2, 3, 4, 5, 6, 8}, NT2[] = {2, 4}, NT1[] = {1, 2};
  const int* nts = WQ == 4 ? NT4 : (WQ == 2 ? NT2 : NT1);
  const int n_nts = WQ == 4 ? 6 : 2;
  const int BQ = 16 * WQ;
  const int qblocks = (nq + BQ - 1) / BQ;
  for (int i = 0; i < n_nts; ++i) {
    const int NT = nts[i], BN = 16 * NT * (4 / WQ);
    const long long tiles = (long long)((n_range + BN - 1) / BN) * qblocks;
    // split K only when the tiles alone cannot give every CU a work-group (the partial sums cost
    // KS x Q x N x 8 B of extra traffic and a longer rounding chain)
    int KS = 1;
    // (one block of queries over >= 64 row tiles: two work-groups per CU overlap each other's LDS hand-offs;
    // measured at 64 x 10 000 and 25 x 4541 x 4096: -4 / -3 us; 128 x 16 000 and 32 x 2000: +4 us, so not there)
    static const long long wgs_env = getenv("GLOC3D_MFMA_WGS") ? atoll(getenv("GLOC3D_MFMA_WGS")) : 0;  // developer override
    const long long want_wgs = wgs_env ? wgs_env : ((qblocks == 1 && tiles >= 64) ? 400 : 200);
    while (tiles * KS < want_wgs && KS < 16 && (dim % (64 * KS * 2)) == 0 && dim / (KS * 2) >= 128) KS *= 2;
    const int klen = (dim + KS - 1) / KS;
    const long long wgs = tiles * KS;
    const long long rounds = (wgs + 255) / 256;
    // per-WG time ~ klen * (MFMA issue for BN rows + staging of BQ+BN rows)
    const double per_wg = (double)klen * ((double)BN * 1.0 + (double)(BQ + BN) * 0.35);
    const double cost = (double)rounds * per_wg * (1.0 + 0.03 * (KS - 1)) + (wgs < 128 ? 1e7 : 0);
    if (cost < best_cost) {
      best_cost = cost;
      best = MfmaPlan{WQ, NT, KS, BQ, BN};
    }
  }
  // Many rounds of work-groups per CU (a shard of a large database): the 32 x 32 x 2 tiles -- half the LDS operand reads
  // per flop.  Measured at 64 x 125 000 x 4096: 794 us against 922 with the 16 x 16 x 4 tiles (BN = 128, K-step 32, three
  // work-groups per CU).  At 64 x 10 000 the launch is ONE round of work-groups and the tile that divides 10 000 rows
  // into 500 of them (BN = 80, split-K 4) wins: 82 us against 92 - 116 for every 32-wide plan.
  static const bool no_t32 = getenv("GLOC3D_MFMA_NO_T32") != nullptr;
  if (nq > 32 && !no_t32 && (long long)((n_range + 127) / 128) * ((nq + 63) / 64) >= 3 * 256) {
    best = MfmaPlan{4, 2, 1, 64, 128};
    best.t32 = 1;
  }
  // developer override: GLOC3D_MFMA_T32="NT,KS" -- the 32 x 32 x 2 tiles with NT tiles per wave and KS splits of K
  if (const char* e = getenv("GLOC3D_MFMA_T32")) {
    int nt = 0, ks = 0;
    if (sscanf(e, "%d,%d", &nt, &ks) == 2 && (nt == 1 || nt == 2) && ks > 0 && nq > 32) {
      best = MfmaPlan{4, nt, ks, 64, 64 * nt};
      best.t32 = 1;
    }
  }
  // developer override: GLOC3D_MFMA_PLAN="NT,KS"
  if (const char* e = getenv("GLOC3D_MFMA_PLAN")) {
    int nt = 0, ks = 0;
    if (sscanf(e, "%d,%d", &nt, &ks) == 2 && nt > 0 && ks > 0)
      best = MfmaPlan{WQ, nt, ks, BQ, 16 * nt * (4 / WQ)};
  }
  return best;
}

template <int WQ, int NT>
void launch_mfma_inst(gloc_knn* h, const MfmaPlan& p, const float* d_q, int nq, size_t first,
                      int n_range, size_t ld, size_t strideP) {
  dim3 grid((unsigned)((n_range + p.BN - 1) / p.BN), (unsigned)((nq + p.BQ - 1) / p.BQ),
            (unsigned)p.KS);
  const int kps = (((int)h->dim + p.KS - 1) / p.KS + 63) & ~63;
  static const int bk = getenv("GLOC3D_MFMA_BK") ? atoi(getenv("GLOC3D_MFMA_BK")) : 64;
  if (bk == 32 || kps < 128)
    hipLaunchKernelGGL((dist_mfma_kernel<WQ, NT, 8>), grid, dim3(256), 0, h->stream,
                       h->rows.as<float>(), d_q, h->dist.as<float>(), (int)h->dim, first, n_range,
                       nq, kps, ld, strideP);
  else
    hipLaunchKernelGGL((dist_mfma_kernel<WQ, NT, 16>), grid, dim3(256), 0, h->stream,
                       h->rows.as<float>(), d_q, h->dist.as<float>(), (int)h->dim, first, n_range,
                       nq, kps, ld, strideP);
}

// *finalized: the result (indices, distances) has been written through `fo` already -- no finalize launch
int run_mfma(gloc_knn* h, const float* d_q, int nq, int k, size_t first, int n_range,
             uint64_t* d_keys_out, const FinalOut& fo, bool* finalized, bool fp32_only) {
  *finalized = false;
  const MfmaPlan p = plan_mfma(nq, n_range, (int)h->dim, fp32_only);
  const int KC = std::max(h->candidates, std::min(64, k + 12));
  const size_t ld = ((size_t)n_range + 63) & ~(size_t)63;
  const size_t qpad = (size_t)((nq + p.BQ - 1) / p.BQ) * p.BQ;
  const size_t strideP = qpad * ld;
  GLOC_TRY(h->dist.ensure((size_t)p.KS * strideP * sizeof(float), h->stream));
  GLOC_TRY(h->qnorm.ensure((size_t)nq * sizeof(float), h->stream));
  GLOC_TRY(h->keys.ensure((size_t)nq * KC * sizeof(uint64_t), h->stream));
  GLOC_TRY(h->flags.ensure((size_t)nq * sizeof(int), h->stream));
  if (!h->n_incomplete.p) {
    GLOC_TRY(h->n_incomplete.ensure(sizeof(unsigned long long), h->stream));
    GLOC_HIP(hipMemsetAsync(h->n_incomplete.p, 0, sizeof(unsigned long long), h->stream));
  }
  bool use_bmin = false;
  int n_blocks = 0;
  if (p.b3) {
    // few work-groups: each splits its queries itself; many: once, ahead of the launch
    dim3 grid((unsigned)((n_range + p.BN - 1) / p.BN), (unsigned)((nq + p.BQ - 1) / p.BQ), (unsigned)p.KS);
    static const int qraw_env = getenv("GLOC3D_KNN_B3_QRAW") ? atoi(getenv("GLOC3D_KNN_B3_QRAW")) : -1;  // developer switch
    const bool qraw = qraw_env >= 0 ? qraw_env != 0 : (long long)grid.x * grid.y * grid.z <= 768;  // (64 x 125 000, 977 work-groups: 456 us split ahead, 461 in-kernel)
    const float* qsrc = d_q;
    if (!qraw) {
      ProfScope ps(h->prof, "split_queries", h->stream);
      GLOC_TRY(h->qsplit.ensure((size_t)nq * h->dim * sizeof(float), h->stream));
      const size_t n8 = (size_t)nq * h->dim / 8;
      hipLaunchKernelGGL(split_queries_kernel, dim3((unsigned)((n8 + 255) / 256)), dim3(256), 0, h->stream, d_q, n8,
                         h->qsplit.as<float>());
      qsrc = h->qsplit.as<float>();
    }
    ProfScope ps(h->prof, "dist_mfma", h->stream);
    const int kps3 = (((int)h->dim + p.KS - 1) / p.KS + 63) & ~63;
    static const bool no_mirror = getenv("GLOC3D_KNN_NO_MIRROR") != nullptr;  // developer switch: the row-major kernel
    if (!no_mirror && h->mirror.p) {
      // the rows from their tiled, pre-split mirror: contiguous 8-KB runs per tile and step (round 6)
      const dim3 tgrid((unsigned)((first % MIR_ROWS + (size_t)n_range + p.BN - 1) / p.BN), grid.y, grid.z);
      // a large window in one K-split: the epilogue leaves block minima for the selection (select_blocks_body)
      static const bool no_bmin = getenv("GLOC3D_KNN_NO_BLOCKMIN") != nullptr;  // developer switch: the slices
      n_blocks = (int)tgrid.x * (p.BN / 32);
      use_bmin = !no_bmin && n_range > SELQ_MAX_ROWS && p.KS == 1 && n_blocks <= SELQ_MAX_ROWS && KC <= SRR_KC &&
                 (int)h->dim <= 4 * SRR_G && !getenv("GLOC3D_KNN_NO_FUSED_RERANK");
      if (use_bmin) {
        GLOC_TRY(h->bmin.ensure((size_t)nq * n_blocks * sizeof(float), h->stream));
      }
#define B3T(NT_, QR_)                                                                                                  \
  do {                                                                                                                \
    constexpr int lds_bytes = b3_lds_bytes<NT_, 4>();                                                                 \
    static std::atomic<uint64_t> attr_set{0};                                                                         \
    const uint64_t dev_bit = 1ull << (h->device & 63);                                                                \
    if (lds_bytes > 48 * 1024 && !(attr_set.load(std::memory_order_relaxed) & dev_bit)) {                            \
      GLOC_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&dist_bf16x3_tiled_kernel<NT_, QR_>),                \
                                   hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes));                          \
      attr_set.fetch_or(dev_bit, std::memory_order_relaxed);                                                          \
    }                                                                                                                 \
    if (use_bmin) {                                                                                                   \
      static std::atomic<uint64_t> attr_set_b{0};                                                                     \
      if (lds_bytes > 48 * 1024 && !(attr_set_b.load(std::memory_order_relaxed) & dev_bit)) {                         \
        GLOC_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&dist_bf16x3_tiled_kernel<NT_, QR_, true>),        \
                                     hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes));                        \
        attr_set_b.fetch_or(dev_bit, std::memory_order_relaxed);                                                      \
      }                                                                                                               \
      hipLaunchKernelGGL((dist_bf16x3_tiled_kernel<NT_, QR_, true>), tgrid, dim3(256), lds_bytes, h->stream,          \
                         h->mirror.as<u32x4>(), qsrc, h->dist.as<float>(), (int)h->dim, first, n_range, nq, kps3, ld, \
                         strideP, h->norms.as<float>(), h->bmin.as<float>(), n_blocks);                               \
    } else {                                                                                                          \
      hipLaunchKernelGGL((dist_bf16x3_tiled_kernel<NT_, QR_>), tgrid, dim3(256), lds_bytes, h->stream,                \
                         h->mirror.as<u32x4>(), qsrc, h->dist.as<float>(), (int)h->dim, first, n_range, nq, kps3, ld, \
                         strideP, (const float*)nullptr, (float*)nullptr, 0);                                         \
    }                                                                                                                 \
  } while (0)
      if (p.NT == 1) { if (qraw) B3T(1, true); else B3T(1, false); }
      else { if (qraw) B3T(2, true); else B3T(2, false); }
#undef B3T
      GLOC_HIP(hipGetLastError());
    } else {
    static const int phase = getenv("GLOC3D_KNN_B3_PHASE") ? atoi(getenv("GLOC3D_KNN_B3_PHASE")) : 5;  // developer switch
#define B3(NT_, QR_, KO_)                                                                                             \
  do {                                                                                                                \
    constexpr int lds_bytes = b3_lds_bytes<NT_, KO_>();                                                               \
    static std::atomic<uint64_t> attr_set{0}; /* bit d: done on device d (the attribute belongs to the device's copy) */ \
    const uint64_t dev_bit = 1ull << (h->device & 63);                                                                \
    if (lds_bytes > 48 * 1024 && !(attr_set.load(std::memory_order_relaxed) & dev_bit)) {                            \
      GLOC_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&dist_bf16x3_kernel<NT_, QR_, KO_>),                 \
                                   hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes));                          \
      attr_set.fetch_or(dev_bit, std::memory_order_relaxed);                                                          \
    }                                                                                                                 \
    hipLaunchKernelGGL((dist_bf16x3_kernel<NT_, QR_, KO_>), grid, dim3(256), lds_bytes, h->stream,                    \
                       h->rows.as<float>(), qsrc, h->dist.as<float>(), (int)h->dim, first, n_range, nq, kps3, ld,     \
                       strideP, phase);                                                                               \
  } while (0)
#define B3Q(NT_, KO_)          \
  do {                         \
    if (qraw)                  \
      B3(NT_, true, KO_);      \
    else                       \
      B3(NT_, false, KO_);     \
  } while (0)
    // (steps of 64 k -- KO = 8, 256 contiguous bytes of a row per step -- measured no faster at either size: 444 / 479 us
    // against 456 / 448 at 64 x 125 000, and slower at 64 x 10 000; only the 32-k instances are built)
    if (p.NT == 1) B3Q(1, 4);
    else B3Q(2, 4);
#undef B3Q
#undef B3
    GLOC_HIP(hipGetLastError());
    }
  } else if (p.t32) {
    ProfScope ps(h->prof, "dist_mfma", h->stream);
    dim3 grid((unsigned)((n_range + p.BN - 1) / p.BN), (unsigned)((nq + p.BQ - 1) / p.BQ), (unsigned)p.KS);
    const int kps32 = (((int)h->dim + p.KS - 1) / p.KS + 63) & ~63;
    static const int bk32 = getenv("GLOC3D_MFMA_BK") ? atoi(getenv("GLOC3D_MFMA_BK")) : 32;
#define MF32(NT_, KQ_)                                                                                             \
  hipLaunchKernelGGL((dist_mfma32_kernel<NT_, KQ_>), grid, dim3(256), 0, h->stream, h->rows.as<float>(), d_q,      \
                     h->dist.as<float>(), (int)h->dim, first, n_range, nq, kps32, ld, strideP)
    if (bk32 == 32 || kps32 < 128) {
      if (p.NT == 1) MF32(1, 8); else MF32(2, 8);
    } else {
      if (p.NT == 1) MF32(1, 16); else MF32(2, 16);
    }
#undef MF32
    GLOC_HIP(hipGetLastError());
  } else {
    ProfScope ps(h->prof, "dist_mfma", h->stream);
#define MF(WQ_, NT_)                                                        \
  if (p.WQ == WQ_ && p.NT == NT_) {                                         \
    launch_mfma_inst<WQ_, NT_>(h, p, d_q, nq, first, n_range, ld, strideP); \
  } else
    MF(4, 2) MF(4, 3) MF(4, 4) MF(4, 5) MF(4, 6) MF(4, 8) MF(2, 2) MF(2, 4) MF(1, 1) MF(1, 2) {
      set_err("internal: no MFMA instance for WQ=%d NT=%d", p.WQ, p.NT);
      return GLOC_ERR_STATE;
    }
#undef MF
    GLOC_HIP(hipGetLastError());
  }
  // rounding bound of the coarse distance against the reference-order distance:
  //   reference chain          (D/4 + 4) u d2
  //   MFMA chains of <= 64 fma, nch partial sums, KS split sums, norms (D/64 + 6), 3 final ops
  const float u = 5.9604645e-8f;
  const int kps = (((int)h->dim + p.KS - 1) / p.KS + 63) & ~63;
  const float eps_rel_d = 1.05f * u * (float)(h->dim / 4 + 4);
  static const float eps_scale = getenv("GLOC3D_KNN_EPS_SCALE") ? (float)atof(getenv("GLOC3D_KNN_EPS_SCALE")) : 1.f;  // dev
  //   split-bf16 form: the dropped product terms 3.03 * 2^-16 = 776 u (knn_kernels.hpp), and its chains are 3 x 64
  //   products long with the accumulation inside an MFMA priced as truncating adds (2 u each): 384 for the 64
  const float chain_u = p.b3 ? 776.f + 384.f : 64.f;
  const float eps_rel_n = eps_scale * 1.05f * u * (chain_u + (float)((kps + 63) / 64 + p.KS + h->dim / 64 + 6 + 3 + 4));
  static const bool no_fused = getenv("GLOC3D_KNN_NO_FUSED_RERANK") != nullptr;  // developer switch: the three launches
  const bool large = n_range > SELQ_MAX_ROWS;  // slices first; an incomplete query is flagged for the host
  SlicePlan sl{1, 0};
  const bool fused = (!large || use_bmin || plan_slices(n_range, nq, KC, &sl)) && KC <= SRR_KC && (int)h->dim <= 4 * SRR_G && !no_fused;
  if (fused) {
    // select + re-rank + completeness check in one launch, one work-group per query
    if (large && use_bmin) {  // round 6: from the coarse kernel's block minima -- 32 x KC partial dots per query, not the window's:
      // inside the select + re-rank launch itself (select_blocks_body)
      GLOC_TRY(h->klists.ensure((size_t)nq * SELB_LIST * sizeof(uint64_t), h->stream));
    } else if (large) {
      ProfScope ps(h->prof, "select", h->stream);
      GLOC_TRY(launch_slices<1>(h, d_q, nq, KC, first, n_range, ld, strideP, p.KS, sl, nullptr));
    }
    ProfScope ps(h->prof, "select_rerank", h->stream);
#define SRR_ARGS                                                                                                          \
  dim3(nq), dim3(SELQ_THREADS), 0, h->stream, h->dist.as<float>(), ld, strideP, p.KS, d_q, (int)h->dim,                  \
      h->norms.as<float>(), first, n_range, KC, k, h->rows.as<float>(), h->dn_max.as<uint32_t>(), eps_rel_d, eps_rel_n, \
      h->qnorm.as<float>(), d_keys_out, h->flags.as<int>(), h->n_incomplete.as<unsigned long long>(), fo,                \
      h->dist.as<float>(), h->dev_trace.as<unsigned long long>()
    if (large)
      hipLaunchKernelGGL(select_rerank_kernel<true>, SRR_ARGS, h->klists.as<uint64_t>(), use_bmin ? SELB_LIST : sl.S * KC,
                         use_bmin ? h->bmin.as<float>() : (const float*)nullptr, n_blocks, h->klists.as<uint64_t>());
    else
      hipLaunchKernelGGL(select_rerank_kernel<false>, SRR_ARGS, (const uint64_t*)nullptr, 0, (const float*)nullptr, 0, (uint64_t*)nullptr);
#undef SRR_ARGS
    GLOC_HIP(hipGetLastError());
    // (incomplete queries -- rare -- are redone exactly by their own work-group inside the same launch: no
    // read-back, no host synchronisation, no further launch)
    h->stats.last_n_tile = (uint32_t)p.BN;
    h->stats.last_k_split = (uint32_t)p.KS;
    h->stats.last_candidates = (uint32_t)KC;
    if (!large) {
      *finalized = fo.idx != nullptr;
      return GLOC_OK;
    }
    // (large windows: on to the flagged-only exact pass below -- a work-group per query cannot redo a window of that size)
  } else {
  // (the query norms of the coarse form are made by the select kernel, which leaves them in h->qnorm)
  GLOC_TRY(run_select<1>(h, d_q, nq, KC, first, n_range, ld, strideP, p.KS, h->keys.as<uint64_t>()));
  {
    // rounding bound of the coarse distance against the reference-order distance:
    //   reference chain          (D/4 + 4) u d2
    //   MFMA chains of <= 64 fma, nch partial sums, KS split sums, norms (D/64 + 6), 3 final ops
    ProfScope ps(h->prof, "rerank", h->stream);
    GLOC_TRY(h->exact.ensure((size_t)nq * KC * sizeof(float), h->stream));
    hipLaunchKernelGGL(rerank_dist_kernel, dim3((KC + RR - 1) / RR, nq), dim3(64), 0, h->stream,
                       h->rows.as<float>(), d_q, (int)h->dim, h->keys.as<uint64_t>(), KC, k,
                       h->qnorm.as<float>(), h->dn_max.as<uint32_t>(), eps_rel_d, eps_rel_n,
                       h->exact.as<float>());
    hipLaunchKernelGGL(rerank_final_kernel, dim3(nq), dim3(64), 0, h->stream,
                       h->keys.as<uint64_t>(), h->exact.as<float>(), KC, k, n_range,
                       h->qnorm.as<float>(), h->dn_max.as<uint32_t>(), eps_rel_d, eps_rel_n,
                       d_keys_out, h->flags.as<int>(), h->n_incomplete.as<unsigned long long>());
    GLOC_HIP(hipGetLastError());
  }
  }
  h->stats.last_n_tile = (uint32_t)p.BN;
  h->stats.last_k_split = (uint32_t)p.KS;
  h->stats.last_candidates = (uint32_t)KC;
  SlicePlan fb_sl;
  if ((n_range <= SELECT_ONE_BLOCK_MAX && k <= 2048 / std::max(1, (n_range + 255) / 256)) ||
      (k <= 64 && plan_slices(n_range, nq, k, &fb_sl))) {
    // Incomplete queries are redone on the exact path ON THE DEVICE: the kernels are always
    // launched and leave at once unless the query's flag is set -- no read-back, no host synchronisation
    // (windows above 16 384 rows too since round 4: the read-back of the flags stalled the launches of a run of searches
    // behind a host synchronisation, ~35 us of a 460-us search over a 125 000-row shard).
    // (The coarse partial dots in h->dist are dead by now: the exact distances of the flagged queries
    // reuse the buffer, row q at q * ld.)
    static const bool no_redo1 = getenv("GLOC3D_KNN_NO_REDO1") != nullptr;  // developer switch: the three launches
    if (fused && large && k <= 64 && !no_redo1) {
      // ONE launch (round 6): every work-group walks the flags and leaves when none is set; a flagged query's exact
      // distances, slice selections and final selection happen inside it (flagged_redo_kernel)
      int S = std::max(1, (n_range + 2047) / 2048);
      while ((long long)S * k > SELQ_MAX_ROWS) S = (S + 1) / 2;
      int L = (((n_range + S - 1) / S) + 63) & ~63;
      S = (n_range + L - 1) / L;  // (no empty slice)
      if (L <= SELQ_MAX_ROWS) {
        const size_t before = h->redo_tickets.cap;
        GLOC_TRY(h->redo_tickets.ensure((size_t)nq * sizeof(unsigned int), h->stream));
        if (h->redo_tickets.cap != before) GLOC_HIP(hipMemsetAsync(h->redo_tickets.p, 0, h->redo_tickets.cap, h->stream));
        GLOC_TRY(h->klists.ensure((size_t)nq * S * k * sizeof(uint64_t), h->stream));
        ProfScope ps(h->prof, "dist_exact", h->stream);
        hipLaunchKernelGGL(flagged_redo_kernel, dim3((unsigned)S), dim3(SELQ_THREADS), 0, h->stream, h->rows.as<float>(), d_q,
                           (int)h->dim, first, n_range, L, k, h->dist.as<float>(), ld, h->klists.as<uint64_t>(),
                           h->redo_tickets.as<unsigned int>(), h->flags.as<int>(), nq, d_keys_out, fo);
        GLOC_HIP(hipGetLastError());
        *finalized = fo.idx != nullptr;
        return GLOC_OK;
      }
    }
    GLOC_TRY(launch_dist_exact(h, d_q, nq, first, n_range, ld, h->flags.as<int>()));
    if (fused && large) {  // the fused kernel has written the result through `fo`: the flagged queries' is replaced
      bool fin = false;
      GLOC_TRY(run_select<0>(h, d_q, nq, k, first, n_range, ld, 0, 1, d_keys_out, h->flags.as<int>(), fo, &fin));
      *finalized = fo.idx != nullptr;
    } else {
      GLOC_TRY(run_select<0>(h, d_q, nq, k, first, n_range, ld, 0, 1, d_keys_out, h->flags.as<int>()));
    }
    return GLOC_OK;
  }
  // windows too large for that (the slices' lists no longer fit one work-group): completeness flags -> host;
  // incomplete queries are redone on the exact path
  if (h->h_flags_cap < (size_t)nq) {
    if (h->h_flags) (void)hipHostFree(h->h_flags);
    h->h_flags = nullptr;
    GLOC_HIP(hipHostMalloc((void**)&h->h_flags, sizeof(int) * (size_t)nq * 2));
    h->h_flags_cap = (size_t)nq * 2;
  }
  GLOC_HIP(hipMemcpyAsync(h->h_flags, h->flags.p, sizeof(int) * (size_t)nq,
                          hipMemcpyDeviceToHost, h->stream));
  GLOC_HIP(hipStreamSynchronize(h->stream));
  bool redone = false;
  for (int q = 0; q < nq; ++q) {
    if (h->h_flags[q]) {
      GLOC_TRY(run_exact(h, d_q + (size_t)q * h->dim, 1, k, first, n_range,
                         d_keys_out + (size_t)q * k));
      redone = true;
    }
  }
  if (fused && !redone) *finalized = fo.idx != nullptr;  // the fused kernel's own result stands
  return GLOC_OK;
}

int search_device_impl(gloc_knn* h, const float* d_q, size_t nq, size_t k, size_t first_row,
                       size_t last_row, uint64_t index_offset, uint64_t* d_idx, float* d_d2,
                       uint64_t index_stride = 1) {
  GLOC_REQUIRE(h && d_q && d_idx && d_d2, GLOC_ERR_INVALID, "null argument");
  GLOC_REQUIRE(k >= 1 && k <= 256, GLOC_ERR_INVALID, "k = %zu outside [1,256]", k);
  GLOC_REQUIRE(nq >= 1 && nq <= (1u << 20), GLOC_ERR_INVALID, "nq = %zu outside [1,2^20]", nq);
  GLOC_HIP(hipSetDevice(h->device));
  view_sync(h);
  if (last_row > h->n) last_row = h->n;
  if (first_row > last_row) first_row = last_row;
  const size_t range = last_row - first_row;
  GLOC_REQUIRE(range < (1ull << 31), GLOC_ERR_INVALID, "row window too large");
  GLOC_TRY(h->keys2.ensure(nq * k * sizeof(uint64_t), h->stream));
  uint64_t* keys_out = h->keys2.as<uint64_t>();
  bool all_final = true;  // every block of queries left its result in d_idx / d_d2 itself
  if (range == 0) {
    all_final = false;
    GLOC_HIP(hipMemsetAsync(keys_out, 0xFF, nq * k * sizeof(uint64_t), h->stream));
  } else {
    int algo = h->algo;
    const bool mfma_ok = (h->dim % 4 == 0) && k <= 52 && range >= 64;
    if (algo == GLOC_KNN_ALGO_AUTO) algo = (nq > 8 && mfma_ok) ? GLOC_KNN_ALGO_MFMA : GLOC_KNN_ALGO_EXACT;
    if ((algo == GLOC_KNN_ALGO_MFMA || algo == GLOC_KNN_ALGO_MFMA_FP32) && !mfma_ok) algo = GLOC_KNN_ALGO_EXACT;
    // process queries in blocks that bound the distance workspace (<= 2 GiB)
    const size_t ld = (range + 63) & ~(size_t)63;
    size_t qblk = (size_t)(2ull << 30) / (ld * sizeof(float) * 4);
    qblk = std::min<size_t>(1024, std::max<size_t>(64, qblk / 64 * 64));
    // the coarse form of this search (see h_inc): look at the last tracked search's count if its copy has landed
    static const bool no_adapt = getenv("GLOC3D_KNN_NO_ADAPT") != nullptr;  // developer switch
    bool tracked = false;
    if (algo == GLOC_KNN_ALGO_MFMA && !no_adapt) {
      if (h->inc_pending && hipEventQuery(h->inc_ev) == hipSuccess) {
        const unsigned long long now = *h->h_inc, delta = now - h->inc_seen;
        h->inc_seen = now;
        h->inc_pending = false;
        if (delta * 4 > h->inc_queries) h->coarse_fp32_left = 64;
      }
      if (h->coarse_fp32_left > 0) {
        h->coarse_fp32_left--;
        algo = GLOC_KNN_ALGO_MFMA_FP32;
      } else {
        tracked = !h->inc_pending;
      }
    }
    for (size_t q0 = 0; q0 < nq; q0 += qblk) {
      const int cnt = (int)std::min(qblk, nq - q0);
      if (algo == GLOC_KNN_ALGO_MFMA || algo == GLOC_KNN_ALGO_MFMA_FP32) {
        h->stats.searches_mfma++;
        bool fin = false;
        const FinalOut fo{d_idx + q0 * k, d_d2 + q0 * k, index_offset, index_stride};
        GLOC_TRY(run_mfma(h, d_q + q0 * h->dim, cnt, (int)k, first_row, (int)range, keys_out + q0 * k, fo, &fin,
                          algo == GLOC_KNN_ALGO_MFMA_FP32));
        all_final = all_final && fin;
      } else {
        h->stats.searches_exact++;
        bool fin = false;
        const FinalOut fo{d_idx + q0 * k, d_d2 + q0 * k, index_offset, index_stride};
        GLOC_TRY(run_exact(h, d_q + q0 * h->dim, cnt, (int)k, first_row, (int)range, keys_out + q0 * k, fo, &fin));
        all_final = all_final && fin;
      }
    }
    if (tracked && h->n_incomplete.p) {  // the fallback count as of this search, for the next one to look at
      if (!h->h_inc) {
        GLOC_HIP(hipHostMalloc((void**)&h->h_inc, sizeof(unsigned long long)));
        *h->h_inc = 0;
        GLOC_HIP(hipEventCreateWithFlags(&h->inc_ev, hipEventDisableTiming));
      }
      GLOC_HIP(hipMemcpyAsync(h->h_inc, h->n_incomplete.p, sizeof(unsigned long long), hipMemcpyDeviceToHost, h->stream));
      GLOC_HIP(hipEventRecord(h->inc_ev, h->stream));
      h->inc_pending = true;
      h->inc_queries = nq;
    }
  }
  h->stats.queries_total += nq;
  if (!all_final) {
    ProfScope ps(h->prof, "finalize", h->stream);
    const size_t total = nq * k;
    hipLaunchKernelGGL(finalize_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0,
                       h->stream, keys_out, total, index_offset, index_stride, d_idx, d_d2);
    GLOC_HIP(hipGetLastError());
  }
  return GLOC_OK;
}

}  // namespace

extern "C" {

int gloc_knn_create(int device, size_t dim, gloc_knn** out) {
  GLOC_REQUIRE(out, GLOC_ERR_INVALID, "out is null");
  *out = nullptr;
  GLOC_REQUIRE(dim >= 1 && dim <= (1u << 20), GLOC_ERR_INVALID, "dim = %zu outside [1,2^20]", dim);
  GLOC_TRY(select_device(device));
  gloc_knn* h = new (std::nothrow) gloc_knn;
  GLOC_REQUIRE(h, GLOC_ERR_NOMEM, "host allocation failed");
  h->device = device;
  h->dim = dim;
  hipError_t e = hipStreamCreateWithFlags(&h->own_stream, hipStreamNonBlocking);
  if (e != hipSuccess) {
    set_err("hipStreamCreate failed: %s", hipGetErrorString(e));
    delete h;
    return GLOC_ERR_HIP;
  }
  h->stream = h->own_stream;
  *out = h;
  return GLOC_OK;
}

int gloc_knn_create_view(gloc_knn* parent, gloc_knn** out) {
  GLOC_REQUIRE(parent && out, GLOC_ERR_INVALID, "null argument");
  *out = nullptr;
  GLOC_REQUIRE(!parent->parent, GLOC_ERR_INVALID, "a view of a view: take it of the owning handle");
  gloc_knn* v = nullptr;
  GLOC_TRY(gloc_knn_create(parent->device, parent->dim, &v));
  v->parent = parent;
  v->algo = parent->algo;
  v->candidates = parent->candidates;
  parent->views++;
  view_sync(v);
  *out = v;
  return GLOC_OK;
}

int gloc_knn_destroy(gloc_knn* h) {
  if (!h) return GLOC_OK;
  GLOC_REQUIRE(h->views == 0, GLOC_ERR_STATE, "%d view(s) of this handle are still alive (gloc_knn_create_view): destroy them first", h->views);
  (void)hipSetDevice(h->device);
  (void)hipStreamSynchronize(h->stream);
  h->prof.destroy();
  if (h->parent) {  // the rows are the parent's
    h->rows = DevBuf{};
    h->norms = DevBuf{};
    h->mirror = DevBuf{};
    h->dn_max = DevBuf{};
    h->parent->views--;
  }
  h->rows.release();
  h->norms.release();
  h->mirror.release();
  h->dn_max.release();
  h->dist.release();
  h->keys.release();
  h->n_incomplete.release();
  h->shard_ws.release();
  h->klists.release();
  h->klists2.release();
  h->exact.release();
  h->keys2.release();
  h->qnorm.release();
  h->qsplit.release();
  h->dev_trace.release();
  h->flags.release();
  h->redo_tickets.release();
  h->bmin.release();
  h->stage_q.release();
  h->stage_idx.release();
  h->stage_d2.release();
  if (h->h_flags) (void)hipHostFree(h->h_flags);
  if (h->h_inc) (void)hipHostFree(h->h_inc);
  if (h->inc_ev) (void)hipEventDestroy(h->inc_ev);
  if (h->own_stream) (void)hipStreamDestroy(h->own_stream);
  delete h;
  return GLOC_OK;
}

int gloc_knn_set_stream(gloc_knn* h, void* hip_stream) {
  GLOC_REQUIRE(h, GLOC_ERR_INVALID, "null handle");
  GLOC_HIP(hipSetDevice(h->device));
  GLOC_HIP(hipStreamSynchronize(h->stream));
  h->stream = hip_stream ? (hipStream_t)hip_stream : h->own_stream;
  return GLOC_OK;
}

// Developer aid (not part of include/gloc3d.h): phase stamps [nq][16] of the LAST fused select + re-rank launch.
int gloc_knn_debug_trace(gloc_knn* h, int enable_nq, unsigned long long* out, size_t n_words) {
  GLOC_REQUIRE(h, GLOC_ERR_INVALID, "null handle");
  GLOC_HIP(hipSetDevice(h->device));
  if (out && h->dev_trace.p) {
    GLOC_HIP(hipMemcpyAsync(out, h->dev_trace.p, n_words * 8, hipMemcpyDeviceToHost, h->stream));
    GLOC_HIP(hipStreamSynchronize(h->stream));
  }
  if (enable_nq > 0) {
    GLOC_TRY(h->dev_trace.ensure((size_t)enable_nq * 128, h->stream));
  } else if (enable_nq < 0) {
    GLOC_HIP(hipStreamSynchronize(h->stream));
    h->dev_trace.release();
  }
  return GLOC_OK;
}

int gloc_knn_synchronize(gloc_knn* h) {
  GLOC_REQUIRE(h, GLOC_ERR_INVALID, "null handle");
  GLOC_HIP(hipSetDevice(h->device));
  GLOC_HIP(hipStreamSynchronize(h->stream));
  return GLOC_OK;
}

int gloc_knn_set_option(gloc_knn* h, int option, int64_t value) {
  GLOC_REQUIRE(h, GLOC_ERR_INVALID, "null handle");
  switch (option) {
    case GLOC_KNN_OPT_ALGO:
      GLOC_REQUIRE(value >= 0 && value <= 3, GLOC_ERR_INVALID, "bad algorithm %lld", (long long)value);
      h->algo = (int)value;
      return GLOC_OK;
    case GLOC_KNN_OPT_CANDIDATES:
      GLOC_REQUIRE(value >= 1 && value <= 64, GLOC_ERR_INVALID, "candidates %lld outside [1,64]",
                   (long long)value);
      h->candidates = (int)value;
      return GLOC_OK;
    case GLOC_KNN_OPT_PROFILE:
      h->prof.enabled = value != 0;
      return GLOC_OK;
    default:
      set_err("unknown option %d", option);
      return GLOC_ERR_INVALID;
  }
}

int gloc_knn_reserve(gloc_knn* h, size_t n_rows) {
  GLOC_REQUIRE(h, GLOC_ERR_INVALID, "null handle");
  GLOC_HIP(hipSetDevice(h->device));
  return ensure_rows(h, n_rows);
}

int gloc_knn_add(gloc_knn* h, const float* rows, size_t n) {
  GLOC_REQUIRE(h && (rows || n == 0), GLOC_ERR_INVALID, "null argument");
  if (n == 0) return GLOC_OK;
  GLOC_REQUIRE(h->n + n < (1ull << 32) - 1, GLOC_ERR_INVALID, "database limited to 2^32-2 rows");
  GLOC_HIP(hipSetDevice(h->device));
  GLOC_TRY(ensure_rows(h, h->n + n));
  GLOC_HIP(hipMemcpyAsync(h->rows.as<float>() + h->n * h->dim, rows, n * h->dim * sizeof(float),
                          hipMemcpyHostToDevice, h->stream));
  GLOC_TRY(update_norms(h, h->n, n));
  GLOC_HIP(hipStreamSynchronize(h->stream));  // the caller may free `rows` on return
  h->n += n;
  return GLOC_OK;
}

int gloc_knn_add_device(gloc_knn* h, const float* d_rows, size_t n) {
  GLOC_REQUIRE(h && (d_rows || n == 0), GLOC_ERR_INVALID, "null argument");
  if (n == 0) return GLOC_OK;
  GLOC_REQUIRE(h->n + n < (1ull << 32) - 1, GLOC_ERR_INVALID, "database limited to 2^32-2 rows");
  GLOC_HIP(hipSetDevice(h->device));
  GLOC_TRY(ensure_rows(h, h->n + n));
  GLOC_HIP(hipMemcpyAsync(h->rows.as<float>() + h->n * h->dim, d_rows,
                          n * h->dim * sizeof(float), hipMemcpyDeviceToDevice, h->stream));
  GLOC_TRY(update_norms(h, h->n, n));
  h->n += n;
  return GLOC_OK;
}

int gloc_knn_add_synthetic(gloc_knn* h, int kind, uint64_t seed, uint64_t first_row, size_t n,
                           uint64_t row_stride) {
  GLOC_REQUIRE(h, GLOC_ERR_INVALID, "null handle");
  GLOC_REQUIRE(kind == 0 || kind == 1, GLOC_ERR_INVALID, "kind must be 0 (iid) or 1 (trajectory)");
  if (n == 0) return GLOC_OK;
  GLOC_REQUIRE(h->n + n < (1ull << 32) - 1, GLOC_ERR_INVALID, "database limited to 2^32-2 rows");
  GLOC_HIP(hipSetDevice(h->device));
  GLOC_TRY(ensure_rows(h, h->n + n));
  gloc::synth::launch_fill(h->stream, kind, seed, first_row, n, h->dim, row_stride ? row_stride : 1,
                           h->rows.as<float>() + h->n * h->dim);
  GLOC_HIP(hipGetLastError());
  GLOC_TRY(update_norms(h, h->n, n));
  h->n += n;
  return GLOC_OK;
}

int gloc_synth_fill_device(int device, void* hip_stream, int kind, uint64_t seed,
                           uint64_t first_row, size_t n, size_t dim, uint64_t row_stride,
                           float* d_out) {
  GLOC_REQUIRE(d_out || n == 0, GLOC_ERR_INVALID, "null output");
  GLOC_REQUIRE(kind == 0 || kind == 1, GLOC_ERR_INVALID, "kind must be 0 (iid) or 1 (trajectory)");
  GLOC_TRY(select_device(device));
  if (n == 0) return GLOC_OK;
  gloc::synth::launch_fill((hipStream_t)hip_stream, kind, seed, first_row, n, dim,
                           row_stride ? row_stride : 1, d_out);
  GLOC_HIP(hipGetLastError());
  return GLOC_OK;
}

int gloc_knn_clear(gloc_knn* h) {
  GLOC_REQUIRE(h, GLOC_ERR_INVALID, "null handle");
  GLOC_NOT_VIEW(h);
  GLOC_HIP(hipSetDevice(h->device));
  GLOC_HIP(hipStreamSynchronize(h->stream));
  h->n = 0;
  if (h->dn_max.p) GLOC_HIP(hipMemsetAsync(h->dn_max.p, 0, sizeof(uint32_t), h->stream));
  return GLOC_OK;
}

int gloc_knn_size(const gloc_knn* h, size_t* n_rows) {
  GLOC_REQUIRE(h && n_rows, GLOC_ERR_INVALID, "null argument");
  *n_rows = h->parent ? h->parent->n : h->n;
  return GLOC_OK;
}

int gloc_knn_dim(const gloc_knn* h, size_t* dim) {
  GLOC_REQUIRE(h && dim, GLOC_ERR_INVALID, "null argument");
  *dim = h->dim;
  return GLOC_OK;
}

int gloc_knn_device_rows(const gloc_knn* h, const float** d_rows) {
  GLOC_REQUIRE(h && d_rows, GLOC_ERR_INVALID, "null argument");
  *d_rows = h->rows.as<float>();
  return GLOC_OK;
}

int gloc_knn_save(gloc_knn* h, const char* path) {
  GLOC_REQUIRE(h && path, GLOC_ERR_INVALID, "null argument");
  GLOC_HIP(hipSetDevice(h->device));
  FILE* f = fopen(path, "wb");
  GLOC_REQUIRE(f, GLOC_ERR_INVALID, "cannot open %s for writing", path);
  const uint32_t hdr[2] = {(uint32_t)h->n, (uint32_t)h->dim};
  bool ok = fwrite("GLOCDESC", 1, 8, f) == 8 && fwrite(hdr, 4, 2, f) == 2;
  std::vector<float> buf;
  const size_t chunk = std::max<size_t>(1, (64u << 20) / (h->dim * sizeof(float)));  // 64 MiB pieces
  for (size_t r = 0; ok && r < h->n; r += chunk) {
    const size_t cnt = std::min(chunk, h->n - r);
    buf.resize(cnt * h->dim);
    if (hipMemcpyAsync(buf.data(), h->rows.as<float>() + r * h->dim, buf.size() * sizeof(float),
                       hipMemcpyDeviceToHost, h->stream) != hipSuccess ||
        hipStreamSynchronize(h->stream) != hipSuccess)
      ok = false;
    else
      ok = fwrite(buf.data(), sizeof(float), buf.size(), f) == buf.size();
  }
  ok = (fclose(f) == 0) && ok;
  GLOC_REQUIRE(ok, GLOC_ERR_STATE, "writing %s failed", path);
  return GLOC_OK;
}

int gloc_knn_load(gloc_knn* h, const char* path) {
  GLOC_REQUIRE(h && path, GLOC_ERR_INVALID, "null argument");
  FILE* f = fopen(path, "rb");
  GLOC_REQUIRE(f, GLOC_ERR_INVALID, "cannot open %s", path);
  char magic[8];
  uint32_t hdr[2] = {0, 0};
  if (fread(magic, 1, 8, f) != 8 || memcmp(magic, "GLOCDESC", 8) != 0 || fread(hdr, 4, 2, f) != 2) {
    fclose(f);
    set_err("%s is not a GLOCDESC file", path);
    return GLOC_ERR_INVALID;
  }
  if (hdr[1] != h->dim) {
    fclose(f);
    set_err("%s holds %u-D rows, the index is %zu-D", path, hdr[1], h->dim);
    return GLOC_ERR_INVALID;
  }
  // the header must agree with the file's size before a single row is added
  const long pos = ftell(f);
  if (pos < 0 || fseek(f, 0, SEEK_END) != 0) {
    fclose(f);
    set_err("cannot size %s", path);
    return GLOC_ERR_INVALID;
  }
  const long end = ftell(f);
  if (end < 0 || (unsigned long long)(end - pos) < (unsigned long long)hdr[0] * h->dim * sizeof(float)) {
    fclose(f);
    set_err("%s is truncated: the header announces %u rows", path, hdr[0]);
    return GLOC_ERR_INVALID;
  }
  (void)fseek(f, pos, SEEK_SET);
  std::vector<float> buf;
  const size_t chunk = std::max<size_t>(1, (64u << 20) / (h->dim * sizeof(float)));
  const size_t n0 = h->n;  // on any failure below the index is rolled back to this many rows
  int rc = GLOC_OK;
  for (size_t r = 0; rc == GLOC_OK && r < hdr[0]; r += chunk) {
    const size_t cnt = std::min<size_t>(chunk, hdr[0] - r);
    buf.resize(cnt * h->dim);
    if (fread(buf.data(), sizeof(float), buf.size(), f) != buf.size()) {
      set_err("%s is truncated", path);
      rc = GLOC_ERR_INVALID;
    } else {
      rc = gloc_knn_add(h, buf.data(), cnt);
    }
  }
  fclose(f);
  if (rc != GLOC_OK) h->n = n0;  // (the running maximum norm may stay larger: it only widens the re-rank window)
  return rc;
}

int gloc_knn_search_device(gloc_knn* h, const float* d_queries, size_t nq, size_t k,
                           size_t first_row, size_t last_row, uint64_t index_offset,
                           uint64_t* d_out_idx, float* d_out_d2) {
  GLOC_REQUIRE(h, GLOC_ERR_INVALID, "null handle");
  return search_device_impl(h, d_queries, nq, k, first_row, last_row, index_offset, d_out_idx,
                            d_out_d2);
}

int gloc_knn_search_sharded(gloc_knn* h, gloc_comm* comm, const float* d_queries, size_t nq, size_t k,
                            uint64_t index_stride, uint64_t index_offset, uint64_t* d_out_idx,
                            float* d_out_d2) {
  GLOC_REQUIRE(h && comm && d_queries && d_out_idx && d_out_d2, GLOC_ERR_INVALID, "null argument");
  GLOC_REQUIRE(comm->device == h->device, GLOC_ERR_INVALID, "communicator on device %d, index on %d", comm->device,
               h->device);
  GLOC_REQUIRE(index_stride >= 1, GLOC_ERR_INVALID, "index_stride must be >= 1");
  GLOC_REQUIRE((size_t)comm->world * k <= 1024, GLOC_ERR_INVALID, "shards x k = %zu exceeds 1024",
               (size_t)comm->world * k);
  const size_t G = (size_t)comm->world, cnt = nq * k;
  // [local idx | local d2 | gathered idx | gathered d2]
  GLOC_TRY(h->shard_ws.ensure(cnt * 12 + G * cnt * 12 + 64, h->stream));
  uint64_t* li = h->shard_ws.as<uint64_t>();
  uint64_t* gi = li + cnt;
  float* ld = reinterpret_cast<float*>(gi + G * cnt);
  float* gd = ld + cnt;
  // this shard's top-k with GLOBAL row indices ...  (profile families: shard_local / shard_gather / shard_merge)
  {
    ProfScope ps(h->prof, "shard_local", h->stream);
    GLOC_TRY(search_device_impl(h, d_queries, nq, k, 0, (size_t)-1, index_offset, li, ld, index_stride));
  }
  if (G == 1) {
    GLOC_HIP(hipMemcpyAsync(d_out_idx, li, cnt * sizeof(uint64_t), hipMemcpyDeviceToDevice, h->stream));
    GLOC_HIP(hipMemcpyAsync(d_out_d2, ld, cnt * sizeof(float), hipMemcpyDeviceToDevice, h->stream));
    return GLOC_OK;
  }
  // ... all-gathered over xGMI: one fused launch for the two small arrays ([shard][nq][k], what K3 reads) ...
  {
    ProfScope ps(h->prof, "shard_gather", h->stream);
    GLOC_TRY(gloc::comm::group_begin());
    int rc = gloc::comm::all_gather(comm, li, gi, cnt * sizeof(uint64_t), h->stream);
    if (rc == GLOC_OK) rc = gloc::comm::all_gather(comm, ld, gd, cnt * sizeof(float), h->stream);
    const int rc2 = gloc::comm::group_end();
    GLOC_TRY(rc);
    GLOC_TRY(rc2);
  }
  // ... and merged on every rank in the same (d2, idx) order: a replicated result, equal to the one-GPU search
  ProfScope ps(h->prof, "shard_merge", h->stream);
  hipLaunchKernelGGL(merge_kernel, dim3((unsigned)nq), dim3(64), 0, h->stream, gi, gd, (int)G, (int)nq, (int)k,
                     d_out_idx, d_out_d2);
  GLOC_HIP(hipGetLastError());
  return GLOC_OK;
}

int gloc_knn_search_sharded_host(gloc_knn* h, gloc_comm* comm, const float* queries, size_t nq, size_t k,
                                 uint64_t index_stride, uint64_t index_offset, uint64_t* out_idx, float* out_d2) {
  GLOC_REQUIRE(h && queries && out_idx && out_d2, GLOC_ERR_INVALID, "null argument");
  GLOC_REQUIRE(k >= 1 && k <= 256 && nq >= 1 && nq <= (1u << 20), GLOC_ERR_INVALID, "bad nq / k");
  GLOC_NOT_VIEW(h);
  GLOC_HIP(hipSetDevice(h->device));
  GLOC_TRY(h->stage_q.ensure(nq * h->dim * sizeof(float), h->stream));
  GLOC_TRY(h->stage_idx.ensure(nq * k * sizeof(uint64_t), h->stream));
  GLOC_TRY(h->stage_d2.ensure(nq * k * sizeof(float), h->stream));
  GLOC_HIP(hipMemcpyAsync(h->stage_q.p, queries, nq * h->dim * sizeof(float), hipMemcpyHostToDevice, h->stream));
  GLOC_TRY(gloc_knn_search_sharded(h, comm, h->stage_q.as<float>(), nq, k, index_stride, index_offset,
                                   h->stage_idx.as<uint64_t>(), h->stage_d2.as<float>()));
  GLOC_HIP(hipMemcpyAsync(out_idx, h->stage_idx.p, nq * k * sizeof(uint64_t), hipMemcpyDeviceToHost, h->stream));
  GLOC_HIP(hipMemcpyAsync(out_d2, h->stage_d2.p, nq * k * sizeof(float), hipMemcpyDeviceToHost, h->stream));
  GLOC_HIP(hipStreamSynchronize(h->stream));
  return GLOC_OK;
}

int gloc_knn_search(gloc_knn* h, const float* queries, size_t nq, size_t k, size_t first_row,
                    size_t last_row, uint64_t* out_idx, float* out_d2) {
  GLOC_REQUIRE(h && queries && out_idx && out_d2, GLOC_ERR_INVALID, "null argument");
  GLOC_REQUIRE(k >= 1 && k <= 256, GLOC_ERR_INVALID, "k = %zu outside [1,256]", k);
  GLOC_REQUIRE(nq >= 1 && nq <= (1u << 20), GLOC_ERR_INVALID, "nq = %zu outside [1,2^20]", nq);
  GLOC_HIP(hipSetDevice(h->device));
  GLOC_TRY(h->stage_q.ensure(nq * h->dim * sizeof(float), h->stream));
  GLOC_TRY(h->stage_idx.ensure(nq * k * sizeof(uint64_t), h->stream));
  GLOC_TRY(h->stage_d2.ensure(nq * k * sizeof(float), h->stream));
  GLOC_HIP(hipMemcpyAsync(h->stage_q.p, queries, nq * h->dim * sizeof(float),
                          hipMemcpyHostToDevice, h->stream));
  GLOC_TRY(search_device_impl(h, h->stage_q.as<float>(), nq, k, first_row, last_row, 0,
                              h->stage_idx.as<uint64_t>(), h->stage_d2.as<float>()));
  GLOC_HIP(hipMemcpyAsync(out_idx, h->stage_idx.p, nq * k * sizeof(uint64_t),
                          hipMemcpyDeviceToHost, h->stream));
  GLOC_HIP(hipMemcpyAsync(out_d2, h->stage_d2.p, nq * k * sizeof(float), hipMemcpyDeviceToHost,
                          h->stream));
  GLOC_HIP(hipStreamSynchronize(h->stream));
  return GLOC_OK;
}

int gloc_topk_merge_device(int device, void* hip_stream, const uint64_t* d_idx, const float* d_d2,
                           size_t n_lists, size_t nq, size_t k, uint64_t* d_out_idx,
                           float* d_out_d2) {
  GLOC_REQUIRE(d_idx && d_d2 && d_out_idx && d_out_d2, GLOC_ERR_INVALID, "null argument");
  GLOC_REQUIRE(n_lists >= 1 && k >= 1 && n_lists * k <= 1024, GLOC_ERR_INVALID,
               "n_lists * k = %zu outside [1,1024]", n_lists * k);
  GLOC_REQUIRE(nq >= 1 && nq <= (1u << 20), GLOC_ERR_INVALID, "nq outside [1,2^20]");
  GLOC_TRY(select_device(device));
  hipLaunchKernelGGL(merge_kernel, dim3((unsigned)nq), dim3(64), 0, (hipStream_t)hip_stream, d_idx,
                     d_d2, (int)n_lists, (int)nq, (int)k, d_out_idx, d_out_d2);
  GLOC_HIP(hipGetLastError());
  return GLOC_OK;
}

int gloc_knn_get_stats(const gloc_knn* h, gloc_knn_stats* out) {
  GLOC_REQUIRE(h && out, GLOC_ERR_INVALID, "null argument");
  *out = h->stats;
  if (h->n_incomplete.p) {  // the fallback count lives on the device (no read-back on the search path)
    unsigned long long c = 0;
    GLOC_HIP(hipSetDevice(h->device));
    GLOC_HIP(hipStreamSynchronize(h->stream));
    GLOC_HIP(hipMemcpy(&c, h->n_incomplete.p, sizeof(c), hipMemcpyDeviceToHost));
    out->queries_fallback = c;
  }
  return GLOC_OK;
}

int gloc_knn_profile(gloc_knn* h, const char* kernel, double* total_ms, uint64_t* launches) {
  GLOC_REQUIRE(h && kernel, GLOC_ERR_INVALID, "null argument");
  GLOC_HIP(hipSetDevice(h->device));
  GLOC_TRY(h->prof.collect(h->stream));
  auto it = h->prof.fam.find(kernel);
  if (total_ms) *total_ms = it == h->prof.fam.end() ? 0.0 : it->second.total_ms;
  if (launches) *launches = it == h->prof.fam.end() ? 0 : it->second.launches;
  return GLOC_OK;
}

int gloc_knn_profile_reset(gloc_knn* h) {
  GLOC_REQUIRE(h, GLOC_ERR_INVALID, "null handle");
  GLOC_HIP(hipSetDevice(h->device));
  GLOC_HIP(hipStreamSynchronize(h->stream));
  h->prof.reset();
  return GLOC_OK;
}

}  // extern "C"
