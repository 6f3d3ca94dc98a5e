// reg.hip -- C ABI of the batched candidate registration (include/gloc3d.h) over reg_kernels.hpp.
// Replaces icp_match_3d (registration/global_registration.cpp:237-248) and the per-candidate RANSAC
// transform estimate (registration/loop_detector.cpp:256-257) of the reference, batched over the
// top-k candidates of GlocEvaluator::global_registraion (registration/global_localization.cpp:511-574).
#include <algorithm>
#include <cmath>
#include <new>
#include <vector>

#include <hipcub/hipcub.hpp>

#include "common.hpp"
#include "nn_culled.hpp"
#include "reg_kernels.hpp"

using namespace gloc;
using namespace gloc::reg;

// A scan resident in HBM: original-order xyz plus its search index (Morton-sorted copy with the
// original indices, chunk boxes, sorted keys, inverse permutation).  One allocation per scan.
struct DevScan {
  void* block = nullptr;
  float* xyz = nullptr;
  size_t n = 0;
  ScanIndexDev idx{};
  uint32_t* order = nullptr;  // source groups of 64 * order_cs sorted points, widest first (launch order)
  mutable int order_cs = 0;   // 0: not built yet
};

struct gloc_reg {
  int device = 0;
  hipStream_t own_stream = nullptr, stream = nullptr;
  std::vector<DevScan> scans;  // resident scan store
  DevBuf cands, states;        // CandDesc[], CandState[]
  DevBuf corr, d2, pairs;      // [cand][ld]
  DevBuf Rt, valid, inliers;   // RANSAC hypotheses
  DevBuf partials;
  DevBuf ccands;                          // CulledCand[]
  DevBuf sort_tmp, sort_keys, sort_vals, sort_perm;  // scan indexing scratch
  DevBuf counters;                        // [0] = pairs evaluated by nn_culled_kernel
  std::vector<CandState> h_states;
  int nn_mode = 0;                        // 0 culled + compacted (default), 1 exhaustive, 2 culled + broadcast
  bool trace_on = false;                  // dev only: per-wave trace of the culled kernel
  DevBuf trace;
  size_t trace_waves = 0;
  int nn_src_per_lane = 2;                // culled kernel: source points per lane (1, 2, 4)
  float nn_heavy_frac = 1.0f;             // share of a candidate's work-groups launched candidate-fastest, widest first
  uint64_t nn_launches = 0;
  Profiler prof;
};

namespace {

int upload_packed(gloc_reg* h, const float* pts, size_t n, size_t stride, float* d_dst) {
  if (n == 0) return GLOC_OK;
  if (stride == 3) {
    GLOC_HIP(hipMemcpyAsync(d_dst, pts, n * 3 * sizeof(float), hipMemcpyHostToDevice, h->stream));
  } else {
    GLOC_HIP(hipMemcpy2DAsync(d_dst, 3 * sizeof(float), pts, stride * sizeof(float),
                              3 * sizeof(float), n, hipMemcpyHostToDevice, h->stream));
  }
  return GLOC_OK;
}

void free_scan(DevScan& s) {
  if (s.block) (void)hipFree(s.block);
  s = DevScan{};
}

// Allocate a scan, upload its points and build its search index (Morton sort + chunk boxes).
int make_scan(gloc_reg* h, const float* pts, size_t n, size_t stride, DevScan* out) {
  DevScan s;
  s.n = n;
  const size_t nch = (n + CH - 1) / CH;
  const size_t nsb = (n + SB - 1) / SB;
  const size_t n1 = std::max<size_t>(n, 1), c1 = std::max<size_t>(nch, 1), b1 = std::max<size_t>(nsb, 1);
  // layout: pts4 | box_lo | box_hi | sb_lo | sb_hi | sup_lo | sup_hi | xyz | keys | inv | order
  // (16-byte aligned parts first)
  const size_t g1 = (n1 + 63) / 64;
  const size_t nsup = (nch + 63) / 64, u1 = std::max<size_t>(nsup, 1);
  const size_t bytes = sizeof(f32x4) * (n1 + 2 * c1 + 2 * b1 + 2 * u1) + sizeof(float) * 3 * n1 +
                       sizeof(uint32_t) * (2 * n1 + g1);
  GLOC_HIP(hipMalloc(&s.block, bytes));
  f32x4* p4 = reinterpret_cast<f32x4*>(s.block);
  f32x4* lo = p4 + n1;
  f32x4* hi = lo + c1;
  f32x4* slo = hi + c1;
  f32x4* shi = slo + b1;
  f32x4* ulo = shi + b1;
  f32x4* uhi = ulo + u1;
  s.xyz = reinterpret_cast<float*>(uhi + u1);
  uint32_t* keys = reinterpret_cast<uint32_t*>(s.xyz + 3 * n1);
  uint32_t* inv = keys + n1;
  s.order = inv + n1;
  int rc = upload_packed(h, pts, n, stride, s.xyz);
  if (rc != GLOC_OK) {
    free_scan(s);
    return rc;
  }
  // bounding box on the host (the caller's buffer is at hand) -> Morton grid of 1024^3 cells
  float mn[3] = {0, 0, 0}, mx[3] = {0, 0, 0};
  for (size_t i = 0; i < n; ++i)
    for (int a = 0; a < 3; ++a) {
      const float v = pts[i * stride + a];
      if (i == 0 || v < mn[a]) mn[a] = v;
      if (i == 0 || v > mx[a]) mx[a] = v;
    }
  const float ext = std::max(std::max(mx[0] - mn[0], mx[1] - mn[1]), mx[2] - mn[2]);
  const float cell = std::max(0.25f, ext / 1023.0f);
  s.idx = ScanIndexDev{p4, lo, hi, slo, shi, keys, inv, (uint32_t)n, (uint32_t)nch,
                       mn[0], mn[1], mn[2], 1.0f / cell, ulo, uhi, (uint32_t)nsup};
  if (n) {
    hipStream_t st = h->stream;
    auto fail = [&](int code) { free_scan(s); return code; };
    if (h->sort_keys.ensure(sizeof(uint32_t) * n, st) || h->sort_vals.ensure(sizeof(uint32_t) * n, st) ||
        h->sort_perm.ensure(sizeof(uint32_t) * n, st))
      return fail(GLOC_ERR_NOMEM);
    hipLaunchKernelGGL(morton_keys_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, s.xyz,
                       (uint32_t)n, mn[0], mn[1], mn[2], 1.0f / cell, h->sort_keys.as<uint32_t>(),
                       h->sort_vals.as<uint32_t>());
    size_t tmp_bytes = 0;
    if (hipcub::DeviceRadixSort::SortPairs(nullptr, tmp_bytes, h->sort_keys.as<uint32_t>(), keys,
                                           h->sort_vals.as<uint32_t>(), h->sort_perm.as<uint32_t>(),
                                           (int)n, 0, 30, st) != hipSuccess)
      return fail(GLOC_ERR_HIP);
    if (h->sort_tmp.ensure(std::max<size_t>(tmp_bytes, 16), st)) return fail(GLOC_ERR_NOMEM);
    if (hipcub::DeviceRadixSort::SortPairs(h->sort_tmp.p, tmp_bytes, h->sort_keys.as<uint32_t>(), keys,
                                           h->sort_vals.as<uint32_t>(), h->sort_perm.as<uint32_t>(),
                                           (int)n, 0, 30, st) != hipSuccess)
      return fail(GLOC_ERR_HIP);
    hipLaunchKernelGGL(gather_sorted_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, s.xyz,
                       h->sort_perm.as<uint32_t>(), (uint32_t)n, p4, inv);
    hipLaunchKernelGGL(chunk_boxes_kernel, dim3((unsigned)nch), dim3(64), 0, st, p4, (uint32_t)n, lo, hi);
    hipLaunchKernelGGL(subblock_boxes_kernel, dim3((unsigned)((nsb + 255) / 256)), dim3(256), 0, st, p4,
                       (uint32_t)n, (uint32_t)nsb, slo, shi);
    hipLaunchKernelGGL(super_boxes_kernel, dim3((unsigned)nsup), dim3(64), 0, st, lo, hi, (uint32_t)nch, ulo, uhi);
    if (hipGetLastError() != hipSuccess || hipStreamSynchronize(st) != hipSuccess) {
      set_err("scan indexing failed: %s", hipGetErrorString(hipGetLastError()));
      return fail(GLOC_ERR_HIP);
    }
  }
  *out = s;
  return GLOC_OK;
}

// Launch order of a source scan for the culled search: groups of 64*cs sorted points, widest first.
int build_order(gloc_reg* h, const DevScan& s, int cs) {
  if (s.order_cs == cs || s.n == 0) return GLOC_OK;
  hipStream_t st = h->stream;
  const uint32_t group = 64u * (uint32_t)cs;
  const uint32_t ng = (uint32_t)((s.n + group - 1) / group);
  GLOC_TRY(h->sort_keys.ensure(sizeof(float) * ng, st));
  GLOC_TRY(h->sort_vals.ensure(sizeof(uint32_t) * ng, st));
  GLOC_TRY(h->sort_perm.ensure(sizeof(float) * ng, st));
  hipLaunchKernelGGL(group_extent_kernel, dim3((ng + 3) / 4), dim3(256), 0, st, s.idx.pts, (uint32_t)s.n, group,
                     ng, h->sort_keys.as<float>(), h->sort_vals.as<uint32_t>());
  size_t tmp_bytes = 0;
  GLOC_HIP(hipcub::DeviceRadixSort::SortPairsDescending(nullptr, tmp_bytes, h->sort_keys.as<float>(),
                                                        h->sort_perm.as<float>(), h->sort_vals.as<uint32_t>(),
                                                        s.order, (int)ng, 0, 32, st));
  GLOC_TRY(h->sort_tmp.ensure(std::max<size_t>(tmp_bytes, 16), st));
  GLOC_HIP(hipcub::DeviceRadixSort::SortPairsDescending(h->sort_tmp.p, tmp_bytes, h->sort_keys.as<float>(),
                                                        h->sort_perm.as<float>(), h->sort_vals.as<uint32_t>(),
                                                        s.order, (int)ng, 0, 32, st));
  GLOC_HIP(hipGetLastError());
  s.order_cs = cs;
  return GLOC_OK;
}

void init_state(CandState& st, const float* T16, uint32_t ransac_iters = 0) {
  memset(&st, 0, sizeof(st));
  static const float I16[16] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1};
  const float* T = T16 ? T16 : I16;
  for (int i = 0; i < 3; ++i) {
    for (int j = 0; j < 3; ++j) {
      st.Tf[3 * i + j] = T[4 * i + j];
      st.Td[3 * i + j] = (double)T[4 * i + j];
    }
    st.Tf[9 + i] = T[4 * i + 3];
    st.Td[9 + i] = (double)T[4 * i + 3];
  }
  st.best_h = 0xFFFFFFFFu;
  st.niters = ransac_iters;
}

int launch_nn(gloc_reg* h, const DevScan& src, int n_cand, size_t ld, bool warm) {
  ProfScope ps(h->prof, "nn", h->stream);
  const uint32_t n_src = (uint32_t)src.n;
  h->nn_launches++;
  if (h->nn_mode == 1) {
    dim3 grid((n_src + 256 * NN_S - 1) / (256 * NN_S), (unsigned)n_cand);
    hipLaunchKernelGGL(nn_kernel, grid, dim3(256), 0, h->stream, src.xyz, n_src,
                       h->cands.as<CandDesc>(), h->states.as<CandState>(), h->corr.as<uint32_t>(),
                       h->d2.as<float>(), ld);
  } else {
    const int cs = h->nn_src_per_lane;
    dim3 grid((n_src + 256 * cs - 1) / (256 * cs), (unsigned)n_cand);
    if (h->trace_on) {
      h->trace_waves = (size_t)grid.x * grid.y * 4;
      if (h->trace.ensure(h->trace_waves * 16, h->stream)) return GLOC_ERR_NOMEM;
      GLOC_HIP(hipMemsetAsync(h->trace.p, 0, h->trace_waves * 16, h->stream));
    }
#define LAUNCH_CULLED(CS_)                                                                       \
  hipLaunchKernelGGL(KERNEL_<CS_>, grid, dim3(256), 0, h->stream, src.idx.pts, n_src,            \
                     h->ccands.as<CulledCand>(), h->states.as<CandState>(),                      \
                     warm ? h->corr.as<uint32_t>() : (const uint32_t*)nullptr,                   \
                     h->corr.as<uint32_t>(), h->d2.as<float>(), ld,                              \
                     h->prof.enabled ? h->counters.as<unsigned long long>()                      \
                                     : (unsigned long long*)nullptr,                             \
                     h->trace_on ? h->trace.as<uint32_t>() : (uint32_t*)nullptr)
    if (h->nn_mode == 2) {
#define KERNEL_ nn_culled_kernel
      if (cs == 1) LAUNCH_CULLED(1);
      else if (cs == 2) LAUNCH_CULLED(2);
      else LAUNCH_CULLED(4);
#undef KERNEL_
    } else {
      GLOC_TRY(build_order(h, src, cs));
      const uint32_t n_groups = (n_src + 64 * cs - 1) / (64 * cs), n_wg = (n_groups + 3) / 4;
      const uint32_t heavy = std::min<uint32_t>(n_wg, (uint32_t)(n_wg * h->nn_heavy_frac + 0.5f));
#define LAUNCH_COMPACT(CS_)                                                                         \
  hipLaunchKernelGGL(nn_compact_kernel<CS_>, dim3(n_wg * (unsigned)n_cand), dim3(256), 0, h->stream, \
                     src.idx.pts, n_src, h->ccands.as<CulledCand>(), h->states.as<CandState>(),    \
                     warm ? h->corr.as<uint32_t>() : (const uint32_t*)nullptr,                      \
                     h->corr.as<uint32_t>(), h->d2.as<float>(), ld, src.order, n_groups,            \
                     (uint32_t)n_cand, heavy,                                                       \
                     h->prof.enabled ? h->counters.as<unsigned long long>()                         \
                                     : (unsigned long long*)nullptr,                                \
                     h->trace_on ? h->trace.as<uint32_t>() : (uint32_t*)nullptr)
      if (cs == 1) LAUNCH_COMPACT(1);
      else if (cs == 2) LAUNCH_COMPACT(2);
      else LAUNCH_COMPACT(4);
#undef LAUNCH_COMPACT
    }
#undef LAUNCH_CULLED
  }
  GLOC_HIP(hipGetLastError());
  return GLOC_OK;
}

// The whole per-query pipeline, device resident: S1 -> S2 (RANSAC + refit) -> S3 (ICP).
int run_batch(gloc_reg* h, const DevScan& src, const std::vector<const DevScan*>& tg,
              const uint32_t* stream_ids, const float* init_T, const gloc_reg_params* prm,
              float* out_T, float* out_rmse, uint32_t* out_inliers, int* out_ok) {
  const int n_cand = (int)tg.size();
  const uint32_t n_src = (uint32_t)src.n;
  const float* d_src = src.xyz;
  std::vector<CandDesc> cds(n_cand);
  std::vector<CulledCand> ccs(n_cand);
  for (int c = 0; c < n_cand; ++c) {
    cds[c] = CandDesc{tg[c]->xyz, (uint32_t)tg[c]->n, stream_ids ? stream_ids[c] : (uint32_t)c};
    ccs[c] = CulledCand{tg[c]->idx, tg[c]->xyz};
  }
  const size_t ld = ((size_t)n_src + 63) & ~(size_t)63;
  hipStream_t s = h->stream;
  h->h_states.resize(n_cand);
  for (int c = 0; c < n_cand; ++c)
    init_state(h->h_states[c], init_T ? init_T + 16 * c : nullptr, prm->ransac_iters);
  GLOC_TRY(h->cands.ensure(sizeof(CandDesc) * n_cand, s));
  GLOC_TRY(h->ccands.ensure(sizeof(CulledCand) * n_cand, s));
  if (!h->counters.p) {
    GLOC_TRY(h->counters.ensure(64, s));
    GLOC_HIP(hipMemsetAsync(h->counters.p, 0, 64, s));
  }
  GLOC_HIP(hipMemcpyAsync(h->ccands.p, ccs.data(), sizeof(CulledCand) * n_cand,
                          hipMemcpyHostToDevice, s));
  GLOC_TRY(h->states.ensure(sizeof(CandState) * n_cand, s));
  GLOC_TRY(h->corr.ensure(sizeof(uint32_t) * ld * n_cand, s));
  GLOC_TRY(h->d2.ensure(sizeof(float) * ld * n_cand, s));
  const int nblocks = (int)((n_src + ACC_PER_BLOCK - 1) / ACC_PER_BLOCK);
  GLOC_TRY(h->partials.ensure(sizeof(double) * ACC_NV * std::max(nblocks, 1) * n_cand, s));
  GLOC_HIP(hipMemcpyAsync(h->cands.p, cds.data(), sizeof(CandDesc) * n_cand,
                          hipMemcpyHostToDevice, s));
  GLOC_HIP(hipMemcpyAsync(h->states.p, h->h_states.data(), sizeof(CandState) * n_cand,
                          hipMemcpyHostToDevice, s));
  const bool can = n_src >= 3;
  bool have_corr = false;  // corr holds a previous pass's result: warm start for the next one
  bool any_tgt = false;
  for (auto& c : cds) any_tgt |= c.n_tgt >= 1;

  if (can && any_tgt && prm->ransac_iters > 0) {
    const uint32_t H = prm->ransac_iters;
    GLOC_TRY(h->pairs.ensure(sizeof(f32x4) * 2 * ld * n_cand, s));
    GLOC_TRY(h->Rt.ensure(sizeof(float) * 12 * (size_t)H * n_cand, s));
    GLOC_TRY(h->valid.ensure(sizeof(uint32_t) * (size_t)H * n_cand, s));
    GLOC_TRY(h->inliers.ensure(sizeof(uint32_t) * (size_t)H * n_cand, s));
    GLOC_TRY(launch_nn(h, src, n_cand, ld, false));
    have_corr = true;
    {
      ProfScope ps(h->prof, "transform", s);
      hipLaunchKernelGGL(gather_pairs_kernel, dim3((n_src + 255) / 256, n_cand), dim3(256), 0, s,
                         d_src, n_src, h->cands.as<CandDesc>(), h->states.as<CandState>(),
                         h->corr.as<uint32_t>(), ld, h->pairs.as<f32x4>());
      GLOC_HIP(hipGetLastError());
    }
    const uint32_t HA = std::min<uint32_t>(H, prm->ransac_confidence > 0.f && prm->ransac_confidence < 1.f ? 64 : 256);
    GLOC_HIP(hipMemsetAsync(h->valid.p, 0, sizeof(uint32_t) * (size_t)H * n_cand, s));  // never-generated = invalid
    {
      ProfScope ps(h->prof, "ransac_hyp", s);  // phase A's hypotheses; the rest only where still needed
      hipLaunchKernelGGL(ransac_hyp_kernel, dim3((HA + 127) / 128, n_cand), dim3(128), 0, s,
                         h->pairs.as<f32x4>(), ld, n_src, h->cands.as<CandDesc>(), prm->seed, H, 0u, HA,
                         (const CandState*)nullptr, h->Rt.as<float>(), h->valid.as<uint32_t>());
      GLOC_HIP(hipGetLastError());
    }
    GLOC_HIP(hipMemsetAsync(h->inliers.p, 0, sizeof(uint32_t) * (size_t)H * n_cand, s));
    const float thr2 = prm->inlier_thresh * prm->inlier_thresh;
    {
      // phase A: the first 64 hypotheses (with ~90 % inliers the adaptive count is reached after a
      // handful); phase B: the rest, skipped per candidate once the adaptive iteration count has been
      // reached (then its blocks exit at once).  The split does not change the result.
      ProfScope ps(h->prof, "ransac_score", s);
      const uint32_t hpbA = HA <= 64 ? 64u : 256u;
      const unsigned cchunks = (n_src + SC_CHUNK - 1) / SC_CHUNK;
      hipLaunchKernelGGL(ransac_score_kernel, dim3((HA + hpbA - 1) / hpbA, cchunks, n_cand), dim3(256), 0, s,
                         h->pairs.as<f32x4>(), ld, n_src, H, 0u, hpbA, h->Rt.as<float>(),
                         h->valid.as<uint32_t>(), thr2, (const CandState*)nullptr,
                         h->inliers.as<uint32_t>());
      if (H > HA) {
        hipLaunchKernelGGL(ransac_scan_kernel<false>, dim3(n_cand), dim3(64), 0, s,
                           h->inliers.as<uint32_t>(), h->valid.as<uint32_t>(), h->Rt.as<float>(), H,
                           0u, HA, n_src, prm->ransac_confidence, prm->min_inlier_ratio,
                           h->states.as<CandState>());
        hipLaunchKernelGGL(ransac_hyp_kernel, dim3((H - HA + 127) / 128, n_cand), dim3(128), 0, s,
                           h->pairs.as<f32x4>(), ld, n_src, h->cands.as<CandDesc>(), prm->seed, H, HA, H,
                           h->states.as<CandState>(), h->Rt.as<float>(), h->valid.as<uint32_t>());
        hipLaunchKernelGGL(ransac_score_kernel, dim3((H - HA + 255) / 256, cchunks, n_cand), dim3(256),
                           0, s, h->pairs.as<f32x4>(), ld, n_src, H, HA, 256u, h->Rt.as<float>(),
                           h->valid.as<uint32_t>(), thr2, h->states.as<CandState>(),
                           h->inliers.as<uint32_t>());
        hipLaunchKernelGGL(ransac_scan_kernel<true>, dim3(n_cand), dim3(64), 0, s,
                           h->inliers.as<uint32_t>(), h->valid.as<uint32_t>(), h->Rt.as<float>(), H,
                           HA, H, n_src, prm->ransac_confidence, prm->min_inlier_ratio,
                           h->states.as<CandState>());
      } else {
        hipLaunchKernelGGL(ransac_scan_kernel<true>, dim3(n_cand), dim3(64), 0, s,
                           h->inliers.as<uint32_t>(), h->valid.as<uint32_t>(), h->Rt.as<float>(), H,
                           0u, HA, n_src, prm->ransac_confidence, prm->min_inlier_ratio,
                           h->states.as<CandState>());
      }
      GLOC_HIP(hipGetLastError());
    }
    {
      ProfScope ps(h->prof, "accum", s);
      hipLaunchKernelGGL(accum_kernel<1>, dim3(nblocks, n_cand), dim3(ACC_THREADS), 0, s, d_src,
                         n_src, h->cands.as<CandDesc>(), h->states.as<CandState>(),
                         h->corr.as<uint32_t>(), h->d2.as<float>(), h->pairs.as<f32x4>(), ld, thr2,
                         h->partials.as<double>());
      GLOC_HIP(hipGetLastError());
    }
    {
      ProfScope ps(h->prof, "solve", s);
      if (prm->icp_iters == 0) {
        hipLaunchKernelGGL(sumd2_kernel, dim3(n_cand), dim3(256), 0, s, h->d2.as<float>(), ld,
                           n_src, n_cand, h->states.as<CandState>());
      }
      hipLaunchKernelGGL(solve_kernel<1>, dim3(n_cand), dim3(64), 0, s,
                         h->partials.as<double>(), nblocks, n_cand, h->states.as<CandState>());
      GLOC_HIP(hipGetLastError());
    }
  }
  const float gate2 = prm->max_corr_dist > 0.f ? prm->max_corr_dist * prm->max_corr_dist : 0.f;
  for (uint32_t it = 0; it < prm->icp_iters && can && any_tgt; ++it) {
    GLOC_TRY(launch_nn(h, src, n_cand, ld, have_corr));
    have_corr = true;
    {
      ProfScope ps(h->prof, "accum", s);
      hipLaunchKernelGGL(accum_kernel<0>, dim3(nblocks, n_cand), dim3(ACC_THREADS), 0, s, d_src,
                         n_src, h->cands.as<CandDesc>(), h->states.as<CandState>(),
                         h->corr.as<uint32_t>(), h->d2.as<float>(), (const f32x4*)nullptr, ld,
                         gate2, h->partials.as<double>());
      GLOC_HIP(hipGetLastError());
    }
    {
      ProfScope ps(h->prof, "solve", s);
      hipLaunchKernelGGL(solve_kernel<0>, dim3(n_cand), dim3(64), 0, s,
                         h->partials.as<double>(), nblocks, n_cand, h->states.as<CandState>());
      GLOC_HIP(hipGetLastError());
    }
  }
  GLOC_HIP(hipMemcpyAsync(h->h_states.data(), h->states.p, sizeof(CandState) * n_cand,
                          hipMemcpyDeviceToHost, s));
  GLOC_HIP(hipStreamSynchronize(s));
  for (int c = 0; c < n_cand; ++c) {
    const CandState& st = h->h_states[c];
    float* T = out_T + 16 * c;
    for (int i = 0; i < 3; ++i) {
      for (int j = 0; j < 3; ++j) T[4 * i + j] = st.Tf[3 * i + j];
      T[4 * i + 3] = st.Tf[9 + i];
    }
    T[12] = 0.f; T[13] = 0.f; T[14] = 0.f; T[15] = 1.f;
    if (out_rmse) out_rmse[c] = n_src ? (float)std::sqrt(st.sum_d2 / (double)n_src) : 0.f;
    if (out_inliers) out_inliers[c] = st.best_inl;
    if (out_ok) out_ok[c] = st.ok;
  }
  return GLOC_OK;
}

int check_params(const gloc_reg_params* p) {
  GLOC_REQUIRE(p, GLOC_ERR_INVALID, "params is null");
  GLOC_REQUIRE(p->ransac_iters <= (1u << 20), GLOC_ERR_INVALID, "ransac_iters too large");
  GLOC_REQUIRE(p->icp_iters <= 10000, GLOC_ERR_INVALID, "icp_iters too large");
  GLOC_REQUIRE(p->ransac_iters == 0 || p->inlier_thresh > 0.f, GLOC_ERR_INVALID,
               "inlier_thresh must be > 0");
  return GLOC_OK;
}

}  // namespace

extern "C" {

void gloc_reg_default_params(gloc_reg_params* p) {
  if (!p) return;
  p->ransac_iters = 3000;   // registration/loop_detector.cpp:257
  p->inlier_thresh = 0.6f;  // 3 * 0.2 m: loop_detector.cpp:257, loop_detector.h:116
  p->min_inlier_ratio = 0.3f;
  p->icp_iters = 30;  // registration/global_registration.cpp:242
  p->max_corr_dist = 0.f;
  p->seed = 1234;
  p->ransac_confidence = 0.99f;  // cv::estimateAffinePartial2D's default, used by the reference
  p->reserved_ = 0;
}

int gloc_reg_create(int device, gloc_reg** out) {
  GLOC_REQUIRE(out, GLOC_ERR_INVALID, "out is null");
  *out = nullptr;
  GLOC_TRY(select_device(device));
  gloc_reg* h = new (std::nothrow) gloc_reg;
  GLOC_REQUIRE(h, GLOC_ERR_NOMEM, "host allocation failed");
  h->device = device;
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

int gloc_reg_scan_clear(gloc_reg* h) {
  GLOC_REQUIRE(h, GLOC_ERR_INVALID, "null handle");
  GLOC_HIP(hipSetDevice(h->device));
  GLOC_HIP(hipStreamSynchronize(h->stream));
  for (auto& s : h->scans) free_scan(s);
  h->scans.clear();
  return GLOC_OK;
}

int gloc_reg_destroy(gloc_reg* h) {
  if (!h) return GLOC_OK;
  (void)hipSetDevice(h->device);
  (void)hipStreamSynchronize(h->stream);
  (void)gloc_reg_scan_clear(h);
  h->prof.destroy();
  for (DevBuf* b : {&h->cands, &h->states, &h->corr, &h->d2, &h->pairs,
                    &h->Rt, &h->valid, &h->inliers, &h->partials, &h->ccands, &h->sort_tmp,
                    &h->sort_keys, &h->sort_vals, &h->sort_perm, &h->counters})
    b->release();
  if (h->own_stream) (void)hipStreamDestroy(h->own_stream);
  delete h;
  return GLOC_OK;
}

int gloc_reg_set_stream(gloc_reg* h, void* hip_stream) {
  GLOC_REQUIRE(h, GLOC_ERR_INVALID, "null handle");
  GLOC_HIP(hipSetDevice(h->device));
  GLOC_HIP(hipStreamSynchronize(h->stream));
  h->stream = hip_stream ? (hipStream_t)hip_stream : h->own_stream;
  return GLOC_OK;
}

int gloc_reg_synchronize(gloc_reg* h) {
  GLOC_REQUIRE(h, GLOC_ERR_INVALID, "null handle");
  GLOC_HIP(hipSetDevice(h->device));
  GLOC_HIP(hipStreamSynchronize(h->stream));
  return GLOC_OK;
}

int gloc_reg_set_option(gloc_reg* h, int option, int64_t value) {
  GLOC_REQUIRE(h, GLOC_ERR_INVALID, "null handle");
  if (option == GLOC_REG_OPT_PROFILE) {
    h->prof.enabled = value != 0;
    return GLOC_OK;
  }
  if (option == GLOC_REG_OPT_NN_MODE) {
    GLOC_REQUIRE(value == GLOC_REG_NN_CULLED || value == GLOC_REG_NN_EXHAUSTIVE ||
                     value == GLOC_REG_NN_CULLED_BROADCAST, GLOC_ERR_INVALID,
                 "bad nn mode %lld", (long long)value);
    h->nn_mode = (int)value;
    return GLOC_OK;
  }
  if (option == GLOC_REG_OPT_NN_SRC_PER_LANE) {
    GLOC_REQUIRE(value == 1 || value == 2 || value == 4, GLOC_ERR_INVALID, "must be 1, 2 or 4");
    h->nn_src_per_lane = (int)value;
    return GLOC_OK;
  }
  if (option == GLOC_REG_OPT_NN_HEAVY_PERMILLE) {
    GLOC_REQUIRE(value >= 0 && value <= 1000, GLOC_ERR_INVALID, "must be in [0, 1000]");
    h->nn_heavy_frac = (float)value / 1000.0f;
    return GLOC_OK;
  }
  set_err("unknown option %d", option);
  return GLOC_ERR_INVALID;
}

int gloc_reg_scan_upload(gloc_reg* h, const float* pts, size_t n, size_t stride_floats,
                         uint32_t* scan_id) {
  GLOC_REQUIRE(h && scan_id && (pts || n == 0), GLOC_ERR_INVALID, "null argument");
  GLOC_REQUIRE(stride_floats >= 3 && stride_floats <= 16, GLOC_ERR_INVALID,
               "stride_floats = %zu outside [3,16]", stride_floats);
  GLOC_REQUIRE(n < (1ull << 31), GLOC_ERR_INVALID, "scan too large");
  GLOC_HIP(hipSetDevice(h->device));
  DevScan s;
  GLOC_TRY(make_scan(h, pts, n, stride_floats, &s));
  h->scans.push_back(s);
  *scan_id = (uint32_t)(h->scans.size() - 1);
  return GLOC_OK;
}

int gloc_reg_scan_count(const gloc_reg* h, size_t* n_scans) {
  GLOC_REQUIRE(h && n_scans, GLOC_ERR_INVALID, "null argument");
  *n_scans = h->scans.size();
  return GLOC_OK;
}

int gloc_reg_batch(gloc_reg* h, const float* q_xyz, size_t nq_pts, const float* const* cand_xyz,
                   const size_t* cand_npts, size_t n_cand, const uint32_t* cand_stream_ids,
                   const float* init_T, const gloc_reg_params* params, float* out_T,
                   float* out_rmse, uint32_t* out_inliers, int* out_ok) {
  GLOC_REQUIRE(h && out_T && (q_xyz || nq_pts == 0), GLOC_ERR_INVALID, "null argument");
  GLOC_REQUIRE(n_cand >= 1 && n_cand <= 4096 && cand_xyz && cand_npts, GLOC_ERR_INVALID,
               "n_cand = %zu outside [1,4096] or null candidate arrays", n_cand);
  GLOC_REQUIRE(nq_pts < (1ull << 31), GLOC_ERR_INVALID, "query scan too large");
  GLOC_TRY(check_params(params));
  GLOC_HIP(hipSetDevice(h->device));
  for (size_t c = 0; c < n_cand; ++c) {
    GLOC_REQUIRE(cand_xyz[c] || cand_npts[c] == 0, GLOC_ERR_INVALID, "candidate %zu is null", c);
    GLOC_REQUIRE(cand_npts[c] < (1ull << 31), GLOC_ERR_INVALID, "candidate scan too large");
  }
  // temporary resident copies (uploaded + indexed), released on return
  std::vector<DevScan> tmp(n_cand + 1);
  int rc = make_scan(h, q_xyz, nq_pts, 3, &tmp[0]);
  for (size_t c = 0; rc == GLOC_OK && c < n_cand; ++c)
    rc = make_scan(h, cand_xyz[c], cand_npts[c], 3, &tmp[c + 1]);
  if (rc == GLOC_OK) {
    std::vector<const DevScan*> tg(n_cand);
    for (size_t c = 0; c < n_cand; ++c) tg[c] = &tmp[c + 1];
    rc = run_batch(h, tmp[0], tg, cand_stream_ids, init_T, params, out_T, out_rmse, out_inliers,
                   out_ok);
  }
  (void)hipStreamSynchronize(h->stream);
  for (auto& s : tmp) free_scan(s);
  return rc;
}

int gloc_reg_batch_ids(gloc_reg* h, uint32_t q_scan_id, const uint32_t* cand_scan_ids,
                       size_t n_cand, const uint32_t* cand_stream_ids, const float* init_T,
                       const gloc_reg_params* params, float* out_T, float* out_rmse,
                       uint32_t* out_inliers, int* out_ok) {
  GLOC_REQUIRE(h && out_T && cand_scan_ids, GLOC_ERR_INVALID, "null argument");
  GLOC_REQUIRE(n_cand >= 1 && n_cand <= 4096, GLOC_ERR_INVALID, "n_cand = %zu outside [1,4096]",
               n_cand);
  GLOC_REQUIRE(q_scan_id < h->scans.size(), GLOC_ERR_INVALID, "unknown query scan id %u", q_scan_id);
  GLOC_TRY(check_params(params));
  GLOC_HIP(hipSetDevice(h->device));
  std::vector<const DevScan*> tg(n_cand);
  for (size_t c = 0; c < n_cand; ++c) {
    GLOC_REQUIRE(cand_scan_ids[c] < h->scans.size(), GLOC_ERR_INVALID, "unknown scan id %u",
                 cand_scan_ids[c]);
    tg[c] = &h->scans[cand_scan_ids[c]];
  }
  return run_batch(h, h->scans[q_scan_id], tg, cand_stream_ids, init_T, params, out_T, out_rmse,
                   out_inliers, out_ok);
}

int gloc_reg_select_first_ok(const int* ok, size_t n_cand) {
  if (!ok) return -1;
  for (size_t i = 0; i < n_cand; ++i)
    if (ok[i]) return (int)i;
  return -1;
}

int gloc_reg_nn(gloc_reg* h, const float* src_xyz, size_t n_src, const float* tgt_xyz,
                size_t n_tgt, const float* T16, uint32_t* out_idx, float* out_d2) {
  GLOC_REQUIRE(h && out_idx && out_d2 && (src_xyz || !n_src) && (tgt_xyz || !n_tgt),
               GLOC_ERR_INVALID, "null argument");
  GLOC_REQUIRE(n_src < (1ull << 31) && n_tgt < (1ull << 31), GLOC_ERR_INVALID, "scan too large");
  if (n_src == 0) return GLOC_OK;
  GLOC_HIP(hipSetDevice(h->device));
  hipStream_t s = h->stream;
  const size_t ld = (n_src + 63) & ~(size_t)63;
  DevScan src, tgt;
  GLOC_TRY(make_scan(h, src_xyz, n_src, 3, &src));
  int rc = make_scan(h, tgt_xyz, n_tgt, 3, &tgt);
  auto done = [&](int code) {
    (void)hipStreamSynchronize(s);
    free_scan(src);
    free_scan(tgt);
    return code;
  };
  if (rc != GLOC_OK) return done(rc);
  if (h->cands.ensure(sizeof(CandDesc), s) || h->ccands.ensure(sizeof(CulledCand), s) ||
      h->states.ensure(sizeof(CandState), s) || h->corr.ensure(sizeof(uint32_t) * ld, s) ||
      h->d2.ensure(sizeof(float) * ld, s) || h->counters.ensure(64, s))
    return done(GLOC_ERR_NOMEM);
  CandDesc cd{tgt.xyz, (uint32_t)n_tgt, 0};
  CulledCand cc{tgt.idx, tgt.xyz};
  CandState st;
  init_state(st, T16);
  if (hipMemcpyAsync(h->cands.p, &cd, sizeof(cd), hipMemcpyHostToDevice, s) != hipSuccess ||
      hipMemcpyAsync(h->ccands.p, &cc, sizeof(cc), hipMemcpyHostToDevice, s) != hipSuccess ||
      hipMemcpyAsync(h->states.p, &st, sizeof(st), hipMemcpyHostToDevice, s) != hipSuccess)
    return done(GLOC_ERR_HIP);
  rc = launch_nn(h, src, 1, ld, false);
  if (rc != GLOC_OK) return done(rc);
  if (hipMemcpyAsync(out_idx, h->corr.p, sizeof(uint32_t) * n_src, hipMemcpyDeviceToHost, s) != hipSuccess ||
      hipMemcpyAsync(out_d2, h->d2.p, sizeof(float) * n_src, hipMemcpyDeviceToHost, s) != hipSuccess ||
      hipStreamSynchronize(s) != hipSuccess) {
    set_err("gloc_reg_nn: copy back failed: %s", hipGetErrorString(hipGetLastError()));
    return done(GLOC_ERR_HIP);
  }
  return done(GLOC_OK);
}

int gloc_reg_ransac_hypotheses(gloc_reg* h, const float* src_xyz, const float* tgt_xyz,
                               const uint32_t* corr, size_t n, uint64_t seed, uint32_t cand,
                               uint32_t n_hyp, float* out_Rt, uint32_t* out_valid,
                               uint32_t* out_inliers, float inlier_thresh) {
  GLOC_REQUIRE(h && src_xyz && tgt_xyz && corr && out_Rt && out_valid && out_inliers,
               GLOC_ERR_INVALID, "null argument");
  GLOC_REQUIRE(n >= 3 && n < (1ull << 31) && n_hyp >= 1 && n_hyp <= (1u << 20), GLOC_ERR_INVALID,
               "bad sizes");
  GLOC_HIP(hipSetDevice(h->device));
  hipStream_t s = h->stream;
  const size_t ld = (n + 63) & ~(size_t)63;
  // pairs are built on the host from (src, tgt[corr]) -- src is taken as already moved
  std::vector<float> hp(ld * 8, 0.f);
  for (size_t i = 0; i < n; ++i) {
    for (int a = 0; a < 3; ++a) {
      hp[i * 8 + a] = src_xyz[3 * i + a];
      hp[i * 8 + 4 + a] = tgt_xyz[3 * (size_t)corr[i] + a];
    }
  }
  GLOC_TRY(h->pairs.ensure(sizeof(float) * 8 * ld, s));
  GLOC_TRY(h->cands.ensure(sizeof(CandDesc), s));
  GLOC_TRY(h->Rt.ensure(sizeof(float) * 12 * (size_t)n_hyp, s));
  GLOC_TRY(h->valid.ensure(sizeof(uint32_t) * (size_t)n_hyp, s));
  GLOC_TRY(h->inliers.ensure(sizeof(uint32_t) * (size_t)n_hyp, s));
  CandDesc cd{nullptr, 0, cand};
  GLOC_HIP(hipMemcpyAsync(h->cands.p, &cd, sizeof(cd), hipMemcpyHostToDevice, s));
  GLOC_HIP(hipMemcpyAsync(h->pairs.p, hp.data(), sizeof(float) * 8 * ld, hipMemcpyHostToDevice, s));
  GLOC_HIP(hipMemsetAsync(h->inliers.p, 0, sizeof(uint32_t) * (size_t)n_hyp, s));
  hipLaunchKernelGGL(ransac_hyp_kernel, dim3((n_hyp + 127) / 128, 1), dim3(128), 0, s,
                     h->pairs.as<f32x4>(), ld, (uint32_t)n, h->cands.as<CandDesc>(), seed, n_hyp, 0u, n_hyp,
                     (const CandState*)nullptr, h->Rt.as<float>(), h->valid.as<uint32_t>());
  GLOC_HIP(hipGetLastError());
  dim3 grid((n_hyp + 255) / 256, (unsigned)((n + SC_CHUNK - 1) / SC_CHUNK), 1);
  hipLaunchKernelGGL(ransac_score_kernel, grid, dim3(256), 0, s, h->pairs.as<f32x4>(), ld,
                     (uint32_t)n, n_hyp, 0u, 256u /* thread <-> hypothesis */, h->Rt.as<float>(), h->valid.as<uint32_t>(),
                     inlier_thresh * inlier_thresh, (const CandState*)nullptr,
                     h->inliers.as<uint32_t>());
  GLOC_HIP(hipGetLastError());
  GLOC_HIP(hipMemcpyAsync(out_Rt, h->Rt.p, sizeof(float) * 12 * (size_t)n_hyp,
                          hipMemcpyDeviceToHost, s));
  GLOC_HIP(hipMemcpyAsync(out_valid, h->valid.p, sizeof(uint32_t) * (size_t)n_hyp,
                          hipMemcpyDeviceToHost, s));
  GLOC_HIP(hipMemcpyAsync(out_inliers, h->inliers.p, sizeof(uint32_t) * (size_t)n_hyp,
                          hipMemcpyDeviceToHost, s));
  GLOC_HIP(hipStreamSynchronize(s));
  return GLOC_OK;
}

int gloc_reg_nn_stats(gloc_reg* h, uint64_t* pairs_evaluated, uint64_t* launches) {
  GLOC_REQUIRE(h, GLOC_ERR_INVALID, "null handle");
  GLOC_HIP(hipSetDevice(h->device));
  unsigned long long c = 0;
  if (h->counters.p) {
    GLOC_HIP(hipMemcpyAsync(&c, h->counters.p, sizeof(c), hipMemcpyDeviceToHost, h->stream));
    GLOC_HIP(hipStreamSynchronize(h->stream));
  }
  if (pairs_evaluated) *pairs_evaluated = c;
  if (launches) *launches = h->nn_launches;
  return GLOC_OK;
}

// Developer aid (not part of include/gloc3d.h): per-wave trace of the LAST culled 1-NN launch.
int gloc_reg_debug_trace(gloc_reg* h, int enable, uint32_t* out, size_t cap_waves, size_t* n_waves) {
  GLOC_REQUIRE(h, GLOC_ERR_INVALID, "null handle");
  GLOC_HIP(hipSetDevice(h->device));
  h->trace_on = enable != 0;
  if (n_waves) *n_waves = h->trace_waves;
  if (out && h->trace.p) {
    const size_t n = std::min(cap_waves, h->trace_waves);
    GLOC_HIP(hipMemcpyAsync(out, h->trace.p, n * 16, hipMemcpyDeviceToHost, h->stream));
    GLOC_HIP(hipStreamSynchronize(h->stream));
  }
  return GLOC_OK;
}

int gloc_reg_profile(gloc_reg* h, const char* kernel, double* total_ms, uint64_t* launches) {
  GLOC_REQUIRE(h && kernel, GLOC_ERR_INVALID, "null argument");
  GLOC_HIP(hipSetDevice(h->device));
  GLOC_TRY(h->prof.collect(h->stream));
  auto it = h->prof.fam.find(kernel);
  if (total_ms) *total_ms = it == h->prof.fam.end() ? 0.0 : it->second.total_ms;
  if (launches) *launches = it == h->prof.fam.end() ? 0 : it->second.launches;
  return GLOC_OK;
}

int gloc_reg_profile_reset(gloc_reg* h) {
  GLOC_REQUIRE(h, GLOC_ERR_INVALID, "null handle");
  GLOC_HIP(hipSetDevice(h->device));
  GLOC_HIP(hipStreamSynchronize(h->stream));
  h->prof.reset();
  h->nn_launches = 0;
  if (h->counters.p) GLOC_HIP(hipMemsetAsync(h->counters.p, 0, 64, h->stream));
  return GLOC_OK;
}

}  // extern "C"
