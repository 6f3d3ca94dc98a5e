// ground.hip -- C ABI of the ground pre-alignment (include/gloc3d.h, "next" row N3 of SURVEY.md 8f).
// Replaces GroundEstimator::EsitmateGroundAndTransform (registration/ground_estimator.cpp:196-228) and
// its helpers FilterGroundByNormals (:63-161), EstimateGround (:19-61), TransformPointsToGround
// (:163-194).  Steps G1..G7 are stated in oracle/ground_oracle.c; the host-side parts here (bin choice,
// the sequential RANSAC rule, T_l2g from the plane) are a few dozen scalar operations.
#include <algorithm>
#include <cmath>
#include <new>

#include "common.hpp"
#include "ground_kernels.hpp"
#include "seg_sort.hpp"

using namespace gloc;
using namespace gloc::ground;

struct gloc_ground {
  int device = 0;
  hipStream_t own_stream = nullptr, stream = nullptr;
  DevBuf stage_in, stage_out;            // host-pointer API staging
  DevBuf flag, sel, count, tile_cnt;     // stream compaction
  DevBuf sort_segs, sort_hist;           // the segmented radix sort's descriptor and scratch (seg_sort.hpp)
  DevBuf near, knn_idx, knn_d2, knn_pidx, knn_pd2, bins, hist, normals;
  DevBuf skeys, svals, skeys2, sperm, spts, cbox_lo, cbox_hi;  // culled 10-NN: Hilbert-sorted copy + chunk boxes
  int knn_exhaustive = 0;                // 1: the exhaustive form (kept for comparison)
  DevBuf gpts, planes, valid, inliers, T12;
  Profiler prof;
};

namespace {

// Smallest k with (1 - w^3)^k <= 1 - conf by repeated multiplication (the adaptive iteration count of
// pcl::RandomSampleConsensus, transcendental-free so that every implementation agrees), capped.
uint32_t needed_iters(uint32_t inl, uint32_t n, float conf, uint32_t max_iters) {
  const double w = (double)inl / (double)n;
  const double q = 1.0 - (w * w) * w;
  const double target = 1.0 - (double)conf;
  double pw = 1.0;
  uint32_t k = 0;
  while (pw > target && k < max_iters) {
    pw = pw * q;
    k++;
  }
  return k;
}

void identity16(float* T) {
  for (int i = 0; i < 16; ++i) T[i] = (i % 5 == 0) ? 1.f : 0.f;
}

// TransformPointsToGround (ground_estimator.cpp:163-194): Eigen::Quaternionf::FromTwoVectors(n, z),
// toRotationMatrix().eulerAngles(2, 1, 0) with Eigen 3.3's branch (first angle in [0, pi]),
// cartographer::transform::RollPitchYaw(roll, pitch, 0), translation (0, 0, |d| / |n|).
void transform_from_plane(const float plane[4], float* T16) {
  double n[3] = {plane[0], plane[1], plane[2]};
  const double len = std::sqrt((n[0] * n[0] + n[1] * n[1]) + n[2] * n[2]);
  const double height = std::fabs((double)plane[3]) / len;
  const double sgn = plane[2] < 0.f ? -1.0 : 1.0;  // "make sure the normal vector is upward"
  for (int a = 0; a < 3; ++a) n[a] = sgn * n[a] / len;
  double q[4];  // w x y z
  const double c = n[2];
  if (c < -1.0 + 1e-12) {
    q[0] = 0; q[1] = 1; q[2] = 0; q[3] = 0;
  } else {
    const double ax[3] = {n[1] * 1.0 - n[2] * 0.0, n[2] * 0.0 - n[0] * 1.0, 0.0};
    const double s = std::sqrt((1.0 + c) * 2.0), inv = 1.0 / s;
    q[0] = s * 0.5; q[1] = ax[0] * inv; q[2] = ax[1] * inv; q[3] = ax[2] * inv;
  }
  const double qn = std::sqrt(((q[0] * q[0] + q[1] * q[1]) + q[2] * q[2]) + q[3] * q[3]);
  for (int a = 0; a < 4; ++a) q[a] /= qn;
  const double w = q[0], x = q[1], y = q[2], z = q[3];
  const double R[9] = {1 - 2 * (y * y + z * z), 2 * (x * y - z * w),     2 * (x * z + y * w),
                       2 * (x * y + z * w),     1 - 2 * (x * x + z * z), 2 * (y * z - x * w),
                       2 * (x * z - y * w),     2 * (y * z + x * w),     1 - 2 * (x * x + y * y)};
  double e0 = std::atan2(R[3], R[0]);
  const double c2 = std::sqrt(R[8] * R[8] + R[7] * R[7]);
  double e1;
  if (e0 < 0.0) {
    e0 += M_PI;
    e1 = std::atan2(-R[6], -c2);
  } else {
    e1 = std::atan2(-R[6], c2);
  }
  const double s1 = std::sin(e0), c1 = std::cos(e0);
  const double e2 = std::atan2(s1 * R[2] - c1 * R[5], c1 * R[4] - s1 * R[1]);
  const double cp = std::cos(e1), sp = std::sin(e1), cr = std::cos(e2), sr = std::sin(e2);
  const double Rn[9] = {cp, sp * sr, sp * cr, 0.0, cr, -sr, -sp, cp * sr, cp * cr};
  for (int i = 0; i < 16; ++i) T16[i] = 0.f;
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) T16[4 * i + j] = (float)Rn[3 * i + j];
  T16[11] = (float)height;
  T16[15] = 1.f;
}

int check_params(const gloc_ground_params* p) {
  GLOC_REQUIRE(p, GLOC_ERR_INVALID, "params is NULL");
  GLOC_REQUIRE(p->knn >= 3 && p->knn <= (uint32_t)KMAX, GLOC_ERR_INVALID, "knn must be in [3, %d]", KMAX);
  GLOC_REQUIRE(p->near_range2 > 0.f && p->plane_thresh > 0.f, GLOC_ERR_INVALID, "range and threshold must be positive");
  GLOC_REQUIRE(p->ransac_iters >= 1 && p->ransac_iters <= 65536, GLOC_ERR_INVALID, "ransac_iters must be in [1, 65536]");
  return GLOC_OK;
}

// Stable compaction of the indices whose flag is set: sel[0 .. *count) ascending.
int select_flagged(gloc_ground* h, uint32_t n, uint32_t* h_count) {
  hipStream_t s = h->stream;
  GLOC_TRY(h->sel.ensure(sizeof(uint32_t) * std::max<uint32_t>(n, 1), s));
  GLOC_TRY(h->count.ensure(sizeof(uint32_t), s));
  const uint32_t n_tiles = (n + SEL_TILE - 1) / SEL_TILE;
  GLOC_TRY(h->tile_cnt.ensure(sizeof(uint32_t) * std::max<uint32_t>(n_tiles, 1), s));
  if (n_tiles) {
    hipLaunchKernelGGL(flag_count_kernel, dim3(n_tiles), dim3(256), 0, s, h->flag.as<uint8_t>(), n, h->tile_cnt.as<uint32_t>());
    hipLaunchKernelGGL(flag_scatter_kernel, dim3(n_tiles), dim3(256), 0, s, h->flag.as<uint8_t>(), n, h->tile_cnt.as<uint32_t>(),
                       n_tiles, h->sel.as<uint32_t>(), h->count.as<uint32_t>());
    GLOC_HIP(hipGetLastError());
  } else {
    GLOC_HIP(hipMemsetAsync(h->count.p, 0, sizeof(uint32_t), s));
  }
  GLOC_HIP(hipMemcpyAsync(h_count, h->count.p, sizeof(uint32_t), hipMemcpyDeviceToHost, s));
  GLOC_HIP(hipStreamSynchronize(s));
  return GLOC_OK;
}

__global__ void one_segment_kernel(segsort::Seg* seg, uint32_t n) { *seg = segsort::Seg{0u, n}; }

// key_range: every coordinate of the cloud lies in [-key_range, key_range] (the range filter's radius);
// points outside are clamped into the outermost cells, which only loosens the order, not the result.
int knn_device(gloc_ground* h, const f32x4* d_pts, uint32_t m, uint32_t k, float key_range) {
  hipStream_t s = h->stream;
  GLOC_TRY(h->knn_idx.ensure(sizeof(uint32_t) * (size_t)m * k, s));
  GLOC_TRY(h->knn_d2.ensure(sizeof(float) * (size_t)m * k, s));
  if (!h->knn_exhaustive && m > 4 * KCH) {
    ProfScope ps(h->prof, "ground_knn", s);
    GLOC_TRY(h->skeys.ensure(sizeof(uint32_t) * m, s));
    GLOC_TRY(h->svals.ensure(sizeof(uint32_t) * m, s));
    GLOC_TRY(h->skeys2.ensure(sizeof(uint32_t) * m, s));
    GLOC_TRY(h->sperm.ensure(sizeof(uint32_t) * m, s));
    GLOC_TRY(h->spts.ensure(sizeof(f32x4) * m, s));
    const uint32_t nch = (m + KCH - 1) / KCH;
    GLOC_TRY(h->cbox_lo.ensure(sizeof(f32x4) * nch, s));
    GLOC_TRY(h->cbox_hi.ensure(sizeof(f32x4) * nch, s));
    hipLaunchKernelGGL(hilbert_keys_kernel, dim3((m + 255) / 256), dim3(256), 0, s, d_pts, m, -key_range,
                       1023.0f / (2.0f * key_range), h->skeys.as<uint32_t>(), h->svals.as<uint32_t>());
    // the curve keys' order: the repo's own stable segmented radix sort (round 4: hipcub::DeviceRadixSort before), one
    // segment, four 8-bit digits
    // (the one-segment descriptor is written by a one-thread kernel: a copy from a stack variable needed a host
    // synchronisation inside every scan's ground stage -- ADVICE r4)
    GLOC_TRY(h->sort_segs.ensure(sizeof(segsort::Seg), s));
    GLOC_TRY(h->sort_hist.ensure(segsort::scratch_bytes(1, m), s));
    hipLaunchKernelGGL(one_segment_kernel, dim3(1), dim3(1), 0, s, h->sort_segs.as<segsort::Seg>(), m);
    uint32_t* vbuf[2] = {h->svals.as<uint32_t>(), h->sperm.as<uint32_t>()};
    const int cur = segsort::sort_pairs<uint32_t, 8>(s, h->skeys.as<uint32_t>(), h->skeys2.as<uint32_t>(), vbuf[0], vbuf[1],
                                                     h->sort_segs.as<segsort::Seg>(), 1, m, 0, 32, h->sort_hist.as<uint32_t>());
    GLOC_HIP(hipGetLastError());
    const uint32_t* perm = vbuf[cur];
    hipLaunchKernelGGL(gather_sorted_f4_kernel, dim3((m + 255) / 256), dim3(256), 0, s, d_pts, perm, m, h->spts.as<f32x4>());
    hipLaunchKernelGGL(kchunk_boxes_kernel, dim3(nch), dim3(64), 0, s, h->spts.as<f32x4>(), m,
                       h->cbox_lo.as<f32x4>(), h->cbox_hi.as<f32x4>());
    hipLaunchKernelGGL(knn_culled_kernel, dim3((nch + 3) / 4), dim3(256), 0, s, h->spts.as<f32x4>(), m,
                       h->cbox_lo.as<f32x4>(), h->cbox_hi.as<f32x4>(), nch, (int)k, h->knn_idx.as<uint32_t>(),
                       h->knn_d2.as<float>());
    GLOC_HIP(hipGetLastError());
    return GLOC_OK;
  }
  // enough target slices for ~8 waves per SIMD (a wave of this kernel is latency-bound), each a whole
  // number of LDS tiles
  const uint32_t src_blocks = (m + KNN_BLOCK - 1) / KNN_BLOCK;
  uint32_t slices = std::max<uint32_t>(1, std::min<uint32_t>(32, 4096 / std::max<uint32_t>(src_blocks, 1)));
  slices = std::min<uint32_t>(slices, (m + KNN_TILE - 1) / KNN_TILE);
  const uint32_t slice_len = ((m + slices - 1) / slices + KNN_TILE - 1) / KNN_TILE * KNN_TILE;
  slices = (m + slice_len - 1) / slice_len;
  ProfScope ps(h->prof, "ground_knn", s);
  if (slices == 1) {
    hipLaunchKernelGGL(knn_self_kernel, dim3(src_blocks, 1), dim3(KNN_BLOCK), 0, s, d_pts, m, (int)k, slice_len,
                       h->knn_idx.as<uint32_t>(), h->knn_d2.as<float>());
  } else {
    GLOC_TRY(h->knn_pidx.ensure(sizeof(uint32_t) * (size_t)slices * m * k, s));
    GLOC_TRY(h->knn_pd2.ensure(sizeof(float) * (size_t)slices * m * k, s));
    hipLaunchKernelGGL(knn_self_kernel, dim3(src_blocks, slices), dim3(KNN_BLOCK), 0, s, d_pts, m, (int)k, slice_len,
                       h->knn_pidx.as<uint32_t>(), h->knn_pd2.as<float>());
    hipLaunchKernelGGL(knn_merge_kernel, dim3((m + 255) / 256), dim3(256), 0, s, h->knn_pidx.as<uint32_t>(),
                       h->knn_pd2.as<float>(), m, (int)k, (int)slices, h->knn_idx.as<uint32_t>(),
                       h->knn_d2.as<float>());
  }
  GLOC_HIP(hipGetLastError());
  return GLOC_OK;
}

int normals_device(gloc_ground* h, const f32x4* d_pts, uint32_t m, uint32_t k, bool want_normals) {
  hipStream_t s = h->stream;
  GLOC_TRY(h->bins.ensure(std::max<uint32_t>(m, 16), s));
  GLOC_TRY(h->hist.ensure(sizeof(uint32_t) * 18, s));
  if (want_normals) GLOC_TRY(h->normals.ensure(sizeof(float) * 3 * (size_t)m, s));
  GLOC_HIP(hipMemsetAsync(h->hist.p, 0, sizeof(uint32_t) * 18, s));
  ProfScope ps(h->prof, "ground_normals", s);
  hipLaunchKernelGGL(normals_kernel, dim3((m + 255) / 256), dim3(256), 0, s, d_pts, m, h->knn_idx.as<uint32_t>(),
                     (int)k, want_normals ? h->normals.as<float>() : (float*)nullptr, h->bins.as<uint8_t>(),
                     h->hist.as<uint32_t>());
  GLOC_HIP(hipGetLastError());
  return GLOC_OK;
}

int estimate_device(gloc_ground* h, const float* d_xyz, size_t n, size_t stride, const gloc_ground_params* p,
                    float* T16, gloc_ground_info* info, float* d_out) {
  GLOC_TRY(check_params(p));
  GLOC_REQUIRE(stride >= 3 && stride <= 16, GLOC_ERR_INVALID, "stride_floats must be in [3, 16]");
  GLOC_REQUIRE(T16, GLOC_ERR_INVALID, "T16 is NULL");
  GLOC_REQUIRE(n <= 0x7FFFFFFFu, GLOC_ERR_INVALID, "too many points");
  GLOC_HIP(hipSetDevice(h->device));
  hipStream_t s = h->stream;
  gloc_ground_info local;
  std::memset(&local, 0, sizeof(local));
  local.ground_bin = -1;
  identity16(T16);
  const uint32_t N = (uint32_t)n;
  auto finish = [&]() -> int {  // transformed cloud (identity when nothing was found) + info
    if (d_out && N) {
      float T12[12];
      for (int i = 0; i < 3; ++i) {
        for (int j = 0; j < 3; ++j) T12[3 * i + j] = T16[4 * i + j];
        T12[9 + i] = T16[4 * i + 3];
      }
      GLOC_TRY(h->T12.ensure(sizeof(T12), s));
      GLOC_HIP(hipMemcpyAsync(h->T12.p, T12, sizeof(T12), hipMemcpyHostToDevice, s));
      ProfScope ps(h->prof, "ground_transform", s);
      hipLaunchKernelGGL(transform_cloud_kernel, dim3((N + 255) / 256), dim3(256), 0, s, d_xyz, N, (int)stride,
                         h->T12.as<float>(), d_out);
      GLOC_HIP(hipGetLastError());
      GLOC_HIP(hipStreamSynchronize(s));  // T12 is a stack buffer
    }
    if (info) *info = local;
    return GLOC_OK;
  };
  if (N == 0) return finish();

  // G1: range filter, compacted in original order
  GLOC_TRY(h->flag.ensure(N, s));
  hipLaunchKernelGGL(near_flag_kernel, dim3((N + 255) / 256), dim3(256), 0, s, d_xyz, N, (int)stride,
                     p->near_range2, h->flag.as<uint8_t>());
  uint32_t m = 0;
  GLOC_TRY(select_flagged(h, N, &m));
  local.n_near = m;
  if (m < 3) return finish();
  GLOC_TRY(h->near.ensure(sizeof(f32x4) * m, s));
  hipLaunchKernelGGL(gather_points_kernel, dim3((m + 255) / 256), dim3(256), 0, s, d_xyz, (int)stride,
                     h->sel.as<uint32_t>(), m, h->near.as<f32x4>());
  // G2..G4
  GLOC_TRY(knn_device(h, h->near.as<f32x4>(), m, p->knn, std::sqrt(p->near_range2)));
  GLOC_TRY(normals_device(h, h->near.as<f32x4>(), m, p->knn, false));
  GLOC_HIP(hipMemcpyAsync(local.hist, h->hist.p, sizeof(uint32_t) * 18, hipMemcpyDeviceToHost, s));
  GLOC_HIP(hipStreamSynchronize(s));
  // G5: the fullest bin outside 5..12 (ground_estimator.cpp:104-127); ties -> the lower bin
  int gb = -1;
  for (int b = 0; b < 18; ++b) {
    if (b > 4 && b < 13) continue;
    if (gb < 0 || local.hist[b] > local.hist[gb]) gb = b;
  }
  if (gb < 0 || local.hist[gb] < 3) return finish();
  local.ground_bin = gb;
  hipLaunchKernelGGL(bin_flag_kernel, dim3((m + 255) / 256), dim3(256), 0, s, h->bins.as<uint8_t>(), m, gb,
                     h->flag.as<uint8_t>());
  uint32_t ng = 0;
  GLOC_TRY(select_flagged(h, m, &ng));
  local.n_ground = ng;
  GLOC_TRY(h->gpts.ensure(sizeof(f32x4) * std::max<uint32_t>(ng, 1), s));
  hipLaunchKernelGGL(gather_f4_kernel, dim3((ng + 255) / 256), dim3(256), 0, s, h->near.as<f32x4>(),
                     h->sel.as<uint32_t>(), ng, h->gpts.as<f32x4>());
  // G6: all hypotheses scored on the device, the sequential rule applied to the counts on the host
  const uint32_t H = p->ransac_iters;
  GLOC_TRY(h->planes.ensure(sizeof(float) * 4 * H, s));
  GLOC_TRY(h->valid.ensure(sizeof(uint32_t) * H, s));
  GLOC_TRY(h->inliers.ensure(sizeof(uint32_t) * H, s));
  GLOC_HIP(hipMemsetAsync(h->inliers.p, 0, sizeof(uint32_t) * H, s));
  {
    ProfScope ps(h->prof, "ground_plane", s);
    hipLaunchKernelGGL(plane_hyp_kernel, dim3((H + 255) / 256), dim3(256), 0, s, h->gpts.as<f32x4>(), ng, p->seed, H,
                       h->planes.as<float>(), h->valid.as<uint32_t>());
    const uint32_t slabs = std::max<uint32_t>(1, std::min<uint32_t>(64, (ng + 4095) / 4096));
    const uint32_t slab = (ng + slabs - 1) / slabs;
    hipLaunchKernelGGL(plane_score_kernel, dim3((H + 255) / 256, slabs), dim3(256), 0, s, h->gpts.as<f32x4>(), ng,
                       h->planes.as<float>(), h->valid.as<uint32_t>(), H, p->plane_thresh, slab,
                       h->inliers.as<uint32_t>());
    GLOC_HIP(hipGetLastError());
  }
  std::vector<uint32_t> inl(H), val(H);
  GLOC_HIP(hipMemcpyAsync(inl.data(), h->inliers.p, sizeof(uint32_t) * H, hipMemcpyDeviceToHost, s));
  GLOC_HIP(hipMemcpyAsync(val.data(), h->valid.p, sizeof(uint32_t) * H, hipMemcpyDeviceToHost, s));
  GLOC_HIP(hipStreamSynchronize(s));
  uint32_t best_h = 0xFFFFFFFFu, best_inl = 0, niters = H;
  for (uint32_t hh = 0; hh < niters; ++hh) {  // first strictly better hypothesis wins; adaptive stop
    if (!val[hh] || inl[hh] <= best_inl) continue;
    best_inl = inl[hh];
    best_h = hh;
    if (p->ransac_conf > 0.f && p->ransac_conf < 1.f)
      niters = std::min(niters, needed_iters(best_inl, ng, p->ransac_conf, H));
  }
  local.best_hyp = best_h;
  local.inliers = best_inl;
  local.iters_used = niters;
  if (best_h != 0xFFFFFFFFu) {
    GLOC_HIP(hipMemcpyAsync(local.plane, h->planes.as<float>() + 4 * (size_t)best_h, sizeof(float) * 4,
                            hipMemcpyDeviceToHost, s));
    GLOC_HIP(hipStreamSynchronize(s));
    transform_from_plane(local.plane, T16);
    local.found = 1;
  }
  return finish();
}

}  // namespace

extern "C" {

int gloc_ground_default_params(gloc_ground_params* p) {
  GLOC_REQUIRE(p, GLOC_ERR_INVALID, "params is NULL");
  std::memset(p, 0, sizeof(*p));
  p->near_range2 = 400.f;   // ground_estimator.cpp:203
  p->knn = 10;              // :79
  p->plane_thresh = 0.1f;   // :27
  p->ransac_iters = 1000;   // pcl::SampleConsensus default max_iterations_
  p->ransac_conf = 0.99f;   // pcl::SampleConsensus default probability_
  p->seed = 0;
  return GLOC_OK;
}

int gloc_ground_create(int device, gloc_ground** out) {
  GLOC_REQUIRE(out, GLOC_ERR_INVALID, "out is NULL");
  GLOC_TRY(select_device(device));
  gloc_ground* h = new (std::nothrow) gloc_ground();
  GLOC_REQUIRE(h, GLOC_ERR_NOMEM, "out of host memory");
  h->device = device;
  hipError_t e = hipStreamCreateWithFlags(&h->own_stream, hipStreamNonBlocking);
  if (e != hipSuccess) {
    delete h;
    set_err("hipStreamCreate failed: %s", hipGetErrorString(e));
    return GLOC_ERR_HIP;
  }
  h->stream = h->own_stream;
  *out = h;
  return GLOC_OK;
}

int gloc_ground_destroy(gloc_ground* h) {
  if (!h) return GLOC_OK;
  (void)hipSetDevice(h->device);
  (void)hipStreamSynchronize(h->stream);
  h->prof.destroy();
  for (DevBuf* b : {&h->stage_in, &h->stage_out, &h->flag, &h->sel, &h->count, &h->tile_cnt, &h->sort_segs, &h->sort_hist, &h->near, &h->knn_idx,
                    &h->knn_d2, &h->knn_pidx, &h->knn_pd2, &h->skeys, &h->svals, &h->skeys2, &h->sperm, &h->spts,
                    &h->cbox_lo, &h->cbox_hi, &h->bins, &h->hist, &h->normals, &h->gpts, &h->planes, &h->valid, &h->inliers,
                    &h->T12})
    b->release();
  if (h->own_stream) (void)hipStreamDestroy(h->own_stream);
  delete h;
  return GLOC_OK;
}

int gloc_ground_set_stream(gloc_ground* h, void* hip_stream) {
  GLOC_REQUIRE(h, GLOC_ERR_INVALID, "handle is NULL");
  GLOC_HIP(hipSetDevice(h->device));
  GLOC_HIP(hipStreamSynchronize(h->stream));
  h->stream = hip_stream ? static_cast<hipStream_t>(hip_stream) : h->own_stream;
  return GLOC_OK;
}

int gloc_ground_estimate(gloc_ground* h, const float* xyz, size_t n, size_t stride_floats,
                         const gloc_ground_params* p, float* T16, gloc_ground_info* info, float* out_xyz) {
  GLOC_REQUIRE(h && (xyz || n == 0), GLOC_ERR_INVALID, "NULL argument");
  GLOC_REQUIRE(stride_floats >= 3 && stride_floats <= 16, GLOC_ERR_INVALID, "stride_floats must be in [3, 16]");
  GLOC_HIP(hipSetDevice(h->device));
  hipStream_t s = h->stream;
  const size_t bytes = sizeof(float) * n * stride_floats;
  GLOC_TRY(h->stage_in.ensure(std::max<size_t>(bytes, 16), s));
  if (n) GLOC_HIP(hipMemcpyAsync(h->stage_in.p, xyz, bytes, hipMemcpyHostToDevice, s));
  float* d_out = nullptr;
  if (out_xyz && n) {
    GLOC_TRY(h->stage_out.ensure(bytes, s));
    d_out = h->stage_out.as<float>();
  }
  GLOC_TRY(estimate_device(h, h->stage_in.as<float>(), n, stride_floats, p, T16, info, d_out));
  if (d_out) {
    GLOC_HIP(hipMemcpyAsync(out_xyz, d_out, bytes, hipMemcpyDeviceToHost, s));
    GLOC_HIP(hipStreamSynchronize(s));
  }
  return GLOC_OK;
}

int gloc_ground_estimate_device(gloc_ground* h, const float* d_xyz, size_t n, size_t stride_floats,
                                const gloc_ground_params* p, float* T16, gloc_ground_info* info,
                                float* d_out_xyz) {
  GLOC_REQUIRE(h && (d_xyz || n == 0), GLOC_ERR_INVALID, "NULL argument");
  return estimate_device(h, d_xyz, n, stride_floats, p, T16, info, d_out_xyz);
}

int gloc_ground_knn(gloc_ground* h, const float* xyz, size_t n, uint32_t k, uint32_t* out_idx, float* out_d2) {
  GLOC_REQUIRE(h && xyz && out_idx && out_d2 && n > 0 && n <= 0x7FFFFFFFu, GLOC_ERR_INVALID, "bad argument");
  GLOC_REQUIRE(k >= 1 && k <= (uint32_t)KMAX, GLOC_ERR_INVALID, "k must be in [1, %d]", KMAX);
  GLOC_HIP(hipSetDevice(h->device));
  hipStream_t s = h->stream;
  const uint32_t m = (uint32_t)n;
  std::vector<f32x4> pts(m);
  for (uint32_t i = 0; i < m; ++i) pts[i] = f32x4{xyz[3 * (size_t)i], xyz[3 * (size_t)i + 1], xyz[3 * (size_t)i + 2], 0.f};
  GLOC_TRY(h->near.ensure(sizeof(f32x4) * m, s));
  GLOC_HIP(hipMemcpyAsync(h->near.p, pts.data(), sizeof(f32x4) * m, hipMemcpyHostToDevice, s));
  float key_range = 1.f;
  for (size_t i = 0; i < 3 * (size_t)m; ++i)
    if (std::isfinite(xyz[i])) key_range = std::max(key_range, std::fabs(xyz[i]));
  GLOC_TRY(knn_device(h, h->near.as<f32x4>(), m, k, key_range));
  GLOC_HIP(hipMemcpyAsync(out_idx, h->knn_idx.p, sizeof(uint32_t) * (size_t)m * k, hipMemcpyDeviceToHost, s));
  GLOC_HIP(hipMemcpyAsync(out_d2, h->knn_d2.p, sizeof(float) * (size_t)m * k, hipMemcpyDeviceToHost, s));
  GLOC_HIP(hipStreamSynchronize(s));
  return GLOC_OK;
}

int gloc_ground_normals(gloc_ground* h, const float* xyz, size_t n, uint32_t k, float* out_normals,
                        uint8_t* out_bins) {
  GLOC_REQUIRE(h && xyz && out_normals && out_bins && n > 0 && n <= 0x7FFFFFFFu, GLOC_ERR_INVALID, "bad argument");
  GLOC_REQUIRE(k >= 3 && k <= (uint32_t)KMAX, GLOC_ERR_INVALID, "k must be in [3, %d]", KMAX);
  GLOC_HIP(hipSetDevice(h->device));
  hipStream_t s = h->stream;
  const uint32_t m = (uint32_t)n;
  std::vector<f32x4> pts(m);
  for (uint32_t i = 0; i < m; ++i) pts[i] = f32x4{xyz[3 * (size_t)i], xyz[3 * (size_t)i + 1], xyz[3 * (size_t)i + 2], 0.f};
  GLOC_TRY(h->near.ensure(sizeof(f32x4) * m, s));
  GLOC_HIP(hipMemcpyAsync(h->near.p, pts.data(), sizeof(f32x4) * m, hipMemcpyHostToDevice, s));
  float key_range = 1.f;
  for (size_t i = 0; i < 3 * (size_t)m; ++i)
    if (std::isfinite(xyz[i])) key_range = std::max(key_range, std::fabs(xyz[i]));
  GLOC_TRY(knn_device(h, h->near.as<f32x4>(), m, k, key_range));
  GLOC_TRY(normals_device(h, h->near.as<f32x4>(), m, k, true));
  GLOC_HIP(hipMemcpyAsync(out_normals, h->normals.p, sizeof(float) * 3 * (size_t)m, hipMemcpyDeviceToHost, s));
  GLOC_HIP(hipMemcpyAsync(out_bins, h->bins.p, m, hipMemcpyDeviceToHost, s));
  GLOC_HIP(hipStreamSynchronize(s));
  return GLOC_OK;
}

int gloc_ground_transform_from_plane(const float* plane4, float* T16) {
  GLOC_REQUIRE(plane4 && T16, GLOC_ERR_INVALID, "NULL argument");
  const double l2 = (double)plane4[0] * plane4[0] + (double)plane4[1] * plane4[1] + (double)plane4[2] * plane4[2];
  GLOC_REQUIRE(l2 > 0.0, GLOC_ERR_INVALID, "zero plane normal");
  transform_from_plane(plane4, T16);
  return GLOC_OK;
}

int gloc_ground_set_option(gloc_ground* h, int option, int64_t value) {
  GLOC_REQUIRE(h, GLOC_ERR_INVALID, "handle is NULL");
  if (option == GLOC_GROUND_OPT_KNN_EXHAUSTIVE) {
    h->knn_exhaustive = value != 0;
    return GLOC_OK;
  }
  set_err("unknown option %d", option);
  return GLOC_ERR_INVALID;
}

int gloc_ground_set_profile(gloc_ground* h, int enable) {
  GLOC_REQUIRE(h, GLOC_ERR_INVALID, "handle is NULL");
  h->prof.enabled = enable != 0;
  return GLOC_OK;
}

int gloc_ground_profile(gloc_ground* h, const char* kernel, double* total_ms, uint64_t* launches) {
  GLOC_REQUIRE(h && kernel, GLOC_ERR_INVALID, "NULL argument");
  GLOC_HIP(hipSetDevice(h->device));
  GLOC_TRY(h->prof.collect(h->stream));
  auto it = h->prof.fam.find(kernel);
  if (total_ms) *total_ms = it == h->prof.fam.end() ? 0.0 : it->second.total_ms;
  if (launches) *launches = it == h->prof.fam.end() ? 0 : it->second.launches;
  return GLOC_OK;
}

}  // extern "C"
