// bev.hip -- C ABI of the BEV occupancy projection (include/gloc3d.h, "next" row N1 of SURVEY.md 8f).
// Replaces RpyPCLoopDetector::get_projected_grid / crop_pad_occupancy and the tensor packing of
// get_place_feature (registration/loop_detector.cpp:83-106,122-151).
#include <algorithm>
#include <cmath>
#include <new>

#include "common.hpp"
#include "bev_kernels.hpp"

using namespace gloc;
using namespace gloc::bev;

struct gloc_bev {
  int device = 0;
  hipStream_t own_stream = nullptr, stream = nullptr;
  DevBuf zcol, multi, meta, offsets;   // per column [n_scans][S][S]: one z index (u32), span flag (u8)
  DevBuf stage_in, stage_out, raw;     // host-pointer API staging
  std::vector<ScanMeta> h_meta;        // last projection, host copy (filled when infos were requested)
  bool meta_on_host = false;
  size_t last_scans = 0;
  int last_R = 0, last_S = 0;
  Profiler prof;
};

namespace {

int check_params(const gloc_bev_params* p, int* R) {
  GLOC_REQUIRE(p, GLOC_ERR_INVALID, "params is NULL");
  GLOC_REQUIRE(p->resolution > 0.f && std::isfinite(p->resolution), GLOC_ERR_INVALID,
               "resolution must be positive");
  GLOC_REQUIRE(p->max_range > 0.f && std::isfinite(p->max_range), GLOC_ERR_INVALID,
               "max_range must be positive");
  const double cells = std::ceil((double)p->max_range / (double)p->resolution);
  // the reference's grid is limited to +-8192 cells (3d/hybrid_grid.h:468); half of that keeps the
  // column tables (5 B x (2R+1)^2 per scan) under 340 MB
  GLOC_REQUIRE(cells <= 4094.0, GLOC_ERR_INVALID, "max_range / resolution = %.0f exceeds 4094 cells", cells);
  GLOC_REQUIRE(p->out_width > 0 && p->out_height > 0 && p->out_width <= 16384 && p->out_height <= 16384,
               GLOC_ERR_INVALID, "output size %ux%u out of range", p->out_width, p->out_height);
  GLOC_REQUIRE(p->format == GLOC_BEV_U8_HWC3 || p->format == GLOC_BEV_F32_CHW, GLOC_ERR_INVALID,
               "unknown image format %u", p->format);
  *R = (int)cells + 2;
  return GLOC_OK;
}

size_t image_bytes(const gloc_bev_params* p) {
  return (size_t)p->out_width * p->out_height * 3 * (p->format == GLOC_BEV_F32_CHW ? 4 : 1);
}

void fill_info(const ScanMeta& m, const gloc_bev_params* p, gloc_bev_info* info) {
  std::memset(info, 0, sizeof(*info));
  info->resolution = (double)p->resolution;  // hybrid_grid->resolution() widened, submap_3d.cpp:241
  info->n_returns = m.n_returns;
  if (m.min_ix > m.max_ix) {
    info->empty = 1;
    return;
  }
  info->min_ix = m.min_ix; info->min_iy = m.min_iy; info->max_ix = m.max_ix; info->max_iy = m.max_iy;
  info->width = (uint32_t)(m.max_ix - m.min_ix + 1);
  info->height = (uint32_t)(m.max_iy - m.min_iy + 1);
  info->ox = m.min_ix * info->resolution;  // submap_3d.cpp:281-282
  info->oy = m.min_iy * info->resolution;
}

int project_device(gloc_bev* h, const float* d_xyz, const uint64_t* offsets, size_t n_scans,
                   size_t stride, const gloc_bev_params* p, void* d_out, gloc_bev_info* infos) {
  int R = 0;
  GLOC_TRY(check_params(p, &R));
  GLOC_REQUIRE(stride >= 3 && stride <= 64, GLOC_ERR_INVALID, "stride_floats must be in [3, 64]");
  GLOC_REQUIRE(n_scans > 0 && n_scans <= 65535, GLOC_ERR_INVALID, "n_scans must be in [1, 65535]");
  GLOC_REQUIRE(offsets && d_out, GLOC_ERR_INVALID, "NULL argument");
  uint64_t max_n = 0;
  for (size_t i = 0; i < n_scans; ++i) {
    GLOC_REQUIRE(offsets[i + 1] >= offsets[i], GLOC_ERR_INVALID, "offsets must be non-decreasing");
    max_n = std::max<uint64_t>(max_n, offsets[i + 1] - offsets[i]);
  }
  GLOC_REQUIRE(offsets[n_scans] == 0 || d_xyz, GLOC_ERR_INVALID, "points pointer is NULL");
  GLOC_HIP(hipSetDevice(h->device));
  hipStream_t s = h->stream;
  const int S = 2 * R + 1;
  const size_t ncols = (size_t)n_scans * S * S, bytes16 = (ncols + 15) / 16;
  GLOC_TRY(h->zcol.ensure(ncols * 4, s));
  GLOC_TRY(h->multi.ensure(bytes16 * 16, s));
  GLOC_TRY(h->meta.ensure(sizeof(ScanMeta) * n_scans, s));
  GLOC_TRY(h->offsets.ensure(sizeof(uint64_t) * (n_scans + 1), s));
  GLOC_HIP(hipMemcpyAsync(h->offsets.p, offsets, sizeof(uint64_t) * (n_scans + 1), hipMemcpyHostToDevice, s));
  h->last_scans = n_scans; h->last_R = R; h->last_S = S; h->meta_on_host = false;
  {
    ProfScope ps(h->prof, "bev_clear", s);
    hipLaunchKernelGGL(bev_clear_kernel, dim3((unsigned)((std::max<size_t>(bytes16, n_scans) + 255) / 256)),
                       dim3(256), 0, s, h->multi.as<uint4>(), bytes16, h->meta.as<ScanMeta>(), (int)n_scans);
    GLOC_HIP(hipGetLastError());
  }
  if (max_n) {
    const dim3 grid((unsigned)((max_n + 256 * PTS_PER_THREAD - 1) / (256 * PTS_PER_THREAD)), (unsigned)n_scans);
    {
      ProfScope ps(h->prof, "bev_mark", s);
      hipLaunchKernelGGL(bev_points_kernel<0>, grid, dim3(256), 0, s, d_xyz, h->offsets.as<uint64_t>(),
                         (int)stride, p->resolution, p->max_range, (float)(int)p->max_range, R, S,
                         h->zcol.as<uint32_t>(), h->multi.as<uint8_t>(), h->meta.as<ScanMeta>());
    }
    {
      ProfScope ps(h->prof, "bev_flag", s);
      hipLaunchKernelGGL(bev_points_kernel<1>, grid, dim3(256), 0, s, d_xyz, h->offsets.as<uint64_t>(),
                         (int)stride, p->resolution, p->max_range, (float)(int)p->max_range, R, S,
                         h->zcol.as<uint32_t>(), h->multi.as<uint8_t>(), h->meta.as<ScanMeta>());
    }
    GLOC_HIP(hipGetLastError());
  }
  {
    ProfScope ps(h->prof, "bev_image", s);
    const uint32_t pad = (uint32_t)p->pad_bgr[0] | ((uint32_t)p->pad_bgr[1] << 8) | ((uint32_t)p->pad_bgr[2] << 16);
    const size_t quads = (size_t)((p->out_width + 3) / 4) * p->out_height;
    const dim3 grid((unsigned)((quads + 255) / 256), (unsigned)n_scans);
    if (p->format == GLOC_BEV_U8_HWC3)
      hipLaunchKernelGGL(bev_image_kernel<GLOC_BEV_U8_HWC3>, grid, dim3(256), 0, s, h->multi.as<uint8_t>(),
                         h->meta.as<ScanMeta>(), R, S, (int)p->out_width, (int)p->out_height, pad, d_out);
    else
      hipLaunchKernelGGL(bev_image_kernel<GLOC_BEV_F32_CHW>, grid, dim3(256), 0, s, h->multi.as<uint8_t>(),
                         h->meta.as<ScanMeta>(), R, S, (int)p->out_width, (int)p->out_height, pad, d_out);
    GLOC_HIP(hipGetLastError());
  }
  if (infos) {
    h->h_meta.resize(n_scans);
    GLOC_HIP(hipMemcpyAsync(h->h_meta.data(), h->meta.p, sizeof(ScanMeta) * n_scans, hipMemcpyDeviceToHost, s));
    GLOC_HIP(hipStreamSynchronize(s));
    h->meta_on_host = true;
    for (size_t i = 0; i < n_scans; ++i) fill_info(h->h_meta[i], p, &infos[i]);
  }
  return GLOC_OK;
}

}  // namespace

extern "C" {

int gloc_bev_default_params(gloc_bev_params* p) {
  GLOC_REQUIRE(p, GLOC_ERR_INVALID, "params is NULL");
  std::memset(p, 0, sizeof(*p));
  p->resolution = 0.2f;   // loop_detector.h:116
  p->max_range = 100.f;   // loop_detector.h:115, loop_detector.cpp:113
  p->out_width = 768;     // loop_detector.cpp:142-143
  p->out_height = 768;
  p->format = GLOC_BEV_U8_HWC3;
  p->pad_bgr[0] = 255;    // cv::Mat::ones(h, w, CV_8UC3) * 255: channel 0 only (loop_detector.cpp:84)
  return GLOC_OK;
}

int gloc_bev_create(int device, gloc_bev** out) {
  GLOC_REQUIRE(out, GLOC_ERR_INVALID, "out is NULL");
  GLOC_TRY(select_device(device));
  gloc_bev* h = new (std::nothrow) gloc_bev();
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

int gloc_bev_destroy(gloc_bev* h) {
  if (!h) return GLOC_OK;
  (void)hipSetDevice(h->device);
  (void)hipStreamSynchronize(h->stream);
  h->prof.destroy();
  for (DevBuf* b : {&h->zcol, &h->multi, &h->meta, &h->offsets, &h->stage_in, &h->stage_out, &h->raw}) b->release();
  if (h->own_stream) (void)hipStreamDestroy(h->own_stream);
  delete h;
  return GLOC_OK;
}

int gloc_bev_set_stream(gloc_bev* h, void* hip_stream) {
  GLOC_REQUIRE(h, GLOC_ERR_INVALID, "handle is NULL");
  GLOC_HIP(hipSetDevice(h->device));
  GLOC_HIP(hipStreamSynchronize(h->stream));
  h->stream = hip_stream ? static_cast<hipStream_t>(hip_stream) : h->own_stream;
  return GLOC_OK;
}

int gloc_bev_synchronize(gloc_bev* h) {
  GLOC_REQUIRE(h, GLOC_ERR_INVALID, "handle is NULL");
  GLOC_HIP(hipSetDevice(h->device));
  GLOC_HIP(hipStreamSynchronize(h->stream));
  return GLOC_OK;
}

int gloc_bev_project(gloc_bev* h, const float* xyz, size_t n, size_t stride_floats,
                     const gloc_bev_params* p, void* out_image, gloc_bev_info* info) {
  GLOC_REQUIRE(h && p && out_image, GLOC_ERR_INVALID, "NULL argument");
  GLOC_REQUIRE(n == 0 || xyz, GLOC_ERR_INVALID, "points pointer is NULL");
  int R = 0;
  GLOC_TRY(check_params(p, &R));
  GLOC_REQUIRE(stride_floats >= 3 && stride_floats <= 64, GLOC_ERR_INVALID, "stride_floats must be in [3, 64]");
  GLOC_HIP(hipSetDevice(h->device));
  hipStream_t s = h->stream;
  const size_t in_bytes = sizeof(float) * n * stride_floats, out_bytes = image_bytes(p);
  GLOC_TRY(h->stage_in.ensure(std::max<size_t>(in_bytes, 16), s));
  GLOC_TRY(h->stage_out.ensure(out_bytes, s));
  if (n) GLOC_HIP(hipMemcpyAsync(h->stage_in.p, xyz, in_bytes, hipMemcpyHostToDevice, s));
  const uint64_t offsets[2] = {0, (uint64_t)n};
  gloc_bev_info local;
  GLOC_TRY(project_device(h, h->stage_in.as<float>(), offsets, 1, stride_floats, p, h->stage_out.p, &local));
  GLOC_HIP(hipMemcpyAsync(out_image, h->stage_out.p, out_bytes, hipMemcpyDeviceToHost, s));
  GLOC_HIP(hipStreamSynchronize(s));
  if (info) *info = local;
  return GLOC_OK;
}

int gloc_bev_project_batch_device(gloc_bev* h, const float* d_xyz, const uint64_t* offsets,
                                  size_t n_scans, size_t stride_floats, const gloc_bev_params* p,
                                  void* d_out_images, gloc_bev_info* infos) {
  GLOC_REQUIRE(h, GLOC_ERR_INVALID, "handle is NULL");
  return project_device(h, d_xyz, offsets, n_scans, stride_floats, p, d_out_images, infos);
}

int gloc_bev_raw_image(gloc_bev* h, size_t scan, uint8_t* out, size_t capacity) {
  GLOC_REQUIRE(h && out, GLOC_ERR_INVALID, "NULL argument");
  GLOC_REQUIRE(scan < h->last_scans, GLOC_ERR_STATE, "scan %zu is not part of the last projection (%zu scans)",
               scan, h->last_scans);
  GLOC_HIP(hipSetDevice(h->device));
  hipStream_t s = h->stream;
  if (!h->meta_on_host) {
    h->h_meta.resize(h->last_scans);
    GLOC_HIP(hipMemcpyAsync(h->h_meta.data(), h->meta.p, sizeof(ScanMeta) * h->last_scans,
                            hipMemcpyDeviceToHost, s));
    GLOC_HIP(hipStreamSynchronize(s));
    h->meta_on_host = true;
  }
  const ScanMeta& m = h->h_meta[scan];
  GLOC_REQUIRE(m.min_ix <= m.max_ix, GLOC_ERR_STATE, "scan %zu projected to an empty image", scan);
  const int w = m.max_ix - m.min_ix + 1, ht = m.max_iy - m.min_iy + 1;
  const size_t bytes = (size_t)w * ht;
  GLOC_REQUIRE(capacity >= bytes, GLOC_ERR_INVALID, "raw image is %dx%d = %zu bytes, capacity %zu", w, ht,
               bytes, capacity);
  GLOC_TRY(h->raw.ensure(bytes, s));
  hipLaunchKernelGGL(bev_raw_kernel, dim3((unsigned)((bytes + 255) / 256)), dim3(256), 0, s,
                     h->multi.as<uint8_t>() + scan * (size_t)h->last_S * h->last_S, m.min_ix, m.min_iy, w, ht,
                     h->last_R, h->last_S, h->raw.as<uint8_t>());
  GLOC_HIP(hipGetLastError());
  GLOC_HIP(hipMemcpyAsync(out, h->raw.p, bytes, hipMemcpyDeviceToHost, s));
  GLOC_HIP(hipStreamSynchronize(s));
  return GLOC_OK;
}

int gloc_bev_device_flags(gloc_bev* h, size_t scan, const uint8_t** d_flags, int* R, int* S) {
  GLOC_REQUIRE(h && d_flags && R && S, GLOC_ERR_INVALID, "NULL argument");
  GLOC_REQUIRE(scan < h->last_scans, GLOC_ERR_STATE, "scan %zu is not part of the last projection (%zu scans)",
               scan, h->last_scans);
  *d_flags = h->multi.as<uint8_t>() + scan * (size_t)h->last_S * h->last_S;
  *R = h->last_R;
  *S = h->last_S;
  return GLOC_OK;
}

int gloc_bev_set_profile(gloc_bev* h, int enable) {
  GLOC_REQUIRE(h, GLOC_ERR_INVALID, "handle is NULL");
  h->prof.enabled = enable != 0;
  return GLOC_OK;
}

int gloc_bev_profile(gloc_bev* h, const char* kernel, double* total_ms, uint64_t* launches) {
  GLOC_REQUIRE(h && kernel, GLOC_ERR_INVALID, "NULL argument");
  GLOC_HIP(hipSetDevice(h->device));
  GLOC_TRY(h->prof.collect(h->stream));
  auto it = h->prof.fam.find(kernel);
  if (total_ms) *total_ms = it == h->prof.fam.end() ? 0.0 : it->second.total_ms;
  if (launches) *launches = it == h->prof.fam.end() ? 0 : it->second.launches;
  return GLOC_OK;
}

}  // extern "C"
