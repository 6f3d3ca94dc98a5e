// vlad.hip -- C ABI of the NetVLAD-FC pooling head (include/gloc3d.h, "next" row N2 of SURVEY.md 8f).
// Replaces NetVLAD.forward (model/netvlad_fc.py:73-109), i.e. the tail of the TorchScript module the
// reference runs in RpyPCLoopDetector::get_place_feature (registration/loop_detector.cpp:152-163).
#include <algorithm>
#include <new>

#include "common.hpp"
#include "vlad_kernels.hpp"

using namespace gloc;
using namespace gloc::vlad;

struct gloc_vlad {
  int device = 0;
  size_t C = 0, K = 0, Kp = 0, out_dim = 0;
  int normalize_input = 1;
  bool has_bias = false;
  hipStream_t own_stream = nullptr, stream = nullptr;
  DevBuf conv_w, conv_b, centroids, fc_w;       // parameters, resident
  DevBuf gate_w, gate_scale, gate_shift, gate_tmp;  // optional GatingContext
  bool gating = false;
  DevBuf partV, partS, vlad, nrm2, fc_part;     // workspace
  DevBuf stage_in, stage_out;                   // host-pointer API staging
  Profiler prof;
};

namespace {

int forward_device(gloc_vlad* h, const float* d_feat, size_t n, size_t hw, float* d_out) {
  hipStream_t s = h->stream;
  const int C = (int)h->C, K = (int)h->K, Kp = (int)h->Kp, HW = (int)hw, OD = (int)h->out_dim;
  const int ntiles = (HW + VP - 1) / VP;
  const size_t KC = (size_t)K * C;
  GLOC_TRY(h->partV.ensure(sizeof(float) * n * ntiles * Kp * C, s));
  GLOC_TRY(h->partS.ensure(sizeof(float) * n * ntiles * Kp, s));
  GLOC_TRY(h->vlad.ensure(sizeof(float) * n * KC, s));
  GLOC_TRY(h->nrm2.ensure(sizeof(float) * n * K, s));
  const int nslabs = (int)((KC + FC_ROWS - 1) / FC_ROWS);
  GLOC_TRY(h->fc_part.ensure(sizeof(float) * (size_t)nslabs * FC_NB * OD, s));
  const size_t lds = sizeof(float) * ((size_t)C * VPITCH + (size_t)Kp * VPITCH + 2 * 4 * VP + VP);
  GLOC_REQUIRE(lds <= 160 * 1024, GLOC_ERR_INVALID, "feature dim %d needs %zu B of LDS (> 160 KiB)", C, lds);
  GLOC_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(vlad_tile_kernel),
                               hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  {
    ProfScope ps(h->prof, "vlad_tile", s);
    hipLaunchKernelGGL(vlad_tile_kernel, dim3(ntiles, (unsigned)n), dim3(256), lds, s, d_feat, C, HW, K,
                       Kp, h->conv_w.as<float>(), h->has_bias ? h->conv_b.as<float>() : (const float*)nullptr,
                       h->normalize_input, h->partV.as<float>(), h->partS.as<float>());
    GLOC_HIP(hipGetLastError());
  }
  {
    ProfScope ps(h->prof, "vlad_cluster", s);
    hipLaunchKernelGGL(vlad_cluster_kernel, dim3(K, (unsigned)n), dim3(256), 0, s, h->partV.as<float>(),
                       h->partS.as<float>(), ntiles, C, K, Kp, h->centroids.as<float>(),
                       h->vlad.as<float>(), h->nrm2.as<float>());
    GLOC_HIP(hipGetLastError());
  }
  {
    ProfScope ps(h->prof, "vlad_fc", s);
    for (size_t n0 = 0; n0 < n; n0 += FC_NB) {
      const int nb = (int)std::min<size_t>(FC_NB, n - n0);
      hipLaunchKernelGGL(vlad_fc_kernel, dim3(nslabs, (OD + 255) / 256), dim3(256), 0, s,
                         h->vlad.as<float>(), (int)n0, nb, (int)KC, OD, h->fc_w.as<float>(),
                         h->fc_part.as<float>());
      hipLaunchKernelGGL(vlad_fc_reduce_kernel, dim3((OD + 255) / 256, nb), dim3(256), 0, s,
                         h->fc_part.as<float>(), nslabs, (int)n0, nb, OD, h->nrm2.as<float>(), K, d_out);
    }
    GLOC_HIP(hipGetLastError());
  }
  if (h->gating) {
    ProfScope ps(h->prof, "vlad_gate", s);
    GLOC_TRY(h->gate_tmp.ensure(sizeof(float) * n * OD, s));
    hipLaunchKernelGGL(vlad_gate_kernel, dim3((OD + 63) / 64, (unsigned)n), dim3(64), sizeof(float) * OD, s, d_out,
                       (int)n, OD, h->gate_w.as<float>(), h->gate_scale.as<float>(), h->gate_shift.as<float>(),
                       h->gate_tmp.as<float>());
    GLOC_HIP(hipGetLastError());
    GLOC_HIP(hipMemcpyAsync(d_out, h->gate_tmp.p, sizeof(float) * n * OD, hipMemcpyDeviceToDevice, s));
  }
  return GLOC_OK;
}

int upload(gloc_vlad* h, DevBuf& b, const float* src, size_t count) {
  GLOC_TRY(b.ensure(sizeof(float) * count, h->stream));
  GLOC_HIP(hipMemcpyAsync(b.p, src, sizeof(float) * count, hipMemcpyHostToDevice, h->stream));
  return GLOC_OK;
}

}  // namespace

extern "C" {

int gloc_vlad_create(int device, size_t dim, size_t clusters, size_t out_dim, const float* conv_w,
                     const float* conv_b, const float* centroids, const float* fc_w,
                     int normalize_input, gloc_vlad** out) {
  GLOC_REQUIRE(out && conv_w && centroids && fc_w, GLOC_ERR_INVALID, "null argument");
  *out = nullptr;
  GLOC_REQUIRE(dim >= 1 && dim <= 560 && clusters >= 1 && clusters <= 64 && out_dim >= 1 &&
                   out_dim <= 65536,
               GLOC_ERR_INVALID, "need dim <= 560 (LDS tile), clusters <= 64, out_dim <= 65536");
  GLOC_TRY(select_device(device));
  gloc_vlad* h = new (std::nothrow) gloc_vlad;
  GLOC_REQUIRE(h, GLOC_ERR_NOMEM, "host allocation failed");
  h->device = device;
  h->C = dim;
  h->K = clusters;
  h->Kp = (clusters + 15) / 16 * 16;
  h->out_dim = out_dim;
  h->normalize_input = normalize_input;
  h->has_bias = conv_b != nullptr;
  if (hipStreamCreateWithFlags(&h->own_stream, hipStreamNonBlocking) != hipSuccess) {
    set_err("hipStreamCreate failed");
    delete h;
    return GLOC_ERR_HIP;
  }
  h->stream = h->own_stream;
  int rc = upload(h, h->conv_w, conv_w, clusters * dim);
  if (rc == GLOC_OK && conv_b) rc = upload(h, h->conv_b, conv_b, clusters);
  if (rc == GLOC_OK) rc = upload(h, h->centroids, centroids, clusters * dim);
  if (rc == GLOC_OK) rc = upload(h, h->fc_w, fc_w, clusters * dim * out_dim);
  if (rc == GLOC_OK && hipStreamSynchronize(h->stream) != hipSuccess) rc = GLOC_ERR_HIP;
  if (rc != GLOC_OK) {
    gloc_vlad_destroy(h);
    return rc;
  }
  *out = h;
  return GLOC_OK;
}

int gloc_vlad_set_gating(gloc_vlad* h, const float* gating_w, const float* scale, const float* shift) {
  GLOC_REQUIRE(h, GLOC_ERR_INVALID, "null handle");
  GLOC_HIP(hipSetDevice(h->device));
  GLOC_HIP(hipStreamSynchronize(h->stream));
  if (!gating_w) {
    h->gating = false;
    return GLOC_OK;
  }
  GLOC_REQUIRE(scale && shift, GLOC_ERR_INVALID, "scale and shift are required with gating weights");
  GLOC_REQUIRE(h->out_dim <= 16384, GLOC_ERR_INVALID, "gating needs out_dim <= 16384");
  GLOC_TRY(upload(h, h->gate_w, gating_w, h->out_dim * h->out_dim));
  GLOC_TRY(upload(h, h->gate_scale, scale, h->out_dim));
  GLOC_TRY(upload(h, h->gate_shift, shift, h->out_dim));
  GLOC_HIP(hipStreamSynchronize(h->stream));
  h->gating = true;
  return GLOC_OK;
}

int gloc_vlad_destroy(gloc_vlad* h) {
  if (!h) return GLOC_OK;
  (void)hipSetDevice(h->device);
  (void)hipStreamSynchronize(h->stream);
  h->prof.destroy();
  for (DevBuf* b : {&h->gate_w, &h->gate_scale, &h->gate_shift, &h->gate_tmp, &h->conv_w, &h->conv_b, &h->centroids, &h->fc_w, &h->partV, &h->partS, &h->vlad,
                    &h->nrm2, &h->fc_part, &h->stage_in, &h->stage_out})
    b->release();
  if (h->own_stream) (void)hipStreamDestroy(h->own_stream);
  delete h;
  return GLOC_OK;
}

int gloc_vlad_set_stream(gloc_vlad* h, void* hip_stream) {
  GLOC_REQUIRE(h, GLOC_ERR_INVALID, "null handle");
  GLOC_HIP(hipSetDevice(h->device));
  GLOC_HIP(hipStreamSynchronize(h->stream));
  h->stream = hip_stream ? (hipStream_t)hip_stream : h->own_stream;
  return GLOC_OK;
}

int gloc_vlad_forward_device(gloc_vlad* h, const float* d_feat, size_t n, size_t hw, float* d_out) {
  GLOC_REQUIRE(h && d_feat && d_out, GLOC_ERR_INVALID, "null argument");
  GLOC_REQUIRE(n >= 1 && n <= (1u << 20) && hw >= 1 && hw <= (1u << 24), GLOC_ERR_INVALID, "bad sizes");
  GLOC_HIP(hipSetDevice(h->device));
  return forward_device(h, d_feat, n, hw, d_out);
}

int gloc_vlad_forward(gloc_vlad* h, const float* feat, size_t n, size_t hw, float* out) {
  GLOC_REQUIRE(h && feat && out, GLOC_ERR_INVALID, "null argument");
  GLOC_REQUIRE(n >= 1 && n <= (1u << 20) && hw >= 1 && hw <= (1u << 24), GLOC_ERR_INVALID, "bad sizes");
  GLOC_HIP(hipSetDevice(h->device));
  const size_t in_count = n * h->C * hw;
  GLOC_TRY(h->stage_in.ensure(sizeof(float) * in_count, h->stream));
  GLOC_TRY(h->stage_out.ensure(sizeof(float) * n * h->out_dim, h->stream));
  GLOC_HIP(hipMemcpyAsync(h->stage_in.p, feat, sizeof(float) * in_count, hipMemcpyHostToDevice, h->stream));
  GLOC_TRY(forward_device(h, h->stage_in.as<float>(), n, hw, h->stage_out.as<float>()));
  GLOC_HIP(hipMemcpyAsync(out, h->stage_out.p, sizeof(float) * n * h->out_dim, hipMemcpyDeviceToHost, h->stream));
  GLOC_HIP(hipStreamSynchronize(h->stream));
  return GLOC_OK;
}

int gloc_vlad_set_profile(gloc_vlad* h, int enable) {
  GLOC_REQUIRE(h, GLOC_ERR_INVALID, "null handle");
  h->prof.enabled = enable != 0;
  return GLOC_OK;
}

int gloc_vlad_profile(gloc_vlad* h, const char* kernel, double* total_ms, uint64_t* launches) {
  GLOC_REQUIRE(h && kernel, GLOC_ERR_INVALID, "null argument");
  GLOC_HIP(hipSetDevice(h->device));
  GLOC_TRY(h->prof.collect(h->stream));
  auto it = h->prof.fam.find(kernel);
  if (total_ms) *total_ms = it == h->prof.fam.end() ? 0.0 : it->second.total_ms;
  if (launches) *launches = it == h->prof.fam.end() ? 0 : it->second.launches;
  return GLOC_OK;
}

}  // extern "C"
