// reg.hip -- placeholder; replaced by the registration kernels (K4-K6).
#include "common.hpp"
#define NI { gloc::set_err("registration not built yet"); return GLOC_ERR_STATE; }
extern "C" {
void gloc_reg_default_params(gloc_reg_params* p) {
  if (!p) return;
  p->ransac_iters = 3000; p->inlier_thresh = 0.6f; p->min_inlier_ratio = 0.3f;
  p->icp_iters = 30; p->max_corr_dist = 0.f; p->seed = 1234;
}
int gloc_reg_create(int, gloc_reg**) NI
int gloc_reg_destroy(gloc_reg*) NI
int gloc_reg_set_stream(gloc_reg*, void*) NI
int gloc_reg_synchronize(gloc_reg*) NI
int gloc_reg_set_option(gloc_reg*, int, int64_t) NI
int gloc_reg_scan_upload(gloc_reg*, const float*, size_t, size_t, uint32_t*) NI
int gloc_reg_scan_count(const gloc_reg*, size_t*) NI
int gloc_reg_scan_clear(gloc_reg*) NI
int gloc_reg_batch(gloc_reg*, const float*, size_t, const float* const*, const size_t*, size_t, const float*, const gloc_reg_params*, float*, float*, uint32_t*, int*) NI
int gloc_reg_batch_ids(gloc_reg*, uint32_t, const uint32_t*, size_t, const float*, const gloc_reg_params*, float*, float*, uint32_t*, int*) NI
int gloc_reg_select_first_ok(const int* ok, size_t n) { for (size_t i = 0; i < n; ++i) if (ok[i]) return (int)i; return -1; }
int gloc_reg_nn(gloc_reg*, const float*, size_t, const float*, size_t, const float*, uint32_t*, float*) NI
int gloc_reg_ransac_hypotheses(gloc_reg*, const float*, const float*, const uint32_t*, size_t, uint64_t, uint32_t, uint32_t, float*, uint32_t*, uint32_t*, float) NI
int gloc_reg_profile(gloc_reg*, const char*, double*, uint64_t*) NI
int gloc_reg_profile_reset(gloc_reg*) NI
}
