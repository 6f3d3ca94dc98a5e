/*
 * bev_oracle.c -- CPU restatement of the BEV occupancy projection ("next" row N1).
 *
 * TEST INFRASTRUCTURE ONLY (see gloc_oracle.h).  PARITY UNPINNED: the reference's implementation
 * needs Eigen, glog, OpenCV and PCL, none of which are in this image, and the reference holds no
 * fixture for this path; the restatement below follows the reference's source step by step so that
 * the algebraic shortcut the HIP kernels take (see gloc3d_amd/csrc/bev_kernels.hpp) is checked
 * against the long form and not against itself.
 *
 * Path restated (all under /root/reference/registration):
 *   RpyPCLoopDetector::get_projected_grid            loop_detector.cpp:122-135
 *     point_cloud_to_range_data                      loop_detector.cpp:108-120
 *     Submap3D::InsertRangeData                      3d/submap_3d.cpp:162-177
 *       FilterRangeDataByMaxRange                    3d/submap_3d.cpp:43-52
 *       RangeDataInserter3D::Insert                  3d/range_data_inserter_3d.cpp:63-78
 *         InsertMissesIntoGrid                       3d/range_data_inserter_3d.cpp:27-52
 *         HybridGrid::ApplyLookupTable/FinishUpdate  3d/hybrid_grid.h:491-519
 *     ProjectToCvMat                                 3d/submap_3d.cpp:238-326
 *   RpyPCLoopDetector::crop_pad_occupancy            loop_detector.cpp:83-106
 *   probability <-> uint16 tables                    3d/probability_values.{h,cpp}
 */
#include "gloc_oracle.h"

#include <limits.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>

/* ---- 3d/probability_values.h:30-47, 3d/probability_values.cpp:25-33,67-79 -------------------- */

#define K_UPDATE_MARKER 32768u
static const float kMinProbability = 0.1f;

static float clampf(float v, float lo, float hi) { return v < lo ? lo : (v > hi ? hi : v); }

static uint16_t probability_to_value(float p) {
  const float kMaxProbability = 1.f - kMinProbability;
  const int value =
      (int)lroundf((clampf(p, kMinProbability, kMaxProbability) - kMinProbability) *
                   (32766.f / (kMaxProbability - kMinProbability))) + 1;
  return (uint16_t)value;
}

static float slow_value_to_probability(unsigned value) {
  const float kMaxProbability = 1.f - kMinProbability;
  if (value == 0) return kMinProbability; /* unknown */
  const float kScale = (kMaxProbability - kMinProbability) / 32766.f;
  return (float)value * kScale + (kMinProbability - kScale);
}

static float odds(float p) { return p / (1.f - p); }
static float probability_from_odds(float o) { return o / (o + 1.f); }

typedef struct {
  float value_to_probability[65536];
  uint16_t hit_table[32768], miss_table[32768];
} bev_tables;

static void compute_lookup_table(const bev_tables* t, float o, uint16_t* out) {
  out[0] = (uint16_t)(probability_to_value(probability_from_odds(o)) + K_UPDATE_MARKER);
  for (int cell = 1; cell != 32768; ++cell)
    out[cell] = (uint16_t)(probability_to_value(
                               probability_from_odds(o * odds(t->value_to_probability[cell]))) +
                           K_UPDATE_MARKER);
}

static bev_tables* make_tables(void) {
  bev_tables* t = (bev_tables*)malloc(sizeof(bev_tables));
  for (int repeat = 0; repeat != 2; ++repeat)
    for (unsigned v = 0; v != 32768; ++v)
      t->value_to_probability[repeat * 32768 + v] = slow_value_to_probability(v);
  compute_lookup_table(t, odds(0.55f), t->hit_table);   /* range_data_inserter_3d.cpp:57-61 */
  compute_lookup_table(t, odds(0.49f), t->miss_table);
  return t;
}

/* ---- the voxel store: stands in for HybridGrid (3d/hybrid_grid.h), whose nested/dynamic layout
 * only affects iteration order; nothing below depends on that order. ----------------------------- */

typedef struct { int32_t x, y, z; uint16_t value; uint8_t used; } cell_t;
typedef struct {
  cell_t* slots; size_t cap, count;
  size_t* order; /* insertion order, for iteration */
  size_t* updated; size_t n_updated;
} grid_t;

static uint64_t cell_hash(int32_t x, int32_t y, int32_t z) {
  uint64_t h = (uint64_t)(uint32_t)x * 0x9E3779B97F4A7C15ull;
  h ^= ((uint64_t)(uint32_t)y + 0x7F4A7C15ull) * 0xC2B2AE3D27D4EB4Full;
  h ^= ((uint64_t)(uint32_t)z + 0x165667B1ull) * 0xD6E8FEB86659FD93ull;
  h ^= h >> 29;
  return h * 0xBF58476D1CE4E5B9ull;
}

static void grid_init(grid_t* g, size_t max_cells) {
  size_t cap = 64;
  while (cap < 2 * max_cells + 16) cap <<= 1;
  g->slots = (cell_t*)calloc(cap, sizeof(cell_t));
  g->cap = cap; g->count = 0;
  g->order = (size_t*)malloc((max_cells + 1) * sizeof(size_t));
  g->updated = (size_t*)malloc((max_cells + 1) * sizeof(size_t));
  g->n_updated = 0;
}
static void grid_free(grid_t* g) { free(g->slots); free(g->order); free(g->updated); }

static size_t grid_mutable(grid_t* g, int32_t x, int32_t y, int32_t z) {
  size_t i = (size_t)(cell_hash(x, y, z) >> 20) & (g->cap - 1);
  for (;;) {
    cell_t* c = &g->slots[i];
    if (!c->used) {
      c->used = 1; c->x = x; c->y = y; c->z = z; c->value = 0;
      g->order[g->count++] = i;
      return i;
    }
    if (c->x == x && c->y == y && c->z == z) return i;
    i = (i + 1) & (g->cap - 1);
  }
}

/* hybrid_grid.h:491-508 */
static void apply_lookup_table(grid_t* g, int32_t x, int32_t y, int32_t z, const uint16_t* table) {
  const size_t i = grid_mutable(g, x, y, z);
  cell_t* c = &g->slots[i];
  if (c->value >= K_UPDATE_MARKER) return;
  g->updated[g->n_updated++] = i;
  c->value = table[c->value];
}
/* hybrid_grid.h:483-489 */
static void finish_update(grid_t* g) {
  while (g->n_updated) g->slots[g->updated[--g->n_updated]].value -= K_UPDATE_MARKER;
}

/* hybrid_grid.h:429-434 with port.h:41 */
static void cell_index(const float p[3], float resolution, int32_t out[3]) {
  for (int a = 0; a < 3; ++a) out[a] = (int32_t)lroundf(p[a] / resolution);
}

static int imax3abs(const int32_t d[3]) {
  int m = abs(d[0]);
  if (abs(d[1]) > m) m = abs(d[1]);
  if (abs(d[2]) > m) m = abs(d[2]);
  return m;
}

int oracle_bev_project(const float* xyz, size_t n, size_t stride_floats, float resolution,
                       float max_range, uint8_t** out_img, oracle_bev_info* info) {
  bev_tables* tab = make_tables();
  memset(info, 0, sizeof(*info));
  *out_img = NULL;

  /* point_cloud_to_range_data (loop_detector.cpp:108-120): sqrt(x*x + y*y + z*z) > 100. sends the
   * point to `misses` (never read again on this path); then FilterRangeDataByMaxRange
   * (submap_3d.cpp:43-52) keeps a return when (hit - origin).norm() <= max_range, Eigen's 3-vector
   * reduction being x*x + (y*y + z*z).  Both tests are kept, each in its own order. */
  float* ret = (float*)malloc((n ? n : 1) * 3 * sizeof(float));
  size_t nr = 0;
  for (size_t i = 0; i < n; ++i) {
    const float x = xyz[i * stride_floats], y = xyz[i * stride_floats + 1], z = xyz[i * stride_floats + 2];
    const float s1 = (x * x + y * y) + z * z;
    if ((double)sqrtf(s1) > (double)max_range) continue;
    const float hx = x - 0.f, hy = y - 0.f, hz = z - 0.f; /* hit - origin; TransformRangeData by identity is exact */
    const float s2 = hx * hx + (hy * hy + hz * hz);
    if (!(sqrtf(s2) <= (float)(int)max_range)) continue; /* the callee takes the range as int (submap_3d.h) */
    ret[nr * 3] = x; ret[nr * 3 + 1] = y; ret[nr * 3 + 2] = z;
    ++nr;
  }
  info->n_returns = (uint32_t)nr;

  grid_t g;
  grid_init(&g, 3 * nr);
  /* RangeDataInserter3D::Insert (range_data_inserter_3d.cpp:63-78): hits first ... */
  for (size_t i = 0; i < nr; ++i) {
    int32_t c[3];
    cell_index(&ret[i * 3], resolution, c);
    apply_lookup_table(&g, c[0], c[1], c[2], tab->hit_table);
  }
  /* ... then the last two free-space voxels of every ray (InsertMissesIntoGrid, :27-52) */
  {
    const float origin[3] = {0.f, 0.f, 0.f};
    int32_t oc[3];
    cell_index(origin, resolution, oc);
    for (size_t i = 0; i < nr; ++i) {
      int32_t hc[3], d[3];
      cell_index(&ret[i * 3], resolution, hc);
      for (int a = 0; a < 3; ++a) d[a] = hc[a] - oc[a];
      const int num_samples = imax3abs(d);
      const int start = num_samples - 2 > 0 ? num_samples - 2 : 0;
      for (int position = start; position < num_samples; ++position)
        apply_lookup_table(&g, oc[0] + d[0] * position / num_samples,
                           oc[1] + d[1] * position / num_samples,
                           oc[2] + d[2] * position / num_samples, tab->miss_table);
    }
  }
  finish_update(&g);
  info->n_cells_known = (uint32_t)g.count;

  /* ProjectToCvMat (submap_3d.cpp:238-326) with the identity transform: gravity_aligned is the
   * identity rotation, so cell_center_aligned == cell centre. */
  const float resolution_inverse = 1.f / resolution;
  int32_t mn[3] = {INT_MAX, INT_MAX, INT_MAX}, mx[3] = {INT_MIN, INT_MIN, INT_MIN};
  int32_t* vox = (int32_t*)malloc((g.count + 1) * 4 * sizeof(int32_t));
  size_t nv = 0;
  for (size_t k = 0; k < g.count; ++k) {
    const cell_t* c = &g.slots[g.order[k]];
    const float probability = tab->value_to_probability[c->value];
    if (probability < 0.501f) continue;
    const float centre[3] = {(float)c->x * resolution, (float)c->y * resolution, (float)c->z * resolution};
    int32_t* v = &vox[nv * 4];
    for (int a = 0; a < 3; ++a) {
      v[a] = (int32_t)lroundf(centre[a] * resolution_inverse);
      if (v[a] < mn[a]) mn[a] = v[a];
      if (v[a] > mx[a]) mx[a] = v[a];
    }
    v[3] = c->value;
    ++nv;
  }
  info->n_cells_obstructed = (uint32_t)nv;
  info->resolution = (double)resolution;
  if (nv == 0) { /* the reference computes a negative size here and aborts inside cv::Mat */
    free(vox); grid_free(&g); free(ret); free(tab);
    return 1;
  }
  info->min_ix = mn[0]; info->min_iy = mn[1]; info->min_iz = mn[2];
  info->max_ix = mx[0]; info->max_iy = mx[1]; info->max_iz = mx[2];
  info->ox = mn[0] * (double)resolution;
  info->oy = mn[1] * (double)resolution;
  const int width = mx[0] - mn[0] + 1, height = mx[1] - mn[1] + 1;
  info->width = (uint32_t)width; info->height = (uint32_t)height;

  float* psum = (float*)calloc((size_t)width * height, sizeof(float));
  for (size_t k = 0; k < nv; ++k) {
    const int32_t* v = &vox[k * 4];
    if (v[0] < mn[0] || v[1] < mn[1] || v[0] > mx[0] || v[1] > mx[1]) continue;
    psum[(size_t)(v[1] - mn[1]) * width + (v[0] - mn[0])] += tab->value_to_probability[v[3]];
  }
  const float kMaxProbability = 1.f - kMinProbability;
  uint8_t* img = (uint8_t*)malloc((size_t)width * height);
  for (size_t k = 0; k < (size_t)width * height; ++k) img[k] = psum[k] > kMaxProbability ? 0 : 255;
  *out_img = img;
  free(psum); free(vox); grid_free(&g); free(ret); free(tab);
  return 0;
}

/* crop_pad_occupancy (loop_detector.cpp:83-106).  dst starts as cv::Mat::ones(h, w, CV_8UC3) * 255:
 * OpenCV's Mat::ones sets only the FIRST channel of a multi-channel matrix, so the padding is
 * (255, 0, 0); the grey source is expanded to three equal channels (CV_GRAY2BGR). */
void oracle_bev_crop_pad(const uint8_t* src, uint32_t src_w, uint32_t src_h, uint32_t out_w,
                         uint32_t out_h, uint8_t* dst_hwc3) {
  for (size_t i = 0; i < (size_t)out_w * out_h; ++i) {
    dst_hwc3[i * 3] = 255; dst_hwc3[i * 3 + 1] = 0; dst_hwc3[i * 3 + 2] = 0;
  }
  const int cw = src_w >= out_w ? (int)out_w : (int)src_w;
  const int ch = src_h >= out_h ? (int)out_h : (int)src_h;
  const int sx = (int)floor(((int)src_w - cw) / 2.), sy = (int)floor(((int)src_h - ch) / 2.);
  const int dx = (int)floor(((int)out_w - cw) / 2.), dy = (int)floor(((int)out_h - ch) / 2.);
  for (int y = 0; y < ch; ++y)
    for (int x = 0; x < cw; ++x) {
      const uint8_t v = src[(size_t)(sy + y) * src_w + (sx + x)];
      uint8_t* d = &dst_hwc3[((size_t)(dy + y) * out_w + (dx + x)) * 3];
      d[0] = v; d[1] = v; d[2] = v;
    }
}

/* get_place_feature's tensor (loop_detector.cpp:146-151): convertTo(CV_32FC3, 1/255) then NHWC ->
 * NCHW.  Inputs are 0 or 255 only, for which u8 * (1/255) is exactly 0 or 1 in either precision. */
void oracle_bev_to_chw_f32(const uint8_t* hwc3, uint32_t w, uint32_t h, float* out_chw) {
  for (uint32_t c = 0; c < 3; ++c)
    for (size_t i = 0; i < (size_t)w * h; ++i)
      out_chw[(size_t)c * w * h + i] = (float)((double)hwc3[i * 3 + c] * (1.0 / 255.0));
}

void oracle_free(void* p) { free(p); }
