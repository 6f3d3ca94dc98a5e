/*
 * coarse_oracle.c -- CPU restatement of the coarse global (x, y, yaw) match on BEV occupancy grids
 * (TEST INFRASTRUCTURE ONLY, see gloc_oracle.h).  Plain C; build with -ffp-contract=off.
 *
 * PARITY UNPINNED: the reference finds the coarse pose with SURF + FLANN + cv::estimateAffinePartial2D
 * (RpyPCLoopDetector::match, registration/loop_detector.cpp:192-288) -- OpenCV 3.3.1 with the contrib
 * xfeatures2d module, neither under /root/reference nor in this image, and it holds no fixture for it.
 * What is kept from the reference is the interface: inputs = two occupancy images with (ox, oy, res),
 * binarised at 100 (loop_detector.cpp:196-197), metric = tl + pixel * res (:245-252); output =
 * (x, y, yaw) with p_db = R(yaw) p_q + (x, y) (:276-281), or "no match".
 *
 * The search itself (this build's design, mirrored by gloc3d_amd/csrc/coarse_kernels.hpp):
 *   grid    occupied pixels, by their integer voxel index ix = lround(ox / res) + x, -> cells of
 *           cell_px x cell_px pixels on a 512 x 512 grid centred on the sensor (cell index =
 *           floor(ix / cell_px) + 256); bit map, 3x3 dilation, projections hx / hy;
 *   yaw     for k in 0..n_yaw-1: rotate the query's occupied cell centres (pixel units) by 2 pi k / n_yaw
 *           (fp32: c*x - s*y, s*x + c*y, un-fused; cos / sin computed in fp64 and rounded once), round to
 *           a pixel (halves away from zero), re-bin, project;
 *           correlate each projection with the database grid's over lags -max_shift..max_shift; per
 *           axis the largest sum, ties -> the most negative lag; score = sx + sy;
 *   verify  the top_yaw rotations by score (ties -> smaller k), then the identity (k = 0, lag 0): for
 *           every shift within `refine` of the lags, count the rotated query cells that land on a set bit
 *           of the DILATED database map; largest count, ties -> first in (dy, dx) row-major order;
 *   result  the rotation candidate with the largest count (ties -> the earlier one) if it has more
 *           than 1.2 x the identity's count, else the identity; ok iff n_query >= 16 and
 *           count >= min_overlap * n_query (fp32 comparison);
 *   scale   (round 3: the `scale` of the reference's match and its acceptance |1 - scale| < 0.1,
 *           loop_detector.cpp:262-272) under the chosen rotation the overlap search is repeated with the query's cells
 *           scaled about the sensor by 0.88, 0.92 .. 1.12 (rounded after scaling), each over the shifts within
 *           `refine` of the chosen one; scale = the factor with the largest overlap (ties: nearest 1, then the smaller),
 *           refined by the parabola through its neighbours' overlaps unless it is an end of the range; no overlap at
 *           all -> 0; ok is withdrawn unless |1 - scale| < 0.1 (fp32).
 */
#include "gloc_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

#define CG 512
#define CGW (CG / 32)
#define CHALF (CG / 2)

struct oracle_coarse_grid {
  uint32_t bits[CG * CGW], dil[CG * CGW];
  uint32_t hx[CG], hy[CG];
  uint32_t n;
  uint32_t* cells; /* (v << 16) | u, row-major order */
};

static int round_half_away_f(float v) {
  float r = (float)(int)v;
  const float d = v - r;
  if (d >= 0.5f) r += 1.f;
  else if (d <= -0.5f) r -= 1.f;
  return (int)r;
}

static int cell_of_px(int ix, int cell_px) {
  const int u = (ix >= 0 ? ix / cell_px : -((-ix + cell_px - 1) / cell_px)) + CHALF;
  return (u >= 0 && u < CG) ? u : -1;
}

static float cell_centre_px(int u, int cell_px) {
  return (float)((u - CHALF) * cell_px) + 0.5f * (float)(cell_px - 1);
}

static int get_bit(const uint32_t* b, int u, int v) { return (int)((b[v * CGW + (u >> 5)] >> (u & 31)) & 1u); }

static void finish(oracle_coarse_grid* g) {
  uint32_t n = 0;
  memset(g->hx, 0, sizeof(g->hx));
  memset(g->hy, 0, sizeof(g->hy));
  memset(g->dil, 0, sizeof(g->dil));
  for (int v = 0; v < CG; ++v)
    for (int u = 0; u < CG; ++u)
      if (get_bit(g->bits, u, v)) n++;
  g->n = n;
  g->cells = (uint32_t*)malloc(sizeof(uint32_t) * (n ? n : 1));
  uint32_t k = 0;
  for (int v = 0; v < CG; ++v)
    for (int u = 0; u < CG; ++u) {
      if (!get_bit(g->bits, u, v)) continue;
      g->cells[k++] = ((uint32_t)v << 16) | (uint32_t)u;
      g->hx[u]++;
      g->hy[v]++;
      for (int dv = -1; dv <= 1; ++dv)
        for (int du = -1; du <= 1; ++du) {
          const int uu = u + du, vv = v + dv;
          if (uu >= 0 && uu < CG && vv >= 0 && vv < CG) g->dil[vv * CGW + (uu >> 5)] |= 1u << (uu & 31);
        }
    }
}

oracle_coarse_grid* oracle_coarse_grid_from_image(const uint8_t* img, uint32_t w, uint32_t h, float ox, float oy,
                                                  float res, uint32_t cell_px) {
  oracle_coarse_grid* g = (oracle_coarse_grid*)calloc(1, sizeof(oracle_coarse_grid));
  /* xy_res = min voxel index * resolution (loop_detector.cpp:133) */
  const int ix0 = (int)lround((double)ox / (double)res), iy0 = (int)lround((double)oy / (double)res);
  for (uint32_t y = 0; y < h; ++y)
    for (uint32_t x = 0; x < w; ++x) {
      if (img[(size_t)y * w + x] >= 100) continue; /* cv::threshold(..., 100, 255, THRESH_BINARY_INV) */
      const int u = cell_of_px(ix0 + (int)x, (int)cell_px), v = cell_of_px(iy0 + (int)y, (int)cell_px);
      if (u >= 0 && v >= 0) g->bits[v * CGW + (u >> 5)] |= 1u << (u & 31);
    }
  finish(g);
  return g;
}

void oracle_coarse_grid_free(oracle_coarse_grid* g) {
  if (!g) return;
  free(g->cells);
  free(g);
}

uint32_t oracle_coarse_grid_cells(const oracle_coarse_grid* g, uint32_t* out) {
  if (out) memcpy(out, g->cells, sizeof(uint32_t) * g->n);
  return g->n;
}

static void rotate_cell(uint32_t uv, float c, float s, int cell_px, int* u, int* v) {
  const float x = cell_centre_px((int)(uv & 0xFFFF), cell_px), y = cell_centre_px((int)(uv >> 16), cell_px);
  const float a0 = c * x, a1 = s * y, b0 = s * x, b1 = c * y;
  *u = cell_of_px(round_half_away_f(a0 - a1), cell_px);
  *v = cell_of_px(round_half_away_f(b0 + b1), cell_px);
}

static uint32_t overlap_at(const oracle_coarse_grid* q, const oracle_coarse_grid* d, float c, float s, int cell,
                           int tx, int ty) {
  uint32_t cnt = 0;
  for (uint32_t i = 0; i < q->n; ++i) {
    int u, v;
    rotate_cell(q->cells[i], c, s, cell, &u, &v);
    if (u < 0 || v < 0) continue;
    u += tx;
    v += ty;
    if (u < 0 || u >= CG || v < 0 || v >= CG) continue;
    cnt += (uint32_t)get_bit(d->dil, u, v);
  }
  return cnt;
}

#define N_SCALES 7
static float scale_factor(int j) { return 0.88f + 0.04f * (float)j; }

static uint32_t overlap_scaled(const oracle_coarse_grid* q, const oracle_coarse_grid* d, float c, float s, float f, int cell,
                               int tx, int ty) {
  uint32_t cnt = 0;
  for (uint32_t i = 0; i < q->n; ++i) {
    const uint32_t uv = q->cells[i];
    const float x = cell_centre_px((int)(uv & 0xFFFF), cell), y = cell_centre_px((int)(uv >> 16), cell);
    const float a0 = c * x, a1 = s * y, b0 = s * x, b1 = c * y;
    const float rx = a0 - a1, ry = b0 + b1;
    int u = cell_of_px(round_half_away_f(rx * f), cell), v = cell_of_px(round_half_away_f(ry * f), cell);
    if (u < 0 || v < 0) continue;
    u += tx;
    v += ty;
    if (u < 0 || u >= CG || v < 0 || v >= CG) continue;
    cnt += (uint32_t)get_bit(d->dil, u, v);
  }
  return cnt;
}

/* the overlap search repeated with the query scaled about the sensor by 0.88 .. 1.12, each over the shifts within
 * `refine` of the chosen one; the factor with the largest overlap (ties: nearest 1, then the smaller), refined by the
 * parabola through its neighbours when it is not an end of the range */
static float estimate_scale(const oracle_coarse_grid* q, const oracle_coarse_grid* d, float c, float s, int cell, int tx,
                            int ty, int W, uint32_t* n_matched) {
  uint32_t o[N_SCALES];
  for (int j = 0; j < N_SCALES; ++j) {
    uint32_t best = 0;
    for (int dy = -W; dy <= W; ++dy)
      for (int dx = -W; dx <= W; ++dx) {
        const uint32_t v = overlap_scaled(q, d, c, s, scale_factor(j), cell, tx + dx, ty + dy);
        if (v > best) best = v;
      }
    o[j] = best;
  }
  if (n_matched) *n_matched = o[N_SCALES / 2];
  int best = N_SCALES / 2;
  for (int dd = 1; dd <= N_SCALES / 2; ++dd) {
    if (o[N_SCALES / 2 - dd] > o[best]) best = N_SCALES / 2 - dd;
    if (o[N_SCALES / 2 + dd] > o[best]) best = N_SCALES / 2 + dd;
  }
  if (o[best] == 0u) return 0.f;
  double est = (double)scale_factor(best);
  if (best > 0 && best < N_SCALES - 1) {
    const double a = (double)o[best - 1], b0 = (double)o[best], cc = (double)o[best + 1];
    const double den = a - 2.0 * b0 + cc;
    if (den < 0.0) est += 0.5 * 0.04 * (a - cc) / den;
  }
  return (float)est;
}

void oracle_coarse_match(const oracle_coarse_grid* q, const oracle_coarse_grid* d, float res, uint32_t cell_px,
                         uint32_t n_yaw, uint32_t max_shift, uint32_t top_yaw, uint32_t refine, float min_overlap,
                         float* out_xy_yaw, float* out_ratio, int* out_ok, uint32_t* out_overlap,
                         uint32_t* out_k) {
  oracle_coarse_match_scale(q, d, res, cell_px, n_yaw, max_shift, top_yaw, refine, min_overlap, out_xy_yaw, out_ratio,
                            out_ok, out_overlap, out_k, NULL, NULL);
}

void oracle_coarse_match_scale(const oracle_coarse_grid* q, const oracle_coarse_grid* d, float res, uint32_t cell_px,
                               uint32_t n_yaw, uint32_t max_shift, uint32_t top_yaw, uint32_t refine, float min_overlap,
                               float* out_xy_yaw, float* out_ratio, int* out_ok, uint32_t* out_overlap,
                               uint32_t* out_k, float* out_scale, uint32_t* out_matched) {
  const int T = (int)max_shift, W = (int)refine, cell = (int)cell_px;
  const float cell_m = (float)cell_px * res;
  float* cs = (float*)malloc(sizeof(float) * 2 * n_yaw);
  uint64_t* score = (uint64_t*)malloc(sizeof(uint64_t) * n_yaw);
  int* lag = (int*)malloc(sizeof(int) * 2 * n_yaw);
  for (uint32_t k = 0; k < n_yaw; ++k) {
    const double a = 2.0 * M_PI * (double)k / (double)n_yaw;
    cs[2 * k] = (float)cos(a);
    cs[2 * k + 1] = (float)sin(a);
    uint32_t hq[2][CG];
    memset(hq, 0, sizeof(hq));
    for (uint32_t i = 0; i < q->n; ++i) {
      int u, v;
      rotate_cell(q->cells[i], cs[2 * k], cs[2 * k + 1], cell, &u, &v);
      if (u >= 0 && v >= 0) {
        hq[0][u]++;
        hq[1][v]++;
      }
    }
    uint32_t best[2] = {0, 0};
    int bl[2] = {-T, -T};
    for (int axis = 0; axis < 2; ++axis) {
      const uint32_t* hd = axis == 0 ? d->hx : d->hy;
      for (int t = -T; t <= T; ++t) {
        uint32_t acc = 0;
        const int i0 = t < 0 ? -t : 0, i1 = t > 0 ? CG - t : CG;
        for (int i = i0; i < i1; ++i) acc += hq[axis][i] * hd[i + t];
        if (t == -T || acc > best[axis]) { /* ties -> the most negative lag */
          best[axis] = acc;
          bl[axis] = t;
        }
      }
    }
    score[k] = (uint64_t)best[0] + (uint64_t)best[1];
    /* (sx + sy) is summed in 32 bits on the device: reproduce the wrap (never reached in practice) */
    score[k] = (uint32_t)score[k];
    lag[2 * k] = bl[0];
    lag[2 * k + 1] = bl[1];
  }
  /* candidates: top_yaw rotations by score (ties -> smaller k), then the identity */
  uint32_t best_over = 0, best_k = 0xFFFFFFFFu;
  int best_tx = 0, best_ty = 0, have = 0;
  char* used = (char*)calloc(n_yaw, 1);
  for (uint32_t m = 0; m < top_yaw; ++m) {
    int bk = -1;
    for (uint32_t k = 0; k < n_yaw; ++k)
      if (!used[k] && (bk < 0 || score[k] > score[bk])) bk = (int)k;
    if (bk < 0) break;
    used[bk] = 1;
    uint32_t bo = 0;
    int btx = 0, bty = 0, first = 1;
    for (int dy = -W; dy <= W; ++dy)
      for (int dx = -W; dx <= W; ++dx) {
        const uint32_t o = overlap_at(q, d, cs[2 * bk], cs[2 * bk + 1], cell, lag[2 * bk] + dx, lag[2 * bk + 1] + dy);
        if (first || o > bo) {
          bo = o;
          btx = lag[2 * bk] + dx;
          bty = lag[2 * bk + 1] + dy;
          first = 0;
        }
      }
    if (!have || bo > best_over) {
      best_over = bo;
      best_k = (uint32_t)bk;
      best_tx = btx;
      best_ty = bty;
      have = 1;
    }
  }
  uint32_t id_over = 0;
  int id_tx = 0, id_ty = 0, first = 1;
  for (int dy = -W; dy <= W; ++dy)
    for (int dx = -W; dx <= W; ++dx) {
      const uint32_t o = overlap_at(q, d, cs[0], cs[1], cell, dx, dy);
      if (first || o > id_over) {
        id_over = o;
        id_tx = dx;
        id_ty = dy;
        first = 0;
      }
    }
  if (!have || !((uint64_t)best_over * 5u > (uint64_t)id_over * 6u)) {
    best_over = id_over;
    best_k = 0;
    best_tx = id_tx;
    best_ty = id_ty;
  }
  const double a = 2.0 * M_PI * (double)best_k / (double)n_yaw;
  out_xy_yaw[0] = (float)best_tx * cell_m;
  out_xy_yaw[1] = (float)best_ty * cell_m;
  out_xy_yaw[2] = (float)(a > M_PI ? a - 2.0 * M_PI : a);
  const double ab = 2.0 * M_PI * (double)best_k / (double)n_yaw;
  const float scale = estimate_scale(q, d, (float)cos(ab), (float)sin(ab), cell, best_tx, best_ty, W, out_matched);
  if (out_scale) *out_scale = scale;
  if (out_ratio) *out_ratio = q->n ? (float)best_over / (float)q->n : 0.f;
  if (out_ok)
    *out_ok = (q->n >= 16 && (float)best_over >= min_overlap * (float)q->n && fabsf(1.f - scale) < 0.1f) ? 1 : 0;
  if (out_overlap) *out_overlap = best_over;
  if (out_k) *out_k = best_k;
  free(cs);
  free(score);
  free(lag);
  free(used);
}
