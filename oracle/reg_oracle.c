/*
 * reg_oracle.c -- CPU restatement of the 3-D registration half of the hot path
 * (TEST INFRASTRUCTURE ONLY, see gloc_oracle.h).  Plain C; build with -ffp-contract=off.
 *
 * PARITY UNPINNED for RANSAC/ICP: the reference performs this arithmetic inside PCL
 * (pcl::IterativeClosestPoint, registration/global_registration.cpp:237-248) and OpenCV
 * (cv::estimateAffinePartial2D(..., RANSAC, 3*res, 3000), registration/loop_detector.cpp:256-257),
 * neither of which is under /root/reference nor version-pinned, and holds no fixture for it.
 * What IS pinned: the exact 1-NN (oracle_nn3*) against the reference's vendored nanoflann
 * instantiated for 3-D points with L2_Simple_Adaptor (registration/nanoflann.hpp:509-539).
 *
 * Semantics:
 *   S1  j(i) = argmin_j ||T p_i - q_j||^2, exact, tie -> smallest j
 *   S2  RANSAC: H hypotheses, each a Kabsch fit (fp64, SVD) to 3 sampled correspondences; inlier
 *       iff ||R p + t - q|| < thr; best = max inliers, tie -> min h; refit on the best's inliers
 *   S3  ICP: repeat {S1; Kabsch over all pairs; T <- dT * T}
 * Constants mirrored from the reference: 3000 iterations and 0.6 m threshold
 * (loop_detector.cpp:257, loop_detector.h:116), 30 ICP iterations (global_registration.cpp:242),
 * success metric (global_localization.cpp:288-306).
 */
#include "gloc_oracle.h"

#include <float.h>
#include <math.h>
#include <pthread.h>
#include <stdlib.h>
#include <string.h>

/* ---- fp32 point arithmetic (fixed order, mirrored by the HIP kernels) ---------------------- */

static inline void xform_f32(const float R[9], const float t[3], const float* p, float* o) {
  const float x = p[0], y = p[1], z = p[2];
  o[0] = ((R[0] * x + R[1] * y) + R[2] * z) + t[0];
  o[1] = ((R[3] * x + R[4] * y) + R[5] * z) + t[1];
  o[2] = ((R[6] * x + R[7] * y) + R[8] * z) + t[2];
}

/* nanoflann.hpp:521-532 (L2_Simple_Adaptor::evalMetric): result += diff*diff, d = 0,1,2 */
static inline float dist2_f32(const float* a, const float* b) {
  const float dx = a[0] - b[0];
  const float dy = a[1] - b[1];
  const float dz = a[2] - b[2];
  return (dx * dx + dy * dy) + dz * dz;
}

void oracle_transform_points(const float* T16, const float* xyz, size_t n, float* out_xyz) {
  const float R[9] = {T16[0], T16[1], T16[2], T16[4], T16[5], T16[6], T16[8], T16[9], T16[10]};
  const float t[3] = {T16[3], T16[7], T16[11]};
  for (size_t i = 0; i < n; ++i) xform_f32(R, t, xyz + 3 * i, out_xyz + 3 * i);
}

void oracle_nn3(const float* src_xyz, size_t n_src, const float* tgt_xyz, size_t n_tgt,
                uint32_t* out_idx, float* out_d2) {
  for (size_t i = 0; i < n_src; ++i) {
    float best = FLT_MAX;
    uint32_t bj = UINT32_MAX;
    for (size_t j = 0; j < n_tgt; ++j) {
      const float d2 = dist2_f32(src_xyz + 3 * i, tgt_xyz + 3 * j);
      if (d2 < best) {
        best = d2;
        bj = (uint32_t)j;
      }
    }
    out_idx[i] = bj;
    out_d2[i] = best;
  }
}

/* Uniform-grid exact NN: identical result to oracle_nn3 (same fp32 d2, smallest index on ties),
 * for scans too large for the exhaustive scan.  Rings of cells are visited outward until the
 * ring's lower bound exceeds the best distance (with a relative margin covering fp32 rounding). */
typedef struct {
  float lo[3];
  double inv_cell, cell;
  int dim[3];
  uint32_t* cell_start; /* dim0*dim1*dim2 + 1 */
  uint32_t* order;      /* target ids sorted by cell, ascending id inside a cell */
} grid3;

static int clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }

static void grid_build(grid3* g, const float* tgt, size_t n) {
  float lo[3] = {FLT_MAX, FLT_MAX, FLT_MAX}, hi[3] = {-FLT_MAX, -FLT_MAX, -FLT_MAX};
  for (size_t j = 0; j < n; ++j)
    for (int a = 0; a < 3; ++a) {
      if (tgt[3 * j + a] < lo[a]) lo[a] = tgt[3 * j + a];
      if (tgt[3 * j + a] > hi[a]) hi[a] = tgt[3 * j + a];
    }
  double vol = 1.0;
  for (int a = 0; a < 3; ++a) vol *= (double)(hi[a] - lo[a]) + 1e-3;
  double cell = cbrt(vol / ((double)n / 4.0 + 1.0));
  if (cell < 0.25) cell = 0.25;
  g->cell = cell;
  g->inv_cell = 1.0 / cell;
  size_t total = 1;
  for (int a = 0; a < 3; ++a) {
    g->lo[a] = lo[a];
    g->dim[a] = (int)(((double)(hi[a] - lo[a])) * g->inv_cell) + 1;
    total *= (size_t)g->dim[a];
  }
  g->cell_start = (uint32_t*)calloc(total + 1, sizeof(uint32_t));
  g->order = (uint32_t*)malloc(sizeof(uint32_t) * (n ? n : 1));
  uint32_t* cid = (uint32_t*)malloc(sizeof(uint32_t) * (n ? n : 1));
  for (size_t j = 0; j < n; ++j) {
    int c[3];
    for (int a = 0; a < 3; ++a)
      c[a] = clampi((int)(((double)tgt[3 * j + a] - (double)lo[a]) * g->inv_cell), 0, g->dim[a] - 1);
    cid[j] = (uint32_t)((c[2] * g->dim[1] + c[1]) * g->dim[0] + c[0]);
    g->cell_start[cid[j] + 1]++;
  }
  for (size_t c = 0; c < total; ++c) g->cell_start[c + 1] += g->cell_start[c];
  uint32_t* fill = (uint32_t*)malloc(sizeof(uint32_t) * (total ? total : 1));
  memcpy(fill, g->cell_start, sizeof(uint32_t) * total);
  for (size_t j = 0; j < n; ++j) g->order[fill[cid[j]]++] = (uint32_t)j; /* ascending id */
  free(fill);
  free(cid);
}

static void grid_free(grid3* g) {
  free(g->cell_start);
  free(g->order);
}

void oracle_nn3_grid(const float* src_xyz, size_t n_src, const float* tgt_xyz, size_t n_tgt,
                     uint32_t* out_idx, float* out_d2) {
  if (n_tgt == 0) {
    for (size_t i = 0; i < n_src; ++i) {
      out_idx[i] = UINT32_MAX;
      out_d2[i] = FLT_MAX;
    }
    return;
  }
  grid3 g;
  grid_build(&g, tgt_xyz, n_tgt);
  const int maxr = g.dim[0] > g.dim[1] ? (g.dim[0] > g.dim[2] ? g.dim[0] : g.dim[2])
                                       : (g.dim[1] > g.dim[2] ? g.dim[1] : g.dim[2]);
  for (size_t i = 0; i < n_src; ++i) {
    const float* p = src_xyz + 3 * i;
    double f[3];
    int c[3];
    /* distance from p to the clamped cell (points outside the grid) */
    double out_d = 0.0;
    for (int a = 0; a < 3; ++a) {
      f[a] = ((double)p[a] - (double)g.lo[a]) * g.inv_cell;
      int ci = (int)floor(f[a]);
      c[a] = clampi(ci, 0, g.dim[a] - 1);
      double below = (double)g.lo[a] - (double)p[a];
      double above = (double)p[a] - ((double)g.lo[a] + g.cell * (double)g.dim[a]);
      double o = below > 0 ? below : (above > 0 ? above : 0.0);
      out_d += o * o;
    }
    float best = FLT_MAX;
    uint32_t bj = UINT32_MAX;
    for (int r = 0; r <= maxr + 1; ++r) {
      /* lower bound on the distance from p to any point in a cell at Chebyshev ring r around
       * the clamped cell c: at least (r-1) whole cells away along one axis */
      if (r >= 1) {
        double lb = (double)(r - 1) * g.cell;
        double lb2 = lb * lb;
        if (lb2 < out_d) lb2 = out_d;
        if (bj != UINT32_MAX && lb2 > (double)best * (1.0 + 1e-5) + 1e-9) break;
      }
      int any = 0;
      for (int dz = -r; dz <= r; ++dz) {
        int z = c[2] + dz;
        if (z < 0 || z >= g.dim[2]) continue;
        for (int dy = -r; dy <= r; ++dy) {
          int y = c[1] + dy;
          if (y < 0 || y >= g.dim[1]) continue;
          int shell_yz = (abs(dz) == r) || (abs(dy) == r);
          for (int dx = -r; dx <= r; dx += (shell_yz ? 1 : (r > 0 ? 2 * r : 1))) {
            int x = c[0] + dx;
            if (x < 0 || x >= g.dim[0]) continue;
            any = 1;
            size_t cell = ((size_t)z * (size_t)g.dim[1] + (size_t)y) * (size_t)g.dim[0] + (size_t)x;
            for (uint32_t s = g.cell_start[cell]; s < g.cell_start[cell + 1]; ++s) {
              uint32_t j = g.order[s];
              float d2 = dist2_f32(p, tgt_xyz + 3 * (size_t)j);
              if (d2 < best || (d2 == best && j < bj)) {
                best = d2;
                bj = j;
              }
            }
          }
        }
      }
      (void)any;
    }
    out_idx[i] = bj;
    out_d2[i] = best;
  }
  grid_free(&g);
}

/* ---- fp64 3x3 helpers --------------------------------------------------------------------- */

/* Cyclic Jacobi eigen-decomposition of a symmetric 3x3 (row-major A), fixed sweep order
 * (0,1),(0,2),(1,2).  Uses only + - * / sqrt, so CPU and GPU fp64 agree bit for bit.
 * On return A's diagonal holds the eigenvalues and V's columns the eigenvectors. */
void oracle_jacobi_eig3(double A[9], double V[9]) {
  V[0] = 1; V[1] = 0; V[2] = 0;
  V[3] = 0; V[4] = 1; V[5] = 0;
  V[6] = 0; V[7] = 0; V[8] = 1;
  static const int PP[3] = {0, 0, 1}, QQ[3] = {1, 2, 2};
  for (int sweep = 0; sweep < 16; ++sweep) {
    const double off = A[1] * A[1] + A[2] * A[2] + A[5] * A[5];
    const double diag = A[0] * A[0] + A[4] * A[4] + A[8] * A[8];
    if (off <= 1e-32 * diag || off == 0.0) break;
    for (int e = 0; e < 3; ++e) {
      const int p = PP[e], q = QQ[e];
      const double apq = A[3 * p + q];
      if (apq == 0.0) continue;
      const double app = A[3 * p + p], aqq = A[3 * q + q];
      const double theta = (aqq - app) / (2.0 * apq);
      const double at = theta < 0 ? -theta : theta;
      double t = 1.0 / (at + sqrt(theta * theta + 1.0));
      if (theta < 0) t = -t;
      const double c = 1.0 / sqrt(t * t + 1.0);
      const double s = t * c;
      /* A <- J^T A J, J = rotation in the (p,q) plane */
      for (int k = 0; k < 3; ++k) {
        const double akp = A[3 * k + p], akq = A[3 * k + q];
        A[3 * k + p] = c * akp - s * akq;
        A[3 * k + q] = s * akp + c * akq;
      }
      for (int k = 0; k < 3; ++k) {
        const double apk = A[3 * p + k], aqk = A[3 * q + k];
        A[3 * p + k] = c * apk - s * aqk;
        A[3 * q + k] = s * apk + c * aqk;
      }
      for (int k = 0; k < 3; ++k) {
        const double vkp = V[3 * k + p], vkq = V[3 * k + q];
        V[3 * k + p] = c * vkp - s * vkq;
        V[3 * k + q] = s * vkp + c * vkq;
      }
    }
  }
}

static inline void cross3(const double a[3], const double b[3], double o[3]) {
  o[0] = a[1] * b[2] - a[2] * b[1];
  o[1] = a[2] * b[0] - a[0] * b[2];
  o[2] = a[0] * b[1] - a[1] * b[0];
}

void oracle_kabsch_from_cov(const double M[9], const double pbar[3], const double qbar[3],
                            double R[9], double t[3]) {
  /* A = M^T M (symmetric), eigenvectors = right singular vectors of M (q-space) */
  double A[9], V[9];
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j)
      A[3 * i + j] = (M[0 + i] * M[0 + j] + M[3 + i] * M[3 + j]) + M[6 + i] * M[6 + j];
  oracle_jacobi_eig3(A, V);
  /* order the eigenvalues descending (stable: ties keep lower column first) */
  int o0 = 0, o1 = 1, o2 = 2;
  double l0 = A[0], l1 = A[4], l2 = A[8];
  if (l1 > l0) { int ti = o0; o0 = o1; o1 = ti; double td = l0; l0 = l1; l1 = td; }
  if (l2 > l1) { int ti = o1; o1 = o2; o2 = ti; double td = l1; l1 = l2; l2 = td; }
  if (l1 > l0) { int ti = o0; o0 = o1; o1 = ti; double td = l0; l0 = l1; l1 = td; }
  (void)o2; (void)l2;
  double v1[3] = {V[0 + o0], V[3 + o0], V[6 + o0]};
  double v2[3] = {V[0 + o1], V[3 + o1], V[6 + o1]};
  double v3[3];
  cross3(v1, v2, v3); /* right-handed V */
  /* left singular vectors u_i ~ M v_i (p-space), Gram-Schmidt, u3 = u1 x u2 */
  double u1[3], u2[3], u3[3];
  for (int i = 0; i < 3; ++i) {
    u1[i] = (M[3 * i + 0] * v1[0] + M[3 * i + 1] * v1[1]) + M[3 * i + 2] * v1[2];
    u2[i] = (M[3 * i + 0] * v2[0] + M[3 * i + 1] * v2[1]) + M[3 * i + 2] * v2[2];
  }
  double n1 = sqrt((u1[0] * u1[0] + u1[1] * u1[1]) + u1[2] * u1[2]);
  if (!(n1 > 1e-300)) { /* no spread at all: identity */
    for (int i = 0; i < 9; ++i) R[i] = (i % 4 == 0) ? 1.0 : 0.0;
    for (int i = 0; i < 3; ++i) t[i] = qbar[i] - pbar[i];
    return;
  }
  for (int i = 0; i < 3; ++i) u1[i] = u1[i] / n1;
  const double d12 = (u1[0] * u2[0] + u1[1] * u2[1]) + u1[2] * u2[2];
  for (int i = 0; i < 3; ++i) u2[i] = u2[i] - d12 * u1[i];
  double n2 = sqrt((u2[0] * u2[0] + u2[1] * u2[1]) + u2[2] * u2[2]);
  if (!(n2 > 1e-300)) {
    /* rank one: any unit vector orthogonal to u1, built deterministically */
    double ax[3] = {0, 0, 0};
    const double a0 = u1[0] < 0 ? -u1[0] : u1[0], a1 = u1[1] < 0 ? -u1[1] : u1[1],
                 a2 = u1[2] < 0 ? -u1[2] : u1[2];
    ax[(a0 <= a1 && a0 <= a2) ? 0 : ((a1 <= a2) ? 1 : 2)] = 1.0;
    cross3(u1, ax, u2);
    n2 = sqrt((u2[0] * u2[0] + u2[1] * u2[1]) + u2[2] * u2[2]);
  }
  for (int i = 0; i < 3; ++i) u2[i] = u2[i] / n2;
  cross3(u1, u2, u3);
  /* R = V U^T = sum_i v_i u_i^T  (maps p-space to q-space; proper rotation because both
   * bases are right-handed: this is V diag(1,1,det(V U^T)) U^T of the textbook form) */
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) R[3 * i + j] = (v1[i] * u1[j] + v2[i] * u2[j]) + v3[i] * u3[j];
  for (int i = 0; i < 3; ++i)
    t[i] = qbar[i] - ((R[3 * i + 0] * pbar[0] + R[3 * i + 1] * pbar[1]) + R[3 * i + 2] * pbar[2]);
}

/* ---- RANSAC -------------------------------------------------------------------------------- */

static inline uint32_t mulhi_idx(uint64_t u, uint32_t n) {
  return (uint32_t)(((unsigned __int128)u * (unsigned __int128)n) >> 64);
}

void oracle_ransac_sample(uint64_t seed, uint32_t cand, uint32_t hyp, uint32_t n, uint32_t out[3]) {
  const uint64_t key = oracle_rng_key(seed, ((uint64_t)cand << 32) | (uint64_t)hyp);
  uint64_t ctr = 0;
  out[0] = mulhi_idx(oracle_rng_draw(key, ctr++), n);
  out[1] = out[0];
  out[2] = out[0];
  for (int tries = 0; tries < 16 && out[1] == out[0]; ++tries)
    out[1] = mulhi_idx(oracle_rng_draw(key, ctr++), n);
  for (int tries = 0; tries < 16 && (out[2] == out[0] || out[2] == out[1]); ++tries)
    out[2] = mulhi_idx(oracle_rng_draw(key, ctr++), n);
}

int oracle_ransac_hypothesis(const float* src_xyz, const float* tgt_xyz, const uint32_t* corr,
                             uint32_t n, uint64_t seed, uint32_t cand, uint32_t hyp, float R[9],
                             float t[3]) {
  uint32_t s[3];
  oracle_ransac_sample(seed, cand, hyp, n, s);
  if (s[0] == s[1] || s[0] == s[2] || s[1] == s[2]) return 0;
  double p[3][3], q[3][3];
  for (int k = 0; k < 3; ++k)
    for (int a = 0; a < 3; ++a) {
      p[k][a] = (double)src_xyz[3 * (size_t)s[k] + a];
      q[k][a] = (double)tgt_xyz[3 * (size_t)corr[s[k]] + a];
    }
  /* degenerate (near-collinear) source triangle */
  double a[3], b[3], c[3];
  for (int i = 0; i < 3; ++i) {
    a[i] = p[1][i] - p[0][i];
    b[i] = p[2][i] - p[0][i];
  }
  cross3(a, b, c);
  const double aa = (a[0] * a[0] + a[1] * a[1]) + a[2] * a[2];
  const double bb = (b[0] * b[0] + b[1] * b[1]) + b[2] * b[2];
  const double cc = (c[0] * c[0] + c[1] * c[1]) + c[2] * c[2];
  if (!(aa > 1e-12) || !(bb > 1e-12) || !(cc > 1e-6 * (aa * bb))) return 0;
  double pbar[3], qbar[3], M[9];
  for (int i = 0; i < 3; ++i) {
    pbar[i] = ((p[0][i] + p[1][i]) + p[2][i]) / 3.0;
    qbar[i] = ((q[0][i] + q[1][i]) + q[2][i]) / 3.0;
  }
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j)
      M[3 * i + j] = ((p[0][i] - pbar[i]) * (q[0][j] - qbar[j]) +
                      (p[1][i] - pbar[i]) * (q[1][j] - qbar[j])) +
                     (p[2][i] - pbar[i]) * (q[2][j] - qbar[j]);
  double Rd[9], td[3];
  oracle_kabsch_from_cov(M, pbar, qbar, Rd, td);
  for (int i = 0; i < 9; ++i) R[i] = (float)Rd[i];
  for (int i = 0; i < 3; ++i) t[i] = (float)td[i];
  return 1;
}

uint32_t oracle_count_inliers(const float* src_xyz, const float* tgt_xyz, const uint32_t* corr,
                              uint32_t n, const float R[9], const float t[3], float thr) {
  const float thr2 = thr * thr;
  uint32_t cnt = 0;
  for (uint32_t i = 0; i < n; ++i) {
    float pp[3];
    xform_f32(R, t, src_xyz + 3 * (size_t)i, pp);
    if (dist2_f32(pp, tgt_xyz + 3 * (size_t)corr[i]) < thr2) cnt++;
  }
  return cnt;
}

uint32_t oracle_ransac_needed_iters(uint32_t inl, uint32_t n, float conf, uint32_t max_iters) {
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

/* Kabsch over pairs (src_i, tgt_corr[i]) with optional gate d2(R0 src_i + t0, tgt) < gate2
 * (gate2 <= 0: all pairs).  Raw moments accumulated in fp64, sequentially. */
/* step_rms (may be NULL): the RMS displacement this update gives the points it was fitted on -- with c their centroid,
 * s2 the trace of their covariance: |R c + t - c|^2 + (|R - I|_F^2 / 2) s2 (the second term is the rotation's
 * 2 (1 - cos theta) times the points' mean squared distance from c: exact for points in the plane of the rotation,
 * an upper bound otherwise).  The convergence measure of the ICP (gloc_reg_params.max_final_step). */
static uint32_t kabsch_pairs(const float* src, const float* tgt, const uint32_t* corr, uint32_t n,
                             const float* gateR, const float* gatet, float gate2, double R[9],
                             double t[3], double* step_rms) {
  double sp[3] = {0, 0, 0}, sq[3] = {0, 0, 0}, spq[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0}, spp = 0.0;
  uint32_t cnt = 0;
  for (uint32_t i = 0; i < n; ++i) {
    const float* p = src + 3 * (size_t)i;
    const float* q = tgt + 3 * (size_t)corr[i];
    if (gate2 > 0.0f) {
      float pp[3];
      xform_f32(gateR, gatet, p, pp);
      if (!(dist2_f32(pp, q) < gate2)) continue;
    }
    cnt++;
    for (int a = 0; a < 3; ++a) {
      sp[a] += (double)p[a];
      sq[a] += (double)q[a];
      for (int b = 0; b < 3; ++b) spq[3 * a + b] += (double)p[a] * (double)q[b];
    }
    spp += ((double)p[0] * (double)p[0] + (double)p[1] * (double)p[1]) + (double)p[2] * (double)p[2];
  }
  if (step_rms) *step_rms = 0.0;
  if (cnt < 3) return cnt;
  const double inv = 1.0 / (double)cnt;
  double pbar[3], qbar[3], M[9];
  for (int a = 0; a < 3; ++a) {
    pbar[a] = sp[a] * inv;
    qbar[a] = sq[a] * inv;
  }
  for (int a = 0; a < 3; ++a)
    for (int b = 0; b < 3; ++b) M[3 * a + b] = spq[3 * a + b] - (double)cnt * (pbar[a] * qbar[b]);
  oracle_kabsch_from_cov(M, pbar, qbar, R, t);
  if (step_rms) {
    double s2 = spp * inv - ((pbar[0] * pbar[0] + pbar[1] * pbar[1]) + pbar[2] * pbar[2]), dc2 = 0.0, f2 = 0.0;
    if (s2 < 0.0) s2 = 0.0;
    for (int a = 0; a < 3; ++a) {
      const double d = (((R[3 * a + 0] * pbar[0] + R[3 * a + 1] * pbar[1]) + R[3 * a + 2] * pbar[2]) + t[a]) - pbar[a];
      dc2 += d * d;
      for (int b = 0; b < 3; ++b) {
        const double e = R[3 * a + b] - (a == b ? 1.0 : 0.0);
        f2 += e * e;
      }
    }
    *step_rms = sqrt(dc2 + 0.5 * f2 * s2);
  }
  return cnt;
}

static void compose(const double Ra[9], const double ta[3], const double Rb[9], const double tb[3],
                    double Ro[9], double to[3]) { /* (Ra,ta) o (Rb,tb) */
  for (int i = 0; i < 3; ++i) {
    for (int j = 0; j < 3; ++j)
      Ro[3 * i + j] = (Ra[3 * i + 0] * Rb[0 + j] + Ra[3 * i + 1] * Rb[3 + j]) + Ra[3 * i + 2] * Rb[6 + j];
    to[i] = ((Ra[3 * i + 0] * tb[0] + Ra[3 * i + 1] * tb[1]) + Ra[3 * i + 2] * tb[2]) + ta[i];
  }
}

void oracle_reg_one_nn_step(const float* src_xyz, size_t n_src, const float* tgt_xyz, size_t n_tgt,
                            const float* init_T, const oracle_reg_params* prm, uint32_t cand_id,
                            const oracle_nn_backend* nn, float* out_T, float* out_rmse,
                            uint32_t* out_inliers, uint32_t* out_best_hyp, int* out_ok, float* out_final_step) {
  const uint32_t n = (uint32_t)n_src;
  /* a search structure over the (fixed) target, built once and queried every pass -- what PCL's ICP
   * does with its KdTreeFLANN (global_registration.cpp:241-247) */
  void* nn_handle = (nn && nn->build && n_tgt >= 1) ? nn->build(tgt_xyz, n_tgt) : NULL;
  double Rc[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1}, tc[3] = {0, 0, 0}; /* current absolute T */
  if (init_T) {
    for (int i = 0; i < 3; ++i) {
      for (int j = 0; j < 3; ++j) Rc[3 * i + j] = (double)init_T[4 * i + j];
      tc[i] = (double)init_T[4 * i + 3];
    }
  }
  float* moved = (float*)malloc(sizeof(float) * 3 * (n_src ? n_src : 1));
  uint32_t* corr = (uint32_t*)malloc(sizeof(uint32_t) * (n_src ? n_src : 1));
  float* d2 = (float*)malloc(sizeof(float) * (n_src ? n_src : 1));
  float Rf[9], tf[3];
  double sum_d2 = 0.0;
  uint32_t best_inl = 0, best_h = UINT32_MAX;
  int ok = 0;

#define CAST_T()                                   \
  do {                                             \
    for (int i_ = 0; i_ < 9; ++i_) Rf[i_] = (float)Rc[i_]; \
    for (int i_ = 0; i_ < 3; ++i_) tf[i_] = (float)tc[i_]; \
  } while (0)
#define MOVE_AND_MATCH()                                                          \
  do {                                                                            \
    CAST_T();                                                                     \
    for (uint32_t i_ = 0; i_ < n; ++i_) xform_f32(Rf, tf, src_xyz + 3 * (size_t)i_, moved + 3 * (size_t)i_); \
    if (nn_handle)                                                                \
      nn->query(nn_handle, moved, n_src, corr, d2);                               \
    else if (n_src * n_tgt > (size_t)64 * 1024 * 1024)                            \
      oracle_nn3_grid(moved, n_src, tgt_xyz, n_tgt, corr, d2);                    \
    else                                                                          \
      oracle_nn3(moved, n_src, tgt_xyz, n_tgt, corr, d2);                         \
    sum_d2 = 0.0;                                                                 \
    for (uint32_t i_ = 0; i_ < n; ++i_) sum_d2 += (double)d2[i_];                 \
  } while (0)

  if (n >= 3 && n_tgt >= 1 && prm->ransac_iters > 0) {
    /* S1 under T0, S2 on (T0 p, q) pairs */
    MOVE_AND_MATCH();
    float bR[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1}, bt[3] = {0, 0, 0};
    /* sequential RANSAC with OpenCV's adaptive iteration count (the reference calls
     * cv::estimateAffinePartial2D with its default confidence 0.99, loop_detector.cpp:256-257):
     * every improvement may lower the number of iterations still to run */
    const int adaptive = prm->ransac_confidence > 0.0f && prm->ransac_confidence < 1.0f;
    uint32_t niters = prm->ransac_iters;
    for (uint32_t h = 0; h < niters; ++h) {
      float R[9], t[3];
      if (!oracle_ransac_hypothesis(moved, tgt_xyz, corr, n, prm->seed, cand_id, h, R, t)) continue;
      const uint32_t inl = oracle_count_inliers(moved, tgt_xyz, corr, n, R, t, prm->inlier_thresh);
      if (inl > best_inl) { /* tie -> min h */
        best_inl = inl;
        best_h = h;
        memcpy(bR, R, sizeof(bR));
        memcpy(bt, t, sizeof(bt));
        if (adaptive) {
          const uint32_t need = oracle_ransac_needed_iters(inl, n, prm->ransac_confidence, prm->ransac_iters);
          if (need < niters) niters = need;
        }
      }
    }
    uint32_t min_inl = (uint32_t)(prm->min_inlier_ratio * (float)n);
    if (min_inl < 3) min_inl = 3;
    ok = best_inl >= min_inl;
    if (best_h != UINT32_MAX) {
      /* refit on the inliers of the best hypothesis, then T <- T_r * T0 */
      double Rr[9], tr[3], Rn[9], tn[3];
      const float thr2 = prm->inlier_thresh * prm->inlier_thresh;
      const uint32_t used = kabsch_pairs(moved, tgt_xyz, corr, n, bR, bt, thr2, Rr, tr, NULL);
      if (used < 3) {
        for (int i = 0; i < 9; ++i) Rr[i] = (double)bR[i];
        for (int i = 0; i < 3; ++i) tr[i] = (double)bt[i];
      }
      compose(Rr, tr, Rc, tc, Rn, tn);
      memcpy(Rc, Rn, sizeof(Rn));
      memcpy(tc, tn, sizeof(tn));
    }
  }

  const float gate2 = prm->max_corr_dist > 0.0f ? prm->max_corr_dist * prm->max_corr_dist : 0.0f;
  double last_step = 0.0; /* of the last ICP update (0 without one) */
  static const float I9[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1}, Z3[3] = {0, 0, 0};
  for (uint32_t it = 0; it < prm->icp_iters && n >= 3 && n_tgt >= 1; ++it) {
    MOVE_AND_MATCH();
    double Rd[9], td[3], Rn[9], tn[3];
    const uint32_t used = kabsch_pairs(moved, tgt_xyz, corr, n, I9, Z3, gate2, Rd, td, &last_step);
    if (used < 3) { /* the ICP stops for want of correspondences: NOT converged, whatever its earlier updates were */
      last_step = (double)INFINITY;
      break;
    }
    compose(Rd, td, Rc, tc, Rn, tn);
    memcpy(Rc, Rn, sizeof(Rn));
    memcpy(tc, tn, sizeof(tn));
  }
  CAST_T();
  for (int i = 0; i < 3; ++i) {
    for (int j = 0; j < 3; ++j) out_T[4 * i + j] = Rf[3 * i + j];
    out_T[4 * i + 3] = tf[i];
  }
  out_T[12] = 0; out_T[13] = 0; out_T[14] = 0; out_T[15] = 1;
  const float rmse = n ? (float)sqrt(sum_d2 / (double)n) : 0.0f;
  if (out_rmse) *out_rmse = rmse;
  if (out_inliers) *out_inliers = best_inl;
  if (out_best_hyp) *out_best_hyp = best_h;
  if (out_final_step) *out_final_step = (float)last_step;
  /* plausibility of the estimate (the analogue of the reference's |1 - scale| < 0.1, loop_detector.cpp:268-272):
   * the ICP must have CONVERGED -- its last update moved the scan by no more than max_final_step (RMS over the
   * matched points) -- and, optionally, the final rmse is bounded */
  if (out_ok)
    *out_ok = ok && !(prm->max_rmse > 0.0f && !(rmse <= prm->max_rmse)) &&
              !(prm->max_final_step > 0.0f && prm->icp_iters > 0 && !((float)last_step <= prm->max_final_step));
  free(moved);
  free(corr);
  free(d2);
  if (nn_handle) nn->free_(nn_handle);
#undef CAST_T
#undef MOVE_AND_MATCH
}

void oracle_reg_one_nn(const float* src_xyz, size_t n_src, const float* tgt_xyz, size_t n_tgt,
                       const float* init_T, const oracle_reg_params* prm, uint32_t cand_id,
                       const oracle_nn_backend* nn, float* out_T, float* out_rmse,
                       uint32_t* out_inliers, uint32_t* out_best_hyp, int* out_ok) {
  oracle_reg_one_nn_step(src_xyz, n_src, tgt_xyz, n_tgt, init_T, prm, cand_id, nn, out_T, out_rmse, out_inliers,
                         out_best_hyp, out_ok, NULL);
}

void oracle_reg_one(const float* src_xyz, size_t n_src, const float* tgt_xyz, size_t n_tgt,
                    const float* init_T, const oracle_reg_params* prm, uint32_t cand_id,
                    float* out_T, float* out_rmse, uint32_t* out_inliers, uint32_t* out_best_hyp,
                    int* out_ok) {
  oracle_reg_one_nn(src_xyz, n_src, tgt_xyz, n_tgt, init_T, prm, cand_id, NULL, out_T, out_rmse,
                    out_inliers, out_best_hyp, out_ok);
}

/* Candidates of one query over threads (the all-cores leg of bench.py's cpu_baseline). */
typedef struct {
  const float* src;
  size_t n_src;
  const float* const* tgt;
  const size_t* n_tgt;
  size_t n_cand, first, stride;
  const oracle_reg_params* prm;
  const uint32_t* cand_ids;
  const oracle_nn_backend* nn;
  float* T;
  float* rmse;
  uint32_t* inl;
  int* ok;
} reg_mt_arg;

static void* reg_mt_worker(void* p) {
  reg_mt_arg* a = (reg_mt_arg*)p;
  for (size_t c = a->first; c < a->n_cand; c += a->stride) {
    uint32_t hyp;
    oracle_reg_one_nn(a->src, a->n_src, a->tgt[c], a->n_tgt[c], NULL, a->prm,
                      a->cand_ids ? a->cand_ids[c] : (uint32_t)c, a->nn, a->T + 16 * c, a->rmse + c,
                      a->inl + c, &hyp, a->ok + c);
  }
  return NULL;
}

void oracle_reg_many_mt(const float* src_xyz, size_t n_src, const float* const* tgt_xyz,
                        const size_t* n_tgt, size_t n_cand, const oracle_reg_params* prm,
                        const uint32_t* cand_ids, const oracle_nn_backend* nn, int threads,
                        float* out_T, float* out_rmse, uint32_t* out_inliers, int* out_ok) {
  if (threads < 1) threads = 1;
  if ((size_t)threads > n_cand) threads = (int)n_cand;
  pthread_t* th = (pthread_t*)malloc(sizeof(pthread_t) * (size_t)(threads ? threads : 1));
  reg_mt_arg* args = (reg_mt_arg*)malloc(sizeof(reg_mt_arg) * (size_t)(threads ? threads : 1));
  for (int t = 0; t < threads; ++t) {
    args[t] = (reg_mt_arg){src_xyz, n_src, tgt_xyz, n_tgt, n_cand, (size_t)t, (size_t)threads, prm,
                           cand_ids, nn, out_T, out_rmse, out_inliers, out_ok};
    pthread_create(&th[t], NULL, reg_mt_worker, &args[t]);
  }
  for (int t = 0; t < threads; ++t) pthread_join(th[t], NULL);
  free(th);
  free(args);
}

void oracle_pose_error(const float* Tg, const float* Te, float* err_rot_deg, float* err_pos) {
  /* global_localization.cpp:291-306: err_R = gt_rot^T * R_restored */
  float tr = 0.0f;
  for (int i = 0; i < 3; ++i)
    for (int k = 0; k < 3; ++k) tr += Tg[4 * k + i] * Te[4 * k + i];
  float off = 0.5f * (tr - 1.0f);
  if (off < -0.999999f) off = -0.999999f;
  if (off > 0.999999f) off = 0.999999f;
  float er = fabsf(acosf(off)) * (float)(180.0 / M_PI);
  const float dx = Tg[3] - Te[3], dy = Tg[7] - Te[7], dz = Tg[11] - Te[11];
  if (fabsf(er - 180.0f) < 5.0f) er = fabsf(er - 180.0f);
  *err_rot_deg = er;
  *err_pos = sqrtf(dx * dx + dy * dy + dz * dz);
}
