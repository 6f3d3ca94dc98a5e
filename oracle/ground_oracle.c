/*
 * ground_oracle.c -- CPU restatement of the ground pre-alignment ("next" row N3).
 *
 * TEST INFRASTRUCTURE ONLY (see gloc_oracle.h).  PARITY UNPINNED: the reference's implementation
 * (registration/ground_estimator.cpp) is a thin driver over PCL (NormalEstimationOMP, KdTree,
 * RandomSampleConsensus<SampleConsensusModelPlane>) and Eigen, none of which are in this image; PCL's
 * choices that the reference does not fix (neighbour order among equal distances, the eigen-solver,
 * the random sampler) are replaced by this repository's deterministic ones, stated at each step.
 *
 * Steps (all line numbers: registration/ground_estimator.cpp)
 *   G1  keep points with x*x + y*y + z*z < 400                               :201-210
 *   G2  k = 10 nearest neighbours of every kept point among the kept points  :72-80
 *   G3  normal = smallest-eigenvalue eigenvector of the neighbours' covariance, flipped towards the
 *       sensor origin (pcl::NormalEstimation: computePointNormal + flipNormalTowardsViewpoint)
 *   G4  theta = (atan2(nz, sqrt(nx^2+ny^2)) + pi/2) in degrees, bin = floor(theta / 10)   :86-102
 *   G5  ground bin = fullest bin outside 5..12                                :104-127
 *   G6  plane RANSAC over that bin's points, threshold 0.1, no refit          :19-36
 *   G7  T_l2g: rotate the (upward) plane normal onto z, drop the yaw, lift by the sensor height :163-194
 */
#include "gloc_oracle.h"

#include <float.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>

#define GROUND_STREAM 0x47524E44u /* 'GRND': the RNG stream of the plane sampler */

static float d2f(const float* a, const float* b) {
  const float dx = a[0] - b[0], dy = a[1] - b[1], dz = a[2] - b[2];
  return (dx * dx + dy * dy) + dz * dz;
}

/* G2.  Exhaustive, targets in index order, strict '<' insertion: equal distances keep the smaller
 * index first (PCL/FLANN leave that order unspecified). */
void oracle_ground_knn(const float* xyz, size_t n, uint32_t k, uint32_t* idx, float* d2) {
  for (size_t i = 0; i < n; ++i) {
    uint32_t* bi = idx + i * k;
    float* bd = d2 + i * k;
    for (uint32_t s = 0; s < k; ++s) { bi[s] = UINT32_MAX; bd[s] = FLT_MAX; }
    for (size_t j = 0; j < n; ++j) {
      const float d = d2f(xyz + 3 * i, xyz + 3 * j);
      if (!(d < bd[k - 1])) continue;
      uint32_t s = k - 1;
      while (s > 0 && d < bd[s - 1]) { bd[s] = bd[s - 1]; bi[s] = bi[s - 1]; --s; }
      bd[s] = d; bi[s] = (uint32_t)j;
    }
  }
}

/* sin of the bin edges -80, -70, ..., +80 degrees: bin b holds elevations [10b - 90, 10b - 80).
 * The reference bins atan2() of the normal; the elevation's sine is nz / |n|, and comparing it with
 * these constants gives the same bin without a transcendental call (CPU and GPU then agree). */
static const double kSinEdge[17] = {
    -0.98480775301220805937, -0.93969262078590838405, -0.86602540378443864676, -0.76604444311897803520,
    -0.64278760968653932632, -0.50000000000000000000, -0.34202014332566873304, -0.17364817766693034885,
    0.0,
    0.17364817766693034885,  0.34202014332566873304,  0.50000000000000000000,  0.64278760968653932632,
    0.76604444311897803520,  0.86602540378443864676,  0.93969262078590838405,  0.98480775301220805937};

static int elevation_bin(const double nrm[3]) {
  const double len = sqrt((nrm[0] * nrm[0] + nrm[1] * nrm[1]) + nrm[2] * nrm[2]);
  const double s = len > 0.0 ? nrm[2] / len : 0.0;
  int b = 0;
  while (b < 17 && s >= kSinEdge[b]) ++b;
  return b; /* 0..17 (the reference's floor() would give 18 for a normal exactly along +z: clamped) */
}

/* G3 + G4 for one point: neighbours in the given order; fewer than 3 valid neighbours -> (0,0,0), bin 9 */
static void normal_of(const float* xyz, const float* p, const uint32_t* nb, uint32_t k, double out[3]) {
  double mean[3] = {0, 0, 0};
  uint32_t m = 0;
  for (uint32_t s = 0; s < k; ++s) {
    if (nb[s] == UINT32_MAX) continue;
    const float* q = xyz + 3 * (size_t)nb[s];
    for (int a = 0; a < 3; ++a) mean[a] += (double)q[a];
    ++m;
  }
  out[0] = out[1] = out[2] = 0.0;
  if (m < 3) return;
  for (int a = 0; a < 3; ++a) mean[a] = mean[a] / (double)m;
  double C[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
  for (uint32_t s = 0; s < k; ++s) {
    if (nb[s] == UINT32_MAX) continue;
    const float* q = xyz + 3 * (size_t)nb[s];
    const double d[3] = {(double)q[0] - mean[0], (double)q[1] - mean[1], (double)q[2] - mean[2]};
    for (int a = 0; a < 3; ++a)
      for (int b = a; b < 3; ++b) C[3 * a + b] += d[a] * d[b];
  }
  C[3] = C[1]; C[6] = C[2]; C[7] = C[5];
  double V[9];
  oracle_jacobi_eig3(C, V);
  int col = 0; /* smallest eigenvalue; ties keep the lower column */
  if (C[4] < C[3 * col + col]) col = 1;
  if (C[8] < C[3 * col + col]) col = 2;
  double nrm[3] = {V[0 + col], V[3 + col], V[6 + col]};
  /* flipNormalTowardsViewpoint with the viewpoint at the origin: (0 - p) . n < 0 -> flip */
  const double dot = ((-(double)p[0]) * nrm[0] + (-(double)p[1]) * nrm[1]) + (-(double)p[2]) * nrm[2];
  if (dot < 0.0) { nrm[0] = -nrm[0]; nrm[1] = -nrm[1]; nrm[2] = -nrm[2]; }
  out[0] = nrm[0]; out[1] = nrm[1]; out[2] = nrm[2];
}

void oracle_ground_normals(const float* xyz, size_t n, const uint32_t* knn_idx, uint32_t k,
                           float* normals, uint8_t* bins) {
  for (size_t i = 0; i < n; ++i) {
    double nrm[3];
    normal_of(xyz, xyz + 3 * i, knn_idx + i * k, k, nrm);
    for (int a = 0; a < 3; ++a) normals[3 * i + a] = (float)nrm[a];
    bins[i] = (uint8_t)elevation_bin(nrm);
  }
}

/* G7.  Eigen::Quaternionf::FromTwoVectors(n, z), toRotationMatrix(), eulerAngles(2, 1, 0) with
 * Eigen 3.3's branch (first angle in [0, pi]), cartographer's RollPitchYaw(roll, pitch, 0). */
void oracle_ground_transform_from_plane(const float plane[4], float* T16) {
  double n[3] = {plane[0], plane[1], plane[2]};
  const double len = sqrt((n[0] * n[0] + n[1] * n[1]) + n[2] * n[2]);
  const double height = fabs((double)plane[3]) / len; /* the ground is below the sensor */
  const double sgn = plane[2] < 0.f ? -1.0 : 1.0;     /* upward normal */
  for (int a = 0; a < 3; ++a) n[a] = sgn * n[a] / len;
  /* FromTwoVectors(n, z): axis = n x z, w = 1 + n.z, normalised (n is never close to -z here:
   * the normal was flipped upward; the antipodal case falls back to a rotation about x) */
  double q[4]; /* w x y z */
  const double c = n[2];
  if (c < -1.0 + 1e-12) {
    q[0] = 0; q[1] = 1; q[2] = 0; q[3] = 0;
  } else {
    const double ax[3] = {n[1] * 1.0 - n[2] * 0.0, n[2] * 0.0 - n[0] * 1.0, 0.0};
    const double s = sqrt((1.0 + c) * 2.0), inv = 1.0 / s;
    q[0] = s * 0.5; q[1] = ax[0] * inv; q[2] = ax[1] * inv; q[3] = ax[2] * inv;
  }
  const double qn = sqrt(((q[0] * q[0] + q[1] * q[1]) + q[2] * q[2]) + q[3] * q[3]);
  for (int a = 0; a < 4; ++a) q[a] /= qn;
  const double w = q[0], x = q[1], y = q[2], z = q[3];
  const double R[9] = {1 - 2 * (y * y + z * z), 2 * (x * y - z * w),     2 * (x * z + y * w),
                       2 * (x * y + z * w),     1 - 2 * (x * x + z * z), 2 * (y * z - x * w),
                       2 * (x * z - y * w),     2 * (y * z + x * w),     1 - 2 * (x * x + y * y)};
  /* eulerAngles(2, 1, 0): i = 2, j = 1, k = 0, odd */
  double e0 = atan2(R[3 * 1 + 0], R[3 * 0 + 0]);
  const double c2 = sqrt(R[3 * 2 + 2] * R[3 * 2 + 2] + R[3 * 2 + 1] * R[3 * 2 + 1]);
  double e1;
  if (e0 < 0.0) {
    e0 += M_PI;
    e1 = atan2(-R[3 * 2 + 0], -c2);
  } else {
    e1 = atan2(-R[3 * 2 + 0], c2);
  }
  const double s1 = sin(e0), c1 = cos(e0);
  const double e2 = atan2(s1 * R[3 * 0 + 2] - c1 * R[3 * 1 + 2], c1 * R[3 * 1 + 1] - s1 * R[3 * 0 + 1]);
  /* RollPitchYaw(roll = e2, pitch = e1, yaw = 0) = Ry(pitch) * Rx(roll) */
  const double cp = cos(e1), sp = sin(e1), cr = cos(e2), sr = sin(e2);
  const double Rn[9] = {cp, sp * sr, sp * cr, 0.0, cr, -sr, -sp, cp * sr, cp * cr};
  for (int i = 0; i < 16; ++i) T16[i] = 0.f;
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) T16[4 * i + j] = (float)Rn[3 * i + j];
  T16[11] = (float)height;
  T16[15] = 1.f;
}

static void identity16(float* T) {
  for (int i = 0; i < 16; ++i) T[i] = (i % 5 == 0) ? 1.f : 0.f;
}

int oracle_ground_estimate(const float* xyz, size_t n, size_t stride, const oracle_ground_params* prm,
                           float* T16, oracle_ground_info* info) {
  memset(info, 0, sizeof(*info));
  info->ground_bin = -1;
  identity16(T16);
  /* G1 */
  float* near = (float*)malloc((n ? n : 1) * 3 * sizeof(float));
  size_t m = 0;
  for (size_t i = 0; i < n; ++i) {
    const float x = xyz[i * stride], y = xyz[i * stride + 1], z = xyz[i * stride + 2];
    if ((x * x + y * y) + z * z < prm->near_range2) {
      near[3 * m] = x; near[3 * m + 1] = y; near[3 * m + 2] = z;
      ++m;
    }
  }
  info->n_near = (uint32_t)m;
  if (m < 3) { free(near); return 0; }
  /* G2..G4 */
  const uint32_t k = prm->knn;
  uint32_t* nb = (uint32_t*)malloc(m * k * sizeof(uint32_t));
  float* nd = (float*)malloc(m * k * sizeof(float));
  float* normals = (float*)malloc(m * 3 * sizeof(float));
  uint8_t* bins = (uint8_t*)malloc(m);
  oracle_ground_knn(near, m, k, nb, nd);
  oracle_ground_normals(near, m, nb, k, normals, bins);
  for (size_t i = 0; i < m; ++i) info->hist[bins[i]]++;
  /* G5: the fullest bin outside 5..12 (argsort descending, first admissible; ties -> lower bin) */
  int gb = -1;
  for (int b = 0; b < 18; ++b) {
    if (b > 4 && b < 13) continue;
    if (gb < 0 || info->hist[b] > info->hist[gb]) gb = b;
  }
  if (gb < 0 || info->hist[gb] < 3) { free(near); free(nb); free(nd); free(normals); free(bins); return 0; }
  info->ground_bin = gb;
  float* g = (float*)malloc((size_t)info->hist[gb] * 3 * sizeof(float));
  uint32_t ng = 0;
  for (size_t i = 0; i < m; ++i)
    if (bins[i] == gb) { memcpy(g + 3 * ng, near + 3 * i, 3 * sizeof(float)); ++ng; }
  info->n_ground = ng;
  /* G6: plane through 3 sampled points, |a x + b y + c z + d| < thr, first strictly better wins,
   * iteration count adapted like pcl::RandomSampleConsensus (k = log(1-p) / log(1-w^3)) */
  uint32_t best_h = UINT32_MAX, best_inl = 0, niters = prm->ransac_iters, h = 0;
  float best_pl[4] = {0, 0, 0, 0};
  for (; h < niters; ++h) {
    uint32_t s[3];
    oracle_ransac_sample(prm->seed, GROUND_STREAM, h, ng, s);
    if (s[0] == s[1] || s[0] == s[2] || s[1] == s[2]) continue;
    const float *p0 = g + 3 * s[0], *p1 = g + 3 * s[1], *p2 = g + 3 * s[2];
    const double a[3] = {(double)p1[0] - p0[0], (double)p1[1] - p0[1], (double)p1[2] - p0[2]};
    const double b[3] = {(double)p2[0] - p0[0], (double)p2[1] - p0[1], (double)p2[2] - p0[2]};
    const double c[3] = {a[1] * b[2] - a[2] * b[1], a[2] * b[0] - a[0] * b[2], a[0] * b[1] - a[1] * b[0]};
    const double aa = (a[0] * a[0] + a[1] * a[1]) + a[2] * a[2];
    const double bb = (b[0] * b[0] + b[1] * b[1]) + b[2] * b[2];
    const double cc = (c[0] * c[0] + c[1] * c[1]) + c[2] * c[2];
    if (!(aa > 1e-12) || !(bb > 1e-12) || !(cc > 1e-6 * (aa * bb))) continue; /* collinear sample */
    const double len = sqrt(cc);
    const float pl[4] = {(float)(c[0] / len), (float)(c[1] / len), (float)(c[2] / len),
                         (float)(-((c[0] / len * p0[0] + c[1] / len * p0[1]) + c[2] / len * p0[2]))};
    uint32_t inl = 0;
    for (uint32_t i = 0; i < ng; ++i) {
      const float* q = g + 3 * i;
      const float dist = ((pl[0] * q[0] + pl[1] * q[1]) + pl[2] * q[2]) + pl[3];
      if (fabsf(dist) < prm->plane_thresh) ++inl;
    }
    if (inl > best_inl) {
      best_inl = inl; best_h = h;
      memcpy(best_pl, pl, sizeof(pl));
      if (prm->ransac_conf > 0.f && prm->ransac_conf < 1.f) {
        const uint32_t need = oracle_ransac_needed_iters(inl, ng, prm->ransac_conf, prm->ransac_iters);
        if (need < niters) niters = need;
      }
    }
  }
  info->best_hyp = best_h; info->inliers = best_inl; info->iters_used = niters;
  if (best_h != UINT32_MAX) {
    memcpy(info->plane, best_pl, sizeof(best_pl));
    oracle_ground_transform_from_plane(best_pl, T16);
    info->found = 1;
  }
  free(g); free(near); free(nb); free(nd); free(normals); free(bins);
  return info->found;
}
