/*
 * knn_oracle.c -- CPU restatement of the reference's descriptor kNN (TEST INFRASTRUCTURE ONLY,
 * see gloc_oracle.h).  Plain C, no dependencies.  Build with -ffp-contract=off.
 *
 * Follows:
 *   registration/nanoflann.hpp:453-487   L2_Adaptor::evalMetric (accumulation order)
 *   registration/nanoflann.hpp:200-235   KNNResultSet::addPoint / worstDist (result-set semantics)
 *   registration/nanoflann.hpp:1448-1468 findNeighbors (exact search, eps = 0)
 *   registration/KDTreeVectorOfVectorsAdaptor.h:95-102 query()
 *   registration/global_localization.cpp:221-268, main.py:336-348  recall@N
 *
 * The KD-tree traversal of the reference visits rows in a tree-dependent order; because the search
 * is exact, its result equals the exhaustive scan below except for the order among rows at
 * exactly equal distance (nanoflann.hpp:212 strict '>').  The scan visits rows in ascending index,
 * so equal distances come out in ascending index.
 */
#include "gloc_oracle.h"

#include <float.h>
#include <pthread.h>
#include <stdlib.h>
#include <string.h>

float oracle_l2_eval(const float* a, const float* b, size_t dim) {
  float result = 0.0f;
  size_t d = 0;
  /* nanoflann.hpp:463-477: "while (a < lastgroup)", lastgroup = last - 3 */
  while (d + 3 < dim) {
    const float diff0 = a[d + 0] - b[d + 0];
    const float diff1 = a[d + 1] - b[d + 1];
    const float diff2 = a[d + 2] - b[d + 2];
    const float diff3 = a[d + 3] - b[d + 3];
    result += diff0 * diff0 + diff1 * diff1 + diff2 * diff2 + diff3 * diff3;
    d += 4;
  }
  /* nanoflann.hpp:480-485 */
  while (d < dim) {
    const float diff0 = a[d] - b[d];
    result += diff0 * diff0;
    d++;
  }
  return result;
}

/* nanoflann.hpp:200-233 with count/capacity kept by the caller */
static void result_add(uint64_t* indices, float* dists, size_t capacity, size_t* count, float dist,
                       uint64_t index) {
  size_t i;
  for (i = *count; i > 0; --i) {
    if (dists[i - 1] > dist) {
      if (i < capacity) {
        dists[i] = dists[i - 1];
        indices[i] = indices[i - 1];
      }
    } else {
      break;
    }
  }
  if (i < capacity) {
    dists[i] = dist;
    indices[i] = index;
  }
  if (*count < capacity) (*count)++;
}

static void knn_one(const float* db, size_t dim, const float* q, size_t k, size_t first_row,
                    size_t last_row, uint64_t* out_idx, float* out_d2) {
  size_t count = 0;
  for (size_t i = 0; i < k; ++i) {
    out_idx[i] = UINT64_MAX;
    out_d2[i] = FLT_MAX; /* nanoflann.hpp:188: dists[capacity-1] = max */
  }
  if (k == 0) return;
  for (size_t j = first_row; j < last_row; ++j) {
    const float d2 = oracle_l2_eval(q, db + j * dim, dim);
    /* searchLevel only offers a point when dist < worstDist (nanoflann.hpp:1611) */
    if (d2 < out_d2[k - 1]) result_add(out_idx, out_d2, k, &count, d2, (uint64_t)j);
  }
}

void oracle_knn_search(const float* db, size_t n_rows, size_t dim, const float* queries, size_t nq,
                       size_t k, size_t first_row, size_t last_row, uint64_t* out_idx,
                       float* out_d2) {
  if (last_row > n_rows) last_row = n_rows;
  if (first_row > last_row) first_row = last_row;
  for (size_t qi = 0; qi < nq; ++qi)
    knn_one(db, dim, queries + qi * dim, k, first_row, last_row, out_idx + qi * k, out_d2 + qi * k);
}

typedef struct {
  const float* db;
  size_t dim;
  const float* queries;
  size_t k, first_row, last_row;
  uint64_t* out_idx;
  float* out_d2;
  size_t q_begin, q_end;
} knn_job;

static void* knn_worker(void* arg) {
  knn_job* j = (knn_job*)arg;
  for (size_t qi = j->q_begin; qi < j->q_end; ++qi)
    knn_one(j->db, j->dim, j->queries + qi * j->dim, j->k, j->first_row, j->last_row,
            j->out_idx + qi * j->k, j->out_d2 + qi * j->k);
  return NULL;
}

void oracle_knn_search_mt(const float* db, size_t n_rows, size_t dim, const float* queries,
                          size_t nq, size_t k, size_t first_row, size_t last_row,
                          uint64_t* out_idx, float* out_d2, int n_threads) {
  if (last_row > n_rows) last_row = n_rows;
  if (first_row > last_row) first_row = last_row;
  if (n_threads < 1) n_threads = 1;
  if ((size_t)n_threads > nq) n_threads = (int)(nq ? nq : 1);
  pthread_t* th = (pthread_t*)malloc(sizeof(pthread_t) * (size_t)n_threads);
  knn_job* jobs = (knn_job*)malloc(sizeof(knn_job) * (size_t)n_threads);
  for (int t = 0; t < n_threads; ++t) {
    knn_job jb = {db, dim, queries, k, first_row, last_row, out_idx, out_d2,
                  nq * (size_t)t / (size_t)n_threads, nq * (size_t)(t + 1) / (size_t)n_threads};
    jobs[t] = jb;
    pthread_create(&th[t], NULL, knn_worker, &jobs[t]);
  }
  for (int t = 0; t < n_threads; ++t) pthread_join(th[t], NULL);
  free(th);
  free(jobs);
}

size_t oracle_recall_at(const uint64_t* idx, size_t nq, size_t k, const uint64_t* pos,
                        const size_t* pos_off, const int* k_values, size_t n_k, float* recalls) {
  size_t valid = 0;
  for (size_t i = 0; i < n_k; ++i) recalls[i] = 0.0f;
  for (size_t q = 0; q < nq; ++q) {
    const size_t pb = pos_off[q], pe = pos_off[q + 1];
    if (pb == pe) continue; /* global_localization.cpp:226 */
    valid++;
    for (size_t ki = 0; ki < n_k; ++ki) {
      size_t top = (size_t)k_values[ki];
      if (top > k) top = k;
      int hit = 0;
      for (size_t j = 0; j < top && !hit; ++j)
        for (size_t p = pb; p < pe; ++p)
          if (pos[p] == idx[q * k + j]) {
            hit = 1;
            break;
          }
      if (hit) recalls[ki] += 1.0f;
    }
  }
  if (valid > 0)
    for (size_t i = 0; i < n_k; ++i) recalls[i] /= (float)valid;
  return valid;
}

/* ---- deterministic synthetic inputs --------------------------------------------------------
 * splitmix64 finaliser as a random-access counter RNG; the "gaussian" is an Irwin-Hall sum of the
 * draw's four 16-bit fields, so every platform (C, numpy, HIP) gets the same bits with no libm. */
static inline uint64_t mix64(uint64_t z) {
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
  return z ^ (z >> 31);
}

uint64_t oracle_rng_key(uint64_t seed, uint64_t stream) {
  return mix64(mix64(seed + 0x9E3779B97F4A7C15ULL) ^ (stream * 0xD1B54A32D192ED03ULL + 1ULL));
}

uint64_t oracle_rng_draw(uint64_t key, uint64_t ctr) {
  return mix64(key + (ctr + 1ULL) * 0x9E3779B97F4A7C15ULL);
}

float oracle_rng_gauss(uint64_t key, uint64_t ctr) {
  const uint64_t u = oracle_rng_draw(key, ctr);
  const int32_t s = (int32_t)((u & 0xFFFF) + ((u >> 16) & 0xFFFF) + ((u >> 32) & 0xFFFF) +
                              ((u >> 48) & 0xFFFF)) -
                    131070;
  return (float)s * (1.0f / 37837.227f);
}

void oracle_synth_iid(uint64_t seed, size_t first_row, size_t n_rows, size_t dim, float* out) {
  const float scale = (float)(1.0 / __builtin_sqrt((double)dim));
  for (size_t r = 0; r < n_rows; ++r) {
    const uint64_t key = oracle_rng_key(seed, (uint64_t)(first_row + r));
    for (size_t c = 0; c < dim; ++c) out[r * dim + c] = oracle_rng_gauss(key, (uint64_t)c) * scale;
  }
}
