// ground_kernels.hpp -- HIP kernels of the ground pre-alignment ("next" row N3 of SURVEY.md 8f).
// GroundEstimator::EsitmateGroundAndTransform (registration/ground_estimator.cpp:196-228): keep the
// points within 20 m, estimate a normal per point from its 10 nearest neighbours, take the fullest
// 10-degree elevation bin outside 5..12 as "ground", fit a plane to it by RANSAC and build the
// roll/pitch/height transform T_l2g.  The reference delegates every numeric step to PCL / Eigen; the
// arithmetic here is the one stated in oracle/ground_oracle.c (steps G1..G7), bit for bit where it is
// + - * / sqrt (neighbours, normals, bins, plane hypotheses, inlier counts).
//
// The heavy step is G2: exact 10-NN of ~30-60 k points among themselves.  It is done exhaustively:
// 1e9..4e9 pairs at 8 flop each, targets staged through LDS as wave-uniform float4 reads, a sorted
// top-k list per lane in registers whose insertion path is entered only when a lane's candidate
// beats its current k-th distance (rare after the first few hundred targets).
#pragma once
#include <hip/hip_runtime.h>

#include <cfloat>
#include <cstdint>

#include "math3.hpp"          // f32x4, dist2, xform, jacobi_eig3, cross3, mulhi_idx, NN_FAR
#include "synth_kernels.hpp"  // rng_key / rng_draw

namespace gloc {
namespace ground {

using reg::f32x4;

constexpr int KMAX = 16;          // neighbours per point the register list can hold
constexpr int KNN_TILE = 256;     // targets per LDS tile
constexpr int KNN_BLOCK = 128;    // sources per work-group (one per thread)
constexpr uint32_t GROUND_STREAM = 0x47524E44u;  // 'GRND': RNG stream of the plane sampler
constexpr float NN_FAR_ = reg::NN_FAR;

// G1: flag the points within the range filter (x*x + y*y + z*z < r2, fp32, left to right).
__global__ __launch_bounds__(256) void near_flag_kernel(const float* __restrict__ xyz, uint32_t n, int stride,
                                                         float r2, uint8_t* __restrict__ flag) {
  const uint32_t i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const float* p = xyz + (size_t)i * stride;
  const float x = p[0], y = p[1], z = p[2];
  flag[i] = ((x * x + y * y) + z * z < r2) ? 1 : 0;
}

// Gather selected points (ascending original index) as (x, y, z, bits(original index)).
__global__ __launch_bounds__(256) void gather_points_kernel(const float* __restrict__ xyz, int stride,
                                                             const uint32_t* __restrict__ sel, uint32_t m,
                                                             f32x4* __restrict__ out) {
  const uint32_t i = blockIdx.x * 256 + threadIdx.x;
  if (i >= m) return;
  const uint32_t j = sel[i];
  const float* p = xyz + (size_t)j * stride;
  out[i] = f32x4{p[0], p[1], p[2], __uint_as_float(j)};
}

// G2: exact k nearest neighbours of every point among the same points, ascending (d2, index).
// Targets arrive in index order and insertion needs a strictly smaller distance, so equal distances
// keep the smaller index first.  The targets are cut into gridDim.y slices (more waves than m / 64
// alone would give; a wave here is latency-bound); slice s writes its own sorted list to
// [s][m][k] and knn_merge_kernel folds the slices.
__global__ __launch_bounds__(KNN_BLOCK) void knn_self_kernel(const f32x4* __restrict__ pts, uint32_t m, int k,
                                                             uint32_t slice_len, uint32_t* __restrict__ idx,
                                                             float* __restrict__ d2) {
  __shared__ f32x4 tile[KNN_TILE];
  const uint32_t i = blockIdx.x * KNN_BLOCK + threadIdx.x;
  const uint32_t s_begin = blockIdx.y * slice_len;
  const uint32_t s_end = (s_begin + slice_len) < m ? (s_begin + slice_len) : m;
  const f32x4 p = pts[i < m ? i : m - 1];
  float bd[KMAX];
  uint32_t bi[KMAX];
#pragma unroll
  for (int s = 0; s < KMAX; ++s) { bd[s] = FLT_MAX; bi[s] = 0xFFFFFFFFu; }
  float worst = FLT_MAX;  // bd[k - 1]
  auto insert = [&](float d, uint32_t j) {
    float cd = d;
    uint32_t ci = j;
    bool shifting = false;  // from the first entry the candidate beats, everything moves back one
#pragma unroll
    for (int s = 0; s < KMAX; ++s) {
      if (s < k) {
        shifting = shifting || (cd < bd[s]);
        if (shifting) {
          const float td = bd[s]; const uint32_t ti = bi[s];
          bd[s] = cd; bi[s] = ci;
          cd = td; ci = ti;
        }
      }
    }
#pragma unroll
    for (int s = 0; s < KMAX; ++s)
      if (s == k - 1) worst = bd[s];
  };
  for (uint32_t t0 = s_begin; t0 < s_end; t0 += KNN_TILE) {
    __syncthreads();
    for (uint32_t t = threadIdx.x; t < KNN_TILE; t += KNN_BLOCK) {
      f32x4 v = {NN_FAR_, NN_FAR_, NN_FAR_, 0.f};  // padding: farther than any real point, never inserted
      if (t0 + t < s_end) v = pts[t0 + t];
      tile[t] = v;
    }
    __syncthreads();
    // eight targets per step: the distances are independent (ILP), the insertion path is rare
    for (uint32_t t = 0; t < KNN_TILE; t += 8) {
      float d[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const f32x4 q = tile[t + u];
        d[u] = reg::dist2(p.x, p.y, p.z, q.x, q.y, q.z);
      }
      const float dm = fminf(fminf(fminf(d[0], d[1]), fminf(d[2], d[3])), fminf(fminf(d[4], d[5]), fminf(d[6], d[7])));
      if (dm < worst) {
#pragma unroll
        for (int u = 0; u < 8; ++u)
          if (d[u] < worst && t0 + t + u < s_end) insert(d[u], t0 + t + u);
      }
    }
  }
  if (i < m) {
    const size_t o = ((size_t)blockIdx.y * m + i) * k;
#pragma unroll
    for (int s = 0; s < KMAX; ++s)
      if (s < k) {
        idx[o + s] = bi[s];
        d2[o + s] = bd[s];
      }
  }
}

// Fold the per-slice lists: thread per point, the k smallest by (d2, index) over n_slices sorted lists
// (slice order = index order, so on equal distances the earlier slice's entry goes first).
__global__ __launch_bounds__(256) void knn_merge_kernel(const uint32_t* __restrict__ pidx, const float* __restrict__ pd2,
                                                         uint32_t m, int k, int n_slices, uint32_t* __restrict__ idx,
                                                         float* __restrict__ d2) {
  const uint32_t i = blockIdx.x * 256 + threadIdx.x;
  if (i >= m) return;
  int head[32];  // n_slices <= 32
  for (int s = 0; s < n_slices; ++s) head[s] = 0;
  for (int r = 0; r < k; ++r) {
    float best = FLT_MAX;
    uint32_t bj = 0xFFFFFFFFu;
    int bs = -1;
    for (int s = 0; s < n_slices; ++s) {
      if (head[s] >= k) continue;
      const size_t o = ((size_t)s * m + i) * k + head[s];
      const uint32_t j = pidx[o];
      const float d = pd2[o];
      if (j != 0xFFFFFFFFu && d < best) { best = d; bj = j; bs = s; }  // strict: earlier slice wins ties
    }
    idx[(size_t)i * k + r] = bj;
    d2[(size_t)i * k + r] = bs >= 0 ? best : FLT_MAX;
    if (bs >= 0) head[bs]++;
  }
}

// ---- culled form of G2 ---------------------------------------------------------------------------
// The kept points are Hilbert-sorted and cut into chunks of 64 with bounding boxes (the machinery of
// the registration's culled 1-NN).  A wave takes one chunk as its 64 sources; it fills every lane's
// list from the own chunk, then sweeps the chunk boxes 64 per ballot against (own box, the wave's
// largest k-th distance) and evaluates only chunks some lane can still improve on.  Same lists as
// knn_self_kernel: ascending (d2, index) with the index of the UNSORTED cloud as the tie-break, which
// the insertion compares explicitly (targets no longer arrive in index order).
constexpr int KCH = 64;  // points per chunk

__global__ __launch_bounds__(256) void hilbert_keys_kernel(const f32x4* __restrict__ pts, uint32_t m, float origin,
                                                            float inv_cell, uint32_t* __restrict__ keys,
                                                            uint32_t* __restrict__ vals) {
  const uint32_t i = blockIdx.x * 256 + threadIdx.x;
  if (i >= m) return;
  const f32x4 p = pts[i];
  keys[i] = reg::morton_key(p.x, p.y, p.z, origin, origin, origin, inv_cell);
  vals[i] = i;
}

// ---- stable compaction of the flagged indices (round 4: hipcub::DeviceSelect::Flagged before) ---------------------------
// sel[0 .. count) = the indices i with flag[i] != 0, ascending.  Two launches, tiles of 2048 flags: a count per tile;
// then every tile sums the counts before it (a few hundred at most) and ranks its own flags with ballots -- waves in
// order, lanes in order -- so the order of the input is kept.
constexpr int SEL_TILE = 2048;

__global__ __launch_bounds__(256) void flag_count_kernel(const uint8_t* __restrict__ flag, uint32_t n,
                                                          uint32_t* __restrict__ tile_cnt) {
  __shared__ uint32_t wc[4];
  const uint32_t base = blockIdx.x * SEL_TILE;
  uint32_t c = 0;
#pragma unroll
  for (int k = 0; k < SEL_TILE / 256; ++k) {
    const uint32_t e = base + k * 256 + threadIdx.x;
    c += (uint32_t)__popcll(__ballot(e < n && flag[e] != 0));  // (per wave, the same in all its lanes)
  }
  if ((threadIdx.x & 63) == 0) wc[threadIdx.x >> 6] = c;
  __syncthreads();
  if (threadIdx.x == 0) tile_cnt[blockIdx.x] = (wc[0] + wc[1]) + (wc[2] + wc[3]);
}

__global__ __launch_bounds__(256) void flag_scatter_kernel(const uint8_t* __restrict__ flag, uint32_t n,
                                                            const uint32_t* __restrict__ tile_cnt, uint32_t n_tiles,
                                                            uint32_t* __restrict__ sel, uint32_t* __restrict__ count) {
  __shared__ uint32_t red[4], wave_cnt[4], off_s;
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  uint32_t before = 0;  // flags set in the tiles before this one
  for (uint32_t t = threadIdx.x; t < blockIdx.x; t += 256) before += tile_cnt[t];
  for (int o = 32; o > 0; o >>= 1) before += __shfl_xor(before, o);
  if (lane == 0) red[w] = before;
  __syncthreads();
  uint32_t off = (red[0] + red[1]) + (red[2] + red[3]);
  const uint32_t base = blockIdx.x * SEL_TILE;
  for (int k = 0; k < SEL_TILE / 256; ++k) {
    const uint32_t e = base + k * 256 + threadIdx.x;
    const bool f = e < n && flag[e] != 0;
    const unsigned long long m = __ballot(f);
    __syncthreads();  // (the previous round's wave_cnt has been read)
    if (lane == 0) wave_cnt[w] = (uint32_t)__popcll(m);
    __syncthreads();
    uint32_t wo = off;
    for (int i = 0; i < w; ++i) wo += wave_cnt[i];
    if (f) sel[wo + (uint32_t)__popcll(m & ((1ull << lane) - 1ull))] = e;
    off += (wave_cnt[0] + wave_cnt[1]) + (wave_cnt[2] + wave_cnt[3]);
  }
  if (blockIdx.x == n_tiles - 1 && threadIdx.x == 0) *count = off;
  (void)off_s;
}

// spts[s] = (x, y, z, bits(index in the unsorted cloud)) of the s-th point in key order
__global__ __launch_bounds__(256) void gather_sorted_f4_kernel(const f32x4* __restrict__ pts,
                                                                const uint32_t* __restrict__ perm, uint32_t m,
                                                                f32x4* __restrict__ spts) {
  const uint32_t s = blockIdx.x * 256 + threadIdx.x;
  if (s >= m) return;
  const uint32_t o = perm[s];
  const f32x4 p = pts[o];
  spts[s] = f32x4{p.x, p.y, p.z, __uint_as_float(o)};
}

// one wave per chunk of KCH sorted points
__global__ __launch_bounds__(64) void kchunk_boxes_kernel(const f32x4* __restrict__ spts, uint32_t m,
                                                          f32x4* __restrict__ lo, f32x4* __restrict__ hi) {
  const uint32_t j = blockIdx.x * KCH + threadIdx.x;
  float mn[3] = {3.4e38f, 3.4e38f, 3.4e38f}, mx[3] = {-3.4e38f, -3.4e38f, -3.4e38f};
  if (j < m) {
    const f32x4 p = spts[j];
    mn[0] = mx[0] = p.x; mn[1] = mx[1] = p.y; mn[2] = mx[2] = p.z;
  }
  for (int o = 32; o > 0; o >>= 1)
    for (int a = 0; a < 3; ++a) {
      mn[a] = fminf(mn[a], __shfl_xor(mn[a], o));
      mx[a] = fmaxf(mx[a], __shfl_xor(mx[a], o));
    }
  if (threadIdx.x == 0) {
    lo[blockIdx.x] = f32x4{mn[0], mn[1], mn[2], 0.f};
    hi[blockIdx.x] = f32x4{mx[0], mx[1], mx[2], 0.f};
  }
}

// 4 independent waves per work-group, one chunk of sources each.
__global__ __launch_bounds__(256) void knn_culled_kernel(const f32x4* __restrict__ spts, uint32_t m,
                                                          const f32x4* __restrict__ box_lo,
                                                          const f32x4* __restrict__ box_hi, uint32_t nchunks, int k,
                                                          uint32_t* __restrict__ idx, float* __restrict__ d2) {
  __shared__ f32x4 stage_all[4][KCH];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  f32x4* stage = stage_all[w];
  const uint32_t own = blockIdx.x * 4 + w;
  if (own >= nchunks) return;  // whole wave idle; no work-group barrier below
  const uint32_t si = own * KCH + lane;
  const bool valid = si < m;
  const f32x4 p = spts[valid ? si : m - 1];
  float bd[KMAX];
  uint32_t bi[KMAX];
#pragma unroll
  for (int s = 0; s < KMAX; ++s) { bd[s] = FLT_MAX; bi[s] = 0xFFFFFFFFu; }
  float worst = FLT_MAX;
  uint32_t worst_i = 0xFFFFFFFFu;  // (bd, bi)[k - 1]
  auto insert = [&](float d, uint32_t j) {
    float cd = d;
    uint32_t ci = j;
    bool shifting = false;
#pragma unroll
    for (int s = 0; s < KMAX; ++s) {
      if (s < k) {
        shifting = shifting || cd < bd[s] || (cd == bd[s] && ci < bi[s]);
        if (shifting) {
          const float td = bd[s]; const uint32_t ti = bi[s];
          bd[s] = cd; bi[s] = ci;
          cd = td; ci = ti;
        }
      }
    }
#pragma unroll
    for (int s = 0; s < KMAX; ++s)
      if (s == k - 1) { worst = bd[s]; worst_i = bi[s]; }
  };
  // evaluate one staged chunk: all lanes against its points (wave-uniform LDS reads)
  auto eval_chunk = [&](uint32_t c) {
    const uint32_t j = c * KCH + lane;
    f32x4 v = {NN_FAR_, NN_FAR_, NN_FAR_, __uint_as_float(0xFFFFFFFFu)};
    if (j < m) v = spts[j];
    __builtin_amdgcn_wave_barrier();  // earlier reads of the slice are done
    stage[lane] = v;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    const uint32_t lim = (m - c * KCH) < (uint32_t)KCH ? (m - c * KCH) : (uint32_t)KCH;
    for (uint32_t t = 0; t < (uint32_t)KCH; t += 8) {
      float d[8];
      uint32_t o[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const f32x4 q = stage[t + u];
        d[u] = reg::dist2(p.x, p.y, p.z, q.x, q.y, q.z);
        o[u] = __float_as_uint(q.w);
      }
      const float dm = fminf(fminf(fminf(d[0], d[1]), fminf(d[2], d[3])), fminf(fminf(d[4], d[5]), fminf(d[6], d[7])));
      if (dm <= worst) {
#pragma unroll
        for (int u = 0; u < 8; ++u)
          if (t + u < lim && (d[u] < worst || (d[u] == worst && o[u] < worst_i))) insert(d[u], o[u]);
      }
    }
  };
  eval_chunk(own);
  auto wave_max = [&]() {
    float mx = valid ? worst : -1.f;
    for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o));
    return mx;
  };
  float wmax = wave_max();
  const f32x4 olo = box_lo[own], ohi = box_hi[own];
  for (uint32_t c0 = 0; c0 < nchunks; c0 += 64) {
    const uint32_t cl = c0 + lane;
    float lbw = FLT_MAX;
    f32x4 blo = {0.f, 0.f, 0.f, 0.f}, bhi = {0.f, 0.f, 0.f, 0.f};
    if (cl < nchunks && cl != own) {
      blo = box_lo[cl]; bhi = box_hi[cl];
      const float ex = fmaxf(fmaxf(blo.x - ohi.x, olo.x - bhi.x), 0.f);
      const float ey = fmaxf(fmaxf(blo.y - ohi.y, olo.y - bhi.y), 0.f);
      const float ez = fmaxf(fmaxf(blo.z - ohi.z, olo.z - bhi.z), 0.f);
      lbw = ((ex * ex + ey * ey) + ez * ez) * 0.99999905f;
    }
    unsigned long long mask = __ballot(lbw <= wmax);
    while (mask) {
      const int b = __ffsll((long long)mask) - 1;
      mask &= mask - 1;
      if (__int_as_float(__builtin_amdgcn_readlane(__float_as_int(lbw), b)) > wmax) continue;
      f32x4 lo, hi;
      lo.x = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(blo.x), b));
      lo.y = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(blo.y), b));
      lo.z = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(blo.z), b));
      hi.x = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(bhi.x), b));
      hi.y = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(bhi.y), b));
      hi.z = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(bhi.z), b));
      const bool need = valid && reg::box_lb(p.x, p.y, p.z, lo, hi) <= worst;  // <=: equal distances may win on the index
      if (!__any(need)) continue;
      eval_chunk(c0 + b);
      wmax = wave_max();
    }
  }
  if (valid) {
    const uint32_t orig = __float_as_uint(p.w);
#pragma unroll
    for (int s = 0; s < KMAX; ++s)
      if (s < k) {
        idx[(size_t)orig * k + s] = bi[s];
        d2[(size_t)orig * k + s] = bd[s];
      }
  }
}

// sin of the bin edges -80 .. +80 degrees: bin b holds elevations [10b - 90, 10b - 80) (oracle: kSinEdge)
__device__ __constant__ double kSinEdge[17] = {
    -0.98480775301220805937, -0.93969262078590838405, -0.86602540378443864676, -0.76604444311897803520,
    -0.64278760968653932632, -0.50000000000000000000, -0.34202014332566873304, -0.17364817766693034885,
    0.0,
    0.17364817766693034885,  0.34202014332566873304,  0.50000000000000000000,  0.64278760968653932632,
    0.76604444311897803520,  0.86602540378443864676,  0.93969262078590838405,  0.98480775301220805937};

// G3 + G4: one thread per point.  fp64 mean and covariance in neighbour order, smallest-eigenvalue
// eigenvector (cyclic Jacobi), flipped towards the origin, 10-degree elevation bin; 18-bin histogram.
__global__ __launch_bounds__(256) void normals_kernel(const f32x4* __restrict__ pts, uint32_t m,
                                                       const uint32_t* __restrict__ nb, int k,
                                                       float* __restrict__ normals /* may be null */,
                                                       uint8_t* __restrict__ bins, uint32_t* __restrict__ hist) {
  __shared__ uint32_t lh[18];
  if (threadIdx.x < 18) lh[threadIdx.x] = 0;
  __syncthreads();
  const uint32_t i = blockIdx.x * 256 + threadIdx.x;
  if (i < m) {
    double mean[3] = {0, 0, 0};
    uint32_t cnt = 0;
    for (int s = 0; s < k; ++s) {
      const uint32_t j = nb[(size_t)i * k + s];
      if (j == 0xFFFFFFFFu) continue;
      const f32x4 q = pts[j];
      mean[0] += (double)q.x; mean[1] += (double)q.y; mean[2] += (double)q.z;
      ++cnt;
    }
    double nrm[3] = {0, 0, 0};
    if (cnt >= 3) {
      for (int a = 0; a < 3; ++a) mean[a] = mean[a] / (double)cnt;
      double C[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
      for (int s = 0; s < k; ++s) {
        const uint32_t j = nb[(size_t)i * k + s];
        if (j == 0xFFFFFFFFu) continue;
        const f32x4 q = pts[j];
        const double d[3] = {(double)q.x - mean[0], (double)q.y - mean[1], (double)q.z - mean[2]};
        C[0] += d[0] * d[0]; C[1] += d[0] * d[1]; C[2] += d[0] * d[2];
        C[4] += d[1] * d[1]; C[5] += d[1] * d[2];
        C[8] += d[2] * d[2];
      }
      C[3] = C[1]; C[6] = C[2]; C[7] = C[5];
      double V[9];
      reg::jacobi_eig3(C, V);
      int col = 0;  // smallest eigenvalue; ties keep the lower column
      if (C[4] < C[0]) col = 1;
      if (C[8] < C[4 * col]) col = 2;
      nrm[0] = V[0 + col]; nrm[1] = V[3 + col]; nrm[2] = V[6 + col];
      const f32x4 p = pts[i];
      const double dot = ((-(double)p.x) * nrm[0] + (-(double)p.y) * nrm[1]) + (-(double)p.z) * nrm[2];
      if (dot < 0.0) { nrm[0] = -nrm[0]; nrm[1] = -nrm[1]; nrm[2] = -nrm[2]; }
    }
    const double len = sqrt((nrm[0] * nrm[0] + nrm[1] * nrm[1]) + nrm[2] * nrm[2]);
    const double sn = len > 0.0 ? nrm[2] / len : 0.0;
    int b = 0;
    while (b < 17 && sn >= kSinEdge[b]) ++b;
    bins[i] = (uint8_t)b;
    if (normals) {
      normals[3 * (size_t)i + 0] = (float)nrm[0];
      normals[3 * (size_t)i + 1] = (float)nrm[1];
      normals[3 * (size_t)i + 2] = (float)nrm[2];
    }
    atomicAdd(&lh[b], 1u);
  }
  __syncthreads();
  if (threadIdx.x < 18 && lh[threadIdx.x]) atomicAdd(&hist[threadIdx.x], lh[threadIdx.x]);
}

__global__ __launch_bounds__(256) void bin_flag_kernel(const uint8_t* __restrict__ bins, uint32_t m, int bin,
                                                        uint8_t* __restrict__ flag) {
  const uint32_t i = blockIdx.x * 256 + threadIdx.x;
  if (i < m) flag[i] = bins[i] == bin ? 1 : 0;
}

// Compact by a selection list: out[i] = in[sel[i]].
__global__ __launch_bounds__(256) void gather_f4_kernel(const f32x4* __restrict__ in, const uint32_t* __restrict__ sel,
                                                         uint32_t m, f32x4* __restrict__ out) {
  const uint32_t i = blockIdx.x * 256 + threadIdx.x;
  if (i < m) out[i] = in[sel[i]];
}

// G6a: thread per hypothesis -- sample 3 ground points, plane through them (fp64, unit normal).
__global__ __launch_bounds__(256) void plane_hyp_kernel(const f32x4* __restrict__ g, uint32_t ng, uint64_t seed,
                                                         uint32_t n_hyp, float* __restrict__ planes /* [n_hyp][4] */,
                                                         uint32_t* __restrict__ valid) {
  const uint32_t h = blockIdx.x * 256 + threadIdx.x;
  if (h >= n_hyp) return;
  valid[h] = 0;
  if (ng < 3) return;
  const uint64_t key = synth::rng_key(seed, ((uint64_t)GROUND_STREAM << 32) | (uint64_t)h);
  uint64_t ctr = 0;
  uint32_t s0 = reg::mulhi_idx(synth::rng_draw(key, ctr++), ng), s1 = s0, s2 = s0;
  for (int tries = 0; tries < 16 && s1 == s0; ++tries) s1 = reg::mulhi_idx(synth::rng_draw(key, ctr++), ng);
  for (int tries = 0; tries < 16 && (s2 == s0 || s2 == s1); ++tries)
    s2 = reg::mulhi_idx(synth::rng_draw(key, ctr++), ng);
  if (s0 == s1 || s0 == s2 || s1 == s2) return;
  const f32x4 p0 = g[s0], p1 = g[s1], p2 = g[s2];
  const double a[3] = {(double)p1.x - p0.x, (double)p1.y - p0.y, (double)p1.z - p0.z};
  const double b[3] = {(double)p2.x - p0.x, (double)p2.y - p0.y, (double)p2.z - p0.z};
  double c[3];
  reg::cross3(a, b, c);
  const double aa = (a[0] * a[0] + a[1] * a[1]) + a[2] * a[2];
  const double bb = (b[0] * b[0] + b[1] * b[1]) + b[2] * b[2];
  const double cc = (c[0] * c[0] + c[1] * c[1]) + c[2] * c[2];
  if (!(aa > 1e-12) || !(bb > 1e-12) || !(cc > 1e-6 * (aa * bb))) return;
  const double len = sqrt(cc);
  float* pl = planes + 4 * (size_t)h;
  pl[0] = (float)(c[0] / len);
  pl[1] = (float)(c[1] / len);
  pl[2] = (float)(c[2] / len);
  pl[3] = (float)(-((c[0] / len * p0.x + c[1] / len * p0.y) + c[2] / len * p0.z));
  valid[h] = 1;
}

// G6b: thread per hypothesis, ground points streamed through LDS; |a x + b y + c z + d| < thr.
// grid = (ceil(n_hyp / 256), point slabs); counts are added atomically (integers: order-free).
constexpr int PLANE_TILE = 512;
__global__ __launch_bounds__(256) void plane_score_kernel(const f32x4* __restrict__ g, uint32_t ng,
                                                           const float* __restrict__ planes,
                                                           const uint32_t* __restrict__ valid, uint32_t n_hyp,
                                                           float thr, uint32_t slab, uint32_t* __restrict__ inliers) {
  __shared__ f32x4 tile[PLANE_TILE];
  const uint32_t h = blockIdx.x * 256 + threadIdx.x;
  const bool hv = h < n_hyp && valid[h];
  float pa = 0.f, pb = 0.f, pc = 0.f, pd = 0.f;
  if (hv) { pa = planes[4 * (size_t)h]; pb = planes[4 * (size_t)h + 1]; pc = planes[4 * (size_t)h + 2]; pd = planes[4 * (size_t)h + 3]; }
  const uint32_t i0 = blockIdx.y * slab, i1 = (i0 + slab) < ng ? (i0 + slab) : ng;
  uint32_t cnt = 0;
  for (uint32_t t0 = i0; t0 < i1; t0 += PLANE_TILE) {
    __syncthreads();
    for (uint32_t t = threadIdx.x; t < PLANE_TILE; t += 256)
      if (t0 + t < i1) tile[t] = g[t0 + t];
    __syncthreads();
    const uint32_t lim = (i1 - t0) < (uint32_t)PLANE_TILE ? (i1 - t0) : (uint32_t)PLANE_TILE;
    if (hv)
      for (uint32_t t = 0; t < lim; ++t) {
        const f32x4 q = tile[t];
        const float dist = ((pa * q.x + pb * q.y) + pc * q.z) + pd;
        cnt += (fabsf(dist) < thr) ? 1u : 0u;
      }
  }
  if (hv && cnt) atomicAdd(&inliers[h], cnt);
}

// G7 output: p' = R p + t for the whole cloud (other channels of a strided cloud are copied).
__global__ __launch_bounds__(256) void transform_cloud_kernel(const float* __restrict__ in, uint32_t n, int stride,
                                                               const float* __restrict__ T12 /* R row-major | t */,
                                                               float* __restrict__ out) {
  const uint32_t i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const float* p = in + (size_t)i * stride;
  float* o = out + (size_t)i * stride;
  float x, y, z;
  reg::xform(T12, p[0], p[1], p[2], x, y, z);
  o[0] = x; o[1] = y; o[2] = z;
  for (int c = 3; c < stride; ++c) o[c] = p[c];
}

}  // namespace ground
}  // namespace gloc
