// coarse_kernels.hpp -- coarse global (x, y, yaw) match of two BEV occupancy grids.
//
// The reference gets its coarse pose from the 2-D images: SURF keypoints + FLANN matching + a RANSAC
// partial-affine fit (RpyPCLoopDetector::match, registration/loop_detector.cpp:192-288; OpenCV + its
// contrib module, neither in this image).  What that step delivers -- p_db = R(yaw) p_q + (x, y), or
// "no match" -- is delivered here by an exhaustive, integer-valued search the GPU is good at:
//   grid    a scan's occupied BEV columns (pixel value 0 of the reference's occupancy image: columns whose
//           hits span two or more z voxels, i.e. vertical structure) are binned, by their integer voxel
//           index, into cells of cell_px x cell_px pixels on a G x G grid centred on the sensor: a bit
//           map, its 3x3 dilation, the list of occupied cells and the two projections (occupied cells
//           per column / per row);
//   yaw     for each of n_yaw rotations the query's cells are rotated, projected onto x and y, and
//           each projection is correlated with the database grid's over all lags |t| <= max_shift:
//           the best lags (tx, ty) and the score sx + sy (one work-group per rotation);
//   verify  the top_yaw rotations by score, plus the identity, are checked in 2-D: the number of rotated
//           query cells that fall on the DILATED database bit map, for every shift within `refine`
//           cells of (tx, ty) (one work-group per rotation, the bit map in LDS);
//   result  the (rotation, shift) with the largest overlap; the identity wins unless another rotation
//           beats it by 20 %; ok iff overlap >= min_overlap x (number of query cells).
// Everything that decides is an integer count; the only floating-point step is rotating a cell centre
// (in pixel units), in a fixed un-fused fp32 order with cos / sin tables made by the host in fp64 -- oracle/coarse_oracle.c
// computes the same numbers and the tests compare them exactly.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace gloc {
namespace coarse {

constexpr int G = 512;        // cells per axis
constexpr int GW = G / 32;    // 32-bit words per bit-map row
constexpr int HALF = G / 2;

struct GridDev {              // one scan's coarse grid, device resident (one allocation)
  uint32_t* bits;             // [G][GW]
  uint32_t* dil;              // [G][GW], 3x3 dilation
  uint32_t* hx;               // [G] occupied cells per column (x index)
  uint32_t* hy;               // [G] per row
  uint32_t* cells;            // [cap] (V << 16) | U
  uint32_t* count;            // number of cells
};

// std::lround: halves away from zero (as the BEV projection, bev_kernels.hpp)
__host__ __device__ inline int round_half_away_f(float v) {
  float r = (float)(int)v;  // trunc: |v| < 2^23 here
  const float d = v - r;
  if (d >= 0.5f) r += 1.f;
  else if (d <= -0.5f) r -= 1.f;
  return (int)r;
}

// voxel (pixel) index -> cell index in [0, G) or -1: floor(ix / cell_px) + G / 2
__host__ __device__ inline int cell_of_px(int ix, int cell_px) {
  const int u = (ix >= 0 ? ix / cell_px : -((-ix + cell_px - 1) / cell_px)) + HALF;
  return (u >= 0 && u < G) ? u : -1;
}
// centre of cell u in pixel units: the middle of the cell_px pixels it spans
__host__ __device__ inline float cell_centre_px(int u, int cell_px) {
  return (float)((u - HALF) * cell_px) + 0.5f * (float)(cell_px - 1);
}
// a cell (v << 16 | u) rotated by (c, s) about the sensor -> cell indices, or -1
__host__ __device__ inline void rotate_cell(uint32_t uv, float c, float s, int cell_px, int& u, int& v) {
  const float x = cell_centre_px((int)(uv & 0xFFFF), cell_px), y = cell_centre_px((int)(uv >> 16), cell_px);
  const float a0 = c * x, a1 = s * y, b0 = s * x, b1 = c * y;
  u = cell_of_px(round_half_away_f(a0 - a1), cell_px);
  v = cell_of_px(round_half_away_f(b0 + b1), cell_px);
}

// occupied columns of a BEV flag plane (multi[(iy + R) * S + (ix + R)] != 0) -> bit map
__global__ __launch_bounds__(256) void mark_from_flags_kernel(const uint8_t* __restrict__ flags, int R, int S,
                                                              int cell_px, uint32_t* __restrict__ bits) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= (size_t)S * S || !flags[i]) return;
  const int u = cell_of_px((int)(i % S) - R, cell_px), v = cell_of_px((int)(i / S) - R, cell_px);
  if (u >= 0 && v >= 0) atomicOr(&bits[v * GW + (u >> 5)], 1u << (u & 31));
}

// the same from the reference's occupancy image ([h][w] u8, below 100 = occupied as the reference's
// threshold, loop_detector.cpp:196; pixel (x, y) = voxel (ix0 + x, iy0 + y) with ix0 = lround(ox / res))
__global__ __launch_bounds__(256) void mark_from_image_kernel(const uint8_t* __restrict__ img, int w, int h,
                                                              int ix0, int iy0, int cell_px,
                                                              uint32_t* __restrict__ bits) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= (size_t)w * h || img[i] >= 100) return;
  const int u = cell_of_px(ix0 + (int)(i % w), cell_px), v = cell_of_px(iy0 + (int)(i / w), cell_px);
  if (u >= 0 && v >= 0) atomicOr(&bits[v * GW + (u >> 5)], 1u << (u & 31));
}

// bit map -> dilation, projections, cell list.  One work-group of 512 threads: thread = row.
__global__ __launch_bounds__(512) void finish_grid_kernel(GridDev g, uint32_t cap) {
  __shared__ uint32_t colcnt[G];
  const int v = threadIdx.x;
  colcnt[v] = 0;
  __syncthreads();
  uint32_t rowcnt = 0;
  for (int w = 0; w < GW; ++w) {
    const uint32_t b = g.bits[v * GW + w];
    rowcnt += __popc(b);
    // dilation: OR of the three rows, each OR-ed with its one-bit shifts (carry across words)
    uint32_t d = 0;
    for (int dv = -1; dv <= 1; ++dv) {
      const int vv = v + dv;
      if (vv < 0 || vv >= G) continue;
      const uint32_t c = g.bits[vv * GW + w];
      const uint32_t l = w > 0 ? g.bits[vv * GW + w - 1] : 0u;
      const uint32_t r = w + 1 < GW ? g.bits[vv * GW + w + 1] : 0u;
      d |= c | (c << 1) | (c >> 1) | (l >> 31) | (r << 31);
    }
    g.dil[v * GW + w] = d;
    uint32_t m = b;
    while (m) {
      const int bit = __ffs(m) - 1;
      m &= m - 1;
      const int u = w * 32 + bit;
      atomicAdd(&colcnt[u], 1u);
      const uint32_t pos = atomicAdd(g.count, 1u);
      if (pos < cap) g.cells[pos] = ((uint32_t)v << 16) | (uint32_t)u;
    }
  }
  g.hy[v] = rowcnt;
  __syncthreads();
  g.hx[v] = colcnt[v];
}

struct YawOut {  // per (pair, rotation)
  int tx, ty;
  uint32_t sx, sy;
};

// grid (n_yaw, n_pairs), 256 threads.  trig: [n_yaw][2] = (cos, sin).
__global__ __launch_bounds__(256) void yaw_kernel(const GridDev* __restrict__ qgrids, const GridDev* __restrict__ dgrids,
                                                  const uint32_t* __restrict__ pair_q, const uint32_t* __restrict__ pair_d,
                                                  const float* __restrict__ trig, int cell_px, int max_shift,
                                                  uint32_t n_yaw, YawOut* __restrict__ out) {
  __shared__ uint32_t hq[2][G], hd[2][G];
  __shared__ unsigned long long best[2];
  const int tid = threadIdx.x;
  const uint32_t k = blockIdx.x, pair = blockIdx.y;
  const GridDev q = qgrids[pair_q[pair]], d = dgrids[pair_d[pair]];
  for (int i = tid; i < G; i += 256) {
    hq[0][i] = 0; hq[1][i] = 0;
    hd[0][i] = d.hx[i]; hd[1][i] = d.hy[i];
  }
  if (tid < 2) best[tid] = 0ull;
  __syncthreads();
  const float c = trig[2 * k], s = trig[2 * k + 1];
  const uint32_t n = *q.count;
  for (uint32_t i = tid; i < n; i += 256) {
    int u, v;
    rotate_cell(q.cells[i], c, s, cell_px, u, v);
    if (u >= 0 && v >= 0) {
      atomicAdd(&hq[0][u], 1u);
      atomicAdd(&hq[1][v], 1u);
    }
  }
  __syncthreads();
  // lags -max_shift .. +max_shift of both axes over the threads; key = (score << 32) | (~lag index):
  // the largest score wins, ties go to the smallest lag index (most negative lag first)
  const int nl = 2 * max_shift + 1;
  for (int j = tid; j < 2 * nl; j += 256) {
    const int axis = j / nl, li = j % nl, t = li - max_shift;
    uint32_t acc = 0;
    const int i0 = t < 0 ? -t : 0, i1 = t > 0 ? G - t : G;
    for (int i = i0; i < i1; ++i) acc += hq[axis][i] * hd[axis][i + t];
    atomicMax(&best[axis], ((unsigned long long)acc << 32) | (uint32_t)(0xFFFFFFFFu - (uint32_t)li));
  }
  __syncthreads();
  if (tid == 0) {
    YawOut o;
    o.sx = (uint32_t)(best[0] >> 32);
    o.tx = (int)(0xFFFFFFFFu - (uint32_t)best[0]) - max_shift;
    o.sy = (uint32_t)(best[1] >> 32);
    o.ty = (int)(0xFFFFFFFFu - (uint32_t)best[1]) - max_shift;
    out[(size_t)pair * n_yaw + k] = o;
  }
}

// per pair: the top_yaw rotations by sx + sy (ties: smaller rotation index), then the identity (k = 0,
// shift 0) as candidate number top_yaw.  One wave per pair.
__global__ __launch_bounds__(64) void top_kernel(const YawOut* __restrict__ yo, uint32_t n_yaw, uint32_t top_yaw,
                                                 uint32_t* __restrict__ cand /* [pair][top_yaw + 1][3] = k, tx, ty */) {
  const uint32_t pair = blockIdx.x, lane = threadIdx.x;
  unsigned long long prev = ~0ull;
  for (uint32_t m = 0; m < top_yaw; ++m) {
    unsigned long long bestk = 0ull;
    for (uint32_t k = lane; k < n_yaw; k += 64) {
      const YawOut o = yo[(size_t)pair * n_yaw + k];
      const unsigned long long key = ((unsigned long long)(o.sx + o.sy) << 32) | (uint32_t)(0xFFFFFFFFu - k);
      if (key < prev && key > bestk) bestk = key;
    }
    for (int o = 32; o > 0; o >>= 1) {
      const unsigned long long other = __shfl_xor(bestk, o);
      bestk = other > bestk ? other : bestk;
    }
    prev = bestk;
    if (lane == 0) {
      uint32_t* c = cand + ((size_t)pair * (top_yaw + 1) + m) * 3;
      if (bestk == 0ull) {  // fewer rotations than top_yaw
        c[0] = 0xFFFFFFFFu; c[1] = 0; c[2] = 0;
      } else {
        const uint32_t k = 0xFFFFFFFFu - (uint32_t)bestk;
        const YawOut o = yo[(size_t)pair * n_yaw + k];
        c[0] = k; c[1] = (uint32_t)o.tx; c[2] = (uint32_t)o.ty;
      }
    }
  }
  if (lane == 0) {
    uint32_t* c = cand + ((size_t)pair * (top_yaw + 1) + top_yaw) * 3;
    c[0] = 0; c[1] = 0; c[2] = 0;
  }
}

struct VerifyOut {
  uint32_t overlap;
  int tx, ty;
  uint32_t k;
};

// grid (top_yaw + 1, n_pairs), 256 threads: the database grid's dilated bit map in LDS; for every shift
// within `refine` of the candidate's (tx, ty), the number of rotated query cells on a set bit.
__global__ __launch_bounds__(256) void verify_kernel(const GridDev* __restrict__ qgrids, const GridDev* __restrict__ dgrids,
                                                     const uint32_t* __restrict__ pair_q, const uint32_t* __restrict__ pair_d,
                                                     const float* __restrict__ trig, int cell_px, int refine,
                                                     uint32_t n_cand, const uint32_t* __restrict__ cand,
                                                     VerifyOut* __restrict__ out) {
  __shared__ uint32_t bm[G * GW];
  __shared__ uint32_t red[4];
  __shared__ unsigned long long best;
  const int tid = threadIdx.x;
  const uint32_t m = blockIdx.x, pair = blockIdx.y;
  const uint32_t* cd = cand + ((size_t)pair * n_cand + m) * 3;
  const uint32_t k = cd[0];
  VerifyOut vo{0u, 0, 0, k};
  if (k == 0xFFFFFFFFu) {
    if (tid == 0) out[(size_t)pair * n_cand + m] = vo;
    return;
  }
  const int tx0 = (int)cd[1], ty0 = (int)cd[2];
  const GridDev q = qgrids[pair_q[pair]], d = dgrids[pair_d[pair]];
  for (int i = tid; i < G * GW; i += 256) bm[i] = d.dil[i];
  if (tid == 0) best = 0ull;
  __syncthreads();
  const float c = trig[2 * k], s = trig[2 * k + 1];
  const uint32_t n = *q.count;
  const int side = 2 * refine + 1;
  for (int o = 0; o < side * side; ++o) {
    const int dy = o / side - refine, dx = o % side - refine;
    const int tx = tx0 + dx, ty = ty0 + dy;
    uint32_t cnt = 0;
    for (uint32_t i = tid; i < n; i += 256) {
      int u, v;
      rotate_cell(q.cells[i], c, s, cell_px, u, v);
      if (u < 0 || v < 0) continue;
      u += tx;
      v += ty;
      if (u < 0 || u >= G || v < 0 || v >= G) continue;
      cnt += (bm[v * GW + (u >> 5)] >> (u & 31)) & 1u;
    }
    for (int sft = 32; sft > 0; sft >>= 1) cnt += __shfl_xor(cnt, sft);
    if ((tid & 63) == 0) red[tid >> 6] = cnt;
    __syncthreads();
    if (tid == 0) {
      const uint32_t total = red[0] + red[1] + red[2] + red[3];
      // largest overlap; ties: the earliest shift in (dy, dx) row-major order
      const unsigned long long key = ((unsigned long long)total << 32) | (uint32_t)(0xFFFFFFFFu - (uint32_t)o);
      if (key > best) best = key;
    }
    __syncthreads();
  }
  if (tid == 0) {
    const int o = (int)(0xFFFFFFFFu - (uint32_t)best);
    vo.overlap = (uint32_t)(best >> 32);
    vo.tx = tx0 + o % side - refine;
    vo.ty = ty0 + o / side - refine;
    out[(size_t)pair * n_cand + m] = vo;
  }
}

struct MatchOut {  // per pair
  float x, y, yaw, ratio;
  uint32_t overlap, n_query, k;
  int ok;
  int tx, ty;      // the shift in cells (scale_kernel)
  float scale;     // least-squares scale of the matched cells (scale_kernel)
  uint32_t n_matched;
};

// per pair: the candidate with the largest overlap (ties: the earlier candidate); the identity candidate
// (the last one) wins unless the best rotation has more than 1.2 x its overlap.
__global__ void final_kernel(const VerifyOut* __restrict__ vo, const GridDev* __restrict__ qgrids,
                             const uint32_t* __restrict__ pair_q, uint32_t n_cand, uint32_t n_pairs,
                             uint32_t n_yaw, float cell /* metres */, float min_overlap, const float* __restrict__ yaw_of,
                             MatchOut* __restrict__ out) {
  const uint32_t pair = blockIdx.x * blockDim.x + threadIdx.x;
  if (pair >= n_pairs) return;
  const VerifyOut* v = vo + (size_t)pair * n_cand;
  uint32_t bi = 0;
  for (uint32_t m = 1; m + 1 < n_cand; ++m)
    if (v[m].overlap > v[bi].overlap) bi = m;
  const VerifyOut id = v[n_cand - 1];
  VerifyOut b = v[bi];
  if (n_cand == 1 || b.k == 0xFFFFFFFFu || !((unsigned long long)b.overlap * 5ull > (unsigned long long)id.overlap * 6ull)) b = id;
  const uint32_t nq = *qgrids[pair_q[pair]].count;
  MatchOut o;
  o.x = (float)b.tx * cell;
  o.y = (float)b.ty * cell;
  o.yaw = yaw_of[b.k];
  o.overlap = b.overlap;
  o.n_query = nq;
  o.k = b.k;
  o.ratio = nq ? (float)b.overlap / (float)nq : 0.f;
  o.ok = (nq >= 16 && (float)b.overlap >= min_overlap * (float)nq) ? 1 : 0;
  o.tx = b.tx;
  o.ty = b.ty;
  o.scale = 0.f;
  o.n_matched = 0;
  out[pair] = o;
}

// The reference's match also estimates a SCALE (cv::estimateAffinePartial2D fits a similarity) and accepts only
// |1 - scale| < 0.1 (loop_detector.cpp:262-272).  Here: under the chosen rotation, the overlap search of verify_kernel
// is repeated with the query's cells scaled about the sensor by each of N_SCALES factors 0.88 .. 1.12 (p_db = s R p_q + t;
// a cell 30 m out moves by three cells per 4 %, so the overlap is sharply peaked in s), each over the shifts within
// `refine` of the chosen one; the estimate is the factor with the largest overlap (ties: the one nearest 1, then the
// smaller), refined by the parabola through its neighbours' overlaps when it is not an end of the range -- integer
// counts and a few fp64 operations, so the CPU restatement gets the same bits.  An end of the range means "10 % or
// more": ok is withdrawn unless |1 - scale| < 0.1.
constexpr int N_SCALES = 7;
__host__ __device__ inline float scale_factor(int j) { return 0.88f + 0.04f * (float)j; }
__host__ __device__ inline void rotate_scale_cell(uint32_t uv, float c, float s, float f, int cell_px, int& u, int& v) {
  const float x = cell_centre_px((int)(uv & 0xFFFF), cell_px), y = cell_centre_px((int)(uv >> 16), cell_px);
  const float a0 = c * x, a1 = s * y, b0 = s * x, b1 = c * y;
  const float rx = a0 - a1, ry = b0 + b1;
  u = cell_of_px(round_half_away_f(rx * f), cell_px);
  v = cell_of_px(round_half_away_f(ry * f), cell_px);
}
// overlaps[N_SCALES] -> the estimate
__host__ __device__ inline float scale_from_overlaps(const uint32_t* o) {
  int best = N_SCALES / 2;  // 1.0 first: ties go to the factor nearest 1, then to the smaller one
  for (int d = 1; d <= N_SCALES / 2; ++d) {
    if (o[N_SCALES / 2 - d] > o[best]) best = N_SCALES / 2 - d;
    if (o[N_SCALES / 2 + d] > o[best]) best = N_SCALES / 2 + d;
  }
  if (o[best] == 0u) return 0.f;
  double est = (double)scale_factor(best);
  if (best > 0 && best < N_SCALES - 1) {
    const double a = (double)o[best - 1], b0 = (double)o[best], c = (double)o[best + 1];
    const double den = a - 2.0 * b0 + c;
    if (den < 0.0) est += 0.5 * 0.04 * (a - c) / den;
  }
  return (float)est;
}

// grid (N_SCALES, n_pairs), 256 threads: as verify_kernel, for the pair's chosen rotation and shift, one scale each
__global__ __launch_bounds__(256) void scale_kernel(const GridDev* __restrict__ qgrids, const GridDev* __restrict__ dgrids,
                                                    const uint32_t* __restrict__ pair_q, const uint32_t* __restrict__ pair_d,
                                                    const float* __restrict__ trig, int cell_px, int refine,
                                                    const MatchOut* __restrict__ mo_all, uint32_t* __restrict__ overlaps) {
  __shared__ uint32_t bm[G * GW];
  __shared__ uint32_t red[4];
  __shared__ uint32_t best;
  const int tid = threadIdx.x;
  const uint32_t j = blockIdx.x, pair = blockIdx.y;
  const MatchOut mo = mo_all[pair];
  const GridDev q = qgrids[pair_q[pair]], d = dgrids[pair_d[pair]];
  for (int i = tid; i < G * GW; i += 256) bm[i] = d.dil[i];
  if (tid == 0) best = 0u;
  __syncthreads();
  const float c = trig[2 * mo.k], s = trig[2 * mo.k + 1], f = scale_factor((int)j);
  const uint32_t n = *q.count;
  const int side = 2 * refine + 1;
  for (int o = 0; o < side * side; ++o) {
    const int tx = mo.tx + o % side - refine, ty = mo.ty + o / side - refine;
    uint32_t cnt = 0;
    for (uint32_t i = tid; i < n; i += 256) {
      int u, v;
      rotate_scale_cell(q.cells[i], c, s, f, cell_px, u, v);
      if (u < 0 || v < 0) continue;
      u += tx;
      v += ty;
      if (u < 0 || u >= G || v < 0 || v >= G) continue;
      cnt += (bm[v * GW + (u >> 5)] >> (u & 31)) & 1u;
    }
    for (int sft = 32; sft > 0; sft >>= 1) cnt += __shfl_xor(cnt, sft);
    if ((tid & 63) == 0) red[tid >> 6] = cnt;
    __syncthreads();
    if (tid == 0) {
      const uint32_t total = red[0] + red[1] + red[2] + red[3];
      if (total > best) best = total;
    }
    __syncthreads();
  }
  if (tid == 0) overlaps[(size_t)pair * N_SCALES + j] = best;
}

__global__ void scale_final_kernel(const uint32_t* __restrict__ overlaps, uint32_t n_pairs, MatchOut* __restrict__ out) {
  const uint32_t pair = blockIdx.x * blockDim.x + threadIdx.x;
  if (pair >= n_pairs) return;
  MatchOut mo = out[pair];
  mo.scale = scale_from_overlaps(overlaps + (size_t)pair * N_SCALES);
  mo.n_matched = overlaps[(size_t)pair * N_SCALES + N_SCALES / 2];
  if (!(fabsf(1.f - mo.scale) < 0.1f)) mo.ok = 0;
  out[pair] = mo;
}

}  // namespace coarse
}  // namespace gloc
