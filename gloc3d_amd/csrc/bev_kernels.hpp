// bev_kernels.hpp -- HIP kernels of the BEV occupancy projection ("next" row N1 of SURVEY.md 8f).
//
// What the reference computes (registration/loop_detector.cpp:122-135): every kept point of ONE scan
// is inserted as a hit into a fresh probability voxel grid (3d/range_data_inserter_3d.cpp:63-78),
// the last two voxels of every ray are inserted as misses (:27-52), and the grid is x-ray projected
// (ProjectToCvMat, 3d/submap_3d.cpp:238-326): a pixel becomes 0 when the probabilities of the
// column's voxels with p >= 0.501 sum to more than 0.9, else 255.
//
// What that is, for a fresh grid and one scan (the only way this path is ever called):
//   * a voxel touched by a hit holds exactly hit_table[0] = value(0.55) -- hits are inserted first
//     and the update marker makes every later touch of the same voxel a no-op until FinishUpdate
//     (3d/hybrid_grid.h:491-508), so multiplicity and misses cannot change it;
//   * a voxel touched only by misses holds value(0.49) < 0.501 and is skipped by the projection;
//   * the identity pose makes the projected voxel index equal to the grid index;
//   * so the column sum is 0.55 x (number of distinct hit voxels in the column), which exceeds 0.9
//     exactly when that number is >= 2, i.e. when the column's hit voxels span two z indices.
// The kernels therefore keep, per (x, y) column, ONE z index (any one of those that hit it) and a
// one-byte flag "some hit of this column has a different z index".  No voxel grid, no ray walking, no
// tables, and NO ATOMICS on the column data: global atomics execute at the memory side on this chip
// (one 64-B request per scattered lane, ~6 G/s measured here), so the flag is built from two passes of
// plain stores instead:
//   pass A (bev_mark):  every kept point stores its z index into zcol[column]   (racing stores: one wins)
//   pass B (bev_flag):  every kept point compares its z index with the winner; if different it stores
//                       1 into multi[column]                                     (all writers agree)
// multi[column] ends up 1 exactly when the column's hits span two or more z indices, whatever order
// the stores of pass A landed in; the kernel boundary between the passes makes pass A's stores
// visible chip-wide.  zcol needs no clearing (pass B only reads columns pass A wrote).
// The oracle (oracle/bev_oracle.c) restates the long form; the tests compare the two byte for byte.
//
// HBM traffic per scan: 2 x 12..16 B per point in, 1 B x S^2 flags cleared (S = 2R+1 columns per
// axis, R = ceil(max_range / resolution) + 2), scattered 4-B and 1-B stores that merge in L2, and the
// output image.
#pragma once
#include <hip/hip_runtime.h>

#include <climits>
#include <cstdint>

namespace gloc {
namespace bev {

struct ScanMeta {  // per scan, device resident
  int min_ix, min_iy, max_ix, max_iy;
  unsigned n_returns, pad_[3];
};

constexpr int PTS_PER_THREAD = 8;  // points per thread: few blocks per scan -> few box atomics

// std::lround (3d/port.h:41): halves away from zero.  v - trunc(v) is exact for |v| < 2^23.
__device__ inline int round_half_away(float v) {
  float r = truncf(v);
  const float d = v - r;
  if (d >= 0.5f) r += 1.f;
  else if (d <= -0.5f) r -= 1.f;
  return (int)r;
}

// Clear the flags and reset the per-scan boxes.  bytes16 = n_scans*S*S/16 rounded up (padded buffer).
__global__ __launch_bounds__(256) void bev_clear_kernel(uint4* multi16, size_t bytes16, ScanMeta* meta,
                                                         int n_scans) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i < bytes16) multi16[i] = make_uint4(0, 0, 0, 0);
  if (i < (size_t)n_scans) {
    ScanMeta m;
    m.min_ix = INT_MAX; m.min_iy = INT_MAX; m.max_ix = INT_MIN; m.max_iy = INT_MIN;
    m.n_returns = 0; m.pad_[0] = m.pad_[1] = m.pad_[2] = 0;
    meta[i] = m;
  }
}

// The reference's two range tests and its voxel index for one point.  false: the point is dropped.
__device__ inline bool voxel_of(const float* __restrict__ p, float resolution, float range_split,
                                float range_filter, int R, int& ix, int& iy, int& iz) {
  const float x = p[0], y = p[1], z = p[2];
  // loop_detector.cpp:113: sqrt(x*x + y*y + z*z) > 100. -> a "miss" (never read on this path)
  const float s1 = (x * x + y * y) + z * z;
  if (__builtin_sqrtf(s1) > range_split) return false;  // correctly rounded (__fsqrt_rn is the native one)
  // submap_3d.cpp:47: (hit - origin).norm() <= max_range, Eigen's 3-term order, range taken as int
  const float s2 = x * x + (y * y + z * z);
  if (!(__builtin_sqrtf(s2) <= range_filter)) return false;
  // hybrid_grid.h:429-434
  ix = round_half_away(x / resolution);
  iy = round_half_away(y / resolution);
  iz = round_half_away(z / resolution);
  return !(ix < -R || ix > R || iy < -R || iy > R || iz < -R || iz > R);  // always true: |p| <= range
}

// PASS: 0 = mark (store z, build the box and the count), 1 = flag.  blockIdx.y = scan; a block
// covers 256 * PTS_PER_THREAD consecutive points.
template <int PASS>
__global__ __launch_bounds__(256) void bev_points_kernel(const float* __restrict__ xyz,
                                                          const uint64_t* __restrict__ offsets, int stride,
                                                          float resolution, float range_split,
                                                          float range_filter, int R, int S,
                                                          uint32_t* __restrict__ zcol,
                                                          uint8_t* __restrict__ multi,
                                                          ScanMeta* __restrict__ meta) {
  const int scan = blockIdx.y;
  const uint64_t base = offsets[scan];
  const uint64_t n = offsets[scan + 1] - base;
  const uint64_t first = (uint64_t)blockIdx.x * (256 * PTS_PER_THREAD);
  if (first >= n) return;
  const size_t plane = (size_t)scan * S * S;
  int mnx = INT_MAX, mny = INT_MAX, mxx = INT_MIN, mxy = INT_MIN;
  unsigned kept = 0;
#pragma unroll
  for (int k = 0; k < PTS_PER_THREAD; ++k) {
    const uint64_t i = first + (uint64_t)k * 256 + threadIdx.x;
    if (i >= n) break;
    int ix, iy, iz;
    if (!voxel_of(xyz + (base + i) * (uint64_t)stride, resolution, range_split, range_filter, R, ix, iy, iz))
      continue;
    const size_t c = plane + (size_t)(iy + R) * S + (ix + R);
    const uint32_t zb = (uint32_t)(iz + R);
    if (PASS == 0) {
      zcol[c] = zb;
      ++kept;
      mnx = min(mnx, ix); mxx = max(mxx, ix);
      mny = min(mny, iy); mxy = max(mxy, iy);
    } else {
      if (zcol[c] != zb) multi[c] = 1;
    }
  }
  if (PASS == 0) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      mnx = min(mnx, __shfl_xor(mnx, o)); mny = min(mny, __shfl_xor(mny, o));
      mxx = max(mxx, __shfl_xor(mxx, o)); mxy = max(mxy, __shfl_xor(mxy, o));
      kept += __shfl_xor(kept, o);
    }
    __shared__ int red[4][5];
    const int wave = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) {
      red[wave][0] = mnx; red[wave][1] = mny; red[wave][2] = mxx; red[wave][3] = mxy; red[wave][4] = (int)kept;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
      for (int w = 1; w < 4; ++w) {
        mnx = min(mnx, red[w][0]); mny = min(mny, red[w][1]);
        mxx = max(mxx, red[w][2]); mxy = max(mxy, red[w][3]);
        kept += (unsigned)red[w][4];
      }
      if (kept) {
        ScanMeta* m = meta + scan;
        atomicMin(&m->min_ix, mnx); atomicMin(&m->min_iy, mny);
        atomicMax(&m->max_ix, mxx); atomicMax(&m->max_iy, mxy);
        atomicAdd(&m->n_returns, kept);
      }
    }
  }
}

struct CropGeom {  // crop_pad_occupancy's two ROIs (loop_detector.cpp:85-101)
  int w, h, cw, ch, sx, sy, dx, dy;
  bool empty;
};

__device__ inline CropGeom crop_geom(const ScanMeta& m, int out_w, int out_h) {
  CropGeom g;
  g.empty = m.min_ix > m.max_ix;
  g.w = g.empty ? 0 : m.max_ix - m.min_ix + 1;
  g.h = g.empty ? 0 : m.max_iy - m.min_iy + 1;
  g.cw = g.w >= out_w ? out_w : g.w;
  g.ch = g.h >= out_h ? out_h : g.h;
  g.sx = (g.w - g.cw) / 2;  // floor((src.cols - cw) / 2.), operands non-negative
  g.sy = (g.h - g.ch) / 2;
  g.dx = (out_w - g.cw) / 2;
  g.dy = (out_h - g.ch) / 2;
  return g;
}

// Four consecutive pixels of one output row per thread (flat over rows); blockIdx.y = scan.
template <int FORMAT>
__global__ __launch_bounds__(256) void bev_image_kernel(const uint8_t* __restrict__ multi,
                                                         const ScanMeta* __restrict__ meta, int R, int S,
                                                         int out_w, int out_h, uint32_t pad_bgr,
                                                         void* __restrict__ out) {
  const int scan = blockIdx.y;
  const int wq = (out_w + 3) >> 2;
  const unsigned q = blockIdx.x * 256 + threadIdx.x;
  if (q >= (unsigned)wq * (unsigned)out_h) return;
  const int y = (int)(q / (unsigned)wq), x0 = (int)(q % (unsigned)wq) * 4;
  const ScanMeta m = meta[scan];
  const CropGeom g = crop_geom(m, out_w, out_h);
  const uint8_t* col = multi + (size_t)scan * S * S;
  const bool row_in = !g.empty && y >= g.dy && y < g.dy + g.ch;
  const size_t src_row = row_in ? (size_t)(m.min_iy + g.sy + (y - g.dy) + R) * S : 0;
  uint8_t px[4][3];
#pragma unroll
  for (int p = 0; p < 4; ++p) {
    const int x = x0 + p;
    const bool in = row_in && x >= g.dx && x < g.dx + g.cw;
    if (in) {
      const uint8_t v = col[src_row + (size_t)(m.min_ix + g.sx + (x - g.dx) + R)] ? 0 : 255;
      px[p][0] = v; px[p][1] = v; px[p][2] = v;
    } else {
      px[p][0] = (uint8_t)pad_bgr; px[p][1] = (uint8_t)(pad_bgr >> 8); px[p][2] = (uint8_t)(pad_bgr >> 16);
    }
  }
  const size_t plane = (size_t)out_w * out_h;
  if (FORMAT == GLOC_BEV_U8_HWC3) {
    uint8_t* o = static_cast<uint8_t*>(out) + ((size_t)scan * plane + (size_t)y * out_w + x0) * 3;
    if ((out_w & 3) == 0) {  // 12 bytes, 4-byte aligned
      const uint8_t* b = &px[0][0];
      uint32_t* o4 = reinterpret_cast<uint32_t*>(o);
#pragma unroll
      for (int k = 0; k < 3; ++k)
        o4[k] = (uint32_t)b[4 * k] | ((uint32_t)b[4 * k + 1] << 8) | ((uint32_t)b[4 * k + 2] << 16) |
                ((uint32_t)b[4 * k + 3] << 24);
    } else {
      for (int p = 0; p < 4 && x0 + p < out_w; ++p)
        for (int c = 0; c < 3; ++c) o[p * 3 + c] = px[p][c];
    }
  } else {  // u8 * (1/255), loop_detector.cpp:146: the bytes are 0 or 255 (or the pad colour)
    float* o = static_cast<float*>(out) + (size_t)scan * 3 * plane + (size_t)y * out_w + x0;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      float f[4];
#pragma unroll
      for (int p = 0; p < 4; ++p) f[p] = (float)((double)px[p][c] * (1.0 / 255.0));
      if ((out_w & 3) == 0) {
        *reinterpret_cast<float4*>(o + (size_t)c * plane) = make_float4(f[0], f[1], f[2], f[3]);
      } else {
        for (int p = 0; p < 4 && x0 + p < out_w; ++p) o[(size_t)c * plane + p] = f[p];
      }
    }
  }
}

// The uncropped image of one scan: [h][w] u8.
__global__ __launch_bounds__(256) void bev_raw_kernel(const uint8_t* __restrict__ col, int min_ix,
                                                       int min_iy, int w, int h, int R, int S,
                                                       uint8_t* __restrict__ out) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= (size_t)w * h) return;
  const int x = (int)(i % w), y = (int)(i / w);
  out[i] = col[(size_t)(min_iy + y + R) * S + (min_ix + x + R)] ? 0 : 255;
}

}  // namespace bev
}  // namespace gloc
