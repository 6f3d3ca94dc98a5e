// nn_culled.hpp -- exact 1-NN with spatial culling (K4, default mode).
//
// Same result, bit for bit, as the exhaustive nn_kernel (and the oracle): the distance of every
// evaluated pair is the un-fused fp32 ((dx dx + dy dy) + dz dz), ties go to the smallest ORIGINAL
// target index.  What changes is which pairs are evaluated:
//   * every scan is Morton-sorted once (scan_prepare) and cut into chunks of 128 points with an
//     axis-aligned bounding box each;
//   * a wave owns 256 consecutive (hence spatially compact) sorted source points, 4 per lane;
//   * upper bounds come from the previous ICP pass's correspondence (warm start) or from the
//     Morton neighbourhood of the point in the target's key order;
//   * 64 chunk boxes at a time are tested against the wave's box by the 64 lanes (one ballot), the
//     survivors against each lane's own points, and only chunks that can still hold a nearer (or
//     equally near) point are staged through LDS and evaluated.
// A chunk is skipped only when a conservative lower bound of its distance exceeds the current best
// of every point of the wave, so no candidate for the minimum (or for a tie) is ever missed.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "reg_kernels.hpp"

namespace gloc {
namespace reg {

#define GPTR(T) const T __attribute__((address_space(1)))*
constexpr int CH = 128;  // points per chunk
constexpr int SB = 16;   // points per sub-block (second-level boxes, argmin bookkeeping)

struct ScanIndexDev {
  const f32x4* pts;      // Morton order: x, y, z, bits(original index)
  const f32x4* box_lo;   // per chunk of CH points
  const f32x4* box_hi;
  const f32x4* sb_lo;    // per sub-block of SB points
  const f32x4* sb_hi;
  const uint32_t* keys;  // sorted Morton keys
  const uint32_t* inv;   // original index -> sorted position
  uint32_t n, nchunks;
  float ox, oy, oz, inv_cell;
};

struct CulledCand {
  ScanIndexDev idx;
  const float* xyz;  // original order, packed
};

__device__ __forceinline__ uint32_t spread10(uint32_t v) {
  v &= 0x3FF;
  v = (v | (v << 16)) & 0x030000FF;
  v = (v | (v << 8)) & 0x0300F00F;
  v = (v | (v << 4)) & 0x030C30C3;
  v = (v | (v << 2)) & 0x09249249;
  return v;
}
__device__ __forceinline__ uint32_t morton_key(float x, float y, float z, float ox, float oy,
                                               float oz, float inv_cell) {
  const float fx = fminf(fmaxf((x - ox) * inv_cell, 0.f), 1023.f);
  const float fy = fminf(fmaxf((y - oy) * inv_cell, 0.f), 1023.f);
  const float fz = fminf(fmaxf((z - oz) * inv_cell, 0.f), 1023.f);
  // Hilbert curve (Skilling's transpose form) rather than Morton: no long jumps between
  // consecutive cells, so the 128-point chunks and 16-point sub-blocks get tighter boxes.
  uint32_t X[3] = {(uint32_t)fx, (uint32_t)fy, (uint32_t)fz};
  const uint32_t M = 1u << 9;
  for (uint32_t Q = M; Q > 1; Q >>= 1) {
    const uint32_t P = Q - 1;
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      if (X[i] & Q) {
        X[0] ^= P;
      } else {
        const uint32_t t = (X[0] ^ X[i]) & P;
        X[0] ^= t;
        X[i] ^= t;
      }
    }
  }
  X[1] ^= X[0];
  X[2] ^= X[1];
  uint32_t t = 0;
  for (uint32_t Q = M; Q > 1; Q >>= 1)
    if (X[2] & Q) t ^= Q - 1;
  X[0] ^= t; X[1] ^= t; X[2] ^= t;
  return (spread10(X[0]) << 2) | (spread10(X[1]) << 1) | spread10(X[2]);
}

__global__ void morton_keys_kernel(const float* __restrict__ xyz, uint32_t n, float ox, float oy,
                                   float oz, float inv_cell, uint32_t* __restrict__ keys,
                                   uint32_t* __restrict__ vals) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  keys[i] = morton_key(xyz[3 * (size_t)i], xyz[3 * (size_t)i + 1], xyz[3 * (size_t)i + 2], ox, oy,
                       oz, inv_cell);
  vals[i] = i;
}

__global__ void gather_sorted_kernel(const float* __restrict__ xyz, const uint32_t* __restrict__ perm,
                                     uint32_t n, f32x4* __restrict__ pts,
                                     uint32_t* __restrict__ inv) {
  const uint32_t s = blockIdx.x * blockDim.x + threadIdx.x;
  if (s >= n) return;
  const uint32_t o = perm[s];
  pts[s] = f32x4{xyz[3 * (size_t)o], xyz[3 * (size_t)o + 1], xyz[3 * (size_t)o + 2],
                 __uint_as_float(o)};
  inv[o] = s;
}

// one wave per chunk
__global__ __launch_bounds__(64) void chunk_boxes_kernel(const f32x4* __restrict__ pts, uint32_t n,
                                                         f32x4* __restrict__ lo,
                                                         f32x4* __restrict__ hi) {
  const uint32_t c = blockIdx.x, lane = threadIdx.x;
  float mn[3] = {3.4e38f, 3.4e38f, 3.4e38f}, mx[3] = {-3.4e38f, -3.4e38f, -3.4e38f};
  for (uint32_t t = lane; t < CH; t += 64) {
    const uint32_t j = c * CH + t;
    if (j < n) {
      const f32x4 p = pts[j];
      mn[0] = fminf(mn[0], p.x); mx[0] = fmaxf(mx[0], p.x);
      mn[1] = fminf(mn[1], p.y); mx[1] = fmaxf(mx[1], p.y);
      mn[2] = fminf(mn[2], p.z); mx[2] = fmaxf(mx[2], p.z);
    }
  }
  for (int o = 32; o > 0; o >>= 1)
    for (int a = 0; a < 3; ++a) {
      mn[a] = fminf(mn[a], __shfl_xor(mn[a], o));
      mx[a] = fmaxf(mx[a], __shfl_xor(mx[a], o));
    }
  if (lane == 0) {
    lo[c] = f32x4{mn[0], mn[1], mn[2], 0.f};
    hi[c] = f32x4{mx[0], mx[1], mx[2], 0.f};
  }
}

// one thread per sub-block of SB points
__global__ void subblock_boxes_kernel(const f32x4* __restrict__ pts, uint32_t n, uint32_t nsb,
                                      f32x4* __restrict__ lo, f32x4* __restrict__ hi) {
  const uint32_t b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= nsb) return;
  float mn[3] = {3.4e38f, 3.4e38f, 3.4e38f}, mx[3] = {-3.4e38f, -3.4e38f, -3.4e38f};
  for (uint32_t t = 0; t < SB; ++t) {
    const uint32_t j = b * SB + t;
    if (j < n) {
      const f32x4 p = pts[j];
      mn[0] = fminf(mn[0], p.x); mx[0] = fmaxf(mx[0], p.x);
      mn[1] = fminf(mn[1], p.y); mx[1] = fmaxf(mx[1], p.y);
      mn[2] = fminf(mn[2], p.z); mx[2] = fmaxf(mx[2], p.z);
    }
  }
  lo[b] = f32x4{mn[0], mn[1], mn[2], 0.f};
  hi[b] = f32x4{mx[0], mx[1], mx[2], 0.f};
}

// squared distance from a point to a box, scaled down by 2^-20 so that fp32 rounding can never
// push it above the (fp32) distance of any point inside the box
__device__ __forceinline__ float box_lb(float px, float py, float pz, const f32x4& lo,
                                        const f32x4& hi) {
  const float ex = fmaxf(fmaxf(lo.x - px, px - hi.x), 0.f);
  const float ey = fmaxf(fmaxf(lo.y - py, py - hi.y), 0.f);
  const float ez = fmaxf(fmaxf(lo.z - pz, pz - hi.z), 0.f);
  return ((ex * ex + ey * ey) + ez * ez) * 0.99999905f;
}


// CS = source points per lane.  grid = (ceil(n_src / (256*CS)), n_cand); work-group = 4 independent
// waves of 64*CS sources each.
// src4: Morton-sorted source points (x, y, z, bits(original index)).
// prev_corr (may be null): the previous pass's correspondences [cand][ld] by original source index.
template <int CS>
__global__ __launch_bounds__(256) void nn_culled_kernel(
    const f32x4* __restrict__ src4, uint32_t n_src, const CulledCand* __restrict__ ccands,
    const CandState* __restrict__ states, const uint32_t* __restrict__ prev_corr,
    uint32_t* __restrict__ corr, float* __restrict__ d2out, size_t ld,
    unsigned long long* __restrict__ stat_chunks /* pairs evaluated */,
    uint32_t* __restrict__ trace /* dev only: [wave][4] = cycles, candidate chunks, chunks, sub-blocks */) {
  __shared__ f32x4 stage_all[4][CH];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  f32x4* stage = stage_all[w];
  const int cand = blockIdx.y;
  // The index arrays are reached through pointers loaded from memory, which the compiler would
  // treat as generic (flat_load): view them in the global address space explicitly.
  struct IndexView {
    GPTR(f32x4) pts; GPTR(f32x4) box_lo; GPTR(f32x4) box_hi; GPTR(f32x4) sb_lo; GPTR(f32x4) sb_hi;
    GPTR(uint32_t) keys; GPTR(uint32_t) inv;
    uint32_t n, nchunks;
    float ox, oy, oz, inv_cell;
  };
  const ScanIndexDev ixg = ccands[cand].idx;
  const IndexView ix{(GPTR(f32x4))ixg.pts, (GPTR(f32x4))ixg.box_lo, (GPTR(f32x4))ixg.box_hi,
                     (GPTR(f32x4))ixg.sb_lo, (GPTR(f32x4))ixg.sb_hi, (GPTR(uint32_t))ixg.keys,
                     (GPTR(uint32_t))ixg.inv, ixg.n, ixg.nchunks, ixg.ox, ixg.oy, ixg.oz, ixg.inv_cell};
  GPTR(float) txyz = (GPTR(float))ccands[cand].xyz;
  float T[12];
#pragma unroll
  for (int i = 0; i < 12; ++i) T[i] = states[cand].Tf[i];

  const uint32_t wave_base = (blockIdx.x * 4 + w) * (64 * CS);
  if (wave_base >= n_src) return;
  const unsigned long long t_start = trace ? __builtin_amdgcn_s_memtime() : 0ull;
  uint32_t n_cand_chunks = 0;  // whole wave idle (no barriers are used below)

  float px[CS], py[CS], pz[CS], best[CS];
  uint32_t orig[CS], bch[CS];  // bch: index of the 16-target sub-block holding the minimum
  bool valid[CS], tie[CS];
  float wlo[3] = {3.4e38f, 3.4e38f, 3.4e38f}, whi[3] = {-3.4e38f, -3.4e38f, -3.4e38f};
#pragma unroll
  for (int s = 0; s < CS; ++s) {
    const uint32_t i = wave_base + s * 64 + lane;
    valid[s] = i < n_src;
    const f32x4 p = src4[valid[s] ? i : (n_src - 1)];
    orig[s] = __float_as_uint(p.w);
    xform(T, p.x, p.y, p.z, px[s], py[s], pz[s]);
    wlo[0] = fminf(wlo[0], px[s]); whi[0] = fmaxf(whi[0], px[s]);
    wlo[1] = fminf(wlo[1], py[s]); whi[1] = fmaxf(whi[1], py[s]);
    wlo[2] = fminf(wlo[2], pz[s]); whi[2] = fmaxf(whi[2], pz[s]);
    best[s] = 3.402823466e+38f;
    bch[s] = 0;
    tie[s] = false;
  }
  for (int o = 32; o > 0; o >>= 1)
#pragma unroll
    for (int a = 0; a < 3; ++a) {
      wlo[a] = fminf(wlo[a], __shfl_xor(wlo[a], o));
      whi[a] = fmaxf(whi[a], __shfl_xor(whi[a], o));
    }

  // ---- upper bounds -------------------------------------------------------------------------
  if (ix.n) {
#pragma unroll
    for (int s = 0; s < CS; ++s) {
      uint32_t j = 0xFFFFFFFFu;
      if (prev_corr) j = prev_corr[(size_t)cand * ld + orig[s]];
      if (j < ix.n) {
        best[s] = dist2(px[s], py[s], pz[s], txyz[3 * (size_t)j], txyz[3 * (size_t)j + 1],
                        txyz[3 * (size_t)j + 2]);
        bch[s] = ix.inv[j] / SB;
      } else {
        const uint32_t key = morton_key(px[s], py[s], pz[s], ix.ox, ix.oy, ix.oz, ix.inv_cell);
        uint32_t lo = 0, hi = ix.n;  // lower_bound over the sorted keys
        while (lo < hi) {
          const uint32_t mid = (lo + hi) >> 1;
          if (ix.keys[mid] < key) lo = mid + 1; else hi = mid;
        }
        for (int d = -2; d <= 2; ++d) {
          long long jj = (long long)lo + d;
          jj = jj < 0 ? 0 : (jj >= (long long)ix.n ? (long long)ix.n - 1 : jj);
          const f32x4 t = ix.pts[jj];
          const float dd = dist2(px[s], py[s], pz[s], t.x, t.y, t.z);
          if (dd < best[s]) {
            best[s] = dd;
            bch[s] = (uint32_t)jj / SB;
          }
        }
      }
    }
  }
  auto wave_max_best = [&]() {
    float m = -1.f;
#pragma unroll
    for (int s = 0; s < CS; ++s) m = fmaxf(m, valid[s] ? best[s] : -1.f);
    for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
    return m;
  };
  float wmax = wave_max_best();
  unsigned long long n_processed = 0, n_sub = 0;

  // ---- sweep: 64 chunk boxes per ballot -----------------------------------------------------
  f32x4 nlo = {0.f, 0.f, 0.f, 0.f}, nhi = {0.f, 0.f, 0.f, 0.f};  // next batch's boxes, in flight
  if ((uint32_t)lane < ix.nchunks) {
    nlo = ix.box_lo[lane];
    nhi = ix.box_hi[lane];
  }
  for (uint32_t c0 = 0; c0 < ix.nchunks; c0 += 64) {
    const uint32_t cl = c0 + lane;
    const f32x4 blo = nlo, bhi = nhi;
    if (cl + 64 < ix.nchunks) {  // loads of the following batch overlap this batch's work
      nlo = ix.box_lo[cl + 64];
      nhi = ix.box_hi[cl + 64];
    }
    float lbw = 3.402823466e+38f;
    if (cl < ix.nchunks) {
      const float ex = fmaxf(fmaxf(blo.x - whi[0], wlo[0] - bhi.x), 0.f);
      const float ey = fmaxf(fmaxf(blo.y - whi[1], wlo[1] - bhi.y), 0.f);
      const float ez = fmaxf(fmaxf(blo.z - whi[2], wlo[2] - bhi.z), 0.f);
      lbw = ((ex * ex + ey * ey) + ez * ez) * 0.99999905f;
    }
    unsigned long long mask = __ballot(lbw <= wmax);
    while (mask) {
      const int b = __ffsll((long long)mask) - 1;
      mask &= mask - 1;
      // the wave's bound may have tightened since the ballot
      if (__int_as_float(__builtin_amdgcn_readlane(__float_as_int(lbw), b)) > wmax) continue;
      const uint32_t c = c0 + b;
      n_cand_chunks++;
      f32x4 lo, hi;
      lo.x = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(blo.x), b));
      lo.y = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(blo.y), b));
      lo.z = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(blo.z), b));
      hi.x = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(bhi.x), b));
      hi.y = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(bhi.y), b));
      hi.z = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(bhi.z), b));
      bool need = false;
#pragma unroll
      for (int s = 0; s < CS; ++s)
        need |= valid[s] && (box_lb(px[s], py[s], pz[s], lo, hi) <= best[s]);
      if (!__any(need)) continue;
      n_processed++;
      // stage the chunk (wave-private LDS slice; padding never wins)
#pragma unroll
      for (int u = 0; u < CH / 64; ++u) {
        const uint32_t j = c * CH + u * 64 + lane;
        f32x4 v = {NN_FAR, NN_FAR, NN_FAR, 0.f};
        if (j < ix.n) v = ix.pts[j];
        stage[u * 64 + lane] = v;
      }
      // the 8 sub-block boxes: lanes 0..7 load one each (issued together with the staging loads)
      f32x4 sbl = {0.f, 0.f, 0.f, 0.f}, sbh = {0.f, 0.f, 0.f, 0.f};
      {
        const uint32_t blk_l = c * (CH / SB) + (lane & 7);
        if (blk_l * SB < ix.n) {
          sbl = ix.sb_lo[blk_l];
          sbh = ix.sb_hi[blk_l];
        }
      }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
      bool changed = false;
      for (int b0 = 0; b0 < CH; b0 += SB) {
        const uint32_t blk = (c * CH + b0) / SB;
        if (blk * SB >= ix.n) break;
        {  // second-level test: does any point of the wave still need this sub-block?
          const int bi = b0 / SB;
          f32x4 slo, shi;
          slo.x = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(sbl.x), bi));
          slo.y = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(sbl.y), bi));
          slo.z = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(sbl.z), bi));
          shi.x = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(sbh.x), bi));
          shi.y = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(sbh.y), bi));
          shi.z = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(sbh.z), bi));
          bool need_sb = false;
#pragma unroll
          for (int s = 0; s < CS; ++s)
            need_sb |= valid[s] && (box_lb(px[s], py[s], pz[s], slo, shi) <= best[s]);
          if (!__any(need_sb)) continue;
          n_sub++;
        }
        float m[CS];
#pragma unroll
        for (int s = 0; s < CS; ++s) m[s] = 3.402823466e+38f;
#pragma unroll 2  // deeper unrolling costs 42 VGPRs (126 vs 84) and a wave per SIMD: 5 % slower
        for (int t = b0; t < b0 + SB; t += 2) {
          const f32x4 q0 = stage[t];
          const f32x4 q1 = stage[t + 1];
#pragma unroll
          for (int s = 0; s < CS; ++s) {
            const float d0 = dist2(px[s], py[s], pz[s], q0.x, q0.y, q0.z);
            const float d1 = dist2(px[s], py[s], pz[s], q1.x, q1.y, q1.z);
            m[s] = fminf(fminf(m[s], d0), d1);
          }
        }
#pragma unroll
        for (int s = 0; s < CS; ++s) {
          if (m[s] < best[s]) {
            best[s] = m[s];
            bch[s] = blk;
            tie[s] = false;
            changed = true;
          } else if (m[s] == best[s] && blk != bch[s]) {
            tie[s] = true;  // an equally near point elsewhere: resolved by original index below
          }
        }
      }
      __builtin_amdgcn_wave_barrier();  // all reads done before the slice is overwritten
      if (__any(changed)) wmax = wave_max_best();
    }
  }
  (void)n_processed;
  if (stat_chunks && lane == 0) atomicAdd(stat_chunks, n_sub * (unsigned long long)(64 * CS * SB));

  // ---- index recovery: smallest ORIGINAL index among the targets at the minimum distance ----
#pragma unroll
  for (int s = 0; s < CS; ++s) {
    if (!valid[s]) continue;
    uint32_t bj = 0xFFFFFFFFu;
    if (ix.n) {
      if (!tie[s]) {
        const uint32_t j0 = bch[s] * SB;
        const uint32_t j1 = (j0 + SB) < ix.n ? (j0 + SB) : ix.n;
        for (uint32_t j = j0; j < j1; ++j) {
          const f32x4 t = ix.pts[j];
          if (dist2(px[s], py[s], pz[s], t.x, t.y, t.z) == best[s]) {
            const uint32_t o = __float_as_uint(t.w);
            bj = o < bj ? o : bj;
          }
        }
      } else {  // rare: every chunk that can hold a point at the minimum distance
        for (uint32_t c = 0; c < ix.nchunks; ++c) {
          const f32x4 clo = ix.box_lo[c], chi = ix.box_hi[c];
          if (box_lb(px[s], py[s], pz[s], clo, chi) > best[s]) continue;
          const uint32_t j0 = c * CH;
          const uint32_t j1 = (j0 + CH) < ix.n ? (j0 + CH) : ix.n;
          for (uint32_t j = j0; j < j1; ++j) {
            const f32x4 t = ix.pts[j];
            if (dist2(px[s], py[s], pz[s], t.x, t.y, t.z) == best[s]) {
              const uint32_t o = __float_as_uint(t.w);
              bj = o < bj ? o : bj;
            }
          }
        }
      }
    }
    corr[(size_t)cand * ld + orig[s]] = bj;
    d2out[(size_t)cand * ld + orig[s]] = best[s];
  }
  if (trace && lane == 0) {
    const size_t wid = ((size_t)blockIdx.y * gridDim.x + blockIdx.x) * 4 + w;
    trace[4 * wid + 0] = (uint32_t)(__builtin_amdgcn_s_memtime() - t_start);
    trace[4 * wid + 1] = n_cand_chunks;
    trace[4 * wid + 2] = (uint32_t)n_processed;
    trace[4 * wid + 3] = (uint32_t)n_sub;
  }
}

}  // namespace reg
}  // namespace gloc
