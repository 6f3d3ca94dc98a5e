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

typedef float f32x2 __attribute__((ext_vector_type(2)));

#define GPTR(T) const T __attribute__((address_space(1)))*
constexpr int CH = 128;  // points per chunk (256 / 32 measured: -9 % throughput with three queries in flight)
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
  const f32x4* sup_lo;   // per super-chunk of 64 chunks (8192 points)
  const f32x4* sup_hi;
  uint32_t nsup;
};

struct CulledCand {
  ScanIndexDev idx;
  const float* xyz;  // original order, packed
};

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

// one wave per super-chunk: the union of 64 chunk boxes
__global__ __launch_bounds__(64) void super_boxes_kernel(const f32x4* __restrict__ lo, const f32x4* __restrict__ hi,
                                                         uint32_t nchunks, f32x4* __restrict__ slo,
                                                         f32x4* __restrict__ shi) {
  const uint32_t c = blockIdx.x * 64 + threadIdx.x;
  float mn[3] = {3.4e38f, 3.4e38f, 3.4e38f}, mx[3] = {-3.4e38f, -3.4e38f, -3.4e38f};
  if (c < nchunks) {
    const f32x4 a = lo[c], b = hi[c];
    mn[0] = a.x; mn[1] = a.y; mn[2] = a.z;
    mx[0] = b.x; mx[1] = b.y; mx[2] = b.z;
  }
  for (int o = 32; o > 0; o >>= 1)
    for (int a = 0; a < 3; ++a) {
      mn[a] = fminf(mn[a], __shfl_xor(mn[a], o));
      mx[a] = fmaxf(mx[a], __shfl_xor(mx[a], o));
    }
  if (threadIdx.x == 0) {
    slo[blockIdx.x] = f32x4{mn[0], mn[1], mn[2], 0.f};
    shi[blockIdx.x] = f32x4{mx[0], mx[1], mx[2], 0.f};
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

// Spatial extent of every group of `group` consecutive (Hilbert-sorted) points: the squared diagonal
// of its bounding box.  A wave's sweep cost grows with the extent of its sources (more chunk boxes
// pass the wave-level test), so launching the widest groups first keeps the stragglers off the tail.
// One wave per group.
__global__ __launch_bounds__(256) void group_extent_kernel(const f32x4* __restrict__ pts, uint32_t n,
                                                            uint32_t group, uint32_t n_groups,
                                                            float* __restrict__ ext, uint32_t* __restrict__ ids) {
  const uint32_t g = blockIdx.x * 4 + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (g >= n_groups) return;
  float lo[3] = {3.4e38f, 3.4e38f, 3.4e38f}, hi[3] = {-3.4e38f, -3.4e38f, -3.4e38f};
  for (uint32_t i = g * group + lane; i < (g + 1) * group && i < n; i += 64) {
    const f32x4 p = pts[i];
    lo[0] = fminf(lo[0], p.x); hi[0] = fmaxf(hi[0], p.x);
    lo[1] = fminf(lo[1], p.y); hi[1] = fmaxf(hi[1], p.y);
    lo[2] = fminf(lo[2], p.z); hi[2] = fmaxf(hi[2], p.z);
  }
  for (int o = 32; o > 0; o >>= 1)
#pragma unroll
    for (int a = 0; a < 3; ++a) {
      lo[a] = fminf(lo[a], __shfl_xor(lo[a], o));
      hi[a] = fmaxf(hi[a], __shfl_xor(hi[a], o));
    }
  if (lane == 0) {
    const float dx = hi[0] - lo[0], dy = hi[1] - lo[1], dz = hi[2] - lo[2];
    ext[g] = dx * dx + dy * dy + dz * dz;
    ids[g] = g;
  }
}

// ---- compacted evaluation --------------------------------------------------------------------
// Same sweep and the same tests as nn_culled_kernel, but a sub-block is no longer evaluated by the
// whole wave when any lane needs it (that form spends ~85 % of its distance evaluations on lanes
// that did not ask).  Here every (source, sub-block) pair that passes the box test becomes a work
// item in a wave-private LDS queue.  The tests themselves are compacted too: the sources that passed
// the chunk-level test are listed (ballot + mbcnt) and the (source, sub-block) box tests run one pair
// per lane over that list only.  The wave then consumes the queue 64 items at a time: a lane fetches ITS item's source from LDS,
// walks ITS sub-block's 16 staged targets and folds the minimum into the source's packed
// (d2 bits << 32 | sub-block) key with an LDS atomic min.  Lanes of a round that share a sub-block
// read the same LDS address (broadcast); different sub-blocks are walked with a rotated start so
// their 16-B reads fall into disjoint bank groups.  An equal minimum from a different sub-block
// raises the source's tie flag (resolved below by smallest original index, as before).
// The result is the one nn_culled_kernel and nn_kernel produce: same d2 bits, same indices.
//
// Launch order (1-D grid of n_wg * n_cand work-groups; `order` lists the source groups widest first):
// the first heavy_wgs work-groups of EVERY candidate come first (candidate fastest), so the few very
// long waves start at t = 0 and finish under cover of the bulk; the rest run candidate by candidate,
// which keeps one candidate's points and boxes hot in the L2s.
template <int CS>
__global__ __launch_bounds__(256) void nn_compact_kernel(
    const f32x4* __restrict__ src4, uint32_t n_src, const CulledCand* __restrict__ ccands,
    const CandState* __restrict__ states, const uint32_t* __restrict__ prev_corr,
    uint32_t* __restrict__ corr, float* __restrict__ d2out, size_t ld,
    const uint32_t* __restrict__ order, uint32_t n_groups, uint32_t n_cand, uint32_t heavy_wgs,
    unsigned long long* __restrict__ stat_pairs /* pairs evaluated */,
    uint32_t* __restrict__ trace /* dev only: [wave][4] = cycles, candidate chunks, chunks, rounds */) {
  constexpr int S = 64 * CS;        // sources per wave
  constexpr int NSB = CH / SB;      // sub-blocks per chunk
  // The staged chunk is kept as PAIRS of targets, structure-of-arrays: pair i of a sub-block is
  // (x0, x1, y0, y1 | z0, z1, -, -), 32 B, so that one packed fp32 instruction handles two targets
  // (v_pk_add/mul_f32 round each half like the scalar forms: same bits).  Sub-blocks are 288 B apart:
  // the extra 32 B shift sub-block b's pair i into bank group (i + b) % 8, so lanes that walk
  // different sub-blocks in lock step never collide.
  constexpr int SB_STRIDE = (SB / 2) * 8 + 8;  // floats per sub-block: SB/2 pairs x 8 floats + 8 of shift
  struct WaveLds {
    float stage[NSB * SB_STRIDE];   // the chunk being evaluated
    f32x4 src[S];                   // moved source points
    unsigned long long key[S];      // (bits(best d2) << 32) | sub-block holding it
    uint8_t tie[S];                 // (sizes are chosen so that CS = 2 stays under 8 KB per wave: 5 work-groups per CU)
    f32x4 sblo[NSB], sbhi[NSB];     // the chunk's sub-block boxes
    uint16_t list[S];               // source slots that passed the chunk-level test
    uint16_t queue[S * NSB];        // work items: (source slot << 3) | sub-block within the chunk
  };
  static_assert(SB % 16 == 0 && NSB == 8, "items pack the sub-block into 3 bits");
  __shared__ WaveLds lds_all[4];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  WaveLds& L = lds_all[w];
  const uint32_t n_wg = (n_groups + 3) / 4;
  uint32_t cand, wg;
  if (blockIdx.x < heavy_wgs * n_cand) {
    cand = blockIdx.x % n_cand;
    wg = blockIdx.x / n_cand;
  } else {
    const uint32_t rest = blockIdx.x - heavy_wgs * n_cand, per = n_wg - heavy_wgs;
    cand = rest / per;
    wg = heavy_wgs + rest % per;
  }
  const uint32_t gi = wg * 4 + w;
  if (gi >= n_groups) return;  // whole wave idle (no work-group barriers are used below)
  struct IndexView {
    GPTR(f32x4) pts; GPTR(f32x4) box_lo; GPTR(f32x4) box_hi; GPTR(f32x4) sb_lo; GPTR(f32x4) sb_hi;
    GPTR(uint32_t) keys; GPTR(uint32_t) inv;
    uint32_t n, nchunks;
    float ox, oy, oz, inv_cell;
    GPTR(f32x4) sup_lo; GPTR(f32x4) sup_hi;
    uint32_t nsup;
  };
  const ScanIndexDev ixg = ccands[cand].idx;
  const IndexView ix{(GPTR(f32x4))ixg.pts, (GPTR(f32x4))ixg.box_lo, (GPTR(f32x4))ixg.box_hi,
                     (GPTR(f32x4))ixg.sb_lo, (GPTR(f32x4))ixg.sb_hi, (GPTR(uint32_t))ixg.keys,
                     (GPTR(uint32_t))ixg.inv, ixg.n, ixg.nchunks, ixg.ox, ixg.oy, ixg.oz, ixg.inv_cell,
                     (GPTR(f32x4))ixg.sup_lo, (GPTR(f32x4))ixg.sup_hi, ixg.nsup};
  GPTR(float) txyz = (GPTR(float))ccands[cand].xyz;
  float T[12];
#pragma unroll
  for (int i = 0; i < 12; ++i) T[i] = states[cand].Tf[i];

  const uint32_t wave_base = order[gi] * S;
  const unsigned long long t_start = trace ? __builtin_amdgcn_s_memtime() : 0ull;
  uint32_t n_cand_chunks = 0, n_processed = 0, n_rounds = 0;
  unsigned long long n_items = 0;

  float px[CS], py[CS], pz[CS], best[CS];
  uint32_t orig[CS];
  bool valid[CS];
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
  }
  for (int o = 32; o > 0; o >>= 1)
#pragma unroll
    for (int a = 0; a < 3; ++a) {
      wlo[a] = fminf(wlo[a], __shfl_xor(wlo[a], o));
      whi[a] = fmaxf(whi[a], __shfl_xor(whi[a], o));
    }

  // ---- upper bounds (as nn_culled_kernel) -> LDS state ----------------------------------------
#pragma unroll
  for (int s = 0; s < CS; ++s) {
    uint32_t b0 = 0;
    if (ix.n) {
      uint32_t j = 0xFFFFFFFFu;
      if (prev_corr) j = prev_corr[(size_t)cand * ld + orig[s]];
      if (j < ix.n) {
        best[s] = dist2(px[s], py[s], pz[s], txyz[3 * (size_t)j], txyz[3 * (size_t)j + 1],
                        txyz[3 * (size_t)j + 2]);
        b0 = ix.inv[j] / SB;
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
            b0 = (uint32_t)jj / SB;
          }
        }
      }
    }
    const int slot = s * 64 + lane;
    L.src[slot] = f32x4{px[s], py[s], pz[s], 0.f};
    L.key[slot] = ((unsigned long long)__float_as_uint(best[s]) << 32) | b0;
    L.tie[slot] = 0;
  }
  auto wave_max_best = [&]() {
    float m = -1.f;
#pragma unroll
    for (int s = 0; s < CS; ++s) m = fmaxf(m, valid[s] ? best[s] : -1.f);
    for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
    return m;
  };
  float wmax = wave_max_best();

  // ---- sweep: super-chunk boxes first (64 per ballot), then 64 chunk boxes per surviving batch ----
  auto box_box_lb = [&](const f32x4& blo, const f32x4& bhi) {
    const float ex = fmaxf(fmaxf(blo.x - whi[0], wlo[0] - bhi.x), 0.f);
    const float ey = fmaxf(fmaxf(blo.y - whi[1], wlo[1] - bhi.y), 0.f);
    const float ez = fmaxf(fmaxf(blo.z - whi[2], wlo[2] - bhi.z), 0.f);
    return ((ex * ex + ey * ey) + ez * ez) * 0.99999905f;
  };
  for (uint32_t s0 = 0; s0 < ix.nsup; s0 += 64) {
    float lbs = 3.402823466e+38f;
    if (s0 + lane < ix.nsup) {
      const f32x4 ulo = ix.sup_lo[s0 + lane], uhi = ix.sup_hi[s0 + lane];
      lbs = box_box_lb(ulo, uhi);
    }
    unsigned long long smask = __ballot(lbs <= wmax);
    // the chunk boxes of the NEXT surviving batch are in flight while the current one is worked on
    f32x4 nlo = {0.f, 0.f, 0.f, 0.f}, nhi = {0.f, 0.f, 0.f, 0.f};
    int cur = -1;
    if (smask) {
      cur = __ffsll((long long)smask) - 1;
      smask &= smask - 1;
      const uint32_t cl = (s0 + cur) * 64 + lane;
      if (cl < ix.nchunks) { nlo = ix.box_lo[cl]; nhi = ix.box_hi[cl]; }
    }
    while (cur >= 0) {
    const uint32_t c0 = (s0 + cur) * 64;
    const bool live = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(lbs), cur)) <= wmax;
    const uint32_t cl = c0 + lane;
    const f32x4 blo = nlo, bhi = nhi;
    cur = -1;
    if (smask) {
      cur = __ffsll((long long)smask) - 1;
      smask &= smask - 1;
      const uint32_t cn = (s0 + cur) * 64 + lane;
      if (cn < ix.nchunks) { nlo = ix.box_lo[cn]; nhi = ix.box_hi[cn]; }
    }
    if (!live) continue;  // the wave's bound tightened since the super-chunk ballot
    float lbw = 3.402823466e+38f;
    if (cl < ix.nchunks) lbw = box_box_lb(blo, bhi);
    unsigned long long mask = __ballot(lbw <= wmax);
    while (mask) {
      const int b = __ffsll((long long)mask) - 1;
      mask &= mask - 1;
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
      bool need[CS], any_need = false;
#pragma unroll
      for (int s = 0; s < CS; ++s) {
        need[s] = valid[s] && (box_lb(px[s], py[s], pz[s], lo, hi) <= best[s]);
        any_need |= need[s];
      }
      if (!__any(any_need)) continue;
      n_processed++;
      // stage the chunk (wave-private LDS; padding never wins) and fetch its 8 sub-block boxes
#pragma unroll
      for (int u = 0; u < CH / 64; ++u) {
        const uint32_t tl = u * 64 + lane, j = c * CH + tl;
        f32x4 v = {NN_FAR, NN_FAR, NN_FAR, 0.f};
        if (j < ix.n) v = ix.pts[j];
        float* d = &L.stage[(tl / SB) * SB_STRIDE + ((tl % SB) >> 1) * 8 + (tl & 1)];
        d[0] = v.x; d[2] = v.y; d[4] = v.z;
      }
      f32x4 sbl = {0.f, 0.f, 0.f, 0.f}, sbh = {0.f, 0.f, 0.f, 0.f};
      {
        const uint32_t blk_l = c * NSB + (lane & 7);
        if (blk_l * SB < ix.n) {
          sbl = ix.sb_lo[blk_l];
          sbh = ix.sb_hi[blk_l];
        }
      }
      // sources that passed the chunk-level test, compacted
      uint32_t k = 0;
#pragma unroll
      for (int s = 0; s < CS; ++s) {
        const unsigned long long m = __ballot(need[s]);
        if (need[s])
          L.list[k + __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u))] =
              (uint16_t)(s * 64 + lane);
        k += (uint32_t)__popcll(m);
      }
      if (lane < NSB) {
        L.sblo[lane] = sbl;
        L.sbhi[lane] = sbh;
      }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
      // sub-block tests, one (source, sub-block) pair per lane: only the listed sources are tested,
      // against their CURRENT bound; the passing pairs become the work items
      const uint32_t left = ix.n - c * CH;  // > 0: the chunk exists
      const uint32_t nsb_valid = left >= (uint32_t)CH ? (uint32_t)NSB : (left + SB - 1) / SB;
      uint32_t total = 0;
      // four steps of 64 pairs at a time: their LDS reads and box tests are independent, so a wave
      // (latency-bound when few share the SIMD) overlaps them; only the queue positions are serial
      constexpr int TU = 4;
      for (uint32_t t0 = 0; t0 < k * NSB; t0 += 64 * TU) {
        uint32_t si[TU];
        bool act[TU];
#pragma unroll
        for (int u = 0; u < TU; ++u) {
          const uint32_t t = t0 + u * 64 + lane;
          act[u] = t < k * NSB;
          si[u] = L.list[act[u] ? (t >> 3) : 0];
        }
        const uint32_t sb = lane & 7;  // (t & 7): t0 and u * 64 are multiples of 8
        const f32x4 slo = L.sblo[sb], shi = L.sbhi[sb];
        f32x4 p[TU];
        float bst[TU];
#pragma unroll
        for (int u = 0; u < TU; ++u) {
          p[u] = L.src[si[u]];
          bst[u] = __uint_as_float((uint32_t)(L.key[si[u]] >> 32));
        }
#pragma unroll
        for (int u = 0; u < TU; ++u) {
          const bool nd = act[u] && sb < nsb_valid && (box_lb(p[u].x, p[u].y, p[u].z, slo, shi) <= bst[u]);
          const unsigned long long m = __ballot(nd);
          if (nd)
            L.queue[total + __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u))] =
                (uint16_t)((si[u] << 3) | sb);
          total += (uint32_t)__popcll(m);
        }
      }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
      n_items += total;
      for (uint32_t r = 0; r < total; r += 64) {
        n_rounds++;
        const uint32_t it = r + lane;
        const bool act = it < total;
        const uint32_t item = L.queue[act ? it : r];
        const uint32_t slot = item >> 3, bi = item & 7;
        const f32x4 p = L.src[slot];
        const float* sb = &L.stage[bi * SB_STRIDE];
        const f32x2 ppx = {p.x, p.x}, ppy = {p.y, p.y}, ppz = {p.z, p.z};
        float m = 3.402823466e+38f;
#pragma unroll
        for (int i = 0; i < SB / 2; ++i) {
          const f32x4 xy = *reinterpret_cast<const f32x4*>(sb + i * 8);
          const f32x2 zz = *reinterpret_cast<const f32x2*>(sb + i * 8 + 4);
          // dist2() on two targets at once: d = p - q per axis, (dx*dx + dy*dy) + dz*dz, un-fused
          const f32x2 dx = ppx - f32x2{xy.x, xy.y}, dy = ppy - f32x2{xy.z, xy.w}, dz = ppz - zz;
          const f32x2 d2 = (dx * dx + dy * dy) + dz * dz;
          m = fminf(fminf(m, d2.x), d2.y);
        }
        if (act) {
          const uint32_t blk = c * NSB + bi;
          const unsigned long long key = ((unsigned long long)__float_as_uint(m) << 32) | blk;
          const unsigned long long old = atomicMin(&L.key[slot], key);
          if ((uint32_t)(old >> 32) == __float_as_uint(m) && (uint32_t)old != blk) L.tie[slot] = 1;
        }
      }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();  // all reads of the stage done, all key updates visible
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
      bool changed = false;
#pragma unroll
      for (int s = 0; s < CS; ++s) {
        const float nb = __uint_as_float((uint32_t)(L.key[s * 64 + lane] >> 32));
        changed |= nb < best[s];
        best[s] = nb;
      }
      if (__any(changed)) wmax = wave_max_best();
    }
    }  // batches of this super-chunk group
  }
  if (stat_pairs && lane == 0) atomicAdd(stat_pairs, n_items * (unsigned long long)SB);

  // ---- index recovery: smallest ORIGINAL index among the targets at the minimum distance ----
#pragma unroll
  for (int s = 0; s < CS; ++s) {
    if (!valid[s]) continue;
    const int slot = s * 64 + lane;
    const uint32_t bch = (uint32_t)L.key[slot];
    const bool tie = L.tie[slot] != 0;
    uint32_t bj = 0xFFFFFFFFu;
    if (ix.n) {
      if (!tie) {
        // all 16 loads in flight at once (clamped, so that they are unconditional): one memory
        // round trip instead of sixteen -- a lone wave is latency-bound here
        for (uint32_t j0 = bch * SB; j0 < (bch + 1) * SB; j0 += 16) {  // 16 loads (64 VGPRs) at a time
          f32x4 t[16];
#pragma unroll
          for (int u = 0; u < 16; ++u) t[u] = ix.pts[(j0 + u) < ix.n ? (j0 + u) : (ix.n - 1)];
#pragma unroll
          for (int u = 0; u < 16; ++u) {
            if ((j0 + u) < ix.n && dist2(px[s], py[s], pz[s], t[u].x, t[u].y, t[u].z) == best[s]) {
              const uint32_t o = __float_as_uint(t[u].w);
              bj = o < bj ? o : bj;
            }
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
    const size_t wid = ((size_t)cand * n_wg + wg) * 4 + w;
    trace[4 * wid + 0] = (uint32_t)(__builtin_amdgcn_s_memtime() - t_start);
    trace[4 * wid + 1] = n_cand_chunks;
    trace[4 * wid + 2] = n_processed;
    trace[4 * wid + 3] = n_rounds;
  }
}

}  // namespace reg
}  // namespace gloc
