// knn_kernels.hpp -- device kernels of the descriptor kNN (gfx950).  Included only by knn.hip.
//
// Pipeline (kernels K1/K1b/K2/K3):
//   exact path : dist_exact -> select(k)                          -> finalize
//   MFMA  path : row_norms(q) -> dist_mfma -> select(k') -> rerank -> finalize   (+ exact fallback)
//
// Bit-exactness contract: every distance that reaches the caller is computed by exact_pairs_wave(),
// which reproduces nanoflann's L2_Adaptor::evalMetric (registration/nanoflann.hpp:453-487) --
// groups of four differences, ((d0^2 + d1^2) + d2^2) + d3^2, added sequentially over d, scalar tail,
// no FMA contraction (this translation unit is compiled with -ffp-contract=off).
#pragma once
#include <type_traits>
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "lane_ops.hpp"

namespace gloc {
namespace knn {

typedef float f32x4 __attribute__((ext_vector_type(4)));
struct __attribute__((aligned(4))) f4u {  // float4 with 4-byte alignment (rows start at any dword)
  float x, y, z, w;
};

constexpr uint64_t KEY_SENTINEL = ~0ull;

// order-preserving float -> uint32 (handles the small negative values the MFMA form can produce)
__device__ __forceinline__ uint32_t f2ord(float f) {
  uint32_t b = __float_as_uint(f);
  return (b & 0x80000000u) ? ~b : (b | 0x80000000u);
}
__device__ __forceinline__ float ord2f(uint32_t o) {
  uint32_t b = (o & 0x80000000u) ? (o & 0x7FFFFFFFu) : ~o;
  return __uint_as_float(b);
}
__device__ __forceinline__ uint64_t make_key(float d, uint32_t idx) {
  return ((uint64_t)f2ord(d) << 32) | (uint64_t)idx;
}

// ---------------------------------------------------------------------------------------------
// K2: squared row norms (any summation order: they only feed the coarse MFMA form) + running max.
// One wave per row.
__global__ __launch_bounds__(256) void row_norms_kernel(const float* __restrict__ rows, size_t n,
                                                        int dim, float* __restrict__ out,
                                                        uint32_t* __restrict__ max_bits) {
  const int lane = threadIdx.x & 63;
  const size_t row = (size_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= n) return;
  const float* r = rows + row * (size_t)dim;
  float s = 0.f;
  for (int d = lane; d < dim; d += 64) s += r[d] * r[d];
  for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
  if (lane == 0) {
    out[row] = s;
    if (max_bits) atomicMax(max_bits, __float_as_uint(s));  // s >= 0: uint order == float order
  }
}

// ---------------------------------------------------------------------------------------------
// Reference-order distances for up to 64 (query,row) pairs per wave.
//   phase 1 (lane <-> group of 4 dims, coalesced 1-KiB row reads): s_g = ((d0^2+d1^2)+d2^2)+d3^2
//   phase 2 (lane <-> pair): acc += s_g for g ascending -- the reference's sequential chain.
// The group sums cross lanes through a per-wave LDS tile S[64 pairs][68].
// Pair p = qt * RW + r  (qt < QT queries, r < RW rows).  Row r is row_of(r).
constexpr int S_PITCH = 68;  // floats; 272 B rows keep ds_read_b128 aligned and conflict-free

template <int QT, class RowOf>
__device__ __forceinline__ float exact_pairs_wave(const float* __restrict__ db,
                                                  const float* __restrict__ q /* QT rows */,
                                                  int dim, int RW, RowOf row_of, int n_valid_q,
                                                  float* __restrict__ S /* [64][S_PITCH] */) {
  const int lane = threadIdx.x & 63;
  const int G = dim >> 2;
  const int P = QT * RW;
  float acc = 0.f;
  for (int c0 = 0; c0 < G; c0 += 64) {
    const int g = c0 + lane;
    const bool gv = g < G;
    f4u qv[QT];
#pragma unroll
    for (int t = 0; t < QT; ++t) {
      if (gv && t < n_valid_q)
        qv[t] = *reinterpret_cast<const f4u*>(q + (size_t)t * dim + 4 * g);
      else
        qv[t] = f4u{0.f, 0.f, 0.f, 0.f};
    }
    constexpr int RU = 4;  // rows in flight per lane (8 and 16 measured slower: fewer waves per SIMD)
    for (int r0 = 0; r0 < RW; r0 += RU) {
      f4u dv[RU];
#pragma unroll
      for (int u = 0; u < RU; ++u) {
        const int r = r0 + u;
        const long long row = (r < RW) ? row_of(r) : -1;
        if (gv && row >= 0)
          dv[u] = *reinterpret_cast<const f4u*>(db + (size_t)row * dim + 4 * g);
        else
          dv[u] = f4u{0.f, 0.f, 0.f, 0.f};
      }
#pragma unroll
      for (int u = 0; u < RU; ++u) {
        const int r = r0 + u;
        if (r < RW) {
#pragma unroll
          for (int t = 0; t < QT; ++t) {
            const float e0 = qv[t].x - dv[u].x;
            const float e1 = qv[t].y - dv[u].y;
            const float e2 = qv[t].z - dv[u].z;
            const float e3 = qv[t].w - dv[u].w;
            const float s = e0 * e0 + e1 * e1 + e2 * e2 + e3 * e3;  // left-to-right, no FMA
            S[(t * RW + r) * S_PITCH + lane] = s;
          }
        }
      }
    }
    __syncthreads();
    const int ng = (G - c0) < 64 ? (G - c0) : 64;
    if (lane < P) {
      const float* sp = S + lane * S_PITCH;
      int i = 0;
      for (; i + 4 <= ng; i += 4) {
        const f32x4 v = *reinterpret_cast<const f32x4*>(sp + i);
        acc += v.x;
        acc += v.y;
        acc += v.z;
        acc += v.w;
      }
      for (; i < ng; ++i) acc += sp[i];
    }
    __syncthreads();
  }
  // scalar tail, dims 4G..dim-1 (nanoflann.hpp:480-485)
  if ((dim & 3) && lane < P) {
    const int t = lane / RW, r = lane % RW;
    const long long row = row_of(r);
    if (row >= 0 && t < n_valid_q) {
      for (int d = 4 * G; d < dim; ++d) {
        const float e = q[(size_t)t * dim + d] - db[(size_t)row * dim + d];
        acc += e * e;
      }
    }
  }
  return acc;
}

// K1 (exact form): dist[q][j - first] for rows [first, first + n_range).
// grid.x = row tiles (4 waves x RW rows), grid.y = query groups of QT.
template <int QT>
__global__ __launch_bounds__(256) void dist_exact_kernel(const float* __restrict__ db,
                                                         const float* __restrict__ queries,
                                                         float* __restrict__ dist, int dim,
                                                         size_t first_row, int n_range, int nq,
                                                         int RW, size_t ld,
                                                         const int* __restrict__ only_flagged /* QT = 1: run only
                                                         for queries whose flag is set (device-side fallback) */) {
  __shared__ __attribute__((aligned(16))) float S_all[4 * 64 * S_PITCH];
  const int w = threadIdx.x >> 6, lane = threadIdx.x & 63;
  float* S = S_all + w * 64 * S_PITCH;
  const int row0 = (blockIdx.x * 4 + w) * RW;  // relative to first_row
  auto row_of = [&](int r) -> long long {
    const int rr = row0 + r;
    return rr < n_range ? (long long)(first_row + (size_t)rr) : -1;
  };
  // (the grid's y may be shorter than the query groups: the flagged pass over a large window is launched with y = 1
  // and walks the flags -- a work-group per (row tile, query) that leaves at once is 31 000 dispatches at 64 x 125 000)
  auto one = [&](int q0) {
    const int nvq = (nq - q0) < QT ? (nq - q0) : QT;
    const float acc =
        exact_pairs_wave<QT>(db, queries + (size_t)q0 * dim, dim, RW, row_of, nvq, S);
    if (lane < QT * RW) {
      const int t = lane / RW, r = lane % RW;
      if (t < nvq && row0 + r < n_range) dist[(size_t)(q0 + t) * ld + (size_t)(row0 + r)] = acc;
    }
  };
  if (only_flagged && gridDim.y == 1 && QT == 1) {  // the walk: 256 flags per round trip, usually none set
    for (int base = 0; base < nq; base += 256) {
      const int f = (base + (int)threadIdx.x < nq) ? only_flagged[base + threadIdx.x] : 0;
      if (!__syncthreads_or(f)) continue;
      for (int q0 = base; q0 < nq && q0 < base + 256; ++q0)
        if (only_flagged[q0]) one(q0);  // uniform over the work-group
    }
    return;
  }
  const int q0 = blockIdx.y * QT;
  if (only_flagged && !only_flagged[q0]) return;  // uniform over the work-group
  one(q0);
}

// K1 (exact form, few queries -- the reference's own pattern: kdtree_->query one descriptor at a time,
// loop_detector.cpp:45).  One wave per work-group, QT queries x RW rows = P <= 8 pairs.  The group sums
// of ALL dims are formed first, with no barrier in the loop, so the row reads pipeline (a wave keeps
// (QT + RW) x U 16-byte loads in flight); then P lanes run the reference's sequential chain over them
// from LDS.  The general kernel above interleaves the two per 64 groups and pays a dependent memory
// round trip plus two barriers sixteen times per row: 33 us for 1 x 4541 x 4096, 28 % of the HBM rate.
// Dynamic LDS: P x (Gs + 4) floats, Gs = min(dim / 4, EXS_G) groups per pass.
constexpr int EXS_G = 2048;

template <int QT, int RW>
__global__ __launch_bounds__(64) void dist_exact_small_kernel(const float* __restrict__ db,
                                                              const float* __restrict__ queries,
                                                              float* __restrict__ dist, int dim, size_t first_row,
                                                              int n_range, int nq, size_t ld,
                                                              const int* __restrict__ only_flagged) {
  extern __shared__ __attribute__((aligned(16))) float S_dyn[];
  constexpr int P = QT * RW;
  constexpr int U = (QT + RW) <= 5 ? 4 : 2;
  const int lane = threadIdx.x;
  const int row0 = blockIdx.x * RW;  // relative to first_row
  const int q0 = blockIdx.y * QT;
  if (only_flagged && !only_flagged[q0]) return;
  const int nvq = (nq - q0) < QT ? (nq - q0) : QT;
  const int G = dim >> 2;
  const int Gs = G < EXS_G ? G : EXS_G, pitch = Gs + 4;
  const float* rp[RW];
  const float* qp[QT];
#pragma unroll
  for (int r = 0; r < RW; ++r) {
    const int rr = (row0 + r) < n_range ? (row0 + r) : (n_range - 1);  // clamped: unconditional loads
    rp[r] = db + (first_row + (size_t)rr) * dim;
  }
#pragma unroll
  for (int t = 0; t < QT; ++t) qp[t] = queries + (size_t)(q0 + (t < nvq ? t : 0)) * dim;
  float acc = 0.f;  // lanes 0..P-1: pair (t = lane / RW, r = lane % RW)
  for (int g0 = 0; g0 < G; g0 += Gs) {
    const int gn = (G - g0) < Gs ? (G - g0) : Gs;
    for (int gb = 0; gb < gn; gb += 64 * U) {
      f4u qv[QT][U], dv[RW][U];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int g = gb + u * 64 + lane, gc = g < gn ? g : gn - 1;
#pragma unroll
        for (int t = 0; t < QT; ++t) qv[t][u] = *reinterpret_cast<const f4u*>(qp[t] + 4 * (size_t)(g0 + gc));
#pragma unroll
        for (int r = 0; r < RW; ++r) dv[r][u] = *reinterpret_cast<const f4u*>(rp[r] + 4 * (size_t)(g0 + gc));
      }
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int g = gb + u * 64 + lane;
        if (g < gn) {
#pragma unroll
          for (int t = 0; t < QT; ++t)
#pragma unroll
            for (int r = 0; r < RW; ++r) {
              const float e0 = qv[t][u].x - dv[r][u].x, e1 = qv[t][u].y - dv[r][u].y;
              const float e2 = qv[t][u].z - dv[r][u].z, e3 = qv[t][u].w - dv[r][u].w;
              S_dyn[(t * RW + r) * pitch + g] = e0 * e0 + e1 * e1 + e2 * e2 + e3 * e3;  // left-to-right, no FMA
            }
        }
      }
    }
    __syncthreads();
    if (lane < P) {
      const float* sp = S_dyn + lane * pitch;
      int i = 0;
      for (; i + 4 <= gn; i += 4) {
        const f32x4 v = *reinterpret_cast<const f32x4*>(sp + i);
        acc += v.x;
        acc += v.y;
        acc += v.z;
        acc += v.w;
      }
      for (; i < gn; ++i) acc += sp[i];
    }
    __syncthreads();
  }
  if (lane < P) {
    const int t = lane / RW, r = lane % RW;
    if (t < nvq && row0 + r < n_range) {
      for (int d = 4 * G; d < dim; ++d) {  // scalar tail (nanoflann.hpp:480-485)
        const float e = qp[t][d] - rp[r][d];
        acc += e * e;
      }
      dist[(size_t)(q0 + t) * ld + (size_t)(row0 + r)] = acc;
    }
  }
}

// ---------------------------------------------------------------------------------------------
// K1 (MFMA form): partial dot products P[split][q][j - first] = sum_{k in split} Q[q][k] D[j][k]
// with v_mfma_f32_16x16x4_f32 (exact fp32 fma chain).  Work-group = 4 waves;
// WQ waves along the queries (16 each), 4/WQ along the rows, NT 16-row tiles per wave.
// LDS image per K-step of 32: [kq = k/4][row][4 floats], rows 0..BQ-1 queries, BQ.. db rows,
// one 16-B slot of padding per kq plane (conflict-free ds_write_b128, <=2-way ds_read_b128).
template <int WQ, int NT, int KQ /* K-step = 4*KQ floats: 8 -> 32, 16 -> 64 */>
__global__ __launch_bounds__(256) void dist_mfma_kernel(const float* __restrict__ db,
                                                        const float* __restrict__ queries,
                                                        float* __restrict__ P, int dim,
                                                        size_t first_row, int n_range, int nq,
                                                        int k_per_split, size_t ldP,
                                                        size_t strideP) {
  constexpr int WN = 4 / WQ;
  constexpr int BQ = 16 * WQ;
  constexpr int BN = 16 * NT * WN;
  constexpr int ROWS = BQ + BN;
  constexpr int PLANE = ROWS + 1;  // float4 slots per kq plane
  constexpr int BK = 4 * KQ;
  constexpr int NL = (ROWS * KQ + 255) / 256;
  __shared__ f32x4 lds[2 * KQ * PLANE];

  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int wq = w % WQ, wn = w / WQ;
  const int n0 = blockIdx.x * BN;
  const int q0 = blockIdx.y * BQ;
  const int kbeg = blockIdx.z * k_per_split;
  const int kend = (kbeg + k_per_split) < dim ? (kbeg + k_per_split) : dim;

  // per-thread load slots: slot -> (row, kq); KQ consecutive threads read one 16*KQ-byte row segment
  const float* src[NL];  // never null: rows outside the tile read row 0 and are masked to zero
  int dst[NL], kq4[NL];
  bool ok[NL], rowv[NL];
#pragma unroll
  for (int i = 0; i < NL; ++i) {
    const int slot = tid + i * 256;
    const int row = slot / KQ, kq = slot % KQ;
    ok[i] = slot < ROWS * KQ;
    dst[i] = kq * PLANE + row;
    kq4[i] = kq * 4;
    src[i] = db + first_row * dim;
    rowv[i] = false;
    if (ok[i]) {
      if (row < BQ) {
        const int qq = q0 + row;
        if (qq < nq) {
          src[i] = queries + (size_t)qq * dim;
          rowv[i] = true;
        }
      } else {
        const int jj = n0 + (row - BQ);
        if (jj < n_range) {
          src[i] = db + (first_row + (size_t)jj) * dim;
          rowv[i] = true;
        }
      }
    }
  }
  // Two register sets: the global loads of K-steps s+1 and s+2 are in flight while step s is
  // multiplied, so a load has two MFMA phases to land (one phase was measured to leave the kernel
  // bound by bytes in flight: 2.0 TB/s at one wave per SIMD).
  f32x4 preA[NL], preB[NL];
  // Unconditional loads (addresses clamped into the row, result masked): straight-line code lets
  // the compiler wait with a COUNTED vmcnt instead of draining both register sets.
  auto gload = [&](f32x4* pre, int k) {
#pragma unroll
    for (int i = 0; i < NL; ++i) {
      const int kk = k + kq4[i];
      const int kc = kk < dim - 4 ? kk : dim - 4;
      const f4u v = *reinterpret_cast<const f4u*>(src[i] + kc);
      const bool use = rowv[i] && kk < kend;
      pre[i] = use ? f32x4{v.x, v.y, v.z, v.w} : f32x4{0.f, 0.f, 0.f, 0.f};
    }
  };
  auto lstore = [&](const f32x4* pre, int buf) {
#pragma unroll
    for (int i = 0; i < NL; ++i)
      if (ok[i]) lds[buf * KQ * PLANE + dst[i]] = pre[i];
  };

  // acc: chain of at most 64 fused multiply-adds; tot: sum of those partial chains.  Keeping the
  // chains short is what makes the rounding bound of the coarse distance tight.
  f32x4 acc[NT], tot[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t) {
    acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
    tot[t] = f32x4{0.f, 0.f, 0.f, 0.f};
  }
  const int a_row = wq * 16 + (lane & 15);
  const int b_row0 = BQ + wn * NT * 16 + (lane & 15);
  auto compute = [&](int buf, int k) {
    const f32x4* L = lds + buf * KQ * PLANE;
#pragma unroll
    for (int sub = 0; sub < KQ / 4; ++sub) {
      const int kq = sub * 4 + (lane >> 4);
      const f32x4 a = L[kq * PLANE + a_row];
      f32x4 b[NT];
#pragma unroll
      for (int t = 0; t < NT; ++t) b[t] = L[kq * PLANE + b_row0 + t * 16];
      // k-slot outer, tile inner: consecutive MFMAs hit different accumulators (the 16x16x4 form
      // has a 40-cycle dependent latency against a 32-cycle issue interval)
#pragma unroll
      for (int t = 0; t < NT; ++t) acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.x, b[t].x, acc[t], 0, 0, 0);
#pragma unroll
      for (int t = 0; t < NT; ++t) acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.y, b[t].y, acc[t], 0, 0, 0);
#pragma unroll
      for (int t = 0; t < NT; ++t) acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.z, b[t].z, acc[t], 0, 0, 0);
#pragma unroll
      for (int t = 0; t < NT; ++t) acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.w, b[t].w, acc[t], 0, 0, 0);
    }
    if (BK >= 64 || ((k - kbeg) & 32) || (k + BK) >= kend) {  // every 64 k and at the end: flush
#pragma unroll
      for (int t = 0; t < NT; ++t) {
        tot[t] += acc[t];
        acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
      }
    }
  };

  gload(preA, kbeg);
  lstore(preA, 0);
  gload(preA, kbeg + BK);      // step 1 (zeros beyond kend)
  gload(preB, kbeg + 2 * BK);  // step 2
  __syncthreads();
  for (int k = kbeg; k < kend; k += 2 * BK) {
    // even step: data in buffer 0; preA holds step+1, preB step+2
    compute(0, k);
    if (k + BK < kend) {
      lstore(preA, 1);
      gload(preA, k + 3 * BK);
      __syncthreads();
      compute(1, k + BK);
      if (k + 2 * BK < kend) {
        lstore(preB, 0);
        gload(preB, k + 4 * BK);
        __syncthreads();
      }
    }
  }
  // C/D map of the 16x16 forms: col = lane & 15, row = 4 * (lane >> 4) + reg
  float* Pz = P + (size_t)blockIdx.z * strideP;
#pragma unroll
  for (int t = 0; t < NT; ++t) {
    const int j = n0 + (wn * NT + t) * 16 + (lane & 15);
    if (j < n_range) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int qq = q0 + wq * 16 + (lane >> 4) * 4 + r;
        Pz[(size_t)qq * ldP + j] = tot[t][r];  // rows q >= nq land in the padded part of P
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------
// K1 (MFMA form, 32 x 32 tiles; round 4): the same partial dots with v_mfma_f32_32x32x2_f32.  A wave owns 32 queries x
// (32 * NT) rows: one ds_read_b128 per operand row feeds FOUR MFMAs of 64 cycles each (the 16x16x4 form: four of 32),
// so a flop costs half the LDS operand reads, and the 16 accumulator registers of a tile give the 64-cycle dependent
// latency of this form nothing to wait for once two tiles (or two waves per SIMD) interleave.  Work-group = 4 waves,
// 2 along the queries (64) x 2 along the rows: BN = 64 * NT.  LDS image as above: [k / 4][row][4 floats], the two
// half-waves read planes kq and kq + 1 (K = 2 per MFMA: lane l holds row l % 32, k slot l / 32).
// Rounding: every accumulator element is still one k-ordered fmaf chain flushed every 64 k -- the bound of the coarse
// distance (knn.hip) does not change.
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int NT, int KQ /* K-step = 4*KQ floats */>
__global__ __launch_bounds__(256) void dist_mfma32_kernel(const float* __restrict__ db,
                                                          const float* __restrict__ queries,
                                                          float* __restrict__ P, int dim,
                                                          size_t first_row, int n_range, int nq,
                                                          int k_per_split, size_t ldP,
                                                          size_t strideP) {
  constexpr int BQ = 64;
  constexpr int BN = 64 * NT;
  constexpr int ROWS = BQ + BN;
  constexpr int PLANE = ROWS + 1;  // float4 slots per kq plane
  constexpr int BK = 4 * KQ;
  constexpr int NL = (ROWS * KQ + 255) / 256;
  __shared__ f32x4 lds[2 * KQ * PLANE];

  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int wq = w & 1, wn = w >> 1;
  const int n0 = blockIdx.x * BN;
  const int q0 = blockIdx.y * BQ;
  const int kbeg = blockIdx.z * k_per_split;
  const int kend = (kbeg + k_per_split) < dim ? (kbeg + k_per_split) : dim;

  const float* src[NL];  // never null: rows outside the tile read row 0 and are masked to zero
  int dst[NL], kq4[NL];
  bool ok[NL], rowv[NL];
#pragma unroll
  for (int i = 0; i < NL; ++i) {
    const int slot = tid + i * 256;
    const int row = slot / KQ, kq = slot % KQ;
    ok[i] = slot < ROWS * KQ;
    dst[i] = kq * PLANE + row;
    kq4[i] = kq * 4;
    src[i] = db + first_row * dim;
    rowv[i] = false;
    if (ok[i]) {
      if (row < BQ) {
        const int qq = q0 + row;
        if (qq < nq) {
          src[i] = queries + (size_t)qq * dim;
          rowv[i] = true;
        }
      } else {
        const int jj = n0 + (row - BQ);
        if (jj < n_range) {
          src[i] = db + (first_row + (size_t)jj) * dim;
          rowv[i] = true;
        }
      }
    }
  }
  f32x4 preA[NL], preB[NL];
  auto gload = [&](f32x4* pre, int k) {
#pragma unroll
    for (int i = 0; i < NL; ++i) {
      const int kk = k + kq4[i];
      const int kc = kk < dim - 4 ? kk : dim - 4;
      const f4u v = *reinterpret_cast<const f4u*>(src[i] + kc);
      const bool use = rowv[i] && kk < kend;
      pre[i] = use ? f32x4{v.x, v.y, v.z, v.w} : f32x4{0.f, 0.f, 0.f, 0.f};
    }
  };
  auto lstore = [&](const f32x4* pre, int buf) {
#pragma unroll
    for (int i = 0; i < NL; ++i)
      if (ok[i]) lds[buf * KQ * PLANE + dst[i]] = pre[i];
  };

  f32x16 acc[NT], tot[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      acc[t][r] = 0.f;
      tot[t][r] = 0.f;
    }
  const int a_row = wq * 32 + (lane & 31);
  const int b_row0 = BQ + wn * NT * 32 + (lane & 31);
  auto compute = [&](int buf, int k) {
    const f32x4* L = lds + buf * KQ * PLANE;
#pragma unroll
    for (int sub = 0; sub < KQ / 2; ++sub) {
      const int kq = sub * 2 + (lane >> 5);
      const f32x4 a = L[kq * PLANE + a_row];
      f32x4 b[NT];
#pragma unroll
      for (int t = 0; t < NT; ++t) b[t] = L[kq * PLANE + b_row0 + t * 32];
#pragma unroll
      for (int t = 0; t < NT; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, b[t].x, acc[t], 0, 0, 0);
#pragma unroll
      for (int t = 0; t < NT; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, b[t].y, acc[t], 0, 0, 0);
#pragma unroll
      for (int t = 0; t < NT; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, b[t].z, acc[t], 0, 0, 0);
#pragma unroll
      for (int t = 0; t < NT; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, b[t].w, acc[t], 0, 0, 0);
    }
    if (BK >= 64 || ((k - kbeg) & 32) || (k + BK) >= kend) {  // every 64 k and at the end: flush
#pragma unroll
      for (int t = 0; t < NT; ++t) {
        tot[t] += acc[t];
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
      }
    }
  };

  gload(preA, kbeg);
  lstore(preA, 0);
  gload(preA, kbeg + BK);
  gload(preB, kbeg + 2 * BK);
  __syncthreads();
  for (int k = kbeg; k < kend; k += 2 * BK) {
    compute(0, k);
    if (k + BK < kend) {
      lstore(preA, 1);
      gload(preA, k + 3 * BK);
      __syncthreads();
      compute(1, k + BK);
      if (k + 2 * BK < kend) {
        lstore(preB, 0);
        gload(preB, k + 4 * BK);
        __syncthreads();
      }
    }
  }
  // C/D map of the 32x32 forms: col = lane & 31, row = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5)
  float* Pz = P + (size_t)blockIdx.z * strideP;
#pragma unroll
  for (int t = 0; t < NT; ++t) {
    const int j = n0 + (wn * NT + t) * 32 + (lane & 31);
    if (j < n_range) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int qq = q0 + wq * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
        Pz[(size_t)qq * ldP + j] = tot[t][r];  // rows q >= nq land in the padded part of P
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------
// K1 (split-bf16 form; round 4): the partial dots on the bf16 matrix cores (16 x the fp32 MFMA rate), so that the
// coarse pass is bound by the stream of the rows from HBM and not by v_mfma_f32_32x32x2_f32 (64 x 125 000 x 4096:
// 417 us of fp32 MFMA at its peak against 256 us of HBM).  Every fp32 operand x is cut in two bf16 values on its way
// into LDS, h = bf16(x) and m = bf16(x - h) (x - h is exact in fp32; both conversions round to nearest), and a
// product is taken as  qh dh + qh dm + qm dh  -- three v_mfma_f32_32x32x16_bf16 (32 cycles each for K = 16) in place of
// eight v_mfma_f32_32x32x2_f32 (64 cycles each).  What is dropped, qm dm + (q - qh - qm) d + (qh + qm)(d - dh - dm), is
// at most 3.03 * 2^-16 |q_i| |d_i| per term (|x - h| <= 2^-8 |x|, |x - h - m| <= 2^-16 |x|): the coarse distance stays a
// PROVEN approximation -- the bound is in knn.hip, the exact re-rank and the completeness proof are unchanged, and so
// is every bit of the result.  Products of two bf16 values are exact in fp32; the accumulation inside an MFMA is
// priced as K sequential truncating adds (two units in the last place each), flushed every 64 k as in the fp32 forms.
// The queries are split once per search (split_queries_kernel: [q][k / 8][8 bf16 h | 8 bf16 m], the fp32 row's own
// addressing); the rows are split by the work-group that streams them (3 VALU operations per element).
// Work-group = 4 waves, 2 along the queries (64) x 2 along the rows (BN = 64 NT); LDS image [h | m][k / 8][row][16 B]:
// one ds_read_b128 is a lane's whole K = 16 fragment half (lane l: row l % 32, k = 8 (l / 32) + j).
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ uint32_t cvt_pk_bf16(float lo, float hi) {  // round to nearest even; lo in bits 0..15
  uint32_t r;
  asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(r) : "v"(lo), "v"(hi));
  return r;
}
__device__ __forceinline__ void bf16_split2(float x0, float x1, uint32_t& h, uint32_t& m) {
  h = cvt_pk_bf16(x0, x1);
  const float r0 = x0 - __uint_as_float(h << 16), r1 = x1 - __uint_as_float(h & 0xffff0000u);  // exact
  m = cvt_pk_bf16(r0, r1);
}
__device__ __forceinline__ void bf16_split8(const f32x4& a, const f32x4& b, u32x4& h, u32x4& m) {
  uint32_t h0, h1, h2, h3, m0, m1, m2, m3;
  bf16_split2(a.x, a.y, h0, m0);
  bf16_split2(a.z, a.w, h1, m1);
  bf16_split2(b.x, b.y, h2, m2);
  bf16_split2(b.z, b.w, h3, m3);
  h = u32x4{h0, h1, h2, h3};
  m = u32x4{m0, m1, m2, m3};
}

// out: [nq][dim / 8][8 floats' worth: 16 B of h, 16 B of m]; dim % 8 == 0
__global__ __launch_bounds__(256) void split_queries_kernel(const float* __restrict__ q, size_t n8 /* nq * dim / 8 */,
                                                            float* __restrict__ out) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n8) return;
  const f4u a = *reinterpret_cast<const f4u*>(q + i * 8), b = *reinterpret_cast<const f4u*>(q + i * 8 + 4);
  u32x4 h, m;
  bf16_split8(f32x4{a.x, a.y, a.z, a.w}, f32x4{b.x, b.y, b.z, b.w}, h, m);
  u32x4* o = reinterpret_cast<u32x4*>(out + i * 8);
  o[0] = h;
  o[1] = m;
}

// QRAW: `qsplit` holds the fp32 queries themselves and the work-group splits them as it does the rows (a launch of few
// work-groups: cheaper than the split kernel's launch ahead of it).
// KO: planes of 8 k per step (4: steps of 32 k, 128 contiguous bytes of a row per step; 8: steps of 64 k, 256 bytes).
// LDS is dynamic: 2 buffers x [h | m] x KO planes x PLANE slots of 16 B.
template <int NT, int KO>
constexpr int b3_lds_bytes() {
  return 2 * 2 * KO * (64 + 64 * NT + 2) * 16;
}
template <int NT, bool QRAW, int KO>
__global__ __launch_bounds__(256) void dist_bf16x3_kernel(const float* __restrict__ db,
                                                          const float* __restrict__ qsplit /* split_queries_kernel */,
                                                          float* __restrict__ P, int dim, size_t first_row, int n_range,
                                                          int nq, int k_per_split, size_t ldP, size_t strideP,
                                                          int phase /* groups of 64 k per row tile, 0: none */) {
  constexpr int BQ = 64;
  constexpr int BN = 64 * NT;
  constexpr int ROWS = BQ + BN;
  // 16-B slots per k / 8 plane, = 2 mod 8: the eight lanes of a ds_write_b128 group (two rows x four planes) fall on
  // eight different slots of the 128-B bank row (PLANE = ROWS + 1: two-way, a third of all LDS cycles by the counters)
  constexpr int PLANE = ROWS + 2;
  constexpr int BK = 8 * KO;
  static_assert(KO == 4 || KO == 8, "steps of 32 or 64 k");
  constexpr int NQ = BQ * KO / 256;  // 8-float chunks per thread and step: of the queries,
  constexpr int ND = BN * KO / 256;  // of the rows
  extern __shared__ u32x4 lds[];     // [buffer][h | m][plane][row]

  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int wq = w & 1, wn = w >> 1;
  const int n0 = blockIdx.x * BN;
  const int q0 = blockIdx.y * BQ;
  const int kbeg = blockIdx.z * k_per_split;
  const int kend = (kbeg + k_per_split) < dim ? (kbeg + k_per_split) : dim;
  // Every work-group starts its walk over k at another group of 64 k and wraps around (rows are 16 KB apart: work-groups
  // that walk in step ask for the same offset of every row at the same time, and the addresses of one moment differ in
  // their upper bits only -- measured with loads alone: 5.5 TB/s in step, 6.4 TB/s out of step).  The groups' sums are
  // added in the rotated order: the same number of additions, the same bound.  (Only when the split is whole groups.)
  const int klen = kend - kbeg;
  const int rot = (phase > 0 && klen % 64 == 0) ? (int)(((unsigned)blockIdx.x * (unsigned)phase) % (unsigned)(klen / 64)) * 64 : 0;

  // slot = (row, plane): thread tid holds plane tid % KO of the tile rows (tid + 256 i) / KO -- queries first
  const int ko8 = (tid % KO) * 8;
  const float* qsrc[NQ];  // never null: rows outside the window (queries past nq) read the first one
  const float* dsrc[ND];
#pragma unroll
  for (int i = 0; i < NQ; ++i) {
    const int qq = q0 + (tid + 256 * i) / KO;
    qsrc[i] = qsplit + (size_t)(qq < nq ? qq : 0) * dim;
  }
#pragma unroll
  for (int i = 0; i < ND; ++i) {
    const int jj = n0 + (tid + 256 * i) / KO;
    dsrc[i] = db + (first_row + (size_t)(jj < n_range ? jj : 0)) * dim;
  }
  struct Pre {
    f32x4 q[NQ][2], d[ND][2];
  };
  // (the loads are left raw -- no select on a loaded value before its stage is stored, or the wait for it lands right
  // behind the load and the two stages in flight are none: rows outside the window compute garbage nobody stores, and
  // the chunks that can lie beyond kend, when the split's length is no multiple of the step, are zeroed at the store)
  auto gload = [&](Pre& pre, int k) {
    int kr = k + rot;  // (steps never straddle the wrap: rot and the steps are multiples of the step)
    kr = kr >= kend ? kr - klen : kr;
    const int kk = kr + ko8;
    const int kc = kk < dim - 8 ? kk : dim - 8;
#pragma unroll
    for (int i = 0; i < NQ; ++i) {
      const f4u a = *reinterpret_cast<const f4u*>(qsrc[i] + kc), b = *reinterpret_cast<const f4u*>(qsrc[i] + kc + 4);
      pre.q[i][0] = f32x4{a.x, a.y, a.z, a.w};
      pre.q[i][1] = f32x4{b.x, b.y, b.z, b.w};
    }
#pragma unroll
    for (int i = 0; i < ND; ++i) {
      const f4u a = *reinterpret_cast<const f4u*>(dsrc[i] + kc), b = *reinterpret_cast<const f4u*>(dsrc[i] + kc + 4);
      pre.d[i][0] = f32x4{a.x, a.y, a.z, a.w};
      pre.d[i][1] = f32x4{b.x, b.y, b.z, b.w};
    }
  };
  auto lstore = [&](const Pre& pre, int buf, int k) {
    u32x4* Lh = lds + buf * 2 * KO * PLANE;
    u32x4* Lm = Lh + KO * PLANE;
    const int pl = (tid % KO) * PLANE;
    const bool tail = k + BK > kend;             // uniform
    const bool out = tail && k + ko8 >= kend;    // (kend - kbeg and dim are multiples of 8: a chunk is in or out whole)
    const u32x4 z = {0u, 0u, 0u, 0u};
#pragma unroll
    for (int i = 0; i < NQ; ++i) {
      const int row = (tid + 256 * i) / KO;
      if constexpr (QRAW) {
        u32x4 h, m;
        bf16_split8(pre.q[i][0], pre.q[i][1], h, m);
        Lh[pl + row] = out ? z : h;
        Lm[pl + row] = out ? z : m;
      } else {
        Lh[pl + row] = out ? z : __builtin_bit_cast(u32x4, pre.q[i][0]);
        Lm[pl + row] = out ? z : __builtin_bit_cast(u32x4, pre.q[i][1]);
      }
    }
#pragma unroll
    for (int i = 0; i < ND; ++i) {
      u32x4 h, m;
      bf16_split8(pre.d[i][0], pre.d[i][1], h, m);
      const int row = BQ + (tid + 256 * i) / KO;
      Lh[pl + row] = out ? z : h;
      Lm[pl + row] = out ? z : m;
    }
  };

  // acc: the chain of one group of 64 k (its first MFMA takes a literal zero as C: nothing to clear), added to tot at the
  // group's end.  The loop below walks two steps per turn, buffer 0 then buffer 1: with steps of 32 k buffer 0 opens
  // a group and buffer 1 ends it; with steps of 64 k every step is a group.
  f32x16 acc[NT], tot[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) tot[t][r] = 0.f;
  const int a_row = wq * 32 + (lane & 31);
  const int b_row0 = BQ + wn * NT * 32 + (lane & 31);
  auto compute = [&](int buf, auto opens) {
    const u32x4* Lh = lds + buf * 2 * KO * PLANE;
    const u32x4* Lm = Lh + KO * PLANE;
    const f32x16 zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int ks = 0; ks < BK / 16; ++ks) {
      const int pl = (ks * 2 + (lane >> 5)) * PLANE;
      const bf16x8 ah = __builtin_bit_cast(bf16x8, Lh[pl + a_row]), am = __builtin_bit_cast(bf16x8, Lm[pl + a_row]);
      bf16x8 bh[NT], bm[NT];
#pragma unroll
      for (int t = 0; t < NT; ++t) {
        bh[t] = __builtin_bit_cast(bf16x8, Lh[pl + b_row0 + t * 32]);
        bm[t] = __builtin_bit_cast(bf16x8, Lm[pl + b_row0 + t * 32]);
      }
#pragma unroll
      for (int t = 0; t < NT; ++t)
        acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bm[t], (decltype(opens)::value && ks == 0) ? zero : acc[t], 0, 0, 0);
#pragma unroll
      for (int t = 0; t < NT; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am, bh[t], acc[t], 0, 0, 0);
#pragma unroll
      for (int t = 0; t < NT; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh[t], acc[t], 0, 0, 0);
    }
  };
  auto flush = [&]() {
#pragma unroll
    for (int t = 0; t < NT; ++t) tot[t] += acc[t];
  };
  constexpr bool STEP_IS_GROUP = BK == 64;

  Pre preA, preB;
  gload(preA, kbeg);
  lstore(preA, 0, kbeg);
  gload(preA, kbeg + BK);
  gload(preB, kbeg + 2 * BK);
  __syncthreads();
  for (int k = kbeg; k < kend; k += 2 * BK) {
    compute(0, std::true_type{});
    if constexpr (STEP_IS_GROUP) flush();
    if (k + BK < kend) {
      lstore(preA, 1, k + BK);
      gload(preA, k + 3 * BK);
      __syncthreads();
      compute(1, std::integral_constant<bool, STEP_IS_GROUP>{});
      if constexpr (STEP_IS_GROUP) flush();
      if (k + 2 * BK < kend) {
        lstore(preB, 0, k + 2 * BK);
        gload(preB, k + 4 * BK);
        __syncthreads();
      }
    }
    if constexpr (!STEP_IS_GROUP) flush();  // the group's 64 k (fewer at the end of the split)
  }
  // C/D map of the 32x32 forms: col = lane & 31, row = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5)
  float* Pz = P + (size_t)blockIdx.z * strideP;
#pragma unroll
  for (int t = 0; t < NT; ++t) {
    const int j = n0 + (wn * NT + t) * 32 + (lane & 31);
    if (j < n_range) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int qq = q0 + wq * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
        Pz[(size_t)qq * ldP + j] = tot[t][r];  // rows q >= nq land in the padded part of P
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------
// Round 6 -- the coarse pass reads a MIRROR of the database: the rows already split into their two bf16 values and laid
// out by TILES of 64 rows,  [tile][k / 8][h | m][row in tile][16 B = 8 bf16],  so that what a work-group needs for a step
// of 32 k -- four planes of its tile -- is ONE contiguous run of 8 KB instead of 64 pieces of 128 B from rows 16 KB apart.
// Why: the row-major kernel streams 5.0 - 5.3 TB/s however it walks (tools/dev_stream_pattern.hip, round 4: 128 rows per
// work-group 5.1 - 5.6 TB/s whatever the piece per row, 8 rows per work-group 6.4 - 6.5): 768 resident work-groups x 128
// row streams are ~100 000 open rows against the stacks' ~10 000 banks, 768 contiguous streams are not.  The mirror
// costs the database's size again in HBM (16 GB at a million rows of 288 GB) and is kept current by every add
// (mirror_rows_kernel); the fp32 rows stay what the exact kernels and the re-rank read.  The split (3 VALU per element per
// SEARCH in the row-major kernel) moves to the add as well.
constexpr int MIR_ROWS = 64;  // rows per tile
__host__ __device__ __forceinline__ size_t mirror_tile_u32x4(int dim) { return (size_t)(dim / 8) * 2 * MIR_ROWS; }  // 16-B slots per tile

// rows [first, first + count) of the fp32 database -> their places in the mirror.  One thread per (row, 8 k): rows fastest
// (a wave writes 64 x 16 B = 1 KB contiguous of h and of m; its reads are 32-B pieces of 64 rows, once per row ever).
__global__ __launch_bounds__(256) void mirror_rows_kernel(const float* __restrict__ db, size_t first, size_t count, int dim,
                                                          u32x4* __restrict__ mirror) {
  const size_t r_in = (size_t)blockIdx.x * 64 + (threadIdx.x & 63);
  const int p = blockIdx.y * 4 + (threadIdx.x >> 6);  // plane: k / 8
  if (r_in >= count || p >= dim / 8) return;
  const size_t row = first + r_in;
  const float* src = db + row * (size_t)dim + (size_t)p * 8;
  const f4u a = *reinterpret_cast<const f4u*>(src), b = *reinterpret_cast<const f4u*>(src + 4);
  u32x4 h, m;
  bf16_split8(f32x4{a.x, a.y, a.z, a.w}, f32x4{b.x, b.y, b.z, b.w}, h, m);
  u32x4* t = mirror + (row / MIR_ROWS) * mirror_tile_u32x4(dim) + (size_t)p * 2 * MIR_ROWS + (row % MIR_ROWS);
  t[0] = h;
  t[MIR_ROWS] = m;
}

// The coarse pass over the mirror: the arithmetic, the LDS image and the MFMA sequence of dist_bf16x3_kernel; only where
// the rows come from differs.  Tiles are aligned to ABSOLUTE row numbers: a window that starts inside a tile computes the
// tile's leading rows too and stores nothing for them (row - first_row < 0).
// BMIN (round 6, one K-split, windows above 16 384 rows): the epilogue also leaves, per query and per BLOCK of 32 rows (a
// wave's 32 x 32 accumulator), the minimum of  |d|^2 - 2 q.d  over the block's rows of the window -- the coarse distance
// without the query's norm, which does not change an order.  The KC smallest coarse distances of the window lie in blocks
// whose minimum is <= the KC-th smallest block minimum (that one is the minimum of KC distinct rows), so the selection that
// follows reads 32 x KC partial dots per query instead of the window's (select_blocks_body): at 64 x 125 000, 1 MB instead
// of 32 MB, one launch of 9 us instead of the slices' 28 - 34.
template <int NT, bool QRAW, bool BMIN = false>
__global__ __launch_bounds__(256) void dist_bf16x3_tiled_kernel(const u32x4* __restrict__ mirror,
                                                                const float* __restrict__ qsplit /* split_queries_kernel, or raw */,
                                                                float* __restrict__ P, int dim, size_t first_row, int n_range,
                                                                int nq, int k_per_split, size_t ldP, size_t strideP,
                                                                const float* __restrict__ dn = nullptr /* BMIN: the rows' norms */,
                                                                float* __restrict__ bmin = nullptr /* BMIN: [n_blocks][nq] */,
                                                                int n_blocks = 0) {
  constexpr int KO = 4;
  constexpr int BQ = 64;
  constexpr int BN = 64 * NT;
  constexpr int ROWS = BQ + BN;
  constexpr int PLANE = ROWS + 2;
  constexpr int BK = 8 * KO;
  constexpr int NQ = BQ * KO / 256;       // 8-float chunks of the queries per thread and step
  constexpr int ND = NT * 2;              // 16-B slots of the rows per thread and step: NT tiles x 8 KB / (256 x 16 B)
  extern __shared__ u32x4 lds[];          // [buffer][h | m][plane][row]

  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int wq = w & 1, wn = w >> 1;
  const size_t tile0 = first_row / MIR_ROWS + (size_t)blockIdx.x * NT;          // the work-group's first tile
  const long long j0 = (long long)(tile0 * MIR_ROWS) - (long long)first_row;    // its first row, relative to the window (may be < 0)
  const int q0 = blockIdx.y * BQ;
  const int kbeg = blockIdx.z * k_per_split;
  const int kend = (kbeg + k_per_split) < dim ? (kbeg + k_per_split) : dim;
  const size_t tile_slots = mirror_tile_u32x4(dim);
  float dn_pre[NT];  // BMIN: the norms of this lane's rows, fetched before the walk over k (their latency is then nobody's)
#pragma unroll
  for (int t = 0; t < NT; ++t) {
    dn_pre[t] = 0.f;
    if constexpr (BMIN) {
      const long long j = j0 + ((tid >> 7) * NT + t) * 32 + (tid & 31);  // (wn = tid >> 7, lane & 31 = tid & 31)
      if (j >= 0 && j < (long long)n_range) dn_pre[t] = dn[first_row + (size_t)j];
    }
  }

  const int ko8 = (tid % KO) * 8;
  const float* qsrc[NQ];
#pragma unroll
  for (int i = 0; i < NQ; ++i) {
    const int qq = q0 + (tid + 256 * i) / KO;
    qsrc[i] = qsplit + (size_t)(qq < nq ? qq : 0) * dim;
  }
  // slot i of the thread: tile i / 2, then 16-B slot (i % 2) * 256 + tid of the step's 8-KB run = [plane][h | m][row]
  struct Pre {
    f32x4 q[NQ][2];
    u32x4 d[ND];
  };
  auto gload = [&](Pre& pre, int k) {
    const int kk = k + ko8;
    const int kc = kk < dim - 8 ? kk : dim - 8;
#pragma unroll
    for (int i = 0; i < NQ; ++i) {
      const f4u a = *reinterpret_cast<const f4u*>(qsrc[i] + kc), b = *reinterpret_cast<const f4u*>(qsrc[i] + kc + 4);
      pre.q[i][0] = f32x4{a.x, a.y, a.z, a.w};
      pre.q[i][1] = f32x4{b.x, b.y, b.z, b.w};
    }
    // (a prefetch past the end reads the last step again -- never stored; a dim below one step: step 0.  The planes of a step
    // that lie beyond dim / 8 -- dim no multiple of 32 -- are read from whatever follows the tile, the next tile or the
    // allocation's padding, and zeroed at the store)
    const int klast = dim - BK > 0 ? dim - BK : 0;
    const int kp = (k < dim ? k : klast) / 8;
#pragma unroll
    for (int i = 0; i < ND; ++i)
      pre.d[i] = mirror[(tile0 + i / 2) * tile_slots + (size_t)kp * 2 * MIR_ROWS + (i % 2) * 256 + tid];
  };
  auto lstore = [&](const Pre& pre, int buf, int k) {
    u32x4* Lh = lds + buf * 2 * KO * PLANE;
    u32x4* Lm = Lh + KO * PLANE;
    const int pl = (tid % KO) * PLANE;
    const bool tail = k + BK > kend;
    const bool out = tail && k + ko8 >= kend;
    const u32x4 z = {0u, 0u, 0u, 0u};
#pragma unroll
    for (int i = 0; i < NQ; ++i) {
      const int row = (tid + 256 * i) / KO;
      if constexpr (QRAW) {
        u32x4 h, m;
        bf16_split8(pre.q[i][0], pre.q[i][1], h, m);
        Lh[pl + row] = out ? z : h;
        Lm[pl + row] = out ? z : m;
      } else {
        Lh[pl + row] = out ? z : __builtin_bit_cast(u32x4, pre.q[i][0]);
        Lm[pl + row] = out ? z : __builtin_bit_cast(u32x4, pre.q[i][1]);
      }
    }
#pragma unroll
    for (int i = 0; i < ND; ++i) {
      const int s = (i % 2) * 256 + tid;           // slot in the run: [plane 0..3][h | m][row 0..63]
      const int p = s >> 7, hm = (s >> 6) & 1, r = s & 63;
      const bool pout = tail && k + p * 8 >= kend;  // (the split's length is a multiple of 8: a plane is in or out whole)
      (hm ? Lm : Lh)[p * PLANE + BQ + (i / 2) * MIR_ROWS + r] = pout ? z : pre.d[i];
    }
  };

  f32x16 acc[NT], tot[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) tot[t][r] = 0.f;
  const int a_row = wq * 32 + (lane & 31);
  const int b_row0 = BQ + wn * NT * 32 + (lane & 31);
  auto compute = [&](int buf, auto opens) {
    const u32x4* Lh = lds + buf * 2 * KO * PLANE;
    const u32x4* Lm = Lh + KO * PLANE;
    const f32x16 zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int ks = 0; ks < BK / 16; ++ks) {
      const int pl = (ks * 2 + (lane >> 5)) * PLANE;
      const bf16x8 ah = __builtin_bit_cast(bf16x8, Lh[pl + a_row]), am = __builtin_bit_cast(bf16x8, Lm[pl + a_row]);
      bf16x8 bh[NT], bm[NT];
#pragma unroll
      for (int t = 0; t < NT; ++t) {
        bh[t] = __builtin_bit_cast(bf16x8, Lh[pl + b_row0 + t * 32]);
        bm[t] = __builtin_bit_cast(bf16x8, Lm[pl + b_row0 + t * 32]);
      }
#pragma unroll
      for (int t = 0; t < NT; ++t)
        acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bm[t], (decltype(opens)::value && ks == 0) ? zero : acc[t], 0, 0, 0);
#pragma unroll
      for (int t = 0; t < NT; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am, bh[t], acc[t], 0, 0, 0);
#pragma unroll
      for (int t = 0; t < NT; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh[t], acc[t], 0, 0, 0);
    }
  };
  auto flush = [&]() {
#pragma unroll
    for (int t = 0; t < NT; ++t) tot[t] += acc[t];
  };

  Pre preA, preB;
  gload(preA, kbeg);
  lstore(preA, 0, kbeg);
  gload(preA, kbeg + BK);
  gload(preB, kbeg + 2 * BK);
  __syncthreads();
  for (int k = kbeg; k < kend; k += 2 * BK) {
    compute(0, std::true_type{});
    if (k + BK < kend) {
      lstore(preA, 1, k + BK);
      gload(preA, k + 3 * BK);
      __syncthreads();
      compute(1, std::false_type{});
      if (k + 2 * BK < kend) {
        lstore(preB, 0, k + 2 * BK);
        gload(preB, k + 4 * BK);
        __syncthreads();
      }
    }
    flush();  // the group's 64 k (fewer at the end of the split)
  }
  // C/D map of the 32x32 forms: col = lane & 31, row = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5)
  float* Pz = P + (size_t)blockIdx.z * strideP;
#pragma unroll
  for (int t = 0; t < NT; ++t) {
    const long long j = j0 + (wn * NT + t) * 32 + (lane & 31);
    if (j >= 0 && j < (long long)n_range) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int qq = q0 + wq * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
        Pz[(size_t)qq * ldP + (size_t)j] = tot[t][r];  // rows q >= nq land in the padded part of P
      }
    }
    if constexpr (BMIN) {
      const bool in = j >= 0 && j < (long long)n_range;
      const float dnj = dn_pre[t];
      const int b = blockIdx.x * (BN / 32) + wn * NT + t;
      // the minimum over the 32 lanes that hold a query's 32 rows (lanes 0-31: one query, 32-63: another), for the 16
      // accumulator registers at once: four DPP steps inside the rows of 16 lanes, then row 0 -> 1 and row 2 -> 3; lanes 31
      // and 63 end with it.  The 16 chains are interleaved step by step, so a register is read by a DPP operand 15
      // instructions after it was written: only the first step needs wait states (one chain at a time with its own s_nops:
      // +9 us on the 378-us kernel at 64 x 125 000).
      float v[16];
#pragma unroll
      for (int r = 0; r < 16; ++r) v[r] = in ? dnj - 2.f * tot[t][r] : __builtin_inff();
#define GLOC_BM_STEP(CTRL)                                                                                              \
  "v_min_f32_dpp %0, %0, %0 " CTRL "\n\tv_min_f32_dpp %1, %1, %1 " CTRL "\n\tv_min_f32_dpp %2, %2, %2 " CTRL "\n\t"      \
  "v_min_f32_dpp %3, %3, %3 " CTRL "\n\tv_min_f32_dpp %4, %4, %4 " CTRL "\n\tv_min_f32_dpp %5, %5, %5 " CTRL "\n\t"      \
  "v_min_f32_dpp %6, %6, %6 " CTRL "\n\tv_min_f32_dpp %7, %7, %7 " CTRL "\n\tv_min_f32_dpp %8, %8, %8 " CTRL "\n\t"      \
  "v_min_f32_dpp %9, %9, %9 " CTRL "\n\tv_min_f32_dpp %10, %10, %10 " CTRL "\n\tv_min_f32_dpp %11, %11, %11 " CTRL "\n\t" \
  "v_min_f32_dpp %12, %12, %12 " CTRL "\n\tv_min_f32_dpp %13, %13, %13 " CTRL "\n\tv_min_f32_dpp %14, %14, %14 " CTRL "\n\t" \
  "v_min_f32_dpp %15, %15, %15 " CTRL "\n\t"
      asm("s_nop 4\n\t"
          GLOC_BM_STEP("quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf")
          GLOC_BM_STEP("quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf")
          GLOC_BM_STEP("row_half_mirror row_mask:0xf bank_mask:0xf")
          GLOC_BM_STEP("row_mirror row_mask:0xf bank_mask:0xf")
          GLOC_BM_STEP("row_bcast:15 row_mask:0xa bank_mask:0xf")
          "s_nop 0"
          : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]), "+v"(v[4]), "+v"(v[5]), "+v"(v[6]), "+v"(v[7]), "+v"(v[8]), "+v"(v[9]),
            "+v"(v[10]), "+v"(v[11]), "+v"(v[12]), "+v"(v[13]), "+v"(v[14]), "+v"(v[15]));
#undef GLOC_BM_STEP
      if ((lane & 31) == 31 && b < n_blocks) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int qq = q0 + wq * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
          if (qq < nq) bmin[(size_t)b * nq + qq] = v[r];  // block-major: a block's 64 queries are 256 contiguous bytes (query-major:
                                                           // 250 000 scattered 4-byte writes a launch, each a partial line)
        }
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------
// K1 (top-k part): per-query top-K of a distance row, LDS-staged, in two kernels.
//
// select_chunk_kernel: grid (chunks, nq), 256 threads, E elements per thread (chunk = 256 E).
//   pass A  each thread's minimum -> the K-th smallest of the 256 minima is an upper bound tau of
//           the chunk's K-th smallest (K distinct elements are <= it);
//   pass B  elements <= tau are appended to an LDS list (at most E*K of them: only the K threads
//           whose minimum is <= tau contribute), which is bitonic-sorted and cut to K.
// select_merge_kernel: one work-group per query sorts the chunks' K-lists (<= 2048 keys) in LDS.
// Keys are (ordered(d) << 32 | row): one integer order = (distance, row).
// MODE 0: dist holds final distances.  MODE 1: dist holds MFMA partial dots; the coarse distance
// (qn + dn[j]) - 2 * sum_splits P is formed here.
constexpr int SEL_LIST = 2048;

__device__ __forceinline__ void bitonic_sort_lds(uint64_t* buf, int n /* pow2 */, int tid,
                                                 int nthreads) {
  for (int size = 2; size <= n; size <<= 1) {
    for (int stride = size >> 1; stride > 0; stride >>= 1) {
      __syncthreads();
      for (int i = tid; i < (n >> 1); i += nthreads) {
        const int lo = ((i / stride) * stride * 2) + (i % stride);
        const int hi = lo + stride;
        const bool up = ((lo & size) == 0);
        const uint64_t a = buf[lo], b = buf[hi];
        if ((a > b) == up) {
          buf[lo] = b;
          buf[hi] = a;
        }
      }
    }
  }
  __syncthreads();
}

template <int MODE>
__device__ __forceinline__ uint64_t select_key(const float* __restrict__ row, size_t strideP,
                                               int n_splits, float qnv,
                                               const float* __restrict__ dn, size_t first_row,
                                               int j) {
  float d;
  if (MODE == 1) {
    float dot = row[j];
    for (int s = 1; s < n_splits; ++s) dot += row[(size_t)s * strideP + j];
    d = (qnv + dn[first_row + (size_t)j]) - 2.f * dot;
  } else {
    d = row[j];
  }
  return make_key(d, (uint32_t)(first_row + (size_t)j));
}

// MODE 1 forms the query's squared norm itself (any summation order: it only feeds the coarse form; each
// thread a chain of dim / 256 products, then a fixed tree) and leaves it in qn[q] for the re-rank kernels.
// only_flagged (MODE 0): run only for queries whose flag is set (device-side fallback).
template <int MODE>
__global__ __launch_bounds__(256) void select_chunk_kernel(
    const float* __restrict__ dist, size_t ld, size_t strideP, int n_splits,
    float* __restrict__ qn, const float* __restrict__ queries, int dim, const float* __restrict__ dn,
    size_t first_row, int n_range, int K, int E, uint64_t* __restrict__ out_keys /* [nq][chunks][K] */,
    const int* __restrict__ only_flagged) {
  __shared__ uint64_t buf[SEL_LIST];
  __shared__ int cnt;
  __shared__ float qred[256];
  const int tid = threadIdx.x;
  const int q = blockIdx.y, chunk = blockIdx.x, nchunks = gridDim.x;
  if (only_flagged && !only_flagged[q]) return;  // uniform over the work-group
  const int j0 = chunk * 256 * E;
  const float* row = dist + (size_t)q * ld;
  float qnv = 0.f;
  if (MODE == 1) {
    const float* qr = queries + (size_t)q * dim;
    float s = 0.f;
    for (int d = tid; d < dim; d += 256) s += qr[d] * qr[d];
    qred[tid] = s;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
      if (tid < o) qred[tid] += qred[tid + o];
      __syncthreads();
    }
    qnv = qred[0];
    if (chunk == 0 && tid == 0) qn[q] = qnv;
  }
  // pass A: thread minima
  uint64_t mn = KEY_SENTINEL;
  for (int e = 0; e < E; ++e) {
    const int j = j0 + e * 256 + tid;
    if (j < n_range) {
      const uint64_t key = select_key<MODE>(row, strideP, n_splits, qnv, dn, first_row, j);
      mn = key < mn ? key : mn;
    }
  }
  buf[tid] = mn;
  if (tid == 0) cnt = 0;
  bitonic_sort_lds(buf, 256, tid, 256);
  const uint64_t tau = (K <= 256) ? buf[K - 1] : KEY_SENTINEL;
  __syncthreads();
  // pass B: everything <= tau (the same keys are recomputed: identical bits)
  for (int e = 0; e < E; ++e) {
    const int j = j0 + e * 256 + tid;
    if (j < n_range) {
      const uint64_t key = select_key<MODE>(row, strideP, n_splits, qnv, dn, first_row, j);
      if (key <= tau && key != KEY_SENTINEL) {
        const int pos = atomicAdd(&cnt, 1);
        if (pos < SEL_LIST) buf[pos] = key;
      }
    }
  }
  __syncthreads();
  const int c = cnt < SEL_LIST ? cnt : SEL_LIST;
  int n2 = 64;
  while (n2 < c) n2 <<= 1;
  for (int i = c + tid; i < n2; i += 256) buf[i] = KEY_SENTINEL;
  bitonic_sort_lds(buf, n2, tid, 256);
  uint64_t* o = out_keys + ((size_t)q * nchunks + chunk) * K;
  for (int i = tid; i < K; i += 256) o[i] = (i < c) ? buf[i] : KEY_SENTINEL;
}

// ---------------------------------------------------------------------------------------------
// K1 (top-k part), one launch: ONE work-group of 1024 threads per query selects the K <= 64 smallest
// keys of up to 16 384 rows -- no chunk lists, no merge launch, no sort through LDS barriers:
//   1  every thread forms its <= 16 keys (registers) and their minimum;
//   2  each wave sorts its 64 minima across the lanes (bitonic, register exchanges);
//   3  a 4-round tournament merges the 16 sorted lists pairwise (the 64 smallest of two sorted lists
//      are min(a[i], b[63 - i]), a bitonic sequence: six exchange steps sort it) -- wave 0 ends with the
//      64 smallest thread minima; tau = the K-th of them is an upper bound of the K-th smallest key
//      (K distinct keys are <= it) and exactly K threads hold a key <= tau;
//   4  keys <= tau (at most 16 K) are appended to an LDS list; 5  one wave sorts them (<= 64 in
//      registers, more -- rare -- through the LDS sort) and writes the first K.
// The round-1 form (256 threads, two LDS bitonic sorts with a barrier per step, five chunk lists and a
// merge launch) took 21.9 + 9.9 us at 64 x 10 000; this one launch replaces both.
constexpr int SELQ_THREADS = 1024;
constexpr int SELQ_EPT = 16;                             // keys per thread
constexpr int SELQ_MAX_ROWS = SELQ_THREADS * SELQ_EPT;   // 16 384

// one compare-exchange step of the bitonic network at lane distance STRIDE inside blocks of SIZE
template <int SIZE, int STRIDE>
__device__ __forceinline__ uint64_t bitonic_step(uint64_t x, int lane) {
  const uint64_t y = xor_lane_u64<STRIDE>(x);
  const bool up = SIZE == 64 || (lane & SIZE) == 0;
  const bool lower = (lane & STRIDE) == 0;
  // the lower lane of an ascending pair keeps the minimum: keep x when (x < y) says what this lane wants
  // (equal keys are sentinels: either will do)
  return ((x < y) == (lower == up)) ? x : y;
}
template <int SIZE>
__device__ __forceinline__ uint64_t bitonic_merge(uint64_t x, int lane) {  // the strides SIZE/2 ... 1
  if constexpr (SIZE >= 64) x = bitonic_step<SIZE, 32>(x, lane);
  if constexpr (SIZE >= 32) x = bitonic_step<SIZE, 16>(x, lane);
  if constexpr (SIZE >= 16) x = bitonic_step<SIZE, 8>(x, lane);
  if constexpr (SIZE >= 8) x = bitonic_step<SIZE, 4>(x, lane);
  if constexpr (SIZE >= 4) x = bitonic_step<SIZE, 2>(x, lane);
  return bitonic_step<SIZE, 1>(x, lane);
}
// ascending bitonic sort of one key per lane over the wave (lane exchanges through DPP / permlane swaps:
// lane_ops.hpp); FULL = false: the input is already bitonic (only the last merge stage runs)
template <bool FULL>
__device__ __forceinline__ uint64_t wave_sort_u64(uint64_t x, int lane) {
  if constexpr (FULL) {
    x = bitonic_merge<2>(x, lane);
    x = bitonic_merge<4>(x, lane);
    x = bitonic_merge<8>(x, lane);
    x = bitonic_merge<16>(x, lane);
    x = bitonic_merge<32>(x, lane);
  }
  return bitonic_merge<64>(x, lane);
}

// Where a kernel that ends a search leaves its result directly (what finalize_kernel would make of its keys).
struct FinalOut {
  uint64_t* idx;   // [nq][k], null: keys only
  float* d2;
  uint64_t offset, stride;
};
__device__ __forceinline__ void final_store(const FinalOut& fo, size_t i, uint64_t key) {
  if (!fo.idx) return;
  if (key == KEY_SENTINEL) {
    fo.idx[i] = ~0ull;
    fo.d2[i] = 3.402823466e+38f;
  } else {
    fo.idx[i] = (uint64_t)(uint32_t)key * fo.stride + fo.offset;
    fo.d2[i] = ord2f((uint32_t)(key >> 32));
  }
}

// The selection proper, for the calling work-group's query: leaves the K smallest keys, sorted, in
// buf[0..K) (padded with the sentinel) and returns after a barrier.  MODE 1 also returns the query's norm.
template <int MODE>
__device__ __forceinline__ float selq_select(const float* __restrict__ row, size_t strideP, int n_splits,
                                             const float* __restrict__ qr, int dim, const float* __restrict__ dn,
                                             size_t first_row, int n_range, int K, uint64_t* buf /* [SEL_LIST] */,
                                             float* qred /* [16] */, uint64_t* tau_s, int* cnt,
                                             unsigned long long* st = nullptr /* dev: 4 stamps */) {
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  // the loads of the keys' operands go out before the norm's reduction (one memory round trip, not two)
  // (split by split with every key slot's load issued before the first use: the first form walked the
  // splits inside the slot loop and waited for ten round trips in turn -- 6.5 us of a 14-us selection)
  // MODE 2 (round 4): `row` holds n_range ready keys -- the slice lists of select_slices_kernel
  float dot[SELQ_EPT], dnv[SELQ_EPT];
  uint64_t key[SELQ_EPT];
#pragma unroll
  for (int e = 0; e < SELQ_EPT; ++e) {
    const int j = e * SELQ_THREADS + tid;
    if (MODE == 2) {
      key[e] = (j < n_range) ? reinterpret_cast<const uint64_t*>(row)[j] : KEY_SENTINEL;
    } else {
      dot[e] = (j < n_range) ? row[j] : 0.f;
      dnv[e] = (MODE == 1 && j < n_range) ? dn[first_row + (size_t)j] : 0.f;
    }
  }
  if (MODE == 1) {
    if (n_splits > 1) {  // the usual second split rides in the first batch of loads
      float part[SELQ_EPT];
#pragma unroll
      for (int e = 0; e < SELQ_EPT; ++e) {
        const int j = e * SELQ_THREADS + tid;
        part[e] = (j < n_range) ? row[strideP + j] : 0.f;
      }
#pragma unroll
      for (int e = 0; e < SELQ_EPT; ++e) dot[e] += part[e];
    }
    for (int sp = 2; sp < n_splits; ++sp) {
      float part[SELQ_EPT];
#pragma unroll
      for (int e = 0; e < SELQ_EPT; ++e) {
        const int j = e * SELQ_THREADS + tid;
        part[e] = (j < n_range) ? row[(size_t)sp * strideP + j] : 0.f;
      }
#pragma unroll
      for (int e = 0; e < SELQ_EPT; ++e) dot[e] += part[e];
    }
  }
  float qnv = 0.f;
  if (MODE == 1 || (MODE == 2 && qr)) {  // (the same sum in every kernel that forms it: the same bits)
    float sq = 0.f;
    for (int d = tid; d < dim; d += SELQ_THREADS) sq += qr[d] * qr[d];
    for (int o = 32; o > 0; o >>= 1) sq += __shfl_xor(sq, o);
    if (lane == 0) qred[w] = sq;
    __syncthreads();
    for (int i = 0; i < SELQ_THREADS / 64; ++i) qnv += qred[i];  // a fixed order, the same in every thread
  }
  // 1: keys and the thread minimum
  uint64_t mn = KEY_SENTINEL;
#pragma unroll
  for (int e = 0; e < SELQ_EPT; ++e) {
    const int j = e * SELQ_THREADS + tid;
    if (MODE != 2) {
      key[e] = KEY_SENTINEL;
      if (j < n_range) {
        const float d = (MODE == 1) ? (qnv + dnv[e]) - 2.f * dot[e] : dot[e];  // as select_key<MODE>
        key[e] = make_key(d, (uint32_t)(first_row + (size_t)j));
      }
    }
    mn = key[e] < mn ? key[e] : mn;
  }
  if (st && tid == 0) st[0] = __builtin_amdgcn_s_memtime();
  // 2: the wave's minima, sorted over its lanes
  uint64_t x = wave_sort_u64<true>(mn, lane);
  if (st && tid == 0) st[1] = __builtin_amdgcn_s_memtime();
  // 3: tournament
  if (tid == 0) *cnt = 0;
#pragma unroll
  for (int r = 1; r < SELQ_THREADS / 64; r <<= 1) {
    if ((w & (2 * r - 1)) == r) buf[w * 64 + lane] = x;  // the partner list
    __syncthreads();
    if ((w & (2 * r - 1)) == 0) {
      const uint64_t y = buf[(w + r) * 64 + (63 - lane)];
      x = wave_sort_u64<false>(x < y ? x : y, lane);
    }
    __syncthreads();
  }
  if (w == 0 && lane == K - 1) *tau_s = x;
  __syncthreads();
  if (st && tid == 0) st[2] = __builtin_amdgcn_s_memtime();
  const uint64_t tau = *tau_s;
  // 4: everything <= tau (one LDS atomic per wave and key slot)
#pragma unroll
  for (int e = 0; e < SELQ_EPT; ++e) {
    const bool take = key[e] <= tau && key[e] != KEY_SENTINEL;
    const unsigned long long m = __builtin_amdgcn_ballot_w64(take);
    if (m) {
      int base = 0;
      if (lane == 0) base = atomicAdd(cnt, (int)__popcll(m));
      base = __builtin_amdgcn_readfirstlane(base);
      const int pos = base + (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
      if (take && pos < SEL_LIST) buf[pos] = key[e];
    }
  }
  __syncthreads();
  if (st && tid == 0) st[3] = __builtin_amdgcn_s_memtime();
  const int c = *cnt < SEL_LIST ? *cnt : SEL_LIST;
  if (c <= 64) {  // 5: the usual case -- one wave, in registers
    uint64_t v = KEY_SENTINEL;
    if (w == 0) v = wave_sort_u64<true>(lane < c ? buf[lane] : KEY_SENTINEL, lane);
    __syncthreads();
    if (w == 0) buf[lane] = v;
  } else {
    int n2 = 128;
    while (n2 < c) n2 <<= 1;
    for (int i = c + tid; i < n2; i += SELQ_THREADS) buf[i] = KEY_SENTINEL;
    bitonic_sort_lds(buf, n2, tid, SELQ_THREADS);
  }
  __syncthreads();
  return qnv;
}

template <int MODE>
__global__ __launch_bounds__(SELQ_THREADS) void select_query_kernel(
    const float* __restrict__ dist, size_t ld, size_t strideP, int n_splits,
    float* __restrict__ qn, const float* __restrict__ queries, int dim, const float* __restrict__ dn,
    size_t first_row, int n_range, int K, uint64_t* __restrict__ out_keys /* [nq][K] */,
    const int* __restrict__ only_flagged, FinalOut fo) {
  __shared__ uint64_t buf[SEL_LIST];     // tournament lists (16 x 64), then the candidate list
  __shared__ float qred[SELQ_THREADS / 64];
  __shared__ uint64_t tau_s;
  __shared__ int cnt;
  const int tid = threadIdx.x;
  const int q = blockIdx.x;
  if (only_flagged && !only_flagged[q]) return;  // uniform over the work-group
  const float qnv = selq_select<MODE>(dist + (size_t)q * ld, strideP, n_splits,
                                      MODE == 2 ? nullptr : queries + (size_t)q * dim, dim, dn, first_row, n_range, K,
                                      buf, qred, &tau_s, &cnt);
  if (MODE == 1 && tid == 0) qn[q] = qnv;
  if (tid < K) {
    out_keys[(size_t)q * K + tid] = buf[tid];
    final_store(fo, (size_t)q * K + tid, buf[tid]);
  }
}

// Windows above 16 384 rows (round 4; they went through select_chunk_kernel's 256-thread LDS sorts + a merge before:
// 108 us of a 800-us search over a 125 000-row shard): the window is cut in S slices of L <= 16 384 rows, one
// work-group per (slice, query) leaves the slice's K smallest keys, and the selection over the S x K keys (MODE 2
// above: select_query_kernel<2>, or select_rerank_kernel<true>) ends the search.  The K smallest keys of the window
// are among the slices' K smallest, and keys are distinct (the row is part of the key): the same result.
// WALK: the flagged pass -- y = 1 and every work-group walks the flags (as dist_exact_kernel); a template flag because the
// loop around the selection costs the MODE 1 instance 28 spilled registers.
template <int MODE, bool WALK>
__global__ __launch_bounds__(SELQ_THREADS) void select_slices_kernel(
    const float* __restrict__ dist, size_t ld, size_t strideP, int n_splits, float* __restrict__ qn,
    const float* __restrict__ queries, int dim, const float* __restrict__ dn, size_t first_row, int n_range, int L, int K,
    uint64_t* __restrict__ lists /* [nq][S][K] */, const int* __restrict__ only_flagged, int nq) {
  __shared__ uint64_t buf[SEL_LIST];
  __shared__ float qred[SELQ_THREADS / 64];
  __shared__ uint64_t tau_s;
  __shared__ int cnt;
  const int tid = threadIdx.x;
  const int sl = blockIdx.x, S = gridDim.x;
  const int j0 = sl * L;
  const int n = n_range - j0 < L ? n_range - j0 : L;  // >= 1: the host drops empty slices
  auto one = [&](int q) {
    const float qnv = selq_select<MODE>(dist + (size_t)q * ld + j0, strideP, n_splits, queries + (size_t)q * dim, dim,
                                        dn, first_row + (size_t)j0, n, K, buf, qred, &tau_s, &cnt);
    if (MODE == 1 && sl == 0 && tid == 0) qn[q] = qnv;
    if (tid < K) lists[((size_t)q * S + sl) * K + tid] = buf[tid];
  };
  if constexpr (WALK) {  // SELQ_THREADS flags per round trip, usually none set
    for (int base = 0; base < nq; base += SELQ_THREADS) {
      const int f = (base + tid < nq) ? only_flagged[base + tid] : 0;
      if (!__syncthreads_or(f)) continue;
      for (int q = base; q < nq && q < base + SELQ_THREADS; ++q) {
        if (!only_flagged[q]) continue;  // uniform over the work-group
        one(q);
        __syncthreads();  // buf is the next query's
      }
    }
  } else {
    if (only_flagged && !only_flagged[blockIdx.y]) return;  // uniform over the work-group
    one(blockIdx.y);
  }
}

// Round 6 -- the exact redo of the queries a LARGE window's proof has flagged, in ONE launch (three before: exact distances,
// slices, the lists' selection -- always launched, each leaving at once when no flag is set, 5.3 us apiece = 16 us of a
// 440-us search over a 125 000-row shard).  Work-group s owns rows [s L, s L + L) of the window.  Every work-group walks
// the flags (one round trip per 1 024 of them; none set: it leaves); for a flagged query it
//   A  computes the reference-order distances of ITS rows (16 waves x RW rows at a time: exact_pairs_wave, as
//      dist_exact_kernel) into the query's row of the workspace,
//   B  selects their K smallest keys (selq_select<0>) into the query's list s,
//   C  draws the query's ticket: the LAST of the S work-groups selects the K smallest of the S x K keys (the K smallest of
//      the window are among the slices' K smallest; keys are distinct) and writes the result over the unproven one.
// No work-group waits for another (the ticket decides who goes on), so nothing can hang; a flagged query costs ~0.5 ms
// (62 work-groups stream 2 GB between them) instead of three full-width launches -- it is the rare path (no query of any
// bench or golden configuration is flagged), as the one-work-group redo of a window below 16 384 rows has been.
constexpr int REDO_RW = 8;  // rows per wave and step
__global__ __launch_bounds__(SELQ_THREADS) void flagged_redo_kernel(
    const float* __restrict__ db, const float* __restrict__ queries, int dim, size_t first_row, int n_range, int L, int K,
    float* __restrict__ dist, size_t ld, uint64_t* __restrict__ lists /* [nq][S][K] */, unsigned int* __restrict__ tickets /* [nq]: 0 between searches */,
    const int* __restrict__ flags, int nq, uint64_t* __restrict__ out_keys /* [nq][K] */, FinalOut fo) {
  __shared__ uint64_t buf[SEL_LIST];
  __shared__ float qred[SELQ_THREADS / 64];
  __shared__ uint64_t tau_s;
  __shared__ int cnt;
  __shared__ unsigned int arrived;
  __shared__ __attribute__((aligned(16))) float S_all[(SELQ_THREADS / 64) * REDO_RW * S_PITCH];
  const int tid = threadIdx.x, w = tid >> 6, lane = tid & 63;
  const int sl = blockIdx.x, S = gridDim.x;
  const int j0 = sl * L;
  const int n = n_range - j0 < L ? n_range - j0 : L;  // >= 1: the host launches no empty slice
  float* Sw = S_all + w * REDO_RW * S_PITCH;
  auto one = [&](int q) {
    // A: this work-group's rows, REDO_RW per wave and step (the loop is uniform: exact_pairs_wave holds barriers)
    const float* qp = queries + (size_t)q * dim;
    float* drow = dist + (size_t)q * ld;
    for (int r0 = 0; r0 < n; r0 += (SELQ_THREADS / 64) * REDO_RW) {
      const int row0 = j0 + r0 + w * REDO_RW;
      auto row_of = [&](int r) -> long long {
        const int rr = row0 + r;
        return rr < j0 + n ? (long long)(first_row + (size_t)rr) : -1;
      };
      const float acc = exact_pairs_wave<1>(db, qp, dim, REDO_RW, row_of, 1, Sw);
      if (lane < REDO_RW && row0 + lane < j0 + n) drow[row0 + lane] = acc;
    }
    __threadfence_block();
    __syncthreads();
    // B: the slice's K smallest
    selq_select<0>(drow + j0, 0, 1, qp, dim, nullptr, first_row + (size_t)j0, n, K, buf, qred, &tau_s, &cnt);
    if (tid < K) lists[((size_t)q * S + sl) * K + tid] = buf[tid];
    // C: the last work-group of the query ends it
    __threadfence();
    __syncthreads();
    if (tid == 0) arrived = atomicAdd(&tickets[q], 1u);
    __syncthreads();
    if (arrived != (unsigned)(S - 1)) return;  // uniform over the work-group
    __threadfence();
    selq_select<2>(reinterpret_cast<const float*>(lists + (size_t)q * S * K), 0, 1, nullptr, dim, nullptr, 0, S * K, K, buf, qred,
                   &tau_s, &cnt);
    if (tid < K) {
      out_keys[(size_t)q * K + tid] = buf[tid];
      final_store(fo, (size_t)q * K + tid, buf[tid]);
    }
    if (tid == 0) tickets[q] = 0u;
  };
  for (int base = 0; base < nq; base += SELQ_THREADS) {
    const int f = (base + tid < nq) ? flags[base + tid] : 0;
    if (!__syncthreads_or(f)) continue;
    for (int q = base; q < nq && q < base + SELQ_THREADS; ++q) {
      if (!flags[q]) continue;  // uniform over the work-group
      one(q);
      __syncthreads();  // buf is the next query's
    }
  }
}

// Round 6 -- the selection over a LARGE window from the coarse kernel's block minima (dist_bf16x3_tiled_kernel<.., BMIN>):
// one work-group per query finds the KC smallest block minima, takes EVERY block whose minimum is <= the KC-th (ties
// included; more than SELB_MAX_BLOCKS of them: the query is flagged for the exact redo), and writes the coarse keys of
// those blocks' rows -- (|q|^2 + |d|^2) - 2 q.d exactly as selq_select<1> forms them -- as a list of 32 x SELB_MAX_BLOCKS
// ready keys for the selection + re-rank that follows.  Replaces select_slices_kernel<1> there (which read the whole window again).
constexpr int SELB_MAX_BLOCKS = 64;
constexpr int SELB_LIST = 32 * SELB_MAX_BLOCKS;
// (A device function of select_rerank_kernel<true> since the same round: as a kernel of its own it cost a launch floor and a
// round trip of the list through memory -- 14.3 + 27.1 us in two launches against ~36 in one.)  Returns whether more blocks
// tied at the threshold than the list holds (uniform over the work-group); ends with a barrier.
__device__ __forceinline__ bool select_blocks_body(const float* __restrict__ bq /* this query's block minima: bq[b * es] */, size_t es, int NB,
                                                   const float* __restrict__ Pq /* this query's partial dots */,
                                                   const float* __restrict__ qp, int dim, const float* __restrict__ dn,
                                                   size_t first_row, int n_range, int KC, uint64_t* __restrict__ out /* [SELB_LIST] */,
                                                   uint64_t* buf, float* qred, uint64_t* tau_s, int* cnt, int* blk /* [SELB_MAX_BLOCKS] */,
                                                   int* nblk, float* smin /* LDS scratch, >= NB floats */) {
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  // the query's minima are es floats apart in memory ([block][query]: the coarse kernel writes whole lines that way): ONE
  // strided pass into LDS, the selection and the second look at them from there (two strided passes: +4 us)
  for (int b = tid; b < NB; b += SELQ_THREADS) smin[b] = bq[(size_t)b * es];
  __syncthreads();
  // A threshold that is AT LEAST the KC-th smallest block minimum, and nearly always equal to it, without the full
  // selection (its 16-wave tournament: 10 of this stage's 15 us): every thread's minimum over its blocks, each wave's FOUR
  // smallest of those (one register sort), the KC-th smallest of the 64 (one more) -- 64 distinct blocks' values, so their
  // KC-th smallest bounds the KC-th smallest of all from above, and it is the true one unless a wave holds more than four
  // of the KC smallest.  (Each wave's m-th smallest and the MAXIMUM of those was tried first: sound, but ~100 blocks pass.)
  // More blocks at or below it than the list holds (ties, or a loose bound): the exact KC-th smallest decides.
  static_assert(SELQ_THREADS / 64 * 4 == 64, "four per wave: 64 values, one wave sorts them (KC <= 64)");
  float mymin = __builtin_inff();
  for (int b = tid; b < NB; b += SELQ_THREADS) mymin = fminf(mymin, smin[b]);
  {
    const uint64_t x = wave_sort_u64<true>(make_key(mymin, (uint32_t)tid), lane);
    if (lane < 4) buf[w * 4 + lane] = x;
    __syncthreads();
    if (w == 0) {
      const uint64_t y = wave_sort_u64<true>(buf[lane], lane);
      if (lane == KC - 1) *tau_s = y;
    }
    __syncthreads();
  }
  uint32_t tau_ord = (uint32_t)(*tau_s >> 32);
  auto collect = [&]() {
    if (tid == 0) *nblk = 0;
    __syncthreads();
    for (int b = tid; b < NB; b += SELQ_THREADS)
      if (f2ord(smin[b]) <= tau_ord) {
        const int pos = atomicAdd(nblk, 1);
        if (pos < SELB_MAX_BLOCKS) blk[pos] = b;
      }
    __syncthreads();
  };
  collect();
  if (*nblk > SELB_MAX_BLOCKS) {  // (uniform) the exact KC-th smallest block minimum (fewer blocks than KC: the last one)
    __syncthreads();
    selq_select<0>(smin, 0, 1, qp, dim, nullptr, 0, NB, KC, buf, qred, tau_s, cnt);
    const int kk = NB < KC ? NB : KC;
    tau_ord = (uint32_t)(buf[kk - 1] >> 32);
    __syncthreads();
    collect();
  }
  // the query's norm, summed as selq_select sums it (the same bits in every kernel that forms it)
  float sq = 0.f;
  for (int d = tid; d < dim; d += SELQ_THREADS) sq += qp[d] * qp[d];
  for (int o = 32; o > 0; o >>= 1) sq += __shfl_xor(sq, o);
  if (lane == 0) qred[w] = sq;
  __syncthreads();
  float qnv = 0.f;
  for (int i = 0; i < SELQ_THREADS / 64; ++i) qnv += qred[i];
  const int n_found = *nblk;
  const int nb = n_found < SELB_MAX_BLOCKS ? n_found : SELB_MAX_BLOCKS;
  const long long off = (long long)(first_row % MIR_ROWS);
  for (int i = tid; i < SELB_LIST; i += SELQ_THREADS) {
    uint64_t key = KEY_SENTINEL;
    const int bi = i >> 5;
    if (bi < nb) {
      const long long j = (long long)blk[bi] * 32 - off + (i & 31);
      if (j >= 0 && j < (long long)n_range) {
        const float d = (qnv + dn[first_row + (size_t)j]) - 2.f * Pq[j];  // as selq_select<1>
        key = make_key(d, (uint32_t)(first_row + (size_t)j));
      }
    }
    out[i] = key;
  }
  __threadfence_block();  // (the list is read back by other threads of this work-group: selq_select<2>)
  __syncthreads();
  return n_found > SELB_MAX_BLOCKS;
}

// in: [nq][nlists][K]; group g of `per_group` lists -> out [nq][ngroups][K].  grid (ngroups, nq).
__global__ __launch_bounds__(256) void select_merge_kernel(const uint64_t* __restrict__ in_keys,
                                                           int nlists, int per_group, int K,
                                                           uint64_t* __restrict__ out_keys) {
  __shared__ uint64_t buf[SEL_LIST];
  const int tid = threadIdx.x, q = blockIdx.y, g = blockIdx.x, ngroups = gridDim.x;
  const int l0 = g * per_group;
  const int nl = (nlists - l0) < per_group ? (nlists - l0) : per_group;
  const int total = nl * K;  // <= SEL_LIST (host guarantees)
  int n2 = 64;
  while (n2 < total) n2 <<= 1;
  const uint64_t* in = in_keys + ((size_t)q * nlists + l0) * K;
  for (int i = tid; i < n2; i += 256) buf[i] = i < total ? in[i] : KEY_SENTINEL;
  bitonic_sort_lds(buf, n2, tid, 256);
  uint64_t* o = out_keys + ((size_t)q * ngroups + g) * K;
  for (int i = tid; i < K; i += 256) o[i] = (i < total) ? buf[i] : KEY_SENTINEL;
}

// ---------------------------------------------------------------------------------------------
// K1b: exact re-rank of the MFMA path's candidates + completeness check.
//   theta    = coarse_d[k-1] + 2 eps   (eps: rounding bound of coarse vs reference-order distance,
//              eps_rel_d * max(coarse_d[k-1], 0) + eps_rel_n * (qn + dn_max))
//   m        = candidates with coarse_d <= theta (a prefix): those get reference-order distances
//   complete = fewer candidates than KC exist, or coarse_d[KC-1] > theta
// rerank_dist_kernel: grid (ceil(KC/RR), nq), one wave per RR candidates of one query.
// rerank_final_kernel: one wave per query ranks the exact keys and raises flags[q] when the
// candidate set may be incomplete (the host then redoes that query on the exact path).
constexpr int RR = 4;  // candidates per wave in rerank_dist_kernel

__device__ __forceinline__ float rerank_theta(float dk, float qnv, float dn_max, float eps_rel_d,
                                              float eps_rel_n) {
  const float eps = eps_rel_d * fmaxf(dk, 0.f) + eps_rel_n * (qnv + dn_max);
  return dk + 2.f * eps;
}

// Group sums of ALL dims first (no barrier in the loop: the row loads pipeline), then the reference's
// sequential chain over them by RR lanes: the first form (one chunk of 64 groups at a time, a dependent
// global load per chunk) spent 22 us of a 152-us search waiting for sixteen round trips.
constexpr int RR_G = 2048;  // groups held in LDS per pass: dim <= 8192 in one pass

__global__ __launch_bounds__(64) void rerank_dist_kernel(
    const float* __restrict__ db, const float* __restrict__ queries, int dim,
    const uint64_t* __restrict__ cand, int KC, int k, const float* __restrict__ qn,
    const uint32_t* __restrict__ dn_max_bits, float eps_rel_d, float eps_rel_n,
    float* __restrict__ exact /* [nq][KC] */) {
  __shared__ __attribute__((aligned(16))) float S[RR][RR_G];
  const int lane = threadIdx.x;
  const int q = blockIdx.y, p0 = blockIdx.x * RR;
  const uint64_t key = (lane < KC) ? cand[(size_t)q * KC + lane] : KEY_SENTINEL;
  const bool valid = key != KEY_SENTINEL;
  const float dco = ord2f((uint32_t)(key >> 32));
  const int n_valid = __popcll(__ballot(valid));
  const int kk = k < n_valid ? k : n_valid;
  if (kk == 0) return;
  const float theta = rerank_theta(__shfl(dco, kk - 1), qn[q], __uint_as_float(*dn_max_bits),
                                   eps_rel_d, eps_rel_n);
  const int m = __popcll(__ballot(valid && dco <= theta));
  if (p0 >= m) return;  // wave-uniform
  const int rw = (m - p0) < RR ? (m - p0) : RR;
  const float* rowp[RR];
#pragma unroll
  for (int r = 0; r < RR; ++r) {
    const uint32_t row = (uint32_t)__shfl((int)(uint32_t)key, (p0 + (r < rw ? r : 0)) & 63);
    rowp[r] = db + (size_t)row * dim;
  }
  const float* qp = queries + (size_t)q * dim;
  const int G = dim >> 2;
  float acc = 0.f;  // lanes 0..rw-1: the chain of pair (q, row r = lane)
  for (int g0 = 0; g0 < G; g0 += RR_G) {
    const int gn = (G - g0) < RR_G ? (G - g0) : RR_G;
    // four groups per lane and step, all (RR + 1) x 4 loads issued before the first use (clamped, so
    // that they are unconditional): a wave is alone on its work-group and must hide the latency itself
    for (int gb = 0; gb < gn; gb += 256) {
      f4u qv[4], dv[RR][4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int g = gb + u * 64 + lane, gc = g < gn ? g : gn - 1;
        qv[u] = *reinterpret_cast<const f4u*>(qp + 4 * (size_t)(g0 + gc));
#pragma unroll
        for (int r = 0; r < RR; ++r) dv[r][u] = *reinterpret_cast<const f4u*>(rowp[r] + 4 * (size_t)(g0 + gc));
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int g = gb + u * 64 + lane;
        if (g < gn) {
#pragma unroll
          for (int r = 0; r < RR; ++r) {
            const float e0 = qv[u].x - dv[r][u].x, e1 = qv[u].y - dv[r][u].y, e2 = qv[u].z - dv[r][u].z,
                        e3 = qv[u].w - dv[r][u].w;
            S[r][g] = e0 * e0 + e1 * e1 + e2 * e2 + e3 * e3;  // left-to-right, no FMA (as exact_pairs_wave)
          }
        }
      }
    }
    __syncthreads();
    if (lane < rw) {
      const float* sp = S[lane];
      int i = 0;
      for (; i + 4 <= gn; i += 4) {
        const f32x4 v = *reinterpret_cast<const f32x4*>(sp + i);
        acc += v.x;
        acc += v.y;
        acc += v.z;
        acc += v.w;
      }
      for (; i < gn; ++i) acc += sp[i];
    }
    __syncthreads();
  }
  // (no scalar tail: the MFMA path, the only caller, requires dim % 4 == 0)
  if (lane < rw) exact[(size_t)q * KC + p0 + lane] = acc;
}

__global__ __launch_bounds__(64) void rerank_final_kernel(
    const uint64_t* __restrict__ cand, const float* __restrict__ exact, int KC, int k, int n_range,
    const float* __restrict__ qn, const uint32_t* __restrict__ dn_max_bits, float eps_rel_d,
    float eps_rel_n, uint64_t* __restrict__ out_keys, int* __restrict__ flags,
    unsigned long long* __restrict__ n_incomplete) {
  const int lane = threadIdx.x;
  const int q = blockIdx.x;
  const uint64_t key = (lane < KC) ? cand[(size_t)q * KC + lane] : KEY_SENTINEL;
  const bool valid = key != KEY_SENTINEL;
  const float dco = ord2f((uint32_t)(key >> 32));
  const uint32_t idx = (uint32_t)key;
  const int n_valid = __popcll(__ballot(valid));
  const int kk = k < n_valid ? k : n_valid;
  int m = 0;
  bool complete = true;
  if (kk > 0) {
    const float theta = rerank_theta(__shfl(dco, kk - 1), qn[q], __uint_as_float(*dn_max_bits),
                                     eps_rel_d, eps_rel_n);
    m = __popcll(__ballot(valid && dco <= theta));
    if (n_valid == KC && n_range > KC) complete = __shfl(dco, KC - 1) > theta;
  }
  const uint64_t ek = (lane < m) ? make_key(exact[(size_t)q * KC + lane], idx) : KEY_SENTINEL;
  int rank = 0;  // keys are distinct (distinct rows)
  for (int i = 0; i < 64; ++i) {
    const uint64_t o = __shfl(ek, i);
    rank += (o < ek) ? 1 : 0;
  }
  if (lane < m && rank < k) out_keys[(size_t)q * k + rank] = ek;
  for (int i = m + lane; i < k; i += 64) out_keys[(size_t)q * k + i] = KEY_SENTINEL;
  if (lane == 0) {
    flags[q] = complete ? 0 : 1;
    if (!complete && n_incomplete) atomicAdd(n_incomplete, 1ull);
  }
}

// ---------------------------------------------------------------------------------------------
// K1 + K1b in ONE launch (MFMA path, <= 16 384 rows, KC <= 32 candidates, dim <= 4096): the query's
// work-group selects the KC coarse candidates (selq_select), computes the reference-order distances of
// the prefix that can reach the top k, ranks them and raises the query's flag when the candidate set
// may be incomplete -- what select_query + rerank_dist + rerank_final did in three launches, each with
// a ~4.6 us floor at this size.  Arithmetic identical to rerank_dist_kernel: group sums
// ((e0^2 + e1^2) + e2^2) + e3^2 of 4 dims, then the reference's sequential chain over the groups.
//   R1  wave w forms the group sums of candidates w and w + 16 (the query's groups loaded once) into LDS;
//   R2  4 waves x 8 lanes run the chains (one wave per SIMD: a chain is 1024 dependent adds whichever
//       lanes are active, so the chains of a SIMD should share instructions, not waves);
//   F   wave 0 ranks the exact keys.
constexpr int SRR_KC = 32;
constexpr int SRR_G = 1024;          // groups of 4 dims held per candidate
static_assert(SEL_LIST * 8 >= 4 * SRR_G * 4, "the selection's key list doubles as the staged query");
constexpr int SRR_LD = SRR_G + 4;    // floats per candidate row: 16 bytes of shift spread the chains' reads over the banks

// Reference-order distances of the query (staged in LDS: qs, dim floats) to the m <= 32 rows listed in
// rows_s (LDS), by the whole work-group of 1024 threads; exact_s[c] = distance to rows_s[c].  Ends with a
// barrier.  A chain is 1024 DEPENDENT fp32 adds (12 cycles each: 6 us) whatever else the chip does, so the
// work-group is split: waves 0-3 (8 lanes each, one candidate per lane) run the chains over chunk c of 256
// groups while waves 4-15 form the group sums of chunk c + 1 (three rows per wave: random 16-KB reads from
// HBM); a barrier per chunk hands the chunks over.  (Unpipelined: 6.0 + 5.9 us; the sums now hide behind the chains.)
__device__ __forceinline__ void srr_exact_batch(const float* __restrict__ db, const float* qs, int dim,
                                                const uint32_t* rows_s, int m, float* S, float* exact_s,
                                                unsigned long long* stamp = nullptr /* dev: [1] = after the first chunk's sums */) {
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int G = dim >> 2;
  const int n_chunks = (G + 255) / 256;
  const bool summing = w >= 4;
  // summing waves: candidates w - 4, w + 8, w + 20
  int cs[3];
  const float* rp[3];
#pragma unroll
  for (int r = 0; r < 3; ++r) {
    cs[r] = (w - 4) + 12 * r;
    rp[r] = db + (size_t)rows_s[(summing && cs[r] < m) ? cs[r] : 0] * dim;
  }
  auto sums = [&](int ch) {
    if (cs[0] >= m) return;  // wave-uniform
    const int gb = ch * 256;
    f4u x[3][4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {  // all twelve loads before the first use (clamped: unconditional)
      const int g = gb + u * 64 + lane, gc = g < G ? g : G - 1;
#pragma unroll
      for (int r = 0; r < 3; ++r) x[r][u] = *reinterpret_cast<const f4u*>(rp[r] + 4 * (size_t)gc);
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int g = gb + u * 64 + lane;
      if (g < G) {
        const f32x4 qv = *reinterpret_cast<const f32x4*>(qs + 4 * g);
#pragma unroll
        for (int r = 0; r < 3; ++r) {
          if (cs[r] < m) {
            const float e0 = qv.x - x[r][u].x, e1 = qv.y - x[r][u].y, e2 = qv.z - x[r][u].z, e3 = qv.w - x[r][u].w;
            S[cs[r] * SRR_LD + g] = e0 * e0 + e1 * e1 + e2 * e2 + e3 * e3;  // left-to-right, no FMA (as exact_pairs_wave)
          }
        }
      }
    }
  };
  const int cc = w * 8 + lane;  // chain lanes: waves 0-3, lanes 0-7
  const bool chaining = !summing && lane < 8 && cc < m;
  float acc = 0.f;
  if (summing) sums(0);
  __syncthreads();
  if (stamp && tid == 0) stamp[0] = __builtin_amdgcn_s_memtime();
  for (int ch = 0; ch < n_chunks; ++ch) {
    if (summing) {
      if (ch + 1 < n_chunks) sums(ch + 1);
    } else if (chaining) {
      const float* sp = S + cc * SRR_LD;
      const int end = (ch * 256 + 256) < G ? (ch * 256 + 256) : G;
      int i = ch * 256;
#pragma unroll 8
      for (; i + 4 <= end; i += 4) {
        const f32x4 v = *reinterpret_cast<const f32x4*>(sp + i);
        acc += v.x;
        acc += v.y;
        acc += v.z;
        acc += v.w;
      }
      for (; i < end; ++i) acc += sp[i];
    }
    __syncthreads();
  }
  if (chaining) exact_s[cc] = acc;
  __syncthreads();
}
// the query into LDS (dim <= 4 * SRR_G floats), one group of 4 dims per thread; the caller adds the barrier
__device__ __forceinline__ void srr_stage_query(const float* __restrict__ qp, int dim, float* qs) {
  const int g = threadIdx.x;
  if (g < (dim >> 2)) *reinterpret_cast<f32x4*>(qs + 4 * g) = *reinterpret_cast<const f32x4*>(qp + 4 * (size_t)g);
}

// The device-side fallback: a query whose candidate set cannot be proven complete is searched exactly by its
// own work-group, in the same launch -- reference-order distances of every row of the window (32 at a time:
// srr_exact_batch) into the query's own (dead) row of the coarse workspace, then the selection
// (selq_select<0>).  Slow by design (one CU streams the whole window: ~1.5 ms at 10 000 x 4096) and rare
// (0 of 13 440 queries at cfg B).  Rounds 1-2 enqueued two full-size launches (dist_exact, select) for the
// same purpose, the first form of this round one; now none.
__device__ __forceinline__ void srr_fallback_exact(const float* __restrict__ db, const float* __restrict__ qp, int dim,
                                                   size_t first_row, int n_range, int k, float* __restrict__ drow,
                                                   uint64_t* buf, float* qred, uint64_t* tau_s, int* cnt, float* S,
                                                   float* exact_s, uint32_t* rows_s, uint64_t* __restrict__ out_keys_q,
                                                   const FinalOut& fo, size_t out_base) {
  const int tid = threadIdx.x;
  float* qs = reinterpret_cast<float*>(buf);  // (the selection below takes buf over once the distances are out)
  __syncthreads();
  srr_stage_query(qp, dim, qs);
  for (int base = 0; base < n_range; base += SRR_KC) {
    const int mb = (n_range - base) < SRR_KC ? (n_range - base) : SRR_KC;
    if (tid < SRR_KC) rows_s[tid] = (uint32_t)(first_row + (size_t)base + (tid < mb ? tid : 0));
    __syncthreads();
    srr_exact_batch(db, qs, dim, rows_s, mb, S, exact_s);
    if (tid < mb) drow[base + tid] = exact_s[tid];
  }
  __threadfence_block();
  __syncthreads();
  selq_select<0>(drow, 0, 1, qp, dim, nullptr, first_row, n_range, k, buf, qred, tau_s, cnt);
  if (tid < k) {
    out_keys_q[tid] = buf[tid];
    final_store(fo, out_base + tid, buf[tid]);
  }
}

template <bool LISTS /* the candidates come from n_list ready keys per query (select_slices_kernel), not from P */>
__global__ __launch_bounds__(SELQ_THREADS) void select_rerank_kernel(
    const float* __restrict__ P, size_t ld, size_t strideP, int n_splits, const float* __restrict__ queries,
    int dim, const float* __restrict__ dn, size_t first_row, int n_range, int KC, int k,
    const float* __restrict__ db, const uint32_t* __restrict__ dn_max_bits, float eps_rel_d, float eps_rel_n,
    float* __restrict__ qn_out, uint64_t* __restrict__ out_keys /* [nq][k] */, int* __restrict__ flags,
    unsigned long long* __restrict__ n_incomplete, FinalOut fo, float* __restrict__ dist_scratch /* = P: row q of split 0 is this query's */,
    unsigned long long* __restrict__ dev_trace /* dev only: [nq][8] phase stamps, or null */,
    const uint64_t* __restrict__ lists = nullptr, int n_list = 0,
    const float* __restrict__ bmin = nullptr /* LISTS: != null -- the lists are made HERE from the coarse kernel's block minima [NB][nq] */,
    int NB = 0, uint64_t* __restrict__ blist = nullptr /* [nq][SELB_LIST]: room for them */) {
  const unsigned long long t_start = dev_trace ? __builtin_amdgcn_s_memtime() : 0ull;
  __shared__ uint64_t buf[SEL_LIST];
  __shared__ float qred[SELQ_THREADS / 64];
  __shared__ uint64_t tau_s;
  __shared__ int cnt;
  __shared__ __attribute__((aligned(16))) float S[SRR_KC * SRR_LD];
  __shared__ float exact_s[SRR_KC];
  __shared__ uint32_t rows_s[SRR_KC];
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int q = blockIdx.x;
  const float* qp = queries + (size_t)q * dim;
  float qnv;
  bool forced = false;  // (LISTS from block minima: more tied blocks than the list holds -- the query goes to the exact redo)
  if constexpr (LISTS) {
    __shared__ int blk[SELB_MAX_BLOCKS];
    __shared__ int nblk;
    const uint64_t* lq = lists + (size_t)q * n_list;
    int nl = n_list;
    if (bmin) {
      uint64_t* mine = blist + (size_t)q * SELB_LIST;
      forced = select_blocks_body(bmin + q, (size_t)gridDim.x, NB, P + (size_t)q * ld, qp, dim, dn, first_row, n_range, KC, mine, buf, qred,
                                  &tau_s, &cnt, blk, &nblk, S);  // ([block][query]: gridDim.x = nq; S: the re-rank's scratch, free until then)
      lq = mine;
      nl = SELB_LIST;
    }
    qnv = selq_select<2>(reinterpret_cast<const float*>(lq), 0, 1, qp, dim, dn, 0, nl, KC, buf,
                         qred, &tau_s, &cnt, dev_trace ? dev_trace + q * 16 + 8 : nullptr);
  } else
    qnv = selq_select<1>(P + (size_t)q * ld, strideP, n_splits, qp, dim, dn, first_row, n_range, KC, buf,
                         qred, &tau_s, &cnt, dev_trace ? dev_trace + q * 16 + 8 : nullptr);
  if (tid == 0) qn_out[q] = qnv;
  if (dev_trace && tid == 0) {
    dev_trace[q * 16 + 0] = t_start;
    dev_trace[q * 16 + 1] = __builtin_amdgcn_s_memtime();
  }
  // the prefix of candidates that gets reference-order distances (every wave computes the same values)
  const uint64_t ck = (lane < KC) ? buf[lane] : KEY_SENTINEL;
  const bool valid = ck != KEY_SENTINEL;
  const float dco = ord2f((uint32_t)(ck >> 32));
  const int n_valid = __popcll(__builtin_amdgcn_ballot_w64(valid));
  const int kk = k < n_valid ? k : n_valid;
  int m = 0;
  bool complete = true;
  if (kk > 0) {
    const float theta = rerank_theta(__shfl(dco, kk - 1), qnv, __uint_as_float(*dn_max_bits), eps_rel_d, eps_rel_n);
    m = __popcll(__builtin_amdgcn_ballot_w64(valid && dco <= theta));
    if (n_valid == KC && n_range > KC) complete = __shfl(dco, KC - 1) > theta;
  }
  if (forced) complete = false;  // (uniform: one value per query)
  if (w == 0 && lane < SRR_KC) rows_s[lane] = valid ? (uint32_t)ck : 0u;
  __syncthreads();  // every wave has its candidates in registers: buf is free and takes the query
  float* qs = reinterpret_cast<float*>(buf);
  srr_stage_query(qp, dim, qs);
  __syncthreads();
  if (dev_trace && tid == 0) dev_trace[q * 16 + 2] = __builtin_amdgcn_s_memtime();
  srr_exact_batch(db, qs, dim, rows_s, m, S, exact_s, dev_trace ? dev_trace + q * 16 + 3 : nullptr);
  if (dev_trace && tid == 0) dev_trace[q * 16 + 4] = __builtin_amdgcn_s_memtime();
  // F: rank the exact keys (distinct: distinct rows)
  if (w == 0) {
    const uint64_t ek = (lane < m) ? make_key(exact_s[lane], (uint32_t)ck) : KEY_SENTINEL;
    int rank = 0;
    for (int i = 0; i < m; ++i) {  // (i is uniform: v_readlane, not the LDS crossbar of __shfl -- 3.5 k -> 2.6 k cycles)
      const uint64_t o = ((uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(ek >> 32), i) << 32) |
                         (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)ek, i);
      rank += (o < ek) ? 1 : 0;
    }
    if (lane < m && rank < k) {
      out_keys[(size_t)q * k + rank] = ek;
      final_store(fo, (size_t)q * k + rank, ek);
    }
    for (int i = m + lane; i < k; i += 64) {
      out_keys[(size_t)q * k + i] = KEY_SENTINEL;
      final_store(fo, (size_t)q * k + i, KEY_SENTINEL);
    }
    if (lane == 0) {
      flags[q] = complete ? 0 : 1;
      if (!complete && n_incomplete) atomicAdd(n_incomplete, 1ull);
      if (dev_trace) {
        dev_trace[q * 16 + 5] = __builtin_amdgcn_s_memtime();
        dev_trace[q * 16 + 6] = (unsigned long long)m;
      }
    }
  }
  // (a window above 16 384 rows is not searched exactly by ONE work-group -- 16 GB through one CU at a million rows: the
  // flag goes to the host, which redoes the query on the exact path, knn.hip)
  if constexpr (!LISTS)
  if (!complete)  // uniform over the work-group (every wave computed the same proof)
    srr_fallback_exact(db, qp, dim, first_row, n_range, k, dist_scratch + (size_t)q * ld, buf, qred, &tau_s, &cnt, S, exact_s,
                       rows_s, out_keys + (size_t)q * k, fo, (size_t)q * k);
}

// keys -> (u64 index * stride + offset, f32 d2); sentinel -> (UINT64_MAX, FLT_MAX).  stride / offset place a
// shard's local rows in the global database (contiguous shards: stride 1; interleaved: stride = shards).
__global__ void finalize_kernel(const uint64_t* __restrict__ keys, size_t n, uint64_t offset, uint64_t stride,
                                uint64_t* __restrict__ out_idx, float* __restrict__ out_d2) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const uint64_t k = keys[i];
  if (k == KEY_SENTINEL) {
    out_idx[i] = ~0ull;
    out_d2[i] = 3.402823466e+38f;
  } else {
    out_idx[i] = (uint64_t)(uint32_t)k * stride + offset;
    out_d2[i] = ord2f((uint32_t)(k >> 32));
  }
}

// K3: merge n_lists sorted top-k lists per query by (d2, idx).  One wave per query; k*n_lists
// entries ranked by counting (n_lists*k <= 1024).
__global__ __launch_bounds__(64) void merge_kernel(const uint64_t* __restrict__ idx,
                                                   const float* __restrict__ d2, int n_lists,
                                                   int nq, int k, uint64_t* __restrict__ out_idx,
                                                   float* __restrict__ out_d2) {
  __shared__ float sd[1024];
  __shared__ uint64_t si[1024];
  const int q = blockIdx.x, lane = threadIdx.x;
  const int total = n_lists * k;
  for (int e = lane; e < total; e += 64) {
    const int l = e / k, r = e % k;
    const size_t off = ((size_t)l * nq + q) * k + r;
    sd[e] = d2[off];
    si[e] = idx[off];
  }
  __syncthreads();
  for (int e = lane; e < total; e += 64) {
    const float d = sd[e];
    const uint64_t i = si[e];
    int rank = 0;
    for (int o = 0; o < total; ++o) {
      const float od = sd[o];
      const uint64_t oi = si[o];
      rank += (od < d || (od == d && (oi < i || (oi == i && o < e)))) ? 1 : 0;
    }
    if (rank < k) {
      out_idx[(size_t)q * k + rank] = i;
      out_d2[(size_t)q * k + rank] = d;
    }
  }
}

}  // namespace knn
}  // namespace gloc
