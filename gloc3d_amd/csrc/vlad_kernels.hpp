// vlad_kernels.hpp -- NetVLAD-FC pooling head on gfx950.
// Replaces NetVLAD.forward of the reference (model/netvlad_fc.py:73-109; instantiated without gating
// at main.py:594):  per-position L2 normalisation -> 1x1-conv soft assignment (softmax over clusters)
// -> residual aggregation -> intra-normalisation -> L2 -> FC [K*C -> out].
//
//   vlad_tile_kernel     grid (position tiles of 64, images): the feature tile [C][64] is staged in
//                        LDS once (133 KB at C = 512) and feeds BOTH fp32-MFMA products:
//                        logits[k][p] = W[k][:] . x^[:][p]   and   V[k][c] += soft[k][p] x^[c][p].
//   vlad_cluster_kernel  grid (clusters, images): sum the tiles' partial V, subtract centroid * S,
//                        intra-normalise.
//   vlad_fc_kernel       grid (K*C / 512, out / 256): partial GEMV/GEMM over a 512-row slab of the FC
//                        matrix (HBM-bound: 67 MB at 64 x 512 x 512), up to 8 images per pass.
//   vlad_fc_reduce_kernel  fixed-order sum of the slabs' partials, global L2 scale.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace gloc {
namespace vlad {

typedef float f32x4 __attribute__((ext_vector_type(4)));
constexpr int VP = 64;        // positions per tile
constexpr int VPITCH = VP + 1;
constexpr int FC_ROWS = 512;  // FC slab
constexpr int FC_NB = 8;      // images per FC pass

// dynamic LDS: X [C][VPITCH] | SOFT [Kp][VPITCH] | red [4][VP] (x2) | inv [VP]
__global__ __launch_bounds__(256) void vlad_tile_kernel(
    const float* __restrict__ feat /* [n][C][HW] */, int C, int HW, int K, int Kp,
    const float* __restrict__ conv_w /* [K][C] */, const float* __restrict__ conv_b /* [K] or null */,
    int normalize_input, float* __restrict__ partV /* [n][tiles][Kp][C] */,
    float* __restrict__ partS /* [n][tiles][Kp] */) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* X = smem;
  float* SOFT = X + (size_t)C * VPITCH;
  float* red = SOFT + (size_t)Kp * VPITCH;  // [2][4][VP]
  float* inv = red + 2 * 4 * VP;
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int tile = blockIdx.x, img = blockIdx.y, ntiles = gridDim.x;
  const int p0 = tile * VP;
  const float* f = feat + (size_t)img * C * HW;

  // a. stage the tile (positions are contiguous in NCHW: 256-B rows), zero beyond HW
  for (int e = tid; e < C * VP; e += 256) {
    const int c = e / VP, p = e % VP;
    X[c * VPITCH + p] = (p0 + p < HW) ? f[(size_t)c * HW + p0 + p] : 0.f;
  }
  __syncthreads();
  // b. per-position inverse norm (model/netvlad_fc.py:76-77: F.normalize, eps 1e-12)
  {
    float s = 0.f;
    for (int c = w; c < C; c += 4) {
      const float v = X[c * VPITCH + lane];
      s += v * v;
    }
    red[w * VP + lane] = s;
  }
  __syncthreads();
  if (tid < VP) {
    const float s = (red[tid] + red[VP + tid]) + (red[2 * VP + tid] + red[3 * VP + tid]);
    inv[tid] = normalize_input ? 1.0f / fmaxf(sqrtf(s), 1e-12f) : 1.0f;
  }
  __syncthreads();
  for (int e = tid; e < C * VP; e += 256) {
    const int c = e / VP, p = e % VP;
    X[c * VPITCH + p] *= inv[p];
  }
  __syncthreads();

  // c. logits: wave w <-> cluster tile w (16 clusters) x 4 position tiles, K = C
  const int ktiles = Kp / 16;
  f32x4 lg[4];
#pragma unroll
  for (int t = 0; t < 4; ++t) lg[t] = f32x4{0.f, 0.f, 0.f, 0.f};
  if (w < ktiles) {
    const int krow = w * 16 + (lane & 15);
    const float* wrow = conv_w + (size_t)(krow < K ? krow : 0) * C;
    for (int c0 = 0; c0 < C; c0 += 16) {
      f32x4 a = {0.f, 0.f, 0.f, 0.f};
      const int cb = c0 + 4 * (lane >> 4);
      if (krow < K) {
        a.x = (cb + 0 < C) ? wrow[cb + 0] : 0.f;
        a.y = (cb + 1 < C) ? wrow[cb + 1] : 0.f;
        a.z = (cb + 2 < C) ? wrow[cb + 2] : 0.f;
        a.w = (cb + 3 < C) ? wrow[cb + 3] : 0.f;
      }
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        const int p = t * 16 + (lane & 15);
        const float b0 = (cb + 0 < C) ? X[(cb + 0) * VPITCH + p] : 0.f;
        const float b1 = (cb + 1 < C) ? X[(cb + 1) * VPITCH + p] : 0.f;
        const float b2 = (cb + 2 < C) ? X[(cb + 2) * VPITCH + p] : 0.f;
        const float b3 = (cb + 3 < C) ? X[(cb + 3) * VPITCH + p] : 0.f;
        lg[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.x, b0, lg[t], 0, 0, 0);
        lg[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.y, b1, lg[t], 0, 0, 0);
        lg[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.z, b2, lg[t], 0, 0, 0);
        lg[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.w, b3, lg[t], 0, 0, 0);
      }
    }
  }
  // C/D map: column = lane & 15 (position inside the tile), row = 4 (lane >> 4) + r (cluster)
  // bias, padded clusters -> -inf, softmax over clusters (model/netvlad_fc.py:80-81)
  float mx[4], sm[4];
#pragma unroll
  for (int t = 0; t < 4; ++t) {
    mx[t] = -3.4e38f;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int k = w * 16 + 4 * (lane >> 4) + r;
      float v = lg[t][r];
      if (conv_b && k < K) v += conv_b[k];
      if (k >= K || w >= ktiles) v = -3.4e38f;
      lg[t][r] = v;
      mx[t] = fmaxf(mx[t], v);
    }
    mx[t] = fmaxf(mx[t], __shfl_xor(mx[t], 16));
    mx[t] = fmaxf(mx[t], __shfl_xor(mx[t], 32));
    if (lane < 16) red[w * VP + t * 16 + lane] = mx[t];
  }
  __syncthreads();
#pragma unroll
  for (int t = 0; t < 4; ++t) {
    const int p = t * 16 + (lane & 15);
    const float m = fmaxf(fmaxf(red[p], red[VP + p]), fmaxf(red[2 * VP + p], red[3 * VP + p]));
    sm[t] = 0.f;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const float e = (lg[t][r] > -3.0e38f) ? expf(lg[t][r] - m) : 0.f;
      lg[t][r] = e;
      sm[t] += e;
    }
    sm[t] += __shfl_xor(sm[t], 16);
    sm[t] += __shfl_xor(sm[t], 32);
    if (lane < 16) red[4 * VP + w * VP + t * 16 + lane] = sm[t];
  }
  __syncthreads();
  if (w < ktiles) {
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      const int p = t * 16 + (lane & 15);
      const float* r2 = red + 4 * VP;
      const float tot = (r2[p] + r2[VP + p]) + (r2[2 * VP + p] + r2[3 * VP + p]);
      const float sc = (p0 + p < HW) ? 1.0f / tot : 0.f;  // padded positions carry no weight
#pragma unroll
      for (int r = 0; r < 4; ++r) SOFT[(w * 16 + 4 * (lane >> 4) + r) * VPITCH + p] = lg[t][r] * sc;
    }
  }
  __syncthreads();

  // d. aggregation: V[k][c] = sum_p soft[k][p] x^[c][p]; wave w <-> cluster tile w, all C/16 column
  //    tiles in groups of 8 accumulators
  if (w < ktiles) {
    const int krow = w * 16 + (lane & 15);
    float* pv = partV + (((size_t)img * ntiles + tile) * Kp) * C;
    for (int ct0 = 0; ct0 < (C + 15) / 16; ct0 += 8) {
      f32x4 acc[8];
#pragma unroll
      for (int t = 0; t < 8; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
      for (int s = 0; s < VP / 4; ++s) {
        const int p = 4 * s + (lane >> 4);
        const float a = SOFT[krow * VPITCH + p];
#pragma unroll
        for (int t = 0; t < 8; ++t) {
          const int c = (ct0 + t) * 16 + (lane & 15);
          const float b = (c < C) ? X[c * VPITCH + p] : 0.f;
          acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[t], 0, 0, 0);
        }
      }
#pragma unroll
      for (int t = 0; t < 8; ++t) {
        const int c = (ct0 + t) * 16 + (lane & 15);
        if (c < C) {
#pragma unroll
          for (int r = 0; r < 4; ++r) pv[(size_t)(w * 16 + 4 * (lane >> 4) + r) * C + c] = acc[t][r];
        }
      }
    }
  }
  if (tid < Kp) {  // S[k] = sum_p soft[k][p]
    float s = 0.f;
    for (int p = 0; p < VP; ++p) s += SOFT[tid * VPITCH + p];
    partS[((size_t)img * ntiles + tile) * Kp + tid] = s;
  }
}

// grid (K, n).  vlad[k][:] = (sum_tiles V - centroid[k] * sum_tiles S) / max(||.||, 1e-12)
// (model/netvlad_fc.py:88-99); also the cluster's contribution to the global sum of squares.
__global__ __launch_bounds__(256) void vlad_cluster_kernel(
    const float* __restrict__ partV, const float* __restrict__ partS, int ntiles, int C, int K, int Kp,
    const float* __restrict__ centroids, float* __restrict__ vlad /* [n][K*C] */,
    float* __restrict__ nrm2 /* [n][K] */) {
  __shared__ float red[256];
  const int k = blockIdx.x, img = blockIdx.y, tid = threadIdx.x;
  float S = 0.f;
  for (int t = 0; t < ntiles; ++t) S += partS[((size_t)img * ntiles + t) * Kp + k];
  float ss = 0.f;
  for (int c = tid; c < C; c += 256) {
    float v = 0.f;
    for (int t = 0; t < ntiles; ++t) v += partV[((((size_t)img * ntiles + t) * Kp) + k) * C + c];
    v -= centroids[(size_t)k * C + c] * S;
    vlad[((size_t)img * K + k) * C + c] = v;
    ss += v * v;
  }
  red[tid] = ss;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if (tid < o) red[tid] += red[tid + o];
    __syncthreads();
  }
  const float inv = 1.0f / fmaxf(sqrtf(red[0]), 1e-12f);
  __syncthreads();
  float s2 = 0.f;
  for (int c = tid; c < C; c += 256) {
    const float v = vlad[((size_t)img * K + k) * C + c] * inv;
    vlad[((size_t)img * K + k) * C + c] = v;
    s2 += v * v;
  }
  red[tid] = s2;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if (tid < o) red[tid] += red[tid + o];
    __syncthreads();
  }
  if (tid == 0) nrm2[(size_t)img * K + k] = red[0];
}

// grid (slabs of FC_ROWS rows, out / 256).  part[slab][img][o] = sum_{j in slab} vlad[img][j] W[j][o]
__global__ __launch_bounds__(256) void vlad_fc_kernel(const float* __restrict__ vlad, int n0, int nb,
                                                      int KC, int out_dim,
                                                      const float* __restrict__ fc_w,
                                                      float* __restrict__ part /* [slabs][FC_NB][out] */) {
  __shared__ float v[FC_NB][FC_ROWS];
  const int slab = blockIdx.x, o = blockIdx.y * 256 + threadIdx.x;
  const int j0 = slab * FC_ROWS;
  const int rows = (KC - j0) < FC_ROWS ? (KC - j0) : FC_ROWS;
  for (int e = threadIdx.x; e < FC_NB * FC_ROWS; e += 256) {
    const int i = e / FC_ROWS, j = e % FC_ROWS;
    v[i][j] = (i < nb && j < rows) ? vlad[(size_t)(n0 + i) * KC + j0 + j] : 0.f;
  }
  __syncthreads();
  if (o >= out_dim) return;
  float acc[FC_NB];
#pragma unroll
  for (int i = 0; i < FC_NB; ++i) acc[i] = 0.f;
  const float* wp = fc_w + (size_t)j0 * out_dim + o;
#pragma unroll 4
  for (int j = 0; j < rows; ++j) {
    const float wv = wp[(size_t)j * out_dim];
#pragma unroll
    for (int i = 0; i < FC_NB; ++i) acc[i] += v[i][j] * wv;
  }
#pragma unroll
  for (int i = 0; i < FC_NB; ++i) part[((size_t)slab * FC_NB + i) * out_dim + o] = acc[i];
}

// out[img][o] = (sum_slabs part) / max(sqrt(sum_k nrm2[img][k]), 1e-12)   (netvlad_fc.py:101-105)
__global__ void vlad_fc_reduce_kernel(const float* __restrict__ part, int nslabs, int n0, int nb,
                                      int out_dim, const float* __restrict__ nrm2, int K,
                                      float* __restrict__ out) {
  const int o = blockIdx.x * blockDim.x + threadIdx.x, i = blockIdx.y;
  if (o >= out_dim || i >= nb) return;
  float g = 0.f;
  for (int k = 0; k < K; ++k) g += nrm2[(size_t)(n0 + i) * K + k];
  const float sc = 1.0f / fmaxf(sqrtf(g), 1e-12f);
  float s = 0.f;
  for (int sl = 0; sl < nslabs; ++sl) s += part[((size_t)sl * FC_NB + i) * out_dim + o];
  out[(size_t)(n0 + i) * out_dim + o] = s * sc;
}

// GatingContext (model/netvlad_fc.py:120-146): gates = x W; BatchNorm1d in eval mode or + bias, both as
// gates * scale + shift (the caller folds running_mean / running_var / weight / bias into scale and shift);
// sigmoid; out = x * gates.  One wave per (image, 64 outputs): lanes <-> outputs, the image's descriptor
// broadcast from LDS.  In place: `io` is read completely before it is written.
__global__ __launch_bounds__(64) void vlad_gate_kernel(float* __restrict__ io, int n, int dim,
                                                       const float* __restrict__ gw /* [dim][dim] */,
                                                       const float* __restrict__ scale,
                                                       const float* __restrict__ shift, float* __restrict__ out) {
  extern __shared__ float xs[];
  const int img = blockIdx.y, j = blockIdx.x * 64 + threadIdx.x;
  for (int i = threadIdx.x; i < dim; i += 64) xs[i] = io[(size_t)img * dim + i];
  __syncthreads();
  if (j >= dim) return;
  float acc = 0.f;
  for (int i = 0; i < dim; ++i) acc += xs[i] * gw[(size_t)i * dim + j];  // coalesced over j
  const float g = acc * scale[j] + shift[j];
  const float sg = 1.0f / (1.0f + expf(-g));
  out[(size_t)img * dim + j] = xs[j] * sg;
}

}  // namespace vlad
}  // namespace gloc
