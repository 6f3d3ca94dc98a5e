// synth_kernels.hpp -- on-device twin of gloc3d_amd/synth.py (deterministic synthetic descriptors).
// Integer hashing + single IEEE fp32 operations only (no FMA contraction: -ffp-contract=off), so
// the bits equal the numpy and C generators.  Used for databases too large to upload (cfg E).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace gloc {
namespace synth {

__host__ __device__ __forceinline__ uint64_t mix64(uint64_t z) {
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
  return z ^ (z >> 31);
}
__host__ __device__ __forceinline__ uint64_t rng_key(uint64_t seed, uint64_t stream) {
  return mix64(mix64(seed + 0x9E3779B97F4A7C15ULL) ^ (stream * 0xD1B54A32D192ED03ULL + 1ULL));
}
__host__ __device__ __forceinline__ uint64_t rng_draw(uint64_t key, uint64_t ctr) {
  return mix64(key + (ctr + 1ULL) * 0x9E3779B97F4A7C15ULL);
}
__host__ __device__ __forceinline__ float rng_gauss(uint64_t key, uint64_t ctr) {
  const uint64_t u = rng_draw(key, ctr);
  const int s = (int)((u & 0xFFFF) + ((u >> 16) & 0xFFFF) + ((u >> 32) & 0xFFFF) +
                      ((u >> 48) & 0xFFFF)) -
                131070;
  return (float)s * (1.0f / 37837.227f);
}

constexpr uint64_t TAG_NOISE = 0x6E6F697365ULL;

// kind 0: iid N(0,1)/sqrt(dim).  kind 1: anchored trajectory, stride 16, noise 0.05
// (descriptors_traj in synth.py: ((1-t) a_k + t a_{k+1}) + 0.05 noise).
static __global__ void fill_kernel(int kind, uint64_t seed, uint64_t first_row, size_t n,
                                   size_t dim, uint64_t row_stride, float scale,
                                   float* __restrict__ out) {
  const size_t total = n * dim;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total;
       i += (size_t)gridDim.x * blockDim.x) {
    const uint64_t row = first_row + (i / dim) * row_stride;
    const uint64_t c = i % dim;
    float v;
    if (kind == 0) {
      v = rng_gauss(rng_key(seed, row), c) * scale;
    } else {
      const uint64_t k = row / 16;
      const float t = (float)(row % 16) / 16.0f;
      const float a0 = rng_gauss(rng_key(seed, k), c) * scale;
      const float a1 = rng_gauss(rng_key(seed, k + 1), c) * scale;
      const float ns = rng_gauss(rng_key(seed ^ TAG_NOISE, row), c) * scale;
      const float w0 = 1.0f - t;
      const float x = w0 * a0;
      const float y = t * a1;
      const float z = 0.05f * ns;
      v = (x + y) + z;
    }
    out[i] = v;
  }
}

inline void launch_fill(hipStream_t s, int kind, uint64_t seed, uint64_t first_row, size_t n,
                        size_t dim, uint64_t row_stride, float* d_out) {
  const float scale = (float)(1.0 / __builtin_sqrt((double)dim));
  const size_t total = n * dim;
  unsigned blocks = (unsigned)((total + 255) / 256 < 8192 ? (total + 255) / 256 : 8192);
  hipLaunchKernelGGL(fill_kernel, dim3(blocks), dim3(256), 0, s, kind, seed, first_row, n, dim,
                     row_stride, scale, d_out);
}

}  // namespace synth
}  // namespace gloc
