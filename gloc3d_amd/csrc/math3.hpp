// math3.hpp -- device-side 3-D helpers shared by the registration and the ground kernels (no kernels
// here, so that several translation units can include it).  Fixed, un-fused operation orders: the same
// sequences as oracle/reg_oracle.c, compiled with -ffp-contract=off.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace gloc {
namespace reg {

typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr float NN_FAR = 1.0e18f;  // padding coordinate: squares to +inf, never the minimum

__device__ __forceinline__ void xform(const float* __restrict__ T, float x, float y, float z,
                                      float& ox, float& oy, float& oz) {
  ox = ((T[0] * x + T[1] * y) + T[2] * z) + T[9];
  oy = ((T[3] * x + T[4] * y) + T[5] * z) + T[10];
  oz = ((T[6] * x + T[7] * y) + T[8] * z) + T[11];
}
__device__ __forceinline__ float dist2(float ax, float ay, float az, float bx, float by, float bz) {
  const float dx = ax - bx, dy = ay - by, dz = az - bz;
  return (dx * dx + dy * dy) + dz * dz;
}


// ---- fp64 3x3 helpers: the same operation sequence as oracle/reg_oracle.c --------------------
__device__ inline void jacobi_eig3(double A[9], double V[9]) {
  V[0] = 1; V[1] = 0; V[2] = 0;
  V[3] = 0; V[4] = 1; V[5] = 0;
  V[6] = 0; V[7] = 0; V[8] = 1;
  for (int sweep = 0; sweep < 16; ++sweep) {
    const double off = A[1] * A[1] + A[2] * A[2] + A[5] * A[5];
    const double diag = A[0] * A[0] + A[4] * A[4] + A[8] * A[8];
    if (off <= 1e-32 * diag || off == 0.0) break;
    for (int e = 0; e < 3; ++e) {
      const int p = (e == 2) ? 1 : 0, q = (e == 0) ? 1 : 2;
      const double apq = A[3 * p + q];
      if (apq == 0.0) continue;
      const double app = A[3 * p + p], aqq = A[3 * q + q];
      const double theta = (aqq - app) / (2.0 * apq);
      const double at = theta < 0 ? -theta : theta;
      double t = 1.0 / (at + sqrt(theta * theta + 1.0));
      if (theta < 0) t = -t;
      const double c = 1.0 / sqrt(t * t + 1.0);
      const double s = t * c;
      for (int k = 0; k < 3; ++k) {
        const double akp = A[3 * k + p], akq = A[3 * k + q];
        A[3 * k + p] = c * akp - s * akq;
        A[3 * k + q] = s * akp + c * akq;
      }
      for (int k = 0; k < 3; ++k) {
        const double apk = A[3 * p + k], aqk = A[3 * q + k];
        A[3 * p + k] = c * apk - s * aqk;
        A[3 * q + k] = s * apk + c * aqk;
      }
      for (int k = 0; k < 3; ++k) {
        const double vkp = V[3 * k + p], vkq = V[3 * k + q];
        V[3 * k + p] = c * vkp - s * vkq;
        V[3 * k + q] = s * vkp + c * vkq;
      }
    }
  }
}

__device__ __forceinline__ void cross3(const double a[3], const double b[3], double o[3]) {
  o[0] = a[1] * b[2] - a[2] * b[1];
  o[1] = a[2] * b[0] - a[0] * b[2];
  o[2] = a[0] * b[1] - a[1] * b[0];
}


__device__ __forceinline__ uint32_t mulhi_idx(uint64_t u, uint32_t n) {
  return (uint32_t)__umul64hi(u, (uint64_t)n);
}


// ---- spatial keys and box bounds (shared by the culled 1-NN and the ground 10-NN) ----------------
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


// squared distance from a point to a box, scaled down by 2^-20 so that fp32 rounding can never
// push it above the (fp32) distance of any point inside the box
__device__ __forceinline__ float box_lb(float px, float py, float pz, const f32x4& lo,
                                        const f32x4& hi) {
  const float ex = fmaxf(fmaxf(lo.x - px, px - hi.x), 0.f);
  const float ey = fmaxf(fmaxf(lo.y - py, py - hi.y), 0.f);
  const float ez = fmaxf(fmaxf(lo.z - pz, pz - hi.z), 0.f);
  return ((ex * ex + ey * ey) + ez * ez) * 0.99999905f;
}

}  // namespace reg
}  // namespace gloc
