// lane_ops.hpp -- lane exchanges inside a wave without the LDS crossbar (ds_bpermute): DPP row
// operations below 16 lanes, gfx950's v_permlane16_swap / v_permlane32_swap across rows.
// ds_bpermute costs an LDS-pipe slot and ~100 cycles of latency per dword; a 21-step bitonic sort of
// 64-bit keys built on it is a 4000-cycle dependent chain, the DPP form a few hundred.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace gloc {

// x of lane (l ^ O), O = 1, 2, 4, 8, 16, 32
template <int O>
__device__ __forceinline__ uint32_t xor_lane_u32(uint32_t x) {
  static_assert(O == 1 || O == 2 || O == 4 || O == 8 || O == 16 || O == 32, "a power of two below 64");
  typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
  if constexpr (O == 1) {
    return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0xB1, 0xF, 0xF, true);  // quad_perm [1,0,3,2]
  } else if constexpr (O == 2) {
    return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x4E, 0xF, 0xF, true);  // quad_perm [2,3,0,1]
  } else if constexpr (O == 8) {
    return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x128, 0xF, 0xF, true);  // row_ror:8
  } else if constexpr (O == 4) {
    // row_ror:12 for the banks whose lanes have bit 2 clear (they read lane + 4), row_ror:4 for the others
    const int a = __builtin_amdgcn_update_dpp(0, (int)x, 0x12C, 0xF, 0x5, false);
    return (uint32_t)__builtin_amdgcn_update_dpp(a, (int)x, 0x124, 0xF, 0xA, false);
  } else if constexpr (O == 16) {
    // swap(x, x): .x = rows (0, 0, 2, 2) of x, .y = rows (1, 1, 3, 3): the partner row is .y for even rows
    const u32x2 r = __builtin_amdgcn_permlane16_swap(x, x, false, false);
    return (__lane_id() & 16) ? r.x : r.y;
  } else {
    const u32x2 r = __builtin_amdgcn_permlane32_swap(x, x, false, false);  // .x = (lo, lo), .y = (hi, hi)
    return (__lane_id() & 32) ? r.x : r.y;
  }
}
template <int O>
__device__ __forceinline__ uint64_t xor_lane_u64(uint64_t x) {
  return ((uint64_t)xor_lane_u32<O>((uint32_t)(x >> 32)) << 32) | xor_lane_u32<O>((uint32_t)x);
}
template <int O>
__device__ __forceinline__ double xor_lane(double x) {
  return __builtin_bit_cast(double, xor_lane_u64<O>(__builtin_bit_cast(uint64_t, x)));
}
template <int O>
__device__ __forceinline__ float xor_lane(float x) {
  return __uint_as_float(xor_lane_u32<O>(__float_as_uint(x)));
}

}  // namespace gloc
