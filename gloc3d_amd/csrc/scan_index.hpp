// scan_index.hpp -- the search index of one resident scan and the kernels that build it on the
// device (no host pass over the points, no host synchronisation between the steps):
//   bounding box -> Hilbert keys on a 1024^3 grid -> radix sort -> sorted float4 copy with the
//   original indices, inverse permutation -> boxes per 128-point chunk, per 16-point sub-block and
//   per 64-chunk super-chunk -> launch order of the source groups (widest first).
// A scan that serves as a registration TARGET (a database place) is re-sorted once more, into KD ORDER
// (gloc_scan_store_build_target_index): a complete binary tree over the sorted positions, every aligned
// block of 16 * 2^j positions one kd cell -- at every level the points of a cell are split at the middle
// position along the widest axis of their bounding box.  Cells of one level are disjoint and fit the
// point density, where runs of a space-filling curve have irregular, overlapping bounding boxes: on
// lidar scans a moved query point is within its nearest-neighbour distance of 1.7 chunk boxes and 2.5
// sub-block boxes instead of 2.6 and 3.5, and the culled search evaluates 30 % fewer pairs and tests 35 %
// fewer boxes.  The order of a scan never changes a result: the search is exact either way.
// The structures live here; the kernels that fill them are in scan_store.hip; the 1-NN kernels
// (nn_compact.hpp) read them.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "math3.hpp"

namespace gloc {
namespace reg {

constexpr int CH = 128;  // points per chunk (256 / 32 measured: -9 % throughput with three queries in flight)
constexpr int SB = 16;   // points per sub-block (second-level boxes, argmin bookkeeping)
// sub-block boxes (sb2) are kept as centre c and -h / SB2_RANGE per axis (h: half extent, rounded up): the distance of a
// point to the box along an axis, in units of SB2_RANGE metres and saturated at one unit, is then
// clamp(|p - c| / SB2_RANGE - h / SB2_RANGE) -- one fma with free abs / clamp modifiers.  A power of two: exact.
constexpr float SB2_RANGE = 64.f, SB2_INV_RANGE = 1.f / 64.f;

// Device-resident header of a scan (first 64 bytes of its allocation): written by the indexing
// kernels, read by the search kernels, never by the host.
struct ScanHeader {
  float ox, oy, oz, inv_cell;  // origin and 1 / cell of the 1024^3 key grid
  uint32_t lo[3], hi[3];       // bounding box as order-preserving integers (see f2ord)
  uint32_t pad_[6];
};
static_assert(sizeof(ScanHeader) == 64, "header is 64 bytes");

struct ScanIndexDev {
  const f32x4* pts;      // Hilbert order: x, y, z, bits(original index)
  // per chunk of CH points, since round 4 as (centre, -half extent / SB2_RANGE): box_lo holds the centres, box_hi the
  // negated scaled half extents (the names are the arrays', from when they held the corners)
  const f32x4* box_lo;
  const f32x4* box_hi;
  // boxes of the sub-blocks of SB points, two sub-blocks (2q, 2q+1) interleaved in 3 float4:
  // (lo0.x lo1.x lo0.y lo1.y) (lo0.z lo1.z hi0.x hi1.x) (hi0.y hi1.y hi0.z hi1.z) -- a lane tests one
  // point against both with packed fp32 instructions, the pairs being register pairs as loaded
  const f32x4* sb2;
  // Target index only (null for a scan in plain Hilbert order): position in `pts` of the point with the
  // i-th smallest curve key.  A target index is sorted in kd order (see below), the curve keys stay the
  // cold-start lookup: binary search in `keys`, then through kpos to the points.
  const uint32_t* kpos;
  const uint32_t* keys;  // sorted curve keys
  const uint32_t* inv;   // original index -> sorted position
  const ScanHeader* hdr;
  const f32x4* sup_lo;   // per super-chunk of 64 chunks (8192 points)
  const f32x4* sup_hi;
  uint32_t n, nchunks, nsup, pad_;
};

// order-preserving map float -> uint32 (for atomicMin / atomicMax on coordinates)
__host__ __device__ __forceinline__ uint32_t f2ord(float f) {
  const uint32_t u = __builtin_bit_cast(uint32_t, f);
  return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__host__ __device__ __forceinline__ float ord2f(uint32_t o) {
  const uint32_t u = (o & 0x80000000u) ? (o & 0x7FFFFFFFu) : ~o;
  return __builtin_bit_cast(float, u);
}

}  // namespace reg
}  // namespace gloc
