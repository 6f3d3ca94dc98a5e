// scan_store.hip -- the resident scan store (gloc_scan_store_* of include/gloc3d.h): every database
// scan of the reference's GlocEvaluator (db_files_, read again from disk for every candidate at
// registration/global_localization.cpp:521-525) is kept in HBM together with its search index, shared
// by any number of registration handles; query scans are added, used and released.
// Indexing runs entirely on the device (no host pass over the points), and for ANY NUMBER OF SCANS AT ONCE
// with the same launches: every kernel takes the scan from blockIdx.y and a descriptor table, every sort is
// the segmented radix sort of seg_sort.hpp with one segment per scan.  Adding the 25 query scans of a step is
// one sequence of ~35 launches (round 2: ~25 launches per scan, a third of them inside hipCUB's merge sort);
// re-sorting a KITTI-00-sized database into kd order is 71 batches of 64 scans.
#include <algorithm>
#include <new>

#include "scan_store.hpp"
#include "seg_sort.hpp"
#include <cmath>

#include "raycast_kernels.hpp"
#include "synth_kernels.hpp"

namespace gloc {
namespace reg {

constexpr int PACK_BLOCKS = 128;  // work-groups of the bounding-box pass per scan

// Device-visible description of one scan being indexed (one per blockIdx.y).
struct ScanBuild {
  const float* in;  // source points on the device, `stride` floats apart
  ScanHeader* hdr;
  float* xyz;
  f32x4* pts;
  f32x4* lo;
  f32x4* hi;
  f32x4* sb2;
  f32x4* ulo;
  f32x4* uhi;
  uint32_t* keys;
  uint32_t* inv;
  uint32_t* order;  // the launch order being built (one sources-per-lane setting)
  uint32_t n, stride, n_pad, nch, nsup, npairs, n_groups, group;
  uint32_t key_off;  // this scan's slice of the concatenated sort arrays (points)
  uint32_t grp_off;  // ... and of the group arrays
  uint32_t pad_[2];
};

// strided (x, y, z, ...) -> packed xyz, and the bounding box: every work-group leaves its partial box
// in `part` ([scan][PACK_BLOCKS][6] order-preserving integers), reduced by header_finish_kernel -- no
// atomics (the first version's six atomics per wave on one cache line cost 134 us per 123k-point scan).
__global__ __launch_bounds__(256) void pack_bbox_kernel(const ScanBuild* __restrict__ sbs, uint32_t* __restrict__ part_all) {
  __shared__ uint32_t red[4][6];
  const ScanBuild& sb = sbs[blockIdx.y];
  const uint32_t n = sb.n, stride = sb.stride;
  const float* __restrict__ in = sb.in;
  float* __restrict__ xyz = sb.xyz;
  uint32_t lo[3] = {0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu}, hi[3] = {0u, 0u, 0u};
  for (uint32_t i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) {
    float v[3];
#pragma unroll
    for (int a = 0; a < 3; ++a) v[a] = in[(size_t)i * stride + a];
#pragma unroll
    for (int a = 0; a < 3; ++a) {
      xyz[(size_t)i * 3 + a] = v[a];
      const uint32_t o = f2ord(v[a]);
      lo[a] = o < lo[a] ? o : lo[a];
      hi[a] = o > hi[a] ? o : hi[a];
    }
  }
#pragma unroll
  for (int a = 0; a < 3; ++a)
    for (int o = 32; o > 0; o >>= 1) {
      const uint32_t l2 = __shfl_xor(lo[a], o), h2 = __shfl_xor(hi[a], o);
      lo[a] = l2 < lo[a] ? l2 : lo[a];
      hi[a] = h2 > hi[a] ? h2 : hi[a];
    }
  const int w = threadIdx.x >> 6;
  if ((threadIdx.x & 63) == 0) {
#pragma unroll
    for (int a = 0; a < 3; ++a) {
      red[w][a] = lo[a];
      red[w][3 + a] = hi[a];
    }
  }
  __syncthreads();
  if (threadIdx.x < 6) {
    uint32_t v = red[0][threadIdx.x];
    for (int ww = 1; ww < 4; ++ww) {
      const uint32_t x = red[ww][threadIdx.x];
      v = threadIdx.x < 3 ? (x < v ? x : v) : (x > v ? x : v);
    }
    part_all[((size_t)blockIdx.y * PACK_BLOCKS + blockIdx.x) * 6 + threadIdx.x] = v;
  }
}

// one wave per scan: reduce the partial boxes, derive the key grid (an empty scan gets a default header)
__global__ __launch_bounds__(64) void header_finish_kernel(const ScanBuild* __restrict__ sbs, const uint32_t* __restrict__ part_all) {
  const ScanBuild& sb = sbs[blockIdx.x];
  ScanHeader* hdr = sb.hdr;
  const uint32_t* part = part_all + (size_t)blockIdx.x * PACK_BLOCKS * 6;
  uint32_t lo[3] = {0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu}, hi[3] = {0u, 0u, 0u};
  if (sb.n)
    for (uint32_t b = threadIdx.x; b < PACK_BLOCKS; b += 64)
#pragma unroll
      for (int a = 0; a < 3; ++a) {
        const uint32_t l = part[b * 6 + a], h = part[b * 6 + 3 + a];
        lo[a] = l < lo[a] ? l : lo[a];
        hi[a] = h > hi[a] ? h : hi[a];
      }
#pragma unroll
  for (int a = 0; a < 3; ++a)
    for (int o = 32; o > 0; o >>= 1) {
      const uint32_t l2 = __shfl_xor(lo[a], o), h2 = __shfl_xor(hi[a], o);
      lo[a] = l2 < lo[a] ? l2 : lo[a];
      hi[a] = h2 > hi[a] ? h2 : hi[a];
    }
  if (threadIdx.x != 0) return;
  for (int a = 0; a < 3; ++a) {
    hdr->lo[a] = lo[a];
    hdr->hi[a] = hi[a];
  }
  if (!sb.n) {
    hdr->ox = hdr->oy = hdr->oz = 0.f;
    hdr->inv_cell = 4.f;
    return;
  }
  const float mn0 = ord2f(lo[0]), mn1 = ord2f(lo[1]), mn2 = ord2f(lo[2]);
  const float e0 = ord2f(hi[0]) - mn0, e1 = ord2f(hi[1]) - mn1, e2 = ord2f(hi[2]) - mn2;
  const float ext = fmaxf(fmaxf(e0, e1), e2);
  const float cell = fmaxf(0.25f, ext / 1023.0f);
  hdr->ox = mn0;
  hdr->oy = mn1;
  hdr->oz = mn2;
  hdr->inv_cell = 1.0f / cell;
}

__global__ void curve_keys_kernel(const ScanBuild* __restrict__ sbs, uint32_t* __restrict__ keys, uint32_t* __restrict__ vals) {
  const ScanBuild& sb = sbs[blockIdx.y];
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= sb.n) return;
  const float* xyz = sb.xyz;
  const ScanHeader* hdr = sb.hdr;
  keys[sb.key_off + i] = morton_key(xyz[3 * (size_t)i], xyz[3 * (size_t)i + 1], xyz[3 * (size_t)i + 2], hdr->ox, hdr->oy,
                                    hdr->oz, hdr->inv_cell);
  vals[sb.key_off + i] = i;
}



__global__ void gather_sorted_kernel(const ScanBuild* __restrict__ sbs, const uint32_t* __restrict__ skeys,
                                     const uint32_t* __restrict__ perm) {
  const ScanBuild& sb = sbs[blockIdx.y];
  const uint32_t s = blockIdx.x * blockDim.x + threadIdx.x;
  if (s >= sb.n_pad) return;
  if (s >= sb.n) {  // padding of the last chunk: farther than any real point, an index that never wins a tie
    sb.pts[s] = f32x4{NN_FAR, NN_FAR, NN_FAR, __uint_as_float(0xFFFFFFFFu)};
    return;
  }
  const uint32_t o = perm[sb.key_off + s];
  const float* xyz = sb.xyz;
  sb.pts[s] = f32x4{xyz[3 * (size_t)o], xyz[3 * (size_t)o + 1], xyz[3 * (size_t)o + 2], __uint_as_float(o)};
  sb.inv[o] = s;
  sb.keys[s] = skeys[sb.key_off + s];
}

// one wave per chunk
__global__ __launch_bounds__(64) void chunk_boxes_kernel(const ScanBuild* __restrict__ sbs) {
  const ScanBuild& sb = sbs[blockIdx.y];
  const uint32_t c = blockIdx.x, lane = threadIdx.x;
  if (c >= sb.nch) return;
  const f32x4* __restrict__ pts = sb.pts;
  const uint32_t n = sb.n;
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
    // Stored as CENTRE and NEGATED HALF EXTENT / 64 (scan_index.hpp: SB2_*), like the sub-block boxes since round 3 and
    // for the same reason: a point's (or a box's) distance to the box along an axis is one v_fma_f32 with the abs and
    // clamp modifiers, clamp(|p - c| / 64 + nh).  The half extent is rounded up so that [c - h, c + h] contains the box
    // whatever the rounding of c.
    float cc[3], nh[3];
    for (int a = 0; a < 3; ++a) {
      const float c_ = 0.5f * mn[a] + 0.5f * mx[a];
      const float he = fmaxf(c_ - mn[a], mx[a] - c_) * 1.0000005f + 1.0e-30f;  // (each difference within 2^-24 of exact)
      cc[a] = c_;
      nh[a] = -he * SB2_INV_RANGE;
    }
    sb.lo[c] = f32x4{cc[0], cc[1], cc[2], 0.f};
    sb.hi[c] = f32x4{nh[0], nh[1], nh[2], 0.f};
  }
}

// one wave per super-chunk: the union of 64 chunk boxes
__global__ __launch_bounds__(64) void super_boxes_kernel(const ScanBuild* __restrict__ sbs) {
  const ScanBuild& sb = sbs[blockIdx.y];
  if (blockIdx.x >= sb.nsup) return;
  const uint32_t c = blockIdx.x * 64 + threadIdx.x;
  float mn[3] = {3.4e38f, 3.4e38f, 3.4e38f}, mx[3] = {-3.4e38f, -3.4e38f, -3.4e38f};
  if (c < sb.nch) {
    // the chunk's box back from (centre, -half extent / 64), widened by more than the rounding of the two operations
    const f32x4 a = sb.lo[c], b = sb.hi[c];
    const float cc[3] = {a.x, a.y, a.z}, he[3] = {-b.x * SB2_RANGE, -b.y * SB2_RANGE, -b.z * SB2_RANGE};
    for (int k = 0; k < 3; ++k) {
      const float slack = 2.0e-7f * (fabsf(cc[k]) + he[k]);
      mn[k] = (cc[k] - he[k]) - slack;
      mx[k] = (cc[k] + he[k]) + slack;
    }
  }
  for (int o = 32; o > 0; o >>= 1)
    for (int a = 0; a < 3; ++a) {
      mn[a] = fminf(mn[a], __shfl_xor(mn[a], o));
      mx[a] = fmaxf(mx[a], __shfl_xor(mx[a], o));
    }
  if (threadIdx.x == 0) {
    sb.ulo[blockIdx.x] = f32x4{mn[0], mn[1], mn[2], 0.f};
    sb.uhi[blockIdx.x] = f32x4{mx[0], mx[1], mx[2], 0.f};
  }
}

// one thread per PAIR of sub-blocks; a sub-block past the end of the scan gets an empty (inverted) box
__global__ void subblock_boxes_kernel(const ScanBuild* __restrict__ sbs) {
  const ScanBuild& sb = sbs[blockIdx.y];
  const uint32_t b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= sb.npairs) return;
  const f32x4* __restrict__ pts = sb.pts;
  const uint32_t n = sb.n;
  float mn[2][3], mx[2][3];
  for (int h = 0; h < 2; ++h) {
    for (int a = 0; a < 3; ++a) {
      mn[h][a] = 3.4e38f;
      mx[h][a] = -3.4e38f;
    }
    for (uint32_t t = 0; t < SB; ++t) {
      const uint32_t j = (2 * b + h) * SB + t;
      if (j < n) {
        const f32x4 p = pts[j];
        mn[h][0] = fminf(mn[h][0], p.x); mx[h][0] = fmaxf(mx[h][0], p.x);
        mn[h][1] = fminf(mn[h][1], p.y); mx[h][1] = fmaxf(mx[h][1], p.y);
        mn[h][2] = fminf(mn[h][2], p.z); mx[h][2] = fmaxf(mx[h][2], p.z);
      }
    }
  }
  // Stored as CENTRE and NEGATED HALF EXTENT / 64 (scan_index.hpp: SB2_*), two sub-blocks interleaved for packed fp32:
  // the search's test is e = clamp(|p - c| / 64 - h / 64) per axis -- a packed subtract and one v_fma_f32 with the abs
  // and clamp modifiers (2.5 cycles) where max(lo - p, p - hi, 0) took two subtracts and a v_max3_f32 (4.3).  The half
  // extent is rounded up so that [c - h, c + h] contains the box whatever the rounding of c; an empty sub-block gets a
  // half extent of -1e30 (every point is "64 m away or more").
  float c_[2][3], nh[2][3];
  for (int h = 0; h < 2; ++h)
    for (int a = 0; a < 3; ++a) {
      if (mn[h][a] <= mx[h][a]) {
        const float c = 0.5f * mn[h][a] + 0.5f * mx[h][a];
        const float he = fmaxf(c - mn[h][a], mx[h][a] - c) * 1.0000005f + 1.0e-30f;  // (each difference within 2^-24 of exact)
        c_[h][a] = c;
        nh[h][a] = -he * SB2_INV_RANGE;
      } else {
        c_[h][a] = 0.f;
        nh[h][a] = 1.0e30f;
      }
    }
  f32x4* sb2 = sb.sb2;
  sb2[3 * (size_t)b + 0] = f32x4{c_[0][0], c_[1][0], c_[0][1], c_[1][1]};
  sb2[3 * (size_t)b + 1] = f32x4{c_[0][2], c_[1][2], nh[0][0], nh[1][0]};
  sb2[3 * (size_t)b + 2] = f32x4{nh[0][1], nh[1][1], nh[0][2], nh[1][2]};
}

// Spatial extent of every group of `group` consecutive sorted points: the squared diagonal of its bounding
// box.  A wave's sweep cost grows with the extent of its sources (more chunk boxes pass the wave-level test),
// so launching the widest groups first keeps the stragglers off the tail.  One wave per group.  The sort key
// is the complement of the extent's order-preserving integer: ascending keys = widest first, and equal
// extents keep ascending group ids (the sort is stable).
__global__ __launch_bounds__(256) void group_extent_kernel(const ScanBuild* __restrict__ sbs, uint32_t* __restrict__ keys,
                                                            uint32_t* __restrict__ ids) {
  const ScanBuild& sb = sbs[blockIdx.y];
  const uint32_t g = blockIdx.x * 4 + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (g >= sb.n_groups) return;
  const f32x4* __restrict__ pts = sb.pts;
  const uint32_t n = sb.n, group = sb.group;
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
    keys[sb.grp_off + g] = ~f2ord(dx * dx + dy * dy + dz * dz);
    ids[sb.grp_off + g] = g;
  }
}

__global__ void copy_order_kernel(const ScanBuild* __restrict__ sbs, const uint32_t* __restrict__ ids) {
  const ScanBuild& sb = sbs[blockIdx.y];
  const uint32_t g = blockIdx.x * blockDim.x + threadIdx.x;
  if (g < sb.n_groups) sb.order[g] = ids[sb.grp_off + g];
}


// Synthetic variant of a resident scan (bench / tests): out_i = T p_i + sigma * gauss(key, 3 i + a),
// the transform in the fixed un-fused fp32 order of xform(), the noise from the counter RNG of
// synth_kernels.hpp -- gloc3d_amd/synth.py::scan_variant produces the same bits.
__global__ void scan_variant_kernel(const float* __restrict__ xyz, uint32_t n, const float* __restrict__ T12,
                                    float sigma, uint64_t key, float* __restrict__ out) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  float T[12];
#pragma unroll
  for (int k = 0; k < 12; ++k) T[k] = T12[k];
  float x, y, z;
  xform(T, xyz[3 * (size_t)i], xyz[3 * (size_t)i + 1], xyz[3 * (size_t)i + 2], x, y, z);
  const float nx = sigma * synth::rng_gauss(key, 3ull * i + 0);
  const float ny = sigma * synth::rng_gauss(key, 3ull * i + 1);
  const float nz = sigma * synth::rng_gauss(key, 3ull * i + 2);
  out[3 * (size_t)i + 0] = x + nx;
  out[3 * (size_t)i + 1] = y + ny;
  out[3 * (size_t)i + 2] = z + nz;
}

// ---- kd order (target index): see scan_index.hpp -------------------------------------------------------
// The scan is already in curve order.  P = 16 * 2^L >= n positions form a complete binary tree; at level
// l a node is the aligned block of (P >> l) positions.  Level by level (top down) every node's points are
// sorted along the widest axis of their bounding box, so that its lower half -- the left child -- holds the
// smaller coordinates: one segmented radix sort per level by (node, coordinate), stable, so that equal
// coordinates keep their current (deterministic) order.  Positions >= n are never materialised: they stand
// for points at +infinity, which stay at the end of their node under every sort.  All scans of a batch go
// through every level together (a scan with fewer levels keeps its order once it is done).
struct ScanKd {
  f32x4* pts;      // the scan's sorted points (read first, written back last)
  uint32_t* inv;
  uint32_t* kpos;
  uint32_t n, n_pad, levels;
  uint32_t off;      // slice of the concatenated working arrays
  uint32_t box_off;  // slice of the node boxes (6 words per node)
  uint32_t pad_;
};

__global__ void kd_load_kernel(const ScanKd* __restrict__ ks, f32x4* __restrict__ p, uint32_t* __restrict__ h) {
  const ScanKd& k = ks[blockIdx.y];
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= k.n) return;
  p[k.off + i] = k.pts[i];
  h[k.off + i] = i;  // its position in curve order
}

__global__ void kd_box_init_kernel(const ScanKd* __restrict__ ks, uint32_t level, uint32_t* __restrict__ box) {
  const ScanKd& k = ks[blockIdx.y];
  const uint32_t nd = blockIdx.x * blockDim.x + threadIdx.x;
  if (level >= k.levels || nd >= (1u << level)) return;
  uint32_t* b = box + k.box_off + 6 * (size_t)nd;
#pragma unroll
  for (int a = 0; a < 3; ++a) {
    b[a] = 0xFFFFFFFFu;
    b[3 + a] = 0u;
  }
}

// one thread per block of 16 points; a wave's 64 blocks lie in ONE node when the node holds >= 1024 positions
__global__ __launch_bounds__(256) void kd_node_bbox_kernel(const ScanKd* __restrict__ ks, uint32_t level,
                                                           const f32x4* __restrict__ p_all, uint32_t* __restrict__ box) {
  const ScanKd& k = ks[blockIdx.y];
  if (level >= k.levels) return;
  const uint32_t shift = 4 + k.levels - level;  // log2 of the node size (SB = 16 = 2^4 points per leaf)
  const uint32_t n = k.n;
  const f32x4* __restrict__ p = p_all + k.off;
  const uint32_t b = blockIdx.x * blockDim.x + threadIdx.x;
  const uint32_t i0 = b * 16u;
  uint32_t lo[3] = {0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu}, hi[3] = {0u, 0u, 0u};
  if (i0 < n) {
    const uint32_t i1 = i0 + 16u < n ? i0 + 16u : n;
    for (uint32_t i = i0; i < i1; ++i) {
      const f32x4 v = p[i];
      const uint32_t o[3] = {f2ord(v.x), f2ord(v.y), f2ord(v.z)};
#pragma unroll
      for (int a = 0; a < 3; ++a) {
        lo[a] = o[a] < lo[a] ? o[a] : lo[a];
        hi[a] = o[a] > hi[a] ? o[a] : hi[a];
      }
    }
  }
  uint32_t first = i0;  // a position inside this thread's node
  if (shift >= 10) {    // wave-uniform node: combine first, one set of atomics per wave
#pragma unroll
    for (int a = 0; a < 3; ++a)
      for (int o = 32; o > 0; o >>= 1) {
        const uint32_t l2 = __shfl_xor(lo[a], o), h2 = __shfl_xor(hi[a], o);
        lo[a] = l2 < lo[a] ? l2 : lo[a];
        hi[a] = h2 > hi[a] ? h2 : hi[a];
      }
    first = (b & ~63u) * 16u;
    if ((threadIdx.x & 63) != 0) return;
  }
  if (lo[0] > hi[0]) return;  // no point in this block / wave
  uint32_t* o = box + k.box_off + 6 * (size_t)(first >> shift);
#pragma unroll
  for (int a = 0; a < 3; ++a) {
    atomicMin(o + a, lo[a]);
    atomicMax(o + 3 + a, hi[a]);
  }
}

__global__ void kd_keys_kernel(const ScanKd* __restrict__ ks, uint32_t level, const f32x4* __restrict__ p_all,
                               const uint32_t* __restrict__ box, unsigned long long* __restrict__ keys,
                               uint32_t* __restrict__ vals) {
  const ScanKd& k = ks[blockIdx.y];
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= k.n) return;
  vals[k.off + i] = i;
  if (level >= k.levels) {  // this scan is done: keep its order
    keys[k.off + i] = i;
    return;
  }
  const uint32_t shift = 4 + k.levels - level;
  const uint32_t node = i >> shift;
  const uint32_t* b = box + k.box_off + 6 * (size_t)node;
  const float ex = ord2f(b[3]) - ord2f(b[0]), ey = ord2f(b[4]) - ord2f(b[1]), ez = ord2f(b[5]) - ord2f(b[2]);
  int axis = 0;  // the widest axis; ties -> the lower axis
  float e = ex;
  if (ey > e) { axis = 1; e = ey; }
  if (ez > e) axis = 2;
  const f32x4 v = p_all[k.off + i];
  const float c = axis == 0 ? v.x : (axis == 1 ? v.y : v.z);
  keys[k.off + i] = ((unsigned long long)node << 32) | f2ord(c);
}

__global__ void kd_gather_kernel(const ScanKd* __restrict__ ks, const f32x4* __restrict__ p_in,
                                 const uint32_t* __restrict__ h_in, const uint32_t* __restrict__ perm,
                                 f32x4* __restrict__ p_out, uint32_t* __restrict__ h_out) {
  const ScanKd& k = ks[blockIdx.y];
  const uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= k.n) return;
  const uint32_t s = perm[k.off + j];
  p_out[k.off + j] = p_in[k.off + s];
  h_out[k.off + j] = h_in[k.off + s];
}

// the re-sorted points back into the scan: pts (kd order, padded), inv (original index -> kd position), kpos
// (curve position -> kd position)
__global__ void kd_finish_kernel(const ScanKd* __restrict__ ks, const f32x4* __restrict__ p, const uint32_t* __restrict__ h) {
  const ScanKd& k = ks[blockIdx.y];
  const uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= k.n_pad) return;
  if (j >= k.n) {
    k.pts[j] = f32x4{NN_FAR, NN_FAR, NN_FAR, __uint_as_float(0xFFFFFFFFu)};
    return;
  }
  const f32x4 v = p[k.off + j];
  k.pts[j] = v;
  k.inv[__float_as_uint(v.w)] = j;
  k.kpos[h[k.off + j]] = j;
}

namespace {

struct Layout {
  size_t bytes, n1, np, c1, b1, u1, g1;
};
Layout layout_for(size_t n) {
  Layout L;
  const size_t nch = (n + CH - 1) / CH, nsup = (nch + 63) / 64;
  L.n1 = std::max<size_t>(n, 1);
  L.c1 = std::max<size_t>(nch, 1);
  // the sorted points and the sub-block boxes are padded to whole chunks (far sentinels / empty boxes),
  // so that the search kernel stages a chunk and re-reads a sub-block without bounds checks
  L.np = L.c1 * CH;
  L.b1 = L.c1 * (CH / SB / 2) * 3;  // float4 per scan of the paired sub-block boxes
  L.u1 = std::max<size_t>(nsup, 1);
  L.g1 = (L.n1 + 63) / 64;
  // header | pts4 | box_lo | box_hi | sb2 | sup_lo | sup_hi | xyz | keys | inv | order (cs = 1 | 2 | 4) | kpos
  L.bytes = sizeof(ScanHeader) + sizeof(f32x4) * (L.np + 2 * L.c1 + L.b1 + 2 * L.u1) +
            sizeof(float) * 3 * L.n1 + sizeof(uint32_t) * (3 * L.n1 + 2 * L.g1 + 2);
  return L;
}

int take_block(gloc_scan_store* st, size_t bytes, void** out, size_t* cap) {
  // the smallest cached allocation that fits without wasting more than a third
  auto it = st->free_blocks.lower_bound(bytes);
  if (it != st->free_blocks.end() && it->first <= bytes + bytes / 2 + (256u << 10)) {
    *out = it->second;
    *cap = it->first;
    st->cached_bytes -= it->first;
    st->free_blocks.erase(it);
    return GLOC_OK;
  }
  const size_t want = (bytes + (256u << 10) - 1) & ~(size_t)((256u << 10) - 1);  // 256 KiB granules: reuse
  hipError_t e = hipMalloc(out, want);
  if (e == hipErrorOutOfMemory && !st->free_blocks.empty()) {
    // a store near HBM capacity: give the parked allocations back before reporting failure
    (void)hipGetLastError();
    (void)hipStreamSynchronize(st->stream);
    for (auto& kv : st->free_blocks) (void)hipFree(kv.second);
    st->free_blocks.clear();
    st->cached_bytes = 0;
    e = hipMalloc(out, want);
  }
  if (e != hipSuccess) {
    (void)hipGetLastError();
    set_err("hipMalloc of %zu bytes for a scan failed: %s", want, hipGetErrorString(e));
    return e == hipErrorOutOfMemory ? GLOC_ERR_NOMEM : GLOC_ERR_HIP;
  }
  *cap = want;
  return GLOC_OK;
}

ScanBuild build_desc(const DevScan& s, const Layout& L) {
  ScanBuild b{};
  b.hdr = const_cast<ScanHeader*>(s.idx.hdr);
  b.xyz = s.xyz;
  b.pts = const_cast<f32x4*>(s.idx.pts);
  b.lo = const_cast<f32x4*>(s.idx.box_lo);
  b.hi = const_cast<f32x4*>(s.idx.box_hi);
  b.sb2 = const_cast<f32x4*>(s.idx.sb2);
  b.ulo = const_cast<f32x4*>(s.idx.sup_lo);
  b.uhi = const_cast<f32x4*>(s.idx.sup_hi);
  b.keys = const_cast<uint32_t*>(s.idx.keys);
  b.inv = const_cast<uint32_t*>(s.idx.inv);
  b.n = (uint32_t)s.n;
  b.stride = 3;
  b.n_pad = (uint32_t)L.np;
  b.nch = s.idx.nchunks;
  b.nsup = s.idx.nsup;
  b.npairs = (uint32_t)(L.b1 / 3);
  return b;
}

// boxes of all levels from the sorted points (whatever order they are in)
void launch_boxes(hipStream_t q, const ScanBuild* d_sb, uint32_t count, uint32_t max_nch, uint32_t max_npairs,
                  uint32_t max_nsup) {
  if (!max_nch) return;
  hipLaunchKernelGGL(chunk_boxes_kernel, dim3(max_nch, count), dim3(64), 0, q, d_sb);
  hipLaunchKernelGGL(subblock_boxes_kernel, dim3((max_npairs + 255) / 256, count), dim3(256), 0, q, d_sb);
  hipLaunchKernelGGL(super_boxes_kernel, dim3(max_nsup, count), dim3(64), 0, q, d_sb);
}

// launch orders for `cs` sources per lane of scans whose descriptors (with order / n_groups / group / grp_off set)
// are already on the device; total_groups = sum of n_groups
int launch_orders(gloc_scan_store* st, const ScanBuild* d_sb, const std::vector<ScanBuild>& hb, uint32_t count) {
  hipStream_t q = st->stream;
  uint32_t total = 0, max_ng = 0;
  std::vector<segsort::Seg> segs(count);
  for (uint32_t i = 0; i < count; ++i) {
    segs[i] = segsort::Seg{hb[i].grp_off, hb[i].n_groups};
    total = std::max(total, hb[i].grp_off + hb[i].n_groups);
    max_ng = std::max(max_ng, hb[i].n_groups);
  }
  if (!max_ng) return GLOC_OK;
  GLOC_TRY(st->grp_k0.ensure(sizeof(uint32_t) * total, q));
  GLOC_TRY(st->grp_k1.ensure(sizeof(uint32_t) * total, q));
  GLOC_TRY(st->grp_v0.ensure(sizeof(uint32_t) * total, q));
  GLOC_TRY(st->grp_v1.ensure(sizeof(uint32_t) * total, q));
  GLOC_TRY(st->grp_segs.ensure(sizeof(segsort::Seg) * count, q));
  GLOC_TRY(st->sort_hist.ensure(segsort::scratch_bytes(count, max_ng), q));
  GLOC_HIP(hipMemcpyAsync(st->grp_segs.p, segs.data(), sizeof(segsort::Seg) * count, hipMemcpyHostToDevice, q));
  hipLaunchKernelGGL(group_extent_kernel, dim3((max_ng + 3) / 4, count), dim3(256), 0, q, d_sb, st->grp_k0.as<uint32_t>(),
                     st->grp_v0.as<uint32_t>());
  if (max_ng <= (uint32_t)segsort::SMALL_MAX) {  // one launch: a work-group per scan sorts its groups in LDS
    hipLaunchKernelGGL(segsort::small_sort_kernel, dim3(count), dim3(1024), 0, q, st->grp_k0.as<uint32_t>(),
                       st->grp_v0.as<uint32_t>(), st->grp_segs.as<segsort::Seg>(), st->grp_v1.as<uint32_t>());
    hipLaunchKernelGGL(copy_order_kernel, dim3((max_ng + 255) / 256, count), dim3(256), 0, q, d_sb, st->grp_v1.as<uint32_t>());
  } else {
    const int r = segsort::sort_pairs<uint32_t, 8>(q, st->grp_k0.as<uint32_t>(), st->grp_k1.as<uint32_t>(),
                                                    st->grp_v0.as<uint32_t>(), st->grp_v1.as<uint32_t>(),
                                                    st->grp_segs.as<segsort::Seg>(), count, max_ng, 0, 32,
                                                    st->sort_hist.as<uint32_t>());
    hipLaunchKernelGGL(copy_order_kernel, dim3((max_ng + 255) / 256, count), dim3(256), 0, q, d_sb,
                       r ? st->grp_v1.as<uint32_t>() : st->grp_v0.as<uint32_t>());
  }
  GLOC_HIP(hipGetLastError());
  GLOC_HIP(hipStreamSynchronize(q));  // (segs is a local: the copy must have been consumed)
  return GLOC_OK;
}

}  // namespace

void store_free_scan(gloc_scan_store* st, DevScan& s, bool cache_block) {
  if (s.block) {
    if (cache_block && st->cached_bytes + s.block_bytes <= (size_t(1) << 30)) {
      st->free_blocks.emplace(s.block_bytes, s.block);
      st->cached_bytes += s.block_bytes;
    } else {
      (void)hipFree(s.block);
    }
  }
  s = DevScan{};
}

int store_build_order(gloc_scan_store* st, DevScan& s, int cs) {
  if (cs == 0) {
    s.order = nullptr;
    return GLOC_OK;
  }
  if (cs != 1 && cs != 2 && cs != 4) {
    set_err("sources per lane must be 1, 2 or 4 (got %d)", cs);
    return GLOC_ERR_INVALID;
  }
  s.order = s.order_of(cs);
  if ((s.order_built & (1u << cs)) || s.n == 0) return GLOC_OK;
  hipStream_t q = st->stream;
  std::vector<ScanBuild> hb(1, build_desc(s, layout_for(s.n)));
  hb[0].group = 64u * (uint32_t)cs;
  hb[0].n_groups = (uint32_t)((s.n + hb[0].group - 1) / hb[0].group);
  hb[0].order = s.order;
  GLOC_TRY(st->builds.ensure(sizeof(ScanBuild), q));
  GLOC_HIP(hipMemcpyAsync(st->builds.p, hb.data(), sizeof(ScanBuild), hipMemcpyHostToDevice, q));
  GLOC_TRY(launch_orders(st, st->builds.as<ScanBuild>(), hb, 1));
  s.order_built |= 1u << cs;
  return GLOC_OK;
}

// Index `count` scans in one launch sequence.  out[i] receives scan i (live = false until the caller inserts it);
// on failure everything allocated here is released.  Returns after the work has completed on the store's stream.
int store_make_scans(gloc_scan_store* st, size_t count, const float* const* pts, const size_t* n, size_t stride,
                     bool device_src, DevScan* out) {
  if (!count) return GLOC_OK;
  hipStream_t q = st->stream;
  std::vector<ScanBuild> hb(count);
  std::vector<segsort::Seg> segs(count);
  size_t made = 0;
  auto fail = [&](int code) {
    (void)hipStreamSynchronize(q);
    for (size_t i = 0; i < made; ++i) store_free_scan(st, out[i], true);
    return code;
  };
  size_t stage_floats = 0, total_pts = 0, total_grp = 0;
  uint32_t max_n = 0, max_np = 0, max_nch = 0, max_npairs = 0, max_nsup = 0;
  for (size_t i = 0; i < count; ++i) {
    DevScan s;
    s.n = n[i];
    const Layout L = layout_for(n[i]);
    const size_t nch = (n[i] + CH - 1) / CH, nsup = (nch + 63) / 64;
    if (int rc = take_block(st, L.bytes, &s.block, &s.block_bytes)) return fail(rc);
    ScanHeader* hdr = reinterpret_cast<ScanHeader*>(s.block);
    f32x4* p4 = reinterpret_cast<f32x4*>(hdr + 1);
    f32x4* lo = p4 + L.np;
    f32x4* hi = lo + L.c1;
    f32x4* sb2 = hi + L.c1;
    f32x4* ulo = sb2 + L.b1;
    f32x4* uhi = ulo + L.u1;
    s.xyz = reinterpret_cast<float*>(uhi + L.u1);
    uint32_t* keys = reinterpret_cast<uint32_t*>(s.xyz + 3 * L.n1);
    uint32_t* inv = keys + L.n1;
    s.order_base = inv + L.n1;
    s.order_g1 = L.g1;
    s.order = nullptr;
    s.kpos_mem = s.order_base + 2 * L.g1 + 2;
    s.idx = ScanIndexDev{p4, lo, hi, sb2, nullptr, keys, inv, hdr, ulo, uhi, (uint32_t)n[i], (uint32_t)nch,
                         (uint32_t)nsup, 0u};
    out[i] = s;
    made = i + 1;
    ScanBuild b = build_desc(s, L);
    b.stride = (uint32_t)stride;
    b.group = 128;  // the default sources-per-lane setting (2)
    b.n_groups = (uint32_t)((n[i] + b.group - 1) / b.group);
    b.order = s.order_of(2);
    b.key_off = (uint32_t)total_pts;
    b.grp_off = (uint32_t)total_grp;
    segs[i] = segsort::Seg{b.key_off, b.n};
    total_pts += (n[i] + 3) & ~(size_t)3;
    total_grp += b.n_groups;
    if (total_pts >= (1ull << 31)) {
      set_err("batch of scans too large (%zu points)", total_pts);
      return fail(GLOC_ERR_INVALID);
    }
    stage_floats += device_src ? 0 : stride * n[i];
    max_n = std::max(max_n, b.n);
    max_np = std::max(max_np, b.n_pad);
    max_nch = std::max(max_nch, b.nch);
    max_npairs = std::max(max_npairs, b.npairs);
    max_nsup = std::max(max_nsup, b.nsup);
    hb[i] = b;
  }
  // sources: device pointers as they are; host points travel as they are (stride included) into a staging
  // buffer, the pack kernel drops the extra channels on the device
  if (!device_src) {
    if (int rc = st->stage.ensure(sizeof(float) * std::max<size_t>(stage_floats, 4), q)) return fail(rc);
    size_t off = 0;
    for (size_t i = 0; i < count; ++i) {
      hb[i].in = st->stage.as<float>() + off;
      if (n[i]) {
        const hipError_t eu = hipMemcpyAsync(st->stage.as<float>() + off, pts[i], sizeof(float) * stride * n[i],
                                             hipMemcpyHostToDevice, q);
        if (eu != hipSuccess) {
          (void)hipGetLastError();
          set_err("scan upload failed: %s", hipGetErrorString(eu));
          return fail(GLOC_ERR_HIP);
        }
      }
      off += stride * n[i];
    }
  } else {
    for (size_t i = 0; i < count; ++i) hb[i].in = pts[i];
  }
  const size_t tp = std::max<size_t>(total_pts, 4);
  for (auto need : {std::make_pair(&st->sort_keys, sizeof(uint32_t) * tp), std::make_pair(&st->sort_keys2, sizeof(uint32_t) * tp),
                    std::make_pair(&st->sort_vals, sizeof(uint32_t) * tp), std::make_pair(&st->sort_perm, sizeof(uint32_t) * tp),
                    std::make_pair(&st->part, sizeof(uint32_t) * 6 * PACK_BLOCKS * count),
                    std::make_pair(&st->builds, sizeof(ScanBuild) * count), std::make_pair(&st->segs, sizeof(segsort::Seg) * count),
                    std::make_pair(&st->sort_hist, segsort::scratch_bytes((uint32_t)count, std::max<uint32_t>(max_n, 1)))})
    if (int rc = need.first->ensure(need.second, q)) return fail(rc);  // (the actual code: NOMEM or a HIP error)
  hipError_t e = hipMemcpyAsync(st->builds.p, hb.data(), sizeof(ScanBuild) * count, hipMemcpyHostToDevice, q);
  if (e == hipSuccess) e = hipMemcpyAsync(st->segs.p, segs.data(), sizeof(segsort::Seg) * count, hipMemcpyHostToDevice, q);
  if (e != hipSuccess) {
    (void)hipGetLastError();
    set_err("scan indexing: descriptor upload failed: %s", hipGetErrorString(e));
    return fail(GLOC_ERR_HIP);
  }
  const ScanBuild* d_sb = st->builds.as<ScanBuild>();
  const uint32_t cnt = (uint32_t)count;
  if (max_n) hipLaunchKernelGGL(pack_bbox_kernel, dim3(PACK_BLOCKS, cnt), dim3(256), 0, q, d_sb, st->part.as<uint32_t>());
  hipLaunchKernelGGL(header_finish_kernel, dim3(cnt), dim3(64), 0, q, d_sb, st->part.as<uint32_t>());
  if (max_n) {
    hipLaunchKernelGGL(curve_keys_kernel, dim3((max_n + 255) / 256, cnt), dim3(256), 0, q, d_sb, st->sort_keys.as<uint32_t>(),
                       st->sort_vals.as<uint32_t>());
    const int r = segsort::sort_pairs<uint32_t, 8>(q, st->sort_keys.as<uint32_t>(), st->sort_keys2.as<uint32_t>(),
                                                    st->sort_vals.as<uint32_t>(), st->sort_perm.as<uint32_t>(),
                                                    st->segs.as<segsort::Seg>(), cnt, max_n, 0, 30, st->sort_hist.as<uint32_t>());
    hipLaunchKernelGGL(gather_sorted_kernel, dim3((max_np + 255) / 256, cnt), dim3(256), 0, q, d_sb,
                       r ? st->sort_keys2.as<uint32_t>() : st->sort_keys.as<uint32_t>(),
                       r ? st->sort_perm.as<uint32_t>() : st->sort_vals.as<uint32_t>());
    launch_boxes(q, d_sb, cnt, max_nch, max_npairs, max_nsup);
  }
  const hipError_t ei = hipGetLastError();  // (read once: the call clears the error)
  if (ei != hipSuccess) {
    set_err("scan indexing failed: %s", hipGetErrorString(ei));
    return fail(GLOC_ERR_HIP);
  }
  if (max_n) {
    if (int rc = launch_orders(st, d_sb, hb, cnt)) return fail(rc);  // synchronises the stream
  } else if (hipStreamSynchronize(q) != hipSuccess) {
    return fail(GLOC_ERR_HIP);
  }
  for (size_t i = 0; i < count; ++i) {
    out[i].order_built = n[i] ? (1u << 2) : 0u;
    out[i].live = true;
  }
  return GLOC_OK;
}

int store_make_scan(gloc_scan_store* st, const float* pts, size_t n, size_t stride, bool device_src, DevScan* out) {
  return store_make_scans(st, 1, &pts, &n, stride, device_src, out);
}

// Re-sort `count` indexed scans into kd order (target index).  Scans already in kd order are skipped.
int store_build_target_indices(gloc_scan_store* st, DevScan* const* scans, size_t count) {
  hipStream_t q = st->stream;
  constexpr size_t MAX_BATCH_POINTS = size_t(8) << 20;  // 64 B of scratch per point: 512 MB
  size_t a = 0;
  while (a < count) {
    std::vector<ScanKd> hk;
    std::vector<ScanBuild> hb;
    std::vector<DevScan*> todo;
    std::vector<segsort::Seg> segs;
    size_t total = 0, box_words = 0, total_grp = 0;
    uint32_t max_n = 0, max_np = 0, max_levels = 0, max_nch = 0, max_npairs = 0, max_nsup = 0;
    size_t b = a;
    for (; b < count; ++b) {
      DevScan& s = *scans[b];
      if (s.kd) continue;
      if (s.n <= (size_t)SB) {
        s.kd = true;  // (one sub-block is in kd order as it is)
        continue;
      }
      if (!todo.empty() && total + s.n > MAX_BATCH_POINTS) break;
      uint32_t L = 0;
      while (((size_t)SB << L) < s.n) ++L;  // P = SB * 2^L positions
      const Layout lay = layout_for(s.n);
      ScanKd k{};
      k.pts = const_cast<f32x4*>(s.idx.pts);
      k.inv = const_cast<uint32_t*>(s.idx.inv);
      k.kpos = s.kpos_mem;
      k.n = (uint32_t)s.n;
      k.n_pad = (uint32_t)lay.np;
      k.levels = L;
      k.off = (uint32_t)total;
      k.box_off = (uint32_t)box_words;
      ScanBuild sb = build_desc(s, lay);
      sb.group = 128;
      sb.n_groups = (uint32_t)((s.n + 127) / 128);
      sb.order = s.order_of(2);
      sb.grp_off = (uint32_t)total_grp;
      segs.push_back(segsort::Seg{k.off, k.n});
      total += (s.n + 3) & ~(size_t)3;
      box_words += (size_t)6 << (L - 1);
      total_grp += sb.n_groups;
      max_n = std::max(max_n, k.n);
      max_np = std::max(max_np, k.n_pad);
      max_levels = std::max(max_levels, L);
      max_nch = std::max(max_nch, sb.nch);
      max_npairs = std::max(max_npairs, sb.npairs);
      max_nsup = std::max(max_nsup, sb.nsup);
      hk.push_back(k);
      hb.push_back(sb);
      todo.push_back(&s);
    }
    a = b;
    if (todo.empty()) continue;
    const uint32_t cnt = (uint32_t)todo.size();
    GLOC_TRY(st->kd_k0.ensure(sizeof(unsigned long long) * total, q));
    GLOC_TRY(st->kd_k1.ensure(sizeof(unsigned long long) * total, q));
    GLOC_TRY(st->kd_v0.ensure(sizeof(uint32_t) * total, q));
    GLOC_TRY(st->kd_v1.ensure(sizeof(uint32_t) * total, q));
    GLOC_TRY(st->kd_p0.ensure(sizeof(f32x4) * total, q));
    GLOC_TRY(st->kd_p1.ensure(sizeof(f32x4) * total, q));
    GLOC_TRY(st->kd_h0.ensure(sizeof(uint32_t) * total, q));
    GLOC_TRY(st->kd_h1.ensure(sizeof(uint32_t) * total, q));
    GLOC_TRY(st->kd_box.ensure(sizeof(uint32_t) * box_words, q));
    GLOC_TRY(st->kd_desc.ensure(sizeof(ScanKd) * cnt, q));
    GLOC_TRY(st->builds.ensure(sizeof(ScanBuild) * cnt, q));
    GLOC_TRY(st->segs.ensure(sizeof(segsort::Seg) * cnt, q));
    GLOC_TRY(st->sort_hist.ensure(segsort::scratch_bytes(cnt, max_n), q));
    GLOC_HIP(hipMemcpyAsync(st->kd_desc.p, hk.data(), sizeof(ScanKd) * cnt, hipMemcpyHostToDevice, q));
    GLOC_HIP(hipMemcpyAsync(st->builds.p, hb.data(), sizeof(ScanBuild) * cnt, hipMemcpyHostToDevice, q));
    GLOC_HIP(hipMemcpyAsync(st->segs.p, segs.data(), sizeof(segsort::Seg) * cnt, hipMemcpyHostToDevice, q));
    const ScanKd* d_k = st->kd_desc.as<ScanKd>();
    f32x4* pp[2] = {st->kd_p0.as<f32x4>(), st->kd_p1.as<f32x4>()};
    uint32_t* hh[2] = {st->kd_h0.as<uint32_t>(), st->kd_h1.as<uint32_t>()};
    unsigned long long* kk[2] = {st->kd_k0.as<unsigned long long>(), st->kd_k1.as<unsigned long long>()};
    uint32_t* vv[2] = {st->kd_v0.as<uint32_t>(), st->kd_v1.as<uint32_t>()};
    const dim3 gpt((max_n + 255) / 256, cnt), gblk(((max_n + SB - 1) / SB + 255) / 256, cnt);
    hipLaunchKernelGGL(kd_load_kernel, gpt, dim3(256), 0, q, d_k, pp[0], hh[0]);
    int cur = 0;
    static_assert(SB == 16, "the node size arithmetic assumes 16-point leaves");
    for (uint32_t l = 0; l < max_levels; ++l) {
      hipLaunchKernelGGL(kd_box_init_kernel, dim3(((1u << l) + 255) / 256, cnt), dim3(256), 0, q, d_k, l, st->kd_box.as<uint32_t>());
      hipLaunchKernelGGL(kd_node_bbox_kernel, gblk, dim3(256), 0, q, d_k, l, pp[cur], st->kd_box.as<uint32_t>());
      hipLaunchKernelGGL(kd_keys_kernel, gpt, dim3(256), 0, q, d_k, l, pp[cur], st->kd_box.as<uint32_t>(), kk[0], vv[0]);
      const int r = segsort::sort_pairs<unsigned long long, 8>(q, kk[0], kk[1], vv[0], vv[1], st->segs.as<segsort::Seg>(), cnt,
                                                                max_n, 0, (int)(32 + l), st->sort_hist.as<uint32_t>());
      hipLaunchKernelGGL(kd_gather_kernel, gpt, dim3(256), 0, q, d_k, pp[cur], hh[cur], vv[r], pp[cur ^ 1], hh[cur ^ 1]);
      cur ^= 1;
    }
    // write back and rebuild what depends on the order: inv, kpos, all boxes, the launch orders
    hipLaunchKernelGGL(kd_finish_kernel, dim3((max_np + 255) / 256, cnt), dim3(256), 0, q, d_k, pp[cur], hh[cur]);
    launch_boxes(q, st->builds.as<ScanBuild>(), cnt, max_nch, max_npairs, max_nsup);
    GLOC_HIP(hipGetLastError());
    // From here on the scans ARE in kd order (kd_finish_kernel rewrote their points, inverse permutations and kpos, the
    // boxes follow): the bookkeeping says so before anything else can fail, so that a failure below (the launch orders'
    // scratch) leaves every scan consistent -- cold starts go through kpos, a retry does not re-sort a re-sorted scan,
    // and the launch orders, which listed groups of the old order, are rebuilt on demand.
    for (DevScan* s : todo) {
      s->idx.kpos = s->kpos_mem;
      s->kd = true;
      s->order_built = 0;
    }
    const int rc_orders = launch_orders(st, st->builds.as<ScanBuild>(), hb, cnt);  // synchronises the stream
    if (rc_orders != GLOC_OK) {
      (void)hipStreamSynchronize(q);
      return rc_orders;
    }
    for (DevScan* s : todo) s->order_built = 1u << 2;
  }
  return GLOC_OK;
}

int store_build_target_index(gloc_scan_store* st, DevScan& s) {
  DevScan* p = &s;
  return store_build_target_indices(st, &p, 1);
}

int store_get(gloc_scan_store* st, uint32_t id, int cs, DevScan* out) {
  std::lock_guard<std::mutex> lk(st->mu);
  if (id >= st->scans.size() || !st->scans[id].live) {
    set_err("unknown scan id %u", id);
    return GLOC_ERR_INVALID;
  }
  DevScan& s = st->scans[id];
  GLOC_TRY(store_build_order(st, s, cs));  // builds that cs's own array on first use; never rewrites another
  *out = s;
  out->order = cs ? s.order_of(cs) : nullptr;
  return GLOC_OK;
}

int store_get_pinned(gloc_scan_store* st, const uint32_t* ids, const int* cs, size_t count, DevScan* out) {
  std::lock_guard<std::mutex> lk(st->mu);
  for (size_t i = 0; i < count; ++i)
    if (ids[i] >= st->scans.size() || !st->scans[ids[i]].live) {
      set_err("unknown scan id %u", ids[i]);
      return GLOC_ERR_INVALID;
    }
  for (size_t i = 0; i < count; ++i) GLOC_TRY(store_build_order(st, st->scans[ids[i]], cs[i]));  // (nothing pinned yet if one fails)
  for (size_t i = 0; i < count; ++i) {
    DevScan& s = st->scans[ids[i]];
    s.pins += 1;
    out[i] = s;
    out[i].order = cs[i] ? s.order_of(cs[i]) : nullptr;
  }
  return GLOC_OK;
}

void store_pin(gloc_scan_store* st, const uint32_t* ids, size_t count, int delta) {
  std::lock_guard<std::mutex> lk(st->mu);
  for (size_t i = 0; i < count; ++i)
    if (ids[i] < st->scans.size() && st->scans[ids[i]].live) st->scans[ids[i]].pins += delta;
}

}  // namespace reg
}  // namespace gloc

using namespace gloc;
using namespace gloc::reg;

namespace {

int store_insert(gloc_scan_store* st, const DevScan& s, uint32_t* id) {
  if (!st->free_ids.empty()) {
    *id = st->free_ids.back();
    st->free_ids.pop_back();
    st->scans[*id] = s;
  } else {
    st->scans.push_back(s);
    *id = (uint32_t)(st->scans.size() - 1);
  }
  st->live_count++;
  st->live_bytes += s.block_bytes;
  return GLOC_OK;
}

int add_common(gloc_scan_store* st, const float* pts, size_t n, size_t stride, bool dev, uint32_t* id) {
  GLOC_REQUIRE(st && id && (pts || n == 0), GLOC_ERR_INVALID, "null argument");
  GLOC_REQUIRE(stride >= 3 && stride <= 16, GLOC_ERR_INVALID, "stride_floats = %zu outside [3,16]", stride);
  GLOC_REQUIRE(n < (1ull << 31), GLOC_ERR_INVALID, "scan too large");
  GLOC_HIP(hipSetDevice(st->device));
  std::lock_guard<std::mutex> lk(st->mu);
  DevScan s;
  GLOC_TRY(store_make_scan(st, pts, n, stride, dev, &s));
  return store_insert(st, s, id);
}

}  // namespace

extern "C" {

int gloc_scan_store_create(int device, gloc_scan_store** out) {
  GLOC_REQUIRE(out, GLOC_ERR_INVALID, "out is null");
  *out = nullptr;
  GLOC_TRY(select_device(device));
  gloc_scan_store* st = new (std::nothrow) gloc_scan_store;
  GLOC_REQUIRE(st, GLOC_ERR_NOMEM, "host allocation failed");
  st->device = device;
  hipError_t e = hipStreamCreateWithFlags(&st->stream, hipStreamNonBlocking);
  if (e != hipSuccess) {
    set_err("hipStreamCreate failed: %s", hipGetErrorString(e));
    delete st;
    return GLOC_ERR_HIP;
  }
  *out = st;
  return GLOC_OK;
}

int gloc_scan_store_destroy(gloc_scan_store* st) {
  if (!st) return GLOC_OK;
  GLOC_REQUIRE(st->attached.load() == 0, GLOC_ERR_STATE,
               "%d registration handle(s) still attached to this scan store", st->attached.load());
  (void)hipSetDevice(st->device);
  (void)hipStreamSynchronize(st->stream);
  for (auto& s : st->scans)
    if (s.block) (void)hipFree(s.block);
  for (auto& kv : st->free_blocks) (void)hipFree(kv.second);
  for (DevBuf* b : st->scratch()) b->release();
  (void)hipStreamDestroy(st->stream);
  delete st;
  return GLOC_OK;
}

int gloc_scan_store_add(gloc_scan_store* st, const float* pts, size_t n, size_t stride_floats, uint32_t* scan_id) {
  return add_common(st, pts, n, stride_floats, false, scan_id);
}

int gloc_scan_store_add_device(gloc_scan_store* st, const float* d_pts, size_t n, size_t stride_floats,
                               uint32_t* scan_id) {
  return add_common(st, d_pts, n, stride_floats, true, scan_id);
}

int gloc_scan_store_add_batch(gloc_scan_store* st, const float* const* pts, const size_t* n, size_t count,
                              size_t stride_floats, uint32_t* scan_ids) {
  GLOC_REQUIRE(st && scan_ids && pts && n, GLOC_ERR_INVALID, "null argument");
  GLOC_REQUIRE(count >= 1 && count <= 4096, GLOC_ERR_INVALID, "count = %zu outside [1,4096]", count);
  GLOC_REQUIRE(stride_floats >= 3 && stride_floats <= 16, GLOC_ERR_INVALID, "stride_floats = %zu outside [3,16]", stride_floats);
  for (size_t i = 0; i < count; ++i) {
    GLOC_REQUIRE(pts[i] || n[i] == 0, GLOC_ERR_INVALID, "scan %zu is null", i);
    GLOC_REQUIRE(n[i] < (1ull << 31), GLOC_ERR_INVALID, "scan too large");
  }
  GLOC_HIP(hipSetDevice(st->device));
  std::lock_guard<std::mutex> lk(st->mu);
  std::vector<DevScan> s(count);
  GLOC_TRY(store_make_scans(st, count, pts, n, stride_floats, false, s.data()));
  for (size_t i = 0; i < count; ++i) store_insert(st, s[i], &scan_ids[i]);
  return GLOC_OK;
}

int gloc_scan_store_add_variant(gloc_scan_store* st, uint32_t base_id, const float* T16, float noise_sigma,
                                uint64_t seed, uint32_t* scan_id) {
  GLOC_REQUIRE(st && scan_id, GLOC_ERR_INVALID, "null argument");
  GLOC_HIP(hipSetDevice(st->device));
  std::lock_guard<std::mutex> lk(st->mu);
  GLOC_REQUIRE(base_id < st->scans.size() && st->scans[base_id].live, GLOC_ERR_INVALID, "unknown scan id %u",
               base_id);
  const DevScan base = st->scans[base_id];
  static const float I16[16] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1};
  const float* T = T16 ? T16 : I16;
  float T12[12];
  for (int i = 0; i < 3; ++i) {
    for (int j = 0; j < 3; ++j) T12[3 * i + j] = T[4 * i + j];
    T12[9 + i] = T[4 * i + 3];
  }
  hipStream_t q = st->stream;
  // [T12 | moved points] in the variant scratch (the stage buffer is used by store_make_scan for host uploads only)
  const size_t need = 64 + sizeof(float) * 3 * std::max<size_t>(base.n, 1);
  GLOC_TRY(st->stage.ensure(need, q));
  float* dT = st->stage.as<float>();
  float* dout = dT + 16;
  GLOC_HIP(hipMemcpyAsync(dT, T12, sizeof(T12), hipMemcpyHostToDevice, q));
  if (base.n)
    hipLaunchKernelGGL(scan_variant_kernel, dim3((unsigned)((base.n + 255) / 256)), dim3(256), 0, q, base.xyz,
                       (uint32_t)base.n, dT, noise_sigma, synth::rng_key(seed, 11), dout);
  GLOC_HIP(hipGetLastError());
  DevScan s;
  GLOC_TRY(store_make_scan(st, dout, base.n, 3, true, &s));
  return store_insert(st, s, scan_id);
}

int gloc_scan_store_add_raycast_batch(gloc_scan_store* st, size_t count, const double* box_lo, const double* box_hi,
                                      const uint32_t* box_first, double ground_z, const double* T16, const uint64_t* seeds,
                                      const gloc_raycast_params* prm, uint32_t* scan_ids) {
  GLOC_REQUIRE(st && box_first && T16 && seeds && prm && scan_ids, GLOC_ERR_INVALID, "null argument");
  GLOC_REQUIRE(count >= 1 && count <= 256, GLOC_ERR_INVALID, "count = %zu outside [1,256]", count);
  GLOC_REQUIRE(prm->n_beams >= 1 && prm->n_az >= 1 && (uint64_t)prm->n_beams * prm->n_az <= (1u << 22), GLOC_ERR_INVALID,
               "n_beams x n_az = %u x %u outside [1, 4194304] rays", prm->n_beams, prm->n_az);
  GLOC_REQUIRE(prm->max_range > 0.0 && prm->noise_sigma >= 0.0, GLOC_ERR_INVALID, "max_range must be positive, noise_sigma not negative");
  const size_t n_boxes = box_first[count];
  for (size_t i = 0; i < count; ++i)
    GLOC_REQUIRE(box_first[i] <= box_first[i + 1], GLOC_ERR_INVALID, "box_first is not ascending at scan %zu", i);
  GLOC_REQUIRE(n_boxes == 0 || (box_lo && box_hi), GLOC_ERR_INVALID, "null box arrays");
  GLOC_HIP(hipSetDevice(st->device));
  std::lock_guard<std::mutex> lk(st->mu);
  hipStream_t q = st->stream;
  const uint32_t n_rays = prm->n_beams * prm->n_az;
  const uint32_t n_blocks = (n_rays + raycast::RC_THREADS - 1) / raycast::RC_THREADS;
  const size_t rays_pad = (size_t)n_blocks * raycast::RC_THREADS;
  // host side of the ray table: numpy's linspace (start + i * step, the last one exactly stop) and deg2rad (x * (pi / 180))
  std::vector<double> tab(2 * (size_t)prm->n_beams + 2 * (size_t)prm->n_az);
  {
    const double pi = 3.141592653589793;
    const double e0 = prm->fov_lo_deg, e1 = prm->fov_hi_deg;
    const double estep = prm->n_beams > 1 ? (e1 - e0) / (double)(prm->n_beams - 1) : 0.0;
    for (uint32_t b = 0; b < prm->n_beams; ++b) {
      const double deg = (b + 1 == prm->n_beams && prm->n_beams > 1) ? e1 : e0 + (double)b * estep;
      const double el = deg * (pi / 180.0);
      tab[b] = std::cos(el);
      tab[prm->n_beams + b] = std::sin(el);
    }
    const double astep = (2.0 * pi) / (double)prm->n_az;  // endpoint=False
    for (uint32_t a = 0; a < prm->n_az; ++a) {
      const double az = (double)a * astep;
      tab[2 * (size_t)prm->n_beams + a] = std::cos(az);
      tab[2 * (size_t)prm->n_beams + prm->n_az + a] = std::sin(az);
    }
  }
  std::vector<raycast::RayScan> hs(count);
  for (size_t i = 0; i < count; ++i) {
    const double* T = T16 + 16 * i;
    for (int a = 0; a < 3; ++a) {
      for (int b = 0; b < 3; ++b) hs[i].R[3 * a + b] = T[4 * a + b];
      hs[i].o[a] = T[4 * a + 3];
    }
    hs[i].key = synth::rng_key(seeds[i], 7);
    hs[i].box0 = box_first[i];
    hs[i].box1 = box_first[i + 1];
  }
  // scratch (the stage buffer: store_make_scans uses it for HOST sources only): [tables | boxes lo, hi | scan descriptors |
  // counts, totals | masks | returns by ray | returns compacted]
  auto al = [](size_t b) { return (b + 255) & ~(size_t)255; };
  const size_t o_tab = 0, o_lo = o_tab + al(tab.size() * 8), o_hi = o_lo + al(n_boxes * 24 + 8), o_sc = o_hi + al(n_boxes * 24 + 8);
  const size_t o_cnt = o_sc + al(count * sizeof(raycast::RayScan)), o_tot = o_cnt + al(count * n_blocks * 4);
  const size_t o_msk = o_tot + al(count * 4), o_tmp = o_msk + al(count * rays_pad / 64 * 8), o_out = o_tmp + al(count * rays_pad * 12);
  const size_t need = o_out + al(count * rays_pad * 12);
  GLOC_TRY(st->stage.ensure(need, q));
  char* base = st->stage.as<char>();
  GLOC_HIP(hipMemcpyAsync(base + o_tab, tab.data(), tab.size() * 8, hipMemcpyHostToDevice, q));
  if (n_boxes) {
    GLOC_HIP(hipMemcpyAsync(base + o_lo, box_lo, n_boxes * 24, hipMemcpyHostToDevice, q));
    GLOC_HIP(hipMemcpyAsync(base + o_hi, box_hi, n_boxes * 24, hipMemcpyHostToDevice, q));
  }
  GLOC_HIP(hipMemcpyAsync(base + o_sc, hs.data(), count * sizeof(raycast::RayScan), hipMemcpyHostToDevice, q));
  raycast::RayCfg c{};
  const double* dtab = reinterpret_cast<const double*>(base + o_tab);
  c.ce = dtab;
  c.se = dtab + prm->n_beams;
  c.ca = dtab + 2 * (size_t)prm->n_beams;
  c.sa = c.ca + prm->n_az;
  c.lo = reinterpret_cast<const double*>(base + o_lo);
  c.hi = reinterpret_cast<const double*>(base + o_hi);
  c.n_beams = prm->n_beams;
  c.n_az = prm->n_az;
  c.ground = ground_z;
  c.max_range = prm->max_range;
  c.noise = prm->noise_sigma;
  float* tmp = reinterpret_cast<float*>(base + o_tmp);
  float* out = reinterpret_cast<float*>(base + o_out);
  unsigned long long* masks = reinterpret_cast<unsigned long long*>(base + o_msk);
  uint32_t* counts = reinterpret_cast<uint32_t*>(base + o_cnt);
  uint32_t* totals = reinterpret_cast<uint32_t*>(base + o_tot);
  const dim3 grid(n_blocks, (unsigned)count);
  hipLaunchKernelGGL(raycast::cast_kernel, grid, dim3(raycast::RC_THREADS), 0, q, reinterpret_cast<const raycast::RayScan*>(base + o_sc),
                     c, tmp, masks, counts);
  hipLaunchKernelGGL(raycast::scan_kernel, dim3((unsigned)count), dim3(raycast::RC_THREADS), 0, q, counts, n_blocks, totals);
  hipLaunchKernelGGL(raycast::compact_kernel, grid, dim3(raycast::RC_THREADS), 0, q, tmp, masks, counts, n_rays, out);
  GLOC_HIP(hipGetLastError());
  std::vector<uint32_t> tot(count);
  GLOC_HIP(hipMemcpyAsync(tot.data(), totals, count * 4, hipMemcpyDeviceToHost, q));
  GLOC_HIP(hipStreamSynchronize(q));
  std::vector<const float*> ptrs(count);
  std::vector<size_t> ns(count);
  for (size_t i = 0; i < count; ++i) {
    ptrs[i] = out + 3 * (i * rays_pad);
    ns[i] = tot[i];
  }
  std::vector<DevScan> sc(count);
  GLOC_TRY(store_make_scans(st, count, ptrs.data(), ns.data(), 3, true, sc.data()));
  for (size_t i = 0; i < count; ++i) store_insert(st, sc[i], &scan_ids[i]);
  return GLOC_OK;
}

int gloc_scan_store_build_target_index_batch(gloc_scan_store* st, const uint32_t* scan_ids, size_t count) {
  GLOC_REQUIRE(st && (scan_ids || !count), GLOC_ERR_INVALID, "null argument");
  GLOC_HIP(hipSetDevice(st->device));
  std::lock_guard<std::mutex> lk(st->mu);
  // the re-sort rewrites points, boxes and permutations of a resident scan IN PLACE: a batch in flight that reads the scan
  // would search half-rewritten boxes (missed neighbours, not only a slower search).  Only scans that still need the
  // re-sort AND are pinned by such a batch are refused (round 5; round 4 refused every call while any batch was in flight)
  std::vector<DevScan*> ps(count);
  for (size_t i = 0; i < count; ++i) {
    GLOC_REQUIRE(scan_ids[i] < st->scans.size() && st->scans[scan_ids[i]].live, GLOC_ERR_INVALID, "unknown scan id %u",
                 scan_ids[i]);
    ps[i] = &st->scans[scan_ids[i]];
    GLOC_REQUIRE(ps[i]->kd || ps[i]->n <= (size_t)SB || ps[i]->pins == 0, GLOC_ERR_STATE,
                 "scan %u is read by %d registration batch(es) in flight (gloc_reg_batch_multi_begin without _end)", scan_ids[i],
                 ps[i]->pins);
  }
  return store_build_target_indices(st, ps.data(), count);
}

int gloc_scan_store_build_target_index(gloc_scan_store* st, uint32_t scan_id) {
  return gloc_scan_store_build_target_index_batch(st, &scan_id, 1);
}

int gloc_scan_store_release(gloc_scan_store* st, uint32_t scan_id) {
  GLOC_REQUIRE(st, GLOC_ERR_INVALID, "null store");
  GLOC_HIP(hipSetDevice(st->device));
  std::lock_guard<std::mutex> lk(st->mu);
  GLOC_REQUIRE(scan_id < st->scans.size() && st->scans[scan_id].live, GLOC_ERR_INVALID, "unknown scan id %u",
               scan_id);
  // (its block would go back to the cache and be handed to the next upload while the batch's kernels still read it)
  GLOC_REQUIRE(st->scans[scan_id].pins == 0, GLOC_ERR_STATE,
               "scan %u is read by %d registration batch(es) in flight (gloc_reg_batch_multi_begin without _end)", scan_id,
               st->scans[scan_id].pins);
  st->live_count--;
  st->live_bytes -= st->scans[scan_id].block_bytes;
  store_free_scan(st, st->scans[scan_id], true);
  st->free_ids.push_back(scan_id);
  return GLOC_OK;
}

int gloc_scan_store_clear(gloc_scan_store* st) {
  GLOC_REQUIRE(st, GLOC_ERR_INVALID, "null store");
  GLOC_HIP(hipSetDevice(st->device));
  std::lock_guard<std::mutex> lk(st->mu);
  GLOC_HIP(hipStreamSynchronize(st->stream));
  for (auto& s : st->scans)
    if (s.block) (void)hipFree(s.block);
  for (auto& kv : st->free_blocks) (void)hipFree(kv.second);
  st->scans.clear();
  st->free_ids.clear();
  st->free_blocks.clear();
  st->live_count = 0;
  st->live_bytes = 0;
  st->cached_bytes = 0;
  return GLOC_OK;
}

int gloc_scan_store_count(gloc_scan_store* st, size_t* n_scans) {
  GLOC_REQUIRE(st && n_scans, GLOC_ERR_INVALID, "null argument");
  std::lock_guard<std::mutex> lk(st->mu);
  *n_scans = st->live_count;
  return GLOC_OK;
}

int gloc_scan_store_bytes(gloc_scan_store* st, size_t* live_bytes, size_t* cached_bytes) {
  GLOC_REQUIRE(st, GLOC_ERR_INVALID, "null store");
  std::lock_guard<std::mutex> lk(st->mu);
  if (live_bytes) *live_bytes = st->live_bytes;
  if (cached_bytes) *cached_bytes = st->cached_bytes;
  return GLOC_OK;
}

int gloc_scan_store_points(gloc_scan_store* st, uint32_t scan_id, size_t* n_points) {
  GLOC_REQUIRE(st && n_points, GLOC_ERR_INVALID, "null argument");
  std::lock_guard<std::mutex> lk(st->mu);
  GLOC_REQUIRE(scan_id < st->scans.size() && st->scans[scan_id].live, GLOC_ERR_INVALID, "unknown scan id %u",
               scan_id);
  *n_points = st->scans[scan_id].n;
  return GLOC_OK;
}

// Developer / test aid (not part of include/gloc3d.h): the index of a scan as the search sees it -- per sorted
// position the original index of the point (perm), the sorted curve keys, the curve position -> sorted position
// table of a target index (kpos; identity for a scan in curve order), the launch order for 2 sources per lane
// (order2, ceil(n / 128) entries).  Any output may be null.  *is_kd: the index is in kd order.
int gloc_scan_store_debug_index(gloc_scan_store* st, uint32_t scan_id, uint32_t* perm, uint32_t* keys, uint32_t* kpos,
                                uint32_t* order2, int* is_kd) {
  GLOC_REQUIRE(st, GLOC_ERR_INVALID, "null store");
  GLOC_HIP(hipSetDevice(st->device));
  std::lock_guard<std::mutex> lk(st->mu);
  GLOC_REQUIRE(scan_id < st->scans.size() && st->scans[scan_id].live, GLOC_ERR_INVALID, "unknown scan id %u", scan_id);
  DevScan& s = st->scans[scan_id];
  if (is_kd) *is_kd = s.kd ? 1 : 0;
  if (!s.n) return GLOC_OK;
  hipStream_t q = st->stream;
  if (perm) {
    std::vector<f32x4> p(s.n);
    GLOC_HIP(hipMemcpyAsync(p.data(), s.idx.pts, sizeof(f32x4) * s.n, hipMemcpyDeviceToHost, q));
    GLOC_HIP(hipStreamSynchronize(q));
    // (memcpy, not __builtin_bit_cast on the vector element: this clang's host side hands back element 0 for that)
    for (size_t i = 0; i < s.n; ++i) std::memcpy(&perm[i], reinterpret_cast<const char*>(&p[i]) + 12, 4);
  }
  if (keys) GLOC_HIP(hipMemcpyAsync(keys, s.idx.keys, sizeof(uint32_t) * s.n, hipMemcpyDeviceToHost, q));
  if (kpos) {
    if (s.idx.kpos) {
      GLOC_HIP(hipMemcpyAsync(kpos, s.idx.kpos, sizeof(uint32_t) * s.n, hipMemcpyDeviceToHost, q));
    } else {
      for (size_t i = 0; i < s.n; ++i) kpos[i] = (uint32_t)i;
    }
  }
  if (order2) {
    GLOC_TRY(store_build_order(st, s, 2));
    GLOC_HIP(hipMemcpyAsync(order2, s.order_of(2), sizeof(uint32_t) * ((s.n + 127) / 128), hipMemcpyDeviceToHost, q));
  }
  GLOC_HIP(hipStreamSynchronize(q));
  return GLOC_OK;
}

int gloc_scan_store_download(gloc_scan_store* st, uint32_t scan_id, float* out_xyz, size_t capacity_points) {
  GLOC_REQUIRE(st && out_xyz, GLOC_ERR_INVALID, "null argument");
  GLOC_HIP(hipSetDevice(st->device));
  std::lock_guard<std::mutex> lk(st->mu);
  GLOC_REQUIRE(scan_id < st->scans.size() && st->scans[scan_id].live, GLOC_ERR_INVALID, "unknown scan id %u",
               scan_id);
  const DevScan& s = st->scans[scan_id];
  GLOC_REQUIRE(capacity_points >= s.n, GLOC_ERR_INVALID, "buffer holds %zu points, the scan has %zu",
               capacity_points, s.n);
  if (s.n) {
    GLOC_HIP(hipMemcpyAsync(out_xyz, s.xyz, sizeof(float) * 3 * s.n, hipMemcpyDeviceToHost, st->stream));
    GLOC_HIP(hipStreamSynchronize(st->stream));
  }
  return GLOC_OK;
}

}  // extern "C"
