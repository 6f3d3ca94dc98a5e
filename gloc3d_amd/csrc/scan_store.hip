// scan_store.hip -- the resident scan store (gloc_scan_store_* of include/gloc3d.h): every database
// scan of the reference's GlocEvaluator (db_files_, read again from disk for every candidate at
// registration/global_localization.cpp:521-525) is kept in HBM together with its search index, shared
// by any number of registration handles; query scans are added, used and released.
// Indexing runs entirely on the device (no host pass over the points).
#include <algorithm>
#include <new>

#include <hipcub/hipcub.hpp>

#include "scan_store.hpp"
#include "synth_kernels.hpp"

namespace gloc {
namespace reg {

__global__ void scan_header_init_kernel(ScanHeader* hdr) {
  if (threadIdx.x == 0) {
    hdr->ox = hdr->oy = hdr->oz = 0.f;
    hdr->inv_cell = 4.f;
    for (int a = 0; a < 3; ++a) {
      hdr->lo[a] = 0xFFFFFFFFu;
      hdr->hi[a] = 0u;
    }
  }
}

// strided (x, y, z, ...) -> packed xyz, and the bounding box: every work-group leaves its partial box
// in `part` ([gridDim.x][6] order-preserving integers), reduced by scan_header_finish_kernel -- no
// atomics (the first version's six atomics per wave on one cache line cost 134 us per 123k-point scan).
constexpr int PACK_BLOCKS = 128;
__global__ __launch_bounds__(256) void pack_bbox_kernel(const float* __restrict__ in, uint32_t n,
                                                        uint32_t stride, float* __restrict__ xyz,
                                                        uint32_t* __restrict__ part) {
  __shared__ uint32_t red[4][6];
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
    part[blockIdx.x * 6 + threadIdx.x] = v;
  }
}

// one wave: reduce the partial boxes, derive the key grid
__global__ __launch_bounds__(64) void scan_header_finish_kernel(ScanHeader* hdr, const uint32_t* __restrict__ part,
                                                                uint32_t n_part) {
  uint32_t lo[3] = {0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu}, hi[3] = {0u, 0u, 0u};
  for (uint32_t b = threadIdx.x; b < n_part; b += 64)
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
  const float mn0 = ord2f(lo[0]), mn1 = ord2f(lo[1]), mn2 = ord2f(lo[2]);
  const float e0 = ord2f(hi[0]) - mn0, e1 = ord2f(hi[1]) - mn1, e2 = ord2f(hi[2]) - mn2;
  const float ext = fmaxf(fmaxf(e0, e1), e2);
  const float cell = fmaxf(0.25f, ext / 1023.0f);
  hdr->ox = mn0;
  hdr->oy = mn1;
  hdr->oz = mn2;
  hdr->inv_cell = 1.0f / cell;
}

__global__ void morton_keys_kernel(const float* __restrict__ xyz, uint32_t n,
                                   const ScanHeader* __restrict__ hdr, uint32_t* __restrict__ keys,
                                   uint32_t* __restrict__ vals) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  keys[i] = morton_key(xyz[3 * (size_t)i], xyz[3 * (size_t)i + 1], xyz[3 * (size_t)i + 2], hdr->ox,
                       hdr->oy, hdr->oz, hdr->inv_cell);
  vals[i] = i;
}

__global__ void gather_sorted_kernel(const float* __restrict__ xyz, const uint32_t* __restrict__ perm,
                                     uint32_t n, uint32_t n_pad, f32x4* __restrict__ pts,
                                     uint32_t* __restrict__ inv) {
  const uint32_t s = blockIdx.x * blockDim.x + threadIdx.x;
  if (s >= n_pad) return;
  if (s >= n) {  // padding of the last chunk: farther than any real point, an index that never wins a tie
    pts[s] = f32x4{NN_FAR, NN_FAR, NN_FAR, __uint_as_float(0xFFFFFFFFu)};
    return;
  }
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
// one thread per PAIR of sub-blocks; a sub-block past the end of the scan gets an empty (inverted) box
__global__ void subblock_boxes_kernel(const f32x4* __restrict__ pts, uint32_t n, uint32_t npairs,
                                      f32x4* __restrict__ sb2) {
  const uint32_t b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= npairs) return;
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
  sb2[3 * (size_t)b + 0] = f32x4{mn[0][0], mn[1][0], mn[0][1], mn[1][1]};
  sb2[3 * (size_t)b + 1] = f32x4{mn[0][2], mn[1][2], mx[0][0], mx[1][0]};
  sb2[3 * (size_t)b + 2] = f32x4{mx[0][1], mx[1][1], mx[0][2], mx[1][2]};
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
// smaller coordinates: one device-wide radix sort per level by (node, coordinate), stable, so that equal
// coordinates keep their current (deterministic) order.  Positions >= n are never materialised: they stand
// for points at +infinity, which stay at the end of their node under every sort.
__global__ void kd_box_init_kernel(uint32_t* __restrict__ box, uint32_t nodes) {
  const uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= nodes) return;
#pragma unroll
  for (int a = 0; a < 3; ++a) {
    box[6 * k + a] = 0xFFFFFFFFu;
    box[6 * k + 3 + a] = 0u;
  }
}

// one thread per block of 16 points; a wave's 64 blocks lie in ONE node when the node holds >= 1024 positions
__global__ __launch_bounds__(256) void kd_node_bbox_kernel(const f32x4* __restrict__ p, uint32_t n, uint32_t shift,
                                                           uint32_t* __restrict__ box) {
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
  uint32_t* o = box + 6 * (size_t)(first >> shift);
#pragma unroll
  for (int a = 0; a < 3; ++a) {
    atomicMin(o + a, lo[a]);
    atomicMax(o + 3 + a, hi[a]);
  }
}

__global__ void kd_keys_kernel(const f32x4* __restrict__ p, uint32_t n, uint32_t shift, const uint32_t* __restrict__ box,
                               unsigned long long* __restrict__ keys, uint32_t* __restrict__ vals) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const uint32_t node = i >> shift;
  const uint32_t* b = box + 6 * (size_t)node;
  const float ex = ord2f(b[3]) - ord2f(b[0]), ey = ord2f(b[4]) - ord2f(b[1]), ez = ord2f(b[5]) - ord2f(b[2]);
  int axis = 0;           // the widest axis; ties -> the lower axis
  float e = ex;
  if (ey > e) { axis = 1; e = ey; }
  if (ez > e) axis = 2;
  const f32x4 v = p[i];
  const float c = axis == 0 ? v.x : (axis == 1 ? v.y : v.z);
  keys[i] = ((unsigned long long)node << 32) | f2ord(c);
  vals[i] = i;
}

__global__ void kd_gather_kernel(const f32x4* __restrict__ p_in, const uint32_t* __restrict__ h_in,
                                 const uint32_t* __restrict__ perm, uint32_t n, f32x4* __restrict__ p_out,
                                 uint32_t* __restrict__ h_out) {
  const uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= n) return;
  const uint32_t s = perm[j];
  p_out[j] = p_in[s];
  h_out[j] = h_in[s];
}

__global__ void kd_iota_kernel(uint32_t* __restrict__ h, uint32_t n) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) h[i] = i;
}

// the re-sorted points back into the scan: pts (kd order, padded), inv (original index -> kd position), kpos
// (curve position -> kd position)
__global__ void kd_finish_kernel(const f32x4* __restrict__ p, const uint32_t* __restrict__ h, uint32_t n, uint32_t n_pad,
                                 f32x4* __restrict__ pts, uint32_t* __restrict__ inv, uint32_t* __restrict__ kpos) {
  const uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= n_pad) return;
  if (j >= n) {
    pts[j] = f32x4{NN_FAR, NN_FAR, NN_FAR, __uint_as_float(0xFFFFFFFFu)};
    return;
  }
  const f32x4 v = p[j];
  pts[j] = v;
  inv[__float_as_uint(v.w)] = j;
  kpos[h[j]] = j;
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
  const uint32_t group = 64u * (uint32_t)cs;
  const uint32_t ng = (uint32_t)((s.n + group - 1) / group);
  GLOC_TRY(st->sort_keys.ensure(sizeof(float) * std::max<size_t>(ng, s.n), q));
  GLOC_TRY(st->sort_vals.ensure(sizeof(uint32_t) * std::max<size_t>(ng, s.n), q));
  GLOC_TRY(st->sort_perm.ensure(sizeof(float) * std::max<size_t>(ng, s.n), q));
  hipLaunchKernelGGL(group_extent_kernel, dim3((ng + 3) / 4), dim3(256), 0, q, s.idx.pts, (uint32_t)s.n, group,
                     ng, st->sort_keys.as<float>(), st->sort_vals.as<uint32_t>());
  size_t tmp_bytes = 0;
  GLOC_HIP(hipcub::DeviceRadixSort::SortPairsDescending(nullptr, tmp_bytes, st->sort_keys.as<float>(),
                                                        st->sort_perm.as<float>(), st->sort_vals.as<uint32_t>(),
                                                        s.order, (int)ng, 0, 32, q));
  GLOC_TRY(st->sort_tmp.ensure(std::max<size_t>(tmp_bytes, 16), q));
  GLOC_HIP(hipcub::DeviceRadixSort::SortPairsDescending(st->sort_tmp.p, tmp_bytes, st->sort_keys.as<float>(),
                                                        st->sort_perm.as<float>(), st->sort_vals.as<uint32_t>(),
                                                        s.order, (int)ng, 0, 32, q));
  GLOC_HIP(hipGetLastError());
  GLOC_HIP(hipStreamSynchronize(q));
  s.order_built |= 1u << cs;
  return GLOC_OK;
}


int store_build_target_index(gloc_scan_store* st, DevScan& s) {
  if (s.kd || s.n <= (size_t)SB) {
    s.kd = true;  // (a scan of one sub-block is in kd order as it is)
    return GLOC_OK;
  }
  hipStream_t q = st->stream;
  const uint32_t n = (uint32_t)s.n;
  uint32_t L = 0;
  while (((size_t)SB << L) < s.n) ++L;  // P = SB * 2^L positions
  const Layout lay = layout_for(s.n);
  GLOC_TRY(st->kd_k0.ensure(sizeof(unsigned long long) * n, q));
  GLOC_TRY(st->kd_k1.ensure(sizeof(unsigned long long) * n, q));
  GLOC_TRY(st->kd_v0.ensure(sizeof(uint32_t) * n, q));
  GLOC_TRY(st->kd_v1.ensure(sizeof(uint32_t) * n, q));
  GLOC_TRY(st->kd_p0.ensure((sizeof(f32x4) + sizeof(uint32_t)) * (size_t)n, q));
  GLOC_TRY(st->kd_p1.ensure((sizeof(f32x4) + sizeof(uint32_t)) * (size_t)n, q));
  GLOC_TRY(st->kd_box.ensure(sizeof(uint32_t) * 6 * ((size_t)1 << (L ? L - 1 : 0)), q));
  f32x4* pp[2] = {st->kd_p0.as<f32x4>(), st->kd_p1.as<f32x4>()};
  uint32_t* hh[2] = {reinterpret_cast<uint32_t*>(pp[0] + n), reinterpret_cast<uint32_t*>(pp[1] + n)};
  const unsigned nb = (n + 255) / 256, nblk = ((n + SB - 1) / SB + 255) / 256;
  GLOC_HIP(hipMemcpyAsync(pp[0], s.idx.pts, sizeof(f32x4) * n, hipMemcpyDeviceToDevice, q));
  hipLaunchKernelGGL(kd_iota_kernel, dim3(nb), dim3(256), 0, q, hh[0], n);
  int cur = 0;
  size_t tmp_cap = 0;
  for (uint32_t l = 0; l < L; ++l) {
    const uint32_t shift = 4 + L - l;  // log2 of the node size at this level (SB = 16 = 2^4)
    static_assert(SB == 16, "shift arithmetic assumes 16-point leaves");
    const uint32_t nodes = 1u << l;
    hipLaunchKernelGGL(kd_box_init_kernel, dim3((nodes + 255) / 256), dim3(256), 0, q, st->kd_box.as<uint32_t>(), nodes);
    hipLaunchKernelGGL(kd_node_bbox_kernel, dim3(nblk), dim3(256), 0, q, pp[cur], n, shift, st->kd_box.as<uint32_t>());
    hipLaunchKernelGGL(kd_keys_kernel, dim3(nb), dim3(256), 0, q, pp[cur], n, shift, st->kd_box.as<uint32_t>(),
                       st->kd_k0.as<unsigned long long>(), st->kd_v0.as<uint32_t>());
    GLOC_HIP(hipGetLastError());
    size_t tmp_bytes = 0;
    GLOC_HIP(hipcub::DeviceRadixSort::SortPairs(nullptr, tmp_bytes, st->kd_k0.as<unsigned long long>(),
                                                st->kd_k1.as<unsigned long long>(), st->kd_v0.as<uint32_t>(),
                                                st->kd_v1.as<uint32_t>(), (int)n, 0, (int)(32 + l), q));
    if (tmp_bytes > tmp_cap) {
      GLOC_TRY(st->sort_tmp.ensure(std::max<size_t>(tmp_bytes, 16), q));
      tmp_cap = tmp_bytes;
    }
    GLOC_HIP(hipcub::DeviceRadixSort::SortPairs(st->sort_tmp.p, tmp_bytes, st->kd_k0.as<unsigned long long>(),
                                                st->kd_k1.as<unsigned long long>(), st->kd_v0.as<uint32_t>(),
                                                st->kd_v1.as<uint32_t>(), (int)n, 0, (int)(32 + l), q));
    hipLaunchKernelGGL(kd_gather_kernel, dim3(nb), dim3(256), 0, q, pp[cur], hh[cur], st->kd_v1.as<uint32_t>(), n,
                       pp[cur ^ 1], hh[cur ^ 1]);
    cur ^= 1;
  }
  // write back and rebuild what depends on the order: inv, kpos, all boxes, the launch orders
  f32x4* p4 = const_cast<f32x4*>(s.idx.pts);
  uint32_t* inv = const_cast<uint32_t*>(s.idx.inv);
  const size_t nch = (s.n + CH - 1) / CH, nsup = (nch + 63) / 64;
  hipLaunchKernelGGL(kd_finish_kernel, dim3((unsigned)((lay.np + 255) / 256)), dim3(256), 0, q, pp[cur], hh[cur], n,
                     (uint32_t)lay.np, p4, inv, s.kpos_mem);
  hipLaunchKernelGGL(chunk_boxes_kernel, dim3((unsigned)nch), dim3(64), 0, q, p4, n, const_cast<f32x4*>(s.idx.box_lo),
                     const_cast<f32x4*>(s.idx.box_hi));
  hipLaunchKernelGGL(subblock_boxes_kernel, dim3((unsigned)((lay.b1 / 3 + 255) / 256)), dim3(256), 0, q, p4, n,
                     (uint32_t)(lay.b1 / 3), const_cast<f32x4*>(s.idx.sb2));
  hipLaunchKernelGGL(super_boxes_kernel, dim3((unsigned)nsup), dim3(64), 0, q, s.idx.box_lo, s.idx.box_hi, (uint32_t)nch,
                     const_cast<f32x4*>(s.idx.sup_lo), const_cast<f32x4*>(s.idx.sup_hi));
  GLOC_HIP(hipGetLastError());
  s.idx.kpos = s.kpos_mem;
  s.kd = true;
  s.order_built = 0;  // the launch orders list groups of the old order
  return store_build_order(st, s, 2);  // synchronises the stream
}

int store_make_scan(gloc_scan_store* st, const float* pts, size_t n, size_t stride, bool device_src,
                    DevScan* out) {
  DevScan s;
  s.n = n;
  const Layout L = layout_for(n);
  const size_t nch = (n + CH - 1) / CH, nsup = (nch + 63) / 64;
  GLOC_TRY(take_block(st, L.bytes, &s.block, &s.block_bytes));
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
  s.idx = ScanIndexDev{p4, lo, hi, sb2, nullptr, keys, inv, hdr, ulo, uhi, (uint32_t)n, (uint32_t)nch,
                       (uint32_t)nsup, 0u};
  hipStream_t q = st->stream;
  auto fail = [&](int code) {
    (void)hipStreamSynchronize(q);
    store_free_scan(st, s, true);
    return code;
  };
  hipLaunchKernelGGL(scan_header_init_kernel, dim3(1), dim3(64), 0, q, hdr);
  if (n) {
    const float* d_in = pts;
    if (!device_src) {
      // host points travel as they are (stride included) into a staging buffer; the pack kernel
      // drops the extra channels on the device
      if (int rc = st->stage.ensure(sizeof(float) * stride * n, q)) return fail(rc);
      const hipError_t eu = hipMemcpyAsync(st->stage.p, pts, sizeof(float) * stride * n, hipMemcpyHostToDevice, q);
      if (eu != hipSuccess) {
        (void)hipGetLastError();
        set_err("scan upload failed: %s", hipGetErrorString(eu));
        return fail(GLOC_ERR_HIP);
      }
      d_in = st->stage.as<float>();
    }
    for (auto need : {std::make_pair(&st->sort_keys, sizeof(uint32_t) * n), std::make_pair(&st->sort_vals, sizeof(uint32_t) * n),
                      std::make_pair(&st->sort_perm, sizeof(uint32_t) * n),
                      std::make_pair(&st->sort_tmp, sizeof(uint32_t) * 6 * PACK_BLOCKS)})
      if (int rc = need.first->ensure(need.second, q)) return fail(rc);  // (the actual code: NOMEM or a HIP error)
    const unsigned nb = (unsigned)((n + 255) / 256);
    const unsigned npk = std::min<unsigned>(nb, PACK_BLOCKS);
    uint32_t* part = st->sort_tmp.as<uint32_t>();  // free until the radix sort below
    hipLaunchKernelGGL(pack_bbox_kernel, dim3(npk), dim3(256), 0, q, d_in, (uint32_t)n, (uint32_t)stride, s.xyz, part);
    hipLaunchKernelGGL(scan_header_finish_kernel, dim3(1), dim3(64), 0, q, hdr, part, npk);
    hipLaunchKernelGGL(morton_keys_kernel, dim3(nb), dim3(256), 0, q, s.xyz, (uint32_t)n, hdr,
                       st->sort_keys.as<uint32_t>(), st->sort_vals.as<uint32_t>());
    size_t tmp_bytes = 0;
    if (hipcub::DeviceRadixSort::SortPairs(nullptr, tmp_bytes, st->sort_keys.as<uint32_t>(), keys,
                                           st->sort_vals.as<uint32_t>(), st->sort_perm.as<uint32_t>(),
                                           (int)n, 0, 30, q) != hipSuccess)
      return fail(GLOC_ERR_HIP);
    if (int rc = st->sort_tmp.ensure(std::max<size_t>(tmp_bytes, 16), q)) return fail(rc);
    if (hipcub::DeviceRadixSort::SortPairs(st->sort_tmp.p, tmp_bytes, st->sort_keys.as<uint32_t>(), keys,
                                           st->sort_vals.as<uint32_t>(), st->sort_perm.as<uint32_t>(),
                                           (int)n, 0, 30, q) != hipSuccess)
      return fail(GLOC_ERR_HIP);
    hipLaunchKernelGGL(gather_sorted_kernel, dim3((unsigned)((L.np + 255) / 256)), dim3(256), 0, q, s.xyz,
                       st->sort_perm.as<uint32_t>(), (uint32_t)n, (uint32_t)L.np, p4, inv);
    hipLaunchKernelGGL(chunk_boxes_kernel, dim3((unsigned)nch), dim3(64), 0, q, p4, (uint32_t)n, lo, hi);
    hipLaunchKernelGGL(subblock_boxes_kernel, dim3((unsigned)((L.b1 / 3 + 255) / 256)), dim3(256), 0, q, p4,
                       (uint32_t)n, (uint32_t)(L.b1 / 3), sb2);
    hipLaunchKernelGGL(super_boxes_kernel, dim3((unsigned)nsup), dim3(64), 0, q, lo, hi, (uint32_t)nch, ulo, uhi);
    const hipError_t ei = hipGetLastError();  // (read once: the call clears the error)
    if (ei != hipSuccess) {
      set_err("scan indexing failed: %s", hipGetErrorString(ei));
      return fail(GLOC_ERR_HIP);
    }
    int rc = store_build_order(st, s, 2);  // the default sources-per-lane; synchronises the stream
    if (rc != GLOC_OK) return fail(rc);
  } else if (hipStreamSynchronize(q) != hipSuccess) {
    return fail(GLOC_ERR_HIP);
  }
  s.live = true;
  *out = s;
  return GLOC_OK;
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
  for (DevBuf* b : {&st->sort_tmp, &st->sort_keys, &st->sort_vals, &st->sort_perm, &st->stage, &st->kd_k0, &st->kd_k1,
                    &st->kd_v0, &st->kd_v1, &st->kd_p0, &st->kd_p1, &st->kd_box})
    b->release();
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

int gloc_scan_store_build_target_index(gloc_scan_store* st, uint32_t scan_id) {
  GLOC_REQUIRE(st, GLOC_ERR_INVALID, "null store");
  GLOC_HIP(hipSetDevice(st->device));
  std::lock_guard<std::mutex> lk(st->mu);
  GLOC_REQUIRE(scan_id < st->scans.size() && st->scans[scan_id].live, GLOC_ERR_INVALID, "unknown scan id %u", scan_id);
  return store_build_target_index(st, st->scans[scan_id]);
}

int gloc_scan_store_release(gloc_scan_store* st, uint32_t scan_id) {
  GLOC_REQUIRE(st, GLOC_ERR_INVALID, "null store");
  GLOC_HIP(hipSetDevice(st->device));
  std::lock_guard<std::mutex> lk(st->mu);
  GLOC_REQUIRE(scan_id < st->scans.size() && st->scans[scan_id].live, GLOC_ERR_INVALID, "unknown scan id %u",
               scan_id);
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
