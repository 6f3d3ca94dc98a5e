// seg_sort.hpp -- segmented least-significant-digit radix sort of (key, value) pairs, hand-written for gfx950.
//
// Every sort on the scan-indexing path goes through here: the curve keys of a scan (30 bits), the extents of its
// source groups (32 bits), the (node, coordinate) keys of a kd level (up to 49 bits) -- for ONE scan or for a
// whole batch of scans with the same three launches per digit: a segment is one scan's slice of the
// concatenated key / value arrays, blockIdx.y walks the segments.  (Rounds 1-2 called hipCUB per scan: at 124 k
// elements rocPRIM picks a merge sort of 9 launches, a kd level costs 14 launches, a KITTI-00-sized database
// 1.2 M launches; batched, the launch count does not depend on the number of scans.)
//
// Per digit:  hist_kernel     a work-group counts the digits of its tile of 2048 elements
//             scan_kernel     one work-group per segment: exclusive prefix over (digit, tile), digit-major
//             scatter_kernel  a work-group re-reads its tile, ranks every element among the equal digits before
//                             it (stable: tiles in order, waves in order, chunks of 64 in order, lanes in order)
//                             and writes it to its place
// Equal digits inside a wave are found with one ballot per bit of the digit instead of LDS atomics:
// the kd keys of the upper levels and the top digit of the curve keys have very few distinct digits, and a
// thousand same-address atomics per tile would serialise.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace gloc {
namespace segsort {

constexpr int TILE = 2048;  // elements per work-group
constexpr int THREADS = 256;
constexpr int CHUNKS = TILE / THREADS;  // chunks of 64 per wave: a wave owns 512 consecutive elements of the tile
// Digit width: a template parameter (8, 10 or 11 bits).  Wider digits mean fewer passes of three launches each, but
// more ballots per element and 4 - 8 x the counters to clear, write and scan per tile: measured on 124 k-element
// segments 8 bits wins (one scan indexed in 0.22 ms against 0.31 ms with 10-bit digits for the curve keys, a kd re-sort
// in 2.1 ms against 7.3 ms with 11-bit digits), so every caller uses 8.

struct Seg {
  uint32_t begin, n;  // elements [begin, begin + n) of the key / value arrays
};

// lanes (among the active ones) that hold the same digit as this lane
template <int BITS>
__device__ __forceinline__ unsigned long long match_digit(uint32_t d, bool active) {
  unsigned long long peers = __builtin_amdgcn_ballot_w64(active);
#pragma unroll
  for (int b = 0; b < BITS; ++b) {
    const bool bit = (d >> b) & 1u;
    const unsigned long long m = __builtin_amdgcn_ballot_w64(bit);
    peers &= bit ? m : ~m;
  }
  return peers;
}

template <typename K, int BITS>
__global__ __launch_bounds__(THREADS) void hist_kernel(const K* __restrict__ keys, const Seg* __restrict__ segs,
                                                       uint32_t max_tiles, uint32_t shift, uint32_t* __restrict__ hist) {
  constexpr int RADIX = 1 << BITS;
  __shared__ uint32_t cnt[RADIX];
  const Seg sg = segs[blockIdx.y];
  const uint32_t t0 = blockIdx.x * TILE;
  if (t0 >= sg.n) return;  // uniform over the work-group
  for (int i = threadIdx.x; i < RADIX; i += THREADS) cnt[i] = 0;
  __syncthreads();
  const K* k = keys + sg.begin + t0;
  const uint32_t m = sg.n - t0 < (uint32_t)TILE ? sg.n - t0 : (uint32_t)TILE;
  const int lane = threadIdx.x & 63;
#pragma unroll
  for (int c = 0; c < CHUNKS; ++c) {
    const uint32_t e = c * THREADS + threadIdx.x;  // (counting does not care about the order)
    const bool act = e < m;
    const uint32_t d = act ? (uint32_t)(k[e] >> shift) & (uint32_t)(RADIX - 1) : 0u;
    const unsigned long long peers = match_digit<BITS>(d, act);
    if (act && lane == __builtin_ctzll(peers)) atomicAdd(&cnt[d], (uint32_t)__popcll(peers));  // one add per (wave, digit)
  }
  __syncthreads();
  for (int i = threadIdx.x; i < RADIX; i += THREADS) hist[((size_t)blockIdx.y * RADIX + i) * max_tiles + blockIdx.x] = cnt[i];
}

// hist[seg][digit][tile] -> where the first element of that digit of that tile goes, relative to the start of the
// segment = base[seg][digit] (exclusive prefix of the digit totals) + the exclusive prefix over the tiles of that digit,
// which replaces the count in place.  One work-group of 16 waves per segment; a wave takes one digit at a time, its lanes
// the tiles (64 at a time, the running sum carried on): the first version -- a thread per digit walking the tiles one
// by one -- was a chain of dependent global round trips and took longer than the two kernels around it.
constexpr int SCAN_THREADS = 1024;
template <int BITS>
__global__ __launch_bounds__(SCAN_THREADS) void scan_kernel(const Seg* __restrict__ segs, uint32_t max_tiles,
                                                            uint32_t* __restrict__ hist, uint32_t* __restrict__ base) {
  constexpr int RADIX = 1 << BITS;
  __shared__ uint32_t tot[RADIX];
  const Seg sg = segs[blockIdx.x];
  const uint32_t nt = (sg.n + TILE - 1) / TILE;
  const int w = threadIdx.x >> 6, lane = threadIdx.x & 63;
  // four digits of the wave at a time: their loads are in flight together (one digit after the other was a chain of
  // sixteen load -> scan -> store round trips per wave: 13 us of a 27-us pass on one 124 k-element segment)
  constexpr int DW = SCAN_THREADS / 64, U = 4;
  static_assert(RADIX % (DW * U) == 0, "digits per wave a multiple of the unroll");
  for (int d0 = w; d0 < RADIX; d0 += DW * U) {
    uint32_t run[U];
#pragma unroll
    for (int u = 0; u < U; ++u) run[u] = 0;
    for (uint32_t t0 = 0; t0 < nt; t0 += 64) {
      const uint32_t t = t0 + lane;
      uint32_t v[U], inc[U];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const uint32_t* h = hist + ((size_t)blockIdx.x * RADIX + (d0 + u * DW)) * max_tiles;
        v[u] = t < nt ? h[t] : 0u;
      }
#pragma unroll
      for (int u = 0; u < U; ++u) {
        inc[u] = v[u];  // inclusive prefix over the lanes
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
          const uint32_t up = __shfl_up(inc[u], o);
          if (lane >= o) inc[u] += up;
        }
      }
#pragma unroll
      for (int u = 0; u < U; ++u) {
        uint32_t* h = hist + ((size_t)blockIdx.x * RADIX + (d0 + u * DW)) * max_tiles;
        if (t < nt) h[t] = run[u] + inc[u] - v[u];
        run[u] += __shfl(inc[u], 63);
      }
    }
#pragma unroll
    for (int u = 0; u < U; ++u)
      if (lane == 0) tot[d0 + u * DW] = run[u];
  }
  __syncthreads();
  // exclusive prefix of the digit totals (RADIX values: every thread sums what lies before its digits)
  for (int d = threadIdx.x; d < RADIX; d += SCAN_THREADS) {
    uint32_t b = 0;
    for (int j = 0; j < d; ++j) b += tot[j];
    base[(size_t)blockIdx.x * RADIX + d] = b;
  }
}

template <typename K, int BITS>
__global__ __launch_bounds__(THREADS) void scatter_kernel(const K* __restrict__ kin, const uint32_t* __restrict__ vin,
                                                          K* __restrict__ kout, uint32_t* __restrict__ vout,
                                                          const Seg* __restrict__ segs, uint32_t max_tiles, uint32_t shift,
                                                          const uint32_t* __restrict__ hist, const uint32_t* __restrict__ base) {
  constexpr int RADIX = 1 << BITS;
  __shared__ uint32_t cnt[THREADS / 64][RADIX];  // per wave: digit counts, then the running output position
  const Seg sg = segs[blockIdx.y];
  const uint32_t t0 = blockIdx.x * TILE;
  if (t0 >= sg.n) return;
  const int w = threadIdx.x >> 6, lane = threadIdx.x & 63;
  for (int i = threadIdx.x; i < (THREADS / 64) * RADIX; i += THREADS) (&cnt[0][0])[i] = 0;
  __syncthreads();
  const uint32_t m = sg.n - t0 < (uint32_t)TILE ? sg.n - t0 : (uint32_t)TILE;
  const size_t in0 = (size_t)sg.begin + t0;
  K key[CHUNKS];
  uint32_t val[CHUNKS];
  bool act[CHUNKS];
  // a wave owns elements [w * 512, w * 512 + 512) of the tile, chunk c = 64 consecutive elements: order is kept
#pragma unroll
  for (int c = 0; c < CHUNKS; ++c) {
    const uint32_t e = w * (TILE / (THREADS / 64)) + c * 64 + lane;
    act[c] = e < m;
    key[c] = act[c] ? kin[in0 + e] : (K)0;
    val[c] = act[c] ? vin[in0 + e] : 0u;
  }
  // 1: digit counts of this wave's elements (one lane per distinct digit adds: no atomics, no conflicts)
#pragma unroll
  for (int c = 0; c < CHUNKS; ++c) {
    const uint32_t d = (uint32_t)(key[c] >> shift) & (uint32_t)(RADIX - 1);
    const unsigned long long peers = match_digit<BITS>(d, act[c]);
    if (act[c] && lane == __builtin_ctzll(peers)) cnt[w][d] += (uint32_t)__popcll(peers);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  }
  __syncthreads();
  // 2: where each wave's first element of every digit goes: the tile's place for that digit + the earlier waves' counts
  for (uint32_t d = threadIdx.x; d < (uint32_t)RADIX; d += THREADS) {
    uint32_t b = hist[((size_t)blockIdx.y * RADIX + d) * max_tiles + blockIdx.x] + base[(size_t)blockIdx.y * RADIX + d];
#pragma unroll
    for (int ww = 0; ww < THREADS / 64; ++ww) {
      const uint32_t c = cnt[ww][d];
      cnt[ww][d] = b;
      b += c;
    }
  }
  __syncthreads();
  // 3: rank among the equal digits before it, write out, advance the running position
#pragma unroll
  for (int c = 0; c < CHUNKS; ++c) {
    const uint32_t d = (uint32_t)(key[c] >> shift) & (uint32_t)(RADIX - 1);
    const unsigned long long peers = match_digit<BITS>(d, act[c]);
    const uint32_t pos = cnt[w][d] + (uint32_t)__popcll(peers & ((1ull << lane) - 1ull));
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();  // every lane has read the position before a leader moves it on
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    if (act[c] && lane == __builtin_ctzll(peers)) cnt[w][d] += (uint32_t)__popcll(peers);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    if (act[c]) {
      kout[(size_t)sg.begin + pos] = key[c];
      vout[(size_t)sg.begin + pos] = val[c];
    }
  }
}

// bytes of scratch (tile histograms + digit bases) for a sort of n_segs segments of at most max_n elements (any digit
// width up to 11 bits)
inline size_t scratch_bytes(uint32_t n_segs, uint32_t max_n) {
  const size_t max_tiles = ((size_t)max_n + TILE - 1) / TILE;
  return sizeof(uint32_t) * 2048 * ((max_tiles ? max_tiles : 1) + 1) * (n_segs ? n_segs : 1);
}

// Sorts every segment by bits [begin_bit, end_bit) of its keys (bits at and above end_bit must be zero), stable,
// BITS bits per pass.  The pairs ping-pong between (k0, v0) and (k1, v1); returns the index (0 / 1) of the buffers
// that hold the result.
template <typename K, int BITS>
int sort_pairs(hipStream_t q, K* k0, K* k1, uint32_t* v0, uint32_t* v1, const Seg* d_segs, uint32_t n_segs,
               uint32_t max_n, int begin_bit, int end_bit, uint32_t* d_hist) {
  static_assert(BITS == 8 || BITS == 10 || BITS == 11, "digit widths with kernels instantiated");
  if (!n_segs || !max_n) return 0;
  const uint32_t max_tiles = (max_n + TILE - 1) / TILE;
  uint32_t* d_base = d_hist + (size_t)(1 << BITS) * max_tiles * n_segs;  // (scratch_bytes() leaves room for it)
  K* k[2] = {k0, k1};
  uint32_t* v[2] = {v0, v1};
  int cur = 0;
  for (int shift = begin_bit; shift < end_bit; shift += BITS) {
    hipLaunchKernelGGL((hist_kernel<K, BITS>), dim3(max_tiles, n_segs), dim3(THREADS), 0, q, k[cur], d_segs, max_tiles,
                       (uint32_t)shift, d_hist);
    hipLaunchKernelGGL((scan_kernel<BITS>), dim3(n_segs), dim3(SCAN_THREADS), 0, q, d_segs, max_tiles, d_hist, d_base);
    hipLaunchKernelGGL((scatter_kernel<K, BITS>), dim3(max_tiles, n_segs), dim3(THREADS), 0, q, k[cur], v[cur], k[cur ^ 1],
                       v[cur ^ 1], d_segs, max_tiles, (uint32_t)shift, d_hist, d_base);
    cur ^= 1;
  }
  return cur;
}

// ---- small segments in one launch: a work-group sorts one segment of up to 4096 (key, value) pairs in LDS ----------
// Bitonic network on (key << 32 | value): the value makes every element distinct, so the result is the stable order
// whenever values ascend with the input position (they do: values are the positions).  Used for the launch order of a
// scan's source groups (969 groups at 124 k points): one launch instead of the twelve of a four-pass radix sort.
constexpr int SMALL_MAX = 4096;
static __global__ __launch_bounds__(1024) void small_sort_kernel(const uint32_t* __restrict__ keys, const uint32_t* __restrict__ vals,
                                                          const Seg* __restrict__ segs, uint32_t* __restrict__ vals_out) {
  __shared__ unsigned long long a[SMALL_MAX];
  const Seg sg = segs[blockIdx.x];
  uint32_t m = 1;
  while (m < sg.n) m <<= 1;  // padded to a power of two with keys that sort last
  for (uint32_t i = threadIdx.x; i < m; i += 1024)
    a[i] = i < sg.n ? ((unsigned long long)keys[sg.begin + i] << 32) | vals[sg.begin + i] : ~0ull;
  __syncthreads();
  for (uint32_t k = 2; k <= m; k <<= 1)
    for (uint32_t j = k >> 1; j > 0; j >>= 1) {
      for (uint32_t i = threadIdx.x; i < m; i += 1024) {
        const uint32_t p = i ^ j;
        if (p > i) {
          const unsigned long long x = a[i], y = a[p];
          const bool up = (i & k) == 0;
          if ((x > y) == up) {
            a[i] = y;
            a[p] = x;
          }
        }
      }
      __syncthreads();
    }
  for (uint32_t i = threadIdx.x; i < sg.n; i += 1024) vals_out[sg.begin + i] = (uint32_t)a[i];
}

}  // namespace segsort
}  // namespace gloc
