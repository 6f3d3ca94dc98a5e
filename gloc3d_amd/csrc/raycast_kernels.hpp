// raycast_kernels.hpp -- bench / test support: a spinning 64-beam lidar ray-cast over a procedural world (ground plane +
// axis-aligned boxes) ON THE DEVICE, so that a KITTI-00-sized store of 4541 DISTINCT views (SURVEY.md 8d cfg D: "a
// 4541-pose loop trajectory through one procedural world") is made in seconds.  The twin of gloc3d_amd/synth.py::lidar_scan
// (same rays, same slab test, same counter-RNG range noise, fp64 throughout, results rounded to fp32 once): equal to it
// up to the last bits of the fp64 products (numpy's matmul may fuse), i.e. far below the 2 cm range noise; a ray whose
// range sits within that of max_range may be kept by one and dropped by the other.  Not part of the hot path.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "synth_kernels.hpp"

namespace gloc {
namespace raycast {

struct RayScan {
  double R[9];     // world <- sensor rotation, row-major
  double o[3];     // sensor origin in the world
  uint64_t key;    // rng_key(seed, 7): range noise
  uint32_t box0, box1;  // this scan's boxes: [box0, box1) of the lo / hi arrays
};

struct RayCfg {
  const double* ce;  // [n_beams] cos(elevation)
  const double* se;  // [n_beams] sin(elevation)
  const double* ca;  // [n_az]    cos(azimuth)
  const double* sa;  // [n_az]    sin(azimuth)
  const double* lo;  // [n_boxes][3]
  const double* hi;
  uint32_t n_beams, n_az;
  double ground, max_range, noise;
};

constexpr int RC_THREADS = 256;

// Pass 1: ray r = beam * n_az + az of scan blockIdx.y -> its return (sensor frame, fp32) + a validity mask per wave +
// a count per work-group.
static __global__ __launch_bounds__(RC_THREADS) void cast_kernel(const RayScan* __restrict__ scans, RayCfg c,
                                                                float* __restrict__ tmp, unsigned long long* __restrict__ masks,
                                                                uint32_t* __restrict__ counts) {
  const RayScan& s = scans[blockIdx.y];
  const uint32_t n_rays = c.n_beams * c.n_az;
  const uint32_t r = blockIdx.x * RC_THREADS + threadIdx.x;
  const size_t ray_base = (size_t)blockIdx.y * ((size_t)gridDim.x * RC_THREADS);
  bool ok = false;
  float px = 0.f, py = 0.f, pz = 0.f;
  if (r < n_rays) {
    const uint32_t b = r / c.n_az, a = r - b * c.n_az;
    const double d[3] = {c.ce[b] * c.ca[a], c.ce[b] * c.sa[a], c.se[b]};
    double dw[3], inv[3];
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      dw[i] = (d[0] * s.R[3 * i] + d[1] * s.R[3 * i + 1]) + d[2] * s.R[3 * i + 2];
      inv[i] = 1.0 / dw[i];
    }
    double tbest = __builtin_huge_val();
    const double tg = (c.ground - s.o[2]) / dw[2];
    if (dw[2] < 0.0 && tg > 0.0) tbest = tg;
    for (uint32_t k = s.box0; k < s.box1; ++k) {
      double tn = -__builtin_huge_val(), tf = __builtin_huge_val();
      bool any = false;
#pragma unroll
      for (int i = 0; i < 3; ++i) {
        const double t1 = (c.lo[3 * k + i] - s.o[i]) * inv[i], t2 = (c.hi[3 * k + i] - s.o[i]) * inv[i];
        if (t1 != t1 || t2 != t2) continue;  // 0 * inf: the ray runs inside that slab's plane (numpy: nanmax / nanmin skip it)
        const double mn = t1 < t2 ? t1 : t2, mx = t1 < t2 ? t2 : t1;
        tn = mn > tn ? mn : tn;
        tf = mx < tf ? mx : tf;
        any = true;
      }
      if (any && tn <= tf && tn > 0.5 && tn < tbest) tbest = tn;
    }
    ok = tbest < c.max_range;
    if (ok) {
      const double rr = tbest + (double)synth::rng_gauss(s.key, (uint64_t)r) * c.noise;
      px = (float)(d[0] * rr);
      py = (float)(d[1] * rr);
      pz = (float)(d[2] * rr);
    }
    float* t = tmp + 3 * (ray_base + r);
    t[0] = px;
    t[1] = py;
    t[2] = pz;
  }
  const unsigned long long m = __ballot(ok);
  __shared__ uint32_t wcnt[RC_THREADS / 64];
  const int w = threadIdx.x >> 6;
  if ((threadIdx.x & 63) == 0) {
    masks[(ray_base + blockIdx.x * RC_THREADS) / 64 + w] = m;
    wcnt[w] = (uint32_t)__popcll(m);
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    uint32_t t = 0;
    for (int i = 0; i < RC_THREADS / 64; ++i) t += wcnt[i];
    counts[(size_t)blockIdx.y * gridDim.x + blockIdx.x] = t;
  }
}

// Pass 2: exclusive scan of a scan's work-group counts (one work-group per scan) -> offsets, total.
static __global__ __launch_bounds__(RC_THREADS) void scan_kernel(uint32_t* __restrict__ counts, uint32_t n_blocks,
                                                                uint32_t* __restrict__ totals) {
  uint32_t* c = counts + (size_t)blockIdx.x * n_blocks;
  __shared__ uint32_t part[RC_THREADS];
  const uint32_t per = (n_blocks + RC_THREADS - 1) / RC_THREADS;
  const uint32_t a = threadIdx.x * per, b = a + per < n_blocks ? a + per : n_blocks;
  uint32_t t = 0;
  for (uint32_t i = a; i < b; ++i) t += c[i];
  part[threadIdx.x] = t;
  __syncthreads();
  if (threadIdx.x == 0) {
    uint32_t run = 0;
    for (int i = 0; i < RC_THREADS; ++i) {
      const uint32_t v = part[i];
      part[i] = run;
      run += v;
    }
    totals[blockIdx.x] = run;
  }
  __syncthreads();
  uint32_t run = part[threadIdx.x];
  for (uint32_t i = a; i < b; ++i) {
    const uint32_t v = c[i];
    c[i] = run;
    run += v;
  }
}

// Pass 3: stable compaction (ray order kept, as numpy's boolean mask does).
static __global__ __launch_bounds__(RC_THREADS) void compact_kernel(const float* __restrict__ tmp,
                                                                   const unsigned long long* __restrict__ masks,
                                                                   const uint32_t* __restrict__ offsets, uint32_t n_rays,
                                                                   float* __restrict__ out) {
  const size_t ray_base = (size_t)blockIdx.y * ((size_t)gridDim.x * RC_THREADS);
  const uint32_t r = blockIdx.x * RC_THREADS + threadIdx.x;
  const int w = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const unsigned long long* mw = masks + (ray_base + blockIdx.x * RC_THREADS) / 64;
  uint32_t rank = offsets[(size_t)blockIdx.y * gridDim.x + blockIdx.x];
  for (int i = 0; i < w; ++i) rank += (uint32_t)__popcll(mw[i]);
  const unsigned long long m = mw[w];
  if (r < n_rays && ((m >> lane) & 1ull)) {
    rank += (uint32_t)__popcll(m & ((1ull << lane) - 1ull));
    const float* t = tmp + 3 * (ray_base + r);
    float* o = out + 3 * (ray_base + rank);
    o[0] = t[0];
    o[1] = t[1];
    o[2] = t[2];
  }
}

}  // namespace raycast
}  // namespace gloc
