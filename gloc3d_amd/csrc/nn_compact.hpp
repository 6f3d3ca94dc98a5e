// nn_compact.hpp -- exact 1-NN with spatial culling and compacted evaluation (K4, default mode).
//
// Same result, bit for bit, as the exhaustive nn_kernel (and the oracle): the distance that is returned
// is the un-fused fp32 ((dx dx + dy dy) + dz dz), ties go to the smallest ORIGINAL target index.  (The
// search itself compares FUSED distances, 2e-7 away at most, and re-decides contested minima un-fused:
// NN_NEAR.)  What changes is which pairs are evaluated:
//   * every scan is Hilbert-sorted once (scan_store.hip) and cut into chunks of 128 points, sub-blocks
//     of 16 and super-chunks of 64 chunks, each with an axis-aligned bounding box; a scan that serves as a
//     target (a database place) is re-sorted into kd order, whose chunks and sub-blocks are disjoint cells
//     (scan_index.hpp) -- nothing below depends on which order it is;
//   * a wave owns 64*CS consecutive (hence spatially compact) sorted source points, CS per lane;
//   * upper bounds come from the previous ICP pass's correspondence (warm start) or from the curve
//     neighbourhood of the point in the target's key order;
//   * 64 boxes at a time are tested against the wave's box by the 64 lanes (one ballot), the
//     survivors against each lane's own points; the sources that pass a chunk are listed, and a lane tests
//     one listed source against TWO of the chunk's sub-block boxes (stored interleaved for packed fp32);
//     every (source, sub-block) pair that can still hold a nearer (or equally near) point becomes a work
//     item in a wave-private LDS queue, consumed 64 items at a time: a lane walks ITS sub-block's 16 staged
//     targets and folds the minimum into the source's packed (d2 bits << 32 | sub-block << 2 | quarter) key
//     with an LDS atomic min; the index recovery re-reads the 4 targets of the winning quarter.
// A box is skipped only when a conservative lower bound of its distance exceeds the current best
// of the point (or of every point of the wave), so no candidate for the minimum (or for a tie) is missed.
//
// Everything is indexed in SORTED space: source slot i is the i-th point of the source's Hilbert
// order, a correspondence is the target's sorted position.  (Round 1 kept original indices: the warm
// start then cost three dependent random gathers per point and 5x the algorithmic HBM traffic.)
// The epilogue writes, besides corr / d2: the (moved source, matched target) pair for the RANSAC
// stage (PAIRS) and the wave's fp64 raw moments for the ICP step (one partial per wave, reduced in a
// fixed order by solve_kernel), so no separate gather / accumulate pass reads the points again.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <type_traits>

#include "lane_ops.hpp"
#include "reg_kernels.hpp"

namespace gloc {
namespace reg {

typedef float f32x2 __attribute__((ext_vector_type(2)));

#define GPTR(T) const T __attribute__((address_space(1)))*
// the same memory seen as constant: a load through it with a wave-uniform address is a scalar load (s_load), which
// costs no vector instruction -- the boxes of a scan do not change while a search runs
#define CPTR(T) const T __attribute__((address_space(4)))*
// relative window inside which two FUSED distances count as contested (nn_compact_kernel: NEAR); fused and un-fused
// (dx dx + dy dy) + dz dz differ by < 2e-7
constexpr float NN_NEAR = 1.0e-6f;
// a box's lower bound is scaled down by this before it is compared with a point's bound: covers the rounding of the
// bound itself (a few 2^-24) AND the bound being a fused distance, up to 2e-7 below the un-fused one (0.99999905 until
// the evaluation was fused)
constexpr float NN_LB_SCALE = 0.999998f;
#ifndef GLOC_NN_THIN_MIN
#define GLOC_NN_THIN_MIN 32  // survivors of a batch of 64 chunk boxes before the batch is thinned (nn_compact_kernel)
#endif
#ifdef GLOC_NN_MARKS
#define NN_MARK(n) asm volatile("; NN_MARK " n)
#else
#define NN_MARK(n)
#endif
#ifndef GLOC_NN_WPB
#define GLOC_NN_WPB 1  // waves per work-group; waves never synchronise with each other, and 1 measured 4 % faster than 4
#endif
constexpr int NN_WPB = GLOC_NN_WPB;
constexpr uint32_t NN_STAT_SLOTS = 4096;  // partial counters of the pairs-evaluated statistic
constexpr uint32_t NN_TRACE_WORDS = 32;   // dev trace: words per wave

// Wave-wide min / max without the LDS crossbar: four DPP steps inside every row of 16 lanes (quad
// swaps, then the half-row and row mirrors: after each step the lanes already paired hold one value,
// so a mirror reaches the same partner set as an xor), then the four row values through scalar registers.
template <int CTRL>
__device__ __forceinline__ float dpp_f32(float x) {
  return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), CTRL, 0xF, 0xF, true));
}
// Round 5: the six steps are v_min/v_max_f32 WITH the DPP operand, written out.  From fminf(x, dpp(x)) the compiler
// makes three instructions a step (v_mov_b32_dpp, a v_max x, x that quiets a signalling NaN, the v_min) and four for
// the two row steps: 22 vector instructions a reduction, seven reductions in a wave's prologue and one after most
// processed chunks -- 255 of the wave's 2 208 (ISA count).  v_min/v_max_f32 return the other operand when one is a NaN,
// as fminf / fmaxf do; the hazard (a VALU write of a register two or fewer slots before a DPP read of it) is covered
// by the s_nop the compiler cannot place inside an asm.
#if defined(__HIP_DEVICE_COMPILE__) && !defined(__gfx950__) && !defined(__gfx942__)
#error "wave_minmax / wave_box: hand-placed DPP hazards (row_bcast:15 / :31, s_nop padding) are written for gfx942 / gfx950 only"
#endif
// (the leading s_nop 4: five wait states cover a VALU write of EXEC right before the first DPP as well as a VALU write of
// the operand -- the compiler cannot see into the asm; the later steps read what the previous DPP wrote: two states)
template <bool MAX>
__device__ __forceinline__ float wave_minmax(float x) {
  // across the four rows: row_bcast:15 hands a row's value (all its lanes hold it) to the next row -- rows 1 and 3
  // take it -- then row_bcast:31 hands rows 0-1's to rows 2 and 3; lanes left out keep x.
  // Lane 63 ends with the wave's value: one v_readlane instead of four plus three combines.
  if constexpr (MAX)
    asm("s_nop 4\n\tv_max_f32_dpp %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
        "s_nop 1\n\tv_max_f32_dpp %0, %0, %0 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"
        "s_nop 1\n\tv_max_f32_dpp %0, %0, %0 row_half_mirror row_mask:0xf bank_mask:0xf\n\t"
        "s_nop 1\n\tv_max_f32_dpp %0, %0, %0 row_mirror row_mask:0xf bank_mask:0xf\n\t"
        "s_nop 1\n\tv_max_f32_dpp %0, %0, %0 row_bcast:15 row_mask:0xa bank_mask:0xf\n\t"
        "s_nop 1\n\tv_max_f32_dpp %0, %0, %0 row_bcast:31 row_mask:0xc bank_mask:0xf\n\ts_nop 0"
        : "+v"(x));
  else
    asm("s_nop 4\n\tv_min_f32_dpp %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
        "s_nop 1\n\tv_min_f32_dpp %0, %0, %0 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"
        "s_nop 1\n\tv_min_f32_dpp %0, %0, %0 row_half_mirror row_mask:0xf bank_mask:0xf\n\t"
        "s_nop 1\n\tv_min_f32_dpp %0, %0, %0 row_mirror row_mask:0xf bank_mask:0xf\n\t"
        "s_nop 1\n\tv_min_f32_dpp %0, %0, %0 row_bcast:15 row_mask:0xa bank_mask:0xf\n\t"
        "s_nop 1\n\tv_min_f32_dpp %0, %0, %0 row_bcast:31 row_mask:0xc bank_mask:0xf\n\ts_nop 0"
        : "+v"(x));
  return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(x), 63));
}
// The wave's box: three minima and three maxima, the six chains interleaved -- a register is read by a DPP operand
// five instructions after it was written, so only the first step needs the wait states.
__device__ __forceinline__ void wave_box(float (&lo)[3], float (&hi)[3]) {
#define GLOC_BOX_STEP(CTRL)                                      \
  "v_min_f32_dpp %0, %0, %0 " CTRL "\n\tv_max_f32_dpp %3, %3, %3 " CTRL "\n\t" \
  "v_min_f32_dpp %1, %1, %1 " CTRL "\n\tv_max_f32_dpp %4, %4, %4 " CTRL "\n\t" \
  "v_min_f32_dpp %2, %2, %2 " CTRL "\n\tv_max_f32_dpp %5, %5, %5 " CTRL "\n\t"
  asm("s_nop 4\n\t"
      GLOC_BOX_STEP("quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf")
      GLOC_BOX_STEP("quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf")
      GLOC_BOX_STEP("row_half_mirror row_mask:0xf bank_mask:0xf")
      GLOC_BOX_STEP("row_mirror row_mask:0xf bank_mask:0xf")
      GLOC_BOX_STEP("row_bcast:15 row_mask:0xa bank_mask:0xf")
      GLOC_BOX_STEP("row_bcast:31 row_mask:0xc bank_mask:0xf")
      "s_nop 0"
      : "+v"(lo[0]), "+v"(lo[1]), "+v"(lo[2]), "+v"(hi[0]), "+v"(hi[1]), "+v"(hi[2]));
#undef GLOC_BOX_STEP
#pragma unroll
  for (int a = 0; a < 3; ++a) {
    lo[a] = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(lo[a]), 63));
    hi[a] = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(hi[a]), 63));
  }
}

// (xor_lane<O>: lane_ops.hpp -- DPP below 16 lanes; ds_bpermute kept the LDS pipe busy and was worth 6 % of the launch)
// The reduce-scatter exchange at distance 32 (16): a lane with that bit clear keeps x and hands y to its
// partner, a lane with it set keeps y and hands x over.  gfx950's v_permlane32_swap (v_permlane16_swap)
// moves the handed-over halves in place: afterwards EVERY lane holds (kept, received) in some order in
// (x, y), so the node's sum is x + y (addition commutes: same bits as kept + received).
template <int O>
__device__ __forceinline__ double exchange_add(double x, double y) {
  static_assert(O == 32 || O == 16, "whole rows");
  typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
  const unsigned long long ux = __builtin_bit_cast(unsigned long long, x), uy = __builtin_bit_cast(unsigned long long, y);
  u32x2 lo, hi;
  if (O == 32) {
    lo = __builtin_amdgcn_permlane32_swap((unsigned)ux, (unsigned)uy, false, false);
    hi = __builtin_amdgcn_permlane32_swap((unsigned)(ux >> 32), (unsigned)(uy >> 32), false, false);
  } else {
    lo = __builtin_amdgcn_permlane16_swap((unsigned)ux, (unsigned)uy, false, false);
    hi = __builtin_amdgcn_permlane16_swap((unsigned)(ux >> 32), (unsigned)(uy >> 32), false, false);
  }
  const double nx = __builtin_bit_cast(double, ((unsigned long long)hi.x << 32) | lo.x);
  const double ny = __builtin_bit_cast(double, ((unsigned long long)hi.y << 32) | lo.y);
  return nx + ny;
}

template <int O>
__device__ __forceinline__ float exchange_add(float x, float y) {
  static_assert(O == 32 || O == 16, "whole rows");
  typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
  u32x2 r;
  if (O == 32) r = __builtin_amdgcn_permlane32_swap(__float_as_uint(x), __float_as_uint(y), false, false);
  else r = __builtin_amdgcn_permlane16_swap(__float_as_uint(x), __float_as_uint(y), false, false);
  return __uint_as_float(r.x) + __uint_as_float(r.y);
}

// grid = n_wg * n_jobs work-groups of 4 independent waves.  Jobs are taken `job_group` at a time; within
// a group the job index runs fastest (every job's widest source groups -- `order` lists them widest
// first -- start together and finish under cover of the bulk), so that at any time the work-groups in
// flight touch the scans of about one group of jobs.  Work-groups are dealt round-robin over the 8 XCDs,
// so with job_group a multiple of 8 all work-groups of a job run on ONE XCD and its candidate scan (2 MB of
// points + boxes) stays in that XCD's 4 MB L2: measured L2-miss traffic per launch of 500 jobs 3.4 GB at
// job_group 60, 1.07 GB at 24 (FETCH_SIZE; profiles/r02_*), i.e. 1.1x the algorithmic bytes.
// WARM: every pass after a batch's first -- prev_corr holds the previous correspondences, and the cold start's code (a
// search in the target's curve keys) is not in the kernel at all: measured, its mere presence cost the warm passes
// -- 92 % of the 1-NN time -- 3 % (registers, code layout).
// ---- the chained launch (NnChain, reg_kernels.hpp) -----------------------------------------------------------------
// A bounded wait for *p >= need, by the whole wave (lane 0 polls).  false: a wait ran out somewhere (err is set).
// FRESH: the word is at an address nobody reads before it can have its final value's predecessor written through -- the
// first look is an ordinary load (served by the XCD's L2 to the 1 200 other waves of the job; a copy from before the
// value was complete only sends the wave on to the loads past the caches).
// Returns 0: a wait ran out; 1: the ordinary look sufficed; 2: the first look past the caches; 3: after polling.
template <bool FRESH>
__device__ __forceinline__ int chain_wait(const uint32_t* p, uint32_t need, uint32_t* err) {
  const int lane = threadIdx.x & 63;
  uint32_t v = 0;
  if constexpr (FRESH) {
    asm volatile("global_load_dword %0, %1, off\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
    if (__builtin_amdgcn_readfirstlane(v) >= need) return 1;
    v = 0;
  }
  if (lane == 0) v = ld_u32<true>(p);
  if (__builtin_amdgcn_readfirstlane(v) >= need) return 2;
  const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
  for (;;) {
    __builtin_amdgcn_s_sleep(16);
    uint32_t e = 0;
    if (lane == 0) {
      v = ld_u32<true>(p);
      e = ld_u32<true>(err);
    }
    if (__builtin_amdgcn_readfirstlane(v) >= need) return 3;
    if (__builtin_amdgcn_readfirstlane(e) != 0u) return 0;
    if (__builtin_amdgcn_s_memrealtime() - t0 > NN_CHAIN_WAIT_TICKS) {
      if (lane == 0) st_u32<true>(err, 1u);
      return 0;
    }
  }
}

// dev: a stamp of the 100 MHz clock in slot k of (pass, job) -- first / last writer as `mode` says
__device__ __forceinline__ void chain_stamp(const NnChain& ch, uint32_t pass, uint32_t job, uint32_t n_jobs, int k, int mode /* 0 store, 1 min, 2 max */) {
  if (ch.dbg && (threadIdx.x & 63) == 0) {
    uint32_t* p = ch.dbg + ((size_t)pass * n_jobs + job) * 16 + k;
    const uint32_t t = (uint32_t)__builtin_amdgcn_s_memrealtime();
    if (mode == 0) *p = t;
    else if (mode == 1) atomicMin(p, t);
    else atomicMax(p, t);
  }
}

// everything this wave has stored or added is where the other XCDs see it (acknowledged), THEN the counter
__device__ __forceinline__ uint32_t chain_arrive(uint32_t* counter) {
  __builtin_amdgcn_s_waitcnt(0);
  asm volatile("" ::: "memory");
  uint32_t a = 0;
  if ((threadIdx.x & 63) == 0) a = __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  return __builtin_amdgcn_readfirstlane(a);
}

// Reducer r of a job (solve_kernel's wave r): the partials r * 64 + lane, + 1024, ... in that order, the xor butterfly,
// the sub-sum stored; the reducer that arrives last adds the 16 sub-sums in order and solves -- the additions of
// solve_kernel<0>, one for one.
__device__ __forceinline__ void chain_reduce_solve(uint32_t job, uint32_t pass, uint32_t r, uint32_t n_jobs, const Job* __restrict__ jobs,
                                                   CandState* states, const double* partials, uint32_t n_part, const NnChain& ch,
                                                   double* ws /* LDS, this wave's */) {
  const int lane = threadIdx.x & 63;
  const uint32_t cnt = jobs[job].n_groups;
  const float* base = reinterpret_cast<const float*>(partials + (size_t)job * n_part * ACC_NV);
  double v[ACC_NV];
#pragma unroll
  for (int k = 0; k < ACC_NV; ++k) v[k] = 0.0;
  for (uint32_t g = r * 64u + (uint32_t)lane; g < cnt; g += SOLVE_THREADS) {
    const unsigned long long* f2 = reinterpret_cast<const unsigned long long*>(base + (size_t)g * (2 * ACC_NV));
    float f[WAVE_PARTIAL_FLOATS + 1];
#pragma unroll
    for (int i = 0; i < (WAVE_PARTIAL_FLOATS + 1) / 2; ++i) {
      const unsigned long long t = ld_u64<true>(f2 + i);
      f[2 * i] = __uint_as_float((uint32_t)t);
      f[2 * i + 1] = __uint_as_float((uint32_t)(t >> 32));
    }
    add_wave_partial(f, v);
  }
  double* sub = ch.sub + ((size_t)job * NN_CHAIN_RED + r) * ACC_NV;
#pragma unroll
  for (int k = 0; k < ACC_NV; ++k) {
    double x = v[k];
    x += xor_lane<32>(x);
    x += xor_lane<16>(x);
    x += xor_lane<8>(x);
    x += xor_lane<4>(x);
    x += xor_lane<2>(x);
    x += xor_lane<1>(x);
    if (lane == 0) st_f64<true>(sub + k, x);
  }
  if (r == 0) chain_stamp(ch, pass, job, n_jobs, 5, 0);  // reducer 0 has stored its sub-sum
  if (chain_arrive(ch.sdone + (size_t)pass * n_jobs + job) != NN_CHAIN_RED - 1u) return;
  chain_stamp(ch, pass, job, n_jobs, 6, 0);  // the last reducer is in
  // the 16 sub-sums in order, a lane per moment; the state's fp64 pose beside them (loads past the caches: all in flight together)
  {
    const double* all = ch.sub + (size_t)job * NN_CHAIN_RED * ACC_NV;
    double acc = 0.0, td_in = 0.0;
    uint32_t frozen = 0;
    if (lane < ACC_NV) {
      double x[NN_CHAIN_RED];
#pragma unroll
      for (uint32_t rr = 0; rr < NN_CHAIN_RED; ++rr) x[rr] = ld_f64<true>(all + (size_t)rr * ACC_NV + lane);
#pragma unroll
      for (uint32_t rr = 0; rr < NN_CHAIN_RED; ++rr) acc += x[rr];
    } else if (lane < ACC_NV + 12) {
      td_in = ld_f64<true>(&states[job].Td[lane - ACC_NV]);
    } else if (lane == ACC_NV + 12) {
      frozen = ld_u32<true>(reinterpret_cast<const uint32_t*>(&states[job].frozen));
    }
    if (lane < ACC_NV) ws[lane] = acc;
    else if (lane < ACC_NV + 12) ws[lane] = td_in;
    frozen = (uint32_t)__builtin_amdgcn_readlane((int)frozen, ACC_NV + 12);
    __builtin_amdgcn_s_waitcnt(0);  // (the LDS stores before lane 0 reads them: one wave, in order -- the counter makes it explicit)
    asm volatile("" ::: "memory");
    chain_stamp(ch, pass, job, n_jobs, 7, 0);  // sums and pose loaded
    // (the next pass's pose goes to its own address too: solve_compose's T_also -- a job whose pose does not change any
    // more, frozen or short of correspondences, hands on the one it was given)
    float* t_next = pass + 1u < ch.n_pass ? ch.Tp + ((size_t)(pass + 1u) * n_jobs + job) * NN_CHAIN_T_STRIDE : nullptr;
    const float* t_now = pass ? ch.Tp + ((size_t)pass * n_jobs + job) * NN_CHAIN_T_STRIDE : states[job].Tf;
    if (lane == 0) solve_compose<0, true, true>(ws, states + job, ws + ACC_NV, frozen, ws + ACC_NV + 12, t_next, t_now);
    chain_stamp(ch, pass, job, n_jobs, 8, 0);  // solved
  }
  if (pass + 1u < ch.n_pass) (void)chain_arrive(ch.ready + ((size_t)(pass + 1u) * n_jobs + job) * NN_CHAIN_PAD);
}

// The plan of the job's NEXT pass by one wave: solve_kernel's planner, 64 groups at a time (the same table: first the
// groups that get 4 or 8 waves, then those that get 2, each class in launch order).
__device__ __forceinline__ void chain_plan(uint32_t job, uint32_t pass, uint32_t n_jobs, const Job* __restrict__ jobs, const NnSplit& sp,
                                           uint32_t n_part, const NnChain& ch) {
  const int lane = threadIdx.x & 63;
  if (sp.hx) {
    constexpr uint32_t REACH = 8 * SOLVE_THREADS;  // (solve_kernel's PLAN_TILES)
    const uint32_t n_groups = jobs[job].n_groups, ng = n_groups < REACH ? n_groups : REACH;
    uint32_t* work = sp.work + (size_t)job * n_part;
    // (the tables of pass + 1, at their own addresses; behind the last pass nobody reads a plan: the batch's own arrays take it)
    const bool last = pass + 1u >= ch.n_pass;
    uint32_t* plan = (last ? sp.plan : ch.planp + (size_t)pass * n_jobs * n_part) + (size_t)job * n_part;
    uint32_t* helper = (last ? sp.helper : ch.helperp + (size_t)pass * n_jobs * sp.hx) + (size_t)job * sp.hx;
    // (the estimates of 1024 groups at a time, 16 loads a lane in flight together: one round trip a tile, not one per 64 groups)
    uint32_t tot_h = 0, tot_gh = 0;
    for (uint32_t t0 = 0; t0 < ng; t0 += SOLVE_THREADS) {
      uint32_t wk[16];
#pragma unroll
      for (int c = 0; c < 16; ++c) {
        const uint32_t g = t0 + 64u * (uint32_t)c + (uint32_t)lane;
        wk[c] = g < ng ? ld_u32<true>(work + g) : 0u;
      }
#pragma unroll
      for (int c = 0; c < 16; ++c) {
        const uint32_t parts = nn_parts_for(wk[c], sp.thresh);  // (0 beyond ng: an estimate of 0)
        const unsigned long long b4 = __ballot(parts == 4), b8 = __ballot(parts == 8);
        tot_h += 4u * (uint32_t)__popcll(b4) + 8u * (uint32_t)__popcll(b8);
        tot_gh += (uint32_t)__popcll(b4 | b8);
      }
    }
    for (uint32_t e = (uint32_t)lane; e < sp.hx; e += 64) st_u32<true>(helper + e, NN_NO_HELPER);  // (slots nobody takes below)
    __builtin_amdgcn_s_waitcnt(0);  // (before another lane's entry for the same slot)
    asm volatile("" ::: "memory");
    uint32_t run_h = 0, run_l = tot_h, run_gh = 0, run_gl = tot_gh;
    const unsigned long long below = (1ull << lane) - 1ull;
    for (uint32_t t0 = 0; t0 < ng; t0 += SOLVE_THREADS) {
      uint32_t wk[16];
#pragma unroll
      for (int c = 0; c < 16; ++c) {
        const uint32_t g = t0 + 64u * (uint32_t)c + (uint32_t)lane;
        wk[c] = g < ng ? ld_u32<true>(work + g) : 0u;
      }
#pragma unroll
      for (int c = 0; c < 16; ++c) {
        const uint32_t g = t0 + 64u * (uint32_t)c + (uint32_t)lane;
        const uint32_t parts = nn_parts_for(wk[c], sp.thresh);
        if (g < ng) st_u32<true>(work + g, 0u);  // consumed
        const unsigned long long b2 = __ballot(parts == 2), b4 = __ballot(parts == 4), b8 = __ballot(parts == 8);
        const uint32_t pre = parts >= 4 ? 4u * (uint32_t)__popcll(b4 & below) + 8u * (uint32_t)__popcll(b8 & below) : 2u * (uint32_t)__popcll(b2 & below);
        const uint32_t gpre = parts >= 4 ? (uint32_t)__popcll((b4 | b8) & below) : (uint32_t)__popcll(b2 & below);
        const uint32_t first = (parts >= 4 ? run_h : run_l) + pre, hid = (parts >= 4 ? run_gh : run_gl) + gpre;
        if (g < ng) {
          uint32_t word = 0;
          if (parts > 1 && first + parts <= sp.hx) {  // (groups past the last slot keep their one wave at their own rank)
            for (uint32_t p = 0; p < parts; ++p) st_u32<true>(helper + first + p, g | (p << 20) | (parts << 24));
            word = (hid << 8) | parts;
          }
          st_u32<true>(plan + g, word);
        }
        run_h += 4u * (uint32_t)__popcll(b4) + 8u * (uint32_t)__popcll(b8);
        run_l += 2u * (uint32_t)__popcll(b2);
        run_gh += (uint32_t)__popcll(b4 | b8);
        run_gl += (uint32_t)__popcll(b2);
      }
    }
    for (uint32_t g = ng + (uint32_t)lane; g < n_groups; g += 64) st_u32<true>(plan + g, 0u);  // (beyond the planner's reach)
  }
  if (pass + 1u < ch.n_pass) (void)chain_arrive(ch.ready + ((size_t)(pass + 1u) * n_jobs + job) * NN_CHAIN_PAD);
}


struct NnPos {
  uint32_t grp, slot, wgv;
  uint32_t role = 0, pass = 0, job = 0;  // chained launch only: 0 a search wave; 1 + r: reducer r of `job`; 1 + NN_CHAIN_RED: its planner
};

template <int CS, bool PAIRS, bool TRACE = false, bool SPLIT = false /* with the plan for heavy groups (NnSplit; sp.hx > 0) */,
          bool WARM = false, bool HEAVY = false /* the second launch of a cold pass: the groups its waves gave up (NnHeavy) */,
          bool CHAIN = false /* a pass of the chained launch (NnChain): what other waves of the SAME launch wrote or will read goes past the caches */>
__device__ __forceinline__ void nn_compact_body(
    const NnPos pos /* the work-group's place in the launch order of ONE pass: blockIdx of the plain launch */,
    const Job* __restrict__ jobs, uint32_t n_jobs, uint32_t job_group, uint32_t n_wg /* per slot */, uint32_t subs,
    const CandState* __restrict__ states,
    const uint32_t* prev_corr /* may alias corr; null: cold start */, uint32_t* corr, float* __restrict__ d2out,
    f32x4* __restrict__ pairs, double* __restrict__ partials /* [job][n_part][ACC_NV] */, uint32_t n_part,
    size_t ld, float gate2, NnSplit sp, NnHeavy hv, unsigned long long* __restrict__ stat_pairs /* [NN_STAT_SLOTS] pairs evaluated, or null */,
    uint32_t* __restrict__ trace /* dev only: [wave][NN_TRACE_WORDS]: counts in words 0-7, cycles per region in 8-19 (tools/dev_nn_trace3.py) */,
    const NnChain ch = NnChain{}) {
  constexpr int S = 64 * CS;        // sources per wave
  constexpr int NSB = CH / SB;      // sub-blocks per chunk
  // The staged chunk is kept as PAIRS of targets, structure-of-arrays: pair i of a sub-block is
  // (x0, x1, y0, y1 | z0, z1, -, -), 32 B, so that one packed fp32 instruction handles two targets
  // (v_pk_add/mul_f32 round each half like the scalar forms: same bits).  Sub-blocks are 288 B apart:
  // the extra 32 B shift sub-block b's pair i into bank group (i + b) % 8, so lanes that walk
  // different sub-blocks in lock step never collide.
  constexpr int SB_STRIDE = (SB / 2) * 8 + 8;  // floats per sub-block: SB/2 pairs x 8 floats + 8 of shift
  // queue entries: a chunk queues 51 items on average; a test step adds at most 128 to the at most 63 that wait for a
  // full round (below).  A wave's LDS: 6.3 KB.
#ifndef GLOC_NN_QCAP
#define GLOC_NN_QCAP 256
#endif
#ifndef GLOC_NN_GRAN
#define GLOC_NN_GRAN 64  // items evaluated between two test steps come in multiples of this (64: full rounds; 32, 16: also a two- / four-lane round)
#endif
  constexpr int QCAP = GLOC_NN_QCAP;
  struct WaveLds {
    float stage[NSB * SB_STRIDE];   // the chunk being evaluated
    f32x4 src[S + 1];               // moved source points; [S]: a dummy the padded tail of the list points at
    unsigned long long key[S + 1];  // (bits(best d2) << 32) | (sub-block holding it << 2) | its quarter; [S]: bound -1 (nothing passes)
    uint8_t tie[S];                 // (CS = 2: 6.3 KB per wave, 78 VGPRs: six waves per SIMD)
    uint16_t list[S + 16];          // source slots that passed the chunk-level test, padded to a multiple of 16 with S
    uint16_t queue[QCAP];           // work items: (source slot << 3) | sub-block within the chunk
#ifdef GLOC_NN_LDS_PAD
    uint8_t pad_[GLOC_NN_LDS_PAD];  // dev: occupancy experiments
#endif
  };
  static_assert(SB % 16 == 0 && NSB == 8, "items pack the sub-block into 3 bits");
  __shared__ WaveLds lds_all[NN_WPB];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  WaveLds& L = lds_all[w];
  if constexpr (CHAIN) {
    // a role wave of the chained launch: its arrays live in this wave's (otherwise unused) staging area -- the solve has
    // to fit the search's register budget
    static_assert(sizeof(L.stage) >= 160 * sizeof(double), "the solve's workspace");
    if (pos.role != 0u) {
      if (pos.role <= NN_CHAIN_RED)
        chain_reduce_solve(pos.job, pos.pass, pos.role - 1u, n_jobs, jobs, const_cast<CandState*>(states), partials, n_part, ch,
                           reinterpret_cast<double*>(L.stage));
      else
        chain_plan(pos.job, pos.pass, n_jobs, jobs, sp, n_part, ch);
      return;
    }
  }
  // blockIdx -> (job, work-group of the job).  The launch order walks `job_group` SLOTS at a time, slot fastest, and
  // work-groups go to the XCDs round-robin (XCD = blockIdx % 8, every XCD working through its own share independently):
  // with job_group a multiple of 8 a slot stays on one XCD.  A slot holds one job -- or, with `subs` > 1 (few jobs:
  // fewer than the XCDs can balance), one of `subs` interleaved shares of a job's work-groups.  Which job a slot holds
  // rotates: within a row of 8 slots by the row number, and from one group of slots to the next -- any regular pattern
  // in the jobs' costs (bench.py: every fourth candidate is from a different world, 22 % dearer) would otherwise
  // load the same XCDs in every group: measured, XCDs 1 and 5 of 8 carried every such job and the launch waited for them.
  // (round 5: the grid is (slots of a group, work-groups of a slot, groups) -- the same linear order, slot fastest, and the
  // three divisions of a one-dimensional block index by run-time values are gone: 12 vector + 60 scalar instructions a wave)
  static_assert(!HEAVY || (SPLIT && !WARM && NN_WPB == 1), "the redo of a cold pass's heavy groups folds its parts as the planned split does");
  uint32_t lin_block, job, wg = 0, heavy_e = 0;
  if constexpr (HEAVY) {
    // all parts of a list entry on ONE XCD (work-groups go to the XCDs round-robin: XCD = blockIdx % 8), so that the
    // target's boxes and points are fetched into one L2, not eight: block b -> XCD x = b % 8, part (b / 8) % PARTS,
    // entry 8 * (b / 8 / PARTS) + x.  (The list's length is read here: the host never sees it.)
    lin_block = blockIdx.x;
    heavy_e = 8u * (blockIdx.x / (8u * NN_HEAVY_PARTS)) + (blockIdx.x & 7u);
    const uint32_t n_listed = *(CPTR(uint32_t))hv.count;
    if (heavy_e >= (n_listed < hv.cap ? n_listed : hv.cap)) return;
    job = ((CPTR(uint32_t))hv.list)[2 * heavy_e];
  } else {
    const uint32_t grp = pos.grp, slot = pos.slot, wgv = pos.wgv;
    lin_block = slot + job_group * (wgv + n_wg * grp);
    uint32_t vin = slot;  // the virtual job of the slot, within the group
    if ((job_group & 7u) == 0u) vin = (slot & ~7u) | ((slot - (slot >> 3) - grp) & 7u);
    const uint32_t vjob = grp * job_group + vin;
    if (vjob >= n_jobs * subs) return;
    job = vjob;
    wg = wgv;
    if (subs != 1u) {  // (uniform; shares of a job only in small batches: the usual launch pays no division)
      job = vjob / subs;
      wg = wgv * subs + vjob % subs;
    }
  }
  const Job& J = jobs[job];
  const uint32_t n_src = J.n_src;
  // the first sp.hx waves of a job are helpers (they start with the job's widest groups); then one wave per group
  uint32_t gi, part = 0, parts = 1, hid = 0, plan_word = 0;
  bool own_wave = true;  // the group's own wave at its rank (not a helper)
  if constexpr (HEAVY) {
    gi = ((CPTR(uint32_t))hv.list)[2 * heavy_e + 1];
    part = (blockIdx.x >> 3) % NN_HEAVY_PARTS;
    parts = NN_HEAVY_PARTS;
    hid = heavy_e;
    own_wave = false;
    if (gi >= J.n_groups) return;
  } else {
    const uint32_t wv = __builtin_amdgcn_readfirstlane(wg * NN_WPB + w);
    if (SPLIT && wv < sp.hx) {
      const uint32_t hw = ((CPTR(uint32_t))sp.helper)[(size_t)job * sp.hx + wv];  // (wave-uniform: scalar loads)
      if (hw == NN_NO_HELPER) return;
      gi = hw & 0xFFFFFu;
      part = (hw >> 20) & 15u;
      parts = hw >> 24;
      own_wave = false;
    } else {
      gi = wv - (SPLIT ? sp.hx : 0u);
    }
    if (gi >= J.n_groups) return;  // whole wave idle (no work-group barriers are used below)
    // (looked at below, once the wave's other loads are in flight: a dependent round trip at the head of every wave
    // -- 1 us of its 19 -- otherwise)
    if constexpr (SPLIT) plan_word = ((CPTR(uint32_t))sp.plan)[(size_t)job * n_part + gi];
  }
  // candidate chunks of this part: bit b of a batch of 64 chunk boxes (the batch starts at a multiple of 64)
  const unsigned long long pmask =
      !SPLIT || parts == 1 ? ~0ull : ((parts == 2 ? 0x5555555555555555ull : (parts == 4 ? 0x1111111111111111ull : (parts == 8 ? 0x0101010101010101ull : 0x0001000100010001ull))) << part);
  uint32_t w_cand = 0;
  struct IndexView {
    GPTR(f32x4) pts; GPTR(f32x4) box_lo; GPTR(f32x4) box_hi; GPTR(f32x4) sb2;
    GPTR(uint32_t) keys; GPTR(uint32_t) kpos; GPTR(uint32_t) inv; GPTR(ScanHeader) hdr;
    uint32_t n, nchunks;
    GPTR(f32x4) sup_lo; GPTR(f32x4) sup_hi;
    uint32_t nsup;
    CPTR(f32x4) cbox_lo; CPTR(f32x4) cbox_hi;
  };
  // The index arrays are reached through pointers loaded from memory, which the compiler would
  // treat as generic (flat_load): view them in the global address space explicitly.
  const ScanIndexDev ixg = J.tgt;
  const IndexView ix{(GPTR(f32x4))ixg.pts, (GPTR(f32x4))ixg.box_lo, (GPTR(f32x4))ixg.box_hi,
                     (GPTR(f32x4))ixg.sb2, (GPTR(uint32_t))ixg.keys, (GPTR(uint32_t))ixg.kpos,
                     (GPTR(uint32_t))ixg.inv, (GPTR(ScanHeader))ixg.hdr, ixg.n, ixg.nchunks,
                     (GPTR(f32x4))ixg.sup_lo, (GPTR(f32x4))ixg.sup_hi, ixg.nsup,
                     (CPTR(f32x4))ixg.box_lo, (CPTR(f32x4))ixg.box_hi};
  GPTR(f32x4) src4 = (GPTR(f32x4))J.src_pts;
  float T[12];
#pragma unroll
  for (int i = 0; i < 12; ++i) {
    // (chained: the pose of THIS pass at an address nobody has read before its solve wrote it -- an ordinary load)
    if constexpr (CHAIN) T[i] = pos.pass ? ((CPTR(float))ch.Tp)[((size_t)pos.pass * n_jobs + job) * NN_CHAIN_T_STRIDE + i] : states[job].Tf[i];
    else T[i] = states[job].Tf[i];
  }

  const uint32_t wave_base = ((CPTR(uint32_t))J.src_order)[gi] * S;
  if constexpr (SPLIT && !HEAVY) {
    if (own_wave && (plan_word & 0xFFu) != 0u) return;  // the helpers have this group, all its parts
    hid = plan_word >> 8;
  }
  const unsigned long long t_start = TRACE ? __builtin_amdgcn_s_memtime() : 0ull;
  const unsigned long long rt_start = TRACE ? __builtin_amdgcn_s_memrealtime() : 0ull;
  uint32_t n_processed = 0, n_rounds = 0, n_live_sb = 0, n_live_pairs = 0, n_steps = 0, n_cand = 0, round_hist = 0, n_ties = 0;
  unsigned long long n_items = 0;
  // dev trace: cycles per region (s_memtime; a wave's wall clock among the other waves of its SIMD)
  auto now = [&]() { return TRACE ? __builtin_amdgcn_s_memtime() : 0ull; };
  unsigned long long a_cand = 0, a_thin = 0, a_stage = 0, a_tests = 0, a_rounds = 0, a_refresh = 0, a_batch = 0;

  NN_MARK("load_xform_box");
  float px[CS], py[CS], pz[CS], best[CS];
  bool valid[CS];
  float wlo[3] = {3.4e38f, 3.4e38f, 3.4e38f}, whi[3] = {-3.4e38f, -3.4e38f, -3.4e38f};
  // The wave's loads in as few dependent round trips as there are: the previous pass's correspondences (warm start) go
  // out with the source points, the gather of the matched targets as soon as they are back -- before the wave's box is
  // reduced (48 DPP instructions that need no memory).  (Until round 4: points, box, THEN correspondences, THEN the
  // gather -- two more exposed round trips at the head of every wave, 1 - 2 us of its 19.)
  uint32_t pj[CS];
  f32x4 pt[CS];
#pragma unroll
  for (int s = 0; s < CS; ++s) {
    pj[s] = 0xFFFFFFFFu;
    if (prev_corr && ix.n) pj[s] = prev_corr[(size_t)job * ld + (wave_base + s * 64 + lane < n_src ? wave_base + s * 64 + lane : 0)];
  }
  f32x4 psrc[CS];
#pragma unroll
  for (int s = 0; s < CS; ++s) {
    const uint32_t i = wave_base + s * 64 + lane;
    valid[s] = i < n_src;
    psrc[s] = src4[valid[s] ? i : (n_src - 1)];
  }
#pragma unroll
  for (int s = 0; s < CS; ++s) {
    pt[s] = f32x4{0.f, 0.f, 0.f, 0.f};
    if (pj[s] < ix.n) pt[s] = ix.pts[pj[s]];
  }
#pragma unroll
  for (int s = 0; s < CS; ++s) {
    const f32x4 p = psrc[s];
    xform(T, p.x, p.y, p.z, px[s], py[s], pz[s]);
    wlo[0] = fminf(wlo[0], px[s]); whi[0] = fmaxf(whi[0], px[s]);
    wlo[1] = fminf(wlo[1], py[s]); whi[1] = fmaxf(whi[1], py[s]);
    wlo[2] = fminf(wlo[2], pz[s]); whi[2] = fmaxf(whi[2], pz[s]);
    best[s] = 3.402823466e+38f;
  }
  wave_box(wlo, whi);

  const unsigned long long t_box = now();
  NN_MARK("upper_bounds");
  // ---- upper bounds -> LDS state -----------------------------------------------------------------
  // warm: the previous pass's correspondence (a sorted position: one coherent 16-byte gather);
  // cold: the five curve neighbours of the point's key in the target's order
  uint32_t b0s[CS];
#pragma unroll
  for (int s = 0; s < CS; ++s) {
    b0s[s] = 0;
    if (ix.n && pj[s] < ix.n) {
      const f32x4 t = pt[s];
      best[s] = dist2(px[s], py[s], pz[s], t.x, t.y, t.z);
      b0s[s] = pj[s] / (SB / 4);  // the key's low word: (sub-block << 2) | quarter of the sub-block
    }
  }
  uint8_t tie0[CS];
#pragma unroll
  for (int s = 0; s < CS; ++s) tie0[s] = 0;
  if constexpr (HEAVY) {  // the bounds (and contested flags) of the wave that gave the group up: hv.hkey
#pragma unroll
    for (int s = 0; s < CS; ++s) {
      const unsigned long long k = hv.hkey[(size_t)hid * S + s * 64 + lane];
      best[s] = __uint_as_float((uint32_t)(k >> 32));
      b0s[s] = (uint32_t)k & 0x7FFFFFFFu;
      tie0[s] = (uint8_t)(((uint32_t)k >> 31) & 1u);
    }
  } else if constexpr (!WARM) {  // (a warm pass: a point that found no neighbour -- a NaN -- searches without a bound)
    // lower_bound over the sorted keys, (KP + 1)-way: KP independent probes per round trip.  The binary search took 17
    // dependent trips at 124 k keys (46 % of a cold wave's time, traced); nine-way (8 probes, 7 trips, 56 loads) made the
    // memory pipe the limit (every lane its own addresses: 40 %); five-way is 8 trips of 4.  The lane's CS sources search
    // TOGETHER (round 4, late): one after the other they were 2 x 10 dependent trips, 40 % of a cold wave's time -- cold launch
    // of 500 jobs 3.47 -> 3.07 ms; with the trips shared the loads count again: 3 / 4 / 6 / 8 probes 2.96 / 3.07 / 3.32 / 3.22 ms.
#ifndef GLOC_NN_KEY_PROBES
#define GLOC_NN_KEY_PROBES 3
#endif
    constexpr int KP = GLOC_NN_KEY_PROBES;
    bool cold[CS];
    uint32_t key[CS], lo[CS], hi[CS];
    bool any = false;
#pragma unroll
    for (int s = 0; s < CS; ++s) {
      cold[s] = ix.n && !(pj[s] < ix.n);
      key[s] = morton_key(px[s], py[s], pz[s], ix.hdr->ox, ix.hdr->oy, ix.hdr->oz, ix.hdr->inv_cell);
      lo[s] = 0;
      hi[s] = cold[s] ? ix.n : 0u;  // the answer is in [lo, hi]; a source with a bound searches nothing
      any |= cold[s];
    }
    if (any) {
      for (;;) {
        bool go = false;
#pragma unroll
        for (int s = 0; s < CS; ++s) go |= hi[s] - lo[s] > (uint32_t)KP;
        if (!go) break;
        uint32_t pos[CS][KP], kv[CS][KP];
#pragma unroll
        for (int s = 0; s < CS; ++s) {
          const uint32_t len = hi[s] - lo[s];
#pragma unroll
          for (int i = 0; i < KP; ++i) {  // lo < pos < hi, ascending (an interval already short: probes it will not use)
            pos[s][i] = lo[s] + (uint32_t)(((unsigned long long)len * (uint32_t)(i + 1)) / (uint32_t)(KP + 1));
            pos[s][i] = pos[s][i] < ix.n ? pos[s][i] : ix.n - 1;
          }
        }
#pragma unroll
        for (int s = 0; s < CS; ++s)
#pragma unroll
          for (int i = 0; i < KP; ++i) kv[s][i] = ix.keys[pos[s][i]];
#pragma unroll
        for (int s = 0; s < CS; ++s) {
          if (!(hi[s] - lo[s] > (uint32_t)KP)) continue;
          uint32_t nlo = lo[s], nhi = hi[s];
#pragma unroll
          for (int i = KP - 1; i >= 0; --i)
            if (!(kv[s][i] < key[s])) nhi = pos[s][i];  // the first probe that is not below the key bounds the answer from above
#pragma unroll
          for (int i = 0; i < KP; ++i)
            if (kv[s][i] < key[s]) nlo = pos[s][i] + 1;  // the last probe below it, from below
          lo[s] = nlo;
          hi[s] = nhi;
        }
      }
      uint32_t kl[CS][KP];
#pragma unroll
      for (int s = 0; s < CS; ++s)
#pragma unroll
        for (int i = 0; i < KP; ++i) kl[s][i] = (lo[s] + i < hi[s]) ? ix.keys[lo[s] + i] : 0xFFFFFFFFu;
#pragma unroll
      for (int s = 0; s < CS; ++s) {
        uint32_t below = 0;
#pragma unroll
        for (int i = 0; i < KP; ++i)
          if (lo[s] + i < hi[s] && kl[s][i] < key[s]) below++;
        lo[s] += below;  // (sorted: the keys below the key are a prefix of [lo, hi))
      }
      // the five curve neighbours of the key's place
      uint32_t pjj[CS][5];
#pragma unroll
      for (int s = 0; s < CS; ++s)
#pragma unroll
        for (int d = -2; d <= 2; ++d) {
          long long jj = (long long)lo[s] + d;
          jj = jj < 0 ? 0 : (jj >= (long long)ix.n ? (long long)ix.n - 1 : jj);
          // (a target index is in kd order: the curve neighbour's place among the points comes from kpos)
          pjj[s][d + 2] = !cold[s] ? 0u : (ix.kpos ? ix.kpos[jj] : (uint32_t)jj);
        }
      f32x4 tn[CS][5];
#pragma unroll
      for (int s = 0; s < CS; ++s)
#pragma unroll
        for (int d = 0; d < 5; ++d) tn[s][d] = ix.pts[pjj[s][d]];
#pragma unroll
      for (int s = 0; s < CS; ++s) {
        if (!cold[s]) continue;
#pragma unroll
        for (int d = 0; d < 5; ++d) {
          const float dd = dist2(px[s], py[s], pz[s], tn[s][d].x, tn[s][d].y, tn[s][d].z);
          if (dd < best[s]) {
            best[s] = dd;
            b0s[s] = pjj[s][d] / (SB / 4);
          }
        }
      }
    }
  }
#pragma unroll
  for (int s = 0; s < CS; ++s) {
    const uint32_t b0 = b0s[s];
    if (!valid[s]) best[s] = -1.f;  // a lane without a point: a bound no box lower bound (>= 0) passes
    const int slot = s * 64 + lane;
    L.src[slot] = f32x4{px[s], py[s], pz[s], 0.f};
    L.key[slot] = ((unsigned long long)__float_as_uint(best[s]) << 32) | b0;
    L.tie[slot] = tie0[s];
  }
  if (lane == 0) {  // the dummy slot: a negative bound -- no box lower bound (>= 0) passes it
    L.src[S] = f32x4{0.f, 0.f, 0.f, 0.f};
    L.key[S] = (unsigned long long)__float_as_uint(-1.f) << 32;
  }
  // Through the sweep a source's bound lives in LDS (the key) and, for the lane-level tests, in a register as
  // bests[s] = bound x KS: the units of the chunk-box tests (below).  best[s] itself is read back after the sweep.
  constexpr float KS = SB2_INV_RANGE * SB2_INV_RANGE / NN_LB_SCALE;
  float bests[CS];
#pragma unroll
  for (int s = 0; s < CS; ++s) bests[s] = best[s] * KS;
  auto wave_max_best = [&]() {
    float m = -1.f;
#pragma unroll
    for (int s = 0; s < CS; ++s) m = fmaxf(m, bests[s]);  // (negative where there is no point)
    return wave_minmax<true>(m);
  };
  float wmax_s = wave_max_best();  // the wave's bound, in those units
  const unsigned long long t_pro = TRACE ? __builtin_amdgcn_s_memtime() : 0ull;
  unsigned long long t_chunks = 0;

  NN_MARK("sweep");
#if defined(GLOC_NN_RET) && GLOC_NN_RET == 1  // dev (timing only, wrong results): stop after the prologue
  if (wmax_s > -2.f) return;
#endif
  // ---- sweep: super-chunk boxes first (64 per ballot), then 64 chunk boxes per surviving batch ----
  auto box_box_lb = [&](const f32x4& blo, const f32x4& bhi) {  // (super-chunk boxes: corners; the result in units of (64 m)^2)
    const float ex = fmaxf(fmaxf(blo.x - whi[0], wlo[0] - bhi.x), 0.f);
    const float ey = fmaxf(fmaxf(blo.y - whi[1], wlo[1] - bhi.y), 0.f);
    const float ez = fmaxf(fmaxf(blo.z - whi[2], wlo[2] - bhi.z), 0.f);
    return __builtin_fmaf(ez, ez, __builtin_fmaf(ey, ey, ex * ex)) * (SB2_INV_RANGE * SB2_INV_RANGE);  // (a bound, not a distance: fused is fine)
  };
  // Chunk boxes are (centre, -half extent / 64) since round 4 (scan_store.hip: chunk_boxes_kernel): the distance of the
  // wave's box -- kept the same way -- or of a point to a chunk's box along an axis is clamp(|c - c'| / 64 + nh + nh'), one
  // v_fma_f32 with the abs and clamp modifiers; squared sums in units of (64 m)^2, saturated at one unit per axis (a
  // bound stays a bound), compared with the bounds scaled once: KS = 1 / 64^2 / NN_LB_SCALE.
  float wc[3], wnh[3];
#pragma unroll
  for (int a = 0; a < 3; ++a) {
    wc[a] = 0.5f * wlo[a] + 0.5f * whi[a];
    wnh[a] = -(fmaxf(wc[a] - wlo[a], whi[a] - wc[a]) * 1.0000005f + 1.0e-30f) * SB2_INV_RANGE;
  }
  auto axis_e = [](float d, float nh) {
    float e;
    asm("v_fma_f32 %0, |%1|, %2, %3 clamp" : "=v"(e) : "v"(d), "v"(SB2_INV_RANGE), "v"(nh));
    return e;
  };
  auto axis_es = [](float d, float nh_uniform) {  // the half extent in a scalar register (a chunk's box loaded with s_load)
    float e;
    asm("v_fma_f32 %0, |%1|, %2, %3 clamp" : "=v"(e) : "v"(d), "v"(SB2_INV_RANGE), "s"(nh_uniform));
    return e;
  };
  auto chunk_box_lb = [&](const f32x4& bc, const f32x4& bnh) {  // this lane's chunk box against the wave's box
    const float ex = axis_e(bc.x - wc[0], bnh.x + wnh[0]);
    const float ey = axis_e(bc.y - wc[1], bnh.y + wnh[1]);
    const float ez = axis_e(bc.z - wc[2], bnh.z + wnh[2]);
    return __builtin_fmaf(ez, ez, __builtin_fmaf(ey, ey, ex * ex));
  };
    // the lane-level test of one chunk box: which of the lane's sources can still use the chunk
    auto lane_test = [&](const f32x4& bc, const f32x4& bnh, bool (&need)[CS], unsigned long long (&nm)[CS]) {
      unsigned long long nm_any = 0ull;  // the ballots, taken where the comparisons are made
#pragma unroll
      for (int s = 0; s < CS; ++s) {  // (CS = 2: the three subtractions pair up in packed instructions)
        const float ex = axis_es(px[s] - bc.x, bnh.x), ey = axis_es(py[s] - bc.y, bnh.y), ez = axis_es(pz[s] - bc.z, bnh.z);
        need[s] = __builtin_fmaf(ez, ez, __builtin_fmaf(ey, ey, ex * ex)) <= bests[s];  // (no point: the bound -1, nothing passes)
        nm[s] = __builtin_amdgcn_ballot_w64(need[s]);
        nm_any |= nm[s];
      }
      return nm_any;
    };
    // One candidate chunk: the lane-level test of its box, and -- if a source can still use it -- staging, listing, the
    // sub-block tests, the evaluation rounds, the refresh of the bounds.  A lambda since round 6: the cold pass calls it
    // for its SEED chunks before the sweep as well (below); the warm kernel has the one call site it always had.
    bool gave_up = false;  // (cold pass: the wave has handed its group to the second launch -- NnHeavy)
    auto visit = [&](const uint32_t c, const bool last, const unsigned long long t_k0) {
      // the chunk's box straight into scalar registers (round 3: six v_readlane from the lane that tested it before)
      const f32x4 lo = ix.cbox_lo[c], hi = ix.cbox_hi[c];
      bool need[CS];
      unsigned long long nm[CS];
#ifdef GLOC_NN_DUP_CAND  // dev: the lane-level test of a candidate chunk twice
      {
        bool need2[CS];
        unsigned long long nm2[CS];
        f32x4 lo2 = lo;
        asm volatile("" : "+s"(lo2.x));
        unsigned long long sink_ = lane_test(lo2, hi, need2, nm2);
        asm volatile("" : : "s"(sink_));
      }
#endif
      const unsigned long long nm_any = lane_test(lo, hi, need, nm);
      if constexpr (TRACE) a_cand += now() - t_k0;
      if (nm_any == 0ull) return;
#if defined(GLOC_NN_RET) && GLOC_NN_RET == 2  // dev (timing only): candidates are tested, none is processed
      if (nm_any != 0x12345ull) return;
#endif
      n_processed++;  NN_MARK("candidate_tested");
      if constexpr (!WARM && !HEAVY) {
        // a cold wave this deep into its sweep is one of the heavy ones (p99 of a cold wave: ~30 chunks): hand the group to
        // the second launch and leave -- nothing has been written yet (outputs are the epilogue's)
        if (hv.cap != 0u && parts == 1 && n_processed == hv.thresh) {
          uint32_t e = 0;
          if (lane == 0) e = atomicAdd(hv.count, 1u);
          e = __builtin_amdgcn_readfirstlane(e);
          if (e < hv.cap) {
            if (lane == 0) {
              hv.list[2 * e] = job;
              hv.list[2 * e + 1] = gi;
            }
#pragma unroll
            for (int s = 0; s < CS; ++s)  // (the previous chunk's rounds have all been committed: visit() ends behind a barrier)
              hv.hkey[(size_t)e * S + s * 64 + lane] = L.key[s * 64 + lane] | (L.tie[s * 64 + lane] ? 0x80000000ull : 0ull);
            gave_up = true;
            return;
          }  // (the list is full: the wave goes on by itself)
        }
      }

      const unsigned long long t_c0 = now();
      // stage the chunk (wave-private LDS; padding never wins) and fetch its 8 sub-block boxes
#pragma unroll
      for (int u = 0; u < CH / 64; ++u) {
        const uint32_t tl = u * 64 + lane;
        const f32x4 v = ix.pts[c * CH + tl];  // the store pads the last chunk with far sentinels
        float* d = &L.stage[(tl / SB) * SB_STRIDE + ((tl % SB) >> 1) * 8 + (tl & 1)];
        d[0] = v.x; d[2] = v.y; d[4] = v.z;
      }
      // the boxes of the two sub-blocks this lane will test (2q, 2q + 1, q = lane % 4), straight into
      // registers; sub-blocks past the end of the scan have empty, inverted boxes: their bound is +inf
      GPTR(f32x4) bp = ix.sb2 + ((size_t)c * (NSB / 2) + (lane & 3)) * 3;
      const f32x4 bA = bp[0], bB = bp[1], bC = bp[2];
      // sources that passed the chunk-level test, compacted
      uint32_t k = 0;
#pragma unroll
      for (int s = 0; s < CS; ++s) {
        const unsigned long long m = nm[s];
        if (need[s])
          L.list[k + __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u))] =
              (uint16_t)(s * 64 + lane);
        k += (uint32_t)__popcll(m);
      }
      if (lane < 16) L.list[k + lane] = (uint16_t)S;  // (the test steps then need no bounds check)
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
      if constexpr (TRACE) a_stage += now() - t_c0;
  NN_MARK("staged_listed");
      // sub-block tests: a lane takes one listed source and TWO sub-blocks (packed fp32: both boxes per
      // instruction), against the source's CURRENT bound; the passing pairs become the work items
      const f32x2 bcx = {bA.x, bA.y}, bcy = {bA.z, bA.w}, bcz = {bB.x, bB.y};   // centres of the lane's two sub-blocks
      const f32x2 nhx = {bB.z, bB.w}, nhy = {bC.x, bC.y}, nhz = {bC.z, bC.w};   // -(half extent) / SB2_RANGE
      uint32_t total = 0, sbmask = 0;
      // evaluate the queued work items: all of them, or (between two test steps: FULL_ONLY) the full rounds only -- the
      // items of an incomplete round stay queued (moved to the front) and wait for company
      auto run_rounds = [&](auto full_tag) {
        constexpr bool FULL_ONLY = decltype(full_tag)::value;
        const unsigned long long t_r0 = now();
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        const uint32_t all_queued = total;
        if constexpr (FULL_ONLY) total &= ~(uint32_t)(GLOC_NN_GRAN - 1);
        n_items += total;
  NN_MARK("rounds_begin");
        // One round = up to 64 items.  A full round gives every lane one item: its sub-block's 16 staged targets, the
        // minimum of each quarter (the key records the winning quarter, so the index recovery re-reads 4 targets; a
        // minimum attained in two quarters -- two targets at the minimum distance -- takes the tie path).  A chunk queues
        // 55 items on average, so 37 % of the rounds are the tail of a chunk with at most 32 items (traced: 23 % have at
        // most 16): those give an item to TWO or FOUR lanes, each walking half / a quarter of the targets, and hand the
        // quarter minima round inside the quad (four DPP moves); lane 0 of the group commits.  Same values, same key.
        auto round_body = [&](uint32_t r, auto np_tag) {
          constexpr int NP = decltype(np_tag)::value;  // lanes per item: 1, 2 or 4
          const uint32_t it = r + (uint32_t)lane / NP;
          const int part = lane & (NP - 1);
          const bool act = it < total;
          const uint32_t item = L.queue[act ? it : r];
          const uint32_t slot = item >> 3, bi = item & 7;
          const f32x4 p = L.src[slot];
          const float* sb = &L.stage[bi * SB_STRIDE] + part * ((SB / 2 / NP) * 8);
          const f32x2 ppx = {p.x, p.x}, ppy = {p.y, p.y}, ppz = {p.z, p.z};
          constexpr int QL = 4 / NP;  // quarters this lane walks
          float ml[QL];
#pragma unroll
          for (int j = 0; j < QL; ++j) ml[j] = 3.402823466e+38f;
#pragma unroll
          for (int i = 0; i < SB / 2 / NP; ++i) {
            const f32x4 xy = *reinterpret_cast<const f32x4*>(sb + i * 8);
            const f32x2 zz = *reinterpret_cast<const f32x2*>(sb + i * 8 + 4);
            // two targets at once, FUSED (one multiply and two fma instead of three multiplies and two adds): within
            // 2e-7 of dist2()'s un-fused value, which is all the search needs -- see NEAR below; the distance that is
            // stored is computed by the index recovery, un-fused
            const f32x2 dx = ppx - f32x2{xy.x, xy.y}, dy = ppy - f32x2{xy.z, xy.w}, dz = ppz - zz;
            const f32x2 d2 = __builtin_elementwise_fma(dz, dz, __builtin_elementwise_fma(dy, dy, dx * dx));
            ml[i / (SB / 8)] = fminf(fminf(ml[i / (SB / 8)], d2.x), d2.y);
          }
          float mq[4];
          if constexpr (NP == 1) {
#pragma unroll
            for (int j = 0; j < 4; ++j) mq[j] = ml[j];
          } else if constexpr (NP == 2) {  // lanes (2k, 2k + 1) hold quarters (0, 1) and (2, 3)
            mq[0] = dpp_f32<0xA0>(ml[0]);  // quad_perm [0,0,2,2]
            mq[1] = dpp_f32<0xA0>(ml[1]);
            mq[2] = dpp_f32<0xF5>(ml[0]);  // quad_perm [1,1,3,3]
            mq[3] = dpp_f32<0xF5>(ml[1]);
          } else {  // the quad's lane j holds quarter j
            mq[0] = dpp_f32<0x00>(ml[0]);
            mq[1] = dpp_f32<0x55>(ml[0]);
            mq[2] = dpp_f32<0xAA>(ml[0]);
            mq[3] = dpp_f32<0xFF>(ml[0]);
          }
          const float m = fminf(fminf(fminf(mq[0], mq[1]), mq[2]), mq[3]);
          if (act && part == 0) {
            const uint32_t blk = c * NSB + bi;
            const bool e0 = mq[0] == m, e1 = mq[1] == m, e2 = mq[2] == m;
            const uint32_t quarter = e0 ? 0u : (e1 ? 1u : (e2 ? 2u : 3u));
            // NEAR: the evaluated distances are fused, the reference's are not; the two differ by < 2e-7 relative, so
            // the un-fused minimum lies among the targets whose fused distance is within NN_NEAR = 1e-6 of the fused
            // minimum.  Normally that is one quarter of one sub-block (the recovery re-reads its 4 targets and takes the
            // un-fused minimum); when a second quarter or another sub-block comes that close -- an exact tie included --
            // the source is flagged and the wave looks at every target that close, un-fused (the tie path).
            const float near_m = m * (1.0f + NN_NEAR);
            const bool twice = ((mq[0] <= near_m) + (mq[1] <= near_m) + (mq[2] <= near_m) + (mq[3] <= near_m)) > 1;
            const unsigned long long key = ((unsigned long long)__float_as_uint(m) << 32) | (blk * 4u + quarter);
            const unsigned long long old = atomicMin(&L.key[slot], key);
            const float od = __uint_as_float((uint32_t)(old >> 32));
            const bool close = od <= near_m && m <= od * (1.0f + NN_NEAR);  // (od = FLT_MAX: the product is +inf, the first test fails)
            const bool maybe = close || twice;
            if (__builtin_amdgcn_ballot_w64(maybe) != 0ull) {
              if ((close && ((uint32_t)old >> 2) != blk) || (twice && m <= od * (1.0f + NN_NEAR))) L.tie[slot] = 1;
            }
          }
        };
#ifdef GLOC_NN_DUP_ROUNDS  // dev: every evaluation round twice (idempotent: the same keys again) -- what the rounds cost
        for (uint32_t r = 0; r < total; r += 64) {
          const uint32_t left = total - r;
          if (left <= 16) round_body(r, std::integral_constant<int, 4>{});
          else if (left <= 32) round_body(r, std::integral_constant<int, 2>{});
          else round_body(r, std::integral_constant<int, 1>{});
        }
#endif
        for (uint32_t r = 0; r < total; r += 64) {
          n_rounds++;
          const uint32_t left = total - r;
          if constexpr (TRACE) round_hist += 1u << (8 * (left >= 64 ? 3u : (left - 1) / 16));  // occupancy, in quarters
          if (left <= 16) round_body(r, std::integral_constant<int, 4>{});
          else if (left <= 32) round_body(r, std::integral_constant<int, 2>{});
          else round_body(r, std::integral_constant<int, 1>{});
        }
  NN_MARK("rounds_end");
        if constexpr (FULL_ONLY) {
          const uint32_t rem = all_queued - total;  // (< 64 <= total: the two ranges do not overlap)
          uint16_t keep = 0;
          if ((uint32_t)lane < rem) keep = L.queue[total + lane];
          if ((uint32_t)lane < rem) L.queue[lane] = keep;
          total = rem;
          __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
          __builtin_amdgcn_wave_barrier();  // the keys the rounds lowered are what the next test step reads
          __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        } else {
          total = 0;
        }
        if constexpr (TRACE) a_rounds += now() - t_r0;
      };
  NN_MARK("teststeps");
      const unsigned long long t_s0 = now(), a_r_before = a_rounds;
      const uint32_t sb0 = (lane & 3) * 2;
      const uint16_t* list_lane = &L.list[lane >> 2];
      // one step of 16 listed sources (x 4 sub-block pairs) at a time: four steps unrolled together measured
      // 6 % slower -- a processed chunk lists 58 sources on average, many far fewer
      constexpr int TU = 1;
#ifdef GLOC_NN_DUP_TESTS  // dev: the chunk's test steps twice (the second pass rewrites the same queue entries)
      for (int rep_ = 0; rep_ < 2; ++rep_) {
      if (rep_ == 1) total = 0;
#endif
      for (uint32_t t0 = 0; t0 < k * (NSB / 2); t0 += 64 * TU) {
        // Between two steps every FULL round that is waiting is evaluated (round 4; until then only when the queue might
        // overflow): its lanes are as busy as they get, and the bounds it lowers save the chunk's later steps their items.
        static_assert(QCAP >= 63 + 128 * TU, "the tail of a round and one step's items fit the queue");
#ifdef GLOC_NN_EAGER  // test variant (lib/libgloc3d_smallq.so): everything queued is evaluated before every step
        if (total) run_rounds(std::false_type{});
#else
        if (total >= (uint32_t)GLOC_NN_GRAN) run_rounds(std::true_type{});
#endif
        uint32_t si[TU];
        bool act[TU];
#pragma unroll
        for (int u = 0; u < TU; ++u) {
          act[u] = true;  // entries past k are the dummy slot: its bound (-1) rejects every box
          si[u] = list_lane[(t0 + u * 64) >> 2];  // = L.list[(t0 + u * 64 + lane) >> 2]: one add per step
        }
        f32x4 p[TU];
        float bst[TU];
#pragma unroll
        for (int u = 0; u < TU; ++u) {
          p[u] = L.src[si[u]];
          bst[u] = __uint_as_float((uint32_t)(L.key[si[u]] >> 32));
        }
#pragma unroll
        for (int u = 0; u < TU; ++u) {
          // the distance to both boxes at once, per axis clamp(|p - c| / 64 - h / 64): one packed subtract, then one
          // v_fma_f32 with the abs and clamp modifiers per box (2.5 cycles; max(lo - p, p - hi, 0) was two packed
          // subtracts and a v_max3_f32 of 4.3).  Units of 64 m, saturated at one: a bound stays a bound.
          const f32x2 qx = {p[u].x, p[u].x}, qy = {p[u].y, p[u].y}, qz = {p[u].z, p[u].z};
          const f32x2 dx = qx - bcx, dy = qy - bcy, dz = qz - bcz;
          auto axis = [](float d, float nh) {
            float e;
            asm("v_fma_f32 %0, |%1|, %2, %3 clamp" : "=v"(e) : "v"(d), "v"(SB2_INV_RANGE), "v"(nh));
            return e;
          };
          const f32x2 ex = {axis(dx.x, nhx.x), axis(dx.y, nhx.y)};
          const f32x2 ey = {axis(dy.x, nhy.x), axis(dy.y, nhy.y)};
          const f32x2 ez = {axis(dz.x, nhz.x), axis(dz.y, nhz.y)};
          // sums of squares with scalar fma (2.6 cycles each; the packed forms are 4.4): a bound, not a distance
          const f32x2 lb = {__builtin_fmaf(ez.x, ez.x, __builtin_fmaf(ey.x, ey.x, ex.x * ex.x)),
                            __builtin_fmaf(ez.y, ez.y, __builtin_fmaf(ey.y, ey.y, ex.y * ex.y))};
          const float bs = bst[u] * (SB2_INV_RANGE * SB2_INV_RANGE / NN_LB_SCALE);  // the source's bound in those units
          const bool nd0 = act[u] && lb.x <= bs, nd1 = act[u] && lb.y <= bs;
          const unsigned long long m0 = __builtin_amdgcn_ballot_w64(nd0), m1 = __builtin_amdgcn_ballot_w64(nd1);
          const uint32_t c0n = (uint32_t)__popcll(m0);
          if (nd0)
            L.queue[total + __builtin_amdgcn_mbcnt_hi((uint32_t)(m0 >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m0, 0u))] =
                (uint16_t)((si[u] << 3) | sb0);
          if (nd1)
            L.queue[total + c0n + __builtin_amdgcn_mbcnt_hi((uint32_t)(m1 >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m1, 0u))] =
                (uint16_t)((si[u] << 3) | (sb0 + 1));
          total += c0n + (uint32_t)__popcll(m1);
          if constexpr (TRACE) sbmask |= (nd0 ? (1u << sb0) : 0u) | (nd1 ? (2u << sb0) : 0u);
        }
      }
  NN_MARK("teststeps_end");
#ifdef GLOC_NN_DUP_TESTS
      }
#endif
      if constexpr (TRACE) {
        uint32_t lm = 0;
        for (int b = 0; b < 8; ++b) lm |= __builtin_amdgcn_ballot_w64((sbmask >> b) & 1u) ? (1u << b) : 0u;
        n_live_sb += (uint32_t)__popc(lm);
        n_live_pairs += (uint32_t)__popc((lm | (lm >> 1)) & 0x55u);
        n_steps += (k + 15) / 16;
      }
      if constexpr (TRACE) a_tests += (now() - t_s0) - (a_rounds - a_r_before);
      run_rounds(std::false_type{});
      const unsigned long long t_f0 = now();
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();  // all reads of the stage done, all key updates visible
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  NN_MARK("refresh");
      // (the registers' copies of the bounds serve the tests of LATER candidates: after the wave's last one -- no bit left
      // in this batch, no batch prefetched, no further group of super-chunks -- nobody reads them)
      if (last) return;
      if constexpr (HEAVY) {
        // The parts of a group SHARE their bounds (hv.hkey: a board of one key per source, device-scope atomic min): a part
        // sees an eighth of the chunks, and on its own would search with the nearest point of ITS eighth as the bound --
        // on dense ground a ball as full as the whole wave's (measured: the second launch took as long as the waves it
        // replaced).  Every fourth processed chunk a part puts its keys on the board and adopts what is smaller there --
        // distance AND location: a location is a place in the target, so any part can recover it.  A minimum that is
        // contested ACROSS parts (within NN_NEAR of the adopted one, in another sub-block) takes the tie path like one
        // contested inside a wave; so does an adopted key whose owner had flagged it.
        if ((n_processed & 3u) == 0u) {
#pragma unroll
          for (int s = 0; s < CS; ++s) {
            const int slot = s * 64 + lane;
            const unsigned long long k = L.key[slot];
            const unsigned long long old = __hip_atomic_fetch_min(hv.hkey + (size_t)hid * S + slot, k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const unsigned long long oldk = old & ~0x80000000ull;
            if (oldk < k) {
              const float od = __uint_as_float((uint32_t)(oldk >> 32)), md = __uint_as_float((uint32_t)(k >> 32));
              if ((old & 0x80000000ull) != 0ull || (md <= od * (1.0f + NN_NEAR) && ((uint32_t)oldk >> 2) != ((uint32_t)k >> 2))) L.tie[slot] = 1;
              L.key[slot] = oldk;
            }
          }
        }
      }
      bool changed = false;
#pragma unroll
      for (int s = 0; s < CS; ++s) {
        const float nb = __uint_as_float((uint32_t)(L.key[s * 64 + lane] >> 32));
        const float nbs = nb * KS;
        changed |= nbs < bests[s];
        bests[s] = nbs;
      }
      // A warm pass starts from bounds that are nearly final (the previous pass's neighbour): the WAVE's bound hardly moves,
      // and re-reducing it after every chunk cost more than the candidates it spared (round 5, same box: 38.16 -> 37.87 ms of
      // 1-NN per step; dropping the lanes' refresh as well: 38.30).  The stale value is still an upper bound: same result.
      // (A warm wave with a source that STARTED without a bound -- its previous correspondence invalid, a NaN point -- has a
      // wave bound near FLT_MAX x KS and would cull no chunk at the wave level for the whole sweep: it refreshes as a cold wave does.)
      if (!WARM || wmax_s > 1.0e30f) {
        if (__builtin_amdgcn_ballot_w64(changed) != 0ull) wmax_s = wave_max_best();
      }
      if constexpr (TRACE) {
        const unsigned long long t_e = now();
        a_refresh += t_e - t_f0;
        t_chunks += t_e - t_c0;
      }
    };
  // ---- cold pass only: the SEED chunks first (round 6) -------------------------------------------------------
  // A cold source's bound is its distance to the best of five curve neighbours: within 1.2 x the true distance for half of
  // the sources, but 8 x for a tenth and 50 x for a hundredth (CPU model on bench.py's distinct ray-casts) -- and the sweep
  // visits the chunks in index order, so such a source drags every chunk its loose ball touches through the tests until its
  // own neighbourhood comes up.  On the sensor's near field, where a ball of 3 m holds 10 000 ground points, that made
  // waves of 3.6 M cycles (1.7 ms: as long as the whole launch of 160 jobs).  The chunks that HOLD those curve neighbours
  // -- kd cells of 128 points around them -- are therefore visited before the sweep, up to NSEED of them (the first
  // unserved source's, then the next's ...): the bounds the sweep then starts from are those of a warm pass.  The sweep
  // skips them (they are done: bounds only tighten).  Same result: every chunk that can hold a nearer-or-equal point is
  // still visited, once, with bounds that are upper bounds.
  constexpr int NSEED = 4;
  uint32_t seed_c[NSEED];
#pragma unroll
  for (int k = 0; k < NSEED; ++k) seed_c[k] = 0xFFFFFFFFu;
#ifndef GLOC_NN_R5_COLD  // (-DGLOC_NN_R5_COLD: round 5's cold pass -- no seeds, no second launch -- for same-box A/Bs: tools/dev_ab.sh)
  if constexpr (!WARM) {
    if (parts == 1 && ix.nchunks) {
      bool pend[CS];
      uint32_t sc[CS];
#pragma unroll
      for (int s = 0; s < CS; ++s) {
        sc[s] = b0s[s] / (uint32_t)(CH / (SB / 4));  // the chunk of the source's bound (b0 = sorted position / (SB / 4))
        pend[s] = valid[s] && best[s] < 3.0e38f;      // (a source without a bound has no seed)
      }
      for (int k = 0; k < NSEED; ++k) {
        uint32_t c = 0xFFFFFFFFu;
#pragma unroll
        for (int s = CS - 1; s >= 0; --s) {
          const unsigned long long m = __builtin_amdgcn_ballot_w64(pend[s]);
          if (m) c = (uint32_t)__builtin_amdgcn_readlane((int)sc[s], __ffsll((long long)m) - 1);
        }
        if (c == 0xFFFFFFFFu) break;
        c = c < ix.nchunks ? c : ix.nchunks - 1;
#pragma unroll
        for (int j = 0; j < NSEED; ++j)
          if (j == k) seed_c[j] = c;
#pragma unroll
        for (int s = 0; s < CS; ++s) pend[s] = pend[s] && sc[s] != c;
        if constexpr (TRACE) n_cand++;
        visit(c, false, now());
        if (gave_up) return;
      }
    }
  }
#endif
#if defined(GLOC_NN_RET) && GLOC_NN_RET == 6  // dev (timing only): no sweep at all -- prologue + epilogue
  for (uint32_t s0 = 0; s0 < (wmax_s > -2.f ? 0u : ix.nsup); s0 += 64) {
#else
  for (uint32_t s0 = 0; s0 < ix.nsup; s0 += 64) {
#endif
    float lbs = __builtin_inff();  // (not FLT_MAX: a wave whose bound is still FLT_MAX -- a non-finite source point -- must not pass lanes past the end)
    if (s0 + lane < ix.nsup) {
      const f32x4 ulo = ix.sup_lo[s0 + lane], uhi = ix.sup_hi[s0 + lane];
      lbs = box_box_lb(ulo, uhi);
    }
    unsigned long long smask = __builtin_amdgcn_ballot_w64(lbs <= wmax_s);
    // the chunk boxes of the NEXT surviving batch are in flight while the current one is worked on
    f32x4 nlo = {0.f, 0.f, 0.f, 0.f}, nhi = {0.f, 0.f, 0.f, 0.f};
    int cur = -1;
    if (smask) {
      cur = __ffsll((long long)smask) - 1;
      smask &= smask - 1;
      const uint32_t cl = (s0 + cur) * 64 + lane;
      if (cl < ix.nchunks) { nlo = ix.box_lo[cl]; nhi = ix.box_hi[cl]; }
    }
    while (cur >= 0) {
    const unsigned long long t_b0 = now();
    const uint32_t c0 = (s0 + cur) * 64;
    const bool live = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(lbs), cur)) <= wmax_s;
    const uint32_t cl = c0 + lane;
    const f32x4 blo = nlo, bhi = nhi;
    cur = -1;
    if (smask) {
      cur = __ffsll((long long)smask) - 1;
      smask &= smask - 1;
      const uint32_t cn = (s0 + cur) * 64 + lane;
      if (cn < ix.nchunks) { nlo = ix.box_lo[cn]; nhi = ix.box_hi[cn]; }
    }
    if (!live) continue;  // the wave's bound tightened since the super-chunk ballot
    float lbw = __builtin_inff();
    if (cl < ix.nchunks) lbw = chunk_box_lb(blo, bhi);
    unsigned long long mask = __builtin_amdgcn_ballot_w64(lbw <= wmax_s);
    if constexpr (SPLIT) {
      mask &= pmask;
      w_cand += (uint32_t)__popcll(mask);
    }
    if constexpr (TRACE) a_batch += now() - t_b0;
  NN_MARK("batch_tested");
    // A wave over a sparse stretch of the curve (the far field: 128 points in a box of 60 m x 130 m, each 0.2 m from
    // its neighbour) lists hundreds of candidates here and fails nearly all of them at the lane level: it is the wave
    // a launch of few jobs waits for, and a scalar load per candidate (a miss in the scalar cache: ~400 cycles, nothing
    // to overlap it with) is most of its time.  A batch with many survivors is therefore thinned first, with each box
    // taken from the lane that holds it (v_readlane: no memory): what remains goes through the loop below.
    if (__popcll(mask) > GLOC_NN_THIN_MIN) {
      const unsigned long long t_t0 = now();
      unsigned long long keep = 0ull;
      while (mask) {
        const int b = __ffsll((long long)mask) - 1;
        mask &= mask - 1;
        auto rl = [&](float x) { return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(x), b)); };
        const f32x4 lo = {rl(blo.x), rl(blo.y), rl(blo.z), 0.f}, hi = {rl(bhi.x), rl(bhi.y), rl(bhi.z), 0.f};
        bool need[CS];
        unsigned long long nm[CS];
        if constexpr (TRACE) n_cand++;
        if (lane_test(lo, hi, need, nm) != 0ull) keep |= 1ull << b;
      }
      mask = keep;
      if constexpr (TRACE) a_thin += now() - t_t0;
    }
    while (mask) {
      const int b = __ffsll((long long)mask) - 1;
      mask &= mask - 1;
      const unsigned long long t_k0 = now();
      if (__int_as_float(__builtin_amdgcn_readlane(__float_as_int(lbw), b)) > wmax_s) continue;
      const uint32_t c = __builtin_amdgcn_readfirstlane(c0 + b);
      if constexpr (!WARM) {  // (a seed chunk: visited before the sweep)
        bool seeded = false;
#pragma unroll
        for (int k = 0; k < NSEED; ++k) seeded |= c == seed_c[k];
        if (seeded) continue;
      }
      if constexpr (TRACE) n_cand++;
      visit(c, mask == 0ull && cur < 0 && s0 + 64 >= ix.nsup, t_k0);
      if constexpr (!WARM && !HEAVY) {
        if (gave_up) return;
      }
    }
    }  // batches of this super-chunk group
  }
  NN_MARK("sweep_end");
  // (one counter for the whole grid serialised the launch: 483 k atomics on one address took 12.6 ns
  // each, which WAS the launch time of the profiled runs of rounds 1 and 2 until this was found)
  if (stat_pairs && lane == 0) atomicAdd(stat_pairs + (lin_block % NN_STAT_SLOTS), n_items * (unsigned long long)SB);
  if (SPLIT && lane == 0 && (!HEAVY || sp.work != nullptr)) {
    const uint32_t wk = (part == 0 ? NN_W_FIXED : 0u) + NN_W_CAND * w_cand + NN_W_CHUNK * n_processed + NN_W_ITEM * (uint32_t)n_items;
    uint32_t* wp = sp.work + (size_t)job * n_part + gi;
    if (parts == 1) st_u32<CHAIN>(wp, wk); else atomicAdd(wp, wk);  // (the planner left a zero)
  }
  const unsigned long long t_sweep = TRACE ? __builtin_amdgcn_s_memtime() : 0ull;

#if defined(GLOC_NN_RET) && GLOC_NN_RET == 3  // dev (timing only): no index recovery, no outputs
  if (wmax_s > -2.f) return;
#endif
  // (the bounds as the sweep left them: the fused minima -- the keys' high words)
#pragma unroll
  for (int s = 0; s < CS; ++s) best[s] = __uint_as_float((uint32_t)(L.key[s * 64 + lane] >> 32));
  NN_MARK("recovery");
  // ---- index recovery: smallest ORIGINAL index among the targets at the minimum distance ----
  // bpos = that target's sorted position (what is stored), (qx, qy, qz) its coordinates
  uint32_t bpos[CS];
  {
    // (both sources' quarters are fetched before either is measured: one dependent round trip per wave, not CS)
    uint32_t bch[CS];
    bool rec[CS];
    f32x4 t[CS][SB / 4];
#pragma unroll
    for (int s = 0; s < CS; ++s) {
      const int slot = s * 64 + lane;
      bch[s] = (uint32_t)L.key[slot];  // (sub-block << 2) | quarter: 4 targets
      rec[s] = valid[s] && ix.n && L.tie[slot] == 0;
      // (the store pads the scan to whole chunks, so the loads are unconditional; a lane with nothing to recover reads quarter 0)
      GPTR(f32x4) tp = ix.pts + (size_t)(rec[s] ? bch[s] : 0u) * (SB / 4);
#pragma unroll
      for (int u = 0; u < SB / 4; ++u) t[s][u] = tp[u];
    }
#pragma unroll
    for (int s = 0; s < CS; ++s) {
      bpos[s] = 0xFFFFFFFFu;
      if (!rec[s]) continue;
      uint32_t bj = 0xFFFFFFFFu;
      float bd = __builtin_inff();
      const f32x2 pxy = {px[s], py[s]};
#pragma unroll
      for (int u = 0; u < SB / 4; ++u) {
        // dist2(): x and y share one packed instruction (a loaded point's x, y are a register pair);
        // same roundings as the scalar form
        const f32x2 dxy = pxy - f32x2{t[s][u].x, t[s][u].y};
        const f32x2 sxy = dxy * dxy;
        const float dz = pz[s] - t[s][u].z;
        const float d2 = (sxy.x + sxy.y) + dz * dz;
        const uint32_t o = __float_as_uint(t[s][u].w);  // padding carries 0xFFFFFFFF: never smaller
        // the un-fused minimum of the quarter (the search compared fused distances), smallest original index first.
        // (Selects, not branches: from `a || (b && c)` the compiler made four exec-mask branches a target, 32 a wave.)
        const bool take = (d2 < bd) | ((d2 == bd) & (o < bj));
        bd = take ? d2 : bd;
        bj = take ? o : bj;
        bpos[s] = take ? bch[s] * (SB / 4) + u : bpos[s];
      }
      if (bpos[s] != 0xFFFFFFFFu) best[s] = bd;
    }
  }
  const unsigned long long t_rec = now();
  NN_MARK("tie");
  // rare: a source whose minimum is contested -- two targets at the same distance, or fused distances within NN_NEAR
  // of each other (tie flag).  The wave looks for it together: lanes <-> chunk boxes, then lanes <-> the targets of every
  // chunk that can hold a point that close, with dist2() itself; the smallest un-fused distance wins, the smallest
  // original index among equals.  (Until round 3 the lane searched alone, chunk by chunk: ~400 k cycles, the longest
  // wave of a launch whenever it happened.)
#pragma unroll
  for (int s = 0; s < CS; ++s) {
    unsigned long long tm = __builtin_amdgcn_ballot_w64(valid[s] && ix.n && L.tie[s * 64 + lane] != 0);
    if constexpr (TRACE) n_ties += (uint32_t)__popcll(tm);
    while (tm) {
      const int tl = __ffsll((long long)tm) - 1;
      tm &= tm - 1;
      auto rl = [&](float x) { return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(x), tl)); };
      const float qx = rl(px[s]), qy = rl(py[s]), qz = rl(pz[s]);
      const float qb = rl(best[s]) * (1.0f + 2.0f * NN_NEAR);  // (the fused minimum: everything this close is looked at)
      unsigned long long bk = ~0ull;  // (bits of the un-fused distance << 32) | original index
      uint32_t bp = 0xFFFFFFFFu;      // its sorted position
      for (uint32_t c0 = 0; c0 < ix.nchunks; c0 += 64) {
        const uint32_t cl = c0 + lane;
        bool hit = false;
        if (cl < ix.nchunks) {
          const f32x4 bc = ix.box_lo[cl], bnh = ix.box_hi[cl];  // (centre, -half extent / 64)
          const float ex = axis_e(qx - bc.x, bnh.x), ey = axis_e(qy - bc.y, bnh.y), ez = axis_e(qz - bc.z, bnh.z);
          hit = !(__builtin_fmaf(ez, ez, __builtin_fmaf(ey, ey, ex * ex)) > qb * KS);
        }
        unsigned long long cm = __builtin_amdgcn_ballot_w64(hit);
        while (cm) {
          const uint32_t cc = c0 + (uint32_t)(__ffsll((long long)cm) - 1);
          cm &= cm - 1;
#pragma unroll
          for (int u = 0; u < CH / 64; ++u) {
            const uint32_t j = cc * CH + u * 64 + lane;
            if (j < ix.n) {
              const f32x4 t = ix.pts[j];
              const float d2 = dist2(qx, qy, qz, t.x, t.y, t.z);
              const unsigned long long k = ((unsigned long long)__float_as_uint(d2) << 32) | __float_as_uint(t.w);
              if (d2 == d2 && k < bk) {  // (non-negative floats order like their bits; a NaN target never wins)
                bk = k;
                bp = j;
              }
            }
          }
        }
      }
      unsigned long long wk = bk;
      for (int o = 32; o > 0; o >>= 1) {
        const unsigned long long ok = __shfl_xor(wk, o);
        wk = ok < wk ? ok : wk;
      }
      if (wk != ~0ull) {
        const int owner = __ffsll((long long)__builtin_amdgcn_ballot_w64(bk == wk)) - 1;  // (original indices are unique)
        const uint32_t wp = (uint32_t)__builtin_amdgcn_readlane((int)bp, owner);
        if (lane == tl) {
          bpos[s] = wp;
          best[s] = __uint_as_float((uint32_t)(wk >> 32));
        }
      }
    }
  }

  // ---- a group searched by several waves: fold the parts, the last one to arrive goes on ---------------------
  if (SPLIT && parts > 1) {
    unsigned long long* sk = HEAVY ? hv.skey + (size_t)hid * S : sp.skey + ((size_t)job * sp.hx + hid) * S;
    uint32_t* tick = HEAVY ? hv.ticket + hid : sp.ticket + (size_t)job * sp.hx + hid;
#pragma unroll
    for (int s = 0; s < CS; ++s) {
      if (bpos[s] == 0xFFFFFFFFu) continue;
      const uint32_t o = __float_as_uint(ix.pts[bpos[s]].w);  // (just read: an L1 hit)
      __hip_atomic_fetch_min(sk + s * 64 + lane, ((unsigned long long)__float_as_uint(best[s]) << 32) | o, __ATOMIC_RELAXED,
                             __HIP_MEMORY_SCOPE_AGENT);
    }
    // The keys before the ticket.  Only atomics carry data between the parts (device-scope read-modify-writes and loads,
    // performed where all XCDs see them), so it is enough that this wave's have been acknowledged before its ticket is
    // drawn: no cache write-back or invalidate (__threadfence() here -- a write-back of the XCD's whole L2 per part --
    // made a launch with 2 500 parts take twice as long).
    __builtin_amdgcn_s_waitcnt(0);  // vmcnt = expcnt = lgkmcnt = 0
    asm volatile("" ::: "memory");
    uint32_t arrived = 0;
    if (lane == 0) arrived = __hip_atomic_fetch_add(tick, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    arrived = __builtin_amdgcn_readfirstlane(arrived);
    if (arrived != parts - 1) return;  // (all lanes: no barrier below)
#pragma unroll
    for (int s = 0; s < CS; ++s) {
      const unsigned long long k = __hip_atomic_load(sk + s * 64 + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      st_u64<CHAIN>(sk + s * 64 + lane, ~0ull);  // as the next pass expects it
      if (k != ~0ull) {
        best[s] = __uint_as_float((uint32_t)(k >> 32));
        bpos[s] = ix.inv[(uint32_t)k];
      }
    }
    if (lane == 0) st_u32<CHAIN>(tick, 0u);
  }
  const unsigned long long t_tie = now();
  NN_MARK("outputs");
  // ---- outputs: corr / d2 (sorted slots), pairs, the wave's moments ------------------------------
  // Moments about the WAVE'S OWN centre, in fp32 (round 3; fp64 raw moments until then: 11 % of the launch).  The ICP
  // step needs n, sum p, sum q, sum p q^T to ~1e-9 of their size, which for raw coordinates (100 m, 124 k points) takes
  // fp64.  About the centre c of the wave's box the coordinates are at most half the wave's extent W (and the matched
  // target's at most that plus the correspondence distance): a 24-bit sum of 128 products errs by at most
  // 128 x 2^-24 x (W / 2 + d)^2 -- 2e-5 m^2 for an ordinary wave (W ~ 2 m), 0.04 m^2 for one of the ~10 far-field waves of
  // a scan (W up to 130 m, d up to tens of metres when no gate is set) -- against entries of sum p q^T of order
  // 124 000 x (30 m)^2 = 1e8 m^2: 4e-10 relative in the worst case, 1e-13 ordinarily.  solve_kernel puts the raw moments back
  // together in fp64 (sum p q^T = sum p' q'^T + c sum q'^T + sum p' c^T + n c c^T: exact products of fp32 values), in the
  // fixed order of the partials.  The pose that results is within 1e-5 of the fp64-moment path's
  // (tests/test_reg_gpu.py::test_every_pass_bit_identical_to_the_brute_force_kernel) and 1e-4 of the oracle's.
  float mv[17], pp = 0.f;  // (pp: sum |p'|^2, for the RMS size of the ICP update -- solve_kernel)
#pragma unroll
  for (int k = 0; k < 17; ++k) mv[k] = 0.f;
  const float cen[3] = {0.5f * (wlo[0] + whi[0]), 0.5f * (wlo[1] + whi[1]), 0.5f * (wlo[2] + whi[2])};
#pragma unroll
  for (int s = 0; s < CS; ++s) {
    if (!valid[s]) continue;
    const size_t o = (size_t)job * ld + wave_base + s * 64 + lane;
    st_u32<CHAIN>(corr + o, bpos[s]);  // (chained: written through -- the group's next pass may run on another XCD)
    st_f32<CHAIN>(d2out + o, best[s]);
    f32x4 q = {NN_FAR, NN_FAR, NN_FAR, 0.f};  // no correspondence (empty target): never an inlier
    if (bpos[s] != 0xFFFFFFFFu) {
      q = ix.pts[bpos[s]];  // just loaded above: an L1 hit
      q.w = 0.f;
      const float d2 = best[s];
      if constexpr (!PAIRS) mv[16] += d2;
      if (!PAIRS && (!(gate2 > 0.f) || d2 < gate2)) {  // (the pass that writes the pairs is refitted from them: no moments)
        const float P[3] = {px[s] - cen[0], py[s] - cen[1], pz[s] - cen[2]};
        const float Q[3] = {q.x - cen[0], q.y - cen[1], q.z - cen[2]};
        mv[0] += 1.f;
        pp = __builtin_fmaf(P[2], P[2], __builtin_fmaf(P[1], P[1], __builtin_fmaf(P[0], P[0], pp)));
#pragma unroll
        for (int a = 0; a < 3; ++a) {
          mv[1 + a] += P[a];
          mv[4 + a] += Q[a];
#pragma unroll
          for (int b = 0; b < 3; ++b) mv[7 + 3 * a + b] = __builtin_fmaf(P[a], Q[b], mv[7 + 3 * a + b]);
        }
      }
    }
    if (PAIRS) {
      pairs[o * 2 + 0] = f32x4{px[s], py[s], pz[s], 0.f};
      pairs[o * 2 + 1] = q;
    }
  }
  const unsigned long long t_out = now();
  NN_MARK("reduce");
  if (!PAIRS && partials) {
    // Sum over the 64 lanes in the order of the xor butterfly (o = 32, 16, ..., 1), but as a
    // reduce-scatter: at every step a lane keeps half of its values and hands the other half to its
    // partner, so 16 values cost 8 + 4 + 2 + 1 + 1 + 1 exchanges instead of 16 x 6.  Lane l ends with moment
    // (l >> 2) & 15 (all four lanes of a quad).  The partial: 17 floats + the centre (WavePartial).
    static_assert(ACC_NV == 18, "16 scattered moments, the d2 sum, sum |p'|^2");
    static_assert(sizeof(double) * ACC_NV >= sizeof(float) * WAVE_PARTIAL_FLOATS, "a wave partial fits a partial's slot");
    float* out = reinterpret_cast<float*>(partials + ((size_t)job * n_part + gi) * ACC_NV);
#pragma unroll
    for (int k = 0; k < 8; ++k) mv[k] = exchange_add<32>(mv[k], mv[k + 8]);
#pragma unroll
    for (int k = 0; k < 4; ++k) mv[k] = exchange_add<16>(mv[k], mv[k + 4]);
    {
      const bool up = (lane & 8) != 0;
#pragma unroll
      for (int k = 0; k < 2; ++k) {
        const float keep = up ? mv[k + 2] : mv[k], send = up ? mv[k] : mv[k + 2];
        mv[k] = keep + xor_lane<8>(send);
      }
    }
    {
      const bool up = (lane & 4) != 0;
      const float keep = up ? mv[1] : mv[0], send = up ? mv[0] : mv[1];
      mv[0] = keep + xor_lane<4>(send);
    }
    float x = mv[0], y = mv[16];
    x += xor_lane<2>(x);
    x += xor_lane<1>(x);
    y = exchange_add<32>(y, pp);  // the two whole-wave sums share the butterfly: lanes 0-31 end with sum d2, 32-63 with sum |p'|^2
    y = exchange_add<16>(y, y);
    y += xor_lane<8>(y);
    y += xor_lane<4>(y);
    y += xor_lane<2>(y);
    y += xor_lane<1>(y);
    if ((lane & 3) == 0) st_f32<CHAIN>(out + ((lane >> 2) & 15), x);
    if (lane == 0) {
      st_f32<CHAIN>(out + 16, y);
      st_f32<CHAIN>(out + 17, cen[0]);
      st_f32<CHAIN>(out + 18, cen[1]);
      st_f32<CHAIN>(out + 19, cen[2]);
    }
    if (lane == 32) st_f32<CHAIN>(out + 20, y);
  }
  NN_MARK("end");
  if (TRACE && trace && lane == 0) {
    const size_t wid = (size_t)lin_block * NN_WPB + w;
    uint32_t* tw = trace + NN_TRACE_WORDS * wid;
    const unsigned long long t_end = now();
    tw[0] = (uint32_t)(t_end - t_start);
    tw[1] = round_hist;  // rounds with <= 16 / 32 / 48 / 64 items, a byte each
    tw[2] = n_processed;
    tw[3] = (n_rounds & 0xFFFFu) | (n_cand << 16);  // rounds | candidate chunks (passed the wave-level test)
    tw[4] = (uint32_t)n_items;
    tw[5] = ((uint32_t)(t_pro - t_start) & 0xFFFFFFu) | (n_ties << 24);  // prologue cycles | contested sources
    tw[6] = job;
    tw[7] = (n_live_sb << 20) | (n_live_pairs << 10) | n_steps;  // per wave: live sub-blocks, live pairs, test steps
    // cycles per region
    tw[8] = (uint32_t)(t_box - t_start);   // load + transform + wave box
    tw[9] = (uint32_t)(t_pro - t_box);     // upper bounds -> LDS state
    tw[10] = (uint32_t)a_batch;            // batches of 64 chunk boxes (wave level), incl. the wait for their loads
    tw[11] = (uint32_t)a_thin;             // thinning of many-survivor batches
    tw[12] = (uint32_t)a_cand;             // lane-level tests of candidate chunks
    tw[13] = (uint32_t)a_stage;            // staging + listing
    tw[14] = (uint32_t)a_tests;            // sub-block test steps
    tw[15] = (uint32_t)a_rounds;           // evaluation rounds
    tw[16] = (uint32_t)a_refresh;          // bound refresh
    tw[17] = (uint32_t)(t_sweep - t_pro);  // the whole sweep
    tw[18] = (uint32_t)(t_rec - t_sweep);  // index recovery
    tw[19] = (uint32_t)(t_tie - t_rec);    // contested minima
    tw[20] = (uint32_t)(t_out - t_tie);    // outputs + moments
    tw[21] = (uint32_t)(t_end - t_out);    // reduction of the moments
    tw[22] = gi | (part << 20) | (parts << 24);
    tw[23] = (uint32_t)rt_start;  // the constant 100 MHz clock all XCDs share (the shader clock above is not synchronised)
    tw[24] = (uint32_t)__builtin_amdgcn_s_memrealtime();
    tw[25] = __builtin_amdgcn_s_getreg((31 << 11) | 20);  // XCC_ID
    tw[26] = __builtin_amdgcn_s_getreg((31 << 11) | 4);   // HW_ID: wave, simd, cu, sh, se
    (void)t_chunks;
  }
}

#define NN_COMPACT_PARAMS                                                                                              \
  const Job *__restrict__ jobs, uint32_t n_jobs, uint32_t job_group, uint32_t n_wg, uint32_t subs,                     \
      const CandState *__restrict__ states, const uint32_t *prev_corr, uint32_t *corr, float *__restrict__ d2out,      \
      f32x4 *__restrict__ pairs, double *__restrict__ partials, uint32_t n_part, size_t ld, float gate2, NnSplit sp,   \
      NnHeavy hv, unsigned long long *__restrict__ stat_pairs, uint32_t *__restrict__ trace
#define NN_COMPACT_ARGS jobs, n_jobs, job_group, n_wg, subs, states, prev_corr, corr, d2out, pairs, partials, n_part, ld, gate2, sp, hv, stat_pairs, trace

template <int CS, bool PAIRS, bool TRACE = false, bool SPLIT = false, bool WARM = false>
__global__ __launch_bounds__(64 * NN_WPB) void nn_compact_kernel(NN_COMPACT_PARAMS) {
  nn_compact_body<CS, PAIRS, TRACE, SPLIT, WARM>(NnPos{blockIdx.z, blockIdx.x, blockIdx.y}, NN_COMPACT_ARGS);
}

// The second launch of a cold pass: the groups its waves gave up, NN_HEAVY_PARTS waves each (NnHeavy).
template <int CS, bool PAIRS>
__global__ __launch_bounds__(64) void nn_compact_heavy_kernel(NN_COMPACT_PARAMS) {
  nn_compact_body<CS, PAIRS, false, true, false, true>(NnPos{0u, 0u, 0u}, NN_COMPACT_ARGS);
}

// The warm moments pass with the split plan in it -- what one query alone runs 20 times -- held to the register budget
// of six waves per SIMD (80 VGPRs; the compiler takes 82 left alone, one scalar spill more with the limit): the plain
// kernel fits by itself.
template <int CS>
__global__ __launch_bounds__(64 * NN_WPB) __attribute__((amdgpu_waves_per_eu(6, 6))) void nn_compact_split_warm_kernel(NN_COMPACT_PARAMS) {
  nn_compact_body<CS, false, false, true, true>(NnPos{blockIdx.z, blockIdx.x, blockIdx.y}, NN_COMPACT_ARGS);
}

// All warm moments passes of a small batch in ONE launch (NnChain): a one-dimensional grid of n_pass x pass_size
// single-wave work-groups; a pass is `groups` runs of [job_group * n_wg searches | the group's roles | padding to a
// multiple of 8, so that a slot keeps its XCD from pass to pass].  Held to six waves per SIMD like the warm kernel it chains.
template <int CS, bool SPLIT = true>
__global__ __launch_bounds__(64 * NN_WPB) __attribute__((amdgpu_waves_per_eu(6, 6))) void nn_chain_kernel(NN_COMPACT_PARAMS, NnChain ch) {
  static_assert(NN_WPB == 1, "a work-group is a wave: the roles and the arrival counts are per wave");
  const uint32_t b = blockIdx.x;
  const uint32_t pass = b / ch.pass_size, r1 = b - pass * ch.pass_size;
  const uint32_t grp = r1 / ch.grp_size, r2 = r1 - grp * ch.grp_size;
  const uint32_t n_search = job_group * n_wg;
  if (r2 >= n_search) {  // a role of one of the group's jobs
    const uint32_t si = r2 - n_search, jl = si / NN_CHAIN_ROLES, role = si - jl * NN_CHAIN_ROLES;
    const uint32_t job = grp * ch.jobs_per_grp + jl;
    if (jl >= ch.jobs_per_grp || job >= n_jobs) return;  // padding
    if (role == 0) chain_stamp(ch, pass, job, n_jobs, 3, 0);  // reducer 0 starts to wait
    // The search waves only add to `done` and leave (they do not wait for the sum to come back: 2 us of a slot, 8 % of
    // a wave); ONE wave per job, its planner, polls the counter and raises `go` for the reducers, who poll that -- a line
    // the 1 225 additions do not go through.
    const size_t cell = ((size_t)pass * n_jobs + job) * NN_CHAIN_PAD;
    if (role == NN_CHAIN_RED) {
      if (!chain_wait<false>(ch.done + cell, ch.expected, ch.err)) return;
      if ((threadIdx.x & 63) == 0) st_u32<true>(ch.go + cell, 1u);
      chain_stamp(ch, pass, job, n_jobs, 2, 0);  // the job's pass is seen complete
    } else if (!chain_wait<false>(ch.go + cell, 1u, ch.err)) {
      return;
    }
    if (role == 0) chain_stamp(ch, pass, job, n_jobs, 4, 0);  // ... and sees the pass done
    nn_compact_body<CS, false, false, SPLIT, true, false, true>(NnPos{0u, 0u, 0u, 1u + role, pass, job}, jobs, n_jobs, job_group, n_wg, subs, states,
                                                                corr, corr, d2out, pairs, partials, n_part, ld, gate2, sp, hv, stat_pairs, trace, ch);
    return;
  }
  const uint32_t wgv = r2 / job_group, slot = r2 - wgv * job_group;
  // (the body's own decoding of the slot -- only the job is needed here)
  uint32_t vin = slot;
  if ((job_group & 7u) == 0u) vin = (slot & ~7u) | ((slot - (slot >> 3) - grp) & 7u);
  const uint32_t vjob = grp * job_group + vin;
  if (vjob >= n_jobs * subs) return;
  const uint32_t job = vjob / subs;
  const bool first_wave = wgv == 0u && vjob % subs == 0u;  // (dev stamps: the first wave of the job's first share)
  if (first_wave) chain_stamp(ch, pass, job, n_jobs, 0, 0);  // a first search wave of the job's pass here
  const size_t cell = ((size_t)pass * n_jobs + job) * NN_CHAIN_PAD;
  NnSplit spp = sp;  // the pass's own plan and helpers' table (pass 0: the batch's, written by the launch before)
  if (pass > 0u) {
    const int how = chain_wait<true>(ch.ready + cell, 2u, ch.err);
    if (!how) return;
    if (ch.dbg && (threadIdx.x & 63) == 0) atomicAdd(ch.dbg + ((size_t)pass * n_jobs + job) * 16 + 8 + how, 1u);  // dev: slots 9, 10, 11
    if constexpr (SPLIT) {
      spp.plan = ch.planp + (size_t)(pass - 1u) * n_jobs * n_part;
      spp.helper = ch.helperp + (size_t)(pass - 1u) * n_jobs * sp.hx;
    }
  }
  if (first_wave) chain_stamp(ch, pass, job, n_jobs, 1, 0);  // ... and past the wait
  nn_compact_body<CS, false, false, SPLIT, true, false, true>(NnPos{grp, slot, wgv, 0u, pass, job}, jobs, n_jobs, job_group, n_wg, subs, states, corr, corr,
                                                              d2out, pairs, partials, n_part, ld, gate2, spp, hv, stat_pairs, trace, ch);
  // (this wave's stores acknowledged, then its count -- not waited for)
  __builtin_amdgcn_s_waitcnt(0);
  asm volatile("" ::: "memory");
  if ((threadIdx.x & 63) == 0) (void)__hip_atomic_fetch_add(ch.done + cell, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
#undef NN_COMPACT_PARAMS
#undef NN_COMPACT_ARGS

}  // namespace reg
}  // namespace gloc
