// reg_kernels.hpp -- device kernels of the batched candidate registration (gfx950).
//
//   K4 nn_kernel            exact brute-force 1-NN (coalesced LDS-staged targets, per-lane running
//                           min, per-chunk argmin bookkeeping, one-chunk rescan for the index)
//   K5 ransac_hyp_kernel    one thread per hypothesis: counter-RNG sample, 3-point Kabsch (fp64 SVD)
//      ransac_score_kernel  thread <-> hypothesis, correspondences broadcast from LDS, inlier counts
//      ransac_best_kernel   argmax (inliers, -h) per candidate
//   K6 accum_kernel         fp64 raw moments of the (moved source, matched target) pairs
//      solve_kernel         Kabsch from the moments, T <- dT * T (fp64 state), fp32 copy for K4
//
// All fp32 point arithmetic uses the fixed, un-fused order of oracle/reg_oracle.c (this unit is
// compiled with -ffp-contract=off): p' = ((r0 x + r1 y) + r2 z) + t, d2 = (dx dx + dy dy) + dz dz
// (nanoflann L2_Simple order, registration/nanoflann.hpp:521-532).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "math3.hpp"          // f32x4, xform, dist2, jacobi_eig3, cross3, mulhi_idx, NN_FAR
#include "lane_ops.hpp"
#include "scan_index.hpp"     // ScanIndexDev, CH, SB
#include "synth_kernels.hpp"  // mix64 / rng_key / rng_draw

namespace gloc {
namespace reg {


// One registration job = (query scan, candidate scan).  A batch holds the jobs of ALL queries in
// flight (n_queries x n_cand), so that every launch covers them all.  Everything downstream of the
// scans is indexed in SORTED space: slot i of job c is the i-th point of the query's Hilbert order,
// corr[c][i] the matched target's sorted position (0xFFFFFFFF: none).
struct Job {
  const f32x4* src_pts;       // query scan, Hilbert order: x, y, z, bits(original index)
  const uint32_t* src_order;  // source groups, widest first (launch order of the culled search)
  const uint32_t* src_inv;    // original source index -> sorted slot (RANSAC samples original ids)
  const float* tgt_xyz;       // candidate scan, original order, packed (exhaustive search)
  ScanIndexDev tgt;           // candidate scan, search index
  uint32_t n_src, n_groups;   // n_groups = ceil(n_src / (64 * sources per lane))
  uint32_t cand_id;           // RANSAC stream id (rank in the retrieval list)
  uint32_t pad_;
};

// ---- heavy source groups over several waves (round 4) -------------------------------------------------------------
// A launch cannot end before its longest wave does, and a few waves are 3 - 6 x the mean: the far-field groups of every
// scan (128 points in a box of 60 m x 130 m: ~880 of the 969 chunk boxes pass the wave-level test and are rejected lane
// by lane) and, against a different scene, groups whose points are metres from everything (40+ processed chunks).  With
// few jobs per launch (one query alone: 20) that wave IS the launch time.  Such a group is searched by P = 2, 4 or 8
// waves: part p takes the candidate chunks c with c % P == p (the wave-level ballot is masked), runs the unchanged
// search over them -- every part starts from the same upper bounds, so no part can cull the true neighbour -- and folds
// its (un-fused distance, original index) per source into a 64-bit key in global memory (atomicMin: the smallest
// distance, the smallest original index among equals: the reference's rule); the part that arrives last (a ticket)
// reads the keys back and writes the outputs and the moments the single wave would have written, bit for bit.
// Which groups: a work estimate per (job, group) written by every pass (candidate chunks, processed chunks, items, in
// units of the cycles they cost), turned into next pass's plan by solve_kernel, which runs between two passes anyway.
// The waves of a planned group ("helpers", all its parts) are the first sp.hx work-groups of a job in the launch order,
// the heaviest class first: a heavy wave that starts late outlasts the launch however it is split.  The group's own
// wave, at its rank in the launch order, then exits at once.
struct NnSplit {
  uint32_t* work;             // [job][n_part]: the estimate, accumulated by the parts, consumed + zeroed by the planner
  uint32_t* plan;             // [job][n_part]: (ordinal among the job's split groups << 8) | parts; 0: not planned (its own wave)
  uint32_t* helper;           // [job][hx]: rank | part << 20 | parts << 24, or NN_NO_HELPER
  unsigned long long* skey;   // [job][hx][64 * CS]: (bits(d2) << 32) | original index; ~0 between passes
  uint32_t* ticket;           // [job][hx]: parts arrived; 0 between passes
  uint32_t hx;                // helper slots per job (0: no plan, the pointers are null)
  uint32_t thresh;            // estimate above which a group is split (cycles)
};
// Round 6 -- a COLD pass has no earlier pass to plan from, and some of its waves are heavy by the geometry, not by a loose
// bound: a source on an object next to the sensor whose nearest target is 1.5 m off has 3 000 ground points inside that
// ball, and 128 such sources made waves of 3 M cycles (1.4 ms -- the launch of 500 jobs waited 2.5 ms for the last of
// them; one query alone waits for its slowest).  A cold wave that has processed `thresh` chunks therefore GIVES UP:
// it appends (job, group) to a list and leaves without output; a second launch behind the pass (nn_compact_heavy_kernel)
// searches every listed group again with NN_HEAVY_PARTS waves, candidate chunks dealt c mod parts, folded through
// device-scope atomics as the planned split does (skey / ticket, self-resetting).  Same bits as the single wave.
// The wave hands its bounds over (hkey): a part sees an eighth of the chunks, and left to its own finds would search with
// the nearest point of ITS eighth as the bound -- on dense ground a ball as full as the whole wave's (measured: the second
// launch took as long as the waves it replaced); with the bounds of the wave that gave up the parts only add what is
// nearer, and skip the cold start's key search.
struct NnHeavy {
  uint32_t* count;            // entries appended (may exceed cap: the surplus went on by itself)
  uint32_t* list;             // [cap][2]: job, group rank
  unsigned long long* skey;   // [cap][64 * CS]; ~0 between passes
  unsigned long long* hkey;   // [cap][64 * CS]: the search state of the wave that gave up -- per source (bits(best d2) << 32) |
                              // tie flag << 31 | (sub-block << 2 | quarter) holding it: the parts START from these bounds
  uint32_t* ticket;           // [cap]; 0 between passes
  uint32_t cap;               // 0: off
  uint32_t thresh;            // processed chunks at which a wave gives up
};
#ifndef GLOC_NN_HEAVY_PARTS
#define GLOC_NN_HEAVY_PARTS 8  // 2, 4, 8 or 16
#endif
constexpr uint32_t NN_HEAVY_PARTS = GLOC_NN_HEAVY_PARTS;
// Round 6, late -- the ICP passes of a SMALL batch CHAINED in one launch (nn_chain_kernel, nn_compact.hpp).  One query
// alone is 20 jobs x 21 passes: 21 x (a search launch of 19 380 waves over 6 144 slots that ramps up, drains and waits for
// its slowest wave, + a solve launch with the chip idle) = 121 + 14 us per pass where the waves themselves need ~80.  The
// jobs of a batch never depend on each other, only job j's pass p + 1 on job j's solve of pass p.  The chained launch is
// the grid of ALL warm passes laid end to end -- pass-major, inside a pass the launch order of one pass, and behind
// every group of jobs the few waves that reduce its moments, solve and plan ("roles") -- and a wave of pass p + 1
// waits at its head until its job's pass p has been solved and planned (`ready`).  Work-groups of a launch are handed
// out in index order (per XCD, each XCD walking its own eighth), and everything a wave waits for has a smaller index:
// the wave with the smallest unfinished index never waits for anything unfinished, so the launch always advances, and
// while one job's solve is under way the other jobs' waves fill the chip.  No host round trip, no launch boundary, no
// ramp: the chip stays full from the first pass to the last.
// What crosses from one wave to another inside the launch goes past the XCDs' caches (relaxed device-scope atomics:
// T, the plan, the helpers' table, the work estimates, the fold keys and tickets, the moments; corr and d2 are written
// through so that no XCD keeps a stale dirty line for the end of the launch) -- never a cache write-back.  A warm wave
// that reads an OLDER pass's correspondence for its bound (a line its XCD still holds) searches from a looser, still valid,
// bound: same bits.  Every wait is bounded (NN_CHAIN_WAIT_TICKS of the 100 MHz clock): a wave that runs out sets `err`,
// every waiting wave leaves when it sees it, and the host runs the batch again launch by launch and stops chaining on the handle.
// Hot spots (measured, the first version: 21 ms for one query where launch by launch takes 3.2): 1 225 waves per job and
// pass reading the SAME pose / plan / flag words past the caches are served one after the other where the line lives
// (~50 ns each: 1 ms per pass), and counters of different jobs in one line serialise the jobs.  Hence: whatever a pass's
// waves read lives at an address of ITS OWN per pass (pose, plan, helpers' table, the ready flag), written through
// before the flag, and nobody reads it before that -- an ORDINARY load then finds no stale copy anywhere and the XCD's L2
// serves the other 1 200 waves; only a wave that finds the flag not yet set polls past the caches.  Every counter and
// flag has a 256-byte line to itself.
struct NnChain {
  uint32_t* ready;   // [pass][job] x NN_CHAIN_PAD: 2 when the pass before has been solved and planned for the job (pass 0: never read)
  uint32_t* done;    // [pass][job] x NN_CHAIN_PAD: search waves of the pass that have left
  uint32_t* go;      // [pass][job] x NN_CHAIN_PAD: set by the last of them (what the role waves poll: not the counters' line)
  uint32_t* sdone;   // [pass][job] x NN_CHAIN_PAD: reducer waves that have stored their sub-sums
  float* Tp;         // [pass][job][NN_CHAIN_T_STRIDE]: the pose the pass's waves move the sources by (pass 0: the state's)
  uint32_t* planp;   // [pass - 1][job][n_part]: NnSplit::plan of passes 1 .. (pass 0: the batch's own)
  uint32_t* helperp; // [pass - 1][job][hx]: NnSplit::helper of passes 1 ..
  double* sub;       // [job][NN_CHAIN_RED][ACC_NV]: the reducers' sub-sums of the moments (solve_kernel's sub[][])
  uint32_t* err;     // != 0: a wait ran out
  uint32_t* dbg;     // dev (gloc_reg_debug_chain_trace): [pass][job][16] stamps of the 100 MHz clock, or null
  uint32_t n_pass;
  uint32_t expected;      // search waves per job and pass
  uint32_t jobs_per_grp;  // jobs of a group of slots (job_group / subs)
  uint32_t grp_size;      // work-groups of a group in one pass: job_group * n_wg searches + the roles, padded to a multiple of 8
  uint32_t pass_size;     // work-groups of a pass
};
constexpr uint32_t NN_CHAIN_PAD = 64;        // words between two counters / flags (256 B)
constexpr uint32_t NN_CHAIN_T_STRIDE = 32;   // floats per (pass, job) pose slot (128 B: a line of its own)
constexpr uint32_t NN_CHAIN_RED = 16;                  // reducer waves per job: solve_kernel's 16 waves, one each
constexpr uint32_t NN_CHAIN_ROLES = NN_CHAIN_RED + 1;  // + the planner
constexpr unsigned long long NN_CHAIN_WAIT_TICKS = 300000000ull;  // 3 s of the 100 MHz clock
constexpr uint32_t NN_NO_HELPER = 0xFFFFFFFFu;
constexpr uint32_t NN_MAX_PARTS = 8;
// the estimate, from the trace's regression of wave cycles on its counts (tools/dev_nn_trace3.py)
constexpr uint32_t NN_W_FIXED = 21000, NN_W_CAND = 300, NN_W_CHUNK = 2800, NN_W_ITEM = 37;

// waves for an estimate: 0 = the group's own wave at its rank in the launch order; 2, 4, 8 = that many parts in helper slots
// (a class "one wave, but started early" for groups above half the threshold was tried: it used up the slots)
__host__ __device__ __forceinline__ uint32_t nn_parts_for(uint32_t w, uint32_t thresh) {
  if (thresh == 0 || w <= thresh) return 0;
  return w > 4 * (unsigned long long)thresh ? 8u : (w > 2 * (unsigned long long)thresh ? 4u : 2u);
}

// per-candidate state, device resident
struct CandState {
  double Td[12];      // current absolute transform (R row-major 9, t 3), fp64
  float Tf[12];       // fp32 copy used to move points
  float bestRt[12];   // best RANSAC hypothesis (fp32)
  uint32_t best_h, best_inl;
  int ok, frozen;
  double sum_d2;      // of the last S1 pass
  uint32_t niters;    // RANSAC: iterations still allowed (adaptive stop), starts at ransac_iters
  int ransac_done;
  float last_step;    // RMS displacement of the matched points by the last ICP update (gloc_reg_params.max_final_step)
  float pad_;
};

// ---------------------------------------------------------------------------------------------
// K4 (exhaustive form, the kernel north_star names).  grid = (ceil(max n_src / (256*NN_S)), n_jobs).
// Each lane owns NN_S source points (sorted slots); the targets are streamed IN ORIGINAL ORDER, 256 at
// a time into LDS as float4 and read back as wave-uniform broadcasts, so "first minimum" is the
// smallest original index.  The stored correspondence is the winner's sorted position.
constexpr int NN_S = 4;
constexpr int NN_TC = 256;

__global__ __launch_bounds__(256) void nn_kernel(const Job* __restrict__ jobs,
                                                 const CandState* __restrict__ states,
                                                 uint32_t* __restrict__ corr,
                                                 float* __restrict__ d2out, size_t ld) {
  __shared__ f32x4 tl[2][NN_TC];
  const int tid = threadIdx.x;
  const int job = blockIdx.y;
  const Job& J = jobs[job];
  const uint32_t n_src = J.n_src;
  const uint32_t base = blockIdx.x * (256 * NN_S);
  if (base >= n_src) return;  // uniform over the work-group
  const float* __restrict__ tgt = J.tgt_xyz;
  const uint32_t n_tgt = J.tgt.n;
  float T[12];
#pragma unroll
  for (int i = 0; i < 12; ++i) T[i] = states[job].Tf[i];

  float px[NN_S], py[NN_S], pz[NN_S], best[NN_S];
  uint32_t bchunk[NN_S];
#pragma unroll
  for (int s = 0; s < NN_S; ++s) {
    const uint32_t i = base + s * 256 + tid;
    f32x4 p = {0.f, 0.f, 0.f, 0.f};
    if (i < n_src) p = J.src_pts[i];
    xform(T, p.x, p.y, p.z, px[s], py[s], pz[s]);
    best[s] = 3.402823466e+38f;
    bchunk[s] = 0;
  }
  const uint32_t nchunks = (n_tgt + NN_TC - 1) / NN_TC;
  auto stage = [&](uint32_t c, int buf) {
    const uint32_t j = c * NN_TC + tid;
    f32x4 v = {NN_FAR, NN_FAR, NN_FAR, 0.f};
    if (j < n_tgt) {
      v.x = tgt[3 * (size_t)j + 0];
      v.y = tgt[3 * (size_t)j + 1];
      v.z = tgt[3 * (size_t)j + 2];
    }
    tl[buf][tid] = v;
  };
  if (nchunks) stage(0, 0);
  __syncthreads();
  for (uint32_t c = 0; c < nchunks; ++c) {
    const int buf = c & 1;
    if (c + 1 < nchunks) stage(c + 1, buf ^ 1);
    float m[NN_S];
#pragma unroll
    for (int s = 0; s < NN_S; ++s) m[s] = 3.402823466e+38f;
#pragma unroll 8
    for (int t = 0; t < NN_TC; t += 2) {
      const f32x4 q0 = tl[buf][t];
      const f32x4 q1 = tl[buf][t + 1];
#pragma unroll
      for (int s = 0; s < NN_S; ++s) {
        const float d0 = dist2(px[s], py[s], pz[s], q0.x, q0.y, q0.z);
        const float d1 = dist2(px[s], py[s], pz[s], q1.x, q1.y, q1.z);
        m[s] = fminf(fminf(m[s], d0), d1);  // v_min3_f32
      }
    }
#pragma unroll
    for (int s = 0; s < NN_S; ++s) {
      if (m[s] < best[s]) {  // strict: the earliest chunk holding the minimum wins
        best[s] = m[s];
        bchunk[s] = c;
      }
    }
    __syncthreads();
  }
  // index recovery: first target of the winning chunk whose distance equals the minimum
#pragma unroll
  for (int s = 0; s < NN_S; ++s) {
    const uint32_t i = base + s * 256 + tid;
    if (i >= n_src) continue;
    uint32_t bj = 0xFFFFFFFFu;
    if (n_tgt) {
      const uint32_t j0 = bchunk[s] * NN_TC;
      const uint32_t j1 = (j0 + NN_TC) < n_tgt ? (j0 + NN_TC) : n_tgt;
      for (uint32_t j = j0; j < j1; ++j) {
        const float d = dist2(px[s], py[s], pz[s], tgt[3 * (size_t)j], tgt[3 * (size_t)j + 1],
                              tgt[3 * (size_t)j + 2]);
        if (d == best[s]) {
          bj = J.tgt.inv[j];
          break;
        }
      }
    }
    corr[(size_t)job * ld + i] = bj;
    d2out[(size_t)job * ld + i] = best[s];
  }
}

// ---------------------------------------------------------------------------------------------
// (moved source, matched target) pairs as 2 x float4 per sorted slot, so the RANSAC scorer reads them
// coalesced.  Only the exhaustive search needs this pass: the culled search writes the pairs itself.
__global__ void gather_pairs_kernel(const Job* __restrict__ jobs, const CandState* __restrict__ states,
                                    const uint32_t* __restrict__ corr, size_t ld,
                                    f32x4* __restrict__ pairs) {
  const int job = blockIdx.y;
  const Job& J = jobs[job];
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= J.n_src) return;
  float T[12];
#pragma unroll
  for (int k = 0; k < 12; ++k) T[k] = states[job].Tf[k];
  const f32x4 p = J.src_pts[i];
  float x, y, z;
  xform(T, p.x, p.y, p.z, x, y, z);
  const uint32_t j = corr[(size_t)job * ld + i];
  f32x4 q = {NN_FAR, NN_FAR, NN_FAR, 0.f};  // no correspondence (empty target): never an inlier
  if (j < J.tgt.n) {
    q = J.tgt.pts[j];
    q.w = 0.f;
  }
  pairs[((size_t)job * ld + i) * 2 + 0] = f32x4{x, y, z, 0.f};
  pairs[((size_t)job * ld + i) * 2 + 1] = q;
}

// Correspondences in the caller's terms (gloc_reg_nn): original source index -> original target index.
__global__ void export_corr_kernel(const Job* __restrict__ jobs, const uint32_t* __restrict__ corr,
                                   const float* __restrict__ d2in, size_t ld,
                                   uint32_t* __restrict__ out_idx, float* __restrict__ out_d2) {
  const int job = blockIdx.y;
  const Job& J = jobs[job];
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= J.n_src) return;
  const uint32_t o = __float_as_uint(J.src_pts[i].w);
  const uint32_t j = corr[(size_t)job * ld + i];
  out_idx[(size_t)job * ld + o] = j < J.tgt.n ? __float_as_uint(J.tgt.pts[j].w) : 0xFFFFFFFFu;
  out_d2[(size_t)job * ld + o] = d2in[(size_t)job * ld + i];
}

// The small arrays of the fp64 solve: registers -- or, WS, a workspace the caller hands in (LDS: the chained launch's
// solver waves run inside the search kernel's register budget, nn_compact.hpp).  Same operations either way.
#define GLOC_WS_ARR(name, n)          \
  double name##_priv[WS ? 1 : (n)];   \
  double* const name = WS ? ws : name##_priv; \
  if (WS) ws += (n)
template <bool WS>
__device__ inline void kabsch_from_cov_t(const double* M /* [9] */, const double* pbar /* [3] */,
                                                  const double* qbar /* [3] */, double* R /* [9] */, double* t /* [3] */, double* ws) {
  GLOC_WS_ARR(A, 9);
  GLOC_WS_ARR(V, 9);
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j)
      A[3 * i + j] = (M[0 + i] * M[0 + j] + M[3 + i] * M[3 + j]) + M[6 + i] * M[6 + j];
  jacobi_eig3(A, V);
  int o0 = 0, o1 = 1, o2 = 2;
  double l0 = A[0], l1 = A[4], l2 = A[8];
  if (l1 > l0) { int ti = o0; o0 = o1; o1 = ti; double td = l0; l0 = l1; l1 = td; }
  if (l2 > l1) { int ti = o1; o1 = o2; o2 = ti; double td = l1; l1 = l2; l2 = td; }
  if (l1 > l0) { int ti = o0; o0 = o1; o1 = ti; double td = l0; l0 = l1; l1 = td; }
  (void)o2; (void)l2;
  GLOC_WS_ARR(v1, 3);
  GLOC_WS_ARR(v2, 3);
  GLOC_WS_ARR(v3, 3);
  v1[0] = V[0 + o0]; v1[1] = V[3 + o0]; v1[2] = V[6 + o0];
  v2[0] = V[0 + o1]; v2[1] = V[3 + o1]; v2[2] = V[6 + o1];
  cross3(v1, v2, v3);
  GLOC_WS_ARR(u1, 3);
  GLOC_WS_ARR(u2, 3);
  GLOC_WS_ARR(u3, 3);
  for (int i = 0; i < 3; ++i) {
    u1[i] = (M[3 * i + 0] * v1[0] + M[3 * i + 1] * v1[1]) + M[3 * i + 2] * v1[2];
    u2[i] = (M[3 * i + 0] * v2[0] + M[3 * i + 1] * v2[1]) + M[3 * i + 2] * v2[2];
  }
  double n1 = sqrt((u1[0] * u1[0] + u1[1] * u1[1]) + u1[2] * u1[2]);
  if (!(n1 > 1e-300)) {
    for (int i = 0; i < 9; ++i) R[i] = (i % 4 == 0) ? 1.0 : 0.0;
    for (int i = 0; i < 3; ++i) t[i] = qbar[i] - pbar[i];
    return;
  }
  for (int i = 0; i < 3; ++i) u1[i] = u1[i] / n1;
  const double d12 = (u1[0] * u2[0] + u1[1] * u2[1]) + u1[2] * u2[2];
  for (int i = 0; i < 3; ++i) u2[i] = u2[i] - d12 * u1[i];
  double n2 = sqrt((u2[0] * u2[0] + u2[1] * u2[1]) + u2[2] * u2[2]);
  if (!(n2 > 1e-300)) {
    GLOC_WS_ARR(ax, 3);
    ax[0] = 0; ax[1] = 0; ax[2] = 0;
    const double a0 = u1[0] < 0 ? -u1[0] : u1[0], a1 = u1[1] < 0 ? -u1[1] : u1[1],
                 a2 = u1[2] < 0 ? -u1[2] : u1[2];
    ax[(a0 <= a1 && a0 <= a2) ? 0 : ((a1 <= a2) ? 1 : 2)] = 1.0;
    cross3(u1, ax, u2);
    n2 = sqrt((u2[0] * u2[0] + u2[1] * u2[1]) + u2[2] * u2[2]);
  }
  for (int i = 0; i < 3; ++i) u2[i] = u2[i] / n2;
  cross3(u1, u2, u3);
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) R[3 * i + j] = (v1[i] * u1[j] + v2[i] * u2[j]) + v3[i] * u3[j];
  for (int i = 0; i < 3; ++i)
    t[i] = qbar[i] - ((R[3 * i + 0] * pbar[0] + R[3 * i + 1] * pbar[1]) + R[3 * i + 2] * pbar[2]);
}
__device__ inline void kabsch_from_cov(const double M[9], const double pbar[3], const double qbar[3], double R[9], double t[3]) {
  kabsch_from_cov_t<false>(M, pbar, qbar, R, t, nullptr);
}

// K5a.  One thread per (job, hypothesis).  pairs: [job][ld][2] float4 by sorted slot; the three
// sampled ids are ORIGINAL source indices (as the oracle samples them), mapped through src_inv.
// Hypotheses [h_begin, h_end) of every job; with `states`, jobs whose adaptive iteration count has
// already been reached are skipped (their later hypotheses are never looked at).
__global__ void ransac_hyp_kernel(const f32x4* __restrict__ pairs, size_t ld,
                                  const Job* __restrict__ jobs, uint64_t seed,
                                  uint32_t n_hyp, uint32_t h_begin, uint32_t h_end,
                                  const CandState* __restrict__ states,
                                  float* __restrict__ Rt /* [job][n_hyp][12] */,
                                  uint32_t* __restrict__ valid) {
  const int cand = blockIdx.y;
  const uint32_t h = h_begin + blockIdx.x * blockDim.x + threadIdx.x;
  if (h >= h_end) return;
  if (states && states[cand].ransac_done) return;
  const size_t o = (size_t)cand * n_hyp + h;
  valid[o] = 0;
  const uint32_t n = jobs[cand].n_src;
  if (n < 3) return;
  const uint32_t* __restrict__ inv = jobs[cand].src_inv;
  const uint64_t key = synth::rng_key(seed, ((uint64_t)jobs[cand].cand_id << 32) | (uint64_t)h);
  uint64_t ctr = 0;
  uint32_t s0 = mulhi_idx(synth::rng_draw(key, ctr++), n), s1 = s0, s2 = s0;
  for (int tries = 0; tries < 16 && s1 == s0; ++tries) s1 = mulhi_idx(synth::rng_draw(key, ctr++), n);
  for (int tries = 0; tries < 16 && (s2 == s0 || s2 == s1); ++tries)
    s2 = mulhi_idx(synth::rng_draw(key, ctr++), n);
  if (s0 == s1 || s0 == s2 || s1 == s2) return;
  const uint32_t sidx[3] = {s0, s1, s2};
  double p[3][3], q[3][3];
  for (int k = 0; k < 3; ++k) {
    const uint32_t slot = inv ? inv[sidx[k]] : sidx[k];
    const f32x4 pv = pairs[((size_t)cand * ld + slot) * 2 + 0];
    const f32x4 qv = pairs[((size_t)cand * ld + slot) * 2 + 1];
    p[k][0] = (double)pv.x; p[k][1] = (double)pv.y; p[k][2] = (double)pv.z;
    q[k][0] = (double)qv.x; q[k][1] = (double)qv.y; q[k][2] = (double)qv.z;
    if (qv.x >= 0.5f * NN_FAR) return;  // sampled a point without correspondence (empty target scan)
  }
  double a[3], b[3], c[3];
  for (int i = 0; i < 3; ++i) {
    a[i] = p[1][i] - p[0][i];
    b[i] = p[2][i] - p[0][i];
  }
  cross3(a, b, c);
  const double aa = (a[0] * a[0] + a[1] * a[1]) + a[2] * a[2];
  const double bb = (b[0] * b[0] + b[1] * b[1]) + b[2] * b[2];
  const double cc = (c[0] * c[0] + c[1] * c[1]) + c[2] * c[2];
  if (!(aa > 1e-12) || !(bb > 1e-12) || !(cc > 1e-6 * (aa * bb))) return;
  double pbar[3], qbar[3], M[9];
  for (int i = 0; i < 3; ++i) {
    pbar[i] = ((p[0][i] + p[1][i]) + p[2][i]) / 3.0;
    qbar[i] = ((q[0][i] + q[1][i]) + q[2][i]) / 3.0;
  }
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j)
      M[3 * i + j] = ((p[0][i] - pbar[i]) * (q[0][j] - qbar[j]) +
                      (p[1][i] - pbar[i]) * (q[1][j] - qbar[j])) +
                     (p[2][i] - pbar[i]) * (q[2][j] - qbar[j]);
  double Rd[9], td[3];
  kabsch_from_cov(M, pbar, qbar, Rd, td);
  float* out = Rt + o * 12;
  for (int i = 0; i < 9; ++i) out[i] = (float)Rd[i];
  for (int i = 0; i < 3; ++i) out[9 + i] = (float)td[i];
  valid[o] = 1;
}

// K5b.  thread <-> hypothesis (R,t in registers); a chunk of SC correspondences per work-group
// streams through LDS and is read as wave-uniform broadcasts.  grid = (hyp tiles, corr chunks, cand)
constexpr int SC_CHUNK = 4096;
constexpr int SC_STAGE = 256;

__global__ __launch_bounds__(256) void ransac_score_kernel(const f32x4* __restrict__ pairs,
                                                           size_t ld, const Job* __restrict__ jobs, uint32_t n_hyp,
                                                           uint32_t h_begin, uint32_t hyp_per_block,
                                                           const float* __restrict__ Rt,
                                                           const uint32_t* __restrict__ valid,
                                                           float thr2,
                                                           const CandState* __restrict__ states,
                                                           uint32_t* __restrict__ inliers,
                                                           const uint32_t* __restrict__ alive_idx /* [cand][n_hyp] or null */,
                                                           const uint32_t* __restrict__ alive_cnt, uint32_t chunk0,
                                                           uint32_t chunk_len /* pairs per work-group: a multiple of SC_STAGE (SC_CHUNK; less in small batches) */) {
  // staged pairs, two correspondences per entry, structure-of-arrays: (px0 px1 py0 py1)(pz0 pz1 qx0 qx1)(qy0 qy1 qz0 qz1)
  // so that one packed fp32 instruction moves / measures two correspondences.
  //
  // The count must be the oracle's: residual = dist2(xform(T, p), q), every product and sum rounded, inlier iff < thr2.
  // Round 3: the residual is first computed FUSED (three packed fma per coordinate, one multiply and two fma for the
  // square: 15 packed instructions per two pairs instead of 26), which differs from the un-fused value by a bounded
  // amount: a coordinate of xform() by at most 9 roundings of partial sums no larger than B = max_row |T| x M + |t|
  // (M: the largest coordinate of the staged tile, found while it is staged), the difference to q by two more of
  // B + M, the squares and sums by 3e-7 relative.  A fused residual outside thr2 -+ band is decided as it stands; inside
  // the band (a few pairs in ten thousand) the un-fused form is computed and decides.  Same counts, bit for bit.
  __shared__ f32x4 sp[3 * (SC_STAGE / 2)];
  __shared__ float tile_max[4];
  typedef float f32x2 __attribute__((ext_vector_type(2)));
  const int cand = blockIdx.z;
  if (states && states[cand].ransac_done) return;  // adaptive stop reached in an earlier phase
  const uint32_t n = jobs[cand].n_src;
  const uint32_t chunk = blockIdx.y + chunk0;
  if (chunk * chunk_len >= n) return;
  // hyp_per_block = 256: thread <-> hypothesis.  64 / 16: the four waves share that many hypotheses, each thread taking
  // a quarter / a sixteenth of every staged tile (the first phases of the adaptive RANSAC need few hypotheses).
  const uint32_t sub = threadIdx.x / hyp_per_block, nsub = 256 / hyp_per_block;
  uint32_t h = h_begin + blockIdx.x * hyp_per_block + threadIdx.x % hyp_per_block;
  bool hv = h < n_hyp && h < h_begin + (blockIdx.x + 1) * hyp_per_block && valid[(size_t)cand * n_hyp + h];
  if (alive_idx) {  // the hypotheses still in the race, compacted (ransac_alive_kernel): thread <-> list entry
    const uint32_t na = alive_cnt[cand], e = blockIdx.x * hyp_per_block + threadIdx.x % hyp_per_block;
    if (blockIdx.x * hyp_per_block >= na) return;  // (uniform over the work-group)
    hv = e < na;
    h = hv ? alive_idx[(size_t)cand * n_hyp + e] : 0u;
  }
  float T[12];
#pragma unroll
  for (int i = 0; i < 12; ++i) T[i] = hv ? Rt[((size_t)cand * n_hyp + h) * 12 + i] : 0.f;
  const float row = fmaxf(fmaxf(fabsf(T[0]) + fabsf(T[1]) + fabsf(T[2]), fabsf(T[3]) + fabsf(T[4]) + fabsf(T[5])),
                          fabsf(T[6]) + fabsf(T[7]) + fabsf(T[8]));
  const float tmax = fmaxf(fmaxf(fabsf(T[9]), fabsf(T[10])), fabsf(T[11]));
  const float thr = sqrtf(thr2);
  const uint32_t i0 = chunk * chunk_len;
  const uint32_t i1 = (i0 + chunk_len) < n ? (i0 + chunk_len) : n;
  uint32_t cnt = 0;
  float* spf = reinterpret_cast<float*>(sp);
  for (uint32_t b = i0; b < i1; b += SC_STAGE) {
    const uint32_t i = b + threadIdx.x;
    f32x4 pv = {0.f, 0.f, 0.f, 0.f}, qv = {NN_FAR, NN_FAR, NN_FAR, 0.f};  // padding: never inlier
    if (i < i1) {
      pv = pairs[((size_t)cand * ld + i) * 2 + 0];
      qv = pairs[((size_t)cand * ld + i) * 2 + 1];
    }
    // the tile's largest coordinate (pairs without a target carry NN_FAR: they are decided by magnitude alone)
    float mx = fmaxf(fmaxf(fabsf(pv.x), fabsf(pv.y)), fabsf(pv.z));
    if (qv.x < 0.5f * NN_FAR) mx = fmaxf(mx, fmaxf(fmaxf(fabsf(qv.x), fabsf(qv.y)), fabsf(qv.z)));
    for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o));
    __syncthreads();
    if ((threadIdx.x & 63) == 0) tile_max[threadIdx.x >> 6] = mx;
    {
      float* d = spf + (threadIdx.x >> 1) * 12 + (threadIdx.x & 1);
      d[0] = pv.x; d[2] = pv.y; d[4] = pv.z; d[6] = qv.x; d[8] = qv.y; d[10] = qv.z;
    }
    __syncthreads();
    const float M = fmaxf(fmaxf(tile_max[0], tile_max[1]), fmaxf(tile_max[2], tile_max[3]));
    const float B = row * M + tmax;
    const float dd = 1.7321f * 8.5e-7f * (B + M);            // |fused - un-fused| of the residual's length, at most
    const float band = (2.f * thr * dd + dd * dd) + 2.0e-6f * thr2;
    const float lo2 = thr2 - band, hi2 = thr2 + band;        // (a NaN or infinite bound sends every pair to the exact form)
    const int t_begin = (int)(sub * (SC_STAGE / 2 / nsub)), t_end = (int)((sub + 1) * (SC_STAGE / 2 / nsub));
#pragma unroll 4
    for (int t = t_begin; t < t_end; ++t) {
      const f32x4 a = sp[3 * t], bq = sp[3 * t + 1], c = sp[3 * t + 2];
      const f32x2 px = {a.x, a.y}, py = {a.z, a.w}, pz = {bq.x, bq.y}, qx = {bq.z, bq.w}, qy = {c.x, c.y}, qz = {c.z, c.w};
      auto fma2 = [](f32x2 u, f32x2 v, f32x2 w) { return __builtin_elementwise_fma(u, v, w); };
      const f32x2 xf = fma2(f32x2{T[0], T[0]}, px, fma2(f32x2{T[1], T[1]}, py, fma2(f32x2{T[2], T[2]}, pz, f32x2{T[9], T[9]})));
      const f32x2 yf = fma2(f32x2{T[3], T[3]}, px, fma2(f32x2{T[4], T[4]}, py, fma2(f32x2{T[5], T[5]}, pz, f32x2{T[10], T[10]})));
      const f32x2 zf = fma2(f32x2{T[6], T[6]}, px, fma2(f32x2{T[7], T[7]}, py, fma2(f32x2{T[8], T[8]}, pz, f32x2{T[11], T[11]})));
      const f32x2 ex = xf - qx, ey = yf - qy, ez = zf - qz;
      const f32x2 df = fma2(ez, ez, fma2(ey, ey, ex * ex));
      const bool in0 = df.x < lo2, in1 = df.y < lo2;
      cnt += (in0 ? 1u : 0u) + (in1 ? 1u : 0u);
      // neither clearly inside nor clearly outside (written so that a NaN bound or residual lands here too)
      const bool un0 = !in0 && !(df.x >= hi2), un1 = !in1 && !(df.y >= hi2);
      if (__builtin_amdgcn_ballot_w64(un0 || un1) != 0ull) {
        // xform(): ((r0 x + r1 y) + r2 z) + t, then dist2(): (dx dx + dy dy) + dz dz -- the oracle's roundings
        const f32x2 x = ((f32x2{T[0], T[0]} * px + f32x2{T[1], T[1]} * py) + f32x2{T[2], T[2]} * pz) + f32x2{T[9], T[9]};
        const f32x2 y = ((f32x2{T[3], T[3]} * px + f32x2{T[4], T[4]} * py) + f32x2{T[5], T[5]} * pz) + f32x2{T[10], T[10]};
        const f32x2 z = ((f32x2{T[6], T[6]} * px + f32x2{T[7], T[7]} * py) + f32x2{T[8], T[8]} * pz) + f32x2{T[11], T[11]};
        const f32x2 dx = x - qx, dy = y - qy, dz = z - qz;
        const f32x2 d2 = (dx * dx + dy * dy) + dz * dz;
        cnt += (un0 && d2.x < thr2) ? 1u : 0u;
        cnt += (un1 && d2.y < thr2) ? 1u : 0u;
      }
    }
  }
  if (hyp_per_block < 256u) {  // (uniform; every early return above is the whole work-group's)
    // The threads that share a hypothesis add up first: one atomic per hypothesis and work-group, not one per thread
    // (16 hypotheses: 16 threads each -- 256 atomics on one 64-byte line per work-group, and the line serves them one
    // after the other: the first phase of one query's 20 jobs took 75 us, 620 work-groups x 256).
    __shared__ uint32_t red[4][64];
    uint32_t c = hv ? cnt : 0u;
    if (hyp_per_block == 16u) {
      c += xor_lane_u32<16>(c);
      c += xor_lane_u32<32>(c);
    }
    red[threadIdx.x >> 6][threadIdx.x & 63] = c;
    __syncthreads();
    if (threadIdx.x < hyp_per_block) {
      const uint32_t tot = (red[0][threadIdx.x] + red[1][threadIdx.x]) + (red[2][threadIdx.x] + red[3][threadIdx.x]);
      if (hv && tot) atomicAdd(&inliers[(size_t)cand * n_hyp + h], tot);
    }
    return;
  }
  if (hv && cnt) atomicAdd(&inliers[(size_t)cand * n_hyp + h], cnt);
}

// K5b'.  Scoring EVERY hypothesis (ransac_confidence off: SURVEY App. B's wording of S2) without scoring every pair of
// every hypothesis.  The pairs are scored a part (an eighth) at a time; before a part, a hypothesis whose count so far plus
// all pairs still to come cannot exceed the best FULL count known (the winner of the first 256 hypotheses, which are
// scored completely first) is out of the race -- it can neither win nor tie ahead of that winner, which has a smaller
// index -- and the survivors are compacted into a list the scorer's threads map to.  Sound: the selected hypothesis and
// its count are the every-pair scorer's (the counts of dropped hypotheses stay partial, below the winner's).
// One work-group per candidate; the list keeps ascending order.
__global__ __launch_bounds__(1024) void ransac_alive_kernel(const uint32_t* __restrict__ inliers,
                                                            const uint32_t* __restrict__ valid, uint32_t n_hyp,
                                                            uint32_t h0, uint32_t h1, const Job* __restrict__ jobs,
                                                            const CandState* __restrict__ states, uint32_t first_pair,
                                                            uint32_t* __restrict__ alive_idx, uint32_t* __restrict__ alive_cnt) {
  __shared__ uint32_t wave_cnt[16], base_s;
  const int cand = blockIdx.x, tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const uint32_t n = jobs[cand].n_src, best = states[cand].best_inl;
  const uint32_t left = n > first_pair ? n - first_pair : 0u;
  if (tid == 0) base_s = 0;
  __syncthreads();
  for (uint32_t t0 = h0; t0 < h1; t0 += 1024) {
    const uint32_t h = t0 + (uint32_t)tid;
    bool keep = false;
    if (h < h1 && valid[(size_t)cand * n_hyp + h]) keep = inliers[(size_t)cand * n_hyp + h] + left > best;
    const unsigned long long m = __ballot(keep);
    if (lane == 0) wave_cnt[w] = (uint32_t)__popcll(m);
    __syncthreads();
    uint32_t off = base_s, tot = 0;
    for (int i = 0; i < 16; ++i) {
      if (i < w) off += wave_cnt[i];
      tot += wave_cnt[i];
    }
    if (keep) alive_idx[(size_t)cand * n_hyp + off + (uint32_t)__popcll(m & ((1ull << lane) - 1ull))] = h;
    __syncthreads();
    if (tid == 0) base_s += tot;
    __syncthreads();
  }
  if (tid == 0) alive_cnt[cand] = base_s;
}

// Iterations after which a 3-point RANSAC reaches `conf` with inlier ratio inl/n: smallest k with
// (1 - w^3)^k <= 1 - conf by repeated multiplication (no libm, so CPU and GPU agree), capped.
__device__ inline uint32_t ransac_needed_iters(uint32_t inl, uint32_t n, float conf,
                                               uint32_t max_iters) {
  const double w = (double)inl / (double)n;
  const double q = 1.0 - (w * w) * w;
  const double target = 1.0 - (double)conf;
  // A record with few inliers needs more iterations than the budget: the loop below would run to the cap, max_iters
  // dependent fp64 multiplications in one lane (3000: 17 us, and a job sees several such records -- the scan of the
  // first 64 hypotheses took 111 us of a 4 ms query).  q^max_iters above the target by a margin no rounding of the
  // loop can bridge (each product is within 2^-53 of exact: 3000 of them, 4e-13) says so without running it.
  const double lq = q > 0.0 ? log(q) : 0.0, lt = log(target);
  if (q > 0.0 && (double)max_iters * lq > lt + 1.0e-6) return max_iters;
  // Round 6: the count without the loop where no rounding can make it differ.  The loop returns the smallest k whose running
  // product fl(q^k) is <= target; the exact answer is ceil(ln target / ln q).  The running product is within k 2^-53 of
  // q^k, the quotient of the two logarithms within 1e-9 of its value for k <= 2^20 (check_params), and a relative error e
  // of the product moves the threshold in k by e / |ln q| <= e k / |ln target| -- under 3e-5 for a million iterations: a
  // quotient further than 1e-3 from the next integer decides.  (A record with a fifth of the points as inliers took 570
  // dependent multiplications, 5 us in one lane, and the different-world candidates of a query see several: the scan of
  // the first 16 hypotheses of one query's 20 jobs was 54 us.)
  if (q > 0.0 && q < 1.0 && target > 0.0 && target < 1.0) {
    const double ke = lt / lq, kc = ceil(ke);
    if (kc - ke > 1.0e-3 && kc - ke < 1.0 - 1.0e-3 && kc >= 1.0 && kc < (double)max_iters) return (uint32_t)kc;
  }
  double pw = 1.0;
  uint32_t k = 0;
  while (pw > target && k < max_iters) {
    pw = pw * q;
    k++;
  }
  return k;
}

// test aid (gloc_reg_debug_needed_iters): the device's count for arrays of (inliers, points)
__global__ void needed_iters_kernel(const uint32_t* __restrict__ inl, const uint32_t* __restrict__ n, float conf, uint32_t max_iters,
                                    uint32_t* __restrict__ out, uint32_t count) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < count) out[i] = ransac_needed_iters(inl[i], n[i], conf, max_iters);
}

// K5c.  The sequential RANSAC rule over hypotheses [h0, h1): best = first hypothesis with strictly
// more inliers than all before it; every new best may lower the iteration count (OpenCV's adaptive
// stop, which the reference runs with its default confidence); hypotheses at or beyond the count
// are never considered.  One wave per candidate: 64 hypotheses per step, records found with a
// prefix maximum and handled in order.  FINAL: also publish ok / best (R,t).
template <bool FINAL>
__global__ __launch_bounds__(64) void ransac_scan_kernel(const uint32_t* __restrict__ inliers,
                                                         const uint32_t* __restrict__ valid,
                                                         const float* __restrict__ Rt,
                                                         uint32_t n_hyp, uint32_t h0, uint32_t h1,
                                                         const Job* __restrict__ jobs, float conf,
                                                         float min_inlier_ratio,
                                                         CandState* __restrict__ states) {
  const int cand = blockIdx.x;
  const int lane = threadIdx.x;
  const uint32_t n = jobs[cand].n_src;
  CandState& st = states[cand];
  uint32_t niters = st.niters, best_inl = st.best_inl, best_h = st.best_h;
  int done = st.ransac_done;
  const bool adaptive = conf > 0.f && conf < 1.f;
  for (uint32_t base = h0; base < h1 && !done; base += 64) {
    if (base >= niters) {
      done = 1;
      break;
    }
    const uint32_t h = base + lane;
    uint32_t inl = 0;
    if (h < h1 && valid[(size_t)cand * n_hyp + h]) inl = inliers[(size_t)cand * n_hyp + h];
    // exclusive prefix maximum over the lanes, seeded with the best so far
    uint32_t pm = inl;
    for (int o = 1; o < 64; o <<= 1) {
      const uint32_t t = __shfl_up(pm, o);
      if (lane >= o) pm = pm > t ? pm : t;
    }
    uint32_t before = __shfl_up(pm, 1);
    if (lane == 0) before = 0;
    before = before > best_inl ? before : best_inl;
    unsigned long long rec = __ballot(inl > before);
    while (rec) {
      const int b = __ffsll((long long)rec) - 1;
      rec &= rec - 1;
      const uint32_t hh = base + b;
      if (hh >= niters) {
        done = 1;
        break;
      }
      best_inl = (uint32_t)__builtin_amdgcn_readlane((int)inl, b);
      best_h = hh;
      if (adaptive) {
        // only min(need, niters) matters: stop counting at the current budget
        const uint32_t need = ransac_needed_iters(best_inl, n, conf, niters);
        niters = need < niters ? need : niters;
      }
    }
  }
  if (!done && h1 >= niters) done = 1;
  if (lane == 0) {
    st.niters = niters;
    st.best_inl = best_inl;
    st.best_h = best_h;
    st.ransac_done = done;
    if (FINAL) {
      if (best_h == 0xFFFFFFFFu) {
        st.ok = 0;
      } else {
        uint32_t min_inl = (uint32_t)(min_inlier_ratio * (float)n);
        if (min_inl < 3) min_inl = 3;
        st.ok = best_inl >= min_inl;
        for (int i = 0; i < 12; ++i) st.bestRt[i] = Rt[((size_t)cand * n_hyp + best_h) * 12 + i];
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------
// K6a.  fp64 raw moments over pairs, by sorted slot.  MODE 0 (ICP step of the exhaustive search; the
// culled search accumulates in its own epilogue): source moved by Tf on the fly, targets via corr;
// optional gate d2 < gate2 on the pair distance.  MODE 1 (RANSAC refit): pre-gathered pairs, gate =
// inlier of states[job].bestRt.  Each work-group writes ACC_NV partial sums
// (n, sp[3], sq[3], spq[9], sum_d2_all, spp = sum |p|^2) -- reduced in a fixed order by solve_kernel.
constexpr int ACC_THREADS = 256;
constexpr int ACC_PER_BLOCK = 2048;
constexpr int ACC_NV = 18;

template <int MODE>
__global__ __launch_bounds__(ACC_THREADS) void accum_kernel(
    const Job* __restrict__ jobs, const CandState* __restrict__ states, const uint32_t* __restrict__ corr,
    const float* __restrict__ d2in, const f32x4* __restrict__ pairs, size_t ld, float gate2,
    double* __restrict__ partials /* [job][n_part][ACC_NV] */, uint32_t n_part) {
  __shared__ double red[ACC_THREADS / 64][ACC_NV];
  const int cand = blockIdx.y;
  const Job& J = jobs[cand];
  const uint32_t n_src = J.n_src;
  const uint32_t b0 = blockIdx.x * ACC_PER_BLOCK;
  if (b0 >= n_src) return;  // uniform over the work-group
  float T[12];
#pragma unroll
  for (int k = 0; k < 12; ++k) T[k] = (MODE == 0) ? states[cand].Tf[k] : states[cand].bestRt[k];
  double v[ACC_NV];
#pragma unroll
  for (int k = 0; k < ACC_NV; ++k) v[k] = 0.0;
  for (uint32_t i = b0 + threadIdx.x; i < b0 + ACC_PER_BLOCK && i < n_src; i += ACC_THREADS) {
    float px, py, pz, qx, qy, qz;
    bool use;
    if (MODE == 0) {
      const f32x4 p = J.src_pts[i];
      xform(T, p.x, p.y, p.z, px, py, pz);
      const uint32_t j = corr[(size_t)cand * ld + i];
      if (j >= J.tgt.n) continue;  // no correspondence (empty target scan)
      const f32x4 t = J.tgt.pts[j];
      qx = t.x; qy = t.y; qz = t.z;
      const float d2 = d2in[(size_t)cand * ld + i];
      v[16] += (double)d2;
      use = !(gate2 > 0.f) || (d2 < gate2);
    } else {
      const f32x4 p = pairs[((size_t)cand * ld + i) * 2 + 0];
      const f32x4 q = pairs[((size_t)cand * ld + i) * 2 + 1];
      px = p.x; py = p.y; pz = p.z;
      qx = q.x; qy = q.y; qz = q.z;
      if (qx >= 0.5f * NN_FAR) continue;  // no correspondence
      v[16] += (double)dist2(px, py, pz, qx, qy, qz);  // = the 1-NN pass's d2, bit for bit
      float x, y, z;
      xform(T, px, py, pz, x, y, z);
      use = dist2(x, y, z, qx, qy, qz) < gate2;
    }
    if (use) {
      const double P[3] = {(double)px, (double)py, (double)pz};
      const double Q[3] = {(double)qx, (double)qy, (double)qz};
      v[0] += 1.0;
#pragma unroll
      for (int a = 0; a < 3; ++a) {
        v[1 + a] += P[a];
        v[4 + a] += Q[a];
#pragma unroll
        for (int b = 0; b < 3; ++b) v[7 + 3 * a + b] += P[a] * Q[b];
      }
      v[17] += (P[0] * P[0] + P[1] * P[1]) + P[2] * P[2];
    }
  }
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
#pragma unroll
  for (int k = 0; k < ACC_NV; ++k) {
    double x = v[k];
    for (int o = 32; o > 0; o >>= 1) x += __shfl_xor(x, o);
    if (lane == 0) red[w][k] = x;
  }
  __syncthreads();
  if (threadIdx.x < ACC_NV) {
    double s = 0.0;
    for (int ww = 0; ww < ACC_THREADS / 64; ++ww) s += red[ww][threadIdx.x];
    partials[((size_t)cand * n_part + blockIdx.x) * ACC_NV + threadIdx.x] = s;
  }
}

// K6b.  One work-group per job: the job's partials (one per source group when the culled search
// accumulated them, PER_GROUP, else one per ACC_PER_BLOCK slots) are summed in a fixed order --
// thread (k, r) takes partials r, r + 56, ... of moment k, thread k then the 56 sub-sums in order --
// thread 0 solves Kabsch and composes.
// MODE 0 (ICP step): T <- dT * T.   MODE 1 (RANSAC refit): T <- T_r * T0, falling back to the
// un-refitted best hypothesis when fewer than 3 inliers, or to T0 when no hypothesis was valid.
constexpr int WAVE_PARTIAL_FLOATS = 21;  // the culled search's partial: 17 fp32 moments about the wave's centre, the centre, sum |p'|^2
constexpr int SOLVE_R = 56;  // (56 x 18 moments = 1008 threads)
constexpr int SOLVE_THREADS = 1024;
static_assert(SOLVE_R * ACC_NV <= SOLVE_THREADS, "a thread per (stride class, moment)");

// memory that every XCD sees without a cache write-back or invalidate: relaxed device-scope atomics (loads and stores
// that go past the XCD's own L2) -- what the chained passes (NnChain, nn_compact.hpp) hand from one wave to another
// INSIDE a launch; COH = false: ordinary loads and stores (data crosses a launch boundary)
template <bool COH>
__device__ __forceinline__ uint32_t ld_u32(const uint32_t* p) {
  if constexpr (COH) return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  else return *p;
}
template <bool COH>
__device__ __forceinline__ void st_u32(uint32_t* p, uint32_t v) {
  if constexpr (COH) __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  else *p = v;
}
template <bool COH>
__device__ __forceinline__ unsigned long long ld_u64(const unsigned long long* p) {
  if constexpr (COH) return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  else return *p;
}
template <bool COH>
__device__ __forceinline__ void st_u64(unsigned long long* p, unsigned long long v) {
  if constexpr (COH) __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  else *p = v;
}
template <bool COH>
__device__ __forceinline__ double ld_f64(const double* p) {
  return __builtin_bit_cast(double, ld_u64<COH>(reinterpret_cast<const unsigned long long*>(p)));
}
template <bool COH>
__device__ __forceinline__ void st_f64(double* p, double v) {
  st_u64<COH>(reinterpret_cast<unsigned long long*>(p), __builtin_bit_cast(unsigned long long, v));
}
template <bool COH>
__device__ __forceinline__ float ld_f32(const float* p) {
  return __uint_as_float(ld_u32<COH>(reinterpret_cast<const uint32_t*>(p)));
}
template <bool COH>
__device__ __forceinline__ void st_f32(float* p, float v) {
  st_u32<COH>(reinterpret_cast<uint32_t*>(p), __float_as_uint(v));
}

// One wave partial of the culled search (nn_compact.hpp: fp32 moments about the wave's own centre, the centre, sum |p'|^2)
// added to raw fp64 moments: sum p q^T = sum p' q'^T + c sum q'^T + sum p' c^T + n c c^T, every product of two fp32
// values exact in fp64.
__device__ __forceinline__ void add_wave_partial(const float* f /* [WAVE_PARTIAL_FLOATS + 1] */, double* v /* [ACC_NV] */) {
  const double n = (double)f[0];
  const double c[3] = {(double)f[17], (double)f[18], (double)f[19]};
  v[0] += n;
  v[16] += (double)f[16];
  // sum |p|^2 = sum |p'|^2 + 2 c . sum p' + n |c|^2
  v[17] += ((double)f[20] + 2.0 * ((c[0] * (double)f[1] + c[1] * (double)f[2]) + c[2] * (double)f[3])) +
           n * ((c[0] * c[0] + c[1] * c[1]) + c[2] * c[2]);
#pragma unroll
  for (int a = 0; a < 3; ++a) {
    v[1 + a] += (double)f[1 + a] + n * c[a];
    v[4 + a] += (double)f[4 + a] + n * c[a];
#pragma unroll
    for (int b = 0; b < 3; ++b)
      v[7 + 3 * a + b] += (((double)f[7 + 3 * a + b] + c[a] * (double)f[4 + b]) + (double)f[1 + a] * c[b]) + n * (c[a] * c[b]);
  }
}

// Kabsch from the raw moments and T <- dT * T (MODE 0: an ICP step) or T <- T_r * T0 (MODE 1: the RANSAC refit), by ONE
// thread.  COH: the state is read and written past the caches (the chained passes: a wave on another XCD reads it in the
// same launch).
template <int MODE, bool COH, bool WS = false>
__device__ __forceinline__ void solve_compose(const double* v /* [ACC_NV] */, CandState* st, const double* Td /* st->Td, or a copy of it */,
                                              uint32_t frozen /* st->frozen */, double* ws = nullptr,
                                              float* T_also = nullptr /* a second place for the new fp32 pose (the chained launch's next pass) ... */,
                                              const float* T_was = nullptr /* ... and where the pose it was given is, should it stay */) {
  GLOC_WS_ARR(Rd, 9);
  GLOC_WS_ARR(td, 3);
  bool have = false;
  if (MODE == 0) {
    if (frozen) {
      if (T_also)
        for (int i = 0; i < 12; ++i) st_f32<COH>(T_also + i, T_was[i]);
      return;
    }
    st_f64<COH>(&st->sum_d2, v[16]);
    if (v[0] < 3.0) {
      st_u32<COH>(reinterpret_cast<uint32_t*>(&st->frozen), 1u);
      st_f32<COH>(&st->last_step, __builtin_inff());  // an ICP that stopped for want of correspondences has not converged (max_final_step)
      if (T_also)
        for (int i = 0; i < 12; ++i) st_f32<COH>(T_also + i, T_was[i]);
      return;
    }
  } else {
    st_f64<COH>(&st->sum_d2, v[16]);
    if (st->best_h == 0xFFFFFFFFu) return;  // keep T0
    if (v[0] < 3.0) {
      for (int i = 0; i < 9; ++i) Rd[i] = (double)st->bestRt[i];
      for (int i = 0; i < 3; ++i) td[i] = (double)st->bestRt[9 + i];
      have = true;
    }
  }
  if (!have) {
    const double cnt = v[0];
    const double inv = 1.0 / cnt;
    GLOC_WS_ARR(pbar, 3);
    GLOC_WS_ARR(qbar, 3);
    GLOC_WS_ARR(M, 9);
    for (int a = 0; a < 3; ++a) {
      pbar[a] = v[1 + a] * inv;
      qbar[a] = v[4 + a] * inv;
    }
    for (int a = 0; a < 3; ++a)
      for (int b = 0; b < 3; ++b) M[3 * a + b] = v[7 + 3 * a + b] - cnt * (pbar[a] * qbar[b]);
    kabsch_from_cov_t<WS>(M, pbar, qbar, Rd, td, ws);
    if (MODE == 0) {
      // how far this update moves the points it was fitted on, RMS: |R c + t - c|^2 + (|R - I|_F^2 / 2) tr cov
      // (oracle/reg_oracle.c: kabsch_pairs) -- the ICP's convergence measure, read by the host after the last pass
      double s2 = v[17] * inv - ((pbar[0] * pbar[0] + pbar[1] * pbar[1]) + pbar[2] * pbar[2]), dc2 = 0.0, f2 = 0.0;
      if (s2 < 0.0) s2 = 0.0;
      for (int a = 0; a < 3; ++a) {
        const double d = (((Rd[3 * a + 0] * pbar[0] + Rd[3 * a + 1] * pbar[1]) + Rd[3 * a + 2] * pbar[2]) + td[a]) - pbar[a];
        dc2 += d * d;
        for (int b = 0; b < 3; ++b) {
          const double e = Rd[3 * a + b] - (a == b ? 1.0 : 0.0);
          f2 += e * e;
        }
      }
      st_f32<COH>(&st->last_step, (float)sqrt(dc2 + 0.5 * f2 * s2));
    }
  }
  // (Rd,td) o (Td)
  GLOC_WS_ARR(Rn, 9);
  GLOC_WS_ARR(tn, 3);
  for (int i = 0; i < 3; ++i) {
    for (int j = 0; j < 3; ++j)
      Rn[3 * i + j] = (Rd[3 * i + 0] * Td[0 + j] + Rd[3 * i + 1] * Td[3 + j]) +
                      Rd[3 * i + 2] * Td[6 + j];
    tn[i] = ((Rd[3 * i + 0] * Td[9] + Rd[3 * i + 1] * Td[10]) + Rd[3 * i + 2] * Td[11]) +
            td[i];
  }
  for (int i = 0; i < 9; ++i) {
    st_f64<COH>(&st->Td[i], Rn[i]);
    st_f32<COH>(&st->Tf[i], (float)Rn[i]);
    if (T_also) st_f32<COH>(T_also + i, (float)Rn[i]);
  }
  for (int i = 0; i < 3; ++i) {
    st_f64<COH>(&st->Td[9 + i], tn[i]);
    st_f32<COH>(&st->Tf[9 + i], (float)tn[i]);
    if (T_also) st_f32<COH>(T_also + 9 + i, (float)tn[i]);
  }
}

#undef GLOC_WS_ARR

template <int MODE>
__global__ __launch_bounds__(SOLVE_THREADS) void solve_kernel(const double* __restrict__ partials,
                                                              uint32_t n_part, bool per_group,
                                                              const Job* __restrict__ jobs,
                                                              CandState* __restrict__ states, NnSplit sp, uint32_t n_jobs) {
  __shared__ double sub[SOLVE_R][ACC_NV];
  __shared__ double tot[ACC_NV];
  const int tid = threadIdx.x;
  // The plan of the NEXT culled 1-NN pass (nn_compact.hpp, NnSplit): groups whose work estimate of the pass just done
  // exceeds the threshold get 2, 4 or 8 waves, in launch order (widest group first) until the helper slots run out.
  // A plan moves time, never a result.  The estimates are consumed.  Made by a work-group of its own -- the grid is
  // 2 x jobs with a plan, the second half plans -- beside the job's solve, not in front of it (in front it added 4 us to
  // the 14 of a kernel that sits between every two passes of one query alone).
  if (sp.hx && blockIdx.x >= n_jobs) {
    const int cand = (int)(blockIdx.x - n_jobs);
    // Slots: first the groups that get 4 or 8 waves, then those that get 2, each class in launch order (the widest group
    // first) -- the prefix sums from ballots (a group's need is 0, 2, 4 or 8), one barrier between counting and placing.
    constexpr int PLAN_TILES = 8;  // 8192 groups: a million-point scan
    __shared__ uint32_t cnt_h[PLAN_TILES][SOLVE_THREADS / 64], cnt_l[PLAN_TILES][SOLVE_THREADS / 64], grp_h[PLAN_TILES][SOLVE_THREADS / 64],
        grp_l[PLAN_TILES][SOLVE_THREADS / 64];
    const uint32_t ng = jobs[cand].n_groups < (uint32_t)(PLAN_TILES * SOLVE_THREADS) ? jobs[cand].n_groups : (uint32_t)(PLAN_TILES * SOLVE_THREADS);
    uint32_t* work = sp.work + (size_t)cand * n_part;
    uint32_t* plan = sp.plan + (size_t)cand * n_part;
    uint32_t* helper = sp.helper + (size_t)cand * sp.hx;
    const int lane = tid & 63, w = tid >> 6;
    const uint32_t n_tiles = (ng + SOLVE_THREADS - 1) / SOLVE_THREADS;
    uint32_t my_parts[PLAN_TILES], my_pre[PLAN_TILES], my_gpre[PLAN_TILES];
#pragma unroll
    for (int t = 0; t < PLAN_TILES; ++t) {
      my_parts[t] = my_pre[t] = my_gpre[t] = 0;
      if ((uint32_t)t >= n_tiles) continue;  // (uniform)
      const uint32_t g = (uint32_t)t * SOLVE_THREADS + (uint32_t)tid;
      uint32_t parts = 0;
      if (g < ng) {
        parts = nn_parts_for(work[g], sp.thresh);
        work[g] = 0;  // consumed
      }
      const unsigned long long b2 = __ballot(parts == 2), b4 = __ballot(parts == 4), b8 = __ballot(parts == 8);
      const unsigned long long below = (1ull << lane) - 1ull;
      my_parts[t] = parts;
      // need and group ordinal among the lanes before this one, within the lane's own class (heavy: 4 / 8; light: 2)
      my_pre[t] = parts >= 4 ? 4u * (uint32_t)__popcll(b4 & below) + 8u * (uint32_t)__popcll(b8 & below) : 2u * (uint32_t)__popcll(b2 & below);
      my_gpre[t] = parts >= 4 ? (uint32_t)__popcll((b4 | b8) & below) : (uint32_t)__popcll(b2 & below);
      if (lane == 0) {
        cnt_h[t][w] = 4u * (uint32_t)__popcll(b4) + 8u * (uint32_t)__popcll(b8);
        cnt_l[t][w] = 2u * (uint32_t)__popcll(b2);
        grp_h[t][w] = (uint32_t)__popcll(b4 | b8);
        grp_l[t][w] = (uint32_t)__popcll(b2);
      }
    }
    for (uint32_t e = (uint32_t)tid; e < sp.hx; e += SOLVE_THREADS) helper[e] = NN_NO_HELPER;  // (slots nobody takes below)
    __syncthreads();
    uint32_t tot_h = 0, tot_gh = 0;
    for (uint32_t t = 0; t < n_tiles; ++t)
      for (int i = 0; i < SOLVE_THREADS / 64; ++i) {
        tot_h += cnt_h[t][i];
        tot_gh += grp_h[t][i];
      }
    uint32_t run_h = 0, run_l = tot_h, run_gh = 0, run_gl = tot_gh;  // slots / ordinals before the current (tile, wave)
#pragma unroll
    for (int t = 0; t < PLAN_TILES; ++t) {
      if ((uint32_t)t >= n_tiles) continue;
      uint32_t bh = run_h, bl = run_l, bgh = run_gh, bgl = run_gl;
      for (int i = 0; i < SOLVE_THREADS / 64; ++i) {
        if (i < w) {
          bh += cnt_h[t][i];
          bl += cnt_l[t][i];
          bgh += grp_h[t][i];
          bgl += grp_l[t][i];
        }
        run_h += cnt_h[t][i];
        run_l += cnt_l[t][i];
        run_gh += grp_h[t][i];
        run_gl += grp_l[t][i];
      }
      const uint32_t g = (uint32_t)t * SOLVE_THREADS + (uint32_t)tid, parts = my_parts[t];
      if (g >= ng) continue;
      const uint32_t first = (parts >= 4 ? bh : bl) + my_pre[t], hid = (parts >= 4 ? bgh : bgl) + my_gpre[t];
      uint32_t word = 0;
      if (parts > 1 && first + parts <= sp.hx) {  // (groups past the last slot keep their one wave at their own rank)
        for (uint32_t p = 0; p < parts; ++p) helper[first + p] = g | (p << 20) | (parts << 24);
        word = (hid << 8) | parts;
      }
      plan[g] = word;
    }
    for (uint32_t g = ng + (uint32_t)tid; g < jobs[cand].n_groups; g += SOLVE_THREADS) plan[g] = 0;  // (beyond the planner's reach)
    return;
  }
  const int cand = blockIdx.x;
  const uint32_t cnt = per_group ? jobs[cand].n_groups
                                 : (jobs[cand].n_src + ACC_PER_BLOCK - 1) / ACC_PER_BLOCK;
  if (per_group) {
    // The culled search's partials: fp32 moments about each wave's own centre (nn_compact.hpp) -> raw moments in
    // fp64.  sum p q^T = sum p' q'^T + c sum q'^T + sum p' c^T + n c c^T, every product of two fp32 values exact in
    // fp64.  A thread takes whole partials (one each at 124 k points: 969 waves), the work-group sums them in a
    // fixed order: xor butterfly inside the wave, the 16 waves in order.
    static_assert(SOLVE_THREADS / 64 <= SOLVE_R, "a row of sub[][] per wave");
    const float* base = reinterpret_cast<const float*>(partials + (size_t)cand * n_part * ACC_NV);
    double v[ACC_NV];
#pragma unroll
    for (int k = 0; k < ACC_NV; ++k) v[k] = 0.0;
    for (uint32_t g = (uint32_t)tid; g < cnt; g += SOLVE_THREADS) {
      typedef float f32x2_ __attribute__((ext_vector_type(2)));
      const f32x2_* f2 = reinterpret_cast<const f32x2_*>(base + (size_t)g * (2 * ACC_NV));  // (a slot is 136 B: 8-byte aligned)
      float f[WAVE_PARTIAL_FLOATS + 1];
#pragma unroll
      for (int i = 0; i < (WAVE_PARTIAL_FLOATS + 1) / 2; ++i) {
        const f32x2_ t = f2[i];
        f[2 * i] = t.x;
        f[2 * i + 1] = t.y;
      }
      add_wave_partial(f, v);
    }
#pragma unroll
    for (int k = 0; k < ACC_NV; ++k) {  // (DPP / permlane exchanges: the LDS crossbar's __shfl_xor made this 8 us of chain)
      double x = v[k];
      x += xor_lane<32>(x);
      x += xor_lane<16>(x);
      x += xor_lane<8>(x);
      x += xor_lane<4>(x);
      x += xor_lane<2>(x);
      x += xor_lane<1>(x);
      v[k] = x;
    }
    if ((tid & 63) == 0) {
#pragma unroll
      for (int k = 0; k < ACC_NV; ++k) sub[tid >> 6][k] = v[k];
    }
  } else if (tid < SOLVE_R * ACC_NV) {
    const int k = tid % ACC_NV, r = tid / ACC_NV;
    double acc = 0.0;
    uint32_t b = (uint32_t)r;
    const double* pp = partials + (size_t)cand * n_part * ACC_NV + k;
    for (; b + 3 * SOLVE_R < cnt; b += 4 * SOLVE_R) {  // four independent loads in flight, summed in order
      const double v0 = pp[(size_t)b * ACC_NV], v1 = pp[(size_t)(b + SOLVE_R) * ACC_NV];
      const double v2 = pp[(size_t)(b + 2 * SOLVE_R) * ACC_NV], v3 = pp[(size_t)(b + 3 * SOLVE_R) * ACC_NV];
      acc += v0; acc += v1; acc += v2; acc += v3;
    }
    for (; b < cnt; b += SOLVE_R) acc += pp[(size_t)b * ACC_NV];
    sub[r][k] = acc;
  }
  __syncthreads();
  if (tid < ACC_NV) {
    double acc = 0.0;
    const int rows = per_group ? SOLVE_THREADS / 64 : SOLVE_R;  // (a row per wave / per stride class)
    for (int r = 0; r < rows; ++r) acc += sub[r][tid];
    tot[tid] = acc;
  }
  __syncthreads();
  if (tid != 0) return;
  double v[ACC_NV];
#pragma unroll
  for (int k = 0; k < ACC_NV; ++k) v[k] = tot[k];
  solve_compose<MODE, false>(v, &states[cand], states[cand].Td, (uint32_t)states[cand].frozen);
}

}  // namespace reg
}  // namespace gloc
