// reg.hip -- C ABI of the batched candidate registration (include/gloc3d.h) over reg_kernels.hpp.
// Replaces icp_match_3d (registration/global_registration.cpp:237-248) and the per-candidate RANSAC
// transform estimate (registration/loop_detector.cpp:256-257) of the reference, batched over the
// top-k candidates of GlocEvaluator::global_registraion (registration/global_localization.cpp:511-574)
// and over any number of queries in flight (one launch covers all their candidates).
#include <algorithm>
#include <cmath>
#include <new>
#include <vector>

#include "common.hpp"
#include <future>

#include "nn_compact.hpp"
#include "reg_kernels.hpp"
#include "scan_store.hpp"

using namespace gloc;
using namespace gloc::reg;

struct gloc_reg {
  int device = 0;
  hipStream_t own_stream = nullptr, stream = nullptr;
  hipEvent_t done_ev = nullptr;  // a batch waits for ITS OWN results, not for the stream: handles that share a stream
                                 // (gloc_reg_set_stream) queue their batches back to back while the host reads results
  gloc_scan_store* store = nullptr;  // scans are looked up here (attached, or the handle's own)
  gloc_scan_store* own_store = nullptr;
  DevBuf jobs, states;           // Job[], CandState[]
  DevBuf corr, d2, pairs;        // [job][ld]
  DevBuf Rt, valid, inliers;     // RANSAC hypotheses
  DevBuf alive;                  // all-hypotheses RANSAC: [job][H] list of the hypotheses still in the race + [job] counts
  DevBuf partials;               // [job][n_part][ACC_NV] fp64
  DevBuf export_idx, export_d2;  // gloc_reg_nn: results in the caller's index space
  DevBuf counters;               // pairs evaluated by the culled search: NN_STAT_SLOTS partial counts (profiling only)
  // pinned host staging: job table up, per-job state up and down.  Pageable memory would make the "async" copies
  // wait for the stream, i.e. for the batch BEFORE this one when handles share a stream
  void* pin = nullptr;
  size_t pin_cap = 0;
  CandState* h_states = nullptr;  // [n_jobs], inside pin
  Job* h_jobs = nullptr;          // [n_jobs], inside pin
  // a batch between gloc_reg_batch_multi_begin and _end
  struct Pending {
    bool active = false;
    std::vector<size_t> slot;     // job -> output row
    std::vector<size_t> n_src;    // per job
    std::vector<float> def_T;     // output rows before the jobs' results go in (initial guess / identity)
    size_t total = 0;
    float max_rmse = 0.f, max_final_step = 0.f;
    bool icp = false;
    gloc_scan_store* store = nullptr;  // the store whose scans this batch pins ...
    std::vector<uint32_t> pinned;      // ... and their ids (store_pin)
  } pending;
  std::vector<float> last_final_step;  // per job of the last collected batch (gloc_reg_final_steps)
  int nn_mode = 0;        // 0 culled + compacted (default), 1 exhaustive
  bool trace_on = false;  // dev only: per-wave trace of the culled kernel
  DevBuf trace;
  size_t trace_waves = 0;
  int nn_src_per_lane = 2;  // culled kernel: source points per lane (1, 2, 4)
  int nn_job_group = 24;    // culled kernel: jobs interleaved in the launch order (a multiple of 8: see nn_compact.hpp)
  bool nn_job_group_set = false;  // by the caller (else a small batch takes its own: launch_order)
  int nn_sub_jobs = 0;      // culled kernel: interleaved shares of a job's work-groups that get their own slot (0: by batch size)
  bool temp_target_index = false;  // kd-ordered target index for the temporary scans of the host-buffer calls
  // heavy source groups over several waves (nn_compact.hpp, NnSplit): helper waves per job (-1: by batch size, 0: off)
  // and the work estimate (cycles) above which a group is split
  int nn_split_helpers = -1;
  uint32_t nn_split_thresh = 60000;
  bool nn_split_thresh_set = false;  // by the caller (GLOC_REG_OPT_NN_SPLIT_THRESH): else 60000, or 85000 where the passes are chained
  DevBuf split_zero, split_ff;     // [work | plan | ticket] and [skey | helper] of the batch
  NnSplit split{};                 // views into them for the batch being enqueued (hx = 0: off)
  // the groups a cold pass's waves give up and a second launch searches with NN_HEAVY_PARTS waves each (NnHeavy)
  int nn_heavy_thresh = 32;        // processed chunks at which a cold wave gives up (0: off)
  static constexpr uint32_t MAX_SUB = 8;
  DevBuf heavy_buf;                // per sub-batch a region [count | list | ticket | skey | hkey]
  size_t heavy_cap = 0;            // entries a region holds (its ticket / skey parts are self-resetting)
  size_t heavy_regions = 0;        // regions the buffer holds
  NnHeavy heavy_of[MAX_SUB]{};     // the views for the batch being enqueued, one per sub-batch (cap = 0: off)
  // sub-batches of a small batch, each on its own stream (enqueue_jobs): -1 by batch size (4 for 8 .. 64 jobs), 0 / 1 off
  int sub_batches = -1;
  hipStream_t sub_stream[MAX_SUB - 1] = {};
  hipEvent_t fork_ev = nullptr, join_ev[MAX_SUB - 1] = {};
  std::atomic<uint64_t> nn_launches{0};
  // the warm passes of a small batch chained in one launch (NnChain): 1 on (default), 0 off; stopped for good on a handle
  // whose chain once ran out of time
  int nn_chain = 1;
  bool chain_broken = false;
  bool chain_trace = false;       // dev (gloc_reg_debug_chain_trace): stamps per (pass, job)
  DevBuf chain_dbg;
  uint32_t chain_dbg_pass = 0, chain_dbg_jobs = 0;
  bool chain_stall = false;       // test aid (gloc_reg_debug_chain_stall): the solvers wait for one wave more than there is
  DevBuf chain_buf;               // [ready | done | sdone | err] then the reducers' sub-sums
  uint32_t* h_chain_err = nullptr; // pinned: the batch's err word, copied behind the states
  bool chain_in_batch = false;    // the batch enqueued last ran a chained launch
  // ... and what it takes to run that batch again launch by launch should the chain's waits run out (collect_jobs):
  // the jobs as enqueue_jobs saw them (JobHost, kept as bytes: the type is this file's) with their initial poses, the parameters
  std::vector<unsigned char> retry_jobs;
  std::vector<float> retry_T;
  gloc_reg_params retry_prm{};
  std::atomic<uint64_t> chain_launches{0}, chain_timeouts{0};
  size_t last_ld = 0;      // shape of the last batch (gloc_reg_debug_corr)
  uint32_t last_jobs = 0;
  Profiler prof;
};

namespace {

struct JobHost {
  DevScan src, tgt;
  uint32_t stream_id;
  const float* init_T;  // 16 floats or null
};

void init_state(CandState& st, const float* T16, uint32_t ransac_iters = 0) {
  memset(&st, 0, sizeof(st));
  static const float I16[16] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1};
  const float* T = T16 ? T16 : I16;
  for (int i = 0; i < 3; ++i) {
    for (int j = 0; j < 3; ++j) {
      st.Tf[3 * i + j] = T[4 * i + j];
      st.Td[3 * i + j] = (double)T[4 * i + j];
    }
    st.Tf[9 + i] = T[4 * i + 3];
    st.Td[9 + i] = (double)T[4 * i + 3];
  }
  st.best_h = 0xFFFFFFFFu;
  st.niters = ransac_iters;
  st.last_step = 0.f;
}

struct BatchDims {
  uint32_t n_jobs, max_src, max_groups, n_part;
  size_t ld;
};

// A batch -- or, round 6, one of the sub-batches a small batch is cut into, each on its own stream -- as the launches see
// it: its jobs' slices of the handle's workspaces (every per-job array offset by the sub-batch's first job; strides --
// bd.ld, bd.n_part, the hypothesis count -- are the whole batch's).
struct WsView {
  hipStream_t s;
  uint32_t job0, n_jobs;
  Job* jobs;
  CandState* states;
  uint32_t* corr;
  float* d2;
  f32x4* pairs;
  float* Rt;
  uint32_t *valid, *inliers;
  double* partials;
  uint32_t *a_idx, *a_cnt;  // (ransac_alive_kernel's lists; set when they are allocated)
  NnSplit split;
  NnHeavy heavy;
};

// The split plan's buffers for a batch (NnSplit): everything starts as "no group is split"; the first pass of a batch
// therefore runs one wave per group and leaves the estimates the first plan is made from.
int setup_split(gloc_reg* h, const BatchDims& bd, int cs) {
  h->split = NnSplit{};
  if (h->nn_mode == 1 || h->nn_split_helpers == 0 || h->nn_split_thresh == 0) return GLOC_OK;
  // one query alone (20 jobs) is the case that needs it: its launch is as long as its longest wave.  A launch of hundreds
  // of jobs only loses its tail to such waves (5 % at 500 jobs, measured per XCD) and the kernel with the plan in it is
  // 3 % slower: off by default there
  uint32_t hx = h->nn_split_helpers > 0 ? (uint32_t)h->nn_split_helpers : (bd.n_jobs <= 64 ? 256u : (bd.n_jobs <= 256 ? 64u : 0u));
  if (hx == 0) return GLOC_OK;
  hx = std::min<uint32_t>((hx + NN_WPB - 1) / NN_WPB * NN_WPB, 1u << 12);
  const size_t S = 64 * (size_t)cs, nj = bd.n_jobs, np = bd.n_part;
  const size_t zero_words = nj * np * 2 + nj * hx;
  const size_t ff_bytes = nj * hx * S * 8 + nj * hx * 4;
  hipStream_t s = h->stream;
  GLOC_TRY(h->split_zero.ensure(zero_words * 4, s));
  GLOC_TRY(h->split_ff.ensure(ff_bytes, s));
  GLOC_HIP(hipMemsetAsync(h->split_zero.p, 0, zero_words * 4, s));
  GLOC_HIP(hipMemsetAsync(h->split_ff.p, 0xFF, ff_bytes, s));
  NnSplit& sp = h->split;
  sp.work = h->split_zero.as<uint32_t>();
  sp.plan = sp.work + nj * np;
  sp.ticket = sp.plan + nj * np;
  sp.skey = h->split_ff.as<unsigned long long>();
  sp.helper = reinterpret_cast<uint32_t*>(sp.skey + nj * hx * S);
  sp.hx = hx;
  // (the chained launch hides a job's longest wave behind the other jobs' work, so fewer groups need splitting: one query
  // alone, registration of 20 jobs, threshold 45 / 60 / 75 / 90 / 120 thousand cycles: 2.92 / 2.89 / 2.79 / 2.78 / 3.03 ms)
  static const bool chain_off = getenv("GLOC3D_NN_NO_CHAIN") != nullptr;  // developer switch (chain_passes)
  const bool chains = h->nn_chain && !h->chain_broken && !h->prof.enabled && !h->trace_on && cs == 2 && bd.n_jobs < 48 && !chain_off;
  sp.thresh = h->nn_split_thresh_set || !chains ? h->nn_split_thresh : 85000u;
  return GLOC_OK;
}

// The lists of a cold pass's heavy groups (NnHeavy), one per sub-batch: 16 entries per job, at most 16 384; only at two
// sources per lane.
int setup_heavy(gloc_reg* h, const BatchDims& bd, int cs, uint32_t G = 1) {
  for (uint32_t g = 0; g < gloc_reg::MAX_SUB; ++g) h->heavy_of[g] = NnHeavy{};
#ifdef GLOC_NN_R5_COLD
  static const bool off = true;
#else
  static const bool off = getenv("GLOC3D_NN_NO_HEAVY") != nullptr;  // developer switch
#endif
  if (h->nn_mode == 1 || cs != 2 || h->nn_heavy_thresh <= 0 || off || h->trace_on) return GLOC_OK;
  const size_t per = ((size_t)bd.n_jobs + G - 1) / G;
  const size_t cap = std::min<size_t>(per * 16, 16384), S = 64 * (size_t)cs;
  hipStream_t s = h->stream;
  auto region_bytes = [&](size_t c) { return ((256 + c * 8 + c * 4 + 255) & ~(size_t)255) + 2 * c * S * 8; };
  if (cap > h->heavy_cap || G > h->heavy_regions) {
    const size_t cap2 = std::min<size_t>(std::max<size_t>(cap, h->heavy_cap), 16384), R = std::max<size_t>(G, h->heavy_regions);
    GLOC_TRY(h->heavy_buf.ensure(R * region_bytes(cap2) + 256, s));
    for (size_t g = 0; g < R; ++g) {
      char* b = h->heavy_buf.as<char>() + g * region_bytes(cap2);
      const size_t keys_at = (256 + cap2 * 8 + cap2 * 4 + 255) & ~(size_t)255;
      GLOC_HIP(hipMemsetAsync(b, 0, keys_at, s));                      // count, list, tickets
      GLOC_HIP(hipMemsetAsync(b + keys_at, 0xFF, cap2 * S * 8, s));     // fold keys (hkey behind them: written before read)
    }
    h->heavy_cap = cap2;
    h->heavy_regions = R;
  }
  for (uint32_t g = 0; g < G; ++g) {
    char* b = h->heavy_buf.as<char>() + g * region_bytes(h->heavy_cap);
    const size_t head = 256 + h->heavy_cap * 8, keys_at = (head + h->heavy_cap * 4 + 255) & ~(size_t)255;
    NnHeavy& hv = h->heavy_of[g];
    hv.count = reinterpret_cast<uint32_t*>(b);
    hv.list = reinterpret_cast<uint32_t*>(b + 256);
    hv.ticket = reinterpret_cast<uint32_t*>(b + head);
    hv.skey = reinterpret_cast<unsigned long long*>(b + keys_at);
    hv.hkey = hv.skey + h->heavy_cap * S;
    hv.cap = (uint32_t)cap;
    hv.thresh = (uint32_t)h->nn_heavy_thresh;
  }
  return GLOC_OK;
}

int ensure_sub_streams(gloc_reg* h, uint32_t G) {
  if (!h->fork_ev) GLOC_HIP(hipEventCreateWithFlags(&h->fork_ev, hipEventDisableTiming));
  for (uint32_t g = 1; g < G; ++g) {
    if (!h->sub_stream[g - 1]) GLOC_HIP(hipStreamCreateWithFlags(&h->sub_stream[g - 1], hipStreamNonBlocking));
    if (!h->join_ev[g - 1]) GLOC_HIP(hipEventCreateWithFlags(&h->join_ev[g - 1], hipEventDisableTiming));
  }
  return GLOC_OK;
}

// The launch order of a batch's 1-NN passes (nn_compact.hpp): slots per group and interleaved shares of a job.  A batch of
// 48 jobs or more: groups of 24 slots, a job per slot (the headline's launches: tuned in rounds 2-6).  A SMALL batch left
// alone (no GLOC_REG_OPT_NN_JOB_GROUP / NN_SUB_JOBS): rows of 8 slots -- one per XCD -- and as few shares of a job as
// spread the batch evenly over the 8 XCDs (20 jobs: 2 shares, 40 slots, five rows; an odd number of jobs: 8): an XCD then
// works on ONE job's share at a time and its L2 holds that job's target instead of three.  Round 5 chose 24 slots x 8 shares on
// rigid copies; on the distinct casts, one query alone, registration of 20 jobs, slots x shares 24 x 8 / 24 x 4 / 24 x 2 /
// 8 x 4 / 8 x 2: launch by launch 3.24 / 3.21 / 3.10 / 3.21 / 3.08 ms, chained 2.69 / 2.66 / 2.72 / 2.68 / 2.62
// (profiles/r06_chain_split_sweep.txt).
void launch_order(const gloc_reg* h, uint32_t n_jobs, uint32_t& jg, uint32_t& subs) {
  subs = h->nn_sub_jobs > 0 ? (uint32_t)h->nn_sub_jobs : (n_jobs < 48 ? 8u : 1u);
  jg = (uint32_t)h->nn_job_group;
  if (h->nn_sub_jobs > 0 || h->nn_job_group_set || n_jobs >= 48) return;
  jg = 8u;
  subs = n_jobs % 4u == 0u ? 2u : (n_jobs % 2u == 0u ? 4u : 8u);
}

// S1 for every job of the batch.  warm: corr holds the previous pass's result.  want_pairs: write the (moved
// source, matched target) pairs INSTEAD of the moments (the RANSAC stage refits from the pairs: accum_kernel<1>).  The culled search leaves the wave partials of the
// fp64 moments in h->partials (per source group); the exhaustive one needs accum_kernel<0> afterwards.
int launch_nn(gloc_reg* h, const BatchDims& bd, const WsView& v, bool warm, bool want_pairs, float gate2) {
  ProfScope ps(h->prof, "nn", v.s);
  // (the first pass of a batch -- no previous correspondence, or the pass that writes the pairs -- is another
  // instantiation and ~1.6 x a warm pass: also counted in its own family, so that "nn" - "nn_cold" is the warm passes alone)
  Profiler no_prof;
  ProfScope ps_cold((want_pairs || !warm) ? h->prof : no_prof, "nn_cold", v.s);
  h->nn_launches++;
  if (h->nn_mode == 1) {
    dim3 grid((bd.max_src + 256 * NN_S - 1) / (256 * NN_S), v.n_jobs);
    hipLaunchKernelGGL(nn_kernel, grid, dim3(256), 0, v.s, v.jobs, v.states,
                       v.corr, v.d2, bd.ld);
    if (want_pairs)
      hipLaunchKernelGGL(gather_pairs_kernel, dim3((bd.max_src + 255) / 256, v.n_jobs), dim3(256), 0, v.s,
                         v.jobs, v.states, v.corr, bd.ld,
                         v.pairs);
  } else {
    const int cs = h->nn_src_per_lane;
    const bool cold = want_pairs || !warm;
    const NnHeavy hv = cold ? v.heavy : NnHeavy{};
    if (hv.cap) GLOC_HIP(hipMemsetAsync(hv.count, 0, 4, v.s));
    const uint32_t n_wg_job = (bd.max_groups + v.split.hx + NN_WPB - 1) / NN_WPB;  // helper waves first, then one per group
    // slots of the launch order (nn_compact.hpp): a job each, or -- few jobs -- `subs` interleaved shares of a job, so
    // that the 8 XCDs get equal numbers of slots
    // (8 shares -- one per XCD -- since round 5: one query alone 105.4 -> 104.1 / 104.4 us per pass against 4)
    uint32_t jg, subs;
    launch_order(h, v.n_jobs, jg, subs);
    // (one group of all the slots of a small batch, so that every job's helpers start at the head of the launch, was
    // tried: one query alone 0.115 ms per pass against 0.108 with groups of 24)
    const uint32_t n_slots = v.n_jobs * subs;
    const uint32_t n_wg = (n_wg_job + subs - 1) / subs;
    const unsigned grid = n_wg * jg * ((n_slots + jg - 1) / jg);
    // (launched as (slots of a group, work-groups of a slot, groups): the same linear order without the divisions)
    GLOC_REQUIRE(n_wg <= 65535u && (n_slots + jg - 1) / jg <= 65535u, GLOC_ERR_INVALID,
                 "the culled search's grid: %u work-groups per job slot (scans above ~8 M points) or %u job groups exceed 65535", n_wg,
                 (n_slots + jg - 1) / jg);
    if (h->trace_on) {
      h->trace_waves = (size_t)grid * NN_WPB;
      if (h->trace.ensure(h->trace_waves * 4 * NN_TRACE_WORDS, v.s)) return GLOC_ERR_NOMEM;
      GLOC_HIP(hipMemsetAsync(h->trace.p, 0, h->trace_waves * 4 * NN_TRACE_WORDS, v.s));
    }
// instantiations: sources per lane x (pairs | moments) x (first pass: cold start | later: warm) x (with the split plan);
// the per-wave trace only at two sources per lane
#define LAUNCH_COMPACT(CS_, P_, W_)                                                                      \
  do {                                                                                                  \
    if (h->trace_on && (CS_) == 2) {                                                                    \
      if (v.split.hx) LAUNCH_COMPACT_T(2, P_, true, true, W_);                                         \
      else LAUNCH_COMPACT_T(2, P_, true, false, W_);                                                    \
    } else if (v.split.hx) {                                                                           \
      if (!(P_) && (W_) && (CS_) == 2) LAUNCH_COMPACT_K((nn_compact_split_warm_kernel<2>), P_);         \
      else LAUNCH_COMPACT_T(CS_, P_, false, true, W_);                                                  \
    } else {                                                                                            \
      LAUNCH_COMPACT_T(CS_, P_, false, false, W_);                                                      \
    }                                                                                                   \
  } while (0)
#define LAUNCH_COMPACT_T(CS_, P_, T_, S_, W_) LAUNCH_COMPACT_K((nn_compact_kernel<CS_, P_, T_, S_, W_>), P_)
#define LAUNCH_COMPACT_K(K_, P_)                                                                         \
  hipLaunchKernelGGL(K_, dim3(jg, n_wg, (n_slots + jg - 1) / jg), dim3(64 * NN_WPB), 0, v.s, v.jobs, \
                     v.n_jobs, jg, n_wg, subs, v.states,                               \
                     warm ? v.corr : (const uint32_t*)nullptr, v.corr,   \
                     v.d2, v.pairs, (P_) ? (double*)nullptr : v.partials, bd.n_part, bd.ld, \
                     gate2, v.split, hv,                                                                \
                     h->prof.enabled ? h->counters.as<unsigned long long>() : (unsigned long long*)nullptr, \
                     h->trace_on ? h->trace.as<uint32_t>() : (uint32_t*)nullptr)
#define LAUNCH_COMPACT_CS(P_, W_)                                                                        \
  do {                                                                                                  \
    if (cs == 1) LAUNCH_COMPACT(1, P_, W_);                                                             \
    else if (cs == 2) LAUNCH_COMPACT(2, P_, W_);                                                        \
    else LAUNCH_COMPACT(4, P_, W_);                                                                     \
  } while (0)
    if (grid) {
      if (want_pairs) LAUNCH_COMPACT_CS(true, false);  // (the pass that writes the pairs is a batch's first)
      else if (warm) LAUNCH_COMPACT_CS(false, true);
      else LAUNCH_COMPACT_CS(false, false);
      if (hv.cap) {  // the groups the cold pass's waves gave up: NN_HEAVY_PARTS waves each (the list's length stays on the device)
#define LAUNCH_HEAVY(P_)                                                                                              \
  hipLaunchKernelGGL((nn_compact_heavy_kernel<2, P_>), dim3(hv.cap * NN_HEAVY_PARTS), dim3(64), 0, v.s, v.jobs, \
                     v.n_jobs, jg, n_wg, subs, v.states, (const uint32_t*)nullptr, v.corr,       \
                     v.d2, v.pairs, (P_) ? (double*)nullptr : v.partials, bd.n_part, bd.ld, \
                     gate2, v.split, hv,                                                                               \
                     h->prof.enabled ? h->counters.as<unsigned long long>() : (unsigned long long*)nullptr, (uint32_t*)nullptr)
        if (want_pairs) LAUNCH_HEAVY(true);
        else LAUNCH_HEAVY(false);
#undef LAUNCH_HEAVY
      }
    }
#undef LAUNCH_COMPACT_CS
#undef LAUNCH_COMPACT
#undef LAUNCH_COMPACT_T
#undef LAUNCH_COMPACT_K
  }
  GLOC_HIP(hipGetLastError());
  return GLOC_OK;
}

// The remaining warm moments passes of a small batch -- search, reduce, solve, plan, `n_pass` times -- in ONE launch
// (NnChain, reg_kernels.hpp; nn_chain_kernel, nn_compact.hpp).  0 passes: the batch does not qualify (the caller goes on
// launch by launch).
uint32_t chain_passes(const gloc_reg* h, const BatchDims& bd, const WsView& v, uint32_t remaining) {
  static const bool off = getenv("GLOC3D_NN_NO_CHAIN") != nullptr;  // developer switch
  static const bool force = getenv("GLOC3D_NN_CHAIN_FORCE") != nullptr;  // developer switch: any batch size, with or without the plan, under the per-kernel events
  if (off || !h->nn_chain || h->chain_broken || h->nn_mode == 1 || h->nn_src_per_lane != 2 || h->trace_on || remaining < 2 || NN_WPB != 1) return 0;
  if (!force && (!v.split.hx || h->prof.enabled)) return 0;
  uint32_t jg, subs;
  launch_order(h, v.n_jobs, jg, subs);
  if ((!force && v.n_jobs >= 48) || (jg & 7u) || jg % subs) return 0;  // (small batches; a group of slots holds whole jobs)
  if ((bd.n_part & 31u) || (v.split.hx & 31u)) return 0;  // (a job's rows of the per-pass tables are whole cache lines)
  const uint32_t n_wg = ((bd.max_groups + v.split.hx + NN_WPB - 1) / NN_WPB + subs - 1) / subs;
  const uint32_t groups = (v.n_jobs * subs + jg - 1) / jg;
  const uint64_t grp_size = ((uint64_t)jg * n_wg + (jg / subs) * NN_CHAIN_ROLES + 7) & ~7ull;
  if (grp_size * groups * remaining >= (1ull << 31)) return 0;
  return remaining;
}

int launch_nn_chain(gloc_reg* h, const BatchDims& bd, const WsView& v, uint32_t n_pass, float gate2) {
  uint32_t jg, subs;
  launch_order(h, v.n_jobs, jg, subs);
  const uint32_t n_wg = ((bd.max_groups + v.split.hx + NN_WPB - 1) / NN_WPB + subs - 1) / subs;
  const uint32_t groups = (v.n_jobs * subs + jg - 1) / jg;
  NnChain ch{};
  ch.n_pass = n_pass;
  ch.expected = subs * n_wg * NN_WPB + (h->chain_stall ? 1u : 0u);
  ch.jobs_per_grp = jg / subs;
  ch.grp_size = (jg * n_wg + ch.jobs_per_grp * NN_CHAIN_ROLES + 7u) & ~7u;
  ch.pass_size = ch.grp_size * groups;
  // [ready | done | go | sdone] a 256-byte line per (pass, job), the err word's line; then (never cleared: written before
  // read) the per-pass poses, plans and helpers' tables, the reducers' sub-sums
  const size_t cells = (size_t)n_pass * v.n_jobs, line = NN_CHAIN_PAD * 4;
  const size_t head = (4 * cells + 1) * line;
  const size_t t_bytes = cells * NN_CHAIN_T_STRIDE * 4;
  const size_t plan_bytes = (size_t)(n_pass - 1) * v.n_jobs * bd.n_part * 4, help_bytes = (size_t)(n_pass - 1) * v.n_jobs * v.split.hx * 4;
  const size_t sub_bytes = sizeof(double) * ACC_NV * NN_CHAIN_RED * v.n_jobs;
  GLOC_TRY(h->chain_buf.ensure(head + t_bytes + plan_bytes + help_bytes + sub_bytes + 256, v.s));
  GLOC_HIP(hipMemsetAsync(h->chain_buf.p, 0, head, v.s));
  char* base = h->chain_buf.as<char>();
  ch.ready = reinterpret_cast<uint32_t*>(base);
  ch.done = reinterpret_cast<uint32_t*>(base + cells * line);
  ch.go = reinterpret_cast<uint32_t*>(base + 2 * cells * line);
  ch.sdone = reinterpret_cast<uint32_t*>(base + 3 * cells * line);
  ch.err = reinterpret_cast<uint32_t*>(base + 4 * cells * line);
  ch.Tp = reinterpret_cast<float*>(base + head);
  ch.planp = reinterpret_cast<uint32_t*>(base + head + t_bytes);
  ch.helperp = reinterpret_cast<uint32_t*>(base + head + t_bytes + plan_bytes);
  ch.sub = reinterpret_cast<double*>(base + head + t_bytes + plan_bytes + help_bytes);
  if (h->chain_trace) {
    const size_t nb = (size_t)n_pass * v.n_jobs * 16 * 4;
    GLOC_TRY(h->chain_dbg.ensure(nb, v.s));
    GLOC_HIP(hipMemsetAsync(h->chain_dbg.p, 0, nb, v.s));
    for (uint32_t q = 0; q < n_pass * v.n_jobs; ++q)  // (slots 0 and 1 take a minimum)
      GLOC_HIP(hipMemsetAsync(h->chain_dbg.as<uint32_t>() + (size_t)q * 16, 0xFF, 8, v.s));
    ch.dbg = h->chain_dbg.as<uint32_t>();
    h->chain_dbg_pass = n_pass;
    h->chain_dbg_jobs = v.n_jobs;
  }
  h->nn_launches += n_pass;
  h->chain_launches++;
  h->chain_in_batch = true;
  {
    ProfScope ps(h->prof, "nn", v.s);  // (only with GLOC3D_NN_CHAIN_FORCE: the events otherwise switch the chain off)
    if (v.split.hx)
      hipLaunchKernelGGL((nn_chain_kernel<2, true>), dim3(ch.pass_size * n_pass), dim3(64 * NN_WPB), 0, v.s, v.jobs, v.n_jobs, jg, n_wg, subs,
                         v.states, v.corr, v.corr, v.d2, v.pairs, v.partials, bd.n_part, bd.ld, gate2, v.split, NnHeavy{},
                         (unsigned long long*)nullptr, (uint32_t*)nullptr, ch);
    else
      hipLaunchKernelGGL((nn_chain_kernel<2, false>), dim3(ch.pass_size * n_pass), dim3(64 * NN_WPB), 0, v.s, v.jobs, v.n_jobs, jg, n_wg, subs,
                         v.states, v.corr, v.corr, v.d2, v.pairs, v.partials, bd.n_part, bd.ld, gate2, v.split, NnHeavy{},
                         (unsigned long long*)nullptr, (uint32_t*)nullptr, ch);
  }
  GLOC_HIP(hipGetLastError());
  GLOC_HIP(hipMemcpyAsync(h->h_chain_err, ch.err, 4, hipMemcpyDeviceToHost, v.s));
  return GLOC_OK;
}

int ensure_pinned(gloc_reg* h, uint32_t n_jobs) {
  const size_t need = (sizeof(CandState) + sizeof(Job)) * (size_t)n_jobs + 64;
  if (need > h->pin_cap) {
    if (h->pin) (void)hipHostFree(h->pin);
    h->pin = nullptr;
    h->pin_cap = 0;
    const size_t cap = need + need / 2 + 4096;
    GLOC_HIP(hipHostMalloc(&h->pin, cap, hipHostMallocDefault));
    h->pin_cap = cap;
  }
  h->h_states = reinterpret_cast<CandState*>(h->pin);
  h->h_jobs = reinterpret_cast<Job*>(h->h_states + n_jobs);
  h->h_chain_err = reinterpret_cast<uint32_t*>(h->h_jobs + n_jobs);
  return GLOC_OK;
}

// The launches of a batch -- or of one sub-batch on its own stream: S1 -> S2 (RANSAC + refit) -> S3 (ICP) over the view's jobs.
int enqueue_pipeline(gloc_reg* h, const BatchDims& bd, const gloc_reg_params* prm, const WsView& v, bool can, bool any_tgt,
                     uint32_t nblocks) {
  const uint32_t n_jobs = v.n_jobs;
  hipStream_t s = v.s;
  const bool culled = h->nn_mode != 1;
  const float gate2 = prm->max_corr_dist > 0.f ? prm->max_corr_dist * prm->max_corr_dist : 0.f;
  bool have_corr = false;  // corr holds a previous pass's result: warm start for the next one

  if (can && any_tgt && prm->ransac_iters > 0) {
    const uint32_t H = prm->ransac_iters;
    GLOC_TRY(launch_nn(h, bd, v, false, true, 0.f));
    have_corr = true;
    // Phases of hypotheses, each generated, scored and scanned before the next: with the adaptive stop (the
    // reference's call: confidence 0.99) and ~85 % inliers the iteration count drops to 5 - 8 at the first good
    // hypothesis, so [0, 16) settles nearly every job, [16, 64) most of the rest; a job that is done is skipped by the
    // later phases (its blocks exit at once).  The rule is sequential in h (ransac_scan_kernel), so the split does not
    // change the result.  (Round 2 scored 64 first: 2.1 ms per step of 500 jobs, 0.6 with 16.)
    const bool adaptive = prm->ransac_confidence > 0.f && prm->ransac_confidence < 1.f;
    uint32_t bounds[4] = {0u, 0u, 0u, 0u};
    int n_ph = 0;
    for (uint32_t b : {adaptive ? 16u : 256u, adaptive ? 64u : H, H})
      if (b <= H && b > bounds[n_ph]) bounds[++n_ph] = b;
    if (bounds[n_ph] < H) bounds[++n_ph] = H;
    GLOC_HIP(hipMemsetAsync(v.valid, 0, sizeof(uint32_t) * (size_t)H * n_jobs, s));  // never-generated = invalid
    GLOC_HIP(hipMemsetAsync(v.inliers, 0, sizeof(uint32_t) * (size_t)H * n_jobs, s));
    const float thr2 = prm->inlier_thresh * prm->inlier_thresh;
    // pairs per work-group of the scoring: 4096 -- or 1024 in a small batch (one query alone: 20 jobs x 31 chunks = 620
    // work-groups for 256 CUs, each walking 16 tiles behind two barriers: 75 us for the first 16 hypotheses)
    static const unsigned chunk_env = getenv("GLOC3D_RANSAC_CHUNK") ? (unsigned)atoi(getenv("GLOC3D_RANSAC_CHUNK")) : 0u;  // developer override
    const uint32_t chunk_len = chunk_env ? chunk_env : ((size_t)n_jobs * ((bd.max_src + SC_CHUNK - 1) / SC_CHUNK) >= 2048 ? (uint32_t)SC_CHUNK : 1024u);
    static_assert(1024 % SC_STAGE == 0 && SC_CHUNK % SC_STAGE == 0, "whole tiles");
    const unsigned cchunks = (bd.max_src + chunk_len - 1) / chunk_len;
    for (int ph = 0; ph < n_ph; ++ph) {
      const uint32_t h0 = bounds[ph], h1 = bounds[ph + 1], len = h1 - h0;
      const CandState* st = ph ? v.states : (const CandState*)nullptr;
      {
        ProfScope ps(h->prof, "ransac_hyp", s);
        hipLaunchKernelGGL(ransac_hyp_kernel, dim3((len + 127) / 128, n_jobs), dim3(128), 0, s, v.pairs, bd.ld,
                           v.jobs, prm->seed, H, h0, h1, st, v.Rt, v.valid);
        GLOC_HIP(hipGetLastError());
      }
      ProfScope ps(h->prof, "ransac_score", s);
      // hypotheses per work-group: 16 / 64 (its four waves share them and split every staged tile) or thread <-> hypothesis
      const uint32_t hpb = len <= 16 ? 16u : (len <= 64 ? 64u : 256u);
      static const unsigned parts_env = getenv("GLOC3D_RANSAC_PARTS") ? (unsigned)atoi(getenv("GLOC3D_RANSAC_PARTS")) : 0u;  // developer override
      const unsigned NP = parts_env ? parts_env : 8u;
      if (!adaptive && ph > 0 && len >= 512 && cchunks >= NP) {
        // every hypothesis scored, not every pair of every hypothesis: an eighth of the pairs at a time, the hypotheses
        // that can no longer beat the first phase's winner dropped in between (ransac_alive_kernel)
        uint32_t* a_idx = v.a_idx;
        uint32_t* a_cnt = v.a_cnt;
        for (unsigned q = 0; q < NP; ++q) {
          const unsigned c0 = q * cchunks / NP, c1 = (q + 1) * cchunks / NP;
          hipLaunchKernelGGL(ransac_alive_kernel, dim3(n_jobs), dim3(1024), 0, s, v.inliers, v.valid,
                             H, h0, h1, v.jobs, v.states, (uint32_t)(c0 * chunk_len), a_idx, a_cnt);
          hipLaunchKernelGGL(ransac_score_kernel, dim3((len + 255) / 256, c1 - c0, n_jobs), dim3(256), 0, s,
                             v.pairs, bd.ld, v.jobs, H, h0, 256u, v.Rt,
                             v.valid, thr2, st, v.inliers, a_idx, a_cnt, (uint32_t)c0, chunk_len);
        }
      } else {
        hipLaunchKernelGGL(ransac_score_kernel, dim3((len + hpb - 1) / hpb, cchunks, n_jobs), dim3(256), 0, s,
                           v.pairs, bd.ld, v.jobs, H, h0, hpb, v.Rt,
                           v.valid, thr2, st, v.inliers, (const uint32_t*)nullptr,
                           (const uint32_t*)nullptr, 0u, chunk_len);
      }
      if (ph + 1 < n_ph)
        hipLaunchKernelGGL(ransac_scan_kernel<false>, dim3(n_jobs), dim3(64), 0, s, v.inliers,
                           v.valid, v.Rt, H, h0, h1, v.jobs, prm->ransac_confidence,
                           prm->min_inlier_ratio, v.states);
      else
        hipLaunchKernelGGL(ransac_scan_kernel<true>, dim3(n_jobs), dim3(64), 0, s, v.inliers,
                           v.valid, v.Rt, H, h0, h1, v.jobs, prm->ransac_confidence,
                           prm->min_inlier_ratio, v.states);
      GLOC_HIP(hipGetLastError());
    }
    {
      ProfScope ps(h->prof, "accum", s);  // refit on the best hypothesis' inliers
      hipLaunchKernelGGL(accum_kernel<1>, dim3(nblocks, n_jobs), dim3(ACC_THREADS), 0, s, v.jobs,
                         v.states, v.corr, v.d2,
                         v.pairs, bd.ld, thr2, v.partials, bd.n_part);
      GLOC_HIP(hipGetLastError());
    }
    {
      ProfScope ps(h->prof, "solve", s);
      hipLaunchKernelGGL(solve_kernel<1>, dim3(v.split.hx ? 2 * n_jobs : n_jobs), dim3(SOLVE_THREADS), 0, s, v.partials,
                         bd.n_part, false, v.jobs, v.states, v.split, n_jobs);
      GLOC_HIP(hipGetLastError());
    }
  }
  for (uint32_t it = 0; it < prm->icp_iters && can && any_tgt; ++it) {
    if (have_corr && v.job0 == 0 && v.n_jobs == bd.n_jobs) {
      const uint32_t n_chain = chain_passes(h, bd, v, prm->icp_iters - it);
      if (n_chain) {  // a small batch: all the passes that are left in one launch
        GLOC_TRY(launch_nn_chain(h, bd, v, n_chain, gate2));
        break;
      }
    }
    GLOC_TRY(launch_nn(h, bd, v, have_corr, false, gate2));
    have_corr = true;
    if (!culled) {
      ProfScope ps(h->prof, "accum", s);
      hipLaunchKernelGGL(accum_kernel<0>, dim3(nblocks, n_jobs), dim3(ACC_THREADS), 0, s, v.jobs,
                         v.states, v.corr, v.d2,
                         (const f32x4*)nullptr, bd.ld, gate2, v.partials, bd.n_part);
      GLOC_HIP(hipGetLastError());
    }
    {
      ProfScope ps(h->prof, "solve", s);
      hipLaunchKernelGGL(solve_kernel<0>, dim3(v.split.hx ? 2 * n_jobs : n_jobs), dim3(SOLVE_THREADS), 0, s, v.partials,
                         bd.n_part, culled, v.jobs, v.states, v.split, n_jobs);
      GLOC_HIP(hipGetLastError());
    }
  }
  return GLOC_OK;
}

// The whole pipeline for a batch of jobs, device resident: S1 -> S2 (RANSAC + refit) -> S3 (ICP), ENQUEUED on the
// handle's stream with the copy of the per-job results behind it and an event behind that: returns without waiting.
int enqueue_jobs(gloc_reg* h, const std::vector<JobHost>& jh, const gloc_reg_params* prm) {
  const uint32_t n_jobs = (uint32_t)jh.size();
  if (n_jobs == 0) return GLOC_OK;
  const int cs = h->nn_src_per_lane;
  BatchDims bd{n_jobs, 0, 0, 0, 0};
  GLOC_TRY(ensure_pinned(h, n_jobs));
  h->chain_in_batch = false;
  *h->h_chain_err = 0u;
  Job* jd = h->h_jobs;
  bool can = false, any_tgt = false;
  for (uint32_t c = 0; c < n_jobs; ++c) {
    const DevScan &s = jh[c].src, &t = jh[c].tgt;
    const uint32_t ng = (uint32_t)((s.n + 64 * cs - 1) / (64 * cs));
    jd[c] = Job{s.idx.pts, s.order, s.idx.inv, t.xyz, t.idx, (uint32_t)s.n, ng, jh[c].stream_id, 0u};
    bd.max_src = std::max<uint32_t>(bd.max_src, (uint32_t)s.n);
    bd.max_groups = std::max(bd.max_groups, ng);
    can |= s.n >= 3;
    any_tgt |= t.n >= 1;
  }
  const uint32_t nblocks = (bd.max_src + ACC_PER_BLOCK - 1) / ACC_PER_BLOCK;
  bd.n_part = (std::max<uint32_t>(std::max(bd.max_groups, nblocks), 1) + 31u) & ~31u;  // (a job's row of a [job][n_part] table: whole 128-byte lines)
  bd.ld = ((size_t)bd.max_src + 127) & ~(size_t)127;
  h->last_ld = bd.ld;
  h->last_jobs = n_jobs;
  hipStream_t s = h->stream;
  for (uint32_t c = 0; c < n_jobs; ++c) {
    init_state(h->h_states[c], jh[c].init_T, prm->ransac_iters);
    if (jh[c].src.n < 3) h->h_states[c].frozen = 1;  // nothing to estimate: T stays the initial guess
  }
  GLOC_TRY(h->jobs.ensure(sizeof(Job) * n_jobs, s));
  if (!h->counters.p) {
    GLOC_TRY(h->counters.ensure(8 * NN_STAT_SLOTS, s));
    GLOC_HIP(hipMemsetAsync(h->counters.p, 0, 8 * NN_STAT_SLOTS, s));
  }
  GLOC_TRY(h->states.ensure(sizeof(CandState) * n_jobs, s));
  GLOC_TRY(h->corr.ensure(sizeof(uint32_t) * std::max<size_t>(bd.ld, 1) * n_jobs, s));
  GLOC_TRY(h->d2.ensure(sizeof(float) * std::max<size_t>(bd.ld, 1) * n_jobs, s));
  GLOC_TRY(h->partials.ensure(sizeof(double) * ACC_NV * (size_t)bd.n_part * n_jobs, s));
  GLOC_HIP(hipMemcpyAsync(h->jobs.p, jd, sizeof(Job) * n_jobs, hipMemcpyHostToDevice, s));
  GLOC_HIP(hipMemcpyAsync(h->states.p, h->h_states, sizeof(CandState) * n_jobs, hipMemcpyHostToDevice, s));
  GLOC_TRY(setup_split(h, bd, cs));
  const bool ransac = can && any_tgt && prm->ransac_iters > 0;
  const size_t H = prm->ransac_iters;
  const bool adaptive = prm->ransac_confidence > 0.f && prm->ransac_confidence < 1.f;
  if (ransac) {
    GLOC_TRY(h->pairs.ensure(sizeof(f32x4) * 2 * bd.ld * n_jobs, s));
    GLOC_TRY(h->Rt.ensure(sizeof(float) * 12 * H * n_jobs, s));
    GLOC_TRY(h->valid.ensure(sizeof(uint32_t) * H * n_jobs, s));
    GLOC_TRY(h->inliers.ensure(sizeof(uint32_t) * H * n_jobs, s));
    if (!adaptive) GLOC_TRY(h->alive.ensure(sizeof(uint32_t) * (H + 1) * n_jobs, s));
  }
  // Sub-batches (round 6): a SMALL batch -- one query alone: 20 jobs -- is cut into G runs of jobs, each enqueued on its
  // own stream.  The jobs of a batch never depend on each other, but one stream makes all of them wait at every launch
  // boundary: 21 x (search launch + solve) with the chip ramping up and draining 42 times (a pass of 20 jobs is 19 380
  // waves over 6 144 slots: 69 us at full occupancy, 105 - 120 measured, + 14 us of solve with the chip idle).  With G
  // streams one sub-batch's solve and ramp run under the others' searches.  Results are the same bits (a job's
  // arithmetic never sees the batch).  Not with the per-kernel events on (they bracket launches on ONE stream), nor the trace.
  static const int g_env = getenv("GLOC3D_REG_SUBBATCHES") ? atoi(getenv("GLOC3D_REG_SUBBATCHES")) : -1;  // developer override
  uint32_t G = 1;
  if (!h->prof.enabled && !h->trace_on && h->nn_mode != 1) {
    if (g_env >= 1) G = (uint32_t)g_env;
    else if (h->sub_batches > 0) G = (uint32_t)h->sub_batches;
    // (-1, the default, is OFF: measured for one query alone -- 20 jobs -- 3.20 ms on one stream, 3.28 with 2 sub-batches,
    // 3.9 with 4, 5.0 with 8, enqueued one after the other or from a host thread each: a search launch of even 5 jobs has
    // 4 845 + helper waves for 6 144 slots, so the streams' kernels mostly run one after the other, each with its own ramp)
  }
  G = std::max<uint32_t>(1, std::min<uint32_t>(std::min<uint32_t>(G, n_jobs), gloc_reg::MAX_SUB));
  GLOC_TRY(setup_heavy(h, bd, cs, G));
  std::vector<WsView> views(G);
  for (uint32_t g = 0; g < G; ++g) {
    const uint32_t j0 = (uint32_t)((uint64_t)n_jobs * g / G), j1 = (uint32_t)((uint64_t)n_jobs * (g + 1) / G);
    WsView& v = views[g];
    v = WsView{};
    v.s = s;
    v.job0 = j0;
    v.n_jobs = j1 - j0;
    v.jobs = h->jobs.as<Job>() + j0;
    v.states = h->states.as<CandState>() + j0;
    v.corr = h->corr.as<uint32_t>() + (size_t)j0 * bd.ld;
    v.d2 = h->d2.as<float>() + (size_t)j0 * bd.ld;
    v.partials = h->partials.as<double>() + (size_t)j0 * bd.n_part * ACC_NV;
    if (ransac) {
      v.pairs = h->pairs.as<f32x4>() + 2 * bd.ld * (size_t)j0;
      v.Rt = h->Rt.as<float>() + 12 * H * j0;
      v.valid = h->valid.as<uint32_t>() + H * j0;
      v.inliers = h->inliers.as<uint32_t>() + H * j0;
      if (!adaptive) {
        v.a_idx = h->alive.as<uint32_t>() + H * j0;
        v.a_cnt = h->alive.as<uint32_t>() + H * n_jobs + j0;
      }
    }
    v.split = h->split;
    if (h->split.hx) {  // [job][...] arrays: the sub-batch's jobs
      const size_t S = 64 * (size_t)cs;
      v.split.work += (size_t)j0 * bd.n_part;
      v.split.plan += (size_t)j0 * bd.n_part;
      v.split.ticket += (size_t)j0 * h->split.hx;
      v.split.helper += (size_t)j0 * h->split.hx;
      v.split.skey += (size_t)j0 * h->split.hx * S;
    }
    v.heavy = h->heavy_of[g];
  }
  if (G > 1) {
    GLOC_TRY(ensure_sub_streams(h, G));
    GLOC_HIP(hipEventRecord(h->fork_ev, s));  // uploads, memsets and everything earlier on the handle's stream
    for (uint32_t g = 1; g < G; ++g) {
      views[g].s = h->sub_stream[g - 1];
      GLOC_HIP(hipStreamWaitEvent(views[g].s, h->fork_ev, 0));
    }
  }
  int rc = GLOC_OK;
  if (G > 1) {
    // one host thread per sub-batch: 4 x 60 launches enqueued one stream after the other leave the later streams empty
    // while the first runs (measured: 3.2 -> 3.9 ms for one query alone); side by side the streams fill together
    std::vector<std::future<int>> fut;
    for (uint32_t g = 1; g < G; ++g)
      fut.push_back(std::async(std::launch::async, [&, g]() {
        if (hipSetDevice(h->device) != hipSuccess) return (int)GLOC_ERR_HIP;
        return enqueue_pipeline(h, bd, prm, views[g], can, any_tgt, nblocks);
      }));
    rc = enqueue_pipeline(h, bd, prm, views[0], can, any_tgt, nblocks);
    for (auto& f : fut) {
      const int r = f.get();
      if (rc == GLOC_OK) rc = r;
    }
  } else {
    rc = enqueue_pipeline(h, bd, prm, views[0], can, any_tgt, nblocks);
  }
  for (uint32_t g = 1; g < G; ++g) {  // (also after a failure part-way: whatever was launched is joined)
    (void)hipEventRecord(h->join_ev[g - 1], views[g].s);
    (void)hipStreamWaitEvent(s, h->join_ev[g - 1], 0);
  }
  GLOC_TRY(rc);
  GLOC_HIP(hipMemcpyAsync(h->h_states, h->states.p, sizeof(CandState) * n_jobs, hipMemcpyDeviceToHost, s));
  GLOC_HIP(hipEventRecord(h->done_ev, s));
  if (h->chain_in_batch) {  // (a small batch: a few KB)
    h->retry_jobs.resize(sizeof(JobHost) * n_jobs);
    memcpy(h->retry_jobs.data(), jh.data(), sizeof(JobHost) * n_jobs);
    h->retry_T.assign((size_t)16 * n_jobs, 0.f);
    for (uint32_t c = 0; c < n_jobs; ++c)
      if (jh[c].init_T) memcpy(&h->retry_T[(size_t)16 * c], jh[c].init_T, sizeof(float) * 16);
    h->retry_prm = *prm;
  }
  return GLOC_OK;
}

// Waits for the results of the batch enqueue_jobs() queued last (its own event: not for the stream, which may carry the
// next batch of a handle sharing it) and unpacks them, per job, in job order.
int collect_jobs(gloc_reg* h, uint32_t n_jobs, const size_t* n_src_of, float max_rmse, float max_final_step, float* out_T,
                 float* out_rmse, uint32_t* out_inliers, int* out_ok) {
  h->last_final_step.assign(n_jobs, 0.f);
  if (n_jobs == 0) return GLOC_OK;
  GLOC_HIP(hipEventSynchronize(h->done_ev));
  if (h->chain_in_batch && *h->h_chain_err) {
    // A wait inside the chained launch ran out (NN_CHAIN_WAIT_TICKS): its waves left without finishing the passes -- the
    // poses are not results.  The handle goes back to one launch per pass for good, and THIS batch is run again that
    // way, here (its scans are still pinned: the batch has not been collected): the caller gets the launch-by-launch bits,
    // late.  Counted (gloc_reg_debug_chain) and said on stderr once per handle.
    const bool first = !h->chain_broken;
    h->chain_broken = true;
    h->chain_timeouts++;
    if (first)
      fprintf(stderr, "[gloc3d] the chained ICP passes of a batch of %u jobs timed out on the device (a wait for a job's solve ran out): "
                      "the batch is run again launch by launch, and this handle launches pass by pass from now on\n", n_jobs);
    GLOC_REQUIRE(h->retry_jobs.size() == sizeof(JobHost) * n_jobs, GLOC_ERR_HIP,
                 "the chained ICP passes of a batch of %u jobs timed out on the device and the batch cannot be run again", n_jobs);
    std::vector<JobHost> jh(n_jobs);
    memcpy(jh.data(), h->retry_jobs.data(), sizeof(JobHost) * n_jobs);
    for (uint32_t c = 0; c < n_jobs; ++c)
      if (jh[c].init_T) jh[c].init_T = &h->retry_T[(size_t)16 * c];
    const gloc_reg_params prm = h->retry_prm;
    GLOC_TRY(enqueue_jobs(h, jh, &prm));
    GLOC_HIP(hipEventSynchronize(h->done_ev));
  }
  static const bool heavy_dbg = getenv("GLOC3D_NN_HEAVY_DEBUG") != nullptr;  // developer switch: the length of the last cold pass's list
  if (heavy_dbg && h->heavy_of[0].cap) {
    uint32_t cnt = 0;
    (void)hipMemcpy(&cnt, h->heavy_of[0].count, 4, hipMemcpyDeviceToHost);
    fprintf(stderr, "[gloc3d] cold pass of %u jobs: %u groups given up in the first sub-batch (list of %u)\n", n_jobs, cnt, h->heavy_of[0].cap);
  }
  for (uint32_t c = 0; c < n_jobs; ++c) {
    const CandState& st = h->h_states[c];
    const size_t n_src = n_src_of[c];
    float* T = out_T + 16 * (size_t)c;
    for (int i = 0; i < 3; ++i) {
      for (int j = 0; j < 3; ++j) T[4 * i + j] = st.Tf[3 * i + j];
      T[4 * i + 3] = st.Tf[9 + i];
    }
    T[12] = 0.f; T[13] = 0.f; T[14] = 0.f; T[15] = 1.f;
    const float rmse = n_src ? (float)std::sqrt(st.sum_d2 / (double)n_src) : 0.f;
    if (out_rmse) out_rmse[c] = rmse;
    if (out_inliers) out_inliers[c] = st.best_inl;
    // plausibility of the estimate (include/gloc3d.h: max_final_step, max_rmse): the ICP converged, the residual is bounded
    h->last_final_step[c] = st.last_step;
    if (out_ok)
      out_ok[c] = st.ok && !(max_rmse > 0.f && !(rmse <= max_rmse)) && !(max_final_step > 0.f && !(st.last_step <= max_final_step));
  }
  return GLOC_OK;
}

int run_jobs(gloc_reg* h, const std::vector<JobHost>& jh, const gloc_reg_params* prm, float* out_T, float* out_rmse,
             uint32_t* out_inliers, int* out_ok) {
  GLOC_TRY(enqueue_jobs(h, jh, prm));
  std::vector<size_t> n_src(jh.size());
  for (size_t c = 0; c < jh.size(); ++c) n_src[c] = jh[c].src.n;
  return collect_jobs(h, (uint32_t)jh.size(), n_src.data(), prm->max_rmse, prm->icp_iters ? prm->max_final_step : 0.f, out_T,
                      out_rmse, out_inliers, out_ok);
}

// Every entry point that runs jobs on the handle's workspaces (enqueue_jobs / launch_nn: pinned staging, job table,
// states, corr, partials, the done event) is refused while a batch is between gloc_reg_batch_multi_begin and _end:
// it would overwrite what that batch's D2H copy and unpacking still read.
#define GLOC_NOT_PENDING(h) \
  GLOC_REQUIRE(!(h)->pending.active, GLOC_ERR_STATE, "a batch is in flight on this handle: call gloc_reg_batch_multi_end first")

int check_params(const gloc_reg_params* p) {
  GLOC_REQUIRE(p, GLOC_ERR_INVALID, "params is null");
  GLOC_REQUIRE(p->ransac_iters <= (1u << 20), GLOC_ERR_INVALID, "ransac_iters too large");
  GLOC_REQUIRE(p->icp_iters <= 10000, GLOC_ERR_INVALID, "icp_iters too large");
  GLOC_REQUIRE(p->ransac_iters == 0 || p->inlier_thresh > 0.f, GLOC_ERR_INVALID,
               "inlier_thresh must be > 0");
  return GLOC_OK;
}

int ensure_store(gloc_reg* h) {
  if (h->store) return GLOC_OK;
  GLOC_TRY(gloc_scan_store_create(h->device, &h->own_store));
  h->store = h->own_store;
  h->store->attached++;
  return GLOC_OK;
}

// Temporary resident copies of caller-owned host scans (uploaded + indexed), released by the caller.
struct TempScans {
  gloc_scan_store* st;
  std::vector<DevScan> scans;
  explicit TempScans(gloc_scan_store* s) : st(s) {}
  // Scan 0 = the source (launch order for `cs` sources per lane), scans 1.. = targets: uploaded and indexed in ONE
  // launch sequence (round 2: one sequence of ~25 launches per scan, 21 scans for a top-20 registration)
  int add_all(const float* src, size_t n_src, int cs, const float* const* tgt, const size_t* n_tgt, size_t n_tgts,
              bool target_index) {
    std::lock_guard<std::mutex> lk(st->mu);
    std::vector<const float*> p(1 + n_tgts);
    std::vector<size_t> n(1 + n_tgts);
    p[0] = src;
    n[0] = n_src;
    for (size_t c = 0; c < n_tgts; ++c) {
      p[1 + c] = tgt[c];
      n[1 + c] = n_tgt[c];
    }
    scans.resize(1 + n_tgts);
    const int rc = store_make_scans(st, 1 + n_tgts, p.data(), n.data(), 3, false, scans.data());
    if (rc != GLOC_OK) {
      scans.clear();  // (store_make_scans released what it had allocated)
      return rc;
    }
    if (target_index && n_tgts) {
      std::vector<DevScan*> ps(n_tgts);
      for (size_t c = 0; c < n_tgts; ++c) ps[c] = &scans[1 + c];
      GLOC_TRY(store_build_target_indices(st, ps.data(), n_tgts));
    }
    GLOC_TRY(store_build_order(st, scans[0], cs));
    for (size_t c = 0; c < n_tgts; ++c) scans[1 + c].order = nullptr;  // a target's launch order is never read
    return GLOC_OK;
  }
  ~TempScans() {
    std::lock_guard<std::mutex> lk(st->mu);
    for (auto& s : scans) store_free_scan(st, s, true);
  }
};

}  // namespace

extern "C" {

void gloc_reg_default_params(gloc_reg_params* p) {
  if (!p) return;
  p->ransac_iters = 3000;   // registration/loop_detector.cpp:257
  p->inlier_thresh = 0.6f;  // 3 * 0.2 m: loop_detector.cpp:257, loop_detector.h:116
  p->min_inlier_ratio = 0.3f;
  p->icp_iters = 30;  // registration/global_registration.cpp:242
  p->max_corr_dist = 0.f;
  p->seed = 1234;
  p->ransac_confidence = 0.99f;  // cv::estimateAffinePartial2D's default, used by the reference
  p->max_rmse = 0.f;
  p->max_final_step = 0.f;  // off, as the reference: its 3-D stage takes what the ICP returns (GLOC_REG_FINAL_STEP_SUGGESTED: see the header)
}

int gloc_reg_create(int device, gloc_reg** out) {
  GLOC_REQUIRE(out, GLOC_ERR_INVALID, "out is null");
  *out = nullptr;
  GLOC_TRY(select_device(device));
  gloc_reg* h = new (std::nothrow) gloc_reg;
  GLOC_REQUIRE(h, GLOC_ERR_NOMEM, "host allocation failed");
  h->device = device;
  hipError_t e = hipStreamCreateWithFlags(&h->own_stream, hipStreamNonBlocking);
  if (e != hipSuccess) {
    set_err("hipStreamCreate failed: %s", hipGetErrorString(e));
    delete h;
    return GLOC_ERR_HIP;
  }
  h->stream = h->own_stream;
  e = hipEventCreateWithFlags(&h->done_ev, hipEventDisableTiming);
  if (e != hipSuccess) {
    set_err("hipEventCreate failed: %s", hipGetErrorString(e));
    (void)hipStreamDestroy(h->own_stream);
    delete h;
    return GLOC_ERR_HIP;
  }
  *out = h;
  return GLOC_OK;
}

int gloc_reg_attach_store(gloc_reg* h, gloc_scan_store* store) {
  GLOC_REQUIRE(h, GLOC_ERR_INVALID, "null handle");
  GLOC_REQUIRE(!store || store->device == h->device, GLOC_ERR_INVALID, "store lives on device %d, the handle on %d",
               store ? store->device : -1, h->device);
  GLOC_HIP(hipSetDevice(h->device));
  GLOC_HIP(hipStreamSynchronize(h->stream));
  if (h->store) h->store->attached--;
  h->store = store ? store : h->own_store;
  if (h->store) h->store->attached++;
  return GLOC_OK;
}

int gloc_reg_scan_clear(gloc_reg* h) {
  GLOC_REQUIRE(h, GLOC_ERR_INVALID, "null handle");
  GLOC_HIP(hipSetDevice(h->device));
  GLOC_HIP(hipStreamSynchronize(h->stream));
  if (!h->store) return GLOC_OK;
  return gloc_scan_store_clear(h->store);
}

int gloc_reg_destroy(gloc_reg* h) {
  if (!h) return GLOC_OK;
  (void)hipSetDevice(h->device);
  (void)hipStreamSynchronize(h->stream);
  if (h->pending.active && h->pending.store) store_pin(h->pending.store, h->pending.pinned.data(), h->pending.pinned.size(), -1);
  h->pending.pinned.clear();
  h->pending.active = false;
  if (h->store) h->store->attached--;
  h->store = nullptr;
  if (h->own_store) (void)gloc_scan_store_destroy(h->own_store);
  h->prof.destroy();
  for (DevBuf* b : {&h->jobs, &h->states, &h->corr, &h->d2, &h->pairs, &h->Rt, &h->valid, &h->inliers,
                    &h->partials, &h->export_idx, &h->export_d2, &h->counters, &h->trace, &h->split_zero, &h->split_ff, &h->alive, &h->heavy_buf})
    b->release();
  if (h->done_ev) (void)hipEventDestroy(h->done_ev);
  if (h->fork_ev) (void)hipEventDestroy(h->fork_ev);
  for (uint32_t g = 0; g + 1 < gloc_reg::MAX_SUB; ++g) {
    if (h->join_ev[g]) (void)hipEventDestroy(h->join_ev[g]);
    if (h->sub_stream[g]) (void)hipStreamDestroy(h->sub_stream[g]);
  }
  if (h->pin) (void)hipHostFree(h->pin);
  if (h->own_stream) (void)hipStreamDestroy(h->own_stream);
  delete h;
  return GLOC_OK;
}

int gloc_reg_set_stream(gloc_reg* h, void* hip_stream) {
  GLOC_REQUIRE(h, GLOC_ERR_INVALID, "null handle");
  GLOC_HIP(hipSetDevice(h->device));
  GLOC_HIP(hipStreamSynchronize(h->stream));
  h->stream = hip_stream ? (hipStream_t)hip_stream : h->own_stream;
  return GLOC_OK;
}

int gloc_reg_synchronize(gloc_reg* h) {
  GLOC_REQUIRE(h, GLOC_ERR_INVALID, "null handle");
  GLOC_HIP(hipSetDevice(h->device));
  GLOC_HIP(hipStreamSynchronize(h->stream));
  return GLOC_OK;
}

int gloc_reg_set_option(gloc_reg* h, int option, int64_t value) {
  GLOC_REQUIRE(h, GLOC_ERR_INVALID, "null handle");
  if (option == GLOC_REG_OPT_PROFILE) {
    h->prof.enabled = value != 0;
    return GLOC_OK;
  }
  if (option == GLOC_REG_OPT_NN_MODE) {
    GLOC_REQUIRE(value == GLOC_REG_NN_CULLED || value == GLOC_REG_NN_EXHAUSTIVE, GLOC_ERR_INVALID,
                 "bad nn mode %lld", (long long)value);
    h->nn_mode = (int)value;
    return GLOC_OK;
  }
  if (option == GLOC_REG_OPT_NN_JOB_GROUP) {
    GLOC_REQUIRE(value >= 1 && value <= 65536, GLOC_ERR_INVALID, "must be in [1, 65536]");
    h->nn_job_group = (int)value;
    h->nn_job_group_set = true;
    return GLOC_OK;
  }
  if (option == GLOC_REG_OPT_TEMP_TARGET_INDEX) {
    h->temp_target_index = value != 0;
    return GLOC_OK;
  }
  if (option == GLOC_REG_OPT_NN_SUB_JOBS) {
    GLOC_REQUIRE(value >= 0 && value <= 64, GLOC_ERR_INVALID, "must be in [0, 64]");
    h->nn_sub_jobs = (int)value;
    return GLOC_OK;
  }
  if (option == GLOC_REG_OPT_NN_SPLIT_HELPERS) {
    GLOC_REQUIRE(value >= -1 && value <= 4096, GLOC_ERR_INVALID, "must be in [-1, 4096]");
    h->nn_split_helpers = (int)value;
    return GLOC_OK;
  }
  if (option == GLOC_REG_OPT_SUB_BATCHES) {
    GLOC_REQUIRE(value >= -1 && value <= (int64_t)gloc_reg::MAX_SUB, GLOC_ERR_INVALID, "must be in [-1, %u]", gloc_reg::MAX_SUB);
    h->sub_batches = (int)value;
    return GLOC_OK;
  }
  if (option == GLOC_REG_OPT_NN_CHAIN) {
    GLOC_REQUIRE(value == 0 || value == 1, GLOC_ERR_INVALID, "must be 0 or 1");
    h->nn_chain = (int)value;
    if (value) h->chain_broken = false;
    return GLOC_OK;
  }
  if (option == GLOC_REG_OPT_NN_HEAVY_THRESH) {
    GLOC_REQUIRE(value >= 0 && value <= 65535, GLOC_ERR_INVALID, "must be in [0, 65535]");
    h->nn_heavy_thresh = (int)value;
    return GLOC_OK;
  }
  if (option == GLOC_REG_OPT_NN_SPLIT_THRESH) {
    GLOC_REQUIRE(value >= 0 && value <= 0x3FFFFFFF, GLOC_ERR_INVALID, "must be in [0, 2^30)");
    h->nn_split_thresh = (uint32_t)value;
    h->nn_split_thresh_set = true;
    return GLOC_OK;
  }
  if (option == GLOC_REG_OPT_NN_SRC_PER_LANE) {
    GLOC_REQUIRE(value == 1 || value == 2 || value == 4, GLOC_ERR_INVALID, "must be 1, 2 or 4");
    h->nn_src_per_lane = (int)value;
    return GLOC_OK;
  }
  set_err("unknown option %d", option);
  return GLOC_ERR_INVALID;
}

int gloc_reg_scan_upload(gloc_reg* h, const float* pts, size_t n, size_t stride_floats,
                         uint32_t* scan_id) {
  GLOC_REQUIRE(h, GLOC_ERR_INVALID, "null handle");
  GLOC_HIP(hipSetDevice(h->device));
  GLOC_TRY(ensure_store(h));
  return gloc_scan_store_add(h->store, pts, n, stride_floats, scan_id);
}

int gloc_reg_scan_build_target_index(gloc_reg* h, uint32_t scan_id) {
  GLOC_REQUIRE(h, GLOC_ERR_INVALID, "null handle");
  GLOC_REQUIRE(h->store, GLOC_ERR_INVALID, "unknown scan id %u", scan_id);
  GLOC_HIP(hipSetDevice(h->device));
  GLOC_HIP(hipStreamSynchronize(h->stream));  // no launch of this handle may still read the scan
  return gloc_scan_store_build_target_index(h->store, scan_id);
}

int gloc_reg_scan_release(gloc_reg* h, uint32_t scan_id) {
  GLOC_REQUIRE(h, GLOC_ERR_INVALID, "null handle");
  GLOC_REQUIRE(h->store, GLOC_ERR_INVALID, "unknown scan id %u", scan_id);
  GLOC_HIP(hipSetDevice(h->device));
  GLOC_HIP(hipStreamSynchronize(h->stream));  // no launch of this handle may still read the scan
  return gloc_scan_store_release(h->store, scan_id);
}

int gloc_reg_scan_count(const gloc_reg* h, size_t* n_scans) {
  GLOC_REQUIRE(h && n_scans, GLOC_ERR_INVALID, "null argument");
  *n_scans = 0;
  if (!h->store) return GLOC_OK;
  return gloc_scan_store_count(h->store, n_scans);
}

int gloc_reg_batch(gloc_reg* h, const float* q_xyz, size_t nq_pts, const float* const* cand_xyz,
                   const size_t* cand_npts, size_t n_cand, const uint32_t* cand_stream_ids,
                   const float* init_T, const gloc_reg_params* params, float* out_T,
                   float* out_rmse, uint32_t* out_inliers, int* out_ok) {
  GLOC_REQUIRE(h && out_T && (q_xyz || nq_pts == 0), GLOC_ERR_INVALID, "null argument");
  GLOC_REQUIRE(n_cand >= 1 && n_cand <= 4096 && cand_xyz && cand_npts, GLOC_ERR_INVALID,
               "n_cand = %zu outside [1,4096] or null candidate arrays", n_cand);
  GLOC_REQUIRE(nq_pts < (1ull << 31), GLOC_ERR_INVALID, "query scan too large");
  GLOC_NOT_PENDING(h);
  GLOC_TRY(check_params(params));
  GLOC_HIP(hipSetDevice(h->device));
  for (size_t c = 0; c < n_cand; ++c) {
    GLOC_REQUIRE(cand_xyz[c] || cand_npts[c] == 0, GLOC_ERR_INVALID, "candidate %zu is null", c);
    GLOC_REQUIRE(cand_npts[c] < (1ull << 31), GLOC_ERR_INVALID, "candidate scan too large");
  }
  GLOC_TRY(ensure_store(h));
  TempScans tmp(h->store);  // released on return
  GLOC_TRY(tmp.add_all(q_xyz, nq_pts, h->nn_src_per_lane, cand_xyz, cand_npts, n_cand, h->temp_target_index));
  std::vector<JobHost> jh(n_cand);
  for (size_t c = 0; c < n_cand; ++c)
    jh[c] = JobHost{tmp.scans[0], tmp.scans[c + 1], cand_stream_ids ? cand_stream_ids[c] : (uint32_t)c,
                    init_T ? init_T + 16 * c : nullptr};
  int rc = run_jobs(h, jh, params, out_T, out_rmse, out_inliers, out_ok);
  (void)hipStreamSynchronize(h->stream);
  return rc;
}

int gloc_reg_batch_multi_begin(gloc_reg* h, size_t n_queries, const uint32_t* q_scan_ids, const uint32_t* cand_scan_ids,
                               size_t n_cand, const uint32_t* cand_stream_ids, const float* init_T,
                               const gloc_reg_params* params) {
  GLOC_REQUIRE(h && q_scan_ids && cand_scan_ids, GLOC_ERR_INVALID, "null argument");
  GLOC_REQUIRE(n_queries >= 1 && n_cand >= 1 && n_queries * n_cand <= 65536, GLOC_ERR_INVALID,
               "n_queries x n_cand = %zu x %zu outside [1, 65536]", n_queries, n_cand);
  GLOC_REQUIRE(h->store, GLOC_ERR_INVALID, "no scan store: upload scans or attach a store first");
  GLOC_REQUIRE(!h->pending.active, GLOC_ERR_STATE, "a batch is already in flight on this handle: call gloc_reg_batch_multi_end first");
  GLOC_TRY(check_params(params));
  GLOC_HIP(hipSetDevice(h->device));
  const size_t total = n_queries * n_cand;
  std::vector<JobHost> jh;
  gloc_reg::Pending& P = h->pending;
  P.slot.clear();
  P.n_src.clear();
  // The scans the jobs read: views and pins in ONE step under the store's mutex, BEFORE anything is launched
  // (gloc_scan_store_build_target_index refuses to re-sort them in place and gloc_scan_store_release to free them until
  // _end); released again if the enqueue fails part-way -- whatever was launched is waited for first.
  P.pinned.assign(q_scan_ids, q_scan_ids + n_queries);
  std::vector<int> cs_of(n_queries, h->nn_src_per_lane);
  for (size_t o = 0; o < total; ++o)
    if (cand_scan_ids[o] != 0xFFFFFFFFu) {  // (0xFFFFFFFF: "no candidate", a retrieval list shorter than k)
      P.pinned.push_back(cand_scan_ids[o]);
      cs_of.push_back(0);  // a target's launch order is never read
    }
  std::vector<DevScan> view(P.pinned.size());
  {
    const int rc = store_get_pinned(h->store, P.pinned.data(), cs_of.data(), P.pinned.size(), view.data());
    if (rc != GLOC_OK) {
      P.pinned.clear();
      return rc;
    }
  }
  jh.reserve(total);
  size_t vi = n_queries;
  for (size_t q = 0; q < n_queries; ++q) {
    const DevScan& src = view[q];
    for (size_t c = 0; c < n_cand; ++c) {
      const size_t o = q * n_cand + c;
      if (cand_scan_ids[o] == 0xFFFFFFFFu) continue;
      JobHost j;
      j.src = src;
      j.tgt = view[vi++];
      j.stream_id = cand_stream_ids ? cand_stream_ids[o] : (uint32_t)c;
      j.init_T = init_T ? init_T + 16 * o : nullptr;
      jh.push_back(j);
      P.slot.push_back(o);
      P.n_src.push_back(src.n);
    }
  }
  // rows without a candidate keep the initial guess (identity), not ok
  P.def_T.resize(16 * total);
  for (size_t o = 0; o < total; ++o)
    for (int i = 0; i < 16; ++i) P.def_T[16 * o + i] = init_T ? init_T[16 * o + i] : ((i % 5 == 0) ? 1.f : 0.f);
  P.total = total;
  P.max_rmse = params->max_rmse;
  P.max_final_step = params->icp_iters ? params->max_final_step : 0.f;
  const int rc = enqueue_jobs(h, jh, params);
  if (rc != GLOC_OK) {
    (void)hipStreamSynchronize(h->stream);
    store_pin(h->store, P.pinned.data(), P.pinned.size(), -1);
    P.pinned.clear();
    return rc;
  }
  P.active = true;
  P.store = h->store;
  return GLOC_OK;
}

int gloc_reg_batch_multi_end(gloc_reg* h, float* out_T, float* out_rmse, uint32_t* out_inliers, int* out_ok) {
  GLOC_REQUIRE(h && out_T, GLOC_ERR_INVALID, "null argument");
  GLOC_REQUIRE(h->pending.active, GLOC_ERR_STATE, "no batch in flight on this handle");
  GLOC_HIP(hipSetDevice(h->device));
  gloc_reg::Pending& P = h->pending;
  P.active = false;
  struct Done {  // whatever happens below, the batch no longer reads the store's scans once its event has been waited for
    gloc_reg* h;
    gloc_scan_store* st;
    ~Done() {
      (void)hipEventSynchronize(h->done_ev);
      if (st) store_pin(st, h->pending.pinned.data(), h->pending.pinned.size(), -1);
      h->pending.pinned.clear();
    }
  } done{h, P.store};
  P.store = nullptr;
  for (size_t o = 0; o < P.total; ++o) {
    std::copy(P.def_T.begin() + 16 * o, P.def_T.begin() + 16 * (o + 1), out_T + 16 * o);
    if (out_rmse) out_rmse[o] = 0.f;
    if (out_inliers) out_inliers[o] = 0;
    if (out_ok) out_ok[o] = 0;
  }
  const size_t nj = P.slot.size();
  std::vector<float> T(16 * std::max<size_t>(nj, 1)), rm(std::max<size_t>(nj, 1));
  std::vector<uint32_t> inl(std::max<size_t>(nj, 1));
  std::vector<int> ok(std::max<size_t>(nj, 1));
  GLOC_TRY(collect_jobs(h, (uint32_t)nj, P.n_src.data(), P.max_rmse, P.max_final_step, T.data(), rm.data(), inl.data(), ok.data()));
  for (size_t j = 0; j < nj; ++j) {
    const size_t o = P.slot[j];
    std::copy(T.begin() + 16 * j, T.begin() + 16 * (j + 1), out_T + 16 * o);
    if (out_rmse) out_rmse[o] = rm[j];
    if (out_inliers) out_inliers[o] = inl[j];
    if (out_ok) out_ok[o] = ok[j];
  }
  return GLOC_OK;
}

int gloc_reg_batch_multi(gloc_reg* h, size_t n_queries, const uint32_t* q_scan_ids,
                         const uint32_t* cand_scan_ids, size_t n_cand, const uint32_t* cand_stream_ids,
                         const float* init_T, const gloc_reg_params* params, float* out_T,
                         float* out_rmse, uint32_t* out_inliers, int* out_ok) {
  GLOC_REQUIRE(out_T, GLOC_ERR_INVALID, "null argument");
  GLOC_TRY(gloc_reg_batch_multi_begin(h, n_queries, q_scan_ids, cand_scan_ids, n_cand, cand_stream_ids, init_T, params));
  return gloc_reg_batch_multi_end(h, out_T, out_rmse, out_inliers, out_ok);
}

int gloc_reg_first_success_multi(gloc_reg* h, size_t n_queries, const uint32_t* q_scan_ids,
                                 const uint32_t* cand_scan_ids, size_t n_cand, const float* init_T,
                                 const gloc_reg_params* params, int* out_rank, float* out_T, float* out_rmse,
                                 uint32_t* out_inliers, uint64_t* out_jobs_run) {
  GLOC_REQUIRE(h && q_scan_ids && cand_scan_ids && out_rank && out_T, GLOC_ERR_INVALID, "null argument");
  GLOC_NOT_PENDING(h);
  GLOC_REQUIRE(n_queries >= 1 && n_cand >= 1 && n_queries * n_cand <= 65536, GLOC_ERR_INVALID,
               "n_queries x n_cand = %zu x %zu outside [1, 65536]", n_queries, n_cand);
  GLOC_REQUIRE(h->store, GLOC_ERR_INVALID, "no scan store: upload scans or attach a store first");
  GLOC_TRY(check_params(params));
  GLOC_HIP(hipSetDevice(h->device));
  std::vector<DevScan> src(n_queries);
  for (size_t q = 0; q < n_queries; ++q) {
    GLOC_TRY(store_get(h->store, q_scan_ids[q], h->nn_src_per_lane, &src[q]));
    out_rank[q] = -1;
    float* T = out_T + 16 * q;
    for (int i = 0; i < 16; ++i) T[i] = (i % 5 == 0) ? 1.f : 0.f;
    if (out_rmse) out_rmse[q] = 0.f;
    if (out_inliers) out_inliers[q] = 0;
  }
  std::vector<size_t> pending(n_queries);
  for (size_t q = 0; q < n_queries; ++q) pending[q] = q;
  uint64_t jobs_run = 0;
  std::vector<JobHost> jh;
  std::vector<size_t> who;
  std::vector<float> T, rm;
  std::vector<uint32_t> inl;
  std::vector<int> ok;
  // rank by rank, as GlocEvaluator::global_registraion walks a query's candidates
  // (registration/global_localization.cpp:519-572), but for all pending queries at once
  for (size_t r = 0; r < n_cand && !pending.empty(); ++r) {
    jh.clear();
    who.clear();
    for (size_t q : pending) {
      const size_t o = q * n_cand + r;
      if (cand_scan_ids[o] == 0xFFFFFFFFu) continue;
      JobHost j;
      j.src = src[q];
      GLOC_TRY(store_get(h->store, cand_scan_ids[o], 0, &j.tgt));  // a target's launch order is never read
      j.stream_id = (uint32_t)r;  // the RANSAC stream of retrieval rank r: the same job as in gloc_reg_batch_multi
      j.init_T = init_T ? init_T + 16 * o : nullptr;
      jh.push_back(j);
      who.push_back(q);
    }
    if (jh.empty()) continue;
    const size_t nj = jh.size();
    T.resize(16 * nj);
    rm.resize(nj);
    inl.resize(nj);
    ok.resize(nj);
    GLOC_TRY(run_jobs(h, jh, params, T.data(), rm.data(), inl.data(), ok.data()));
    jobs_run += nj;
    std::vector<size_t> still;
    size_t k = 0;
    for (size_t q : pending) {
      if (k < nj && who[k] == q) {
        if (ok[k]) {
          out_rank[q] = (int)r;
          std::copy(T.begin() + 16 * k, T.begin() + 16 * (k + 1), out_T + 16 * q);
          if (out_rmse) out_rmse[q] = rm[k];
          if (out_inliers) out_inliers[q] = inl[k];
        } else {
          still.push_back(q);
        }
        ++k;
      } else {
        still.push_back(q);  // no candidate at this rank
      }
    }
    pending.swap(still);
  }
  if (out_jobs_run) *out_jobs_run = jobs_run;
  return GLOC_OK;
}

int gloc_reg_batch_ids(gloc_reg* h, uint32_t q_scan_id, const uint32_t* cand_scan_ids,
                       size_t n_cand, const uint32_t* cand_stream_ids, const float* init_T,
                       const gloc_reg_params* params, float* out_T, float* out_rmse,
                       uint32_t* out_inliers, int* out_ok) {
  GLOC_REQUIRE(n_cand >= 1 && n_cand <= 4096, GLOC_ERR_INVALID, "n_cand = %zu outside [1,4096]", n_cand);
  if (cand_scan_ids)
    for (size_t c = 0; c < n_cand; ++c)
      GLOC_REQUIRE(cand_scan_ids[c] != 0xFFFFFFFFu, GLOC_ERR_INVALID, "unknown scan id %u", cand_scan_ids[c]);
  return gloc_reg_batch_multi(h, 1, &q_scan_id, cand_scan_ids, n_cand, cand_stream_ids, init_T, params, out_T,
                              out_rmse, out_inliers, out_ok);
}

int gloc_reg_final_steps(gloc_reg* h, float* out, size_t n) {
  GLOC_REQUIRE(h && (out || !n), GLOC_ERR_INVALID, "null argument");
  GLOC_REQUIRE(n <= h->last_final_step.size(), GLOC_ERR_INVALID, "the last batch had %zu jobs", h->last_final_step.size());
  std::copy(h->last_final_step.begin(), h->last_final_step.begin() + n, out);
  return GLOC_OK;
}

int gloc_reg_select_first_ok(const int* ok, size_t n_cand) {
  if (!ok) return -1;
  for (size_t i = 0; i < n_cand; ++i)
    if (ok[i]) return (int)i;
  return -1;
}

int gloc_reg_nn(gloc_reg* h, const float* src_xyz, size_t n_src, const float* tgt_xyz,
                size_t n_tgt, const float* T16, uint32_t* out_idx, float* out_d2) {
  GLOC_REQUIRE(h && out_idx && out_d2 && (src_xyz || !n_src) && (tgt_xyz || !n_tgt),
               GLOC_ERR_INVALID, "null argument");
  GLOC_REQUIRE(n_src < (1ull << 31) && n_tgt < (1ull << 31), GLOC_ERR_INVALID, "scan too large");
  GLOC_NOT_PENDING(h);
  if (n_src == 0) return GLOC_OK;
  GLOC_HIP(hipSetDevice(h->device));
  GLOC_TRY(ensure_store(h));
  hipStream_t s = h->stream;
  TempScans tmp(h->store);
  GLOC_TRY(tmp.add_all(src_xyz, n_src, h->nn_src_per_lane, &tgt_xyz, &n_tgt, 1, h->temp_target_index));
  const int cs = h->nn_src_per_lane;
  const uint32_t ng = (uint32_t)((n_src + 64 * cs - 1) / (64 * cs));
  BatchDims bd{1, (uint32_t)n_src, ng, std::max<uint32_t>(ng, 1), ((size_t)n_src + 127) & ~(size_t)127};
  auto done = [&](int code) {
    (void)hipStreamSynchronize(s);
    return code;
  };
  if (h->jobs.ensure(sizeof(Job), s) || h->states.ensure(sizeof(CandState), s) ||
      h->corr.ensure(sizeof(uint32_t) * bd.ld, s) || h->d2.ensure(sizeof(float) * bd.ld, s) ||
      h->export_idx.ensure(sizeof(uint32_t) * bd.ld, s) || h->export_d2.ensure(sizeof(float) * bd.ld, s) ||
      h->partials.ensure(sizeof(double) * ACC_NV * bd.n_part, s))
    return done(GLOC_ERR_NOMEM);
  const DevScan &sc = tmp.scans[0], &tg = tmp.scans[1];
  Job jd{sc.idx.pts, sc.order, sc.idx.inv, tg.xyz, tg.idx, (uint32_t)n_src, ng, 0u, 0u};
  CandState st;
  init_state(st, T16);
  if (hipMemcpyAsync(h->jobs.p, &jd, sizeof(jd), hipMemcpyHostToDevice, s) != hipSuccess ||
      hipMemcpyAsync(h->states.p, &st, sizeof(st), hipMemcpyHostToDevice, s) != hipSuccess)
    return done(GLOC_ERR_HIP);
  h->split = NnSplit{};  // (one cold pass: there is no estimate to plan from)
  if (int rc_h = setup_heavy(h, bd, cs)) return done(rc_h);
  WsView v{};
  v.s = s;
  v.n_jobs = 1;
  v.jobs = h->jobs.as<Job>();
  v.states = h->states.as<CandState>();
  v.corr = h->corr.as<uint32_t>();
  v.d2 = h->d2.as<float>();
  v.partials = h->partials.as<double>();
  v.heavy = h->heavy_of[0];
  int rc = launch_nn(h, bd, v, false, false, 0.f);
  if (rc != GLOC_OK) return done(rc);
  hipLaunchKernelGGL(export_corr_kernel, dim3((unsigned)((n_src + 255) / 256), 1), dim3(256), 0, s,
                     h->jobs.as<Job>(), h->corr.as<uint32_t>(), h->d2.as<float>(), bd.ld,
                     h->export_idx.as<uint32_t>(), h->export_d2.as<float>());
  if (hipMemcpyAsync(out_idx, h->export_idx.p, sizeof(uint32_t) * n_src, hipMemcpyDeviceToHost, s) != hipSuccess ||
      hipMemcpyAsync(out_d2, h->export_d2.p, sizeof(float) * n_src, hipMemcpyDeviceToHost, s) != hipSuccess ||
      hipStreamSynchronize(s) != hipSuccess) {
    set_err("gloc_reg_nn: copy back failed: %s", hipGetErrorString(hipGetLastError()));
    return done(GLOC_ERR_HIP);
  }
  return done(GLOC_OK);
}

int gloc_reg_ransac_hypotheses(gloc_reg* h, const float* src_xyz, const float* tgt_xyz,
                               const uint32_t* corr, size_t n, uint64_t seed, uint32_t cand,
                               uint32_t n_hyp, float* out_Rt, uint32_t* out_valid,
                               uint32_t* out_inliers, float inlier_thresh) {
  GLOC_REQUIRE(h && src_xyz && tgt_xyz && corr && out_Rt && out_valid && out_inliers,
               GLOC_ERR_INVALID, "null argument");
  GLOC_REQUIRE(n >= 3 && n < (1ull << 31) && n_hyp >= 1 && n_hyp <= (1u << 20), GLOC_ERR_INVALID,
               "bad sizes");
  GLOC_NOT_PENDING(h);
  GLOC_HIP(hipSetDevice(h->device));
  hipStream_t s = h->stream;
  const size_t ld = (n + 127) & ~(size_t)127;
  // pairs are built on the host from (src, tgt[corr]) -- src is taken as already moved; slots are the
  // caller's indices (no inverse permutation)
  std::vector<float> hp(ld * 8, 0.f);
  for (size_t i = 0; i < n; ++i) {
    for (int a = 0; a < 3; ++a) {
      hp[i * 8 + a] = src_xyz[3 * i + a];
      hp[i * 8 + 4 + a] = tgt_xyz[3 * (size_t)corr[i] + a];
    }
  }
  GLOC_TRY(h->pairs.ensure(sizeof(float) * 8 * ld, s));
  GLOC_TRY(h->jobs.ensure(sizeof(Job), s));
  GLOC_TRY(h->Rt.ensure(sizeof(float) * 12 * (size_t)n_hyp, s));
  GLOC_TRY(h->valid.ensure(sizeof(uint32_t) * (size_t)n_hyp, s));
  GLOC_TRY(h->inliers.ensure(sizeof(uint32_t) * (size_t)n_hyp, s));
  Job jd{};
  jd.n_src = (uint32_t)n;
  jd.cand_id = cand;
  GLOC_HIP(hipMemcpyAsync(h->jobs.p, &jd, sizeof(jd), hipMemcpyHostToDevice, s));
  GLOC_HIP(hipMemcpyAsync(h->pairs.p, hp.data(), sizeof(float) * 8 * ld, hipMemcpyHostToDevice, s));
  GLOC_HIP(hipMemsetAsync(h->inliers.p, 0, sizeof(uint32_t) * (size_t)n_hyp, s));
  hipLaunchKernelGGL(ransac_hyp_kernel, dim3((n_hyp + 127) / 128, 1), dim3(128), 0, s,
                     h->pairs.as<f32x4>(), ld, h->jobs.as<Job>(), seed, n_hyp, 0u, n_hyp,
                     (const CandState*)nullptr, h->Rt.as<float>(), h->valid.as<uint32_t>());
  GLOC_HIP(hipGetLastError());
  dim3 grid((n_hyp + 255) / 256, (unsigned)((n + SC_CHUNK - 1) / SC_CHUNK), 1);
  hipLaunchKernelGGL(ransac_score_kernel, grid, dim3(256), 0, s, h->pairs.as<f32x4>(), ld,
                     h->jobs.as<Job>(), n_hyp, 0u, 256u /* thread <-> hypothesis */, h->Rt.as<float>(),
                     h->valid.as<uint32_t>(), inlier_thresh * inlier_thresh, (const CandState*)nullptr,
                     h->inliers.as<uint32_t>(), (const uint32_t*)nullptr, (const uint32_t*)nullptr, 0u, (uint32_t)SC_CHUNK);
  GLOC_HIP(hipGetLastError());
  GLOC_HIP(hipMemcpyAsync(out_Rt, h->Rt.p, sizeof(float) * 12 * (size_t)n_hyp,
                          hipMemcpyDeviceToHost, s));
  GLOC_HIP(hipMemcpyAsync(out_valid, h->valid.p, sizeof(uint32_t) * (size_t)n_hyp,
                          hipMemcpyDeviceToHost, s));
  GLOC_HIP(hipMemcpyAsync(out_inliers, h->inliers.p, sizeof(uint32_t) * (size_t)n_hyp,
                          hipMemcpyDeviceToHost, s));
  GLOC_HIP(hipStreamSynchronize(s));
  return GLOC_OK;
}

int gloc_reg_nn_stats(gloc_reg* h, uint64_t* pairs_evaluated, uint64_t* launches) {
  GLOC_REQUIRE(h, GLOC_ERR_INVALID, "null handle");
  GLOC_HIP(hipSetDevice(h->device));
  unsigned long long c = 0;
  if (h->counters.p) {
    std::vector<unsigned long long> part(NN_STAT_SLOTS);
    GLOC_HIP(hipMemcpyAsync(part.data(), h->counters.p, 8 * NN_STAT_SLOTS, hipMemcpyDeviceToHost, h->stream));
    GLOC_HIP(hipStreamSynchronize(h->stream));
    for (unsigned long long v : part) c += v;
  }
  if (pairs_evaluated) *pairs_evaluated = c;
  if (launches) *launches = h->nn_launches;
  return GLOC_OK;
}

// Developer / test aid (not part of include/gloc3d.h): the correspondences of the LAST 1-NN pass of the
// last batch for job `job`, in the caller's index space (original source index -> original target index).
int gloc_reg_debug_corr(gloc_reg* h, uint32_t job, uint32_t n_src, uint32_t* out_idx, float* out_d2) {
  GLOC_REQUIRE(h && out_idx && out_d2, GLOC_ERR_INVALID, "null argument");
  GLOC_HIP(hipSetDevice(h->device));
  hipStream_t s = h->stream;
  const size_t ld = h->last_ld;
  GLOC_NOT_PENDING(h);
  GLOC_REQUIRE(job < h->last_jobs && n_src <= ld, GLOC_ERR_INVALID, "no such job in the last batch");
  GLOC_TRY(h->export_idx.ensure(sizeof(uint32_t) * ld * h->last_jobs, s));
  GLOC_TRY(h->export_d2.ensure(sizeof(float) * ld * h->last_jobs, s));
  hipLaunchKernelGGL(export_corr_kernel, dim3((unsigned)((ld + 255) / 256), h->last_jobs), dim3(256), 0, s,
                     h->jobs.as<Job>(), h->corr.as<uint32_t>(), h->d2.as<float>(), ld,
                     h->export_idx.as<uint32_t>(), h->export_d2.as<float>());
  GLOC_HIP(hipGetLastError());
  GLOC_HIP(hipMemcpyAsync(out_idx, h->export_idx.as<uint32_t>() + (size_t)job * ld, sizeof(uint32_t) * n_src,
                          hipMemcpyDeviceToHost, s));
  GLOC_HIP(hipMemcpyAsync(out_d2, h->export_d2.as<float>() + (size_t)job * ld, sizeof(float) * n_src,
                          hipMemcpyDeviceToHost, s));
  GLOC_HIP(hipStreamSynchronize(s));
  return GLOC_OK;
}

// Developer / test aid (not part of include/gloc3d.h): chained launches enqueued and chained launches that timed out.
int gloc_reg_debug_chain(gloc_reg* h, uint64_t* launches, uint64_t* timeouts) {
  GLOC_REQUIRE(h, GLOC_ERR_INVALID, "null handle");
  if (launches) *launches = h->chain_launches;
  if (timeouts) *timeouts = h->chain_timeouts;
  return GLOC_OK;
}

// Developer aid (not part of include/gloc3d.h): stamps of the last chained launch, [pass][job][16] words of the 100 MHz clock
// (tools/dev_chain_trace.py): 0 first search wave arrives, 1 first one past its wait, 2 last one leaves, 3 reducer 0 starts
// to wait, 4 sees the pass done, 5 has stored its sub-sum, 6 last reducer in, 7 sums loaded, 8 solved.
int gloc_reg_debug_chain_trace(gloc_reg* h, int enable, uint32_t* out, size_t cap_words, uint32_t* n_pass, uint32_t* n_jobs) {
  GLOC_REQUIRE(h, GLOC_ERR_INVALID, "null handle");
  GLOC_HIP(hipSetDevice(h->device));
  h->chain_trace = enable != 0;
  if (n_pass) *n_pass = h->chain_dbg_pass;
  if (n_jobs) *n_jobs = h->chain_dbg_jobs;
  if (out && h->chain_dbg.p) {
    const size_t n = std::min(cap_words, (size_t)h->chain_dbg_pass * h->chain_dbg_jobs * 16);
    GLOC_HIP(hipStreamSynchronize(h->stream));
    GLOC_HIP(hipMemcpy(out, h->chain_dbg.p, n * 4, hipMemcpyDeviceToHost));
  }
  return GLOC_OK;
}

// Test aid (not part of include/gloc3d.h): the adaptive stop's iteration count as the DEVICE computes it, for `count`
// (inliers, points) pairs -- compared with the oracle's loop in tests/test_reg_gpu.py.
int gloc_reg_debug_needed_iters(gloc_reg* h, const uint32_t* inl, const uint32_t* n, uint32_t count, float conf, uint32_t max_iters,
                                uint32_t* out) {
  GLOC_REQUIRE(h && inl && n && out, GLOC_ERR_INVALID, "null argument");
  GLOC_HIP(hipSetDevice(h->device));
  GLOC_NOT_PENDING(h);
  hipStream_t s = h->stream;
  GLOC_TRY(h->export_idx.ensure(sizeof(uint32_t) * 3 * (size_t)count, s));
  uint32_t* d = h->export_idx.as<uint32_t>();
  GLOC_HIP(hipMemcpyAsync(d, inl, 4 * (size_t)count, hipMemcpyHostToDevice, s));
  GLOC_HIP(hipMemcpyAsync(d + count, n, 4 * (size_t)count, hipMemcpyHostToDevice, s));
  hipLaunchKernelGGL(needed_iters_kernel, dim3((count + 255) / 256), dim3(256), 0, s, d, d + count, conf, max_iters, d + 2 * (size_t)count, count);
  GLOC_HIP(hipGetLastError());
  GLOC_HIP(hipMemcpyAsync(out, d + 2 * (size_t)count, 4 * (size_t)count, hipMemcpyDeviceToHost, s));
  GLOC_HIP(hipStreamSynchronize(s));
  return GLOC_OK;
}

// Test aid (not part of include/gloc3d.h): the next chained launches wait for a wave that never comes -- every wait runs
// out, the launch ends by itself, the batch is run again launch by launch (tests/test_reg_gpu.py: the bounded waits).
int gloc_reg_debug_chain_stall(gloc_reg* h, int on) {
  GLOC_REQUIRE(h, GLOC_ERR_INVALID, "null handle");
  h->chain_stall = on != 0;
  return GLOC_OK;
}

// Developer aid (not part of include/gloc3d.h): per-wave trace of the LAST culled 1-NN launch.
int gloc_reg_debug_trace(gloc_reg* h, int enable, uint32_t* out, size_t cap_waves, size_t* n_waves) {
  GLOC_REQUIRE(h, GLOC_ERR_INVALID, "null handle");
  GLOC_HIP(hipSetDevice(h->device));
  h->trace_on = enable != 0;
  if (n_waves) *n_waves = h->trace_waves;
  if (out && h->trace.p) {
    const size_t n = std::min(cap_waves, h->trace_waves);
    GLOC_HIP(hipMemcpyAsync(out, h->trace.p, n * 4 * NN_TRACE_WORDS, hipMemcpyDeviceToHost, h->stream));
    GLOC_HIP(hipStreamSynchronize(h->stream));
  }
  return GLOC_OK;
}

int gloc_reg_profile(gloc_reg* h, const char* kernel, double* total_ms, uint64_t* launches) {
  GLOC_REQUIRE(h && kernel, GLOC_ERR_INVALID, "null argument");
  GLOC_HIP(hipSetDevice(h->device));
  GLOC_TRY(h->prof.collect(h->stream));
  auto it = h->prof.fam.find(kernel);
  if (total_ms) *total_ms = it == h->prof.fam.end() ? 0.0 : it->second.total_ms;
  if (launches) *launches = it == h->prof.fam.end() ? 0 : it->second.launches;
  return GLOC_OK;
}

int gloc_reg_profile_reset(gloc_reg* h) {
  GLOC_REQUIRE(h, GLOC_ERR_INVALID, "null handle");
  GLOC_HIP(hipSetDevice(h->device));
  GLOC_HIP(hipStreamSynchronize(h->stream));
  h->prof.reset();
  h->nn_launches = 0;
  if (h->counters.p) GLOC_HIP(hipMemsetAsync(h->counters.p, 0, 8 * NN_STAT_SLOTS, h->stream));
  return GLOC_OK;
}

}  // extern "C"
