// coarse.hip -- C ABI of the coarse global (x, y, yaw) match on BEV occupancy grids (include/gloc3d.h;
// SURVEY.md 8a row a-12).  Replaces RpyPCLoopDetector::match(q_grid, db_idx, xy_yaw, scale)
// (registration/loop_detector.cpp:186-288): same inputs (two occupancy grids), same output convention
// (p_db = R(yaw) p_q + (x, y); scale is 1 by construction), found by the exhaustive search described in
// coarse_kernels.hpp instead of SURF + FLANN + RANSAC.
#include <algorithm>
#include <cmath>
#include <map>
#include <mutex>
#include <new>
#include <vector>

#include "coarse_kernels.hpp"
#include "common.hpp"
#include "scan_store.hpp"

using namespace gloc;
using namespace gloc::coarse;

struct gloc_coarse {
  int device = 0;
  hipStream_t own_stream = nullptr, stream = nullptr;
  std::vector<void*> blocks;      // one allocation per grid (null: released)
  std::vector<GridDev> grids;     // host copies of the device views
  std::vector<uint32_t> counts;   // occupied cells per grid
  std::vector<uint32_t> cell_px;  // the geometry each grid was built with: a match with other parameters
  std::vector<float> res;         // would silently return a wrong (x, y, yaw), so it is rejected
  std::vector<uint32_t> free_ids;
  std::vector<size_t> block_words;                    // size class of each grid's allocation
  size_t cached_bytes = 0;                            // ... bytes parked in free_blocks (capped: COARSE_CACHE_BYTES)
  std::map<size_t, std::vector<void*>> free_blocks;   // released allocations by size class: a stream of queries
                                                      // adds and releases 25 grids per step -- no hipMalloc / hipFree
                                                      // (both synchronise the device) once the classes are warm
  std::vector<uint32_t> h_bits;                       // host copy of the batch's bit maps (counted on the host)
  DevBuf scratch_bits, scratch_cnt, stage_img, tiny_img;
  DevBuf d_grids, d_pq, d_pd, d_trig, d_yaw, d_yawout, d_cand, d_verify, d_out, d_scale;
  bool grids_dirty = true;
  uint32_t trig_n = 0;
  gloc_bev* bev = nullptr;        // created on first add_scan
};

namespace {

int check_params(const gloc_coarse_params* p) {
  GLOC_REQUIRE(p, GLOC_ERR_INVALID, "params is NULL");
  GLOC_REQUIRE(p->resolution > 0.f && std::isfinite(p->resolution), GLOC_ERR_INVALID, "resolution must be positive");
  GLOC_REQUIRE(p->cell_px >= 1 && p->cell_px <= 16, GLOC_ERR_INVALID, "cell_px = %u outside [1,16]", p->cell_px);
  GLOC_REQUIRE(p->n_yaw >= 1 && p->n_yaw <= 3600, GLOC_ERR_INVALID, "n_yaw = %u outside [1,3600]", p->n_yaw);
  GLOC_REQUIRE(p->max_shift <= 255, GLOC_ERR_INVALID, "max_shift = %u outside [0,255]", p->max_shift);
  GLOC_REQUIRE(p->top_yaw <= 64 && p->top_yaw <= p->n_yaw, GLOC_ERR_INVALID, "top_yaw = %u outside [0,min(64,n_yaw)]",
               p->top_yaw);
  GLOC_REQUIRE(p->refine <= 8, GLOC_ERR_INVALID, "refine = %u outside [0,8]", p->refine);
  return GLOC_OK;
}

// bit maps in h->scratch_bits ([n][G * GW]) -> n new grids.  Two synchronisations for the whole batch (the counts
// come to the host to size the cell lists; the grids are complete when the call returns) and no allocation once the
// size classes are warm: a grid made alone used to cost three synchronisations and a hipMalloc, 2.3 ms in a busy
// stream against 0.25 ms of device work.
constexpr size_t CELL_CLASS = 4096;  // cell lists are sized in steps of 4096 cells
constexpr size_t COARSE_CACHE_BYTES = 256u << 20;  // released grid blocks kept for reuse, at most

int take_block(gloc_coarse* h, size_t words, void** blk) {
  auto it = h->free_blocks.find(words);
  if (it != h->free_blocks.end() && !it->second.empty()) {
    *blk = it->second.back();
    it->second.pop_back();
    h->cached_bytes -= sizeof(uint32_t) * words;
    return GLOC_OK;
  }
  hipError_t e = hipMalloc(blk, sizeof(uint32_t) * words);
  if (e == hipErrorOutOfMemory) {  // drop what the pool holds for other size classes, then try once more
    (void)hipGetLastError();
    for (auto& kv : h->free_blocks) {
      for (void* b : kv.second) (void)hipFree(b);
      kv.second.clear();
    }
    h->cached_bytes = 0;
    e = hipMalloc(blk, sizeof(uint32_t) * words);
  }
  if (e != hipSuccess) {
    (void)hipGetLastError();
    *blk = nullptr;
    set_err("hipMalloc of %zu bytes for a coarse grid failed: %s", sizeof(uint32_t) * words, hipGetErrorString(e));
    return e == hipErrorOutOfMemory ? GLOC_ERR_NOMEM : GLOC_ERR_HIP;
  }
  return GLOC_OK;
}

int finish_grids(gloc_coarse* h, const gloc_coarse_params* prm, size_t n_grids, uint32_t* grid_ids) {
  hipStream_t s = h->stream;
  h->h_bits.resize(n_grids * G * GW);
  GLOC_HIP(hipMemcpyAsync(h->h_bits.data(), h->scratch_bits.p, sizeof(uint32_t) * G * GW * n_grids, hipMemcpyDeviceToHost, s));
  GLOC_HIP(hipStreamSynchronize(s));
  std::vector<void*> blks(n_grids, nullptr);
  std::vector<GridDev> gs(n_grids);
  std::vector<uint32_t> cnt(n_grids, 0);
  std::vector<size_t> words(n_grids, 0);
  int rc = GLOC_OK;
  for (size_t i = 0; i < n_grids && rc == GLOC_OK; ++i) {
    uint32_t n = 0;
    for (size_t w = 0; w < (size_t)G * GW; ++w) n += (uint32_t)__builtin_popcount(h->h_bits[i * G * GW + w]);
    cnt[i] = n;
    const size_t cap = (std::max<size_t>(n, 1) + CELL_CLASS - 1) / CELL_CLASS * CELL_CLASS;
    words[i] = (size_t)2 * G * GW + 2 * G + cap + 4;
    rc = take_block(h, words[i], &blks[i]);
    if (rc != GLOC_OK) break;
    uint32_t* p = reinterpret_cast<uint32_t*>(blks[i]);
    GridDev& g = gs[i];
    g.bits = p;
    g.dil = g.bits + G * GW;
    g.hx = g.dil + G * GW;
    g.hy = g.hx + G;
    g.cells = g.hy + G;
    g.count = g.cells + cap;
    hipError_t e = hipMemcpyAsync(g.bits, h->scratch_bits.as<uint32_t>() + i * G * GW, sizeof(uint32_t) * G * GW,
                                  hipMemcpyDeviceToDevice, s);
    if (e == hipSuccess) e = hipMemsetAsync(g.count, 0, sizeof(uint32_t), s);
    if (e == hipSuccess) {
      hipLaunchKernelGGL(finish_grid_kernel, dim3(1), dim3(G), 0, s, g, (uint32_t)cap);
      e = hipGetLastError();  // (read once: the call clears the error)
    }
    if (e != hipSuccess) {
      set_err("coarse grid setup failed: %s", hipGetErrorString(e));
      rc = GLOC_ERR_HIP;
    }
  }
  if (rc == GLOC_OK) {
    const hipError_t e = hipStreamSynchronize(s);
    if (e != hipSuccess) {
      set_err("coarse grid kernel failed: %s", hipGetErrorString(e));
      rc = GLOC_ERR_HIP;
    }
  } else {
    (void)hipStreamSynchronize(s);
  }
  if (rc != GLOC_OK) {
    for (size_t i = 0; i < n_grids; ++i)
      if (blks[i]) {
        h->free_blocks[words[i]].push_back(blks[i]);
        h->cached_bytes += sizeof(uint32_t) * words[i];
      }
    return rc;
  }
  for (size_t i = 0; i < n_grids; ++i) {
    uint32_t id;
    if (!h->free_ids.empty()) {
      id = h->free_ids.back();
      h->free_ids.pop_back();
      h->blocks[id] = blks[i];
      h->grids[id] = gs[i];
      h->counts[id] = cnt[i];
      h->cell_px[id] = prm->cell_px;
      h->res[id] = prm->resolution;
      h->block_words[id] = words[i];
    } else {
      id = (uint32_t)h->grids.size();
      h->blocks.push_back(blks[i]);
      h->grids.push_back(gs[i]);
      h->counts.push_back(cnt[i]);
      h->cell_px.push_back(prm->cell_px);
      h->res.push_back(prm->resolution);
      h->block_words.push_back(words[i]);
    }
    grid_ids[i] = id;
  }
  h->grids_dirty = true;
  return GLOC_OK;
}

int finish_grid(gloc_coarse* h, const gloc_coarse_params* prm, uint32_t* grid_id) { return finish_grids(h, prm, 1, grid_id); }

// the bit maps of n scans of a store -> h->scratch_bits, nothing synchronised
int mark_store_scans(gloc_coarse* h, gloc_scan_store* store, const uint32_t* scan_ids, size_t n,
                     const gloc_coarse_params* params) {
  hipStream_t s = h->stream;
  if (!h->bev) {
    GLOC_TRY(gloc_bev_create(h->device, &h->bev));
    GLOC_TRY(gloc_bev_set_stream(h->bev, (void*)s));
  }
  gloc_bev_params bp;
  gloc_bev_default_params(&bp);
  bp.resolution = params->resolution;
  bp.out_width = 4;
  bp.out_height = 4;
  GLOC_TRY(h->tiny_img.ensure(64, s));
  GLOC_TRY(h->scratch_bits.ensure(sizeof(uint32_t) * G * GW * n, s));
  GLOC_HIP(hipMemsetAsync(h->scratch_bits.p, 0, sizeof(uint32_t) * G * GW * n, s));
  for (size_t i = 0; i < n; ++i) {
    DevScan sc;
    GLOC_TRY(gloc::reg::store_get(store, scan_ids[i], 0, &sc));  // only the points are read
    const uint64_t offsets[2] = {0, (uint64_t)sc.n};
    // (no info requested: the projection does not synchronise; a scan that projects to nothing leaves cleared flags)
    GLOC_TRY(gloc_bev_project_batch_device(h->bev, sc.xyz, offsets, 1, 3, &bp, h->tiny_img.p, nullptr));
    const uint8_t* d_flags = nullptr;
    int R = 0, S = 0;
    GLOC_TRY(gloc_bev_device_flags(h->bev, 0, &d_flags, &R, &S));
    const size_t px = (size_t)S * S;
    hipLaunchKernelGGL(mark_from_flags_kernel, dim3((unsigned)((px + 255) / 256)), dim3(256), 0, s, d_flags, R, S,
                       (int)params->cell_px, h->scratch_bits.as<uint32_t>() + i * G * GW);
    GLOC_HIP(hipGetLastError());
  }
  return GLOC_OK;
}

}  // namespace

extern "C" {

int gloc_coarse_default_params(gloc_coarse_params* p) {
  GLOC_REQUIRE(p, GLOC_ERR_INVALID, "params is NULL");
  p->resolution = 0.2f;   // the BEV pixel (loop_detector.h:116)
  p->cell_px = 2;         // search cell = 2 x 2 pixels = 0.4 m
  p->n_yaw = 360;         // 1-degree steps: 0.9 m at 50 m, inside the ICP basin
  p->max_shift = 64;      // cells: +-25.6 m
  p->top_yaw = 12;
  p->refine = 4;          // cells around the projection optimum
  p->min_overlap = 0.25f;
  p->reserved_ = 0;
  return GLOC_OK;
}

int gloc_coarse_create(int device, gloc_coarse** out) {
  GLOC_REQUIRE(out, GLOC_ERR_INVALID, "out is null");
  *out = nullptr;
  GLOC_TRY(select_device(device));
  gloc_coarse* h = new (std::nothrow) gloc_coarse;
  GLOC_REQUIRE(h, GLOC_ERR_NOMEM, "host allocation failed");
  h->device = device;
  hipError_t e = hipStreamCreateWithFlags(&h->own_stream, hipStreamNonBlocking);
  if (e != hipSuccess) {
    set_err("hipStreamCreate failed: %s", hipGetErrorString(e));
    delete h;
    return GLOC_ERR_HIP;
  }
  h->stream = h->own_stream;
  *out = h;
  return GLOC_OK;
}

int gloc_coarse_destroy(gloc_coarse* h) {
  if (!h) return GLOC_OK;
  (void)hipSetDevice(h->device);
  (void)hipStreamSynchronize(h->stream);
  for (void* b : h->blocks)
    if (b) (void)hipFree(b);
  for (auto& kv : h->free_blocks)
    for (void* b : kv.second) (void)hipFree(b);
  for (DevBuf* b : {&h->scratch_bits, &h->scratch_cnt, &h->stage_img, &h->tiny_img, &h->d_grids, &h->d_pq, &h->d_pd, &h->d_trig,
                    &h->d_yaw, &h->d_yawout, &h->d_cand, &h->d_verify, &h->d_out, &h->d_scale})
    b->release();
  if (h->bev) (void)gloc_bev_destroy(h->bev);
  if (h->own_stream) (void)hipStreamDestroy(h->own_stream);
  delete h;
  return GLOC_OK;
}

int gloc_coarse_add_image(gloc_coarse* h, const uint8_t* occupancy, uint32_t width, uint32_t height, float ox,
                          float oy, float resolution, const gloc_coarse_params* params, uint32_t* grid_id) {
  GLOC_REQUIRE(h && grid_id && (occupancy || !width || !height), GLOC_ERR_INVALID, "null argument");
  GLOC_REQUIRE(width <= 16384 && height <= 16384 && resolution > 0.f, GLOC_ERR_INVALID, "bad image geometry");
  GLOC_TRY(check_params(params));
  GLOC_REQUIRE(std::fabs(resolution - params->resolution) <= 1e-6f * params->resolution, GLOC_ERR_INVALID,
               "image resolution %g differs from the matcher's %g", (double)resolution, (double)params->resolution);
  // the image's first pixel is voxel (ox / res, oy / res) (xy_res = min index * resolution, loop_detector.cpp:133)
  const int ix0 = (int)std::lround((double)ox / (double)resolution), iy0 = (int)std::lround((double)oy / (double)resolution);
  GLOC_HIP(hipSetDevice(h->device));
  hipStream_t s = h->stream;
  const size_t px = (size_t)width * height;
  GLOC_TRY(h->scratch_bits.ensure(sizeof(uint32_t) * G * GW, s));
  GLOC_HIP(hipMemsetAsync(h->scratch_bits.p, 0, sizeof(uint32_t) * G * GW, s));
  if (px) {
    GLOC_TRY(h->stage_img.ensure(px, s));
    GLOC_HIP(hipMemcpyAsync(h->stage_img.p, occupancy, px, hipMemcpyHostToDevice, s));
    hipLaunchKernelGGL(mark_from_image_kernel, dim3((unsigned)((px + 255) / 256)), dim3(256), 0, s,
                       h->stage_img.as<uint8_t>(), (int)width, (int)height, ix0, iy0, (int)params->cell_px,
                       h->scratch_bits.as<uint32_t>());
    GLOC_HIP(hipGetLastError());
  }
  return finish_grid(h, params, grid_id);
}

int gloc_coarse_add_scan(gloc_coarse* h, const float* xyz, size_t n, size_t stride_floats,
                         const gloc_coarse_params* params, uint32_t* grid_id) {
  GLOC_REQUIRE(h && grid_id && (xyz || n == 0), GLOC_ERR_INVALID, "null argument");
  GLOC_TRY(check_params(params));
  GLOC_HIP(hipSetDevice(h->device));
  hipStream_t s = h->stream;
  if (!h->bev) {
    GLOC_TRY(gloc_bev_create(h->device, &h->bev));
    GLOC_TRY(gloc_bev_set_stream(h->bev, (void*)s));
  }
  gloc_bev_params bp;
  gloc_bev_default_params(&bp);  // 100 m: the reference's occupancy image (loop_detector.h:115-116)
  bp.resolution = params->resolution;
  bp.out_width = 4;
  bp.out_height = 4;
  uint8_t tiny[4 * 4 * 3];
  gloc_bev_info info;
  GLOC_TRY(gloc_bev_project(h->bev, xyz, n, stride_floats, &bp, tiny, &info));
  const uint8_t* d_flags = nullptr;
  int R = 0, S = 0;
  GLOC_TRY(gloc_bev_device_flags(h->bev, 0, &d_flags, &R, &S));
  GLOC_TRY(h->scratch_bits.ensure(sizeof(uint32_t) * G * GW, s));
  GLOC_HIP(hipMemsetAsync(h->scratch_bits.p, 0, sizeof(uint32_t) * G * GW, s));
  if (!info.empty) {
    const size_t px = (size_t)S * S;
    hipLaunchKernelGGL(mark_from_flags_kernel, dim3((unsigned)((px + 255) / 256)), dim3(256), 0, s, d_flags, R, S,
                       (int)params->cell_px, h->scratch_bits.as<uint32_t>());
    GLOC_HIP(hipGetLastError());
  }
  return finish_grid(h, params, grid_id);
}

int gloc_coarse_release(gloc_coarse* h, uint32_t grid_id) {
  GLOC_REQUIRE(h, GLOC_ERR_INVALID, "null handle");
  GLOC_REQUIRE(grid_id < h->blocks.size() && h->blocks[grid_id], GLOC_ERR_INVALID, "unknown grid id %u", grid_id);
  GLOC_HIP(hipSetDevice(h->device));
  GLOC_HIP(hipStreamSynchronize(h->stream));  // (matches are synchronous: nothing of this handle still reads it)
  // parked for the next grid of its size class -- up to COARSE_CACHE_BYTES in all (a long stream of grids of varying
  // cell counts would otherwise pile up device memory that no other allocator can see); beyond that, freed
  const size_t bytes = sizeof(uint32_t) * h->block_words[grid_id];
  if (h->cached_bytes + bytes <= COARSE_CACHE_BYTES) {
    h->free_blocks[h->block_words[grid_id]].push_back(h->blocks[grid_id]);
    h->cached_bytes += bytes;
  } else {
    (void)hipFree(h->blocks[grid_id]);
  }
  h->blocks[grid_id] = nullptr;
  h->counts[grid_id] = 0;
  h->free_ids.push_back(grid_id);
  return GLOC_OK;
}

int gloc_coarse_cells(gloc_coarse* h, uint32_t grid_id, uint32_t* n_cells, uint32_t* out_cells, size_t capacity) {
  GLOC_REQUIRE(h && n_cells, GLOC_ERR_INVALID, "null argument");
  GLOC_REQUIRE(grid_id < h->blocks.size() && h->blocks[grid_id], GLOC_ERR_INVALID, "unknown grid id %u", grid_id);
  GLOC_HIP(hipSetDevice(h->device));
  *n_cells = h->counts[grid_id];
  if (out_cells) {
    GLOC_REQUIRE(capacity >= h->counts[grid_id], GLOC_ERR_INVALID, "buffer holds %zu cells, the grid has %u", capacity,
                 h->counts[grid_id]);
    if (h->counts[grid_id]) {
      GLOC_HIP(hipMemcpyAsync(out_cells, h->grids[grid_id].cells, sizeof(uint32_t) * h->counts[grid_id],
                              hipMemcpyDeviceToHost, h->stream));
      GLOC_HIP(hipStreamSynchronize(h->stream));
    }
  }
  return GLOC_OK;
}

int gloc_coarse_add_store_scans(gloc_coarse* h, gloc_scan_store* store, const uint32_t* scan_ids, size_t n,
                                const gloc_coarse_params* params, uint32_t* grid_ids) {
  GLOC_REQUIRE(h && store && (n == 0 || (scan_ids && grid_ids)), GLOC_ERR_INVALID, "null argument");
  GLOC_REQUIRE(store->device == h->device, GLOC_ERR_INVALID, "store on device %d, matcher on %d", store->device,
               h->device);
  GLOC_REQUIRE(n <= 4096, GLOC_ERR_INVALID, "at most 4096 scans per call (%zu)", n);
  GLOC_TRY(check_params(params));
  if (!n) return GLOC_OK;
  GLOC_HIP(hipSetDevice(h->device));
  GLOC_TRY(mark_store_scans(h, store, scan_ids, n, params));
  return finish_grids(h, params, n, grid_ids);
}

int gloc_coarse_add_store_scan(gloc_coarse* h, gloc_scan_store* store, uint32_t scan_id,
                               const gloc_coarse_params* params, uint32_t* grid_id) {
  GLOC_REQUIRE(grid_id, GLOC_ERR_INVALID, "null argument");
  return gloc_coarse_add_store_scans(h, store, &scan_id, 1, params, grid_id);
}

int gloc_coarse_match(gloc_coarse* h, uint32_t q_grid, const uint32_t* db_grids, size_t n_db,
                      const gloc_coarse_params* params, float* out_xy_yaw, float* out_ratio, int* out_ok,
                      float* out_scale) {
  GLOC_REQUIRE(h && db_grids, GLOC_ERR_INVALID, "null argument");
  GLOC_REQUIRE(n_db >= 1 && n_db <= 65536, GLOC_ERR_INVALID, "n_db = %zu outside [1,65536]", n_db);
  std::vector<uint32_t> q(n_db, q_grid);
  return gloc_coarse_match_pairs(h, q.data(), db_grids, n_db, params, out_xy_yaw, out_ratio, out_ok, out_scale);
}

int gloc_coarse_match_pairs(gloc_coarse* h, const uint32_t* q_grids, const uint32_t* db_grids, size_t n_pairs_in,
                            const gloc_coarse_params* params, float* out_xy_yaw, float* out_ratio, int* out_ok,
                            float* out_scale) {
  GLOC_REQUIRE(h && q_grids && db_grids && out_xy_yaw, GLOC_ERR_INVALID, "null argument");
  GLOC_REQUIRE(n_pairs_in >= 1 && n_pairs_in <= 65536, GLOC_ERR_INVALID, "n_pairs = %zu outside [1,65536]", n_pairs_in);
  GLOC_TRY(check_params(params));
  for (size_t i = 0; i < n_pairs_in; ++i) {
    GLOC_REQUIRE(q_grids[i] < h->blocks.size() && h->blocks[q_grids[i]], GLOC_ERR_INVALID, "unknown grid id %u",
                 q_grids[i]);
    GLOC_REQUIRE(db_grids[i] < h->blocks.size() && h->blocks[db_grids[i]], GLOC_ERR_INVALID, "unknown grid id %u",
                 db_grids[i]);
    for (uint32_t g : {q_grids[i], db_grids[i]})
      GLOC_REQUIRE(h->cell_px[g] == params->cell_px && h->res[g] == params->resolution, GLOC_ERR_INVALID,
                   "grid %u was built with cell_px %u / resolution %g, the match asks for %u / %g", g, h->cell_px[g],
                   (double)h->res[g], params->cell_px, (double)params->resolution);
  }
  GLOC_HIP(hipSetDevice(h->device));
  hipStream_t s = h->stream;
  const uint32_t n_pairs = (uint32_t)n_pairs_in, n_yaw = params->n_yaw, n_cand = params->top_yaw + 1;
  if (h->grids_dirty) {
    GLOC_TRY(h->d_grids.ensure(sizeof(GridDev) * h->grids.size(), s));
    GLOC_HIP(hipMemcpyAsync(h->d_grids.p, h->grids.data(), sizeof(GridDev) * h->grids.size(), hipMemcpyHostToDevice, s));
    h->grids_dirty = false;
  }
  if (h->trig_n != n_yaw) {
    // cos / sin in fp64 on the host, rounded once to fp32: oracle/coarse_oracle.c makes the same table
    std::vector<float> trig(2 * (size_t)n_yaw), yaw(n_yaw);
    for (uint32_t k = 0; k < n_yaw; ++k) {
      const double a = 2.0 * M_PI * (double)k / (double)n_yaw;
      trig[2 * k] = (float)std::cos(a);
      trig[2 * k + 1] = (float)std::sin(a);
      yaw[k] = (float)(a > M_PI ? a - 2.0 * M_PI : a);
    }
    GLOC_TRY(h->d_trig.ensure(sizeof(float) * 2 * n_yaw, s));
    GLOC_TRY(h->d_yaw.ensure(sizeof(float) * n_yaw, s));
    GLOC_HIP(hipMemcpyAsync(h->d_trig.p, trig.data(), sizeof(float) * 2 * n_yaw, hipMemcpyHostToDevice, s));
    GLOC_HIP(hipMemcpyAsync(h->d_yaw.p, yaw.data(), sizeof(float) * n_yaw, hipMemcpyHostToDevice, s));
    GLOC_HIP(hipStreamSynchronize(s));  // the vectors go out of scope
    h->trig_n = n_yaw;
  }
  std::vector<uint32_t> pq(q_grids, q_grids + n_pairs), pd(db_grids, db_grids + n_pairs);
  GLOC_TRY(h->d_pq.ensure(sizeof(uint32_t) * n_pairs, s));
  GLOC_TRY(h->d_pd.ensure(sizeof(uint32_t) * n_pairs, s));
  GLOC_TRY(h->d_yawout.ensure(sizeof(YawOut) * (size_t)n_pairs * n_yaw, s));
  GLOC_TRY(h->d_cand.ensure(sizeof(uint32_t) * 3 * (size_t)n_pairs * n_cand, s));
  GLOC_TRY(h->d_verify.ensure(sizeof(VerifyOut) * (size_t)n_pairs * n_cand, s));
  GLOC_TRY(h->d_out.ensure(sizeof(MatchOut) * n_pairs, s));
  GLOC_HIP(hipMemcpyAsync(h->d_pq.p, pq.data(), sizeof(uint32_t) * n_pairs, hipMemcpyHostToDevice, s));
  GLOC_HIP(hipMemcpyAsync(h->d_pd.p, pd.data(), sizeof(uint32_t) * n_pairs, hipMemcpyHostToDevice, s));
  const GridDev* dg = h->d_grids.as<GridDev>();
  hipLaunchKernelGGL(yaw_kernel, dim3(n_yaw, n_pairs), dim3(256), 0, s, dg, dg, h->d_pq.as<uint32_t>(),
                     h->d_pd.as<uint32_t>(), h->d_trig.as<float>(), (int)params->cell_px, (int)params->max_shift, n_yaw,
                     h->d_yawout.as<YawOut>());
  hipLaunchKernelGGL(top_kernel, dim3(n_pairs), dim3(64), 0, s, h->d_yawout.as<YawOut>(), n_yaw, params->top_yaw,
                     h->d_cand.as<uint32_t>());
  hipLaunchKernelGGL(verify_kernel, dim3(n_cand, n_pairs), dim3(256), 0, s, dg, dg, h->d_pq.as<uint32_t>(),
                     h->d_pd.as<uint32_t>(), h->d_trig.as<float>(), (int)params->cell_px, (int)params->refine, n_cand,
                     h->d_cand.as<uint32_t>(), h->d_verify.as<VerifyOut>());
  hipLaunchKernelGGL(final_kernel, dim3((n_pairs + 63) / 64), dim3(64), 0, s, h->d_verify.as<VerifyOut>(), dg,
                     h->d_pq.as<uint32_t>(), n_cand, n_pairs, n_yaw, (float)params->cell_px * params->resolution,
                     params->min_overlap,
                     h->d_yaw.as<float>(), h->d_out.as<MatchOut>());
  GLOC_TRY(h->d_scale.ensure(sizeof(uint32_t) * N_SCALES * (size_t)n_pairs, s));
  hipLaunchKernelGGL(scale_kernel, dim3(N_SCALES, n_pairs), dim3(256), 0, s, dg, dg, h->d_pq.as<uint32_t>(),
                     h->d_pd.as<uint32_t>(), h->d_trig.as<float>(), (int)params->cell_px, (int)params->refine,
                     h->d_out.as<MatchOut>(), h->d_scale.as<uint32_t>());
  hipLaunchKernelGGL(scale_final_kernel, dim3((n_pairs + 63) / 64), dim3(64), 0, s, h->d_scale.as<uint32_t>(), n_pairs,
                     h->d_out.as<MatchOut>());
  GLOC_HIP(hipGetLastError());
  std::vector<MatchOut> mo(n_pairs);
  GLOC_HIP(hipMemcpyAsync(mo.data(), h->d_out.p, sizeof(MatchOut) * n_pairs, hipMemcpyDeviceToHost, s));
  GLOC_HIP(hipStreamSynchronize(s));
  for (uint32_t i = 0; i < n_pairs; ++i) {
    out_xy_yaw[3 * i + 0] = mo[i].x;
    out_xy_yaw[3 * i + 1] = mo[i].y;
    out_xy_yaw[3 * i + 2] = mo[i].yaw;
    if (out_ratio) out_ratio[i] = mo[i].ratio;
    if (out_ok) out_ok[i] = mo[i].ok;
    if (out_scale) out_scale[i] = mo[i].scale;
  }
  return GLOC_OK;
}

}  // extern "C"
