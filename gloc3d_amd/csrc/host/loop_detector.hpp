// loop_detector.hpp -- host-side mirror of the reference's RpyPCLoopDetector
// (registration/loop_detector.h:41-119) for the hot path: same method names, argument meaning and
// guards, over the C ABI (include/gloc3d.h).  Differences, all forced by scope:
//   * the descriptor comes from the caller (the CNN backbone is upstream of the hot path);
//     get_projected_grid / get_place_input are the BEV projection in front of it;
//   * match(scan, db_idx, xy_yaw, scale) is the coarse 2-D match on the BEV grids (an exhaustive yaw x shift
//     search instead of SURF + FLANN + RANSAC-affine); the batched match() is the 3-D registration
//     (RANSAC-SVD + ICP) of north_star, seeded by it.
// Not thread-safe; single caller thread, like the reference.
#pragma once
#include <cmath>
#include <cstdint>
#include <stdexcept>
#include <string>
#include <vector>

#include "gloc3d.h"
#include "gloc_io.hpp"

namespace gloc_host {

class RpyPCLoopDetector {
 public:
  explicit RpyPCLoopDetector(size_t k_dim = 512, int device = 0) : k_dim_(k_dim), device_(device) {
    if (gloc_knn_create(device, k_dim_, &knn_) != GLOC_OK) throw std::runtime_error(gloc_last_error());
    if (gloc_reg_create(device, &reg_) != GLOC_OK) {
      std::string e = gloc_last_error();
      gloc_knn_destroy(knn_);
      throw std::runtime_error(e);
    }
    gloc_reg_default_params(&reg_params_);
    if (gloc_coarse_create(device, &coarse_) != GLOC_OK) {
      std::string e = gloc_last_error();
      gloc_reg_destroy(reg_);
      gloc_knn_destroy(knn_);
      throw std::runtime_error(e);
    }
    gloc_coarse_default_params(&coarse_params_);
  }
  ~RpyPCLoopDetector() {
    gloc_coarse_destroy(coarse_);
    gloc_bev_destroy(bev_);
    gloc_reg_destroy(reg_);
    gloc_knn_destroy(knn_);
  }
  RpyPCLoopDetector(const RpyPCLoopDetector&) = delete;
  RpyPCLoopDetector& operator=(const RpyPCLoopDetector&) = delete;

  const int NUM_EXCLUDE_RECENT = 30;  // loop_detector.h:77

  // Multi-GPU (the reference is single-GPU): with a communicator attached BEFORE the first
  // add_keyframe, place g's descriptor lives on rank g % world (interleaved row shards), every rank keeps
  // all scans and grids, and detect() is collective: local top-k -> RCCL all-gather -> merge, the same
  // result on every rank as on one GPU.  The SLAM-mode detect is not available in this mode.
  void attach_comm(gloc_comm* comm) {
    if (db_size_) throw std::runtime_error("attach_comm: the database is not empty");
    comm_ = comm;
    rank_ = 0;
    world_ = 1;
    if (comm && gloc_comm_rank(comm, &rank_, &world_) != GLOC_OK) throw std::runtime_error(gloc_last_error());
  }
  int rank() const { return rank_; }
  int world() const { return world_; }

  // add_keyframe (loop_detector.cpp:10-20): one descriptor + its scan (x,y,z,i quadruples).
  void add_keyframe(const std::vector<float>& descriptor, const float* scan_xyzi, size_t n_pts) {
    if (descriptor.size() != k_dim_) throw std::runtime_error("descriptor length != k_dim_");
    if (!comm_ || (int)(db_size_ % (size_t)world_) == rank_) check(gloc_knn_add(knn_, descriptor.data(), 1));
    uint32_t sid = 0;
    check(gloc_reg_scan_upload(reg_, scan_xyzi, n_pts, 4, &sid));
    check(gloc_reg_scan_build_target_index(reg_, sid));  // a database place is a registration target from now on
    db_scan_ids_.push_back(sid);
    uint32_t gid = 0;  // db_grids_.push_back(grid), loop_detector.cpp:16-19
    check(gloc_coarse_add_scan(coarse_, scan_xyzi, n_pts, 4, &coarse_params_, &gid));
    db_grid_ids_.push_back(gid);
    db_size_++;
  }

  // match(q_grid, db_idx, xy_yaw, scale) of the reference (loop_detector.cpp:186-288): the coarse pose of
  // the query in place db_idx's frame, p_db = R(yaw) p_q + (x, y), and the scale estimated from the matched cells;
  // true iff the grids overlap enough AND |1 - scale| < 0.1 (loop_detector.cpp:268-272).  Here the query goes in as
  // its scan (its grid is made on the device).
  bool match(const float* q_scan_xyzi, size_t n_pts, size_t db_idx, float xy_yaw[3], double& estimated_scale) {
    std::vector<float> out(3), sc(1);
    std::vector<int> ok(1);
    match_2d(q_scan_xyzi, n_pts, {db_idx}, out, ok, &sc);
    xy_yaw[0] = out[0]; xy_yaw[1] = out[1]; xy_yaw[2] = out[2];
    estimated_scale = (double)sc[0];
    return ok[0] != 0;
  }
  void match_2d(const float* q_scan_xyzi, size_t n_pts, const std::vector<size_t>& db_indices,
                std::vector<float>& xy_yaw /* n x 3 */, std::vector<int>& ok, std::vector<float>* scale = nullptr) {
    const size_t n = db_indices.size();
    xy_yaw.assign(3 * n, 0.f);
    ok.assign(n, 0);
    if (scale) scale->assign(n, 0.f);
    if (n == 0) return;
    uint32_t qg = 0;
    check(gloc_coarse_add_scan(coarse_, q_scan_xyzi, n_pts, 4, &coarse_params_, &qg));
    std::vector<uint32_t> ids(n);
    for (size_t i = 0; i < n; ++i) ids[i] = db_grid_ids_.at(db_indices[i]);
    const int rc = gloc_coarse_match(coarse_, qg, ids.data(), n, &coarse_params_, xy_yaw.data(), nullptr, ok.data(),
                                     scale ? scale->data() : nullptr);
    gloc_coarse_release(coarse_, qg);
    check(rc);
  }
  // (x, y, yaw) -> 4x4: R = RollPitchYaw(0, 0, yaw), t = (x, y, 0) (global_localization.cpp:526-530)
  static Mat4 embed_3d(const float* xy_yaw) {
    Mat4 T = identity4();
    const float c = std::cos(xy_yaw[2]), s = std::sin(xy_yaw[2]);
    T[0] = c; T[1] = -s; T[4] = s; T[5] = c;
    T[3] = xy_yaw[0]; T[7] = xy_yaw[1];
    return T;
  }
  bool use_coarse_match = true;  // match(): seed the 3-D registration with the 2-D match when no guess is given

  // global localization (loop_detector.cpp:22-46).  Leaves the vectors untouched when the
  // database is too small ("Not enough keyframes in database.", :27-30).
  void detect(const std::vector<float>& q_descriptor, std::vector<size_t>& loop_indices,
              std::vector<float>& out_dists_sqr) {
    if (db_size_ <= num_exclude_recent_ + top_k_) {
      std::printf("Not enough keyframes in database.\n");
      return;
    }
    query(q_descriptor.data(), 0, db_size_, loop_indices, out_dists_sqr);
  }

  // SLAM mode (loop_detector.cpp:48-81): the newest frame is the query; every
  // tree_making_period_ calls the searchable window is refreshed to db[0 : end-30]; a loop is
  // accepted iff the best SQUARED distance is below 0.8 (:54, loop_detector.h:103).
  bool detect(size_t& q_idx, size_t& loop_idx, const std::vector<float>& last_descriptor) {
    if (comm_) throw std::runtime_error("SLAM-mode detect is not available with a sharded database");
    if (db_size_ <= num_exclude_recent_ + top_k_) return false;
    if (tree_making_period_counter_ % tree_making_period_ == 0)
      searchable_end_ = db_size_ - num_exclude_recent_;
    tree_making_period_counter_++;
    std::vector<size_t> idx;
    std::vector<float> d2;
    query(last_descriptor.data(), 0, searchable_end_, idx, d2);
    if (idx.empty()) return false;
    if (d2[0] < loop_metric_dist_th_) {
      q_idx = db_size_ - 1;
      loop_idx = idx[0];
      return true;
    }
    return false;
  }

  // Register a query scan against retrieved candidates in one batch; returns the rank of the first
  // successful candidate (global_localization.cpp:519-572) or -1, and its pose (query -> db).
  // init_guess (optional): one initial pose (query -> db) per candidate, as icp_match_3d's
  // initial_guess (global_registration.cpp:237-248); identity when absent.
  int match(const float* q_scan_xyzi, size_t n_pts, const std::vector<size_t>& db_indices,
            Mat4& pose_in_db, std::vector<Mat4>* all_poses = nullptr, std::vector<int>* all_ok = nullptr,
            const std::vector<Mat4>* init_guess = nullptr) {
    std::vector<Mat4> coarse_init;
    if (!init_guess && use_coarse_match && !db_indices.empty()) {
      std::vector<float> xy_yaw;
      std::vector<int> ok2d;
      match_2d(q_scan_xyzi, n_pts, db_indices, xy_yaw, ok2d);
      coarse_init.resize(db_indices.size());
      for (size_t i = 0; i < db_indices.size(); ++i)
        coarse_init[i] = ok2d[i] ? embed_3d(&xy_yaw[3 * i]) : identity4();
      init_guess = &coarse_init;
    }
    uint32_t qid = 0;
    check(gloc_reg_scan_upload(reg_, q_scan_xyzi, n_pts, 4, &qid));
    int r;
    try {
      r = match_ids(qid, db_indices, pose_in_db, all_poses, all_ok, init_guess);
    } catch (...) {
      gloc_reg_scan_release(reg_, qid);
      throw;
    }
    check(gloc_reg_scan_release(reg_, qid));  // the query scan is transient: HBM stays flat over a run
    return r;
  }

  // Number of scans resident in HBM (the database's; query scans are released after match()).
  size_t resident_scans() const {
    size_t n = 0;
    gloc_reg_scan_count(reg_, &n);
    return n;
  }

  int match_ids(uint32_t q_scan_id, const std::vector<size_t>& db_indices, Mat4& pose_in_db,
                std::vector<Mat4>* all_poses, std::vector<int>* all_ok,
                const std::vector<Mat4>* init_guess = nullptr) {
    const size_t n = db_indices.size();
    if (n == 0) return -1;
    std::vector<uint32_t> ids(n);
    for (size_t i = 0; i < n; ++i) ids[i] = db_scan_ids_.at(db_indices[i]);
    std::vector<float> init;
    if (!all_poses && !all_ok) {
      // nobody asked for every candidate's pose: walk the candidates as the reference does, stopping at the
      // first success (global_localization.cpp:519-572) -- the same rank and pose, fewer registrations
      if (init_guess) {
        if (init_guess->size() != n) throw std::runtime_error("one initial guess per candidate expected");
        init.resize(16 * n);
        for (size_t i = 0; i < n; ++i) std::copy((*init_guess)[i].begin(), (*init_guess)[i].end(), init.begin() + 16 * i);
      }
      int rank = -1;
      float Tq[16];
      check(gloc_reg_first_success_multi(reg_, 1, &q_scan_id, ids.data(), n, init_guess ? init.data() : nullptr,
                                         &reg_params_, &rank, Tq, nullptr, nullptr, nullptr));
      if (rank >= 0) std::copy(Tq, Tq + 16, pose_in_db.begin());
      return rank;
    }
    std::vector<float> T(16 * n);
    std::vector<int> ok(n);
    if (init_guess) {
      if (init_guess->size() != n) throw std::runtime_error("one initial guess per candidate expected");
      init.resize(16 * n);
      for (size_t i = 0; i < n; ++i) std::copy((*init_guess)[i].begin(), (*init_guess)[i].end(), init.begin() + 16 * i);
    }
    check(gloc_reg_batch_ids(reg_, q_scan_id, ids.data(), n, nullptr, init_guess ? init.data() : nullptr,
                             &reg_params_, T.data(), nullptr, nullptr, ok.data()));
    if (all_poses) {
      all_poses->resize(n);
      for (size_t i = 0; i < n; ++i) std::copy(T.begin() + 16 * i, T.begin() + 16 * (i + 1), (*all_poses)[i].begin());
    }
    if (all_ok) *all_ok = ok;
    const int r = gloc_reg_select_first_ok(ok.data(), n);
    if (r >= 0) std::copy(T.begin() + 16 * r, T.begin() + 16 * (r + 1), pose_in_db.begin());
    return r;
  }

  // get_projected_grid (loop_detector.cpp:122-135): the occupancy image ([height][width] u8, 0 =
  // occupied column) of one scan and xy_res = (ox, oy, resolution).
  struct Grid {
    std::vector<uint8_t> image;
    uint32_t width = 0, height = 0;
    double ox = 0, oy = 0, resolution = 0;
  };
  Grid get_projected_grid(const float* scan_xyzi, size_t n_pts) {
    gloc_bev_params p = bev_params();
    std::vector<uint8_t> scratch((size_t)p.out_width * p.out_height * 3);
    gloc_bev_info info;
    check(gloc_bev_project(projector(), scan_xyzi, n_pts, 4, &p, scratch.data(), &info));
    if (info.empty) throw std::runtime_error("no point of the scan lies within range");  // the reference aborts
    Grid g;
    g.width = info.width; g.height = info.height;
    g.ox = info.ox; g.oy = info.oy; g.resolution = info.resolution;
    g.image.resize((size_t)g.width * g.height);
    check(gloc_bev_raw_image(projector(), 0, g.image.data(), g.image.size()));
    return g;
  }

  // The tensor get_place_feature feeds the descriptor network (loop_detector.cpp:137-151):
  // [3][768][768] f32 in {0, 1} (crop_pad_occupancy, / 255, NHWC -> NCHW).
  std::vector<float> get_place_input(const float* scan_xyzi, size_t n_pts, gloc_bev_info* info = nullptr) {
    gloc_bev_params p = bev_params();
    p.format = GLOC_BEV_F32_CHW;
    std::vector<float> chw((size_t)3 * p.out_width * p.out_height);
    check(gloc_bev_project(projector(), scan_xyzi, n_pts, 4, &p, chw.data(), info));
    return chw;
  }

  gloc_reg_params& registration_params() { return reg_params_; }
  size_t size() const { return db_size_; }
  size_t top_k() const { return top_k_; }

 private:
  void check(int rc) {
    if (rc != GLOC_OK) throw std::runtime_error(gloc_last_error());
  }
  gloc_bev* projector() {  // created on first use
    if (!bev_) check(gloc_bev_create(device_, &bev_));
    return bev_;
  }
  gloc_bev_params bev_params() const {
    gloc_bev_params p;
    gloc_bev_default_params(&p);
    p.resolution = high_resolution_;
    p.max_range = high_resolution_max_range_;
    return p;
  }
  void query(const float* q, size_t first, size_t last, std::vector<size_t>& idx, std::vector<float>& d2) {
    std::vector<uint64_t> i64(top_k_);
    idx.resize(top_k_);   // loop_detector.cpp:42-43
    d2.resize(top_k_);
    if (comm_)
      check(gloc_knn_search_sharded_host(knn_, comm_, q, 1, top_k_, (uint64_t)world_, (uint64_t)rank_, i64.data(),
                                         d2.data()));
    else
      check(gloc_knn_search(knn_, q, 1, top_k_, first, last, i64.data(), d2.data()));
    for (size_t i = 0; i < top_k_; ++i) idx[i] = (size_t)i64[i];
  }

  const size_t k_dim_;
  const size_t top_k_ = 20;                // loop_detector.h:98
  const size_t num_exclude_recent_ = 30;   // :99
  const size_t tree_making_period_ = 30;   // :100
  size_t tree_making_period_counter_ = 0;  // :101
  const float loop_metric_dist_th_ = 0.8f; // :103
  const float high_resolution_max_range_ = 100.f;  // :115
  const float high_resolution_ = 0.2f;             // :116
  size_t db_size_ = 0, searchable_end_ = 0;
  gloc_knn* knn_ = nullptr;
  gloc_reg* reg_ = nullptr;
  gloc_bev* bev_ = nullptr;
  gloc_coarse* coarse_ = nullptr;
  gloc_comm* comm_ = nullptr;
  int rank_ = 0, world_ = 1;
  gloc_coarse_params coarse_params_{};
  std::vector<uint32_t> db_grid_ids_;
  int device_ = 0;
  gloc_reg_params reg_params_{};
  std::vector<uint32_t> db_scan_ids_;
};

}  // namespace gloc_host
