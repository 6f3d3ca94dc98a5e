// global_localization -- drop-in for the reference's evaluator
// (registration/global_localization.cpp:577-600):
//
//   global_localization VALSET POSES DESCRIPTORS [x]
//
// Same argv positions and the same report: recall@{1,5,10,20}, success rate, rot/pos error
// mean +/- std, failed_detect_indices.txt and failed_registration_indices.txt in the CWD.
// argv[3] is the descriptor file that stands in for the TorchScript model (the CNN is out of the
// hot path's scope): "GLOCDESC" u32 n u32 dim, then db descriptors followed by query descriptors in
// valset order.  A 4th argument selects ground alignment as in the reference (:584-588): every db
// and query scan is pre-aligned by gloc_ground_estimate (:431-436, :495-499), registration runs on
// the aligned clouds and the pose is carried back with Tdb_l2g^-1 * T * Tq_l2g (:527-541).
//
// Multi-GPU (one process per GPU; SURVEY.md 8e -- the reference is single-GPU): start N copies with
//   GLOC_WORLD=N GLOC_RANK=r GLOC_COMM_ID_FILE=/shared/path [GLOC_DEVICE=ordinal, default r]
// The descriptor database is interleave-sharded over the ranks; every query's retrieval is collective
// (local top-k -> RCCL all-gather over xGMI -> merge: gloc_knn_search_sharded); query q is registered by
// rank q % N against that rank's replica of the scan store; rank 0 gathers the results and reports.
#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <thread>
#include <memory>
#include <stdexcept>

#include "loop_detector.hpp"

using namespace gloc_host;

namespace {

struct GlocEvaluator {
  Valset vs;
  std::vector<Mat4> poses_db_q;
  std::vector<float> desc;
  size_t n_desc = 0, dim = 0;
  std::unique_ptr<RpyPCLoopDetector> det;
  std::vector<std::vector<size_t>> queried_idx;
  std::vector<std::pair<size_t, Mat4>> located;  // {db idx, pose in db}
  double time_sum_match = 0, times_call_match = 0;
  bool align_ground = false;
  int rank = 0, world = 1, device = 0;
  gloc_comm* comm = nullptr;
  gloc_ground* ground = nullptr;
  std::vector<Mat4> db_rpz_estimates;  // T_l2g per database scan (:393)

  ~GlocEvaluator() {
    gloc_ground_destroy(ground);
    det.reset();
    gloc_comm_destroy(comm);
  }

  // GLOC_WORLD / GLOC_RANK / GLOC_COMM_ID_FILE: rank 0 publishes the RCCL id through the file
  void init_comm() {
    const char* w = std::getenv("GLOC_WORLD");
    world = w ? std::atoi(w) : 1;
    rank = std::getenv("GLOC_RANK") ? std::atoi(std::getenv("GLOC_RANK")) : 0;
    device = std::getenv("GLOC_DEVICE") ? std::atoi(std::getenv("GLOC_DEVICE")) : (w ? rank : 0);
    if (!w) return;
    const char* path = std::getenv("GLOC_COMM_ID_FILE");
    if (world < 1 || rank < 0 || rank >= world || !path) throw std::runtime_error("GLOC_WORLD / GLOC_RANK / GLOC_COMM_ID_FILE");
    uint8_t id[128];
    if (rank == 0) {
      if (gloc_comm_unique_id(id) != GLOC_OK) throw std::runtime_error(gloc_last_error());
      const std::string tmp = std::string(path) + ".tmp";
      std::ofstream(tmp, std::ios::binary).write(reinterpret_cast<const char*>(id), 128);
      if (std::rename(tmp.c_str(), path) != 0) throw std::runtime_error("cannot publish the communicator id");
    } else {
      bool got = false;
      for (int i = 0; i < 1200 && !got; ++i) {  // up to two minutes
        std::ifstream f(path, std::ios::binary);
        got = f && f.read(reinterpret_cast<char*>(id), 128) && f.gcount() == 128;
        if (!got) std::this_thread::sleep_for(std::chrono::milliseconds(100));
      }
      if (!got) throw std::runtime_error("timed out waiting for the communicator id file");
    }
    if (gloc_comm_create(device, rank, world, id, &comm) != GLOC_OK) throw std::runtime_error(gloc_last_error());
  }

  // EsitmateGroundAndTransform: the scan is replaced by its ground-aligned copy; returns T_l2g
  Mat4 align(std::vector<float>& scan) {
    if (!ground && gloc_ground_create(0, &ground) != GLOC_OK) throw std::runtime_error(gloc_last_error());
    gloc_ground_params p;
    gloc_ground_default_params(&p);
    Mat4 T = identity4();
    std::vector<float> out(scan.size());
    if (gloc_ground_estimate(ground, scan.data(), scan.size() / 4, 4, &p, T.data(), nullptr, out.data()) != GLOC_OK)
      throw std::runtime_error(gloc_last_error());
    scan.swap(out);
    return T;
  }

  static double now_ms() {
    return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count();
  }

  void construct_db() {  // :419-449
    det.reset(new RpyPCLoopDetector(dim, device));
    if (comm) det->attach_comm(comm);
    double t_add = 0, t_align = 0;
    for (size_t i = 0; i < vs.db_files.size(); ++i) {
      std::vector<float> scan = read_lidar_kitti(vs.db_files[i]);
      std::vector<float> d(desc.begin() + i * dim, desc.begin() + (i + 1) * dim);
      if (align_ground) {
        const double ta = now_ms();
        db_rpz_estimates.push_back(align(scan));
        t_align += now_ms() - ta;
      }
      const double t0 = now_ms();
      det->add_keyframe(d, scan.data(), scan.size() / 4);
      t_add += now_ms() - t0;
    }
    if (align_ground)
      std::printf("time cost for align to ground: %f ms.\n", t_align / std::max<size_t>(1, vs.db_files.size()));
    std::printf("time cost for add_keyframe (upload + index): %f ms.\n", t_add / std::max<size_t>(1, vs.db_files.size()));
  }

  void locate_all_query() {  // :211-217, :482-509, :342-356
    const size_t ndb = vs.db_files.size();
    double t_det = 0;
    queried_idx.clear();
    located.assign(vs.q_files.size(), {ndb + 1, identity4()});
    for (size_t q = 0; q < vs.q_files.size(); ++q) {
      std::vector<float> d(desc.begin() + (ndb + q) * dim, desc.begin() + (ndb + q + 1) * dim);
      std::vector<size_t> idx;
      std::vector<float> d2;
      const double t0 = now_ms();
      det->detect(d, idx, d2);
      t_det += now_ms() - t0;
      queried_idx.push_back(idx);
      if (idx.empty()) continue;
      if ((int)(q % (size_t)world) != rank) continue;  // registered by another rank
      std::vector<float> scan = read_lidar_kitti(vs.q_files[q]);
      Mat4 Tq_l2g = identity4();
      if (align_ground) Tq_l2g = align(scan);
      Mat4 pose = identity4();
      // Without alignment the registration starts from the identity (sensor frames alike).  The same
      // prior expressed between the ground frames is Tdb_l2g * Tq_l2g^-1; it also carries the half
      // turn about z by which two T_l2g can differ (the first Euler angle is kept in [0, pi]).
      std::vector<Mat4> init;
      if (align_ground)
        for (size_t c : idx) init.push_back(mul4(db_rpz_estimates[c], rigid_inverse(Tq_l2g)));
      const double t1 = now_ms();
      const int r = det->match(scan.data(), scan.size() / 4, idx, pose, nullptr, nullptr,
                               align_ground ? &init : nullptr);
      time_sum_match += now_ms() - t1;
      times_call_match += 1;
      if (r >= 0) {
        const size_t db_idx = idx[(size_t)r];
        if (align_ground)  // back to the sensor frames: Tdb_l2g^-1 * T_qg_dbg * Tq_l2g (:541)
          pose = mul4(rigid_inverse(db_rpz_estimates[db_idx]), mul4(pose, Tq_l2g));
        located[q] = {db_idx, pose};
      }
    }
    std::printf("Each query cost: %f ms.\n", t_det / std::max<size_t>(1, vs.q_files.size()));
    if (comm && world > 1) {  // every rank's (db idx, pose) rows -> all ranks; row q is valid on rank q % world
      const size_t nq = vs.q_files.size(), row = 17;
      std::vector<float> mine(nq * row, 0.f), all(nq * row * (size_t)world);
      for (size_t q = 0; q < nq; ++q) {
        mine[q * row] = (float)located[q].first;  // < 2^24: exact
        std::copy(located[q].second.begin(), located[q].second.end(), mine.begin() + q * row + 1);
      }
      if (gloc_comm_all_gather_host(comm, mine.data(), all.data(), mine.size() * sizeof(float)) != GLOC_OK)
        throw std::runtime_error(gloc_last_error());
      for (size_t q = 0; q < nq; ++q) {
        const float* r = all.data() + ((q % (size_t)world) * nq + q) * row;
        located[q].first = (size_t)r[0];
        std::copy(r + 1, r + 17, located[q].second.begin());
      }
    }
  }

  void recognition_recalls() {  // :221-268
    const int k_values[4] = {1, 5, 10, 20};
    float k_recalls[4] = {0, 0, 0, 0};
    int valid = 0;
    std::vector<size_t> failed;
    for (size_t i = 0; i < vs.q_files.size(); ++i) {
      if (i >= vs.pos_idx.size() || vs.pos_idx[i].empty()) continue;
      valid++;
      if (queried_idx[i].empty()) {
        failed.push_back(i);
        continue;
      }
      bool detected = false;
      for (int k = 0; k < 4; ++k)
        for (int j = 0; j < k_values[k] && j < (int)queried_idx[i].size(); ++j)
          if (std::find(vs.pos_idx[i].begin(), vs.pos_idx[i].end(), queried_idx[i][j]) != vs.pos_idx[i].end()) {
            k_recalls[k] += 1;
            detected = true;
            break;
          }
      if (!detected) failed.push_back(i);
    }
    if (valid > 0)
      for (int i = 0; i < 4; ++i) std::printf("Recall @ %d: %g\n", k_values[i], k_recalls[i] / valid);
    std::ofstream ofs("failed_detect_indices.txt", std::ios::out);
    for (size_t idx : failed) ofs << idx << " ";
    ofs << "\n";
  }

  void registration_recalls() {  // :270-335
    const size_t ndb = vs.db_files.size();
    int all_tests = (int)located.size(), succeed = 0;
    std::vector<double> rot_err, pos_err;
    std::vector<size_t> failed;
    for (size_t i = 0; i < located.size(); ++i) {
      const size_t db_idx = located[i].first;
      if (db_idx >= ndb) {
        failed.push_back(i);
        continue;
      }
      const Mat4 q2db = mul4(rigid_inverse(poses_db_q[db_idx]), poses_db_q[i + ndb]);
      float er, ep;
      pose_error(q2db, located[i].second, er, ep);
      if (ep < 1.0f && er < 5.f) {
        succeed++;
        rot_err.push_back(er);
        pos_err.push_back(ep);
      }
    }
    double mr = 0, sr = 0, mp = 0, sp = 0;
    if (!pos_err.empty()) {
      mean_std(pos_err, mp, sp);
      mean_std(rot_err, mr, sr);
    }
    std::printf("%d, %d\n", succeed, all_tests);
    std::printf("Success rate: %g\n", all_tests ? (float)succeed / (float)all_tests : 0.f);
    std::printf("Rot error: %g, %g\n", mr, sr);
    std::printf("Pos error: %g, %g\n", mp, sp);
    std::ofstream ofs("failed_registration_indices.txt", std::ios::out);
    for (size_t idx : failed) ofs << idx << " ";
    ofs << "\n";
    std::printf("Average 3D match costs %g ms.\n", times_call_match ? time_sum_match / times_call_match : 0.0);
  }
};

}  // namespace

int main(int argc, char* argv[]) {
  if (argc < 4) {
    std::fprintf(stderr, "usage: %s VALSET POSES DESCRIPTORS [x]\n", argv[0]);
    return 2;
  }
  GlocEvaluator g;
  g.align_ground = argc == 5;  // :584-588
  if (!read_valset(argv[1], g.vs) || !read_valset_pose(argv[2], g.poses_db_q)) return 1;
  std::printf("db_num and db_files: %zu\nq_num and q_files: %zu\nq_num and q_pos_index: %zu\n", g.vs.db_files.size(),
              g.vs.q_files.size(), g.vs.pos_idx.size());
  std::printf("Read poses with size: %zu\n", g.poses_db_q.size());
  if (!read_descriptors(argv[3], g.desc, g.n_desc, g.dim)) {
    std::fprintf(stderr,
                 "%s is not a descriptor file (GLOCDESC header).  The reference loads a TorchScript model here; the "
                 "descriptor network is upstream of this build's hot path -- export its outputs for the valset's db "
                 "then query scans and pass that file instead.\n", argv[3]);
    return 1;
  }
  if (g.n_desc != g.vs.db_files.size() + g.vs.q_files.size() ||
      g.poses_db_q.size() != g.vs.db_files.size() + g.vs.q_files.size()) {
    std::fprintf(stderr, "descriptor/pose count does not match the valset\n");
    return 1;
  }
  try {
    g.init_comm();
    g.construct_db();
    g.locate_all_query();
  } catch (const std::exception& e) {
    std::fprintf(stderr, "fatal: %s\n", e.what());
    return 1;
  }
  if (g.rank == 0) {  // the report (and the two failure files) once
    g.recognition_recalls();
    g.registration_recalls();
  }
  return 0;
}
