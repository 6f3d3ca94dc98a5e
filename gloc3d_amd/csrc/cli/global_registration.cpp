// global_registration -- drop-in for the reference's pairwise registration evaluator
// (registration/global_registration.cpp:1198-1444; no CMake target upstream):
//
//   global_registration VALSET POSES [kitti|nclt|auto]
//
// For every query scan and every ground-truth positive database scan: register the pair, print
// "err_pos, err_rot" (:1417), then the success rate (<1 m and <5 deg) and mean/std (:1432-1442).
// The reference composes a 2-D SURF match with ground alignment and optionally refines with PCL ICP
// (:1342-1398, use_icp=false :1222); here the pair goes through the 3-D RANSAC-SVD + ICP hot path.
// No GUI windows are opened.  Scans: the reference reads NCLT raw records unconditionally (:1239,1304) although
// its comments say KITTI (:1224-1226); here the optional third argument (or GLOC_SCAN_FORMAT) names the format,
// and "auto" (default) decides by CONTENT -- never by file size: an NCLT file with an even number of 8-byte
// records is a multiple of 16 bytes as well (host/gloc_io.hpp: looks_like_kitti).
#include <cstdio>
#include <fstream>
#include <memory>

#include "loop_detector.hpp"

using namespace gloc_host;

static ScanFormat g_format = ScanFormat::Auto;
static std::vector<float> read_scan(const std::string& path) { return read_lidar_any(path, g_format); }

int main(int argc, char* argv[]) {
  if (argc < 3) {
    std::fprintf(stderr, "usage: %s VALSET POSES [kitti|nclt|auto]\n", argv[0]);
    return 2;
  }
  g_format = scan_format_from_string(argc > 3 ? argv[3] : getenv("GLOC_SCAN_FORMAT"));
  Valset vs;
  std::vector<Mat4> poses;
  if (!read_valset(argv[1], vs) || !read_valset_pose(argv[2], poses)) return 1;
  if (vs.q_files.size() != vs.pos_idx.size()) {  // CHECK at :1210
    std::fprintf(stderr, "Check failed: q_files.size()==gt_q_pos_idx.size()\n");
    return 1;
  }
  gloc_reg* reg = nullptr;
  if (gloc_reg_create(0, &reg) != GLOC_OK) {
    std::fprintf(stderr, "fatal: %s\n", gloc_last_error());
    return 1;
  }
  gloc_reg_params prm;
  gloc_reg_default_params(&prm);
  const size_t ndb = vs.db_files.size();
  int all_tests = 0, succeed = 0;
  std::vector<double> rot_err, pos_err;
  for (size_t i = 0; i < vs.q_files.size(); ++i) {
    if (vs.pos_idx[i].empty()) continue;
    const std::vector<float> q = read_scan(vs.q_files[i]);
    uint32_t qid = 0;
    if (gloc_reg_scan_upload(reg, q.data(), q.size() / 4, 4, &qid) != GLOC_OK) {
      std::fprintf(stderr, "fatal: %s\n", gloc_last_error());
      return 1;
    }
    std::vector<uint32_t> ids;
    for (size_t j : vs.pos_idx[i]) {
      const std::vector<float> d = read_scan(vs.db_files.at(j));
      uint32_t sid = 0;
      if (gloc_reg_scan_upload(reg, d.data(), d.size() / 4, 4, &sid) != GLOC_OK) {
        std::fprintf(stderr, "fatal: %s\n", gloc_last_error());
        return 1;
      }
      ids.push_back(sid);
    }
    std::vector<float> T(16 * ids.size());
    if (gloc_reg_batch_ids(reg, qid, ids.data(), ids.size(), nullptr, nullptr, &prm, T.data(), nullptr, nullptr,
                           nullptr) != GLOC_OK) {
      std::fprintf(stderr, "fatal: %s\n", gloc_last_error());
      return 1;
    }
    for (size_t c = 0; c < ids.size(); ++c) {
      const size_t j = vs.pos_idx[i][c];
      const Mat4 q2db = mul4(rigid_inverse(poses.at(j)), poses.at(ndb + i));
      Mat4 est;
      std::copy(T.begin() + 16 * c, T.begin() + 16 * (c + 1), est.begin());
      float er, ep;
      pose_error(q2db, est, er, ep);
      std::printf("err_pos, err_rot: %g, %g\n", ep, er);
      all_tests++;
      if (ep < 1.0f && er < 5.f) {
        succeed++;
        rot_err.push_back(er);
        pos_err.push_back(ep);
      }
    }
    gloc_reg_scan_clear(reg);
  }
  double mr = 0, sr = 0, mp = 0, sp = 0;
  if (pos_err.size() > 1) {
    mean_std(pos_err, mp, sp);
    mean_std(rot_err, mr, sr);
  }
  std::printf("%d, %d\n", succeed, all_tests);
  std::printf("Success rate: %g\n", all_tests ? (float)succeed / (float)all_tests : 0.f);
  std::printf("Rot error: %g, %g\n", mr, sr);
  std::printf("Pos error: %g, %g\n", mp, sp);
  gloc_reg_destroy(reg);
  return 0;
}
