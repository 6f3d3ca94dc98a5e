// gloc_io.hpp -- the file formats the reference's command lines read (host side, header only).
//   valset : registration/global_localization.cpp:64-122   (ReadValset)
//   poses  : registration/global_localization.cpp:124-156  (ReadValsetPose)
//   scans  : KITTI .bin  registration/global_localization.cpp:160-182 (read_lidar_data)
//            NCLT raw    registration/global_registration.cpp:181-209 (read_lidar_data_nclt)
// plus the descriptor file that stands in for the TorchScript model (the CNN is out of scope).
#pragma once
#include <array>
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <sstream>
#include <string>
#include <vector>

namespace gloc_host {

using Mat4 = std::array<float, 16>;  // row-major

inline Mat4 identity4() { return Mat4{1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1}; }

inline Mat4 mul4(const Mat4& a, const Mat4& b) {
  Mat4 o{};
  for (int i = 0; i < 4; ++i)
    for (int j = 0; j < 4; ++j) {
      float s = 0.f;
      for (int k = 0; k < 4; ++k) s += a[4 * i + k] * b[4 * k + j];
      o[4 * i + j] = s;
    }
  return o;
}

inline Mat4 rigid_inverse(const Mat4& t) {  // poses are rigid: [R t; 0 1]^-1 = [R^T  -R^T t]
  Mat4 o = identity4();
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) o[4 * i + j] = t[4 * j + i];
  for (int i = 0; i < 3; ++i)
    o[4 * i + 3] = -(o[4 * i + 0] * t[3] + o[4 * i + 1] * t[7] + o[4 * i + 2] * t[11]);
  return o;
}

// tokens separated by any of `delims`, empty tokens dropped (strtok semantics, :44-58)
inline std::vector<std::string> split(const std::string& s, const char* delims) {
  std::vector<std::string> out;
  size_t i = 0;
  while (i < s.size()) {
    const size_t b = s.find_first_not_of(delims, i);
    if (b == std::string::npos) break;
    size_t e = s.find_first_of(delims, b);
    if (e == std::string::npos) e = s.size();
    out.push_back(s.substr(b, e - b));
    i = e;
  }
  return out;
}

struct Valset {
  std::vector<std::string> db_files, q_files;
  std::vector<std::vector<size_t>> pos_idx;  // may be shorter than q_files (reader stops at EOF/blank)
};

// Quirks reproduced: the "<qIdx>" token before ':' is parsed but ignored -- positives are assigned
// positionally (:104,116); reading stops at EOF or the first empty line (:95-98).
inline bool read_valset(const std::string& filename, Valset& v) {
  std::ifstream ifs(filename);
  v = Valset{};
  if (!ifs.is_open()) {
    std::printf("failed to open file %s\n", filename.c_str());
    return false;
  }
  std::string line;
  std::getline(ifs, line);
  const auto head = split(line, " ");
  if (head.size() < 2) return false;
  const int db_num = std::atoi(head[0].c_str()), q_num = std::atoi(head[1].c_str());
  for (int i = 0; i < db_num; ++i) {
    std::getline(ifs, line);
    v.db_files.push_back(line);
  }
  for (int i = 0; i < q_num; ++i) {
    std::getline(ifs, line);
    v.q_files.push_back(line);
  }
  for (int i = 0; i < q_num; ++i) {
    if (!std::getline(ifs, line)) break;
    if (line.empty()) break;
    const auto parts = split(line, ":");
    if (parts.size() == 1) {
      v.pos_idx.push_back({});
      continue;
    }
    std::vector<size_t> pos;
    for (const auto& tok : split(parts[1], " ")) pos.push_back((size_t)std::atoi(tok.c_str()));
    v.pos_idx.push_back(pos);
  }
  return true;
}

// "qx qy qz qw x y z" per line, db poses first then query poses (:136-153)
inline bool read_valset_pose(const std::string& filename, std::vector<Mat4>& poses) {
  std::ifstream ifs(filename);
  poses.clear();
  if (!ifs.is_open()) {
    std::printf("failed to open file %s\n", filename.c_str());
    return false;
  }
  std::string line;
  while (std::getline(ifs, line)) {
    const auto t = split(line, " ");
    if (t.size() != 7) {
      std::fprintf(stderr, "Check failed: substrs.size()==7 (%zu) in %s\n", t.size(), filename.c_str());
      std::abort();  // the reference CHECKs (:138)
    }
    const float qx = (float)std::atof(t[0].c_str()), qy = (float)std::atof(t[1].c_str()),
                qz = (float)std::atof(t[2].c_str()), qw = (float)std::atof(t[3].c_str());
    Mat4 p = identity4();  // Eigen::Quaternionf(w,x,y,z).toRotationMatrix()
    p[0] = 1 - 2 * (qy * qy + qz * qz); p[1] = 2 * (qx * qy - qz * qw); p[2] = 2 * (qx * qz + qy * qw);
    p[4] = 2 * (qx * qy + qz * qw); p[5] = 1 - 2 * (qx * qx + qz * qz); p[6] = 2 * (qy * qz - qx * qw);
    p[8] = 2 * (qx * qz - qy * qw); p[9] = 2 * (qy * qz + qx * qw); p[10] = 1 - 2 * (qx * qx + qy * qy);
    p[3] = (float)std::atof(t[4].c_str());
    p[7] = (float)std::atof(t[5].c_str());
    p[11] = (float)std::atof(t[6].c_str());
    poses.push_back(p);
  }
  return true;
}

// KITTI: packed float32 x y z i.  Returns x,y,z,i quadruples (stride 4).
inline std::vector<float> read_lidar_kitti(const std::string& path) {
  std::ifstream f(path, std::ifstream::in | std::ifstream::binary);
  std::vector<float> buf;
  if (!f.is_open()) return buf;
  f.seekg(0, std::ios::end);
  const size_t n = (size_t)f.tellg() / sizeof(float);
  f.seekg(0, std::ios::beg);
  buf.resize(n - n % 4);
  f.read(reinterpret_cast<char*>(buf.data()), (std::streamsize)(buf.size() * sizeof(float)));
  return buf;
}

// NCLT velodyne_sync: u16 x,y,z; u8 intensity, label; metres = v * 0.005 - 100
// (read_lidar_data_nclt, registration/global_registration.cpp:181-209).  The reference's loop tests eof()
// BEFORE it reads (:191-192), so after the last whole record it goes round once more: the reads fail, the
// fields keep their values and the last point is pushed a second time -- a file of n records gives n + 1
// points (and the bytes of a truncated trailing record overwrite the leading fields of that extra point).
// Reproduced here, so that clouds -- and everything computed from them -- equal the reference's.  An empty
// file yields an empty cloud (upstream pushes one point of uninitialised fields).
inline std::vector<float> read_lidar_nclt(const std::string& path) {
  std::ifstream f(path, std::ifstream::in | std::ifstream::binary);
  std::vector<float> out;
  if (!f.is_open()) return out;
  f.seekg(0, std::ios::end);
  const size_t bytes = (size_t)f.tellg();
  f.seekg(0, std::ios::beg);
  if (bytes == 0) return out;
  std::vector<unsigned char> raw(bytes);
  f.read(reinterpret_cast<char*>(raw.data()), (std::streamsize)bytes);
  unsigned char rec[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  out.reserve(4 * (bytes / 8 + 1));
  for (size_t off = 0;; off += 8) {
    const size_t avail = off < bytes ? (bytes - off < 8 ? bytes - off : 8) : 0;
    std::memcpy(rec, raw.data() + (off < bytes ? off : 0), avail);  // a failed read leaves the rest untouched
    uint16_t v[3];
    std::memcpy(v, rec, 6);
    out.push_back((float)v[0] * 0.005f + -100.0f);
    out.push_back((float)v[1] * 0.005f + -100.0f);
    out.push_back((float)v[2] * 0.005f + -100.0f);
    out.push_back((float)rec[6]);
    if (avail < 8) break;  // that read hit the end of the file: eof() is set, the loop ends
  }
  return out;
}

enum class ScanFormat { Auto, Kitti, Nclt };

// "kitti" / "nclt" / "auto" (anything else: Auto)
inline ScanFormat scan_format_from_string(const char* s) {
  if (!s) return ScanFormat::Auto;
  const std::string t(s);
  if (t == "kitti" || t == "KITTI") return ScanFormat::Kitti;
  if (t == "nclt" || t == "NCLT") return ScanFormat::Nclt;
  return ScanFormat::Auto;
}

// Tell a KITTI float file from an NCLT raw file by CONTENT (the file size says nothing: an NCLT file with an
// even number of records is a multiple of 16 bytes too).  KITTI: whole x y z i float quadruples, every coordinate
// finite and within a kilometre, none of them a denormal / vanishing non-zero.  NCLT records read as floats pair
// two u16 fields: (x | y << 16) has the exponent of y's top bits -- 5e8 for a point at y = 0 -- and
// (z | (i | l << 8) << 16) is a denormal for label 0.  The first 256 points decide.
inline bool looks_like_kitti(const std::string& path) {
  std::ifstream f(path, std::ifstream::in | std::ifstream::binary);
  if (!f.is_open()) return false;
  f.seekg(0, std::ios::end);
  const size_t bytes = (size_t)f.tellg();
  f.seekg(0, std::ios::beg);
  if (bytes == 0 || bytes % 16 != 0) return bytes == 0;
  const size_t n = std::min<size_t>(bytes / 16, 256);
  std::vector<float> v(4 * n);
  f.read(reinterpret_cast<char*>(v.data()), (std::streamsize)(16 * n));
  for (size_t i = 0; i < n; ++i)
    for (int a = 0; a < 3; ++a) {
      const float c = v[4 * i + a];
      if (!std::isfinite(c) || std::fabs(c) > 1000.f || (c != 0.f && std::fabs(c) < 1e-20f)) return false;
    }
  return true;
}

// The reference's global_registration reads every scan with the NCLT reader (:1239,1304), its
// global_localization with the KITTI one (global_localization.cpp:160-182); a drop-in for both takes the format
// from the caller (trailing argument / GLOC_SCAN_FORMAT) and otherwise from the content.
inline std::vector<float> read_lidar_any(const std::string& path, ScanFormat fmt = ScanFormat::Auto) {
  if (fmt == ScanFormat::Auto) fmt = looks_like_kitti(path) ? ScanFormat::Kitti : ScanFormat::Nclt;
  return fmt == ScanFormat::Kitti ? read_lidar_kitti(path) : read_lidar_nclt(path);
}

// Descriptor file standing in for MODEL: "GLOCDESC" u32 n u32 dim, then n*dim float32
// (db rows first, then query rows -- the order of the valset).
inline bool read_descriptors(const std::string& path, std::vector<float>& data, size_t& n, size_t& dim) {
  std::ifstream f(path, std::ifstream::in | std::ifstream::binary);
  char magic[8];
  uint32_t hdr[2];
  if (!f.is_open() || !f.read(magic, 8) || std::memcmp(magic, "GLOCDESC", 8) != 0 ||
      !f.read(reinterpret_cast<char*>(hdr), 8))
    return false;
  n = hdr[0];
  dim = hdr[1];
  data.resize(n * dim);
  return (bool)f.read(reinterpret_cast<char*>(data.data()), (std::streamsize)(data.size() * sizeof(float)));
}

inline void mean_std(const std::vector<double>& v, double& mean, double& stdev) {  // :185-196 (n-1)
  double sum = 0;
  for (double d : v) sum += d;
  mean = sum / (double)v.size();
  double acc = 0;
  for (double d : v) acc += (d - mean) * (d - mean);
  stdev = std::sqrt(acc / (double)(v.size() - 1));
}

// err_rot (deg, with the ~180-degree forgiveness) and err_pos of an estimate vs ground truth,
// registration/global_localization.cpp:288-306.
inline void pose_error(const Mat4& q2db_gt, const Mat4& est, float& err_rot_deg, float& err_pos) {
  float trace = 0.f;
  for (int i = 0; i < 3; ++i)
    for (int k = 0; k < 3; ++k) trace += q2db_gt[4 * k + i] * est[4 * k + i];
  float off = 0.5f * (trace - 1.f);
  off = off < -0.999999f ? -0.999999f : off;
  off = off > 0.999999f ? 0.999999f : off;
  float er = std::fabs(std::acos(off)) * (float)(180. / M_PI);
  const float dx = q2db_gt[3] - est[3], dy = q2db_gt[7] - est[7], dz = q2db_gt[11] - est[11];
  err_pos = std::sqrt(dx * dx + dy * dy + dz * dz);
  if (std::fabs(er - 180.f) < 5.f) er = std::fabs(er - 180.f);
  err_rot_deg = er;
}

}  // namespace gloc_host
