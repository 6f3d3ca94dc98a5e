// Test helper (CPU only): prints what csrc/host/gloc_io.hpp's scan readers return for a file, so that
// tests/test_host_cpu.py can compare the C++ host mirror with the Python one and with a literal emulation of
// the reference's read loop (registration/global_registration.cpp:181-209).
//   host_io_probe PATH [kitti|nclt|auto]   ->   "<format sniffed> <n points>" then one "x y z i" line per point (%.9g)
#include <cstdio>

#include "gloc_io.hpp"

int main(int argc, char** argv) {
  if (argc < 2) return 2;
  const gloc_host::ScanFormat fmt = gloc_host::scan_format_from_string(argc > 2 ? argv[2] : nullptr);
  const std::vector<float> v = gloc_host::read_lidar_any(argv[1], fmt);
  std::printf("%s %zu\n", gloc_host::looks_like_kitti(argv[1]) ? "kitti" : "nclt", v.size() / 4);
  for (size_t i = 0; i + 3 < v.size(); i += 4) std::printf("%.9g %.9g %.9g %.9g\n", v[i], v[i + 1], v[i + 2], v[i + 3]);
  return 0;
}
