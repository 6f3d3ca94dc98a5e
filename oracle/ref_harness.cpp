// ref_harness.cpp -- C entry points around the REFERENCE's own vendored nanoflann.
//
// TEST INFRASTRUCTURE ONLY.  This file contains no reference code: it includes the two headers
// where they lie under /root/reference/registration (-I given by oracle/Makefile) and exposes
// them over a C ABI so tests/ and bench.py's cpu_baseline leg can call the reference itself:
//   registration/nanoflann.hpp                    (KD-tree, L2_Adaptor, L2_Simple_Adaptor)
//   registration/KDTreeVectorOfVectorsAdaptor.h   (InvKeyTree, as used by loop_detector.h:27-32)
// Output goes to oracle/_ref/ only (git-ignored; travels to the GPU box as a built .so).
#include <cstddef>
#include <cstdint>
#include <memory>
#include <vector>

#include "KDTreeVectorOfVectorsAdaptor.h"
#include "nanoflann.hpp"

namespace {

// Same aliases as registration/loop_detector.h:31-32.
using KeyMat = std::vector<std::vector<float>>;
using InvKeyTree = KDTreeVectorOfVectorsAdaptor<KeyMat, float>;

struct RefKnn {
  KeyMat db;  // the adaptor keeps a const-ref to this (KDTreeVectorOfVectorsAdaptor.h:88)
  std::unique_ptr<InvKeyTree> tree;
  size_t dim = 0;
};

// 3-D point cloud adaptor for the stand-in of PCL's KdTreeFLANN (exact 1-NN, L2_Simple).
struct Cloud3 {
  const float* xyz;
  size_t n;
  inline size_t kdtree_get_point_count() const { return n; }
  inline float kdtree_get_pt(const size_t idx, const size_t d) const { return xyz[idx * 3 + d]; }
  template <class BBOX>
  bool kdtree_get_bbox(BBOX&) const {
    return false;
  }
};
using Tree3 = nanoflann::KDTreeSingleIndexAdaptor<nanoflann::L2_Simple_Adaptor<float, Cloud3>,
                                                  Cloud3, 3, uint32_t>;
struct RefNn3 {
  std::vector<float> pts;
  Cloud3 cloud;
  std::unique_ptr<Tree3> tree;
};

}  // namespace

extern "C" {

// InvKeyTree(k_dim_, db_features_, 10): registration/loop_detector.cpp:36
void* ref_knn_build(const float* db, size_t n_rows, size_t dim) {
  auto* h = new RefKnn;
  h->dim = dim;
  h->db.resize(n_rows);
  for (size_t i = 0; i < n_rows; ++i) h->db[i].assign(db + i * dim, db + (i + 1) * dim);
  h->tree.reset(new InvKeyTree(dim, h->db, 10));
  return h;
}

// kdtree_->query(&feat[0], top_k_, &ret_indexes[0], &out_dists_sqr[0]): loop_detector.cpp:45
void ref_knn_query(void* handle, const float* queries, size_t nq, size_t k, uint64_t* out_idx,
                   float* out_d2) {
  auto* h = static_cast<RefKnn*>(handle);
  std::vector<size_t> idx(k);
  for (size_t q = 0; q < nq; ++q) {
    h->tree->query(queries + q * h->dim, k, idx.data(), out_d2 + q * k);
    for (size_t j = 0; j < k; ++j) out_idx[q * k + j] = static_cast<uint64_t>(idx[j]);
  }
}

void ref_knn_free(void* handle) { delete static_cast<RefKnn*>(handle); }

void* ref_nn3_build(const float* xyz, size_t n) {
  auto* h = new RefNn3;
  h->pts.assign(xyz, xyz + 3 * n);
  h->cloud.xyz = h->pts.data();
  h->cloud.n = n;
  h->tree.reset(new Tree3(3, h->cloud, nanoflann::KDTreeSingleIndexAdaptorParams(10)));
  return h;
}

void ref_nn3_query(void* handle, const float* src_xyz, size_t n_src, uint32_t* out_idx,
                   float* out_d2) {
  auto* h = static_cast<RefNn3*>(handle);
  for (size_t i = 0; i < n_src; ++i) {
    nanoflann::KNNResultSet<float, uint32_t> rs(1);
    rs.init(out_idx + i, out_d2 + i);
    h->tree->findNeighbors(rs, src_xyz + 3 * i);
  }
}

void ref_nn3_free(void* handle) { delete static_cast<RefNn3*>(handle); }

}  // extern "C"
