// scan_store.hpp -- host-side view of the resident scan store (gloc_scan_store of include/gloc3d.h),
// shared by scan_store.hip (which owns it) and reg.hip (whose registration handles read it).
#pragma once
#include <atomic>
#include <map>
#include <mutex>
#include <vector>

#include "common.hpp"
#include "scan_index.hpp"

// A scan resident in HBM: original-order xyz plus its search index (Hilbert-sorted copy with the
// original indices, boxes, sorted keys, inverse permutation, launch order).  ONE allocation per scan.
struct DevScan {
  void* block = nullptr;
  size_t block_bytes = 0;
  float* xyz = nullptr;  // original order, packed
  size_t n = 0;
  gloc::reg::ScanIndexDev idx{};
  // Launch order of the source groups (groups of 64 * cs sorted points, widest first), ONE ARRAY PER cs in
  // {1, 2, 4}, each built once (cs = 2 at upload, the others on first request) and never rewritten: a
  // DevScan copy handed out by store_get() keeps pointing at valid, unchanging data whatever other
  // handles or later calls ask for.  `order` of a copy returned by store_get(cs) is the array of that cs
  // (null for cs = 0: a target's order is never read).
  uint32_t* order_base = nullptr;
  uint32_t* order = nullptr;
  unsigned order_built = 0;  // bit (cs) set: order_of(cs) is valid
  uint32_t* kpos_mem = nullptr;  // room for the target index's curve position -> kd position table
  size_t order_g1 = 0;       // groups at cs = 1
  uint32_t* order_of(int cs) const {
    return order_base + (cs == 1 ? 0 : cs == 2 ? order_g1 : order_g1 + (order_g1 + 1) / 2);
  }
  bool live = false;
  bool kd = false;  // the index is in kd order (target index)
  int pins = 0;     // batches in flight (gloc_reg_batch_multi_begin .. _end) whose jobs hold a by-value view of THIS scan:
                    // it may not be re-sorted in place meanwhile (store_pin / store_unpin, under the store's mutex)
};

struct gloc_scan_store {
  int device = 0;
  std::mutex mu;  // guards the tables, the store's stream and its scratch
  hipStream_t stream = nullptr;
  std::vector<DevScan> scans;
  std::vector<uint32_t> free_ids;
  std::multimap<size_t, void*> free_blocks;  // released allocations by capacity, reused by later adds
  size_t live_count = 0, live_bytes = 0, cached_bytes = 0;
  // scratch of the indexing pipeline (scan_store.hip): sort ping-pong arrays over all scans of a batch, descriptor
  // tables, bounding-box partials, radix histograms; the same for the source-group sort and the kd re-sort
  gloc::DevBuf sort_keys, sort_keys2, sort_vals, sort_perm, sort_hist, stage, part, builds, segs;
  gloc::DevBuf grp_k0, grp_k1, grp_v0, grp_v1, grp_segs;
  gloc::DevBuf kd_k0, kd_k1, kd_v0, kd_v1, kd_p0, kd_p1, kd_h0, kd_h1, kd_box, kd_desc;
  std::vector<gloc::DevBuf*> scratch() {
    return {&sort_keys, &sort_keys2, &sort_vals, &sort_perm, &sort_hist, &stage, &part, &builds, &segs, &grp_k0, &grp_k1,
            &grp_v0, &grp_v1, &grp_segs, &kd_k0, &kd_k1, &kd_v0, &kd_v1, &kd_p0, &kd_p1, &kd_h0, &kd_h1, &kd_box, &kd_desc};
  }
  std::atomic<int> attached{0};  // registration handles using this store
};

namespace gloc {
namespace reg {

// Build a scan from host (`device_src` false) or device memory into a fresh or recycled allocation;
// returns after the indexing work has completed on the store's stream.  Caller holds store->mu.
int store_make_scan(gloc_scan_store* st, const float* pts, size_t n, size_t stride, bool device_src,
                    DevScan* out);
// The same for `count` scans in ONE launch sequence (kernels take the scan from blockIdx.y, sorts are segmented).
int store_make_scans(gloc_scan_store* st, size_t count, const float* const* pts, const size_t* n, size_t stride,
                     bool device_src, DevScan* out);
void store_free_scan(gloc_scan_store* st, DevScan& s, bool cache_block);
// Re-sort an indexed scan into kd order (target index) and rebuild everything that depends on the order.
// Caller holds store->mu; nothing may be reading the scan; returns after the work has completed.
int store_build_target_index(gloc_scan_store* st, DevScan& s);
int store_build_target_indices(gloc_scan_store* st, DevScan* const* scans, size_t count);  // batches of <= 8 M points
// Build (once) the launch order of a scan for `cs` source points per lane into its own array and point
// s.order at it.  Caller holds store->mu.
int store_build_order(gloc_scan_store* st, DevScan& s, int cs);
// Copy of scan `id` (by value: the table may grow under another thread) with `order` = the launch order
// for `cs` sources per lane (cs = 0: the scan is used as a target only, no order).  GLOC_ERR_INVALID if unknown.
int store_get(gloc_scan_store* st, uint32_t id, int cs, DevScan* out);
// A batch about to be enqueued takes the by-value views of the scans its jobs read AND pins them under ONE acquisition of
// the store's mutex (round 6; before, the views were taken first and pinned afterwards: a re-sort or a release by another
// thread in between left the batch with a stale view).  All or nothing: an unknown id or a failed launch-order build
// pins nothing.  ids may repeat (pins are counted).  gloc_scan_store_build_target_index refuses a pinned scan that still
// needs the re-sort, gloc_scan_store_release refuses any pinned scan.
int store_get_pinned(gloc_scan_store* st, const uint32_t* ids, const int* cs, size_t count, DevScan* out);
// The pins' release (delta = -1) once the batch's event has been waited for.  Ids no longer live are skipped.
void store_pin(gloc_scan_store* st, const uint32_t* ids, size_t count, int delta);

}  // namespace reg
}  // namespace gloc
