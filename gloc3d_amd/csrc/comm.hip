// comm.hip -- gloc_comm_*: one RCCL communicator per process (= per GPU) for the row-sharded descriptor
// database of SURVEY.md 8e / BASELINE.json configs[4] (the reference is single-GPU; nothing to cite).
// Only two collectives exist on the path: the all-gather of per-shard top-k lists and the all-gather of
// result tables -- both a few KiB, latency-bound; xGMI is point-to-point, so each is ONE fused launch.
#include <dlfcn.h>

#include <cstring>
#include <mutex>
#include <new>

#include "comm.hpp"
#include "common.hpp"

namespace {

// the slice of the RCCL API this library uses (rccl.h of ROCm 7.2)
typedef struct { char internal[128]; } NcclUniqueId;
typedef int (*fn_get_id)(NcclUniqueId*);
typedef int (*fn_init_rank)(void**, int, NcclUniqueId, int);
typedef int (*fn_destroy)(void*);
typedef int (*fn_allgather)(const void*, void*, size_t, int /*ncclDataType_t*/, void*, hipStream_t);
typedef int (*fn_group)(void);
typedef const char* (*fn_errstr)(int);

struct Rccl {
  void* lib = nullptr;
  fn_get_id get_id = nullptr;
  fn_init_rank init_rank = nullptr;
  fn_destroy destroy = nullptr;
  fn_allgather allgather = nullptr;
  fn_group group_start = nullptr, group_end = nullptr;
  fn_errstr errstr = nullptr;
};

Rccl* rccl() {
  static Rccl r;
  static std::once_flag once;  // two host threads may make their first gloc_comm_* / sharded call together
  std::call_once(once, [] {
    // a copy already in the process first (PyTorch's), then GLOC3D_RCCL, then the system's
    const char* env = getenv("GLOC3D_RCCL");
    const char* names[] = {env, "librccl.so", "librccl.so.1", "/opt/rocm/lib/librccl.so"};
    r.lib = dlopen("librccl.so", RTLD_NOW | RTLD_NOLOAD | RTLD_GLOBAL);
    for (int i = 0; !r.lib && i < 4; ++i)
      if (names[i]) r.lib = dlopen(names[i], RTLD_NOW | RTLD_GLOBAL);
    if (r.lib) {
      r.get_id = (fn_get_id)dlsym(r.lib, "ncclGetUniqueId");
      r.init_rank = (fn_init_rank)dlsym(r.lib, "ncclCommInitRank");
      r.destroy = (fn_destroy)dlsym(r.lib, "ncclCommDestroy");
      r.allgather = (fn_allgather)dlsym(r.lib, "ncclAllGather");
      r.group_start = (fn_group)dlsym(r.lib, "ncclGroupStart");
      r.group_end = (fn_group)dlsym(r.lib, "ncclGroupEnd");
      r.errstr = (fn_errstr)dlsym(r.lib, "ncclGetErrorString");
      if (!r.get_id || !r.init_rank || !r.destroy || !r.allgather || !r.group_start || !r.group_end) r.lib = nullptr;
    }
  });
  return r.lib ? &r : nullptr;
}

int nccl_fail(const char* what, int rc) {
  Rccl* r = rccl();
  gloc::set_err("%s failed: %s", what, (r && r->errstr) ? r->errstr(rc) : "RCCL error");
  return GLOC_ERR_HIP;
}

}  // namespace

namespace gloc {
namespace comm {

int group_begin() {
  Rccl* r = rccl();
  GLOC_REQUIRE(r, GLOC_ERR_STATE, "librccl.so could not be loaded");
  const int rc = r->group_start();
  return rc == 0 ? GLOC_OK : nccl_fail("ncclGroupStart", rc);
}

int group_end() {
  Rccl* r = rccl();
  GLOC_REQUIRE(r, GLOC_ERR_STATE, "librccl.so could not be loaded");
  const int rc = r->group_end();
  return rc == 0 ? GLOC_OK : nccl_fail("ncclGroupEnd", rc);
}

int all_gather(gloc_comm* c, const void* d_send, void* d_recv, size_t bytes, hipStream_t s) {
  Rccl* r = rccl();
  GLOC_REQUIRE(r && c && c->nccl, GLOC_ERR_STATE, "no RCCL communicator");
  const int rc = r->allgather(d_send, d_recv, bytes, 1 /* ncclUint8 */, c->nccl, s);
  return rc == 0 ? GLOC_OK : nccl_fail("ncclAllGather", rc);
}

}  // namespace comm
}  // namespace gloc

using namespace gloc;

extern "C" {

int gloc_comm_unique_id(uint8_t* id128) {
  GLOC_REQUIRE(id128, GLOC_ERR_INVALID, "null argument");
  Rccl* r = rccl();
  GLOC_REQUIRE(r, GLOC_ERR_STATE, "librccl.so could not be loaded (set GLOC3D_RCCL)");
  NcclUniqueId u;
  const int rc = r->get_id(&u);
  if (rc != 0) return nccl_fail("ncclGetUniqueId", rc);
  std::memcpy(id128, u.internal, 128);
  return GLOC_OK;
}

int gloc_comm_create(int device, int rank, int world, const uint8_t* id128, gloc_comm** out) {
  GLOC_REQUIRE(out && id128, GLOC_ERR_INVALID, "null argument");
  *out = nullptr;
  GLOC_REQUIRE(world >= 1 && rank >= 0 && rank < world, GLOC_ERR_INVALID, "rank %d outside [0, %d)", rank, world);
  GLOC_TRY(select_device(device));
  Rccl* r = rccl();
  GLOC_REQUIRE(r, GLOC_ERR_STATE, "librccl.so could not be loaded (set GLOC3D_RCCL)");
  gloc_comm* c = new (std::nothrow) gloc_comm;
  GLOC_REQUIRE(c, GLOC_ERR_NOMEM, "host allocation failed");
  c->device = device;
  c->rank = rank;
  c->world = world;
  NcclUniqueId u;
  std::memcpy(u.internal, id128, 128);
  const int rc = r->init_rank(&c->nccl, world, u, rank);
  if (rc != 0) {
    delete c;
    return nccl_fail("ncclCommInitRank", rc);
  }
  *out = c;
  return GLOC_OK;
}

int gloc_comm_destroy(gloc_comm* c) {
  if (!c) return GLOC_OK;
  Rccl* r = rccl();
  // collectives of this communicator may still be in flight on streams this library does not own
  // (gloc_knn_search_sharded's handle stream, the caller's stream of all_gather_device): drain the device first
  if (hipSetDevice(c->device) == hipSuccess) (void)hipDeviceSynchronize();
  if (r && c->nccl) (void)r->destroy(c->nccl);
  delete c;
  return GLOC_OK;
}

int gloc_comm_rank(const gloc_comm* c, int* rank, int* world) {
  GLOC_REQUIRE(c, GLOC_ERR_INVALID, "null communicator");
  if (rank) *rank = c->rank;
  if (world) *world = c->world;
  return GLOC_OK;
}

int gloc_comm_all_gather_device(gloc_comm* c, const void* d_send, void* d_recv, size_t bytes_per_rank,
                                void* hip_stream) {
  GLOC_REQUIRE(c && d_send && d_recv, GLOC_ERR_INVALID, "null argument");
  GLOC_HIP(hipSetDevice(c->device));
  return comm::all_gather(c, d_send, d_recv, bytes_per_rank, (hipStream_t)hip_stream);
}

int gloc_comm_all_gather_host(gloc_comm* c, const void* send, void* recv, size_t bytes_per_rank) {
  GLOC_REQUIRE(c && send && recv, GLOC_ERR_INVALID, "null argument");
  GLOC_HIP(hipSetDevice(c->device));
  void* d = nullptr;
  const size_t total = bytes_per_rank * (size_t)(c->world + 1);
  GLOC_HIP(hipMalloc(&d, total ? total : 16));
  char* ds = static_cast<char*>(d);
  char* dr = ds + bytes_per_rank;
  int rc = GLOC_OK;
  hipError_t e = hipMemcpy(ds, send, bytes_per_rank, hipMemcpyHostToDevice);
  if (e == hipSuccess) {
    rc = comm::all_gather(c, ds, dr, bytes_per_rank, nullptr);  // (on failure its own message stays in place)
    if (rc == GLOC_OK) {
      e = hipStreamSynchronize(nullptr);
      if (e == hipSuccess) e = hipMemcpy(recv, dr, bytes_per_rank * (size_t)c->world, hipMemcpyDeviceToHost);
    }
  }
  if (e != hipSuccess) {
    (void)hipGetLastError();
    set_err("gloc_comm_all_gather_host: %s", hipGetErrorString(e));
    rc = GLOC_ERR_HIP;
  }
  (void)hipFree(d);
  return rc;
}

}  // extern "C"
