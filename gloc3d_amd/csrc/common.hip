// common.hip -- error reporting, device selection, device buffers, HIP-event profiler.
#include "common.hpp"

namespace gloc {

char* err_buf() {
  static thread_local char buf[512] = {0};
  return buf;
}

void set_err(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(err_buf(), 512, fmt, ap);
  va_end(ap);
}

int select_device(int device) {
  int n = 0;
  hipError_t e = hipGetDeviceCount(&n);
  if (e != hipSuccess || n <= 0) {
    set_err("no HIP device visible (%s): the gloc3d hot path has no CPU fallback",
            e == hipSuccess ? "device count 0" : hipGetErrorString(e));
    return GLOC_ERR_NODEVICE;
  }
  if (device < 0 || device >= n) {
    set_err("device ordinal %d out of range [0,%d)", device, n);
    return GLOC_ERR_INVALID;
  }
  hipDeviceProp_t prop;
  GLOC_HIP(hipGetDeviceProperties(&prop, device));
  if (strncmp(prop.gcnArchName, "gfx950", 6) != 0) {
    set_err("device %d is %s; this library is built for gfx950 (MI355X) only", device,
            prop.gcnArchName);
    return GLOC_ERR_NODEVICE;
  }
  GLOC_HIP(hipSetDevice(device));
  return GLOC_OK;
}

int DevBuf::ensure(size_t bytes, hipStream_t s, bool keep, size_t used) {
  if (bytes <= cap) return GLOC_OK;
  size_t ncap = cap ? cap : 256;
  while (ncap < bytes) ncap += ncap / 2 + 256;
  void* np = nullptr;
  GLOC_HIP(hipMalloc(&np, ncap));
  if (keep && p && used) {
    hipError_t e = hipMemcpyAsync(np, p, used, hipMemcpyDeviceToDevice, s);
    if (e == hipSuccess) e = hipStreamSynchronize(s);
    if (e != hipSuccess) {
      (void)hipFree(np);  // the old buffer stays valid
      set_err("device buffer growth failed: %s", hipGetErrorString(e));
      return GLOC_ERR_HIP;
    }
  }
  if (p) GLOC_HIP(hipFree(p));
  p = np;
  cap = ncap;
  return GLOC_OK;
}

void DevBuf::release() {
  if (p) (void)hipFree(p);
  p = nullptr;
  cap = 0;
}

hipEvent_t Profiler::get_event() {
  if (!pool.empty()) {
    hipEvent_t e = pool.back();
    pool.pop_back();
    return e;
  }
  hipEvent_t e;
  (void)hipEventCreate(&e);
  return e;
}

void Profiler::begin(const char* name, hipStream_t s) {
  Family& f = fam[name];
  Span sp{get_event(), get_event()};
  (void)hipEventRecord(sp.a, s);
  f.open.push_back(sp);
}

void Profiler::end(const char* name, hipStream_t s) {
  Family& f = fam[name];
  if (f.open.empty()) return;
  (void)hipEventRecord(f.open.back().b, s);
  f.launches++;
  // bound the number of live events: fold when many are open
  size_t live = 0;
  for (auto& kv : fam) live += kv.second.open.size();
  if (live > 8192) (void)collect(s);
}

int Profiler::collect(hipStream_t s) {
  GLOC_HIP(hipStreamSynchronize(s));
  for (auto& kv : fam) {
    for (Span& sp : kv.second.open) {
      float ms = 0.f;
      if (hipEventElapsedTime(&ms, sp.a, sp.b) == hipSuccess) kv.second.total_ms += ms;
      pool.push_back(sp.a);
      pool.push_back(sp.b);
    }
    kv.second.open.clear();
  }
  return GLOC_OK;
}

void Profiler::reset() {
  for (auto& kv : fam) {
    for (Span& sp : kv.second.open) {
      pool.push_back(sp.a);
      pool.push_back(sp.b);
    }
    kv.second.open.clear();
    kv.second.total_ms = 0;
    kv.second.launches = 0;
  }
}

void Profiler::destroy() {
  reset();
  for (hipEvent_t e : pool) (void)hipEventDestroy(e);
  pool.clear();
  fam.clear();
}

}  // namespace gloc

extern "C" {

const char* gloc_last_error(void) { return gloc::err_buf(); }
int gloc_abi_version(void) { return GLOC3D_ABI_VERSION; }
int gloc_device_count(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) return 0;
  return n;
}

}  // extern "C"
