// common.hpp -- shared host-side plumbing for the gloc3d C ABI (error reporting, HIP checks,
// per-kernel HIP-event profiler, device buffers).  gfx950 only; no CPU fallback anywhere.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <map>
#include <string>
#include <vector>

#include "../../include/gloc3d.h"

namespace gloc {

char* err_buf();
void set_err(const char* fmt, ...);

#define GLOC_HIP(expr)                                                                    \
  do {                                                                                    \
    hipError_t e_ = (expr);                                                               \
    if (e_ != hipSuccess) {                                                               \
      ::gloc::set_err("%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__,    \
                      __LINE__);                                                          \
      return (e_ == hipErrorOutOfMemory) ? GLOC_ERR_NOMEM : GLOC_ERR_HIP;                 \
    }                                                                                     \
  } while (0)

#define GLOC_REQUIRE(cond, code, ...)  \
  do {                                 \
    if (!(cond)) {                     \
      ::gloc::set_err(__VA_ARGS__);    \
      return (code);                   \
    }                                  \
  } while (0)

#define GLOC_TRY(expr)            \
  do {                            \
    int rc_ = (expr);             \
    if (rc_ != GLOC_OK) return rc_; \
  } while (0)

int select_device(int device);  // validates ordinal + gfx950, hipSetDevice

// Growable device buffer (never shrinks).  Keeps contents on growth if `keep`.
struct DevBuf {
  void* p = nullptr;
  size_t cap = 0;
  int ensure(size_t bytes, hipStream_t s, bool keep = false, size_t used = 0);
  void release();
  template <class T>
  T* as() const {
    return reinterpret_cast<T*>(p);
  }
};

// HIP-event profiler: brackets kernel launches of one family and sums elapsed time lazily.
struct Profiler {
  bool enabled = false;
  struct Span {
    hipEvent_t a, b;
  };
  struct Family {
    std::vector<Span> open;
    double total_ms = 0;
    uint64_t launches = 0;
  };
  std::map<std::string, Family> fam;
  std::vector<hipEvent_t> pool;
  hipEvent_t get_event();
  void begin(const char* name, hipStream_t s);
  void end(const char* name, hipStream_t s);
  int collect(hipStream_t s);  // sync + fold open spans into totals
  void reset();
  void destroy();
};

struct ProfScope {
  Profiler& p;
  const char* name;
  hipStream_t s;
  ProfScope(Profiler& p_, const char* n, hipStream_t s_) : p(p_), name(n), s(s_) {
    if (p.enabled) p.begin(name, s);
  }
  ~ProfScope() {
    if (p.enabled) p.end(name, s);
  }
};

}  // namespace gloc
