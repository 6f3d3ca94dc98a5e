// dev: how fast does the chip start one-wave work-groups?  The culled 1-NN launch of 500 jobs is 483 000 work-groups of
// ONE wave (6.3 KB of LDS, 78 VGPRs) that live ~20 us each: if the dispatcher cannot refill 6 144 wave slots that fast,
// the launch is bound by it and not by the waves' instructions.  A work-group here owns LDS bytes and registers like the
// real one and spins on s_memrealtime (100 MHz) for a given time; printed: work-groups per us, and the launch time
// against slots x time.  hipcc --offload-arch=gfx950 -O3 tools/dev_dispatch_rate.hip -o tools/_bin/dispatch_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

template <int LDS_BYTES, int WAVES>
__global__ __launch_bounds__(64 * WAVES) __attribute__((amdgpu_waves_per_eu(6, 6))) void spin(unsigned ticks, unsigned* out, unsigned per_wg) {
  __shared__ unsigned char lds[LDS_BYTES];
  const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
  lds[threadIdx.x] = (unsigned char)threadIdx.x;
  unsigned n = 0;
  for (unsigned it = 0; it < per_wg; ++it) {  // per_wg > 1: a "persistent" wave doing several items back to back
    const unsigned long long t1 = __builtin_amdgcn_s_memrealtime();
    while (__builtin_amdgcn_s_memrealtime() - t1 < ticks) {
      __builtin_amdgcn_s_sleep(4);
      ++n;
    }
  }
  if (n == 0xFFFFFFFFu) out[blockIdx.x] = lds[(threadIdx.x + 1) & 63] + (unsigned)t0;
}

template <int LDS_BYTES, int WAVES>
static void run(const char* name, unsigned wgs, unsigned ticks, unsigned per_wg, unsigned* d) {
  hipEvent_t a, b;
  hipEventCreate(&a);
  hipEventCreate(&b);
  for (int rep = 0; rep < 2; ++rep) {
    hipEventRecord(a);
    hipLaunchKernelGGL((spin<LDS_BYTES, WAVES>), dim3(wgs), dim3(64 * WAVES), 0, 0, ticks, d, per_wg);
    hipEventRecord(b);
    hipEventSynchronize(b);
  }
  float ms = 0;
  hipEventElapsedTime(&ms, a, b);
  const double waves = (double)wgs * WAVES;
  const double ideal_ms = waves * per_wg * (ticks / 100.0) / 6144.0 / 1000.0;  // 256 CUs x 24 wave slots
  std::printf("%-28s wgs %7u x %d waves, %2u items of %5.1f us each: %8.3f ms = %7.1f wg/us; slots x time = %7.3f ms (%.2f of the launch)\n",
              name, wgs, WAVES, per_wg, ticks / 100.0, ms, wgs / (ms * 1000.0), ideal_ms, ideal_ms / ms);
  hipEventDestroy(a);
  hipEventDestroy(b);
}

int main() {
  unsigned* d = nullptr;
  hipMalloc(&d, sizeof(unsigned) << 20);
  const unsigned N = 483000;
  for (unsigned ticks : {0u, 200u, 500u, 1000u, 2000u}) {
    run<6336, 1>("lds 6336, 1 wave", N, ticks, 1, d);
    run<64, 1>("lds   64, 1 wave", N, ticks, 1, d);
    run<12672, 2>("lds 12672, 2 waves", N / 2, ticks, 1, d);
    run<25344, 4>("lds 25344, 4 waves", N / 4, ticks, 1, d);
  }
  // the same work as 6 144 resident waves each doing its share of the items
  for (unsigned ticks : {500u, 1000u, 2000u}) {
    run<6336, 1>("persistent: 6144 waves", 6144, ticks, N / 6144, d);
    run<6336, 1>("2 items per wave", N / 2, ticks, 2, d);
    run<6336, 1>("4 items per wave", N / 4, ticks, 4, d);
    run<6336, 1>("8 items per wave", N / 8, ticks, 8, d);
  }
  hipFree(d);
  return 0;
}
