// ubench_mfma.hip -- sustained fp32 MFMA rate on gfx950 (v_mfma_f32_16x16x4_f32), by accumulators.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
template <int NACC>
__global__ __launch_bounds__(256) void k(float* out, int iters, float a0, float b0) {
  f32x4 acc[NACC];
  for (int i = 0; i < NACC; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
  float a = a0 + threadIdx.x * 1e-3f, b = b0 - threadIdx.x * 1e-3f;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 8; ++u)
#pragma unroll
      for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[i], 0, 0, 0);
  }
  float s = 0;
  for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int NACC>
void run(int wg_per_cu, int iters) {
  float* out; (void)hipMalloc(&out, 256 * 256 * wg_per_cu * 4);
  hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b);
  k<NACC><<<256 * wg_per_cu, 256>>>(out, 10, 0.5f, 0.25f); (void)hipDeviceSynchronize();
  (void)hipEventRecord(a); k<NACC><<<256 * wg_per_cu, 256>>>(out, iters, 0.5f, 0.25f); (void)hipEventRecord(b); (void)hipEventSynchronize(b);
  float ms; (void)hipEventElapsedTime(&ms, a, b);
  const double flop = (double)iters * 8 * NACC * 2048.0 * (256.0 * wg_per_cu * 4);
  printf("NACC=%d waves/SIMD=%d iters=%d: %.3f ms  %.1f TFLOP/s\n", NACC, wg_per_cu, iters, ms, flop / (ms * 1e-3) / 1e12);
  (void)hipFree(out);
}
int main() {
  for (int iters : {200, 2000, 20000}) { run<1>(1, iters); run<2>(1, iters); run<5>(1, iters); run<5>(2, iters); run<8>(1, iters); }
  return 0;
}
