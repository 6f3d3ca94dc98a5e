// ubench_valu.hip -- VALU issue-rate probe for gfx950: scalar vs packed fp32 ops, by waves/SIMD.
// Build: hipcc --offload-arch=gfx950 -O3 tools/ubench_valu.hip -o gpurun_out/ubench_valu
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f2 __attribute__((ext_vector_type(2)));

template <int MODE>
__global__ void k(float* out, int iters) {
  float a0 = threadIdx.x * 1e-3f, a1 = a0 + 1.f, a2 = a0 + 2.f, a3 = a0 + 3.f;
  float a4 = a0 + 4.f, a5 = a0 + 5.f, a6 = a0 + 6.f, a7 = a0 + 7.f;
  f2 p0 = {a0, a1}, p1 = {a2, a3}, p2 = {a4, a5}, p3 = {a6, a7};
  f2 p4 = p0 + 1.f, p5 = p1 + 1.f, p6 = p2 + 1.f, p7 = p3 + 1.f;
  const float c = 0.999f, d = 1e-6f;
  const f2 c2 = {0.999f, 0.998f}, d2 = {1e-6f, 2e-6f};
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int u = 0; u < 16; ++u) {
      if (MODE == 0) {  // v_fma_f32 x8
        asm volatile("v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %1, %1, %8, %9\n v_fma_f32 %2, %2, %8, %9\n v_fma_f32 %3, %3, %8, %9\n"
                     "v_fma_f32 %4, %4, %8, %9\n v_fma_f32 %5, %5, %8, %9\n v_fma_f32 %6, %6, %8, %9\n v_fma_f32 %7, %7, %8, %9\n"
                     : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(c), "v"(d));
      } else if (MODE == 1) {  // v_pk_fma_f32 x8
        asm volatile("v_pk_fma_f32 %0, %0, %8, %9\n v_pk_fma_f32 %1, %1, %8, %9\n v_pk_fma_f32 %2, %2, %8, %9\n v_pk_fma_f32 %3, %3, %8, %9\n"
                     "v_pk_fma_f32 %4, %4, %8, %9\n v_pk_fma_f32 %5, %5, %8, %9\n v_pk_fma_f32 %6, %6, %8, %9\n v_pk_fma_f32 %7, %7, %8, %9\n"
                     : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3), "+v"(p4), "+v"(p5), "+v"(p6), "+v"(p7) : "v"(c2), "v"(d2));
      } else if (MODE == 2) {  // v_mul_f32 / v_add_f32 alternating x8
        asm volatile("v_mul_f32 %0, %0, %8\n v_add_f32 %1, %1, %9\n v_mul_f32 %2, %2, %8\n v_add_f32 %3, %3, %9\n"
                     "v_mul_f32 %4, %4, %8\n v_add_f32 %5, %5, %9\n v_mul_f32 %6, %6, %8\n v_add_f32 %7, %7, %9\n"
                     : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(c), "v"(d));
      } else if (MODE == 3) {  // v_pk_mul_f32 / v_pk_add_f32 alternating x8
        asm volatile("v_pk_mul_f32 %0, %0, %8\n v_pk_add_f32 %1, %1, %9\n v_pk_mul_f32 %2, %2, %8\n v_pk_add_f32 %3, %3, %9\n"
                     "v_pk_mul_f32 %4, %4, %8\n v_pk_add_f32 %5, %5, %9\n v_pk_mul_f32 %6, %6, %8\n v_pk_add_f32 %7, %7, %9\n"
                     : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3), "+v"(p4), "+v"(p5), "+v"(p6), "+v"(p7) : "v"(c2), "v"(d2));
      } else if (MODE == 4) {  // v_min_f32 x8
        asm volatile("v_min_f32 %0, %0, %8\n v_min_f32 %1, %1, %9\n v_min_f32 %2, %2, %8\n v_min_f32 %3, %3, %9\n"
                     "v_min_f32 %4, %4, %8\n v_min_f32 %5, %5, %9\n v_min_f32 %6, %6, %8\n v_min_f32 %7, %7, %9\n"
                     : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(c), "v"(d));
      } else if (MODE == 5) {  // v_min3_f32 x8
        asm volatile("v_min3_f32 %0, %0, %8, %9\n v_min3_f32 %1, %1, %8, %9\n v_min3_f32 %2, %2, %8, %9\n v_min3_f32 %3, %3, %8, %9\n"
                     "v_min3_f32 %4, %4, %8, %9\n v_min3_f32 %5, %5, %8, %9\n v_min3_f32 %6, %6, %8, %9\n v_min3_f32 %7, %7, %8, %9\n"
                     : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(c), "v"(d));
      }
    }
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + p0.x + p1.y + p2.x + p3.y + p4.x + p5.y + p6.x + p7.y;
}

template <int MODE>
void run(const char* name, int waves_per_simd) {
  const int threads = 256, blocks = 256 * waves_per_simd;  // 4 waves/block -> 1 wave/SIMD per block/CU
  float* out; hipMalloc(&out, sizeof(float) * threads * blocks);
  const int iters = 20000;
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  k<MODE><<<blocks, threads>>>(out, 100); hipDeviceSynchronize();
  hipEventRecord(a); k<MODE><<<blocks, threads>>>(out, iters); hipEventRecord(b); hipEventSynchronize(b);
  float ms; hipEventElapsedTime(&ms, a, b);
  const double instr = (double)iters * 16 * 8 * (blocks * threads / 64);  // wave-instructions
  const double per_simd_cycle = instr / (ms * 1e-3) / (1024.0 * 2.4e9);   // wave-instr per SIMD per cycle @2.4GHz
  printf("%-22s waves/SIMD=%d  %.3f ms  %.2f Twave-instr-lanes/s  cycles/instr/SIMD(@2.4GHz)=%.2f\n", name, waves_per_simd, ms,
         instr * 64 / (ms * 1e-3) / 1e12, 1.0 / per_simd_cycle);
  hipFree(out);
}

int main() {
  for (int w : {1, 2, 4, 8}) {
    run<0>("v_fma_f32", w); run<1>("v_pk_fma_f32", w); run<2>("v_mul/add_f32", w);
    run<3>("v_pk_mul/add_f32", w); run<4>("v_min_f32", w); run<5>("v_min3_f32", w);
  }
  return 0;
}
