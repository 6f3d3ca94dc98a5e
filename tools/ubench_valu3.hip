// ubench_valu3.hip -- issue cost of the instruction forms the culled 1-NN kernel is made of (gfx950), 8 independent chains.
#include <hip/hip_runtime.h>
#include <cstdio>
#define CHAINS(BODY)                                                                                  \
  for (int i = 0; i < iters; ++i) {                                                                   \
    _Pragma("unroll") for (int u = 0; u < 16; ++u) { BODY(a0) BODY(a1) BODY(a2) BODY(a3) BODY(a4) BODY(a5) BODY(a6) BODY(a7) } \
  }
#define KERNEL32(name, BODY)                                                                          \
  __global__ void k_##name(float* out, int iters) {                                                   \
    float a0 = threadIdx.x * 1e-3f, a1 = a0 + 1.f, a2 = a0 + 2.f, a3 = a0 + 3.f, a4 = a0 + 4.f, a5 = a0 + 5.f, a6 = a0 + 6.f, a7 = a0 + 7.f; \
    const float c = 0.999f, d = 1.001f;                                                               \
    CHAINS(BODY)                                                                                      \
    out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;               \
  }
#define B_MAX3(x) asm volatile("v_max3_f32 %0, %0, %1, %2" : "+v"(x) : "v"(c), "v"(d));
#define B_MIN3(x) asm volatile("v_min3_f32 %0, %0, %1, %2" : "+v"(x) : "v"(c), "v"(d));
#define B_FMA(x) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x) : "v"(c), "v"(d));
#define B_CNDMASK(x) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(x) : "v"(c));
#define B_CMP(x) asm volatile("v_cmp_lt_f32 vcc, %0, %1" : : "v"(x), "v"(c) : "vcc");
#define B_LSHLADD(x) asm volatile("v_lshl_add_u32 %0, %0, 1, %1" : "+v"(x) : "v"(c));
#define B_MBCNT(x) asm volatile("v_mbcnt_lo_u32_b32 %0, -1, %0" : "+v"(x));
#define B_DPP(x) asm volatile("v_mov_b32_dpp %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf" : "+v"(x));
#define B_ADDDPP(x) asm volatile("v_add_f32_dpp %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf" : "+v"(x));
#define B_READLANE(x) asm volatile("v_readlane_b32 s20, %0, 3" : : "v"(x) : "s20");
#define B_OR(x) asm volatile("v_or_b32 %0, %0, %1" : "+v"(x) : "v"(c));
#define B_CVT(x) asm volatile("v_cvt_u32_f32 %0, %0" : "+v"(x));
KERNEL32(max3, B_MAX3) KERNEL32(min3, B_MIN3) KERNEL32(fma, B_FMA) KERNEL32(cndmask, B_CNDMASK) KERNEL32(cmp, B_CMP)
KERNEL32(lshladd, B_LSHLADD) KERNEL32(mbcnt, B_MBCNT) KERNEL32(dpp, B_DPP) KERNEL32(adddpp, B_ADDDPP) KERNEL32(readlane, B_READLANE)
KERNEL32(orb, B_OR) KERNEL32(cvt, B_CVT)
#define KERNEL64(name, BODY)                                                                          \
  __global__ void k_##name(float* out, int iters) {                                                   \
    double a0 = threadIdx.x * 1e-3, a1 = a0 + 1., a2 = a0 + 2., a3 = a0 + 3., a4 = a0 + 4., a5 = a0 + 5., a6 = a0 + 6., a7 = a0 + 7.; \
    const double c = 0.999, d = 1.001;                                                                \
    CHAINS(BODY)                                                                                      \
    out[blockIdx.x * blockDim.x + threadIdx.x] = (float)(a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7);     \
  }
#define B_ADD64(x) asm volatile("v_add_f64 %0, %0, %1" : "+v"(x) : "v"(c));
#define B_MUL64(x) asm volatile("v_mul_f64 %0, %0, %1" : "+v"(x) : "v"(c));
#define B_FMA64(x) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(x) : "v"(c), "v"(d));
#define B_PKADD(x) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(x) : "v"(c));
#define B_PKMUL(x) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(x) : "v"(c));
#define B_CMP64(x) asm volatile("v_cmp_lt_u64 vcc, %0, %1" : : "v"(x), "v"(c) : "vcc");
KERNEL64(add64, B_ADD64) KERNEL64(mul64, B_MUL64) KERNEL64(fma64, B_FMA64) KERNEL64(pkadd, B_PKADD) KERNEL64(pkmul, B_PKMUL) KERNEL64(cmp64, B_CMP64)

template <class K>
void run(const char* name, K kern, int w) {
  const int threads = 256, blocks = 256 * w;
  float* out; (void)hipMalloc(&out, sizeof(float) * threads * blocks);
  const int iters = 4000;
  hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b);
  kern<<<blocks, threads>>>(out, 100); (void)hipDeviceSynchronize();
  (void)hipEventRecord(a); kern<<<blocks, threads>>>(out, iters); (void)hipEventRecord(b); (void)hipEventSynchronize(b);
  float ms; (void)hipEventElapsedTime(&ms, a, b);
  const double instr = (double)iters * 128 * (blocks * threads / 64);
  printf("%-18s waves/SIMD=%d %.3f ms cycles/instr/SIMD(@2.4GHz)=%.2f\n", name, w, ms, (ms * 1e-3) * 1024.0 * 2.4e9 / instr);
  (void)hipFree(out);
}
int main() {
  for (int w : {4}) {
    run("v_max3_f32", k_max3, w); run("v_min3_f32", k_min3, w); run("v_fma_f32", k_fma, w); run("v_cndmask_b32", k_cndmask, w);
    run("v_cmp_lt_f32", k_cmp, w); run("v_lshl_add_u32", k_lshladd, w); run("v_mbcnt_lo", k_mbcnt, w); run("v_mov_b32_dpp", k_dpp, w);
    run("v_add_f32_dpp", k_adddpp, w); run("v_readlane_b32", k_readlane, w); run("v_or_b32", k_orb, w); run("v_cvt_u32_f32", k_cvt, w);
    run("v_add_f64", k_add64, w); run("v_mul_f64", k_mul64, w); run("v_fma_f64", k_fma64, w); run("v_pk_add_f32", k_pkadd, w);
    run("v_pk_mul_f32", k_pkmul, w); run("v_cmp_lt_u64", k_cmp64, w);
  }
  return 0;
}
