// ubench_valu4.hip -- issue cost of packed fp16 forms against the packed / scalar fp32 forms the culled 1-NN kernel's box
// tests are made of (gfx950), 8 independent chains, 4 waves per SIMD.
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
#define KERNEL64(name, BODY)                                                                          \
  __global__ void k_##name(float* out, int iters) {                                                   \
    double a0 = threadIdx.x * 1e-3, a1 = a0 + 1., a2 = a0 + 2., a3 = a0 + 3., a4 = a0 + 4., a5 = a0 + 5., a6 = a0 + 6., a7 = a0 + 7.; \
    const double c = 0.999, d = 1.001;                                                                \
    CHAINS(BODY)                                                                                      \
    out[blockIdx.x * blockDim.x + threadIdx.x] = (float)(a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7);     \
  }
#define B_PKADD16(x) asm volatile("v_pk_add_f16 %0, %0, %1" : "+v"(x) : "v"(c));
#define B_PKMUL16(x) asm volatile("v_pk_mul_f16 %0, %0, %1" : "+v"(x) : "v"(c));
#define B_PKMAX16(x) asm volatile("v_pk_max_f16 %0, %0, %1" : "+v"(x) : "v"(c));
#define B_PKFMA16(x) asm volatile("v_pk_fma_f16 %0, %0, %1, %2" : "+v"(x) : "v"(c), "v"(d));
#define B_CVTPK(x) asm volatile("v_cvt_pkrtz_f16_f32 %0, %0, %1" : "+v"(x) : "v"(c));
#define B_CVT16(x) asm volatile("v_cvt_f16_f32 %0, %0" : "+v"(x));
#define B_ADD32(x) asm volatile("v_add_f32 %0, %0, %1" : "+v"(x) : "v"(c));
#define B_SUB32(x) asm volatile("v_sub_f32 %0, %0, %1" : "+v"(x) : "v"(c));
#define B_MUL32(x) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(x) : "v"(c));
#define B_MAX32(x) asm volatile("v_max_f32 %0, %0, %1" : "+v"(x) : "v"(c));
#define B_MAX3(x) asm volatile("v_max3_f32 %0, %0, %1, %2" : "+v"(x) : "v"(c), "v"(d));
#define B_FMA(x) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x) : "v"(c), "v"(d));
#define B_FMACLAMP(x) asm volatile("v_fma_f32 %0, |%0|, %1, %2 clamp" : "+v"(x) : "v"(c), "v"(d));
#define B_SUBABS(x) asm volatile("v_sub_f32 %0, |%0|, %1 clamp" : "+v"(x) : "v"(c));
#define B_PKADD(x) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(x) : "v"(c));
#define B_PKMUL(x) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(x) : "v"(c));
#define B_PKFMA(x) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(x) : "v"(c), "v"(d));
#define B_CMP16(x) asm volatile("v_cmp_lt_f16 vcc, %0, %1" : : "v"(x), "v"(c) : "vcc");
#define B_MAX3_16(x) asm volatile("v_max3_f16 %0, %0, %1, %2" : "+v"(x) : "v"(c), "v"(d));
KERNEL32(pkadd16, B_PKADD16) KERNEL32(pkmul16, B_PKMUL16) KERNEL32(pkmax16, B_PKMAX16) KERNEL32(pkfma16, B_PKFMA16)
KERNEL32(cvtpk, B_CVTPK) KERNEL32(cvt16, B_CVT16) KERNEL32(add32, B_ADD32) KERNEL32(sub32, B_SUB32) KERNEL32(mul32, B_MUL32)
KERNEL32(max32, B_MAX32) KERNEL32(max3, B_MAX3) KERNEL32(fma, B_FMA) KERNEL32(cmp16, B_CMP16) KERNEL32(max3_16, B_MAX3_16)
KERNEL32(fmaclamp, B_FMACLAMP) KERNEL32(subabs, B_SUBABS)
KERNEL64(pkadd, B_PKADD) KERNEL64(pkmul, B_PKMUL) KERNEL64(pkfma, B_PKFMA)

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
  printf("%-22s waves/SIMD=%d %.3f ms cycles/instr/SIMD(@2.4GHz)=%.2f\n", name, w, ms, (ms * 1e-3) * 1024.0 * 2.4e9 / instr);
  (void)hipFree(out);
}
int main() {
  const int w = 4;
  run("v_pk_add_f16", k_pkadd16, w); run("v_pk_mul_f16", k_pkmul16, w); run("v_pk_max_f16", k_pkmax16, w); run("v_pk_fma_f16", k_pkfma16, w);
  run("v_cvt_pkrtz_f16_f32", k_cvtpk, w); run("v_cvt_f16_f32", k_cvt16, w); run("v_cmp_lt_f16", k_cmp16, w); run("v_max3_f16", k_max3_16, w);
  run("v_add_f32", k_add32, w); run("v_sub_f32", k_sub32, w); run("v_mul_f32", k_mul32, w); run("v_max_f32", k_max32, w);
  run("v_max3_f32", k_max3, w); run("v_fma_f32", k_fma, w);
  run("v_fma_f32 |a| clamp", k_fmaclamp, w); run("v_sub_f32 |a| clamp", k_subabs, w);
  run("v_pk_add_f32", k_pkadd, w); run("v_pk_mul_f32", k_pkmul, w); run("v_pk_fma_f32", k_pkfma, w);
  return 0;
}
