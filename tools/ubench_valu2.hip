// ubench_valu2.hip -- single-opcode VALU issue-rate probe (gfx950), 8 independent chains.
#include <hip/hip_runtime.h>
#include <cstdio>
#define OP2(name, ins)                                                                          \
  __global__ void k_##name(float* out, int iters) {                                             \
    float a0 = threadIdx.x * 1e-3f, a1 = a0 + 1.f, a2 = a0 + 2.f, a3 = a0 + 3.f, a4 = a0 + 4.f,  \
          a5 = a0 + 5.f, a6 = a0 + 6.f, a7 = a0 + 7.f;                                           \
    const float c = 0.999f;                                                                     \
    for (int i = 0; i < iters; ++i) {                                                           \
      _Pragma("unroll") for (int u = 0; u < 16; ++u) {                                          \
        asm volatile(ins " %0, %0, %8\n" ins " %1, %1, %8\n" ins " %2, %2, %8\n" ins            \
                         " %3, %3, %8\n" ins " %4, %4, %8\n" ins " %5, %5, %8\n" ins            \
                         " %6, %6, %8\n" ins " %7, %7, %8\n"                                    \
                     : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6),    \
                       "+v"(a7)                                                                 \
                     : "v"(c));                                                                 \
      }                                                                                         \
    }                                                                                           \
    out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;         \
  }
OP2(mul, "v_mul_f32")
OP2(add, "v_add_f32")
OP2(sub, "v_sub_f32")
OP2(minf, "v_min_f32")
OP2(maxf, "v_max_f32")
OP2(minu, "v_min_u32")
OP2(addu, "v_add_u32")
OP2(andb, "v_and_b32")
OP2(lshl, "v_lshlrev_b32")

// the point-NN inner body: 3 sub, 3 mul, 2 add, min  (per target, S sources per lane)
__global__ void k_nnmix(float* out, int iters, float qx, float qy, float qz) {
  float px[4], py[4], pz[4], m[4];
  for (int s = 0; s < 4; ++s) { px[s] = threadIdx.x * 1e-3f + s; py[s] = px[s] * 0.5f; pz[s] = px[s] * 0.25f; m[s] = 1e30f; }
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int u = 0; u < 16; ++u) {
      const float tx = qx + u, ty = qy + i, tz = qz;
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        const float dx = px[s] - tx, dy = py[s] - ty, dz = pz[s] - tz;
        const float d = (dx * dx + dy * dy) + dz * dz;
        m[s] = fminf(m[s], d);
      }
    }
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = m[0] + m[1] + m[2] + m[3];
}

template <class K, class... A>
void run(const char* name, K kern, double ops_per_iter, int w, A... args) {
  const int threads = 256, blocks = 256 * w;
  float* out; (void)hipMalloc(&out, sizeof(float) * threads * blocks);
  const int iters = 10000;
  hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b);
  kern<<<blocks, threads>>>(out, 100, args...); (void)hipDeviceSynchronize();
  (void)hipEventRecord(a); kern<<<blocks, threads>>>(out, iters, args...); (void)hipEventRecord(b); (void)hipEventSynchronize(b);
  float ms; (void)hipEventElapsedTime(&ms, a, b);
  const double instr = (double)iters * ops_per_iter * (blocks * threads / 64);
  printf("%-14s waves/SIMD=%d %.3f ms cycles/instr/SIMD(@2.4GHz)=%.2f\n", name, w, ms, (ms * 1e-3) * 1024.0 * 2.4e9 / instr);
  (void)hipFree(out);
}

int main() {
  for (int w : {2, 8}) {
    run("v_mul_f32", k_mul, 128, w); run("v_add_f32", k_add, 128, w); run("v_sub_f32", k_sub, 128, w);
    run("v_min_f32", k_minf, 128, w); run("v_max_f32", k_maxf, 128, w); run("v_min_u32", k_minu, 128, w);
    run("v_add_u32", k_addu, 128, w); run("v_and_b32", k_andb, 128, w); run("v_lshlrev_b32", k_lshl, 128, w);
    run("nnmix(9/pair)", k_nnmix, 16 * 4 * 9, w, 1.f, 2.f, 3.f);
  }
  return 0;
}
