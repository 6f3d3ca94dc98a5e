// dev: how fast does HBM stream a [rows][4096] fp32 table when every work-group walks R rows in pieces of C bytes per row
// and step (the access shape of an LDS-tiled distance kernel), against whole 1-KiB row pieces?  Loads only + a checksum.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
// work-group of 256 threads owns R consecutive rows; per step every row contributes C bytes (C / 16 lanes per row);
// DEPTH loads in flight per thread
template <int DEPTH>
__global__ __launch_bounds__(256) void walk(const float* __restrict__ db, int dim, int R, int C, float* out) {
  const int tid = threadIdx.x;
  const int lanes_per_row = C / 16;                 // threads covering one row piece
  const int rows_per_pass = 256 / lanes_per_row;    // rows covered by one load instruction of the work-group
  const int passes = R / rows_per_pass;             // load instructions per step
  const size_t row0 = (size_t)blockIdx.x * R;
  const int steps = dim * 4 / C;
  f32x4 acc = {0, 0, 0, 0};
  const int r_in = tid / lanes_per_row, c_in = (tid % lanes_per_row) * 4;
  // flatten (step, pass) into one sequence of loads with DEPTH in flight
  const int total = steps * passes;
  f32x4 ring[DEPTH];
#pragma unroll
  for (int i = 0; i < DEPTH; ++i) {
    const int st = i / passes, ps = i % passes;
    ring[i] = *reinterpret_cast<const f32x4*>(db + (row0 + ps * rows_per_pass + r_in) * dim + st * (C / 4) + c_in);
  }
  for (int i = 0; i < total; i += DEPTH) {
#pragma unroll
    for (int j = 0; j < DEPTH; ++j) {
      acc += ring[j];
      const int n = i + j + DEPTH;
      if (n < total) {
        const int st = n / passes, ps = n % passes;
        ring[j] = *reinterpret_cast<const f32x4*>(db + (row0 + ps * rows_per_pass + r_in) * dim + st * (C / 4) + c_in);
      }
    }
  }
  if (acc.x + acc.y + acc.z + acc.w == 12345.678f) out[0] = 1.f;
}
int main(int argc, char** argv) {
  const int dim = 4096;
  const size_t rows = 125000 / 128 * 128;
  float* db; float* out;
  hipMalloc(&db, rows * dim * 4); hipMalloc(&out, 4);
  hipMemset(db, 0, rows * dim * 4);
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  const int Rs[] = {128, 64, 32, 16, 8};
  const int Cs[] = {128, 256, 512, 1024, 4096};
  for (int R : Rs) for (int C : Cs) {
    if (256 / (C / 16) > R || C / 16 > 256) continue;
    const int grid = rows / R;
    for (int rep = 0; rep < 2; ++rep) hipLaunchKernelGGL(walk<12>, dim3(grid), dim3(256), 0, 0, db, dim, R, C, out);
    hipEventRecord(a);
    for (int rep = 0; rep < 5; ++rep) hipLaunchKernelGGL(walk<12>, dim3(grid), dim3(256), 0, 0, db, dim, R, C, out);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b); ms /= 5;
    printf("rows/WG %4d  bytes/row/step %5d  WGs %5d : %7.1f us  %.2f TB/s\n", R, C, grid, ms * 1e3, rows * dim * 4.0 / ms / 1e9);
  }
  return 0;
}
