// Micro-benchmark: peak rate of back-to-back v_mfma_f32_32x32x2_f32 (no memory traffic), 1-4 waves per SIMD, 4 independent accumulators.
// Build: hipcc --offload-arch=gfx950 -O3 scratch/mfma_peak_bench.hip -o scratch/mpb
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
template <int NACC>
__global__ __launch_bounds__(256) void k(float* out, int iters, float a0, float b0) {
  f32x16 acc[NACC];
  for (int i = 0; i < NACC; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
  float a = a0 + threadIdx.x, b = b0 - threadIdx.x;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 16; ++u)
#pragma unroll
      for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[i], 0, 0, 0);
  }
  float s = 0.f;
  for (int i = 0; i < NACC; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}
int main() {
  float* out; hipMalloc(&out, 4096 * 256 * 4);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int wgs_per_cu : {1, 2, 3, 4}) {
    const int grid = 256 * wgs_per_cu, iters = 2000;
    for (int rep = 0; rep < 2; ++rep) {
      hipEventRecord(e0);
      hipLaunchKernelGGL(k<4>, dim3(grid), dim3(256), 0, 0, out, iters, 1.f, 2.f);
      hipEventRecord(e1); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1);
      const double flops = (double)grid * 4 /*waves*/ * iters * 16 * 4 * 4096.0;
      if (rep) printf("%d workgroups/CU (x4 waves): %.3f ms  %.1f TFLOP/s\n", wgs_per_cu, ms, flops / ms / 1e9);
    }
  }
  return 0;
}
