// How does the stream-K GEMM (static, even split of k-iterations over 768 workgroups) behave when another kernel holds some CUs --
// the situation under data parallelism, where RCCL's all-reduce kernels run beside the CNN/encoder backward?
// A spin kernel of W workgroups (one per CU, 1024 threads each so nothing else fits beside it... or 256 threads: shares the CU) runs on a second stream.
// Build: hipcc --offload-arch=gfx950 -O3 scratch/gemm_contention.hip -Iinclude -Last_amd -lastk -Wl,-rpath,$PWD/ast_amd -o scratch/gc
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include "astk.h"
__global__ void spin(long long ticks, int* sink) {
  const long long t0 = __builtin_readcyclecounter();
  long long t = t0;
  while (t - t0 < ticks) { t = __builtin_readcyclecounter(); }
  if (threadIdx.x == 0 && t == 42) sink[0] = 1;
}
int main() {
  const int M = 6400, N = 3072, K = 1024;   // dgrad shape of the encoder's first layer (NN)
  float *A, *B, *C; int* sink;
  hipMalloc(&A, (size_t)M * K * 4); hipMalloc(&B, (size_t)K * N * 4); hipMalloc(&C, (size_t)M * N * 4); hipMalloc(&sink, 4);
  hipMemset(A, 0, (size_t)M * K * 4); hipMemset(B, 0, (size_t)K * N * 4);
  hipStream_t s1, s2; hipStreamCreate(&s1); hipStreamCreate(&s2);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const int ws[] = {0, 8, 16, 32, 64};
  for (int threads : {256, 1024}) {
    for (int w : ws) {
      float best = 1e9;
      for (int rep = 0; rep < 5; ++rep) {
        if (w) hipLaunchKernelGGL(spin, dim3(w), dim3(threads), 0, s2, 400000LL * 100 / 100, sink);   // ~4 ms at 100 MHz counter
        hipEventRecord(e0, s1);
        for (int i = 0; i < 4; ++i) astk_gemm_f32(1, M, N, K, A, K, B, N, C, N, nullptr, 0, 1, 1, 0, 0, 0, s1);
        hipEventRecord(e1, s1);
        hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        hipDeviceSynchronize();
        if (rep && ms < best) best = ms;
      }
      printf("spin %2d WGs x %4d threads: GEMM %.1f us  (%.1f TFLOP/s)\n", w, threads, best * 250, 2.0 * M * N * K / (best / 4 * 1e-3) / 1e12);
    }
  }
  return 0;
}
