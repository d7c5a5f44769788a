// Does hipExtStreamCreateWithCUMask confine a kernel to the masked CUs on this stack?  (A stream-K GEMM of 768 workgroups on a
// stream masked to 64 of 256 CUs should take ~4x as long; with the complement mask on a second stream both should overlap.)
// Build: hipcc --offload-arch=gfx950 -O3 scratch/cumask_probe.hip -Iinclude -Last_amd -lastk -Wl,-rpath,'$ORIGIN/../ast_amd' -o scratch/cm
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include "astk.h"
static float time_gemm(hipStream_t s, float* A, float* B, float* C, int M, int N, int K) {
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  float best = 1e9;
  for (int rep = 0; rep < 4; ++rep) {
    (void)hipEventRecord(e0, s);
    astk_gemm_f32(1, M, N, K, A, K, B, N, C, N, nullptr, 0, 1, 1, 0, 0, 0, s);
    (void)hipEventRecord(e1, s);
    (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    if (rep && ms < best) best = ms;
  }
  return best * 1e3f;
}
int main() {
  const int M = 6400, N = 3072, K = 1024;
  float *A, *B, *C;
  (void)hipMalloc(&A, (size_t)M * K * 4); (void)hipMalloc(&B, (size_t)K * N * 4); (void)hipMalloc(&C, (size_t)M * N * 4);
  (void)hipMemset(A, 0, (size_t)M * K * 4); (void)hipMemset(B, 0, (size_t)K * N * 4);
  hipStream_t s0; (void)hipStreamCreate(&s0);
  printf("unmasked stream: %.1f us\n", time_gemm(s0, A, B, C, M, N, K));
  for (int variant = 0; variant < 3; ++variant) {
    std::vector<uint32_t> mask(8, 0);      // 256 bits
    const char* name = "";
    if (variant == 0) { name = "bits 0..63"; mask[0] = mask[1] = 0xffffffffu; }
    if (variant == 1) { name = "bits 192..255"; mask[6] = mask[7] = 0xffffffffu; }
    if (variant == 2) { name = "every 4th bit"; for (auto& m : mask) m = 0x11111111u; }
    hipStream_t sm;
    hipError_t e = hipExtStreamCreateWithCUMask(&sm, (uint32_t)mask.size(), mask.data());
    if (e != hipSuccess) { printf("mask %s: create failed: %s\n", name, hipGetErrorString(e)); continue; }
    printf("mask %-14s (64 CUs): %.1f us\n", name, time_gemm(sm, A, B, C, M, N, K));
    (void)hipStreamDestroy(sm);
  }
  return 0;
}
