// Micro-benchmark: rate of v_mfma_f32_32x32x2_f32 when the GEMM main loop's other instruction classes ride along at the GEMM's ratio
// (per 32 MFMAs: 8 ds_read_b128, 4 ds_write_b64x2, 4 global_load_dwordx4, ~64 VALU), 3 workgroups per CU.
// Build: hipcc --offload-arch=gfx950 -O3 scratch/mfma_mix_bench.hip -o scratch/mmb
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
template <int LDSR, int LDSW, int VALU, int GLD, int BAR = 0, int SALU = 0>
__global__ __launch_bounds__(256, 3) void k(float* out, const float* in, int iters) {
  __shared__ __attribute__((aligned(16))) float S[2][128 * 20];
  f32x16 acc[4];
  for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
  const int tid = threadIdx.x;
  for (int i = tid; i < 2 * 128 * 20; i += 256) (&S[0][0])[i] = (float)i * 1e-6f;
  __syncthreads();
  float fa[8] = {1, 2, 3, 4, 5, 6, 7, 8}, fb[8] = {8, 7, 6, 5, 4, 3, 2, 1};
  float4 g[4];
  for (int i = 0; i < 4; ++i) g[i] = make_float4(0.f, 0.f, 0.f, 0.f);
  int vsum = tid;
  int ssum = iters;
  const float* ip = in + (size_t)blockIdx.x * 4096 + tid * 4;
  for (int it = 0; it < iters; ++it) {
    const int cur = it & 1;
    if (LDSR) {
      const float* p = &S[cur][(tid & 127) * 20 + 8 * (tid >> 7)];
      *reinterpret_cast<float4*>(&fa[0]) = *reinterpret_cast<const float4*>(p);
      *reinterpret_cast<float4*>(&fa[4]) = *reinterpret_cast<const float4*>(p + 4);
      *reinterpret_cast<float4*>(&fb[0]) = *reinterpret_cast<const float4*>(p + 640);
      *reinterpret_cast<float4*>(&fb[4]) = *reinterpret_cast<const float4*>(p + 644);
      if (LDSR > 1) {
        const float4 x = *reinterpret_cast<const float4*>(p + 1280), y = *reinterpret_cast<const float4*>(p + 1284);
        const float4 z = *reinterpret_cast<const float4*>(p + 1920), w = *reinterpret_cast<const float4*>(p + 1924);
        fa[0] += x.x + y.y; fb[0] += z.z + w.w;
      }
    }
    if (GLD) {
#pragma unroll
      for (int i = 0; i < 4; ++i) g[i] = *reinterpret_cast<const float4*>(ip + (size_t)((it * 4 + i) & 255) * 1048576 / 256);
    }
    if (LDSW) {
      float* q = &S[cur ^ 1][(tid >> 2) * 20 + 2 * (tid & 3)];
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        *reinterpret_cast<float2*>(q + i * 1280) = make_float2(g[i].x + fa[1], g[i].z);
        *reinterpret_cast<float2*>(q + i * 1280 + 8) = make_float2(g[i].y, g[i].w);
      }
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[j], fb[j], acc[0], 0, 0, 0);
      acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[j], fb[7 - j], acc[1], 0, 0, 0);
      acc[2] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[7 - j], fb[j], acc[2], 0, 0, 0);
      acc[3] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[7 - j], fb[7 - j], acc[3], 0, 0, 0);
      if (VALU) {
#pragma unroll
        for (int v = 0; v < VALU; ++v) vsum = (vsum * 3 + j + v) ^ (vsum >> 3);
      }
    }
    if (LDSW || BAR) __syncthreads();
    if (SALU) {
#pragma unroll
      for (int v = 0; v < SALU; ++v) asm volatile("s_mul_i32 %0, %0, 3" : "+s"(ssum) : : "scc");   // s_add would clobber the loop condition in SCC
    }
  }
  float s = (float)vsum + (float)ssum + g[0].x + g[1].y + g[2].z + g[3].w;
  for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
  out[blockIdx.x * 256 + tid] = s;
}
template <int A, int B, int C, int D, int E = 0, int F = 0>
void run(const char* name, float* out, float* in) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const int grid = 768, iters = 4000;
  for (int rep = 0; rep < 2; ++rep) {
    hipEventRecord(e0);
    hipLaunchKernelGGL((k<A, B, C, D, E, F>), dim3(grid), dim3(256), 0, 0, out, in, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    if (rep) printf("%-46s %.1f TFLOP/s\n", name, (double)grid * 4 * iters * 32 * 4096.0 / ms / 1e9);
  }
}
int main() {
  float *out, *in; hipMalloc(&out, 768 * 256 * 4); hipMalloc(&in, (size_t)768 * 4096 * 4 + (size_t)1048576 * 4 * 4);
  hipMemset(in, 0, (size_t)768 * 4096 * 4 + (size_t)1048576 * 4 * 4);
  run<0, 0, 0, 0>("MFMA only (3 WGs/CU)", out, in);
  run<1, 0, 0, 0>("+ 4 ds_read_b128 per 32 MFMA", out, in);
  run<2, 0, 0, 0>("+ 8 ds_read_b128 per 32 MFMA", out, in);
  run<0, 0, 2, 0>("+ 2x3 VALU per 4 MFMA", out, in);
  run<0, 0, 4, 0>("+ 4x3 VALU per 4 MFMA", out, in);
  run<0, 0, 0, 1>("+ 4 global_load_dwordx4 per 32 MFMA", out, in);
  run<2, 1, 0, 1>("+ ds_read + global + ds_write + barrier", out, in);
  run<2, 1, 2, 1>("+ everything", out, in);
  run<0, 0, 0, 0, 1>("+ barrier per 32 MFMA only", out, in);
  run<0, 1, 0, 0>("+ ds_write + barrier", out, in);
  run<0, 1, 0, 1>("+ global + ds_write + barrier", out, in);
  run<0, 0, 0, 0, 0, 32>("+ 32 SALU per 32 MFMA", out, in);
  run<2, 1, 0, 1, 0, 32>("+ ds_read + global + ds_write + barrier + SALU", out, in);
  return 0;
}
