// Probe (round 4): is r = x - bf16(x) exact through v_dot2(c)_f32_bf16 (hi_pk . (-1, 0) + x), so that the bf16x3 split needs no
// shift / and to widen the packed terms again?   hipcc --offload-arch=gfx950 -O2 -o scratch/dot2_probe scratch/dot2_split_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <cmath>
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32pair __attribute__((ext_vector_type(2)));
__device__ __forceinline__ unsigned cvt_pk(float a, float b) { return __builtin_bit_cast(unsigned, __builtin_convertvector((f32pair){a, b}, bf16x2)); }
__global__ void k(const float* x, int n, const unsigned* masks, unsigned* out) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (2 * i + 1 >= n) return;
  const float x0 = x[2 * i], x1 = x[2 * i + 1];
  const unsigned hi = cvt_pk(x0, x1);
  // reference: widen and subtract
  const float r0 = x0 - __uint_as_float(hi << 16), r1 = x1 - __uint_as_float(hi & 0xffff0000u);
  // dot2 with masks from memory (no inline-constant ambiguity)
  const bf16x2 h = __builtin_bit_cast(bf16x2, hi);
  const float d0 = __builtin_amdgcn_fdot2_f32_bf16(h, __builtin_bit_cast(bf16x2, masks[0]), x0, false);
  const float d1 = __builtin_amdgcn_fdot2_f32_bf16(h, __builtin_bit_cast(bf16x2, masks[1]), x1, false);
  // ... and with literal masks
  const float e0 = __builtin_amdgcn_fdot2_f32_bf16(h, __builtin_bit_cast(bf16x2, 0x0000bf80u), x0, false);
  const float e1 = __builtin_amdgcn_fdot2_f32_bf16(h, __builtin_bit_cast(bf16x2, 0xbf800000u), x1, false);
  out[8 * i + 0] = __float_as_uint(r0); out[8 * i + 1] = __float_as_uint(r1);
  out[8 * i + 2] = __float_as_uint(d0); out[8 * i + 3] = __float_as_uint(d1);
  out[8 * i + 4] = __float_as_uint(e0); out[8 * i + 5] = __float_as_uint(e1);
  out[8 * i + 6] = hi; out[8 * i + 7] = 0;
}
int main() {
  const int n = 1 << 20;
  float* hx = (float*)malloc(n * 4);
  srand(7);
  for (int i = 0; i < n; ++i) {
    unsigned bits = ((unsigned)rand() << 16) ^ (unsigned)rand() ^ ((unsigned)rand() << 31);
    unsigned e = (bits >> 23) & 0xff;
    if (i % 3 == 0) e = 100 + e % 60;            // ordinary magnitudes
    if (e == 255) e = 254;
    if (i % 1001 == 0) e = 1 + (i / 1001) % 12;   // near the bottom of the normal range
    bits = (bits & 0x807fffffu) | (e << 23);
    memcpy(&hx[i], &bits, 4);
  }
  float* dx; unsigned *dm, *dout;
  hipMalloc(&dx, n * 4); hipMalloc(&dm, 8); hipMalloc(&dout, (size_t)n * 16);
  const unsigned masks[2] = {0x0000bf80u, 0xbf800000u};
  hipMemcpy(dx, hx, n * 4, hipMemcpyHostToDevice); hipMemcpy(dm, masks, 8, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k, dim3(n / 2 / 256), dim3(256), 0, 0, dx, n, dm, dout);
  unsigned* ho = (unsigned*)malloc((size_t)n * 16);
  hipMemcpy(ho, dout, (size_t)n * 16, hipMemcpyDeviceToHost);
  long bad_d = 0, bad_e = 0, shown = 0;
  for (int i = 0; i < n / 2; ++i) {
    for (int j = 0; j < 2; ++j) {
      const unsigned r = ho[8 * i + j], d = ho[8 * i + 2 + j], e = ho[8 * i + 4 + j];
      if (d != r) { ++bad_d; if (shown++ < 8) printf("x %08x hi %08x ref %08x dot(mem) %08x dot(lit) %08x\n", *(unsigned*)&hx[2 * i + j], ho[8 * i + 6], r, d, e); }
      if (e != r) ++bad_e;
    }
  }
  printf("pairs %d: mismatches dot2(mask from memory) %ld, dot2(literal mask) %ld\n", n / 2, bad_d, bad_e);
  return 0;
}
