// Micro-benchmark: latency of an sc1 b128 load issued right after a store of each flavour (same wave), and of the
// store acknowledgement itself.  Build: hipcc --offload-arch=gfx950 -O3 scratch/vmem_order_bench.hip -o /tmp/vob
#include <hip/hip_runtime.h>
#include <cstdio>
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ __amdgpu_buffer_rsrc_t mk(const void* p) { return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, 0x7fffffff, 0x00020000); }
__global__ void k(float* buf, float* dst, long long* out, int mode, int iters, int gap) {
  const int tid = threadIdx.x;
  const __amdgpu_buffer_rsrc_t rs = mk(buf);
  long long t_issue = 0, t_wait = 0, t_ack = 0;
  unsigned acc = 0;
  for (int it = 0; it < iters; ++it) {
    const long o = ((long)it * 1024 + blockIdx.x * 65536 + tid) ;
    // the "previous step's" store
    if (mode == 1) dst[o] = (float)it;
    else if (mode == 2) __hip_atomic_store(dst + o, (float)it, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __builtin_amdgcn_sched_barrier(0);
    for (int g = 0; g < gap; ++g) __builtin_amdgcn_s_sleep(1);
    const long long a = wall_clock64();
    __builtin_amdgcn_sched_barrier(0);
    u32x4 v0 = __builtin_amdgcn_raw_buffer_load_b128(rs, (int)((o * 4 * 4) & 0x3ffffff0), 0, 16);
    u32x4 v1 = __builtin_amdgcn_raw_buffer_load_b128(rs, (int)((o * 4 * 4 + 4096) & 0x3ffffff0), 0, 16);
    __builtin_amdgcn_sched_barrier(0);
    const long long b = wall_clock64();
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const long long c = wall_clock64();
    acc += v0.x + v1.y;
    t_issue += b - a; t_wait += c - b;
    // ack alone: store then wait
    if (mode == 3) {
      __hip_atomic_store(dst + o + 512, (float)it, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      const long long d = wall_clock64();
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      t_ack += wall_clock64() - d;
    } else if (mode == 4) {
      dst[o + 512] = (float)it;
      const long long d = wall_clock64();
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      t_ack += wall_clock64() - d;
    }
  }
  if (tid == 0) { out[blockIdx.x * 4 + 0] = t_issue; out[blockIdx.x * 4 + 1] = t_wait; out[blockIdx.x * 4 + 2] = t_ack; out[blockIdx.x * 4 + 3] = acc; }
}
int main() {
  float *buf, *dst; long long* out;
  const size_t n = 64u << 20;
  hipMalloc(&buf, n * 4); hipMalloc(&dst, n * 4); hipMalloc(&out, 256 * 4 * 8);
  hipMemset(buf, 0, n * 4); hipMemset(dst, 0, n * 4);
  const char* names[] = {"no store", "plain store before", "sc1 store before", "(sc1 store ack measured)", "(plain store ack measured)"};
  for (int nb : {1, 192}) for (int gap : {0, 40}) for (int mode = 0; mode < 5; ++mode) {
    const int iters = 2000;
    hipLaunchKernelGGL(k, dim3(nb), dim3(256), 0, 0, buf, dst, out, mode, iters, gap);
    hipDeviceSynchronize();
    long long h[4]; hipMemcpy(h, out, sizeof(h), hipMemcpyDeviceToHost);
    printf("blocks %3d gap %2d mode %d %-28s: load issue %6.2f us  load wait %6.2f us  store ack %6.2f us\n", nb, gap, mode, names[mode],
           h[0] / (double)iters / 100, h[1] / (double)iters / 100, h[2] / (double)iters / 100);
  }
  return 0;
}
