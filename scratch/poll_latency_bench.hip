// Micro-benchmark: how long after a remote write-through (sc1) store does a POLLING consumer see the value, as a function
// of the load flavour, when the consumer started polling long BEFORE the store (early poller) or only after it (late poller)?
// Build: hipcc --offload-arch=gfx950 -O3 scratch/poll_latency_bench.hip -o scratch/plb
#include <hip/hip_runtime.h>
#include <cstdio>
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ __amdgpu_buffer_rsrc_t mk(const void* p) { return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, 0x7fffffff, 0x00020000); }
// block 0 = producer, block `cons` = consumer.  slot = line index (separate 256-B lines per trial)
__global__ void k(unsigned* buf, long long* tstore, long long* tseen, int cons, int flavour, int early, int trials, int delay_us, int pstore) {
  const int tid = threadIdx.x;
  if (blockIdx.x != 0 && blockIdx.x != cons) return;
  const __amdgpu_buffer_rsrc_t rs = mk(buf);
  for (int it = 0; it < trials; ++it) {
    unsigned* line = buf + (size_t)it * 64;           // 256-B apart
    unsigned* go = buf + (size_t)(trials + it) * 64;  // start signal for this trial (written by consumer, read by producer)
    if (blockIdx.x == 0) {
      if (tid == 0) {
        while (__hip_atomic_load(go, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0) {}
        const long long t0 = wall_clock64();
        while (wall_clock64() - t0 < delay_us * 100) {}
        const long long ts = wall_clock64();
        if (pstore == 0) __hip_atomic_store(line, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);          // sc1 store
        else if (pstore == 1) __hip_atomic_store(line, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);    // sc0 sc1
        else atomicExch(line, 1u);                                                                          // device atomic
        tstore[it] = ts;
      }
    } else {
      if (tid == 0) {
        __hip_atomic_store(go, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (!early) { const long long t0 = wall_clock64(); while (wall_clock64() - t0 < (delay_us + 5) * 100) {} }   // start polling after the store
        unsigned v = 0;
        long long n = 0;
        while (v == 0 && n < (1 << 22)) {
          ++n;
          if (flavour == 0) v = __builtin_amdgcn_raw_buffer_load_b32(rs, (int)((size_t)it * 256), 0, 16);        // sc1
          else if (flavour == 1) v = __builtin_amdgcn_raw_buffer_load_b32(rs, (int)((size_t)it * 256), 0, 17);   // sc0 sc1
          else if (flavour == 2) v = __hip_atomic_load(line, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          else if (flavour == 3) v = __hip_atomic_load(line, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
          else if (flavour == 4) v = __builtin_amdgcn_raw_buffer_load_b32(rs, (int)((size_t)it * 256), 0, 18);   // nt sc1
          else v = atomicAdd(line, 0u);                                                                           // RMW poll
        }
        tseen[it] = wall_clock64();
      }
    }
  }
}
int main() {
  unsigned* buf; long long *ts, *tn;
  const int trials = 64;
  hipMalloc(&buf, (size_t)trials * 2 * 256); hipMalloc(&ts, trials * 8); hipMalloc(&tn, trials * 8);
  const char* fl[] = {"buffer sc1", "buffer sc0 sc1", "atomic_load agent", "atomic_load system", "buffer nt sc1", "atomicAdd(0)"};
  const char* ps[] = {"sc1 store", "sc0sc1 store", "atomicExch"};
  for (int cons : {1, 8}) for (int pstore = 0; pstore < 3; ++pstore) for (int early = 1; early >= 0; --early) for (int f = 0; f < 6; ++f) {
    hipMemset(buf, 0, (size_t)trials * 2 * 256);
    hipLaunchKernelGGL(k, dim3(16), dim3(64), 0, 0, buf, ts, tn, cons, f, early, trials, 20, pstore);
    hipDeviceSynchronize();
    long long a[trials], b[trials];
    hipMemcpy(a, ts, sizeof(a), hipMemcpyDeviceToHost); hipMemcpy(b, tn, sizeof(b), hipMemcpyDeviceToHost);
    double sum = 0, mx = 0, mn = 1e9;
    for (int i = 4; i < trials; ++i) { double d = (b[i] - a[i]) / 100.0; if (!early) d -= 5.0; sum += d; mx = d > mx ? d : mx; mn = d < mn ? d : mn; }
    printf("consumer block %d (%s XCD) %-13s %-5s poller %-18s: seen after store: mean %7.2f us  min %7.2f  max %7.2f%s\n", cons, cons == 8 ? "same" : "other",
           ps[pstore], early ? "EARLY" : "late", fl[f], sum / (trials - 4), mn, mx, early ? "" : "  (late: time from first poll, minus the 5 us head start)");
  }
  return 0;
}
