// Micro-benchmark: what would XCD-local chains buy the decoder's hand-offs?
// Two groups of NG workgroups (one per CU) play ping-pong the way two consecutive decoder roles do: every workgroup of the sending group
// stores its slice (SLICE bytes) of a tile and arrives on a counter; every workgroup of the receiving group waits for the NG arrivals,
// reads the WHOLE tile (NG x SLICE bytes), reduces it, and the roles swap.  Time per one-way hand-off =
// wall time / (2 x round trips).  Variants:
//   placement: same  = both groups on ONE XCD (picked by HW_REG_XCC_ID, not by block index)
//              cross = the members of both groups dealt round-robin over the 8 XCDs
//   protocol : sc1   = write-through stores, agent-scope counter, sc1 polls and payload loads (what the decoder kernels use; valid everywhere)
//              l2    = plain stores, counter atomics without sc1, sc0 polls and payload loads (served by the XCD's own L2; valid ONLY when
//                      both groups share an XCD)
//   load     : idle  = the other workgroups exit
//              busy  = the other workgroups spin on a flag with sc1 loads, as the waiting roles of the decoder kernels do
// Build: hipcc --offload-arch=gfx950 -O3 scratch/xcd_hop_bench.hip -o scratch/xcd_hop_bench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

__device__ __forceinline__ __amdgpu_buffer_rsrc_t mk(const void* p) { return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, 0x7fffffff, 0x00020000); }

struct Args {
  unsigned* place;     // [0..7] per-XCD arrival count, [8 + 8*32...] table of block ids per XCD, [600] placement barrier, [601] stop flag
  float* tile[2];      // tile[g]: written by group g, NG*SLICE bytes
  unsigned* ctr;       // ctr[g*64]: arrivals of group g
  long long* out;
  int ng, slice_f4, iters, same, proto, busy;
};

template <int AUX>
__device__ __forceinline__ float4 ld16(__amdgpu_buffer_rsrc_t r, int byte_off) {
  const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(r, byte_off, 0, AUX);
  return make_float4(__uint_as_float(v.x), __uint_as_float(v.y), __uint_as_float(v.z), __uint_as_float(v.w));
}
template <int AUX>
__device__ __forceinline__ void st16(__amdgpu_buffer_rsrc_t r, int byte_off, float4 f) {
  u32x4 v; v.x = __float_as_uint(f.x); v.y = __float_as_uint(f.y); v.z = __float_as_uint(f.z); v.w = __float_as_uint(f.w);
  __builtin_amdgcn_raw_buffer_store_b128(v, r, byte_off, 0, AUX);
}
template <int AUX>
__device__ __forceinline__ unsigned ld4(__amdgpu_buffer_rsrc_t r, int byte_off) { return __builtin_amdgcn_raw_buffer_load_b32(r, byte_off, 0, AUX); }

template <int PROTO>   // 0: sc1 everywhere, 1: XCD-local (sc0 loads, plain stores)
__device__ void pingpong(const Args& a, int g, int idx) {
  constexpr int LD = PROTO == 0 ? 16 : 1, ST = PROTO == 0 ? 16 : 0;
  __shared__ float red[4];
  const int tid = threadIdx.x;
  const __amdgpu_buffer_rsrc_t rmine = mk(a.tile[g]), rother = mk(a.tile[g ^ 1]), rc = mk(a.ctr);
  const int tile_f4 = a.ng * a.slice_f4;
  float acc = (float)idx;
  int stale = 0;
  long long t0 = 0;
  for (int it = 0; it < a.iters; ++it) {
    if (it == 2 && g == 0 && idx == 0 && tid == 0) t0 = wall_clock64();
    for (int half = 0; half < 2; ++half) {
      if (half == g) {      // send: my slice, then arrive
        for (int i = tid; i < a.slice_f4; i += 256) st16<ST>(rmine, (idx * a.slice_f4 + i) * 16, make_float4(acc, acc + 1.f, (float)it, (float)i));
        __builtin_amdgcn_s_waitcnt(0);          // vmcnt(0) among others: the stores have been acknowledged
        __syncthreads();
        if (tid == 0) {
          if (PROTO == 0) __hip_atomic_fetch_add(a.ctr + g * 64, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          else __hip_atomic_fetch_add(a.ctr + g * 64, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
      } else {              // receive: wait for the other group's arrivals of this iteration, read its whole tile
        if (tid == 0) {
          const unsigned target = (unsigned)(a.ng * (it + 1));
          unsigned spins = 0;       // bounded: a missing member must not hang the box
          while (ld4<LD>(rc, (g ^ 1) * 64 * 4) < target && ++spins < (1u << 18)) { __asm__ volatile("" ::: "memory"); }
          if (spins >= (1u << 18)) a.out[4] = 1;
        }
        __syncthreads();
        float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
        for (int i0 = tid; i0 < tile_f4; i0 += 256 * 8) {     // 8 loads in flight per lane
          float4 v[8];
#pragma unroll
          for (int j = 0; j < 8; ++j) v[j] = ld16<LD>(rother, min(i0 + 256 * j, tile_f4 - 1) * 16);
#pragma unroll
          for (int j = 0; j < 8; ++j) if (i0 + 256 * j < tile_f4) {
            s.x += v[j].x; s.y += v[j].y; s.z += v[j].z; s.w += v[j].w;
            if (v[j].z != (float)it) stale++;         // a line of an earlier iteration (L1 / L2 served an old copy)
          }
        }
        float w = s.x + s.y * 1e-3f + s.z * 1e-6f + s.w * 1e-9f;
        for (int o = 32; o > 0; o >>= 1) w += __shfl_xor(w, o);
        if ((tid & 63) == 0) red[tid >> 6] = w;
        __syncthreads();
        acc = (red[0] + red[1] + red[2] + red[3]) * 1e-6f + (float)idx;
        __syncthreads();
      }
    }
  }
  if (g == 0 && idx == 0 && tid == 0) { a.out[0] = wall_clock64() - t0; a.out[1] = (long long)acc; }
  if (stale) atomicAdd((unsigned long long*)(a.out + 3), (unsigned long long)stale);
}

__global__ __launch_bounds__(256, 1) void k(Args a) {
  extern __shared__ float pad[];      // sized so that one workgroup fits per CU
  __shared__ int s_role[3];
  const int tid = threadIdx.x;
  if (tid == 0) {
    unsigned xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    xcc &= 7u;
    const unsigned slot = atomicAdd(a.place + xcc, 1u);
    a.place[8 + xcc * 64 + slot] = blockIdx.x;
    __threadfence();
    atomicAdd(a.place + 600, 1u);
    for (unsigned spins = 0; __hip_atomic_load(a.place + 600, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < gridDim.x && spins < (1u << 18); ++spins) {}
    __threadfence();
    // membership: same  -> group g = slots [g*ng, (g+1)*ng) of the fullest-indexed XCD 0 ; cross -> member m of group g = slot (2*(m/8)+g) of XCD m%8
    int g = -1, idx = -1;
    if (a.same) {
      if (xcc == 0 && (int)slot < 2 * a.ng) { g = slot / a.ng; idx = slot % a.ng; }
    } else {
      const int per = (a.ng + 7) / 8;           // members of one group per XCD
      if ((int)slot < 2 * per) {
        const int gg = slot / per, m = (slot % per) * 8 + xcc;
        if (m < a.ng) { g = gg; idx = m; }
      }
    }
    s_role[0] = g; s_role[1] = idx; s_role[2] = (int)xcc;
    if (blockIdx.x == 0) a.out[2] = a.place[0] | ((long long)a.place[1] << 8) | ((long long)a.place[7] << 16);
  }
  __syncthreads();
  const int g = s_role[0], idx = s_role[1];
  if (g < 0) {
    if (a.busy && tid == 0) {           // a waiting role: sc1 polls until the players are done
      for (unsigned spins = 0; __hip_atomic_load(a.place + 601, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0 && spins < (1u << 20); ++spins) {}
    }
    return;
  }
  if (a.proto == 0) pingpong<0>(a, g, idx); else pingpong<1>(a, g, idx);
  if (g == 0 && idx == 0 && tid == 0) __hip_atomic_store(a.place + 601, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

int main(int argc, char** argv) {
  const bool with_l2 = argc > 1 && atoi(argv[1]) != 0;     // the sc0 variant: measured -- sc0 polls are served by the CU's L1 and never see the
                                                           // counter move (every wait ran into its bound), so it is not a protocol
  const int G = 256;
  setvbuf(stdout, nullptr, _IONBF, 0);
  unsigned* place; float* t0; float* t1; unsigned* ctr; long long* out;
  CK(hipMalloc(&place, 4096)); CK(hipMalloc(&t0, 1 << 22)); CK(hipMalloc(&t1, 1 << 22)); CK(hipMalloc(&ctr, 1024)); CK(hipMalloc(&out, 64));
  CK(hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, 100 * 1024));
  printf("%-6s %-5s %-5s %4s %8s  %s\n", "place", "proto", "load", "NG", "tile", "us per one-way hand-off (3 runs)");
  const int ngs[] = {1, 16, 16, 16};
  const int slices[] = {256, 64, 256, 512};       // float4 per slice: 4 KB, 1 KB, 4 KB, 8 KB  -> tiles 4 KB, 16 KB, 64 KB, 128 KB
  for (int c = 0; c < 4; ++c)
    for (int busy = 0; busy < 2; ++busy)
      for (int v = 0; v < (with_l2 ? 3 : 2); ++v) {          // v: 0 cross/sc1, 1 same/sc1, 2 same/l2
        Args a;
        a.place = place; a.tile[0] = t0; a.tile[1] = t1; a.ctr = ctr; a.out = out;
        a.ng = ngs[c]; a.slice_f4 = slices[c]; a.iters = 202; a.same = v > 0; a.proto = v == 2; a.busy = busy;
        printf("%-6s %-5s %-5s %4d %6dKB ", a.same ? "same" : "cross", a.proto ? "l2" : "sc1", busy ? "busy" : "idle", a.ng, a.ng * a.slice_f4 * 16 / 1024);
        for (int rep = 0; rep < 3; ++rep) {
          CK(hipMemset(place, 0, 4096)); CK(hipMemset(ctr, 0, 1024)); CK(hipMemset(out, 0, 64));
          hipLaunchKernelGGL(k, dim3(G), dim3(256), 100 * 1024, 0, a);
          CK(hipDeviceSynchronize());
          long long h[5];
          CK(hipMemcpy(h, out, sizeof(h), hipMemcpyDeviceToHost));
          printf(" %6.2f", (double)h[0] / 100.0 / (2.0 * (a.iters - 2)));
          if (h[3]) printf(" [%lld STALE float4]", h[3]);
          if (h[4]) printf(" [TIMED OUT]");
          if (rep == 2) printf("   (XCD0/1/7 hold %lld/%lld/%lld blocks)", h[2] & 255, (h[2] >> 8) & 255, (h[2] >> 16) & 255);
        }
        printf("\n");
      }
  return 0;
}
