// Attention scan (seq2seq.py:336-357): the HBM/L2-streaming part of a decoder step.
//
//   forward : s[b,t] = enc[b,t,:].q[b,:] ; alpha = softmax_t(s) (unmasked, quirk Q2) ; cv = sum_t alpha enc[b,t,:]
//             ONE pass over enc_states with an online softmax; the time axis of every batch row is split
//             over `nsplit` workgroups so that B*nsplit >= #CUs; the workgroup that finishes a batch row last
//             merges the partial (max, sum, cv) and normalises alpha inside the same launch.
//   backward: ds[b,t] = alpha (enc.d_cv - cv.d_cv) (the softmax Jacobian's sum_t alpha dalpha equals cv.d_cv)
//             dq = sum_t ds enc[b,t,:]  -- again ONE pass; d_enc is produced once per train step by a deferred
//             batched GEMM over the saved (alpha, ds) (decoder.hip), not here.
// Algorithmic bytes per call = B*T*H*4.  Each wave reads whole 4H-byte rows with 16-byte
// lane loads (1 KiB per instruction); workgroup b + B*split keeps the same batch row on the same XCD
// (blockIdx % 8) across decoder steps, so enc_states (13 MB at cfg 2) is served from the XCD L2s after step 0.
#include "common.h"

namespace astk {

namespace {

constexpr int PART_PAD = 4;   // part row: [m, l, -, -, acc[H]]
constexpr int MAX_SPLIT = 128;

// Cross-workgroup combine inside the launch ("last arriver reduces", cdna_hip_programming.md section 5 item 2, sc1 form):
// partials are stored write-through (sc1), every storing wave drains, the workgroup barriers, one lane takes a ticket on the
// batch row's counter; the workgroup that draws the last ticket re-reads all partials with sc1 loads and finishes the row.
// The counter is reset by the last arriver, so it is zero again for the next launch on the stream.
__device__ __forceinline__ void st_sc1(float* p, float v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ float ld_sc1(const float* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ bool last_arriver(unsigned* cnt, int nsplit, int* s_flag) {
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (threadIdx.x == 0) {
    const unsigned tk = __hip_atomic_fetch_add(cnt, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const int last = tk == (unsigned)(nsplit - 1);
    if (last) __hip_atomic_store(cnt, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    *s_flag = last;
  }
  __syncthreads();
  return *s_flag != 0;
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  return v;
}

template <int NCH>
__device__ __forceinline__ void load_row(const float* p, int H, int lane, float4 (&e)[NCH]) {
#pragma unroll
  for (int c = 0; c < NCH; ++c) {
    const int idx = 256 * c + 4 * lane;
    e[c] = idx < H ? *reinterpret_cast<const float4*>(p + idx) : make_float4(0.f, 0.f, 0.f, 0.f);
  }
}
template <int NCH>
__device__ __forceinline__ float dot_row(const float4 (&a)[NCH], const float4 (&b)[NCH]) {
  float s = 0.f;
#pragma unroll
  for (int c = 0; c < NCH; ++c) s += a[c].x * b[c].x + a[c].y * b[c].y + a[c].z * b[c].z + a[c].w * b[c].w;
  return s;
}

template <int NCH>
__global__ __launch_bounds__(256) void attn_fwd_partial(int B, int T, int H, const float* __restrict__ enc,
                                                        const float* __restrict__ q, long ldq, float* __restrict__ scores,
                                                        int Tp, float* __restrict__ part, int nsplit, int chunk,
                                                        unsigned* __restrict__ cnt, float* __restrict__ cv, long ldcv,
                                                        float* __restrict__ cv2, long ldcv2) {
  extern __shared__ __attribute__((aligned(16))) float sm[];   // [4][H] + 8 + 2*MAX_SPLIT
  __shared__ int s_last;
  const int b = blockIdx.x % B, sp = blockIdx.x / B;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int t0 = sp * chunk, t1 = min(T, t0 + chunk);
  float4 qv[NCH];
  load_row<NCH>(q + (long)b * ldq, H, lane, qv);
  float m = -INFINITY, l = 0.f;
  float4 acc[NCH];
#pragma unroll
  for (int c = 0; c < NCH; ++c) acc[c] = make_float4(0.f, 0.f, 0.f, 0.f);
  const float* base = enc + (long)b * T * H;
  for (int t = t0 + wave; t < t1; t += 8) {
    float4 e0[NCH], e1[NCH];
    const bool two = t + 4 < t1;
    load_row<NCH>(base + (long)t * H, H, lane, e0);
    if (two) load_row<NCH>(base + (long)(t + 4) * H, H, lane, e1);
    {
      const float s = wave_sum(dot_row<NCH>(e0, qv));
      if (lane == 0) st_sc1(&scores[(long)b * Tp + t], s);
      const float mn = fmaxf(m, s);
      const float sc = expf(m - mn), p = expf(s - mn);
      l = l * sc + p;
#pragma unroll
      for (int c = 0; c < NCH; ++c) {
        acc[c].x = acc[c].x * sc + p * e0[c].x; acc[c].y = acc[c].y * sc + p * e0[c].y;
        acc[c].z = acc[c].z * sc + p * e0[c].z; acc[c].w = acc[c].w * sc + p * e0[c].w;
      }
      m = mn;
    }
    if (two) {
      const float s = wave_sum(dot_row<NCH>(e1, qv));
      if (lane == 0) st_sc1(&scores[(long)b * Tp + t + 4], s);
      const float mn = fmaxf(m, s);
      const float sc = expf(m - mn), p = expf(s - mn);
      l = l * sc + p;
#pragma unroll
      for (int c = 0; c < NCH; ++c) {
        acc[c].x = acc[c].x * sc + p * e1[c].x; acc[c].y = acc[c].y * sc + p * e1[c].y;
        acc[c].z = acc[c].z * sc + p * e1[c].z; acc[c].w = acc[c].w * sc + p * e1[c].w;
      }
      m = mn;
    }
  }
  // merge the 4 waves
  float* sacc = sm;            // [4][H]
  float* sml = sm + 4 * H;     // [4] m, [4] l
#pragma unroll
  for (int c = 0; c < NCH; ++c) {
    const int idx = 256 * c + 4 * lane;
    if (idx < H) *reinterpret_cast<float4*>(&sacc[wave * H + idx]) = acc[c];
  }
  if (lane == 0) { sml[wave] = m; sml[4 + wave] = l; }
  __syncthreads();
  const float M = fmaxf(fmaxf(sml[0], sml[1]), fmaxf(sml[2], sml[3]));
  float w[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) w[k] = sml[k] == -INFINITY ? 0.f : expf(sml[k] - M);
  float* prow = part + ((long)b * nsplit + sp) * (H + PART_PAD);
  for (int i = threadIdx.x; i < H; i += 256)
    st_sc1(&prow[PART_PAD + i], sacc[i] * w[0] + sacc[H + i] * w[1] + sacc[2 * H + i] * w[2] + sacc[3 * H + i] * w[3]);
  if (threadIdx.x == 0) {
    st_sc1(&prow[0], M);
    st_sc1(&prow[1], sml[4] * w[0] + sml[5] * w[1] + sml[6] * w[2] + sml[7] * w[3]);
  }
  if (!last_arriver(cnt + b, nsplit, &s_last)) return;
  // ---- this workgroup finishes batch row b: merge the nsplit partials, normalise alpha
  float* wgt = sm + 4 * H + 8;            // [MAX_SPLIT] weights exp(m_s - M)
  float* lsum = wgt + MAX_SPLIT;          // [MAX_SPLIT]
  const float* pb = part + (long)b * nsplit * (H + PART_PAD);
  if (threadIdx.x < nsplit) {
    wgt[threadIdx.x] = ld_sc1(&pb[(long)threadIdx.x * (H + PART_PAD)]);
    lsum[threadIdx.x] = ld_sc1(&pb[(long)threadIdx.x * (H + PART_PAD) + 1]);
  }
  __syncthreads();
  float Mx = -INFINITY;
  for (int s2 = 0; s2 < nsplit; ++s2) Mx = fmaxf(Mx, wgt[s2]);
  __syncthreads();
  if (threadIdx.x < nsplit) wgt[threadIdx.x] = expf(wgt[threadIdx.x] - Mx);
  __syncthreads();
  float L = 0.f;
  for (int s2 = 0; s2 < nsplit; ++s2) L += lsum[s2] * wgt[s2];
  const float inv = 1.f / L;
  for (int i = threadIdx.x; i < H; i += 256) {
    float v = 0.f;
    for (int s2 = 0; s2 < nsplit; ++s2) v += ld_sc1(&pb[(long)s2 * (H + PART_PAD) + PART_PAD + i]) * wgt[s2];
    v *= inv;
    cv[(long)b * ldcv + i] = v;
    if (cv2) cv2[(long)b * ldcv2 + i] = v;
  }
  for (int t = threadIdx.x; t < Tp; t += 256) {
    float* ap = scores + (long)b * Tp + t;
    const float sc = t < T ? ld_sc1(ap) : 0.f;
    *ap = t < T ? expf(sc - Mx) * inv : 0.f;
  }
}

template <int NCH>
__global__ __launch_bounds__(256) void attn_bwd_partial(int B, int T, int H, const float* __restrict__ enc,
                                                        const float* __restrict__ alpha, int Tp, const float* __restrict__ cv,
                                                        long ldcv, const float* __restrict__ dcv, long ld_dcv,
                                                        float* __restrict__ ds, float* __restrict__ part, int nsplit, int chunk,
                                                        unsigned* __restrict__ cnt, float* __restrict__ dq) {
  extern __shared__ __attribute__((aligned(16))) float sm[];   // [4][H]
  __shared__ int s_last;
  const int b = blockIdx.x % B, sp = blockIdx.x / B;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int t0 = sp * chunk, t1 = min(T, t0 + chunk);
  float4 dv[NCH], cvv[NCH];
  load_row<NCH>(dcv + (long)b * ld_dcv, H, lane, dv);
  load_row<NCH>(cv + (long)b * ldcv, H, lane, cvv);
  const float cd = wave_sum(dot_row<NCH>(cvv, dv));
  float4 acc[NCH];
#pragma unroll
  for (int c = 0; c < NCH; ++c) acc[c] = make_float4(0.f, 0.f, 0.f, 0.f);
  const float* base = enc + (long)b * T * H;
  for (int t = t0 + wave; t < t1; t += 8) {
    float4 e0[NCH], e1[NCH];
    const bool two = t + 4 < t1;
    load_row<NCH>(base + (long)t * H, H, lane, e0);
    if (two) load_row<NCH>(base + (long)(t + 4) * H, H, lane, e1);
    {
      const float da = wave_sum(dot_row<NCH>(e0, dv));
      const float d = alpha[(long)b * Tp + t] * (da - cd);
      if (lane == 0) ds[(long)b * Tp + t] = d;
#pragma unroll
      for (int c = 0; c < NCH; ++c) {
        acc[c].x += d * e0[c].x; acc[c].y += d * e0[c].y; acc[c].z += d * e0[c].z; acc[c].w += d * e0[c].w;
      }
    }
    if (two) {
      const float da = wave_sum(dot_row<NCH>(e1, dv));
      const float d = alpha[(long)b * Tp + t + 4] * (da - cd);
      if (lane == 0) ds[(long)b * Tp + t + 4] = d;
#pragma unroll
      for (int c = 0; c < NCH; ++c) {
        acc[c].x += d * e1[c].x; acc[c].y += d * e1[c].y; acc[c].z += d * e1[c].z; acc[c].w += d * e1[c].w;
      }
    }
  }
#pragma unroll
  for (int c = 0; c < NCH; ++c) {
    const int idx = 256 * c + 4 * lane;
    if (idx < H) *reinterpret_cast<float4*>(&sm[wave * H + idx]) = acc[c];
  }
  __syncthreads();
  float* prow = part + ((long)b * nsplit + sp) * H;
  for (int i = threadIdx.x; i < H; i += 256) st_sc1(&prow[i], sm[i] + sm[H + i] + sm[2 * H + i] + sm[3 * H + i]);
  if (!last_arriver(cnt + b, nsplit, &s_last)) return;
  for (int i = threadIdx.x; i < H; i += 256) {
    float v = 0.f;
    for (int s2 = 0; s2 < nsplit; ++s2) v += ld_sc1(&part[((long)b * nsplit + s2) * H + i]);
    dq[(long)b * H + i] = v;
  }
}

inline void split_for(int B, int T, int& nsplit, int& chunk) {
  int want = cdiv(512, B);                 // ~2 workgroups per CU
  if (want < 1) want = 1;
  chunk = cdiv(T, want);
  if (chunk < 8) chunk = 8;                // at least 2 rows per wave
  if (chunk > T) chunk = T;
  if (cdiv(T, chunk) > MAX_SPLIT) chunk = cdiv(T, MAX_SPLIT);
  nsplit = cdiv(T, chunk);
}

}  // namespace

size_t attn_ws_bytes(int B, int T, int H) {
  int nsplit, chunk;
  split_for(B, T, nsplit, chunk);
  return align_up((size_t)B * nsplit * (H + PART_PAD) * sizeof(float), 256) + align_up((size_t)B * sizeof(unsigned), 256);
}
// the ticket counters live behind the partials; they must be zero before the first launch (attn_ws_init) and stay zero
unsigned* attn_counters(void* ws, int B, int T, int H) {
  int nsplit, chunk;
  split_for(B, T, nsplit, chunk);
  return (unsigned*)((char*)ws + align_up((size_t)B * nsplit * (H + PART_PAD) * sizeof(float), 256));
}
int attn_ws_init(void* ws, int B, int T, int H, hipStream_t s) {
  ASTK_HIP(hipMemsetAsync(attn_counters(ws, B, T, H), 0, (size_t)B * sizeof(unsigned), s));
  return 0;
}

int attn_fwd_launch(int B, int T, int H, const float* enc, const float* q, long ldq, float* alpha, float* cv, long ldcv,
                    float* cv2, long ldcv2, void* ws, hipStream_t s) {
  ASTK_CHECK(B > 0 && T > 0 && H > 0 && (H % 4) == 0 && H <= 2048, "attn: need H %% 4 == 0 and H <= 2048 (H=%d)", H);
  ASTK_CHECK(enc && q && alpha && cv && ws && aligned16(enc) && aligned16(q) && (ldq % 4) == 0, "attn_fwd: bad pointers");
  int nsplit, chunk;
  split_for(B, T, nsplit, chunk);
  const int Tp = (T + 3) / 4 * 4;
  float* part = (float*)ws;
  const size_t shm = (size_t)(4 * H + 8 + 2 * MAX_SPLIT) * sizeof(float);
  dim3 grid(B * nsplit), blk(256);
  unsigned* cnt = attn_counters(ws, B, T, H);
  const int nch = cdiv(H, 256);
  {
  ProfScope prof(PROF_ATTN_FWD, s);
  if (nch <= 1) hipLaunchKernelGGL((attn_fwd_partial<1>), grid, blk, shm, s, B, T, H, enc, q, ldq, alpha, Tp, part, nsplit, chunk, cnt, cv, ldcv, cv2, ldcv2);
  else if (nch <= 2) hipLaunchKernelGGL((attn_fwd_partial<2>), grid, blk, shm, s, B, T, H, enc, q, ldq, alpha, Tp, part, nsplit, chunk, cnt, cv, ldcv, cv2, ldcv2);
  else if (nch <= 4) hipLaunchKernelGGL((attn_fwd_partial<4>), grid, blk, shm, s, B, T, H, enc, q, ldq, alpha, Tp, part, nsplit, chunk, cnt, cv, ldcv, cv2, ldcv2);
  else hipLaunchKernelGGL((attn_fwd_partial<8>), grid, blk, shm, s, B, T, H, enc, q, ldq, alpha, Tp, part, nsplit, chunk, cnt, cv, ldcv, cv2, ldcv2);
  }
  ASTK_LAUNCH_CHECK();
  return 0;
}

int attn_bwd_launch(int B, int T, int H, const float* enc, const float* alpha, const float* cv, long ldcv, const float* d_cv,
                    long ld_dcv, float* ds, float* dq, void* ws, hipStream_t s) {
  ASTK_CHECK(B > 0 && T > 0 && H > 0 && (H % 4) == 0 && H <= 2048, "attn: need H %% 4 == 0 and H <= 2048 (H=%d)", H);
  ASTK_CHECK(enc && alpha && cv && d_cv && ds && dq && ws && (ldcv % 4) == 0 && (ld_dcv % 4) == 0 && aligned16(cv) &&
                 aligned16(d_cv), "attn_bwd: bad pointers");
  int nsplit, chunk;
  split_for(B, T, nsplit, chunk);
  const int Tp = (T + 3) / 4 * 4;
  float* part = (float*)ws;
  const size_t shm = (size_t)(4 * H) * sizeof(float);
  dim3 grid(B * nsplit), blk(256);
  unsigned* cnt = attn_counters(ws, B, T, H);
  const int nch = cdiv(H, 256);
  {
  ProfScope prof(PROF_ATTN_BWD, s);
  if (nch <= 1) hipLaunchKernelGGL((attn_bwd_partial<1>), grid, blk, shm, s, B, T, H, enc, alpha, Tp, cv, ldcv, d_cv, ld_dcv, ds, part, nsplit, chunk, cnt, dq);
  else if (nch <= 2) hipLaunchKernelGGL((attn_bwd_partial<2>), grid, blk, shm, s, B, T, H, enc, alpha, Tp, cv, ldcv, d_cv, ld_dcv, ds, part, nsplit, chunk, cnt, dq);
  else if (nch <= 4) hipLaunchKernelGGL((attn_bwd_partial<4>), grid, blk, shm, s, B, T, H, enc, alpha, Tp, cv, ldcv, d_cv, ld_dcv, ds, part, nsplit, chunk, cnt, dq);
  else hipLaunchKernelGGL((attn_bwd_partial<8>), grid, blk, shm, s, B, T, H, enc, alpha, Tp, cv, ldcv, d_cv, ld_dcv, ds, part, nsplit, chunk, cnt, dq);
  }
  ASTK_LAUNCH_CHECK();
  return 0;
}

}  // namespace astk

using namespace astk;

extern "C" {

size_t astk_attn_workspace_bytes(int B, int T, int H) { return attn_ws_bytes(B, T, H); }

// alpha is (B, Tp) with Tp = round_up(T, 4) (the layout the decoder keeps it in); cv (B, H).
int astk_attn_step_fwd(int B, int T, int H, const float* enc, const float* q, float* alpha, float* cv, void* ws,
                       size_t ws_bytes, void* stream) {
  ASTK_CHECK(ws_bytes >= attn_ws_bytes(B, T, H), "attn_step_fwd: workspace too small");
  ASTK_TRY(attn_ws_init(ws, B, T, H, (hipStream_t)stream));
  return attn_fwd_launch(B, T, H, enc, q, H, alpha, cv, H, nullptr, 0, ws, (hipStream_t)stream);
}

int astk_attn_step_bwd(int B, int T, int H, const float* enc, const float* alpha, const float* cv, const float* d_cv,
                       float* ds, float* dq, void* ws, size_t ws_bytes, void* stream) {
  ASTK_CHECK(ws_bytes >= attn_ws_bytes(B, T, H), "attn_step_bwd: workspace too small");
  ASTK_TRY(attn_ws_init(ws, B, T, H, (hipStream_t)stream));
  return attn_bwd_launch(B, T, H, enc, alpha, cv, H, d_cv, H, ds, dq, ws, (hipStream_t)stream);
}

}  // extern "C"
