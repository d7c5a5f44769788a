// Attention scan (seq2seq.py:336-357): the HBM/L2-streaming part of a decoder step.
//
//   forward : s[b,t] = enc[b,t,:].q[b,:] ; alpha = softmax_t(s) (unmasked, quirk Q2) ; cv = sum_t alpha enc[b,t,:]
//             ONE pass over enc_states with an online softmax; the time axis of every batch row is split
//             over `nsplit` workgroups so that B*nsplit >= #CUs, partial (max, sum, cv) are merged by a
//             tiny second kernel that also normalises alpha.
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
                                                        int Tp, float* __restrict__ part, int nsplit, int chunk) {
  extern __shared__ __attribute__((aligned(16))) float sm[];   // [4][H] + 8
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
      if (lane == 0) scores[(long)b * Tp + t] = s;
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
      if (lane == 0) scores[(long)b * Tp + t + 4] = s;
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
    prow[PART_PAD + i] = sacc[i] * w[0] + sacc[H + i] * w[1] + sacc[2 * H + i] * w[2] + sacc[3 * H + i] * w[3];
  if (threadIdx.x == 0) {
    prow[0] = M;
    prow[1] = sml[4] * w[0] + sml[5] * w[1] + sml[6] * w[2] + sml[7] * w[3];
  }
}

__global__ __launch_bounds__(256) void attn_fwd_combine(int B, int T, int H, float* __restrict__ alpha, int Tp,
                                                        const float* __restrict__ part, int nsplit, float* __restrict__ cv,
                                                        long ldcv, float* __restrict__ cv2, long ldcv2) {
  const int b = blockIdx.x;
  const float* pb = part + (long)b * nsplit * (H + PART_PAD);
  float M = -INFINITY;
  for (int s = 0; s < nsplit; ++s) M = fmaxf(M, pb[(long)s * (H + PART_PAD)]);
  float L = 0.f;
  for (int s = 0; s < nsplit; ++s) L += pb[(long)s * (H + PART_PAD) + 1] * expf(pb[(long)s * (H + PART_PAD)] - M);
  const float inv = 1.f / L;
  for (int i = threadIdx.x; i < H; i += 256) {
    float v = 0.f;
    for (int s = 0; s < nsplit; ++s) v += pb[(long)s * (H + PART_PAD) + PART_PAD + i] * expf(pb[(long)s * (H + PART_PAD)] - M);
    v *= inv;
    cv[(long)b * ldcv + i] = v;
    if (cv2) cv2[(long)b * ldcv2 + i] = v;
  }
  for (int t = threadIdx.x; t < Tp; t += 256) {
    float* a = alpha + (long)b * Tp + t;
    *a = t < T ? expf(*a - M) * inv : 0.f;
  }
}

template <int NCH>
__global__ __launch_bounds__(256) void attn_bwd_partial(int B, int T, int H, const float* __restrict__ enc,
                                                        const float* __restrict__ alpha, int Tp, const float* __restrict__ cv,
                                                        long ldcv, const float* __restrict__ dcv, long ld_dcv,
                                                        float* __restrict__ ds, float* __restrict__ part, int nsplit, int chunk) {
  extern __shared__ __attribute__((aligned(16))) float sm[];   // [4][H]
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
  for (int i = threadIdx.x; i < H; i += 256) prow[i] = sm[i] + sm[H + i] + sm[2 * H + i] + sm[3 * H + i];
}

__global__ __launch_bounds__(256) void attn_bwd_combine(int B, int H, const float* __restrict__ part, int nsplit,
                                                        float* __restrict__ dq) {
  const int b = blockIdx.x;
  for (int i = threadIdx.x; i < H; i += 256) {
    float v = 0.f;
    for (int s = 0; s < nsplit; ++s) v += part[((long)b * nsplit + s) * H + i];
    dq[(long)b * H + i] = v;
  }
}

inline void split_for(int B, int T, int& nsplit, int& chunk) {
  int want = cdiv(512, B);                 // ~2 workgroups per CU
  if (want < 1) want = 1;
  chunk = cdiv(T, want);
  if (chunk < 8) chunk = 8;                // at least 2 rows per wave
  if (chunk > T) chunk = T;
  nsplit = cdiv(T, chunk);
}

}  // namespace

size_t attn_ws_bytes(int B, int T, int H) {
  int nsplit, chunk;
  split_for(B, T, nsplit, chunk);
  return align_up((size_t)B * nsplit * (H + PART_PAD) * sizeof(float), 256);
}

int attn_fwd_launch(int B, int T, int H, const float* enc, const float* q, long ldq, float* alpha, float* cv, long ldcv,
                    float* cv2, long ldcv2, void* ws, hipStream_t s) {
  ASTK_CHECK(B > 0 && T > 0 && H > 0 && (H % 4) == 0 && H <= 2048, "attn: need H %% 4 == 0 and H <= 2048 (H=%d)", H);
  ASTK_CHECK(enc && q && alpha && cv && ws && aligned16(enc) && aligned16(q) && (ldq % 4) == 0, "attn_fwd: bad pointers");
  int nsplit, chunk;
  split_for(B, T, nsplit, chunk);
  const int Tp = (T + 3) / 4 * 4;
  float* part = (float*)ws;
  const size_t shm = (size_t)(4 * H + 8) * sizeof(float);
  dim3 grid(B * nsplit), blk(256);
  const int nch = cdiv(H, 256);
  {
  ProfScope prof(PROF_ATTN_FWD, s);
  if (nch <= 1) hipLaunchKernelGGL((attn_fwd_partial<1>), grid, blk, shm, s, B, T, H, enc, q, ldq, alpha, Tp, part, nsplit, chunk);
  else if (nch <= 2) hipLaunchKernelGGL((attn_fwd_partial<2>), grid, blk, shm, s, B, T, H, enc, q, ldq, alpha, Tp, part, nsplit, chunk);
  else if (nch <= 4) hipLaunchKernelGGL((attn_fwd_partial<4>), grid, blk, shm, s, B, T, H, enc, q, ldq, alpha, Tp, part, nsplit, chunk);
  else hipLaunchKernelGGL((attn_fwd_partial<8>), grid, blk, shm, s, B, T, H, enc, q, ldq, alpha, Tp, part, nsplit, chunk);
  }
  ASTK_LAUNCH_CHECK();
  hipLaunchKernelGGL(attn_fwd_combine, dim3(B), blk, 0, s, B, T, H, alpha, Tp, part, nsplit, cv, ldcv, cv2, ldcv2);
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
  const int nch = cdiv(H, 256);
  {
  ProfScope prof(PROF_ATTN_BWD, s);
  if (nch <= 1) hipLaunchKernelGGL((attn_bwd_partial<1>), grid, blk, shm, s, B, T, H, enc, alpha, Tp, cv, ldcv, d_cv, ld_dcv, ds, part, nsplit, chunk);
  else if (nch <= 2) hipLaunchKernelGGL((attn_bwd_partial<2>), grid, blk, shm, s, B, T, H, enc, alpha, Tp, cv, ldcv, d_cv, ld_dcv, ds, part, nsplit, chunk);
  else if (nch <= 4) hipLaunchKernelGGL((attn_bwd_partial<4>), grid, blk, shm, s, B, T, H, enc, alpha, Tp, cv, ldcv, d_cv, ld_dcv, ds, part, nsplit, chunk);
  else hipLaunchKernelGGL((attn_bwd_partial<8>), grid, blk, shm, s, B, T, H, enc, alpha, Tp, cv, ldcv, d_cv, ld_dcv, ds, part, nsplit, chunk);
  }
  ASTK_LAUNCH_CHECK();
  hipLaunchKernelGGL(attn_bwd_combine, dim3(B), blk, 0, s, B, H, part, nsplit, dq);
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
  return attn_fwd_launch(B, T, H, enc, q, H, alpha, cv, H, nullptr, 0, ws, (hipStream_t)stream);
}

int astk_attn_step_bwd(int B, int T, int H, const float* enc, const float* alpha, const float* cv, const float* d_cv,
                       float* ds, float* dq, void* ws, size_t ws_bytes, void* stream) {
  ASTK_CHECK(ws_bytes >= attn_ws_bytes(B, T, H), "attn_step_bwd: workspace too small");
  return attn_bwd_launch(B, T, H, enc, alpha, cv, H, d_cv, H, ds, dq, ws, (hipStream_t)stream);
}

}  // extern "C"
