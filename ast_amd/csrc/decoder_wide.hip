// Persistent forward loop for the WIDE decoder of BASELINE configs[4] (H = A = 1024, E = 128, one layer, one attention head, input
// feeding; seq2seq.py:361-397): all decoder steps in ONE launch.
//
// decoder_persist.hip keeps every decoder weight in registers; at this width they are 77 MB (cell 35.6, context 8.4, logits 32.8), 300 of
// the 512 registers of every lane on the chip.  This kernel drops the logits from the chain (decoder.hip scores all steps with one product
// behind the loop; only the steps whose argmax is fed back compute logits here, from weights streamed through L2) and shares the remaining
// weight slices between the two 16-row batch tiles:
//   CELL  all 256 workgroups: 4 hidden units (the 16 gate rows 16w..16w+15 of Chainer's interleaved layout), K = E + A + H = 2176 split over
//         the 4 waves: 34 float4 of weights per lane; embedding rows gathered straight from the table
//   Q     workgroups 64..127:  q = Wa h + ba, 16 columns each (16 float4 per lane)
//   ATT   workgroup (b, chunk): its slice of enc_states stays in LDS for all steps (32 x 8 slices of 25 rows x 4 KB at T'' = 200)
//   CMB   workgroups 192..192+B-1: merges a batch row's partial softmax / context sums, normalises alpha
//   CTX   workgroups 128..191: ht = tanh(Wc [cv; h] + bc), 16 columns each (32 float4 per lane)
//   LOG / ARG (only when the next step is not teacher-forced): every workgroup streams its 16-column logits tiles, one workgroup per batch
//         row reduces the per-tile maxima to the class that is fed back
// Chain per step: CELL -> Q -> ATT -> CMB -> CTX -> next CELL; hand-offs as in decoder_persist.hip (write-through stores, drained, one arrival
// add on a sharded counter; consumers poll, then sc1 loads).  The h part of the next cell product needs only CELL, so it runs while CTX is in
// flight.  Saved state = decoder.hip's DecPlan, so its per-launch backward runs on it unchanged.  All spins are bounded (abort word).
#include "decoder_wide.h"

namespace astk {

namespace {

constexpr int WH = 1024, WA = 1024, WE = 128, WXI = WE + WA;
constexpr int WG_ = 256;              // workgroups (one per CU)
constexpr int CTRS = 64;              // counter stride in words (256 B)
constexpr int NSH = 32;               // shards of a phase counter
constexpr int Q0 = 64, CTX0 = 128, CMB0 = 192;   // first workgroup of the Q / CTX / CMB roles
constexpr int NQ = WH / 16, NCTX = WA / 16;
enum { C_CELL = 0, C_Q, C_CMB, C_CTX, C_LOG, C_ARG, C_N };
constexpr int PARTW = WH + 4;

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

struct WideArgs {
  int B, S, L, T, Tp, V, s0, s1, nsplit, chunk;
  const float *embed, *Wu, *bias, *Wl, *Wa, *ba, *Wc, *bc, *Wo, *bo;
  const float* enc;
  const int32_t *y, *use_truth;
  int32_t* PRED;
  float* LMAX;         // [B][ntile][2]: per (batch row, 16-column logits tile) maximum and its class id (fed-back steps only)
  int ntile;
  const float *emb_mask, *rnn_mask;
  int32_t* TOK;
  float *X0, *G, *C, *HR, *Q, *ALPHA, *CVH, *HT, *PART;
  unsigned* ctr;       // [C_N][NSH] lines, then [B] lines (per batch row: ATT -> CMB)
  AbortCtl ab;
};

__device__ __forceinline__ unsigned ld_flag(const unsigned* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void st_sc1(float* p, float v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ float ld_sc1(const float* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ int ldi_sc1(const int* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void sti_sc1(int* p, int v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const void* p) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, 0x7fffffff, 0x00020000);
}
// Handed-off activations are read by every workgroup of an XCD (the 32 rows x 1024 columns of h, ht, [cv; h]: 128-256 KB per workgroup
// and step).  ASTK_WIDE_SC1 = 1 (default): sc1 loads (every load goes to the memory side, as in decoder_persist.hip); 0: an agent-scope
// acquire behind every successful wait (invalidates the XCD's L2 copies of such lines) and ordinary loads, so that the 32 workgroups of an
// XCD share one fetch of each line -- measured: the ten L2 invalidations per step cost more than the shared fetches save (configs[4] shape:
// 18.7 ms per train step against 17.9 with sc1 loads; 18.7 on the per-launch loop).
#ifndef ASTK_WIDE_SC1
#define ASTK_WIDE_SC1 1
#endif
__device__ __forceinline__ float4 ldb128_sc1(__amdgpu_buffer_rsrc_t r, long float_off) {
  const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(r, (int)(float_off * 4), 0, ASTK_WIDE_SC1 ? 16 : 0);
  return make_float4(__uint_as_float(v.x), __uint_as_float(v.y), __uint_as_float(v.z), __uint_as_float(v.w));
}
__device__ __forceinline__ void acquire_handoff() {
#if !ASTK_WIDE_SC1
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
#endif
}
// lanes 0..NSH-1 of wave 0 poll the shards of a phase counter until `n_items` items have arrived `steps` times each
__device__ __forceinline__ bool wait_sh(const unsigned* base, int n_items, int steps, const AbortCtl& ab, int* s_flag) {
  if (threadIdx.x < 64) {
    const int lane = threadIdx.x;
    const unsigned target = lane < NSH ? (unsigned)(((n_items - lane + NSH - 1) / NSH) * steps) : 0u;
    bool ok = true;
    unsigned spins = 0;
    for (;;) {
      const bool mine = (lane < NSH && target > 0) ? ld_flag(base + lane * CTRS) >= target : true;
      if (__all(mine)) break;
      if (++spins > ab.limit) { abort_raise(ab); ok = false; break; }
      if ((spins & 63u) == 0 && abort_seen(ab)) { ok = false; break; }
    }
    if (lane == 0) *s_flag = ok ? 1 : 0;
  }
  __syncthreads();
  const bool ok = *s_flag != 0;
  __syncthreads();
  acquire_handoff();
  return ok;
}
__device__ __forceinline__ bool wait_one(const unsigned* ctr, unsigned target, const AbortCtl& ab, int* s_flag) {
  if (threadIdx.x == 0) {
    bool ok = true;
    unsigned spins = 0;
    while (ld_flag(ctr) < target) {
      if (++spins > ab.limit) { abort_raise(ab); ok = false; break; }
      if ((spins & 63u) == 0 && abort_seen(ab)) { ok = false; break; }
    }
    *s_flag = ok ? 1 : 0;
  }
  __syncthreads();
  const bool ok = *s_flag != 0;
  __syncthreads();
  acquire_handoff();
  return ok;
}
__device__ __forceinline__ void publish(unsigned* ctr) {
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (threadIdx.x == 0) __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ float sigm_fast(float x) { return __builtin_amdgcn_rcpf(1.f + __expf(-x)); }
__device__ __forceinline__ float tanh_fast(float x) { return 2.f * __builtin_amdgcn_rcpf(1.f + __expf(-2.f * x)) - 1.f; }
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  return v;
}

// NB k-blocks of 16 floats per wave (block wave + 4 i): both 16-row batch tiles against the same resident weight fragments.
// `ra` addresses the activation matrix, off0 / off1 are the float offsets of this lane's row in tile 0 / 1 (incl. 4 q).
template <int NB>
__device__ __forceinline__ void mac2(f32x4 (&acc)[2], const float4* w, __amdgpu_buffer_rsrc_t ra, long off0, long off1, int wave) {
  float4 a0[NB], a1[NB];
#pragma unroll
  for (int i = 0; i < NB; ++i) {
    a0[i] = ldb128_sc1(ra, off0 + 16 * (wave + 4 * i));
    a1[i] = ldb128_sc1(ra, off1 + 16 * (wave + 4 * i));
  }
  __builtin_amdgcn_sched_barrier(0);
  f32x4 b0 = {0.f, 0.f, 0.f, 0.f}, b1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int i = 0; i < NB; ++i) {
    acc[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0[i].x, w[i].x, acc[0], 0, 0, 0);
    acc[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1[i].x, w[i].x, acc[1], 0, 0, 0);
    b0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a0[i].y, w[i].y, b0, 0, 0, 0);
    b1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a1[i].y, w[i].y, b1, 0, 0, 0);
    acc[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0[i].z, w[i].z, acc[0], 0, 0, 0);
    acc[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1[i].z, w[i].z, acc[1], 0, 0, 0);
    b0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a0[i].w, w[i].w, b0, 0, 0, 0);
    b1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a1[i].w, w[i].w, b1, 0, 0, 0);
  }
  acc[0] += b0;
  acc[1] += b1;
}
// resident weight fragments: k-blocks wave + 4 i of W row `row` (K-contiguous), starting at column c0
template <int NB>
__device__ __forceinline__ void wload(float4* w, const float* W, long ldw, int row, int c0, int lane, int wave) {
  const int q = lane >> 4;
#pragma unroll
  for (int i = 0; i < NB; ++i) w[i] = *reinterpret_cast<const float4*>(W + (long)row * ldw + c0 + 16 * (wave + 4 * i) + 4 * q);
}
// 4-wave reduction of the two tiles' accumulators: thread tid gets elements (row = tid>>4, col = tid&15) of tile 0 and tile 1
__device__ __forceinline__ void reduce2(const f32x4 (&acc)[2], float (&v)[2], float* red /* [2][4][256] */) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  *reinterpret_cast<f32x4*>(&red[(wave * 64 + lane) * 4]) = acc[0];
  *reinterpret_cast<f32x4*>(&red[1024 + (wave * 64 + lane) * 4]) = acc[1];
  __syncthreads();
  const int row = tid >> 4, col = tid & 15;
  const int src = ((row >> 2) * 16 + col) * 4 + (row & 3);
  v[0] = red[src] + red[256 + src] + red[512 + src] + red[768 + src];
  v[1] = red[1024 + src] + red[1280 + src] + red[1536 + src] + red[1792 + src];
  __syncthreads();
}

__global__ __launch_bounds__(256, 1) void decoder_wide_fwd(WideArgs a) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  // LDS: enc slice [chunk][WH] | sacc [4][WH] | red [2048] | zt [512] | sml [16] | scr [chunk padded]
  float* s_enc = lds;
  float* s_acc = s_enc + (size_t)a.chunk * WH;
  float* s_red = s_acc + 4 * WH;
  float* s_zt = s_red + 2048;
  float* s_ml = s_zt + 512;
  __shared__ int s_flag;
  const int w = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r = lane & 15, q4 = (lane >> 4) * 4;
  const int B = a.B;
  const int row0 = min(r, B - 1), row1 = min(16 + r, B - 1);
  unsigned* c_cell = a.ctr + (size_t)C_CELL * NSH * CTRS;
  unsigned* c_q = a.ctr + (size_t)C_Q * NSH * CTRS;
  unsigned* c_cmb = a.ctr + (size_t)C_CMB * NSH * CTRS;
  unsigned* c_ctx = a.ctr + (size_t)C_CTX * NSH * CTRS;
  unsigned* c_log = a.ctr + (size_t)C_LOG * NSH * CTRS;
  unsigned* c_arg = a.ctr + (size_t)C_ARG * NSH * CTRS;
  unsigned* c_row = a.ctr + (size_t)C_N * NSH * CTRS;
  const bool is_q = w >= Q0 && w < Q0 + NQ, is_ctx = w >= CTX0 && w < CTX0 + NCTX, is_cmb = w >= CMB0 && w < CMB0 + B;
  const bool is_att = w < B * a.nsplit;
  const int ab_ = w % B, asp = w / B;                         // attention item (batch row, chunk)
  const int t0 = asp * a.chunk, nrow = is_att ? max(0, min(a.chunk, a.T - t0)) : 0;

  // ---- resident weights
  float4 wE[2], wA[16], wHh[16];                              // cell: gate row 16w + r, k-blocks wave + 4i of [emb | ht | h]
  wload<2>(wE, a.Wu, WXI, 16 * w + r, 0, lane, wave);
  wload<16>(wA, a.Wu, WXI, 16 * w + r, WE, lane, wave);
  wload<16>(wHh, a.Wl, WH, 16 * w + r, 0, lane, wave);
  float4 wX[32];                                              // Q: Wa row (16 k-blocks); CTX: Wc row (32 k-blocks)
  if (is_q) wload<16>(wX, a.Wa, WH, 16 * (w - Q0) + r, 0, lane, wave);
  if (is_ctx) wload<32>(wX, a.Wc, 2 * WH, 16 * (w - CTX0) + r, 0, lane, wave);
  // ---- this workgroup's slice of enc_states -> LDS (stays for the whole launch)
  for (int i = tid; i < nrow * (WH / 4); i += 256) {
    const int j = i / (WH / 4), c = (i % (WH / 4)) * 4;
    *reinterpret_cast<float4*>(&s_enc[j * WH + c]) = *reinterpret_cast<const float4*>(a.enc + ((long)ab_ * a.T + t0 + j) * WH + c);
  }
  // cell state of this thread's (batch row, unit): threads 0..127, row = tid >> 2, unit = 4w + (tid & 3)
  const int crow = tid >> 2, cu = 4 * w + (tid & 3);
  float c_state = 0.f;
  if (tid < 128 && crow < B) c_state = a.C[((long)a.s0 * B + crow) * WH + cu];
  const float4 bz = tid < 128 ? *reinterpret_cast<const float4*>(a.bias + 4 * cu) : make_float4(0.f, 0.f, 0.f, 0.f);
  const __amdgpu_buffer_rsrc_t r_hr = make_rsrc(a.HR), r_x0 = make_rsrc(a.X0), r_cvh = make_rsrc(a.CVH), r_q = make_rsrc(a.Q),
                               r_ht = make_rsrc(a.HT);
  int nfed = 0;                                               // steps of this launch whose argmax was fed back so far
  __syncthreads();

  for (int s = a.s0; s <= a.s1; ++s) {
    const int n = s - a.s0 + 1;                               // arrivals per item up to and including this step
    const bool fed_in = s > a.s0 && a.use_truth[s] == 0;      // this step's token is the previous step's argmax, found inside this launch
    // ================= CELL
    f32x4 acc[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
    if (n > 1 && !wait_sh(c_cell, WG_, n - 1, a.ab, &s_flag)) return;      // h_{s-1} of every unit
    mac2<16>(acc, wHh, r_hr, ((long)s * B + row0) * WH + q4, ((long)s * B + row1) * WH + q4, wave);
    if (fed_in && !wait_sh(c_arg, B, nfed, a.ab, &s_flag)) return;          // the fed-back tokens (ARG of step s - 1)
    {   // embedding part (off the chain unless the token is fed back): rows gathered from the table, times the embedding dropout mask
      int tok[2];
#pragma unroll
      for (int mt = 0; mt < 2; ++mt) {
        const int rw = mt ? row1 : row0;
        int t = (a.use_truth[s] || s == 0) ? a.y[(long)rw * a.L + s] : ldi_sc1(a.PRED + (long)(s - 1) * B + rw);
        tok[mt] = t < 0 ? 0 : (t >= a.V ? a.V - 1 : t);
      }
      f32x4 b0 = {0.f, 0.f, 0.f, 0.f}, b1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const int k = 16 * (wave + 4 * i) + q4;
        float4 e0 = *reinterpret_cast<const float4*>(a.embed + (long)tok[0] * WE + k);
        float4 e1 = *reinterpret_cast<const float4*>(a.embed + (long)tok[1] * WE + k);
        if (a.emb_mask) {
          const float4 m0 = *reinterpret_cast<const float4*>(a.emb_mask + ((long)s * B + row0) * WE + k);
          const float4 m1 = *reinterpret_cast<const float4*>(a.emb_mask + ((long)s * B + row1) * WE + k);
          e0.x *= m0.x; e0.y *= m0.y; e0.z *= m0.z; e0.w *= m0.w;
          e1.x *= m1.x; e1.y *= m1.y; e1.z *= m1.z; e1.w *= m1.w;
        }
        acc[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(e0.x, wE[i].x, acc[0], 0, 0, 0);
        acc[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(e1.x, wE[i].x, acc[1], 0, 0, 0);
        b0 = __builtin_amdgcn_mfma_f32_16x16x4f32(e0.y, wE[i].y, b0, 0, 0, 0);
        b1 = __builtin_amdgcn_mfma_f32_16x16x4f32(e1.y, wE[i].y, b1, 0, 0, 0);
        acc[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(e0.z, wE[i].z, acc[0], 0, 0, 0);
        acc[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(e1.z, wE[i].z, acc[1], 0, 0, 0);
        b0 = __builtin_amdgcn_mfma_f32_16x16x4f32(e0.w, wE[i].w, b0, 0, 0, 0);
        b1 = __builtin_amdgcn_mfma_f32_16x16x4f32(e1.w, wE[i].w, b1, 0, 0, 0);
      }
      acc[0] += b0;
      acc[1] += b1;
      // the saved embedding rows (the weight gradients' operand) and tokens: workgroup b writes row b
      if (w < B) {
        int t = (a.use_truth[s] || s == 0) ? a.y[(long)w * a.L + s] : ldi_sc1(a.PRED + (long)(s - 1) * B + w);
        t = t < 0 ? 0 : (t >= a.V ? a.V - 1 : t);
        if (tid == 0) a.TOK[(long)s * B + w] = t;
        if (tid < WE) {
          float v = a.embed[(long)t * WE + tid];
          if (a.emb_mask) v *= a.emb_mask[((long)s * B + w) * WE + tid];
          a.X0[((long)s * B + w) * WXI + tid] = v;
        }
      }
    }
    if (n > 1 && !wait_sh(c_ctx, NCTX, n - 1, a.ab, &s_flag)) return;      // ht_{s-1} (input feeding)
    mac2<16>(acc, wA, r_x0, ((long)s * B + row0) * WXI + WE + q4, ((long)s * B + row1) * WXI + WE + q4, wave);
    {
      float v[2];
      reduce2(acc, v, s_red);
      s_zt[tid] = v[0];                    // zt[tile][row 0..15][col 0..15], col = 4 * unit + gate
      s_zt[256 + tid] = v[1];
      __syncthreads();
      if (tid < 128 && crow < B) {
        const int mt = crow >> 4, rr = crow & 15, u = tid & 3;
        float4 z = *reinterpret_cast<const float4*>(&s_zt[mt * 256 + rr * 16 + 4 * u]);
        z.x += bz.x; z.y += bz.y; z.z += bz.z; z.w += bz.w;
        const float ga = tanh_fast(z.x), gi = sigm_fast(z.y), gf = sigm_fast(z.z), go = sigm_fast(z.w);
        const float c = ga * gi + gf * c_state;
        const float hh = go * tanh_fast(c);
        c_state = c;
        const long bu = (long)crow * WH + cu;
        *reinterpret_cast<float4*>(a.G + ((long)s * B + crow) * 4 * WH + 4 * cu) = make_float4(ga, gi, gf, go);
        a.C[(long)(s + 1) * B * WH + bu] = c;
        st_sc1(a.HR + (long)(s + 1) * B * WH + bu, hh);
        const float hd = a.rnn_mask ? hh * a.rnn_mask[(long)s * B * WH + bu] : hh;
        st_sc1(a.CVH + ((long)s * B + crow) * 2 * WH + WH + cu, hd);
      }
      publish(c_cell + (w & (NSH - 1)) * CTRS);
    }
    // ================= Q: q = Wa h + ba
    if (is_q) {
      if (!wait_sh(c_cell, WG_, n, a.ab, &s_flag)) return;
      f32x4 aq[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
      mac2<16>(aq, wX, r_cvh, ((long)s * B + row0) * 2 * WH + WH + q4, ((long)s * B + row1) * 2 * WH + WH + q4, wave);
      float v[2];
      reduce2(aq, v, s_red);
      const int col = 16 * (w - Q0) + (tid & 15);
      const float bb = a.ba[col];
#pragma unroll
      for (int mt = 0; mt < 2; ++mt) {
        const int rw = 16 * mt + (tid >> 4);
        if (rw < B) st_sc1(a.Q + ((long)s * B + rw) * WH + col, v[mt] + bb);
      }
      publish(c_q + ((w - Q0) & (NSH - 1)) * CTRS);
    }
    // ================= ATT: scores and partial context of (batch row, chunk)
    if (is_att) {
      if (!wait_sh(c_q, NQ, n, a.ab, &s_flag)) return;
      float4 qv[4], ac[4];
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        qv[c] = ldb128_sc1(r_q, ((long)s * B + ab_) * WH + 256 * c + 4 * lane);
        ac[c] = make_float4(0.f, 0.f, 0.f, 0.f);
      }
      float m = -INFINITY, l = 0.f;
      for (int j = wave; j < nrow; j += 4) {
        float4 e[4];
        float d = 0.f;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          e[c] = *reinterpret_cast<const float4*>(&s_enc[j * WH + 256 * c + 4 * lane]);
          d += e[c].x * qv[c].x + e[c].y * qv[c].y + e[c].z * qv[c].z + e[c].w * qv[c].w;
        }
        const float sc_ = wave_sum(d);
        if (lane == 0) st_sc1(a.ALPHA + ((long)s * B + ab_) * a.Tp + t0 + j, sc_);
        const float mn = fmaxf(m, sc_);
        const float f = __expf(m - mn), p = __expf(sc_ - mn);
        l = l * f + p;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          ac[c].x = ac[c].x * f + p * e[c].x; ac[c].y = ac[c].y * f + p * e[c].y;
          ac[c].z = ac[c].z * f + p * e[c].z; ac[c].w = ac[c].w * f + p * e[c].w;
        }
        m = mn;
      }
#pragma unroll
      for (int c = 0; c < 4; ++c) *reinterpret_cast<float4*>(&s_acc[wave * WH + 256 * c + 4 * lane]) = ac[c];
      if (lane == 0) { s_ml[wave] = m; s_ml[4 + wave] = l; }
      __syncthreads();
      const float M = fmaxf(fmaxf(s_ml[0], s_ml[1]), fmaxf(s_ml[2], s_ml[3]));
      float wg[4];
#pragma unroll
      for (int k = 0; k < 4; ++k) wg[k] = s_ml[k] == -INFINITY ? 0.f : __expf(s_ml[k] - M);
      float* prow = a.PART + ((long)ab_ * a.nsplit + asp) * PARTW;
      for (int i = tid; i < WH; i += 256)
        st_sc1(&prow[4 + i], s_acc[i] * wg[0] + s_acc[WH + i] * wg[1] + s_acc[2 * WH + i] * wg[2] + s_acc[3 * WH + i] * wg[3]);
      if (tid == 0) {
        st_sc1(&prow[0], M);
        st_sc1(&prow[1], s_ml[4] * wg[0] + s_ml[5] * wg[1] + s_ml[6] * wg[2] + s_ml[7] * wg[3]);
      }
      publish(c_row + (size_t)ab_ * CTRS);
    }
    // ================= CMB: one batch row's context vector and normalised alpha
    if (is_cmb) {
      const int b = w - CMB0;
      // (chunks behind the last frame hold no rows: their partials are (-inf, 0, 0) and weigh nothing)
      if (!wait_one(c_row + (size_t)b * CTRS, (unsigned)(a.nsplit * n), a.ab, &s_flag)) return;
      // the chunks' (max, sum) pairs once, through LDS; then every thread merges its 4 columns of all partial sums with independent 16-byte loads
      // (one exposed round trip instead of one per partial)
      const __amdgpu_buffer_rsrc_t r_part = make_rsrc(a.PART + (long)b * a.nsplit * PARTW);
      if (tid < a.nsplit) {
        const float4 ml = ldb128_sc1(r_part, (long)tid * PARTW);
        s_acc[tid] = ml.x;
        s_acc[64 + tid] = ml.y;
      }
      __syncthreads();
      float M = -INFINITY;
      for (int k = 0; k < a.nsplit; ++k) M = fmaxf(M, s_acc[k]);
      float Lsum = 0.f;
      for (int k = 0; k < a.nsplit; ++k) Lsum += s_acc[k] == -INFINITY ? 0.f : s_acc[64 + k] * __expf(s_acc[k] - M);
      const float inv = 1.f / Lsum;
      __syncthreads();
      if (tid < a.nsplit) s_acc[128 + tid] = s_acc[tid] == -INFINITY ? 0.f : __expf(s_acc[tid] - M) * inv;      // weight of chunk tid
      __syncthreads();
      {
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        for (int k0 = 0; k0 < a.nsplit; k0 += 8) {
          float4 p[8];
#pragma unroll
          for (int k = 0; k < 8; ++k) p[k] = ldb128_sc1(r_part, (long)min(k0 + k, a.nsplit - 1) * PARTW + 4 + 4 * tid);
#pragma unroll
          for (int k = 0; k < 8; ++k) {
            const float wk = k0 + k < a.nsplit ? s_acc[128 + k0 + k] : 0.f;
            v.x += p[k].x * wk; v.y += p[k].y * wk; v.z += p[k].z * wk; v.w += p[k].w * wk;
          }
        }
        float* cv = a.CVH + ((long)s * B + b) * 2 * WH + 4 * tid;
        st_sc1(cv, v.x); st_sc1(cv + 1, v.y); st_sc1(cv + 2, v.z); st_sc1(cv + 3, v.w);
      }
      float* al = a.ALPHA + ((long)s * B + b) * a.Tp;
      for (int t = tid; t < a.Tp; t += 256) al[t] = t < a.T ? __expf(ld_sc1(&al[t]) - M) * inv : 0.f;
      publish(c_cmb + (b & (NSH - 1)) * CTRS);
    }
    // ================= CTX: ht = tanh(Wc [cv; h] + bc)
    if (is_ctx) {
      f32x4 ax[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
      const long o0 = ((long)s * B + row0) * 2 * WH + q4, o1 = ((long)s * B + row1) * 2 * WH + q4;
      // the h half of [cv; h] (k-blocks wave + 4i, i >= 16: columns [1024, 2048)) is complete since CELL: off the chain, in front of the wait
      if (!is_att && !wait_sh(c_cell, WG_, n, a.ab, &s_flag)) return;        // (attention workgroups have seen Q, which saw CELL)
      mac2<16>(ax, wX + 16, r_cvh, o0 + 1024, o1 + 1024, wave);
      if (!wait_sh(c_cmb, B, n, a.ab, &s_flag)) return;
      mac2<16>(ax, wX, r_cvh, o0, o1, wave);                                  // i < 16: columns [0, 1024), the context vectors
      float v[2];
      reduce2(ax, v, s_red);
      const int col = 16 * (w - CTX0) + (tid & 15);
      const float bb = a.bc[col];
#pragma unroll
      for (int mt = 0; mt < 2; ++mt) {
        const int rw = 16 * mt + (tid >> 4);
        if (rw < B) {
          const float ht = tanh_fast(v[mt] + bb);
          st_sc1(a.HT + ((long)(s + 1) * B + rw) * WA + col, ht);
          if (s + 1 < a.S) st_sc1(a.X0 + ((long)(s + 1) * B + rw) * WXI + WE + col, ht);
        }
      }
      publish(c_ctx + ((w - CTX0) & (NSH - 1)) * CTRS);
    }
    // ================= LOG / ARG: only when the NEXT step is not teacher-forced (its token is this step's argmax; every step's loss is
    // scored behind the loop by decoder.hip).  The logits weights are not resident: each workgroup streams the rows of its 16-column tiles
    // (two at V = 8004: 128 KB) from L2 / Infinity Cache, keeps a per-(row, tile) maximum, and one workgroup per batch row picks the class.
    if (s + 1 < a.S && s < a.s1 && a.use_truth[s + 1] == 0) {
      ++nfed;
      if (!wait_sh(c_ctx, NCTX, n, a.ab, &s_flag)) return;
      for (int t = w; t < a.ntile; t += WG_) {
        f32x4 lg[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
        const float* wrow = a.Wo + (long)min(16 * t + r, a.V - 1) * WA + q4;
#pragma unroll
        for (int half = 0; half < 2; ++half) {
          float4 wt[8];
#pragma unroll
          for (int i = 0; i < 8; ++i) wt[i] = *reinterpret_cast<const float4*>(wrow + 512 * half + 16 * (wave + 4 * i));
          mac2<8>(lg, wt, r_ht, ((long)(s + 1) * B + row0) * WA + q4 + 512 * half, ((long)(s + 1) * B + row1) * WA + q4 + 512 * half, wave);
        }
        float v[2];
        reduce2(lg, v, s_red);
        const int cls = 16 * t + (tid & 15);
        const float bb = a.bo[min(cls, a.V - 1)];
#pragma unroll
        for (int mt = 0; mt < 2; ++mt) {
          float m = cls < a.V ? v[mt] + bb : -INFINITY;
          int mi = cls;
#pragma unroll
          for (int o = 8; o > 0; o >>= 1) {           // the 16 lanes of a row: maximum, lowest class id on ties
            const float om = __shfl_xor(m, o);
            const int oi = __shfl_xor(mi, o);
            if (om > m || (om == m && oi < mi)) { m = om; mi = oi; }
          }
          const int rw = 16 * mt + (tid >> 4);
          if ((tid & 15) == 0 && rw < B) {
            float* lm = a.LMAX + ((long)rw * a.ntile + t) * 2;
            st_sc1(lm, m);
            st_sc1(lm + 1, __int_as_float(mi));
          }
        }
      }
      publish(c_log + (w & (NSH - 1)) * CTRS);
      if (w < B) {
        if (!wait_sh(c_log, WG_, nfed, a.ab, &s_flag)) return;
        float m = -INFINITY;
        int mi = 0x7fffffff;
        for (int t = tid; t < a.ntile; t += 256) {
          const float* lm = a.LMAX + ((long)w * a.ntile + t) * 2;
          const float om = ld_sc1(lm);
          const int oi = __float_as_int(ld_sc1(lm + 1));
          if (om > m || (om == m && oi < mi)) { m = om; mi = oi; }
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
          const float om = __shfl_xor(m, o);
          const int oi = __shfl_xor(mi, o);
          if (om > m || (om == m && oi < mi)) { m = om; mi = oi; }
        }
        if (lane == 0) { s_ml[wave] = m; s_ml[8 + wave] = __int_as_float(mi); }
        __syncthreads();
        if (tid == 0) {
          for (int k = 1; k < 4; ++k) {
            const float om = s_ml[k];
            const int oi = __float_as_int(s_ml[8 + k]);
            if (om > m || (om == m && oi < mi)) { m = om; mi = oi; }
          }
          sti_sc1(a.PRED + (long)s * B + w, mi);
        }
        publish(c_arg + (w & (NSH - 1)) * CTRS);
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------- backward loop
// All S steps of the reversed loop in ONE launch (no argmax dependence in the backward).  Per step st, S-1 .. 0:
//   P1    workgroups 0..127:  d_cvh = d_pre Wc                       (16 of the 2048 columns each, K = 1024)
//   ATTB  workgroup (b, chunk): ds_t = alpha_t (e_t . d_cv - cv . d_cv), partial dq = sum_t ds_t e_t   (encoder slice in LDS)
//   DQC   workgroups 128..128+B-1: dq[b] = sum of the chunks' partials
//   P3    workgroups 128..191: dh_top = d_cvh[:, H:] + dq Wa         (16 columns each, K = 1024)
//   CELLB all 256 workgroups (tile j = w / 4 of 16 units, quarter kq = w % 4: units 16j + 4kq ..+3): dh = mask dh_top + sum of the four
//         K-quarter partials of dz_{st+1} Wl, the LSTM cell's pointwise backward -> dz_st (in place of the saved gates), dc carried in registers
//   DZ    all 256 workgroups: K-quarter kq of dz_st (1024 of the 4096 gate rows) against TWO resident weight slices: the recurrent one
//         (partial of dz_st Wl for step st-1's CELLB) and the input-feeding one (partial of the carry dz_st Wu[:, E:])
//   P5R   workgroups 192..255: d_pre[st-1] = (dlogits Wo (batched in front of the launch) + sum of the four carry partials) (1 - ht^2)
// Chain: P1 -> ATTB -> DQC -> P3 -> CELLB -> DZ -> P5R -> next P1.  The embedding columns of d_x0 are one batched product behind the loop.
constexpr int BP1 = 0, BP3 = 128, BDQC = 128, BP5R = 192;
constexpr int NP1 = 2 * WH / 16, NP3 = WH / 16, NTILE = WH / 16;
enum { D_P1 = 0, D_DQC, D_P3, D_CELL, D_DPRE, D_N };

struct WideBwdArgs {
  int B, S, T, Tp, nsplit, chunk;
  const float *WcT, *WaT, *WlT, *WuT;      // K-contiguous transposes: [2H][A], [H][H], [H][4H], [E+A][4H]
  const float* enc;
  const float *ALPHA, *CVH, *HT, *C, *rnn_mask;
  float *G, *DPRE, *DCVH, *DS, *DQ, *DHTOP, *DC0;
  float *PARTB;        // [B][nsplit][H]
  float *PREC, *PCAR;  // [64 tiles][4][32][16] partial sums of the recurrent / carry products
  unsigned* ctr;       // [D_N][NSH] lines | [B] per-row lines | [64] rec-tile lines | [64] carry-tile lines | abort
  AbortCtl ab;
};

__global__ __launch_bounds__(256, 1) void decoder_wide_bwd(WideBwdArgs a) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  float* s_enc = lds;
  float* s_acc = s_enc + (size_t)a.chunk * WH;      // [4][WH]
  float* s_red = s_acc + 4 * WH;                    // [2048]
  float* s_zt = s_red + 2048;                       // [512]
  __shared__ int s_flag;
  const int w = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r = lane & 15, q4 = (lane >> 4) * 4;
  const int B = a.B, S = a.S;
  const int row0 = min(r, B - 1), row1 = min(16 + r, B - 1);
  unsigned* c_p1 = a.ctr + (size_t)D_P1 * NSH * CTRS;
  unsigned* c_dqc = a.ctr + (size_t)D_DQC * NSH * CTRS;
  unsigned* c_p3 = a.ctr + (size_t)D_P3 * NSH * CTRS;
  unsigned* c_cell = a.ctr + (size_t)D_CELL * NSH * CTRS;
  unsigned* c_dpre = a.ctr + (size_t)D_DPRE * NSH * CTRS;
  unsigned* c_row = a.ctr + (size_t)D_N * NSH * CTRS;
  unsigned* c_rec = c_row + (size_t)32 * CTRS;
  unsigned* c_car = c_rec + (size_t)NTILE * CTRS;
  const bool is_p1 = w < NP1, is_p3 = w >= BP3 && w < BP3 + NP3, is_dqc = w >= BDQC && w < BDQC + B, is_p5r = w >= BP5R;
  const bool is_att = w < B * a.nsplit;
  const int ab_ = w % B, asp = w / B;
  const int t0 = asp * a.chunk, nrow = is_att ? max(0, min(a.chunk, a.T - t0)) : 0;
  const int tj = w >> 2, kq = w & 3;                 // CELLB / DZ item: tile of 16 units, K quarter

  // ---- resident weights
  float4 wRec[16], wCar[16], wX[16];
  wload<16>(wRec, a.WlT, 4 * WH, 16 * tj + r, 1024 * kq, lane, wave);            // dz Wl: output unit 16 tj + r, gate rows of quarter kq
  wload<16>(wCar, a.WuT, 4 * WH, WE + 16 * tj + r, 1024 * kq, lane, wave);       // dz Wu[:, E + ...]: carry column 16 tj + r
  if (is_p1) wload<16>(wX, a.WcT, WA, 16 * w + r, 0, lane, wave);
  if (is_p3) wload<16>(wX, a.WaT, WH, 16 * (w - BP3) + r, 0, lane, wave);
  for (int i = tid; i < nrow * (WH / 4); i += 256) {
    const int j = i / (WH / 4), c = (i % (WH / 4)) * 4;
    *reinterpret_cast<float4*>(&s_enc[j * WH + c]) = *reinterpret_cast<const float4*>(a.enc + ((long)ab_ * a.T + t0 + j) * WH + c);
  }
  const int crow = tid >> 2, cu = 16 * tj + 4 * kq + (tid & 3), ccol = 4 * kq + (tid & 3);
  float dc_next = 0.f;
  const __amdgpu_buffer_rsrc_t r_dpre = make_rsrc(a.DPRE), r_dcvh = make_rsrc(a.DCVH), r_dq = make_rsrc(a.DQ), r_g = make_rsrc(a.G),
                               r_part = make_rsrc(a.PARTB);
  __syncthreads();

  for (int st = S - 1; st >= 0; --st) {
    const int n = S - st;                            // arrivals per item up to and including this step
    // ================= P1: d_cvh[st] = d_pre[st] Wc
    if (is_p1) {
      if (n > 1 && !wait_sh(c_dpre, NTILE, n - 1, a.ab, &s_flag)) return;
      f32x4 acc[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
      mac2<16>(acc, wX, r_dpre, ((long)st * B + row0) * WA + q4, ((long)st * B + row1) * WA + q4, wave);
      float v[2];
      reduce2(acc, v, s_red);
#pragma unroll
      for (int mt = 0; mt < 2; ++mt) {
        const int rw = 16 * mt + (tid >> 4);
        if (rw < B) st_sc1(a.DCVH + ((long)st * B + rw) * 2 * WH + 16 * w + (tid & 15), v[mt]);
      }
      publish(c_p1 + (w & (NSH - 1)) * CTRS);
    }
    // ================= ATTB
    if (is_att) {
      if (!wait_sh(c_p1, NP1, n, a.ab, &s_flag)) return;
      float4 dcv[4], cvv[4], ac[4];
      float c0 = 0.f;
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        dcv[c] = ldb128_sc1(r_dcvh, ((long)st * B + ab_) * 2 * WH + 256 * c + 4 * lane);
        cvv[c] = *reinterpret_cast<const float4*>(a.CVH + ((long)st * B + ab_) * 2 * WH + 256 * c + 4 * lane);
        c0 += dcv[c].x * cvv[c].x + dcv[c].y * cvv[c].y + dcv[c].z * cvv[c].z + dcv[c].w * cvv[c].w;
        ac[c] = make_float4(0.f, 0.f, 0.f, 0.f);
      }
      c0 = wave_sum(c0);
      for (int j = wave; j < nrow; j += 4) {
        float4 e[4];
        float d = 0.f;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          e[c] = *reinterpret_cast<const float4*>(&s_enc[j * WH + 256 * c + 4 * lane]);
          d += e[c].x * dcv[c].x + e[c].y * dcv[c].y + e[c].z * dcv[c].z + e[c].w * dcv[c].w;
        }
        const float al = a.ALPHA[((long)st * B + ab_) * a.Tp + t0 + j];
        const float ds = al * (wave_sum(d) - c0);
        if (lane == 0) a.DS[((long)st * B + ab_) * a.Tp + t0 + j] = ds;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          ac[c].x += ds * e[c].x; ac[c].y += ds * e[c].y; ac[c].z += ds * e[c].z; ac[c].w += ds * e[c].w;
        }
      }
#pragma unroll
      for (int c = 0; c < 4; ++c) *reinterpret_cast<float4*>(&s_acc[wave * WH + 256 * c + 4 * lane]) = ac[c];
      __syncthreads();
      float* prow = a.PARTB + ((long)ab_ * a.nsplit + asp) * WH;
      for (int i = tid; i < WH; i += 256) st_sc1(&prow[i], s_acc[i] + s_acc[WH + i] + s_acc[2 * WH + i] + s_acc[3 * WH + i]);
      publish(c_row + (size_t)ab_ * CTRS);
    }
    // ================= DQC: dq[b] = sum of the chunks' partials
    if (is_dqc) {
      const int b = w - BDQC;
      if (!wait_one(c_row + (size_t)b * CTRS, (unsigned)(a.nsplit * n), a.ab, &s_flag)) return;
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      for (int k0 = 0; k0 < a.nsplit; k0 += 8) {
        float4 p[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) p[k] = ldb128_sc1(r_part, ((long)b * a.nsplit + min(k0 + k, a.nsplit - 1)) * WH + 4 * tid);
#pragma unroll
        for (int k = 0; k < 8; ++k)
          if (k0 + k < a.nsplit) { v.x += p[k].x; v.y += p[k].y; v.z += p[k].z; v.w += p[k].w; }
      }
      float* dq = a.DQ + ((long)st * B + b) * WH + 4 * tid;
      st_sc1(dq, v.x); st_sc1(dq + 1, v.y); st_sc1(dq + 2, v.z); st_sc1(dq + 3, v.w);
      publish(c_dqc + (b & (NSH - 1)) * CTRS);
    }
    // ================= P3: dh_top = d_cvh[:, H:] + dq Wa
    if (is_p3) {
      if (!wait_sh(c_dqc, B, n, a.ab, &s_flag)) return;
      f32x4 acc[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
      mac2<16>(acc, wX, r_dq, ((long)st * B + row0) * WH + q4, ((long)st * B + row1) * WH + q4, wave);
      float v[2];
      reduce2(acc, v, s_red);
      const int col = 16 * (w - BP3) + (tid & 15);
#pragma unroll
      for (int mt = 0; mt < 2; ++mt) {
        const int rw = 16 * mt + (tid >> 4);
        if (rw < B) st_sc1(a.DHTOP + (long)rw * WH + col, v[mt] + ld_sc1(a.DCVH + ((long)st * B + rw) * 2 * WH + WH + col));
      }
      publish(c_p3 + ((w - BP3) & (NSH - 1)) * CTRS);
    }
    // ================= CELLB: pointwise LSTM backward of this workgroup's 4 units
    {
      if (!wait_sh(c_p3, NP3, n, a.ab, &s_flag)) return;
      if (n > 1 && !wait_one(c_rec + (size_t)tj * CTRS, (unsigned)(4 * (n - 1)), a.ab, &s_flag)) return;
      if (tid < 128 && crow < B) {
        const long bu = (long)crow * WH + cu;
        float dy = ld_sc1(a.DHTOP + bu);
        if (a.rnn_mask) dy *= a.rnn_mask[(long)st * B * WH + bu];
        float dh = dy;
        if (n > 1) {
          const float* pr = a.PREC + ((long)tj * 4 * 32 + crow) * 16 + ccol;
          dh += ld_sc1(pr) + ld_sc1(pr + 512) + ld_sc1(pr + 1024) + ld_sc1(pr + 1536);
        }
        float* gp = a.G + ((long)st * B + crow) * 4 * WH + 4 * cu;
        const float4 g = ldb128_sc1(r_g, ((long)st * B + crow) * 4 * WH + 4 * cu);      // the saved gates a, i, f, o (dz goes in their place)
        const float ga = g.x, gi = g.y, gf = g.z, go = g.w;
        const float tc = tanhf(a.C[(long)(st + 1) * B * WH + bu]);
        const float cp = a.C[(long)st * B * WH + bu];
        const float dc = dh * go * (1.f - tc * tc) + dc_next;
        st_sc1(gp, dc * gi * (1.f - ga * ga));
        st_sc1(gp + 1, dc * ga * gi * (1.f - gi));
        st_sc1(gp + 2, dc * cp * gf * (1.f - gf));
        st_sc1(gp + 3, dh * tc * go * (1.f - go));
        dc_next = dc * gf;
        if (st == 0) a.DC0[bu] = dc_next;
      }
      publish(c_cell + (w & (NSH - 1)) * CTRS);
    }
    if (st == 0) break;
    // ================= DZ: quarter kq of dz_st against the recurrent and the carry weight slices
    {
      if (!wait_sh(c_cell, WG_, n, a.ab, &s_flag)) return;
      const long o0 = ((long)st * B + row0) * 4 * WH + 1024 * kq + q4, o1 = ((long)st * B + row1) * 4 * WH + 1024 * kq + q4;
      float4 a0[16], a1[16];
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        a0[i] = ldb128_sc1(r_g, o0 + 16 * (wave + 4 * i));
        a1[i] = ldb128_sc1(r_g, o1 + 16 * (wave + 4 * i));
      }
      __builtin_amdgcn_sched_barrier(0);
      f32x4 rec[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}}, car[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        rec[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0[i].x, wRec[i].x, rec[0], 0, 0, 0);
        rec[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1[i].x, wRec[i].x, rec[1], 0, 0, 0);
        car[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0[i].x, wCar[i].x, car[0], 0, 0, 0);
        car[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1[i].x, wCar[i].x, car[1], 0, 0, 0);
        rec[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0[i].y, wRec[i].y, rec[0], 0, 0, 0);
        rec[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1[i].y, wRec[i].y, rec[1], 0, 0, 0);
        car[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0[i].y, wCar[i].y, car[0], 0, 0, 0);
        car[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1[i].y, wCar[i].y, car[1], 0, 0, 0);
        rec[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0[i].z, wRec[i].z, rec[0], 0, 0, 0);
        rec[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1[i].z, wRec[i].z, rec[1], 0, 0, 0);
        car[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0[i].z, wCar[i].z, car[0], 0, 0, 0);
        car[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1[i].z, wCar[i].z, car[1], 0, 0, 0);
        rec[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0[i].w, wRec[i].w, rec[0], 0, 0, 0);
        rec[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1[i].w, wRec[i].w, rec[1], 0, 0, 0);
        car[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0[i].w, wCar[i].w, car[0], 0, 0, 0);
        car[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1[i].w, wCar[i].w, car[1], 0, 0, 0);
      }
      float v[2];
      float* pc = a.PCAR + (long)w * 512;            // [tile][quarter] = [w]: [32 rows][16 cols]
      reduce2(car, v, s_red);
      st_sc1(pc + tid, v[0]);
      st_sc1(pc + 256 + tid, v[1]);
      publish(c_car + (size_t)tj * CTRS);          // the carry first: it is on the chain
      float* pr = a.PREC + (long)w * 512;
      reduce2(rec, v, s_red);
      st_sc1(pr + tid, v[0]);
      st_sc1(pr + 256 + tid, v[1]);
      publish(c_rec + (size_t)tj * CTRS);
    }
    // ================= P5R: d_pre[st-1] = (dlogits Wo + carry) (1 - ht_{st-1}^2)
    if (is_p5r) {
      const int j = w - BP5R;
      if (!wait_one(c_car + (size_t)j * CTRS, (unsigned)(4 * n), a.ab, &s_flag)) return;
      const float* pc = a.PCAR + (long)j * 4 * 512;
#pragma unroll
      for (int mt = 0; mt < 2; ++mt) {
        const int rw = 16 * mt + (tid >> 4), col = 16 * j + (tid & 15);
        if (rw < B) {
          const int e = mt * 256 + tid;
          const float carry = ld_sc1(pc + e) + ld_sc1(pc + 512 + e) + ld_sc1(pc + 1024 + e) + ld_sc1(pc + 1536 + e);
          float* dp = a.DPRE + ((long)(st - 1) * B + rw) * WA + col;
          const float y = a.HT[((long)st * B + rw) * WA + col];
          st_sc1(dp, (*dp + carry) * (1.f - y * y));
        }
      }
      publish(c_dpre + (j & (NSH - 1)) * CTRS);
    }
  }
}

size_t wide_lds_bytes(int chunk) { return ((size_t)chunk * WH + 4 * WH + 2048 + 512 + 16) * sizeof(float); }

}  // namespace

// Applicable: configs[4]'s decoder shape, one layer, one attention head, input feeding, no LayerNorm; the device's CUs hold the grid; the
// time slices fit LDS.  (decoder.hip additionally needs the host copy of the teacher-forcing flags to cut the loop into segments.)
bool decoder_wide_applicable(const astk_decoder_desc* d, int* nsplit_out, int* chunk_out) {
  if (!tune_on(TUNE_DEC_WIDE)) return false;
  if (d->H != WH || d->A != WA || d->E != WE || d->n_layers != 1 || d->n_attn > 1 || d->no_feed_attn || d->ln) return false;
  if (d->B < 1 || d->B > 32 || d->T < 1 || d->V < 2 || device_cu_count() < WG_) return false;
  int nsplit = WG_ / d->B;
  if (nsplit > d->T) nsplit = d->T;
  if (nsplit > 64) nsplit = 64;
  const int chunk = (d->T + nsplit - 1) / nsplit;
  nsplit = (d->T + chunk - 1) / chunk;
  if (wide_lds_bytes(chunk) > 158 * 1024) return false;      // slices of up to 32 rows (T'' <= 256 at batch 32)
  if (nsplit_out) *nsplit_out = nsplit;
  if (chunk_out) *chunk_out = chunk;
  return true;
}

size_t decoder_wide_part_floats(const astk_decoder_desc* d) {
  int ns = 1, ch = 1;
  return decoder_wide_applicable(d, &ns, &ch) ? (size_t)d->B * ns * PARTW + (size_t)d->B * ((d->V + 15) / 16) * 2 : 4;
}
size_t decoder_wide_ctr_words(const astk_decoder_desc* d) {
  return decoder_wide_applicable(d, nullptr, nullptr) ? ((size_t)C_N * NSH + d->B + 1) * CTRS : 4;
}

int decoder_wide_fwd_launch(const astk_decoder_desc* d, const astk_decoder_params* prm, const float* enc, const int32_t* y,
                            const int32_t* use_truth, const float* emb_mask, const float* rnn_mask, const DecWideBuffers& bf, int s0, int s1,
                            hipStream_t s) {
  int nsplit = 1, chunk = 1;
  ASTK_CHECK(decoder_wide_applicable(d, &nsplit, &chunk), "decoder_wide: not applicable");
  ASTK_CHECK(s0 >= 0 && s1 >= s0 && s1 < d->L - 1, "decoder_wide: bad segment [%d, %d]", s0, s1);
  WideArgs a;
  memset(&a, 0, sizeof(a));
  a.B = d->B; a.S = d->L - 1; a.L = d->L; a.T = d->T; a.Tp = (d->T + 3) / 4 * 4; a.V = d->V; a.s0 = s0; a.s1 = s1;
  a.nsplit = nsplit; a.chunk = chunk;
  a.embed = prm->embed; a.Wu = prm->lstm[0].Wu; a.bias = prm->lstm[0].b; a.Wl = prm->lstm[0].Wl;
  a.Wa = prm->Wa; a.ba = prm->ba; a.Wc = prm->Wc; a.bc = prm->bc; a.Wo = prm->Wo; a.bo = prm->bo;
  a.ntile = (d->V + 15) / 16;
  a.LMAX = bf.PART + (size_t)d->B * nsplit * PARTW;
  a.enc = enc; a.y = y; a.use_truth = use_truth; a.PRED = bf.PRED; a.emb_mask = emb_mask; a.rnn_mask = rnn_mask;
  a.TOK = bf.TOK; a.X0 = bf.X0; a.G = bf.G; a.C = bf.C; a.HR = bf.HR; a.Q = bf.Q; a.ALPHA = bf.ALPHA; a.CVH = bf.CVH; a.HT = bf.HT;
  a.PART = bf.PART; a.ctr = bf.ctr;
  const size_t nctr = ((size_t)C_N * NSH + d->B + 1) * CTRS;
  a.ab = abort_ctl(bf.ctr + ((size_t)C_N * NSH + d->B) * CTRS, PERSIST_DEC_FWD);
  ASTK_HIP(hipMemsetAsync(bf.ctr, 0, nctr * sizeof(unsigned), s));
  const size_t shm = wide_lds_bytes(chunk);
  static bool attr_done = false;
  if (!attr_done) {
    (void)hipFuncSetAttribute((const void*)decoder_wide_fwd, hipFuncAttributeMaxDynamicSharedMemorySize, 158 * 1024);
    attr_done = true;
  }
  hipLaunchKernelGGL(decoder_wide_fwd, dim3(WG_), dim3(256), shm, s, a);
  ASTK_LAUNCH_CHECK();
  return 0;
}

size_t decoder_wide_bwd_floats(const astk_decoder_desc* d) {       // PARTB + PREC + PCAR
  int ns = 1, ch = 1;
  return decoder_wide_applicable(d, &ns, &ch) ? (size_t)d->B * ns * WH + 2 * (size_t)WG_ * 512 : 4;
}
size_t decoder_wide_bwd_ctr_words(const astk_decoder_desc* d) {
  return decoder_wide_applicable(d, nullptr, nullptr) ? ((size_t)D_N * NSH + 32 + 2 * NTILE + 1) * CTRS : 4;
}

int decoder_wide_bwd_launch(const astk_decoder_desc* d, const float* enc, const float* rnn_mask, const DecWideBwdBuffers& bf, hipStream_t s) {
  int nsplit = 1, chunk = 1;
  ASTK_CHECK(decoder_wide_applicable(d, &nsplit, &chunk), "decoder_wide_bwd: not applicable");
  WideBwdArgs a;
  memset(&a, 0, sizeof(a));
  a.B = d->B; a.S = d->L - 1; a.T = d->T; a.Tp = (d->T + 3) / 4 * 4; a.nsplit = nsplit; a.chunk = chunk;
  a.WcT = bf.WcT; a.WaT = bf.WaT; a.WlT = bf.WlT; a.WuT = bf.WuT; a.enc = enc;
  a.ALPHA = bf.ALPHA; a.CVH = bf.CVH; a.HT = bf.HT; a.C = bf.C; a.rnn_mask = rnn_mask;
  a.G = bf.G; a.DPRE = bf.DPRE; a.DCVH = bf.DCVH; a.DS = bf.DS; a.DQ = bf.DQ; a.DHTOP = bf.DHTOP; a.DC0 = bf.DC0;
  a.PARTB = bf.scratch;
  a.PREC = bf.scratch + (size_t)d->B * nsplit * WH;
  a.PCAR = a.PREC + (size_t)WG_ * 512;
  a.ctr = bf.ctr;
  const size_t nctr = ((size_t)D_N * NSH + 32 + 2 * NTILE + 1) * CTRS;
  a.ab = abort_ctl(bf.ctr + nctr - CTRS, PERSIST_DEC_BWD);
  ASTK_HIP(hipMemsetAsync(bf.ctr, 0, nctr * sizeof(unsigned), s));
  const size_t shm = wide_lds_bytes(chunk);
  static bool attr_done = false;
  if (!attr_done) {
    (void)hipFuncSetAttribute((const void*)decoder_wide_bwd, hipFuncAttributeMaxDynamicSharedMemorySize, 158 * 1024);
    attr_done = true;
  }
  hipLaunchKernelGGL(decoder_wide_bwd, dim3(WG_), dim3(256), shm, s, a);
  ASTK_LAUNCH_CHECK();
  return 0;
}

}  // namespace astk
