// Persistent decoder-loop kernels (seq2seq.py:361-473): ONE launch runs all S = L-1 decoder steps.
//
// Why: a decoder step is ~9 tiny dependent kernels (M = batch rows); at ~4.4 us per launch the loop is launch-floor
// bound and the attention scan can never exceed ~3 TB/s.  Here every phase of a step is a set of fixed "items"
// owned by fixed workgroups for the whole loop:
//   * the weight slice of an item lives in VGPRs as f32-MFMA B fragments (all decoder weights = 14.6 MB = 57 KB per
//     workgroup), loaded once;
//   * each attention item keeps its (batch row, time chunk) slice of enc_states in LDS for all steps, so after the
//     first step the attention scan moves no global memory at all;
//   * phases hand activations over with the same protocol as lstm_persist.hip (write-through stores, drained, one
//     arrival add per item on a per-(phase, batch-tile) counter that lives on its own 256-byte line; consumers poll
//     with one lane, barrier, then sc1 loads).  The decoder has no cross-batch-row dependency, so a consumer only
//     waits for the producers of ITS 16 batch rows.
// Forward phases per step (critical chain P1 -> P2 -> P3 -> P3b -> P4 -> next P1; P5/P6 only gate the next step when
// it is not teacher-forced):
//   P1 embed + LSTM cell (16 rows x 8 units)      P2 q = Wa h + ba        P3 attention partial (b, chunk)
//   P3b attention combine per b (cv, alpha)       P4 ht = tanh(Wc[cv;h]+bc)   P5 logits tile + per-tile CE stats
//   P6 CE combine per batch tile: lse, loss row, argmax (feedback token)
// The saved-state layout is exactly decoder.hip's DecPlan, so either backward works on it.
// Decoder layers 1..NL-1 (the shipped es_en_20h / asr_gpfr models have 3): layer l at step s needs layer l-1 at step s and its own
// step s-1, and layer 0 needs ht_{s-1} (input feeding), so the layers of one step are strictly sequential: one more hand-off
// per layer on the chain.  Every cell item (batch tile, 8 units) of every layer has a fixed owner: layers below the top live on
// the workgroups that own the same item of layer 0, the top layer on the otherwise lightly loaded upper half of the grid
// (ctx / logits owners), so that no workgroup holds more than 68 float4 (272 VGPRs) of weights.
// Applicability (else the per-launch path of decoder.hip runs): 1-3 decoder layers, H,A multiples of 16, sizes within the
// register budgets below, grid of 256 workgroups fully resident.  All spins are bounded (abort word).
#include "common.h"

// -DASTK_PDEC_TIMING_ALL=1: the per-phase timers of ASTK_PERSIST_DBG in the multi-layer variants too (16 more registers per lane there)
// (the timers exist only in the test-hook build, libastk_test.so: the product library's kernels see dbg = 0 as a constant)
#ifdef ASTK_TEST_HOOKS
static int persist_dbg_env() { const char* e = getenv("ASTK_PERSIST_DBG"); return e ? atoi(e) : 0; }
#define PERSIST_DBG(a) ((a).dbg)
#else
static int persist_dbg_env() { return 0; }
#define PERSIST_DBG(a) 0
#endif
#ifndef ASTK_PDEC_TIMING_ALL
#define ASTK_PDEC_TIMING_ALL 0
#endif
namespace astk {

namespace {

constexpr int CTRS = 64;          // counter stride in words (256 B)
constexpr int G = 256;            // workgroups (one per CU)
enum Phase { PH_CELL = 0 /* + layer: 0..2 */, PH_CELL1, PH_CELL2, PH_CMB, PH_CTX, PH_LOG, PH_CE, PH_N };
constexpr int PDEC_MAX_LAYERS = 3;
// Specialised attention phase (H = 512): at most this many rows of a (batch row, time chunk) slice stay resident in LDS (2 x 28 x 2 KB =
// 112 KB of enc + encA); a chunk's remaining rows (chunks of up to 60 rows: T'' <= 480 at batch 32, i.e. the loader's longest
// utterances of 1680 frames) are streamed from L2 / Infinity Cache every step -- they are the same bytes for every step of the loop.
constexpr int PDEC_RES_ROWS = 28, PDEC_CHUNK_MAX = 60;
constexpr int NPHASE_SLOTS = 8;   // counter lines reserved per batch tile ahead of the abort word and the per-row counters
// ... and the hand-off between decoder LAYERS the same way (round 5, ASTK_PDEC_SENT_HD=1): HD[l] -- the dropped output of layer l, one slot per
// step, read by nobody but the cells of layer l + 1 -- sentinel-filled before the launch and polled itself by the waves that multiply it.
// MEASURED AND OFF: es_en_20h (3 layers) 6.79 -> 6.83 ms in a same-box A/B -- 4 waves x 128 workgroups re-reading 8 KB each per poll cost the
// chain more than the drain + counter they replace (the top cell's hand-off to the attention scan polls 2 KB per workgroup, once).
#ifndef ASTK_PDEC_SENT_HD
#define ASTK_PDEC_SENT_HD 0
#endif
#ifndef ASTK_PDEC_SENT_H
#define ASTK_PDEC_SENT_H 1
#endif
constexpr unsigned PDEC_SENTINEL = 0xffffffffu;

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

struct PDecArgs {
  int B, S, L, T, Tp, H, E, A, V, Vp, XI, nbt, nsplit, chunk, ntile_v;
  float inv_count;         // 1 / (rows in the cross-entropy mean): 1 / B, or 1 / (the whole batch) when this launch scores a slice of it
  const float *embed, *Wa, *ba, *Wc, *bc, *Wo, *bo, *cw;
  const float *Wu[PDEC_MAX_LAYERS], *bias[PDEC_MAX_LAYERS], *Wl[PDEC_MAX_LAYERS];     // per decoder layer
  const float* enc;
  const float* encA;       // enc . Wa  (B,T,H): score = encA.h + enc.ba
  const int32_t* y;
  const int32_t* ytgt;     // (B,L) class ids scored by the CE role (column s+1): y, or forward_loss's random_out replacements
  const int32_t* use_truth;
  const float* emb_mask;   // [S][B][E] or null
  const float* rnn_mask[PDEC_MAX_LAYERS];   // [S][B][H] per layer, or null
  int32_t* TOK; int32_t* PRED;
  float *Gt[PDEC_MAX_LAYERS], *Cst[PDEC_MAX_LAYERS], *HR[PDEC_MAX_LAYERS];
  float* HD[PDEC_MAX_LAYERS];                // [S][B][H] dropped outputs of the layers below the top (the top's go to CVH[:, H:])
  float *X0, *Q, *ALPHA, *CVH, *HT, *LOGITS, *LOSSROWS, *LSE;
  float* PART;             // [S][B][nsplit][H+4]
  float* ML;               // [S][B][2] softmax max and 1/sum of every attention row
  float* CESTAT;           // [S][B][ntile_v][4]
  unsigned* ctr;           // [PH_N][nbt] * CTRS
  AbortCtl ab;
  int dbg;
  float* tick_out;         // profiler: [G] accumulated attention-phase microseconds per workgroup, [G+0] launches
};

__device__ __forceinline__ unsigned ld_flag(const unsigned* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void st_sc1(float* p, float v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void sti_sc1(int* p, int v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
// A UNIFORM base pointer, re-formed where it is used (opaque to loop-invariant code motion).  An element address is then base (scalar
// registers) + one 32-bit lane offset shared by all arrays of the same shape, instead of a hoisted 64-bit per-lane pointer per array: in the
// multi-layer kernels (512 registers, the weights resident) ~30 such pointers were spilled, and a scratch reload between two written-through
// stores waits (vmcnt counts in order) for the store in front of it -- a memory round trip per reload, on the decoder's chain.
// ua(base, i): &base[i] with `base` uniform and i this lane's 32-bit element offset -- a scalar base + a 32-bit BYTE offset register, the
// form the global instructions take as it is (a 64-bit element offset would be shifted and added per lane, and kept, and spilled).
template <class T>
__device__ __forceinline__ T* ua(T* base, unsigned i) {
  unsigned long long v = (unsigned long long)base;
  asm volatile("" : "+s"(v));
  // (integer -> GLOBAL pointer: through a generic pointer the accesses become FLAT)
  return (T*)((char __attribute__((address_space(1)))*)v + i * (unsigned)sizeof(T));
}
// Written-through store to base[i], `base` uniform, i this lane's 32-bit element offset: the instruction's scalar-base form, by hand -- for
// an atomic store the compiler adds base and offset per lane into a 64-bit register pair (and keeps the extended offset alive, and spills it).
// (The compiler does not count this store in its s_waitcnt bookkeeping: harmless -- memory operations return in order, an uncounted one can
//  only make a wait longer -- and every publish() drains with an explicit s_waitcnt vmcnt(0).)
__device__ __forceinline__ void st_sc1_u(float* base, unsigned i, float v) {
  asm volatile("global_store_dword %0, %1, %2 sc1" ::"v"(i * 4u), "v"(v), "s"(base) : "memory");
}
__device__ __forceinline__ float ld_sc1(const float* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ int ldi_sc1(const int* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const void* p) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, 0x7fffffff, 0x00020000);
}
__device__ __forceinline__ float4 ldb128_sc1(__amdgpu_buffer_rsrc_t r, long float_off) {
  const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(r, (int)(float_off * 4), 0, 16);
  return make_float4(__uint_as_float(v.x), __uint_as_float(v.y), __uint_as_float(v.z), __uint_as_float(v.w));
}
__device__ __forceinline__ bool wait_ge(const unsigned* ctr, unsigned target, const AbortCtl& ab) {
  unsigned spins = 0;
  while (ld_flag(ctr) < target) {
    if (++spins > ab.limit) {
      abort_raise(ab);
      return false;
    }
    if ((spins & 63u) == 0 && abort_seen(ab)) return false;
  }
  return true;
}
// workgroup-wide wait: lane 0 polls, everyone learns the outcome
__device__ __forceinline__ bool wg_wait(const unsigned* ctr, unsigned target, const AbortCtl& ab, int* s_flag) {
  if (threadIdx.x == 0) *s_flag = wait_ge(ctr, target, ab) ? 1 : 0;
  __syncthreads();
  const bool ok = *s_flag != 0;
  __syncthreads();            // s_flag may be rewritten by the next wait
  return ok;
}
// workgroup-wide wait on `count` (<= 64) counters `stride` words apart: lane i of wave 0 polls counter i
__device__ __forceinline__ bool wg_wait_multi(const unsigned* base, int stride, int count, unsigned target, const AbortCtl& ab, int* s_flag) {
  if (threadIdx.x < 64) {
    const int lane = threadIdx.x;
    bool ok = true;
    unsigned spins = 0;
    for (;;) {
      const bool mine = lane < count ? ld_flag(base + (long)lane * stride) >= target : true;
      if (__all(mine)) break;
      if (++spins > ab.limit) { abort_raise(ab); ok = false; break; }
      if ((spins & 63u) == 0 && abort_seen(ab)) { ok = false; break; }
    }
    if (lane == 0) *s_flag = ok ? 1 : 0;
  }
  __syncthreads();
  const bool ok = *s_flag != 0;
  __syncthreads();
  return ok;
}
// Sharded phase counters: the items of a (phase, batch tile) bump one of NSH words (item % NSH), each on its own 256-byte line, and a
// waiter polls the NSH words with NSH lanes of one wave.  Same-address atomics retire one after the other (~12 ns each in isolation,
// far more under load): with 32-128 arrivals per hand-off on ONE word the decoder kernels ran 1.00 / 1.00 ms; 4 / 8 / 16 / 32 / 64 words:
// 0.84/0.82, 0.80/0.78, 0.79/0.75, 0.78/0.73, 0.78/0.75 ms (forward / backward).
constexpr int NSH = 32;
__device__ __forceinline__ bool wg_wait_sh(const unsigned* base, int n_items, int steps, const AbortCtl& ab, int* s_flag) {
  if (threadIdx.x < 64) {
    const int lane = threadIdx.x;
    const unsigned target = lane < NSH ? (unsigned)(((n_items - lane + NSH - 1) / NSH) * steps) : 0u;   // items with idx % NSH == lane
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
  return ok;
}
__device__ __forceinline__ void publish(unsigned* ctr) {
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (threadIdx.x == 0) __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void publish_sh(unsigned* base, int item) { publish(base + (item & (NSH - 1)) * CTRS); }
__device__ __forceinline__ float sigm(float x) { return 1.f / (1.f + expf(-x)); }
// gate / tanh activations on the decoder chain: v_exp_f32 / v_rcp_f32 based (absolute error <= ~2e-7), as in lstm_persist.hip
__device__ __forceinline__ float sigm_fast(float x) { return __builtin_amdgcn_rcpf(1.f + __expf(-x)); }
__device__ __forceinline__ float tanh_fast(float x) { return 2.f * __builtin_amdgcn_rcpf(1.f + __expf(-2.f * x)) - 1.f; }
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  return v;
}

// Resident weight fragments (NB k-blocks of 16 floats per wave): MFMA column j = lane&15 -> W row `row`.
template <int NB>
__device__ __forceinline__ void wload(float4* w, const float* W, long ldw, int row, int K, int lane, int wave) {
  const int q = lane >> 4;
#pragma unroll
  for (int i = 0; i < NB; ++i) {
    const int k = 16 * (wave + 4 * i) + 4 * q;
    w[i] = k < K ? *reinterpret_cast<const float4*>(W + (long)row * ldw + k) : make_float4(0.f, 0.f, 0.f, 0.f);
  }
}
// A fragments of a handed-off row block: 16-byte loads (AUX = 16: sc1) at float offset a_off + k, k = 16 (wave + 4 (i0 + i)) + 4 q.
// Blocks that all lie inside K (the shipped widths): ONE address register and immediate offsets.  Otherwise k is clamped (the matching w
// is zero beyond K) -- one address register per load: in the multi-layer kernels those were spilled, and a scratch reload between two
// buffer loads waits for every load issued before it (vmcnt counts in order): the 16 loads of a product came back in 8 round trips.
template <int NB, int AUX = 16>
__device__ __forceinline__ void aload_sc1(float4* a, __amdgpu_buffer_rsrc_t ra, long a_off, int K, int lane, int wave, int i0 = 0) {
  const int kb = 16 * (wave + 4 * i0) + 4 * (lane >> 4);
  if (64 * (i0 + NB) <= K) {
    const int vo = (int)((a_off + kb) * 4);
#pragma unroll
    for (int i = 0; i < NB; ++i) {
      const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(ra, vo + 256 * i, 0, AUX);
      a[i] = make_float4(__uint_as_float(v.x), __uint_as_float(v.y), __uint_as_float(v.z), __uint_as_float(v.w));
    }
  } else {
    int kbv = kb;
    asm volatile("" : "+v"(kbv));      // (opaque: the NB clamped offsets are formed here, not hoisted out of the step loop as NB live registers)
#pragma unroll
    for (int i = 0; i < NB; ++i) {
      const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(ra, (int)((a_off + min(kbv + 64 * i, K - 4)) * 4), 0, AUX);
      a[i] = make_float4(__uint_as_float(v.x), __uint_as_float(v.y), __uint_as_float(v.z), __uint_as_float(v.w));
    }
  }
}
// acc += A[arow][0:K] . W^T : A read with sc1 loads through `ra` at float offset a_off (+k)
template <int NB>
__device__ __forceinline__ void wmac(f32x4& acc, const float4* w, __amdgpu_buffer_rsrc_t ra, long a_off, int K, int lane, int wave) {
  float4 a[NB];
  aload_sc1<NB>(a, ra, a_off, K, lane, wave);
  __builtin_amdgcn_sched_barrier(0);
  f32x4 acc2 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int i = 0; i < NB; ++i) {
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i].x, w[i].x, acc, 0, 0, 0);
    acc2 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i].y, w[i].y, acc2, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i].z, w[i].z, acc, 0, 0, 0);
    acc2 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i].w, w[i].w, acc2, 0, 0, 0);
  }
  acc += acc2;
}

// 4-wave reduction of one 16x16 accumulator: returns element (row = tid>>4, col = tid&15)
__device__ __forceinline__ float reduce16(f32x4 acc, float* red /* [4][256] */) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  *reinterpret_cast<f32x4*>(&red[(wave * 64 + lane) * 4]) = acc;
  __syncthreads();
  const int row = tid >> 4, col = tid & 15;
  const int src = ((row >> 2) * 16 + col) * 4 + (row & 3);
  const float v = red[src] + red[256 + src] + red[512 + src] + red[768 + src];
  __syncthreads();
  return v;
}

// Register budgets (k-blocks of 16 per wave): cell emb-part E <= 128, ht-part A <= 512, h-part H <= 512; ctx 2H <= 1024;
// logits A <= 512.  A workgroup is either a cell owner or a ctx/logits owner, so both roles share one register array:
// cell = 2 tiles x (we[2] + wa[8] + wh[8]) = 36 float4 ; other = wc[16] + wl[2][8] = 32 float4.
constexpr int NB_E = 2, NB_A = 8, NB_H = 8, NB_C = 16, NB_L = 8;
constexpr int CELLW = NB_E + NB_A + NB_H;            // per gate tile
constexpr int OFF_WC = 0, OFF_WL = NB_C, NWREG = 2 * CELLW;
static_assert(OFF_WL + 2 * NB_L <= NWREG, "register budget");
// NL > 1: two cell slots of 2 tiles x (8 + 8) k-blocks = 32 float4 each: [0, 32) layer 0 ([ht | lateral]; its embedding columns are
// re-read every step, off the chain) or ctx / logits, [32, 64) a layer >= 1 ([upward | lateral]).  64 float4 = 256 registers = the
// whole AGPR file.  (Streaming the lateral weights of the second slot from L2 instead -- 48 float4 resident -- halved the spills of
// the specialised attention phase but cost 4 us per decoder step: the reads compete with the chain's hand-offs.  Not kept.)
constexpr int CELLW2 = 2 * NB_H, OFF_C2 = 2 * CELLW2, NWREG_ML = 4 * CELLW2;
static_assert(OFF_WL + 2 * NB_L <= OFF_C2, "register budget");

template <int NB>
__device__ __forceinline__ void mfma_blocks(f32x4& acc, const float4* a, const float4* w) {
  f32x4 acc2 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int i = 0; i < NB; ++i) {
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i].x, w[i].x, acc, 0, 0, 0);
    acc2 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i].y, w[i].y, acc2, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i].z, w[i].z, acc, 0, 0, 0);
    acc2 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i].w, w[i].w, acc2, 0, 0, 0);
  }
  acc += acc2;
}

// NC > 0: H = 64 * NC and chunk <= PDEC_CHUNK_MAX are compile-time facts for the attention phase (fully unrolled, batched reads);
// NC = 0: generic loops.
// XS: the slice has more rows than stay resident (chunk > PDEC_RES_ROWS): the streamed-row code is compiled into its own instantiation
// so that the common short-chunk case keeps its register budget.
template <int NC, int NL, bool XS>
__global__ __launch_bounds__(256, 1) void decoder_persist_fwd(PDecArgs a) {
  extern __shared__ __attribute__((aligned(16))) float lds[];   // enc slice [chunk][H], encA slice [chunk][H], scratch
  __shared__ __attribute__((aligned(16))) float red[4 * 256];
  __shared__ __attribute__((aligned(16))) float zt[2 * 256];
  __shared__ int s_flag;
  __shared__ int yS[16 * 192];         // targets of this workgroup's batch tile (L <= 192)
  __shared__ int flagS[192];           // teacher-forcing flags
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wg = blockIdx.x;
  const int B = a.B, S = a.S, H = a.H, E = a.E, A = a.A, V = a.V, XI = a.XI, T = a.T, Tp = a.Tp;
  const int nbt = a.nbt;
  unsigned* ctr = a.ctr;
#define CTR(ph, bt) (ctr + ((ph) * nbt + (bt)) * NSH * CTRS)
#define ROWCTR(row) (ctr + ((long)NPHASE_SLOTS * NSH * nbt + 2 + (row)) * CTRS)

  // ---------------- static item ownership
  const int n_cell = nbt * (H / 8);                       // cell items (bt, 8 units): workgroups [0, n_cell)
  const bool has_cell = wg < n_cell;
  const int cell_bt = has_cell ? wg / (H / 8) : 0, cell_u0 = has_cell ? (wg % (H / 8)) * 8 : 0;
  const int n_c = nbt * (A / 16);                         // ctx items: first non-cell workgroups
  const int rest = G - n_cell;
  const int r_idx = wg - n_cell;
  const bool has_c = r_idx >= 0 && r_idx < n_c;
  const int c_bt = has_c ? r_idx / (A / 16) : 0, c_n0 = has_c ? (r_idx % (A / 16)) * 16 : 0;
  const int n_l = nbt * a.ntile_v;                        // logits items: up to 2 per non-cell workgroup
  int l_item[2] = {-1, -1};
  if (r_idx >= 0) {
    if (r_idx < n_l) l_item[0] = r_idx;
    if (r_idx + rest < n_l) l_item[1] = r_idx + rest;
  }
  const int n_att = B * a.nsplit;                         // attention items (b, split): one per workgroup
  const bool has_att = wg < n_att;
  const int att_b = has_att ? wg % B : 0, att_sp = has_att ? wg / B : 0;
  const int cmb_rank = wg - (G - B);                      // combine items (b): the last B workgroups
  const bool has_cmb = cmb_rank >= 0 && cmb_rank < B;
  const int cmb_b = has_cmb ? cmb_rank : 0;
  const int ce_rank = wg - (G - B - nbt);                 // CE items (bt)
  const bool has_ce = ce_rank >= 0 && ce_rank < nbt;

  // upper decoder layers (NL > 1): layers 1..NL-2 on the owner of the same layer-0 item, the top layer on the upper workgroups
  constexpr int TOP = NL - 1;
  const bool has_mid = NL > 2 && has_cell;
  const int top_rank = wg - (G - n_cell);
  const bool has_top = NL > 1 && top_rank >= 0;
  const int top_bt = has_top ? top_rank / (H / 8) : 0, top_u0 = has_top ? (top_rank % (H / 8)) * 8 : 0;

  // ---------------- resident weights
  constexpr int NW = NL > 1 ? NWREG_ML : NWREG;
  float4 wreg[NW];
#pragma unroll
  for (int i = 0; i < NW; ++i) wreg[i] = make_float4(0.f, 0.f, 0.f, 0.f);
  const int r16 = lane & 15;
  if constexpr (NL > 1) {
    // second cell slot: [tile][upward 8 | lateral 8] of layer 1 (mid, NL == 3) or of the top layer
    const int l2 = has_mid ? 1 : TOP;
    if (has_mid || has_top) {
      const int u0 = has_mid ? cell_u0 : top_u0;
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        const int row = 4 * u0 + 16 * t + r16;
        wload<NB_H>(wreg + OFF_C2 + t * CELLW2, a.Wu[l2] + (long)row * H, 0, 0, H, lane, wave);
        wload<NB_H>(wreg + OFF_C2 + t * CELLW2 + NB_H, a.Wl[l2] + (long)row * H, 0, 0, H, lane, wave);
      }
    }
  }
  // layer-0 slot: [emb | ht | lateral] per tile with one layer, [ht | lateral] with more (the embedding columns are streamed)
  constexpr bool EMB_RES = NL == 1;
  constexpr int CW0 = EMB_RES ? CELLW : CELLW2, W0_A = EMB_RES ? NB_E : 0, W0_H = W0_A + NB_A;
  if (has_cell) {   // two 16-row gate tiles (units u0..u0+3, u0+4..u0+7): [emb cols | ht cols] of Wu, then Wl
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      const int row = 4 * cell_u0 + 16 * t + r16;
      if constexpr (EMB_RES) wload<NB_E>(wreg + t * CW0, a.Wu[0] + (long)row * XI, 0, 0, E, lane, wave);
      wload<NB_A>(wreg + t * CW0 + W0_A, a.Wu[0] + (long)row * XI + E, 0, 0, A, lane, wave);
      wload<NB_H>(wreg + t * CW0 + W0_H, a.Wl[0] + (long)row * H, 0, 0, H, lane, wave);
    }
  } else {
    if (has_c) wload<NB_C>(wreg + OFF_WC, a.Wc, 2 * H, c_n0 + r16, 2 * H, lane, wave);
#pragma unroll
    for (int t = 0; t < 2; ++t)
      if (l_item[t] >= 0) {
        const int n0 = (l_item[t] % a.ntile_v) * 16;
        wload<NB_L>(wreg + OFF_WL + t * NB_L, a.Wo, A, min(n0 + r16, V - 1), A, lane, wave);
      }
  }
  if (has_cell) {
    for (int i = tid; i < 16 * a.L; i += 256) yS[i] = a.y[(long)min(cell_bt * 16 + i / a.L, B - 1) * a.L + (i % a.L)];
    for (int i = tid; i < S; i += 256) flagS[i] = a.use_truth[i];
  } else if (has_ce) {
    for (int i = tid; i < S; i += 256) flagS[i] = a.use_truth[i];     // (which steps feed their argmax back: decides when P6 may be deferred)
  }
  // ---------------- resident slices of enc_states and encA = enc Wa in LDS: rows [t0, t1) of batch row att_b
  const int t0 = att_sp * a.chunk, t1 = min(T, t0 + a.chunk);
  const int nrow = has_att ? t1 - t0 : 0;
  const int cres = NC > 0 ? min(a.chunk, PDEC_RES_ROWS) : a.chunk;      // rows of the slice kept in LDS
  const int nres = min(nrow, cres);
  float* encS = lds;
  float* encAS = lds + cres * H;
  float* ebS = lds + 2 * cres * H;               // [chunk] enc.ba, computed once
  float* scr = ebS + ((a.chunk + 3) & ~3);       // per-step scratch: hS[H] | score[chunk] | p[chunk]  -- or the combine's partial rows
  const float* gEnc = a.enc + ((long)att_b * T + t0) * H;       // this slice in global memory (rows >= nres are read from here every step)
  const float* gEncA = a.encA + ((long)att_b * T + t0) * H;
  if (has_att) {
    const int n4 = nres * H / 4;
    for (int i = tid; i < n4; i += 256) {
      reinterpret_cast<float4*>(encS)[i] = reinterpret_cast<const float4*>(gEnc)[i];
      reinterpret_cast<float4*>(encAS)[i] = reinterpret_cast<const float4*>(gEncA)[i];
    }
    __syncthreads();
    for (int t = tid; t < nrow; t += 256) {       // eb[t] = enc[t,:] . ba  (score = encA.h + eb)
      float d = 0.f;
      for (int k = 0; k < H; ++k) d += gEnc[(long)t * H + k] * a.ba[k];
      ebS[t] = d;
    }
  }
  __syncthreads();

  const __amdgpu_buffer_rsrc_t r_x0 = make_rsrc(a.X0), r_hr = make_rsrc(a.HR[0]), r_cvh = make_rsrc(a.CVH), r_ht = make_rsrc(a.HT);
  bool att_dead = false;     // this wave gave up polling a sentinel hand-off (abort / time-out): it goes on without waiting
  const __amdgpu_buffer_rsrc_t r_part = make_rsrc(a.PART), r_ces = make_rsrc(a.CESTAT);
  // cell epilogue ownership: threads 0..127: tile = tid>>6, (row = (tid>>2)&15, unit = tid&3)
  const int ce_tile = tid >> 6, ce_row = (tid >> 2) & 15, ce_u = tid & 3;
  const int cell_b = cell_bt * 16 + ce_row, cell_u = cell_u0 + 4 * ce_tile + ce_u;
  float c_state = 0.f;
  if (has_cell && tid < 128 && cell_b < B) c_state = a.Cst[0][(long)cell_b * H + cell_u];   // C[0] = c0
  float4 cbias = make_float4(0.f, 0.f, 0.f, 0.f);
  if (has_cell && tid < 128) cbias = *reinterpret_cast<const float4*>(a.bias[0] + 4 * cell_u);
  // second cell slot: state and bias of this thread's (row, unit) of layer l2
  const int l2 = has_mid ? 1 : TOP;
  const int c2_bt = has_mid ? cell_bt : top_bt, c2_u0 = has_mid ? cell_u0 : top_u0;
  const int c2_b = c2_bt * 16 + ce_row, c2_u = c2_u0 + 4 * ce_tile + ce_u;
  // this thread's (row, unit) element of a [.][B][H] array, as ONE 32-bit offset per cell slot (the arrays' uniform bases: ua(), st_sc1_u())
  const unsigned cell_off = (unsigned)(cell_b * H + cell_u), c2_off = (unsigned)(c2_b * H + c2_u);
  const bool has_c2 = NL > 1 && (has_mid || has_top);
  float c_state2 = 0.f;
  float4 cbias2 = make_float4(0.f, 0.f, 0.f, 0.f);
  if (has_c2 && tid < 128) {
    if (c2_b < B) c_state2 = a.Cst[l2][(long)c2_b * H + c2_u];
    cbias2 = *reinterpret_cast<const float4*>(a.bias[l2] + 4 * c2_u);
  }

  long long tk[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) tk[i] = 0;
  // per-phase debug timers (ASTK_PERSIST_DBG): 50 registers when live -- compiled in only where they fit without spills (one layer)
  const bool timing = (NL == 1 || ASTK_PDEC_TIMING_ALL) && PERSIST_DBG(a) != 0;
  long long tlast = timing ? wall_clock64() : 0;
  long long tk_att = 0;
  long long tq[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#define TQ(i) if (timing) { const long long now_ = wall_clock64(); tq[i] += now_ - tlast; tlast = now_; }
#define TICK(i) if (timing) { const long long now_ = wall_clock64(); tk[i] += now_ - tlast; tlast = now_; }
  int p6_pending = -1;
  auto run_p6 = [&](const int s) -> bool {
      const int bt = ce_rank, m0 = bt * 16;
      if (!wg_wait_sh(CTR(PH_LOG, bt), a.ntile_v, s + 1, a.ab, &s_flag)) return false;
      const int row = m0 + (tid >> 4), sub = tid & 15;       // 16 threads per row sweep the tiles
      float mx = -INFINITY, se = 0.f, xt = 0.f;
      int mi = 0x7fffffff;
      if (row < B)
        for (int k = sub; k < a.ntile_v; k += 16) {
          const float4 cs = ldb128_sc1(r_ces, (((long)s * B + row) * a.ntile_v + k) * 4);
          const float tm = cs.x, ts = cs.y, tx = cs.w;
          const int ti = __float_as_int(cs.z);
          const float nm = fmaxf(mx, tm);
          se = se * expf(mx - nm) + ts * expf(tm - nm);
          if (tm > mx || (tm == mx && ti < mi)) mi = ti;
          mx = nm;
          xt += tx;
        }
#pragma unroll
      for (int o = 8; o > 0; o >>= 1) {
        const float om = __shfl_xor(mx, o), os = __shfl_xor(se, o), ox = __shfl_xor(xt, o);
        const int oi = __shfl_xor(mi, o);
        const float nm = fmaxf(mx, om);
        se = (mx == -INFINITY ? 0.f : se * expf(mx - nm)) + (om == -INFINITY ? 0.f : os * expf(om - nm));
        if (om > mx || (om == mx && oi < mi)) mi = oi;
        mx = nm;
        xt += ox;
      }
      if (sub == 0 && row < B) {
        const float lse = mx + logf(se);
        const int tgt = *ua(a.ytgt + s + 1, (unsigned)(row * a.L));
        const float w = a.cw ? a.cw[tgt < 0 ? 0 : (tgt >= V ? V - 1 : tgt)] : 1.f;
        *ua(a.LSE + (long)s * B, (unsigned)row) = lse;
        *ua(a.LOSSROWS + (long)s * B, (unsigned)row) = -(xt - lse) * w * a.inv_count;
        sti_sc1(ua(a.PRED + (long)s * B, (unsigned)row), mi);
      }
      publish_sh(CTR(PH_CE, bt), 0);
      TICK(12)
    return true;
  };
  for (int s = 0; s < S; ++s) {
    // ================= P1: embed + LSTM cell =================
    if (has_cell) {
      const int bt = cell_bt, m0 = bt * 16;
      const int brow = min(m0 + r16, B - 1);             // this lane's A-operand batch row
      const bool truth = s == 0 || flagS[s] != 0;
      if (!truth) { if (!wg_wait_sh(CTR(PH_CE, bt), 1, s, a.ab, &s_flag)) return; }
      int tok = truth ? yS[r16 * a.L + s] : ldi_sc1(ua(a.PRED + (long)(s - 1) * B, (unsigned)brow));
      tok = tok < 0 ? 0 : (tok >= V ? V - 1 : tok);
      if (s > 0) { if (!wg_wait_sh(CTR(PH_CELL, bt), H / 8, s, a.ab, &s_flag)) return; }
      f32x4 acc[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
      {
        // (a) embedding part and (b) recurrent part: neither depends on this step's ht, so they run before the wait on P4
        const int q = lane >> 4;
        float4 ae[NB_E], ah[NB_H];
#pragma unroll
        for (int i = 0; i < NB_E; ++i) {
          const int k = min(16 * (wave + 4 * i) + 4 * q, E - 4);
          float4 v = *reinterpret_cast<const float4*>(a.embed + (long)tok * E + k);
          if (a.emb_mask) {
            const float4 mk = *reinterpret_cast<const float4*>(a.emb_mask + ((long)s * B + brow) * E + k);
            v.x *= mk.x; v.y *= mk.y; v.z *= mk.z; v.w *= mk.w;
          }
          ae[i] = v;
        }
        aload_sc1<NB_H>(ah, r_hr, ((long)s * B + brow) * H, H, lane, wave);     // h_{s-1}: published a whole step ago (waited above)
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int t = 0; t < 2; ++t) {
          if constexpr (EMB_RES) mfma_blocks<NB_E>(acc[t], ae, wreg + t * CW0);
          else {
            float4 we[NB_E];
            wload<NB_E>(we, a.Wu[0] + (long)(4 * cell_u0 + 16 * t + r16) * XI, 0, 0, E, lane, wave);
            mfma_blocks<NB_E>(acc[t], ae, we);
          }
          mfma_blocks<NB_H>(acc[t], ah, wreg + t * CW0 + W0_H);
        }
      }
      TICK(0)
      // (c) input-feeding part: ht_{s-1}, written into X0[s][:, E:] by P4 of step s-1
      if (s > 0) {
        if (!wg_wait_sh(CTR(PH_CTX, bt), A / 16, s, a.ab, &s_flag)) return;
        TICK(1)
        float4 at[NB_A];
        aload_sc1<NB_A>(at, r_x0, ((long)s * B + brow) * XI + E, A, lane, wave);
        __builtin_amdgcn_sched_barrier(0);
        mfma_blocks<NB_A>(acc[0], at, wreg + W0_A);
        mfma_blocks<NB_A>(acc[1], at, wreg + CW0 + W0_A);
      }
      const float z0 = reduce16(acc[0], red);
      const float z1 = reduce16(acc[1], red);
      zt[tid] = z0;
      zt[256 + tid] = z1;
      __syncthreads();
      float4 gsave = make_float4(0.f, 0.f, 0.f, 0.f);
      const bool ev = tid < 128 && cell_b < B;
      if (ev) {
        const float4 z = *reinterpret_cast<const float4*>(&zt[ce_tile * 256 + ce_row * 16 + ce_u * 4]);
        const float ga = tanh_fast(z.x + cbias.x), gi = sigm_fast(z.y + cbias.y), gf = sigm_fast(z.z + cbias.z), go = sigm_fast(z.w + cbias.w);
        c_state = ga * gi + gf * c_state;
        const float hh = go * tanh_fast(c_state);
        const float hd = a.rnn_mask[0] ? hh * *ua(a.rnn_mask[0] + (long)s * B * H, cell_off) : hh;
        gsave = make_float4(ga, gi, gf, go);
        st_sc1_u(a.HR[0] + (long)(s + 1) * B * H, cell_off, hh);
        if constexpr (NL > 1) st_sc1_u(a.HD[0] + (long)s * B * H, cell_off, hd);     // input of layer 1
        else st_sc1_u(a.CVH + (long)s * B * 2 * H + H, cell_off + cell_b * H, hd);
      }
      TICK(2)
      publish_sh(CTR(PH_CELL, bt), cell_u0 / 8);
      TICK(3)
      if (ev) {                                    // saved for the backward (plain stores, off the critical path)
        *ua(reinterpret_cast<float4*>(a.Gt[0] + (long)s * B * 4 * H), cell_off) = gsave;
        *ua(a.Cst[0] + (long)(s + 1) * B * H, cell_off) = c_state;
      }
    }
    // ================= P1b: decoder layers 1..NL-1 (one cell item per workgroup: layer 1 of a 3-layer stack on the lower
    // workgroups, the top layer on the upper ones).  z = Wu hd_{l-1,s} + Wl h_{l,s-1} + b: the recurrent half runs before the wait.
    if constexpr (NL > 1) {
#pragma unroll
      for (int pass = 0; pass < 2; ++pass) {
        // pass 0: layer 1 as a MID layer (NL == 3 only); pass 1: the TOP layer
        const bool mine = pass == 0 ? has_mid : has_top;
        if ((pass == 0 && NL < 3) || !mine) continue;
        const int l = pass == 0 ? 1 : TOP;
        const int bt = c2_bt, m0 = bt * 16;
        const int brow = min(m0 + r16, B - 1);
        if (s > 0) { if (!wg_wait_sh(CTR(PH_CELL + l, bt), H / 8, s, a.ab, &s_flag)) return; }
        f32x4 acc[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
        {
          float4 ah[NB_H];
          aload_sc1<NB_H>(ah, make_rsrc(a.HR[l]), ((long)s * B + brow) * H, H, lane, wave);     // h_{l,s-1} (h0 at s = 0)
          __builtin_amdgcn_sched_barrier(0);
          mfma_blocks<NB_H>(acc[0], ah, wreg + OFF_C2 + NB_H);
          mfma_blocks<NB_H>(acc[1], ah, wreg + OFF_C2 + CELLW2 + NB_H);
        }
#if !(ASTK_PDEC_SENT_HD && ASTK_PDEC_SENT_H)
        if (!wg_wait_sh(CTR(PH_CELL + l - 1, bt), H / 8, s + 1, a.ab, &s_flag)) return;
#endif
        {
          float4 ax[NB_H];
#if ASTK_PDEC_SENT_HD && ASTK_PDEC_SENT_H
          // the data is the flag: every wave re-reads ITS fragments of the 16 rows until none of them holds the sentinel -- no drain,
          // counter, counter poll or barrier between the lower cell's stores and this product
          {
            const __amdgpu_buffer_rsrc_t r_hd = make_rsrc(a.HD[l - 1]);
            unsigned spins = 0;
            for (;;) {
              aload_sc1<NB_H>(ax, r_hd, ((long)s * B + brow) * H, H, lane, wave);
              bool ok = true;
#pragma unroll
              for (int i = 0; i < NB_H; ++i)
                ok = ok & (__float_as_uint(ax[i].x) != PDEC_SENTINEL) & (__float_as_uint(ax[i].y) != PDEC_SENTINEL) &
                     (__float_as_uint(ax[i].z) != PDEC_SENTINEL) & (__float_as_uint(ax[i].w) != PDEC_SENTINEL);
              if (__all(ok) || att_dead) break;
              if (++spins > (a.ab.limit >> 1)) { abort_raise(a.ab); att_dead = true; }
              else if ((spins & 63u) == 0 && abort_seen(a.ab)) att_dead = true;
            }
          }
#else
          aload_sc1<NB_H>(ax, make_rsrc(a.HD[l - 1]), ((long)s * B + brow) * H, H, lane, wave);  // dropped output of the layer below
#endif
          __builtin_amdgcn_sched_barrier(0);
          mfma_blocks<NB_H>(acc[0], ax, wreg + OFF_C2);
          mfma_blocks<NB_H>(acc[1], ax, wreg + OFF_C2 + CELLW2);
        }
        const float z0 = reduce16(acc[0], red);
        const float z1 = reduce16(acc[1], red);
        zt[tid] = z0;
        zt[256 + tid] = z1;
        __syncthreads();
        float4 gsave = make_float4(0.f, 0.f, 0.f, 0.f);
        const bool ev = tid < 128 && c2_b < B;
        if (ev) {
          const float4 z = *reinterpret_cast<const float4*>(&zt[ce_tile * 256 + ce_row * 16 + ce_u * 4]);
          const float ga = tanh_fast(z.x + cbias2.x), gi = sigm_fast(z.y + cbias2.y), gf = sigm_fast(z.z + cbias2.z), go = sigm_fast(z.w + cbias2.w);
          c_state2 = ga * gi + gf * c_state2;
          const float hh = go * tanh_fast(c_state2);
          const float hd = a.rnn_mask[l] ? hh * *ua(a.rnn_mask[l] + (long)s * B * H, c2_off) : hh;
          gsave = make_float4(ga, gi, gf, go);
          st_sc1_u(a.HR[l] + (long)(s + 1) * B * H, c2_off, hh);
          if (l == TOP) st_sc1_u(a.CVH + (long)s * B * 2 * H + H, c2_off + c2_b * H, hd);
          else st_sc1_u(a.HD[l] + (long)s * B * H, c2_off, hd);
        }
        publish_sh(CTR(PH_CELL + l, bt), c2_u0 / 8);
        if (ev) {
          *ua(reinterpret_cast<float4*>(a.Gt[l] + (long)s * B * 4 * H), c2_off) = gsave;
          *ua(a.Cst[l] + (long)(s + 1) * B * H, c2_off) = c_state2;
        }
      }
    }
    // ================= P3: attention over the LDS-resident slices: score = encA.h + eb, p = exp(score - max), cv partial =========
    if (has_att) {
      const int b = att_b, bt = b / 16;
      TICK(15)
      const int c4 = (a.chunk + 3) & ~3;
      float* hS = scr;                 // [H]
      float* scS = scr + H;            // [c4] raw scores (tail padded with -inf)
      float* pS = scr + H + c4;        // [c4] exp(score - m) (tail 0)
#if ASTK_PDEC_SENT_H
      // The data is the flag (lstm_persist.hip): the h half of CVH is sentinel-filled before the launch and this row's 2 KB are polled
      // themselves -- no drain, counter, counter poll or barrier between the top cell's stores and the scan.
      if (tid < H / 4) {
        const int off = (int)((((long)s * B + b) * 2 * H + H + 4 * tid) * 4);
        u32x4 v;
        unsigned spins = 0;
        for (;;) {
          v = __builtin_amdgcn_raw_buffer_load_b128(r_cvh, off, 0, 16);
          if (__all((v.x != PDEC_SENTINEL) & (v.y != PDEC_SENTINEL) & (v.z != PDEC_SENTINEL) & (v.w != PDEC_SENTINEL)) || att_dead) break;
          if (++spins > (a.ab.limit >> 1)) { abort_raise(a.ab); att_dead = true; }
          else if ((spins & 63u) == 0 && abort_seen(a.ab)) att_dead = true;
        }
        *reinterpret_cast<float4*>(hS + 4 * tid) = make_float4(__uint_as_float(v.x), __uint_as_float(v.y), __uint_as_float(v.z), __uint_as_float(v.w));
      }
      TICK(4)
      const long long ta0 = a.tick_out ? wall_clock64() : 0;
#else
      if (!wg_wait_sh(CTR(PH_CELL + TOP, bt), H / 8, s + 1, a.ab, &s_flag)) return;
      TICK(4)
      const long long ta0 = a.tick_out ? wall_clock64() : 0;
      if (tid < H / 4) *reinterpret_cast<float4*>(hS + 4 * tid) = ldb128_sc1(r_cvh, ((long)s * B + b) * 2 * H + H + 4 * tid);
#endif
      float m, l = 0.f, my_score;
      float* prow = a.PART + (((long)s * B + b) * a.nsplit + att_sp) * (H + 4);
      if constexpr (NC > 0) {
        // ---- specialised scan: H = 64 NC, nrow <= 60: rows [0, nres) from LDS, rows [nres, nrow) (nx <= 32 of them) from global memory.
        // Every read of a pass is issued before its first use.
        constexpr int HH = 64 * NC;
        const int nx = XS ? nrow - nres : 0;
        float* const pS_ = scr + HH + 64;                // scratch: hS[H] | scores[64] | p[64] | fold
        if (tid >= nrow && tid < 64) scS[tid] = -INFINITY;
        __syncthreads();
        TICK(13)
        {
          // pass 1: 16 lanes per row; group g owns rows g and g + 16; lane l covers floats 4l + 64c of the row
          const int grp = tid >> 4, l16 = tid & 15;
          const int ta = min(grp, nres - 1), tb2 = min(grp + 16, nres - 1);
          const float* e1 = encAS + ta * HH + 4 * l16;
          const float* e2 = encAS + tb2 * HH + 4 * l16;
          float4 hv[NC], x1[NC], x2[NC];
#pragma unroll
          for (int c = 0; c < NC; ++c) {
            hv[c] = *reinterpret_cast<const float4*>(hS + 64 * c + 4 * l16);
            x1[c] = *reinterpret_cast<const float4*>(e1 + 64 * c);
            x2[c] = *reinterpret_cast<const float4*>(e2 + 64 * c);
          }
          float d1 = 0.f, d2 = 0.f;
#pragma unroll
          for (int c = 0; c < NC; ++c) {
            d1 += x1[c].x * hv[c].x + x1[c].y * hv[c].y + x1[c].z * hv[c].z + x1[c].w * hv[c].w;
            d2 += x2[c].x * hv[c].x + x2[c].y * hv[c].y + x2[c].z * hv[c].z + x2[c].w * hv[c].w;
          }
#pragma unroll
          for (int o = 8; o > 0; o >>= 1) { d1 += __shfl_xor(d1, o); d2 += __shfl_xor(d2, o); }
          if (l16 == 0) {
            if (grp < nres) scS[grp] = d1 + ebS[grp];
            if (grp + 16 < nres) scS[grp + 16] = d2 + ebS[grp + 16];
          }
          if (nx > 0) {        // streamed rows nres + g, nres + g + 16 (workgroup-uniform branch)
            const int ua = nres + min(grp, nx - 1), ub = nres + min(grp + 16, nx - 1);
            const float* g1 = gEncA + (long)ua * HH + 4 * l16;
            const float* g2 = gEncA + (long)ub * HH + 4 * l16;
#pragma unroll
            for (int c = 0; c < NC; ++c) {
              x1[c] = *reinterpret_cast<const float4*>(g1 + 64 * c);
              x2[c] = *reinterpret_cast<const float4*>(g2 + 64 * c);
            }
            d1 = 0.f; d2 = 0.f;
#pragma unroll
            for (int c = 0; c < NC; ++c) {
              d1 += x1[c].x * hv[c].x + x1[c].y * hv[c].y + x1[c].z * hv[c].z + x1[c].w * hv[c].w;
              d2 += x2[c].x * hv[c].x + x2[c].y * hv[c].y + x2[c].z * hv[c].z + x2[c].w * hv[c].w;
            }
#pragma unroll
            for (int o = 8; o > 0; o >>= 1) { d1 += __shfl_xor(d1, o); d2 += __shfl_xor(d2, o); }
            if (l16 == 0) {
              if (grp < nx) scS[nres + grp] = d1 + ebS[nres + grp];
              if (grp + 16 < nx) scS[nres + grp + 16] = d2 + ebS[nres + grp + 16];
            }
          }
        }
        TQ(0)
        __syncthreads();
        TQ(1)
        {
          // chunk max / exp / sum: wave 0 holds one (padded) score per lane -- two shuffle reductions instead of 64-entry sweeps in
          // every thread (32 float4 registers that the multi-layer variants, whose weights fill the AGPR file, do not have)
          my_score = 0.f;
          m = 0.f;
          if (tid < 64) {
            const float sc = scS[tid];
            float mx = sc;
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o));
            const float pe = tid < nrow ? expf(sc - mx) : 0.f;
            pS_[tid] = pe;
            l = wave_sum(pe);
            m = mx;
            my_score = tid < nrow ? sc : 0.f;
          }
        }
        TQ(2)
        __syncthreads();
        TQ(3)
        {
          // pass 2: context partial.  Thread (cq, rh): columns 4cq..4cq+3, rows rh, rh+RH, ... (16-byte reads), the RH row groups
          // are folded through LDS and row group 0 publishes with 16-byte write-through stores.
          constexpr int RH = 256 / (16 * NC);            // row groups (2 for H = 512)
          constexpr int RPT = (32 + RH - 1) / RH;        // rows per thread (max) of each half (resident / streamed)
          const int cq = tid % (16 * NC), rh = tid / (16 * NC);
          float4 ev[RPT];
          float pv[RPT];
#pragma unroll
          for (int i = 0; i < RPT; ++i) {
            const int t = rh + RH * i;
            const int tc = min(t, nres - 1);
            ev[i] = *reinterpret_cast<const float4*>(encS + tc * HH + 4 * cq);
            pv[i] = t < nres ? pS_[t] : 0.f;
          }
          float4 acc4 = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
          for (int i = 0; i < RPT; ++i) {
            acc4.x += pv[i] * ev[i].x; acc4.y += pv[i] * ev[i].y; acc4.z += pv[i] * ev[i].z; acc4.w += pv[i] * ev[i].w;
          }
          if (nx > 0) {
#pragma unroll
            for (int i = 0; i < RPT; ++i) {
              const int t = rh + RH * i;
              ev[i] = *reinterpret_cast<const float4*>(gEnc + (long)(nres + min(t, nx - 1)) * HH + 4 * cq);
              pv[i] = t < nx ? pS_[nres + t] : 0.f;
            }
#pragma unroll
            for (int i = 0; i < RPT; ++i) {
              acc4.x += pv[i] * ev[i].x; acc4.y += pv[i] * ev[i].y; acc4.z += pv[i] * ev[i].z; acc4.w += pv[i] * ev[i].w;
            }
          }
          float4* fold = reinterpret_cast<float4*>(scr + HH + 128);   // [RH-1][16 NC] float4 (the launcher checks the scratch size)
          if (RH > 1) {
            if (rh > 0) fold[(rh - 1) * (16 * NC) + cq] = acc4;
            __syncthreads();
          }
          if (rh == 0) {
#pragma unroll
            for (int k = 1; k < RH; ++k) {
              const float4 o = fold[(k - 1) * (16 * NC) + cq];
              acc4.x += o.x; acc4.y += o.y; acc4.z += o.z; acc4.w += o.w;
            }
            u32x4 u;
            u.x = __float_as_uint(acc4.x); u.y = __float_as_uint(acc4.y); u.z = __float_as_uint(acc4.z); u.w = __float_as_uint(acc4.w);
            __builtin_amdgcn_raw_buffer_store_b128(u, r_part, (int)((prow - a.PART + 4 + 4 * cq) * 4), 0, 16);
          }
          TQ(4)
        }
      } else {
        float mg = -INFINITY;
        if (tid >= nrow && tid < c4) scS[tid] = -INFINITY;        // pad the score vector for the float4 sweeps below
        __syncthreads();
        TICK(13)
        {
          // pass 1: 16 lanes per row, two rows per trip; lane l covers floats 4l + 64c (conflict-free LDS reads)
          const int grp = tid >> 4, l16 = tid & 15;
          for (int t = grp; t < nrow; t += 32) {
            const int t2 = t + 16;
            const bool two = t2 < nrow;
            const float* e1 = encAS + t * H + 4 * l16;
            const float* e2 = encAS + (two ? t2 : t) * H + 4 * l16;
            float d1 = 0.f, d2 = 0.f;
  #pragma unroll
            for (int c = 0; c < 16; ++c) {
              if (64 * c < H) {                                     // wave-uniform: H is a multiple of 64, <= 1024
                const float4 hv = *reinterpret_cast<const float4*>(hS + 64 * c + 4 * l16);
                const float4 x1 = *reinterpret_cast<const float4*>(e1 + 64 * c);
                const float4 x2 = *reinterpret_cast<const float4*>(e2 + 64 * c);
                d1 += x1.x * hv.x + x1.y * hv.y + x1.z * hv.z + x1.w * hv.w;
                d2 += x2.x * hv.x + x2.y * hv.y + x2.z * hv.z + x2.w * hv.w;
              }
            }
  #pragma unroll
            for (int o = 8; o > 0; o >>= 1) { d1 += __shfl_xor(d1, o); d2 += __shfl_xor(d2, o); }
            if (l16 == 0) {
              scS[t] = d1 + ebS[t];
              if (two) scS[t2] = d2 + ebS[t2];
            }
          }
        }
        TQ(0)
        __syncthreads();
        TQ(1)
        // chunk max / exp / sum: every thread sweeps the (<= 256) scores with broadcast float4 LDS reads -- no shuffles
        for (int t = 0; t < nrow; t += 4) {
          const float4 v = *reinterpret_cast<const float4*>(scS + t);
          mg = fmaxf(fmaxf(mg, fmaxf(v.x, v.y)), fmaxf(v.z, v.w));
        }
        m = mg;
        if (tid < c4) pS[tid] = tid < nrow ? expf(scS[tid] - m) : 0.f;
        my_score = tid < nrow ? scS[tid] : 0.f;
        TQ(2)
        __syncthreads();
        TQ(3)
        float cvp[4] = {0.f, 0.f, 0.f, 0.f};
        TICK(14)
        {
          // pass 2: context partial; thread owns columns tid + 256 j (consecutive threads -> consecutive LDS words); 4 rows per trip
          const int c0 = min(tid, H - 1), c1 = min(tid + 256, H - 1), c2 = min(tid + 512, H - 1), c3 = min(tid + 768, H - 1);
          for (int t = 0; t < nrow; t += 4) {
            const float4 pv = *reinterpret_cast<const float4*>(pS + t);      // rows beyond nrow have p = 0 and read the next slice rows
            l += (pv.x + pv.y) + (pv.z + pv.w);
            const float* e0 = encS + t * H;
            const int r1 = min(t + 1, nrow - 1) - t, r2 = min(t + 2, nrow - 1) - t, r3 = min(t + 3, nrow - 1) - t;
            cvp[0] += pv.x * e0[c0] + pv.y * e0[r1 * H + c0] + pv.z * e0[r2 * H + c0] + pv.w * e0[r3 * H + c0];
            if (H > 256) cvp[1] += pv.x * e0[c1] + pv.y * e0[r1 * H + c1] + pv.z * e0[r2 * H + c1] + pv.w * e0[r3 * H + c1];
            if (H > 512) {
              cvp[2] += pv.x * e0[c2] + pv.y * e0[r1 * H + c2] + pv.z * e0[r2 * H + c2] + pv.w * e0[r3 * H + c2];
              cvp[3] += pv.x * e0[c3] + pv.y * e0[r1 * H + c3] + pv.z * e0[r2 * H + c3] + pv.w * e0[r3 * H + c3];
            }
          }
  #pragma unroll
          for (int j = 0; j < 4; ++j)
            if (tid + 256 * j < H) st_sc1(&prow[4 + tid + 256 * j], cvp[j]);
          TQ(4)
        }
      }
      if (tid == 0) { st_sc1(&prow[0], m); st_sc1(&prow[1], l); }
      publish(ROWCTR(b));      // per-row counter: 8 arrivals instead of 128 on one word, and the combine waits for ITS row only
      TQ(5)
      if (a.tick_out) tk_att += wall_clock64() - ta0;
      if (tid < nrow) a.ALPHA[((long)s * B + b) * Tp + t0 + tid] = my_score;   // raw score, normalised by the backward (M, 1/L in ML)
      TICK(5)
    }
    if (has_ce && p6_pending >= 0) {       // P6 of the previous step, deferred behind this step's attention partial (see P6 below)
      if (!run_p6(p6_pending)) return;
      p6_pending = -1;
    }
    // ================= P3b: combine the nsplit partials of one batch row =================
    if (has_cmb) {
      const int b = cmb_b, bt = b / 16;
      const int rows_bt = min(16, B - bt * 16);
      TICK(15)
      if (!wg_wait(ROWCTR(b), (unsigned)(a.nsplit * (s + 1)), a.ab, &s_flag)) return;
      TICK(6)
      // No staging, no barrier: lane k of EVERY wave reads the header {max_k, sum_k} of partial k, the softmax weights of the
      // nsplit partials are formed with wave shuffles (identically in every wave), and thread tid < H/4 folds its four columns of the
      // nsplit partial context vectors straight from 16-byte sc1 loads (8 in flight) into one 16-byte write-through store.
      const long pbo = ((long)s * B + b) * a.nsplit * (H + 4);
      float mk = -INFINITY, lk = 0.f;
      if (lane < a.nsplit) {
        const float4 hd = ldb128_sc1(r_part, pbo + (long)lane * (H + 4));
        mk = hd.x; lk = hd.y;
      }
      const int cq = min(tid, H / 4 - 1);
      float4 acc4 = make_float4(0.f, 0.f, 0.f, 0.f);
      float Mx = mk;
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) Mx = fmaxf(Mx, __shfl_xor(Mx, o));
      const float wk_mine = lane < a.nsplit ? expf(mk - Mx) : 0.f;
      const float Ls = wave_sum(lk * wk_mine);
      const float inv = 1.f / Ls;
      for (int k0 = 0; k0 < a.nsplit; k0 += 8) {
        float4 pv[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) pv[j] = ldb128_sc1(r_part, pbo + (long)min(k0 + j, a.nsplit - 1) * (H + 4) + 4 + 4 * cq);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const float wk = __shfl(wk_mine, min(k0 + j, 63));     // 0 for k >= nsplit
          const float wv = k0 + j < a.nsplit ? wk : 0.f;
          acc4.x += wv * pv[j].x; acc4.y += wv * pv[j].y; acc4.z += wv * pv[j].z; acc4.w += wv * pv[j].w;
        }
      }
      if (tid < H / 4) {
        u32x4 u;
        u.x = __float_as_uint(acc4.x * inv); u.y = __float_as_uint(acc4.y * inv); u.z = __float_as_uint(acc4.z * inv); u.w = __float_as_uint(acc4.w * inv);
        __builtin_amdgcn_raw_buffer_store_b128(u, r_cvh, (int)((((long)s * B + b) * 2 * H + 4 * tid) * 4), 0, 16);
      }
      publish_sh(CTR(PH_CMB, bt), b - bt * 16);
      TICK(7)
      if (tid == 0) { a.ML[((long)s * B + b) * 2] = Mx; a.ML[((long)s * B + b) * 2 + 1] = inv; }
    }
    // ================= P4: ht = tanh(Wc [cv;h] + bc) =================
    if (has_c) {
      const int bt = c_bt, m0 = bt * 16;
      const int rows_bt = min(16, B - bt * 16);
      TICK(15)
      f32x4 acc = {0.f, 0.f, 0.f, 0.f};
      const long crow = ((long)s * B + min(m0 + r16, B - 1)) * 2 * H;
      if constexpr (NC > 0) {
        // the h half of [cv ; h] was published by the cells long ago: its fragments (k-blocks NC..2NC-1 of each wave) and their MFMAs
        // run BEFORE the wait on the combine; only the cv half is fetched and multiplied behind it
        if (!wg_wait_sh(CTR(PH_CELL + TOP, bt), H / 8, s + 1, a.ab, &s_flag)) return;
        {
          float4 ahd[NC];
          const int q = lane >> 4;
#pragma unroll
          for (int i = 0; i < NC; ++i) ahd[i] = ldb128_sc1(r_cvh, crow + 16 * (wave + 4 * (NC + i)) + 4 * q);
          __builtin_amdgcn_sched_barrier(0);
          mfma_blocks<NC>(acc, ahd, wreg + OFF_WC + NC);
        }
        if (!wg_wait_sh(CTR(PH_CMB, bt), rows_bt, s + 1, a.ab, &s_flag)) return;
        TICK(8)
        {
          float4 acv[NC];
          const int q = lane >> 4;
#pragma unroll
          for (int i = 0; i < NC; ++i) acv[i] = ldb128_sc1(r_cvh, crow + 16 * (wave + 4 * i) + 4 * q);
          __builtin_amdgcn_sched_barrier(0);
          mfma_blocks<NC>(acc, acv, wreg + OFF_WC);
        }
      } else {
        if (!wg_wait_sh(CTR(PH_CMB, bt), rows_bt, s + 1, a.ab, &s_flag)) return;
        TICK(8)
        wmac<NB_C>(acc, wreg + OFF_WC, r_cvh, crow, 2 * H, lane, wave);
      }
      const float v = reduce16(acc, red);
      const int row = m0 + (tid >> 4), n = c_n0 + (tid & 15);
      if (row < B) {
        const float ht = tanh_fast(v + a.bc[n]);
        st_sc1_u(a.HT + (long)(s + 1) * B * A, (unsigned)(row * A + n), ht);
        if (s + 1 < S) st_sc1_u(a.X0 + (long)(s + 1) * B * XI + E, (unsigned)(row * XI + n), ht);
      }
      publish_sh(CTR(PH_CTX, bt), c_n0 / 16);
      TICK(9)
    }
    // ================= P5: logits tiles + per-tile softmax statistics =================
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      if (l_item[t] < 0) continue;
      const int bt = l_item[t] / a.ntile_v, tile = l_item[t] % a.ntile_v, m0 = bt * 16, n0 = tile * 16;
      TICK(15)
      if (!wg_wait_sh(CTR(PH_CTX, bt), A / 16, s + 1, a.ab, &s_flag)) return;
      TICK(10)
      f32x4 acc = {0.f, 0.f, 0.f, 0.f};
      wmac<NB_L>(acc, wreg + OFF_WL + t * NB_L, r_ht, ((long)(s + 1) * B + min(m0 + r16, B - 1)) * A, A, lane, wave);
      const float v = reduce16(acc, red);
      const int row = m0 + (tid >> 4), n = n0 + (tid & 15);
      const bool ok = row < B && n < V;
      const float x = ok ? v + a.bo[n] : -INFINITY;
      if (ok) *ua(a.LOGITS + (long)s * B * a.Vp, (unsigned)(row * a.Vp + n)) = x;
      else if (row < B && n < a.Vp) *ua(a.LOGITS + (long)s * B * a.Vp, (unsigned)(row * a.Vp + n)) = 0.f;
      float mx = x;
      int mi = n;
#pragma unroll
      for (int o = 8; o > 0; o >>= 1) {
        const float om = __shfl_xor(mx, o);
        const int oi = __shfl_xor(mi, o);
        if (om > mx || (om == mx && oi < mi)) { mx = om; mi = oi; }
      }
      float se = ok ? expf(x - mx) : 0.f;
#pragma unroll
      for (int o = 8; o > 0; o >>= 1) se += __shfl_xor(se, o);
      const int tgt = row < B ? *ua(a.ytgt + s + 1, (unsigned)(row * a.L)) : 0;
      float xt = (ok && n == tgt) ? x : 0.f;
#pragma unroll
      for (int o = 8; o > 0; o >>= 1) xt += __shfl_xor(xt, o);
      if ((tid & 15) == 0 && row < B) {
        float* cs = ua(a.CESTAT + ((long)s * B * a.ntile_v + tile) * 4, (unsigned)(row * a.ntile_v * 4));
        st_sc1(cs, mx); st_sc1(cs + 1, se); st_sc1(cs + 2, __int_as_float(mi)); st_sc1(cs + 3, xt);
      }
      publish_sh(CTR(PH_LOG, bt), tile);
      TICK(11)
    }
    // ================= P6: cross-entropy combine per batch tile =================
    // The CE workgroups also own attention items of the NEXT step's chain.  P6(s) ends ~4 us after h_{s+1} is there (it waits for the
    // last logits tile), so run in program order it delayed those attention partials -- and with them their batch rows' combine, the
    // whole batch tile's ht and every later step -- by 3.3 us per decoder step (phase stamps: the combine of row 31 waited 4.8 us for
    // its partials, that of row 0 1.5 us).  Nothing on the chain needs P6(s) when step s+1 is teacher-forced (its token is the truth):
    // it is then DEFERRED behind the workgroup's P3 of step s+1.  When step s+1 feeds the argmax back its cells wait for PH_CE(s)
    // anyway and P6(s) runs in place (deferring it there would dead-lock: P3(s+1) waits for cells that wait for P6(s)).
    if (has_ce) {
      const bool defer = s + 1 < S && flagS[s + 1] != 0;
      if (defer) p6_pending = s;
      else if (!run_p6(s)) return;
    }
  }
  if (has_ce && p6_pending >= 0) { if (!run_p6(p6_pending)) return; }
  if (a.tick_out && tid == 0 && has_att) {     // attention phase = hand-off satisfied -> partial published (10 ns ticks -> us)
    atomicAdd(&a.tick_out[wg], (float)tk_att * 0.01f / (float)S);
    if (wg == 0) atomicAdd(&a.tick_out[G], 1.0f);
  }
  if (timing && tid == 0 && (wg == 0 || wg == n_cell || wg == G - B || wg == G - 1))
    printf("pdec wg %3d per-step 10ns: P1[pre %lld waitctx %lld ht+epi %lld publish %lld] P3[wait %lld work %lld] P3b[wait %lld work %lld] "
           "P4[wait %lld work %lld] P5[wait %lld work %lld] P6 %lld other %lld | P3: hload %lld pass1+max %lld\n", wg, tk[0] / S, tk[1] / S, tk[2] / S, tk[3] / S,
           tk[4] / S, tk[5] / S, tk[6] / S, tk[7] / S, tk[8] / S, tk[9] / S, tk[10] / S, tk[11] / S, tk[12] / S, tk[15] / S, tk[13] / S, tk[14] / S);
  if (timing && tid == 0 && (wg == 0 || wg == G - 1))
    printf("pdec-att wg %3d per-step 10ns: pass1 %lld barrier %lld max+exp %lld barrier %lld pass2+stores %lld publish %lld\n", wg, tq[0] / S, tq[1] / S,
           tq[2] / S, tq[3] / S, tq[4] / S, tq[5] / S);
#undef TICK
#undef TQ
#undef CTR
#undef ROWCTR
}

// =====================================================================================================================
// Backward loop.  Phases per step (s descending; n = S-1-s steps already done):
//   B1 d_pre = (dlogits Wo + d_ht carried from step s+1) (1 - ht^2)      items (bt, 16 cols of A)
//   B2 d_cvh = d_pre Wc  -> d_cv | dh (direct part)                      items (bt, 32 cols of 2H)
//   B3 attention backward over the LDS-resident slices:  ds = alpha (enc.d_cv - cv.d_cv) ; dh_att partial = sum_t ds encA
//   B5 cell backward: dh = mask (dh_direct + sum_splits dh_att) + dz_{s+1} Wl ; dz ; dc carried in a register
//      NL > 1: one B5 per layer, top first.  A layer below the top takes the gradient of its dropped output from the layer above by
//      multiplying dz_{l+1,s} with its 16 columns of Wu_{l+1} ITSELF (no separate d_x phase, one hand-off per layer):
//      dh_l = mask_l (dz_{l+1,s} Wu_{l+1}) + dz_{l,s+1} Wl_l
//   B6 d_x0 = dz_0 Wu_0  (its [E:] half is the d_ht carry of step s-1)   items (bt, 16 cols of E+A)
// Products that do not depend on the current step's chain (dlogits Wo in B1, dz_{s+1} Wl in B5) run before the wait.
// Weight gradients, d_enc, dq and d_embed are batched products over all steps after the loop (decoder.hip).
enum BPhase { PB1 = 0, PB2, PB5 /* + layer: 0..2 */, PB5_1, PB5_2, PB6, PB_N };

struct PDecBwdArgs {
  int B, S, L, T, Tp, H, E, A, V, Vp, XI, nbt, nsplit, chunk;
  int NL;
  const float *WoT, *WcT;                  // (A,Vp) (2H,A)
  const float *WlT[PDEC_MAX_LAYERS], *WuT[PDEC_MAX_LAYERS];      // per layer: (H,4H), (in,4H) with in = XI for layer 0, H above
  const float *enc, *encA;
  const float *CVH, *HT, *LOGITS, *ML;
  const float *Cst[PDEC_MAX_LAYERS], *rnn_mask[PDEC_MAX_LAYERS];
  float *ALPHA;                            // raw scores in, normalised alpha out
  float *Gt[PDEC_MAX_LAYERS];              // gates -> dz
  float *DPRE, *DCVH, *DS, *DX0, *DHATT;   // DHATT [S][B][nsplit][H]
  float* DXH;                              // b6_split: [2][S][B][A] K-halves of d_x0[:, E:] (the carry of the next step's B1)
  int b6_split;                            // 1: B6 covers only the ht columns, every item split into two K halves (the embedding
                                           //    columns are one batched GEMM after the launch)
                                           // 2: no B6 role and no d_pre hand-off: the B1 items (two K halves each) add dz_{s+1} Wu[:, ht cols]
                                           //    to their half of dlogits Wo and leave the two partial pre-activations in DXH; B2 forms
                                           //    d_pre = (p0 + p1) (1 - ht^2) while it loads its operand (two hand-offs less on the chain; the
                                           //    embedding columns as under 1)
  float *d_c0;                             // [NL][B][H]
  unsigned* ctr;
  AbortCtl ab;
  int dbg;
  float* tick_out;
};

constexpr int NB_B1 = 18, NB_B2 = 8, NB_B5 = 32, NB_B6 = 32;     // k-blocks per wave: Vp <= 1152, A <= 512, 4H <= 2048
constexpr int NWB = 36;
// NL > 1: B5 of layer l on workgroups [l n5, (l+1) n5) with Wl_l^T rows in [0, 32) and (below the top) Wu_{l+1}^T rows in [32, 64);
// B1/B2 on the last max(n1, n2) workgroups in [0, 34); the split B6 items on the last n6 workgroups in [48, 64)
constexpr int NWB_ML = 64, OFF_B5UP = NB_B5, OFF_B6_ML = 48;

// acc += A . W over NB blocks, A fetched in chunks of CH blocks (bounds the live A registers)
template <int NB, int CH>
__device__ __forceinline__ void wmac_chunked(f32x4& acc, const float4* w, __amdgpu_buffer_rsrc_t ra, long a_off, int K, int lane, int wave) {
#pragma unroll
  for (int c0 = 0; c0 < NB; c0 += CH) {
    if (c0 > 0 && 64 * c0 >= K) break;     // a chunk wholly beyond K (narrow models in the full-width register layout): its round trip
                                           // would sit on the chain for nothing
    float4 a[CH];
    aload_sc1<CH>(a, ra, a_off, K, lane, wave, c0);
    __builtin_amdgcn_sched_barrier(0);
    mfma_blocks<CH>(acc, a, w + c0);
  }
}

template <int NC, int NL, bool XS>   // as in decoder_persist_fwd
__global__ __launch_bounds__(256, 1) void decoder_persist_bwd(PDecBwdArgs a) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  __shared__ __attribute__((aligned(16))) float red[4 * 256];
  __shared__ int s_flag;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wg = blockIdx.x;
  const int B = a.B, S = a.S, H = a.H, E = a.E, A = a.A, XI = a.XI, T = a.T, Tp = a.Tp, Vp = a.Vp;
  const int nbt = a.nbt, K4 = 4 * H;
  unsigned* ctr = a.ctr;
#define CTR(ph, bt) (ctr + ((ph) * nbt + (bt)) * NSH * CTRS)
#define ROWCTR(row) (ctr + ((long)NPHASE_SLOTS * NSH * nbt + 2 + (row)) * CTRS)
  // ---------------- roles.  NL == 1: [0,n5) cell bwd, [n5, n5+n6) dx0, [n5+n6, n5+n6+n1) d_pre + d_cvh ; attention: all.
  // NL > 1: [l n5, (l+1) n5) cell bwd of layer l ; d_pre + d_cvh on the last max(n1, n2) workgroups ; dx0 (split) on the last n6
  constexpr int TOP = NL - 1;
  const bool fused6 = a.b6_split == 2;
  const int n5 = nbt * (H / 16), n6 = fused6 ? 0 : (a.b6_split ? 2 * nbt * (A / 16) : nbt * (XI / 16)), n1 = nbt * (A / 16), n2 = nbt * (2 * H / 32);
  const int n12 = n1 > n2 ? n1 : n2;
  const bool has5 = wg < NL * n5;
  const int b5_l = has5 ? wg / n5 : 0, b5_i = has5 ? wg % n5 : 0;
  const int b5_bt = b5_i / (H / 16), b5_u0 = (b5_i % (H / 16)) * 16;
  const int r6 = NL > 1 ? wg - (G - n6) : wg - n5;
  const bool has6 = r6 >= 0 && r6 < n6;
  // B6 item: 16 columns of d_x0 for one batch tile; split mode: ht columns only, item (r6 >> 1), K half (r6 & 1)
  const int b6_half = a.b6_split ? (r6 & 1) : 0;
  const int b6_item = a.b6_split ? (r6 >> 1) : r6;
  const int b6_per_bt = a.b6_split ? A / 16 : XI / 16;
  const int b6_bt = has6 ? b6_item / b6_per_bt : 0, b6_n0 = has6 ? (a.b6_split ? E : 0) + (b6_item % b6_per_bt) * 16 : 0;
  const int b6_k0 = b6_half * (K4 / 2);
  // fused form: 2 n1 half-items of the pre-activation, on other workgroups than the d_cvh items
  const int n1f = fused6 ? 2 * n1 : n1;
  const int r1 = fused6 ? (NL > 1 ? wg - (G - n1f) : wg - n5) : (NL > 1 ? wg - (G - n12) : wg - n5 - n6);
  const bool has1 = r1 >= 0 && r1 < n1f;
  const int b1_half = fused6 ? (r1 & 1) : 0, b1_item = fused6 ? (r1 >> 1) : r1;
  const int b1_bt = has1 ? b1_item / (A / 16) : 0, b1_n0 = has1 ? (b1_item % (A / 16)) * 16 : 0;
  constexpr int KV_HALF = 64 * (NB_B1 / 2);          // fused form: dlogits Wo split at this k (9 k-blocks per wave and half)
  const int b1_kv0 = b1_half * KV_HALF, b1_kvn = b1_half ? Vp - KV_HALF : min(Vp, KV_HALF);
  // (NL > 1: the pre-activation half-items on the last 2 n1 workgroups -- top-layer cell owners among them, never a lower layer's --
  //  and the d_cvh items on the last n2, none of which owns a cell: decoder_persist_bwd_launch checked both)
  const int r2 = fused6 ? (NL > 1 ? wg - (G - n2) : wg - n5 - n1f) : r1;
  const bool has2 = r2 >= 0 && r2 < n2;
  const int b2_bt = has2 ? r2 / (2 * H / 32) : 0, b2_n0 = has2 ? (r2 % (2 * H / 32)) * 32 : 0;
  const int n_att = B * a.nsplit;
  const bool has_att = wg < n_att;
  const int att_b = has_att ? wg % B : 0, att_sp = has_att ? wg / B : 0;

  constexpr int NW = NL > 1 ? NWB_ML : NWB;
  constexpr int OFF_B6 = NL > 1 ? OFF_B6_ML : 0;
  // fused form: the pre-activation half-item's 9 + 16 blocks.  NL > 1: behind the d_cvh item's blocks [NB_B1, NB_B1 + 2 NB_B2) and a
  // top-layer cell owner's [0, NB_B5), either of which the same workgroup may hold
  constexpr int OFF_B1F = NL > 1 ? NB_B1 + 2 * NB_B2 : 0;
  static_assert(OFF_B1F + NB_B1 / 2 + NB_B6 / 2 <= NW && NB_B5 <= NB_B1 + 2 * NB_B2, "fused pre-activation fragments");
  float4 wreg[NW];
#pragma unroll
  for (int i = 0; i < NW; ++i) wreg[i] = make_float4(0.f, 0.f, 0.f, 0.f);
  const int r16 = lane & 15;
  if (has5) {
    wload<NB_B5>(wreg, a.WlT[b5_l], K4, b5_u0 + r16, K4, lane, wave);
    if constexpr (NL > 1) {
      if (b5_l < TOP) wload<NB_B5>(wreg + OFF_B5UP, a.WuT[b5_l + 1], K4, b5_u0 + r16, K4, lane, wave);
    }
  }
  if (has6 && (NL > 1 || !has5)) {
    if (a.b6_split) wload<NB_B6 / 2>(wreg + OFF_B6, a.WuT[0] + b6_k0, K4, b6_n0 + r16, K4 / 2, lane, wave);
    else wload<NB_B6>(wreg + OFF_B6, a.WuT[0], K4, b6_n0 + r16, K4, lane, wave);
  }
  if (fused6) {
    if (has1) {
      wload<NB_B1 / 2>(wreg + OFF_B1F, a.WoT + b1_kv0, Vp, b1_n0 + r16, b1_kvn, lane, wave);
      wload<NB_B6 / 2>(wreg + OFF_B1F + NB_B1 / 2, a.WuT[0] + b1_half * (K4 / 2), K4, E + b1_n0 + r16, K4 / 2, lane, wave);
    }
    if (has2) {
      wload<NB_B2>(wreg + NB_B1, a.WcT, A, b2_n0 + r16, A, lane, wave);
      wload<NB_B2>(wreg + NB_B1 + NB_B2, a.WcT, A, b2_n0 + 16 + r16, A, lane, wave);
    }
  } else if (!has5 && (NL > 1 || !has6)) {
    if (has1) wload<NB_B1>(wreg, a.WoT, Vp, b1_n0 + r16, Vp, lane, wave);
    if (has2) {
      wload<NB_B2>(wreg + NB_B1, a.WcT, A, b2_n0 + r16, A, lane, wave);
      wload<NB_B2>(wreg + NB_B1 + NB_B2, a.WcT, A, b2_n0 + 16 + r16, A, lane, wave);
    }
  }
  const int t0 = att_sp * a.chunk, t1 = min(T, t0 + a.chunk);
  const int nrow = has_att ? t1 - t0 : 0;
  const int cres = NC > 0 ? min(a.chunk, PDEC_RES_ROWS) : a.chunk;      // rows of the slice kept in LDS (as in the forward kernel)
  const int nres = min(nrow, cres);
  float* encS = lds;
  float* encAS = lds + cres * H;
  float* scr = lds + 2 * cres * H;           // dS[H] (d_cv) | cvS[H] | ds[chunk] | wred[8]
  const float* gEnc = a.enc + ((long)att_b * T + t0) * H;
  const float* gEncA = a.encA + ((long)att_b * T + t0) * H;
  if (has_att) {
    const int n4 = nres * H / 4;
    for (int i = tid; i < n4; i += 256) {
      reinterpret_cast<float4*>(encS)[i] = reinterpret_cast<const float4*>(gEnc)[i];
      reinterpret_cast<float4*>(encAS)[i] = reinterpret_cast<const float4*>(gEncA)[i];
    }
  }
  __syncthreads();
  const __amdgpu_buffer_rsrc_t r_dl = make_rsrc(a.LOGITS), r_dpre = make_rsrc(a.DPRE), r_dcvh = make_rsrc(a.DCVH),
                               r_g = make_rsrc(a.Gt[b5_l]), r_g0 = make_rsrc(a.Gt[0]), r_dx0 = make_rsrc(a.DX0), r_dha = make_rsrc(a.DHATT);
  const int e_row = tid >> 4, e_col = tid & 15;
  float dc_state = 0.f;
  long long tk_att = 0;
  long long tb[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) tb[i] = 0;
  const bool timing = (NL == 1 || ASTK_PDEC_TIMING_ALL) && PERSIST_DBG(a) != 0;
  long long tlast = timing ? wall_clock64() : 0;
#define TB(i) if (timing) { const long long now_ = wall_clock64(); tb[i] += now_ - tlast; tlast = now_; }

  for (int s = S - 1; s >= 0; --s) {
    const int n = S - 1 - s;
    // ================= B1 =================
    if (has1) {
      const int bt = b1_bt, m0 = bt * 16;
      f32x4 acc = {0.f, 0.f, 0.f, 0.f};
      const int row = m0 + e_row, col = b1_n0 + e_col;
      if (fused6) {
        // this half of dlogits Wo (no dependency on the chain), then -- in the same accumulators -- this half of the carry
        // d_x0[s+1][:, E + col] = dz_{0,s+1} Wu_0[:, E + col]: the cell backward of step s+1 hands its dz straight to this product
        if (b1_kvn > 0) wmac_chunked<NB_B1 / 2, 9>(acc, wreg + OFF_B1F, r_dl, ((long)s * B + min(m0 + r16, B - 1)) * Vp + b1_kv0, b1_kvn, lane, wave);
        TB(0)
        if (n > 0) {
          if (!wg_wait_sh(CTR(PB5, bt), H / 16, n, a.ab, &s_flag)) return;
          TB(1)
          wmac_chunked<NB_B6 / 2, 16>(acc, wreg + OFF_B1F + NB_B1 / 2, r_g0, ((long)(s + 1) * B + min(m0 + r16, B - 1)) * K4 + b1_half * (K4 / 2), K4 / 2, lane, wave);
        }
        const float v = reduce16(acc, red);
        if (row < B) st_sc1(a.DXH + ((long)(b1_half * S + s) * B + row) * A + col, v);
      } else {
        wmac_chunked<NB_B1, 9>(acc, wreg, r_dl, ((long)s * B + min(m0 + r16, B - 1)) * Vp, Vp, lane, wave);   // no dependency on the chain
        const float v = reduce16(acc, red);
        float carry = 0.f;
        TB(0)
        if (n > 0) {
          if (!wg_wait_sh(CTR(PB6, bt), a.b6_split ? 2 * (A / 16) : XI / 16, n, a.ab, &s_flag)) return;
          TB(1)
          if (row < B) {
            if (a.b6_split) {
              const float c0 = ld_sc1(a.DXH + ((long)(s + 1) * B + row) * A + col);
              const float c1 = ld_sc1(a.DXH + ((long)(S + s + 1) * B + row) * A + col);
              carry = c0 + c1;
            } else carry = ld_sc1(a.DX0 + ((long)(s + 1) * B + row) * XI + E + col);
          }
        }
        if (row < B) {
          const float y = a.HT[((long)(s + 1) * B + row) * A + col];
          st_sc1(a.DPRE + ((long)s * B + row) * A + col, (v + carry) * (1.f - y * y));
        }
      }
      publish_sh(CTR(PB1, bt), fused6 ? 2 * (b1_n0 / 16) + b1_half : b1_n0 / 16);
      TB(2)
    }
    // ================= B2 =================
    if (has2) {
      const int bt = b2_bt, m0 = bt * 16;
      TB(15)
      float4 av[NB_B2];
      if (fused6) {
        // d_pre = (p0 + p1) (1 - ht^2) formed on the operand fragments; ht fetched in front of the wait; every item leaves its share
        // of the 4 NB_B2 fragments of d_pre for the weight gradients behind the launch
        float4 yv[NB_B2], p1[NB_B2];
        const __amdgpu_buffer_rsrc_t r_ht = make_rsrc(a.HT), r_dxh = make_rsrc(a.DXH);
        const long rowo = (long)s * B + min(m0 + r16, B - 1);
        aload_sc1<NB_B2, 0>(yv, r_ht, (rowo + B) * A, A, lane, wave);
        if (!wg_wait_sh(CTR(PB1, bt), 2 * (A / 16), n + 1, a.ab, &s_flag)) return;
        TB(3)
        aload_sc1<NB_B2>(av, r_dxh, rowo * A, A, lane, wave);
        aload_sc1<NB_B2>(p1, r_dxh, ((long)S * B + rowo) * A, A, lane, wave);
        const int nb2 = 2 * H / 32, my = b2_n0 / 32;
#pragma unroll
        for (int i = 0; i < NB_B2; ++i) {
          av[i].x = (av[i].x + p1[i].x) * (1.f - yv[i].x * yv[i].x); av[i].y = (av[i].y + p1[i].y) * (1.f - yv[i].y * yv[i].y);
          av[i].z = (av[i].z + p1[i].z) * (1.f - yv[i].z * yv[i].z); av[i].w = (av[i].w + p1[i].w) * (1.f - yv[i].w * yv[i].w);
          const int k = 16 * (wave + 4 * i) + 4 * (lane >> 4);
          if ((wave * NB_B2 + i) % nb2 == my && k < A && m0 + r16 < B) *reinterpret_cast<float4*>(a.DPRE + rowo * A + k) = av[i];
        }
      } else {
        if (!wg_wait_sh(CTR(PB1, bt), A / 16, n + 1, a.ab, &s_flag)) return;
        TB(3)
        aload_sc1<NB_B2>(av, r_dpre, ((long)s * B + min(m0 + r16, B - 1)) * A, A, lane, wave);
      }
      __builtin_amdgcn_sched_barrier(0);
      f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
      mfma_blocks<NB_B2>(acc0, av, wreg + NB_B1);
      mfma_blocks<NB_B2>(acc1, av, wreg + NB_B1 + NB_B2);
      const float v0 = reduce16(acc0, red);
      const float v1 = reduce16(acc1, red);
      const int row = m0 + e_row;
      if (row < B) {
        st_sc1(a.DCVH + ((long)s * B + row) * 2 * H + b2_n0 + e_col, v0);
        st_sc1(a.DCVH + ((long)s * B + row) * 2 * H + b2_n0 + 16 + e_col, v1);
      }
      publish_sh(CTR(PB2, bt), b2_n0 / 32);
      TB(4)
    }
    // ================= B5, the part that does not depend on this step's chain: dh_rec = dz_{s+1} Wl and the saved forward state.
    // In FRONT of the attention phase in program order: dz_{s+1} has been there since the previous step's cell backward, and a cell
    // workgroup that started this product only after its own attention item kept its batch tile's dz -- the whole chain -- waiting
    // for it (5 us per step in the phase timers) =================
    float v = 0.f;
    float4 g = make_float4(0.f, 0.f, 0.f, 0.f);
    float ccur = 0.f, cp = 0.f, mk = 1.f;
    if (has5) {
      const int bt = b5_bt, m0 = bt * 16, l = b5_l;
      TB(15)
      f32x4 acc = {0.f, 0.f, 0.f, 0.f};
      if (n > 0) {
        if (!wg_wait_sh(CTR(PB5 + l, bt), H / 16, n, a.ab, &s_flag)) return;
        wmac_chunked<NB_B5, 16>(acc, wreg, r_g, ((long)(s + 1) * B + min(m0 + r16, B - 1)) * K4, K4, lane, wave);
      }
      v = reduce16(acc, red);
      const int row = m0 + e_row, u = b5_u0 + e_col;
      if (row < B) {
        g = *reinterpret_cast<const float4*>(a.Gt[l] + ((long)s * B + row) * K4 + 4 * u);
        ccur = a.Cst[l][((long)(s + 1) * B + row) * H + u];
        cp = a.Cst[l][((long)s * B + row) * H + u];
        if (a.rnn_mask[l]) mk = a.rnn_mask[l][((long)s * B + row) * H + u];
      }
      TB(6)
    }
    // ================= B3: attention backward =================
    if (has_att) {
      const int b = att_b, bt = b / 16;
      long long tb0 = 0;
      TB(15)
      if constexpr (NC > 0) {
        // ---- specialised scan (H = 64 NC, nrow <= 60; rows >= nres come from global memory): everything that does not depend on this
        // step's chain (alpha from the saved raw scores, cv) is fetched BEFORE the wait; every read of a pass is issued before its first use
        constexpr int HH = 64 * NC;
        const int nx = XS ? nrow - nres : 0;
        float* dS = scr;                       // d_cv[b][:]
        float* dsS = scr + 2 * HH;             // ds[64] (tail 0)
        float* aS = dsS + 64;                  // alpha[64] (tail 0)
        float* wred = aS + 64;                 // [8]
        float4* fold = reinterpret_cast<float4*>(scr + 2 * HH + 160);
        float* al = a.ALPHA + ((long)s * B + b) * Tp + t0;
        float a_t = 0.f;
        float4 cv4 = make_float4(0.f, 0.f, 0.f, 0.f);
        {
          const float mlM = a.ML[((long)s * B + b) * 2], mlI = a.ML[((long)s * B + b) * 2 + 1];
          const float raw = al[min(tid, nrow - 1)];
          if (tid < HH / 4) cv4 = *reinterpret_cast<const float4*>(a.CVH + ((long)s * B + b) * 2 * HH + 4 * tid);
          a_t = tid < nrow ? expf(raw - mlM) * mlI : 0.f;
        }
        if (!wg_wait_sh(CTR(PB2, bt), 2 * H / 32, n + 1, a.ab, &s_flag)) return;
        tb0 = a.tick_out ? wall_clock64() : 0;
        float cdp = 0.f;                       // cv . d_cv
        if (tid < HH / 4) {
          const float4 d4 = ldb128_sc1(r_dcvh, ((long)s * B + b) * 2 * HH + 4 * tid);
          *reinterpret_cast<float4*>(dS + 4 * tid) = d4;
          cdp = cv4.x * d4.x + cv4.y * d4.y + cv4.z * d4.z + cv4.w * d4.w;
        }
        cdp = wave_sum(cdp);
        if (lane == 0) wred[wave] = cdp;
        if (tid < 64) { aS[tid] = a_t; dsS[tid] = 0.f; }
        if (tid < nrow) al[tid] = a_t;          // normalised alpha for the deferred d_enc product (off the chain)
        __syncthreads();
        const float cd = wred[0] + wred[1] + wred[2] + wred[3];
        {
          const int grp = tid >> 4, l16 = tid & 15;
          const int ta = min(grp, nres - 1), tb2 = min(grp + 16, nres - 1);
          const float* e1 = encS + ta * HH + 4 * l16;
          const float* e2 = encS + tb2 * HH + 4 * l16;
          float4 dv[NC], x1[NC], x2[NC];
#pragma unroll
          for (int c = 0; c < NC; ++c) {
            dv[c] = *reinterpret_cast<const float4*>(dS + 64 * c + 4 * l16);
            x1[c] = *reinterpret_cast<const float4*>(e1 + 64 * c);
            x2[c] = *reinterpret_cast<const float4*>(e2 + 64 * c);
          }
          float d1 = 0.f, d2 = 0.f;
#pragma unroll
          for (int c = 0; c < NC; ++c) {
            d1 += x1[c].x * dv[c].x + x1[c].y * dv[c].y + x1[c].z * dv[c].z + x1[c].w * dv[c].w;
            d2 += x2[c].x * dv[c].x + x2[c].y * dv[c].y + x2[c].z * dv[c].z + x2[c].w * dv[c].w;
          }
#pragma unroll
          for (int o = 8; o > 0; o >>= 1) { d1 += __shfl_xor(d1, o); d2 += __shfl_xor(d2, o); }
          if (l16 == 0) {
            if (grp < nres) {
              const float g1 = aS[grp] * (d1 - cd);
              dsS[grp] = g1;
              a.DS[((long)s * B + b) * Tp + t0 + grp] = g1;
            }
            if (grp + 16 < nres) {
              const float g2 = aS[grp + 16] * (d2 - cd);
              dsS[grp + 16] = g2;
              a.DS[((long)s * B + b) * Tp + t0 + grp + 16] = g2;
            }
          }
          if (nx > 0) {        // streamed rows (workgroup-uniform branch)
            const int ua = nres + min(grp, nx - 1), ub = nres + min(grp + 16, nx - 1);
            const float* g1p = gEnc + (long)ua * HH + 4 * l16;
            const float* g2p = gEnc + (long)ub * HH + 4 * l16;
#pragma unroll
            for (int c = 0; c < NC; ++c) {
              x1[c] = *reinterpret_cast<const float4*>(g1p + 64 * c);
              x2[c] = *reinterpret_cast<const float4*>(g2p + 64 * c);
            }
            d1 = 0.f; d2 = 0.f;
#pragma unroll
            for (int c = 0; c < NC; ++c) {
              d1 += x1[c].x * dv[c].x + x1[c].y * dv[c].y + x1[c].z * dv[c].z + x1[c].w * dv[c].w;
              d2 += x2[c].x * dv[c].x + x2[c].y * dv[c].y + x2[c].z * dv[c].z + x2[c].w * dv[c].w;
            }
#pragma unroll
            for (int o = 8; o > 0; o >>= 1) { d1 += __shfl_xor(d1, o); d2 += __shfl_xor(d2, o); }
            if (l16 == 0) {
              if (grp < nx) {
                const float g1 = aS[nres + grp] * (d1 - cd);
                dsS[nres + grp] = g1;
                a.DS[((long)s * B + b) * Tp + t0 + nres + grp] = g1;
              }
              if (grp + 16 < nx) {
                const float g2 = aS[nres + grp + 16] * (d2 - cd);
                dsS[nres + grp + 16] = g2;
                a.DS[((long)s * B + b) * Tp + t0 + nres + grp + 16] = g2;
              }
            }
          }
        }
        __syncthreads();
        {
          constexpr int RH = 256 / (16 * NC);
          constexpr int RPT = (32 + RH - 1) / RH;
          const int cq = tid % (16 * NC), rh = tid / (16 * NC);
          float4 ev[RPT];
          float gv[RPT];
#pragma unroll
          for (int i = 0; i < RPT; ++i) {
            const int t = rh + RH * i;
            ev[i] = *reinterpret_cast<const float4*>(encAS + min(t, nres - 1) * HH + 4 * cq);
            gv[i] = t < nres ? dsS[t] : 0.f;
          }
          float4 acc4 = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
          for (int i = 0; i < RPT; ++i) {
            acc4.x += gv[i] * ev[i].x; acc4.y += gv[i] * ev[i].y; acc4.z += gv[i] * ev[i].z; acc4.w += gv[i] * ev[i].w;
          }
          if (nx > 0) {
#pragma unroll
            for (int i = 0; i < RPT; ++i) {
              const int t = rh + RH * i;
              ev[i] = *reinterpret_cast<const float4*>(gEncA + (long)(nres + min(t, nx - 1)) * HH + 4 * cq);
              gv[i] = t < nx ? dsS[nres + t] : 0.f;
            }
#pragma unroll
            for (int i = 0; i < RPT; ++i) {
              acc4.x += gv[i] * ev[i].x; acc4.y += gv[i] * ev[i].y; acc4.z += gv[i] * ev[i].z; acc4.w += gv[i] * ev[i].w;
            }
          }
          if (RH > 1) {
            if (rh > 0) fold[(rh - 1) * (16 * NC) + cq] = acc4;
            __syncthreads();
          }
          if (rh == 0) {
#pragma unroll
            for (int k = 1; k < RH; ++k) {
              const float4 o = fold[(k - 1) * (16 * NC) + cq];
              acc4.x += o.x; acc4.y += o.y; acc4.z += o.z; acc4.w += o.w;
            }
            u32x4 u;
            u.x = __float_as_uint(acc4.x); u.y = __float_as_uint(acc4.y); u.z = __float_as_uint(acc4.z); u.w = __float_as_uint(acc4.w);
            __builtin_amdgcn_raw_buffer_store_b128(u, r_dha, (int)(((((long)s * B + b) * a.nsplit + att_sp) * HH + 4 * cq) * 4), 0, 16);
          }
        }
      } else {
        if (!wg_wait_sh(CTR(PB2, bt), 2 * H / 32, n + 1, a.ab, &s_flag)) return;
        tb0 = a.tick_out ? wall_clock64() : 0;
        float* dS = scr;                  // d_cv[b][:]
        float* cvS = scr + H;             // cv[b][:]
        float* dsS = scr + 2 * H;         // ds[c4] (tail 0)
        const int c4 = (a.chunk + 3) & ~3;
        float* wred = dsS + c4;           // [8]
        if (tid >= nrow && tid < c4) dsS[tid] = 0.f;
        if (tid < H / 4) {
          *reinterpret_cast<float4*>(dS + 4 * tid) = ldb128_sc1(r_dcvh, ((long)s * B + b) * 2 * H + 4 * tid);
          *reinterpret_cast<float4*>(cvS + 4 * tid) = *reinterpret_cast<const float4*>(a.CVH + ((long)s * B + b) * 2 * H + 4 * tid);
        }
        __syncthreads();
        float cdp = 0.f;                  // cv . d_cv
        for (int k = tid; k < H; k += 256) cdp += cvS[k] * dS[k];
        cdp = wave_sum(cdp);
        if (lane == 0) wred[wave] = cdp;
        __syncthreads();
        const float cd = wred[0] + wred[1] + wred[2] + wred[3];
        const float mlM = a.ML[((long)s * B + b) * 2], mlI = a.ML[((long)s * B + b) * 2 + 1];
        {
          const int grp = tid >> 4, l16 = tid & 15;
          for (int t = grp; t < nrow; t += 32) {
            const int t2 = t + 16;
            const bool two = t2 < nrow;
            const float* e1 = encS + t * H + 4 * l16;
            const float* e2 = encS + (two ? t2 : t) * H + 4 * l16;
            float d1 = 0.f, d2 = 0.f;
  #pragma unroll
            for (int c = 0; c < 16; ++c) {
              if (64 * c < H) {
                const float4 dv = *reinterpret_cast<const float4*>(dS + 64 * c + 4 * l16);
                const float4 x1 = *reinterpret_cast<const float4*>(e1 + 64 * c);
                const float4 x2 = *reinterpret_cast<const float4*>(e2 + 64 * c);
                d1 += x1.x * dv.x + x1.y * dv.y + x1.z * dv.z + x1.w * dv.w;
                d2 += x2.x * dv.x + x2.y * dv.y + x2.z * dv.z + x2.w * dv.w;
              }
            }
  #pragma unroll
            for (int o = 8; o > 0; o >>= 1) { d1 += __shfl_xor(d1, o); d2 += __shfl_xor(d2, o); }
            if (l16 == 0) {
              float* al = a.ALPHA + ((long)s * B + b) * Tp + t0;
              const float a1 = expf(al[t] - mlM) * mlI;
              al[t] = a1;                                  // normalised alpha for the deferred d_enc product
              const float g1 = a1 * (d1 - cd);
              dsS[t] = g1;
              a.DS[((long)s * B + b) * Tp + t0 + t] = g1;
              if (two) {
                const float a2 = expf(al[t2] - mlM) * mlI;
                al[t2] = a2;
                const float g2 = a2 * (d2 - cd);
                dsS[t2] = g2;
                a.DS[((long)s * B + b) * Tp + t0 + t2] = g2;
              }
            }
          }
        }
        __syncthreads();
        {
          float acc4[4] = {0.f, 0.f, 0.f, 0.f};
          const int c0 = min(tid, H - 1), c1 = min(tid + 256, H - 1), c2 = min(tid + 512, H - 1), c3 = min(tid + 768, H - 1);
          for (int t = 0; t < nrow; t += 4) {
            const float4 gv = *reinterpret_cast<const float4*>(dsS + t);
            const float* e0 = encAS + t * H;
            const int r1 = min(t + 1, nrow - 1) - t, r2 = min(t + 2, nrow - 1) - t, r3 = min(t + 3, nrow - 1) - t;
            acc4[0] += gv.x * e0[c0] + gv.y * e0[r1 * H + c0] + gv.z * e0[r2 * H + c0] + gv.w * e0[r3 * H + c0];
            if (H > 256) acc4[1] += gv.x * e0[c1] + gv.y * e0[r1 * H + c1] + gv.z * e0[r2 * H + c1] + gv.w * e0[r3 * H + c1];
            if (H > 512) {
              acc4[2] += gv.x * e0[c2] + gv.y * e0[r1 * H + c2] + gv.z * e0[r2 * H + c2] + gv.w * e0[r3 * H + c2];
              acc4[3] += gv.x * e0[c3] + gv.y * e0[r1 * H + c3] + gv.z * e0[r2 * H + c3] + gv.w * e0[r3 * H + c3];
            }
          }
          float* out = a.DHATT + (((long)s * B + b) * a.nsplit + att_sp) * H;
  #pragma unroll
          for (int j = 0; j < 4; ++j)
            if (tid + 256 * j < H) st_sc1(&out[tid + 256 * j], acc4[j]);
        }
      }
      publish(ROWCTR(b));
      if (a.tick_out) tk_att += wall_clock64() - tb0;
      TB(5)
    }
    // ================= B5: cell backward (one per decoder layer, top first) =================
    if (has5) {
      const int bt = b5_bt, m0 = bt * 16, l = b5_l;
      const int rows_bt = min(16, B - bt * 16);
      const int row = m0 + e_row, u = b5_u0 + e_col;
      const bool ev = row < B;
      TB(15)
      float dy = 0.f;
      if (NL == 1 || l == TOP) {
        if (!wg_wait_multi(ROWCTR(m0), CTRS, rows_bt, (unsigned)(a.nsplit * (n + 1)), a.ab, &s_flag)) return;
        TB(7)
        if (ev) {
          dy = ld_sc1(a.DCVH + ((long)s * B + row) * 2 * H + H + u);
          float hs = 0.f;
          for (int k0 = 0; k0 < a.nsplit; k0 += 8) {      // 8 partial loads in flight (a plain loop waits for each one)
            float hv[8];
#pragma unroll
            for (int j = 0; j < 8; ++j)
              hv[j] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(
                  r_dha, (int)(((((long)s * B + row) * a.nsplit + min(k0 + j, a.nsplit - 1)) * H + u) * 4), 0, 16));
#pragma unroll
            for (int j = 0; j < 8; ++j) hs += k0 + j < a.nsplit ? hv[j] : 0.f;
          }
          dy += hs;
        }
      } else {
        if constexpr (NL > 1) {
          // gradient of this layer's dropped output: dz_{l+1,s} times this workgroup's 16 columns of Wu_{l+1}
          if (!wg_wait_sh(CTR(PB5 + l + 1, bt), H / 16, n + 1, a.ab, &s_flag)) return;
          TB(7)
          f32x4 accu = {0.f, 0.f, 0.f, 0.f};
          wmac_chunked<NB_B5, 16>(accu, wreg + OFF_B5UP, make_rsrc(a.Gt[l + 1]), ((long)s * B + min(m0 + r16, B - 1)) * K4, K4, lane, wave);
          dy = reduce16(accu, red);
        }
      }
      if (ev) {
        const float dh = v + dy * mk;
        const float tc = tanh_fast(ccur);
        const float dcv = dh * g.w * (1.f - tc * tc) + dc_state;
        u32x4 o;
        o.x = __float_as_uint(dcv * g.y * (1.f - g.x * g.x)); o.y = __float_as_uint(dcv * g.x * g.y * (1.f - g.y));
        o.z = __float_as_uint(dcv * cp * g.z * (1.f - g.z)); o.w = __float_as_uint(dh * tc * g.w * (1.f - g.w));
        __builtin_amdgcn_raw_buffer_store_b128(o, r_g, (int)((((long)s * B + row) * K4 + 4 * u) * 4), 0, 16);
        dc_state = dcv * g.z;
      }
      publish_sh(CTR(PB5 + l, bt), b5_u0 / 16);
      TB(8)
    }
    // ================= B6: d_x0 = dz Wu =================
    if (has6) {
      const int bt = b6_bt, m0 = bt * 16;
      TB(15)
      if (!wg_wait_sh(CTR(PB5, bt), H / 16, n + 1, a.ab, &s_flag)) return;
      TB(9)
      f32x4 acc = {0.f, 0.f, 0.f, 0.f};
      if (a.b6_split) wmac_chunked<NB_B6 / 2, 16>(acc, wreg + OFF_B6, r_g0, ((long)s * B + min(m0 + r16, B - 1)) * K4 + b6_k0, K4 / 2, lane, wave);
      else wmac_chunked<NB_B6, 16>(acc, wreg + OFF_B6, r_g0, ((long)s * B + min(m0 + r16, B - 1)) * K4, K4, lane, wave);
      const float v = reduce16(acc, red);
      const int row = m0 + e_row;
      if (row < B) {
        if (a.b6_split) st_sc1(a.DXH + ((long)(b6_half * S + s) * B + row) * A + (b6_n0 - E) + e_col, v);
        else st_sc1(a.DX0 + ((long)s * B + row) * XI + b6_n0 + e_col, v);
      }
      publish_sh(CTR(PB6, bt), a.b6_split ? 2 * (b6_item % b6_per_bt) + b6_half : b6_item % b6_per_bt);
      TB(10)
    }
  }
  if (timing && tid == 0 && (wg == 0 || wg == n5 || wg == n5 + 20 || wg == 2 * n5 + 1 || wg == n5 + n6 || wg == n5 + n6 + n1 - 1 || wg == G - 1))
    printf("pdecb wg %3d per-step 10ns: B1[pre %lld wait %lld tail %lld] B2[wait %lld work %lld] B3[wait+work %lld] B5[pre %lld wait %lld tail %lld] B6[wait %lld work %lld] other %lld\n",
           wg, tb[0] / S, tb[1] / S, tb[2] / S, tb[3] / S, tb[4] / S, tb[5] / S, tb[6] / S, tb[7] / S, tb[8] / S, tb[9] / S, tb[10] / S, tb[15] / S);
#undef TB
  if (has5) {
    const int row = b5_bt * 16 + e_row, u = b5_u0 + e_col;
    if (row < B) a.d_c0[((long)b5_l * B + row) * H + u] = dc_state;
  }
  if (a.tick_out && tid == 0 && has_att) {
    atomicAdd(&a.tick_out[wg], (float)tk_att * 0.01f / (float)S);
    if (wg == 0) atomicAdd(&a.tick_out[G], 1.0f);
  }
#undef CTR
#undef ROWCTR
}

// Everything that follows the forward loop in ONE launch (a dependent launch costs ~5 us whatever it does), one block per (s, b) row:
//   * TOK[s][b] (the token that was fed) and X0[s][b][:E] = embed[TOK] * mask for the backward's wgrad / scatter,
//   * dlogits = w[t] (softmax - onehot) / B in place of the saved logits,
//   * the copy of the prediction; block 0 also sums the per-row losses (in double).
__global__ __launch_bounds__(256) void k_decoder_post(const float* __restrict__ embed, const int32_t* __restrict__ y, const int32_t* __restrict__ ytgt,
                                                      const int32_t* __restrict__ use_truth,
                                                      const int32_t* __restrict__ pred, const float* __restrict__ emb_mask, int32_t* __restrict__ tok_out,
                                                      float* __restrict__ x0, float* __restrict__ logits, const float* __restrict__ lse,
                                                      const float* __restrict__ cw, const float* __restrict__ lossrows, float* __restrict__ loss,
                                                      int32_t* __restrict__ pred_out, int S, int B, int L, int E, int XI, int V, int Vp, float inv_count,
                                                      const unsigned* __restrict__ status, float* __restrict__ status_dst) {
  const int r = blockIdx.x, s = r / B, b = r % B;
  if (r == 0 && threadIdx.x == 0 && status_dst) *status_dst = (float)(*status);      // (astk_decoder_desc.status_dst: the snapshot without its launch)
  {
    int tok = (s == 0 || use_truth[s]) ? y[(long)b * L + s] : pred[(long)(s - 1) * B + b];
    tok = tok < 0 ? 0 : (tok >= V ? V - 1 : tok);
    if (threadIdx.x == 0) {
      tok_out[r] = tok;
      if (pred_out) pred_out[r] = pred[r];
    }
    for (int e = threadIdx.x; e < E; e += 256) {
      float v = embed[(long)tok * E + e];
      if (emb_mask) v *= emb_mask[(long)r * E + e];
      x0[(long)r * XI + e] = v;
    }
  }
  {
    int t = ytgt[(long)b * L + s + 1];        // the class that was SCORED (y, or forward_loss's random_out replacement)
    t = t < 0 ? 0 : (t >= V ? V - 1 : t);
    const float scale = (cw ? cw[t] : 1.f) * inv_count;
    const float ls = lse[r];
    float* x = logits + (long)r * Vp;
    for (int v = threadIdx.x; v < Vp; v += 256) {
      float g = 0.f;
      if (v < V) {
        g = expf(x[v] - ls) * scale;
        if (v == t) g -= scale;
      }
      x[v] = g;
    }
  }
  if (r == 0 && loss) {
    __shared__ double red[4];
    double acc = 0.0;
    for (int i = threadIdx.x; i < S * B; i += 256) acc += (double)lossrows[i];
    for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) *loss = (float)(red[0] + red[1] + red[2] + red[3]);
  }
}

}  // namespace

struct DecPersistBuffers {
  int32_t *TOK, *PRED;
  float *X0, *Q, *ALPHA, *CVH, *HT, *LOGITS, *LOSSROWS;
  float *G[PDEC_MAX_LAYERS], *C[PDEC_MAX_LAYERS], *HR[PDEC_MAX_LAYERS], *HD[PDEC_MAX_LAYERS];
  float *LSE, *PART, *CESTAT, *ENCA, *ML;
  unsigned* ctr;
  // two small buffers the forward launcher zeroes with its own fill launch (HT of step -1 and the first concat row: decoder.hip)
  void* zero_a; size_t zero_a_bytes; void* zero_b; size_t zero_b_bytes;
  // ... and the initial states (n_layers, B, H) it copies into C[l] / HR[l] with the same launch (nullptr: the caller copied them)
  const float *c0, *h0;
};

static bool pdec_special(int H, int chunk) { return H == 512 && chunk <= PDEC_CHUNK_MAX; }     // the NC = 8 attention phase
static int pdec_res_rows(int H, int chunk) { return pdec_special(H, chunk) && chunk > PDEC_RES_ROWS ? PDEC_RES_ROWS : chunk; }
static size_t pdec_lds_floats(int chunk, int H, int nsplit) {
  size_t scratch = (size_t)H + 2 * (size_t)((chunk + 3) & ~3) + 16;
  if (pdec_special(H, chunk) && scratch < (size_t)H + 128 + 512) scratch = (size_t)H + 128 + 512;
  if (scratch < (size_t)nsplit * (H + 4)) scratch = (size_t)nsplit * (H + 4);
  return 2 * (size_t)pdec_res_rows(H, chunk) * H + (size_t)((chunk + 3) & ~3) + scratch;
}
static size_t pdec_bwd_lds_floats(int chunk, int H) {
  return 2 * (size_t)pdec_res_rows(H, chunk) * H + 2 * (size_t)H + (size_t)((chunk + 3) & ~3) + 16 + 160 + 512;
}

bool decoder_persist_applicable(const astk_decoder_desc* d, int* nsplit_out, int* chunk_out) {
  if (!tune_on(TUNE_DEC_PERSIST)) return false;
  if (d->n_attn > 1 || d->no_feed_attn || d->ln) return false;      // optional model features: per-launch loop (decoder.hip)
  if (d->n_layers < 1 || d->n_layers > PDEC_MAX_LAYERS) return false;
  if (device_cu_count() < G) return false;        // fixed roles over G workgroups, all of them resident (one per CU)
  if ((d->H % 64) || (d->A % 16) || (d->E % 16) || d->A < 16 || d->E < 16) return false;
  const int nbt = (d->B + 15) / 16, ntv = (d->V + 15) / 16;
  if (d->E > 64 * NB_E || d->A > 64 * NB_A || d->H > 64 * NB_H || 2 * d->H > 64 * NB_C || d->A > 64 * NB_L || d->H > 1024) return false;
  const int n_cell = nbt * (d->H / 8);
  const int rest = G - n_cell;
  if (n_cell >= G || rest < nbt * (d->A / 16) || 2 * rest < nbt * ntv) return false;
  if (rest < d->B + nbt || d->B > G) return false;
  if (d->n_layers > 1 && 2 * n_cell > G) return false;       // the top layer's cell items live on the upper workgroups
  int nsplit = G / d->B;
  if (nsplit > 64) nsplit = 64;
  if (nsplit > d->T) nsplit = d->T;
  if (nsplit < 1) return false;
  int chunk = (d->T + nsplit - 1) / nsplit;
  nsplit = (d->T + chunk - 1) / chunk;
  if (chunk > 256 || d->L > 192) return false;
  {   // backward roles and register budgets
    const int XI = d->E + d->A, Vp = (d->V + 3) / 4 * 4;
    if (Vp > 64 * NB_B1 || d->A > 64 * NB_B2 || 4 * d->H > 64 * NB_B5 || (XI % 16) || ((2 * d->H) % 32)) return false;
    if (nbt * (d->H / 16) + nbt * (XI / 16) + nbt * (d->A / 16) > G || nbt * (2 * d->H / 32) > nbt * (d->A / 16) + (G - nbt * (d->H / 16) - nbt * (XI / 16) - nbt * (d->A / 16))) return false;
    if (pdec_bwd_lds_floats(chunk, d->H) * sizeof(float) > 148 * 1024) return false;
    if (d->n_layers > 1) {
      // per-layer cell-backward owners, d_pre/d_cvh owners at the end of the grid, the split d_x0 items on the last n6 workgroups
      // (never on a workgroup that also holds the 64 float4 of a below-top cell backward)
      const int n5 = nbt * (d->H / 16), n6 = 2 * nbt * (d->A / 16), n1 = nbt * (d->A / 16), n2 = nbt * (2 * d->H / 32);
      const int n12 = n1 > n2 ? n1 : n2;
      if ((4 * d->H) % 128) return false;
      if (d->n_layers * n5 + n12 > G || n6 > G - (d->n_layers - 1) * n5) return false;
    }
  }                       // pass-1 bookkeeping uses one thread per row
  if (pdec_lds_floats(chunk, d->H, nsplit) * sizeof(float) > 136 * 1024) return false;
  *nsplit_out = nsplit;
  *chunk_out = chunk;
  return true;
}

// d_x0 phase of the backward kernel as 2 K-halves per item over the ht columns only: fits when the roles still fit 256 workgroups
bool decoder_persist_b6_split(const astk_decoder_desc* d) {
  const int nbt = (d->B + 15) / 16;
  const int n5 = nbt * (d->H / 16), n6 = 2 * nbt * (d->A / 16), n1 = nbt * (d->A / 16), n2 = nbt * (2 * d->H / 32);
  if ((4 * d->H) % 128) return false;
  if (d->n_layers > 1) return true;           // the multi-layer role layout always uses the split form (decoder_persist_applicable checked it)
  if (!tune_on(TUNE_DEC_B6_SPLIT)) return false;
  return n5 + n6 + (n1 > n2 ? n1 : n2) <= G;
}

struct DecPersistBwdBuffers {
  void* zero_ptr; size_t zero_bytes;      // astk_decoder_desc.zero_ptr: zeroed by the launcher's fill launch
  void* zero2_ptr; size_t zero2_bytes;    // d_enc: zeroed there too, its two batched products then ADD into it from one grouped launch
  const float *WoT, *WcT, *ENCA, *CVH, *HT, *LOGITS, *ML;
  const float *WlT[PDEC_MAX_LAYERS], *WuT[PDEC_MAX_LAYERS], *C[PDEC_MAX_LAYERS];
  float *G[PDEC_MAX_LAYERS];
  float *ALPHA, *DPRE, *DCVH, *DS, *DX0, *DHATT, *d_c0;
  float* DXH;        // [2][S][B][A] or null (no K split of the d_x0 phase)
  unsigned* ctr;
};

template <int NL>
static void pdec_launch_bwd(bool special, bool xs, size_t shm, hipStream_t s, const PDecBwdArgs& a) {
  static bool attr_done = false;
  if (!attr_done) {
    (void)hipFuncSetAttribute((const void*)decoder_persist_bwd<0, NL, false>, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);
    (void)hipFuncSetAttribute((const void*)decoder_persist_bwd<8, NL, false>, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);
    (void)hipFuncSetAttribute((const void*)decoder_persist_bwd<8, NL, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);
    attr_done = true;
  }
  if (special && xs) hipLaunchKernelGGL((decoder_persist_bwd<8, NL, true>), dim3(G), dim3(256), shm, s, a);
  else if (special) hipLaunchKernelGGL((decoder_persist_bwd<8, NL, false>), dim3(G), dim3(256), shm, s, a);
  else hipLaunchKernelGGL((decoder_persist_bwd<0, NL, false>), dim3(G), dim3(256), shm, s, a);
}
template <int NL>
static void pdec_launch_fwd(bool special, bool xs, size_t shm, hipStream_t s, const PDecArgs& a) {
  static bool attr_done = false;
  if (!attr_done) {
    (void)hipFuncSetAttribute((const void*)decoder_persist_fwd<0, NL, false>, hipFuncAttributeMaxDynamicSharedMemorySize, 138 * 1024);
    (void)hipFuncSetAttribute((const void*)decoder_persist_fwd<8, NL, false>, hipFuncAttributeMaxDynamicSharedMemorySize, 138 * 1024);
    (void)hipFuncSetAttribute((const void*)decoder_persist_fwd<8, NL, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 138 * 1024);
    attr_done = true;
  }
  if (special && xs) hipLaunchKernelGGL((decoder_persist_fwd<8, NL, true>), dim3(G), dim3(256), shm, s, a);
  else if (special) hipLaunchKernelGGL((decoder_persist_fwd<8, NL, false>), dim3(G), dim3(256), shm, s, a);
  else hipLaunchKernelGGL((decoder_persist_fwd<0, NL, false>), dim3(G), dim3(256), shm, s, a);
}

int decoder_persist_bwd_launch(const astk_decoder_desc* d, const float* enc, const float* rnn_masks, const DecPersistBwdBuffers& bf,
                               hipStream_t s) {
  int nsplit = 1, chunk = 1;
  ASTK_CHECK(decoder_persist_applicable(d, &nsplit, &chunk), "decoder_persist_bwd: not applicable");
  PDecBwdArgs a;
  memset(&a, 0, sizeof(a));
  a.B = d->B; a.S = d->L - 1; a.L = d->L; a.T = d->T; a.Tp = (d->T + 3) / 4 * 4; a.H = d->H; a.E = d->E; a.A = d->A; a.V = d->V;
  a.Vp = (d->V + 3) / 4 * 4; a.XI = d->E + d->A; a.nbt = (d->B + 15) / 16; a.nsplit = nsplit; a.chunk = chunk;
  a.NL = d->n_layers;
  a.WoT = bf.WoT; a.WcT = bf.WcT; a.enc = enc; a.encA = bf.ENCA; a.ALPHA = bf.ALPHA; a.CVH = bf.CVH; a.ML = bf.ML;
  for (int l = 0; l < d->n_layers; ++l) {
    a.WlT[l] = bf.WlT[l]; a.WuT[l] = bf.WuT[l]; a.Cst[l] = bf.C[l]; a.Gt[l] = bf.G[l];
    a.rnn_mask[l] = rnn_masks ? rnn_masks + (size_t)l * (d->L - 1) * d->B * d->H : nullptr;
  }
  a.HT = bf.HT; a.LOGITS = bf.LOGITS; a.DPRE = bf.DPRE; a.DCVH = bf.DCVH; a.DS = bf.DS;
  a.DX0 = bf.DX0; a.DHATT = bf.DHATT; a.d_c0 = bf.d_c0;
  a.DXH = bf.DXH;
  a.b6_split = bf.DXH != nullptr && decoder_persist_b6_split(d) ? 1 : 0;
  if (a.b6_split) {          // astk_set_tuning("dec.b6_fused", 0): the d_x0 role of rounds 2-3 (kept for A/B runs)
    const int n5 = a.nbt * (a.H / 16), n1 = a.nbt * (a.A / 16), n2 = a.nbt * (2 * a.H / 32);
    const bool fits = d->n_layers == 1 ? n5 + 2 * n1 + n2 <= G : ((d->n_layers - 1) * n5 + 2 * n1 <= G && d->n_layers * n5 + n2 <= G);
    if (tune_on(TUNE_DEC_B6_FUSED) && fits) a.b6_split = 2;
  }
  ASTK_CHECK(d->n_layers == 1 || a.b6_split, "decoder_persist_bwd: the multi-layer role layout needs the split d_x0 buffers");
  a.ctr = bf.ctr;
  a.ab = abort_ctl(bf.ctr + (size_t)NPHASE_SLOTS * NSH * a.nbt * CTRS, PERSIST_DEC_BWD);
  a.tick_out = prof_tick_buffer(1);
  a.dbg = persist_dbg_env();
  {      // (a fill kernel, not hipMemsetAsync: the runtime's fill path left a 6 us gap in front of the launch)
    FillSegs f;
    f.n = 0;
    fill_seg_add(f, bf.ctr, ((size_t)NPHASE_SLOTS * NSH * a.nbt + 2 + a.B) * CTRS * sizeof(unsigned));
    if (bf.zero_ptr && bf.zero_bytes) fill_seg_add(f, bf.zero_ptr, bf.zero_bytes);      // (the caller's gradient arena: astk_decoder_desc.zero_ptr)
    if (bf.zero2_ptr && bf.zero2_bytes) fill_seg_add(f, bf.zero2_ptr, bf.zero2_bytes);   // (d_enc: decoder.hip adds both of its products into it)
    ASTK_TRY(fill_u32_segments(f, 0u, s));
  }
  const size_t shm = pdec_bwd_lds_floats(chunk, a.H) * sizeof(float);       // slices + dS/cvS/ds + ds/alpha/fold of the specialised scan
  {
    ProfScope prof(PROF_DEC_BWD, s);
    const bool special = pdec_special(a.H, chunk);
    const bool xs = special && chunk > PDEC_RES_ROWS;
    if (d->n_layers == 1) pdec_launch_bwd<1>(special, xs, shm, s, a);
    else if (d->n_layers == 2) pdec_launch_bwd<2>(special, xs, shm, s, a);
    else pdec_launch_bwd<3>(special, xs, shm, s, a);
  }
  ASTK_LAUNCH_CHECK();
  return 0;
}

int decoder_persist_fwd_launch(const astk_decoder_desc* d, const astk_decoder_params* prm, const float* enc, const int32_t* y,
                               const int32_t* ytgt, const int32_t* use_truth, const float* emb_mask, const float* rnn_masks, const DecPersistBuffers& bf,
                               float* loss, int32_t* pred_out, hipStream_t s) {
  int nsplit = 1, chunk = 1;
  // encA = enc . Wa (one batched GEMM): the score of decoder step s is encA[b,t,:].h_s + enc[b,t,:].ba, so the per-step
  // q = Wa h phase drops out of the sequential chain; Q itself is rebuilt after the loop for the backward.
  ASTK_TRY(gemm_launch(GEMM_NN, gemm_args(d->B * d->T, d->H, d->H, mat(enc, d->H), mat(prm->Wa, d->H), bf.ENCA, d->H), s));
  ASTK_CHECK(decoder_persist_applicable(d, &nsplit, &chunk), "decoder_persist: not applicable");
  PDecArgs a;
  memset(&a, 0, sizeof(a));
  a.B = d->B; a.S = d->L - 1; a.L = d->L; a.T = d->T; a.Tp = (d->T + 3) / 4 * 4; a.H = d->H; a.E = d->E; a.A = d->A; a.V = d->V;
  a.Vp = (d->V + 3) / 4 * 4; a.XI = d->E + d->A; a.nbt = (d->B + 15) / 16; a.nsplit = nsplit; a.chunk = chunk;
  a.ntile_v = (d->V + 15) / 16;
  a.inv_count = 1.f / (float)(d->loss_rows > 0 ? d->loss_rows : d->B);
  a.embed = prm->embed;
  for (int l = 0; l < d->n_layers; ++l) {
    a.Wu[l] = prm->lstm[l].Wu; a.bias[l] = prm->lstm[l].b; a.Wl[l] = prm->lstm[l].Wl;
    a.Gt[l] = bf.G[l]; a.Cst[l] = bf.C[l]; a.HR[l] = bf.HR[l]; a.HD[l] = bf.HD[l];
    a.rnn_mask[l] = rnn_masks ? rnn_masks + (size_t)l * (d->L - 1) * d->B * d->H : nullptr;
  }
  a.Wa = prm->Wa; a.ba = prm->ba; a.Wc = prm->Wc; a.bc = prm->bc; a.Wo = prm->Wo; a.bo = prm->bo; a.cw = prm->class_weight;
  a.enc = enc; a.encA = bf.ENCA; a.y = y; a.ytgt = ytgt ? ytgt : y; a.use_truth = use_truth; a.emb_mask = emb_mask;
  a.TOK = bf.TOK; a.PRED = bf.PRED; a.X0 = bf.X0; a.Q = bf.Q; a.ALPHA = bf.ALPHA;
  a.CVH = bf.CVH; a.HT = bf.HT; a.LOGITS = bf.LOGITS; a.LOSSROWS = bf.LOSSROWS; a.LSE = bf.LSE; a.PART = bf.PART; a.CESTAT = bf.CESTAT; a.ML = bf.ML;
  a.ctr = bf.ctr;
  a.ab = abort_ctl(bf.ctr + (size_t)NPHASE_SLOTS * NSH * a.nbt * CTRS, PERSIST_DEC_FWD);
  a.dbg = persist_dbg_env();
  a.tick_out = prof_tick_buffer(0);
#if ASTK_PDEC_SENT_H
  {
    FillSegs f;
    f.n = 0;
    fill_seg_add(f, bf.ctr, ((size_t)NPHASE_SLOTS * NSH * a.nbt + 2 + a.B) * CTRS * sizeof(unsigned), 0u);
    fill_seg_add(f, bf.CVH, (size_t)a.S * a.B * 2 * a.H * sizeof(float));
#if ASTK_PDEC_SENT_HD
    for (int l = 0; l + 1 < d->n_layers; ++l) fill_seg_add(f, bf.HD[l], (size_t)a.S * a.B * a.H * sizeof(float));      // (the launch's value: the sentinel)
#endif
    if (bf.zero_a) fill_seg_add(f, bf.zero_a, bf.zero_a_bytes, 0u);
    if (bf.zero_b) fill_seg_add(f, bf.zero_b, bf.zero_b_bytes, 0u);
    if (bf.c0 && bf.h0) {
      const size_t bh = (size_t)d->B * d->H;
      ASTK_CHECK(f.n + 2 * d->n_layers <= FILL_SEG_MAX, "decoder_persist: too many fill segments");
      for (int l = 0; l < d->n_layers; ++l) {
        fill_seg_add_copy(f, bf.C[l], bf.c0 + l * bh, bh * sizeof(float));
        fill_seg_add_copy(f, bf.HR[l], bf.h0 + l * bh, bh * sizeof(float));
      }
    }
    ASTK_TRY(fill_u32_segments(f, PDEC_SENTINEL, s));
  }
#else
  if (bf.c0 && bf.h0) {
    CopySegs cp;
    cp.n = 0;
    const size_t bh = (size_t)d->B * d->H;
    for (int l = 0; l < d->n_layers; ++l) {
      copy_seg_add(cp, bf.C[l], bf.c0 + l * bh, bh * sizeof(float));
      copy_seg_add(cp, bf.HR[l], bf.h0 + l * bh, bh * sizeof(float));
    }
    ASTK_TRY(copy_segments(cp, s));
  }
  ASTK_HIP(hipMemsetAsync(bf.ctr, 0, ((size_t)NPHASE_SLOTS * NSH * a.nbt + 2 + a.B) * CTRS * sizeof(unsigned), s));
  if (bf.zero_a) ASTK_HIP(hipMemsetAsync(bf.zero_a, 0, bf.zero_a_bytes, s));
  if (bf.zero_b) ASTK_HIP(hipMemsetAsync(bf.zero_b, 0, bf.zero_b_bytes, s));
#endif
  const size_t shm = pdec_lds_floats(chunk, a.H, nsplit) * sizeof(float);
  {
    ProfScope prof(PROF_DEC_FWD, s);
    const bool special = pdec_special(a.H, chunk);
    const bool xs = special && chunk > PDEC_RES_ROWS;
    if (d->n_layers == 1) pdec_launch_fwd<1>(special, xs, shm, s, a);
    else if (d->n_layers == 2) pdec_launch_fwd<2>(special, xs, shm, s, a);
    else pdec_launch_fwd<3>(special, xs, shm, s, a);
  }
  ASTK_LAUNCH_CHECK();
  // Q[s][b][:] = Wa h_top + ba for all steps (needed by the backward's deferred d_enc product)
  ASTK_TRY(gemm_launch(GEMM_NT, gemm_args(a.S * a.B, a.H, a.H, mat(bf.CVH + a.H, 2 * a.H), mat(prm->Wa, a.H), bf.Q, a.H, prm->ba), s));
  hipLaunchKernelGGL(k_decoder_post, dim3(a.S * a.B), dim3(256), 0, s, prm->embed, y, ytgt ? ytgt : y, use_truth, bf.PRED, emb_mask, bf.TOK, bf.X0, bf.LOGITS, bf.LSE,
                     prm->class_weight, bf.LOSSROWS, loss, pred_out, a.S, a.B, a.L, a.E, a.XI, a.V, a.Vp, a.inv_count, persist_status_word(), d->status_dst);
  ASTK_LAUNCH_CHECK();
  return 0;
}

}  // namespace astk
