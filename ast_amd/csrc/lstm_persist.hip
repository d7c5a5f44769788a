// Persistent, wavefront-scheduled encoder LSTM kernels ("hard part" of section 7).
//
// One launch runs ALL time steps of ALL (direction, layer) cells.  A workgroup owns 16 hidden units x 16 batch
// rows of one cell for the whole sequence; its weight slices are loaded ONCE into VGPRs as f32-MFMA B fragments and stay
// there for all T steps; the cell state c (forward) / dc (backward) lives in a register of the thread that owns
// (batch row, unit).  Layers run as a wavefront: cell (l, t) starts as soon as (l, t-1) and (l-1, t) are there, so the
// sequential depth is T + O(n_layers) steps instead of n_layers * T launches.
//
// Two kernels, two hand-off protocols (DESIGN.md section 4 has the measurements behind the choice):
//   lstm_persist_fwd_g   forward.  Gather: per step a workgroup needs the 16 x h activations of the previous step and of the
//                        layer below (4 KB per wave).  Tag-free SENTINEL hand-off: the saved-activation buffers are filled
//                        with 0xFFFFFFFF before the launch, producers store each value once (sc1), consumers re-read until no
//                        word is the sentinel -- no counter, drain, atomic or barrier on the chain.
//   lstm_persist_bwd_rs  backward.  Reduce-scatter: the owner of 64 gate columns of dz multiplies them with its rows of W_l
//                        (and W_u for the layer below) and hands out 16x16 partial tiles.  Inside a cell (the recurrence's chain)
//                        the partial tiles are a 4-deep SENTINEL ring: a consumer polls its 16 KB themselves and puts the
//                        sentinel back behind the step's barrier.  To the layer below (which lags): COUNTER hand-off (R1 of
//                        cdna_hip_programming.md Guideline 16: write-through stores, every storing wave drains vmcnt,
//                        barrier, ONE lane adds to the arrival counter on its own 256-byte line; the consumer asks one step ahead
//                        with ONE lane and fetches the partials a step early).  Counters are zeroed before every launch.
// Every spin is bounded: on time-out a workgroup raises the abort word, every poll loop checks it, and the grid drains.
// Residency: the launcher only uses this path when the whole grid fits one workgroup per CU (<= 256 workgroups).
#include "common.h"
#include <type_traits>

namespace astk {

namespace {

typedef unsigned long long u64;
// In-kernel instrumentation (phase timers, bit 8; the dawdling slice of the last-arrival regression test, bit 16) exists only in the
// test-hook build (libastk_test.so, -DASTK_TEST_HOOKS): there ASTK_PERSIST_DBG is read at every launch; the product library's kernels see
// the constant 0 and carry none of it.
#ifdef ASTK_TEST_HOOKS
static int persist_dbg_env() { const char* e = getenv("ASTK_PERSIST_DBG"); return e ? atoi(e) : 0; }
#define PERSIST_DBG(a) ((a).dbg)
#else
static int persist_dbg_env() { return 0; }
#define PERSIST_DBG(a) 0
#endif
constexpr int CTR_STRIDE = 64;   // arrival counters live 256 bytes apart: pollers of different cells never share a line

struct PCellF {
  const float* Wl;      // (4h, h)
  const float* Wu;      // (4h, h) layers >= 1, else null
  const float* bias;    // (4h) layers >= 1 (layer 0: already inside zx)
  const float* zx;      // layer 0: (T,B,4h) upward projection incl. bias
  float* gates;         // (T,B,4h)
  float* C;             // (T,B,h)
  float* HR;            // (T,B,h) raw h
  float* HD;            // (T,B,h) dropped output (null: next layer reads HR)
  const float* xin;     // layers >= 1: HD or HR of the layer below
  const float* mask;    // (T,B,h) or null
  float* enc;           // top layer: enc_states + dir*h ; row stride enc_ldb per batch row, enc_ldt per position
  int reverse_pos;      // top layer of direction 1: position = T-1-t
  int layer;
  // layer 0 with its input projection produced in time chunks on a side stream (astk_lstm_stack_desc.side_stream): steps [0, zx_s0) are
  // there when the launch starts; chunk k >= 0 = steps [zx_s0 + k zx_cs, zx_s0 + (k+1) zx_cs) is there when zx_flags[k * CTR_STRIDE] != 0
  // (one word per chunk on its own line, zeroed before the side stream starts, set by a one-lane kernel behind the chunk's product).  null: all there.
  const unsigned* zx_flags;
  int zx_s0, zx_cs;
};
struct PFwdArgs {
  PCellF c[16];
  int ncells, nl, T, B, h, H;
  int dbg;              // timing experiments only (ASTK_PERSIST_DBG): 1 = skip payload loads + MFMA, 2 = skip publish drain
  unsigned* done;       // [ncells][nbt] arrival counters
  AbortCtl ab;
};

struct PCellB {
  const float* Wl;      // (4h, h): the recurrent weight as the parameters hold it
  float* gates_dz;      // (T,B,4h): activated gates in, dz out (in place; the batched products after the launch read it)
  const float* C;       // (T,B,h)
  const float* mask;    // (T,B,h) or null
  const float* d_enc;   // gradient wrt this cell's (dropped-out) output that does not come as partial tiles: the top layer's slice of
                        // d_enc_states (+ dir*h), or -- hoisted form -- the dense (T,B,h) product dz W_u of the layer above; element
                        // (row b, position p, unit u) at d_enc[b * dy_sb + p * dy_st + u]
  long dy_sb, dy_st;
  const float* d_hT;    // (B,h) or null
  const float* d_cT;    // (B,h) or null
  // reduce-scatter path (lstm_persist_bwd_rs):
  const float* Wu;      // (4h, h): upward weight of THIS cell (layers >= 1), for the partials handed DOWN
  float* PR;            // ring [PR_RING][nbt][nslice consumer][nslice producer][16 col][16 row]: partial dh_rec of this cell
  float* PD;            // [T][nbt][nslice consumer][nslice producer][16][16]: partial dx handed to the layer below (null: layer 0)
  const float* PD_up;   // PD of the layer above (null: top layer)
  int up_external;      // PD_up was completed by an EARLIER launch (layer groups): read it without waiting on a counter of this launch
  int reverse_pos;
  int layer;
  u64* amax;            // 16 sharded words for max |dz| of this cell (the fp16x2 GEMMs' operand scale, gemm_amax_reserve), or null
  float* db;            // (4h) bias gradient, += the column sums of dz over all steps and rows (null: the caller sums dz itself)
  float* db_part;       // deterministic calls: [workgroup rows][4h] scratch that receives every row's sums instead of the atomics into db (null: atomics)
  // layer 0 with side-stream consumers of dz (the time-chunked input / weight gradient products): dz goes out written through, and every
  // workgroup of the cell arrives once on *prog per chunk of prog_cs steps (>= 4) -- chunk k = loop steps [k prog_cs, (k+1) prog_cs) counted from
  // the END of the sequence, complete when *prog >= (k+1) x workgroups of the cell.  null: dz is read behind the launch only.
  unsigned* prog;
  int prog_cs;
};
struct PBwdArgs {
  PCellB c[16];
  int ncells, nl, T, B, h, H;
  int dbg;
  unsigned amax_gen;    // generation tag of the maxima (high half of the words)
  unsigned* done;
  AbortCtl ab;
};

__device__ __forceinline__ unsigned ld_flag(const unsigned* p) {
  return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void st4_sc1(float* p, float v) {
  __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// One lane waits until *ctr >= target (or the abort word is raised).  Returns false on abort / time-out.
__device__ __forceinline__ bool wait_ge(const unsigned* ctr, unsigned target, const AbortCtl& ab) {
  unsigned spins = 0;
  while (ld_flag(ctr) < target) {
    if (++spins > ab.limit) {   // ~seconds: something is wrong (grid not resident); drain instead of hanging
      abort_raise(ab);
      return false;
    }
    if ((spins & 63u) == 0 && abort_seen(ab)) return false;
  }
  return true;
}

// publish: every storing wave drains its stores, the workgroup barriers, one lane bumps the arrival counter
// (tid: the caller's thread index inside its -- possibly virtual, see DUO -- workgroup)
__device__ __forceinline__ void publish(unsigned* ctr, int tid) {
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (tid == 0) __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// Gate activations of the persistent kernels' epilogues (on the recurrence's critical path): v_exp_f32 / v_rcp_f32 based,
// absolute error <= ~2e-7 (libdevice's tanhf/expf with full-precision division cost ~0.25 us more per step).
__device__ __forceinline__ float sigm_fast(float x) { return __builtin_amdgcn_rcpf(1.f + __expf(-x)); }
__device__ __forceinline__ float tanh_fast(float x) { return 2.f * __builtin_amdgcn_rcpf(1.f + __expf(-2.f * x)) - 1.f; }

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
// raw buffer descriptor over a hand-off buffer: lets the compiler track 16-byte sc1 loads / stores itself
__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const void* p) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, 0x7fffffff, 0x00020000);
}

#define MFMA4(ACC, A4, W4)                                                   \
  ACC = __builtin_amdgcn_mfma_f32_16x16x4f32((A4).x, (W4).x, ACC, 0, 0, 0);  \
  ACC = __builtin_amdgcn_mfma_f32_16x16x4f32((A4).y, (W4).y, ACC, 0, 0, 0);  \
  ACC = __builtin_amdgcn_mfma_f32_16x16x4f32((A4).z, (W4).z, ACC, 0, 0, 0);  \
  ACC = __builtin_amdgcn_mfma_f32_16x16x4f32((A4).w, (W4).w, ACC, 0, 0, 0);

// four independent accumulators (the gates) interleaved so that no MFMA waits on the previous one's result
#define MFMA4G(ACC, A4, W)                                                                    \
  _Pragma("unroll") for (int g_ = 0; g_ < 4; ++g_) ACC[g_] = __builtin_amdgcn_mfma_f32_16x16x4f32((A4).x, (W)[g_].x, ACC[g_], 0, 0, 0); \
  _Pragma("unroll") for (int g_ = 0; g_ < 4; ++g_) ACC[g_] = __builtin_amdgcn_mfma_f32_16x16x4f32((A4).y, (W)[g_].y, ACC[g_], 0, 0, 0); \
  _Pragma("unroll") for (int g_ = 0; g_ < 4; ++g_) ACC[g_] = __builtin_amdgcn_mfma_f32_16x16x4f32((A4).z, (W)[g_].z, ACC[g_], 0, 0, 0); \
  _Pragma("unroll") for (int g_ = 0; g_ < 4; ++g_) ACC[g_] = __builtin_amdgcn_mfma_f32_16x16x4f32((A4).w, (W)[g_].w, ACC[g_], 0, 0, 0);

// fp16x2 form of the recurrences' products (template parameter X2): the weights a workgroup keeps in registers are split ONCE
// into fp16 hi / lo fragments behind a per-workgroup power-of-two scale, the 16 x K activation fragments of a step are split when they
// arrive (scale 2^10: |h| < 1 and a dropped-out input is at most 1 / (1 - p)), and 32 k of a product are three
// v_mfma_f32_16x16x32_f16 of 16 cycles instead of eight v_mfma_f32_16x16x4_f32 of 32: the matrix part of a step's critical path drops
// from 0.93 us to 0.18 us at the same 2^-22 product accuracy as the batched GEMMs (gemm.hip).
// bf16x3 form (XS = 3): the same structure with THREE bf16 terms per value and six v_mfma_f32_16x16x32_bf16 per 32 k (common.h:
// split8b / MFMA32B): every f32 weight and activation is represented exactly (no scales: bf16 has f32's exponent range), a product is
// exact to 2^-26 -- at least the accuracy of the f32 MFMA chain -- at 6 x 16 cycles per 32 k instead of 8 x 32.  The resident weights take
// 1.5 x the registers of the f32 fragments (192 per lane at h = 256): shapes whose slices do not fit (h = 512) keep the f32 MFMAs.
// XS (0 f32 MFMAs, 2 fp16x2, 3 bf16x3) is a template parameter of both kernels: the launchers pick it from the arithmetic in force for the call
// (descriptor precision, else the process default), so that one process can time the step under every arithmetic.
constexpr float ACT_SCALE = 1024.f, ACT_SCALE_INV = 1.f / 1024.f;
// four independent accumulators (the gates) interleaved, three term products of 32 k each
#define MFMA32HG(ACC, A, W)                                                                                                     \
  _Pragma("unroll") for (int g_ = 0; g_ < 4; ++g_)                                                                              \
    ACC[g_] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(h16x8, (A).lo), __builtin_bit_cast(h16x8, (W)[g_].hi), ACC[g_], 0, 0, 0); \
  _Pragma("unroll") for (int g_ = 0; g_ < 4; ++g_)                                                                              \
    ACC[g_] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(h16x8, (A).hi), __builtin_bit_cast(h16x8, (W)[g_].lo), ACC[g_], 0, 0, 0); \
  _Pragma("unroll") for (int g_ = 0; g_ < 4; ++g_)                                                                              \
    ACC[g_] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(h16x8, (A).hi), __builtin_bit_cast(h16x8, (W)[g_].hi), ACC[g_], 0, 0, 0);
// bf16x3: four independent accumulators (the gates) interleaved, six term products of 32 k each (smallest first)
#define MFMA32BG_T_(ACC, AP, W, WP) _Pragma("unroll") for (int g_ = 0; g_ < 4; ++g_) { MFMA_B16_(ACC[g_], AP, (W)[g_].WP) }
#define MFMA32BG(ACC, A, W)          \
  MFMA32BG_T_(ACC, (A).lo, W, hi)    \
  MFMA32BG_T_(ACC, (A).hi, W, lo)    \
  MFMA32BG_T_(ACC, (A).mid, W, mid)  \
  MFMA32BG_T_(ACC, (A).mid, W, hi)   \
  MFMA32BG_T_(ACC, (A).hi, W, mid)   \
  MFMA32BG_T_(ACC, (A).hi, W, hi)
// fragment type and helpers of a split scheme XS (2: fp16 hi / lo behind a scale, 3: bf16 hi / mid / lo)
// XS = 4 forms of the two macros above: W holds hi / mid, LOP points at this thread's lo fragments in LDS (gate g at LOP[g * 256])
#define MFMA32BG4_T_(ACC, AP, WEXPR) _Pragma("unroll") for (int g_ = 0; g_ < 4; ++g_) { MFMA_B16_(ACC[g_], AP, WEXPR) }
#define MFMA32BG4(ACC, A, W, LOP)              \
  {                                            \
    u32q lo_[4];                               \
    _Pragma("unroll") for (int g_ = 0; g_ < 4; ++g_) lo_[g_] = (LOP)[g_ * 256]; \
    MFMA32BG4_T_(ACC, (A).lo, (W)[g_].hi)      \
    MFMA32BG4_T_(ACC, (A).hi, lo_[g_])         \
    MFMA32BG4_T_(ACC, (A).mid, (W)[g_].mid)    \
    MFMA32BG4_T_(ACC, (A).mid, (W)[g_].hi)     \
    MFMA32BG4_T_(ACC, (A).hi, (W)[g_].mid)     \
    MFMA32BG4_T_(ACC, (A).hi, (W)[g_].hi)      \
  }
#define MFMA32B4(ACC, A, W, LOV)   \
  MFMA_B16_(ACC, (A).lo, (W).hi)  \
  MFMA_B16_(ACC, (A).hi, LOV)     \
  MFMA_B16_(ACC, (A).mid, (W).mid) \
  MFMA_B16_(ACC, (A).mid, (W).hi) \
  MFMA_B16_(ACC, (A).hi, (W).mid) \
  MFMA_B16_(ACC, (A).hi, (W).hi)
template <int XS> struct FragOf { typedef HL8 type; };
template <> struct FragOf<3> { typedef HML8 type; };
template <> struct FragOf<4> { typedef HML8 type; };
// XS = 4: bf16x3 with the LO plane of the resident weight fragments in LDS.  At h = 512 (and for the one product of the hoisted h = 1024
// form) the three bf16 planes of a workgroup's weights are 384 KB: 256 KB (hi, mid) stay in registers -- what the f32 fragments took --
// and the lo plane, which enters ONE of the six term products, lives in 128 KB of LDS, every thread reading back its own 16 bytes per
// fragment (no conflicts, no barrier).  Same six products as XS = 3; the f32-input MFMAs these shapes ran under the default arithmetic
// were 3.9 us of matrix time per step against 1.5.
struct HM8 { u32q hi, mid; };
template <int XS> struct WFragOf { typedef typename FragOf<XS>::type type; };
template <> struct WFragOf<4> { typedef HM8 type; };
constexpr int LO_LDS_FRAGS = 32;          // weight fragments per thread whose lo plane lives in LDS (32 x 256 threads x 16 B = 128 KB)
template <int XS>
__device__ __forceinline__ typename FragOf<XS>::type split_frag(const float4& a, const float4& b, float scl) {
  if constexpr (XS == 3 || XS == 4) return split8b(a, b);
  else return split8(a, b, scl);
}
// workgroup-wide maximum of a per-thread value (256 threads; `red` = 4 floats of LDS scratch; ends with a barrier)
__device__ __forceinline__ float wg_max(float m, float* red) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = m;
  __syncthreads();
  m = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
  __syncthreads();
  return m;
}
// maximum over the 64 lanes of a wave of a NON-NEGATIVE value, in every lane: five v_max_f32 with DPP row / bank operands and one
// cross-half readlane (a ds_bpermute butterfly costs an LDS round trip per step)
__device__ __forceinline__ float wave_max_nonneg(float m) {
  int v = __float_as_int(m);
#define DPP_MAX_STEP(ctrl) v = __float_as_int(fmaxf(__int_as_float(v), __int_as_float(__builtin_amdgcn_update_dpp(0, v, ctrl, 0xf, 0xf, true))));
  DPP_MAX_STEP(0x111)   // row_shr:1
  DPP_MAX_STEP(0x112)   // row_shr:2
  DPP_MAX_STEP(0x114)   // row_shr:4
  DPP_MAX_STEP(0x118)   // row_shr:8   -> lane 15 of every row of 16 holds the row's maximum
#undef DPP_MAX_STEP
  const float r0 = __int_as_float(__builtin_amdgcn_readlane(v, 15)), r1 = __int_as_float(__builtin_amdgcn_readlane(v, 31));
  const float r2 = __int_as_float(__builtin_amdgcn_readlane(v, 47)), r3 = __int_as_float(__builtin_amdgcn_readlane(v, 63));
  return fmaxf(fmaxf(r0, r1), fmaxf(r2, r3));
}
__device__ __forceinline__ float amax4f(float m, const float4& v) { return fmaxf(fmaxf(m, fmaxf(fabsf(v.x), fabsf(v.y))), fmaxf(fabsf(v.z), fabsf(v.w))); }

// ------------------------------------------------------------------ forward, sentinel hand-off
// ---- tag-free hand-off: the data is the flag.
// A hand-off buffer is filled with SENTINEL words (0xFFFFFFFF, a NaN pattern no finite activation and no arithmetic NaN
// has) by a memset node before every launch; producers store each value ONCE, write-through (sc1), every value to its own
// (step, row, column) slot; a consuming WAVE loads exactly the fragments its MFMAs need (sc1) and re-reads them until no
// word is the sentinel.  4-byte stores are single-copy atomic, nothing is reused within a launch, so no tag, no flag, no
// drain, no barrier is needed, and the buffers double as the saved activations the later launches read.
constexpr unsigned SENTINEL = 0xffffffffu;
#ifndef ASTK_FRAG_WAIT_SWEEP
#define ASTK_FRAG_WAIT_SWEEP 3
#endif
template <int NB>
__device__ __forceinline__ void frag_issue(__amdgpu_buffer_rsrc_t rs, int byte_off, int wave, u32x4 (&g)[NB]) {
#pragma unroll
  for (int i = 0; i < NB; ++i) g[i] = __builtin_amdgcn_raw_buffer_load_b128(rs, byte_off + 64 * (wave + 4 * i), 0, 16);   // aux 16 = sc1
}
template <int NB>
__device__ __forceinline__ bool frag_ok(const u32x4 (&g)[NB]) {
  // (the sentinel is the largest unsigned word: one running maximum and ONE compare -- a chain of && compiles to a branch per word,
  //  0.25 us per check on the recurrence's critical path)
  unsigned m = 0;
#pragma unroll
  for (int i = 0; i < NB; ++i) m = max(max(m, g[i].x), max(max(g[i].y, g[i].z), g[i].w));
  return __all(m != SENTINEL);
}
// Slow path: some fragment was not there yet.  Poll ONE fragment per lane until it is complete (waiting waves must not
// flood the fabric with full sweeps), then re-read everything; repeat until complete.  Bounded; a time-out raises the
// abort word, which every other spin checks, and the grid drains.
template <int NB>
__device__ __forceinline__ void frag_wait(__amdgpu_buffer_rsrc_t rs, int byte_off, int wave, u32x4 (&g)[NB], bool& dead, const AbortCtl& ab) {
  unsigned spins = 0;
  while (!dead) {
#if ASTK_FRAG_WAIT_SWEEP
    // the first retries are whole sweeps: a miss then costs one more round trip, not two (poll, then re-read); a producer that is
    // really late is polled with one fragment per lane as before
    if (spins < ASTK_FRAG_WAIT_SWEEP) {
      frag_issue<NB>(rs, byte_off, wave, g);
      if (frag_ok<NB>(g)) return;
      ++spins;
      continue;
    }
#endif
    const u32x4 c = __builtin_amdgcn_raw_buffer_load_b128(rs, byte_off + 64 * wave, 0, 16);
    if (__all((c.x != SENTINEL) & (c.y != SENTINEL) & (c.z != SENTINEL) & (c.w != SENTINEL))) {
      frag_issue<NB>(rs, byte_off, wave, g);
      if (frag_ok<NB>(g)) return;
    }
    if (++spins > (ab.limit >> 1)) { abort_raise(ab); dead = true; }
    else if ((spins & 63u) == 0 && abort_seen(ab)) dead = true;
  }
}
__device__ __forceinline__ float4 frag_vals(const u32x4& g) {
  return make_float4(__uint_as_float(g.x), __uint_as_float(g.y), __uint_as_float(g.z), __uint_as_float(g.w));
}

// Schedule of one step t (wave view).  vmcnt retires in issue order and the compiler cannot count memory operations
// across branches, so every wait in the loop is in effect "everything issued so far"; the schedule is therefore built so
// that at the step's ONE wait point W_t only the young operations of the recurrence's critical path are pending:
//   top:   issue the h_{t-1} fragment loads (this step's hand-off stores of h_{t-1} went out just before)
//          upward MFMAs of step t from REGISTERS (ax: fetched at W_{t-1})
//   W_t:   h fragments complete? (slow path: poll)   x_{t+1} fragments complete? (issued a whole step ago) -> ax
//   after: off-path traffic -- prefetch x_{t+2}, zx/mask of step t+1, the saved gates/cell state of step t-1
//          recurrent MFMAs -> LDS reduction (one barrier) -> gates -> hand-off stores of h_t (HR, and HD for the layer above)
// The layer above therefore trails the layer below by two steps.
// Code shape: everything a wait depends on is UNCONDITIONAL inside the loop (HAS_UP is a template parameter, step 0 is
// peeled, prefetch indices are clamped instead of guarded): a conditionally issued load becomes a phi of "old registers /
// load result", and hipcc then copies the result right behind the load, i.e. waits for it at the point of issue.
// MT = 16-row batch tiles per workgroup (1 or 2).  MT = 2 (round 6): the workgroup's resident weight fragments serve BOTH tiles -- two sets
// of activation fragments and accumulators against the same B operands -- so a batch of 32 rows takes 96 workgroups instead of 192 (160 CUs
// free for the time-chunked layer-0 products on the side stream), batch 64 runs as ONE launch and the 6-layer stacks need half the launches.
// The hand-off protocol is unchanged (every wave polls the fragments of both tiles); the matrix part and the gate epilogue of a step double.
// DUO (round 6; bf16x3 with the lo plane in LDS, h <= 256): a 512-thread workgroup = TWO virtual 256-thread workgroups of the 16-row form, one per
// batch tile (virtual workgroup row 2 blockIdx.y + half), two waves per SIMD.  Each half keeps its own weight fragments (hi / mid in 128
// registers: a wave has 256) and shares the lo plane in LDS (the same weights: both halves write and read identical words); every hand-off,
// counter and buffer is the 16-row form's, per half; the only coupling is the workgroup barrier.  A step's WORK is ~60 % of its time
// (MT above): here the two tiles' work runs on the SAME SIMDs from different waves, so one tile's splits and gate epilogue (vector ALU) overlap
// the other's MFMAs (matrix pipe) and its hand-off wait -- which one wave doing both tiles in turn (MT = 2) cannot.
template <int KB, bool HAS_UP, int XS, int MT, bool DUO = false>
__device__ __forceinline__ void lstm_fwd_steps(const PFwdArgs& a, const PCellF& c, float* red0, float* red1, u32q* lo_lds) {
  static_assert(!DUO || (MT == 1 && XS == 4), "DUO: 16-row halves on the bf16x3 form with the lo plane in LDS");
  const int tid = DUO ? (int)(threadIdx.x & 255) : (int)threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r = lane & 15, q = lane >> 4;
  const int bt = DUO ? (int)(blockIdx.y * 2 + (threadIdx.x >> 8)) : (int)blockIdx.y, j0 = blockIdx.x * 16;
  const int T = a.T, B = a.B, h = a.h, HH = a.H, dbg = PERSIST_DBG(a);
  const AbortCtl ab = a.ab;       // (locals: see the note on the kernel-argument block in lstm_persist_fwd_g)
  const int m0 = bt * 16 * MT;
  bool dead = false;

  auto wl_at = [&](int i, int g) { return *reinterpret_cast<const float4*>(c.Wl + (long)(4 * (j0 + r) + g) * h + 16 * (wave + 4 * i) + 4 * q); };
  auto wu_at = [&](int i, int g) { return *reinterpret_cast<const float4*>(c.Wu + (long)(4 * (j0 + r) + g) * h + 16 * (wave + 4 * i) + 4 * q); };
  // X2: the resident weights as fp16 hi / lo fragments of v_mfma_f32_16x16x32_f16: 16-k blocks i, i+1 of this wave form one 32-k operand
  // (the k order inside a product is free as long as activations and weights agree); one scale for the workgroup's whole slice.
  // Two sweeps over the slice (maximum, then split): holding the f32 values and their fragments at once would take 512 registers.
  constexpr int NPR = (KB + 1) / 2;
  constexpr bool X2 = XS != 0;          // a split scheme (16-bit MFMAs); XS == 2 additionally scales
  typedef typename FragOf<XS>::type Frag;
  const float4 zero4 = make_float4(0.f, 0.f, 0.f, 0.f);
  float zscale = 1.f;
  typedef typename WFragOf<XS>::type WFrag;
  static_assert(XS != 4 || (HAS_UP ? 2 : 1) * NPR * 4 <= LO_LDS_FRAGS, "lo plane does not fit its LDS region");
  u32q* const lo_l = lo_lds + tid, * const lo_u = lo_lds + NPR * 4 * 256 + tid;      // this thread's lo fragments: lateral [p][g], upward behind them
  WFrag wlh[X2 ? NPR : 1][4], wuh[(X2 && HAS_UP) ? NPR : 1][4];
  float4 wl[X2 ? 1 : KB][4], wu[(!X2 && HAS_UP) ? KB : 1][4];
  if constexpr (X2) {
    float wscl = 1.f;
    if constexpr (XS == 2) {
      float m = 0.f;
#pragma unroll
      for (int i = 0; i < KB; ++i)
#pragma unroll
        for (int g = 0; g < 4; ++g) { m = amax4f(m, wl_at(i, g)); if (HAS_UP) m = amax4f(m, wu_at(i, g)); }
      float winv;
      wscl = pow2_scale_for(wg_max(m, red0), winv);
      zscale = winv * ACT_SCALE_INV;
    }
#pragma unroll
    for (int p = 0; p < NPR; ++p)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        if constexpr (XS == 4) {
          const HML8 t = split8b(wl_at(2 * p, g), 2 * p + 1 < KB ? wl_at(2 * p + 1, g) : zero4);
          wlh[p][g].hi = t.hi; wlh[p][g].mid = t.mid;
          lo_l[(p * 4 + g) * 256] = t.lo;
          if constexpr (HAS_UP) {
            const HML8 u = split8b(wu_at(2 * p, g), 2 * p + 1 < KB ? wu_at(2 * p + 1, g) : zero4);
            wuh[p][g].hi = u.hi; wuh[p][g].mid = u.mid;
            lo_u[(p * 4 + g) * 256] = u.lo;
          }
        } else {
        wlh[p][g] = split_frag<XS>(wl_at(2 * p, g), 2 * p + 1 < KB ? wl_at(2 * p + 1, g) : zero4, wscl);
        if constexpr (HAS_UP) wuh[p][g] = split_frag<XS>(wu_at(2 * p, g), 2 * p + 1 < KB ? wu_at(2 * p + 1, g) : zero4, wscl);
        }
      }
  } else {
#pragma unroll
    for (int i = 0; i < KB; ++i)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        wl[i][g] = wl_at(i, g);
        if constexpr (HAS_UP) wu[i][g] = wu_at(i, g);
      }
  }
  const __amdgpu_buffer_rsrc_t r_own = make_rsrc(c.HR);
  const __amdgpu_buffer_rsrc_t r_below = make_rsrc(HAS_UP ? c.xin : c.HR);
  // layer 0: the input projection of all time steps comes from a batched product; with time-chunked products on a side stream
  // (astk_lstm_stack_desc.side_stream) the rows of step t exist once the flag of t's chunk is up -- read with sc1 loads behind the flag (the
  // producing launch ended, i.e. its stores are in memory, before the flag kernel behind it ran)
  const __amdgpu_buffer_rsrc_t r_zx = make_rsrc(HAS_UP ? (const float*)c.HR : c.zx);
  const unsigned* const zflags = HAS_UP ? nullptr : c.zx_flags;
  const int zs0 = c.zx_s0, zcs = max(c.zx_cs, 1);
  const int eu = j0 + (tid & 15);                            // epilogue ownership: unit eu, batch rows eb[mt]
  int frag0[MT], eb[MT];
  long ebc[MT];
  bool evalid[MT];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt) {
    const int arow = min(m0 + 16 * mt + r, B - 1);
    frag0[mt] = (arow * h + 4 * q) * 4;   // byte offset of this lane's fragment row at step 0
    eb[mt] = m0 + 16 * mt + (tid >> 4);
    evalid[mt] = eb[mt] < B;
    ebc[mt] = evalid[mt] ? eb[mt] : 0;
  }
  float4 bias4 = make_float4(0.f, 0.f, 0.f, 0.f);
  if (HAS_UP) bias4 = *reinterpret_cast<const float4*>(c.bias + 4 * eu);
  const bool use_mask = c.mask != nullptr;
  const float* maskp = use_mask ? c.mask : c.C;   // always a readable (T,B,h) buffer: the mask load is unconditional
  float c_state[MT];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt) c_state[mt] = 0.f;
  const int step_bytes = B * h * 4;
  u32x4 gx[MT][KB];
  float4 ax[MT][KB];
  Frag axh[MT][X2 ? NPR : 1];
  auto take_x = [&]() {
    if constexpr (X2) {
#pragma unroll
      for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int p = 0; p < NPR; ++p) axh[mt][p] = split_frag<XS>(ax[mt][2 * p], 2 * p + 1 < KB ? ax[mt][2 * p + 1] : zero4, ACT_SCALE);
    }
  };
  auto zx_load = [&](long tbs) {      // this thread's four pre-activations of (step, row) tbs: sc1 (see r_zx)
    return frag_vals(__builtin_amdgcn_raw_buffer_load_b128(r_zx, (int)((tbs * 4 * h + 4 * eu) * 4), 0, 16));
  };
  // Chunk flags (layer 0 with side-stream chunks only).  z_next = first step of the next chunk this wave has not entered yet, zfp = that
  // chunk's flag word (one per 256-byte line; the word behind the last chunk's stays 0 and is never waited for).  The word is LOOKED AT
  // every step (one uniform load, off the chain), so when the chunk's first rows are wanted the answer is at least a step old; entering a
  // chunk = (wait for the flag if it was not up yet) + ONE agent-scope acquire: the chunk was written with plain stores by a launch that
  // ended before the flag kernel ran, so its bytes are in memory, and the acquire keeps this CU from answering the loads below out of lines
  // it may still hold from an earlier launch (MI355X_MICROARCH.md "Consumer, always").
  int z_next = zflags ? zs0 : 0x7fffffff;
  const unsigned* zfp = zflags;
  unsigned zf_seen = 1u;
  auto zflag_wait = [&](const unsigned* fp, unsigned seen) {
    unsigned spins = 0;
    while (seen == 0u && !dead) {
      seen = ld_flag(fp);
      if (++spins > (ab.limit >> 1)) { abort_raise(ab); dead = true; }
      else if ((spins & 63u) == 0 && abort_seen(ab)) dead = true;
    }
  };
  auto zx_enter = [&](int ts) {              // (uniform) the rows of step ts are about to be loaded
    while (ts >= z_next) {
      if (zf_seen == 0u) zflag_wait(zfp, 0u);
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
      z_next += zcs; zfp += CTR_STRIDE;
      zf_seen = ld_flag(zfp);
    }
  };
  // what only later launches read, stored half a step late
  float4 p_gates[MT];
  float p_hd[MT], p_c[MT];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt) { p_gates[mt] = make_float4(0.f, 0.f, 0.f, 0.f); p_hd[mt] = 0.f; p_c[mt] = 0.f; }
  auto store_saved = [&](int ts) {
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
      if (!evalid[mt]) continue;
      const long tbs = (long)ts * B + eb[mt];
      *reinterpret_cast<float4*>(c.gates + tbs * 4 * h + 4 * eu) = p_gates[mt];
      c.C[tbs * h + eu] = p_c[mt];
      if (c.enc) {
        const int pos = c.reverse_pos ? T - 1 - ts : ts;
        c.enc[((long)eb[mt] * T + pos) * HH + eu] = p_hd[mt];
      }
    }
  };
  long long tk[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  int slow_x = 0, slow_h = 0, nbig2 = 0;
  long long big2 = 0;
  const bool timing = (dbg & 8) != 0;
#define TICK(i, t0) if (timing) { __builtin_amdgcn_sched_barrier(0); const long long now_ = wall_clock64(); tk[i] += now_ - t0; t0 = now_; __builtin_amdgcn_sched_barrier(0); }

  // ---- prologue: inputs of step 0; x_0 into registers, x_1 in flight
  float4 zadd[MT], zadd_n[MT];
  float mk_raw[MT], mk_raw_n[MT];
  if (!HAS_UP && zflags) {
    zf_seen = ld_flag(zfp);
    zx_enter(0);
  }
#pragma unroll
  for (int mt = 0; mt < MT; ++mt) {
    zadd[mt] = bias4; zadd_n[mt] = bias4;
    if (!HAS_UP) zadd[mt] = zx_load(ebc[mt]);
    mk_raw[mt] = maskp[ebc[mt] * h + eu];
  }
  if (HAS_UP) {
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) frag_issue<KB>(r_below, frag0[mt], wave, gx[mt]);
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
      if (!frag_ok<KB>(gx[mt])) frag_wait<KB>(r_below, frag0[mt], wave, gx[mt], dead, ab);
#pragma unroll
      for (int i = 0; i < KB; ++i) ax[mt][i] = frag_vals(gx[mt][i]);
    }
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) frag_issue<KB>(r_below, frag0[mt] + min(1, T - 1) * step_bytes, wave, gx[mt]);
  }

  // One step.  FIRST: no recurrent part (h_{-1} = 0).
  auto step = [&](auto first_tag, int t) {
    constexpr bool FIRST = decltype(first_tag)::value;
    long long t0 = timing ? wall_clock64() : 0;
    f32x4 acc[MT][4];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
      for (int g = 0; g < 4; ++g) acc[mt][g] = f32x4{0.f, 0.f, 0.f, 0.f};
    u32x4 gh[MT][KB];
    if (!FIRST) {
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) frag_issue<KB>(r_own, frag0[mt] + (t - 1) * step_bytes, wave, gh[mt]);   // in flight behind the upward MFMAs
    }
    __builtin_amdgcn_sched_barrier(0);
    if constexpr (HAS_UP) {
      if constexpr (X2) {
        take_x();       // x_t (taken at W_{t-1}) is split HERE, in front of the wait for h, not between that wait and the recurrent MFMAs
#pragma unroll
        for (int p = 0; p < NPR; ++p)
#pragma unroll
          for (int mt = 0; mt < MT; ++mt) {
            if constexpr (XS == 4) { MFMA32BG4(acc[mt], axh[mt][p], wuh[p], lo_u + p * 4 * 256) } else if constexpr (XS == 3) { MFMA32BG(acc[mt], axh[mt][p], wuh[p]) } else { MFMA32HG(acc[mt], axh[mt][p], wuh[p]) }
          }
      } else {
#pragma unroll
        for (int i = 0; i < KB; ++i)
#pragma unroll
          for (int mt = 0; mt < MT; ++mt) { MFMA4G(acc[mt], ax[mt][i], wu[i]) }
      }
    }
    const int t1 = min(t + 1, T - 1), t2 = min(t + 2, T - 1);
    if (HAS_UP) {   // x_{t+1}: issued a whole step ago, taken over HERE, while h_{t-1} is still in flight (behind the wait it was 0.27 us of the chain)
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) {
        if (!frag_ok<KB>(gx[mt])) { ++slow_x; frag_wait<KB>(r_below, frag0[mt] + t1 * step_bytes, wave, gx[mt], dead, ab); }
#pragma unroll
        for (int i = 0; i < KB; ++i) ax[mt][i] = frag_vals(gx[mt][i]);
      }
    }
    __builtin_amdgcn_sched_barrier(0);
    TICK(0, t0)
    // ---- W_t
    if (!FIRST) {
#pragma unroll
      for (int mt = 0; mt < MT; ++mt)
        if (!frag_ok<KB>(gh[mt])) { ++slow_h; frag_wait<KB>(r_own, frag0[mt] + (t - 1) * step_bytes, wave, gh[mt], dead, ab); }
    }
    TICK(1, t0)
    // Everything issued so far has landed (that is what W_t is); saying so explicitly lets the compiler drop its own
    // conservative waits behind the slow path's merge, which would otherwise stall the recurrent MFMAs on the off-path
    // loads issued next.
    __builtin_amdgcn_s_waitcnt(0x0F70);
    { const long long b2_ = tk[2]; TICK(2, t0) if (timing && tk[2] - b2_ > 100) { big2 += tk[2] - b2_; ++nbig2; } }
    if (!FIRST) {
      if constexpr (X2) {
#pragma unroll
        for (int p = 0; p < NPR; ++p)
#pragma unroll
          for (int mt = 0; mt < MT; ++mt) {
            const Frag ah = split_frag<XS>(frag_vals(gh[mt][2 * p]), 2 * p + 1 < KB ? frag_vals(gh[mt][2 * p + 1 < KB ? 2 * p + 1 : 0]) : zero4, ACT_SCALE);
            if constexpr (XS == 4) { MFMA32BG4(acc[mt], ah, wlh[p], lo_l + p * 4 * 256) } else if constexpr (XS == 3) { MFMA32BG(acc[mt], ah, wlh[p]) } else { MFMA32HG(acc[mt], ah, wlh[p]) }
          }
      } else {
#pragma unroll
        for (int i = 0; i < KB; ++i)
#pragma unroll
          for (int mt = 0; mt < MT; ++mt) {
            const float4 ah = frag_vals(gh[mt][i]);
            MFMA4G(acc[mt], ah, wl[i])
          }
      }
    }
    __builtin_amdgcn_sched_barrier(0);
    // ---- off-path traffic: issued BEHIND the recurrent MFMAs (they run in the matrix pipe meanwhile; in front of them these loads, stores and their
    // address arithmetic were 0.18 us of the chain)
    if (HAS_UP) {
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) frag_issue<KB>(r_below, frag0[mt] + t2 * step_bytes, wave, gx[mt]);
    }
    if (!HAS_UP && zflags) {
      zx_enter(t1);                   // (once per chunk: see zx_enter)
      zf_seen = ld_flag(zfp);         // the next chunk's flag, looked at every step
    }
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
      const long tbs = (long)t1 * B + ebc[mt];
      if (!HAS_UP) zadd_n[mt] = zx_load(tbs);
      mk_raw_n[mt] = maskp[tbs * h + eu];
    }
    if (!FIRST) store_saved(t - 1);
    __builtin_amdgcn_sched_barrier(0);
    TICK(3, t0)
    // ---- 4-wave K reduction through LDS (double-buffered: one barrier per step)
    float* rd = (t & 1) ? red1 : red0;
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
      for (int g = 0; g < 4; ++g) *reinterpret_cast<f32x4*>(&rd[mt * 4096 + ((wave * 4 + g) * 64 + lane) * 4]) = acc[mt][g];
    TICK(4, t0)
    __syncthreads();
    TICK(5, t0)
    {
      const int row = tid >> 4, col = tid & 15;
      const int src = ((row >> 2) * 16 + col) * 4 + (row & 3);
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) {
        const float* rm = rd + mt * 4096;
        float z[4];
#pragma unroll
        for (int g = 0; g < 4; ++g)
          z[g] = rm[(0 * 4 + g) * 256 + src] + rm[(1 * 4 + g) * 256 + src] + rm[(2 * 4 + g) * 256 + src] + rm[(3 * 4 + g) * 256 + src];
        if constexpr (XS == 2) {
#pragma unroll
          for (int g = 0; g < 4; ++g) z[g] *= zscale;      // 1 / (weight scale x activation scale)
        }
        const float ga = tanh_fast(z[0] + zadd[mt].x), gi = sigm_fast(z[1] + zadd[mt].y), gf = sigm_fast(z[2] + zadd[mt].z), go = sigm_fast(z[3] + zadd[mt].w);
        c_state[mt] = ga * gi + gf * c_state[mt];
        const float hh = go * tanh_fast(c_state[mt]);
        const float hd = use_mask ? hh * mk_raw[mt] : hh;
        // the hand-off: the values themselves, write-through; nothing to drain or signal
        if (evalid[mt]) {
          const long o = ((long)t * B + eb[mt]) * h + eu;
          st4_sc1(c.HR + o, hh);
          if (c.HD) st4_sc1(c.HD + o, hd);
        }
        p_gates[mt] = make_float4(ga, gi, gf, go);
        p_hd[mt] = hd; p_c[mt] = c_state[mt];
      }
    }
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) { zadd[mt] = zadd_n[mt]; mk_raw[mt] = mk_raw_n[mt]; }
    TICK(6, t0)
  };
  step(std::true_type{}, 0);
  for (int t = 1; t < T; ++t) step(std::false_type{}, t);
  store_saved(T - 1);
  if (timing && lane == 0 && blockIdx.x == 0 && blockIdx.y == 0)
    printf("persist_fwd_g cell %d (layer %d) wave %d: per-step 10ns ticks: issue+xmfma %lld  h_wait %lld  take_x %lld  hmfma+traffic %lld  lds %lld  barrier %lld  epilogue %lld  slow x %d h %d  take_x > 1 us: %d times, %lld ticks\n",
           (int)blockIdx.z, c.layer, wave, tk[0] / T, tk[1] / T, tk[2] / T, tk[3] / T, tk[4] / T, tk[5] / T, tk[6] / T, slow_x, slow_h, nbig2, big2);
#undef TICK
}

// the DUO form of the forward kernel (see lstm_fwd_steps): 512 threads, both halves share the lo plane, each has its own reduction buffers
template <int KB>
__global__ __launch_bounds__(512, 1) void lstm_persist_fwd_duo(PFwdArgs a) {
  __shared__ __attribute__((aligned(16))) float red[2][2][4 * 4 * 256];
  __shared__ __attribute__((aligned(16))) u32q lo_lds[(LO_LDS_FRAGS / 2) * 256];
  static_assert(2 * ((KB + 1) / 2) * 4 <= LO_LDS_FRAGS / 2, "lo plane of both products in 64 KB");
  const PCellF c = a.c[blockIdx.z];
  const int half = threadIdx.x >> 8;
  if (c.Wu != nullptr) { lstm_fwd_steps<KB, true, 4, 1, true>(a, c, red[half][0], red[half][1], lo_lds); return; }
  lstm_fwd_steps<KB, false, 4, 1, true>(a, c, red[half][0], red[half][1], lo_lds);
}

template <int KB, int XS, int MT>
__global__ __launch_bounds__(256, 1) void lstm_persist_fwd_g(PFwdArgs a) {
  __shared__ __attribute__((aligned(16))) float red[2][MT * 4 * 4 * 256];
  // (a COPY, not a reference into the kernel-argument block: fields read through a dynamically indexed reference are re-loaded behind
  //  every global store of the step loop -- the stores might alias them -- and each re-load is a scalar-cache round trip on the chain)
  __shared__ __attribute__((aligned(16))) u32q lo_lds[XS == 4 ? LO_LDS_FRAGS * 256 : 1];
  const PCellF c = a.c[blockIdx.z];
  // a cell multiplies its input itself iff it was given the upward weight: layer 0 -- and, in the hoisted form (h = 1024: the weight
  // fragments of ONE product fill the registers), every layer -- gets the projection of all time steps from a batched GEMM (zx)
  if constexpr (KB <= 8) {
    if (c.Wu != nullptr) { lstm_fwd_steps<KB, true, XS, MT>(a, c, red[0], red[1], lo_lds); return; }
  }
  lstm_fwd_steps<KB, false, XS, MT>(a, c, red[0], red[1], lo_lds);
}

// ------------------------------------------------------------------ backward, reduce-scatter hand-off
// lstm_persist_bwd gathers: every workgroup pulls the whole dz_{t+1} row block of its cell (16 x 4h) and of the layer above
// -- 128 KB per workgroup and step, which is what bounds it (6.2 us/step).  Here the products are turned around: the
// workgroup that OWNS 64 gate columns of dz multiplies them with the matching 64 rows of W_l (and of its own W_u for the layer
// below) as soon as its gate epilogue has produced them, and hands out 16x16 PARTIAL sums, one tile per consuming slice; a
// consumer adds the 16 partial tiles of its slice: 16 KB + 16 KB per workgroup and step instead of 128 KB, same MFMA count.
// Thread ownership is (unit = tid >> 4, row = tid & 15) so that the partial tiles, stored by the MFMA lanes as [col][row] in
// 16-byte pieces, are read back with fully coalesced 4-byte loads.  Counter protocol (R1): partials are written through (sc1),
// drained, one arrival per workgroup and step on counter A (own cell) and, after the second product, counter B (layer below).
constexpr int PR_RING = 4;
// Own-cell hand-off of lstm_persist_bwd_rs (the partial dh_rec tiles, on the recurrence's critical path).  1: the data is the flag, as in
// the forward kernel -- the ring is sentinel-filled before the launch, a consumer polls its 16 KB of partial words themselves and puts the
// sentinel back behind its read; a slot comes round again PR_RING steps later, and the consumer's per-step vmcnt(0) orders its reset in
// front of everything its peers can have seen of it since.  No drain, barrier, counter or counter poll on the chain.  0: counter A.
// Used up to h = 256 (KB <= 4): at h = 512 a consumer's slice is 32 KB per step and the counter form is the faster one (BASELINE
// configs[4] shape: 22.0 against 22.5 ms per train step).
#ifndef ASTK_BWD_SENTINEL
#define ASTK_BWD_SENTINEL 1
#endif
#ifndef ASTK_BWD_SENTINEL_MAXKB
#define ASTK_BWD_SENTINEL_MAXKB 4
#endif
constexpr bool bwd_sentinel(int KB) { return ASTK_BWD_SENTINEL && KB <= ASTK_BWD_SENTINEL_MAXKB; }
// HAS_UP: the cell has a layer above it in this stack (a template parameter so that the loads of that layer's partials are unconditional
// code: a conditionally issued load becomes a phi whose copy makes hipcc wait for the load where it is issued)
// Row stride of the 16 x 64 dz tile in LDS: 68 floats, not 64.  Writers (unit u = tid >> 4, row r = tid & 15) and readers (row r16 = lane & 15,
// k-quad q = lane >> 4) both move 16 bytes per lane with 16 consecutive lanes on 16 DIFFERENT rows of the same columns: with a 256-byte
// row stride those 16 lanes sit on the same four banks (16-way conflicts on every access: 25.8 M SQ_LDS_BANK_CONFLICT cycles per launch
// against 4.9 M in the forward kernel, round-3 PMC pass); 68 moves consecutive rows four banks on and the accesses are conflict-free.
constexpr int DZ_LD = 68;
// MT = 16-row batch tiles per workgroup (see lstm_fwd_steps): the resident weight fragments of both products serve both tiles; the partial
// tile buffers keep their per-16-row-tile layout (tile index bt16 = blockIdx.y * MT + mt), the counters are per workgroup row (blockIdx.y).
// DUO: see lstm_fwd_steps -- two virtual 16-row workgroups (by = 2 blockIdx.y + half) in one 512-thread workgroup, lo planes shared in LDS.
template <int KB, bool HAS_UP, int XS, int MT, bool DUO = false>
__device__ __forceinline__ void lstm_bwd_rs_steps(const PBwdArgs& a, const PCellB& c, float (*dzS2)[MT * 16 * DZ_LD], int* s_ok1, int& s_ok2, u32q* lo_lds) {
  static_assert(!DUO || (MT == 1 && XS == 4), "DUO: 16-row halves on the bf16x3 form with the lo plane in LDS");
  constexpr int NS = 4 * KB;          // slices of a cell = 16x16 output tiles of a product = partial tiles per consumer
  float* const dzS = dzS2[0];
  const int tid = DUO ? (int)(threadIdx.x & 255) : (int)threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r16 = lane & 15, q = lane >> 4;
  const int cell = blockIdx.z, by = DUO ? (int)(blockIdx.y * 2 + (threadIdx.x >> 8)) : (int)blockIdx.y, j = blockIdx.x, j0 = j * 16;
  const int T = a.T, B = a.B, h = a.h, HH = a.H, dbg = PERSIST_DBG(a);
  const AbortCtl ab = a.ab;
  const unsigned amax_gen = a.amax_gen;
  const int nby = DUO ? (int)gridDim.y * 2 : (int)gridDim.y, nbt = nby * MT;       // (virtual) workgroup rows; 16-row tiles the partial buffers are laid out for
  const int K = 4 * h;
  constexpr bool has_up = HAS_UP;
  // h = 1024 (KB = 16): one product's weight fragments fill the registers -- the gradient for the layer below is a batched GEMM behind
  // the launch there (hoisted form), and the code of the down product is compiled out
  constexpr bool CAN_DOWN = KB <= 8;
  const bool has_down = CAN_DOWN && c.PD != nullptr;
  unsigned* ctrA = a.done + (cell * nby + by) * CTR_STRIDE;
  unsigned* ctrB = a.done + ((a.ncells + cell) * nby + by) * CTR_STRIDE;
  const unsigned* upB = has_up ? a.done + ((a.ncells + cell + 1) * nby + by) * CTR_STRIDE : nullptr;
  const int m0 = by * 16 * MT;
  (void)HH;

  // resident weight fragments: product tile tl = wave*KB + nt covers output units 16 tl .. 16 tl + 15; K = this slice's 64 gate columns
  // (read straight from the (4h, h) parameters: four 4-byte loads per fragment quad, 16 consecutive lanes on 64 consecutive bytes, once per
  //  launch -- the transposed copies a launch of its own used to make every step, 8 us, are gone)
  auto w_at = [&](const float* W, int nt, int s4) {
    const float* p = W + (long)(64 * j + 16 * s4 + 4 * q) * h + 16 * (wave * KB + nt) + r16;
    return make_float4(p[0], p[h], p[2 * (long)h], p[3 * (long)h]);
  };
  auto wl_at = [&](int nt, int s4) { return w_at(c.Wl, nt, s4); };
  auto wd_at = [&](int nt, int s4) { return has_down ? w_at(c.Wu, nt, s4) : make_float4(0.f, 0.f, 0.f, 0.f); };
  // X2: fp16 hi / lo fragments behind one power-of-two scale per workgroup (see the forward kernel); the 64 gate columns are two 32-k operands
  constexpr bool X2 = XS != 0;          // a split scheme (16-bit MFMAs); XS == 2 additionally scales
  typedef typename FragOf<XS>::type Frag;
  typedef typename WFragOf<XS>::type WFrag;
  static_assert(XS != 4 || (CAN_DOWN ? 2 : 1) * KB * 2 <= LO_LDS_FRAGS, "lo plane does not fit its LDS region");
  u32q* const lo_l = lo_lds + tid, * const lo_d = lo_lds + KB * 2 * 256 + tid;       // this thread's lo fragments: recurrent [nt][p], down behind them
  WFrag wlh[X2 ? KB : 1][2], wdh[(X2 && CAN_DOWN) ? KB : 1][2];
  float4 wl[X2 ? 1 : KB][4], wd[(!X2 && CAN_DOWN) ? KB : 1][4];
  float winv = 1.f;
  if constexpr (X2) {
    float wscl = 1.f;
    if constexpr (XS == 2) {
      float m = 0.f;
#pragma unroll
      for (int nt = 0; nt < KB; ++nt)
#pragma unroll
        for (int s4 = 0; s4 < 4; ++s4) m = amax4f(amax4f(m, wl_at(nt, s4)), CAN_DOWN ? wd_at(nt, s4) : make_float4(0.f, 0.f, 0.f, 0.f));
      wscl = pow2_scale_for(wg_max(m, dzS), winv);
    }
#pragma unroll
    for (int nt = 0; nt < KB; ++nt)
#pragma unroll
      for (int p = 0; p < 2; ++p) {
        if constexpr (XS == 4) {
          const HML8 t = split8b(wl_at(nt, 2 * p), wl_at(nt, 2 * p + 1));
          wlh[nt][p].hi = t.hi; wlh[nt][p].mid = t.mid;
          lo_l[(nt * 2 + p) * 256] = t.lo;
          if constexpr (CAN_DOWN) {
            const HML8 u = split8b(wd_at(nt, 2 * p), wd_at(nt, 2 * p + 1));
            wdh[nt][p].hi = u.hi; wdh[nt][p].mid = u.mid;
            lo_d[(nt * 2 + p) * 256] = u.lo;
          }
        } else {
        wlh[nt][p] = split_frag<XS>(wl_at(nt, 2 * p), wl_at(nt, 2 * p + 1), wscl);
        if constexpr (CAN_DOWN) wdh[nt][p] = split_frag<XS>(wd_at(nt, 2 * p), wd_at(nt, 2 * p + 1), wscl);
        }
      }
  } else {
#pragma unroll
    for (int nt = 0; nt < KB; ++nt)
#pragma unroll
      for (int s4 = 0; s4 < 4; ++s4) { wl[nt][s4] = wl_at(nt, s4); if constexpr (CAN_DOWN) wd[nt][s4] = wd_at(nt, s4); }
  }
  const __amdgpu_buffer_rsrc_t r_pr = make_rsrc(c.PR);
  const __amdgpu_buffer_rsrc_t r_pd = make_rsrc(has_down ? c.PD : c.PR);
  const __amdgpu_buffer_rsrc_t r_pu = make_rsrc(has_up ? c.PD_up : c.PR);
  const __amdgpu_buffer_rsrc_t r_dz = make_rsrc(c.gates_dz);
  const int u = tid >> 4, r = tid & 15;                   // epilogue ownership: (unit, row)
  const int eu = j0 + u;
  int eb[MT];
  bool evalid[MT];
  long ebc[MT];
  float dc_state[MT], dhadd[MT];
  float4 dbacc[MT];
  float dzmax = 0.f;
  // bias gradient = column sums of dz over all steps and rows: this thread's (row, unit) element of every step summed in registers, the
  // 16 rows and the batch tiles at the end -- the separate pass over the six cells' dz (157 MB, 32 us per train step) is gone
#pragma unroll
  for (int mt = 0; mt < MT; ++mt) {
    eb[mt] = m0 + 16 * mt + r;
    evalid[mt] = eb[mt] < B;
    ebc[mt] = evalid[mt] ? eb[mt] : B - 1;
    dc_state[mt] = 0.f; dhadd[mt] = 0.f;
    dbacc[mt] = make_float4(0.f, 0.f, 0.f, 0.f);
    if (c.d_cT) dc_state[mt] = c.d_cT[ebc[mt] * h + eu];
    if (c.d_hT) dhadd[mt] = c.d_hT[ebc[mt] * h + eu];
  }
  const int tile_bytes = 256 * 4;                          // one 16x16 partial tile
  const long cons_stride = (long)NS * tile_bytes;          // bytes between consumers
  // Partials of the layer above: at rest in HBM when they are wanted (that layer runs ahead), i.e. a full memory latency (2.1-2.6 us in
  // the stamps) in front of the gate epilogue if they are fetched when the step begins.  They are fetched a step early instead, right behind the barrier, when
  // the counter says that the layer above has published them (it has, except while the pipeline fills).
  // In front of the barrier of step t one lane makes sure that the layer above has published step t-1 (asked when the step begins, the
  // answer is a memory round trip away; it blocks only while the pipeline fills), behind the barrier everybody fetches.
  constexpr bool SENT = bwd_sentinel(KB);
#ifndef ASTK_BWD_UP_PREFETCH
#define ASTK_BWD_UP_PREFETCH 1
#endif
  constexpr bool UP_PREFETCH = ASTK_BWD_UP_PREFETCH && SENT && HAS_UP && KB <= 4;      // (32 more live registers do not fit the h = 512 kernel)
  float pu[MT][NS];
  bool alive = true;
  if (UP_PREFETCH) {
    if (!c.up_external) {
      if (tid == 0) s_ok1[0] = wait_ge(upB, (unsigned)NS, ab) ? 1 : 0;
      __syncthreads();
      alive = s_ok1[0] != 0;
      __syncthreads();
    }
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
      const int base = (int)((((long)(T - 1) * nbt + by * MT + mt) * NS + j) * cons_stride) + tid * 4;
#pragma unroll
      for (int p = 0; p < NS; ++p) pu[mt][p] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(r_pu, base + p * tile_bytes, 0, 16));
    }
  }
  bool pending_b = false;
  bool pending_prog = false;     // layer 0 with side-stream consumers: a chunk's arrival on the progress counter is due at the next drain point
  bool dead = false;      // this wave gave up waiting (abort / time-out): it runs the remaining steps without waiting, so that barriers still match
  f32x4 acc2[MT][CAN_DOWN ? KB : 1];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt)
#pragma unroll
    for (int nt = 0; nt < (CAN_DOWN ? KB : 1); ++nt) acc2[mt][nt] = f32x4{0.f, 0.f, 0.f, 0.f};
  auto store_down = [&](int ts) {
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
      for (int nt = 0; nt < (CAN_DOWN ? KB : 0); ++nt) {
        const int tl = wave * KB + nt;
        u32x4 o;
        o.x = __float_as_uint(acc2[mt][nt][0]); o.y = __float_as_uint(acc2[mt][nt][1]); o.z = __float_as_uint(acc2[mt][nt][2]); o.w = __float_as_uint(acc2[mt][nt][3]);
        __builtin_amdgcn_raw_buffer_store_b128(o, r_pd, (int)((((long)ts * nbt + by * MT + mt) * NS + tl) * cons_stride) + j * tile_bytes + (r16 * 16 + 4 * q) * 4, 0, 16);
      }
  };

  // Saved forward state and the incoming gradient of a step: from earlier launches, i.e. fetchable at any time.  Fetched ONE STEP EARLY,
  // right behind the step's barrier: at the top of a step the wave's memory queue is still full of the previous step's write-through
  // stores (partials, sentinel resets, dz), and loads issued there -- and the sweep behind them -- wait for the queue, not for data.
  float4 in_g[MT];
  float in_cc[MT], in_cp[MT], in_mk[MT], in_dye[MT];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt) { in_g[mt] = make_float4(0.f, 0.f, 0.f, 0.f); in_cc[mt] = 0.f; in_cp[mt] = 0.f; in_mk[mt] = 1.f; in_dye[mt] = 0.f; }
  auto fetch_inputs = [&](const int ts) {          // ts >= 0 (callers clamp)
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
      const long tbs = (long)ts * B + ebc[mt];
      in_g[mt] = *reinterpret_cast<const float4*>(c.gates_dz + tbs * K + 4 * eu);
      in_cc[mt] = c.C[tbs * h + eu];
      in_cp[mt] = c.C[max(tbs - B, 0L) * h + eu];
      // (unconditional loads through a selected pointer: a load under `if (c.d_enc)` is a phi of "old value / load result", which hipcc
      //  resolves by waiting for the load where it is issued -- a full memory latency per step, 1.75 us in the top layer's timers)
      const float* const mp = c.mask ? c.mask + tbs * h + eu : c.C + tbs * h + eu;
      const float* const dp = c.d_enc ? c.d_enc + ebc[mt] * c.dy_sb + (c.reverse_pos ? T - 1 - ts : ts) * c.dy_st + eu : c.C + tbs * h + eu;
      const float mv = *mp, dv = *dp;
      in_mk[mt] = c.mask ? mv : 1.f;
      in_dye[mt] = c.d_enc ? dv : 0.f;
    }
  };
  if (alive && T > 0) fetch_inputs(T - 1);
  long long tk[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  const bool timing = (dbg & ~16) != 0;
#define TICK(i, t0) if (timing) { const long long now_ = wall_clock64(); tk[i] += now_ - t0; t0 = now_; }
  for (int t = alive ? T - 1 : -1; t >= 0; --t) {
    long long t0 = timing ? wall_clock64() : 0;
    const int stepno = T - 1 - t;
    float* const dzT = dzS2[stepno & 1];
    // inputs from earlier launches (fetched a step ago, see fetch_inputs)
    float4 g[MT];
    float ccur[MT], cp[MT], mk[MT], dye[MT], v1[MT], v0[MT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
      g[mt] = in_g[mt]; ccur[mt] = in_cc[mt]; cp[mt] = t > 0 ? in_cp[mt] : 0.f; mk[mt] = in_mk[mt]; dye[mt] = in_dye[mt];
      v1[mt] = 0.f; v0[mt] = 0.f;
    }
    unsigned up_seen = 0;
    if (has_up) {                            // partials handed down by the layer above (it runs ahead)
      if (!UP_PREFETCH) {
        if (!c.up_external) {
          if (tid == 0) s_ok1[0] = wait_ge(upB, (unsigned)(NS * (stepno + 1)), ab) ? 1 : 0;
          __syncthreads();
          if (!s_ok1[0]) break;
        }
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
          const int base = (int)((((long)t * nbt + by * MT + mt) * NS + j) * cons_stride) + tid * 4;
#pragma unroll
          for (int p = 0; p < NS; ++p) pu[mt][p] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(r_pu, base + p * tile_bytes, 0, 16));
        }
      }
#pragma unroll
      for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int p = 0; p < NS; ++p) v1[mt] += pu[mt][p];
      if (UP_PREFETCH && !c.up_external && tid == 0) up_seen = ld_flag(upB);
    }
    TICK(0, t0)
    int reset_base[MT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) reset_base[mt] = -1;
    if (stepno > 0) {                        // partial dh_rec tiles of this cell's step t+1
      const int slot = (t + 1) % PR_RING;
      int base[MT];
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) base[mt] = (int)((((long)slot * nbt + by * MT + mt) * NS + j) * cons_stride) + tid * 4;
      if constexpr (SENT) {
      unsigned pw[MT][NS];
      auto sweep = [&]() {
        unsigned mx = 0;      // (the sentinel is the largest unsigned word: a running maximum and one compare, not a branch per word)
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
          for (int p = 0; p < NS; ++p) pw[mt][p] = __builtin_amdgcn_raw_buffer_load_b32(r_pr, base[mt] + p * tile_bytes, 0, 16);
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
          for (int p = 0; p < NS; ++p) mx = max(mx, pw[mt][p]);
        return __all(mx != SENTINEL);
      };
      if (!sweep()) {
        // slow path: poll ONE word per lane (its first missing one) until the wave has them all, then re-read everything; bounded
        unsigned spins = 0;
        while (!dead) {
#if ASTK_FRAG_WAIT_SWEEP
          if (spins < ASTK_FRAG_WAIT_SWEEP) {      // the first retries are whole sweeps (see frag_wait)
            if (sweep()) break;
            ++spins;
            continue;
          }
#endif
          int moff = base[0];
#pragma unroll
          for (int mt = MT - 1; mt >= 0; --mt)
#pragma unroll
            for (int p = NS - 1; p >= 0; --p) moff = pw[mt][p] == SENTINEL ? base[mt] + p * tile_bytes : moff;
          const unsigned cw = __builtin_amdgcn_raw_buffer_load_b32(r_pr, moff, 0, 16);
          if (__all(cw != SENTINEL)) {
            if (sweep()) break;
          }
          if (++spins > (ab.limit >> 1)) { abort_raise(ab); dead = true; }
          else if ((spins & 63u) == 0 && abort_seen(ab)) dead = true;
        }
      }
      TICK(1, t0)
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) {
#pragma unroll
        for (int p = 0; p < NS; ++p) v0[mt] += __uint_as_float(pw[mt][p]);
        reset_base[mt] = base[mt] - tid * 4;      // (the slot's words go back to the sentinel behind this step's barrier and product-1 stores)
      }
      } else {
      if (tid == 0) s_ok2 = wait_ge(ctrA, (unsigned)(NS * stepno), ab) ? 1 : 0;
      __syncthreads();
      if (!s_ok2) break;
      TICK(1, t0)
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) {
        float pv[NS];
#pragma unroll
        for (int p = 0; p < NS; ++p) pv[p] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(r_pr, base[mt] + p * tile_bytes, 0, 16));
#pragma unroll
        for (int p = 0; p < NS; ++p) v0[mt] += pv[p];
      }
      }
    }
    TICK(2, t0)
    // every vector memory operation of this wave up to the partial loads has completed (they were just consumed, vmcnt retires in
    // order; the previous step's down-partials and sentinel resets went out a whole step ago): this is the drain the deferred
    // publish of counter B needs, and what orders a reset in front of the slot's next use
#ifdef ASTK_BWD_DRAIN_B
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#else
    if constexpr (!SENT) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
    float4 dz[MT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
      const float dh = v0[mt] + (v1[mt] + dye[mt]) * mk[mt] + (stepno == 0 ? dhadd[mt] : 0.f);
      const float tc = tanh_fast(ccur[mt]);   // the same function the forward kernel used for tanh(c)
      const float dcv = dh * g[mt].w * (1.f - tc * tc) + dc_state[mt];
      dz[mt] = make_float4(dcv * g[mt].y * (1.f - g[mt].x * g[mt].x), dcv * g[mt].x * g[mt].y * (1.f - g[mt].y), dcv * cp[mt] * g[mt].z * (1.f - g[mt].z),
                           dh * tc * g[mt].w * (1.f - g[mt].w));
      dc_state[mt] = dcv * g[mt].z;
      *reinterpret_cast<float4*>(&dzT[mt * 16 * DZ_LD + r * DZ_LD + 4 * u]) = dz[mt];
      dzmax = fmaxf(fmaxf(dzmax, fmaxf(fabsf(dz[mt].x), fabsf(dz[mt].y))), fmaxf(fabsf(dz[mt].z), fabsf(dz[mt].w)));   // (rows past B repeat row B-1)
    }
    TICK(3, t0)
    if (UP_PREFETCH && !c.up_external && tid == 0) {
      const unsigned want = (unsigned)(NS * (stepno + 2));
      s_ok1[stepno & 1] = (t == 0 || up_seen >= want || wait_ge(upB, want, ab)) ? 1 : 0;
    }
    __syncthreads();
    if (pending_b && tid == 0) __hip_atomic_fetch_add(ctrB, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // down-partials of step t+1
    if (pending_prog) {          // (uniform) dz of the chunk that ended with step t+1: written through a step ago, drained by this step's partial loads
      if (tid == 0) __hip_atomic_fetch_add(c.prog, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      pending_prog = false;
    }
    if (UP_PREFETCH) {
      if (!c.up_external && !s_ok1[stepno & 1]) break;
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) {
        const int base = (int)((((long)max(t - 1, 0) * nbt + by * MT + mt) * NS + j) * cons_stride) + tid * 4;      // (the last step fetches its own again)
#pragma unroll
        for (int p = 0; p < NS; ++p) pu[mt][p] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(r_pu, base + p * tile_bytes, 0, 16));
      }
    }
    fetch_inputs(max(t - 1, 0));
    float4 af[MT][4];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
      for (int s4 = 0; s4 < 4; ++s4) af[mt][s4] = *reinterpret_cast<const float4*>(&dzT[mt * 16 * DZ_LD + r16 * DZ_LD + 16 * s4 + 4 * q]);
    // X2: dz is unbounded: its scale is taken from this step's tile (every wave holds the whole 16 x 64 tile across its lanes)
    float pscale[MT];       // 1 / (weight scale x dz scale), applied to the partial sums
    Frag afh[MT][2];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
      pscale[mt] = 1.f;
      if constexpr (XS == 2) {
        const float m = wave_max_nonneg(amax4f(amax4f(amax4f(amax4f(0.f, af[mt][0]), af[mt][1]), af[mt][2]), af[mt][3]));
        float ainv;
        const float ascl = pow2_scale_for(m, ainv);
        pscale[mt] = ainv * winv;
        afh[mt][0] = split8(af[mt][0], af[mt][1], ascl);
        afh[mt][1] = split8(af[mt][2], af[mt][3], ascl);
      } else if constexpr (XS == 3 || XS == 4) {      // bf16 has f32's exponent range: no scale, however small or large dz is
        afh[mt][0] = split8b(af[mt][0], af[mt][1]);
        afh[mt][1] = split8b(af[mt][2], af[mt][3]);
      }
    }
    // ---- product 1: partial dh_rec for every slice of this cell -> write-through stores
    {
      const int slot = t % PR_RING;
#pragma unroll
      for (int nt = 0; nt < KB; ++nt) {
        f32x4 acc[MT];
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) acc[mt] = f32x4{0.f, 0.f, 0.f, 0.f};
        if constexpr (XS == 4) {
          const u32q l0 = lo_l[(nt * 2) * 256], l1 = lo_l[(nt * 2 + 1) * 256];
#pragma unroll
          for (int mt = 0; mt < MT; ++mt) {
            MFMA32B4(acc[mt], afh[mt][0], wlh[nt][0], l0)
            MFMA32B4(acc[mt], afh[mt][1], wlh[nt][1], l1)
          }
        } else {
#pragma unroll
          for (int mt = 0; mt < MT; ++mt) {
            if constexpr (XS == 2) {
              MFMA32H(acc[mt], afh[mt][0], wlh[nt][0])
              MFMA32H(acc[mt], afh[mt][1], wlh[nt][1])
              acc[mt] *= pscale[mt];
            } else if constexpr (XS == 3) {
              MFMA32B(acc[mt], afh[mt][0], wlh[nt][0])
              MFMA32B(acc[mt], afh[mt][1], wlh[nt][1])
            } else {
#pragma unroll
              for (int s4 = 0; s4 < 4; ++s4) { MFMA4(acc[mt], af[mt][s4], wl[nt][s4]) }
            }
          }
        }
        const int tl = wave * KB + nt;
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
          u32x4 o;
          o.x = __float_as_uint(acc[mt][0]); o.y = __float_as_uint(acc[mt][1]); o.z = __float_as_uint(acc[mt][2]); o.w = __float_as_uint(acc[mt][3]);
          __builtin_amdgcn_raw_buffer_store_b128(o, r_pr, (int)((((long)slot * nbt + by * MT + mt) * NS + tl) * cons_stride) + j * tile_bytes + (r16 * 16 + 4 * q) * 4, 0, 16);
        }
      }
    }
    // (ASTK_PERSIST_DBG & 16, the regression test of the last-arrival rule on counter B: slice 0 of every cell with a layer below dawdles for
    //  30 us between its product-1 stores of step 1 and its down partials -- its peers need nothing else from it to finish the launch)
    if ((dbg & 16) && has_down && j == 0 && t == 1) {
      const long long until = wall_clock64() + 3000;
      while (wall_clock64() < until) __builtin_amdgcn_s_sleep(8);
    }
    // ---- product 2 (partial dx for the layer below): its MFMAs run while the product-1 stores land; its own stores go out
    // behind the publish of counter A and counter B is bumped at the next step's drain point.  (Measured alternatives: product 2
    // behind the publish, with or without holding its stores back: 0.3-0.4 us per step slower -- the peers see counter A only
    // ~2 us after the atomic either way, so hiding the 0.5 us drain is what pays.)
    constexpr int KB1 = (KB + 1) / 2;     // first half of product 2 hides the drain, second half runs behind the publish
    auto product2 = [&](const int nt) {
      if constexpr (CAN_DOWN) {
        if constexpr (XS == 4) {
          const u32q l0 = lo_d[(nt * 2) * 256], l1 = lo_d[(nt * 2 + 1) * 256];
#pragma unroll
          for (int mt = 0; mt < MT; ++mt) {
            acc2[mt][nt] = f32x4{0.f, 0.f, 0.f, 0.f};
            MFMA32B4(acc2[mt][nt], afh[mt][0], wdh[nt][0], l0)
            MFMA32B4(acc2[mt][nt], afh[mt][1], wdh[nt][1], l1)
          }
        } else {
#pragma unroll
          for (int mt = 0; mt < MT; ++mt) {
            acc2[mt][nt] = f32x4{0.f, 0.f, 0.f, 0.f};
            if constexpr (XS == 2) {
              MFMA32H(acc2[mt][nt], afh[mt][0], wdh[nt][0])
              MFMA32H(acc2[mt][nt], afh[mt][1], wdh[nt][1])
              acc2[mt][nt] *= pscale[mt];
            } else if constexpr (XS == 3) {
              MFMA32B(acc2[mt][nt], afh[mt][0], wdh[nt][0])
              MFMA32B(acc2[mt][nt], afh[mt][1], wdh[nt][1])
            } else {
#pragma unroll
              for (int s4 = 0; s4 < 4; ++s4) { MFMA4(acc2[mt][nt], af[mt][s4], wd[nt][s4]) }
            }
          }
        }
      }
    };
    if constexpr (CAN_DOWN) if (has_down) {
#pragma unroll
      for (int nt = 0; nt < KB1; ++nt) product2(nt);
    }
    TICK(4, t0)
    if constexpr (SENT) {
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
    if (reset_base[mt] >= 0) {   // behind the step's barrier every reader of the slot just consumed is done with it: the sentinel goes back (16 KB in a row)
      const u32x4 sent = {SENTINEL, SENTINEL, SENTINEL, SENTINEL};
#pragma unroll
      for (int i = 0; i < NS / 4; ++i) __builtin_amdgcn_raw_buffer_store_b128(sent, r_pr, reset_base[mt] + (i * 256 + tid) * 16, 0, 16);
    }
    } else {
      publish(ctrA, tid);      // drain (product-1 stores only), barrier, one arrival
    }
    TICK(5, t0)
    // dz for the batched products: behind the launch (plain stores), or -- layer 0 with side-stream consumers -- beside it, chunk by chunk:
    // written through (sc1), the chunk's arrival on the progress counter follows at the next step's drain point
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
      if (evalid[mt]) {
        const long tb = (long)t * B + eb[mt];
        if (c.prog) {
          u32x4 o;
          o.x = __float_as_uint(dz[mt].x); o.y = __float_as_uint(dz[mt].y); o.z = __float_as_uint(dz[mt].z); o.w = __float_as_uint(dz[mt].w);
          __builtin_amdgcn_raw_buffer_store_b128(o, r_dz, (int)((tb * K + 4 * eu) * 4), 0, 16);
        } else {
          *reinterpret_cast<float4*>(c.gates_dz + tb * K + 4 * eu) = dz[mt];
        }
      }
      dbacc[mt].x += dz[mt].x; dbacc[mt].y += dz[mt].y; dbacc[mt].z += dz[mt].z; dbacc[mt].w += dz[mt].w;
    }
    if (c.prog && (stepno + 1) % c.prog_cs == 0 && t > 0) pending_prog = true;
    if constexpr (CAN_DOWN) if (has_down) {
#pragma unroll
      for (int nt = KB1; nt < KB; ++nt) product2(nt);
      store_down(t);
      pending_b = true;
    }
    TICK(6, t0)
  }
  if (pending_b) {
    // The LAST arrival on counter B (step 0's down partials) needs nothing from the peers, all earlier ones do: a workgroup reaches the
    // barrier of step s only behind its peers' partials of step s+1, i.e. behind their arrival for step s+2, so the 4 KB slices of a
    // (cell, batch tile) are never more than one arrival apart and "count >= NS * k" means "everybody has published k steps".  Without
    // this wait a workgroup that finished early made its T-th arrival while a peer had made T-2: the count reached NS * (T-1) one arrival
    // short of that peer's, and a consumer of the layer below read the peer's tile of step 1 before it was written -- the previous
    // launch's tile.  Once in ~4000 launches, on the last two steps of a layer-0 cell, 1e-4 of two gradient tensors (found by a soak over
    // 3000 batches; scratch/enc_repeat.py reproduces it with two alternating inputs: with one input the stale tile is a copy of the right one).
    // (a wait that gives up -- abort word set, or its spin bound hit, which sets it -- falls through to the publish: every consumer of the
    //  counter is draining on the same abort word by then and the step's results are discarded, so an early arrival harms nobody)
    if (T > 1) {
      if (tid == 0) (void)wait_ge(ctrB, (unsigned)(NS * (T - 1)), ab);
      __syncthreads();
    }
    publish(ctrB, tid);
  }
  if (c.prog) {
    // the last chunk (and a pending one): every storing wave drains, the workgroup barriers, one lane arrives.  The consumer (a wait kernel
    // on the side stream) wants `chunks x workgroups of the cell` arrivals for the chunk that ends with step 0: one arrival per workgroup and
    // chunk, pending or not, so the total is always ceil(T / prog_cs) per workgroup
    // (the LAST arrival is the one that needs nothing from the peers -- the same shape as counter B's: a workgroup that finishes early must not
    //  lift the count to `workgroups x k` while a peer has made k - 1 arrivals and not yet stored the last dz of chunk k - 1; it waits until
    //  every peer has made all its deferred arrivals.  tests/protocol_model.py: progress_counter_procs finds the stale read without this wait.)
    const int nchunks = (T + c.prog_cs - 1) / c.prog_cs;
    if (nchunks > 1) {
      if (tid == 0) (void)wait_ge(c.prog, (unsigned)(NS * nby * (nchunks - 1)), ab);
      __syncthreads();
    }
    publish(c.prog, tid);
  }
  if (c.db) {
    float4 dbs = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
      if (evalid[mt]) { dbs.x += dbacc[mt].x; dbs.y += dbacc[mt].y; dbs.z += dbacc[mt].z; dbs.w += dbacc[mt].w; }        // (rows past B repeat row B - 1)
#pragma unroll
    for (int o = 8; o > 0; o >>= 1) {        // the 16 rows of a unit are 16 consecutive lanes
      dbs.x += __shfl_xor(dbs.x, o); dbs.y += __shfl_xor(dbs.y, o);
      dbs.z += __shfl_xor(dbs.z, o); dbs.w += __shfl_xor(dbs.w, o);
    }
    // One float atomic per column and workgroup row into the gradient arena.  With at most TWO contributions per element into a zeroed
    // buffer (B <= 32 with 16-row tiles, B <= 64 with 32-row tiles) the sum does not depend on their order: bit-reproducible, which the
    // last-arrival regression test relies on.  With more workgroup rows, or gradients accumulated over several calls, the order of the adds
    // varies from run to run and the bias gradients are reproducible only to float rounding (like every split tile of the batched
    // products) -- unless the call is `deterministic` (db_part below).
    if (r == 0) {
      if (c.db_part) {      // deterministic: this workgroup row's sums to a scratch row; the host-side fold adds the rows in order
        *reinterpret_cast<float4*>(c.db_part + (long)by * 4 * h + 4 * eu) = dbs;
      } else {
        atomicAdd(c.db + 4 * eu, dbs.x); atomicAdd(c.db + 4 * eu + 1, dbs.y);
        atomicAdd(c.db + 4 * eu + 2, dbs.z); atomicAdd(c.db + 4 * eu + 3, dbs.w);
      }
    }
  }
  if (c.amax) {
    // max |dz| of the cell for the batched products behind this launch: block maximum, one 64-bit atomic into one of 16 shards
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) dzmax = fmaxf(dzmax, __shfl_xor(dzmax, o));
    __syncthreads();
    if (lane == 0) dzS[wave] = dzmax;
    __syncthreads();
    if (tid == 0) {
      float m = fmaxf(fmaxf(dzS[0], dzS[1]), fmaxf(dzS[2], dzS[3]));
      if (!(m <= 3.0e38f)) m = 3.0e38f;
      atomicMax(amax_shard(c.amax, blockIdx.x + by * gridDim.x), ((u64)amax_gen << 32) | (u64)__float_as_uint(m));
    }
  }
  if (timing && tid == 0 && blockIdx.x == 0 && blockIdx.y == 0)
    printf("bwd_rs cell %d (layer %d): per-step 10ns: up wait+loads %lld  own wait %lld  own loads+sum %lld  epilogue %lld  barrier+mfma1+stores %lld  publishA %lld  product2+publishB %lld\n",
           cell, c.layer, tk[0] / T, tk[1] / T, tk[2] / T, tk[3] / T, tk[4] / T, tk[5] / T, tk[6] / T);
#undef TICK
}

template <int KB>
__global__ __launch_bounds__(512, 1) void lstm_persist_bwd_duo(PBwdArgs a) {
  __shared__ __attribute__((aligned(16))) float dzS2[2][2][16 * DZ_LD];
  __shared__ int s_ok1[2][2], s_ok2[2];
  __shared__ __attribute__((aligned(16))) u32q lo_lds[(LO_LDS_FRAGS / 2) * 256];
  static_assert(2 * KB * 2 <= LO_LDS_FRAGS / 2, "lo plane of both products in 64 KB");
  const PCellB c = a.c[blockIdx.z];
  const int half = threadIdx.x >> 8;
  if (c.PD_up != nullptr) lstm_bwd_rs_steps<KB, true, 4, 1, true>(a, c, dzS2[half], s_ok1[half], s_ok2[half], lo_lds);
  else lstm_bwd_rs_steps<KB, false, 4, 1, true>(a, c, dzS2[half], s_ok1[half], s_ok2[half], lo_lds);
}

template <int KB, int XS, int MT>
__global__ __launch_bounds__(256, 1) void lstm_persist_bwd_rs(PBwdArgs a) {
  // (two copies of the dz tile, used alternately: with the sentinel hand-off the step's ONE barrier sits between a tile's writes and its
  //  reads, and only the copy keeps a wave that is a step ahead from writing into what a slower wave still reads; the flags likewise)
  __shared__ __attribute__((aligned(16))) float dzS2[2][MT * 16 * DZ_LD];
  __shared__ int s_ok1[2], s_ok2;
  __shared__ __attribute__((aligned(16))) u32q lo_lds[XS == 4 ? LO_LDS_FRAGS * 256 : 1];
  const PCellB c = a.c[blockIdx.z];      // a copy (see lstm_persist_fwd_g)
  if (c.PD_up != nullptr) lstm_bwd_rs_steps<KB, true, XS, MT>(a, c, dzS2, s_ok1, s_ok2, lo_lds);
  else lstm_bwd_rs_steps<KB, false, XS, MT>(a, c, dzS2, s_ok1, s_ok2, lo_lds);
}

}  // namespace

// ---- launchers (called from lstm.hip).  Return 1 if the persistent path is not applicable (caller falls back).
struct PersistCellHost {
  const float *Wl, *Wu, *bias, *zx, *xin, *mask, *WlT, *d_enc, *d_hT, *d_cT;
  float *gates, *C, *HR, *HD, *enc;
  const float* WuT;               // backward: this cell's transposed upward weight (layers >= 1)
  float *PR, *PD;                 // backward, reduce-scatter path: partial-sum buffers of this cell
  const float* PD_up;
  int up_external;
  int reverse_pos, layer;
  unsigned long long* amax;       // backward: where max |dz| of the cell goes (16 sharded words, gemm_amax_reserve), null: not wanted
  float* db;                      // backward: bias gradient accumulated by the recurrence kernel itself (null: not wanted)
  long dy_sb, dy_st;              // backward: strides of d_enc (see PCellB)
  const unsigned* zx_flags; int zx_s0, zx_cs;      // forward, layer 0: chunk flags of the input projection (see PCellF)
  unsigned* prog; int prog_cs;                     // backward, layer 0: progress counter for side-stream consumers of dz (see PCellB)
  float* db_part;                                  // backward: deterministic bias-gradient scratch (see PCellB)
};

// Layers per launch.  One workgroup per CU must hold a launch's whole grid; a stack with more (direction, layer) cells than fit is
// run as consecutive launches over groups of layers (all directions of `lpl` layers each): the wavefront overlap between the
// groups is lost, everything else stays (BASELINE configs[4]: 6 layers x 2 directions x 32 unit slices x 2 batch tiles = 768
// workgroups -> 3 launches of 2 layers; batch 64 at the shipped width: 2 launches).  0 = not applicable.
bool lstm_persist_hoisted(int h);
// Form of the recurrence workgroups (the `rows` argument of everything below): 16 = one 16-row batch tile per 256-thread workgroup;
// 33 = DUO: two 16-row tiles per 512-thread workgroup as two virtual workgroups, two waves per SIMD (bf16x3 arithmetic, h <= 256);
// 32 = MT 2: two tiles per 256-thread workgroup against one set of weight fragments (any arithmetic, h <= 256).  Both 32-row forms halve
// the grid -- batch 64 at the shipped width runs as ONE launch, a batch of 32 leaves 160 CUs free for side-stream work instead of 64.
// What they cost (MEASURED, profiles/r6_ab_side.txt): MT 2 does both tiles' splits, MFMAs, LDS reduction and gate epilogue one after the
// other -- a step is only ~40 % hand-off latency -- 4.4 / 5.3 us per step against 2.9 / 3.6; DUO overlaps one tile's vector-ALU work with the
// other's MFMAs and hand-off wait.  Knob lstm.rows32: -1 (default) = DUO where available when it spares launches or the caller runs
// side-stream work, MT 2 when it spares launches and DUO is not available; 0 = always 16; 1 = MT 2 / 2 = DUO whenever possible.
static bool duo_available(int h) { return h <= 256 && gemm_precision_mode() == 1 && tune_on(TUNE_LSTM_X3) && tune_on(TUNE_LSTM_X4); }
int lstm_persist_rows(int B, int h, int nl, int nd, bool side) {
  const int knob = (int)tune(TUNE_LSTM_ROWS32);
  if (knob == 0 || h > 256 || B <= 16) return 16;
  if (knob == 1) return 32;
  if (knob >= 2) return duo_available(h) ? 33 : 16;
  const long cus = device_cu_count();
  const long wg16 = (long)(h / 16) * ((B + 15) / 16) * nd * nl, wg32 = (long)(h / 16) * ((B + 31) / 32) * nd * nl;
  const long launches16 = (wg16 + cus - 1) / cus, launches32 = (wg32 + cus - 1) / cus;
  if (launches32 < launches16) return duo_available(h) ? 33 : 32;
  return side && duo_available(h) && tune_on(TUNE_LSTM_DUO_SIDE) ? 33 : 16;
}
int lstm_persist_layers_per_launch(int B, int h, int nl, int nd, int rows) {
  const int rw = rows == 16 ? 16 : 32;
  const long per_layer = (long)(h / 16) * ((B + rw - 1) / rw) * nd;
  const long cus = device_cu_count();
  if (per_layer < 1 || per_layer > cus) return 0;
  long lpl = cus / per_layer;
  if (lstm_persist_hoisted(h) && lpl > 1) lpl = 1;
  if (lpl > nl) lpl = nl;
  while (lpl * nd > 16) --lpl;
  return (int)lpl;
}

// workgroups of one launch over `layers` layers of all directions
int lstm_persist_grid_wgs(int B, int h, int layers, int nd, int rows) { const int rw = rows == 16 ? 16 : 32; return (h / 16) * ((B + rw - 1) / rw) * nd * layers; }

// Hoisted form (h = 1024): the weight fragments of one product fill a workgroup's registers, so every layer runs as a launch of its
// own over cells that get their input projection from a batched GEMM in front of it (forward) and leave the gradient for the layer below
// to a batched GEMM behind it (backward) -- the structure of the per-step path with T launches per layer replaced by one.
bool lstm_persist_hoisted(int h) { return h >= 1024; }

bool lstm_persist_applicable(int T, int B, int h, int nl, int nd) {
  if (!(h == 64 || h == 128 || h == 256 || h == 512 || h == 1024)) return false;
  if (B < 1 || T < 1) return false;
  if (lstm_persist_hoisted(h)) {
    if (!tune_on(TUNE_LSTM_HOIST)) return false;
  }
  if (lstm_persist_layers_per_launch(B, h, nl, nd, lstm_persist_rows(B, h, nl, nd, false)) < 1) return false;
  // hand-off buffers are addressed with 32-bit byte offsets
  if ((long)T * B * h * 16 >= (1L << 31) || (long)T * (2 * ((B + 31) / 32)) * (h / 16) * (h / 16) * 1024 >= (1L << 31)) return false;
  if (!tune_on(TUNE_LSTM_PERSIST)) return false;
  return true;
}

int lstm_persist_fwd_launch(const PersistCellHost* cells, int ncells, int nl, int T, int B, int h, int H, unsigned* counters, int rows,
                            hipStream_t s) {
  PFwdArgs a;
  memset(&a, 0, sizeof(a));
  ASTK_CHECK(rows == 16 || ((rows == 32 || rows == 33) && h <= 256), "lstm_persist_fwd: form %d at h = %d", rows, h);
  const bool duo = rows == 33;
  const int nby = (B + (rows == 16 ? 16 : 32) - 1) / (rows == 16 ? 16 : 32);      // workgroup rows of the grid
  const int nbt = duo ? 2 * nby : nby;                                            // (virtual) workgroup rows the counters are laid out for
  for (int i = 0; i < ncells; ++i) {
    const PersistCellHost& c = cells[i];
    PCellF& d = a.c[i];
    d.Wl = c.Wl; d.Wu = c.Wu; d.bias = c.bias; d.zx = c.zx; d.gates = c.gates; d.C = c.C; d.HR = c.HR; d.HD = c.HD;
    d.xin = c.xin; d.mask = c.mask; d.enc = c.enc; d.reverse_pos = c.reverse_pos; d.layer = c.layer;
    d.zx_flags = c.zx_flags; d.zx_s0 = c.zx_s0; d.zx_cs = c.zx_cs;
  }
  a.ncells = ncells; a.nl = nl; a.T = T; a.B = B; a.h = h; a.H = H;
  a.dbg = persist_dbg_env();
  a.done = counters;
  a.ab = abort_ctl(counters + (size_t)ncells * nbt * 64, PERSIST_ENC_FWD);
  dim3 grid(h / 16, nby, ncells), blk(duo ? 512 : 256);
  {
    // hand-off buffers = the saved activations themselves: sentinel-filled before every launch (the counters / abort word ride along, zeroed)
    FillSegs f;
    f.n = 0;
    fill_seg_add(f, counters, ((size_t)ncells * nbt + 1) * 64 * sizeof(unsigned), 0u);
    for (int i = 0; i < ncells; ++i) {
      if (f.n + 2 > FILL_SEG_MAX) { ASTK_TRY(fill_u32_segments(f, 0xffffffffu, s)); f.n = 0; }
      fill_seg_add(f, cells[i].HR, (size_t)T * B * h * sizeof(float));
      if (cells[i].HD) fill_seg_add(f, cells[i].HD, (size_t)T * B * h * sizeof(float));
    }
    ASTK_TRY(fill_u32_segments(f, 0xffffffffu, s));
  }
  ProfScope prof(PROF_CELL, s);
  // arithmetic of the recurrences' products: the mode in force for this call (fp16x2: two-term fp16 splits; bf16x3: three-term bf16 splits
  // where the weight fragments fit the registers, h <= 256; f32, or bf16x3 at h = 512: f32 MFMAs).  "lstm.x3" 0 keeps f32 MFMAs under bf16x3.
  const int mode = gemm_precision_mode();
  const bool x3_off = !tune_on(TUNE_LSTM_X3), x4_off = !tune_on(TUNE_LSTM_X4);
  const int xs = mode == 0 ? 2 : (mode == 1 && !x3_off ? (h <= 256 ? 3 : (x4_off ? 0 : 4)) : 0);       // (4: bf16x3 with the weights' lo plane in LDS)
  if (duo) {
    ASTK_CHECK(xs == 3 && !x4_off, "lstm_persist_fwd: the two-waves-per-SIMD form needs the bf16x3 arithmetic");
    switch (h) {
      case 64: hipLaunchKernelGGL((lstm_persist_fwd_duo<1>), grid, blk, 0, s, a); break;
      case 128: hipLaunchKernelGGL((lstm_persist_fwd_duo<2>), grid, blk, 0, s, a); break;
      default: hipLaunchKernelGGL((lstm_persist_fwd_duo<4>), grid, blk, 0, s, a); break;
    }
    ASTK_LAUNCH_CHECK();
    return 0;
  }
#define ASTK_LSTM_FWD_(KB_, XS_, MT_) hipLaunchKernelGGL((lstm_persist_fwd_g<KB_, XS_, MT_>), grid, blk, 0, s, a)
#define ASTK_LSTM_FWD_XS_(KB_, MT_) { if (xs == 2) ASTK_LSTM_FWD_(KB_, 2, MT_); else if (xs == 3) ASTK_LSTM_FWD_(KB_, 3, MT_); else ASTK_LSTM_FWD_(KB_, 0, MT_); }
  switch (h) {
    case 64: if (rows == 32) ASTK_LSTM_FWD_XS_(1, 2) else ASTK_LSTM_FWD_XS_(1, 1) break;
    case 128: if (rows == 32) ASTK_LSTM_FWD_XS_(2, 2) else ASTK_LSTM_FWD_XS_(2, 1) break;
    case 256: if (rows == 32) ASTK_LSTM_FWD_XS_(4, 2) else ASTK_LSTM_FWD_XS_(4, 1) break;
    case 512: if (xs == 2) ASTK_LSTM_FWD_(8, 2, 1); else if (xs == 4) ASTK_LSTM_FWD_(8, 4, 1); else ASTK_LSTM_FWD_(8, 0, 1); break;
    default: if (xs == 2) ASTK_LSTM_FWD_(16, 2, 1); else if (xs == 4) ASTK_LSTM_FWD_(16, 4, 1); else ASTK_LSTM_FWD_(16, 0, 1); break;
  }
#undef ASTK_LSTM_FWD_XS_
#undef ASTK_LSTM_FWD_
  ASTK_LAUNCH_CHECK();
  return 0;
}

size_t lstm_persist_pr_floats(int B, int h);
int lstm_persist_bwd_launch(const PersistCellHost* cells, int ncells, int nl, int T, int B, int h, int H, unsigned* counters,
                            unsigned amax_gen, int rows, hipStream_t s) {
  PBwdArgs a;
  memset(&a, 0, sizeof(a));
  ASTK_CHECK(rows == 16 || ((rows == 32 || rows == 33) && h <= 256), "lstm_persist_bwd: form %d at h = %d", rows, h);
  const bool duo = rows == 33;
  const int nby = (B + (rows == 16 ? 16 : 32) - 1) / (rows == 16 ? 16 : 32);      // workgroup rows of the grid
  const int nbt = duo ? 2 * nby : nby;                                            // (virtual) workgroup rows: the counters are per (virtual) workgroup row
  for (int i = 0; i < ncells; ++i) {
    const PersistCellHost& c = cells[i];
    PCellB& d = a.c[i];
    d.Wl = c.Wl; d.gates_dz = c.gates; d.C = c.C; d.mask = c.mask; d.d_enc = c.d_enc;
    d.d_hT = c.d_hT; d.d_cT = c.d_cT; d.reverse_pos = c.reverse_pos; d.layer = c.layer;
    d.dy_sb = c.dy_sb; d.dy_st = c.dy_st;
    d.Wu = c.PD ? c.Wu : nullptr; d.PR = c.PR; d.PD = c.PD; d.PD_up = c.PD_up; d.up_external = c.up_external;
    d.amax = (u64*)c.amax;
    d.db = c.db;
    d.db_part = c.db_part;
    d.prog = c.prog; d.prog_cs = c.prog_cs > 0 ? c.prog_cs : 1;
    ASTK_CHECK(!c.prog || c.prog_cs >= 4, "lstm_persist_bwd: progress chunks of %d steps (the arrivals' one-apart invariant needs >= 4)", c.prog_cs);
  }
  a.ncells = ncells; a.nl = nl; a.T = T; a.B = B; a.h = h; a.H = H;
  a.amax_gen = amax_gen;
  a.done = counters;
  a.dbg = persist_dbg_env();
  ASTK_CHECK(cells[0].PR != nullptr, "lstm_persist_bwd: partial-sum buffers missing");
  // counters A and B per (cell, batch tile), then the abort word
  a.ab = abort_ctl(counters + (size_t)2 * ncells * nbt * 64, PERSIST_ENC_BWD);
  if (bwd_sentinel(h / 64)) {
    // the partial dh_rec rings are hand-off buffers of the sentinel kind: filled before every launch (the counters / abort word ride along, zeroed)
    FillSegs f;
    f.n = 0;
    fill_seg_add(f, counters, ((size_t)2 * ncells * nbt + 1) * 64 * sizeof(unsigned), 0u);
    for (int i = 0; i < ncells; ++i) {
      if (f.n + 1 > FILL_SEG_MAX) { ASTK_TRY(fill_u32_segments(f, 0xffffffffu, s)); f.n = 0; }
      fill_seg_add(f, cells[i].PR, lstm_persist_pr_floats(B, h) * sizeof(float));
    }
    ASTK_TRY(fill_u32_segments(f, 0xffffffffu, s));
  } else {
    ASTK_HIP(hipMemsetAsync(counters, 0, ((size_t)2 * ncells * nbt + 1) * 64 * sizeof(unsigned), s));
  }
  dim3 grid(h / 16, nby, ncells), blk(duo ? 512 : 256);
  ProfScope prof(PROF_CELL, s);
  const int mode = gemm_precision_mode();      // (see lstm_persist_fwd_launch)
  const bool x3_off = !tune_on(TUNE_LSTM_X3), x4_off = !tune_on(TUNE_LSTM_X4);
  const int xs = mode == 0 ? 2 : (mode == 1 && !x3_off ? (h <= 256 ? 3 : (x4_off ? 0 : 4)) : 0);
  if (duo) {
    ASTK_CHECK(xs == 3 && !x4_off, "lstm_persist_bwd: the two-waves-per-SIMD form needs the bf16x3 arithmetic");
    switch (h) {
      case 64: hipLaunchKernelGGL((lstm_persist_bwd_duo<1>), grid, blk, 0, s, a); break;
      case 128: hipLaunchKernelGGL((lstm_persist_bwd_duo<2>), grid, blk, 0, s, a); break;
      default: hipLaunchKernelGGL((lstm_persist_bwd_duo<4>), grid, blk, 0, s, a); break;
    }
    ASTK_LAUNCH_CHECK();
    return 0;
  }
#define ASTK_LSTM_BWD_(KB_, XS_, MT_) hipLaunchKernelGGL((lstm_persist_bwd_rs<KB_, XS_, MT_>), grid, blk, 0, s, a)
#define ASTK_LSTM_BWD_XS_(KB_, MT_) { if (xs == 2) ASTK_LSTM_BWD_(KB_, 2, MT_); else if (xs == 3) ASTK_LSTM_BWD_(KB_, 3, MT_); else ASTK_LSTM_BWD_(KB_, 0, MT_); }
  switch (h) {
    case 64: if (rows == 32) ASTK_LSTM_BWD_XS_(1, 2) else ASTK_LSTM_BWD_XS_(1, 1) break;
    case 128: if (rows == 32) ASTK_LSTM_BWD_XS_(2, 2) else ASTK_LSTM_BWD_XS_(2, 1) break;
    case 256: if (rows == 32) ASTK_LSTM_BWD_XS_(4, 2) else ASTK_LSTM_BWD_XS_(4, 1) break;
    case 512: if (xs == 2) ASTK_LSTM_BWD_(8, 2, 1); else if (xs == 4) ASTK_LSTM_BWD_(8, 4, 1); else ASTK_LSTM_BWD_(8, 0, 1); break;
    default: if (xs == 2) ASTK_LSTM_BWD_(16, 2, 1); else if (xs == 4) ASTK_LSTM_BWD_(16, 4, 1); else ASTK_LSTM_BWD_(16, 0, 1); break;
  }
#undef ASTK_LSTM_BWD_XS_
#undef ASTK_LSTM_BWD_
  ASTK_LAUNCH_CHECK();
  return 0;
}

// bytes of the reduce-scatter partial buffers of one cell (lstm.hip sizes the workspace with these)
// (sized for an EVEN number of 16-row tiles: a 32-row workgroup addresses tiles 2 by and 2 by + 1 whether the second one has rows or not)
size_t lstm_persist_pr_floats(int B, int h) { return (size_t)PR_RING * (2 * ((B + 31) / 32)) * (h / 16) * (h / 16) * 256; }
size_t lstm_persist_pd_floats(int T, int B, int h) { return (size_t)T * (2 * ((B + 31) / 32)) * (h / 16) * (h / 16) * 256; }

}  // namespace astk
