// Internal helpers shared by the libastk translation units (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stddef.h>
#include <stdio.h>
#include <string.h>
#include "../../include/astk.h"

namespace astk {

void set_error(const char* fmt, ...);

// Optional per-kernel HIP-event timing on the launch stream (bench.py's roofline legs).  Off by default.
enum ProfCat { PROF_ATTN_FWD = 0, PROF_ATTN_BWD = 1, PROF_GEMM = 2, PROF_CELL = 3, PROF_DEC_FWD = 4, PROF_DEC_BWD = 5, PROF_NCAT = 6 };
bool prof_enabled();
float* prof_tick_buffer(int which);   // device buffer [257] for in-kernel phase timing (0: decoder fwd, 1: decoder bwd) or null
void prof_start(int cat, hipStream_t s, double work);
void prof_stop(int cat, hipStream_t s);
void prof_add_bytes(int cat, double bytes);
struct ProfScope {
  int cat; hipStream_t s; bool on;
  ProfScope(int c, hipStream_t st, double work = 0.0) : cat(c), s(st), on(prof_enabled()) { if (on) prof_start(cat, s, work); }
  ~ProfScope() { if (on) prof_stop(cat, s); }
};

#define ASTK_CHECK(cond, ...)            \
  do {                                   \
    if (!(cond)) {                       \
      astk::set_error(__VA_ARGS__);      \
      return -1;                         \
    }                                    \
  } while (0)

// Every descriptor starts with struct_size (astk.h): a caller built against another layout fails here instead of handing over garbage pointers.
#define ASTK_CHECK_DESC(d, type)                                                                                                        \
  ASTK_CHECK((d) != nullptr && (d)->struct_size == sizeof(type),                                                                         \
             #type ": null descriptor or struct_size %zu != %zu (zero-initialise the struct, set struct_size = sizeof(" #type "), astk.h)", \
             (d) ? (size_t)(d)->struct_size : (size_t)0, sizeof(type))

#define ASTK_HIP(expr)                                                               \
  do {                                                                               \
    hipError_t e_ = (expr);                                                          \
    if (e_ != hipSuccess) {                                                          \
      astk::set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, __LINE__); \
      return -2;                                                                     \
    }                                                                                \
  } while (0)

#define ASTK_LAUNCH_CHECK()                                                          \
  do {                                                                               \
    hipError_t e_ = hipGetLastError();                                               \
    if (e_ != hipSuccess) {                                                          \
      astk::set_error("kernel launch failed: %s (%s:%d)", hipGetErrorString(e_), __FILE__, __LINE__); \
      return -3;                                                                     \
    }                                                                                \
  } while (0)

#define ASTK_TRY(expr)        \
  do {                        \
    int r_ = (expr);          \
    if (r_ != 0) return r_;   \
  } while (0)

// ---- bounded spins of the persistent kernels (lstm_persist.hip, decoder_persist.hip)
// A spin that times out raises the launch's own abort word (in the caller's workspace, zeroed before every launch: every other
// spin of the grid checks it and the grid drains) AND sets this kernel's bit in the library's sticky status word, which the host
// reads back next to the loss (astk_persist_status_snapshot) -- a step whose grid was not fully resident must not train on.
enum PersistBit { PERSIST_ENC_FWD = 1, PERSIST_ENC_BWD = 2, PERSIST_DEC_FWD = 4, PERSIST_DEC_BWD = 8 };
struct AbortCtl {
  unsigned* word;     // per-launch abort word
  unsigned* status;   // sticky status word of the library (device memory owned by util.hip)
  unsigned limit;     // spin bound (polls); astk_set_tuning("persist.spin_limit") overrides the default of 1 << 22 (seconds)
  unsigned bit;       // PersistBit of the launching kernel
};
AbortCtl abort_ctl(unsigned* word, unsigned bit);     // host side: fills status / limit
int device_cu_count();                                 // multiProcessorCount of the current device (cached per device)
// Tuning knobs (astk_set_tuning / astk_get_tuning in astk.h: documented, process-wide, read at every launch -- they replace the environment
// variables the library read in rounds 1-5; the product library reads NO environment variable).
enum TuneKey {
  TUNE_GEMM_TILE, TUNE_GEMM_T256_ABOVE, TUNE_GEMM_GRID, TUNE_GEMM_HYBRID, TUNE_GEMM_CHUNK, TUNE_GEMM_CHUNK_DIV, TUNE_GEMM_LOG, TUNE_GEMM_TICKET,
  TUNE_GEMM_DETERMINISTIC, TUNE_GEMM_FORWARD_PAIRS,
  TUNE_CONV_DIRECT0, TUNE_CONV_SEQ_FWD, TUNE_CONV_SEQ_BWD, TUNE_CONV_SEQ_STATS_BLOCKS, TUNE_CONV_SEQ_APPLY_BLOCKS,
  TUNE_DEC_PERSIST, TUNE_DEC_B6_SPLIT, TUNE_DEC_B6_FUSED, TUNE_DEC_WIDE,
  TUNE_LSTM_PERSIST, TUNE_LSTM_HOIST, TUNE_LSTM_X3, TUNE_LSTM_X4, TUNE_LSTM_ROWS32, TUNE_LSTM_OVERLAP_CHUNK, TUNE_LSTM_SIDE_FWD, TUNE_LSTM_SIDE_BWD, TUNE_LSTM_DUO_SIDE,
  TUNE_ROW_LONGK, TUNE_PERSIST_SPIN_LIMIT, TUNE_COLREDUCE_BLOCKS,
  TUNE_COUNT
};
double tune(TuneKey k);
inline bool tune_on(TuneKey k) { return tune(k) != 0.0; }
__device__ __forceinline__ bool abort_seen(const AbortCtl& ab) {
  return __hip_atomic_load(ab.word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0;
}
__device__ __forceinline__ void abort_raise(const AbortCtl& ab) {
  __hip_atomic_store(ab.word, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  __hip_atomic_fetch_or(ab.status, ab.bit, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

constexpr int AMAX_SLOT_WORDS = 16;
// Producer slots of kernels with thousands of emitting blocks keep their 16 shards on SEPARATE 128-byte lines (same-line atomics of
// thousands of blocks retire one after the other in one L2 channel: with adjacent shards those kernels ran 10-30 us longer than
// without).  Such a slot is 16 x 16 words and its handle carries bit 0 set (pointers are 8-byte aligned).  A consuming GEMM reads a
// slot at the start of every tile, and 16 separate lines THERE cost more than the producers save (0.18 ms per train step): before a
// GEMM sees it a strided slot is folded into one word of a plain slot by a small kernel that runs in between anyway (amax_compact).
constexpr int AMAX_PSLOT_STRIDE = 16;
constexpr int AMAX_PSLOT_WORDS = AMAX_SLOT_WORDS * AMAX_PSLOT_STRIDE;
static inline unsigned long long* amax_pslot_handle(unsigned long long* p) { return p ? (unsigned long long*)((uintptr_t)p | 1u) : nullptr; }
// word of shard `i` behind a handle of either kind (device and host)
static inline __host__ __device__ unsigned long long* amax_shard(const unsigned long long* handle, int i) {
  unsigned long long* base = (unsigned long long*)((uintptr_t)handle & ~(uintptr_t)1);
  return base + (i & (AMAX_SLOT_WORDS - 1)) * (((uintptr_t)handle & 1u) ? AMAX_PSLOT_STRIDE : 1);
}

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

// ---- fp16x2 products (device side; gemm.hip describes the scheme): an f32 value times a power-of-two scale is split into two fp16
// terms, x s = hi + lo, and a product is summed from lo.hi + hi.lo + hi.hi on the fp16 MFMAs with f32 accumulation.
#if defined(__HIPCC__)
typedef _Float16 h16x2 __attribute__((ext_vector_type(2)));
typedef _Float16 h16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32q __attribute__((ext_vector_type(4)));
typedef float f32pair __attribute__((ext_vector_type(2)));
// two values -> one dword of the hi plane and one of the lo plane (5 vector-ALU instructions: v_pk_mul_f32, v_cvt_pk_f16_f32, two
// v_fma_mix_f32 reading the fp16 halves in place, v_cvt_pk_f16_f32)
__device__ __forceinline__ void split2h(float x0, float x1, float scl, unsigned& hi, unsigned& lo) {
  const f32pair xs = (f32pair){x0, x1} * (f32pair){scl, scl};
  const h16x2 h = {(_Float16)xs[0], (_Float16)xs[1]};
  hi = __builtin_bit_cast(unsigned, h);
  float r0, r1;
  asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel_hi:[1,0,0]" : "=v"(r0) : "v"(hi), "v"(xs[0]));
  asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(r1) : "v"(hi), "v"(xs[1]));
  const h16x2 l = {(_Float16)r0, (_Float16)r1};
  lo = __builtin_bit_cast(unsigned, l);
}
// eight values (two float4: k = 0..3 and 4..7 of a lane's slice of a v_mfma_f32_16x16x32_f16 operand) -> hi / lo fragments
struct HL8 { u32q hi, lo; };
__device__ __forceinline__ HL8 split8(const float4& a, const float4& b, float scl) {
  HL8 o;
  unsigned h, l;
  split2h(a.x, a.y, scl, h, l); o.hi.x = h; o.lo.x = l;
  split2h(a.z, a.w, scl, h, l); o.hi.y = h; o.lo.y = l;
  split2h(b.x, b.y, scl, h, l); o.hi.z = h; o.lo.z = l;
  split2h(b.z, b.w, scl, h, l); o.hi.w = h; o.lo.w = l;
  return o;
}
// acc += A B over 32 k on v_mfma_f32_16x16x32_f16, three term products (smallest first)
#define MFMA32H(ACC, A, W)                                                                                                                    \
  ACC = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(h16x8, (A).lo), __builtin_bit_cast(h16x8, (W).hi), ACC, 0, 0, 0);          \
  ACC = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(h16x8, (A).hi), __builtin_bit_cast(h16x8, (W).lo), ACC, 0, 0, 0);          \
  ACC = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(h16x8, (A).hi), __builtin_bit_cast(h16x8, (W).hi), ACC, 0, 0, 0);
// ---- bf16x3 products (device side; gemm.hip describes the scheme): an f32 value is split into three bf16 terms, x = hi + mid + lo
// (round-to-nearest-even each: the three terms represent every normal f32 value EXACTLY, no scale needed: bf16 has f32's exponent
// range), and a product is summed from the six largest term products on the bf16 MFMAs with f32 accumulation (the dropped ones are
// <= 2^-26 of the product).
typedef __bf16 b16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 b16x2 __attribute__((ext_vector_type(2)));
// v_cvt_pk_bf16_f32 (round-to-nearest-even), through the compiler's own lowering of the vector conversion -- NOT inline asm: behind an
// asm statement hipcc does not know that a vector-ALU instruction wrote the register and inserts ONE wait state in front of an MFMA
// that reads it where the hardware needs TWO (the MFMA then reads the register's previous content: seen as run-to-run varying errors
// of the h = 64 recurrence kernels, whose last conversion sits directly in front of the first MFMA).
__device__ __forceinline__ unsigned cvt_pk_bf16_f32(float lo, float hi) {
  return __builtin_bit_cast(unsigned, __builtin_convertvector((f32pair){lo, hi}, b16x2));
}
// two values -> one dword of each plane (element 0 in the low half): 9 vector-ALU instructions (three v_cvt_pk_bf16_f32, two shifts and two
// ands to widen the packed terms again, two v_pk_add_f32 for the residuals)
__device__ __forceinline__ void split2b(float x0, float x1, unsigned& hi, unsigned& mid, unsigned& lo) {
  hi = cvt_pk_bf16_f32(x0, x1);
  const f32pair r = (f32pair){x0, x1} - (f32pair){__uint_as_float(hi << 16), __uint_as_float(hi & 0xffff0000u)};
  mid = cvt_pk_bf16_f32(r[0], r[1]);
  const f32pair q = r - (f32pair){__uint_as_float(mid << 16), __uint_as_float(mid & 0xffff0000u)};
  lo = cvt_pk_bf16_f32(q[0], q[1]);
}
// eight values (two float4: k = 0..3 and 4..7 of a lane's slice of a v_mfma_f32_16x16x32_bf16 operand) -> hi / mid / lo fragments
struct HML8 { u32q hi, mid, lo; };
__device__ __forceinline__ HML8 split8b(const float4& a, const float4& b) {
  HML8 o;
  unsigned h, m, l;
  split2b(a.x, a.y, h, m, l); o.hi.x = h; o.mid.x = m; o.lo.x = l;
  split2b(a.z, a.w, h, m, l); o.hi.y = h; o.mid.y = m; o.lo.y = l;
  split2b(b.x, b.y, h, m, l); o.hi.z = h; o.mid.z = m; o.lo.z = l;
  split2b(b.z, b.w, h, m, l); o.hi.w = h; o.mid.w = m; o.lo.w = l;
  return o;
}
#define MFMA_B16_(ACC, X, Y) ACC = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(b16x8, X), __builtin_bit_cast(b16x8, Y), ACC, 0, 0, 0);
// acc += A B over 32 k on v_mfma_f32_16x16x32_bf16, six term products (smallest first: lo.hi, hi.lo, mid.mid, mid.hi, hi.mid, hi.hi)
#define MFMA32B(ACC, A, W)        \
  MFMA_B16_(ACC, (A).lo, (W).hi)  \
  MFMA_B16_(ACC, (A).hi, (W).lo)  \
  MFMA_B16_(ACC, (A).mid, (W).mid) \
  MFMA_B16_(ACC, (A).mid, (W).hi) \
  MFMA_B16_(ACC, (A).hi, (W).mid) \
  MFMA_B16_(ACC, (A).hi, (W).hi)
// The power of two (as a float) that brings an absolute maximum into [2^14, 2^15), and its reciprocal; 1 for a zero maximum.
__device__ __forceinline__ float pow2_scale_for(float amax, float& inv) {
  const int e = (int)(__float_as_uint(amax) >> 23) & 0xff;
  const int se = e == 0 ? 127 : min(max(268 - e, 1), 253);
  inv = __uint_as_float((unsigned)(254 - se) << 23);
  return __uint_as_float((unsigned)se << 23);
}
#endif

#if defined(__HIPCC__)
// Block-wide (256 threads) maximum of a per-thread NON-NEGATIVE value into a 16-word maximum slot: wave shuffles, 4 floats of LDS, then
// ONE sharded 64-bit atomicMax -- skipped when the shard already holds as much (a plain load: after the first blocks have published a
// large value most blocks skip, so the same-address atomics do not queue up).  Every thread of the block must call it.
__device__ __forceinline__ void amax_emit_block(unsigned long long* slot, float m, float* red4) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
  if ((threadIdx.x & 63) == 0) red4[threadIdx.x >> 6] = m;
  __syncthreads();
  if (threadIdx.x == 0) {
    m = fmaxf(fmaxf(red4[0], red4[1]), fmaxf(red4[2], red4[3]));
    if (!(m <= 3.0e38f)) m = 3.0e38f;
    // (slot: a producer-slot handle, bit 0 set, shards 16 words apart -- or a plain 16-word slot)
    const bool strided = ((uintptr_t)slot & 1u) != 0;
    unsigned long long* base = (unsigned long long*)((uintptr_t)slot & ~(uintptr_t)1);
    unsigned long long* w = base + ((blockIdx.x + blockIdx.y * gridDim.x) & 15) * (strided ? AMAX_PSLOT_STRIDE : 1);
    const unsigned long long v = (unsigned long long)__float_as_uint(m);
    if (__hip_atomic_load(w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < v) atomicMax(w, v);
  }
}
// Folds a strided producer slot into word 0 of a plain 16-word slot (the other words are zero): lanes 0..15 of the calling wave.
// Called by ONE wave of a kernel that runs between the producer and the consuming GEMM (stream order makes the producer's atomics visible).
__device__ __forceinline__ void amax_compact(const unsigned long long* strided_handle, unsigned long long* plain_slot) {
  const int lane = threadIdx.x & 63;
  unsigned long long w = lane < AMAX_SLOT_WORDS ? *amax_shard(strided_handle, lane) : 0ull;
#pragma unroll
  for (int o = 8; o > 0; o >>= 1) {
    const unsigned long long v = __shfl_xor(w, o);
    w = v > w ? v : w;
  }
  if (lane < AMAX_SLOT_WORDS) plain_slot[lane] = lane == 0 ? w : 0ull;
}
#endif

static inline size_t align_up(size_t x, size_t a) { return (x + a - 1) / a * a; }
static inline int cdiv(long a, long b) { return (int)((a + b - 1) / b); }
static inline bool aligned16(const void* p) { return (((uintptr_t)p) & 15) == 0; }

// Bump allocator over a caller-provided workspace (256-byte aligned slices).
struct Carver {
  char* base;
  size_t off;
  explicit Carver(void* p) : base((char*)p), off(0) {}
  template <typename T>
  T* take(size_t n) {
    off = align_up(off, 256);
    T* r = base ? (T*)(base + off) : nullptr;
    off += n * sizeof(T);
    return r;
  }
  size_t total() const { return align_up(off, 256); }
};

// Row-major matrix view with generalised row addressing:
//   rowoff(r) = rowidx ? rowidx[r]*ld : (tn > 0 ? (r / tn) * sg + (r % tn) * st : r * ld)
// (two-level rows express zero-copy im2col windows of the padded channels-last conv activations).
struct MatView {
  const float* p;
  long ld;
  int tn;
  long sg, st;
  const int* rowidx;
  long idx_rows;      // indexed rows: rows of the array p the indices point into (0: the indices are a permutation of the operand's own rows)
};
static inline MatView mat(const float* p, long ld) { return MatView{p, ld, 0, 0, 0, nullptr}; }
static inline MatView mat2(const float* p, int tn, long sg, long st) { return MatView{p, st, tn, sg, st, nullptr}; }
static inline MatView mat_idx(const float* p, long ld, const int* idx) { return MatView{p, ld, 0, 0, 0, idx}; }

enum GemmLayout { GEMM_NT = 0, GEMM_NN = 1, GEMM_TN = 2 };
enum GemmMode { GEMM_STORE = 0, GEMM_ACCUM = 1, GEMM_ATOMIC = 2 };

struct GemmArgs {
  MatView A, B;
  float* C;
  long ldc;
  int c_tn;
  long c_sg, c_st;  // two-level C rows when c_tn > 0
  const float* bias;
  int M, N, K;
  int mode;
  int ksplit;
  int batch;
  long sA, sB, sC;
  int lowp;          // 1: this product may run with fp16 operands when astk_set_low_precision_gemms(1) is in force (K6 / K9 and their backward)
  // filled by gemm_launch (work decomposition)
  const unsigned long long* amaxA;   // fp16x2 path: the operands' absolute-maximum words; null on entry = gemm_launch runs the pass itself
  const unsigned long long* amaxB;
  unsigned spanA, spanB;  // bytes from A.p / B.p to the end of one batch slice (buffer-load range)
  int kt;            // k-iterations per tile = ceil(K / BK)
  int tiles_mn;      // tiles per batch matrix
  long iters_total;  // batch * tiles_mn * kt
  int bm;            // tile order inside a batch matrix: super-rows of bm tile rows, inside a super-row column-major (tile t of a super-row
                     // = row t % bm of column t / bm), so that G / 8 consecutive tiles form a bm x (G / 8 / bm) block; 1 = row-major
  // Chunk-major iteration order (gemm.hip "few tiles, deep K"): cs > 0 = k-iterations per chunk (a divisor of kt); the launch's linear
  // iteration space then runs through ALL tiles of chunk 0, then all tiles of chunk 1, ... instead of all of tile 0's iterations first.
  int cs;
  long chunk_iters;  // tiles * cs
};
static inline GemmArgs gemm_args(int M, int N, int K, MatView A, MatView B, float* C, long ldc,
                                 const float* bias = nullptr, int mode = GEMM_STORE, int ksplit = 1) {
  GemmArgs g;
  memset(&g, 0, sizeof(g));
  g.A = A; g.B = B; g.C = C; g.ldc = ldc; g.bias = bias; g.M = M; g.N = N; g.K = K;
  g.mode = mode; g.ksplit = ksplit; g.batch = 1;
  return g;
}
static inline GemmArgs lowp(GemmArgs g) { g.lowp = 1; return g; }
static inline GemmArgs with_amax_a(GemmArgs g, const unsigned long long* a) { g.amaxA = a; return g; }
static inline GemmArgs with_amax_b(GemmArgs g, const unsigned long long* b) { g.amaxB = b; return g; }
int gemm_launch(int layout, const GemmArgs& g, hipStream_t s);
// The fp16x2 GEMMs scale every operand by a power of two taken from its absolute maximum, normally found by a pass in front of the
// launch.  A caller that feeds the same rows x inner matrix (row stride ld) to several launches runs that pass once, here, and hands
// the result to each of them with with_amax_a / with_amax_b (null when the GEMMs run in another precision: then nothing is needed).
// The handle stays valid for the next 511 calls of gemm_amax; the matrix must not change between the pass and the launches.  A
// maximum taken over a superset of the operand (more rows of the same buffer) is fine: it can only make the scale more conservative.
const unsigned long long* gemm_amax(const float* p, long rows, long ld, int inner, hipStream_t s);
// For a kernel that takes the maximum of a matrix while it writes it: n handles (16 sharded 64-bit words each) and the generation
// tag; the kernel does atomicMax(slot + (block & 15), (u64)gen << 32 | float_bits(block maximum)) from every block.
void gemm_amax_reserve(int n, unsigned long long** slots, unsigned* gen, hipStream_t s);
// A matrix whose entries are bounded by construction (LSTM outputs: |h| < 1, times a dropout scale) needs no pass either: a handle to
// a constant maximum `bound` (a power of two).  Cached per device; the first use of a bound enqueues its one-word initialisation on s.
const unsigned long long* gemm_amax_bound(float bound, hipStream_t s);
// Called by the first op of a train step (no maximum handle is live there): restarts the generation count of the maximum slots when
// it comes within reach of its 32-bit limit (see next_amax_gen in gemm.hip).
void gemm_amax_step_boundary(hipStream_t s);
// Kernels that WRITE a matrix a later GEMM reads can take its maximum on the way (amax_emit_block below): the slot is 16 64-bit words
// in the caller's workspace, zeroed before the producing kernel runs (the generation half of the word stays 0), and is handed to the
// GEMM with with_amax_a / with_amax_b like any other handle.
struct AmaxMatrix { const float* p; long rows, ld; int inner; };
void gemm_amax_many(const AmaxMatrix* m, int n, const unsigned long long** out, hipStream_t s);     // several matrices, one launch
int low_precision_gemms();      // fp16-operand mode in force for this thread's launches (descriptor, else astk_set_low_precision_gemms)
int gemm_precision_mode();      // arithmetic in force for this thread's launches: 0 fp16x2, 1 bf16x3, 2 f32 (descriptor, else the process default)
// The arithmetic a descriptor asks for (ASTK_PREC_* / ASTK_OPERANDS_*; 0 = process default), in force for the launches this thread makes
// while the scope lives.  Every C-ABI entry point that takes a descriptor opens one.
struct PrecScope {
  PrecScope(int precision, int operands);
  ~PrecScope();
  int prev_p, prev_l;
};
// Forward entry points (conv, LSTM stack, decoder forward) open one: the hybrid schedule then runs one data-parallel wave FEWER, so that the
// stream-K remainder holds at least as many tiles as workgroups and every split tile has at most TWO contributors -- two float atomic adds
// into a zeroed tile commute, three or more do not, and with all workgroups entering the remainder together their order varies from run
// to run: 1-ulp differences in the forward pass and, once in a few thousand batches, another fed-back argmax (found by the round-5 soak;
// rounds 1-4 had bit-identical losses between two passes over the same batches, which is what makes the soak a race detector).  Backward
// launches keep the full schedule: their weight-gradient products accumulate with many-contributor atomics anyway.
struct GemmForwardScope {
  GemmForwardScope();
  ~GemmForwardScope();
  int prev;
};
// Deterministic mode (descriptor field `deterministic`, or the process default "gemm.deterministic"): every accumulated sum of the calls made
// by this thread inside the scope is formed in a fixed order -- split tiles of the batched products through the fix-up workspace
// (gemm.hip, "Deterministic split tiles"), column reductions with one block per column group, the embedding scatter per token, the LSTM
// bias gradients through a per-workgroup-row scratch; no side-stream work (the fix-up workspace serves one launch at a time).  The
// backward entry points open one; a soak that compares two passes then sees ANY run-to-run difference as a defect.
struct DetScope {
  explicit DetScope(int requested);
  ~DetScope();
  int prev;
};
bool deterministic_mode();
// Scoped cap on the GRID of the GEMM launches made by this thread (workgroups; 0 = none): work that is meant to run BESIDE a persistent
// recurrence kernel, on a second stream, gets at most the CUs that kernel's grid leaves free -- whichever of the two is dispatched first,
// the recurrence grid still becomes resident (a workgroup of it needs a CU's whole register file, and stream-K workgroups live for the
// whole launch).  Capped launches take operand schemes that need no scale slots (the slots' ring is ordered by ONE stream).
struct GemmWgCap {
  explicit GemmWgCap(int wgs);
  ~GemmWgCap();
  int prev;
};
// Several independent products of the same layout in ONE launch (their k-iterations are concatenated and split evenly
// over the workgroups): used where a phase issues many small products that cannot fill the chip one at a time.
constexpr int GEMM_GROUP_MAX = 12;
struct GemmGroup {
  int n;
  int unit;          // granule of the stream-K split in k-iterations (1, or 2 when every kt is even)
  // Hybrid schedule (gemm.hip, "Work decomposition"): the first dp_waves * G tiles of the launch run data-parallel, one whole tile per
  // workgroup and wave, in XCD-local blocks; only the iterations from rem_start on are split stream-K.  dp_waves = 0: all stream-K.
  int dp_waves;
  int dp_kt;         // the launch's uniform k-iterations per tile (hybrid launches only)
  long rem_start;    // first k-iteration of the stream-K remainder = dp_waves * G * dp_kt
  long iters_total;
  long iter_start[GEMM_GROUP_MAX + 1];
  // Ticket words of the split tiles (gemm.hip, "Split tiles without a zeroing launch"): word (tick_base[problem] + tile) * 8 + wave;
  // nullptr: the split tiles are zeroed in front of the launch and every contribution is an atomic add
  unsigned* tick;
  int tick_base[GEMM_GROUP_MAX];
  GemmArgs g[GEMM_GROUP_MAX];
};
int gemm_launch_group(int layout, const GemmArgs* list, int n, hipStream_t s);

// The persistent kernels' sticky status word (util.hip) and a one-thread copy of it, as a float, to *dst (astk_persist_status_snapshot)
const unsigned* persist_status_word();
int status_snapshot_launch(float* dst, hipStream_t s);

// ---- small utility kernels (util.hip)
int fill_zero(void* p, size_t bytes, hipStream_t s);
// One launch that fills up to FILL_SEG_MAX separate 16-byte aligned regions with a 32-bit pattern (the sentinel fill of the
// persistent kernels' hand-off buffers; a hipMemsetAsync per buffer costs ~5 us each).
constexpr int FILL_SEG_MAX = 16;
struct FillSegs { int n; void* p[FILL_SEG_MAX]; size_t bytes[FILL_SEG_MAX]; unsigned val[FILL_SEG_MAX]; unsigned own[FILL_SEG_MAX]; const void* src[FILL_SEG_MAX]; };
static inline void fill_seg_add(FillSegs& f, void* p, size_t bytes) {          // filled with the launch's value
  if (bytes > 0 && f.n < FILL_SEG_MAX) { f.p[f.n] = p; f.bytes[f.n] = bytes; f.own[f.n] = 0; f.val[f.n] = 0; f.src[f.n] = nullptr; ++f.n; }
}
static inline void fill_seg_add(FillSegs& f, void* p, size_t bytes, unsigned value) {   // filled with its own value
  if (bytes > 0 && f.n < FILL_SEG_MAX) { f.p[f.n] = p; f.bytes[f.n] = bytes; f.own[f.n] = 1; f.val[f.n] = value; f.src[f.n] = nullptr; ++f.n; }
}
static inline void fill_seg_add_copy(FillSegs& f, void* p, const void* src, size_t bytes) {   // a copy riding in the fill launch (16-byte aligned both)
  if (bytes > 0 && f.n < FILL_SEG_MAX) { f.p[f.n] = p; f.bytes[f.n] = bytes; f.own[f.n] = 0; f.val[f.n] = 0; f.src[f.n] = src; ++f.n; }
}
int fill_u32_segments(const FillSegs& f, unsigned value, hipStream_t s);
// One launch for up to FILL_SEG_MAX independent device-to-device copies (16-byte aligned, sizes multiples of 4 bytes) and one for
// up to FILL_SEG_MAX independent transposes: the per-call state shuffling (final states, transposed weights) used to be ~35 tiny
// launches of ~5 us each per train step.
struct CopySegs { int n; void* dst[FILL_SEG_MAX]; const void* src[FILL_SEG_MAX]; size_t bytes[FILL_SEG_MAX]; };
static inline void copy_seg_add(CopySegs& c, void* dst, const void* src, size_t bytes) {
  if (bytes > 0 && c.n < FILL_SEG_MAX) { c.dst[c.n] = dst; c.src[c.n] = src; c.bytes[c.n] = bytes; ++c.n; }
}
int copy_segments(const CopySegs& c, hipStream_t s);
struct TransposeJobs { int n; float* dst[FILL_SEG_MAX]; const float* src[FILL_SEG_MAX]; long ldd[FILL_SEG_MAX], lds[FILL_SEG_MAX]; int rows[FILL_SEG_MAX], cols[FILL_SEG_MAX]; };
static inline void transpose_add(TransposeJobs& t, float* dst, long ldd, const float* src, long lds, int rows, int cols) {   // dst[c][r] = src[r][c]
  if (rows > 0 && cols > 0 && t.n < FILL_SEG_MAX) { t.dst[t.n] = dst; t.ldd[t.n] = ldd; t.src[t.n] = src; t.lds[t.n] = lds; t.rows[t.n] = rows; t.cols[t.n] = cols; ++t.n; }
}
int transpose_batch(const TransposeJobs& t, hipStream_t s);
int copy_f32(float* dst, const float* src, size_t n, hipStream_t s);
int copy2d_f32(float* dst, long ldd, const float* src, long lds, int rows, int cols, int cols_dst, hipStream_t s);
int add2d_f32(float* dst, long ldd, const float* src, long lds, int rows, int cols, hipStream_t s);
int transpose_f32(float* dst, long ldd, const float* src, long lds, int rows, int cols, hipStream_t s);  // dst[c][r] = src[r][c]
// Work enqueued on `to` after this call runs after everything enqueued on `from` before it (event record + stream wait; the
// events come from a small per-thread pool and carry no timing).
int stream_order(hipStream_t from, hipStream_t to);
int colsum_add_f32(float* dst, const float* src, long lds, int rows, int cols, hipStream_t s);           // dst[c] += sum_r src[r][c]
// Several column sums (the bias gradients of one backward phase) in ONE launch: add() collects, flush() launches.
constexpr int COLSUM_BATCH_MAX = 8;
struct ColsumJobs {
  int n;
  float* dst[COLSUM_BATCH_MAX];
  const float* src[COLSUM_BATCH_MAX];
  long lds[COLSUM_BATCH_MAX];
  int rows[COLSUM_BATCH_MAX], cols[COLSUM_BATCH_MAX];
  int gx[COLSUM_BATCH_MAX], gy[COLSUM_BATCH_MAX];   // the job's own colreduce_grid()
};
struct ColsumBatch {
  ColsumJobs j;
  ColsumBatch() { j.n = 0; }
  int add(float* dst, const float* src, long lds, int rows, int cols, hipStream_t s);
  int flush(hipStream_t s);
};
int axpy_rows(float* dst, const float* src, size_t n, hipStream_t s);                                    // dst += src

// ---- row-panel ("skinny", M <= a few dozen) MFMA products used by every recurrent step (rowgemm.hip)
struct RowPair {
  const float* A;  // [M][K], row stride lda
  long lda;
  const float* W;  // [N][K] K-contiguous (Linear layout), row stride ldw
  long ldw;
  int K;
};
enum RowAct { ACT_NONE = 0, ACT_TANH = 1, ACT_DTANH = 2 };
struct RowGemmArgs {
  RowPair p[2];
  int npairs;
  int M, N;
  const float* bias;    // [N] or null
  const float* addend;  // [M][N] (ld_add) or null
  long ld_add;
  const float* aux;     // ACT_DTANH: y = tanh output, out = (acc+addend)*(1-y*y)
  long ld_aux;
  float* out;
  long ld_out;
  float* out2;          // optional second copy of the result (e.g. into the input-feeding concat buffer)
  long ld_out2;
  int act;
  // optional in-place "carry" (decoder backward with input feeding): for columns n >= carry_col0, with j = n - carry_col0,
  //   carry[row][j] = (result + carry[row][j]) * (1 - carry_aux[row][j]^2)
  // i.e. the d_ht this product hands to the previous decoder step, added to that step's batched dlogits Wo and sent through tanh'
  float* carry;
  long ld_carry;
  const float* carry_aux;
  long ld_carry_aux;
  int carry_col0;
};
int rowgemm_launch(const RowGemmArgs& a, hipStream_t s);

struct LstmCellFwdArgs {
  RowPair p[2];          // p[0]: (h_prev, Wl) lateral (K may be 0 at the first encoder step); p[1]: (x, Wu) optional
  int npairs;
  int B, h;
  const float* zx;       // [B][4h] precomputed upward projection incl. bias (row stride ld_zx) or null
  long ld_zx;
  const float* bias;     // [4h] or null (used when zx is null or in addition)
  const float* c_prev;   // [B][h] or null (zeros)
  float* gates;          // [B][4h] activated a,i,f,o interleaved (may alias zx)
  long ld_g;
  float* c_out;          // [B][h]
  float* h_out;          // [B][h] raw
  const float* mask;     // [B][h] scaled keep-mask or null
  float* hd_out;         // [B][h'] dropped output, row stride ld_hd (null: skip)
  long ld_hd;
  float* hd_out2;        // optional second destination (enc_states slice), row stride ld_hd2
  long ld_hd2;
};
int lstm_cell_fwd_launch(const LstmCellFwdArgs* cells, int ncells, hipStream_t s);

struct LstmCellBwdArgs {
  RowPair p[2];          // p[0]: (dz_next [B][4h], WlT [h][4h]) -> dh_rec ; p[1]: (dz_above, WuT_above) -> dx (masked)
  int npairs;
  int B, h;
  const float* dh_add;   // [B][h] un-masked addend (d_hT at the last step) or null
  const float* dy;       // [B][h'] gradient wrt the dropped output (row stride ld_dy) or null
  long ld_dy;
  const float* dy2;      // second gradient wrt the dropped output (e.g. d_enc slice) or null
  long ld_dy2;
  const float* mask;     // [B][h] or null
  const float* dc_next;  // [B][h] or null
  const float* c_prev;   // [B][h] or null (zeros)
  const float* c_cur;    // [B][h]
  float* gates_dz;       // [B][4h]: in: activated gates, out: dz  (row stride ld_g)
  long ld_g;
  float* dc_prev;        // [B][h]
};
int lstm_cell_bwd_launch(const LstmCellBwdArgs* cells, int ncells, hipStream_t s);

// ---- attention (attn.hip)
int attn_fwd_launch(int B, int T, int H, const float* enc, const float* q, long ldq, float* alpha, float* cv, long ldcv,
                    float* cv2, long ldcv2, void* ws, hipStream_t s);
int attn_bwd_launch(int B, int T, int H, const float* enc, const float* alpha, const float* cv, long ldcv,
                    const float* d_cv, long ld_dcv, float* ds, float* dq, void* ws, hipStream_t s);
size_t attn_ws_bytes(int B, int T, int H);
int attn_ws_init(void* ws, int B, int T, int H, hipStream_t s);   // zero the ticket counters (once per workspace use)

// ---- normalisation / small row kernels of the optional model features (norm.hip)
int layernorm_fwd_launch(int rows, int n, const float* x, long ldx, const float* gamma, const float* beta, float eps, float* y, long ldy,
                         hipStream_t s);
int layernorm_bwd_launch(int rows, int n, const float* x, long ldx, const float* gamma, float eps, const float* dy, long lddy, float* dx,
                         long lddx, float* dgamma, float* dbeta, hipStream_t s);     // dgamma / dbeta accumulated (may be null)
int mul_rows_launch(float* x, long ldx, const float* m, long ldm, int rows, int cols, hipStream_t s);   // x[r][c] *= m[r][c]
constexpr float LN_EPS = 1e-6f;        // L.LayerNormalization's default eps

// ---- decoder helpers (decoder.hip)
int softmax_ce_launch(int B, int V, long ld, float* logits, const int32_t* targets, long t_stride, const float* cw,
                      float inv_count, float* loss_rows, int32_t* argmax, hipStream_t s);


// ---- column-reduction skeleton shared by the bias-gradient and BatchNorm statistics kernels.
// Row-major [rows x cols] input(s), leading dimension a multiple of 4 floats.  Grid: x = chunks of 64 columns,
// y = row slabs.  A block is CL = min(cols/4, 16) column lanes (one float4 each: 256 B of a row) x NR = 256/CL row
// lanes; every thread keeps 4 row loads in flight, row lanes are folded through LDS and row lane 0 emits one value per
// column and statistic.  (Narrow, tall blocks on purpose: the emits are same-address atomics, and 400 row slabs x 1024
// columns of them cost more than the whole read -- measured 30 us against 8 us for a 26 MB matrix.)
//   acc(r, c, nvalid, a[NS])  accumulates row r, columns c..c+3 into a[]  (loads must be unconditional: hipcc turns a
//                             guarded load into a branch + vmcnt(0), which would serialise the stream)
//   emit(col, stat, value)    publishes (normally an atomicAdd)
constexpr int COLREDUCE_CL = 16;
// (bx, by, ny): this block's column slab, row slab and the number of row slabs -- blockIdx.x, blockIdx.y, gridDim.y of a launch
// made with colreduce_grid(), or a job's own grid inside a batched launch (k_colsum_batch)
template <int NS, class Acc, class Emit>
__device__ __forceinline__ void colreduce_block(int rows, int cols, Acc acc, Emit emit, int bx, int by, int ny) {
  __shared__ float4 red[NS][256];
  const int q = (cols + 3) >> 2;
  const int CL = q < COLREDUCE_CL ? q : COLREDUCE_CL;
  const int NR = 256 / CL;
  const int cl = threadIdx.x % CL, rl = threadIdx.x / CL;
  const int c4 = bx * CL + cl;
  const bool act = rl < NR && c4 < q;
  const int c = c4 * 4, nvalid = cols - c;
  float4 a[NS];
#pragma unroll
  for (int i = 0; i < NS; ++i) a[i] = make_float4(0.f, 0.f, 0.f, 0.f);
  if (act) {
    // row chunks of 4*NR rows are dealt round-robin to the row slabs (blockIdx.y): at any moment the grid reads one contiguous
    // window of the matrix instead of gridDim.y streams a fixed distance apart
    const int rpc = 4 * NR;
    for (int r = by * rpc + rl; r < rows; r += ny * rpc) {
      if (r + 3 * NR < rows) {
        acc(r, c, a);
        acc(r + NR, c, a);
        acc(r + 2 * NR, c, a);
        acc(r + 3 * NR, c, a);
      } else {
        for (int rr = r; rr < rows && rr < r + rpc; rr += NR) acc(rr, c, a);
      }
    }
  }
  if (NR > 1) {
#pragma unroll
    for (int i = 0; i < NS; ++i) red[i][threadIdx.x] = a[i];
    __syncthreads();
    if (act && rl == 0) {
      for (int k = 1; k < NR; ++k) {
#pragma unroll
        for (int i = 0; i < NS; ++i) {
          const float4 v = red[i][k * CL + cl];
          a[i].x += v.x; a[i].y += v.y; a[i].z += v.z; a[i].w += v.w;
        }
      }
    }
  }
  if (act && rl == 0) {
#pragma unroll
    for (int i = 0; i < NS; ++i) {
      if (nvalid > 0) emit(c + 0, i, a[i].x);
      if (nvalid > 1) emit(c + 1, i, a[i].y);
      if (nvalid > 2) emit(c + 2, i, a[i].z);
      if (nvalid > 3) emit(c + 3, i, a[i].w);
    }
  }
}
template <int NS, class Acc, class Emit>
__device__ __forceinline__ void colreduce_block(int rows, int cols, Acc acc, Emit emit) {
  colreduce_block<NS>(rows, cols, acc, emit, (int)blockIdx.x, (int)blockIdx.y, (int)gridDim.y);
}
// grid for colreduce_block: one block per CU, at least 16 rows per lane.  (Every block ends in same-address double atomics, one per column and
// statistic: with 512 blocks the layer-0 BatchNorm statistics -- 128 columns, i.e. 256 arrivals per address -- took 29 us for a 39 MB read, with
// 256 blocks 19; 128 blocks lose on the wide layers.  astk_set_tuning("colreduce.blocks") overrides.)
// float_sums: the blocks publish FLOAT atomics (bias gradients): under a deterministic call one block per column group sums its rows in
// order -- one add per column.  The BatchNorm statistics' DOUBLE atomics keep their grids: their order changes a float result only when a
// double sum falls within 2^-29 of a float rounding boundary (the forward pass has used them through every bit-identical soak).
static inline dim3 colreduce_grid(int rows, int cols, bool float_sums = false) {
  const int q = (cols + 3) / 4;
  const int CL = q < COLREDUCE_CL ? q : COLREDUCE_CL, NR = 256 / CL;
  const int gx = (q + CL - 1) / CL;
  const int blocks = (int)tune(TUNE_COLREDUCE_BLOCKS);
  int gy = (float_sums && deterministic_mode()) ? 1 : blocks / gx;
  const int max_gy = (rows + 16 * NR - 1) / (16 * NR);
  if (gy > max_gy) gy = max_gy;
  if (gy < 1) gy = 1;
  return dim3((unsigned)gx, (unsigned)gy);
}

}  // namespace astk
