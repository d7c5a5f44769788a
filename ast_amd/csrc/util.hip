// Small memory-bound helpers: copies with re-pitch, transposes (weight re-layouts for the backward
// products), column sums (bias gradients), counter-based RNG fills, and the fused optimizer
// (nn.py:81-119: WeightDecay -> GradientClipping -> Adam(amsgrad) on one flat buffer).
#include "common.h"
#include <algorithm>
#include <stdarg.h>
#include <vector>
#include <atomic>
#include <mutex>

namespace astk {

static thread_local char g_err[512] = "";

void set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}
const char* last_error() { return g_err; }

// ---- profiler
struct ProfState {
  bool on = false;
  std::vector<hipEvent_t> ev[PROF_NCAT];   // start/stop pairs
  double work[PROF_NCAT] = {0, 0, 0, 0, 0, 0};
  double bytes[PROF_NCAT] = {0, 0, 0, 0, 0, 0};     // algorithmic bytes (operands read once + results written once)
};
static ProfState g_prof;
static float* g_tick = nullptr;     // [2][512] in-kernel phase timers of the persistent decoder kernels
float* prof_tick_buffer(int which) { return (g_prof.on && g_tick) ? g_tick + 512 * which : nullptr; }
bool prof_enabled() { return g_prof.on; }
void prof_add_bytes(int cat, double bytes) { if (g_prof.on) g_prof.bytes[cat] += bytes; }
void prof_start(int cat, hipStream_t s, double work) {
  hipEvent_t e;
  if (hipEventCreate(&e) != hipSuccess) return;
  (void)hipEventRecord(e, s);
  g_prof.ev[cat].push_back(e);
  g_prof.work[cat] += work;
}
void prof_stop(int cat, hipStream_t s) {
  if (g_prof.ev[cat].size() % 2 == 0) return;
  hipEvent_t e;
  if (hipEventCreate(&e) != hipSuccess) { g_prof.ev[cat].pop_back(); return; }
  (void)hipEventRecord(e, s);
  g_prof.ev[cat].push_back(e);
}

// ---- sticky status word of the persistent kernels (common.h: AbortCtl)
__device__ unsigned g_persist_status[16];     // word 0: PersistBit mask of the kernels whose bounded spins have timed out
static unsigned* g_status_ptr[16] = {nullptr};
static int g_cu_count[16] = {0};
static unsigned* persist_status_ptr() {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 16) dev = 0;
  if (!g_status_ptr[dev]) {
    void* p = nullptr;
    if (hipGetSymbolAddress(&p, HIP_SYMBOL(g_persist_status)) != hipSuccess) return nullptr;
    g_status_ptr[dev] = (unsigned*)p;
  }
  return g_status_ptr[dev];
}
AbortCtl abort_ctl(unsigned* word, unsigned bit) {
  AbortCtl ab;
  ab.word = word;
  ab.status = persist_status_ptr();
  ab.limit = 1u << 22;
  const double lim = tune(TUNE_PERSIST_SPIN_LIMIT);       // test knob ("persist.spin_limit"): a tiny bound forces the time-out path
  if (lim >= 1.0) ab.limit = (unsigned)lim;
  ab.bit = bit;
  return ab;
}
// ---- tuning knobs (astk_set_tuning): name, default
struct TuneEntry { TuneKey key; const char* name; double def; };
static constexpr TuneEntry g_tune_table[TUNE_COUNT] = {
    {TUNE_GEMM_TILE, "gemm.tile", 0},              // force the block tile: 64 | 128 | 256 (0 = chosen per launch)
    {TUNE_GEMM_T256_ABOVE, "gemm.t256_above", 2e10},     // launches of at least this many flops take the 12-wave 256 x 128 kernel
    {TUNE_GEMM_GRID, "gemm.grid", -1},             // force the stream-K grid (workgroups; <= 0 = chosen per launch)
    {TUNE_GEMM_HYBRID, "gemm.hybrid", 1},            // data-parallel waves of whole tiles in XCD-local blocks + stream-K remainder (0: all stream-K, rounds 1-4)
    {TUNE_GEMM_CHUNK, "gemm.chunk", 1},             // few tiles / deep K accumulating launches run chunk-major
    {TUNE_GEMM_CHUNK_DIV, "gemm.chunk_div", 4},         // ... when tiles * chunk_div <= grid
    {TUNE_GEMM_LOG, "gemm.log", 0},               // print every launch's shape and schedule to stderr
    {TUNE_GEMM_TICKET, "gemm.ticket", 1},            // libastk_test.so only: split tiles handed over by tickets instead of a zeroing launch
    {TUNE_GEMM_DETERMINISTIC, "gemm.deterministic", 0},     // process default of the descriptors' `deterministic` field (fixed-order split-tile sums, astk.h)
    {TUNE_GEMM_FORWARD_PAIRS, "gemm.forward_pairs", 1},     // forward launches outside the hybrid branch: grid shrunk until a split tile has <= 2 contributors (2: the forward rule for EVERY launch)
    {TUNE_CONV_DIRECT0, "conv.direct0", 1},           // layer 0 as a direct convolution (0: im2col + GEMM)
    {TUNE_CONV_SEQ_FWD, "conv.seq_fwd", 1},           // BatchNorm + ReLU written straight into the LSTM's (T'',B,C*F') layout by the tiled kernel
    {TUNE_CONV_SEQ_BWD, "conv.seq_bwd", 1},           // the last layer's BatchNorm backward reads that layout itself
    {TUNE_CONV_SEQ_STATS_BLOCKS, "conv.seq_stats_blocks", 1024},
    {TUNE_CONV_SEQ_APPLY_BLOCKS, "conv.seq_apply_blocks", 1024},
    {TUNE_DEC_PERSIST, "dec.persist", 1},            // persistent decoder loops (0: the per-launch loop)
    {TUNE_DEC_B6_SPLIT, "dec.b6_split", 1},           // K-half items of the d_x0 role (one-layer kernel)
    {TUNE_DEC_B6_FUSED, "dec.b6_fused", 1},           // ... with the d_pre role fused into them
    {TUNE_DEC_WIDE, "dec.wide", 1},               // decoder_wide.hip's persistent loops at H = A = 1024
    {TUNE_LSTM_PERSIST, "lstm.persist", 1},           // persistent encoder recurrences (0: one fused-cell launch per step)
    {TUNE_LSTM_HOIST, "lstm.hoist", 1},             // hoisted form of the persistent kernels at h = 1024
    {TUNE_LSTM_X3, "lstm.x3", 1},                // bf16x3 fragments inside the recurrences (0: f32-input MFMAs under the bf16x3 arithmetic)
    {TUNE_LSTM_X4, "lstm.x4", 1},                // ... with the weights' lo plane in LDS at h = 512 / 1024
    {TUNE_LSTM_ROWS32, "lstm.rows32", -1},           // 32 batch rows per recurrence workgroup: 1 always, 0 never, -1 = when it spares launches
    {TUNE_LSTM_OVERLAP_CHUNK, "lstm.overlap_chunk", 0},     // time steps per chunk of the layer-0 products that run beside the recurrences (side_stream); 0 = from the free CUs
    {TUNE_LSTM_SIDE_FWD, "lstm.side_fwd", 1},          // side_stream: the layer-0 input projection in chunks beside the forward recurrence
    {TUNE_LSTM_SIDE_BWD, "lstm.side_bwd", 0},          // side_stream: this many chunks of the input gradient behind the backward recurrence's progress counter, the
                                   // rest in line (< 0: every chunk).  Off: measured slower at any count, profiles/r6_ab_side_bwd.txt
    {TUNE_LSTM_DUO_SIDE, "lstm.duo_side", 0},          // side_stream + batch <= 32: take the two-waves-per-SIMD form (96 workgroups, 160 CUs free) instead of 192 workgroups of 16 rows
    {TUNE_ROW_LONGK, "row.longk", 2048},           // row-panel kernels split K over eight waves from this K on (0 = never)
    {TUNE_PERSIST_SPIN_LIMIT, "persist.spin_limit", 0},     // bound of the persistent kernels' spins in polls (0 = the default, 2^22)
    {TUNE_COLREDUCE_BLOCKS, "colreduce.blocks", 256},     // blocks of a column reduction
};
// (the table is indexed by TuneKey: a row out of place would make one knob's NAME set another knob's value -- round 6 shipped
//  "lstm.side_bwd" and "lstm.duo_side" swapped for a while, found when an A/B of the first showed the second's effect)
constexpr bool tune_table_in_order() {
  for (int i = 0; i < TUNE_COUNT; ++i)
    if (g_tune_table[i].key != i) return false;
  return true;
}
static_assert(tune_table_in_order(), "g_tune_table rows must follow enum TuneKey");
static std::atomic<double> g_tune[TUNE_COUNT];
static std::atomic<int> g_tune_init{0};
static void tune_init() {
  if (g_tune_init.load(std::memory_order_acquire)) return;
  static std::mutex mu;
  std::lock_guard<std::mutex> lock(mu);
  if (g_tune_init.load()) return;
  for (int i = 0; i < TUNE_COUNT; ++i) g_tune[i].store(g_tune_table[i].def);
  g_tune_init.store(1, std::memory_order_release);
}
double tune(TuneKey k) { tune_init(); return g_tune[k].load(std::memory_order_relaxed); }
static int tune_find(const char* key) {
  if (!key) return -1;
  for (int i = 0; i < TUNE_COUNT; ++i)
    if (!strcmp(key, g_tune_table[i].name)) return i;
  return -1;
}

int device_cu_count() {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 16) return 0;
  if (g_cu_count[dev] == 0) {
    int n = 0;
    if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) return 0;
    g_cu_count[dev] = n;
  }
  return g_cu_count[dev];
}

namespace {

__global__ void k_status_snapshot(const unsigned* status, float* dst) { *dst = (float)(*status); }
__global__ void k_status_reset(unsigned* status) { *status = 0u; }
__global__ void k_status_merge(unsigned* status, const float* summed) { if (*summed != 0.f) atomicOr(status, 16u); }

__global__ void k_copy2d(float* dst, long ldd, const float* src, long lds, int rows, int cols, int cols_dst) {
  const long n = (long)rows * cols_dst;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    const int r = (int)(i / cols_dst), c = (int)(i % cols_dst);
    dst[r * ldd + c] = c < cols ? src[r * lds + c] : 0.f;
  }
}

// init_decoder_state and its backward: both tensors (c, h), all bridged layers, one launch (float4 over units)
__global__ __launch_bounds__(256) void k_bridge_states(float* __restrict__ dec_c, float* __restrict__ dec_h, float* __restrict__ enc_c,
                                                       float* __restrict__ enc_h, int nd, int nl_enc, int n, int B, int h, int to_dec) {
  const int h4 = h / 4;
  const long per = (long)n * B * nd * h4;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < 2 * per; i += (long)gridDim.x * blockDim.x) {
    const bool second = i >= per;
    long r = second ? i - per : i;
    const int u4 = (int)(r % h4); r /= h4;
    const int d = (int)(r % nd); r /= nd;
    const int b = (int)(r % B);
    const int k = (int)(r / B);
    float4* dq = reinterpret_cast<float4*>((second ? dec_h : dec_c) + (((long)k * B + b) * nd + d) * h) + u4;
    float4* eq = reinterpret_cast<float4*>((second ? enc_h : enc_c) + (((long)d * nl_enc + k) * B + b) * h) + u4;
    if (to_dec) *dq = *eq;
    else *eq = *dq;
  }
}
__global__ void k_add2d(float* dst, long ldd, const float* src, long lds, int rows, int cols) {
  const long n = (long)rows * cols;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    const int r = (int)(i / cols), c = (int)(i % cols);
    dst[r * ldd + c] += src[r * lds + c];
  }
}

__global__ void k_axpy(float* dst, const float* src, size_t n) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) dst[i] += src[i];
}

// dst[c][r] = src[r][c] through a 32x33 LDS tile; dst columns in [rows, ldd) are zero-filled.
__global__ void k_transpose(float* dst, long ldd, const float* src, long lds, int rows, int cols) {
  __shared__ float t[32][33];
  const int bx = blockIdx.x * 32, by = blockIdx.y * 32;  // bx: src col block, by: src row block
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5; // 256 threads: ty 0..7
  for (int j = ty; j < 32; j += 8) {
    const int r = by + j, c = bx + tx;
    t[j][tx] = (r < rows && c < cols) ? src[(long)r * lds + c] : 0.f;
  }
  __syncthreads();
  for (int j = ty; j < 32; j += 8) {
    const int c = bx + j, r = by + tx;   // dst row = src col
    if (c < cols && r < ldd) dst[(long)c * ldd + r] = t[tx][j];
  }
}

__global__ void k_transpose_batch(TransposeJobs tj) {
  const int job = blockIdx.z;
  const int rows = tj.rows[job], cols = tj.cols[job];
  const long ldd = tj.ldd[job], lds = tj.lds[job];
  const int bx = blockIdx.x * 32, by = blockIdx.y * 32;
  if (bx >= cols || by >= ldd) return;
  __shared__ float t[32][33];
  const float* src = tj.src[job];
  float* dst = tj.dst[job];
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  for (int j = ty; j < 32; j += 8) {
    const int r = by + j, c = bx + tx;
    t[j][tx] = (r < rows && c < cols) ? src[(long)r * lds + c] : 0.f;
  }
  __syncthreads();
  for (int j = ty; j < 32; j += 8) {
    const int c = bx + j, r = by + tx;
    if (c < cols && r < ldd) dst[(long)c * ldd + r] = t[tx][j];
  }
}

__global__ __launch_bounds__(256) void k_copy_segments(CopySegs cs) {
  const int seg = blockIdx.y;
  const size_t n16 = cs.bytes[seg] / 16;
  const uint4* src = reinterpret_cast<const uint4*>(cs.src[seg]);
  uint4* dst = reinterpret_cast<uint4*>(cs.dst[seg]);
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n16; i += (size_t)gridDim.x * blockDim.x) dst[i] = src[i];
  if (blockIdx.x == 0) {
    const unsigned* sw = reinterpret_cast<const unsigned*>(cs.src[seg]);
    unsigned* dw = reinterpret_cast<unsigned*>(cs.dst[seg]);
    for (size_t i = n16 * 4 + threadIdx.x; i < cs.bytes[seg] / 4; i += blockDim.x) dw[i] = sw[i];
  }
}

// dst[c] += sum_r src[r][c]   (colreduce_block skeleton, one atomic per column and row slab)
__global__ __launch_bounds__(256) void k_colsum(float* dst, const float* __restrict__ src, long lds, int rows, int cols) {
  colreduce_block<1>(
      rows, cols,
      [&](int r, int c, float4* a) {
        // columns past `cols` inside the last quad are readable (lds is a multiple of 4) and never emitted
        const float4 v = *reinterpret_cast<const float4*>(src + (long)r * lds + c);
        a[0].x += v.x; a[0].y += v.y; a[0].z += v.z; a[0].w += v.w;
      },
      [&](int col, int, float v) { atomicAdd(&dst[col], v); });
}

// blockIdx.z = job; blocks outside the job's own grid leave at once
__global__ __launch_bounds__(256) void k_colsum_batch(ColsumJobs j) {
  const int q = blockIdx.z;
  if ((int)blockIdx.x >= j.gx[q] || (int)blockIdx.y >= j.gy[q]) return;
  const float* __restrict__ src = j.src[q];
  float* dst = j.dst[q];
  const long lds = j.lds[q];
  colreduce_block<1>(
      j.rows[q], j.cols[q],
      [&](int r, int c, float4* a) {
        const float4 v = *reinterpret_cast<const float4*>(src + (long)r * lds + c);
        a[0].x += v.x; a[0].y += v.y; a[0].z += v.z; a[0].w += v.w;
      },
      [&](int col, int, float v) { atomicAdd(&dst[col], v); }, (int)blockIdx.x, (int)blockIdx.y, j.gy[q]);
}

// one wave that stays busy for `ticks` of the 100 MHz wall clock (bounded) and then bumps *flag
__global__ void k_spin(unsigned long long ticks, unsigned* flag) {
  const unsigned long long t0 = wall_clock64();
  while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(8);
  if (flag && threadIdx.x == 0) atomicAdd(flag, 1u);
}

__global__ void k_scale(float* x, size_t n, float s) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) x[i] *= s;
}

// ---- counter-based RNG (splitmix64 finaliser over (seed, counter)); quality is ample for dropout / noise
__device__ __forceinline__ uint64_t mix64(uint64_t z) {
  z += 0x9E3779B97F4A7C15ull;
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  return z ^ (z >> 31);
}
__device__ __forceinline__ float u01(uint64_t bits) { return (float)((bits >> 40) + 1) * (1.0f / 16777217.0f); }  // (0,1)

// Dropout keep-masks: element i of a fill is a function of (seed, g = offset + i) alone -- the two halves of ONE 64-bit hash of the pair
// g >> 1 serve the counters 2 (g >> 1) and 2 (g >> 1) + 1 (round 5: the hash is 16 quarter-rate integer multiplies, and one hash per
// element made the 39 MB of encoder masks a 25 us compute-bound fill; fused and separate fills still agree element by element).
__device__ __forceinline__ float u01_32(unsigned bits) { return (float)((bits >> 8) + 1) * (1.0f / 16777217.0f); }  // (0,1)
__device__ __forceinline__ void dropout_fill(float* out, size_t n, float ratio, float scale, uint64_t seed, uint64_t offset, size_t first, size_t stride) {
  if (n == 0) return;
  const uint64_t p0 = offset >> 1, npairs = ((offset + n - 1) >> 1) - p0 + 1;
  for (uint64_t j = first; j < npairs; j += stride) {
    const uint64_t pr = p0 + j, h = mix64(seed ^ mix64(pr));
    const uint64_t g0 = 2 * pr, g1 = g0 + 1;
    if (g0 >= offset) out[g0 - offset] = u01_32((unsigned)h) >= ratio ? scale : 0.f;
    if (g1 < offset + n) out[g1 - offset] = u01_32((unsigned)(h >> 32)) >= ratio ? scale : 0.f;
  }
}
__global__ void k_dropout_mask(float* out, size_t n, float ratio, float scale, uint64_t seed, uint64_t offset) {
  dropout_fill(out, n, ratio, scale, seed, offset, blockIdx.x * (size_t)blockDim.x + threadIdx.x, (size_t)gridDim.x * blockDim.x);
}

// Box-Muller pair i of a normal fill: both uniforms from ONE hash (its halves), the hardware's log2 / sin / cos (the sine unit takes its
// argument in turns, u2 itself); absolute error ~1e-6 of a unit normal, far below what a noise tensor can tell apart (round 5: libdevice's
// logf / sincosf with their range reductions made the 2 M noise values the slowest segment of the step's fused fill).
__device__ __forceinline__ void normal_pair(float* out, size_t n, float mean, float sigma, uint64_t seed, uint64_t offset, size_t i) {
  const uint64_t b = mix64(seed ^ mix64(offset + i));
  const float u1 = u01_32((unsigned)b), u2 = u01_32((unsigned)(b >> 32));
  const float r = sqrtf(-1.3862943611198906f * __builtin_amdgcn_logf(u1));      // -2 ln u1 = -2 ln2 log2 u1
  out[2 * i] = mean + sigma * r * __builtin_amdgcn_cosf(u2);
  if (2 * i + 1 < n) out[2 * i + 1] = mean + sigma * r * __builtin_amdgcn_sinf(u2);
}
__global__ void k_normal(float* out, size_t n, float mean, float sigma, uint64_t seed, uint64_t offset) {
  const size_t pairs = (n + 1) / 2;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < pairs; i += (size_t)gridDim.x * blockDim.x) normal_pair(out, n, mean, sigma, seed, offset, i);
}

// Several independent fills of the two kinds above in ONE launch (the step's speech noise and its dropout masks: four launches of ~5-20 us
// before).  Blocks are dealt to the segments in proportion to their sizes; every element gets exactly the value the single-segment kernels
// give it (same counter), so fused and separate draws are interchangeable.
// (words: up to ASTK_RAND_WORDS_MAX host values that ride in the kernel arguments and are written to words_dst by block 0 -- the step's
//  teacher-forcing flags, astk_fill_random_ex: a pinned-buffer copy of their own cost a launch and a cross-queue gap)
struct RandJobs { int n; int blk_start[ASTK_RAND_SEG_MAX + 1]; astk_rand_seg s[ASTK_RAND_SEG_MAX]; int n_words; int32_t* words_dst; int32_t words[ASTK_RAND_WORDS_MAX]; };
__global__ __launch_bounds__(256) void k_fill_random(RandJobs j) {
  if (blockIdx.x == 0)
    for (int i = threadIdx.x; i < j.n_words; i += 256) j.words_dst[i] = j.words[i];
  if (j.n == 0) return;
  int q = 0;
  while ((int)blockIdx.x >= j.blk_start[q + 1]) ++q;
  const astk_rand_seg sg = j.s[q];
  const size_t nblk = (size_t)(j.blk_start[q + 1] - j.blk_start[q]), blk = blockIdx.x - j.blk_start[q];
  if (sg.kind == ASTK_RAND_DROPOUT) {
    dropout_fill(sg.out, sg.n, sg.a, 1.f / (1.f - sg.a), sg.seed, sg.offset, blk * 256 + threadIdx.x, nblk * 256);
  } else {
    const size_t pairs = (sg.n + 1) / 2;
    for (size_t i = blk * 256 + threadIdx.x; i < pairs; i += nblk * 256) normal_pair(sg.out, sg.n, sg.a, sg.b, sg.seed, sg.offset, i);
  }
}

// dataloader.py:83-93 on the device: utterance b (true length len[b] <= T frames, zero-padded behind) gets int(rate * len[b]) of its
// frames zeroed; the frames are drawn WITH replacement (np.random.choice's default), so fewer distinct frames may be hit.
// (the count is int(drop_rate * len(x_data)) in double precision, exactly as Python evaluates it)
__device__ __forceinline__ int zero_frames_draw(uint64_t seed, uint64_t offset, int b, int T, int i, int nb) {
  const uint64_t bits = mix64(seed ^ mix64(offset + (uint64_t)b * (uint64_t)T + (uint64_t)i));
  return (int)((bits >> 11) % (uint64_t)nb);
}
__global__ void k_zero_frames(float* X, int T, int D, const int32_t* len, double rate, uint64_t seed, uint64_t offset) {
  const int b = blockIdx.x;
  const int nb = min(max(len[b], 0), T);
  const int n = (int)(rate * (double)nb);
  float* xb = X + (size_t)b * T * D;
  for (int i = threadIdx.x / 32; i < n; i += blockDim.x / 32) {        // 32 lanes clear one drawn frame
    const int r = zero_frames_draw(seed, offset, b, T, i, nb);
    for (int d = threadIdx.x % 32; d < D; d += 32) xb[(size_t)r * D + d] = 0.f;
  }
}
// the same draws, written out instead of applied: idx[b][i] = i-th frame drawn for utterance b (i < counts[b] <= max_draws)
__global__ void k_zero_frames_draws(int T, const int32_t* len, double rate, uint64_t seed, uint64_t offset, int32_t* idx, int max_draws,
                                    int32_t* counts) {
  const int b = blockIdx.x;
  const int nb = min(max(len[b], 0), T);
  const int n = (int)(rate * (double)nb);
  if (threadIdx.x == 0) counts[b] = n;
  for (int i = threadIdx.x; i < n && i < max_draws; i += blockDim.x) idx[(size_t)b * max_draws + i] = zero_frames_draw(seed, offset, b, T, i, nb);
}

// ---- optimizer
constexpr unsigned SQNORM_BLOCKS = 512;
__device__ double g_sq_part[SQNORM_BLOCKS];
__device__ unsigned g_sq_ctr;
__global__ void k_sqnorm(const float* g, const float* p, float gsc, float l2, size_t n, double* out) {
  __shared__ double red[4];
  double s = 0.0;
  const size_t n4 = n / 4;
  const float4* g4 = reinterpret_cast<const float4*>(g);
  const float4* p4 = reinterpret_cast<const float4*>(p);
  // (one double atomic per block at the end: same-address atomics retire one after the other, so few blocks with several loads in
  //  flight each -- 2048 blocks spent more time in that queue than reading the two arrays)
  auto term = [&](const float4& a, const float4& b) {
    // __fmul_rn: the scaled gradient is rounded before the decay term is added, exactly as if the buffer had been scaled first
    const float x = __fmul_rn(a.x, gsc) + l2 * b.x, y = __fmul_rn(a.y, gsc) + l2 * b.y, z = __fmul_rn(a.z, gsc) + l2 * b.z, w = __fmul_rn(a.w, gsc) + l2 * b.w;
    return (double)(x * x + y * y) + (double)(z * z + w * w);
  };
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
  for (; i + 3 * stride < n4; i += 4 * stride) {
    const float4 a0 = g4[i], a1 = g4[i + stride], a2 = g4[i + 2 * stride], a3 = g4[i + 3 * stride];
    const float4 b0 = p4[i], b1 = p4[i + stride], b2 = p4[i + 2 * stride], b3 = p4[i + 3 * stride];
    s += (term(a0, b0) + term(a1, b1)) + (term(a2, b2) + term(a3, b3));
  }
  for (; i < n4; i += stride) s += term(g4[i], p4[i]);
  if (blockIdx.x == 0 && threadIdx.x == 0)
    for (size_t i = n4 * 4; i < n; ++i) { float x = __fmul_rn(g[i], gsc) + l2 * p[i]; s += (double)x * x; }
  for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  // Per-block partial sums, folded in block order by the block that arrives last: no zeroing launch in front (the counter goes back to
  // zero here), and the norm is the same sum whatever order the blocks finish in.
  __shared__ int last;
  if (threadIdx.x == 0) {
    __hip_atomic_store(&g_sq_part[blockIdx.x], red[0] + red[1] + red[2] + red[3], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // written through and acknowledged (no fence: its L2 write-back / invalidate costs 10 us here)
    last = __hip_atomic_fetch_add(&g_sq_ctr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == gridDim.x - 1;
  }
  __syncthreads();
  if (!last) return;
  double t = 0.0;
  for (unsigned b = threadIdx.x; b < gridDim.x; b += 256) t += __hip_atomic_load(&g_sq_part[b], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  for (int o = 32; o > 0; o >>= 1) t += __shfl_xor(t, o);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = t;
  __syncthreads();
  if (threadIdx.x == 0) {
    *out = (red[0] + red[1]) + (red[2] + red[3]);
    __hip_atomic_store(&g_sq_ctr, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
}

__global__ void k_amsgrad(float* p, const float* g, float* m, float* v, float* vhat, size_t n, float gsc, float l2, float clip,
                          const double* sqnorm, float lr_t, float b1, float b2, float eps, int amsgrad, const unsigned* status) {
  // a persistent kernel of this step timed out (sticky status word): its loss and gradients are garbage, the host finds out when it
  // reads the loss back one step later -- the parameters and moments must not have moved by then
  if (status && *status != 0u) return;
  const float norm = (float)sqrt(*sqnorm);
  const float rate = clip / norm;                    // A7: r = c / n, applied only when r < 1
  const float gs = rate < 1.f ? rate : 1.f;
  auto one = [&](float& pi, float gi_raw, float& mi, float& vi, float& vh) {      // vh: in = vhat (amsgrad), out = the denominator's v
    const float gi = (__fmul_rn(gi_raw, gsc) + l2 * pi) * gs;
    mi += (1.f - b1) * (gi - mi);
    vi += (1.f - b2) * (gi * gi - vi);
    vh = amsgrad ? fmaxf(vh, vi) : vi;
    pi = pi - lr_t * mi / (sqrtf(vh) + eps);
  };
  const size_t stride = (size_t)gridDim.x * blockDim.x, tid = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
  size_t done = 0;
  // nine streams of 4 bytes per lane ran at 4.0 TB/s; 16 bytes per lane where the five arrays allow it (same arithmetic per element)
  if (((((uintptr_t)p) | ((uintptr_t)g) | ((uintptr_t)m) | ((uintptr_t)v) | ((uintptr_t)(amsgrad ? vhat : p))) & 15) == 0) {
    const size_t n4 = n / 4;
    float4* p4 = reinterpret_cast<float4*>(p);
    const float4* g4 = reinterpret_cast<const float4*>(g);
    float4 *m4 = reinterpret_cast<float4*>(m), *v4 = reinterpret_cast<float4*>(v), *h4 = reinterpret_cast<float4*>(vhat);
    for (size_t i = tid; i < n4; i += stride) {
      float4 P = p4[i], M = m4[i], V = v4[i], H = amsgrad ? h4[i] : make_float4(0.f, 0.f, 0.f, 0.f);
      const float4 G = g4[i];
      one(P.x, G.x, M.x, V.x, H.x); one(P.y, G.y, M.y, V.y, H.y); one(P.z, G.z, M.z, V.z, H.z); one(P.w, G.w, M.w, V.w, H.w);
      m4[i] = M; v4[i] = V;
      if (amsgrad) h4[i] = H;
      p4[i] = P;
    }
    done = n4 * 4;
  }
  for (size_t i = done + tid; i < n; i += stride) {
    float pi = p[i], mi = m[i], vi = v[i], vh = amsgrad ? vhat[i] : 0.f;
    one(pi, g[i], mi, vi, vh);
    m[i] = mi;
    v[i] = vi;
    if (amsgrad) vhat[i] = vh;
    p[i] = pi;
  }
}

// GradientNoise behind WeightDecay and GradientClipping (hook order of nn.py:98-110): g <- clip(g gsc + l2 p) + sigma N(0,1), in place;
// the update kernel then runs on the finished gradient (gsc = 1, l2 = 0, no clip)
__global__ void k_decay_clip_noise(float* g, const float* p, size_t n, float gsc, float l2, float clip, const double* sqnorm, float sigma,
                                   uint64_t seed, uint64_t offset) {
  const float norm = (float)sqrt(*sqnorm);
  const float rate = clip / norm;
  const float gs = rate < 1.f ? rate : 1.f;
  const size_t pairs = (n + 1) / 2;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < pairs; i += (size_t)gridDim.x * blockDim.x) {
    const uint64_t b = mix64(seed ^ mix64(offset + i));
    const float u1 = u01(b), u2 = u01(mix64(b));
    const float r = sqrtf(-2.f * logf(u1));
    float sn, cs;
    sincosf(6.2831853071795864f * u2, &sn, &cs);
    g[2 * i] = (__fmul_rn(g[2 * i], gsc) + l2 * p[2 * i]) * gs + sigma * r * cs;
    if (2 * i + 1 < n) g[2 * i + 1] = (__fmul_rn(g[2 * i + 1], gsc) + l2 * p[2 * i + 1]) * gs + sigma * r * sn;
  }
}

__global__ void k_sgd(float* p, const float* g, size_t n, float gsc, float l2, float clip, const double* sqnorm, float lr, const unsigned* status) {
  if (status && *status != 0u) return;      // (see k_amsgrad)
  const float norm = (float)sqrt(*sqnorm);
  const float rate = clip / norm;
  const float gs = rate < 1.f ? rate : 1.f;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
    p[i] -= lr * (__fmul_rn(g[i], gsc) + l2 * p[i]) * gs;
}

inline unsigned grid_for(size_t n, int per = 256) {
  size_t b = (n + per - 1) / per;
  if (b < 1) b = 1;
  if (b > 2048) b = 2048;
  return (unsigned)b;
}

__global__ __launch_bounds__(256) void k_fill_segments(FillSegs f, unsigned value) {
  const int seg = blockIdx.y;
  if (seg >= f.n) return;
  if (f.own[seg]) value = f.val[seg];
  const size_t n16 = f.bytes[seg] / 16;
  uint4* p = reinterpret_cast<uint4*>(f.p[seg]);
  if (f.src[seg]) {      // a copy riding in this launch (fill_seg_add_copy)
    const uint4* q = reinterpret_cast<const uint4*>(f.src[seg]);
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n16; i += (size_t)gridDim.x * blockDim.x) p[i] = q[i];
    if (blockIdx.x == 0) {
      unsigned* w = reinterpret_cast<unsigned*>(f.p[seg]);
      const unsigned* r = reinterpret_cast<const unsigned*>(f.src[seg]);
      for (size_t i = n16 * 4 + threadIdx.x; i < f.bytes[seg] / 4; i += blockDim.x) w[i] = r[i];
    }
    return;
  }
  const uint4 v = make_uint4(value, value, value, value);
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n16; i += (size_t)gridDim.x * blockDim.x) p[i] = v;
  if (blockIdx.x == 0) {   // tail words
    unsigned* w = reinterpret_cast<unsigned*>(f.p[seg]);
    for (size_t i = n16 * 4 + threadIdx.x; i < f.bytes[seg] / 4; i += blockDim.x) w[i] = value;
  }
}

}  // namespace

int fill_u32_segments(const FillSegs& f, unsigned value, hipStream_t s) {
  if (f.n <= 0) return 0;
  for (int i = 0; i < f.n; ++i)
    ASTK_CHECK(aligned16(f.p[i]) && aligned16(f.src[i]) && (f.bytes[i] % 4) == 0, "fill_u32_segments: segment %d must be 16-byte aligned, size a multiple of 4", i);
  size_t mx = 0;
  for (int i = 0; i < f.n; ++i) mx = f.bytes[i] > mx ? f.bytes[i] : mx;
  unsigned gx = (unsigned)((mx / 16 + 256 * 8 - 1) / (256 * 8));
  if (gx < 1) gx = 1;
  if (gx > 1024) gx = 1024;
  hipLaunchKernelGGL(k_fill_segments, dim3(gx, (unsigned)f.n), dim3(256), 0, s, f, value);
  ASTK_LAUNCH_CHECK();
  return 0;
}
int fill_zero(void* p, size_t bytes, hipStream_t s) {
  if (bytes == 0) return 0;
  ASTK_HIP(hipMemsetAsync(p, 0, bytes, s));
  return 0;
}
int copy_f32(float* dst, const float* src, size_t n, hipStream_t s) {
  if (n == 0) return 0;
  ASTK_HIP(hipMemcpyAsync(dst, src, n * sizeof(float), hipMemcpyDeviceToDevice, s));
  return 0;
}
int copy2d_f32(float* dst, long ldd, const float* src, long lds, int rows, int cols, int cols_dst, hipStream_t s) {
  if (rows <= 0 || cols_dst <= 0) return 0;
  hipLaunchKernelGGL(k_copy2d, dim3(grid_for((size_t)rows * cols_dst)), dim3(256), 0, s, dst, ldd, src, lds, rows, cols, cols_dst);
  ASTK_LAUNCH_CHECK();
  return 0;
}
int add2d_f32(float* dst, long ldd, const float* src, long lds, int rows, int cols, hipStream_t s) {
  if (rows <= 0 || cols <= 0) return 0;
  hipLaunchKernelGGL(k_add2d, dim3(grid_for((size_t)rows * cols)), dim3(256), 0, s, dst, ldd, src, lds, rows, cols);
  ASTK_LAUNCH_CHECK();
  return 0;
}
int axpy_rows(float* dst, const float* src, size_t n, hipStream_t s) {
  if (n == 0) return 0;
  hipLaunchKernelGGL(k_axpy, dim3(grid_for(n)), dim3(256), 0, s, dst, src, n);
  ASTK_LAUNCH_CHECK();
  return 0;
}
int transpose_f32(float* dst, long ldd, const float* src, long lds, int rows, int cols, hipStream_t s) {
  if (rows <= 0 || cols <= 0) return 0;
  // y blocks cover ldd so that the pad columns [rows, ldd) of every dst row are written as zeros
  hipLaunchKernelGGL(k_transpose, dim3(cdiv(cols, 32), cdiv(ldd, 32)), dim3(256), 0, s, dst, ldd, src, lds, rows, cols);
  ASTK_LAUNCH_CHECK();
  return 0;
}
int transpose_batch(const TransposeJobs& t, hipStream_t s) {
  if (t.n <= 0) return 0;
  int gx = 1, gy = 1;
  for (int i = 0; i < t.n; ++i) { gx = std::max(gx, cdiv(t.cols[i], 32)); gy = std::max(gy, cdiv(t.ldd[i], 32)); }
  hipLaunchKernelGGL(k_transpose_batch, dim3(gx, gy, t.n), dim3(256), 0, s, t);
  ASTK_LAUNCH_CHECK();
  return 0;
}
int copy_segments(const CopySegs& c, hipStream_t s) {
  if (c.n <= 0) return 0;
  size_t mx = 0;
  for (int i = 0; i < c.n; ++i) {
    ASTK_CHECK(aligned16(c.dst[i]) && aligned16(c.src[i]) && (c.bytes[i] % 4) == 0, "copy_segments: segment %d must be 16-byte aligned, size a multiple of 4", i);
    mx = std::max(mx, c.bytes[i]);
  }
  unsigned gx = (unsigned)((mx / 16 + 256 * 4 - 1) / (256 * 4));
  gx = gx < 1 ? 1 : (gx > 512 ? 512 : gx);
  hipLaunchKernelGGL(k_copy_segments, dim3(gx, (unsigned)c.n), dim3(256), 0, s, c);
  ASTK_LAUNCH_CHECK();
  return 0;
}
int colsum_add_f32(float* dst, const float* src, long lds, int rows, int cols, hipStream_t s) {
  if (rows <= 0 || cols <= 0) return 0;
  ASTK_CHECK((lds % 4) == 0 && aligned16(src), "colsum: source must be 16-byte aligned with a leading dimension multiple of 4");
  hipLaunchKernelGGL(k_colsum, colreduce_grid(rows, cols, true), dim3(256), 0, s, dst, src, lds, rows, cols);
  ASTK_LAUNCH_CHECK();
  return 0;
}

int stream_order(hipStream_t from, hipStream_t to) {
  if (from == to) return 0;
  constexpr int NEV = 16;
  static thread_local hipEvent_t pool[NEV];
  static thread_local int next = 0;
  hipEvent_t& e = pool[next];
  next = (next + 1) % NEV;
  if (!e) ASTK_HIP(hipEventCreateWithFlags(&e, hipEventDisableTiming));
  ASTK_HIP(hipEventRecord(e, from));
  ASTK_HIP(hipStreamWaitEvent(to, e, 0));   // the wait refers to this record even if the event is recorded again later
  return 0;
}

int ColsumBatch::add(float* dst, const float* src, long lds, int rows, int cols, hipStream_t s) {
  if (rows <= 0 || cols <= 0) return 0;
  ASTK_CHECK((lds % 4) == 0 && aligned16(src), "colsum: source must be 16-byte aligned with a leading dimension multiple of 4");
  if (j.n == COLSUM_BATCH_MAX) ASTK_TRY(flush(s));
  const dim3 g = colreduce_grid(rows, cols, true);
  const int i = j.n++;
  j.dst[i] = dst; j.src[i] = src; j.lds[i] = lds; j.rows[i] = rows; j.cols[i] = cols; j.gx[i] = (int)g.x; j.gy[i] = (int)g.y;
  return 0;
}
int ColsumBatch::flush(hipStream_t s) {
  if (j.n == 0) return 0;
  int gx = 0, gy = 0;
  for (int i = 0; i < j.n; ++i) { gx = std::max(gx, j.gx[i]); gy = std::max(gy, j.gy[i]); }
  hipLaunchKernelGGL(k_colsum_batch, dim3(gx, gy, j.n), dim3(256), 0, s, j);
  ASTK_LAUNCH_CHECK();
  j.n = 0;
  return 0;
}

}  // namespace astk

namespace astk {
const unsigned* persist_status_word() { return persist_status_ptr(); }
int status_snapshot_launch(float* dst, hipStream_t s) {
  unsigned* st = persist_status_ptr();
  ASTK_CHECK(st, "persist_status: no status word");
  hipLaunchKernelGGL(k_status_snapshot, dim3(1), dim3(1), 0, s, st, dst);
  ASTK_LAUNCH_CHECK();
  return 0;
}
}  // namespace astk

using namespace astk;

extern "C" {

int astk_version(void) { return ASTK_VERSION; }
const char* astk_last_error(void) { return astk::last_error(); }

int astk_grad_sqnorm(const float* g, const float* p, float l2, size_t n, double* sqnorm, void* stream) {
  return astk_grad_sqnorm_scaled(g, p, 1.f, l2, n, sqnorm, stream);
}
int astk_grad_sqnorm_scaled(const float* g, const float* p, float grad_scale, float l2, size_t n, double* sqnorm, void* stream) {
  hipStream_t s = (hipStream_t)stream;
  ASTK_CHECK(g && p && sqnorm, "grad_sqnorm: null pointer");
  ASTK_CHECK(aligned16(g) && aligned16(p), "grad_sqnorm: buffers must be 16-byte aligned");
  hipLaunchKernelGGL(k_sqnorm, dim3(std::min(SQNORM_BLOCKS, grid_for(n / 4 + 1))), dim3(256), 0, s, g, p, grad_scale, l2, n, sqnorm);
  ASTK_LAUNCH_CHECK();
  return 0;
}

int astk_decay_clip_amsgrad_step(float* p, const float* g, float* m, float* v, float* vhat, size_t n, float l2, float clip,
                                 const double* sqnorm, float lr_t, float beta1, float beta2, float eps, int amsgrad,
                                 void* stream) {
  return astk_decay_clip_amsgrad_step_scaled(p, g, m, v, vhat, n, 1.f, l2, clip, sqnorm, lr_t, beta1, beta2, eps, amsgrad, stream);
}
int astk_decay_clip_amsgrad_step_scaled(float* p, const float* g, float* m, float* v, float* vhat, size_t n, float grad_scale, float l2,
                                        float clip, const double* sqnorm, float lr_t, float beta1, float beta2, float eps, int amsgrad,
                                        void* stream) {
  ASTK_CHECK(p && g && m && v && sqnorm && (vhat || !amsgrad), "amsgrad_step: null pointer");
  if (n == 0) return 0;
  hipLaunchKernelGGL(k_amsgrad, dim3(grid_for(n)), dim3(256), 0, (hipStream_t)stream, p, g, m, v, vhat, n, grad_scale, l2, clip, sqnorm,
                     lr_t, beta1, beta2, eps, amsgrad, (const unsigned*)persist_status_ptr());
  ASTK_LAUNCH_CHECK();
  return 0;
}

int astk_decay_clip_sgd_step(float* p, const float* g, size_t n, float l2, float clip, const double* sqnorm, float lr,
                             void* stream) {
  return astk_decay_clip_sgd_step_scaled(p, g, n, 1.f, l2, clip, sqnorm, lr, stream);
}
int astk_decay_clip_sgd_step_scaled(float* p, const float* g, size_t n, float grad_scale, float l2, float clip, const double* sqnorm,
                                    float lr, void* stream) {
  ASTK_CHECK(p && g && sqnorm, "sgd_step: null pointer");
  if (n == 0) return 0;
  hipLaunchKernelGGL(k_sgd, dim3(grid_for(n)), dim3(256), 0, (hipStream_t)stream, p, g, n, grad_scale, l2, clip, sqnorm, lr,
                     (const unsigned*)persist_status_ptr());
  ASTK_LAUNCH_CHECK();
  return 0;
}

int astk_decay_clip_noise(float* g, const float* p, size_t n, float grad_scale, float l2, float clip, const double* sqnorm, float sigma,
                          uint64_t seed, uint64_t offset, void* stream) {
  ASTK_CHECK(g && p && sqnorm && sigma >= 0.f, "decay_clip_noise: bad arguments");
  if (n == 0) return 0;
  hipLaunchKernelGGL(k_decay_clip_noise, dim3(grid_for((n + 1) / 2)), dim3(256), 0, (hipStream_t)stream, g, p, n, grad_scale, l2, clip, sqnorm,
                     sigma, seed, offset);
  ASTK_LAUNCH_CHECK();
  return 0;
}

int astk_fill_dropout_mask(float* out, size_t n, float ratio, uint64_t seed, uint64_t offset, void* stream) {
  ASTK_CHECK(out && ratio >= 0.f && ratio < 1.f, "fill_dropout_mask: bad arguments");
  if (n == 0) return 0;
  hipLaunchKernelGGL(k_dropout_mask, dim3(grid_for(n)), dim3(256), 0, (hipStream_t)stream, out, n, ratio, 1.f / (1.f - ratio),
                     seed, offset);
  ASTK_LAUNCH_CHECK();
  return 0;
}

int astk_fill_random(const astk_rand_seg* segs, int n_segs, void* stream) {
  return astk_fill_random_ex(segs, n_segs, nullptr, 0, nullptr, stream);
}
int astk_fill_random_ex(const astk_rand_seg* segs, int n_segs, const int32_t* words, int n_words, int32_t* words_dst, void* stream) {
  ASTK_CHECK((segs || n_segs == 0) && n_segs >= 0 && n_segs <= ASTK_RAND_SEG_MAX, "fill_random: %d segments (max %d)", n_segs, ASTK_RAND_SEG_MAX);
  ASTK_CHECK(n_words >= 0 && n_words <= ASTK_RAND_WORDS_MAX && (n_words == 0 || (words && words_dst)), "fill_random: %d words (max %d)", n_words,
             ASTK_RAND_WORDS_MAX);
  RandJobs j;
  memset(&j, 0, sizeof(j));
  j.n_words = n_words;
  j.words_dst = words_dst;
  for (int i = 0; i < n_words; ++i) j.words[i] = words[i];
  for (int i = 0; i < n_segs; ++i) {
    const astk_rand_seg& g = segs[i];
    if (g.n == 0) continue;
    ASTK_CHECK(g.out && (g.kind == ASTK_RAND_NORMAL || (g.kind == ASTK_RAND_DROPOUT && g.a >= 0.f && g.a < 1.f)), "fill_random: bad segment %d", i);
    const size_t work = (g.n + 1) / 2;        // both kinds produce two elements per hash
    const int blocks = (int)std::min<size_t>(2048, std::max<size_t>(1, (work + 2047) / 2048));      // ~8 elements per thread, bounded
    j.s[j.n] = g;
    j.blk_start[j.n + 1] = j.blk_start[j.n] + blocks;
    ++j.n;
  }
  if (j.n == 0 && n_words == 0) return 0;
  for (int i = j.n; i < ASTK_RAND_SEG_MAX; ++i) j.blk_start[i + 1] = j.blk_start[j.n];
  hipLaunchKernelGGL(k_fill_random, dim3((unsigned)std::max(1, j.blk_start[j.n])), dim3(256), 0, (hipStream_t)stream, j);
  ASTK_LAUNCH_CHECK();
  return 0;
}

int astk_fill_normal(float* out, size_t n, float mean, float sigma, uint64_t seed, uint64_t offset, void* stream) {
  ASTK_CHECK(out, "fill_normal: null pointer");
  if (n == 0) return 0;
  hipLaunchKernelGGL(k_normal, dim3(grid_for((n + 1) / 2)), dim3(256), 0, (hipStream_t)stream, out, n, mean, sigma, seed, offset);
  ASTK_LAUNCH_CHECK();
  return 0;
}

int astk_zero_frames_draws(int B, int T, const int32_t* lengths, double rate, uint64_t seed, uint64_t offset, int32_t* idx, int max_draws,
                           int32_t* counts, void* stream) {
  ASTK_CHECK(lengths && idx && counts && B > 0 && T > 0 && max_draws > 0 && rate >= 0. && rate <= 1., "zero_frames_draws: bad arguments");
  hipLaunchKernelGGL(k_zero_frames_draws, dim3(B), dim3(256), 0, (hipStream_t)stream, T, lengths, rate, seed, offset, idx, max_draws, counts);
  ASTK_LAUNCH_CHECK();
  return 0;
}

int astk_zero_frames(float* X, int B, int T, int D, const int32_t* lengths, double rate, uint64_t seed, uint64_t offset, void* stream) {
  ASTK_CHECK(X && lengths && B > 0 && T > 0 && D > 0 && rate >= 0. && rate <= 1., "zero_frames: bad arguments");
  if (rate == 0.) return 0;
  hipLaunchKernelGGL(k_zero_frames, dim3(B), dim3(256), 0, (hipStream_t)stream, X, T, D, lengths, rate, seed, offset);
  ASTK_LAUNCH_CHECK();
  return 0;
}

int astk_spin(unsigned usec, unsigned* flag, void* stream) {
  ASTK_CHECK(usec <= 100000, "spin: at most 100 ms");
  hipLaunchKernelGGL(k_spin, dim3(1), dim3(64), 0, (hipStream_t)stream, (unsigned long long)usec * 100ull, flag);
  ASTK_LAUNCH_CHECK();
  return 0;
}

int astk_scale_f32(float* x, size_t n, float s, void* stream) {
  ASTK_CHECK(x, "scale: null pointer");
  if (n == 0) return 0;
  hipLaunchKernelGGL(k_scale, dim3(grid_for(n)), dim3(256), 0, (hipStream_t)stream, x, n, s);
  ASTK_LAUNCH_CHECK();
  return 0;
}

int astk_add_f32(float* dst, const float* src, size_t n, void* stream) {
  ASTK_CHECK(dst && src, "add: null pointer");
  return n ? axpy_rows(dst, src, n, (hipStream_t)stream) : 0;
}
int astk_bridge_states(float* dec_c, float* dec_h, float* enc_c, float* enc_h, int nd, int nl_enc, int n, int B, int h, int to_decoder,
                       void* stream) {
  ASTK_CHECK(dec_c && dec_h && enc_c && enc_h, "bridge_states: null pointer");
  ASTK_CHECK(nd >= 1 && nl_enc >= 1 && n >= 0 && n <= nl_enc && B >= 1 && h >= 4 && (h % 4) == 0, "bridge_states: bad dimensions");
  ASTK_CHECK(aligned16(dec_c) && aligned16(dec_h) && aligned16(enc_c) && aligned16(enc_h), "bridge_states: tensors must be 16-byte aligned");
  if (n == 0) return 0;
  hipLaunchKernelGGL(k_bridge_states, dim3(grid_for((size_t)2 * n * B * nd * (h / 4))), dim3(256), 0, (hipStream_t)stream, dec_c, dec_h, enc_c,
                     enc_h, nd, nl_enc, n, B, h, to_decoder);
  ASTK_LAUNCH_CHECK();
  return 0;
}
int astk_colsum_add_f32(float* dst, const float* src, long lds, int rows, int cols, void* stream) {
  ASTK_CHECK(dst && src && rows > 0 && cols > 0 && lds >= cols, "colsum_add: bad arguments");
  return colsum_add_f32(dst, src, lds, rows, cols, (hipStream_t)stream);
}

int astk_gemm_f32_ex(int layout, int M, int N, int K, const float* A, long lda, const float* B, long ldb, float* C, long ldc,
                     const float* bias, int mode, int ksplit, int batch, long sA, long sB, long sC, int precision, void* stream) {
  ASTK_CHECK(precision >= ASTK_PREC_DEFAULT && precision <= ASTK_PREC_F32, "gemm: precision must be 0 (default), 1 (fp16x2), 2 (bf16x3) or 3 (f32): the enum of astk.h");
  PrecScope ps(precision, ASTK_OPERANDS_DEFAULT);
  DetScope ds(0);        // (the process default: astk_set_tuning("gemm.deterministic", 1) makes plain GEMM calls deterministic too)
  GemmArgs g = gemm_args(M, N, K, mat(A, lda), mat(B, ldb), C, ldc, bias, mode, ksplit);
  g.batch = batch < 1 ? 1 : batch;
  g.sA = sA; g.sB = sB; g.sC = sC;
  return gemm_launch(layout, g, (hipStream_t)stream);
}
int astk_gemm_f32(int layout, int M, int N, int K, const float* A, long lda, const float* B, long ldb, float* C, long ldc,
                  const float* bias, int mode, int ksplit, int batch, long sA, long sB, long sC, void* stream) {
  return astk_gemm_f32_ex(layout, M, N, K, A, lda, B, ldb, C, ldc, bias, mode, ksplit, batch, sA, sB, sC, ASTK_PREC_DEFAULT, stream);
}

int astk_persist_status_snapshot(float* dst, void* stream) {
  ASTK_CHECK(dst, "persist_status_snapshot: null pointer");
  return status_snapshot_launch(dst, (hipStream_t)stream);
}

int astk_persist_status_merge(const float* summed, void* stream) {
  ASTK_CHECK(summed, "persist_status_merge: null pointer");
  unsigned* st = persist_status_ptr();
  ASTK_CHECK(st, "persist_status: no status word");
  hipLaunchKernelGGL(k_status_merge, dim3(1), dim3(1), 0, (hipStream_t)stream, st, summed);
  ASTK_LAUNCH_CHECK();
  return 0;
}

int astk_persist_status(unsigned* mask_out, int reset) {
  unsigned* st = persist_status_ptr();
  ASTK_CHECK(st, "persist_status: no status word");
  ASTK_HIP(hipDeviceSynchronize());
  unsigned v = 0;
  ASTK_HIP(hipMemcpy(&v, st, sizeof(v), hipMemcpyDeviceToHost));
  if (mask_out) *mask_out = v;
  if (reset && v != 0) {
    hipLaunchKernelGGL(k_status_reset, dim3(1), dim3(1), 0, (hipStream_t)0, st);
    ASTK_LAUNCH_CHECK();
    // (the clip norm's arrival counter too: a norm launch that died with the aborted step would have left it short, and every later
    //  norm of the process wrong -- round-5 advice)
    const unsigned zero = 0;
    ASTK_HIP(hipMemcpyToSymbol(HIP_SYMBOL(g_sq_ctr), &zero, sizeof(zero)));
    ASTK_HIP(hipDeviceSynchronize());
  }
  return 0;
}

int astk_device_cu_count(void) { return device_cu_count(); }

int astk_prof_begin(void) {
  for (int c = 0; c < PROF_NCAT; ++c) { g_prof.ev[c].clear(); g_prof.work[c] = 0; g_prof.bytes[c] = 0; }
  if (!g_tick) ASTK_HIP(hipMalloc((void**)&g_tick, 1024 * sizeof(float)));
  ASTK_HIP(hipMemset(g_tick, 0, 1024 * sizeof(float)));
  g_prof.on = true;
  return 0;
}
// res[0..1] attention fwd (ms, launches); [2..3] attention bwd; [4..6] GEMM (ms, launches, flops); [7..8] LSTM cells (ms, launches)
int astk_prof_end(double* res) {
  g_prof.on = false;
  ASTK_HIP(hipDeviceSynchronize());
  double ms[PROF_NCAT];
  double n[PROF_NCAT];
  for (int c = 0; c < PROF_NCAT; ++c) {
    ms[c] = 0; n[c] = 0;
    std::vector<hipEvent_t>& v = g_prof.ev[c];
    for (size_t i = 0; i + 1 < v.size(); i += 2) {
      float t = 0.f;
      if (hipEventElapsedTime(&t, v[i], v[i + 1]) == hipSuccess) { ms[c] += t; n[c] += 1; }
    }
    for (hipEvent_t e : v) (void)hipEventDestroy(e);
    v.clear();
  }
  res[0] = ms[PROF_ATTN_FWD]; res[1] = n[PROF_ATTN_FWD];
  res[2] = ms[PROF_ATTN_BWD]; res[3] = n[PROF_ATTN_BWD];
  res[4] = ms[PROF_GEMM]; res[5] = n[PROF_GEMM]; res[6] = g_prof.work[PROF_GEMM];
  res[7] = ms[PROF_CELL]; res[8] = n[PROF_CELL];
  // in-kernel attention-phase timing of the persistent decoder kernels: [9] fwd mean us, [10] fwd max-over-workgroups us,
  // [11] launches ; [12] bwd mean, [13] bwd max, [14] launches
  for (int k = 9; k < 24; ++k) res[k] = 0;
  res[16] = ms[PROF_DEC_FWD]; res[17] = n[PROF_DEC_FWD]; res[18] = ms[PROF_DEC_BWD]; res[19] = n[PROF_DEC_BWD];
  res[20] = g_prof.bytes[PROF_GEMM];
  if (g_tick) {
    float h[1024];
    ASTK_HIP(hipMemcpy(h, g_tick, sizeof(h), hipMemcpyDeviceToHost));
    for (int w = 0; w < 2; ++w) {
      const float* t = h + 512 * w;
      const double launches = t[256];
      if (launches > 0) {
        double sum = 0, mx = 0; int cnt = 0;
        for (int i = 0; i < 256; ++i) if (t[i] > 0) { sum += t[i]; if (t[i] > mx) mx = t[i]; ++cnt; }
        res[9 + 3 * w] = cnt ? sum / cnt / launches : 0;
        res[10 + 3 * w] = mx / launches;
        res[11 + 3 * w] = launches;
      }
    }
  }
  return 0;
}

}  // extern "C"

extern "C" int astk_set_tuning(const char* key, double value) {
  const int i = astk::tune_find(key);
  if (i < 0) { astk::set_error("set_tuning: unknown key '%s'", key ? key : "(null)"); return -1; }
  astk::tune_init();
  astk::g_tune[i].store(value);
  return 0;
}
extern "C" int astk_get_tuning(const char* key, double* value) {
  const int i = astk::tune_find(key);
  if (i < 0 || !value) { astk::set_error("get_tuning: unknown key '%s'", key ? key : "(null)"); return -1; }
  *value = astk::tune((astk::TuneKey)i);
  return 0;
}
extern "C" const char* astk_tuning_key(int index) { return index >= 0 && index < astk::TUNE_COUNT ? astk::g_tune_table[index].name : nullptr; }
