// Encoder LSTM stacks (seq2seq.py:182-242): n_dirs independent stacks of L.LSTM links.
//
// Schedule (one HIP stream, graph-capturable, no host sync):
//   per layer: ONE batched-over-time MFMA GEMM for the upward projection of each direction
//              (ZG[d][l] = X W_u^T + b, rows in loop-step order; the reverse direction reads frames through the
//              Q1 permutation table 0,T-1,...,1 instead of a permuted copy), then T launches of the fused cell
//              kernel (both directions in one launch: lateral product on f32 MFMA + interleaved-gate epilogue,
//              activated gates overwrite ZG in place, dropped output written straight into the (B,T,H)
//              enc_states slice for the top layer -- no concat growth, no flipud copy).
//   backward : per layer, top down: T launches of the fused backward cell (dh_rec = dz_{t+1} Wl via the
//              transposed weight, gate derivatives, dz overwrites the gates in place), then batched GEMMs for
//              dWl, dWu, db and the gradient wrt the layer input.
// HBM layout (per direction d, layer l, all f32, step-major):
//   ZG (T,B,4h) gates -> dz | HR (T,B,h) raw h | CC (T,B,h) cell | HD (T,B,h) dropped output (only with masks)
#include "common.h"
#include <algorithm>
#include <vector>

namespace astk {

struct PersistCellHost {
  const float *Wl, *Wu, *bias, *zx, *xin, *mask, *WlT, *d_enc, *d_hT, *d_cT;
  float *gates, *C, *HR, *HD, *enc;
  const float* WuT;
  float *PR, *PD;
  const float* PD_up;
  int up_external;
  int reverse_pos, layer;
  unsigned long long* amax;
  float* db;
  long dy_sb, dy_st;
  const unsigned* zx_flags; int zx_s0, zx_cs;
  unsigned* prog; int prog_cs;
  float* db_part;
};
bool lstm_persist_hoisted(int h);
bool lstm_persist_applicable(int T, int B, int h, int nl, int nd);
int lstm_persist_rows(int B, int h, int nl, int nd, bool side);
int lstm_persist_layers_per_launch(int B, int h, int nl, int nd, int rows);
int lstm_persist_grid_wgs(int B, int h, int layers, int nd, int rows);
size_t lstm_persist_pr_floats(int B, int h);
size_t lstm_persist_pd_floats(int T, int B, int h);
int lstm_persist_fwd_launch(const PersistCellHost* cells, int ncells, int nl, int T, int B, int h, int H, unsigned* counters, int rows, hipStream_t s);
int lstm_persist_bwd_launch(const PersistCellHost* cells, int ncells, int nl, int T, int B, int h, int H, unsigned* counters, unsigned amax_gen,
                            int rows, hipStream_t s);

namespace {

constexpr int SIDE_CHUNKS_MAX = 60;

// ---- work beside the recurrences (astk_lstm_stack_desc.side_stream): the plan of the forward pass.
// The layer-0 input projection (per direction (T B) x 4h x in, the largest product of the step) is cut along time: steps [0, s0) are
// multiplied in line, at full width, in front of the recurrence launch; the rest in chunks of `cs` steps on the side stream, every launch
// capped at `cap` workgroups (the CUs the recurrence grid leaves free), a one-lane kernel behind each chunk raising its flag.  A chunk holds as
// many 128-row tile rows as give `cap` tiles over all directions, so a chunk launch is ONE WHOLE TILE PER WORKGROUP: plain stores, no split
// tiles, the same sum order in every run.  s0 is the smallest head for which, by a rate model (a tile pass ~ 0.66 us per 16 k + 15 us; a
// recurrence step 2.9 / 4.4 us at 16 / 32 rows per workgroup, 3.8 at h = 512), no chunk is late; if the model is wrong the layer-0 cells wait
// on a flag (bounded like every other hand-off) -- slower, never wrong.  n = 0: everything in line.
struct SidePlan { int s0, cs, n, cap; };
SidePlan plan_side_fwd(const astk_lstm_stack_desc* d, int rows, int wgs_first_launch) {
  SidePlan sp = {d->T, 0, 0, 0};
  if (!d->side_stream || lstm_persist_hoisted(d->h) || !tune_on(TUNE_LSTM_SIDE_FWD) || d->deterministic || tune_on(TUNE_GEMM_DETERMINISTIC)) return sp;
  int cap = device_cu_count() - wgs_first_launch;
  if (d->side_wgs > 0) cap = std::min(cap, d->side_wgs);
  cap = cap / 8 * 8;
  const int tiles_per_row = d->n_dirs * ((4 * d->h + 127) / 128);       // 128 x 128 tiles per 128 rows of all directions
  const int tile_rows = cap / tiles_per_row;
  if (cap < 16 || tile_rows < 1) return sp;
  int cs = (int)tune(TUNE_LSTM_OVERLAP_CHUNK);
  if (cs <= 0) cs = tile_rows * 128 / d->B;
  if (cs < 4 || d->T < 3 * cs / 2) return sp;
  const double t_chunk = ((d->in_dim + 15) / 16) * 0.66 + 15.0;           // one tile pass (every workgroup of a chunk launch does one)
  const double r_step = (d->h > 256 ? 3.8 : rows == 32 ? 4.4 : rows == 33 ? 3.3 : 2.9) * 0.9; // (10 % margin)
  // (the side stream starts together with the in-line head, so the chunks have the head's time -- ~190 TFLOP/s on the whole chip -- on top)
  const double head_step = 0.8 * d->n_dirs * 2.0 * d->B * 4.0 * d->h * d->in_dim / 190e6;
  for (int n = std::min(SIDE_CHUNKS_MAX, (d->T - 2) / cs); n >= 1; --n) {
    const int s0 = d->T - n * cs;
    bool ok = s0 >= 2;
    for (int k = 0; k < n && ok; ++k) ok = (k + 1) * t_chunk <= s0 * head_step + (s0 + (double)k * cs) * r_step;
    if (ok) { sp.s0 = s0; sp.cs = cs; sp.n = n; sp.cap = cap; return sp; }
  }
  return sp;
}
__global__ void k_set_flag(unsigned* f) {
  if (threadIdx.x == 0) __hip_atomic_store(f, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// ---- the backward pass beside its recurrence.  The layer-0 cells write dz through and arrive on a progress counter per direction once per
// chunk of `cs` loop steps (lstm_persist_bwd_rs, PCellB::prog); on the side stream a one-wave kernel waits for a chunk's arrivals -- bounded
// spin with s_sleep; a time-out sets the encoder-backward bit of the sticky status word, so the step is reported, and exits, so nothing
// hangs -- and the input-gradient products of that chunk follow it in stream order (a kernel boundary behind the wait: their loads see
// the written-through dz).  Every launch is capped at the CUs the recurrence grid leaves free.
__global__ void k_wait_progress(const unsigned* p0, const unsigned* p1, unsigned target, AbortCtl ab) {
  if (threadIdx.x != 0) return;
  unsigned spins = 0;
  while (__hip_atomic_load(p0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target ||
         (p1 && __hip_atomic_load(p1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target)) {
    __builtin_amdgcn_s_sleep(32);
    if (++spins > ab.limit) { abort_raise(ab); return; }
    if ((spins & 63u) == 0 && abort_seen(ab)) return;
  }
}
// deterministic calls: db[cell][c] += sum over the workgroup rows, in order, of the sums the persistent backward kernel left per row
struct FoldDbJobs { int n, nby, cols; float* db[16]; const float* part[16]; };
__global__ void k_fold_db(FoldDbJobs j) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x, q = blockIdx.y;
  if (c >= j.cols || q >= j.n) return;
  float sum = 0.f;
  for (int by = 0; by < j.nby; ++by) sum += j.part[q][(long)by * j.cols + c];
  j.db[q][c] += sum;
}
__global__ void k_zero_words(unsigned* p, int n, int stride) {
  if ((int)threadIdx.x < n) __hip_atomic_store(p + threadIdx.x * stride, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
SidePlan plan_side_bwd(const astk_lstm_stack_desc* d, int rows, int lpl, bool want_dx) {
  SidePlan sp = {d->T, 0, 0, 0};
  if (!d->side_stream || !want_dx || lstm_persist_hoisted(d->h) || lpl < d->n_layers || low_precision_gemms() || !tune_on(TUNE_LSTM_SIDE_BWD) || deterministic_mode()) return sp;
  int cap = device_cu_count() - lstm_persist_grid_wgs(d->B, d->h, d->n_layers, d->n_dirs, rows);
  if (d->side_wgs > 0) cap = std::min(cap, d->side_wgs);
  cap = cap / 8 * 8;
  if (cap < 16) return sp;
  int cs = (int)tune(TUNE_LSTM_OVERLAP_CHUNK);
  // (as many 128-row tile rows of ONE direction's product as give at most `cap` tiles: one whole tile per workgroup, no split tiles)
  if (cs <= 0) cs = std::max(4, std::max(1, cap / ((d->in_dim + 127) / 128)) * 128 / d->B);
  cs = std::max(4, cs);
  const int nall = std::min(SIDE_CHUNKS_MAX, (d->T + cs - 1) / cs);
  if (nall < 2) return sp;
  // "lstm.side_bwd" = how many chunks (the LAST loop steps, which the backward passes first) go to the side stream; the rest of dx follows in
  // line, at full width, behind the recurrence.  (< 0: every chunk.)  The side stream is the caller's: what else it has queued there --
  // the decoder's parameter gradients in the train step -- decides how many chunks finish inside the recurrence's duration.
  const int want = (int)tune(TUNE_LSTM_SIDE_BWD);
  sp.cs = cs; sp.n = want < 0 ? nall : std::min(want, nall); sp.cap = cap; sp.s0 = 0;
  return sp;
}

struct LstmPlan {
  int T, B, in, h, nl, nd;
  int* perm;      // [T] frame consumed at loop step i by direction 1: (T-i)%T
  int* inv;       // [T] loop step of direction 1 that consumed frame f
  float* ZG[2][ASTK_MAX_RNN_LAYERS];
  float* HR[2][ASTK_MAX_RNN_LAYERS];
  float* CC[2][ASTK_MAX_RNN_LAYERS];
  float* HD[2][ASTK_MAX_RNN_LAYERS];
  float* WlT[2][ASTK_MAX_RNN_LAYERS];  // (h, 4h)
  float* WuT[2][ASTK_MAX_RNN_LAYERS];  // (h, 4h) transposed upward weights of layers >= 1 (persistent backward)
  unsigned* counters;                  // arrival counters of the persistent kernels
  unsigned* zflags;                    // side-stream chunks: forward [SIDE_CHUNKS_MAX + 2] chunk flags, then 2 progress counters of the backward and the wait kernels' abort word (one word per 256-byte line)
  float* PR[2][ASTK_MAX_RNN_LAYERS];   // persistent backward (reduce-scatter): partial dh_rec ring of each cell
  float* PD[2][ASTK_MAX_RNN_LAYERS];   // partial dx handed to the layer below (layers >= 1)
  unsigned long long* ax;              // the frames' maximum, folded from desc.x_amax into one line by the forward call (read by both calls)
  float* GATH;                         // (T,B,4h) dz of the reverse stack's layer 0 re-ordered to frame order
  float* DX[2];                        // (T,B,h) gradient wrt a layer's input (layers >= 1)
  float* DC[2][2];                     // dc ping-pong (B,h)
  float* DBP[2][ASTK_MAX_RNN_LAYERS];  // deterministic calls: [workgroup rows][4h] bias-gradient sums of the persistent backward kernel
  size_t bytes;
};

int make_plan(const astk_lstm_stack_desc* d, void* ws, bool with_masks, LstmPlan& P) {
  ASTK_CHECK_DESC(d, astk_lstm_stack_desc);
  ASTK_CHECK(d && d->T > 0 && d->B > 0 && d->in_dim > 0 && d->h > 0, "lstm_stack: bad dims");
  ASTK_CHECK(d->n_layers >= 1 && d->n_layers <= ASTK_MAX_RNN_LAYERS && (d->n_dirs == 1 || d->n_dirs == 2), "lstm_stack: layers/dirs");
  ASTK_CHECK((d->in_dim % 4) == 0 && (d->h % 4) == 0, "lstm_stack: in_dim and h must be multiples of 4");
  P.T = d->T; P.B = d->B; P.in = d->in_dim; P.h = d->h; P.nl = d->n_layers; P.nd = d->n_dirs;
  Carver c(ws);
  const size_t tb = (size_t)P.T * P.B;
  P.perm = c.take<int>(P.T);
  P.inv = c.take<int>(P.T);
  for (int dd = 0; dd < P.nd; ++dd) {
    for (int l = 0; l < P.nl; ++l) {
      P.ZG[dd][l] = c.take<float>(tb * 4 * P.h);
      P.HR[dd][l] = c.take<float>(tb * P.h);
      P.CC[dd][l] = c.take<float>(tb * P.h);
      // the workspace size must not depend on whether masks are passed: always reserve HD
      P.HD[dd][l] = c.take<float>(tb * P.h);
      P.WlT[dd][l] = c.take<float>((size_t)P.h * 4 * P.h);
      P.WuT[dd][l] = c.take<float>((size_t)P.h * 4 * P.h);
    }
    P.DX[dd] = c.take<float>(tb * P.h);
    P.DC[dd][0] = c.take<float>((size_t)P.B * P.h);
    P.DC[dd][1] = c.take<float>((size_t)P.B * P.h);
  }
  (void)with_masks;
  for (int dd = 0; dd < P.nd; ++dd)
    for (int l = 0; l < P.nl; ++l) P.DBP[dd][l] = c.take<float>((size_t)(2 * ((P.B + 31) / 32)) * 4 * P.h);
  P.GATH = c.take<float>(P.nd > 1 ? tb * 4 * P.h : 4);
  P.ax = c.take<unsigned long long>(AMAX_SLOT_WORDS);
  P.counters = c.take<unsigned>(((size_t)2 * P.nd * P.nl * (2 * ((P.B + 31) / 32)) + 2) * 64);      // (an even number of 16-row tiles: the 32-row forms)
  P.zflags = c.take<unsigned>((size_t)(SIDE_CHUNKS_MAX + 2 + 3) * 64);      // chunk flags (+ 2 zero words), two progress counters, the wait kernels' abort word
  {
    const bool pp = lstm_persist_applicable(P.T, P.B, P.h, P.nl, P.nd);
    for (int dd = 0; dd < P.nd; ++dd)
      for (int l = 0; l < P.nl; ++l) {
        // (hoisted form: the layers run one launch after the other, every cell of a direction uses the ring of layer 0; no down partials)
        const bool hoist = lstm_persist_hoisted(P.h);
        P.PR[dd][l] = (hoist && l > 0) ? P.PR[dd][0] : c.take<float>(pp ? lstm_persist_pr_floats(P.B, P.h) : 4);
        P.PD[dd][l] = c.take<float>(pp && l > 0 && !hoist ? lstm_persist_pd_floats(P.T, P.B, P.h) : 4);
      }
  }
  P.bytes = c.total();
  return 0;
}

// quirk Q1: the reverse stack reads frame X[-i] = (T - i) % T at step i -- an involution, so the frame permutation is its own inverse.
// One launch writes it (perm = inv) and its expansions to (T*B) row indices: rows[i*B+b] = perm[i]*B + b.
__global__ void k_perm_rows(int* perm, int* inv, int* rows_perm, int* rows_inv, int T, int B, const unsigned long long* fold_src,
                            unsigned long long* fold_dst, unsigned* zflags, int nzf) {
  // (rides along: the frames' maximum arrives in a STRIDED producer slot -- thousands of blocks wrote it -- and the GEMMs read one line)
  if (fold_src && blockIdx.x == 0 && threadIdx.x < 64) amax_compact(fold_src, fold_dst);
  // (rides along: the chunk flags of the side-stream plan go down before the side stream is let loose)
  if (zflags && blockIdx.x == 0 && (int)threadIdx.x < nzf) __hip_atomic_store(zflags + threadIdx.x * 64, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < T) {
    const int f = (T - i) % T;
    perm[i] = f;
    inv[f] = i;
  }
  if (i < T * B) {
    const int r = ((T - i / B) % T) * B + (i % B);
    rows_perm[i] = r;
    rows_inv[i] = r;
  }
}
// dst[r][:] = src[idx[r]][:]  (float4 columns)
__global__ void k_gather_rows(float* __restrict__ dst, const float* __restrict__ src, const int* __restrict__ idx, int rows, int cols4) {
  const long n = (long)rows * cols4;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    const int r = (int)(i / cols4), c = (int)(i % cols4);
    reinterpret_cast<float4*>(dst)[i] = reinterpret_cast<const float4*>(src)[(long)idx[r] * cols4 + c];
  }
}
}  // namespace
}  // namespace astk

using namespace astk;

extern "C" {

int astk_lstm_stack_path(const astk_lstm_stack_desc* d) {
  if (!d || d->struct_size != sizeof(astk_lstm_stack_desc)) return 0;
  return lstm_persist_applicable(d->T, d->B, d->h, d->n_layers, d->n_dirs) ? (lstm_persist_hoisted(d->h) ? 2 : 1) : 0;
}

int astk_lstm_stack_free_cus(const astk_lstm_stack_desc* d) {
  if (!d || d->struct_size != sizeof(astk_lstm_stack_desc)) return 0;
  if (!lstm_persist_applicable(d->T, d->B, d->h, d->n_layers, d->n_dirs)) return 0;
  const int rows = lstm_persist_rows(d->B, d->h, d->n_layers, d->n_dirs, d->side_stream != nullptr);
  const int lpl = lstm_persist_layers_per_launch(d->B, d->h, d->n_layers, d->n_dirs, rows);
  const int wgs = lstm_persist_grid_wgs(d->B, d->h, std::min(lpl, d->n_layers), d->n_dirs, rows);
  return std::max(0, device_cu_count() - wgs);
}

size_t astk_lstm_stack_workspace_bytes(const astk_lstm_stack_desc* d) {
  LstmPlan P;
  if (make_plan(d, nullptr, true, P) != 0) return 0;
  // + two (T*B) row-index tables
  return P.bytes + 2 * align_up((size_t)d->T * d->B * sizeof(int), 256);
}

int astk_lstm_stack_fwd(const astk_lstm_stack_desc* d, const astk_lstm_params* prm, const float* x, const float* masks,
                        float* enc_states, float* cT, float* hT, void* ws, size_t ws_bytes, void* stream) {
  hipStream_t s = (hipStream_t)stream;
  ASTK_CHECK_DESC(d, astk_lstm_stack_desc);
  PrecScope prec_scope(d->precision, d->gemm_operands);
  GemmForwardScope forward_scope;      // split tiles of this op's products have at most two contributors (reproducible forward pass)
  LstmPlan P;
  ASTK_TRY(make_plan(d, ws, masks != nullptr, P));
  const size_t need = astk_lstm_stack_workspace_bytes(d);
  ASTK_CHECK(ws && ws_bytes >= need, "lstm_stack_fwd: workspace too small (%zu < %zu)", ws_bytes, need);
  ASTK_CHECK(prm && x && enc_states, "lstm_stack_fwd: null pointer");
  const int T = P.T, B = P.B, h = P.h, H = P.nd * P.h;
  int* rows_perm = (int*)((char*)ws + P.bytes);
  int* rows_inv = (int*)((char*)rows_perm + align_up((size_t)T * B * sizeof(int), 256));
  // a strided producer slot (bit 0 of the handle) is folded into the plan's plain slot; a plain one is used as it is
  const bool x_strided = d->x_amax && (((uintptr_t)d->x_amax) & 1u);
  const bool persist_path = lstm_persist_applicable(T, B, h, P.nl, P.nd);
  const int rows_wg = persist_path ? lstm_persist_rows(B, h, P.nl, P.nd, d->side_stream != nullptr) : 16;
  const int lpl_f = persist_path ? lstm_persist_layers_per_launch(B, h, P.nl, P.nd, rows_wg) : 0;
  const SidePlan side = persist_path ? plan_side_fwd(d, rows_wg, lstm_persist_grid_wgs(B, h, std::min(lpl_f, P.nl), P.nd, rows_wg)) : SidePlan{T, 0, 0, 0};
  hipStream_t sside = (hipStream_t)d->side_stream;
  hipLaunchKernelGGL(k_perm_rows, dim3(cdiv(T * B, 256)), dim3(256), 0, s, P.perm, P.inv, rows_perm, rows_inv, T, B,
                     x_strided ? (const unsigned long long*)d->x_amax : nullptr, P.ax, side.n > 0 ? P.zflags : nullptr, side.n + 2);
  ASTK_LAUNCH_CHECK();
  const size_t bh = (size_t)B * h;
  if (persist_path) {
    // ---- persistent wavefront path: layer-0 upward projection batched over time, everything else in ONE launch
    PersistCellHost cells[16];
    memset(cells, 0, sizeof(cells));
    // maxima (fp16x2 GEMM scales) of the frames -- both directions multiply the same ones -- and of the two layer-0 upward weights in
    // ONE launch: the grouped projection launch below then needs no maximum pass of its own
    // (the frames' maximum comes with them when the caller passes it on from the kernel that wrote them: desc.x_amax)
    const unsigned long long* ax = x_strided ? P.ax : (const unsigned long long*)d->x_amax;
    const unsigned long long* aw0[2] = {nullptr, nullptr};
    {
      AmaxMatrix am[3] = {{ax ? nullptr : x, (long)T * B, (long)P.in, P.in}, {prm[0].Wu, 4L * h, (long)P.in, P.in},
                          {P.nd > 1 ? prm[P.nl].Wu : nullptr, 4L * h, (long)P.in, P.in}};
      const unsigned long long* out[3];
      gemm_amax_many(am, 3, out, s);
      if (!ax) ax = out[0];
      aw0[0] = out[1]; aw0[1] = out[2];
    }
    GemmArgs k9[2];      // the layer-0 upward projections of both directions: one grouped launch
    for (int dd = 0; dd < P.nd; ++dd) {
      const astk_lstm_params& p0 = prm[dd * P.nl];
      ASTK_CHECK(p0.Wu && p0.b && p0.Wl, "lstm_stack_fwd: null parameter (dir %d layer 0)", dd);
      MatView A = dd == 0 ? mat(x, P.in) : mat_idx(x, P.in, rows_perm);
      k9[dd] = with_amax_b(with_amax_a(lowp(gemm_args(T * B, 4 * h, P.in, A, mat(p0.Wu, P.in), P.ZG[dd][0], 4 * h, p0.b)), ax), aw0[dd]);     // K9
      for (int l = 0; l < P.nl; ++l) {
        const astk_lstm_params& p = prm[dd * P.nl + l];
        ASTK_CHECK(p.Wu && p.b && p.Wl, "lstm_stack_fwd: null parameter (dir %d layer %d)", dd, l);
        PersistCellHost& c = cells[dd * P.nl + l];
        const bool top = l == P.nl - 1;
        c.Wl = p.Wl;
        c.Wu = l > 0 ? p.Wu : nullptr;
        c.bias = l > 0 ? p.b : nullptr;
        c.zx = l == 0 ? P.ZG[dd][0] : nullptr;
        c.gates = P.ZG[dd][l];
        c.C = P.CC[dd][l];
        c.HR = P.HR[dd][l];
        c.HD = (!top && masks) ? P.HD[dd][l] : nullptr;
        c.xin = l > 0 ? (masks ? P.HD[dd][l - 1] : P.HR[dd][l - 1]) : nullptr;
        c.mask = masks ? masks + ((size_t)dd * P.nl + l) * T * bh : nullptr;
        c.enc = top ? enc_states + (size_t)dd * h : nullptr;
        c.reverse_pos = dd == 1;
        c.layer = l;
      }
    }
    if (side.n == 0) {
      ASTK_TRY(gemm_launch_group(GEMM_NT, k9, P.nd, s));
    } else {
      // time-chunked: the head in line, the rest on the side stream beside the recurrence (plan_side_fwd).  Rows are loop-step major in
      // both directions (direction 1 reads its frames through the permutation table), so a chunk is a row range of both products.
      auto rows_of = [&](int s_begin, int s_end, GemmArgs* out) {
        for (int dd = 0; dd < P.nd; ++dd) {
          GemmArgs g = k9[dd];
          const size_t r0 = (size_t)s_begin * B;
          g.M = (s_end - s_begin) * B;
          if (dd == 0) g.A.p = x + r0 * P.in; else { g.A.rowidx = rows_perm + r0; g.A.idx_rows = (long)T * B; }
          g.C = P.ZG[dd][0] + r0 * 4 * h;
          out[dd] = g;
        }
      };
      GemmArgs part[2];
      ASTK_TRY(stream_order(s, sside));                    // the frames, the index table, the lowered flags
      rows_of(0, side.s0, part);
      ASTK_TRY(gemm_launch_group(GEMM_NT, part, P.nd, s));
      {
        GemmWgCap cap_scope(side.cap);
        for (int k = 0; k < side.n; ++k) {
          rows_of(side.s0 + k * side.cs, std::min(T, side.s0 + (k + 1) * side.cs), part);
          ASTK_TRY(gemm_launch_group(GEMM_NT, part, P.nd, sside));
          hipLaunchKernelGGL(k_set_flag, dim3(1), dim3(64), 0, sside, P.zflags + (size_t)k * 64);
          ASTK_LAUNCH_CHECK();
        }
      }
      for (int dd = 0; dd < P.nd; ++dd) {
        cells[dd * P.nl].zx_flags = P.zflags; cells[dd * P.nl].zx_s0 = side.s0; cells[dd * P.nl].zx_cs = side.cs;
      }
    }
    if (lstm_persist_hoisted(h)) {
      // hoisted form: every layer a launch of its own over "layer-0 like" cells -- the input projection of all time steps comes from a
      // batched product in front of the launch (written into the gates buffer, where the cell replaces it step by step)
      for (int l = 0; l < P.nl; ++l) {
        PersistCellHost grp[16];
        GemmArgs up[2];
        for (int dd = 0; dd < P.nd; ++dd) {
          PersistCellHost& c = cells[dd * P.nl + l];
          if (l > 0) up[dd] = gemm_args(T * B, 4 * h, h, mat(c.xin, h), mat(c.Wu, h), P.ZG[dd][l], 4 * h, c.bias);
          c.Wu = nullptr; c.bias = nullptr; c.xin = nullptr;
          c.zx = P.ZG[dd][l];
          grp[dd] = c;
        }
        if (l > 0) ASTK_TRY(gemm_launch_group(GEMM_NT, up, P.nd, s));
        ASTK_TRY(lstm_persist_fwd_launch(grp, P.nd, 1, T, B, h, H, P.counters, 16, s));
      }
    } else {
      // one launch per group of layers (normally a single group: the whole stack); a later group finds the outputs of the layer
      // below complete (its sentinel polls succeed at once)
      const int lpl = lpl_f;
      for (int l0 = 0; l0 < P.nl; l0 += lpl) {
        const int ngl = std::min(lpl, P.nl - l0);
        PersistCellHost grp[16];
        for (int dd = 0; dd < P.nd; ++dd)
          for (int l = 0; l < ngl; ++l) grp[dd * ngl + l] = cells[dd * P.nl + l0 + l];
        ASTK_TRY(lstm_persist_fwd_launch(grp, P.nd * ngl, ngl, T, B, h, H, P.counters, rows_wg, s));
      }
      if (side.n > 0) ASTK_TRY(stream_order(sside, s));      // join: the caller sees one-stream semantics
    }
    CopySegs cp;   // final states of every cell: one launch
    cp.n = 0;
    for (int dd = 0; dd < P.nd; ++dd)
      for (int l = 0; l < P.nl; ++l) {
        if (cp.n + 2 > FILL_SEG_MAX) { ASTK_TRY(copy_segments(cp, s)); cp.n = 0; }
        if (cT) copy_seg_add(cp, cT + ((size_t)dd * P.nl + l) * bh, P.CC[dd][l] + (size_t)(T - 1) * bh, bh * sizeof(float));
        if (hT) copy_seg_add(cp, hT + ((size_t)dd * P.nl + l) * bh, P.HR[dd][l] + (size_t)(T - 1) * bh, bh * sizeof(float));
      }
    ASTK_TRY(copy_segments(cp, s));
    return 0;
  }
  for (int l = 0; l < P.nl; ++l) {
    const int in = l == 0 ? P.in : h;
    for (int dd = 0; dd < P.nd; ++dd) {
      const astk_lstm_params& p = prm[dd * P.nl + l];
      ASTK_CHECK(p.Wu && p.b && p.Wl, "lstm_stack_fwd: null parameter (dir %d layer %d)", dd, l);
      MatView A;
      if (l == 0) A = dd == 0 ? mat(x, in) : mat_idx(x, in, rows_perm);
      else A = mat(masks ? P.HD[dd][l - 1] : P.HR[dd][l - 1], h);
      ASTK_TRY(gemm_launch(GEMM_NT, gemm_args(T * B, 4 * h, in, A, mat(p.Wu, in), P.ZG[dd][l], 4 * h, p.b), s));
    }
    const bool top = l == P.nl - 1;
    for (int i = 0; i < T; ++i) {
      LstmCellFwdArgs cells[2];
      for (int dd = 0; dd < P.nd; ++dd) {
        const astk_lstm_params& p = prm[dd * P.nl + l];
        LstmCellFwdArgs& c = cells[dd];
        memset(&c, 0, sizeof(c));
        c.npairs = 1;
        c.p[0].A = i > 0 ? P.HR[dd][l] + (size_t)(i - 1) * bh : nullptr;
        c.p[0].lda = h;
        c.p[0].W = p.Wl;
        c.p[0].ldw = h;
        c.p[0].K = i > 0 ? h : 0;     // h is None at the first step: lateral skipped (Chainer-sem A1)
        c.B = B; c.h = h;
        c.zx = P.ZG[dd][l] + (size_t)i * B * 4 * h;
        c.ld_zx = 4 * h;
        c.c_prev = i > 0 ? P.CC[dd][l] + (size_t)(i - 1) * bh : nullptr;
        c.gates = P.ZG[dd][l] + (size_t)i * B * 4 * h;
        c.ld_g = 4 * h;
        c.c_out = P.CC[dd][l] + (size_t)i * bh;
        c.h_out = P.HR[dd][l] + (size_t)i * bh;
        c.mask = masks ? masks + (((size_t)dd * P.nl + l) * T + i) * bh : nullptr;
        if (!top && masks) { c.hd_out = P.HD[dd][l] + (size_t)i * bh; c.ld_hd = h; }
        if (top) {
          const int pos = dd == 0 ? i : T - 1 - i;   // flipud of the reverse stack's output list
          c.hd_out2 = enc_states + (size_t)pos * H + (size_t)dd * h;
          c.ld_hd2 = (long)T * H;
        }
      }
      ASTK_TRY(lstm_cell_fwd_launch(cells, P.nd, s));
    }
    for (int dd = 0; dd < P.nd; ++dd) {
      if (cT) ASTK_TRY(copy_f32(cT + ((size_t)dd * P.nl + l) * bh, P.CC[dd][l] + (size_t)(T - 1) * bh, bh, s));
      if (hT) ASTK_TRY(copy_f32(hT + ((size_t)dd * P.nl + l) * bh, P.HR[dd][l] + (size_t)(T - 1) * bh, bh, s));
    }
  }
  return 0;
}

int astk_lstm_stack_bwd(const astk_lstm_stack_desc* d, const astk_lstm_params* prm, const astk_lstm_grads* gr, const float* x,
                        const float* masks, const float* d_enc, const float* d_cT, const float* d_hT, float* dx, void* ws,
                        size_t ws_bytes, void* stream) {
  return astk_lstm_stack_bwd_on(d, prm, gr, x, masks, d_enc, d_cT, d_hT, dx, ws, ws_bytes, stream, nullptr);
}

int astk_lstm_stack_bwd_on(const astk_lstm_stack_desc* d, const astk_lstm_params* prm, const astk_lstm_grads* gr, const float* x,
                           const float* masks, const float* d_enc, const float* d_cT, const float* d_hT, float* dx, void* ws,
                           size_t ws_bytes, void* stream, void* recurrence_stream) {
  hipStream_t s = (hipStream_t)stream;
  hipStream_t sr = recurrence_stream ? (hipStream_t)recurrence_stream : s;
  ASTK_CHECK_DESC(d, astk_lstm_stack_desc);
  PrecScope prec_scope(d->precision, d->gemm_operands);
  DetScope det_scope(d->deterministic);
  LstmPlan P;
  ASTK_TRY(make_plan(d, ws, masks != nullptr, P));
  const size_t need = astk_lstm_stack_workspace_bytes(d);
  ASTK_CHECK(ws && ws_bytes >= need, "lstm_stack_bwd: workspace too small");
  ASTK_CHECK(prm && gr && x && d_enc, "lstm_stack_bwd: null pointer");
  const int T = P.T, B = P.B, h = P.h, H = P.nd * P.h;
  int* rows_perm = (int*)((char*)ws + P.bytes);
  int* rows_inv = (int*)((char*)rows_perm + align_up((size_t)T * B * sizeof(int), 256));
  const size_t bh = (size_t)B * h;
  const bool persist = lstm_persist_applicable(T, B, h, P.nl, P.nd);
  const bool rs_path = persist;
  SidePlan bside = {T, 0, 0, 0};
  // dx (T,B,in) = dz_0 W_u0 of both directions for the loop steps [i0, i1) (side-stream chunks, and the rest in line).  Loop step i of
  // direction 0 is frame i, of direction 1 frame (T - i) % T (quirk Q1).  A product STORES the frames nobody has written yet and
  // ACCUMULATES into the others (the host keeps the book: no zero fill of dx, and the sums are the in-line schedule's).
  std::vector<char> touched;
  auto dx_steps = [&](int dd, int i0, int i1, hipStream_t st, const unsigned long long* amax_dz, const unsigned long long* amax_w) -> int {
    int runs[2][2], nruns = 0;       // frames of these loop steps, as maximal runs [f0, f1)
    if (dd == 0) { runs[0][0] = i0; runs[0][1] = i1; nruns = 1; }
    else {
      const int lo = std::max(i0, 1);                                    // loop steps lo .. i1-1 -> frames T-i1+1 .. T-lo
      if (i1 > lo) { runs[nruns][0] = T - i1 + 1; runs[nruns][1] = T - lo + 1; ++nruns; }
      if (i0 == 0) { runs[nruns][0] = 0; runs[nruns][1] = 1; ++nruns; }   // loop step 0 = frame 0
    }
    for (int r = 0; r < nruns; ++r) {
      int f = runs[r][0];
      while (f < runs[r][1]) {               // sub-runs of equal "written yet?" state
        int g = f;
        while (g < runs[r][1] && touched[g] == touched[f]) ++g;
        const astk_lstm_params& p0 = prm[dd * P.nl];
        const float* dz = P.ZG[dd][0];
        MatView A = dd == 0 ? mat(dz + (size_t)f * B * 4 * h, 4 * h) : mat_idx(dz, 4 * h, rows_inv + (size_t)f * B);
        if (dd == 1) A.idx_rows = (long)T * B;
        ASTK_TRY(gemm_launch(GEMM_NN, with_amax_b(with_amax_a(gemm_args((g - f) * B, P.in, 4 * h, A, mat(p0.Wu, P.in), dx + (size_t)f * B * P.in, P.in, nullptr,
                                                                        touched[f] ? GEMM_ACCUM : GEMM_STORE), amax_dz), amax_w), st));
        for (int q = f; q < g; ++q) touched[q] = 1;
        f = g;
      }
    }
    return 0;
  };
  unsigned long long* dz_amax[16] = {nullptr};
  unsigned dz_amax_gen = 0;
  if (persist) {
    PersistCellHost cells[16];
    memset(cells, 0, sizeof(cells));
    // (the recurrence kernel reads its weight fragments straight from the (4h, h) parameters: no transposed copies)
    for (int dd = 0; dd < P.nd; ++dd)
      for (int l = 0; l < P.nl; ++l) {
        PersistCellHost& c = cells[dd * P.nl + l];
        const bool top = l == P.nl - 1;
        c.Wl = prm[dd * P.nl + l].Wl;
        if (rs_path) {
          c.Wu = l > 0 ? prm[dd * P.nl + l].Wu : nullptr;
          c.PR = P.PR[dd][l];
          c.PD = l > 0 ? P.PD[dd][l] : nullptr;
          c.PD_up = top ? nullptr : P.PD[dd][l + 1];
        }
        c.db = gr[dd * P.nl + l].db;      // the recurrence kernel sums its dz columns itself
        c.db_part = deterministic_mode() ? P.DBP[dd][l] : nullptr;
        c.gates = P.ZG[dd][l];
        c.C = P.CC[dd][l];
        c.mask = masks ? masks + ((size_t)dd * P.nl + l) * T * bh : nullptr;
        c.d_enc = top ? d_enc + (size_t)dd * h : nullptr;
        c.dy_sb = (long)T * H; c.dy_st = H;
        c.d_hT = d_hT ? d_hT + ((size_t)dd * P.nl + l) * bh : nullptr;
        c.d_cT = d_cT ? d_cT + ((size_t)dd * P.nl + l) * bh : nullptr;
        c.reverse_pos = dd == 1;
        c.layer = l;
      }
    // the recurrence kernel leaves max |dz| of every cell for the batched products behind it (fp16x2 GEMM scales)
    gemm_amax_reserve(P.nd * P.nl, dz_amax, &dz_amax_gen, s);
    for (int i = 0; i < P.nd * P.nl; ++i) cells[i].amax = dz_amax[i];
    const int rows_b = lstm_persist_rows(B, h, P.nl, P.nd, d->side_stream != nullptr);
    if (!lstm_persist_hoisted(h)) bside = plan_side_bwd(d, rows_b, lstm_persist_layers_per_launch(B, h, P.nl, P.nd, rows_b), dx != nullptr && sr == s);
    if (bside.n > 0) {
      // The input gradient dx (T,B,in) = dz_0 W_u0 of both directions, chunk by chunk behind the recurrence (see k_wait_progress).  Loop step
      // i of direction 0 is frame i, of direction 1 frame (T - i) % T (quirk Q1): the backward recurrence passes loop steps T-1 .. 0, so
      // direction 0 delivers the high frames first and direction 1 the low ones.  A product STORES the frames nobody has written yet and
      // ACCUMULATES into the others (the host keeps the book: no zero fill of the 79 MB, and the sums are the in-line schedule's).
      hipStream_t sside = (hipStream_t)d->side_stream;
      unsigned* prog = P.zflags + (size_t)(SIDE_CHUNKS_MAX + 2) * 64;
      const unsigned wgs_cell = (unsigned)((h / 16) * (rows_b == 16 ? (B + 15) / 16 : rows_b == 33 ? 2 * ((B + 31) / 32) : (B + 31) / 32));      // arrivals per chunk: (virtual) workgroups of a cell
      for (int dd = 0; dd < P.nd; ++dd) { cells[dd * P.nl].prog = prog + dd * 64; cells[dd * P.nl].prog_cs = bside.cs; }
      hipLaunchKernelGGL(k_zero_words, dim3(1), dim3(64), 0, s, prog, 3, 64);      // the two counters and the wait kernels' abort word
      ASTK_LAUNCH_CHECK();
      ASTK_TRY(stream_order(s, sside));          // everything the products read besides dz (weights, index tables) and the zeroed counters
      touched.assign((size_t)T, 0);
      const AbortCtl wab = abort_ctl(prog + 2 * 64, PERSIST_ENC_BWD);
      GemmWgCap cap_scope(bside.cap);
      for (int k = 0; k < bside.n; ++k) {
        const int i1 = T - k * bside.cs, i0 = std::max(0, i1 - bside.cs);       // loop steps [i0, i1) are final when chunk k has arrived
        hipLaunchKernelGGL(k_wait_progress, dim3(1), dim3(64), 0, sside, prog, P.nd > 1 ? prog + 64 : nullptr, (unsigned)(k + 1) * wgs_cell, wab);
        ASTK_LAUNCH_CHECK();
        for (int dd = 0; dd < P.nd; ++dd) ASTK_TRY(dx_steps(dd, i0, i1, sside, nullptr, nullptr));
      }
    }
    ASTK_TRY(stream_order(s, sr));     // the recurrence kernel may live on its own (CU-masked) stream, see astk.h
    if (lstm_persist_hoisted(h)) {
      // hoisted form: layer by layer from the top; a lower layer's incoming gradient is the dense (T,B,h) product dz W_u of the layer
      // above, one batched product per direction between the launches (no partial tiles handed down)
      for (int l = P.nl - 1; l >= 0; --l) {
        PersistCellHost grp[16];
        for (int dd = 0; dd < P.nd; ++dd) {
          PersistCellHost& c = cells[dd * P.nl + l];
          c.PD = nullptr; c.PD_up = nullptr; c.up_external = 0;
          if (l < P.nl - 1) { c.d_enc = P.DX[dd]; c.dy_sb = h; c.dy_st = (long)B * h; c.reverse_pos = 0; }
          grp[dd] = c;
        }
        ASTK_TRY(lstm_persist_bwd_launch(grp, P.nd, 1, T, B, h, H, P.counters, dz_amax_gen, 16, sr));
        if (l > 0)
          for (int dd = 0; dd < P.nd; ++dd)
            ASTK_TRY(gemm_launch(GEMM_NN, with_amax_a(gemm_args(T * B, h, 4 * h, mat(P.ZG[dd][l], 4 * h), mat(prm[dd * P.nl + l].Wu, h), P.DX[dd], h), dz_amax[dd * P.nl + l]), sr));
      }
    } else {
      // groups of layers, top group first; the top layer of a lower group reads the partial dx tiles the previous launch left
      const int rows_wg = lstm_persist_rows(B, h, P.nl, P.nd, d->side_stream != nullptr);
      const int lpl = lstm_persist_layers_per_launch(B, h, P.nl, P.nd, rows_wg);
      int l1 = P.nl;
      while (l1 > 0) {
        // same grouping as the forward pass (groups start at multiples of lpl)
        const int l0 = ((l1 - 1) / lpl) * lpl;
        const int ngl = l1 - l0;
        PersistCellHost grp[16];
        for (int dd = 0; dd < P.nd; ++dd)
          for (int l = 0; l < ngl; ++l) {
            grp[dd * ngl + l] = cells[dd * P.nl + l0 + l];
            if (l == ngl - 1 && l1 < P.nl) grp[dd * ngl + l].up_external = 1;
          }
        ASTK_TRY(lstm_persist_bwd_launch(grp, P.nd * ngl, ngl, T, B, h, H, P.counters, dz_amax_gen, rows_wg, sr));
        l1 = l0;
      }
    }
    ASTK_TRY(stream_order(sr, s));
    if (deterministic_mode()) {
      FoldDbJobs j;
      j.n = P.nd * P.nl; j.cols = 4 * h;
      const int rows_d = lstm_persist_hoisted(h) ? 16 : rows_b;
      j.nby = rows_d == 16 ? (B + 15) / 16 : rows_d == 33 ? 2 * ((B + 31) / 32) : (B + 31) / 32;
      for (int i = 0; i < j.n; ++i) { j.db[i] = cells[i].db; j.part[i] = cells[i].db_part; }
      hipLaunchKernelGGL(k_fold_db, dim3(cdiv(4 * h, 256), j.n), dim3(256), 0, s, j);
      ASTK_LAUNCH_CHECK();
    }
  }
  GemmArgs wg[GEMM_GROUP_MAX];   // weight-gradient products, issued as grouped launches
  int nwg = 0;
  // Absolute maxima (fp16x2 GEMM scales) of the matrices that feed several products: the frames (B operand of both directions'
  // layer-0 dWu) by a pass here, every cell's dz (dWl, dWu, input gradient) by the recurrence kernel itself on the persistent path.
  // (the forward call of this workspace folded a strided x_amax into P.ax)
  const unsigned long long* ax = (d->x_amax && (((uintptr_t)d->x_amax) & 1u)) ? P.ax : (const unsigned long long*)d->x_amax;
  const unsigned long long* aw0[2] = {nullptr, nullptr};
  {
    AmaxMatrix am[3] = {{ax ? nullptr : x, (long)T * B, (long)P.in, P.in}, {dx ? prm[0].Wu : nullptr, 4L * h, (long)P.in, P.in},
                        {dx && P.nd > 1 ? prm[P.nl].Wu : nullptr, 4L * h, (long)P.in, P.in}};
    const unsigned long long* out[3];
    gemm_amax_many(am, 3, out, s);
    if (!ax) ax = out[0];
    aw0[0] = out[1]; aw0[1] = out[2];
  }
  // the layer outputs are bounded by construction (|h| < 1, times the dropout scale): no pass over them either
  const unsigned long long* ahb = d->out_bound > 0.f ? gemm_amax_bound(exp2f(ceilf(log2f(d->out_bound))), s) : nullptr;
  const unsigned long long* adz_all[16] = {nullptr};
  if (persist)
    for (int i = 0; i < P.nd * P.nl; ++i) adz_all[i] = dz_amax[i];
  ColsumBatch cb;                // bias gradients of all cells: one launch (dz of every cell is final when the recurrence kernel has run)
  for (int l = P.nl - 1; l >= 0; --l) {
    const bool top = l == P.nl - 1;
    const int in = l == 0 ? P.in : h;
    for (int dd = 0; dd < P.nd && !persist; ++dd)
      ASTK_TRY(transpose_f32(P.WlT[dd][l], 4 * h, prm[dd * P.nl + l].Wl, h, 4 * h, h, s));
    for (int i = T - 1; i >= 0 && !persist; --i) {
      LstmCellBwdArgs cells[2];
      for (int dd = 0; dd < P.nd; ++dd) {
        LstmCellBwdArgs& c = cells[dd];
        memset(&c, 0, sizeof(c));
        c.npairs = 1;
        const bool last = i == T - 1;
        c.p[0].A = last ? nullptr : P.ZG[dd][l] + (size_t)(i + 1) * B * 4 * h;   // dz of step i+1
        c.p[0].lda = 4 * h;
        c.p[0].W = P.WlT[dd][l];
        c.p[0].ldw = 4 * h;
        c.p[0].K = last ? 0 : 4 * h;
        c.B = B; c.h = h;
        c.dh_add = (last && d_hT) ? d_hT + ((size_t)dd * P.nl + l) * bh : nullptr;
        if (top) {
          const int pos = dd == 0 ? i : T - 1 - i;
          c.dy2 = d_enc + (size_t)pos * H + (size_t)dd * h;
          c.ld_dy2 = (long)T * H;
        } else {
          c.dy = P.DX[dd] + (size_t)i * bh;
          c.ld_dy = h;
        }
        c.mask = masks ? masks + (((size_t)dd * P.nl + l) * T + i) * bh : nullptr;
        c.dc_next = last ? (d_cT ? d_cT + ((size_t)dd * P.nl + l) * bh : nullptr) : P.DC[dd][(i + 1) & 1];
        c.c_prev = i > 0 ? P.CC[dd][l] + (size_t)(i - 1) * bh : nullptr;
        c.c_cur = P.CC[dd][l] + (size_t)i * bh;
        c.gates_dz = P.ZG[dd][l] + (size_t)i * B * 4 * h;
        c.ld_g = 4 * h;
        c.dc_prev = P.DC[dd][i & 1];
      }
      ASTK_TRY(lstm_cell_bwd_launch(cells, P.nd, s));
    }
    // ---- batched products over all time steps
    for (int dd = 0; dd < P.nd; ++dd) {
      const astk_lstm_params& p = prm[dd * P.nl + l];
      const astk_lstm_grads& g = gr[dd * P.nl + l];
      const float* dz = P.ZG[dd][l];
      const int rows = T * B;
      // dz feeds up to three products (dWl, dWu, the input gradient): one absolute-maximum pass for all of them
      const unsigned long long* adz = persist ? adz_all[dd * P.nl + l] : gemm_amax(dz, rows, 4 * h, 4 * h, s);
      // dWl (4h,h) += sum_{i>=1} dz_i^T h_{i-1}
      if (T > 1) {
        if (nwg == GEMM_GROUP_MAX) { ASTK_TRY(gemm_launch_group(GEMM_TN, wg, nwg, s)); nwg = 0; }
        wg[nwg++] = with_amax_b(with_amax_a(gemm_args(4 * h, h, rows - B, mat(dz + (size_t)B * 4 * h, 4 * h), mat(P.HR[dd][l], h), g.dWl, h, nullptr, GEMM_ATOMIC, 1), adz), ahb);
      }
      // dWu (4h,in) += dz^T X   (reverse stack, layer 0: dz is first re-ordered to frame order, sum_i dz_i^T x[perm i] = sum_f dz[inv f]^T x_f)
      {
        MatView Xv;
        const float* dzu = dz;
        if (l == 0) {
          Xv = mat(x, in);
          if (dd == 1) {
            hipLaunchKernelGGL(k_gather_rows, dim3(2048), dim3(256), 0, s, P.GATH, dz, rows_inv, rows, h);
            ASTK_LAUNCH_CHECK();
            dzu = P.GATH;
          }
        } else Xv = mat(masks ? P.HD[dd][l - 1] : P.HR[dd][l - 1], h);
        if (l == 0 && (dd == 1 || low_precision_gemms())) {   // GATH is a single scratch buffer: issue this product right away
          // (K9's weight gradient; in low-precision mode also direction 0's, which otherwise rides in the grouped launch)
          // (the gathered copy holds the same values as dz: same maximum)
          ASTK_TRY(gemm_launch(GEMM_TN, with_amax_b(with_amax_a(lowp(gemm_args(4 * h, in, rows, mat(dzu, 4 * h), Xv, g.dWu, in, nullptr, GEMM_ATOMIC, 1)), adz), l == 0 ? ax : nullptr), s));
        } else {
          if (nwg == GEMM_GROUP_MAX) { ASTK_TRY(gemm_launch_group(GEMM_TN, wg, nwg, s)); nwg = 0; }
          wg[nwg++] = with_amax_b(with_amax_a(gemm_args(4 * h, in, rows, mat(dzu, 4 * h), Xv, g.dWu, in, nullptr, GEMM_ATOMIC, 1), adz), l == 0 ? ax : ahb);
        }
      }
      if (!persist) ASTK_TRY(cb.add(g.db, dz, 4 * h, rows, 4 * h, s));
      // gradient wrt the layer input
      if (l > 0) {
        if (!persist) ASTK_TRY(gemm_launch(GEMM_NN, with_amax_a(gemm_args(rows, h, 4 * h, mat(dz, 4 * h), mat(p.Wu, h), P.DX[dd], h), adz), s));
      } else if (dx && bside.n == 0) {
        // dx (T,B,in) in frame order: direction 0 stores, direction 1 accumulates through the inverse permutation
        MatView A = dd == 0 ? mat(dz, 4 * h) : mat_idx(dz, 4 * h, rows_inv);
        ASTK_TRY(gemm_launch(GEMM_NN, with_amax_b(with_amax_a(lowp(gemm_args(rows, in, 4 * h, A, mat(p.Wu, in), dx, in, nullptr, dd == 0 ? GEMM_STORE : GEMM_ACCUM)), adz), aw0[dd]), s));
      } else if (dx) {
        // the loop steps the side stream did not take, at full width; the side chunks were sized to be done by now, and their frames are in `touched`
        if (dd == 0) ASTK_TRY(stream_order((hipStream_t)d->side_stream, s));
        const int i1 = T - bside.n * bside.cs;
        if (i1 > 0) ASTK_TRY(dx_steps(dd, 0, i1, s, adz, aw0[dd]));
      }
    }
  }
  ASTK_TRY(cb.flush(s));
  if (nwg > 0) ASTK_TRY(gemm_launch_group(GEMM_TN, wg, nwg, s));
  if (bside.n > 0) ASTK_TRY(stream_order((hipStream_t)d->side_stream, s));      // join: the caller sees one-stream semantics
  return 0;
}

}  // extern "C"
