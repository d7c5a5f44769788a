// Attention decoder loop (seq2seq.py:318-333, 361-473) and its hand-written backward.
//
// Per step s (all on one stream, graph-capturable; the teacher-forcing coin of quirk Q4 is a device flag so
// the launch sequence is static):
//   embed(+dropout) into the [emb ; ht_{s-1}] input-feeding buffer -> fused LSTM cells (row-panel MFMA +
//   gate epilogue) -> q = Wa h + ba -> attention scan (attn.hip) -> ht = tanh(Wc[cv;h] + bc) (written to HT and
//   into the next step's concat buffer) -> logits = Wo ht + bo -> fused softmax-CE / argmax / dlogits.
// Everything the backward needs is saved step-major in the workspace; weight gradients are NOT computed per
// step: the per-step data-path gradients (dz, d_pre, dq, ...) are saved and every dW is one batched TN GEMM
// over S*B rows after the loop; d_enc_states is one batched GEMM over the saved (alpha, ds).
#include "common.h"
#include <mutex>
#include "decoder_wide.h"

namespace astk {

// persistent decoder loop (decoder_persist.hip)
constexpr int PDEC_MAX_LAYERS = 3;
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
bool decoder_persist_applicable(const astk_decoder_desc* d, int* nsplit_out, int* chunk_out);
int decoder_persist_fwd_launch(const astk_decoder_desc* d, const astk_decoder_params* prm, const float* enc, const int32_t* y,
                               const int32_t* ytgt, const int32_t* use_truth, const float* emb_mask, const float* rnn_masks, const DecPersistBuffers& bf,
                               float* loss, int32_t* pred_out, hipStream_t s);

struct DecPersistBwdBuffers {
  void* zero_ptr; size_t zero_bytes;      // astk_decoder_desc.zero_ptr: zeroed by the launcher's fill launch
  void* zero2_ptr; size_t zero2_bytes;    // d_enc: zeroed there too, its two batched products then ADD into it from one grouped launch
  const float *WoT, *WcT, *ENCA, *CVH, *HT, *LOGITS, *ML;
  const float *WlT[PDEC_MAX_LAYERS], *WuT[PDEC_MAX_LAYERS], *C[PDEC_MAX_LAYERS];
  float *G[PDEC_MAX_LAYERS];
  float *ALPHA, *DPRE, *DCVH, *DS, *DX0, *DHATT, *d_c0;
  float* DXH;
  unsigned* ctr;
};
bool decoder_persist_b6_split(const astk_decoder_desc* d);
int decoder_persist_bwd_launch(const astk_decoder_desc* d, const float* enc, const float* rnn_masks, const DecPersistBwdBuffers& bf,
                               hipStream_t s);

namespace {

struct DecPlan {
  int B, L, S, T, Tp, H, E, A, V, Vp, XI, nl;
  int NA, CW;    // attention heads; width of the [cv_0; ..; cv_{NA-1}; h] buffer = (NA+1)*H
  bool feed, ln; // input feeding (ht_{s-1} behind the embedding); LayerNorm behind every LSTM's dropped output
  float* HDL[ASTK_MAX_RNN_LAYERS];   // ln: [S][B][H] dropped outputs BEFORE the LayerNorm (its saved input)
  float* DLN;    // ln: [B][H] gradient wrt a LayerNorm's output / input (scratch)
  float* DLN2;
  int* TOK;      // [S][B] token fed at step s
  int* PRED;     // [S][B] argmax of step s
  float* X0;     // [S][B][XI]  concat(emb, ht_prev)
  float* G[ASTK_MAX_RNN_LAYERS];    // [S][B][4H] gates -> dz
  float* C[ASTK_MAX_RNN_LAYERS];    // [(S+1)][B][H], C[0] = c0
  float* HR[ASTK_MAX_RNN_LAYERS];   // [(S+1)][B][H], HR[0] = h0
  float* HD[ASTK_MAX_RNN_LAYERS];   // [S][B][H] dropped outputs of layers < top (with masks)
  float* Q;      // [NA][S][B][H]
  float* ALPHA;  // [NA][S][B][Tp]
  float* CVH;    // [S][B][CW]  concat(cv_0, .., cv_{NA-1}, h_top_dropped)
  float* HT;     // [(S+1)][B][A], HT[0] = 0
  float* LOGITS; // [S][B][Vp] -> dlogits
  float* LOSSROWS;  // [S][B]
  // backward
  float* DPRE;   // [S][B][A]
  float* DCVH;   // [S][B][CW]
  float* DS;     // [NA][S][B][Tp]
  float* DQ;     // [NA][S][B][H]
  float* DX0;    // [S][B][XI]
  float* DHTOP;  // [B][H]
  float* DC[ASTK_MAX_RNN_LAYERS][2];
  float* WoT;    // [A][Vp]
  float* LG1;    // [B][Vp] logits of a step whose argmax is fed back (when every step is scored behind the loop)
  float* WPART;  // wide persistent forward loop (decoder_wide.hip): partial attention sums, counters
  unsigned* WCTR;
  float* WBWD;   // wide persistent backward loop: partial sums, counters
  unsigned* WBCTR;
  float* WcT;    // [CW][A]
  float* WaT;    // [NA][H][H]
  float* WuT[ASTK_MAX_RNN_LAYERS];  // [in][4H]
  float* WlT[ASTK_MAX_RNN_LAYERS];  // [H][4H]
  float *LSE, *PART, *CESTAT, *ENCA, *MLB, *DXH;       // persistent path only
  unsigned* PCTR;
  void* attn_ws;
  size_t bytes;
};

// Which kernel path a forward call took on a workspace (diagnostics only, never read by a kernel): the path and the workspace carve are
// re-derived from the shape AND the tuning knobs ("dec.persist" / "dec.wide") at every call, so a backward call made under another
// environment than its forward call would read the saved activations at shifted offsets without any error.  The forward records
// (workspace, path), the backward refuses a workspace whose record differs.
struct PathRecord { const void* ws; int path; };
static std::mutex g_path_mu;
static PathRecord g_path_ring[64];
static unsigned g_path_next = 0;
static void path_record(const void* ws, int path) {
  std::lock_guard<std::mutex> lock(g_path_mu);
  for (auto& r : g_path_ring)
    if (r.ws == ws) { r.path = path; return; }
  g_path_ring[g_path_next++ % 64] = PathRecord{ws, path};
}
static int path_lookup(const void* ws) {      // -1: no forward call recorded for this workspace
  std::lock_guard<std::mutex> lock(g_path_mu);
  for (auto& r : g_path_ring)
    if (r.ws == ws) return r.path;
  return -1;
}

int make_plan(const astk_decoder_desc* d, void* ws, DecPlan& P) {
  ASTK_CHECK_DESC(d, astk_decoder_desc);
  ASTK_CHECK(d && d->B > 0 && d->L >= 2 && d->T > 0 && d->V > 1, "decoder: bad dims");
  ASTK_CHECK(d->n_layers >= 1 && d->n_layers <= ASTK_MAX_RNN_LAYERS, "decoder: layers");
  ASTK_CHECK((d->H % 4) == 0 && (d->E % 4) == 0 && (d->A % 4) == 0, "decoder: H, E, A must be multiples of 4");
  P.B = d->B; P.L = d->L; P.S = d->L - 1; P.T = d->T; P.Tp = (d->T + 3) / 4 * 4;
  P.H = d->H; P.E = d->E; P.A = d->A; P.V = d->V; P.Vp = (d->V + 3) / 4 * 4; P.nl = d->n_layers;
  ASTK_CHECK(d->n_attn >= 0 && d->n_attn <= ASTK_MAX_ATTN, "decoder: n_attn %d (max %d)", d->n_attn, ASTK_MAX_ATTN);
  P.NA = d->n_attn > 1 ? d->n_attn : 1;
  P.CW = (P.NA + 1) * d->H;
  P.feed = d->no_feed_attn == 0;
  P.ln = d->ln != 0;
  P.XI = P.feed ? d->E + d->A : d->E;
  Carver c(ws);
  const size_t S = P.S, B = P.B, H = P.H, NA = P.NA, CW = P.CW;
  P.TOK = c.take<int>(S * B);
  P.PRED = c.take<int>(S * B);
  P.X0 = c.take<float>(S * B * P.XI);
  for (int l = 0; l < P.nl; ++l) {
    P.G[l] = c.take<float>(S * B * 4 * H);
    P.C[l] = c.take<float>((S + 1) * B * H);
    P.HR[l] = c.take<float>((S + 1) * B * H);
    P.HD[l] = c.take<float>(S * B * H);
    P.DC[l][0] = c.take<float>(B * H);
    P.DC[l][1] = c.take<float>(B * H);
    const size_t in = l == 0 ? P.XI : H;
    P.WuT[l] = c.take<float>(in * 4 * H);
    P.WlT[l] = c.take<float>(H * 4 * H);
    P.HDL[l] = c.take<float>(P.ln ? S * B * H : 4);
  }
  P.DLN = c.take<float>(B * H);
  P.DLN2 = c.take<float>(B * H);
  P.Q = c.take<float>(NA * S * B * H);
  P.ALPHA = c.take<float>(NA * S * B * P.Tp);
  P.CVH = c.take<float>(S * B * CW);
  P.HT = c.take<float>((S + 1) * B * P.A);
  P.LOGITS = c.take<float>(S * B * P.Vp);
  P.LOSSROWS = c.take<float>(S * B);
  P.DPRE = c.take<float>(S * B * P.A);
  P.DCVH = c.take<float>(S * B * CW);
  P.DS = c.take<float>(NA * S * B * P.Tp);
  P.DQ = c.take<float>(NA * S * B * H);
  P.DX0 = c.take<float>(S * B * P.XI);
  P.DHTOP = c.take<float>(B * H);
  P.WoT = c.take<float>((size_t)P.A * P.Vp);
  P.LG1 = c.take<float>((size_t)P.B * P.Vp);
  P.WPART = c.take<float>(decoder_wide_part_floats(d));
  P.WCTR = c.take<unsigned>(decoder_wide_ctr_words(d));
  P.WBWD = c.take<float>(decoder_wide_bwd_floats(d));
  P.WBCTR = c.take<unsigned>(decoder_wide_bwd_ctr_words(d));
  P.WcT = c.take<float>(CW * P.A);
  P.WaT = c.take<float>(NA * H * H);
  P.attn_ws = c.take<char>(attn_ws_bytes(P.B, P.T, P.H));
  {
    int ns = 1, ch = 1;
    const bool pp = decoder_persist_applicable(d, &ns, &ch);
    P.LSE = c.take<float>(pp ? S * B : 4);
    P.PART = c.take<float>(pp ? S * B * ns * (H + 4) : 4);
    P.CESTAT = c.take<float>(pp ? S * B * (size_t)((P.V + 15) / 16) * 4 : 4);
    P.ENCA = c.take<float>(pp ? B * (size_t)P.T * H : 4);
    P.MLB = c.take<float>(pp ? S * B * 2 : 4);
    P.PCTR = c.take<unsigned>(pp ? (size_t)(8 * 32 * ((P.B + 15) / 16) + 2 + P.B) * 64 : 4);   // sharded phase counters, abort word, per-row counters
    P.DXH = c.take<float>(pp ? 2 * S * B * (size_t)P.A : 4);
  }
  P.bytes = c.total();
  return 0;
}

// tok = use_truth[s] ? y[b][s] : pred_prev[b] ; x0[b][0:E] = embed[tok] * mask   (seq2seq.py:365, 431-436)
__global__ void k_embed(const float* __restrict__ embed, const int32_t* __restrict__ y, int L, int s, const int32_t* __restrict__ use_truth,
                        const int32_t* __restrict__ pred_prev, const int32_t* __restrict__ tokens_direct, int32_t* __restrict__ tok_out,
                        const float* __restrict__ mask, float* __restrict__ x0, int B, int E, int XI, int V) {
  const int b = blockIdx.x;
  int tok;
  if (tokens_direct) tok = tokens_direct[b];
  else tok = (use_truth[s] || !pred_prev) ? y[(long)b * L + s] : pred_prev[b];
  tok = tok < 0 ? 0 : (tok >= V ? V - 1 : tok);
  if (threadIdx.x == 0 && tok_out) tok_out[b] = tok;
  for (int e = threadIdx.x; e < E; e += blockDim.x) {
    float v = embed[(long)tok * E + e];
    if (mask) v *= mask[(long)b * E + e];
    x0[(long)b * XI + e] = v;
  }
}

// d_embed[tok][e] += dx0[s][b][e] * mask  for all (s,b)
__global__ void k_embed_bwd(float* __restrict__ d_embed, const int32_t* __restrict__ tok, const float* __restrict__ dx0,
                            const float* __restrict__ mask, int SB, int E, int XI, int V) {
  const int r = blockIdx.x;
  if (r >= SB) return;
  // (clamped: after a forward launch that timed out -- the step is discarded, but its backward still runs before the host reads the status
  // word -- the token buffer may hold whatever the workspace held; never an index outside the table)
  const int t = min(max(tok[r], 0), V - 1);
  for (int e = threadIdx.x; e < E; e += blockDim.x) {
    float v = dx0[(long)r * XI + e];
    if (mask) v *= mask[(long)r * E + e];
    atomicAdd(&d_embed[(long)t * E + e], v);
  }
}

// deterministic calls: one block per TOKEN adds the rows that embedded it, in row order (no atomics: the table row is the block's own)
__global__ void k_embed_bwd_det(float* __restrict__ d_embed, const int32_t* __restrict__ tok, const float* __restrict__ dx0,
                                const float* __restrict__ mask, int SB, int E, int XI, int V) {
  const int t = blockIdx.x;
  for (int e = threadIdx.x; e < E; e += blockDim.x) {
    float sum = 0.f;
    bool any = false;
    for (int r = 0; r < SB; ++r) {
      if (min(max(tok[r], 0), V - 1) != t) continue;
      float v = dx0[(long)r * XI + e];
      if (mask) v *= mask[(long)r * E + e];
      sum += v;
      any = true;
    }
    if (any) d_embed[(long)t * E + e] += sum;
  }
}

// One block per row: log-softmax, weighted NLL / count, first-max argmax, dlogits in place (Chainer-sem A6).
// Row r is (step r / rows_per_step, batch row r % rows_per_step); its class id is targets[(r % rows_per_step) * t_stride + r / rows_per_step]
// (one decoder step: rows_per_step = B and `targets` points at the step's column).  argmax_only: feedback tokens of a step whose loss is
// scored later (the batched pass); fed_flags (the loop's use_truth, n_steps entries): that batched pass leaves the argmax of the steps
// whose token was fed back (flag of the NEXT step 0) as the loop wrote it.
__global__ __launch_bounds__(256) void k_softmax_ce(int V, long ld, float* logits, const int32_t* __restrict__ targets,
                                                    long t_stride, int rows_per_step, const float* __restrict__ cw, float inv_count,
                                                    float* __restrict__ loss_rows, int32_t* __restrict__ argmax, int argmax_only,
                                                    const int32_t* __restrict__ fed_flags, int n_steps) {
  __shared__ float sv[4];
  __shared__ int si[4];
  const int b = blockIdx.x;
  const int step = b / rows_per_step, brow = b - step * rows_per_step;
  float* x = logits + (long)b * ld;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  float mx = -INFINITY;
  int mi = 0x7fffffff;
  for (int v = threadIdx.x; v < V; v += 256) {
    const float a = x[v];
    if (a > mx) { mx = a; mi = v; }
  }
  for (int o = 32; o > 0; o >>= 1) {
    const float om = __shfl_xor(mx, o);
    const int oi = __shfl_xor(mi, o);
    if (om > mx || (om == mx && oi < mi)) { mx = om; mi = oi; }
  }
  if (lane == 0) { sv[wave] = mx; si[wave] = mi; }
  __syncthreads();
  mx = sv[0]; mi = si[0];
  for (int w = 1; w < 4; ++w)
    if (sv[w] > mx || (sv[w] == mx && si[w] < mi)) { mx = sv[w]; mi = si[w]; }
  __syncthreads();
  if (argmax_only) {
    if (threadIdx.x == 0 && argmax) argmax[b] = mi;
    return;
  }
  float sum = 0.f;
  for (int v = threadIdx.x; v < V; v += 256) sum += expf(x[v] - mx);
  for (int o = 32; o > 0; o >>= 1) sum += __shfl_xor(sum, o);
  if (lane == 0) sv[wave] = sum;
  __syncthreads();
  sum = sv[0] + sv[1] + sv[2] + sv[3];
  const float lse = mx + logf(sum);
  int t = targets[(long)brow * t_stride + step];
  const bool ignore = t < 0;                      // ignore_label = -1 never occurs on this path (PAD is 0)
  t = t < 0 ? 0 : (t >= V ? V - 1 : t);
  const float w = ignore ? 0.f : (cw ? cw[t] : 1.f);
  // The target's logit has to be READ before any thread overwrites the row with the gradient below.  It used to be an ordinary load in
  // front of the barrier whose only use sat behind it; with `logits` declared __restrict__ the compiler was free to sink the load to
  // that use, i.e. behind the barrier, where thread (t % 256) may already have stored the gradient: one loss row in a few thousand
  // came out wrong by the difference, depending on which wave ran first (found at the end of round 3; the gradient was never affected).
  // Now: no __restrict__ on the row, a volatile load, and the loss row is finished in front of the barrier.
  const float xt = *reinterpret_cast<const volatile float*>(&x[t]);
  if (threadIdx.x == 0) {
    if (loss_rows) loss_rows[b] = -(xt - lse) * w * inv_count;
    const bool fed = fed_flags && step + 1 < n_steps && fed_flags[step + 1] == 0;
    if (argmax && !fed) argmax[b] = mi;
  }
  __syncthreads();
  const float scale = w * inv_count;
  for (int v = threadIdx.x; v < ld; v += 256) {
    float g = 0.f;
    if (v < V) {
      g = expf(x[v] - lse) * scale;
      if (v == t) g -= scale;
    }
    x[v] = g;
  }
}

__global__ void k_sum_to(const float* __restrict__ src, int n, float* __restrict__ dst) {
  __shared__ double red[4];
  double s = 0.0;
  for (int i = threadIdx.x; i < n; i += 256) s += (double)src[i];
  for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) *dst = (float)(red[0] + red[1] + red[2] + red[3]);
}

__global__ void k_copy_i32(int32_t* dst, const int32_t* src, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) dst[i] = src[i];
}

// x *= 1 - y^2 (the tanh' factor of d_pre on the rows that take no carry from a later step)
__global__ void k_dtanh_inplace(float* __restrict__ x, const float* __restrict__ y, long n) {
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    const float t = y[i];
    x[i] *= 1.f - t * t;
  }
}

RowGemmArgs rg(int M, int N, const float* A, long lda, const float* W, long ldw, int K, float* out, long ld_out) {
  RowGemmArgs a;
  memset(&a, 0, sizeof(a));
  a.npairs = 1;
  a.p[0].A = A; a.p[0].lda = lda; a.p[0].W = W; a.p[0].ldw = ldw; a.p[0].K = K;
  a.M = M; a.N = N; a.out = out; a.ld_out = ld_out;
  return a;
}

int ksplit_rows(long tiles, int rows) {
  long s = 256 / (tiles > 0 ? tiles : 1);
  if (s < 1) s = 1;
  long smax = rows / 128;
  if (smax < 1) smax = 1;
  return (int)(s > smax ? smax : s);
}

// dW (M x N) += A^T B over `rows` rows (A: rows x M, B: rows x N).  The weight gradients of one backward call are
// independent products with few tiles each: they are collected and issued as ONE grouped launch.
struct WgradBatch {
  GemmArgs list[GEMM_GROUP_MAX];
  int n = 0;
  bool low = false;      // every product of this batch may run with fp16 operands under astk_set_low_precision_gemms(1) (K18 / K24)
  explicit WgradBatch(bool lowp_eligible = false) : low(lowp_eligible) {}
  int add(float* dW, long ldw, int M, int N, const float* A, long lda, const float* Bm, long ldb, int rows, hipStream_t s) {
    if (n == GEMM_GROUP_MAX) ASTK_TRY(flush(s));
    list[n] = gemm_args(M, N, rows, mat(A, lda), mat(Bm, ldb), dW, ldw, nullptr, GEMM_ATOMIC, 1);
    if (low) list[n] = lowp(list[n]);
    ++n;
    return 0;
  }
  int flush(hipStream_t s) {
    const int m = n;
    n = 0;
    return m > 0 ? gemm_launch_group(GEMM_TN, list, m, s) : 0;
  }
};

int cell_fwd(const DecPlan& P, const astk_decoder_params* prm, int l, const float* x_in, long ld_x, int in, const float* h_prev,
             const float* c_prev, float* gates, float* c_out, float* h_out, const float* mask, float* hd_out, long ld_hd,
             hipStream_t s) {
  LstmCellFwdArgs c;
  memset(&c, 0, sizeof(c));
  c.npairs = 2;
  c.p[0].A = h_prev; c.p[0].lda = P.H; c.p[0].W = prm->lstm[l].Wl; c.p[0].ldw = P.H; c.p[0].K = P.H;
  c.p[1].A = x_in; c.p[1].lda = ld_x; c.p[1].W = prm->lstm[l].Wu; c.p[1].ldw = in; c.p[1].K = in;
  c.B = P.B; c.h = P.H;
  c.bias = prm->lstm[l].b;
  c.c_prev = c_prev;
  c.gates = gates; c.ld_g = 4 * P.H;
  c.c_out = c_out; c.h_out = h_out;
  c.mask = mask;
  c.hd_out = hd_out; c.ld_hd = ld_hd;
  return lstm_cell_fwd_launch(&c, 1, s);
}

// ---- row split.  The persistent loop holds at most 32 batch rows at the shipped width (its cell, context and logits items own 16 rows
// each and the grid is one workgroup per CU).  The decoder couples no batch rows -- weights and the 1/B of the loss are all two rows
// share -- so a larger batch runs as TWO persistent launches over halves of its rows, each a complete call of this library on a
// contiguous sub-problem (inputs that are not contiguous per half -- initial states, dropout masks -- are staged; `loss_rows` keeps the
// cross-entropy mean over the WHOLE batch; parameter gradients accumulate).  3.1 ms for batch 64 instead of the per-launch loop.
struct SplitPlan {
  bool on;
  astk_decoder_desc sub[2];
  int off[2];
  void* ws[2];
  size_t wsb[2];
  float *c0[2], *h0[2], *dc0[2], *dh0[2], *emb[2], *rnn[2];
  int32_t* pred[2];
  float* loss2;
  size_t bytes;
};
int make_split(const astk_decoder_desc* d, void* ws, SplitPlan& sp) {
  sp.on = false;
  sp.bytes = 0;
  if (!d || d->B < 2 || d->n_attn > 1 || d->no_feed_attn || d->ln || d->L < 2) return 0;
  int ns = 1, ch = 1;
  if (decoder_persist_applicable(d, &ns, &ch) || decoder_wide_applicable(d, nullptr, nullptr)) return 0;
  const int B0 = ((d->B / 2 + 15) / 16) * 16;
  if (B0 >= d->B) return 0;
  const int Bs[2] = {B0, d->B - B0};
  for (int i = 0; i < 2; ++i) {
    sp.sub[i] = *d;
    sp.sub[i].status_dst = nullptr;      // (the whole op's call takes the snapshot, once)
    sp.sub[i].zero_ptr = nullptr;        // (... and zeroes the caller's buffer, once: astk_decoder_bwd_phase_ex)
    sp.sub[i].zero_bytes = 0;
    sp.sub[i].B = Bs[i];
    sp.sub[i].loss_rows = d->loss_rows > 0 ? d->loss_rows : d->B;
    sp.off[i] = i == 0 ? 0 : B0;
    // (the wide decoder's loops only for more than 32 rows: at 32 rows and slices too long for LDS two half launches are no faster than the
    //  per-launch loop)
    if (!decoder_persist_applicable(&sp.sub[i], &ns, &ch) && !(d->B > 32 && decoder_wide_applicable(&sp.sub[i], nullptr, nullptr))) return 0;
  }
  Carver c(ws);
  const size_t S = d->L - 1, H = d->H, E = d->E, nl = d->n_layers;
  for (int i = 0; i < 2; ++i) {
    DecPlan P;
    ASTK_TRY(make_plan(&sp.sub[i], nullptr, P));
    sp.wsb[i] = P.bytes;
    sp.ws[i] = c.take<char>(P.bytes);
    const size_t b = Bs[i];
    sp.c0[i] = c.take<float>(nl * b * H); sp.h0[i] = c.take<float>(nl * b * H);
    sp.dc0[i] = c.take<float>(nl * b * H); sp.dh0[i] = c.take<float>(nl * b * H);
    sp.emb[i] = c.take<float>(S * b * E);
    sp.rnn[i] = c.take<float>(nl * S * b * H);
    sp.pred[i] = c.take<int32_t>(S * b);
  }
  sp.loss2 = c.take<float>(4);
  sp.bytes = c.total();
  sp.on = true;
  return 0;
}

}  // namespace

int softmax_ce_launch(int B, int V, long ld, float* logits, const int32_t* targets, long t_stride, const float* cw, float inv_count,
                      float* loss_rows, int32_t* argmax, hipStream_t s) {
  ASTK_CHECK(B > 0 && V > 0 && ld >= V && logits && targets, "softmax_ce: bad arguments");
  hipLaunchKernelGGL(k_softmax_ce, dim3(B), dim3(256), 0, s, V, ld, logits, targets, t_stride, B, cw, inv_count, loss_rows, argmax, 0,
                     (const int32_t*)nullptr, 0);
  ASTK_LAUNCH_CHECK();
  return 0;
}

}  // namespace astk

using namespace astk;

extern "C" {

int astk_decoder_path(const astk_decoder_desc* d) {
  int ns = 1, ch = 1;
  if (!d || d->struct_size != sizeof(astk_decoder_desc)) return 0;
  if (!decoder_persist_applicable(d, &ns, &ch)) {
    SplitPlan sp;
    if (make_split(d, nullptr, sp) != 0 || !sp.on) return decoder_wide_applicable(d, nullptr, nullptr) ? 16 : 0;   // 16: wide forward loop (decoder_wide.hip)
    return astk_decoder_path(&sp.sub[0]) | 4;                                // two persistent launches over halves of the rows
  }
  return 1 | ((d->H == 512 && ch <= 60) ? 2 : 0) | (d->n_layers << 8);      // (PDEC_CHUNK_MAX of decoder_persist.hip)
}

size_t astk_decoder_workspace_bytes(const astk_decoder_desc* d) {
  DecPlan P;
  if (make_plan(d, nullptr, P) != 0) return 0;
  SplitPlan sp;
  if (make_split(d, nullptr, sp) == 0 && sp.on && sp.bytes > P.bytes) return sp.bytes;     // (dropout.out keeps the per-launch layout: the larger of the two)
  return P.bytes;
}

int astk_softmax_ce_fwd(int B, int V, long ld, float* logits_inout, const int32_t* targets, long t_stride, const float* class_weight,
                        float inv_count, float* loss_rows, int32_t* argmax, void* stream) {
  return softmax_ce_launch(B, V, ld, logits_inout, targets, t_stride, class_weight, inv_count, loss_rows, argmax, (hipStream_t)stream);
}

int astk_decoder_fwd(const astk_decoder_desc* d, const astk_decoder_params* prm, const float* enc, const float* c0, const float* h0,
                     const int32_t* y, const int32_t* use_truth, const float* emb_mask, const float* rnn_masks, float* loss,
                     int32_t* pred, void* ws, size_t ws_bytes, void* stream) {
  return astk_decoder_fwd_ex(d, prm, enc, c0, h0, y, use_truth, emb_mask, rnn_masks, nullptr, nullptr, loss, pred, ws, ws_bytes, stream);
}

static int decoder_fwd_impl(const astk_decoder_desc* d, const astk_decoder_params* prm, const float* enc, const float* c0, const float* h0,
                            const int32_t* y, const int32_t* use_truth, const float* emb_mask, const float* rnn_masks, const float* out_mask,
                            const int32_t* targets, float* loss, int32_t* pred, void* ws, size_t ws_bytes, void* stream, bool* snapshot_taken);
int astk_decoder_fwd_ex(const astk_decoder_desc* d, const astk_decoder_params* prm, const float* enc, const float* c0, const float* h0,
                        const int32_t* y, const int32_t* use_truth, const float* emb_mask, const float* rnn_masks, const float* out_mask,
                        const int32_t* targets, float* loss, int32_t* pred, void* ws, size_t ws_bytes, void* stream) {
  bool taken = false;
  ASTK_TRY(decoder_fwd_impl(d, prm, enc, c0, h0, y, use_truth, emb_mask, rnn_masks, out_mask, targets, loss, pred, ws, ws_bytes, stream, &taken));
  // status_dst: the persistent loop's scoring kernel wrote it; every other path takes the snapshot with a launch behind the op
  if (d->status_dst && !taken) ASTK_TRY(status_snapshot_launch(d->status_dst, (hipStream_t)stream));
  return 0;
}
static int decoder_fwd_impl(const astk_decoder_desc* d, const astk_decoder_params* prm, const float* enc, const float* c0, const float* h0,
                            const int32_t* y, const int32_t* use_truth, const float* emb_mask, const float* rnn_masks, const float* out_mask,
                            const int32_t* targets, float* loss, int32_t* pred, void* ws, size_t ws_bytes, void* stream, bool* snapshot_taken) {
  hipStream_t s = (hipStream_t)stream;
  ASTK_CHECK_DESC(d, astk_decoder_desc);
  PrecScope prec_scope(d->precision, d->gemm_operands);
  GemmForwardScope forward_scope;      // split tiles of this op's products have at most two contributors (reproducible forward pass)
  {
    SplitPlan sp;
    ASTK_TRY(make_split(d, ws, sp));
    if (sp.on && !out_mask) {
      ASTK_CHECK(ws && ws_bytes >= sp.bytes, "decoder_fwd: workspace too small (%zu < %zu)", ws_bytes, sp.bytes);
      ASTK_CHECK(prm && enc && c0 && h0 && y && use_truth && loss, "decoder_fwd: null pointer");
      const int B = d->B, S = d->L - 1, H = d->H, E = d->E, nl = d->n_layers;
      for (int i = 0; i < 2; ++i) {
        const int b = sp.sub[i].B, off = sp.off[i];
        // initial states (n_layers, B, H) and masks (.., S, B, X): rows of this half, staged contiguously
        ASTK_TRY(copy2d_f32(sp.c0[i], (long)b * H, c0 + (size_t)off * H, (long)B * H, nl, b * H, b * H, s));
        ASTK_TRY(copy2d_f32(sp.h0[i], (long)b * H, h0 + (size_t)off * H, (long)B * H, nl, b * H, b * H, s));
        if (emb_mask) ASTK_TRY(copy2d_f32(sp.emb[i], (long)b * E, emb_mask + (size_t)off * E, (long)B * E, S, b * E, b * E, s));
        if (rnn_masks) ASTK_TRY(copy2d_f32(sp.rnn[i], (long)b * H, rnn_masks + (size_t)off * H, (long)B * H, nl * S, b * H, b * H, s));
        ASTK_TRY(astk_decoder_fwd_ex(&sp.sub[i], prm, enc + (size_t)off * d->T * H, sp.c0[i], sp.h0[i], y + (size_t)off * d->L, use_truth,
                                     emb_mask ? sp.emb[i] : nullptr, rnn_masks ? sp.rnn[i] : nullptr, nullptr,
                                     targets ? targets + (size_t)off * d->L : nullptr, sp.loss2 + i, pred ? sp.pred[i] : nullptr, sp.ws[i], sp.wsb[i], stream));
        if (pred) ASTK_TRY(copy2d_f32((float*)pred + off, B, (const float*)sp.pred[i], b, S, b, b, s));      // (S, b) -> columns of (S, B); a bit copy
      }
      hipLaunchKernelGGL(k_sum_to, dim3(1), dim3(256), 0, s, sp.loss2, 2, loss);
      ASTK_LAUNCH_CHECK();
      return 0;
    }
  }
  DecPlan P;
  ASTK_TRY(make_plan(d, ws, P));
  ASTK_CHECK(ws && ws_bytes >= P.bytes, "decoder_fwd: workspace too small (%zu < %zu)", ws_bytes, P.bytes);
  ASTK_CHECK(prm && enc && c0 && h0 && y && use_truth && loss, "decoder_fwd: null pointer");
  const int B = P.B, S = P.S, H = P.H, E = P.E, A = P.A, V = P.V, XI = P.XI, nl = P.nl, NA = P.NA, CW = P.CW;
  const size_t bh = (size_t)B * H;
  const int32_t* tgt = targets ? targets : y;       // class ids scored at step s: column s + 1
  // initial states and zero attention vector (seq2seq.py:318-333, :420)
  int ns_ = 1, ch_ = 1;
  const bool persist_path = !out_mask && decoder_persist_applicable(d, &ns_, &ch_);
  if (!persist_path) {      // (the persistent launcher copies / zeroes them with its own fill launch: two launches less)
    FillSegs fz;      // one launch: the two zero fills and the state copies
    fz.n = 0;
    fill_seg_add(fz, P.HT, (size_t)B * A * sizeof(float));
    fill_seg_add(fz, P.X0, (size_t)B * XI * sizeof(float));   // ht_{-1} half of the first concat buffer
    for (int l = 0; l < nl; ++l) {
      if (fz.n + 2 > FILL_SEG_MAX) { ASTK_TRY(fill_u32_segments(fz, 0u, s)); fz.n = 0; }
      fill_seg_add_copy(fz, P.C[l], c0 + l * bh, bh * sizeof(float));
      fill_seg_add_copy(fz, P.HR[l], h0 + l * bh, bh * sizeof(float));
    }
    ASTK_TRY(fill_u32_segments(fz, 0u, s));
  }
  const int top = nl - 1;
  {
    int ns = 1, ch = 1;
    // (dropout on the logits is not part of the persistent loop's CE role: per-launch loop)
    const bool persist = !out_mask && decoder_persist_applicable(d, &ns, &ch);
    path_record(ws, persist ? 1 : ((!out_mask && decoder_wide_applicable(d, nullptr, nullptr)) ? 2 : 0));
    if (!persist) ASTK_TRY(attn_ws_init(P.attn_ws, B, P.T, H, s));   // the persistent loop has its own counters
    if (persist) {
      DecPersistBuffers bf;
      memset(&bf, 0, sizeof(bf));
      bf.TOK = P.TOK; bf.PRED = P.PRED; bf.X0 = P.X0; bf.Q = P.Q; bf.ALPHA = P.ALPHA;
      for (int l = 0; l < nl; ++l) { bf.G[l] = P.G[l]; bf.C[l] = P.C[l]; bf.HR[l] = P.HR[l]; bf.HD[l] = P.HD[l]; }
      bf.CVH = P.CVH; bf.HT = P.HT; bf.LOGITS = P.LOGITS; bf.LOSSROWS = P.LOSSROWS; bf.LSE = P.LSE; bf.PART = P.PART;
      bf.CESTAT = P.CESTAT; bf.ENCA = P.ENCA; bf.ML = P.MLB; bf.ctr = P.PCTR;
      bf.zero_a = P.HT; bf.zero_a_bytes = (size_t)B * A * sizeof(float);
      bf.zero_b = P.X0; bf.zero_b_bytes = (size_t)B * XI * sizeof(float);
      bf.c0 = c0; bf.h0 = h0;
      ASTK_TRY(decoder_persist_fwd_launch(d, prm, enc, y, tgt, use_truth, emb_mask, rnn_masks, bf, loss, pred, s));   // (incl. loss sum and predictions)
      *snapshot_taken = d->status_dst != nullptr;
      return 0;
    }
  }
  const int32_t* uth = d->use_truth_host;      // optional host copy of use_truth: which steps feed their argmax back
  const float inv_count = 1.f / (float)(d->loss_rows > 0 ? d->loss_rows : B);
  // The wide decoder (configs[4]: H = A = 1024) on its persistent forward loop (decoder_wide.hip): ONE launch for all steps.  The kernel
  // computes logits itself on the steps whose argmax is fed back (streamed weights) and leaves the class in PRED; every step is scored
  // behind the loop like in the per-launch form with host flags.
  const bool wide = !out_mask && decoder_wide_applicable(d, nullptr, nullptr);
  if (wide) {
    DecWideBuffers wb;
    wb.TOK = P.TOK; wb.PRED = P.PRED; wb.X0 = P.X0; wb.G = P.G[0]; wb.C = P.C[0]; wb.HR = P.HR[0]; wb.Q = P.Q; wb.ALPHA = P.ALPHA;
    wb.CVH = P.CVH; wb.HT = P.HT; wb.PART = P.WPART; wb.ctr = P.WCTR;
    ASTK_TRY(decoder_wide_fwd_launch(d, prm, enc, y, use_truth, emb_mask, rnn_masks, wb, 0, S - 1, s));
  }
  for (int st = 0; st < S && !wide; ++st) {
    float* x0 = P.X0 + (size_t)st * B * XI;
    hipLaunchKernelGGL(k_embed, dim3(B), dim3(128), 0, s, prm->embed, y, P.L, st, use_truth, st > 0 ? P.PRED + (size_t)(st - 1) * B : nullptr,
                       (const int32_t*)nullptr, P.TOK + (size_t)st * B, emb_mask ? emb_mask + (size_t)st * B * E : nullptr, x0, B, E, XI, V);
    ASTK_LAUNCH_CHECK();
    float* cvh = P.CVH + (size_t)st * B * CW;
    float* htop = cvh + (size_t)NA * H;        // the top layer's (dropped, normalised) output sits behind the NA context vectors
    const float* x_in = x0;
    long ld_x = XI;
    int in = XI;
    for (int l = 0; l < nl; ++l) {
      const float* mask = rnn_masks ? rnn_masks + ((size_t)l * S + st) * bh : nullptr;
      float* hd;
      long ld_hd;
      if (l == top) { hd = htop; ld_hd = CW; }
      else { hd = P.HD[l] + (size_t)st * bh; ld_hd = H; }
      if (P.ln) {
        // hs = LN(dropout(LSTM(x))) (seq2seq.py:198-202): the cell leaves the dropped output in HDL, the LayerNorm writes the layer's output
        float* pre = P.HDL[l] + (size_t)st * bh;
        ASTK_TRY(cell_fwd(P, prm, l, x_in, ld_x, in, P.HR[l] + (size_t)st * bh, P.C[l] + (size_t)st * bh, P.G[l] + (size_t)st * B * 4 * H,
                          P.C[l] + (size_t)(st + 1) * bh, P.HR[l] + (size_t)(st + 1) * bh, mask, pre, H, s));
        ASTK_CHECK(prm->ln_gamma[l] && prm->ln_beta[l], "decoder_fwd: ln parameters missing (layer %d)", l);
        ASTK_TRY(layernorm_fwd_launch(B, H, pre, H, prm->ln_gamma[l], prm->ln_beta[l], LN_EPS, hd, ld_hd, s));
      } else {
        ASTK_TRY(cell_fwd(P, prm, l, x_in, ld_x, in, P.HR[l] + (size_t)st * bh, P.C[l] + (size_t)st * bh, P.G[l] + (size_t)st * B * 4 * H,
                          P.C[l] + (size_t)(st + 1) * bh, P.HR[l] + (size_t)(st + 1) * bh, mask, hd, ld_hd, s));
      }
      x_in = hd; ld_x = ld_hd; in = H;
    }
    // every attention head on the same h (seq2seq.py:379-383): q_k = Wa_k h + ba_k, scan -> cv_k
    for (int k = 0; k < NA; ++k) {
      const float* Wa = k == 0 ? prm->Wa : prm->Wa_x[k - 1];
      const float* ba = k == 0 ? prm->ba : prm->ba_x[k - 1];
      ASTK_CHECK(Wa && ba, "decoder_fwd: attention head %d has no parameters", k);
      float* q = P.Q + ((size_t)k * S + st) * bh;
      RowGemmArgs a = rg(B, H, htop, CW, Wa, H, H, q, H);
      a.bias = ba;
      ASTK_TRY(rowgemm_launch(a, s));
      ASTK_TRY(attn_fwd_launch(B, P.T, H, enc, q, H, P.ALPHA + ((size_t)k * S + st) * B * P.Tp, cvh + (size_t)k * H, CW, nullptr, 0, P.attn_ws, s));
    }
    // ht = tanh(Wc [cv..;h] + bc) -> HT[st+1] and (input feeding) the next step's concat buffer
    float* ht = P.HT + (size_t)(st + 1) * B * A;
    {
      RowGemmArgs a = rg(B, A, cvh, CW, prm->Wc, CW, CW, ht, A);
      a.bias = prm->bc;
      a.act = ACT_TANH;
      if (P.feed && st + 1 < S) { a.out2 = P.X0 + (size_t)(st + 1) * B * XI + E; a.ld_out2 = XI; }
      ASTK_TRY(rowgemm_launch(a, s));
    }
    // logits, dropout on the logits (seq2seq.py:394: argmax feedback and loss see the dropped logits; the gradient passes the same mask),
    // softmax cross-entropy.  With the caller's host copy of the flags only the steps whose argmax is FED BACK compute their logits
    // inside the loop (into a scratch panel, argmax only); every step is scored by one product and one launch behind the loop.
    const float* om = out_mask ? out_mask + (size_t)st * B * V : nullptr;
    const bool fed_back = st + 1 < S && (!uth || uth[st + 1] == 0);
    if (!uth || fed_back) {
      float* lg = uth ? P.LG1 : P.LOGITS + (size_t)st * B * P.Vp;
      RowGemmArgs a = rg(B, V, ht, A, prm->Wo, A, A, lg, P.Vp);
      a.bias = prm->bo;
      ASTK_TRY(rowgemm_launch(a, s));
      if (om) ASTK_TRY(mul_rows_launch(lg, P.Vp, om, V, B, V, s));
      if (uth) {
        hipLaunchKernelGGL(k_softmax_ce, dim3(B), dim3(256), 0, s, V, (long)P.Vp, lg, tgt + st + 1, (long)P.L, B, (const float*)nullptr, 1.f,
                           (float*)nullptr, P.PRED + (size_t)st * B, 1, (const int32_t*)nullptr, 0);
        ASTK_LAUNCH_CHECK();
      } else {
        ASTK_TRY(softmax_ce_launch(B, V, P.Vp, lg, tgt + st + 1, P.L, prm->class_weight, inv_count, P.LOSSROWS + (size_t)st * B, P.PRED + (size_t)st * B, s));
        if (om) ASTK_TRY(mul_rows_launch(lg, P.Vp, om, V, B, V, s));
      }
    }
  }
  if (uth || wide) {
    ASTK_TRY(gemm_launch(GEMM_NT, gemm_args(S * B, V, A, mat(P.HT + (size_t)B * A, A), mat(prm->Wo, A), P.LOGITS, P.Vp, prm->bo), s));
    if (out_mask) ASTK_TRY(mul_rows_launch(P.LOGITS, P.Vp, out_mask, V, S * B, V, s));
    hipLaunchKernelGGL(k_softmax_ce, dim3(S * B), dim3(256), 0, s, V, (long)P.Vp, P.LOGITS, tgt + 1, (long)P.L, B, prm->class_weight, inv_count,
                       P.LOSSROWS, P.PRED, 0, use_truth, S);
    ASTK_LAUNCH_CHECK();
    if (out_mask) ASTK_TRY(mul_rows_launch(P.LOGITS, P.Vp, out_mask, V, S * B, V, s));
  }
  hipLaunchKernelGGL(k_sum_to, dim3(1), dim3(256), 0, s, P.LOSSROWS, S * B, loss);
  ASTK_LAUNCH_CHECK();
  if (pred) {
    hipLaunchKernelGGL(k_copy_i32, dim3(cdiv(S * B, 256)), dim3(256), 0, s, pred, P.PRED, S * B);
    ASTK_LAUNCH_CHECK();
  }
  return 0;
}

int astk_decoder_bwd(const astk_decoder_desc* d, const astk_decoder_params* prm, const astk_decoder_grads* g, const float* enc,
                     const float* c0, const float* h0, const int32_t* y, const float* emb_mask, const float* rnn_masks, float* d_enc,
                     float* d_c0, float* d_h0, void* ws, size_t ws_bytes, void* stream) {
  return astk_decoder_bwd_phase_ex(d, prm, g, enc, c0, h0, y, emb_mask, rnn_masks, nullptr, d_enc, d_c0, d_h0, ws, ws_bytes, ASTK_DEC_BWD_ALL, stream);
}

int astk_decoder_bwd_phase(const astk_decoder_desc* d, const astk_decoder_params* prm, const astk_decoder_grads* g, const float* enc,
                           const float* c0, const float* h0, const int32_t* y, const float* emb_mask, const float* rnn_masks,
                           float* d_enc, float* d_c0, float* d_h0, void* ws, size_t ws_bytes, int phase, void* stream) {
  return astk_decoder_bwd_phase_ex(d, prm, g, enc, c0, h0, y, emb_mask, rnn_masks, nullptr, d_enc, d_c0, d_h0, ws, ws_bytes, phase, stream);
}

int astk_decoder_bwd_phase_ex(const astk_decoder_desc* d, const astk_decoder_params* prm, const astk_decoder_grads* g, const float* enc,
                              const float* c0, const float* h0, const int32_t* y, const float* emb_mask, const float* rnn_masks,
                              const float* out_mask, float* d_enc, float* d_c0, float* d_h0, void* ws, size_t ws_bytes, int phase,
                              void* stream) {
  hipStream_t s = (hipStream_t)stream;
  (void)c0; (void)h0; (void)y;
  ASTK_CHECK_DESC(d, astk_decoder_desc);
  PrecScope prec_scope(d->precision, d->gemm_operands);
  DetScope det_scope(d->deterministic);
  ASTK_CHECK(phase == ASTK_DEC_BWD_ALL || phase == ASTK_DEC_BWD_CHAIN || phase == ASTK_DEC_BWD_PARAMS, "decoder_bwd: bad phase %d", phase);
  {
    SplitPlan sp;
    ASTK_TRY(make_split(d, ws, sp));
    if (sp.on && !out_mask) {       // the two halves the forward call ran (their masks are still staged in the workspace)
      ASTK_CHECK(ws && ws_bytes >= sp.bytes, "decoder_bwd: workspace too small");
      ASTK_CHECK(prm && g && enc && d_enc && d_c0 && d_h0, "decoder_bwd: null pointer");
      if (phase != ASTK_DEC_BWD_PARAMS && d->zero_ptr && d->zero_bytes) ASTK_TRY(fill_zero(d->zero_ptr, d->zero_bytes, s));   // (once, in front of both halves)
      const int B = d->B, H = d->H, nl = d->n_layers;
      for (int i = 0; i < 2; ++i) {
        const int b = sp.sub[i].B, off = sp.off[i];
        ASTK_TRY(astk_decoder_bwd_phase_ex(&sp.sub[i], prm, g, enc + (size_t)off * d->T * H, sp.c0[i], sp.h0[i], nullptr, emb_mask ? sp.emb[i] : nullptr,
                                           rnn_masks ? sp.rnn[i] : nullptr, nullptr, d_enc + (size_t)off * d->T * H, sp.dc0[i], sp.dh0[i], sp.ws[i], sp.wsb[i],
                                           phase, stream));
        if (phase != ASTK_DEC_BWD_PARAMS) {
          ASTK_TRY(copy2d_f32(d_c0 + (size_t)off * H, (long)B * H, sp.dc0[i], (long)b * H, nl, b * H, b * H, s));
          ASTK_TRY(copy2d_f32(d_h0 + (size_t)off * H, (long)B * H, sp.dh0[i], (long)b * H, nl, b * H, b * H, s));
        }
      }
      return 0;
    }
  }
  const bool do_chain = phase != ASTK_DEC_BWD_PARAMS, do_params = phase != ASTK_DEC_BWD_CHAIN;
  DecPlan P;
  ASTK_TRY(make_plan(d, ws, P));
  ASTK_CHECK(ws && ws_bytes >= P.bytes, "decoder_bwd: workspace too small");
  ASTK_CHECK(prm && g && enc && d_enc && d_c0 && d_h0, "decoder_bwd: null pointer");
  const int B = P.B, S = P.S, H = P.H, E = P.E, A = P.A, V = P.V, Vp = P.Vp, XI = P.XI, nl = P.nl, T = P.T, Tp = P.Tp, NA = P.NA, CW = P.CW;
  const size_t bh = (size_t)B * H;
  const int top = nl - 1;
  int ns_ = 1, ch_ = 1;
  const bool persist = !out_mask && decoder_persist_applicable(d, &ns_, &ch_);      // (the forward pass took the same decision)
  const bool b6s = persist && decoder_persist_b6_split(d);
  const bool wide_b = !persist && !out_mask && decoder_wide_applicable(d, nullptr, nullptr);     // the whole reversed loop in one launch (decoder_wide.hip)
  {
    const int fwd_path = path_lookup(ws), bwd_path = persist ? 1 : (wide_b ? 2 : 0);
    ASTK_CHECK(fwd_path < 0 || fwd_path == bwd_path, "decoder_bwd: the forward call on this workspace took kernel path %d, this call would take %d "
               "(the dec.persist / dec.wide knobs changed between the two calls?)", fwd_path, bwd_path);
  }
  if (do_chain) {
  // astk_decoder_desc.zero_ptr (the gradient arena): in front of everything this call accumulates -- the persistent launcher's fill takes
  // it along (nothing in front of that launch touches the gradients), every other path fills here
  if (d->zero_ptr && d->zero_bytes) {
    ASTK_CHECK(aligned16(d->zero_ptr) && (d->zero_bytes % 4) == 0, "decoder_bwd: zero_ptr must be 16-byte aligned, zero_bytes a multiple of 4");
    if (!persist) ASTK_TRY(fill_zero(d->zero_ptr, d->zero_bytes, s));
  }
  // transposed weights for the data-path products (dY W as row-panel NT products)
  {
    TransposeJobs tj;
    tj.n = 0;
    if (persist) transpose_add(tj, P.WoT, Vp, prm->Wo, A, V, A);          // (V,A) -> (A,Vp); the per-launch loop batches dlogits Wo over the steps
    transpose_add(tj, P.WcT, A, prm->Wc, CW, A, CW);         // (A,CW) -> (CW,A)
    for (int k = 0; k < NA; ++k) transpose_add(tj, P.WaT + (size_t)k * H * H, H, k == 0 ? prm->Wa : prm->Wa_x[k - 1], H, H, H);
    for (int l = 0; l < nl; ++l) {
      const int in = l == 0 ? XI : H;
      if (tj.n + 2 > FILL_SEG_MAX) { ASTK_TRY(transpose_batch(tj, s)); tj.n = 0; }
      transpose_add(tj, P.WuT[l], 4 * H, prm->lstm[l].Wu, in, 4 * H, in);   // (4H,in) -> (in,4H)
      transpose_add(tj, P.WlT[l], 4 * H, prm->lstm[l].Wl, H, 4 * H, H);     // (4H,H)  -> (H,4H)
    }
    ASTK_TRY(transpose_batch(tj, s));
  }
  if (!persist) ASTK_TRY(attn_ws_init(P.attn_ws, B, T, H, s));   // the persistent loop has its own counters
  if (persist) {
    DecPersistBwdBuffers bf;
    memset(&bf, 0, sizeof(bf));
    bf.WoT = P.WoT; bf.WcT = P.WcT; bf.ENCA = P.ENCA; bf.ALPHA = P.ALPHA; bf.CVH = P.CVH; bf.ML = P.MLB;
    for (int l = 0; l < nl; ++l) { bf.WlT[l] = P.WlT[l]; bf.WuT[l] = P.WuT[l]; bf.C[l] = P.C[l]; bf.G[l] = P.G[l]; }
    bf.HT = P.HT; bf.LOGITS = P.LOGITS; bf.DPRE = P.DPRE; bf.DCVH = P.DCVH; bf.DS = P.DS; bf.DX0 = P.DX0;
    bf.DHATT = P.PART; bf.d_c0 = d_c0; bf.ctr = P.PCTR;
    bf.DXH = b6s ? P.DXH : nullptr;
    bf.zero_ptr = d->zero_ptr; bf.zero_bytes = d->zero_bytes;
    bf.zero2_ptr = d_enc; bf.zero2_bytes = (size_t)B * T * H * sizeof(float);
    ASTK_TRY(decoder_persist_bwd_launch(d, enc, rnn_masks, bf, s));
  }
  if (!persist) {
    // d_pre = (dlogits Wo + d_ht carried from step st+1 through input feeding) * (1 - ht^2).  dlogits Wo does not depend on the recurrence:
    // ONE product over all S*B rows (K = V; as a per-step row panel it is 2*ceil(B/16)*A/16 workgroups walking K = V each), then tanh' on
    // the rows that take no carry -- the last step, or every step without input feeding; the others get carry and tanh' from the epilogue
    // of step st+1's d_x0 product (RowGemmArgs::carry).
    ASTK_TRY(gemm_launch(GEMM_NN, gemm_args(S * B, A, V, mat(P.LOGITS, Vp), mat(prm->Wo, A), P.DPRE, A), s));
    const long r0 = P.feed ? (long)(S - 1) * B * A : 0, n = (long)S * B * A - r0;
    hipLaunchKernelGGL(k_dtanh_inplace, dim3((unsigned)std::min<long>(cdiv(n, 256), 2048)), dim3(256), 0, s, P.DPRE + r0, P.HT + (size_t)B * A + r0, n);
    ASTK_LAUNCH_CHECK();
  }
  if (wide_b) {
    DecWideBwdBuffers wb;
    wb.WcT = P.WcT; wb.WaT = P.WaT; wb.WlT = P.WlT[0]; wb.WuT = P.WuT[0]; wb.ALPHA = P.ALPHA; wb.CVH = P.CVH; wb.HT = P.HT; wb.C = P.C[0];
    wb.G = P.G[0]; wb.DPRE = P.DPRE; wb.DCVH = P.DCVH; wb.DS = P.DS; wb.DQ = P.DQ; wb.DHTOP = P.DHTOP; wb.DC0 = P.DC[0][0];
    wb.scratch = P.WBWD; wb.ctr = P.WBCTR;
    ASTK_TRY(decoder_wide_bwd_launch(d, enc, rnn_masks, wb, s));
  }
  for (int st = S - 1; st >= 0 && !persist && !wide_b; --st) {
    const bool last = st == S - 1;
    float* dpre = P.DPRE + (size_t)st * B * A;
    float* dcvh = P.DCVH + (size_t)st * B * CW;
    ASTK_TRY(rowgemm_launch(rg(B, CW, dpre, A, P.WcT, A, A, dcvh, CW), s));
    float* cvh = P.CVH + (size_t)st * B * CW;
    for (int k = 0; k < NA; ++k)
      ASTK_TRY(attn_bwd_launch(B, T, H, enc, P.ALPHA + ((size_t)k * S + st) * B * Tp, cvh + (size_t)k * H, CW, dcvh + (size_t)k * H, CW,
                               P.DS + ((size_t)k * S + st) * B * Tp, P.DQ + ((size_t)k * S + st) * bh, P.attn_ws, s));
    // gradient wrt the top layer's output: dh_top = dcvh[:, NA*H:] + sum_k dq_k Wa_k   (two heads per row-panel launch)
    {
      const float* add = dcvh + (size_t)NA * H;
      long ld_add = CW;
      float* outb = P.DHTOP;
      for (int k = 0; k < NA; k += 2) {
        RowGemmArgs a = rg(B, H, P.DQ + ((size_t)k * S + st) * bh, H, P.WaT + (size_t)k * H * H, H, H, outb, H);
        if (k + 1 < NA) {
          a.npairs = 2;
          a.p[1].A = P.DQ + ((size_t)(k + 1) * S + st) * bh; a.p[1].lda = H; a.p[1].W = P.WaT + (size_t)(k + 1) * H * H; a.p[1].ldw = H; a.p[1].K = H;
        }
        a.addend = add;
        a.ld_add = ld_add;
        ASTK_TRY(rowgemm_launch(a, s));
        add = outb; ld_add = H;
        outb = outb == P.DHTOP ? P.DLN2 : P.DHTOP;       // (a launch never adds into the buffer it reads)
      }
      if (add != P.DHTOP) ASTK_TRY(copy_f32(P.DHTOP, add, bh, s));
    }
    for (int l = top; l >= 0; --l) {
      LstmCellBwdArgs c;
      memset(&c, 0, sizeof(c));
      c.npairs = 1;
      c.p[0].A = last ? nullptr : P.G[l] + (size_t)(st + 1) * B * 4 * H;
      c.p[0].lda = 4 * H; c.p[0].W = P.WlT[l]; c.p[0].ldw = 4 * H; c.p[0].K = last ? 0 : 4 * H;
      if (P.ln) {
        // gradient wrt the LayerNorm's output (top: dh_top; below: dz_{l+1,st} Wu_{l+1}) -> through the LayerNorm -> the cell's dropped output
        const float* dout = P.DHTOP;
        if (l < top) {
          ASTK_TRY(rowgemm_launch(rg(B, H, P.G[l + 1] + (size_t)st * B * 4 * H, 4 * H, P.WuT[l + 1], 4 * H, 4 * H, P.DLN, H), s));
          dout = P.DLN;
        }
        ASTK_TRY(layernorm_bwd_launch(B, H, P.HDL[l] + (size_t)st * bh, H, prm->ln_gamma[l], LN_EPS, dout, H, P.DLN2, H, g->d_ln_gamma[l],
                                      g->d_ln_beta[l], s));
        c.dy = P.DLN2;
        c.ld_dy = H;
      } else if (l < top) {   // gradient from the layer above at the same step: dz_{l+1,st} Wu_{l+1}
        c.npairs = 2;
        c.p[1].A = P.G[l + 1] + (size_t)st * B * 4 * H;
        c.p[1].lda = 4 * H; c.p[1].W = P.WuT[l + 1]; c.p[1].ldw = 4 * H; c.p[1].K = 4 * H;
      } else {
        c.dy = P.DHTOP;
        c.ld_dy = H;
      }
      c.B = B; c.h = H;
      c.mask = rnn_masks ? rnn_masks + ((size_t)l * S + st) * bh : nullptr;
      c.dc_next = last ? nullptr : P.DC[l][(st + 1) & 1];
      c.c_prev = P.C[l] + (size_t)st * bh;
      c.c_cur = P.C[l] + (size_t)(st + 1) * bh;
      c.gates_dz = P.G[l] + (size_t)st * B * 4 * H;
      c.ld_g = 4 * H;
      c.dc_prev = P.DC[l][st & 1];
      ASTK_TRY(lstm_cell_bwd_launch(&c, 1, s));
    }
    // gradient wrt the layer-0 input [emb ; ht_{st-1}] (or the embedding alone)
    {
      RowGemmArgs a = rg(B, XI, P.G[0] + (size_t)st * B * 4 * H, 4 * H, P.WuT[0], 4 * H, 4 * H, P.DX0 + (size_t)st * B * XI, XI);
      if (P.feed && st > 0) {     // columns E.. are d_ht of step st-1: finish that step's d_pre in place
        a.carry = P.DPRE + (size_t)(st - 1) * B * A; a.ld_carry = A;
        a.carry_aux = P.HT + (size_t)st * B * A; a.ld_carry_aux = A;
        a.carry_col0 = E;
      }
      ASTK_TRY(rowgemm_launch(a, s));
    }
  }
  // ---- gradients wrt the initial states (flow into the encoder's final states, seq2seq.py:326-329)
  for (int l = 0; l < nl; ++l) {
    ASTK_TRY(rowgemm_launch(rg(B, H, P.G[l], 4 * H, P.WlT[l], 4 * H, 4 * H, d_h0 + l * bh, H), s));
    if (!persist) ASTK_TRY(copy_f32(d_c0 + l * bh, P.DC[l][0], bh, s));
  }
  // ---- d_enc[b] = sum_k alpha_k,b^T d_cv_k,b + ds_k,b^T q_k,b   (batched over b, K = S)
  if (persist && 2 * NA <= GEMM_GROUP_MAX) {
    // (the persistent launcher's fill zeroed d_enc: both products of every head ADD into it, all of them in ONE grouped launch -- two
    //  18-us launches of three k-iterations each were one after the other on the backward's chain; two contributions per element: the
    //  same sum in either order)
    GemmArgs list[GEMM_GROUP_MAX];
    int n = 0;
    for (int k = 0; k < NA; ++k) {
      GemmArgs ga = gemm_args(T, H, S, mat(P.ALPHA + (size_t)k * S * B * Tp, (long)B * Tp), mat(P.DCVH + (size_t)k * H, (long)B * CW), d_enc, H, nullptr, GEMM_ATOMIC);
      ga.batch = B; ga.sA = Tp; ga.sB = CW; ga.sC = (long)T * H;
      list[n++] = ga;
      GemmArgs gb = gemm_args(T, H, S, mat(P.DS + (size_t)k * S * B * Tp, (long)B * Tp), mat(P.Q + (size_t)k * S * bh, (long)B * H), d_enc, H, nullptr, GEMM_ATOMIC);
      gb.batch = B; gb.sA = Tp; gb.sB = H; gb.sC = (long)T * H;
      list[n++] = gb;
    }
    ASTK_TRY(gemm_launch_group(GEMM_TN, list, n, s));
  } else
  for (int k = 0; k < NA; ++k) {
    GemmArgs ga = gemm_args(T, H, S, mat(P.ALPHA + (size_t)k * S * B * Tp, (long)B * Tp), mat(P.DCVH + (size_t)k * H, (long)B * CW), d_enc, H, nullptr,
                            k == 0 ? GEMM_STORE : GEMM_ACCUM);
    ga.batch = B; ga.sA = Tp; ga.sB = CW; ga.sC = (long)T * H;
    ASTK_TRY(gemm_launch(GEMM_TN, ga, s));
    GemmArgs gb = gemm_args(T, H, S, mat(P.DS + (size_t)k * S * B * Tp, (long)B * Tp), mat(P.Q + (size_t)k * S * bh, (long)B * H), d_enc, H, nullptr, GEMM_ACCUM);
    gb.batch = B; gb.sA = Tp; gb.sB = H; gb.sC = (long)T * H;
    ASTK_TRY(gemm_launch(GEMM_TN, gb, s));
  }
  }   // do_chain
  if (!do_params) return 0;
  GemmWgCap cap(phase == ASTK_DEC_BWD_PARAMS ? d->side_wgs : 0);   // on its own stream this phase shares the CUs with the encoder's recurrence kernel
  // ==== parameter gradients: read only what the chain phase left in the workspace; nothing downstream of the decoder needs them, so a
  // caller may run this phase on a second stream beside the encoder's backward recurrence (ASTK_DEC_BWD_PARAMS)
  if (wide_b)      // the embedding columns of d_x0 (only the embedding scatter reads them): one batched product over all steps
    ASTK_TRY(gemm_launch(GEMM_NN, lowp(gemm_args(S * B, E, 4 * H, mat(P.G[0], 4 * H), mat(prm->lstm[0].Wu, XI), P.DX0, XI)), s));
  if (persist) {
    // split mode: the embedding columns of d_x0 (only the embedding scatter reads them) are one batched product over all steps
    // dq[s][b][:] = sum_t ds[s][b][t] enc[b][t][:]  (batched over b) -- only the weight gradients of attn_Wa need it
    GemmArgs gq = gemm_args(S, H, T, mat(P.DS, (long)B * Tp), mat(enc, H), P.DQ, (long)B * H);
    gq.batch = B; gq.sA = Tp; gq.sB = (long)T * H; gq.sC = H;
    const GemmArgs gx = lowp(gemm_args(S * B, E, 4 * H, mat(P.G[0], 4 * H), mat(prm->lstm[0].Wu, XI), P.DX0, XI));      // (K18's input gradient)
    if (b6s && low_precision_gemms() == 0) {      // two small independent NN products: one grouped launch (a launch less on the way to the weight gradients)
      const GemmArgs two[2] = {gx, gq};
      ASTK_TRY(gemm_launch_group(GEMM_NN, two, 2, s));
    } else {
      if (b6s) ASTK_TRY(gemm_launch(GEMM_NN, gx, s));
      ASTK_TRY(gemm_launch(GEMM_NN, gq, s));
    }
  }
  // ---- weight gradients: one batched TN GEMM each over the S*B saved rows
  const int SB = S * B;
  // BASELINE configs[4] ("fp16 MFMA GEMMs"; SURVEY 8d: fp16 operands for K6, K9, K18, K24): the batched products of the decoder LSTMs (K18)
  // and of the output layer (K24) form their own grouped launch, eligible for fp16 operands under astk_set_low_precision_gemms(1);
  // attention and context products stay f32-accurate.  (Without that mode both groups run exactly as one did.)
  WgradBatch wb, wbl(true);
  ColsumBatch cb;   // the bias gradients: one launch
  ASTK_TRY(wbl.add(g->dWo, A, V, A, P.LOGITS, Vp, P.HT + (size_t)B * A, A, SB, s));
  ASTK_TRY(cb.add(g->dbo, P.LOGITS, Vp, SB, V, s));
  ASTK_TRY(wb.add(g->dWc, CW, A, CW, P.DPRE, A, P.CVH, CW, SB, s));
  ASTK_TRY(cb.add(g->dbc, P.DPRE, A, SB, A, s));
  for (int k = 0; k < NA; ++k) {
    float* dWa = k == 0 ? g->dWa : g->dWa_x[k - 1];
    float* dba = k == 0 ? g->dba : g->dba_x[k - 1];
    ASTK_CHECK(dWa && dba, "decoder_bwd: attention head %d has no gradient buffers", k);
    ASTK_TRY(wb.add(dWa, H, H, H, P.DQ + (size_t)k * S * bh, H, P.CVH + (size_t)NA * H, CW, SB, s));
    ASTK_TRY(cb.add(dba, P.DQ + (size_t)k * S * bh, H, SB, H, s));
  }
  for (int l = 0; l < nl; ++l) {
    const int in = l == 0 ? XI : H;
    const float* xin;
    long ldx;
    if (l == 0) { xin = P.X0; ldx = XI; }
    else if (rnn_masks || P.ln) { xin = P.HD[l - 1]; ldx = H; }       // (with LayerNorm the layer's input is always the normalised copy)
    else { xin = P.HR[l - 1] + bh; ldx = H; }
    ASTK_TRY(wbl.add(g->lstm[l].dWu, in, 4 * H, in, P.G[l], 4 * H, xin, ldx, SB, s));
    ASTK_TRY(wbl.add(g->lstm[l].dWl, H, 4 * H, H, P.G[l], 4 * H, P.HR[l], H, SB, s));
    ASTK_TRY(cb.add(g->lstm[l].db, P.G[l], 4 * H, SB, 4 * H, s));
  }
  ASTK_TRY(cb.flush(s));
  if (low_precision_gemms()) {
    ASTK_TRY(wb.flush(s));
    ASTK_TRY(wbl.flush(s));
  } else {            // one grouped launch, as before the two were told apart
    for (int i = 0; i < wbl.n; ++i) {
      if (wb.n == GEMM_GROUP_MAX) ASTK_TRY(wb.flush(s));
      wb.list[wb.n++] = wbl.list[i];
    }
    wbl.n = 0;
    ASTK_TRY(wb.flush(s));
  }
  if (deterministic_mode()) hipLaunchKernelGGL(k_embed_bwd_det, dim3(V), dim3(128), 0, s, g->d_embed, P.TOK, P.DX0, emb_mask, SB, E, XI, V);
  else hipLaunchKernelGGL(k_embed_bwd, dim3(SB), dim3(128), 0, s, g->d_embed, P.TOK, P.DX0, emb_mask, SB, E, XI, V);
  ASTK_LAUNCH_CHECK();
  return 0;
}

int astk_decoder_step_infer(const astk_decoder_desc* d, const astk_decoder_params* prm, const float* enc, float* c, float* h, float* ht,
                            const int32_t* tokens, float* logits, float* alpha, int32_t* argmax, void* ws, size_t ws_bytes,
                            void* stream) {
  hipStream_t s = (hipStream_t)stream;
  ASTK_CHECK_DESC(d, astk_decoder_desc);
  PrecScope prec_scope(d->precision, d->gemm_operands);
  DecPlan P;
  ASTK_TRY(make_plan(d, ws, P));
  ASTK_CHECK(ws && ws_bytes >= P.bytes, "decoder_step_infer: workspace too small");
  ASTK_CHECK(prm && enc && c && h && ht && tokens && logits, "decoder_step_infer: null pointer");
  const int B = P.B, H = P.H, E = P.E, A = P.A, V = P.V, XI = P.XI, nl = P.nl, NA = P.NA, CW = P.CW;
  const size_t bh = (size_t)B * H;
  float* x0 = P.X0;
  ASTK_TRY(attn_ws_init(P.attn_ws, B, P.T, H, s));
  hipLaunchKernelGGL(k_embed, dim3(B), dim3(128), 0, s, prm->embed, (const int32_t*)nullptr, 0, 0, (const int32_t*)nullptr,
                     (const int32_t*)nullptr, tokens, (int32_t*)nullptr, (const float*)nullptr, x0, B, E, XI, V);
  ASTK_LAUNCH_CHECK();
  if (P.feed) ASTK_TRY(copy2d_f32(x0 + E, XI, ht, A, B, A, A, s));
  float* cvh = P.CVH;
  float* htop = cvh + (size_t)NA * H;
  const float* x_in = x0;
  long ld_x = XI;
  int in = XI;
  for (int l = 0; l < nl; ++l) {
    float* hd = l == nl - 1 ? htop : P.HD[l];
    const long ld_hd = l == nl - 1 ? CW : H;
    float* raw = P.ln ? P.HDL[l] : hd;          // ln: the cell's output goes through the LayerNorm first
    // new states go to scratch first (the cell reads h_prev while other workgroups write h_out)
    ASTK_TRY(cell_fwd(P, prm, l, x_in, ld_x, in, h + l * bh, c + l * bh, P.G[l], P.C[l], P.HR[l], nullptr, raw, P.ln ? H : ld_hd, s));
    if (P.ln) ASTK_TRY(layernorm_fwd_launch(B, H, raw, H, prm->ln_gamma[l], prm->ln_beta[l], LN_EPS, hd, ld_hd, s));
    ASTK_TRY(copy_f32(c + l * bh, P.C[l], bh, s));
    ASTK_TRY(copy_f32(h + l * bh, P.HR[l], bh, s));
    x_in = hd; ld_x = ld_hd; in = H;
  }
  for (int k = 0; k < NA; ++k) {
    float* q = P.Q + (size_t)k * bh;
    RowGemmArgs a = rg(B, H, htop, CW, k == 0 ? prm->Wa : prm->Wa_x[k - 1], H, H, q, H);
    a.bias = k == 0 ? prm->ba : prm->ba_x[k - 1];
    ASTK_TRY(rowgemm_launch(a, s));
    // (the alphas handed back are the FIRST head's, seq2seq.py:379-383)
    ASTK_TRY(attn_fwd_launch(B, P.T, H, enc, q, H, k == 0 ? P.ALPHA : P.DS, cvh + (size_t)k * H, CW, nullptr, 0, P.attn_ws, s));
  }
  if (alpha) ASTK_TRY(copy2d_f32(alpha, P.T, P.ALPHA, P.Tp, B, P.T, P.T, s));
  {
    RowGemmArgs a = rg(B, A, cvh, CW, prm->Wc, CW, CW, ht, A);
    a.bias = prm->bc;
    a.act = ACT_TANH;
    ASTK_TRY(rowgemm_launch(a, s));
  }
  {
    RowGemmArgs a = rg(B, V, ht, A, prm->Wo, A, A, logits, V);
    a.bias = prm->bo;
    ASTK_TRY(rowgemm_launch(a, s));
  }
  if (argmax) {
    // argmax only: run the CE kernel on a scratch copy so that `logits` stays intact
    ASTK_TRY(copy2d_f32(P.LOGITS, P.Vp, logits, V, B, V, P.Vp, s));
    ASTK_TRY(softmax_ce_launch(B, V, P.Vp, P.LOGITS, tokens, 1, nullptr, 1.f, nullptr, argmax, s));
  }
  return 0;
}

}  // extern "C"
