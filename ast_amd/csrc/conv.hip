// CNN front-end of the speech encoder (seq2seq.py:158-180):
//   [Conv2D(nobias) -> BatchNorm(batch statistics) -> ReLU] x n, then the (T'',B,C*F') re-layout (Q9).
//
// Data layout in HBM (all f32):
//   layer 0 : X (B,T,D) [* noise] --im2col--> P0 [(b,f,t1)][kt*kf padded to 4]  (coalesced reads of the padded
//             (T,D) frame matrix: one workgroup streams whole frame rows) -> Y0 = P0 W0^T  (MFMA GEMM)
//   layer i : activations are kept channels-last and time-padded, HP[(b,f)][pt + T_i + pt][C_i]; because the
//             frequency kernel/stride of layers >= 1 is 1, the im2col row of output (b,f,t) is the CONTIGUOUS
//             window HP[(b,f)][t*st .. t*st+kt)[:] -- the conv runs as a zero-copy "window" GEMM (two-level row
//             addressing in gemm.hip), K = kt*C_{i-1}, against the weight re-packed to [co][kt][ci].
//   backward: wgrad = TN window GEMM with split-K atomics; dgrad = one window GEMM per stride phase over the
//             time-padded dY (transposed convolution without col2im scatter).
//   BatchNorm: column statistics of the raw conv output [rows][C] in float64 atomics, then one fused
//             scale/shift/ReLU pass that writes the next layer's padded layout (or the LSTM layout).
#include "common.h"
#include <atomic>
#include <mutex>
#include <algorithm>
#include <cstdlib>

namespace astk {

namespace {

struct CnnPlan {
  int n;
  int B, T, D;
  // Tn / Fn / rows: what layer i hands on (after its optional max-pool, enc_dec.py:444-456); Tc / Fc / rowsc: its convolution's output.
  // Without pooling (every shipped config) the two sets are equal and YC[i] == Y[i].
  int Tn[ASTK_MAX_CNN_LAYERS], Fn[ASTK_MAX_CNN_LAYERS];
  int Tc[ASTK_MAX_CNN_LAYERS], Fc[ASTK_MAX_CNN_LAYERS];
  int pwt[ASTK_MAX_CNN_LAYERS], pwf[ASTK_MAX_CNN_LAYERS];   // pooling windows (1 = none)
  bool pooled[ASTK_MAX_CNN_LAYERS];
  int Cn[ASTK_MAX_CNN_LAYERS];
  int rows[ASTK_MAX_CNN_LAYERS];    // B*Fn[i]*Tn[i]
  int rowsc[ASTK_MAX_CNN_LAYERS];   // B*Fc[i]*Tc[i]
  int K0, K0p;                      // layer-0 patch size and its padded width
  int JG, K0g, xf_rows;             // direct layer-0 path: padded frequency-kernel width of XF, kt * JG, rows per (b, f) group
  float* XF;
  int padA[ASTK_MAX_CNN_LAYERS];    // time padding of HP_i (= pt of layer i+1), front == back
  int dF[ASTK_MAX_CNN_LAYERS], dB[ASTK_MAX_CNN_LAYERS];  // front/back padding of DYP_i for the dgrad windows (i>=1)
  // workspace slices
  float* P0;
  float* Y[ASTK_MAX_CNN_LAYERS];    // what BatchNorm sees: the convolution's output, max-pooled when the layer pools
  float* YC[ASTK_MAX_CNN_LAYERS];   // the convolution's output (== Y[i] without pooling)
  int32_t* IDX[ASTK_MAX_CNN_LAYERS];  // pooled layers: [rows][C] row of YC that won the window
  float* DYP[ASTK_MAX_CNN_LAYERS];    // pooled layers: gradient wrt Y[i] (un-pooled into DY[i] by k_unpool)
  float* HP[ASTK_MAX_CNN_LAYERS];   // i < n-1
  float* Wr[ASTK_MAX_CNN_LAYERS];   // repacked weights (i=0: padded (C0,K0p); i>=1: (C_i, kt*C_{i-1}))
  double* stat[ASTK_MAX_CNN_LAYERS];  // per layer [2][C] column sums (double)
  size_t zero_fwd_bytes, zero_bwd_bytes;  // one fill from stat[0] zeroes the statistics of all layers (forward) / and the dWr scratch (backward)
  // Absolute maxima (fp16x2 GEMM operand scales) taken by the kernels that WRITE the matrices, 16 words each (common.h amax_emit_block):
  //   a_hp[i]  HP[i], the padded activations (K6's A operand, forward; the weight gradient's B operand)    written by k_bn_relu_rows
  //   a_wr[i]  Wr[i], the re-packed weights (K6's B operand)                                                written by k_repack_w
  //   a_out    the (T'',B,C*F') output = the LSTM stacks' frames (astk_conv_out_amax hands the address on)  written by k_bn_relu_to_seq
  //   a_dy[i]  the padded dY (weight gradient + every stride phase of the input gradient)                   written by k_bn_bwd_apply
  //   a_wd[k]  the phase weights of a grouped dgrad launch                                                  written by k_phase_w
  // The forward slots sit in front of the statistics (zeroed by the forward fill, untouched by the backward one), the backward slots
  // behind them (zeroed by the backward fill).
  // a_hp, a_out and a_dy come from kernels with thousands of emitting blocks: STRIDED producer slots (`_s`, shards on separate lines)
  // that a small kernel running in between anyway folds into the plain slot the GEMMs read (k_repack_w of the next layer folds a_hp, the
  // first k_phase_w folds a_dy, the LSTM stack's k_perm_rows folds a_out); a_wr / a_wd come from small grids: plain slots.
  unsigned long long *a_hp[ASTK_MAX_CNN_LAYERS], *a_wr[ASTK_MAX_CNN_LAYERS], *a_out, *a_dy[ASTK_MAX_CNN_LAYERS], *a_wd[GEMM_GROUP_MAX];
  unsigned long long *a_hp_s[ASTK_MAX_CNN_LAYERS], *a_dy_s[ASTK_MAX_CNN_LAYERS];
  void* zero_fwd_from;
  size_t zero_fwd_amax_bytes;             // the forward slots alone (eval mode: no statistics to zero)
  unsigned* fin_ctr;                      // [layer * 64], inside the forward slots' region
  float* c0_part;                         // direct layer-0 kernel: [c0_tiles][2][C] sums of y and y^2 over each tile's output steps
  int c0_tiles;
  float* bn[ASTK_MAX_CNN_LAYERS];   // [4][C]: mean, inv_std, scale, shift
  float* G;                         // [rows_max][Cmax] gradient wrt post-ReLU output (row layout)
  float* DY[ASTK_MAX_CNN_LAYERS];   // padded dY (i>=1) / plain dY (i=0)
  float* dWr[ASTK_MAX_CNN_LAYERS];    // per layer scratch for re-packed weight gradients
  float* Wd;                        // phase weights for dgrad: wd_copies buffers of wd_stride floats
  size_t wd_stride;
  int wd_copies;
  size_t bytes;
};

int conv_out(int n, int k, int s, int p) { return (n + 2 * p - k) / s + 1; }

int make_plan(const astk_cnn_desc* d, void* ws, CnnPlan& P) {
  ASTK_CHECK_DESC(d, astk_cnn_desc);
  ASTK_CHECK(d && d->n_layers >= 1 && d->n_layers <= ASTK_MAX_CNN_LAYERS, "cnn: 1..%d layers", ASTK_MAX_CNN_LAYERS);
  ASTK_CHECK(d->B > 0 && d->T > 0 && d->D > 0, "cnn: bad input dims");
  P.n = d->n_layers; P.B = d->B; P.T = d->T; P.D = d->D;
  ASTK_CHECK(d->kf[0] >= 1 && d->kf[0] <= d->D && d->sf[0] >= 1, "cnn: bad layer-0 frequency kernel");
  int t = d->T, f = conv_out(d->D, d->kf[0], d->sf[0], 0);
  size_t cmax = 0, rowsmax_c = 0;
  for (int i = 0; i < P.n; ++i) {
    ASTK_CHECK(d->kt[i] >= 1 && d->st[i] >= 1 && d->pt[i] >= 0 && d->C[i] >= 1, "cnn: bad layer %d", i);
    ASTK_CHECK(i == 0 || (d->kf[i] == 1 && d->sf[i] == 1), "cnn: layers >= 1 must have a (kt,1) kernel and frequency stride 1");
    ASTK_CHECK(d->kt[i] >= d->st[i], "cnn: time kernel must be >= time stride (layer %d)", i);
    ASTK_CHECK((d->C[i] % 4) == 0, "cnn: channel counts must be multiples of 4 (layer %d: %d)", i, d->C[i]);
    t = conv_out(t, d->kt[i], d->st[i], d->pt[i]);
    ASTK_CHECK(t >= 1 && f >= 1, "cnn: input too short for layer %d", i);
    P.Tc[i] = t; P.Fc[i] = f;
    P.rowsc[i] = d->B * f * t;
    ASTK_CHECK(d->pool_t[i] >= -1 && d->pool_f[i] >= -1, "cnn: bad pooling window (layer %d)", i);
    P.pwt[i] = d->pool_t[i] == -1 ? t : (d->pool_t[i] > 1 ? d->pool_t[i] : 1);
    P.pwf[i] = d->pool_f[i] == -1 ? f : (d->pool_f[i] > 1 ? d->pool_f[i] : 1);
    P.pooled[i] = P.pwt[i] > 1 || P.pwf[i] > 1;
    t = (t + P.pwt[i] - 1) / P.pwt[i];      // cover_all: the last window may be partial
    f = (f + P.pwf[i] - 1) / P.pwf[i];
    P.Tn[i] = t; P.Fn[i] = f;
    P.Cn[i] = d->C[i];
    P.rows[i] = d->B * f * t;
    cmax = cmax > (size_t)d->C[i] ? cmax : (size_t)d->C[i];
    size_t rc = (size_t)P.rows[i] * d->C[i];
    rowsmax_c = rowsmax_c > rc ? rowsmax_c : rc;
  }
  P.K0 = d->kt[0] * d->kf[0];
  P.K0p = (P.K0 + 3) / 4 * 4;
  Carver c(ws);
  P.P0 = c.take<float>((size_t)P.rowsc[0] * P.K0p);
  // direct layer-0 convolution (k_conv0_fwd_x3): the noisy input in frequency-blocked, time-padded form [(b, f)][pt + T + pt][JG], which
  // the weight gradient reads as a zero-copy window matrix (row (b, f, t1) = the kt*JG floats from row st*t1 on)
  P.JG = (d->kf[0] + 1) & ~1;
  P.K0g = d->kt[0] * P.JG;
  P.xf_rows = (d->T + 2 * d->pt[0] + 1) & ~1;      // (even: the group stride of the window view must be a multiple of 4 floats)
  P.XF = c.take<float>((size_t)d->B * P.Fc[0] * P.xf_rows * P.JG + 16);
  // fused layer-0 statistics of the direct kernel: one row [2][C] of float sums per tile (C0_TT output steps of one (b, f))
  P.c0_tiles = d->B * P.Fc[0] * cdiv(P.Tc[0], 80);
  P.c0_part = c.take<float>((size_t)P.c0_tiles * 2 * P.Cn[0]);
  size_t wd_max = 0;
  for (int i = 0; i < P.n; ++i) {
    P.Y[i] = c.take<float>((size_t)P.rows[i] * P.Cn[i]);
    if (P.pooled[i]) {
      P.YC[i] = c.take<float>((size_t)P.rowsc[i] * P.Cn[i]);
      P.IDX[i] = c.take<int32_t>((size_t)P.rows[i] * P.Cn[i]);
      P.DYP[i] = c.take<float>((size_t)P.rows[i] * P.Cn[i]);
    } else {
      P.YC[i] = P.Y[i]; P.IDX[i] = nullptr; P.DYP[i] = nullptr;
    }
    P.bn[i] = c.take<float>(4 * (size_t)P.Cn[i]);
    if (i < P.n - 1) {
      P.padA[i] = d->pt[i + 1];
      P.HP[i] = c.take<float>((size_t)d->B * P.Fn[i] * (P.Tn[i] + 2 * P.padA[i]) * P.Cn[i]);
    } else {
      P.padA[i] = 0;
      P.HP[i] = nullptr;
    }
    if (i == 0) {
      P.Wr[0] = c.take<float>((size_t)d->C[0] * P.K0p);
      P.dF[0] = P.dB[0] = 0;
      P.DY[0] = c.take<float>((size_t)P.rowsc[0] * P.Cn[0]);
    } else {
      const int KT = d->kt[i], st = d->st[i], pt = d->pt[i];
      P.Wr[i] = c.take<float>((size_t)P.Cn[i] * KT * P.Cn[i - 1]);
      const int na_max = (KT + st - 1) / st;
      int f = na_max - 1 - pt / st;
      P.dF[i] = f > 0 ? f : 0;
      const int qmax = (P.Tn[i - 1] - 1 + pt) / st;
      int bk = qmax - (P.Tc[i] - 1);
      P.dB[i] = bk > 0 ? bk : 0;
      P.DY[i] = c.take<float>((size_t)d->B * P.Fc[i] * (P.Tc[i] + P.dF[i] + P.dB[i]) * P.Cn[i]);
      size_t wd = (size_t)P.Cn[i - 1] * na_max * P.Cn[i];
      wd_max = wd_max > wd ? wd_max : wd;
    }
  }
  // the buffers that have to be zero before use, back to back: ONE fill per pass instead of one per layer and buffer
  const size_t off_afwd = align_up(c.off, 256);
  for (int i = 0; i < P.n; ++i) {
    P.a_hp_s[i] = amax_pslot_handle(c.take<unsigned long long>(AMAX_PSLOT_WORDS));
    P.a_hp[i] = c.take<unsigned long long>(AMAX_SLOT_WORDS);
    P.a_wr[i] = c.take<unsigned long long>(AMAX_SLOT_WORDS);
  }
  P.a_out = amax_pslot_handle(c.take<unsigned long long>(AMAX_PSLOT_WORDS));
  P.fin_ctr = c.take<unsigned>(64 * ASTK_MAX_CNN_LAYERS);       // arrival counters of k_colstats' blocks (the last one finalizes), 256 bytes apart
  P.zero_fwd_from = ws ? (char*)ws + off_afwd : nullptr;
  const size_t off_stat = align_up(c.off, 256);
  P.zero_fwd_amax_bytes = off_stat - off_afwd;
  for (int i = 0; i < P.n; ++i) P.stat[i] = c.take<double>(2 * (size_t)P.Cn[i]);
  const size_t off_abwd = align_up(c.off, 256);
  for (int i = 0; i < P.n; ++i) {
    P.a_dy_s[i] = amax_pslot_handle(c.take<unsigned long long>(AMAX_PSLOT_WORDS));
    P.a_dy[i] = c.take<unsigned long long>(AMAX_SLOT_WORDS);
  }
  for (int k = 0; k < GEMM_GROUP_MAX; ++k) P.a_wd[k] = c.take<unsigned long long>(AMAX_SLOT_WORDS);
  const size_t off_dwr = align_up(c.off, 256);
  for (int i = 0; i < P.n; ++i)
    P.dWr[i] = c.take<float>(i == 0 ? (size_t)d->C[0] * std::max(P.K0p, (P.K0g + 3) & ~3) : (size_t)P.Cn[i] * d->kt[i] * P.Cn[i - 1]);
  P.zero_fwd_bytes = off_abwd - off_afwd;       // forward: the forward slots and the statistics
  P.zero_bwd_bytes = c.off - off_stat;          // backward: statistics, backward slots, dWr scratch
  P.G = c.take<float>(rowsmax_c);
  // one phase-weight buffer per stride phase of a grouped dgrad launch
  int st_max = 1;
  for (int i = 1; i < P.n; ++i) st_max = d->st[i] > st_max ? d->st[i] : st_max;
  P.wd_stride = (wd_max ? wd_max : 4);
  P.wd_copies = st_max < GEMM_GROUP_MAX ? st_max : GEMM_GROUP_MAX;
  P.Wd = c.take<float>(P.wd_stride * P.wd_copies);
  P.bytes = c.total();
  return 0;
}

// ------------------------------------------------------------------ kernels
// P0[(b,f,t1)][k = a*kf + j] = X[b][t1*st - pt + a][f*sf + j] * noise ; one block per (b, t1): the kt frame rows it
// needs are read as whole rows (coalesced), every f patch is written as a contiguous K0p-float row.
__global__ __launch_bounds__(256) void k_im2col0(const float* __restrict__ X, const float* __restrict__ noise, float* __restrict__ P0,
                                                 int B, int T, int D, int F, int T1, int kt, int kf, int st, int sf, int pt, int K0p) {
  const int t1 = blockIdx.x, b = blockIdx.y;
  const int K0 = kt * kf;
  const int total = F * K0p;
  for (int i = threadIdx.x; i < total; i += blockDim.x) {
    const int f = i / K0p, k = i % K0p;
    float v = 0.f;
    if (k < K0) {
      const int a = k / kf, j = k % kf;
      const int t = t1 * st - pt + a;
      if (t >= 0 && t < T) {
        const long idx = ((long)b * T + t) * D + f * sf + j;
        v = X[idx];
        if (noise) v *= noise[idx];
      }
    }
    P0[(((long)b * F + f) * T1 + t1) * K0p + k] = v;
  }
}

// ---- Layer 0 as a DIRECT convolution on the bf16 matrix pipe (default arithmetic, bf16x3).  im2col + GEMM moved 37 MB of patches out and
// back for a product with K = 117: 28 + 47 us, overhead-bound (8 k-iterations per tile).  The frequency stride equals the frequency
// kernel (13), so with the input of one (b, f) kept as [time][JP] in LDS the patch of output step t1 is simply the kt*JP CONTIGUOUS
// elements from flat offset st*JP*t1 on (kt rows of JP, the pads multiplied by zero weights): the MFMA operand fragments are read straight
// from that image -- no patch matrix anywhere.  JP = 20: st*JP*2 bytes = 80 per output step keeps every fragment 16-byte aligned and the
// 16 lanes of a read on 16 different bank groups.  A workgroup owns C0_TT output steps of one (b, f) and all channels (wave w: channels
// 32 w .. 32 w + 31, their weight fragments resident in registers as three bf16 planes); Y^T = W P^T, so a lane ends up with four
// CONSECUTIVE channels of one output row (16-byte stores).  It also leaves the noisy input in the frequency-blocked, time-padded f32
// form XF[(b, f)][pt + T + pt][JG] that the weight gradient reads as a zero-copy window matrix (the P0 patches are not built at all).
constexpr int C0_JP = 20;
constexpr int C0_TT = 80;
constexpr int C0_NKS = 6;           // 32-k MFMA steps over kt * JP <= 192 (kt <= 9)
static int conv0_win_elems(int st) { return st * C0_JP * (C0_TT - 1) + 32 * C0_NKS + 40; }    // (+ the rows the last tile of a group owns in XF)
static bool conv0_direct_shape(const astk_cnn_desc* d) {
  return conv0_win_elems(d->st[0]) <= 512 * 8 && (size_t)d->C[0] * d->kt[0] * d->kf[0] * 4 <= 64 * 1024 && d->kt[0] * C0_JP <= 32 * C0_NKS && d->kf[0] <= 14 && (d->st[0] % 2) == 0 && d->C[0] <= 128 && (d->C[0] % 16) == 0 && d->pool_t[0] <= 1 &&
         d->pool_f[0] <= 1;
}
// the direct path is taken for the shipped layer-0 shapes under the default arithmetic (bf16x3, f32 operands); forward and backward decide
// alike (same descriptor, same process default).  astk_set_tuning("conv.direct0", 0): the im2col + GEMM path always.
static bool conv0_direct(const astk_cnn_desc* d) {
  const bool off = !tune_on(TUNE_CONV_DIRECT0);
  return !off && conv0_direct_shape(d) && gemm_precision_mode() == 1 && low_precision_gemms() == 0;
}
// Which layer-0 path a forward call took on a workspace (host-side record, never read by a kernel): conv0_direct() is re-evaluated by
// the backward call from the arithmetic in force THEN; if the process default moved in between while the descriptor says DEFAULT, the
// backward would read a window matrix XF that was never written (or patches P0 that were never built) -- a silently wrong CNN_0/W
// gradient from stale workspace.  The forward records (workspace, path); the backward refuses a workspace whose record differs.
// (bwd_clean: the forward call's last kernel zeroed the region the backward accumulates into -- statistics, backward maximum slots, dWr
//  scratch -- on its way out, so the backward call that follows needs no fill launch of its own; the backward call takes the mark away.)
struct Conv0PathRecord { const void* ws; int direct; int bwd_clean; };
static std::mutex g_conv0_mu;
static Conv0PathRecord g_conv0_ring[64];
static unsigned g_conv0_next = 0;
static void conv0_path_record(const void* ws, bool direct, bool bwd_clean) {
  std::lock_guard<std::mutex> lock(g_conv0_mu);
  for (auto& r : g_conv0_ring)
    if (r.ws == ws) { r.direct = direct ? 1 : 0; r.bwd_clean = bwd_clean ? 1 : 0; return; }
  g_conv0_ring[g_conv0_next++ % 64] = Conv0PathRecord{ws, direct ? 1 : 0, bwd_clean ? 1 : 0};
}
static int conv0_path_lookup(const void* ws) {      // -1: no forward call recorded for this workspace
  std::lock_guard<std::mutex> lock(g_conv0_mu);
  for (auto& r : g_conv0_ring)
    if (r.ws == ws) return r.direct;
  return -1;
}
static bool conv_take_bwd_clean(const void* ws) {   // true once per marked forward call
  std::lock_guard<std::mutex> lock(g_conv0_mu);
  for (auto& r : g_conv0_ring)
    if (r.ws == ws) { const bool c = r.bwd_clean != 0; r.bwd_clean = 0; return c; }
  return false;
}
// Every thread of the calling grid: zero `bytes` (a multiple of 4) from the 16-byte aligned p.  A fill launch of its own costs ~5 us of
// stream time; the kernels in front of the consumers do it on their way in / out instead.
__device__ __forceinline__ void zero_region(void* p, size_t bytes) {
  if (!p) return;
  const size_t n16 = bytes / 16, n4 = bytes / 4;
  const size_t nth = (size_t)gridDim.x * gridDim.y * blockDim.x;
  const size_t me = ((size_t)blockIdx.y * gridDim.x + blockIdx.x) * blockDim.x + threadIdx.x;
  for (size_t i = me; i < n16; i += nth) reinterpret_cast<uint4*>(p)[i] = make_uint4(0u, 0u, 0u, 0u);
  for (size_t i = n16 * 4 + me; i < n4; i += nth) reinterpret_cast<unsigned*>(p)[i] = 0u;
}
// bn[0]=mean, bn[1]=inv_std, bn[2]=scale, bn[3]=shift ; running statistics per Chainer-sem A4
struct BnFinalize {
  unsigned* ctr;          // k_colstats: arrival counter of the launch's blocks (zero before the launch); nullptr = no fused finalize
  double m;
  const float *gamma, *beta;
  float *avg_mean, *avg_var, *bn;
  float eps, decay;
};
__device__ __forceinline__ void bn_finalize_channel(int c, int C, double s0, double s1, const BnFinalize& f, int train) {
  float mean, inv_std;
  if (train) {
    const double mu = s0 / f.m;
    double var = s1 / f.m - mu * mu;
    if (var < 0) var = 0;
    mean = (float)mu;
    inv_std = (float)(1.0 / sqrt(var + (double)f.eps));
    const double adjust = f.m / (f.m - 1.0 > 1.0 ? f.m - 1.0 : 1.0);
    f.avg_mean[c] = f.decay * f.avg_mean[c] + (1.f - f.decay) * mean;
    f.avg_var[c] = f.decay * f.avg_var[c] + (float)((1.0 - (double)f.decay) * adjust * var);
  } else {
    mean = f.avg_mean[c];
    inv_std = 1.f / sqrtf(f.avg_var[c] + f.eps);
  }
  const float sc = f.gamma[c] * inv_std;
  f.bn[c] = mean;
  f.bn[C + c] = inv_std;
  f.bn[2 * C + c] = sc;
  f.bn[3 * C + c] = f.beta[c] - mean * sc;
}
constexpr int C0_NPAIR = 8;           // pairs of window elements per thread and tile (256 threads x 8 x 2 >= the window)
__global__ __launch_bounds__(256, 2) void k_conv0_fwd_x3(const float* __restrict__ X, const float* __restrict__ noise, const float* __restrict__ W,
                                                         float* __restrict__ Y, float* __restrict__ XF, int B, int T, int D, int F, int T1, int C,
                                                         int kt, int kf, int st, int sf, int pt, int JG, int xf_rows, int tiles_t, int total, int win,
                                                         void* zero_from, size_t zero_bytes, float* __restrict__ part) {
  extern __shared__ __attribute__((aligned(16))) unsigned short c0_planes[];      // [3][win] bf16; first the f32 weights [C][kt*kf]
  // BatchNorm statistics of the layer's output on the way out (part != nullptr; k_colstats read the 262 MB back: 19-21 us): per lane and
  // tile, sums over the steps it stores; folded over the 16 step lanes behind the tile's products; one row [2][C] of float sums per TILE,
  // plain stores -- the statistics kernel behind this one sums 6400 such rows (6.5 MB) instead of the layer's output.
  zero_region(zero_from, zero_bytes);      // the statistics / maximum slots the kernels BEHIND this one accumulate into
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, r = lane & 15, q = lane >> 4;
  const int ch0 = wave * 32;
  const float* const nz = noise ? noise : X;
  // The window of a tile, pairs of elements per thread: element e = (row e / JP, column e % JP), row = padded input time st*t0 + e / JP;
  // four pairs' loads in flight before the first is used (all eight at once, or the next tile's window fetched behind the products:
  // 16 more live registers next to the 144 of the weight fragments spill -- 46 -> 53-58 us).
  // ---- resident weight fragments ("A" operand: 16 channels x 32 k): lane (channel r, k-group q) holds k = 32 ks + 8 q .. + 7 of
  // W16[c][i * JP + j] (zero in the pad columns).  Through LDS: the fragment gather straight from memory touched 16-32 cache lines per
  // load instruction, 96 instructions per lane -- 9 us per workgroup.
  HML8 wf[2][C0_NKS];
  {
    float* const wl = reinterpret_cast<float*>(c0_planes);
    const int K0 = kt * kf;
    for (int i0 = tid; i0 < C * K0; i0 += 256 * 8) {      // (eight loads in flight per thread)
      float w8[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) w8[u] = W[min(i0 + 256 * u, C * K0 - 1)];
#pragma unroll
      for (int u = 0; u < 8; ++u)
        if (i0 + 256 * u < C * K0) wl[i0 + 256 * u] = w8[u];
    }
    __syncthreads();
#pragma unroll
    for (int ct = 0; ct < 2; ++ct)
#pragma unroll
      for (int ks = 0; ks < C0_NKS; ++ks) {
        const int c = ch0 + 16 * ct + r;
        float v[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const int k = 32 * ks + 8 * q + e, i = k / C0_JP, j = k - i * C0_JP;
          const bool ok = c < C && i < kt && j < kf;
          const float w = wl[(min(c, C - 1) * kt + min(i, kt - 1)) * kf + min(j, kf - 1)];
          v[e] = ok ? w : 0.f;
        }
        wf[ct][ks] = split8b(make_float4(v[0], v[1], v[2], v[3]), make_float4(v[4], v[5], v[6], v[7]));
      }
  }
  unsigned short* const ph = c0_planes, * const pm = c0_planes + win, * const pl = c0_planes + 2 * win;
  for (int tile = blockIdx.x; tile < total; tile += gridDim.x) {
    const int tt = tile % tiles_t, bf = tile / tiles_t;
    const int t0 = tt * C0_TT;
    // rows of XF this tile owns: [st*t0, st*(t0 + TT)), the last tile of a group up to the group's end
    const int own_hi = tt == tiles_t - 1 ? xf_rows - st * t0 : st * C0_TT;
    float* const xf = XF + ((long)bf * xf_rows + (long)st * t0) * JG;
    __syncthreads();                      // the previous tile's fragment reads (the first tile: the weight reads) are done
    {
      const int f = bf % F, b = bf / F, tb = st * t0 - pt;
#pragma unroll 1
      for (int u0 = 0; u0 < C0_NPAIR; u0 += 4) {
        float px0[4], px1[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const int e = min(2 * tid + 512 * (u0 + u), win - 2);
          const int tl = e / C0_JP, j = e - tl * C0_JP, t = tb + tl;
          const bool ok0 = t >= 0 && t < T && j < kf;            // (kf odd: the pair's second element may be the first pad column)
          const bool ok1 = ok0 && j + 1 < kf;
          const long idx = ((long)b * T + min(max(t, 0), T - 1)) * D + f * sf + min(j, kf - 1);
          const long idx1 = idx + (j + 1 < kf ? 1 : 0);
          float x0 = X[idx], x1 = X[idx1];
          const float n0 = nz[idx], n1 = nz[idx1];
          if (noise) { x0 *= n0; x1 *= n1; }
          px0[u] = ok0 ? x0 : 0.f;
          px1[u] = ok1 ? x1 : 0.f;
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const int e = 2 * tid + 512 * (u0 + u);
          if (e < win) {
            const int tl = e / C0_JP, j = e - tl * C0_JP;
            unsigned h, m, l;
            split2b(px0[u], px1[u], h, m, l);
            *reinterpret_cast<unsigned*>(ph + e) = h;
            *reinterpret_cast<unsigned*>(pm + e) = m;
            *reinterpret_cast<unsigned*>(pl + e) = l;
            if (tl < own_hi && j < JG) *reinterpret_cast<float2*>(xf + (long)tl * JG + j) = make_float2(px0[u], px1[u]);
          }
        }
      }
    }
    __syncthreads();
    // ---- products: Y^T[32 channels of this wave][16 steps] over kt * JP, two blocks of 16 output steps at a time: four independent
    // accumulators, the six term products of a 32-k step issued term by term across them (back to back on ONE accumulator every MFMA
    // waits for the one before it: 2.5 x the time)
    auto xfrag = [&](int off, int ks) {
      HML8 x;
      x.hi = *reinterpret_cast<const u32q*>(ph + off + 32 * ks);
      x.mid = *reinterpret_cast<const u32q*>(pm + off + 32 * ks);
      x.lo = *reinterpret_cast<const u32q*>(pl + off + 32 * ks);
      return x;
    };
    float sa[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f}, qa[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};      // this tile: sum y, sum y^2 of the lane's 8 channels
    auto store_y = [&](int ti, const f32x4& a0, const f32x4& a1) {
      const int t1 = t0 + 16 * ti + r;
      if (t1 < T1) {
        float* const yr = Y + ((long)bf * T1 + t1) * C;
        const int c = ch0 + 4 * q;
        if (c < C) *reinterpret_cast<float4*>(yr + c) = make_float4(a0[0], a0[1], a0[2], a0[3]);
        if (c + 16 < C) *reinterpret_cast<float4*>(yr + c + 16) = make_float4(a1[0], a1[1], a1[2], a1[3]);
        if (part) {
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            sa[e] += a0[e]; qa[e] += a0[e] * a0[e];
            sa[4 + e] += a1[e]; qa[4 + e] += a1[e] * a1[e];
          }
        }
      }
    };
#define C0_TERM4_(PA, PX) MFMA_B16_(a00, wf[0][ks].PA, xa.PX) MFMA_B16_(a10, wf[1][ks].PA, xa.PX) MFMA_B16_(a01, wf[0][ks].PA, xb.PX) MFMA_B16_(a11, wf[1][ks].PA, xb.PX)
#define C0_TERM2_(PA, PX) MFMA_B16_(a00, wf[0][ks].PA, xa.PX) MFMA_B16_(a10, wf[1][ks].PA, xa.PX)
#pragma unroll 1
    for (int ti = 0; ti + 1 < C0_TT / 16; ti += 2) {
      f32x4 a00 = {0.f, 0.f, 0.f, 0.f}, a10 = a00, a01 = a00, a11 = a00;
      const int off = st * C0_JP * (16 * ti + r) + 8 * q;          // this lane's fragment: output step 16 ti + r, k = 8 q .. 8 q + 7 (+ 32 ks)
#pragma unroll
      for (int ks = 0; ks < C0_NKS; ++ks) {
        const HML8 xa = xfrag(off, ks), xb = xfrag(off + st * C0_JP * 16, ks);
        C0_TERM4_(lo, hi) C0_TERM4_(hi, lo) C0_TERM4_(mid, mid) C0_TERM4_(mid, hi) C0_TERM4_(hi, mid) C0_TERM4_(hi, hi)
      }
      store_y(ti, a00, a10);
      store_y(ti + 1, a01, a11);
    }
    if constexpr ((C0_TT / 16) % 2 == 1) {
      constexpr int ti = C0_TT / 16 - 1;
      f32x4 a00 = {0.f, 0.f, 0.f, 0.f}, a10 = a00;
      const int off = st * C0_JP * (16 * ti + r) + 8 * q;
#pragma unroll
      for (int ks = 0; ks < C0_NKS; ++ks) {
        const HML8 xa = xfrag(off, ks);
        C0_TERM2_(lo, hi) C0_TERM2_(hi, lo) C0_TERM2_(mid, mid) C0_TERM2_(mid, hi) C0_TERM2_(hi, mid) C0_TERM2_(hi, hi)
      }
      store_y(ti, a00, a10);
    }
#undef C0_TERM4_
#undef C0_TERM2_
    if (part) {      // fold the 16 step lanes (same q = same channels); lane r == 0 writes the tile's sums
#pragma unroll
      for (int e = 0; e < 8; ++e) {
#pragma unroll
        for (int o = 8; o > 0; o >>= 1) {
          sa[e] += __shfl_xor(sa[e], o);
          qa[e] += __shfl_xor(qa[e], o);
        }
      }
      if (r == 0) {
        float* const row = part + (long)tile * 2 * C;
        const int c = ch0 + 4 * q;
        if (c < C) {
          *reinterpret_cast<float4*>(row + c) = make_float4(sa[0], sa[1], sa[2], sa[3]);
          *reinterpret_cast<float4*>(row + C + c) = make_float4(qa[0], qa[1], qa[2], qa[3]);
        }
        if (c + 16 < C) {
          *reinterpret_cast<float4*>(row + c + 16) = make_float4(sa[4], sa[5], sa[6], sa[7]);
          *reinterpret_cast<float4*>(row + C + c + 16) = make_float4(qa[4], qa[5], qa[6], qa[7]);
        }
      }
    }
  }
}
// The weight gradients of every layer back in the parameters' layouts, ONE launch behind the layer loop (blockIdx.y = layer):
//   layer 0 (direct path): dW[c][i * kf + j] += dWg[c * ldg + i * JG + j] -- taken over the padded window columns
//   layer i >= 1:          dW[co][ci][kt]    += dWr[co][kt * Ci + ci]
struct UnpackJobs {
  int n;
  const float* src[ASTK_MAX_CNN_LAYERS];
  float* dW[ASTK_MAX_CNN_LAYERS];
  int Co[ASTK_MAX_CNN_LAYERS], Ci[ASTK_MAX_CNN_LAYERS], KT[ASTK_MAX_CNN_LAYERS];      // layer 0: Ci = kf
  int JG[ASTK_MAX_CNN_LAYERS], ldg[ASTK_MAX_CNN_LAYERS];                              // layer 0 only (ldg > 0 marks it)
};
__global__ void k_unpack_dw(UnpackJobs jobs, const unsigned* __restrict__ status, float* __restrict__ status_dst) {
  const int q = blockIdx.y;
  if (status_dst && blockIdx.x == 0 && q == 0 && threadIdx.x == 0) *status_dst = (float)(*status);     // (astk_cnn_desc.status_dst)
  const float* const src = jobs.src[q];
  float* const dW = jobs.dW[q];
  const int Co = jobs.Co[q], Ci = jobs.Ci[q], KT = jobs.KT[q], JG = jobs.JG[q], ldg = jobs.ldg[q];
  const long n = (long)Co * Ci * KT;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    if (ldg > 0) {
      const int j = (int)(i % Ci), t = (int)((i / Ci) % KT), c = (int)(i / ((long)Ci * KT));
      dW[i] += src[(long)c * ldg + t * JG + j];
    } else {
      const int k = (int)(i % KT), ci = (int)((i / KT) % Ci), co = (int)(i / ((long)Ci * KT));
      dW[i] += src[((long)co * KT + k) * Ci + ci];
    }
  }
}

// Wr[co][kt*Ci + ci] = W[co][ci][kt]   (W is (Co,Ci,KT,1))
__global__ __launch_bounds__(256) void k_repack_w(const float* __restrict__ W, float* __restrict__ Wr, int Co, int Ci, int KT, unsigned long long* amax,
                                                  const unsigned long long* fold_src, unsigned long long* fold_dst) {
  __shared__ float red4[4];
  if (fold_src && blockIdx.x == 0 && threadIdx.x < 64) amax_compact(fold_src, fold_dst);     // (the layer below's activation maximum, see CnnPlan)
  const long n = (long)Co * Ci * KT;
  float m = 0.f;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    const int ci = (int)(i % Ci);
    const int k = (int)((i / Ci) % KT);
    const int co = (int)(i / ((long)Ci * KT));
    const float v = W[((long)co * Ci + ci) * KT + k];
    Wr[i] = v;
    m = fmaxf(m, fabsf(v));
  }
  if (amax) amax_emit_block(amax, m, red4);
}
// dgrad phase weights: Wd[ci][a*Co + co] = W[co][ci][kt = r + st*(na-1-a)], the stride phases of one grouped dgrad launch in ONE launch
// (blockIdx.y = phase; a launch per phase cost ~5 us of stream time each)
struct PhaseWJobs {
  int n;
  float* wd[GEMM_GROUP_MAX];
  int r[GEMM_GROUP_MAX], na[GEMM_GROUP_MAX];
  unsigned long long* amax[GEMM_GROUP_MAX];
};
__global__ __launch_bounds__(256) void k_phase_w(const float* __restrict__ W, PhaseWJobs jobs, int Co, int Ci, int KT, int st,
                                                 const unsigned long long* fold_src, unsigned long long* fold_dst) {
  __shared__ float red4[4];
  if (fold_src && blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x < 64) amax_compact(fold_src, fold_dst);     // (the padded dY's maximum, see CnnPlan)
  const int j = blockIdx.y, r = jobs.r[j], na = jobs.na[j];
  float* const Wd = jobs.wd[j];
  const long n = (long)Ci * na * Co;
  float m = 0.f;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    const int co = (int)(i % Co);
    const int a = (int)((i / Co) % na);
    const int ci = (int)(i / ((long)Co * na));
    const int k = r + st * (na - 1 - a);
    const float v = W[((long)co * Ci + ci) * KT + k];
    Wd[i] = v;
    m = fmaxf(m, fabsf(v));
  }
  if (jobs.amax[j]) {      // (amax_emit_block shards by blockIdx.x + blockIdx.y * gridDim.x: fine for a plain slot)
    amax_emit_block(jobs.amax[j], m, red4);
  }
}

__device__ __forceinline__ void colstats_finalize(int C, double* __restrict__ stat, const BnFinalize& fin);
// column sums of Y and Y^2 ( -> double atomics ), colreduce_block skeleton.  fin.ctr: the block that arrives LAST (every block drains its
// atomics, then bumps the counter) turns the sums into scale / shift and the running statistics -- k_bn_finalize without its launch.
__global__ __launch_bounds__(256) void k_colstats(const float* __restrict__ Y, int rows, int C, double* __restrict__ stat, BnFinalize fin) {
  colreduce_block<2>(
      rows, C,
      [&](int r, int c, float4* a) {
        const float4 v = *reinterpret_cast<const float4*>(Y + (long)r * C + c);
        a[0].x += v.x; a[0].y += v.y; a[0].z += v.z; a[0].w += v.w;
        a[1].x += v.x * v.x; a[1].y += v.y * v.y; a[1].z += v.z * v.z; a[1].w += v.w * v.w;
      },
      [&](int col, int st, float v) { atomicAdd(&stat[st * C + col], (double)v); });
  colstats_finalize(C, stat, fin);
}
// ... from rows of PRE-SUMMED statistics [rows][2][C] (k_conv0_fwd_x3's per-tile sums): column sums of a (rows, 2 C) matrix land in stat's
// [2][C] layout as they are
__global__ __launch_bounds__(256) void k_colstats_tiles(const float* __restrict__ part, int rows, int C, double* __restrict__ stat, BnFinalize fin) {
  colreduce_block<1>(
      rows, 2 * C,
      [&](int r, int c, float4* a) {
        const float4 v = *reinterpret_cast<const float4*>(part + (long)r * 2 * C + c);
        a[0].x += v.x; a[0].y += v.y; a[0].z += v.z; a[0].w += v.w;
      },
      [&](int col, int st, float v) { atomicAdd(&stat[col], (double)v); });
  colstats_finalize(C, stat, fin);
}
__device__ __forceinline__ void colstats_finalize(int C, double* __restrict__ stat, const BnFinalize& fin) {
  if (!fin.ctr) return;
  __shared__ int last;
  // this thread's atomics have been performed (at the memory side: device-scope atomics; the last block reads them back with device-scope
  // loads).  NOT __threadfence(): its L2 write-back + invalidate in every wave of 256 blocks cost the kernel 5-15 us.
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (threadIdx.x == 0) last = __hip_atomic_fetch_add(fin.ctr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == gridDim.x * gridDim.y - 1;
  __syncthreads();
  if (!last) return;
  for (int c = threadIdx.x; c < C; c += 256)
    bn_finalize_channel(c, C, __hip_atomic_load(&stat[c], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT),
                        __hip_atomic_load(&stat[C + c], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT), fin, 1);
}

// zero only the pad rows of a time-padded channels-last buffer [(group)][padF + Tn + padB][C]: the interior is rewritten every step.
// (A grid-stride loop every thread of the CALLING kernel runs: the kernels that write the interior rows zero the pads on the way out --
//  a launch of its own cost 5 us twice per step.)
__device__ __forceinline__ void zero_pad_rows(float* __restrict__ buf, int groups, int Tn, int padF, int padB, int C) {
  const int np = padF + padB;
  const long n4 = (long)groups * np * (C / 4);
  const long nthreads = (long)gridDim.x * gridDim.y * blockDim.x;
  for (long i = ((long)blockIdx.y * gridDim.x + blockIdx.x) * blockDim.x + threadIdx.x; i < n4; i += nthreads) {
    const int c4 = (int)(i % (C / 4));
    const long gp = i / (C / 4);
    const int pr = (int)(gp % np);
    const long grp = gp / np;
    const int row = pr < padF ? pr : Tn + pr;     // rows [0, padF) and [padF + Tn, padF + Tn + padB)
    reinterpret_cast<float4*>(buf + (grp * (padF + Tn + padB) + row) * C)[c4] = make_float4(0.f, 0.f, 0.f, 0.f);
  }
}
__global__ __launch_bounds__(256) void k_zero_pads(float* __restrict__ buf, int groups, int Tn, int padF, int padB, int C) {
  const int np = padF + padB;
  const long n4 = (long)groups * np * (C / 4);
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n4; i += (long)gridDim.x * blockDim.x) {
    const int c4 = (int)(i % (C / 4));
    const long gp = i / (C / 4);
    const int pr = (int)(gp % np);
    const long grp = gp / np;
    const int row = pr < padF ? pr : Tn + pr;     // rows [0, padF) and [padF + Tn, padF + Tn + padB)
    reinterpret_cast<float4*>(buf + (grp * (padF + Tn + padB) + row) * C)[c4] = make_float4(0.f, 0.f, 0.f, 0.f);
  }
}

// (the launch of its own: evaluation mode, and training with a statistics exchange between the sums and their use)
__global__ void k_bn_finalize(const double* __restrict__ stat, int C, BnFinalize fin, int train) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  bn_finalize_channel(c, C, train ? stat[c] : 0.0, train ? stat[C + c] : 0.0, fin, train);
}

// cnn_config.bn = false (seq2seq.py:43-57: Convolution2D with bias, no BatchNormalization): the same scale / shift slots the ReLU
// kernels read, filled with scale 1 and shift = bias (mean 0, inv_std 1 for the backward's x_hat, which it then ignores)
__global__ void k_bias_affine(const float* __restrict__ bias, int C, float* __restrict__ bn) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  bn[c] = 0.f;
  bn[C + c] = 1.f;
  bn[2 * C + c] = 1.f;
  bn[3 * C + c] = bias[c];
}
__global__ void k_bias_grad(const double* __restrict__ stat, int C, float* __restrict__ dbias) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c < C) dbias[c] += (float)stat[c];
}

// dst[prow(m)][c] = relu(Y[m][c]*scale + shift), prow(m) = (m/Tn)*(Tn+2*pad) + pad + m%Tn  (float4 over channels)
__global__ __launch_bounds__(256) void k_bn_relu_rows(const float* __restrict__ Y, const float* __restrict__ bn, float* __restrict__ dst, int rows, int C,
                                                      int Tn, int pad, unsigned long long* amax) {
  __shared__ float red4[4];
  const long n4 = (long)rows * C / 4;
  const int C4 = C / 4;
  float mx = 0.f;           // (post-ReLU values: non-negative)
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n4; i += (long)gridDim.x * blockDim.x) {
    const int m = (int)(i / C4), c = (int)(i % C4) * 4;
    const float4 y = *reinterpret_cast<const float4*>(Y + (long)m * C + c);
    const float4 sc = *reinterpret_cast<const float4*>(bn + 2 * C + c);
    const float4 sh = *reinterpret_cast<const float4*>(bn + 3 * C + c);
    float4 o;
    o.x = fmaxf(y.x * sc.x + sh.x, 0.f); o.y = fmaxf(y.y * sc.y + sh.y, 0.f);
    o.z = fmaxf(y.z * sc.z + sh.z, 0.f); o.w = fmaxf(y.w * sc.w + sh.w, 0.f);
    const long pr = (long)(m / Tn) * (Tn + 2 * pad) + pad + (m % Tn);
    *reinterpret_cast<float4*>(dst + pr * C + c) = o;
    mx = fmaxf(fmaxf(mx, fmaxf(o.x, o.y)), fmaxf(o.z, o.w));
  }
  if (pad > 0) zero_pad_rows(dst, rows / Tn, Tn, pad, pad, C);
  if (amax) amax_emit_block(amax, mx, red4);
}

// out[t][b][c*F+f] = relu(bn(Y[(b,f,t)][c]))  -- one block per (t, b); LDS re-orders (f,c) -> (c,f)
__global__ __launch_bounds__(256) void k_bn_relu_to_seq(const float* __restrict__ Y, const float* __restrict__ bn, float* __restrict__ out,
                                                        int B, int F, int Tn, int C, unsigned long long* amax, void* zero_from, size_t zero_bytes) {
  extern __shared__ float tile[];   // [C*F]
  __shared__ float red4[4];
  zero_region(zero_from, zero_bytes);
  const int t = blockIdx.x, b = blockIdx.y;
  const int n = C * F;
  float mx = 0.f;
  for (int i = threadIdx.x; i < n; i += blockDim.x) {
    const int f = i / C, c = i % C;
    const float y = Y[(((long)b * F + f) * Tn + t) * C + c];
    const float v = fmaxf(y * bn[2 * C + c] + bn[3 * C + c], 0.f);
    tile[c * F + f] = v;
    mx = fmaxf(mx, v);
  }
  __syncthreads();
  float* o = out + ((long)t * B + b) * n;
  for (int i = threadIdx.x; i < n; i += blockDim.x) o[i] = tile[i];
  if (amax) amax_emit_block(amax, mx, red4);
}
// G[(b,f,t)][c] = d_out[t][b][c*F+f]
__global__ __launch_bounds__(256) void k_seq_to_rows(const float* __restrict__ d_out, float* __restrict__ G, int B, int F, int Tn, int C) {
  extern __shared__ float tile[];
  const int t = blockIdx.x, b = blockIdx.y;
  const int n = C * F;
  const float* src = d_out + ((long)t * B + b) * n;
  for (int i = threadIdx.x; i < n; i += blockDim.x) tile[i] = src[i];
  __syncthreads();
  for (int i = threadIdx.x; i < n; i += blockDim.x) {
    const int f = i / C, c = i % C;
    G[(((long)b * F + f) * Tn + t) * C + c] = tile[c * F + f];
  }
}

// Max-pool between a layer's convolution and its BatchNorm (enc_dec.py:444-456: F.max_pooling_nd, window = stride, cover_all): pooled row
// (b, f', t') = max over the conv rows (b, f'*pf .. , t'*pt ..) that exist; idx keeps the winning conv row (first maximum in the window's
// (time, frequency) scan order, like Chainer's argmax over the flattened window).  float4 over channels.
__global__ __launch_bounds__(256) void k_maxpool(const float* __restrict__ YC, float* __restrict__ Y, int32_t* __restrict__ idx, int B, int Fc, int Tc,
                                                 int Fn, int Tn, int pf, int pt, int C) {
  const int C4 = C / 4;
  const long n4 = (long)B * Fn * Tn * C4;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n4; i += (long)gridDim.x * blockDim.x) {
    const int c = (int)(i % C4) * 4;
    const long rp = i / C4;
    const int t1 = (int)(rp % Tn), f1 = (int)((rp / Tn) % Fn), b = (int)(rp / ((long)Tn * Fn));
    float4 m = make_float4(-INFINITY, -INFINITY, -INFINITY, -INFINITY);
    int i0 = -1, i1 = -1, i2 = -1, i3 = -1;
    for (int t = t1 * pt; t < min(Tc, (t1 + 1) * pt); ++t)
      for (int f = f1 * pf; f < min(Fc, (f1 + 1) * pf); ++f) {
        const int rc = (b * Fc + f) * Tc + t;
        const float4 y = *reinterpret_cast<const float4*>(YC + (long)rc * C + c);
        if (y.x > m.x || i0 < 0) { m.x = y.x; i0 = rc; }
        if (y.y > m.y || i1 < 0) { m.y = y.y; i1 = rc; }
        if (y.z > m.z || i2 < 0) { m.z = y.z; i2 = rc; }
        if (y.w > m.w || i3 < 0) { m.w = y.w; i3 = rc; }
      }
    *reinterpret_cast<float4*>(Y + rp * C + c) = m;
    *reinterpret_cast<int4*>(idx + rp * C + c) = make_int4(i0, i1, i2, i3);
  }
}
// its backward: the gradient of a pooled element goes to the conv row that won, every other conv row gets zero; written straight into the
// time-padded dY layout the weight- and input-gradient products read (a gather over conv rows: windows do not overlap, no atomics)
__global__ __launch_bounds__(256) void k_unpool(const float* __restrict__ dYp, const int32_t* __restrict__ idx, float* __restrict__ dY, int B, int Fc,
                                                int Tc, int Fn, int Tn, int pf, int pt, int C, int padF, int padB, unsigned long long* amax) {
  __shared__ float red4[4];
  const int C4 = C / 4;
  const long n4 = (long)B * Fc * Tc * C4;
  float mx = 0.f;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n4; i += (long)gridDim.x * blockDim.x) {
    const int c = (int)(i % C4) * 4;
    const long rc = i / C4;
    const int t = (int)(rc % Tc), f = (int)((rc / Tc) % Fc), b = (int)(rc / ((long)Tc * Fc));
    const long rp = ((long)b * Fn + f / pf) * Tn + t / pt;
    const int4 w = *reinterpret_cast<const int4*>(idx + rp * C + c);
    const float4 g = *reinterpret_cast<const float4*>(dYp + rp * C + c);
    float4 v;
    v.x = w.x == (int)rc ? g.x : 0.f; v.y = w.y == (int)rc ? g.y : 0.f;
    v.z = w.z == (int)rc ? g.z : 0.f; v.w = w.w == (int)rc ? g.w : 0.f;
    const long pr = (rc / Tc) * (Tc + padF + padB) + padF + t;
    *reinterpret_cast<float4*>(dY + pr * C + c) = v;
    mx = fmaxf(fmaxf(mx, fmaxf(fabsf(v.x), fabsf(v.y))), fmaxf(fabsf(v.z), fabsf(v.w)));
  }
  if (amax) amax_emit_block(amax, mx, red4);
}

// backward statistics: stat[c] = sum g, stat[C+c] = sum g*xhat, g = G*(bn(Y)>0)   (colreduce_block skeleton)
__global__ __launch_bounds__(256) void k_bn_bwd_stats(const float* __restrict__ Y, const float* __restrict__ G, const float* __restrict__ bn,
                                                      int rows, int C, double* __restrict__ stat) {
  const int q = blockIdx.x * min((C + 3) >> 2, COLREDUCE_CL) + threadIdx.x % min((C + 3) >> 2, COLREDUCE_CL);
  const int cc = min(q * 4, C - 4);   // C % 4 == 0 (checked by the launcher)
  const float4 mean = *reinterpret_cast<const float4*>(bn + cc), inv = *reinterpret_cast<const float4*>(bn + C + cc);
  const float4 sc = *reinterpret_cast<const float4*>(bn + 2 * C + cc), sh = *reinterpret_cast<const float4*>(bn + 3 * C + cc);
  colreduce_block<2>(
      rows, C,
      [&](int r, int c, float4* a) {
        const float4 y = *reinterpret_cast<const float4*>(Y + (long)r * C + c);
        const float4 gr = *reinterpret_cast<const float4*>(G + (long)r * C + c);
        const float g0 = (y.x * sc.x + sh.x > 0.f) ? gr.x : 0.f, g1 = (y.y * sc.y + sh.y > 0.f) ? gr.y : 0.f;
        const float g2 = (y.z * sc.z + sh.z > 0.f) ? gr.z : 0.f, g3 = (y.w * sc.w + sh.w > 0.f) ? gr.w : 0.f;
        a[0].x += g0; a[0].y += g1; a[0].z += g2; a[0].w += g3;
        a[1].x += g0 * (y.x - mean.x) * inv.x; a[1].y += g1 * (y.y - mean.y) * inv.y;
        a[1].z += g2 * (y.z - mean.z) * inv.z; a[1].w += g3 * (y.w - mean.w) * inv.w;
      },
      [&](int col, int st, float v) { atomicAdd(&stat[st * C + col], (double)v); });
}
// dY[prow(m)][c] = scale*(g - (xhat*dgamma + dbeta)/rows) ; also accumulates dgamma/dbeta (block 0)
__global__ __launch_bounds__(256) void k_bn_bwd_apply(const float* __restrict__ Y, const float* __restrict__ G, const float* __restrict__ bn,
                               const double* __restrict__ stat, float* __restrict__ dY, int rows, int C, int Tn, int padF, int padB,
                               float* __restrict__ dgamma, float* __restrict__ dbeta, float invm, unsigned long long* amax) {
  __shared__ float red4[4];
  const int C4 = C / 4;   // C % 4 == 0 (checked by the launcher)
  const long n4 = (long)rows * C4;
  float mx = 0.f;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n4; i += (long)gridDim.x * blockDim.x) {
    const int m = (int)(i / C4), c = (int)(i % C4) * 4;
    const float4 mean = *reinterpret_cast<const float4*>(bn + c), inv = *reinterpret_cast<const float4*>(bn + C + c);
    const float4 sc = *reinterpret_cast<const float4*>(bn + 2 * C + c), sh = *reinterpret_cast<const float4*>(bn + 3 * C + c);
    const float4 y = *reinterpret_cast<const float4*>(Y + i * 4), gr = *reinterpret_cast<const float4*>(G + i * 4);
    const float dg0 = (float)stat[C + c], dg1 = (float)stat[C + c + 1], dg2 = (float)stat[C + c + 2], dg3 = (float)stat[C + c + 3];
    const float db0 = (float)stat[c], db1 = (float)stat[c + 1], db2 = (float)stat[c + 2], db3 = (float)stat[c + 3];
    float4 v;
    v.x = sc.x * (((y.x * sc.x + sh.x > 0.f) ? gr.x : 0.f) - ((y.x - mean.x) * inv.x * dg0 + db0) * invm);
    v.y = sc.y * (((y.y * sc.y + sh.y > 0.f) ? gr.y : 0.f) - ((y.y - mean.y) * inv.y * dg1 + db1) * invm);
    v.z = sc.z * (((y.z * sc.z + sh.z > 0.f) ? gr.z : 0.f) - ((y.z - mean.z) * inv.z * dg2 + db2) * invm);
    v.w = sc.w * (((y.w * sc.w + sh.w > 0.f) ? gr.w : 0.f) - ((y.w - mean.w) * inv.w * dg3 + db3) * invm);
    const long pr = (long)(m / Tn) * (Tn + padF + padB) + padF + (m % Tn);
    *reinterpret_cast<float4*>(dY + pr * C + c) = v;
    mx = fmaxf(fmaxf(mx, fmaxf(fabsf(v.x), fabsf(v.y))), fmaxf(fabsf(v.z), fabsf(v.w)));
  }
  if (padF + padB > 0) zero_pad_rows(dY, rows / Tn, Tn, padF, padB, C);
  if (amax) amax_emit_block(amax, mx, red4);
  if (blockIdx.x == 0 && dgamma)
    for (int c = threadIdx.x; c < C; c += blockDim.x) {
      dgamma[c] += (float)stat[C + c];
      dbeta[c] += (float)stat[c];
    }
}
// ---- The LAST layer's ReLU + BatchNorm backward straight from d_out, the (T'', B, C*F') gradient of the LSTM frames.  k_seq_to_rows used to
// re-order it into rows first (a pass of its own: 79 MB read + 79 MB written, 48 us of the train step); here a block owns SEQ_CH channels, takes
// the contiguous C*F piece of SEQ_PP (t, b) pairs at a time through LDS (re-ordered (c, f) -> [f][c] on the way in) and works on the
// F rows (b, f, t) of each pair -- Y rows and dY rows are touched 256 bytes at a time, like the row kernels do.
constexpr int SEQ_CH = 64;            // channels per block (16 lanes x 4)
constexpr int SEQ_LD = SEQ_CH + 4;    // LDS row stride (floats): consecutive f rows four banks apart
constexpr int SEQ_UNR = 4;            // rows a thread works on per block iteration: 16 row groups x 4 = 64 (pair, f) slots
static int seq_pp(int F) { return 64 / F; }      // (t, b) pairs per block iteration: pp * F <= 64 slots
static size_t seq_tile_bytes(int F) { return std::max((size_t)seq_pp(F) * F * SEQ_LD * sizeof(float), (size_t)2 * 256 * sizeof(float4)); }
static bool seq_bwd_applicable(int C, int F, long npairs) { return (C % SEQ_CH) == 0 && F >= 1 && F <= 16 && npairs < (1L << 30); }
// tile[pl][f][c - c0] = d_out[pair p0 + pl][c * F + f] for the block's channels (pairs past the end: the last pair again, never used)
// in two halves: the 16-byte loads of the iteration's d_out piece (at most SEQ_TL per thread: pp * 16 F <= 1024), issued together with the Y
// rows in front of the barrier, and the re-ordering stores into the tile behind it
constexpr int SEQ_TL = 4;
__device__ __forceinline__ void seq_tile_fetch(const float* __restrict__ d_out, float4 (&v)[SEQ_TL], int p0, int npairs, int PP, int c0, int C, int F) {
  const int per_pair4 = SEQ_CH * F / 4, n4 = PP * per_pair4;
#pragma unroll
  for (int u = 0; u < SEQ_TL; ++u) {
    const int i = min((int)threadIdx.x + 256 * u, n4 - 1);
    const int pl = i / per_pair4, q4 = i % per_pair4;
    const long p = min(p0 + pl, npairs - 1);
    v[u] = *reinterpret_cast<const float4*>(d_out + (p * C + c0) * F + 4 * q4);
  }
}
__device__ __forceinline__ void seq_tile_store(float* tile, const float4 (&v)[SEQ_TL], int PP, int F) {
  const int per_pair4 = SEQ_CH * F / 4, n4 = PP * per_pair4;
#pragma unroll
  for (int u = 0; u < SEQ_TL; ++u) {
    const int i = (int)threadIdx.x + 256 * u;
    if (i < n4) {
      const int pl = i / per_pair4, q4 = i % per_pair4;
      float* tp = tile + pl * F * SEQ_LD;
      int c = (4 * q4) / F, f = (4 * q4) % F;
      tp[f * SEQ_LD + c] = v[u].x; if (++f == F) { f = 0; ++c; }
      tp[f * SEQ_LD + c] = v[u].y; if (++f == F) { f = 0; ++c; }
      tp[f * SEQ_LD + c] = v[u].z; if (++f == F) { f = 0; ++c; }
      tp[f * SEQ_LD + c] = v[u].w;
    }
  }
}
// A thread's four (pair, f) slots are the same in every block iteration (slot = row group + 16 u): set up once.
struct SeqSlots {
  int pl[SEQ_UNR], f[SEQ_UNR], off[SEQ_UNR];
  bool ok[SEQ_UNR];
  __device__ __forceinline__ void init(int g16, int n, int F, int cl) {
#pragma unroll
    for (int u = 0; u < SEQ_UNR; ++u) {
      const int slot = min(g16 + 16 * u, n - 1);
      ok[u] = g16 + 16 * u < n;
      pl[u] = slot / F; f[u] = slot % F;
      off[u] = slot * SEQ_LD + cl;
    }
  }
};
// out[t][b][c * F + f] = relu(bn(Y[(b, f, t)][c])), the forward mirror of the kernels below: a block owns SEQ_CH channels, reads the F conv rows of
// PP (t, b) pairs 256 bytes at a time (16-byte loads), re-orders [f][c] -> [c][f] through LDS and writes each pair's piece of the frames as one
// contiguous run of 16-byte stores (k_bn_relu_to_seq moves 4 bytes per lane both ways: 43 us for 158 MB)
__global__ __launch_bounds__(256) void k_bn_relu_to_seq_tiled(const float* __restrict__ Y, const float* __restrict__ bn, float* __restrict__ out, int B, int F,
                                                              int Tn, int C, int PP, unsigned long long* amax, void* zero_from, size_t zero_bytes) {
  extern __shared__ __attribute__((aligned(16))) float seq_tile[];       // [PP][SEQ_CH * F]: a pair's piece in the frames' own order
  __shared__ float red4[4];
  zero_region(zero_from, zero_bytes);      // what the backward call accumulates into (every user of the forward statistics is behind us in the stream)
  const int c0 = blockIdx.x * SEQ_CH, cl = 4 * (threadIdx.x & 15), c = c0 + cl, g16 = threadIdx.x >> 4;
  const int npairs = Tn * B, piece = SEQ_CH * F;
  const float4 sc = *reinterpret_cast<const float4*>(bn + 2 * C + c), sh = *reinterpret_cast<const float4*>(bn + 3 * C + c);
  SeqSlots sl;
  sl.init(g16, PP * F, F, cl);
  float mx = 0.f;
  for (int p0 = blockIdx.y * PP; p0 < npairs; p0 += gridDim.y * PP) {
    float4 y[SEQ_UNR];
#pragma unroll
    for (int u = 0; u < SEQ_UNR; ++u) {
      const int pp = min(p0 + sl.pl[u], npairs - 1);
      const int t = pp / B, b = pp - t * B;
      y[u] = *reinterpret_cast<const float4*>(Y + (((long)b * F + sl.f[u]) * Tn + t) * C + c);
    }
    __syncthreads();                 // the previous iteration's piece has been written out
#pragma unroll
    for (int u = 0; u < SEQ_UNR; ++u) {
      if (sl.ok[u]) {
        float* tp = seq_tile + sl.pl[u] * piece + cl * F + sl.f[u];
        const float v0 = fmaxf(y[u].x * sc.x + sh.x, 0.f), v1 = fmaxf(y[u].y * sc.y + sh.y, 0.f);
        const float v2 = fmaxf(y[u].z * sc.z + sh.z, 0.f), v3 = fmaxf(y[u].w * sc.w + sh.w, 0.f);
        tp[0] = v0; tp[F] = v1; tp[2 * F] = v2; tp[3 * F] = v3;
        if (p0 + sl.pl[u] < npairs) mx = fmaxf(fmaxf(mx, fmaxf(v0, v1)), fmaxf(v2, v3));
      }
    }
    __syncthreads();
    const int per_pair4 = piece / 4;
    for (int i = threadIdx.x; i < PP * per_pair4; i += 256) {
      const int pl = i / per_pair4, q4 = i - pl * per_pair4;
      if (p0 + pl < npairs)
        *reinterpret_cast<float4*>(out + ((long)(p0 + pl) * C + c0) * F + 4 * q4) = *reinterpret_cast<const float4*>(seq_tile + pl * piece + 4 * q4);
    }
  }
  if (amax) amax_emit_block(amax, mx, red4);
}
// stat[c] += sum g, stat[C + c] += sum g * xhat over all rows, g = d_out * (bn(Y) > 0): grid (C / SEQ_CH, row slabs)
__global__ __launch_bounds__(256) void k_bn_bwd_stats_seq(const float* __restrict__ Y, const float* __restrict__ d_out, const float* __restrict__ bn,
                                                          int B, int F, int Tn, int C, int PP, double* __restrict__ stat) {
  extern __shared__ __attribute__((aligned(16))) float seq_tile[];
  const int c0 = blockIdx.x * SEQ_CH, cl = 4 * (threadIdx.x & 15), c = c0 + cl, g16 = threadIdx.x >> 4;
  const int npairs = Tn * B;
  const float4 mean = *reinterpret_cast<const float4*>(bn + c), inv = *reinterpret_cast<const float4*>(bn + C + c);
  const float4 sc = *reinterpret_cast<const float4*>(bn + 2 * C + c), sh = *reinterpret_cast<const float4*>(bn + 3 * C + c);
  float4 a0 = make_float4(0.f, 0.f, 0.f, 0.f), a1 = a0;
  SeqSlots sl;
  sl.init(g16, PP * F, F, cl);
  for (int p0 = blockIdx.y * PP; p0 < npairs; p0 += gridDim.y * PP) {
    // the rows of Y first (they do not wait for the tile): both streams of the iteration are in flight together
    float4 y[SEQ_UNR];
    float w[SEQ_UNR];
#pragma unroll
    for (int u = 0; u < SEQ_UNR; ++u) {          // (clamped, unconditional loads; a slot past the end counts with weight 0)
      const int pp = min(p0 + sl.pl[u], npairs - 1);
      w[u] = (sl.ok[u] && p0 + sl.pl[u] < npairs) ? 1.f : 0.f;
      const int t = pp / B, b = pp - t * B;
      y[u] = *reinterpret_cast<const float4*>(Y + (((long)b * F + sl.f[u]) * Tn + t) * C + c);
    }
    float4 tv[SEQ_TL];
    seq_tile_fetch(d_out, tv, p0, npairs, PP, c0, C, F);
    __syncthreads();
    seq_tile_store(seq_tile, tv, PP, F);
    __syncthreads();
#pragma unroll
    for (int u = 0; u < SEQ_UNR; ++u) {
      const float4 gr = *reinterpret_cast<const float4*>(seq_tile + sl.off[u]);
      const float g0 = (y[u].x * sc.x + sh.x > 0.f) ? gr.x * w[u] : 0.f, g1 = (y[u].y * sc.y + sh.y > 0.f) ? gr.y * w[u] : 0.f;
      const float g2 = (y[u].z * sc.z + sh.z > 0.f) ? gr.z * w[u] : 0.f, g3 = (y[u].w * sc.w + sh.w > 0.f) ? gr.w * w[u] : 0.f;
      a0.x += g0; a0.y += g1; a0.z += g2; a0.w += g3;
      a1.x += g0 * (y[u].x - mean.x) * inv.x; a1.y += g1 * (y[u].y - mean.y) * inv.y;
      a1.z += g2 * (y[u].z - mean.z) * inv.z; a1.w += g3 * (y[u].w - mean.w) * inv.w;
    }
  }
  __syncthreads();
  float4* red = reinterpret_cast<float4*>(seq_tile);
  red[threadIdx.x] = a0;
  red[256 + threadIdx.x] = a1;
  __syncthreads();
  if (g16 == 0) {
    for (int k = 1; k < 16; ++k) {
      const float4 v0 = red[k * 16 + threadIdx.x], v1 = red[256 + k * 16 + threadIdx.x];
      a0.x += v0.x; a0.y += v0.y; a0.z += v0.z; a0.w += v0.w;
      a1.x += v1.x; a1.y += v1.y; a1.z += v1.z; a1.w += v1.w;
    }
    atomicAdd(&stat[c], (double)a0.x); atomicAdd(&stat[c + 1], (double)a0.y); atomicAdd(&stat[c + 2], (double)a0.z); atomicAdd(&stat[c + 3], (double)a0.w);
    atomicAdd(&stat[C + c], (double)a1.x); atomicAdd(&stat[C + c + 1], (double)a1.y); atomicAdd(&stat[C + c + 2], (double)a1.z); atomicAdd(&stat[C + c + 3], (double)a1.w);
  }
}
// dY[prow(b, f, t)][c] = scale * (g - (xhat * dgamma + dbeta) / rows), the k_bn_bwd_apply of the last layer with g read from d_out
__global__ __launch_bounds__(256) void k_bn_bwd_apply_seq(const float* __restrict__ Y, const float* __restrict__ d_out, const float* __restrict__ bn,
                                                          const double* __restrict__ stat, float* __restrict__ dY, int B, int F, int Tn, int C, int PP,
                                                          int padF, int padB, float* __restrict__ dgamma, float* __restrict__ dbeta, float invm,
                                                          unsigned long long* amax) {
  extern __shared__ __attribute__((aligned(16))) float seq_tile[];
  __shared__ float red4[4];
  const int c0 = blockIdx.x * SEQ_CH, cl = 4 * (threadIdx.x & 15), c = c0 + cl, g16 = threadIdx.x >> 4;
  const int npairs = Tn * B;
  const float4 mean = *reinterpret_cast<const float4*>(bn + c), inv = *reinterpret_cast<const float4*>(bn + C + c);
  const float4 sc = *reinterpret_cast<const float4*>(bn + 2 * C + c), sh = *reinterpret_cast<const float4*>(bn + 3 * C + c);
  const float dg0 = (float)stat[C + c], dg1 = (float)stat[C + c + 1], dg2 = (float)stat[C + c + 2], dg3 = (float)stat[C + c + 3];
  const float db0 = (float)stat[c], db1 = (float)stat[c + 1], db2 = (float)stat[c + 2], db3 = (float)stat[c + 3];
  SeqSlots sl;
  sl.init(g16, PP * F, F, cl);
  float mx = 0.f;
  for (int p0 = blockIdx.y * PP; p0 < npairs; p0 += gridDim.y * PP) {
    float4 y[SEQ_UNR];
    long row[SEQ_UNR];
    bool ok[SEQ_UNR];
#pragma unroll
    for (int u = 0; u < SEQ_UNR; ++u) {
      const int pp = min(p0 + sl.pl[u], npairs - 1);
      ok[u] = sl.ok[u] && p0 + sl.pl[u] < npairs;
      const int t = pp / B, b = pp - t * B;
      row[u] = ((long)b * F + sl.f[u]) * (Tn + padF + padB) + padF + t;
      y[u] = *reinterpret_cast<const float4*>(Y + (((long)b * F + sl.f[u]) * Tn + t) * C + c);
    }
    float4 tv[SEQ_TL];
    seq_tile_fetch(d_out, tv, p0, npairs, PP, c0, C, F);
    __syncthreads();
    seq_tile_store(seq_tile, tv, PP, F);
    __syncthreads();
#pragma unroll
    for (int u = 0; u < SEQ_UNR; ++u) {
      const float4 gr = *reinterpret_cast<const float4*>(seq_tile + sl.off[u]);
      float4 v;      // (the same expression, term for term, as k_bn_bwd_apply)
      v.x = sc.x * (((y[u].x * sc.x + sh.x > 0.f) ? gr.x : 0.f) - ((y[u].x - mean.x) * inv.x * dg0 + db0) * invm);
      v.y = sc.y * (((y[u].y * sc.y + sh.y > 0.f) ? gr.y : 0.f) - ((y[u].y - mean.y) * inv.y * dg1 + db1) * invm);
      v.z = sc.z * (((y[u].z * sc.z + sh.z > 0.f) ? gr.z : 0.f) - ((y[u].z - mean.z) * inv.z * dg2 + db2) * invm);
      v.w = sc.w * (((y[u].w * sc.w + sh.w > 0.f) ? gr.w : 0.f) - ((y[u].w - mean.w) * inv.w * dg3 + db3) * invm);
      if (ok[u]) {
        *reinterpret_cast<float4*>(dY + row[u] * C + c) = v;
        mx = fmaxf(fmaxf(mx, fmaxf(fabsf(v.x), fabsf(v.y))), fmaxf(fabsf(v.z), fabsf(v.w)));
      }
    }
  }
  if (padF + padB > 0) zero_pad_rows(dY, B * F, Tn, padF, padB, C);
  if (amax) amax_emit_block(amax, mx, red4);
  if (blockIdx.y == 0 && dgamma && g16 == 0) {
    dgamma[c] += (float)stat[C + c]; dgamma[c + 1] += (float)stat[C + c + 1]; dgamma[c + 2] += (float)stat[C + c + 2]; dgamma[c + 3] += (float)stat[C + c + 3];
    dbeta[c] += (float)stat[c]; dbeta[c + 1] += (float)stat[c + 1]; dbeta[c + 2] += (float)stat[c + 2]; dbeta[c + 3] += (float)stat[c + 3];
  }
}
// dgamma / dbeta from the LOCAL sums (data-parallel BatchNorm: taken before the statistics are exchanged)
__global__ void k_bn_param_grads(const double* __restrict__ stat, int C, float* __restrict__ dgamma, float* __restrict__ dbeta) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  dgamma[c] += (float)stat[C + c];
  dbeta[c] += (float)stat[c];
}

#ifdef ASTK_TEST_HOOKS
// ---- test instrumentation, libastk_test.so only (tests/test_gpu_model.py: the batch-permutation property): the post-BatchNorm pre-activations of a layer,
// and a list of units whose upstream gradient is dropped by the next backward passes.  Two valid float32 evaluations of the same
// batch (another row order) can differ in the SIGN of a pre-activation that lies within rounding of the ReLU kink; the test names
// those units and checks that nothing else differs.
__global__ void k_preact(const float* __restrict__ Y, const float* __restrict__ bn, float* __restrict__ out, long n, int C) {
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    const int c = (int)(i % C);
    out[i] = Y[i] * bn[2 * C + c] + bn[3 * C + c];
  }
}
__global__ void k_kill_units(float* __restrict__ G, const int32_t* __restrict__ units, int n, int layer, int rows, int C) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n || units[3 * i] != layer) return;
  const int r = units[3 * i + 1], c = units[3 * i + 2];
  if (r >= 0 && r < rows && c >= 0 && c < C) G[(long)r * C + c] = 0.f;
}
const int32_t* g_kill_units = nullptr;
int g_kill_n = 0;
#endif

inline unsigned gridn(size_t n) {
  size_t b = (n + 255) / 256;
  if (b < 1) b = 1;
  if (b > 4096) b = 4096;
  return (unsigned)b;
}

int ksplit_for(long tiles, long K) {
  // aim for ~512 workgroups, at least 256 of K per split
  long s = 512 / (tiles > 0 ? tiles : 1);
  if (s < 1) s = 1;
  long smax = K / 256;
  if (smax < 1) smax = 1;
  if (s > smax) s = smax;
  if (s > 64) s = 64;
  return (int)s;
}

}  // namespace

int cnn_out_dims(const astk_cnn_desc* d, int* T_out, int* F_out, int* feat) {
  CnnPlan P;
  ASTK_TRY(make_plan(d, nullptr, P));
  if (T_out) *T_out = P.Tn[P.n - 1];
  if (F_out) *F_out = P.Fn[P.n - 1];
  if (feat) *feat = P.Cn[P.n - 1] * P.Fn[P.n - 1];
  return 0;
}

}  // namespace astk

using namespace astk;

extern "C" {

int astk_conv_bn_relu_out_dims(const astk_cnn_desc* d, int* T_out, int* F_out, int* feat_dim) {
  return cnn_out_dims(d, T_out, F_out, feat_dim);
}

size_t astk_conv_bn_relu_workspace_bytes(const astk_cnn_desc* d) {
  CnnPlan P;
  if (make_plan(d, nullptr, P) != 0) return 0;
  return P.bytes;
}

int astk_conv_bn_relu_fwd(const astk_cnn_desc* d, const astk_cnn_layer_params* L, const float* X, const float* noise, float* out,
                          void* ws, size_t ws_bytes, int train, void* stream) {
  return astk_conv_bn_relu_fwd_sync(d, L, X, noise, out, ws, ws_bytes, train, nullptr, nullptr, 1, stream);
}

int astk_conv_bn_relu_fwd_sync(const astk_cnn_desc* d, const astk_cnn_layer_params* L, const float* X, const float* noise, float* out,
                               void* ws, size_t ws_bytes, int train, astk_stat_exchange_fn exchange, void* user, int world,
                               void* stream) {
  hipStream_t s = (hipStream_t)stream;
  ASTK_CHECK_DESC(d, astk_cnn_desc);
  PrecScope prec_scope(d->precision, d->gemm_operands);
  GemmForwardScope forward_scope;      // split tiles of this op's products have at most two contributors (reproducible forward pass)
  gemm_amax_step_boundary(s);         // the first op of a step: no operand-maximum handle is live here
  ASTK_CHECK(world >= 1, "conv_bn_relu_fwd: world %d", world);
  if (world == 1 || d->no_bn) exchange = nullptr;
  CnnPlan P;
  ASTK_TRY(make_plan(d, ws, P));
  ASTK_CHECK(ws && ws_bytes >= P.bytes, "conv_bn_relu_fwd: workspace too small (%zu < %zu)", ws_bytes, P.bytes);
  ASTK_CHECK(X && out && L, "conv_bn_relu_fwd: null pointer");
  const int B = P.B;
  // the statistics of every layer (train) and the maximum slots the producing kernels fill: one fill
  const size_t zero_fwd = train ? P.zero_fwd_bytes : P.zero_fwd_amax_bytes;
  if (!conv0_direct(d)) ASTK_TRY(fill_zero(P.zero_fwd_from, zero_fwd, s));       // (the direct layer-0 kernel does it on its way in)
  // the backward's region: zeroed by this call's last kernel (train) -- a fill launch less in the backward call
  const bool seq_out_zeroes = train != 0;
  // (the layer-0 path is recorded now -- the backward must refuse a workspace whose forward took the other path even if this call fails
  //  half-way --, the "backward region is clean" mark only behind the LAST launch below: a forward call that fails before its last kernel
  //  was enqueued must not leave a mark that spares the backward its own zero fill; round-5 advice)
  conv0_path_record(ws, conv0_direct(d), false);
  void* const zb_from = seq_out_zeroes ? (void*)P.stat[0] : nullptr;
  bool c0_stats = false;
  if (conv0_direct(d)) {
    // ---- layer 0: direct convolution (bf16x3 on the matrix pipe), leaves XF for the weight gradient
    const int tiles_t = cdiv(P.Tc[0], C0_TT), total = B * P.Fc[0] * tiles_t, win = conv0_win_elems(d->st[0]);
    const int grid = std::min(total, 2 * device_cu_count());
    const size_t lds0 = std::max((size_t)3 * win * sizeof(unsigned short), (size_t)P.Cn[0] * P.K0 * sizeof(float));
    // the layer's BatchNorm statistics inside the kernel (and, without a statistics exchange, scale / shift and the running statistics)
    c0_stats = train && !d->no_bn && total == P.c0_tiles;      // the layer's BatchNorm statistics: per-tile sums out of the kernel
    hipLaunchKernelGGL(k_conv0_fwd_x3, dim3(grid), dim3(256), lds0, s, X, noise, L[0].W, P.YC[0], P.XF, B, P.T, P.D,
                       P.Fc[0], P.Tc[0], P.Cn[0], d->kt[0], d->kf[0], d->st[0], d->sf[0], d->pt[0], P.JG, P.xf_rows, tiles_t, total, win,
                       P.zero_fwd_from, zero_fwd, c0_stats ? P.c0_part : nullptr);
    ASTK_LAUNCH_CHECK();
  } else {
  // ---- layer 0: im2col + GEMM
  hipLaunchKernelGGL(k_im2col0, dim3(P.Tc[0], B), dim3(256), 0, s, X, noise, P.P0, B, P.T, P.D, P.Fc[0], P.Tc[0], d->kt[0], d->kf[0],
                     d->st[0], d->sf[0], d->pt[0], P.K0p);
  ASTK_LAUNCH_CHECK();
  ASTK_TRY(copy2d_f32(P.Wr[0], P.K0p, L[0].W, P.K0, P.Cn[0], P.K0, P.K0p, s));
  ASTK_TRY(gemm_launch(GEMM_NT, gemm_args(P.rowsc[0], P.Cn[0], P.K0p, mat(P.P0, P.K0p), mat(P.Wr[0], P.K0p), P.YC[0], P.Cn[0]), s));
  }
  for (int i = 0; i < P.n; ++i) {
    const int C = P.Cn[i], rows = P.rows[i], F = P.Fn[i];
    if (i > 0) {
      const int Ci = P.Cn[i - 1], KT = d->kt[i];
      hipLaunchKernelGGL(k_repack_w, dim3(gridn((size_t)C * Ci * KT / 8)), dim3(256), 0, s, L[i].W, P.Wr[i], C, Ci, KT, P.a_wr[i],
                         (const unsigned long long*)P.a_hp_s[i - 1], P.a_hp[i - 1]);
      ASTK_LAUNCH_CHECK();
      const long prow = (long)(P.Tn[i - 1] + 2 * P.padA[i - 1]) * Ci;
      GemmArgs g = gemm_args(P.rowsc[i], C, KT * Ci, mat2(P.HP[i - 1], P.Tc[i], prow, (long)d->st[i] * Ci), mat(P.Wr[i], (long)KT * Ci),
                             P.YC[i], C);
      // (both operands' maxima were taken by the kernels that wrote them: no absolute-maximum pass in front of this launch)
      ASTK_TRY(gemm_launch(GEMM_NT, with_amax_b(with_amax_a(lowp(g), P.a_hp[i - 1]), P.a_wr[i]), s));            // K6
    }
    if (P.pooled[i]) {     // old-path extra: max-pool in front of the BatchNorm (enc_dec.py:444-456)
      hipLaunchKernelGGL(k_maxpool, dim3(gridn((size_t)rows * C / 4)), dim3(256), 0, s, P.YC[i], P.Y[i], P.IDX[i], B, P.Fc[i], P.Tc[i], P.Fn[i], P.Tn[i],
                         P.pwf[i], P.pwt[i], C);
      ASTK_LAUNCH_CHECK();
    }
    // ---- batch statistics -> scale/shift
    if (d->no_bn) {
      ASTK_CHECK(L[i].bias, "conv_bn_relu_fwd: no_bn needs a bias (layer %d)", i);
      hipLaunchKernelGGL(k_bias_affine, dim3(cdiv(C, 256)), dim3(256), 0, s, L[i].bias, C, P.bn[i]);
      ASTK_LAUNCH_CHECK();
    } else {
    BnFinalize fin{nullptr, (double)rows * (exchange ? world : 1), L[i].gamma, L[i].beta, L[i].avg_mean, L[i].avg_var, P.bn[i], d->bn_eps, d->bn_decay};
    const bool fused_fin = train && !exchange;        // the statistics kernel's last block finalizes
    if (train) {
      BnFinalize f2 = fin;
      if (fused_fin) f2.ctr = P.fin_ctr + 64 * i;
      if (i == 0 && c0_stats)      // (layer 0, direct path: the tiles' sums came with the convolution -- 6.5 MB to add up instead of 262)
        hipLaunchKernelGGL(k_colstats_tiles, colreduce_grid(P.c0_tiles, 2 * C), dim3(256), 0, s, P.c0_part, P.c0_tiles, C, P.stat[i], f2);
      else
      hipLaunchKernelGGL(k_colstats, colreduce_grid(rows, C), dim3(256), 0, s, P.Y[i], rows, C, P.stat[i], f2);
      ASTK_LAUNCH_CHECK();
      if (exchange) ASTK_CHECK(exchange(user, P.stat[i], 2 * C, stream) == 0, "conv_bn_relu_fwd: statistics exchange failed (layer %d)", i);
    }
    if (!fused_fin) {
      hipLaunchKernelGGL(k_bn_finalize, dim3(cdiv(C, 256)), dim3(256), 0, s, P.stat[i], C, fin, train);
      ASTK_LAUNCH_CHECK();
    }
    }
    if (i < P.n - 1) {
      // (k_bn_relu_rows zeroes the pad rows of HP[i] itself)
      hipLaunchKernelGGL(k_bn_relu_rows, dim3(gridn((size_t)rows * C / 4)), dim3(256), 0, s, P.Y[i], P.bn[i], P.HP[i], rows, C,
                         P.Tn[i], P.padA[i], P.a_hp_s[i]);
      ASTK_LAUNCH_CHECK();
    } else {
      const bool tiled_off = !tune_on(TUNE_CONV_SEQ_FWD);
      if (!tiled_off && seq_bwd_applicable(C, F, (long)P.Tn[i] * B)) {
        const int pp = seq_pp(F), gy = std::max(1, std::min(2048 / (C / SEQ_CH), cdiv(P.Tn[i] * B, pp)));
        hipLaunchKernelGGL(k_bn_relu_to_seq_tiled, dim3(C / SEQ_CH, gy), dim3(256), (size_t)pp * SEQ_CH * F * sizeof(float), s, P.Y[i], P.bn[i], out, B, F,
                           P.Tn[i], C, pp, P.a_out, zb_from, P.zero_bwd_bytes);
      } else {
      const size_t shm = (size_t)C * F * sizeof(float);
      ASTK_CHECK(shm <= 64 * 1024, "cnn: C*F' too large for the re-layout tile (%zu bytes)", shm);
      hipLaunchKernelGGL(k_bn_relu_to_seq, dim3(P.Tn[i], B), dim3(256), shm, s, P.Y[i], P.bn[i], out, B, F, P.Tn[i], C, P.a_out, zb_from, P.zero_bwd_bytes);
      }
      ASTK_LAUNCH_CHECK();
    }
  }
  if (seq_out_zeroes) conv0_path_record(ws, conv0_direct(d), true);      // every launch of the call went out: its last kernel zeroes the backward's region
  return 0;
}

const void* astk_conv_out_amax(const astk_cnn_desc* d, void* ws, size_t ws_bytes) {
  CnnPlan P;
  if (make_plan(d, ws, P) != 0 || !ws || ws_bytes < P.bytes) return nullptr;
  return P.a_out;
}

#ifdef ASTK_TEST_HOOKS
int astk_conv_debug_preact(const astk_cnn_desc* d, void* ws, size_t ws_bytes, int layer, float* out, void* stream) {
  CnnPlan P;
  ASTK_TRY(make_plan(d, ws, P));
  ASTK_CHECK(ws && ws_bytes >= P.bytes && out && layer >= 0 && layer < P.n, "conv_debug_preact: bad arguments");
  const long n = (long)P.rows[layer] * P.Cn[layer];
  hipLaunchKernelGGL(k_preact, dim3(gridn((size_t)n)), dim3(256), 0, (hipStream_t)stream, P.Y[layer], P.bn[layer], out, n, P.Cn[layer]);
  ASTK_LAUNCH_CHECK();
  return 0;
}

int astk_conv_debug_kill_units(const int32_t* units, int n) {
  g_kill_units = n > 0 ? units : nullptr;
  g_kill_n = n > 0 && units ? n : 0;
  return 0;
}
#endif

int astk_conv_bn_relu_bwd(const astk_cnn_desc* d, const astk_cnn_layer_params* L, const astk_cnn_layer_grads* Gr, float* d_out,
                          void* ws, size_t ws_bytes, void* stream) {
  return astk_conv_bn_relu_bwd_sync(d, L, Gr, d_out, ws, ws_bytes, nullptr, nullptr, 1, stream);
}

int astk_conv_bn_relu_bwd_sync(const astk_cnn_desc* d, const astk_cnn_layer_params* L, const astk_cnn_layer_grads* Gr, float* d_out,
                               void* ws, size_t ws_bytes, astk_stat_exchange_fn exchange, void* user, int world, void* stream) {
  hipStream_t s = (hipStream_t)stream;
  ASTK_CHECK_DESC(d, astk_cnn_desc);
  PrecScope prec_scope(d->precision, d->gemm_operands);
  DetScope det_scope(d->deterministic);
  ASTK_CHECK(world >= 1, "conv_bn_relu_bwd: world %d", world);
  if (world == 1 || d->no_bn) exchange = nullptr;      // (no statistics to exchange without BatchNorm)
  CnnPlan P;
  ASTK_TRY(make_plan(d, ws, P));
  ASTK_CHECK(ws && ws_bytes >= P.bytes, "conv_bn_relu_bwd: workspace too small");
  ASTK_CHECK(d_out && L && Gr, "conv_bn_relu_bwd: null pointer");
  {
    const int fwd_direct = conv0_path_lookup(ws), bwd_direct = conv0_direct(d) ? 1 : 0;
    ASTK_CHECK(fwd_direct < 0 || fwd_direct == bwd_direct, "conv_bn_relu_bwd: the forward call on this workspace took the %s layer-0 path, this call "
               "would take the %s one (the process-default arithmetic changed between the two calls while the descriptor's precision is DEFAULT?)",
               fwd_direct ? "direct-convolution" : "im2col", bwd_direct ? "direct-convolution" : "im2col");
  }
  const int B = P.B;
  // the last layer's BatchNorm backward reads d_out in its (T'', B, C*F') layout itself (k_bn_bwd_*_seq); shapes those kernels do not take
  // (and the test hook that edits G) go through the re-ordered copy G as before.  astk_set_tuning("conv.seq_bwd", 0): always the copy.
  const bool seq_off = !tune_on(TUNE_CONV_SEQ_BWD);
  bool seq_last = !seq_off && seq_bwd_applicable(P.Cn[P.n - 1], P.Fn[P.n - 1], (long)P.Tn[P.n - 1] * B);
#ifdef ASTK_TEST_HOOKS
  if (g_kill_n > 0) seq_last = false;
#endif
  if (!seq_last) {
    const int i = P.n - 1;
    const size_t shm = (size_t)P.Cn[i] * P.Fn[i] * sizeof(float);
    hipLaunchKernelGGL(k_seq_to_rows, dim3(P.Tn[i], B), dim3(256), shm, s, d_out, P.G, B, P.Fn[i], P.Tn[i], P.Cn[i]);
    ASTK_LAUNCH_CHECK();
  }
  UnpackJobs uj;
  uj.n = 0;
  size_t uj_max = 0;
  auto unpack_add = [&](const float* src, float* dW, int Co, int Ci, int KT, int JG, int ldg) {
    uj.src[uj.n] = src; uj.dW[uj.n] = dW; uj.Co[uj.n] = Co; uj.Ci[uj.n] = Ci; uj.KT[uj.n] = KT; uj.JG[uj.n] = JG; uj.ldg[uj.n] = ldg;
    ++uj.n;
    uj_max = std::max(uj_max, (size_t)Co * Ci * KT);
  };
  for (int i = P.n - 1; i >= 0; --i) {
    const int C = P.Cn[i], rows = P.rows[i];
#ifdef ASTK_TEST_HOOKS
    if (g_kill_n > 0) {      // test instrumentation (astk_conv_debug_kill_units): drop the upstream gradient of the listed units
      hipLaunchKernelGGL(k_kill_units, dim3(cdiv(g_kill_n, 256)), dim3(256), 0, s, P.G, g_kill_units, g_kill_n, i, rows, C);
      ASTK_LAUNCH_CHECK();
    }
#endif
    // ---- ReLU + BatchNorm backward: G (grad wrt post-ReLU) -> DY[i] (grad wrt raw conv output)
    // statistics and dWr scratch of every layer: zeroed by the forward call's last kernel (conv0_path_record), else here
    if (i == P.n - 1 && !conv_take_bwd_clean(ws)) ASTK_TRY(fill_zero(P.stat[0], P.zero_bwd_bytes, s));
    const bool seq = seq_last && i == P.n - 1;
    const int seq_gx = C / SEQ_CH;
    if (seq) {
      const int sb = (int)tune(TUNE_CONV_SEQ_STATS_BLOCKS);
      const int gy = std::max(1, std::min(sb / seq_gx, cdiv(P.Tn[i] * B, seq_pp(P.Fn[i]))));
      hipLaunchKernelGGL(k_bn_bwd_stats_seq, dim3(seq_gx, gy), dim3(256), seq_tile_bytes(P.Fn[i]), s, P.Y[i], d_out, P.bn[i], B, P.Fn[i], P.Tn[i], C,
                         seq_pp(P.Fn[i]), P.stat[i]);
    } else
      hipLaunchKernelGGL(k_bn_bwd_stats, colreduce_grid(rows, C), dim3(256), 0, s, P.Y[i], P.G, P.bn[i], rows, C, P.stat[i]);
    ASTK_LAUNCH_CHECK();
    if (exchange) {
      hipLaunchKernelGGL(k_bn_param_grads, dim3(cdiv(C, 256)), dim3(256), 0, s, P.stat[i], C, Gr[i].dgamma, Gr[i].dbeta);
      ASTK_LAUNCH_CHECK();
      ASTK_CHECK(exchange(user, P.stat[i], 2 * C, stream) == 0, "conv_bn_relu_bwd: statistics exchange failed (layer %d)", i);
    }
    const int F = P.Fc[i];             // (b, f) groups of this layer's convolution output = of its input
    const int Tp = P.Tc[i] + P.dF[i] + P.dB[i];
    if (P.dF[i] + P.dB[i] > 0 && P.pooled[i]) {       // (un-pooled layers: the apply kernel zeroes the pad rows of the dY it writes)
      hipLaunchKernelGGL(k_zero_pads, dim3(gridn((size_t)B * F * (P.dF[i] + P.dB[i]) * C / 4)), dim3(256), 0, s, P.DY[i], B * F, P.Tc[i], P.dF[i],
                         P.dB[i], C);
      ASTK_LAUNCH_CHECK();
    }
    // (no_bn: the ReLU mask alone -- scale 1 and a zero 1/m switch the BatchNorm terms off; the column sums of g are the bias gradient)
    // pooled layers: the gradient wrt the POOLED output goes to DYP (plain rows), k_unpool spreads it over the padded dY of the convolution
    if (seq) {
      const int ab = (int)tune(TUNE_CONV_SEQ_APPLY_BLOCKS);
      const int gy = std::max(1, std::min(ab / seq_gx, cdiv(P.Tn[i] * B, seq_pp(P.Fn[i]))));
      hipLaunchKernelGGL(k_bn_bwd_apply_seq, dim3(seq_gx, gy), dim3(256), seq_tile_bytes(P.Fn[i]), s, P.Y[i], d_out, P.bn[i], P.stat[i],
                         P.pooled[i] ? P.DYP[i] : P.DY[i], B, P.Fn[i], P.Tn[i], C, seq_pp(P.Fn[i]), P.pooled[i] ? 0 : P.dF[i], P.pooled[i] ? 0 : P.dB[i],
                         (exchange || d->no_bn) ? nullptr : Gr[i].dgamma, (exchange || d->no_bn) ? nullptr : Gr[i].dbeta,
                         d->no_bn ? 0.f : 1.f / ((float)rows * (exchange ? world : 1)), (i > 0 && !P.pooled[i]) ? P.a_dy_s[i] : nullptr);
    } else
    hipLaunchKernelGGL(k_bn_bwd_apply, dim3(gridn((size_t)rows * C / 4)), dim3(256), 0, s, P.Y[i], P.G, P.bn[i], P.stat[i], P.pooled[i] ? P.DYP[i] : P.DY[i],
                       rows, C, P.Tn[i], P.pooled[i] ? 0 : P.dF[i], P.pooled[i] ? 0 : P.dB[i], (exchange || d->no_bn) ? nullptr : Gr[i].dgamma,
                       (exchange || d->no_bn) ? nullptr : Gr[i].dbeta, d->no_bn ? 0.f : 1.f / ((float)rows * (exchange ? world : 1)),
                       (i > 0 && !P.pooled[i]) ? P.a_dy_s[i] : nullptr);
    ASTK_LAUNCH_CHECK();
    if (P.pooled[i]) {
      hipLaunchKernelGGL(k_unpool, dim3(gridn((size_t)P.rowsc[i] * C / 4)), dim3(256), 0, s, P.DYP[i], P.IDX[i], P.DY[i], B, P.Fc[i], P.Tc[i], P.Fn[i],
                         P.Tn[i], P.pwf[i], P.pwt[i], C, P.dF[i], P.dB[i], i > 0 ? P.a_dy_s[i] : nullptr);
      ASTK_LAUNCH_CHECK();
    }
    if (d->no_bn) {
      ASTK_CHECK(Gr[i].dbias, "conv_bn_relu_bwd: no_bn needs a bias gradient (layer %d)", i);
      hipLaunchKernelGGL(k_bias_grad, dim3(cdiv(C, 256)), dim3(256), 0, s, P.stat[i], C, Gr[i].dbias);
      ASTK_LAUNCH_CHECK();
    }
    if (i == 0) {
      // ---- wgrad layer 0: dW0p[C0][K0p] = DY0^T P0
      const int ks = ksplit_for(1, P.rowsc[0]);
      if (conv0_direct(d)) {
        // the patches as a zero-copy window matrix over XF: row (b, f, t1) = the kt * JG floats from row st * t1 of group (b, f) on
        const int ldg = (P.K0g + 3) & ~3;
        MatView Bm = mat2(P.XF, P.Tc[0], (long)P.xf_rows * P.JG, (long)d->st[0] * P.JG);
        ASTK_TRY(gemm_launch(GEMM_TN, gemm_args(C, P.K0g, P.rowsc[0], mat(P.DY[0], C), Bm, P.dWr[0], ldg, nullptr, GEMM_ATOMIC, ks), s));
        unpack_add(P.dWr[0], Gr[0].dW, C, d->kf[0], d->kt[0], P.JG, ldg);
      } else {
      ASTK_TRY(gemm_launch(GEMM_TN, gemm_args(C, P.K0p, P.rowsc[0], mat(P.DY[0], C), mat(P.P0, P.K0p), P.dWr[0], P.K0p, nullptr, GEMM_ATOMIC, ks), s));
      ASTK_TRY(add2d_f32(Gr[0].dW, P.K0, P.dWr[0], P.K0p, C, P.K0, s));
      }
    } else {
      const int Ci = P.Cn[i - 1], KT = d->kt[i], st = d->st[i], pt = d->pt[i];
      const long dyrow = (long)Tp * C;                                  // per (b,f) group of the padded dY
      const long hprow = (long)(P.Tn[i - 1] + 2 * P.padA[i - 1]) * Ci;  // per (b,f) group of HP[i-1]
      // the padded dY feeds the weight gradient and every stride phase of the input gradient: its maximum (fp16x2 GEMM scale) was taken
      // by k_bn_bwd_apply when it wrote it; the activations' by k_bn_relu_rows in the forward pass
      const unsigned long long* ady = P.a_dy[i];
      // ---- dgrad, part 1: the phase weights of the stride phases rho of the input position t_in = rho + st*j (the first of these small
      // kernels also folds dY's strided maximum slot into the plain one both products read, hence in front of the weight gradient)
      GemmArgs ph[GEMM_GROUP_MAX];
      int nph = 0;
      bool folded = false;
      PhaseWJobs pj;
      pj.n = 0;
      int pj_na_max = 0;
      auto launch_phase_w = [&]() {      // the phase weights gathered so far (wd buffers 0 .. pj.n-1)
        if (pj.n == 0) return;
        hipLaunchKernelGGL(k_phase_w, dim3(gridn((size_t)Ci * pj_na_max * C / 8), pj.n), dim3(256), 0, s, L[i].W, pj, C, Ci, KT, st,
                           folded ? nullptr : (const unsigned long long*)P.a_dy_s[i], P.a_dy[i]);
        folded = true;
        pj.n = 0;
        pj_na_max = 0;
      };
      for (int rho = 0; rho < st && rho < P.Tn[i - 1]; ++rho) {
        const int r = (rho + pt) % st;
        const int na = (KT - r + st - 1) / st;
        if (na <= 0) {   // no tap reaches this phase: gradient is zero
          continue;
        }
        const int nj = (P.Tn[i - 1] - rho + st - 1) / st;
        const int q0 = (rho + pt) / st;
        if (nph == P.wd_copies) {      // (more phases than buffers: flush)
          launch_phase_w();
          ASTK_LAUNCH_CHECK();
          ASTK_TRY(gemm_launch_group(GEMM_NT, ph, nph, s));
          nph = 0;
        }
        float* wd = P.Wd + (size_t)nph * P.wd_stride;
        // (a_wd[k] is zeroed once per backward call: a slot that is reused -- more phases than buffers, a deeper stack -- keeps the larger
        //  of its users' maxima, i.e. a slightly conservative scale for phase weights of similar magnitude)
        pj.wd[pj.n] = wd; pj.r[pj.n] = r; pj.na[pj.n] = na; pj.amax[pj.n] = P.a_wd[nph];
        ++pj.n;
        pj_na_max = std::max(pj_na_max, na);
        const long start = (long)(q0 - na + 1 + P.dF[i]);
        ASTK_CHECK(start >= 0, "cnn dgrad: negative window start");
        GemmArgs g = gemm_args(B * F * nj, Ci, na * C, mat2(P.DY[i] + start * C, nj, dyrow, C), mat(wd, (long)na * C),
                               P.G + (long)rho * Ci, Ci);
        g.c_tn = nj;
        g.c_sg = (long)P.Tn[i - 1] * Ci;
        g.c_st = (long)st * Ci;
        ph[nph] = with_amax_b(with_amax_a(lowp(g), ady), P.a_wd[nph]);
        ++nph;
      }
      launch_phase_w();
      ASTK_LAUNCH_CHECK();
      if (!folded) ady = P.a_dy_s[i];      // (no phase kernel ran: the products read the strided slot itself)
      // ---- wgrad: dWr[co][kt*Ci+ci] = sum_rows DY[row][co] * window(row)[k]
      {
        MatView A = mat2(P.DY[i] + (long)P.dF[i] * C, P.Tc[i], dyrow, C);
        MatView Bm = mat2(P.HP[i - 1], P.Tc[i], hprow, (long)st * Ci);
        const long tiles = (long)cdiv(C, 128) * cdiv(KT * Ci, 128);
        ASTK_TRY(gemm_launch(GEMM_TN, with_amax_b(with_amax_a(lowp(gemm_args(C, KT * Ci, P.rowsc[i], A, Bm, P.dWr[i], (long)KT * Ci, nullptr, GEMM_ATOMIC, ksplit_for(tiles, P.rowsc[i]))), ady), P.a_hp[i - 1]), s));
      }
      unpack_add(P.dWr[i], Gr[i].dW, C, Ci, KT, 0, 0);
      // ---- dgrad, part 2: one window GEMM per stride phase, all phases in one grouped launch
      if (nph > 0) ASTK_TRY(gemm_launch_group(GEMM_NT, ph, nph, s));
    }
  }
  if (uj.n > 0) {
    hipLaunchKernelGGL(k_unpack_dw, dim3(gridn(uj_max), uj.n), dim3(256), 0, s, uj, persist_status_word(), d->status_dst);
    ASTK_LAUNCH_CHECK();
  } else if (d->status_dst) {
    ASTK_TRY(status_snapshot_launch(d->status_dst, s));
  }
  return 0;
}

}  // extern "C"
