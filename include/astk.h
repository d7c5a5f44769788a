/* astk.h -- C ABI of libastk.so: the MI355X (gfx950) encoder-decoder train-step kernels.
 *
 * The reference (0xSameer/ast) has no FFI: its hot path is Python calling Chainer links/functions.
 * Each entry point below replaces the Chainer call sequence of one reference function, cited as
 * /root/reference/<file>:<lines>.  Conventions:
 *   - every pointer is a caller-owned DEVICE pointer (16-byte aligned), sizes are explicit;
 *   - no allocation, no ownership transfer, no implicit synchronisation: functions only enqueue
 *     work on `stream` (a hipStream_t passed as void*), so a caller may capture them in a hipGraph;
 *   - the caller supplies one workspace per op (size from the matching *_workspace_bytes query);
 *     the forward call leaves saved activations in it and the backward call reads them, so it must
 *     stay untouched between the two;
 *   - return 0 on success, <0 on error; astk_last_error() gives the thread-local message;
 *   - weight layouts are Chainer's (A1/A2/A3/A10 of SURVEY.md): Linear W (out,in), LSTM gates
 *     interleaved (unit j, gate k -> row 4j+k, k = a,i,f,o), Conv W (out,in,kh,kw).
 * Storage and accumulation are float32 everywhere, matching the reference's dtype (seq2seq.py:154,302,420).  The PRODUCTS of the
 * batched dense GEMMs run on the 16-bit matrix pipe as bf16x3 splits by default: every f32 operand value is split inside the kernel
 * into three bf16 terms that represent it EXACTLY (f32 exponent range), six term products per useful product, each product exact to
 * 2^-26 -- at least as accurate as an f32 fma chain on any data.  ASTK_PREC_F32 selects the literal f32-input MFMA chain,
 * ASTK_PREC_FP16X2 a faster, NARROWER two-term fp16 split (22 bits, limited exponent range; opt-in, never the default).  Decoder loop,
 * attention, softmax-CE and optimizer are IEEE f32 in every mode; the encoder recurrences follow the mode (bf16x3: weight fragments and
 * every step's activations as exact three-term bf16 splits on v_mfma_f32_16x16x32_bf16; fp16x2: two scaled fp16 terms; f32: f32-input MFMAs).
 * The arithmetic is chosen per call by the descriptors' `precision` field (ASTK_PREC_DEFAULT = the process-wide default).
 */
#ifndef ASTK_H
#define ASTK_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ASTK_VERSION 106
#define ASTK_MAX_CNN_LAYERS 4
#define ASTK_MAX_RNN_LAYERS 8
#define ASTK_MAX_ATTN 4

int astk_version(void);
const char* astk_last_error(void);

/* Arithmetic of an op's f32-accurate products: field `precision` of the descriptors below, argument of astk_gemm_f32_ex.
 *   ASTK_PREC_DEFAULT  the process-wide default (bf16x3 unless astk_set_gemm_precision changed it)
 *   ASTK_PREC_FP16X2   two fp16 terms per operand behind a per-operand power-of-two scale: 22 significant bits, and only for values within
 *                      2^-17 of the operand's maximum (NARROWER than float32: opt-in), three MFMAs per 16 k; encoder recurrences likewise
 *   ASTK_PREC_BF16X3   three bf16 terms per operand (exact representation of every f32 value, f32 exponent range), six MFMAs per 16 k
 *   ASTK_PREC_F32      v_mfma_f32_32x32x2_f32 / 16x16x4_f32: IEEE f32 products, the reference's literal arithmetic
 * `gemm_operands`: ASTK_OPERANDS_FP16 lets the products a caller marks eligible (K6, K9, batched K18 / K24: BASELINE configs[4]) run with
 * ONE fp16 term per operand (reduced precision; the 1e-4 parity gate does not apply); ASTK_OPERANDS_F32 forces the f32-accurate scheme;
 * ASTK_OPERANDS_DEFAULT = the process-wide setting of astk_set_low_precision_gemms. */
enum { ASTK_PREC_DEFAULT = 0, ASTK_PREC_FP16X2 = 1, ASTK_PREC_BF16X3 = 2, ASTK_PREC_F32 = 3 };
enum { ASTK_OPERANDS_DEFAULT = 0, ASTK_OPERANDS_F32 = 1, ASTK_OPERANDS_FP16 = 2 };

/* ---------------------------------------------------------------- generic f32 MFMA GEMM
 * C[M,N] (+)= op(A) op(B) (+ bias[n]).  layout: 0 = "NT"  A[M,K] K-contiguous, B[N,K] K-contiguous (Linear forward)
 *                                               1 = "NN"  A[M,K], B[K,N]                             (dgrad)
 *                                               2 = "TN"  A[K,M], B[K,N]                             (wgrad)
 * mode: 0 store, 1 C += result (single pass), 2 atomic add with split-K over `ksplit` slices.
 * batch > 1 repeats with element strides sA/sB/sC.  Leading dimensions must be multiples of 4.
 * Replaces the cuBLAS sgemm behind chainer.functions.linear / its backward (Chainer-sem A2). */
int astk_gemm_f32(int layout, int M, int N, int K,
                  const float* A, long lda, const float* B, long ldb, float* C, long ldc,
                  const float* bias, int mode, int ksplit, int batch, long sA, long sB, long sC, void* stream);
/* The same product under the arithmetic `precision` names (ASTK_PREC_*); astk_gemm_f32 = ASTK_PREC_DEFAULT. */
int astk_gemm_f32_ex(int layout, int M, int N, int K,
                     const float* A, long lda, const float* B, long ldb, float* C, long ldc,
                     const float* bias, int mode, int ksplit, int batch, long sA, long sB, long sC, int precision, void* stream);

/* BASELINE configs[4] ("fp16 MFMA GEMMs"): mode 1 lets the batched products of the CNN layers >= 1 (K6), of the encoder's layer-0
 * input projection (K9) -- forward and backward -- and of the decoder LSTMs' and the output layer's backward (K18, K24: their weight
 * gradients over all steps, the embedding columns of the input gradient) run with operands rounded to fp16 behind the operand's
 * power-of-two scale where its caller measured it (one v_mfma_f32_32x32x16_f16 per tile, f32 accumulation) instead of the f32-accurate
 * split (below).  The per-step products of K18 / K24 live inside the latency-bound decoder loop kernels and stay on f32 MFMAs.  Reduced precision: the 1e-4 fp32 parity gate does not apply in
 * this mode (SURVEY.md 8d asks for the loss drift instead: tests/test_gpu_model.py).  Process-wide DEFAULT for descriptors whose
 * gemm_operands field is ASTK_OPERANDS_DEFAULT; 0 (default) = off. */
int astk_set_low_precision_gemms(int mode);
int astk_get_low_precision_gemms(void);
/* fp16x2 mode only: launches below `flops` floating-point operations do not repay the absolute-maximum pass the scaled fp16 terms
 * need and run as bf16x3 instead.  Default 3e9; 0 = fp16 terms always.  Process-wide; returns the
 * previous value. */
double astk_set_gemm_bf16_split_below(double flops);
/* The process-wide DEFAULT arithmetic (what ASTK_PREC_DEFAULT resolves to), switchable at run time (bench.py times the same step under each):
 *   0  fp16x2: two fp16 terms per operand behind a power-of-two scale, 22 significant bits per operand value, in the batched GEMMs AND in
 *      the encoder's persistent recurrence kernels -- NARROWER than float32, opt-in;
 *   1  bf16x3 (default): three bf16 terms (exact operands, no scales) in the batched GEMMs, exact-f32 MFMAs in the recurrences;
 *   2  f32: v_mfma_f32_32x32x2_f32 / 16x16x4_f32 everywhere (IEEE f32 products, the reference's literal arithmetic).
 * Returns the previous mode (<0: error). */
int astk_set_gemm_precision(int mode);
int astk_get_gemm_precision(void);
/* Tuning knobs: the ONE documented, process-wide switchboard for A/B measurements and fall-backs (the library reads no environment
 * variable).  Read at every launch; not meant to be flipped between a forward call and its backward call (the decoder / CNN backward
 * refuse a workspace whose forward took another kernel path).  Keys (default):
 *   gemm.tile (0 = per launch; 64 | 128 | 256 forces the block tile)     gemm.t256_above (2e10 flops: the 12-wave 256 x 128 kernel from here on)
 *   gemm.grid (-1 = per launch)   gemm.hybrid (1)   gemm.chunk (1)   gemm.chunk_div (4)   gemm.log (0: 1 prints every launch to stderr)
 *   gemm.deterministic (0: process default of the descriptors' `deterministic` field)   gemm.forward_pairs (1; 2 = the forward ops' two-contributor rule for every launch, plain astk_gemm_f32 calls included)   gemm.ticket (libastk_test.so only)
 *   conv.direct0 (1)   conv.seq_fwd (1)   conv.seq_bwd (1)   conv.seq_stats_blocks (1024)   conv.seq_apply_blocks (1024)
 *   dec.persist (1: 0 = per-launch decoder loop)   dec.b6_split (1)   dec.b6_fused (1)   dec.wide (1)
 *   lstm.persist (1: 0 = one fused-cell launch per step)   lstm.hoist (1)   lstm.x3 (1)   lstm.x4 (1)
 *   lstm.rows32 (-1 = 32 batch rows per recurrence workgroup when that spares launches; 0 never; 1 = one wave set doing two tiles, 2 = two
 *   wave sets per SIMD, whenever possible)   lstm.duo_side (0)
 *   lstm.overlap_chunk (0 = sized from the free CUs; else time steps per side-stream chunk)   lstm.side_fwd (1)   lstm.side_bwd (0 = off; n = chunks of the input gradient on side_stream behind the backward recurrence, the rest in line; < 0 every chunk)
 *   row.longk (2048)   persist.spin_limit (0 = 2^22 polls)   colreduce.blocks (256)
 * astk_set_tuning returns 0, or -1 for an unknown key; astk_tuning_key(i) enumerates the keys (NULL behind the last). */
int astk_set_tuning(const char* key, double value);
int astk_get_tuning(const char* key, double* value);
const char* astk_tuning_key(int index);
#ifdef ASTK_TEST_HOOKS
/* Test hook (libastk_test.so only): sets the generation counter of the fp16x2 scale slots (tests preset it close to the 32-bit wrap). */
int astk_debug_set_amax_generation(unsigned gen);
/* Test hook (libastk_test.so only): ONE grouped launch of n products C_i = op(A_i) op(B_i) (dense row-major operands) the way a forward op
 * issues it -- under the forward ops' two-contributor rule when `forward` is set -- for shapes no op of the step produces. */
int astk_debug_gemm_group(int layout, int n, const int* M, const int* N, const int* K, const float* const* A, const float* const* B,
                          float* const* C, int forward, int precision, void* stream);
#endif

/* ---------------------------------------------------------------- CNN front-end  (seq2seq.py:158-180)
 * [Conv2D(no bias) -> (max-pool) -> BatchNorm(train: batch stats) -> ReLU] x n_layers (or, with no_bn, [Conv2D(bias) -> ReLU]), then the (T'',B,C*F') time-major
 * re-layout with feature index c*F'+f (quirk Q9).  Layer 0: in_channels 1, kernel (kt,kf), stride (st,sf),
 * pad (pt,0).  Layers >= 1: kernel (kt,1), stride (st,1), pad (pt,0) -- the shipped cnn_config. */
typedef struct {
  size_t struct_size;  /* sizeof(astk_cnn_desc) of the header the caller was built against: every entry point refuses a descriptor whose size
                          differs from its own (the descriptors carry pointers the library writes through -- status_dst here, zero_ptr in the
                          decoder's -- so a caller built against an older layout must fail loudly, not hand over garbage).  ZERO-INITIALISE the
                          struct, set struct_size, then the fields you use: every optional field's "off" value is 0 / NULL. */
  int B, T, D;
  int n_layers;
  int C[ASTK_MAX_CNN_LAYERS];
  int kt[ASTK_MAX_CNN_LAYERS], kf[ASTK_MAX_CNN_LAYERS];
  int st[ASTK_MAX_CNN_LAYERS], sf[ASTK_MAX_CNN_LAYERS];
  int pt[ASTK_MAX_CNN_LAYERS];
  float bn_eps;    /* 2e-5  (Chainer-sem A4) */
  float bn_decay;  /* 0.9 */
  int no_bn;       /* cnn_config.bn = false (seq2seq.py:43-57): Conv2D WITH bias -> ReLU, no BatchNorm; 0 = the shipped configs */
  /* OLD-path extra (enc_dec.py:444-456, `cnn_pool`): F.max_pooling_nd(h, (pool_t, pool_f)) between a layer's convolution and its
   * BatchNorm -- window = stride, no padding, cover_all (the last window may be partial: out = ceil(in / window)).  0 or 1 = none,
   * -1 = the whole extent.  No shipped config sets it. */
  int pool_t[ASTK_MAX_CNN_LAYERS], pool_f[ASTK_MAX_CNN_LAYERS];
  int precision;       /* ASTK_PREC_*: arithmetic of this op's products (0 = process default) */
  int gemm_operands;   /* ASTK_OPERANDS_* (0 = process default) */
  float* status_dst;   /* optional (NULL = none): astk_conv_bn_relu_bwd(_sync) also leaves a copy of the persistent kernels' status word
                          there, written by the op's last kernel -- astk_persist_status_snapshot without a launch of its own (the CNN
                          backward is the train step's last op behind the recurrences).  Ignored by the forward call. */
  int deterministic;   /* backward call: 1 = every accumulated sum of this op (split tiles of the weight / input gradient products) is formed in a
                          fixed order -- bit-reproducible from run to run, a few per cent slower; 0 = the process default (astk_set_tuning
                          "gemm.deterministic", default off: float atomics in arrival order) */
} astk_cnn_desc;

typedef struct {
  const float* W;      /* (C, Cin, kt, kf) */
  const float* gamma;  /* (C) */
  const float* beta;   /* (C) */
  float* avg_mean;     /* (C) running stats, updated in train mode */
  float* avg_var;      /* (C) */
  const float* bias;   /* (C) only with no_bn (gamma, beta, avg_* unused then) */
} astk_cnn_layer_params;

typedef struct {
  float* dW;
  float* dgamma;
  float* dbeta;
  float* dbias;        /* only with no_bn */
} astk_cnn_layer_grads;

/* output dims: T_out = T'', F_out = F', feature dim = C_last*F' */
int astk_conv_bn_relu_out_dims(const astk_cnn_desc* d, int* T_out, int* F_out, int* feat_dim);
size_t astk_conv_bn_relu_workspace_bytes(const astk_cnn_desc* d);
/* X (B,T,D); noise (B,T,D) or NULL: X*noise is the speech-noise product of seq2seq.py:297-305;
 * out (T'',B,C_last*F').  train=0 uses running statistics (chainer.config.train False). */
int astk_conv_bn_relu_fwd(const astk_cnn_desc* d, const astk_cnn_layer_params* layers, const float* X,
                          const float* noise, float* out, void* ws, size_t ws_bytes, int train, void* stream);
/* d_out (T'',B,C_last*F') is overwritten.  Gradients are ACCUMULATED into grads (caller zeroes = cleargrads).
 * The backward call must see the descriptor (incl. `precision` / `gemm_operands`) and the process defaults its forward call saw: which
 * of the workspace's layer-0 buffers exists (the patch matrix of im2col + GEMM, or the frequency-blocked input of the direct convolution
 * kernel, conv.hip) follows from them. */
int astk_conv_bn_relu_bwd(const astk_cnn_desc* d, const astk_cnn_layer_params* layers,
                          const astk_cnn_layer_grads* grads, float* d_out, void* ws, size_t ws_bytes, void* stream);
/* Where the forward call leaves the absolute maximum of `out` (16 64-bit words inside ws; taken by the kernel that writes `out`):
 * pass it on as astk_lstm_stack_desc.x_amax.  Valid from the forward call until ws is reused.  NULL: bad arguments. */
const void* astk_conv_out_amax(const astk_cnn_desc* d, void* ws, size_t ws_bytes);

#ifdef ASTK_TEST_HOOKS
/* Test instrumentation, compiled into libastk_test.so only (tests/test_gpu_model.py, the batch-permutation property).  astk_conv_debug_preact: after a forward call, out
 * [(b,f,t)][c] = the post-BatchNorm pre-activation of `layer` (what the ReLU sees), rows = B*F'*T_layer.  astk_conv_debug_kill_units:
 * the following backward calls of this process zero the upstream gradient of the listed units (n triples layer, row, channel in
 * device memory, caller-owned; n = 0 switches it off).  Two valid float32 evaluations of one batch can disagree on the SIGN of a
 * pre-activation that lies within rounding of the ReLU kink; the test names those units and shows that nothing else differs. */
int astk_conv_debug_preact(const astk_cnn_desc* d, void* ws, size_t ws_bytes, int layer, float* out, void* stream);
int astk_conv_debug_kill_units(const int32_t* units, int n);
#endif

/* Data-parallel BatchNorm with GLOBAL batch statistics (SURVEY.md 8e "SyncBN"): the same two calls with an exchange step.
 * After a layer's local per-channel sums are on the device -- forward (sum y, sum y^2), backward (sum g, sum g*xhat), `n` = 2*C
 * doubles at `stat`, inside the caller's workspace -- the library calls `exchange(user, stat, n, stream)` on the host; the callee
 * enqueues, on `stream`, an in-place SUM of that buffer over the `world` replicas (RCCL all-reduce) and returns 0.  Statistics,
 * running averages and the input gradient then use world * rows samples, which is what one process would compute on the
 * concatenated batch; dgamma / dbeta accumulate the LOCAL sums, like every other parameter gradient (the caller's gradient
 * all-reduce adds the replicas).  exchange == NULL (or world == 1) is astk_conv_bn_relu_fwd / _bwd. */
typedef int (*astk_stat_exchange_fn)(void* user, double* stat, int n, void* stream);
int astk_conv_bn_relu_fwd_sync(const astk_cnn_desc* d, const astk_cnn_layer_params* layers, const float* X, const float* noise,
                               float* out, void* ws, size_t ws_bytes, int train, astk_stat_exchange_fn exchange, void* user,
                               int world, void* stream);
int astk_conv_bn_relu_bwd_sync(const astk_cnn_desc* d, const astk_cnn_layer_params* layers, const astk_cnn_layer_grads* grads,
                               float* d_out, void* ws, size_t ws_bytes, astk_stat_exchange_fn exchange, void* user, int world,
                               void* stream);

/* ---------------------------------------------------------------- encoder LSTM stacks  (seq2seq.py:182-242)
 * n_dirs independent uni-directional stacks of n_layers L.LSTM links (Chainer-sem A1), dropout on each
 * layer's *output* copy only.  Direction 1 consumes frames in the reference's order 0,T-1,...,1 (quirk Q1)
 * and its outputs are flipped before the concat, so enc_states[b,p,h:2h] is what seq2seq.py:231-242 builds. */
typedef struct {
  size_t struct_size;  /* sizeof(astk_lstm_stack_desc), see astk_cnn_desc.struct_size */
  int T, B, in_dim, h, n_layers, n_dirs;
  /* Optional hints that spare the f32-accurate GEMMs their absolute-maximum passes (0 / NULL: every launch measures its operands):
   *   out_bound  an upper bound of |layer output| as the next layer and the weight-gradient products read it: 1 without dropout
   *              (|h| < 1), 1 / (1 - ratio) with the masks of astk_fill_dropout_mask;
   *   x_amax     the maximum words of the input frames x, as left by the kernel that wrote them (astk_conv_out_amax). */
  float out_bound;
  const void* x_amax;
  int precision;       /* ASTK_PREC_*: batched products AND the persistent recurrence kernels (0 = process default) */
  int gemm_operands;   /* ASTK_OPERANDS_* (0 = process default) */
  /* Work BESIDE the recurrences (optional; NULL / 0 = everything in line on `stream`).  The persistent recurrence kernels are latency-bound
   * and occupy one workgroup per (direction, layer, 16-unit slice, batch tile): 96-192 of the 256 CUs at the shipped shape.  With a
   * caller-owned ORDINARY second stream in `side_stream` (hipStream_t; no CU mask needed) the layer-0 batched products run on it in time
   * chunks, each launch capped at `side_wgs` workgroups (0 = the CUs the recurrence grid leaves free), gated by flags in the workspace:
   * forward, chunk k of the input projection is produced on the side stream while the recurrence consumes chunk k-1 (the cells of layer 0 wait
   * for the chunk's flag, bounded spins + abort word like every other hand-off); backward, the input / weight gradient of a chunk starts when the
   * recurrence has passed it.  The capped grids can never keep the recurrence grid from becoming resident.  Both calls join the side stream
   * before they return, so callers see one-stream semantics.  Results are bit-identical to the in-line schedule in the forward pass and
   * equal up to the order of float atomics in the backward pass (identical under `deterministic`). */
  void* side_stream;
  int side_wgs;
  int deterministic;   /* backward call: 1 = bias gradients and split tiles summed in a fixed order (see astk_cnn_desc.deterministic); 0 = process default */
} astk_lstm_stack_desc;

typedef struct {
  const float* Wu; /* upward.W  (4h, in)  */
  const float* b;  /* upward.b  (4h)      */
  const float* Wl; /* lateral.W (4h, h)   */
} astk_lstm_params;

typedef struct {
  float* dWu;
  float* db;
  float* dWl;
} astk_lstm_grads;

size_t astk_lstm_stack_workspace_bytes(const astk_lstm_stack_desc* d);
/* x (T,B,in); params[dir*n_layers+layer]; masks: NULL or (n_dirs,n_layers,T,B,h) scaled keep-masks indexed by
 * loop step; enc_states (B,T,n_dirs*h); cT,hT (n_dirs,n_layers,B,h) final UN-dropped states. */
int astk_lstm_stack_fwd(const astk_lstm_stack_desc* d, const astk_lstm_params* params, const float* x,
                        const float* masks, float* enc_states, float* cT, float* hT,
                        void* ws, size_t ws_bytes, void* stream);
/* d_enc_states (B,T,n_dirs*h); d_cT,d_hT (n_dirs,n_layers,B,h) or NULL; dx (T,B,in) written (may be NULL). */
int astk_lstm_stack_bwd(const astk_lstm_stack_desc* d, const astk_lstm_params* params, const astk_lstm_grads* grads,
                        const float* x, const float* masks, const float* d_enc_states, const float* d_cT,
                        const float* d_hT, float* dx, void* ws, size_t ws_bytes, void* stream);
/* Same, with the persistent recurrence kernel (all T steps of all cells in one launch, one workgroup on each of 192 CUs at cfg 2,
 * latency-bound) enqueued on `recurrence_stream` and ordered against `stream` with events on both sides; everything else stays on
 * `stream`.  Meant for a stream created with hipExtStreamCreateWithCUMask: with the recurrence confined to one set of CUs and
 * independent work (astk_decoder_bwd_phase PARAMS) on a stream masked to the others, the two overlap without slowing the chain
 * (scratch/overlap_probe.py).  NULL = astk_lstm_stack_bwd. */
int astk_lstm_stack_bwd_on(const astk_lstm_stack_desc* d, const astk_lstm_params* params, const astk_lstm_grads* grads,
                           const float* x, const float* masks, const float* d_enc_states, const float* d_cT,
                           const float* d_hT, float* dx, void* ws, size_t ws_bytes, void* stream, void* recurrence_stream);

/* ---------------------------------------------------------------- attention step  (seq2seq.py:336-357)
 * q = Wa h + ba is computed by the caller (GEMM); this is the scan over enc_states:
 * s[b,t] = enc[b,t,:].q[b,:]; alpha = softmax_t(s) (no mask, quirk Q2); cv[b,:] = sum_t alpha[b,t] enc[b,t,:].
 * One streaming read of enc_states (online softmax), split over nsplit time chunks per batch row. */
size_t astk_attn_workspace_bytes(int B, int T, int H);
int astk_attn_step_fwd(int B, int T, int H, const float* enc, const float* q, float* alpha, float* cv,
                       void* ws, size_t ws_bytes, void* stream);
/* given d_cv: ds[b,t] = alpha (enc.d_cv - cv.d_cv); dq[b,:] = sum_t ds enc[b,t,:].  One read of enc_states.
 * d_enc is NOT touched here: it is produced once per train step by the deferred batched GEMM in
 * astk_decoder_bwd (d_enc[b] = alpha_b^T d_cv_b + ds_b^T q_b over all steps). */
int astk_attn_step_bwd(int B, int T, int H, const float* enc, const float* alpha, const float* cv,
                       const float* d_cv, float* ds, float* dq, void* ws, size_t ws_bytes, void* stream);

/* ---------------------------------------------------------------- decoder loop  (seq2seq.py:318-333, 361-473)
 * embed(+dropout) -> [emb; ht] (input feeding) -> n_layers LSTM(H) -> attention -> ht = tanh(Wc[cv;h]+bc)
 * -> logits = Wo ht + bo -> argmax feedback when not teacher-forced (quirk Q4) -> class-weighted softmax-CE
 * with denominator B (quirk Q6), summed over the L-1 steps. */
typedef struct {
  size_t struct_size;  /* sizeof(astk_decoder_desc), see astk_cnn_desc.struct_size */
  int B, L, T, H, E, A, V, n_layers;
  /* optional features of the reference's model (rnn_config; zero = the shipped configs).  Any of them set routes the loop through
   * the per-launch kernels (astk_decoder_path returns 0): the persistent loop is built for the shipped model only. */
  int n_attn;        /* attention heads on the same decoder state (seq2seq.py:107-121, 381-383); 0 or 1 = one; context W is (A,(n_attn+1)H) */
  int no_feed_attn;  /* rnn_config.feed_attn = false (seq2seq.py:369-374): the decoder LSTM input is the embedding alone (layer-0 in = E) */
  int ln;            /* rnn_config.ln (seq2seq.py:141-143, 200-202): L.LayerNormalization(H) behind every decoder LSTM's dropped output */
  int loss_rows;     /* denominator of the per-step cross-entropy mean (quirk Q6: the batch size); 0 = B.  Set when this call scores a
                        SLICE of a larger batch (the library's own row split of batches the persistent loop cannot hold in one launch) */
  const int32_t* use_truth_host; /* optional HOST copy of the use_truth flags handed to astk_decoder_fwd(_ex) (L-1 entries), or NULL.  A hint for
                        the per-launch loop only: it then computes logits inside the loop just for the steps whose argmax is fed back and
                        scores every step with one product and one softmax-CE launch behind the loop.  Read during the call, not kept. */
  int precision;       /* ASTK_PREC_*: the batched products around the loop (encA, weight gradients, d_enc); the loop itself is IEEE f32 */
  int gemm_operands;   /* ASTK_OPERANDS_* (0 = process default) */
  float* status_dst;   /* optional (NULL = none): astk_decoder_fwd(_ex) also leaves a copy of the persistent kernels' status word there (the
                          Python shim passes &loss[1]), written by the kernel that writes the loss when the persistent loop runs --
                          astk_persist_status_snapshot without a launch of its own.  Ignored by the backward calls. */
  void* zero_ptr;      /* optional (NULL = none): zero_bytes bytes (16-byte aligned, a multiple of 4) that astk_decoder_bwd(_phase)(_ex) zeroes IN
                          FRONT of everything it accumulates -- the Python shim passes the gradient arena when cleargrads() was called since the
                          last backward pass (model.cleargrads() sits between forward_loss and backward in nn.py:175-189): the persistent loop's
                          launcher does it with the fill launch it has anyway, the other paths with a fill of their own.  Ignored by the forward
                          calls and by ASTK_DEC_BWD_PARAMS. */
  size_t zero_bytes;
  int side_wgs;        /* ASTK_DEC_BWD_PARAMS only: cap on the workgroups of every batched product of the phase (0 = none).  A caller that runs the
                          phase on a second stream beside the encoder's backward recurrence passes the CUs that recurrence leaves free, so that
                          these launches can never keep its grid from becoming resident */
  int deterministic;   /* backward calls: 1 = weight-gradient split tiles summed in a fixed order (see astk_cnn_desc.deterministic); 0 = process default */
} astk_decoder_desc;

typedef struct {
  const float* embed;                         /* (V,E) */
  astk_lstm_params lstm[ASTK_MAX_RNN_LAYERS]; /* layer 0 in = E+A, others in = H */
  const float* Wa; const float* ba;           /* attn_Wa (H,H) */
  const float* Wc; const float* bc;           /* context (A,2H) */
  const float* Wo; const float* bo;           /* out     (V,A)  */
  const float* class_weight;                  /* (V) : mask_pad_id, seq2seq.py:152-156 */
  const float* Wa_x[ASTK_MAX_ATTN - 1];       /* attn_Wa1.. (H,H) when n_attn > 1 */
  const float* ba_x[ASTK_MAX_ATTN - 1];
  const float* ln_gamma[ASTK_MAX_RNN_LAYERS]; /* L{i}_dec_ln (H) when ln */
  const float* ln_beta[ASTK_MAX_RNN_LAYERS];
} astk_decoder_params;

typedef struct {
  float* d_embed;
  astk_lstm_grads lstm[ASTK_MAX_RNN_LAYERS];
  float* dWa; float* dba;
  float* dWc; float* dbc;
  float* dWo; float* dbo;
  float* dWa_x[ASTK_MAX_ATTN - 1];
  float* dba_x[ASTK_MAX_ATTN - 1];
  float* d_ln_gamma[ASTK_MAX_RNN_LAYERS];
  float* d_ln_beta[ASTK_MAX_RNN_LAYERS];
} astk_decoder_grads;

size_t astk_decoder_workspace_bytes(const astk_decoder_desc* d);
/* enc (B,T,H); c0,h0 (n_layers,B,H) initial states (zeros for layers the encoder does not seed);
 * y (B,L) int32 targets; use_truth (L-1) int32 flags; emb_mask NULL or (L-1,B,E); rnn_masks NULL or
 * (n_layers,L-1,B,H); outputs: loss (1) = sum over steps, pred (L-1,B) int32 argmax per step (may be NULL). */
int astk_decoder_fwd(const astk_decoder_desc* d, const astk_decoder_params* p, const float* enc,
                     const float* c0, const float* h0, const int32_t* y, const int32_t* use_truth,
                     const float* emb_mask, const float* rnn_masks, float* loss, int32_t* pred,
                     void* ws, size_t ws_bytes, void* stream);
/* gradients of `loss` (upstream 1.0).  d_enc (B,T,H), d_c0, d_h0 (n_layers,B,H) are written;
 * parameter gradients are ACCUMULATED into g. */
int astk_decoder_bwd(const astk_decoder_desc* d, const astk_decoder_params* p, const astk_decoder_grads* g,
                     const float* enc, const float* c0, const float* h0, const int32_t* y,
                     const float* emb_mask, const float* rnn_masks,
                     float* d_enc, float* d_c0, float* d_h0, void* ws, size_t ws_bytes, void* stream);
/* The loop with the two per-call options of the reference's train step:
 *   out_mask  NULL or (L-1,B,V) scaled keep-masks of dropout.out (seq2seq.py:394: dropout on the LOGITS; argmax feedback and the
 *             loss both see the dropped logits);
 *   targets   NULL or (B,L) int32: the class ids that are SCORED at step s (column s+1), when they differ from the tokens that are
 *             FED (y): forward_loss's random_out replacement (seq2seq.py:456-465), drawn by the caller on the host.
 * astk_decoder_fwd / astk_decoder_bwd_phase are these with both NULL. */
int astk_decoder_fwd_ex(const astk_decoder_desc* d, const astk_decoder_params* p, const float* enc,
                        const float* c0, const float* h0, const int32_t* y, const int32_t* use_truth,
                        const float* emb_mask, const float* rnn_masks, const float* out_mask, const int32_t* targets,
                        float* loss, int32_t* pred, void* ws, size_t ws_bytes, void* stream);
/* The same backward in two phases, for callers that overlap them: ASTK_DEC_BWD_CHAIN runs the reversed loop and writes
 * everything the encoder's backward needs (d_enc, d_c0, d_h0); ASTK_DEC_BWD_PARAMS accumulates the parameter gradients from
 * what the chain phase left in `ws` -- it only has to be ordered after the chain phase (an event), so it can run on a second
 * stream beside the encoder's latency-bound backward recurrence, which leaves most of the device idle.  The caller joins that
 * stream before it reads g or reuses ws.  ASTK_DEC_BWD_ALL = astk_decoder_bwd. */
enum { ASTK_DEC_BWD_ALL = 0, ASTK_DEC_BWD_CHAIN = 1, ASTK_DEC_BWD_PARAMS = 2 };
int astk_decoder_bwd_phase(const astk_decoder_desc* d, const astk_decoder_params* p, const astk_decoder_grads* g,
                           const float* enc, const float* c0, const float* h0, const int32_t* y,
                           const float* emb_mask, const float* rnn_masks,
                           float* d_enc, float* d_c0, float* d_h0, void* ws, size_t ws_bytes, int phase, void* stream);
int astk_decoder_bwd_phase_ex(const astk_decoder_desc* d, const astk_decoder_params* p, const astk_decoder_grads* g,
                              const float* enc, const float* c0, const float* h0, const int32_t* y,
                              const float* emb_mask, const float* rnn_masks, const float* out_mask,
                              float* d_enc, float* d_c0, float* d_h0, void* ws, size_t ws_bytes, int phase, void* stream);
/* eval-mode single step for predict()/beam (seq2seq.py:361-396 under train=False): states (n_layers,B,H)
 * and ht (B,A) are updated in place; logits (B,V) and alpha (B,T) written. */
int astk_decoder_step_infer(const astk_decoder_desc* d, const astk_decoder_params* p, const float* enc,
                            float* c, float* h, float* ht, const int32_t* tokens, float* logits, float* alpha,
                            int32_t* argmax, void* ws, size_t ws_bytes, void* stream);

/* ---------------------------------------------------------------- normalisation layers of the optional encoder variants
 * L.LayerNormalization(units) behind an LSTM (rnn_config.ln; seq2seq.py:85-87, 200-202): rows x n, row strides ld*, per-row mean and
 * BIASED variance, y = (x - mu) / sqrt(var + eps) * gamma + beta (eps 1e-6, the link's default).  Backward recomputes the statistics
 * from x; dgamma / dbeta are ACCUMULATED (NULL: skipped), dx written (NULL: skipped). */
int astk_layernorm_fwd(int rows, int n, const float* x, long ldx, const float* gamma, const float* beta, float eps, float* y, long ldy,
                       void* stream);
int astk_layernorm_bwd(int rows, int n, const float* x, long ldx, const float* gamma, float eps, const float* dy, long lddy, float* dx,
                       long lddx, float* dgamma, float* dbeta, void* stream);
/* The projection between encoder layers of rnn_config.linear_proj (seq2seq.py:89-99, 280-286): out_t = relu(BN(z_t)) for every time
 * step t, z (T,B,C) = the Linear's output (a GEMM of the caller).  The reference calls its BatchNormalization link once per step on
 * a (B,C) matrix: batch statistics over the B rows OF THAT STEP, running averages advanced T times in step order (m = B samples).
 * stats (T,2,C) receives every step's mean and biased variance (read by the backward); train = 0 uses the running statistics. */
int astk_step_bn_relu_fwd(int T, int B, int C, const float* z, const float* gamma, const float* beta, float* avg_mean, float* avg_var,
                          float eps, float decay, int train, float* out, float* stats, void* stream);
/* dz written; dgamma / dbeta ACCUMULATED. */
int astk_step_bn_relu_bwd(int T, int B, int C, const float* z, const float* stats, const float* gamma, float eps, const float* out,
                          const float* d_out, float* dz, float* dgamma, float* dbeta, void* stream);

/* ---------------------------------------------------------------- softmax cross-entropy  (seq2seq.py:468-470)
 * rows = B: loss_rows[b] = -w[t_b] log_softmax(x_b)[t_b] / B ; dlogits = w[t_b](softmax - onehot)/B written in
 * place of logits; argmax (first maximum) written when non-NULL.  (Chainer-sem A6) */
int astk_softmax_ce_fwd(int B, int V, long ld, float* logits_inout, const int32_t* targets, long t_stride,
                        const float* class_weight, float inv_count, float* loss_rows, int32_t* argmax, void* stream);

/* ---------------------------------------------------------------- optimizer  (nn.py:81-119, Chainer-sem A7/A8)
 * One flat parameter / gradient buffer.  ONE norm launch at a time per process: astk_grad_sqnorm(_scaled) folds its per-block partial
 * sums through a process-wide scratch (no zeroing launch in front, a block-order sum whatever order the blocks finish in), so calls must
 * be ordered on one stream at a time -- which a train step's single optimizer is; astk_persist_status(.., reset = 1) also re-arms it.
 * sqnorm[0] = sum (g + l2*p)^2 in float64 (the clip norm of hook order
 * WeightDecay -> GradientClipping); the step applies decay, the clip rate min(1, clip/sqrt(sqnorm)) and
 * AMSGrad-Adam with lr_t = alpha*sqrt(1-b2^t)/(1-b1^t) computed by the caller.  Both update kernels leave p (and the moments)
 * untouched while the persistent kernels' sticky status word is non-zero (a kernel of the step timed out: the gradients are
 * garbage, and the host only learns of it when it reads the loss back).  The word stays set until astk_persist_status(.., reset = 1):
 * a caller must read the status next to the loss at least once per step (the Python shim's loss.data / float(loss) do, and raise) --
 * one that never looks would silently drop every later update while its own step counter advances. */
int astk_grad_sqnorm(const float* g, const float* p, float l2, size_t n, double* sqnorm, void* stream);
int astk_decay_clip_amsgrad_step(float* p, const float* g, float* m, float* v, float* vhat, size_t n,
                                 float l2, float clip, const double* sqnorm, float lr_t, float beta1, float beta2,
                                 float eps, int amsgrad, void* stream);
int astk_decay_clip_sgd_step(float* p, const float* g, size_t n, float l2, float clip, const double* sqnorm,
                             float lr, void* stream);
/* The same three with the gradient read as grad_scale * g: under data parallelism g holds the SUM over the replicas after the
 * all-reduce and grad_scale = 1/world makes it the mean without a separate pass over the buffer (the product is rounded before
 * the decay term is added, exactly as a separate scaling pass would round it). */
int astk_grad_sqnorm_scaled(const float* g, const float* p, float grad_scale, float l2, size_t n, double* sqnorm, void* stream);
int astk_decay_clip_amsgrad_step_scaled(float* p, const float* g, float* m, float* v, float* vhat, size_t n, float grad_scale,
                                        float l2, float clip, const double* sqnorm, float lr_t, float beta1, float beta2,
                                        float eps, int amsgrad, void* stream);
int astk_decay_clip_sgd_step_scaled(float* p, const float* g, size_t n, float grad_scale, float l2, float clip,
                                    const double* sqnorm, float lr, void* stream);

/* ---------------------------------------------------------------- utilities
 * Dropout keep-masks (Chainer-sem A5): out[i] = (u_i >= ratio) / (1-ratio), u from a counter-based hash RNG. */
int astk_fill_dropout_mask(float* out, size_t n, float ratio, uint64_t seed, uint64_t offset, void* stream);
/* chainer.optimizer.GradientNoise behind the other two hooks (nn.py:108-110; hooks run in insertion order WeightDecay -> GradientClipping ->
 * GradientNoise, Chainer-sem A7): g <- clip(g * grad_scale + l2 * p) + sigma * N(0,1) in place, sigma^2 = eta / (1 + t)^0.55 (t = updates done before this one; the caller passes sigma).  The update
 * call that follows takes the finished gradient (grad_scale 1, l2 0, clip off).  Chainer draws from its unseeded global RNG (quirk Q7);
 * this is a counter-based stream (seed, offset) consuming n / 2 counters. */
int astk_decay_clip_noise(float* g, const float* p, size_t n, float grad_scale, float l2, float clip, const double* sqnorm, float sigma,
                          uint64_t seed, uint64_t offset, void* stream);
/* out[i] = 1 + sigma*N(0,1): the multiplicative speech noise of seq2seq.py:300-302, generated on device. */
int astk_fill_normal(float* out, size_t n, float mean, float sigma, uint64_t seed, uint64_t offset, void* stream);
/* Several fills of those two kinds in ONE launch (a train step's speech noise and dropout masks): every element gets exactly the value
 * astk_fill_normal / astk_fill_dropout_mask give it for the same (seed, offset).  kind ASTK_RAND_DROPOUT: a = ratio; ASTK_RAND_NORMAL:
 * a = mean, b = sigma.  Replaces the host-side np.random.normal of seq2seq.py:300 and Chainer's per-call F.dropout masks (A5). */
#define ASTK_RAND_SEG_MAX 8
#define ASTK_RAND_DROPOUT 0
#define ASTK_RAND_NORMAL 1
typedef struct {
  float* out;
  size_t n;
  int kind;
  float a, b;
  uint64_t seed, offset;
} astk_rand_seg;
int astk_fill_random(const astk_rand_seg* segs, int n_segs, void* stream);
/* ... and, in the same launch, n_words (<= ASTK_RAND_WORDS_MAX) HOST values written to the device array words_dst: the step's
 * teacher-forcing flags (seq2seq.py:431-436 draws them on the host) travel in the kernel arguments instead of a copy of their own.
 * n_segs may be 0. */
#define ASTK_RAND_WORDS_MAX 256
int astk_fill_random_ex(const astk_rand_seg* segs, int n_segs, const int32_t* words, int n_words, int32_t* words_dst, void* stream);
int astk_scale_f32(float* x, size_t n, float s, void* stream);
/* dst += src (n floats); dst[c] += sum_r src[r*lds + c] (bias gradients; the sum over time of the linear_proj encoder's reverse-stack
 * input gradient, whose input is one frame fed at every step). */
int astk_add_f32(float* dst, const float* src, size_t n, void* stream);
int astk_colsum_add_f32(float* dst, const float* src, long lds, int rows, int cols, void* stream);
/* init_decoder_state (seq2seq.py:318-333) and its backward as ONE launch each: decoder layer k < n starts from [fwd_k ; rev_k] of the
 * encoder's final states.  enc_c / enc_h: (nd, nl_enc, B, h) as astk_lstm_stack_fwd leaves them; dec_c / dec_h: (>= n, B, nd*h).
 * to_decoder != 0: dec[k][b][d*h + u] = enc[d][k][b][u]; to_decoder == 0 (gradients): enc[d][k][b][u] = dec[k][b][d*h + u].
 * Layers >= n of either side are not touched. */
int astk_bridge_states(float* dec_c, float* dec_h, float* enc_c, float* enc_h, int nd, int nl_enc, int n, int B, int h, int to_decoder,
                       void* stream);
/* Frame zeroing of the training loader (dataloader.py:83-93, `zero_input`), on the padded device batch X (B,T,D): utterance b of true
 * length lengths[b] gets int(rate * lengths[b]) frames zeroed, drawn with replacement from [0, lengths[b]) like
 * np.random.choice(np.arange(T_b), size=n).  The reference's draw is unseeded (quirk Q7); this one is a counter-based stream
 * (seed, offset), consuming B*T counters per call. */
int astk_zero_frames(float* X, int B, int T, int D, const int32_t* lengths, double rate, uint64_t seed, uint64_t offset, void* stream);
/* The draws astk_zero_frames makes for the same (lengths, rate, seed, offset), written out instead of applied: counts[b] =
 * int(rate * lengths[b]) (double precision, as Python evaluates it), idx[b*max_draws + i] = the i-th frame drawn for utterance b.
 * Lets a test feed the device's stream to the oracle's _drop_frames (oracle/loader_ref.py) and compare whole batches bit for bit. */
int astk_zero_frames_draws(int B, int T, const int32_t* lengths, double rate, uint64_t seed, uint64_t offset, int32_t* idx, int max_draws,
                           int32_t* counts, void* stream);
/* One wavefront that keeps `stream` busy for `usec` microseconds (<= 100 000) and then increments *flag (may be NULL).  For probing
 * whether two streams really execute concurrently: HIP multiplexes streams onto a few hardware queues (GPU_MAX_HW_QUEUES, 4 by
 * default), and two streams that share one are serialised whatever their events say (ast_amd/seq2seq.py:_cu_streams). */
int astk_spin(unsigned usec, unsigned* flag, void* stream);

/* Persistent-kernel health.  The encoder / decoder recurrences run as single launches whose workgroups hand activations to each
 * other (lstm_persist.hip, decoder_persist.hip); that needs the whole grid resident (one workgroup per CU), which the launchers
 * check against the device's CU count but cannot guarantee against other tenants of the GPU (RCCL kernels under data
 * parallelism, another process).  Every spin is bounded: a time-out drains the grid and sets a bit in a STICKY status word
 * (1 encoder fwd, 2 encoder bwd, 4 decoder fwd, 8 decoder bwd, 16 a peer rank reported one of these: astk_persist_status_merge); the
 * results of such a step are garbage.
 *   astk_persist_status_snapshot  enqueues a copy of the word (as a float) to *dst on `stream`: the Python shim places it next to
 *                                 the loss scalar, so the loss read-back of nn.py:189 sees it without an extra synchronisation;
 *   astk_persist_status           synchronises the device, returns the word in *mask_out (may be NULL) and clears it if `reset`.
 * astk_set_tuning("persist.spin_limit", polls) shrinks the spin bound; tests use it to force a time-out. */
int astk_persist_status_snapshot(float* dst, void* stream);
/* Data parallelism: `summed` = the SUM over all ranks of the words astk_persist_status_snapshot wrote (the Python shim appends the
 * word to the last gradient range, so the gradient all-reduce carries it: ast_amd/dist.py).  Non-zero: some rank's step is garbage --
 * bit 16 is set in THIS rank's sticky word, so that the update kernels enqueued behind it skip on every rank alike and every rank
 * raises at its next loss read-back instead of waiting in the next collective for a peer that has already failed. */
int astk_persist_status_merge(const float* summed, void* stream);
/* Which path a shape takes on the current device (so that a silent fall-back shows up in logs / bench.py's JSON line):
 *   astk_lstm_stack_path  1 = persistent wavefront kernels (all T steps of all cells in one launch; stacks with more cells than CUs: one launch
 *                         per group of layers), 2 = the hoisted form of the same kernels (h = 1024: one launch per layer, the input projection
 *                         and the gradient for the layer below as batched products between the launches), 0 = one fused-cell launch per step
 *   astk_decoder_path     0 = per-launch decoder loop; otherwise bit 0 = persistent loop, bit 1 = attention phase specialised for
 *                         H = 512 / chunk <= 32, bit 2 = the batch runs as TWO persistent launches over halves of its rows (more than
 *                         32 rows at the shipped width: the decoder couples no batch rows, the halves share nothing but the weights
 *                         and the loss's 1/B), bit 4 (alone) = the wide decoder of configs[4] (H = A = 1024, E = 128, one layer, up to 32 rows,
 *                         T'' <= 256 at 32 rows) on decoder_wide.hip's persistent loops, one launch for the forward and one for the backward
 *                         (with bit 2: two of each, over halves of more than 32 rows); bits 8.. = number of decoder layers fused into the persistent kernels */
int astk_lstm_stack_path(const astk_lstm_stack_desc* d);
/* CUs the persistent recurrence launches of this shape leave FREE on the current device (0: not the persistent path, or none): what a
 * caller may pass as `side_wgs` for work it runs on a second stream beside the recurrences (astk_decoder_desc.side_wgs). */
int astk_lstm_stack_free_cus(const astk_lstm_stack_desc* d);
int astk_decoder_path(const astk_decoder_desc* d);
int astk_persist_status(unsigned* mask_out, int reset);
int astk_device_cu_count(void);

/* Optional per-kernel HIP-event timing on the launch stream (bench.py's roofline legs; off by default).
 * astk_prof_end: res[0..1] attention-scan fwd (ms, launches); [2..3] attention-scan bwd; [4..6] GEMMs (ms, launches, flops);
 * [7..8] fused LSTM cells (ms, launches); [9..11] attention-scan phase inside the persistent decoder forward measured with
 * in-kernel timestamps (mean us per step over workgroups, slowest workgroup's mean us, launches); [12..14] same for the
 * backward; [16..19] persistent decoder forward / backward kernels (ms, launches).  res must hold 24 doubles.  Synchronises the device. */
int astk_prof_begin(void);
int astk_prof_end(double* res);

#ifdef __cplusplus
}
#endif
#endif /* ASTK_H */
