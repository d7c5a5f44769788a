// Normalisation layers of the reference's optional model features, outside the tuned default path:
//   * L.LayerNormalization behind every LSTM when rnn_config.ln is set (seq2seq.py:85-87, 141-143, 200-202): hs = LN(dropout(LSTM(hs)))
//   * the per-time-step BatchNormalization + ReLU of the linear_proj encoder (seq2seq.py:280-286): the link is called once per time
//     step on a (B, units) matrix, so its statistics are those of the B rows of that step, and its running averages advance once
//     per step, in step order.
// Row-wise / per-(step, channel) reductions over a few hundred values: one wavefront or one thread each; HBM-bound and tiny next to
// the recurrences they sit between.
#include "common.h"
#include <algorithm>

namespace astk {

namespace {

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  return v;
}

// one wavefront per row: y = (x - mu) / sqrt(var + eps) * gamma + beta, biased variance (F.layer_normalization)
__global__ __launch_bounds__(256) void k_layernorm_fwd(int rows, int n, const float* __restrict__ x, long ldx, const float* __restrict__ gamma,
                                                       const float* __restrict__ beta, float eps, float* __restrict__ y, long ldy) {
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (row >= rows) return;
  const float* xr = x + (long)row * ldx;
  float s = 0.f;
  for (int c = lane; c < n; c += 64) s += xr[c];
  const float mu = wave_sum(s) / (float)n;
  float q = 0.f;
  for (int c = lane; c < n; c += 64) { const float d = xr[c] - mu; q += d * d; }
  const float inv = 1.f / sqrtf(wave_sum(q) / (float)n + eps);
  float* yr = y + (long)row * ldy;
  for (int c = lane; c < n; c += 64) yr[c] = (xr[c] - mu) * inv * gamma[c] + beta[c];
}

// dx = inv_std (g - mean(g) - x_hat mean(g x_hat)), g = dy gamma; one wavefront per row (statistics recomputed from x)
__global__ __launch_bounds__(256) void k_layernorm_bwd_x(int rows, int n, const float* __restrict__ x, long ldx, const float* __restrict__ gamma,
                                                         float eps, const float* __restrict__ dy, long lddy, float* __restrict__ dx, long lddx) {
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (row >= rows) return;
  const float* xr = x + (long)row * ldx;
  const float* gr = dy + (long)row * lddy;
  float s = 0.f;
  for (int c = lane; c < n; c += 64) s += xr[c];
  const float mu = wave_sum(s) / (float)n;
  float q = 0.f;
  for (int c = lane; c < n; c += 64) { const float d = xr[c] - mu; q += d * d; }
  const float inv = 1.f / sqrtf(wave_sum(q) / (float)n + eps);
  float sg = 0.f, sgx = 0.f;
  for (int c = lane; c < n; c += 64) {
    const float g = gr[c] * gamma[c], xh = (xr[c] - mu) * inv;
    sg += g; sgx += g * xh;
  }
  sg = wave_sum(sg) / (float)n;
  sgx = wave_sum(sgx) / (float)n;
  float* dr = dx + (long)row * lddx;
  for (int c = lane; c < n; c += 64) {
    const float g = gr[c] * gamma[c], xh = (xr[c] - mu) * inv;
    dr[c] = inv * (g - sg - xh * sgx);
  }
}

// dgamma[c] += sum_rows dy x_hat, dbeta[c] += sum_rows dy: one block per 64 columns x a slab of rows; per-row statistics recomputed by
// the 64 lanes of a wave reading the row (rows are short: n <= a few thousand), then column partials folded through LDS
__global__ __launch_bounds__(256) void k_layernorm_bwd_params(int rows, int n, const float* __restrict__ x, long ldx, float eps,
                                                              const float* __restrict__ dy, long lddy, float* __restrict__ dgamma,
                                                              float* __restrict__ dbeta, int rows_per_block) {
  __shared__ float sg[4][64], sb[4][64];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int c = blockIdx.x * 64 + lane;
  const int r0 = blockIdx.y * rows_per_block, r1 = min(rows, r0 + rows_per_block);
  float ag = 0.f, ab = 0.f;
  for (int row = r0 + wave; row < r1; row += 4) {
    const float* xr = x + (long)row * ldx;
    float s = 0.f;
    for (int k = lane; k < n; k += 64) s += xr[k];
    const float mu = wave_sum(s) / (float)n;
    float q = 0.f;
    for (int k = lane; k < n; k += 64) { const float d = xr[k] - mu; q += d * d; }
    const float inv = 1.f / sqrtf(wave_sum(q) / (float)n + eps);
    if (c < n) {
      const float g = dy[(long)row * lddy + c];
      ag += g * (xr[c] - mu) * inv;
      ab += g;
    }
  }
  sg[wave][lane] = ag; sb[wave][lane] = ab;
  __syncthreads();
  if (wave == 0 && c < n) {
    atomicAdd(&dgamma[c], sg[0][lane] + sg[1][lane] + sg[2][lane] + sg[3][lane]);
    atomicAdd(&dbeta[c], sb[0][lane] + sb[1][lane] + sb[2][lane] + sb[3][lane]);
  }
}

// ---- per-step BatchNorm + ReLU (linear_proj).  z, out: (T, B, C); stats: (T, 2, C) = batch mean and BIASED variance of every step
// (train) -- one thread per (step, channel), a loop over the B rows.
__global__ __launch_bounds__(256) void k_step_bn_relu_fwd(int T, int B, int C, const float* __restrict__ z, const float* __restrict__ gamma,
                                                          const float* __restrict__ beta, const float* __restrict__ avg_mean,
                                                          const float* __restrict__ avg_var, float eps, int train, float* __restrict__ out,
                                                          float* __restrict__ stats) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x, t = blockIdx.y;
  if (c >= C) return;
  const float* zt = z + (long)t * B * C + c;
  float mean, var;
  if (train) {
    float s = 0.f;
    for (int b = 0; b < B; ++b) s += zt[(long)b * C];
    mean = s / (float)B;
    float q = 0.f;
    for (int b = 0; b < B; ++b) { const float d = zt[(long)b * C] - mean; q += d * d; }
    var = q / (float)B;
    stats[((long)t * 2 + 0) * C + c] = mean;
    stats[((long)t * 2 + 1) * C + c] = var;
  } else {
    mean = avg_mean[c];
    var = avg_var[c];
  }
  const float sc = gamma[c] / sqrtf(var + eps), sh = beta[c] - mean * sc;
  float* ot = out + (long)t * B * C + c;
  for (int b = 0; b < B; ++b) ot[(long)b * C] = fmaxf(zt[(long)b * C] * sc + sh, 0.f);
}
// running statistics: T sequential updates per channel, in step order (Chainer-sem A4 with m = B samples per call)
__global__ void k_step_bn_running(int T, int B, int C, const float* __restrict__ stats, float* __restrict__ avg_mean, float* __restrict__ avg_var,
                                  float decay) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  const float adjust = (float)B / fmaxf((float)B - 1.f, 1.f);
  float am = avg_mean[c], av = avg_var[c];
  for (int t = 0; t < T; ++t) {
    am = decay * am + (1.f - decay) * stats[((long)t * 2 + 0) * C + c];
    av = decay * av + ((1.f - decay) * adjust) * stats[((long)t * 2 + 1) * C + c];
  }
  avg_mean[c] = am;
  avg_var[c] = av;
}
// backward of relu(BN_t(z_t)): g = d_out (out > 0); dz = gamma inv_std (g - (x_hat dgamma_t + dbeta_t) / B); dgamma += sum_t dgamma_t ...
__global__ __launch_bounds__(256) void k_step_bn_relu_bwd(int T, int B, int C, const float* __restrict__ z, const float* __restrict__ stats,
                                                          const float* __restrict__ gamma, float eps, const float* __restrict__ out,
                                                          const float* __restrict__ d_out, float* __restrict__ dz, float* __restrict__ dgamma,
                                                          float* __restrict__ dbeta) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x, t = blockIdx.y;
  if (c >= C) return;
  const long base = (long)t * B * C + c;
  const float mean = stats[((long)t * 2 + 0) * C + c], inv = 1.f / sqrtf(stats[((long)t * 2 + 1) * C + c] + eps);
  float sg = 0.f, sgx = 0.f;
  for (int b = 0; b < B; ++b) {
    const long i = base + (long)b * C;
    const float g = out[i] > 0.f ? d_out[i] : 0.f;
    sg += g;
    sgx += g * (z[i] - mean) * inv;
  }
  const float sc = gamma[c] * inv, invm = 1.f / (float)B;
  for (int b = 0; b < B; ++b) {
    const long i = base + (long)b * C;
    const float g = out[i] > 0.f ? d_out[i] : 0.f;
    dz[i] = sc * (g - ((z[i] - mean) * inv * sgx + sg) * invm);
  }
  atomicAdd(&dgamma[c], sgx);
  atomicAdd(&dbeta[c], sg);
}

__global__ void k_mul_rows(float* __restrict__ x, long ldx, const float* __restrict__ m, long ldm, int rows, int cols) {
  const long n = (long)rows * cols;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    const int r = (int)(i / cols), c = (int)(i % cols);
    x[(long)r * ldx + c] *= m[(long)r * ldm + c];
  }
}

}  // namespace

int layernorm_fwd_launch(int rows, int n, const float* x, long ldx, const float* gamma, const float* beta, float eps, float* y, long ldy,
                         hipStream_t s) {
  ASTK_CHECK(rows > 0 && n > 0 && x && gamma && beta && y && ldx >= n && ldy >= n, "layernorm_fwd: bad arguments");
  hipLaunchKernelGGL(k_layernorm_fwd, dim3(cdiv(rows, 4)), dim3(256), 0, s, rows, n, x, ldx, gamma, beta, eps, y, ldy);
  ASTK_LAUNCH_CHECK();
  return 0;
}

int layernorm_bwd_launch(int rows, int n, const float* x, long ldx, const float* gamma, float eps, const float* dy, long lddy, float* dx,
                         long lddx, float* dgamma, float* dbeta, hipStream_t s) {
  ASTK_CHECK(rows > 0 && n > 0 && x && gamma && dy && ldx >= n && lddy >= n, "layernorm_bwd: bad arguments");
  if (dx) {
    ASTK_CHECK(lddx >= n, "layernorm_bwd: bad dx stride");
    hipLaunchKernelGGL(k_layernorm_bwd_x, dim3(cdiv(rows, 4)), dim3(256), 0, s, rows, n, x, ldx, gamma, eps, dy, lddy, dx, lddx);
    ASTK_LAUNCH_CHECK();
  }
  if (dgamma && dbeta) {
    const int gx = cdiv(n, 64);
    int gy = 1024 / gx;
    if (gy < 1) gy = 1;
    const int rpb = std::max(8, cdiv(rows, gy));
    hipLaunchKernelGGL(k_layernorm_bwd_params, dim3(gx, cdiv(rows, rpb)), dim3(256), 0, s, rows, n, x, ldx, eps, dy, lddy, dgamma, dbeta, rpb);
    ASTK_LAUNCH_CHECK();
  }
  return 0;
}

int mul_rows_launch(float* x, long ldx, const float* m, long ldm, int rows, int cols, hipStream_t s) {
  long b = ((long)rows * cols + 255) / 256;
  if (b > 2048) b = 2048;
  if (b < 1) b = 1;
  hipLaunchKernelGGL(k_mul_rows, dim3((unsigned)b), dim3(256), 0, s, x, ldx, m, ldm, rows, cols);
  ASTK_LAUNCH_CHECK();
  return 0;
}

}  // namespace astk

using namespace astk;

extern "C" {

int astk_layernorm_fwd(int rows, int n, const float* x, long ldx, const float* gamma, const float* beta, float eps, float* y, long ldy,
                       void* stream) {
  return layernorm_fwd_launch(rows, n, x, ldx, gamma, beta, eps, y, ldy, (hipStream_t)stream);
}

int astk_layernorm_bwd(int rows, int n, const float* x, long ldx, const float* gamma, float eps, const float* dy, long lddy, float* dx,
                       long lddx, float* dgamma, float* dbeta, void* stream) {
  return layernorm_bwd_launch(rows, n, x, ldx, gamma, eps, dy, lddy, dx, lddx, dgamma, dbeta, (hipStream_t)stream);
}

int astk_step_bn_relu_fwd(int T, int B, int C, const float* z, const float* gamma, const float* beta, float* avg_mean, float* avg_var,
                          float eps, float decay, int train, float* out, float* stats, void* stream) {
  hipStream_t s = (hipStream_t)stream;
  ASTK_CHECK(T > 0 && B > 0 && C > 0 && z && gamma && beta && avg_mean && avg_var && out && (stats || !train), "step_bn_relu_fwd: bad arguments");
  hipLaunchKernelGGL(k_step_bn_relu_fwd, dim3(cdiv(C, 256), T), dim3(256), 0, s, T, B, C, z, gamma, beta, avg_mean, avg_var, eps, train, out, stats);
  ASTK_LAUNCH_CHECK();
  if (train) {
    hipLaunchKernelGGL(k_step_bn_running, dim3(cdiv(C, 256)), dim3(256), 0, s, T, B, C, stats, avg_mean, avg_var, decay);
    ASTK_LAUNCH_CHECK();
  }
  return 0;
}

int astk_step_bn_relu_bwd(int T, int B, int C, const float* z, const float* stats, const float* gamma, float eps, const float* out,
                          const float* d_out, float* dz, float* dgamma, float* dbeta, void* stream) {
  ASTK_CHECK(T > 0 && B > 0 && C > 0 && z && stats && gamma && out && d_out && dz && dgamma && dbeta, "step_bn_relu_bwd: bad arguments");
  hipLaunchKernelGGL(k_step_bn_relu_bwd, dim3(cdiv(C, 256), T), dim3(256), 0, (hipStream_t)stream, T, B, C, z, stats, gamma, eps, out, d_out, dz,
                     dgamma, dbeta);
  ASTK_LAUNCH_CHECK();
  return 0;
}

}  // extern "C"
