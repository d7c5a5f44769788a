// Row-panel MFMA products for the recurrent steps (SURVEY.md K10/K11/K18/K19/K23/K24 and their
// backward counterparts): out[M<=~64, N] = sum_p A_p[M,K_p] W_p[N,K_p]^T with a fused epilogue.
//
// These products have M = batch (16-64 rows), so they are weight-streaming / latency bound, not
// MFMA bound.  Design: one workgroup owns a 16-row (32-row above 16 rows: two row tiles on the same weight fragments) x (16*NT)-column output tile and splits K over
// its 4 waves; operands go straight from global memory (L2 / MALL resident) into the
// v_mfma_f32_16x16x4_f32 operand registers with 16-byte loads -- lane (r = l&15, q = l>>4) loads
// floats [16s+4q, 16s+4q+4) of row r for k-block s and feeds them to 4 consecutive MFMAs, so that the
// MFMA's k index is (q, c) <-> k = 16s+4q+c for both operands.  No LDS staging (nothing is reused
// inside a workgroup); LDS is used once, for the 4-way K reduction before the epilogue.
//
// The LSTM cell kernels use NT = 4 tiles that are the four gates of 16 hidden units (Chainer's
// interleaved layout, row 4j+k), so each thread of the epilogue holds a,i,f,o of one (batch row, unit).
#include "common.h"

namespace astk {

namespace {

struct WRows {   // W row for (tile t, column j) = base + j*sj + t*st ; valid columns: j < jmax
  int base, sj, st;
};

// CH k-blocks per wave are loaded back to back (one exposed memory round trip per trip) before their MFMAs issue.
// MT: row tiles (16 rows each) of one workgroup that share the weight fragments -- these products stream their weights once per
// workgroup, so a batch of 17-32 rows as two workgroups per column tile reads every weight twice (round 3: MT = 2 above 16 rows).
// NW: waves of the workgroup that share the K range (4, or 8 when K is long: a wave's share is fetched in ceil(K / 16 / NW / CH) exposed
// round trips, and at K = 2176 / 4096 -- the H = 1024 decoder's cell products -- four waves need 5 / 4 of them).
template <int MT, int NT, int CH, int NW = 4>
__device__ __forceinline__ void tile_dot(const RowPair& pr, int m0, int M, WRows wr, int jvalid, int lane, int wave,
                                         f32x4 (&acc)[MT][NT]) {
  const int K = pr.K;
  if (K <= 0) return;
  const int r = lane & 15, q = lane >> 4;
  const float* ap[MT];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt) ap[mt] = pr.A + (long)min(m0 + 16 * mt + r, M - 1) * pr.lda + 4 * q;
  const int j = min(r, jvalid - 1);
  const float* wp[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t) wp[t] = pr.W + (long)(wr.base + j * wr.sj + t * wr.st) * pr.ldw + 4 * q;
  const int nblk = (K + 15) >> 4;
  f32x4 acc2[MT][NT];   // second accumulator chain: consecutive MFMAs never wait on each other's result
#pragma unroll
  for (int mt = 0; mt < MT; ++mt)
#pragma unroll
    for (int t = 0; t < NT; ++t) acc2[mt][t] = f32x4{0.f, 0.f, 0.f, 0.f};
  for (int sb = wave; sb < nblk; sb += NW * CH) {
    float4 av[MT][CH], wv[CH][NT];
#pragma unroll
    for (int i = 0; i < CH; ++i) {
      // branch-free: the address is clamped inside the row (K % 4 == 0), out-of-range k-blocks are zeroed by a select on the
      // A operand only (0 * w = 0) -- a `cond ? load : 0` would serialise every load behind a vmcnt(0)
      const int s = sb + NW * i;
      const int ko = min(16 * s + 4 * q, K - 4) - 4 * q;
      const bool v = (16 * s + 4 * q) < K;
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) {
        float4 x = *reinterpret_cast<const float4*>(ap[mt] + ko);
        x.x = v ? x.x : 0.f; x.y = v ? x.y : 0.f; x.z = v ? x.z : 0.f; x.w = v ? x.w : 0.f;
        av[mt][i] = x;
      }
#pragma unroll
      for (int t = 0; t < NT; ++t) wv[i][t] = *reinterpret_cast<const float4*>(wp[t] + ko);
    }
    // keep every load of the trip in front of its MFMAs (hipcc otherwise sinks each load next to its use to save VGPRs,
    // which turns the trip into CH dependent memory round trips)
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int i = 0; i < CH; ++i) {
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) {
#pragma unroll
        for (int t = 0; t < NT; ++t) acc[mt][t] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[mt][i].x, wv[i][t].x, acc[mt][t], 0, 0, 0);
#pragma unroll
        for (int t = 0; t < NT; ++t) acc2[mt][t] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[mt][i].y, wv[i][t].y, acc2[mt][t], 0, 0, 0);
#pragma unroll
        for (int t = 0; t < NT; ++t) acc[mt][t] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[mt][i].z, wv[i][t].z, acc[mt][t], 0, 0, 0);
#pragma unroll
        for (int t = 0; t < NT; ++t) acc2[mt][t] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[mt][i].w, wv[i][t].w, acc2[mt][t], 0, 0, 0);
      }
    }
  }
#pragma unroll
  for (int mt = 0; mt < MT; ++mt)
#pragma unroll
    for (int t = 0; t < NT; ++t) acc[mt][t] += acc2[mt][t];
}

// 4-wave K reduction.  D layout of the 16x16 MFMA: col = lane&15, row = (lane>>4)*4 + reg.
// After the call thread tid holds, for each tile t, the full sum of element (row = tid>>4, col = tid&15).
// (NW = 8: threads 256.. only contribute their partial sums; the caller's epilogue runs on threads 0..255.)
template <int NT, int NW = 4>
__device__ __forceinline__ void reduce_waves(f32x4 (&acc)[NT], float (&vals)[NT], float* red /* [NW][NT][256] */) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
#pragma unroll
  for (int t = 0; t < NT; ++t)
    *reinterpret_cast<f32x4*>(&red[((wave * NT + t) * 64 + lane) * 4]) = acc[t];
  __syncthreads();
  const int t256 = tid & 255;
  const int row = t256 >> 4, col = t256 & 15;
  const int src = ((row >> 2) * 16 + col) * 4 + (row & 3);
#pragma unroll
  for (int t = 0; t < NT; ++t) {
    float v = 0.f;
#pragma unroll
    for (int w = 0; w < NW; ++w) v += red[(w * NT + t) * 256 + src];
    vals[t] = v;
  }
}

template <int NT>
__device__ __forceinline__ void zero_acc(f32x4 (&acc)[NT]) {
#pragma unroll
  for (int t = 0; t < NT; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
}

// ------------------------------------------------------------------ generic row-panel GEMM
// One workgroup = 16 rows x 16 columns (N/16 x M/16 workgroups: these products are latency bound, so they are
// spread over as many CUs as possible and each wave's whole K share is fetched in at most a few round trips).
template <int MT, int NW>
__global__ __launch_bounds__(64 * NW) void rowgemm_kernel(RowGemmArgs a) {
  __shared__ __attribute__((aligned(16))) float red[NW * MT * 256];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int n0 = blockIdx.x * 16, m0 = blockIdx.y * 16 * MT;
  f32x4 acc[MT][1];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt) acc[mt][0] = f32x4{0.f, 0.f, 0.f, 0.f};
  const int jv = min(16, a.N - n0);
  for (int p = 0; p < a.npairs; ++p) tile_dot<MT, 1, 8, NW>(a.p[p], m0, a.M, WRows{n0, 1, 0}, jv, lane, wave, acc);
  f32x4 flat[MT];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt) flat[mt] = acc[mt][0];
  float vals[MT];
  reduce_waves<MT, NW>(flat, vals, red);
  if (NW > 4 && threadIdx.x >= 256) return;
  const int n = n0 + (threadIdx.x & 15);
#pragma unroll
  for (int mt = 0; mt < MT; ++mt) {
    const int row = m0 + 16 * mt + (threadIdx.x >> 4);
    if (row >= a.M || n >= a.N) continue;
    float v = vals[mt];
    if (a.bias) v += a.bias[n];
    if (a.addend) v += a.addend[(long)row * a.ld_add + n];
    if (a.act == ACT_TANH) v = tanhf(v);
    else if (a.act == ACT_DTANH) {
      const float y = a.aux[(long)row * a.ld_aux + n];
      v *= (1.f - y * y);
    }
    a.out[(long)row * a.ld_out + n] = v;
    if (a.out2) a.out2[(long)row * a.ld_out2 + n] = v;
    if (a.carry && n >= a.carry_col0) {
      const int j = n - a.carry_col0;
      const float y = a.carry_aux[(long)row * a.ld_carry_aux + j];
      float* c = a.carry + (long)row * a.ld_carry + j;
      *c = (v + *c) * (1.f - y * y);
    }
  }
}

// ------------------------------------------------------------------ LSTM cell forward (Chainer-sem A1)
struct CellFwdBatch {
  LstmCellFwdArgs c[8];
};

__device__ __forceinline__ float sigmoidf_(float x) { return 1.f / (1.f + expf(-x)); }

// One workgroup = 16 batch rows x 4 hidden units: the 16 MFMA columns are the 16 consecutive gate rows 4*j0 .. 4*j0+15 of
// Chainer's interleaved layout, so a cell spreads over (h/4) x (B/16) workgroups and each streams only 16 weight rows.
template <int MT, int NW>
__global__ __launch_bounds__(64 * NW) void lstm_cell_fwd_kernel(CellFwdBatch batch) {
  __shared__ __attribute__((aligned(16))) float red[NW * MT * 256];
  __shared__ __attribute__((aligned(16))) float zt[MT * 256];
  const LstmCellFwdArgs& a = batch.c[blockIdx.z];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int u0 = blockIdx.x * 4, m0 = blockIdx.y * 16 * MT;
  f32x4 acc[MT][1];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt) acc[mt][0] = f32x4{0.f, 0.f, 0.f, 0.f};
  const int jv = min(16, 4 * (a.h - u0));
  for (int p = 0; p < a.npairs; ++p) tile_dot<MT, 1, 8, NW>(a.p[p], m0, a.B, WRows{4 * u0, 1, 0}, jv, lane, wave, acc);
  f32x4 flat[MT];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt) flat[mt] = acc[mt][0];
  float v[MT];
  reduce_waves<MT, NW>(flat, v, red);
  if (threadIdx.x < 256) {
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) zt[mt * 256 + threadIdx.x] = v[mt];               // zt[tile][row][col], col = 4*unit + gate
  }
  __syncthreads();
  if (threadIdx.x >= 64 * MT) return;
  const int mt = threadIdx.x >> 6, t64 = threadIdx.x & 63;
  const int b = m0 + 16 * mt + (t64 >> 2), u = u0 + (t64 & 3);
  if (b >= a.B || u >= a.h) return;
  float4 z = *reinterpret_cast<const float4*>(&zt[mt * 256 + (t64 >> 2) * 16 + (t64 & 3) * 4]);
  if (a.zx) {
    const float4 zx = *reinterpret_cast<const float4*>(a.zx + (long)b * a.ld_zx + 4 * u);
    z.x += zx.x; z.y += zx.y; z.z += zx.z; z.w += zx.w;
  }
  if (a.bias) {
    const float4 bb = *reinterpret_cast<const float4*>(a.bias + 4 * u);
    z.x += bb.x; z.y += bb.y; z.z += bb.z; z.w += bb.w;
  }
  const float ga = tanhf(z.x), gi = sigmoidf_(z.y), gf = sigmoidf_(z.z), go = sigmoidf_(z.w);
  const float cp = a.c_prev ? a.c_prev[(long)b * a.h + u] : 0.f;
  const float c = ga * gi + gf * cp;
  const float hh = go * tanhf(c);
  *reinterpret_cast<float4*>(a.gates + (long)b * a.ld_g + 4 * u) = make_float4(ga, gi, gf, go);
  a.c_out[(long)b * a.h + u] = c;
  a.h_out[(long)b * a.h + u] = hh;
  const float hd = a.mask ? hh * a.mask[(long)b * a.h + u] : hh;
  if (a.hd_out) a.hd_out[(long)b * a.ld_hd + u] = hd;
  if (a.hd_out2) a.hd_out2[(long)b * a.ld_hd2 + u] = hd;
}

// ------------------------------------------------------------------ LSTM cell backward
struct CellBwdBatch {
  LstmCellBwdArgs c[8];
};

template <int MT, int NW>
__global__ __launch_bounds__(64 * NW) void lstm_cell_bwd_kernel(CellBwdBatch batch) {
  constexpr int NT = 2;   // product 0: dh_rec = dz_next WlT ; product 1: dx = dz_above WuT_above
  __shared__ __attribute__((aligned(16))) float red[NW * NT * MT * 256];
  const LstmCellBwdArgs& a = batch.c[blockIdx.z];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int j0 = blockIdx.x * 16, m0 = blockIdx.y * 16 * MT;
  const int jv = min(16, a.h - j0);
  f32x4 accp[NT][MT][1];
#pragma unroll
  for (int p = 0; p < NT; ++p)
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) accp[p][mt][0] = f32x4{0.f, 0.f, 0.f, 0.f};
  tile_dot<MT, 1, 16, NW>(a.p[0], m0, a.B, WRows{j0, 1, 0}, jv, lane, wave, accp[0]);
  if (a.npairs > 1) tile_dot<MT, 1, 16, NW>(a.p[1], m0, a.B, WRows{j0, 1, 0}, jv, lane, wave, accp[1]);
  f32x4 flat[NT * MT];
#pragma unroll
  for (int p = 0; p < NT; ++p)
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) flat[p * MT + mt] = accp[p][mt][0];
  float v[NT * MT];
  reduce_waves<NT * MT, NW>(flat, v, red);
  if (NW > 4 && threadIdx.x >= 256) return;
  const int u = j0 + (threadIdx.x & 15);
#pragma unroll
  for (int mt = 0; mt < MT; ++mt) {
    const int b = m0 + 16 * mt + (threadIdx.x >> 4);
    if (b >= a.B || u >= a.h) continue;
    const long bu = (long)b * a.h + u;
    float dy = v[MT + mt];
    if (a.dy) dy += a.dy[(long)b * a.ld_dy + u];
    if (a.dy2) dy += a.dy2[(long)b * a.ld_dy2 + u];
    if (a.mask) dy *= a.mask[bu];
    float dh = v[mt] + dy;
    if (a.dh_add) dh += a.dh_add[bu];
    float* gp = a.gates_dz + (long)b * a.ld_g + 4 * u;
    const float4 g = *reinterpret_cast<const float4*>(gp);
    const float ga = g.x, gi = g.y, gf = g.z, go = g.w;
    const float tc = tanhf(a.c_cur[bu]);
    const float cp = a.c_prev ? a.c_prev[bu] : 0.f;
    float dc = dh * go * (1.f - tc * tc);
    if (a.dc_next) dc += a.dc_next[bu];
    const float4 dz = make_float4(dc * gi * (1.f - ga * ga), dc * ga * gi * (1.f - gi), dc * cp * gf * (1.f - gf),
                                  dh * tc * go * (1.f - go));
    *reinterpret_cast<float4*>(gp) = dz;
    a.dc_prev[bu] = dc * gf;
  }
}

// eight waves share K once a four-wave workgroup would need more than two exposed round trips per wave (astk_set_tuning("row.longk") overrides the
// threshold, 0 = never: for A/B runs)
bool long_k(int k) {
  const int thr = (int)tune(TUNE_ROW_LONGK);
  return thr > 0 && k >= thr;
}

int check_pair(const RowPair& p, const char* who) {
  if (p.K <= 0) return 0;
  ASTK_CHECK(p.A && p.W, "%s: null operand", who);
  ASTK_CHECK((p.K % 4) == 0 && (p.lda % 4) == 0 && (p.ldw % 4) == 0 && aligned16(p.A) && aligned16(p.W),
             "%s: K, lda, ldw must be multiples of 4 and pointers 16-byte aligned (K=%d lda=%ld ldw=%ld)", who, p.K, p.lda, p.ldw);
  return 0;
}

}  // namespace

int rowgemm_launch(const RowGemmArgs& a, hipStream_t s) {
  ASTK_CHECK(a.M > 0 && a.N > 0 && a.out && a.npairs >= 1 && a.npairs <= 2, "rowgemm: bad arguments");
  for (int p = 0; p < a.npairs; ++p) ASTK_TRY(check_pair(a.p[p], "rowgemm"));
  // two row tiles per workgroup halve the weight traffic but also the number of workgroups: only when the chip stays full
  const bool two = a.M > 16 && (long)cdiv(a.N, 16) * cdiv(a.M, 32) >= device_cu_count();
  int ktot = 0;
  for (int p = 0; p < a.npairs; ++p) ktot += a.p[p].K;
  const dim3 grid(cdiv(a.N, 16), cdiv(a.M, two ? 32 : 16));
  if (long_k(ktot)) {
    if (two) hipLaunchKernelGGL((rowgemm_kernel<2, 8>), grid, dim3(512), 0, s, a);
    else hipLaunchKernelGGL((rowgemm_kernel<1, 8>), grid, dim3(512), 0, s, a);
  } else {
    if (two) hipLaunchKernelGGL((rowgemm_kernel<2, 4>), grid, dim3(256), 0, s, a);
    else hipLaunchKernelGGL((rowgemm_kernel<1, 4>), grid, dim3(256), 0, s, a);
  }
  ASTK_LAUNCH_CHECK();
  return 0;
}

int lstm_cell_fwd_launch(const LstmCellFwdArgs* cells, int ncells, hipStream_t s) {
  ASTK_CHECK(ncells >= 1 && ncells <= 8, "lstm_cell_fwd: 1..8 cells per launch");
  CellFwdBatch batch;
  for (int i = 0; i < ncells; ++i) {
    const LstmCellFwdArgs& c = cells[i];
    ASTK_CHECK(c.B == cells[0].B && c.h == cells[0].h, "lstm_cell_fwd: cells of one launch must share B and h");
    ASTK_CHECK(c.gates && c.c_out && c.h_out && (c.ld_g % 4) == 0 && aligned16(c.gates), "lstm_cell_fwd: bad outputs");
    ASTK_CHECK(!c.zx || ((c.ld_zx % 4) == 0 && aligned16(c.zx)), "lstm_cell_fwd: zx alignment");
    ASTK_CHECK(!c.bias || aligned16(c.bias), "lstm_cell_fwd: bias alignment");
    for (int p = 0; p < c.npairs; ++p) ASTK_TRY(check_pair(c.p[p], "lstm_cell_fwd"));
    batch.c[i] = c;
  }
  ProfScope prof(PROF_CELL, s);
  const bool two = cells[0].B > 16 && (long)cdiv(cells[0].h, 4) * cdiv(cells[0].B, 32) * ncells >= device_cu_count();
  const dim3 grid(cdiv(cells[0].h, 4), cdiv(cells[0].B, two ? 32 : 16), ncells);
  int ktot = 0;
  for (int p = 0; p < cells[0].npairs; ++p) ktot += cells[0].p[p].K;
  if (long_k(ktot)) {
    if (two) hipLaunchKernelGGL((lstm_cell_fwd_kernel<2, 8>), grid, dim3(512), 0, s, batch);
    else hipLaunchKernelGGL((lstm_cell_fwd_kernel<1, 8>), grid, dim3(512), 0, s, batch);
  } else {
    if (two) hipLaunchKernelGGL((lstm_cell_fwd_kernel<2, 4>), grid, dim3(256), 0, s, batch);
    else hipLaunchKernelGGL((lstm_cell_fwd_kernel<1, 4>), grid, dim3(256), 0, s, batch);
  }
  ASTK_LAUNCH_CHECK();
  return 0;
}

int lstm_cell_bwd_launch(const LstmCellBwdArgs* cells, int ncells, hipStream_t s) {
  ASTK_CHECK(ncells >= 1 && ncells <= 8, "lstm_cell_bwd: 1..8 cells per launch");
  CellBwdBatch batch;
  for (int i = 0; i < ncells; ++i) {
    const LstmCellBwdArgs& c = cells[i];
    ASTK_CHECK(c.B == cells[0].B && c.h == cells[0].h, "lstm_cell_bwd: cells of one launch must share B and h");
    ASTK_CHECK(c.gates_dz && c.c_cur && c.dc_prev && (c.ld_g % 4) == 0 && aligned16(c.gates_dz), "lstm_cell_bwd: bad buffers");
    for (int p = 0; p < c.npairs; ++p) ASTK_TRY(check_pair(c.p[p], "lstm_cell_bwd"));
    batch.c[i] = c;
  }
  ProfScope prof(PROF_CELL, s);
  const bool two = cells[0].B > 16 && (long)cdiv(cells[0].h, 16) * cdiv(cells[0].B, 32) * ncells >= device_cu_count();
  const dim3 grid(cdiv(cells[0].h, 16), cdiv(cells[0].B, two ? 32 : 16), ncells);
  int kmax = 0;
  for (int p = 0; p < cells[0].npairs; ++p) kmax = cells[0].p[p].K > kmax ? cells[0].p[p].K : kmax;
  if (long_k(kmax)) {
    if (two) hipLaunchKernelGGL((lstm_cell_bwd_kernel<2, 8>), grid, dim3(512), 0, s, batch);
    else hipLaunchKernelGGL((lstm_cell_bwd_kernel<1, 8>), grid, dim3(512), 0, s, batch);
  } else {
    if (two) hipLaunchKernelGGL((lstm_cell_bwd_kernel<2, 4>), grid, dim3(256), 0, s, batch);
    else hipLaunchKernelGGL((lstm_cell_bwd_kernel<1, 4>), grid, dim3(256), 0, s, batch);
  }
  ASTK_LAUNCH_CHECK();
  return 0;
}

}  // namespace astk
