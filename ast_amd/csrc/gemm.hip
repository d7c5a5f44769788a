// f32-input MFMA GEMM for gfx950 (v_mfma_f32_32x32x2_f32: exact f32, k-ordered fma chain).
//
// One kernel family covers every batched dense product of the train step (SURVEY.md K6, K9, K24 and
// all dgrad / wgrad products): C[M,N] (+)= op(A) op(B) (+bias).
//   128x128x32 block tile, 256 threads = 4 waves as 2x2, each wave 64x64 = 2x2 MFMA 32x32 tiles
//   (64 accumulator VGPRs).  Both operands are staged K-major in LDS (As[k][m], Bs[k][n]) so every MFMA
//   operand read is one conflict-free ds_read_b32 across 32 consecutive floats; operands whose global
//   rows are K-contiguous are transposed on the LDS write (row stride 129 -> conflict-free scalar
//   writes), operands whose rows are M/N-contiguous are written with ds_write_b128 (row stride 132).
//   Global loads are 16 B per lane, register-staged one k-tile ahead of the MFMAs.
// Rows of A, B and C can use two-level or indexed addressing (MatView) so the conv layers run as
// zero-copy "window" GEMMs over padded channels-last activations and the reverse LSTM direction reads
// frames through its permutation table instead of a permuted copy.
#include "common.h"

namespace astk {

namespace {

constexpr int BM = 128, BN = 128, BK = 32;
constexpr int LD_RK = 129;  // LDS row stride (floats): operand staged from K-contiguous global rows
constexpr int LD_KR = 132;  // LDS row stride: operand staged from M/N-contiguous global rows (16 B aligned)

__device__ __forceinline__ long rowoff(const MatView& v, int r) {
  if (v.rowidx) return (long)v.rowidx[r] * v.ld;
  if (v.tn > 0) return (long)(r / v.tn) * v.sg + (long)(r % v.tn) * v.st;
  return (long)r * v.ld;
}

// Branch-free guarded load: the caller passes an address that is ALWAYS readable (clamped inside the matrix); elements
// at index >= nvalid are zeroed with selects.  (A `cond ? load : 0` makes hipcc branch around every load and wait
// vmcnt(0) per element -- the loads of a tile would serialise.)
__device__ __forceinline__ float4 mask4(float4 v, int nvalid) {
  v.x = nvalid > 0 ? v.x : 0.f;
  v.y = nvalid > 1 ? v.y : 0.f;
  v.z = nvalid > 2 ? v.z : 0.f;
  v.w = nvalid > 3 ? v.w : 0.f;
  return v;
}

// Stages one operand tile.  RK: global rows are the tile's M (or N) index, K contiguous (row offsets are computed once,
// any MatView addressing).  !RK: global rows are the k index; TWOLVL selects plain (k*ld) or two-level row offsets --
// both branch-free, so the tile's loads are issued back to back (indexed rows are not supported on this path).
template <bool RK, bool TWOLVL>
struct Stager {
  float4 reg[4];  // raw staged data: masked only when it is written to LDS, so the global loads stay in flight over the MFMAs
  int nv[4];      // valid elements of reg[p]
  long off[4];   // RK only: row offsets (fixed for the whole k loop)
  bool ok[4];    // RK only
  int a, b;      // RK: a = k-quad (0..7), b = row0 (0..31).  KR: a = col-quad (0..31), b = krow0 (0..7)

  __device__ __forceinline__ void init(const MatView& v, int row0, int nrows, int tid) {
    if (RK) {
      a = tid & 7;
      b = tid >> 3;
#pragma unroll
      for (int p = 0; p < 4; ++p) {
        int r = row0 + b + 32 * p;
        ok[p] = r < nrows;
        off[p] = rowoff(v, ok[p] ? r : (nrows - 1));
      }
    } else {
      a = tid & 31;
      b = tid >> 5;
    }
  }
  // kcur: first k of the tile, kend: exclusive end of this block's K range, K: full K extent; col0/ncols: KR only.
  // Leading dimensions are multiples of 4 and >= the extent, so a float4 at any multiple of 4 below round_up(extent,4)
  // stays inside its row: addresses are clamped to that range and the surplus is masked.
  __device__ __forceinline__ void load(const MatView& v, int kcur, int kend, int K, int col0, int ncols) {
    if (RK) {
      const int k = kcur + a * 4;
      const int kc = min(k, ((K + 3) & ~3) - 4);
#pragma unroll
      for (int p = 0; p < 4; ++p) {
        reg[p] = *reinterpret_cast<const float4*>(v.p + off[p] + kc);
        nv[p] = ok[p] ? (kend - k) : 0;
      }
    } else {
      const int col = col0 + a * 4;
      const int colc = min(col, ((ncols + 3) & ~3) - 4);
#pragma unroll
      for (int p = 0; p < 4; ++p) {
        const int kr = kcur + b + 8 * p;
        const int krc = min(kr, kend - 1);
        const long ro = TWOLVL ? (long)(krc / v.tn) * v.sg + (long)(krc % v.tn) * v.st : (long)krc * v.ld;
        reg[p] = *reinterpret_cast<const float4*>(v.p + ro + colc);
        nv[p] = kr < kend ? (ncols - col) : 0;
      }
    }
  }
  __device__ __forceinline__ void store(float* S) const {
    if (RK) {
#pragma unroll
      for (int p = 0; p < 4; ++p) {
        const int r = b + 32 * p;
        const float4 m = mask4(reg[p], nv[p]);
        S[(a * 4 + 0) * LD_RK + r] = m.x;
        S[(a * 4 + 1) * LD_RK + r] = m.y;
        S[(a * 4 + 2) * LD_RK + r] = m.z;
        S[(a * 4 + 3) * LD_RK + r] = m.w;
      }
    } else {
#pragma unroll
      for (int p = 0; p < 4; ++p) *reinterpret_cast<float4*>(&S[(b + 8 * p) * LD_KR + a * 4]) = mask4(reg[p], nv[p]);
    }
  }
};

template <bool A_RK, bool B_RK, bool TWOLVL>
__global__ __launch_bounds__(256) void gemm_f32_kernel(GemmArgs g) {
  constexpr int LDA = A_RK ? LD_RK : LD_KR;
  constexpr int LDB = B_RK ? LD_RK : LD_KR;
  __shared__ __attribute__((aligned(16))) float As[BK * LDA];
  __shared__ __attribute__((aligned(16))) float Bs[BK * LDB];

  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int tiles_n = (g.N + BN - 1) / BN;
  const int tile_m = blockIdx.x / tiles_n, tile_n = blockIdx.x % tiles_n;
  const int m0 = tile_m * BM, n0 = tile_n * BN;
  const int zb = blockIdx.z / g.ksplit, split = blockIdx.z % g.ksplit;

  int kper = (g.K + g.ksplit - 1) / g.ksplit;
  kper = (kper + BK - 1) / BK * BK;
  const int kbeg = split * kper;
  const int kend = min(g.K, kbeg + kper);
  if (kbeg >= kend) return;

  MatView A = g.A, B = g.B;
  A.p += (long)zb * g.sA;
  B.p += (long)zb * g.sB;
  float* C = g.C + (long)zb * g.sC;

  Stager<A_RK, TWOLVL> sa;
  Stager<B_RK, TWOLVL> sb;
  sa.init(A, m0, g.M, tid);
  sb.init(B, n0, g.N, tid);

  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  const int wm = wave >> 1, wn = wave & 1;
  const int li = lane & 31, lk = lane >> 5;
  const int nk = (kend - kbeg + BK - 1) / BK;

  sa.load(A, kbeg, kend, g.K, m0, g.M);
  sb.load(B, kbeg, kend, g.K, n0, g.N);
  sa.store(As);
  sb.store(Bs);
  __syncthreads();

  for (int kt = 0; kt < nk; ++kt) {
    const bool more = kt + 1 < nk;
    if (more) {
      sa.load(A, kbeg + (kt + 1) * BK, kend, g.K, m0, g.M);
      sb.load(B, kbeg + (kt + 1) * BK, kend, g.K, n0, g.N);
    }
    const float* ap = As + lk * LDA + wm * 64 + li;
    const float* bp = Bs + lk * LDB + wn * 64 + li;
    // software-pipelined operand fetch: the LDS reads of k-pair kk+2 are issued before the MFMAs of k-pair kk
    float a0 = ap[0], a1 = ap[32], b0 = bp[0], b1 = bp[32];
#pragma unroll
    for (int kk = 0; kk < BK; kk += 2) {
      float na0 = 0.f, na1 = 0.f, nb0 = 0.f, nb1 = 0.f;
      if (kk + 2 < BK) {
        na0 = ap[(kk + 2) * LDA]; na1 = ap[(kk + 2) * LDA + 32];
        nb0 = bp[(kk + 2) * LDB]; nb1 = bp[(kk + 2) * LDB + 32];
      }
      acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc[0][0], 0, 0, 0);
      acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b1, acc[0][1], 0, 0, 0);
      acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b0, acc[1][0], 0, 0, 0);
      acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1, acc[1][1], 0, 0, 0);
      a0 = na0; a1 = na1; b0 = nb0; b1 = nb1;
    }
    // pin the interleave (hipcc otherwise sinks every LDS read directly in front of its MFMAs with lgkmcnt(0)):
    // reads of k-pair i+1 (2 x ds_read2_b32) go in front of the 4 MFMAs of k-pair i
    __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
#pragma unroll
    for (int i = 0; i < BK / 2 - 1; ++i) {
      __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
      __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);
    }
    __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);
    __syncthreads();
    if (more) {
      sa.store(As);
      sb.store(Bs);
      __syncthreads();
    }
  }

  // epilogue: C/D layout of the 32x32 MFMA: col = lane&31, row = (reg&3) + 8*(reg>>2) + 4*(lane>>5)
  const bool add_bias = g.bias != nullptr && split == 0;
#pragma unroll
  for (int i = 0; i < 2; ++i) {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int row = m0 + wm * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lk;
      if (row >= g.M) continue;
      const long coff = g.c_tn > 0 ? (long)(row / g.c_tn) * g.c_sg + (long)(row % g.c_tn) * g.c_st : (long)row * g.ldc;
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const int col = n0 + wn * 64 + j * 32 + li;
        if (col >= g.N) continue;
        float v = acc[i][j][r];
        if (add_bias) v += g.bias[col];
        float* dst = C + coff + col;
        if (g.mode == GEMM_STORE) *dst = v;
        else if (g.mode == GEMM_ACCUM) *dst += v;
        else atomicAdd(dst, v);
      }
    }
  }
}

}  // namespace

int gemm_launch(int layout, const GemmArgs& g, hipStream_t s) {
  if (g.M <= 0 || g.N <= 0 || g.K <= 0 || g.batch <= 0) return 0;
  ASTK_CHECK(g.A.p && g.B.p && g.C, "gemm: null operand");
  ASTK_CHECK(aligned16(g.A.p) && aligned16(g.B.p), "gemm: A/B must be 16-byte aligned");
  ASTK_CHECK((g.A.ld % 4) == 0 && (g.B.ld % 4) == 0 && (g.A.sg % 4) == 0 && (g.A.st % 4) == 0 &&
                 (g.B.sg % 4) == 0 && (g.B.st % 4) == 0 && (g.sA % 4) == 0 && (g.sB % 4) == 0,
             "gemm: leading dimensions / strides must be multiples of 4 floats (lda=%ld ldb=%ld)", g.A.ld, g.B.ld);
  ASTK_CHECK(g.ksplit >= 1 && (g.ksplit == 1 || g.mode == GEMM_ATOMIC), "gemm: split-K needs atomic mode");
  GemmArgs a = g;
  const bool a_kr = layout == GEMM_TN, b_kr = layout != GEMM_NT;
  ASTK_CHECK(!(a_kr && g.A.rowidx) && !(b_kr && g.B.rowidx), "gemm: indexed rows are only supported on K-contiguous operands");
  const bool twolvl = (a_kr && g.A.tn > 0) || (b_kr && g.B.tn > 0);
  if (twolvl) {   // a plain operand next to a two-level one: express it as one group of INT_MAX rows
    if (a_kr && a.A.tn <= 0) { a.A.tn = 0x7fffffff; a.A.sg = 0; a.A.st = a.A.ld; }
    if (b_kr && a.B.tn <= 0) { a.B.tn = 0x7fffffff; a.B.sg = 0; a.B.st = a.B.ld; }
  }
  const long tiles = (long)cdiv(g.M, BM) * cdiv(g.N, BN);
  dim3 grid((unsigned)tiles, 1, (unsigned)(g.batch * g.ksplit));
  ProfScope prof(PROF_GEMM, s, 2.0 * g.M * g.N * (double)g.K * g.batch);
  switch (layout) {
    case GEMM_NT: hipLaunchKernelGGL((gemm_f32_kernel<true, true, false>), grid, dim3(256), 0, s, a); break;
    case GEMM_NN:
      if (twolvl) hipLaunchKernelGGL((gemm_f32_kernel<true, false, true>), grid, dim3(256), 0, s, a);
      else hipLaunchKernelGGL((gemm_f32_kernel<true, false, false>), grid, dim3(256), 0, s, a);
      break;
    case GEMM_TN:
      if (twolvl) hipLaunchKernelGGL((gemm_f32_kernel<false, false, true>), grid, dim3(256), 0, s, a);
      else hipLaunchKernelGGL((gemm_f32_kernel<false, false, false>), grid, dim3(256), 0, s, a);
      break;
    default: ASTK_CHECK(false, "gemm: bad layout %d", layout);
  }
  ASTK_LAUNCH_CHECK();
  return 0;
}

}  // namespace astk
