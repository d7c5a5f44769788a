// Batched dense products of the train step on the gfx950 matrix pipe (SURVEY.md K6, K9, K24 and all dgrad / wgrad products):
// C[M,N] (+)= op(A) op(B) (+bias), f32 in, f32 out, f32-level accuracy.
//
// One kernel template, three operand schemes (GemmPrec below; chosen per call by the descriptors' `precision`, astk.h):
//   * bf16x3 (DEFAULT): every f32 operand value is split INSIDE the kernel into three bf16 terms that represent it exactly (f32 exponent range, no
//     scales), six v_mfma_f32_32x32x16_bf16 per 16 k -- at least the accuracy of an f32 fma chain on any data; launches of >= 20 GFLOP take the
//     12-wave 256 x 128 kernel
//   * fp16x2 (opt-in, NARROWER than float32: ASTK_PREC_FP16X2 / astk_set_gemm_precision(0)): two fp16 terms behind a per-operand power-of-two scale,
//     three v_mfma_f32_32x32x16_f16 per 16 k; the scales come from an absolute-maximum pass in front of the launch (k_absmax), from the kernel that
//     wrote the operand, or from the caller; launches below 3 GFLOP do not repay that pass and run as bf16x3
//   * f32 (ASTK_PREC_F32 / astk_set_gemm_precision(2)): v_mfma_f32_32x32x2_f32, the exact k-ordered f32 fma chain (round 1's kernel)
// (+ single-term fp16 operands for the launches their callers mark `lowp` when astk_set_low_precision_gemms(1) is in force).
//   128x128x16 block tile (BK = 16), 64x64x16 tiles for launches too small to occupy the chip with 128-tiles; k-iterations split evenly
//   over the grid (stream-K).  Split schemes: 512 threads -- waves 0-3 multiply out of LDS (fragments of tile kt+1 fetched behind tile
//   kt's MFMAs), waves 4-7 stage (4-deep register ring of raw f32 tiles, the split, LDS writes) -- three LDS stages of 16-bit planes,
//   one workgroup per CU.  f32 scheme: 256 threads = 4 waves as 2x2, every wave stages and multiplies, double-buffered f32 tiles in
//   LDS, three workgroups per CU.  An operand whose global rows are K-contiguous ("RK") is kept row-major in LDS, one whose rows are
//   M/N-contiguous ("KR") K-major (split schemes: read with the transposing ds_read_b64_tr_b16).  Global loads are 16 B per lane raw
//   buffer loads off a uniform base (see Stager); K-contiguous operands of the split schemes fetch two k-tiles (one 128-byte line per
//   row) at a time.
// Rows of A, B and C can use two-level or indexed addressing (MatView) so the conv layers run as
// zero-copy "window" GEMMs over padded channels-last activations and the reverse LSTM direction reads
// frames through its permutation table instead of a permuted copy.
#include "common.h"
#include <cstdio>
#include <cstdlib>
#include <algorithm>
#include <atomic>
#include <mutex>
#include <vector>
#include <type_traits>

namespace astk {

namespace {

#ifndef ASTK_GEMM_BK
#define ASTK_GEMM_BK 16
#endif
constexpr int BK = ASTK_GEMM_BK;
// Block tile edge TL: 128 (4 waves x 64x64, the throughput configuration) or 64 (4 waves x 32x32) for products too small to
// give every CU a 128-tile's worth of k-iterations -- a k-iteration of a 128-tile is 32 MFMAs = 0.87 us per wave whatever the
// problem size, so 40 such tiles with K = 512 cannot finish in less than 28 us; with 64-tiles the same product is 160 tiles of 7 us.
#ifndef ASTK_GEMM_RING
#define ASTK_GEMM_RING 4        // register slots of staged tiles on the split paths (must divide 12, the staging loop's trip)
#endif
#ifndef ASTK_GEMM_RING8
#define ASTK_GEMM_RING8 2
#endif
#ifndef ASTK_GEMM_PRIO8
#define ASTK_GEMM_PRIO8 3       // s_setprio of the multiplying waves of the 12-wave kernel
#endif
#ifndef ASTK_GEMM_PRIO8S
#define ASTK_GEMM_PRIO8S 0      // ... and of its staging waves
#endif
#ifndef ASTK_GEMM_PAIR
#define ASTK_GEMM_PAIR 1        // K-contiguous operands: fetch two k-tiles (one 128-byte line per row) at a time
#endif
#ifndef ASTK_GEMM_X3_WGS
#define ASTK_GEMM_X3_WGS 1
#endif
// co-resident workgroups the grid is sized for.  Workgroups of the split schemes have 512 threads (2 waves per SIMD each)
constexpr int wgs_per_cu(int TL, int prec = 0) { return prec != 0 ? (TL == 64 ? 2 : ASTK_GEMM_X3_WGS) : (TL == 64 ? 4 : (BK == 32 ? 2 : 3)); }
constexpr int waves_per_simd(int TL, int prec) { return prec != 0 ? 2 * wgs_per_cu(TL, prec) : wgs_per_cu(TL, prec); }   // co-resident workgroups the grid is sized for (128-tiles,
                                               // BK = 16: 40 KB of LDS and <= 168 registers per workgroup; 2, 3 and 4 per CU are within 2 %)
constexpr int LD_RK = BK + 4;  // LDS row stride (floats) of an operand staged from K-contiguous global rows: kept ROW-major [m][k'] with the
                               // tile's k order permuted to [even k | odd k], so that a thread's global float4 (4 consecutive k) is two
                               // 8-byte LDS writes and the 8 k values an MFMA lane consumes (k = lk, lk+2, ...) are two 16-byte LDS reads
                               // (stride 20 floats: conflict-free for both); was K-major with 4 scalar transposing writes and 8 scalar reads
constexpr int ld_kr(int TL) { return TL; }
                            // LDS row stride of an operand staged from M/N-contiguous global rows: K-major [k][m'], written with 16-byte stores;
                            // the 128 columns of a 128-tile are ordered [wave-row 0 tile 0 | wave-row 1 tile 0 | wave-row 0 tile 1 | wave-row 1 tile 1] so the
                            // two values a lane needs per k (its column of both 32-wide MFMA tiles) are 64 floats apart and consecutive
                            // k-pairs 256 floats: one ds_read2st64_b32 with immediate offsets per k-pair, no address arithmetic

__device__ __forceinline__ long rowoff(const MatView& v, int r) {
  if (v.rowidx) return (long)v.rowidx[r] * v.ld;
  if (v.tn > 0) return (long)(r / v.tn) * v.sg + (long)(r % v.tn) * v.st;
  return (long)r * v.ld;
}

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

// First k-iteration of workgroup w of G (stream-K split of the launch's iteration sequence).  grp.unit = 2 when every product of
// the launch has an even number of k-iterations per tile: ranges then begin on even k-iterations, i.e. on 128-byte lines of a
// K-contiguous f32 operand, which the paired tile loads rely on for whole-line fetches.
// (hybrid launches: the stream-K part covers only [rem_start, iters_total); rem_start is a multiple of unit)
__host__ __device__ __forceinline__ long wg_first_iter(const GemmGroup& grp, unsigned w, unsigned G) {
  return grp.rem_start + ((grp.iters_total - grp.rem_start) / grp.unit) * (long)w / (long)G * grp.unit;
}
// ---- Deterministic split tiles (round 6; template parameter FIX, launches of a `deterministic` call).  A tile whose k range is shared by
// several workgroups is normally zeroed in front of the launch and accumulated with float atomics by every contributor: the sum then depends
// on the order the atomics arrive in, which varies from run to run (2-6e-10 of the clip norm) -- harmless for training, but it blinds a
// soak that compares two passes bit for bit exactly where a hand-off race would show.  FIX kernels: every contributing WAVE stores its 64 x 64
// accumulator block to a slot of g_fix_part (16-byte write-through stores, lane-major), drains them and draws a ticket on the (tile, wave)
// counter; the wave that draws the LAST ticket reads all blocks of its sub-tile back in WORKGROUP ORDER (sc1 loads behind the returned
// ticket: the per-wave form of MI355X_MICROARCH.md's hand-off table), sums them in that fixed order and writes the tile the way `mode` says
// (plain stores for GEMM_STORE: no zeroing launch; ONE atomic add per element for GEMM_ATOMIC / GEMM_ACCUM).  Nobody waits for anybody (no
// spin, residency does not matter); the last arriver puts the counter back to zero.  Slot of (workgroup w, tile X) = 2 w + (X holds the START
// of w's range ? 0 : 1): a stream-K range has at most two split tiles, one at each end.  One launch at a time may use the slots (they are
// process-wide like the scale ring): deterministic calls run everything on ONE stream.  Measured (round 5, as an always-on experiment): every
// launch a few per cent slower -- a weight-gradient tile has up to 14 contributors whose blocks the last one reads one after the other --
// which is why it is a mode, not the default.  Instantiated for the 128 x 128 bf16x3 kernel only; a deterministic call runs all its split
// launches on it (launches of the f32 scheme fall back to one whole tile per workgroup).
constexpr int FIX_WGS = 256;                       // launches of up to this many workgroups
constexpr int FIX_WAVES = 4;
constexpr int FIX_BLOCK = 64 * 64;                 // floats of one wave's accumulator block
__device__ float g_fix_part[(size_t)2 * FIX_WGS * FIX_WAVES * FIX_BLOCK];      // 32 MB
__device__ unsigned g_fix_ctr[2 * FIX_WGS * FIX_WAVES];

// Tile `tile` of a problem (batch-major, then the block order of GemmArgs::bm) -> batch slice and tile coordinates.
__host__ __device__ __forceinline__ void decode_tile(const GemmArgs& g, long tile, int TLM, int TL, int& zb, int& m0, int& n0) {
  const int tiles_n = (g.N + TL - 1) / TL;
  zb = (int)(tile / g.tiles_mn);
  const int tmn = (int)(tile - (long)zb * g.tiles_mn);
  if (g.bm <= 1) { m0 = (tmn / tiles_n) * TLM; n0 = (tmn % tiles_n) * TL; return; }
  const int tiles_m = g.tiles_mn / tiles_n;
  const int per_sr = g.bm * tiles_n;
  const int sr = tmn / per_sr, r = tmn - sr * per_sr;
  const int rows = min(g.bm, tiles_m - sr * g.bm);
  const int n = r / rows;
  m0 = (sr * g.bm + (r - n * rows)) * TLM;
  n0 = n * TL;
}

// Position j (0..BK/2-1) of the k values a lane of k-group lk (0/1) feeds to the MFMAs of one k-iteration is
// k = KROW(j) + 2*lk: a staged float4 (4 consecutive k) then splits into the register pairs (x,y) -> lk 0 and (z,w) -> lk 1.
#define KROW(j) (4 * ((j) >> 1) + ((j) & 1))

// Stages one operand tile: global -> registers (load) -> LDS (store), one tile per call, tiles in k order.
//   RK: global rows are the tile's M (or N) index, K contiguous (any MatView addressing; row offsets are computed once per tile).
//   !RK: global rows are the k index; TWOLVL selects plain (k * ld) or two-level row offsets (indexed rows are not supported here).
// The k loop must not spend vector-ALU instructions: on gfx950 a VALU instruction takes the issue slot of an MFMA pass
// (scratch/mfma_mix_bench.hip: 3 VALU per MFMA cost 18 % of the MFMA rate), and the first version of this stager spent ~56 of
// them per 32 MFMAs on 64-bit addresses, tail selects and register shuffles.  So:
//   * loads are raw buffer loads: voffset = a 32-bit byte offset fixed for the tile, base = a uniform pointer that advances with
//     k in SGPRs (operand slices are limited to 4 GiB, checked on the host);
//   * rows / columns outside the matrix are CLAMPED, never masked: they only feed C elements the epilogue does not write;
//   * the buffer's num_records is the exact extent of the operand slice, so reads past it return zeros: that IS the K tail of a
//     plain K-major operand; a K-contiguous operand may pick up the neighbouring row's data in its last tile, which the
//     owning thread then zeroes in LDS (zero_tail, taken once per tile at most and only when K is not a multiple of BK);
//   * two-level K-major rows (the conv wgrad) are addressed like plain ones while a tile's BK rows lie inside one group (uniform
//     test); only a tile that straddles groups computes per-row offsets.
// ---- bf16x3 operands (PREC_BF16X3): every f32 operand value is split into three bf16 terms, x = hi + mid + lo with
//   hi = bf16(x), mid = bf16(x - hi), lo = bf16(x - hi - mid)   (round-to-nearest-even each; bf16 has f32's exponent range, so no scaling)
// which carry 24 significant bits between them, and a product is summed from the six largest of the nine term products
//   a b ~ lo.hi + hi.lo + mid.mid + mid.hi + hi.mid + hi.hi        (dropped: mid.lo, lo.mid, lo.lo <= 2^-23 |a b|)
// on v_mfma_f32_32x32x16_bf16 with f32 accumulation: 6 MFMAs of 32 cycles for 16 k against 8 f32 MFMAs of 64 cycles, i.e. 2.7x the
// f32-input MFMA rate at an error of the order of f32 rounding itself (a f32 fma chain rounds every product to 2^-24; here a product
// is exact to 2^-23 and a K = 16 block is summed inside the MFMA).  The split runs in the stager (global f32 -> registers -> three
// bf16 planes in LDS), so callers, layouts, addressing, stream-K and the epilogue are those of the f32 kernel.
// PREC_F16 (BASELINE configs[4], "fp16 MFMA GEMMs"): operands rounded to ONE fp16 term (round-to-nearest-even), one
// v_mfma_f32_32x32x16_f16 per tile and 16 k, f32 accumulation.  Reduced precision (11 significant bits): only launches their caller marks
// low-precision-eligible take it, and only when astk_set_low_precision_gemms(1) is in force.
// PREC_F16X2: every operand is scaled by a power of two s (its absolute maximum lands in [2^14, 2^15): fp16 has only 5 exponent bits)
// and split into TWO fp16 terms, x s = hi + lo with hi = fp16(x s), lo = fp16(x s - hi), both round-to-nearest-even (common.h:
// split2h): 22 significant bits for every value down to 2^-17 of the operand's maximum, an absolute floor of 2^-39 of the maximum below
// that.  A product is summed from three of the four term products on v_mfma_f32_32x32x16_f16 (lo.hi + hi.lo + hi.hi; lo.lo <= 2^-22 |a b|
// is dropped) and unscaled by 1 / (s_A s_B) in the epilogue: HALF the matrix-pipe work of bf16x3 at an error of 2^-22 per product.  The
// scales come from an absolute-maximum pass over both operands in front of the launch (k_absmax) or from the caller (gemm_amax).
enum GemmPrec { PREC_F32 = 0, PREC_BF16X3 = 1, PREC_F16 = 2, PREC_F16X2 = 3 };
constexpr int prec_planes(int prec) { return prec == PREC_BF16X3 ? 3 : (prec == PREC_F16X2 ? 2 : 1); }
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
// two values -> one dword of each plane (element 0 in the low half).  (Inline asm is safe HERE: the planes go to LDS, no MFMA reads a
// converted register directly -- see common.h cvt_pk_bf16_f32 for where it is not.)
__device__ __forceinline__ unsigned cvt_pk_bf16(float lo, float hi) {
  unsigned r;
  asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(r) : "v"(lo), "v"(hi));
  return r;
}
// (the residuals through v_dot2c_f32_bf16 -- 7 instead of 11 vector-ALU instructions per pair -- were built, verified bit for bit and measured SLOWER:
//  HISTORY.md round 4, scratch/dot2_split_probe.hip)
__device__ __forceinline__ void split2(float x0, float x1, unsigned& hi, unsigned& mid, unsigned& lo) {
  hi = cvt_pk_bf16(x0, x1);
  const float r0 = x0 - __uint_as_float(hi << 16), r1 = x1 - __uint_as_float(hi & 0xffff0000u);
  mid = cvt_pk_bf16(r0, r1);
  const float s0 = r0 - __uint_as_float(mid << 16), s1 = r1 - __uint_as_float(mid & 0xffff0000u);
  lo = cvt_pk_bf16(s0, s1);
}
// The power of two that brings an operand's absolute maximum into [2^14, 2^15), as the biased exponent of a float (amax: the word
// gemm_absmax left: launch generation in the high half, the maximum's float bits in the low half; null or zero maximum: 1.0).
__device__ __forceinline__ int scale_exp(const unsigned long long* amax) {
  if (amax == nullptr) return 127;
  // (bit 0 of the handle: a producer slot, its 16 shards one 128-byte line apart -- common.h AMAX_PSLOT_STRIDE)
  const int st = ((uintptr_t)amax & 1u) ? AMAX_PSLOT_STRIDE : 1;
  amax = (const unsigned long long*)((uintptr_t)amax & ~(uintptr_t)1);
  unsigned long long w = amax[0];       // AMAX_SHARDS words; one left by an earlier launch (older generation) loses to any of this one's
#pragma unroll
  for (int i = 1; i < 16; ++i) w = max(w, amax[i * st]);
  const int e = (int)(((unsigned)w) >> 23) & 0xff;
  return e == 0 ? 127 : min(max(268 - e, 1), 253);
}
// LDS images of one operand stage on the split paths (bytes).  RK operand (K-contiguous global rows): per plane two k-halves
// [h = k / 8][row][8 k] of TL x 16 B, 16 B apart from a multiple of 128 B so that a stager's 8-byte writes and the MFMA lanes' 16-byte
// row reads are both conflict-free.  KR operand (M/N-contiguous global rows): per plane K-major [k][m] rows of 2 TL + 64 B (the pad makes
// the four k rows of a transposing read land 16 banks apart), read with ds_read_b64_tr_b16.
constexpr int sp_hs(int TL) { return TL * 16 + 16; }          // RK: k-half stride
constexpr int sp_rs(int TL) { return TL * 2 + 64; }           // KR: k-row stride
constexpr int sp_plane(int TL, bool RK) { return RK ? (BK / 8) * sp_hs(TL) : BK * sp_rs(TL); }
constexpr int sp_stage(int TL, bool RK, int prec = PREC_BF16X3) { return prec_planes(prec) * sp_plane(TL, RK); }

// RING: register slots of staged tiles.  The f32 kernel keeps one tile in flight (an iteration is 2600 cycles and three workgroups per
// CU cover each other's waits); an iteration of the split schemes is 600-800 cycles on one workgroup per CU, so their staging waves keep RING = 4 tiles
// (16 loads per thread) in flight to cover the memory latency.
template <int TL, bool RK, bool TWOLVL, int PREC = PREC_F32, int RING = 1>
struct Stager {
  static constexpr int LD_KR = ld_kr(TL);
  static constexpr int KQ = BK / 4;                  // RK: k-quads per row
  static constexpr int CQ = TL / 4;                  // KR: column quads per k row
  static constexpr int RP = 256 / CQ;                // KR: k rows per pass (256 threads x 16 B)
  static constexpr int NP = RK ? TL * KQ / 256 : BK / RP;   // passes over the TL x BK tile
  static_assert(NP >= 1 && (RK || RP * NP == BK), "tile / thread-count mismatch");
  float4 reg[RING][NP];     // staged data
  unsigned voff[NP];  // RK and plain KR: byte offset of this thread's float4 from the tile's uniform base
  int tgrp, trem;     // two-level KR: group and position inside the group of row kcur (uniform)
  int tcol;           // two-level KR: this thread's (clamped) column
  int a, b;           // RK: a = k-quad (0..KQ-1), b = row group (rows NP*b .. NP*b+NP-1).  KR: a = column quad, b = krow0 (0..RP-1)
  int kcur;           // first k of the tile the next load() fetches
  int lds;            // float offset of this thread's first LDS write
  float scl;          // fp16x2: the operand's power-of-two scale

  // row0 / nrows: the tile's first M (N) index and the operand's M (N) extent; [kbeg, kend): this workgroup's K range.
  // Leading dimensions are multiples of 4 and >= the extent, so a float4 at any multiple of 4 below round_up(extent, 4)
  // stays inside its row.
  __device__ __forceinline__ void init(const MatView& v, int row0, int nrows, int kbeg, int kend, int tid) {
    kcur = kbeg;
    if (RK) {
      a = tid % KQ;
      b = tid / KQ;
#pragma unroll
      for (int p = 0; p < NP; ++p) voff[p] = (unsigned)((rowoff(v, min(row0 + NP * b + p, nrows - 1)) + a * 4) * 4);
      lds = NP * b * LD_RK + 2 * a;   // a thread's NP rows are neighbours: one LDS address register serves all its writes
    } else {
      a = tid % CQ;
      b = tid / CQ;
      // LDS column order of a 128-wide KR tile: [wm0 i0 | wm1 i0 | wm0 i1 | wm1 i1] x 32 (see ld_kr); lane a writes LDS columns 4a..4a+3,
      // which hold the tile's columns wm*64 + i*32 + 4*(a&7) with i = a>>4, wm = (a>>3)&1.  A 64-wide tile is stored in order.
      const int m = (TL == 128 && PREC == PREC_F32) ? ((a >> 3) & 1) * 64 + (a >> 4) * 32 + (a & 7) * 4 : 4 * a;   // images of the split schemes: natural order
      const int col = min(row0 + m, ((nrows + 3) & ~3) - 4);
#pragma unroll
      for (int p = 0; p < NP; ++p)
        voff[p] = (unsigned)(((long)(b + RP * p) * (TWOLVL ? v.st : v.ld) + col) * 4);
      if (TWOLVL) { tgrp = kbeg / v.tn; trem = kbeg % v.tn; tcol = col; }
      lds = b * LD_KR + 4 * a;
    }
  }
  // span: bytes from v.p to the end of the operand slice (gemm_prepare)
  __device__ __forceinline__ void load(const MatView& v, unsigned span, int kend, const int slot = 0) {
    if (RK || !TWOLVL) {
      const long adv = RK ? (long)kcur : (long)kcur * v.ld;
      const __amdgpu_buffer_rsrc_t r =
          __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(v.p + adv), 0, (int)(span - (unsigned)(adv * 4)), 0x00020000);
#pragma unroll
      for (int p = 0; p < NP; ++p) {
        const u32x4 u = __builtin_amdgcn_raw_buffer_load_b128(r, (int)voff[p], 0, 0);
        reg[slot][p] = make_float4(__uint_as_float(u.x), __uint_as_float(u.y), __uint_as_float(u.z), __uint_as_float(u.w));
      }
    } else {
      if (trem + BK <= v.tn && kcur + BK <= kend) {   // the tile's BK rows exist and lie in one group (uniform): addressed like the plain case
        const __amdgpu_buffer_rsrc_t r =
            __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(v.p + (long)tgrp * v.sg + (long)trem * v.st), 0, 0x7fffffff, 0x00020000);
#pragma unroll
        for (int p = 0; p < NP; ++p) {
          const u32x4 u = __builtin_amdgcn_raw_buffer_load_b128(r, (int)voff[p], 0, 0);
          reg[slot][p] = make_float4(__uint_as_float(u.x), __uint_as_float(u.y), __uint_as_float(u.z), __uint_as_float(u.w));
        }
      } else {                   // the tile straddles groups (or runs past kend): per-row offsets, rows clamped to kend - 1
#pragma unroll
        for (int p = 0; p < NP; ++p) {
          const int kr = min(kcur + b + RP * p, kend - 1);
          reg[slot][p] = *reinterpret_cast<const float4*>(v.p + (long)(kr / v.tn) * v.sg + (long)(kr % v.tn) * v.st + tcol);
        }
      }
      trem += BK;
      while (trem >= v.tn) { trem -= v.tn; ++tgrp; }
    }
    kcur += BK;
  }
  // RK operands on the split paths: tiles t (-> slot) and t + 1 (-> slot + 1, if `second`) in one go.  A tile row is 64 bytes, half a
  // cache line; fetched one tile at a time the other half is long evicted from the 32 KB vector L1 when the next tile asks for it
  // (4 tiles x 256 rows in flight), so every line crosses the L2 -> L1 path twice.  Issued back to back the two halves are one miss.
  __device__ __forceinline__ void load_pair(const MatView& v, unsigned span, const int slot, bool second) {
    static_assert(RING == 1 || (RING % 2) == 0, "pairs of slots");
    const long adv = (long)kcur;
    const __amdgpu_buffer_rsrc_t r =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(v.p + adv), 0, (int)(span - (unsigned)(adv * 4)), 0x00020000);
#pragma unroll
    for (int p = 0; p < NP; ++p) {
      const u32x4 u = __builtin_amdgcn_raw_buffer_load_b128(r, (int)voff[p], 0, 0);
      reg[slot][p] = make_float4(__uint_as_float(u.x), __uint_as_float(u.y), __uint_as_float(u.z), __uint_as_float(u.w));
      if (second) {
        const u32x4 w = __builtin_amdgcn_raw_buffer_load_b128(r, (int)voff[p] + BK * 4, 0, 0);
        reg[(slot + 1) % RING][p] = make_float4(__uint_as_float(w.x), __uint_as_float(w.y), __uint_as_float(w.z), __uint_as_float(w.w));
      }
    }
    kcur += 2 * BK;
  }
  __device__ __forceinline__ void store(float* S) const {
    if (RK) {
#pragma unroll
      for (int p = 0; p < NP; ++p) {
        *reinterpret_cast<float2*>(&S[lds + p * LD_RK]) = make_float2(reg[0][p].x, reg[0][p].y);
        *reinterpret_cast<float2*>(&S[lds + p * LD_RK + BK / 2]) = make_float2(reg[0][p].z, reg[0][p].w);
      }
    } else {
#pragma unroll
      for (int p = 0; p < NP; ++p) *reinterpret_cast<float4*>(&S[lds + RP * p * LD_KR]) = reg[0][p];
    }
  }
  // split schemes: registers -> the 16-bit planes of the stage at byte address S.  Values with k >= kend are zeroed in registers (uniform
  // branch, last tile of a K range only), so no LDS patching is needed.  ktile: first k of the tile being stored.
  template <bool TAIL>
  __device__ __forceinline__ void store_split(char* S, int ktile, int kend, const int slot) {
    constexpr int PL = sp_plane(TL, RK);
    float4 r[NP];
#pragma unroll
    for (int p = 0; p < NP; ++p) r[p] = reg[slot][p];
    if (TAIL && ktile + BK > kend) {
#pragma unroll
      for (int p = 0; p < NP; ++p) {
        if (RK) {
          const int nv = kend - (ktile + a * 4);
          if (nv < 1) r[p].x = 0.f;
          if (nv < 2) r[p].y = 0.f;
          if (nv < 3) r[p].z = 0.f;
          if (nv < 4) r[p].w = 0.f;
        } else if (ktile + b + RP * p >= kend) r[p] = make_float4(0.f, 0.f, 0.f, 0.f);
      }
    }
#pragma unroll
    for (int p = 0; p < NP; ++p) {
      const int off = RK ? (a >> 1) * sp_hs(TL) + (NP * b + p) * 16 + (a & 1) * 8 : (b + RP * p) * sp_rs(TL) + a * 8;
      if constexpr (PREC == PREC_F16) {
        // (behind the operand's power-of-two scale when its caller measured it -- the backward products' dY / dz would otherwise flush
        //  to zero below 6e-8 and overflow above 65504; scale 1 for operands nobody measured: weights, forward activations)
        const f16x2 lo2 = {(_Float16)(r[p].x * scl), (_Float16)(r[p].y * scl)}, hi2 = {(_Float16)(r[p].z * scl), (_Float16)(r[p].w * scl)};
        *reinterpret_cast<uint2*>(S + off) = make_uint2(__builtin_bit_cast(unsigned, lo2), __builtin_bit_cast(unsigned, hi2));
      } else if constexpr (PREC == PREC_F16X2) {
        unsigned h0, l0, h1, l1;
        split2h(r[p].x, r[p].y, scl, h0, l0);
        split2h(r[p].z, r[p].w, scl, h1, l1);
        *reinterpret_cast<uint2*>(S + off) = make_uint2(h0, h1);
        *reinterpret_cast<uint2*>(S + PL + off) = make_uint2(l0, l1);
      } else {
        unsigned h0, m0, l0, h1, m1, l1;
        split2(r[p].x, r[p].y, h0, m0, l0);
        split2(r[p].z, r[p].w, h1, m1, l1);
        *reinterpret_cast<uint2*>(S + off) = make_uint2(h0, h1);
        *reinterpret_cast<uint2*>(S + PL + off) = make_uint2(m0, m1);
        *reinterpret_cast<uint2*>(S + 2 * PL + off) = make_uint2(l0, l1);
      }
    }
  }
  // Zeroes what this thread's store() wrote for k >= kend (ktile: first k of the tile in S).
  __device__ __forceinline__ void zero_tail(float* S, int ktile, int kend) const {
    if (RK) {
      const int nv = kend - (ktile + a * 4);
#pragma unroll
      for (int p = 0; p < NP; ++p) {
        float* q = &S[lds + p * LD_RK];
        if (nv < 1) q[0] = 0.f;
        if (nv < 2) q[1] = 0.f;
        if (nv < 3) q[BK / 2] = 0.f;
        if (nv < 4) q[BK / 2 + 1] = 0.f;
      }
    } else if (TWOLVL) {
#pragma unroll
      for (int p = 0; p < NP; ++p)
        if (ktile + b + RP * p >= kend) *reinterpret_cast<float4*>(&S[lds + RP * p * LD_KR]) = make_float4(0.f, 0.f, 0.f, 0.f);
    }
  }
};

// Work decomposition ("stream-K"): the launch owns I = tiles * kt k-iterations (kt = ceil(K / BK)), tile-major, and
// workgroup w of G runs the contiguous range [I*w/G, I*(w+1)/G).  A range covers a run of whole tiles plus at most one
// partial tile at each end.  A whole tile is written the way `mode` says; a partial tile is accumulated with atomics
// (for GEMM_STORE the launcher zeroes exactly those tiles first, k_zero_split_tiles).  With G = tiles (one tile each)
// this degenerates to the classical data-parallel launch; with G = 256 or 512 every CU gets the same number of
// k-iterations whatever the tile count -- a 400-tile product no longer runs as "2 waves, the second 56% full".
// The split schemes' kernels run 512 threads: waves 0-3 multiply (LDS fragment reads + MFMAs only), waves 4-7 stage (global loads, the
// f32 -> 3 x bf16 split, LDS writes).  One wave of each kind sits on every SIMD, so the split's vector-ALU work issues in the gaps of
// the other wave's MFMAs; with every wave doing both, the co-resident workgroups ran their VALU and MFMA phases in lockstep and the
// matrix pipe idled more than half of the time (163 instead of 118 TFLOP/s at 4096^3, against > 300 for an MFMA-bound loop).
// MW: multiplying waves of a split-scheme workgroup, 4 (2 x 2 over the tile) or 8 (4 x 2: 256 x 128 tiles, two multiplying waves and one
// staging wave per SIMD); the staging role always has 256 threads.
// Split tiles without a zeroing launch (round 5; GemmGroup::tick): every (split tile, multiplying wave) has one word in a table that is zero
// between launches -- bits 0-14 the k-iterations that have ARRIVED, bit 15 DONE, bits 16-31 the k-iterations that have DEPARTED.  A wave
// with a partial sum adds its nk iterations to the arrivals.  Word was zero: it is the first -- it STORES its sub-tile (written through),
// drains, then raises DONE with its departure.  Otherwise it waits for DONE (the first arrival is inside its epilogue: a bounded wait that
// needs nobody else to be resident) and ADDS with atomics.  The departure that completes the tile's kt iterations writes the word back to
// zero.  Two contributors: first + second, the same sum whoever came first.  (Before: a launch that zeroed the split tiles in front of every
// such GEMM, 6 per train step and ~54 us, and the first arrival's 64 atomics per lane.)
// MEASURED AND SWITCHED OFF (ASTK_GEMM_TICKET = 0, the zeroing launch stays): same box, same call, bench.py events over the family --
// zeroing launch 2.56-2.58 ms; ticket protocol with the release fence the first arrival's plain stores need (L2 write-back of the whole
// XCD) 2.63; without the fence (wrong: the stores can sit in one XCD's L2 while the adds land in memory) 2.53-2.57; with the first arrival's
// stores written through (sc1) or as atomic exchanges instead of the fence -- a THIRD copy of the element loops -- 2.73: the 168-register
// kernel's epilogue then spills ~60 values around the loops and every reload waits, by s_waitcnt vmcnt(0), for the epilogue's own stores.
// The arrivals' round trips cost what the zeroing launches cost; the protocol itself is correct (tests/protocol_model.py, -m gpu test).
// The instrumented build (libastk_test.so, -DASTK_TEST_HOOKS) compiles it IN, so that the protocol keeps being exercised on the GPU
// (test_gemm_split_tiles_ticket_protocol_on_the_instrumented_build).
#ifndef ASTK_GEMM_TICKET
#ifdef ASTK_TEST_HOOKS
#define ASTK_GEMM_TICKET 1
#else
#define ASTK_GEMM_TICKET 0
#endif
#endif
constexpr int TICK_WAVES = 8;
constexpr unsigned TICK_DONE = 1u << 15;
constexpr int gemm_mw(int TLM) { return TLM == 256 ? 8 : 4; }
constexpr int gemm_threads(int PREC, int TLM = 128) { return PREC != PREC_F32 ? 256 + 64 * gemm_mw(TLM) : 256; }
constexpr int gemm_min_waves(int TL, int PREC, int TLM) { return (PREC != PREC_F32 && TLM == 256) ? 3 : waves_per_simd(TL, PREC); }
// TL: tile edge along N (and along M unless TLM says otherwise: the bf16x3 path also runs 256 x 128 tiles -- per k-iteration the split
// costs vector-ALU issue slots in proportion to TLM + TL while the MFMAs grow with TLM x TL, and only from 256 x 128 on do the MFMAs
// (1536 cycles per wave and iteration) outlast the split's issue time on the same SIMD).
template <int TL, bool A_RK, bool B_RK, bool TWOLVL, int PREC, int TLM = TL, bool FIX = false>
__global__ __launch_bounds__(gemm_threads(PREC, TLM), gemm_min_waves(TL, PREC, TLM)) void gemm_f32_kernel(GemmGroup grp) {
  static_assert(!FIX || (PREC == PREC_BF16X3 && TL == 128 && TLM == 128), "the deterministic fix-up is instantiated for the 128 x 128 bf16x3 kernel");
  constexpr int LD_KR = ld_kr(TL);
  constexpr int LDA = A_RK ? LD_RK : LD_KR;
  constexpr int LDB = B_RK ? LD_RK : LD_KR;
  constexpr bool SPLIT = PREC != PREC_F32;     // operands staged as 16-bit planes (three bf16 terms, or one fp16 term)
  constexpr int MW = SPLIT ? gemm_mw(TLM) : 4; // multiplying waves, MW/2 x 2 over the tile
  constexpr int WT = TL / 2, WTM = TLM / (MW / 2);    // the wave's sub-tile is WTM x WT
  constexpr int NA = WT / 32;      // 32x32 accumulator tiles per wave and dimension
  constexpr int NAM = WTM / 32;
  constexpr int NPL = prec_planes(PREC);
  static_assert(SPLIT || TLM == TL, "the f32 path runs square tiles");
  constexpr int A_FLOATS = SPLIT ? sp_stage(TLM, A_RK, PREC) / 4 : (A_RK ? TL * LD_RK : BK * LD_KR);
  constexpr int B_FLOATS = SPLIT ? sp_stage(TL, B_RK, PREC) / 4 : (B_RK ? TL * LD_RK : BK * LD_KR);
  constexpr int NST = SPLIT ? 3 : 2;      // LDS stages (split schemes: the multiplying waves fetch tile kt+1's fragments while they multiply tile kt)
  __shared__ __attribute__((aligned(16))) float As[NST][A_FLOATS];
  __shared__ __attribute__((aligned(16))) float Bs[NST][B_FLOATS];
  static_assert(BK == 16 || BK == 32, "BK / 2 floats per lane and operand, read as BK / 8 float4");

  const bool producer = SPLIT && threadIdx.x >= 64 * MW;   // split schemes: the last four waves stage, waves 0 .. MW-1 multiply; f32: every wave does both
  const int tid = producer ? threadIdx.x - 64 * MW : threadIdx.x;      // stager / multiplier thread index inside its role
  const bool stages = !SPLIT || producer, multiplies = !SPLIT || !producer;
  const int lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int li = lane & 31, lk = lane >> 5;

  // Workgroups are dealt round-robin to the 8 XCDs (blockIdx % 8 share one L2).  Split schemes: give every XCD a CONTIGUOUS eighth of the
  // k-iteration sequence, so that the workgroups sharing an L2 work on neighbouring tiles (same A row panel, adjacent B panels): at
  // 160+ TFLOP/s a 128-tile loop pulls > 5 TB/s of f32 operands into the CUs, which only the L2s can serve.
  if constexpr (SPLIT) {
    // One multiplying and one staging wave share each SIMD's issue port.  At equal priority the arbiter lets the staging wave's
    // vector-ALU instructions queue in front of the next MFMA, and the matrix pipe idles between MFMAs (measured: multiplying waves
    // alone 1100 cycles per k-iteration, staging waves alone 600, both together 1600-1900 = the SUM).  Static priority for the
    // multiplying waves: their MFMAs issue the moment the pipe is free, the split fills the gaps.  (The branch must be provably
    // wave-uniform: s_setprio ignores EXEC.)
    if (__builtin_amdgcn_readfirstlane((int)threadIdx.x) < 64 * MW) __builtin_amdgcn_s_setprio(MW == 8 ? ASTK_GEMM_PRIO8 : 3);
    else if (MW == 8 && ASTK_GEMM_PRIO8S != 0) __builtin_amdgcn_s_setprio(ASTK_GEMM_PRIO8S);
  }
  unsigned wgi = blockIdx.x;
  if (SPLIT && (gridDim.x % 8) == 0) wgi = (blockIdx.x % 8) * (gridDim.x / 8) + blockIdx.x / 8;
  // Hybrid schedule: phases 0 .. dp_waves-1 are data-parallel waves -- in wave ph the gridDim.x / 8 workgroups of XCD x own the
  // consecutive tiles [(8 ph + x) G/8, (8 ph + x + 1) G/8) of the block order, a bm x (G/8/bm) block of the output whose 32 workgroups
  // start together and march through k side by side, so the XCD's L2 serves every operand panel of the block to all its users; the
  // last phase is the stream-K split of what is left.
  const int n_phases = grp.dp_waves + 1;
  for (int ph = 0; ph < n_phases; ++ph) {
  long it, it_end;
  if (ph < grp.dp_waves) {
    const long T = ((long)ph * 8 + blockIdx.x % 8) * (gridDim.x / 8) + blockIdx.x / 8;
    it = T * grp.dp_kt;
    it_end = it + grp.dp_kt;
  } else {
    it = wg_first_iter(grp, wgi, gridDim.x);
    it_end = wg_first_iter(grp, wgi + 1, gridDim.x);
  }
  int prob = 0;

  while (it < it_end) {
    while (it >= grp.iter_start[prob + 1]) ++prob;
    const GemmArgs& g = grp.g[prob];
    const int kt_tile = g.kt;
    const long lit = it - grp.iter_start[prob];
    const long lend = min(it_end, grp.iter_start[prob + 1]) - grp.iter_start[prob];
    long tile;
    int k0, k1;
    if (g.cs > 0) {
      // chunk-major (accumulating products of few tiles and deep K): consecutive workgroups -- the 32 of an XCD -- then work on the SAME
      // k-range of neighbouring tiles and share its operand lines in their L2, instead of 32 different k-ranges of one or two tiles
      const long c = lit / g.chunk_iters, r = lit - c * g.chunk_iters;
      tile = r / g.cs;
      const int kk = (int)(r - tile * g.cs);
      k0 = (int)c * g.cs + kk;
      k1 = (int)min((long)(k0 - kk + g.cs), (long)k0 + (lend - lit));
    } else {
      tile = lit / kt_tile;
      k0 = (int)(lit - tile * kt_tile);
      k1 = (int)min((long)kt_tile, (long)k0 + (lend - lit));
    }
    it += k1 - k0;
    int zb, m0, n0;
    decode_tile(g, tile, TLM, TL, zb, m0, n0);
    const int m0_t = m0, n0_t = n0;
    const int kbeg = k0 * BK;
    const int kend = min(g.K, k1 * BK);
    const bool whole = (k0 == 0) && (k1 == kt_tile);

    MatView A = g.A, B = g.B;
    A.p += (long)zb * g.sA;
    B.p += (long)zb * g.sB;
    float* C = g.C + (long)zb * g.sC;

    // (the 12-wave kernel has 168 registers per lane: two staged tiles of 256 + 128 rows in flight, 48 registers, instead of four)
    constexpr int RING = SPLIT ? (MW == 8 ? ASTK_GEMM_RING8 : ASTK_GEMM_RING) : 1;
    Stager<TLM, A_RK, TWOLVL, PREC, RING> sa;
    Stager<TL, B_RK, TWOLVL, PREC, RING> sb;
    // (every wave runs init: a field assigned only under the role branch, which depends on threadIdx, is a divergent value to hipcc
    //  -- and a divergent kcur puts the operand base pointer into vector registers and a waterfall loop around every buffer load)
    sa.init(A, m0, g.M, kbeg, kend, tid & 255);
    sb.init(B, n0, g.N, kbeg, kend, tid & 255);
    int seA = 127, seB = 127;
    if constexpr (PREC == PREC_F16X2 || PREC == PREC_F16) {
      seA = scale_exp(g.amaxA);
      seB = scale_exp(g.amaxB);
      sa.scl = __uint_as_float((unsigned)seA << 23);
      sb.scl = __uint_as_float((unsigned)seB << 23);
    }

    const int nk = k1 - k0;
    const bool add_bias = g.bias != nullptr && k0 == 0;
    // (GEMM_ACCUM goes out as atomic adds like the split tiles: a read-modify-write per element waits for its own load, 64 times per
    //  lane -- 6400 x 3072 x 1024: 306 us against 232 with atomics)
    const int mode = !whole || g.mode == GEMM_ACCUM ? GEMM_ATOMIC : g.mode;
    // A split tile of a GEMM_STORE product under the ticket protocol (see "Split tiles without a zeroing launch" above the kernel)
    // (carried through the k loop as ONE scalar: the word's index, -1 = no ticket; the address is formed in the epilogue -- `wave` is a
    //  vector value to the compiler, and a per-lane 64-bit address alive through the loop costs the 168-register kernel two registers it
    //  does not have: 60-100 spilled registers and a family 7 % slower, measured)
    const int tick_idx = (ASTK_GEMM_TICKET && !whole && g.mode == GEMM_STORE && grp.tick != nullptr) ? (grp.tick_base[prob] + (int)tile) * TICK_WAVES : -1;
    // epilogue: C/D layout of the 32x32 MFMA: col = lane&31, row = (reg&3) + 8*(reg>>2) + 4*(lane>>5)
    // (everything the element loop needs from the launch arguments is copied into locals first: read through `g`, a reference into
    //  the kernel-argument block, hipcc re-loads the field behind every global store -- the stores might alias it -- and waits for the
    //  scalar load, 64 times per lane and tile)
    const float* const e_bias = add_bias ? g.bias : nullptr;
    // FIX: the tile's first iteration in the launch's sequence, -1 for a whole tile (everything else about the split is derived in the
    // epilogue, behind the k loop)
    const long fix_gs0 = (FIX && !whole) ? grp.iter_start[prob] + tile * kt_tile : -1;
    const int g_mode = g.mode, e_kt = g.kt;
    const int e_M = g.M, e_N = g.N, e_ctn = g.c_tn;
    const long e_ldc = g.ldc, e_csg = g.c_sg, e_cst = g.c_st;
    auto epilogue = [&](f32x16 (&acc)[NAM][NA]) {
      // (the tile's origin through an opaque copy: the row / column offsets below depend on nothing the k loop computes, and hipcc would
      //  otherwise form them in FRONT of the loop and carry ~36 registers of addresses through it -- spills in the 168-register kernel)
      int m0 = m0_t, n0 = n0_t;
      asm volatile("" : "+v"(m0), "+v"(n0));
      if (e_bias) {       // (uniform) this lane's two bias values go into the accumulators first: no load behind the stores
#pragma unroll
        for (int j = 0; j < NA; ++j) {
          const int col = n0 + wn * WT + j * 32 + li;
          const float b = e_bias[min(col, e_N - 1)];
#pragma unroll
          for (int i = 0; i < NAM; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] += b;
        }
      }
      const int col0 = n0 + wn * WT + li;
      int out_mode = mode;
      if constexpr (FIX) {
        if (fix_gs0 >= 0) {                     // (uniform) a split tile of a deterministic launch
          int gl = (int)(fix_gs0 & 0xffffffffL), gh = (int)(fix_gs0 >> 32);
          asm volatile("" : "+v"(gl), "+v"(gh));          // (opaque: keeps this derivation behind the k loop)
          const long gs = ((long)__builtin_amdgcn_readfirstlane(gh) << 32) | (unsigned)__builtin_amdgcn_readfirstlane(gl), ge = gs + e_kt;
          const unsigned Gn = gridDim.x;
          // the contributors: the workgroups whose ranges meet [gs, ge), found by walking from this workgroup's own range
          int lo = (int)wgi, hi = (int)wgi, n = 0;
          while (lo > 0 && wg_first_iter(grp, lo, Gn) > gs) --lo;
          while (hi + 1 < (int)Gn && wg_first_iter(grp, hi + 1, Gn) < ge) ++hi;
          for (int w2 = lo; w2 <= hi; ++w2) n += wg_first_iter(grp, w2 + 1, Gn) > wg_first_iter(grp, w2, Gn) ? 1 : 0;
          if (n > 1) {
            const long fm = wg_first_iter(grp, wgi, Gn), fl = wg_first_iter(grp, lo, Gn);
            const int slot = 2 * (int)wgi + ((fm >= gs && fm < ge) ? 0 : 1);
            const int key = 2 * lo + ((fl >= gs && fl < ge) ? 0 : 1);
            const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(g_fix_part, 0, 0x7fffffff, 0x00020000);
            const int mine = (int)(((size_t)slot * FIX_WAVES + wave) * FIX_BLOCK * 4) + lane * 16;
#pragma unroll
            for (int i = 0; i < NAM; ++i)
#pragma unroll
              for (int j = 0; j < NA; ++j)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                  u32x4 o;
                  o.x = __float_as_uint(acc[i][j][4 * q]); o.y = __float_as_uint(acc[i][j][4 * q + 1]);
                  o.z = __float_as_uint(acc[i][j][4 * q + 2]); o.w = __float_as_uint(acc[i][j][4 * q + 3]);
                  __builtin_amdgcn_raw_buffer_store_b128(o, rs, mine + ((i * NA + j) * 4 + q) * 1024, 0, 16);
                }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // this wave's block has left for memory ...
            unsigned ticket = 0;
            if (lane == 0) ticket = __hip_atomic_fetch_add(&g_fix_ctr[key * FIX_WAVES + wave], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            ticket = __builtin_amdgcn_readfirstlane(ticket);       // ... before its ticket is drawn; the loads below depend on the returned value
            if ((int)ticket != n - 1) return;                      // not the last contributor of this sub-tile: done
            if (lane == 0) __hip_atomic_store(&g_fix_ctr[key * FIX_WAVES + wave], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#pragma unroll
            for (int i = 0; i < NAM; ++i)
#pragma unroll
              for (int j = 0; j < NA; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
            for (int w2 = lo; w2 <= hi; ++w2) {                    // every contributor's block, in workgroup order (its own from memory too)
              const long f2 = wg_first_iter(grp, w2, Gn);
              if (wg_first_iter(grp, w2 + 1, Gn) <= f2) continue;      // (an empty range contributes nothing)
              const int slot2 = 2 * w2 + ((f2 >= gs && f2 < ge) ? 0 : 1);
              const int base = (int)(((size_t)slot2 * FIX_WAVES + wave) * FIX_BLOCK * 4) + lane * 16;
#pragma unroll
              for (int i = 0; i < NAM; ++i)
#pragma unroll
                for (int j = 0; j < NA; ++j) {
                  u32x4 t[4];
#pragma unroll
                  for (int q = 0; q < 4; ++q) t[q] = __builtin_amdgcn_raw_buffer_load_b128(rs, base + ((i * NA + j) * 4 + q) * 1024, 0, 16);
#pragma unroll
                  for (int q = 0; q < 4; ++q) {
                    acc[i][j][4 * q] += __uint_as_float(t[q].x); acc[i][j][4 * q + 1] += __uint_as_float(t[q].y);
                    acc[i][j][4 * q + 2] += __uint_as_float(t[q].z); acc[i][j][4 * q + 3] += __uint_as_float(t[q].w);
                  }
                }
            }
            out_mode = g_mode == GEMM_ACCUM ? GEMM_ATOMIC : g_mode;      // the finished tile goes out the way the caller asked
          }
        }
      }
      // Ticket protocol of a split GEMM_STORE tile.  The element loops below are the two the kernel always had (plain stores / atomic
      // adds), byte for byte: a third copy with written-through stores, or ticket state alive across them, pushed the 168-register kernel's
      // epilogue over its budget -- 100 scratch reloads, each behind an s_waitcnt vmcnt(0) that also drains the epilogue's own stores: the
      // whole family 7 % slower (measured, same box).  So: arrive (and, not being first, wait for DONE) in FRONT of the loops; the first
      // arrival runs the plain-store copy and makes it visible with one release fence (L2 write-back) behind them; the word's address is
      // formed twice, each time from the one scalar that crossed the k loop.
      // Round trips on the way: ONE for the usual second arrival (its arrival returns a word with DONE up and everybody before it
      // departed: no poll, and it knows it is the last to leave -- its reset is a plain store), one plus the fence for the first (its
      // departure returns nothing it needs: nobody can have departed before DONE).
      bool tick_first = false;
      int tick_last = 0;        // 1: the arrival already showed that this wave is the last to leave
      if constexpr (ASTK_GEMM_TICKET) {
        int tick_i = tick_idx;
        asm volatile("" : "+s"(tick_i));      // (opaque: nothing below is formed in front of the loop)
        if (tick_i >= 0) {       // (wave-uniform) this wave's arrival at its quarter / eighth of the split tile
          unsigned* const tick_word = grp.tick + tick_i + __builtin_amdgcn_readfirstlane(wave);
          unsigned old = 0;
          if (lane == 0) old = __hip_atomic_fetch_add(tick_word, (unsigned)nk, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          old = (unsigned)__builtin_amdgcn_readfirstlane((int)old);
          tick_first = old == 0;
          const int arrived = (int)(old & (TICK_DONE - 1u)), departed = (int)(old >> 16);
          tick_last = (old & TICK_DONE) != 0u && departed == arrived && arrived + nk == kt_tile;
          if (!tick_first && !(old & TICK_DONE) && lane == 0) {      // the first arrival is inside its epilogue: a bounded wait
            unsigned spins = 0;
            while (!(__hip_atomic_load(tick_word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) & TICK_DONE) && ++spins < (1u << 26)) __builtin_amdgcn_s_sleep(1);
          }
        }
      }
      auto body = [&](auto storec) {          // one copy of the element loops per output mode (decided once per tile)
        constexpr bool STORE = decltype(storec)::value;
#pragma unroll
        for (int i = 0; i < NAM; ++i) {
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int row = m0 + wm * WTM + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lk;
            if (row >= e_M) continue;
            const long coff = e_ctn > 0 ? (long)(row / e_ctn) * e_csg + (long)(row % e_ctn) * e_cst : (long)row * e_ldc;
            float* rp = C + coff + col0;       // one 64-bit address per row, the column tiles at immediate offsets
#pragma unroll
            for (int j = 0; j < NA; ++j) {
              if (col0 + j * 32 >= e_N) continue;
              if constexpr (STORE) rp[j * 32] = acc[i][j][r];
              else atomicAdd(rp + j * 32, acc[i][j][r]);
            }
          }
        }
      };
      if (out_mode == GEMM_STORE || tick_first) body(std::true_type{});
      else body(std::false_type{});
      if constexpr (ASTK_GEMM_TICKET) {
        int tick_i = tick_idx;
        asm volatile("" : "+s"(tick_i));
        if (tick_i >= 0) {
          // departure: the first arrival's stores are in memory (release fence: this XCD's L2 written back, stores drained) before DONE goes
          // up; the departure that completes the tile's k-iterations puts the word back to zero for the next launch -- every contributor
          // has passed its poll by then
          unsigned* const tick_word = grp.tick + tick_i + __builtin_amdgcn_readfirstlane(wave);
          if (tick_first) __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
          if (lane == 0) {
            if (tick_first) {
              (void)__hip_atomic_fetch_add(tick_word, ((unsigned)nk << 16) | TICK_DONE, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);      // (no one left before DONE: not the last)
            } else if (tick_last) {
              __hip_atomic_store(tick_word, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            } else {
              const unsigned old2 = __hip_atomic_fetch_add(tick_word, (unsigned)nk << 16, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
              if ((int)(old2 >> 16) + nk == kt_tile) __hip_atomic_store(tick_word, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
          }
        }
      }
    };

    if constexpr (SPLIT) {
      // ---- pipeline of the split schemes.  Staging waves: tile t sits in ring slot t % RING (registers) until it is split into LDS buffer t & 1 one
      // iteration before the multiplying waves use it; a slot is reloaded with tile t + RING right after.  The two roles run separate
      // loops with the same number of barriers.  The staging loop's steady state is straight-line code (no tail handling, no
      // conditional loads), so that the compiler's s_waitcnt counts the RING tiles in flight instead of draining them.
      // LDS: three stages; tile t lives in stage t % 3.  Iteration kt: the staging waves write tile kt+2, the multiplying waves read
      // tile kt+1's fragments into registers while their MFMAs run on tile kt's (read one iteration earlier): the LDS read burst of
      // 48 KB per workgroup and iteration (190+ cycles of the LDS pipe, plus latency) is off the matrix pipe's critical path.
      auto stA = [&](int st) { return reinterpret_cast<char*>(As[0]) + st * (A_FLOATS * 4); };
      auto stB = [&](int st) { return reinterpret_cast<char*>(Bs[0]) + st * (B_FLOATS * 4); };
      // refill(st, V, span, t, slot): tile t has just left register slot `slot` = t % RING; fetch tile t + RING into it.  K-contiguous
      // operands fetch PAIRS of tiles (Stager::load_pair) when the odd slot of a pair comes free.
      auto refill = [&](auto& st, const MatView& V, unsigned span, auto rkc, auto steadyc, const int t, const int slot) {
        constexpr bool PAIR = decltype(rkc)::value && ASTK_GEMM_PAIR && BK == 16;
        constexpr bool STEADY = decltype(steadyc)::value;       // every tile asked for exists
        if constexpr (PAIR) {
          if (slot & 1) {
            if (STEADY || t - 1 + RING < nk) st.load_pair(V, span, slot - 1, STEADY || t + RING < nk);
          }
        } else {
          if (STEADY || t + RING < nk) st.load(V, span, kend, slot);
        }
      };
      auto prefetch = [&](auto& st, const MatView& V, unsigned span, auto rkc) {      // tiles 0 .. RING-1 of the range
        constexpr bool PAIR = decltype(rkc)::value && ASTK_GEMM_PAIR && BK == 16;
        if constexpr (PAIR) {
#pragma unroll
          for (int r = 0; r < RING; r += 2)
            if (r < nk) st.load_pair(V, span, r, r + 1 < nk);
        } else {
#pragma unroll
          for (int r = 0; r < RING; ++r)
            if (r < nk) st.load(V, span, kend, r);
        }
      };
      constexpr std::integral_constant<bool, A_RK> ARK{};
      constexpr std::integral_constant<bool, B_RK> BRK{};
      if (producer) {
        prefetch(sa, A, g.spanA, ARK);
        prefetch(sb, B, g.spanB, BRK);
        sa.template store_split<true>(stA(0), kbeg, kend, 0);
        sb.template store_split<true>(stB(0), kbeg, kend, 0);
        refill(sa, A, g.spanA, ARK, std::false_type{}, 0, 0);
        refill(sb, B, g.spanB, BRK, std::false_type{}, 0, 0);
        if (1 < nk) {
          sa.template store_split<true>(stA(1), kbeg + BK, kend, 1);
          sb.template store_split<true>(stB(1), kbeg + BK, kend, 1);
          refill(sa, A, g.spanA, ARK, std::false_type{}, 1, 1);
          refill(sb, B, g.spanB, BRK, std::false_type{}, 1, 1);
        }
        __syncthreads();
        int kt = 0;
        // iteration kt: tile kt+2 -> LDS stage (kt+2) % 3 from slot (kt+2) % RING, then tile kt+2+RING -> that slot
        auto pstep = [&](auto posc, auto tailc, const int kt_) {
          constexpr int pos = decltype(posc)::value;          // kt_ % 12
          constexpr bool TAIL = decltype(tailc)::value;
          constexpr int slot = (pos + 2) % RING, st = (pos + 2) % 3;
          // (A's slot is reloaded right behind A's split: its loads' time in the L1's queue -- 16 cycles per 1 KB instruction -- runs beside B's split)
          if (!TAIL || kt_ + 2 < nk) sa.template store_split<TAIL>(stA(st), kbeg + (kt_ + 2) * BK, kend, slot);
          refill(sa, A, g.spanA, ARK, std::integral_constant<bool, !TAIL>{}, kt_ + 2, slot);
          if (!TAIL || kt_ + 2 < nk) sb.template store_split<TAIL>(stB(st), kbeg + (kt_ + 2) * BK, kend, slot);
          refill(sb, B, g.spanB, BRK, std::integral_constant<bool, !TAIL>{}, kt_ + 2, slot);
          __syncthreads();
        };
        // steady state (12 = lcm(3 stages, 4 slots) iterations per trip): every tile stored in the trip is a full tile, every tile loaded exists
        for (; kt + 13 + RING < nk && kbeg + (kt + 14) * BK <= kend; kt += 12) {
          pstep(std::integral_constant<int, 0>{}, std::false_type{}, kt);
          pstep(std::integral_constant<int, 1>{}, std::false_type{}, kt + 1);
          pstep(std::integral_constant<int, 2>{}, std::false_type{}, kt + 2);
          pstep(std::integral_constant<int, 3>{}, std::false_type{}, kt + 3);
          pstep(std::integral_constant<int, 4>{}, std::false_type{}, kt + 4);
          pstep(std::integral_constant<int, 5>{}, std::false_type{}, kt + 5);
          pstep(std::integral_constant<int, 6>{}, std::false_type{}, kt + 6);
          pstep(std::integral_constant<int, 7>{}, std::false_type{}, kt + 7);
          pstep(std::integral_constant<int, 8>{}, std::false_type{}, kt + 8);
          pstep(std::integral_constant<int, 9>{}, std::false_type{}, kt + 9);
          pstep(std::integral_constant<int, 10>{}, std::false_type{}, kt + 10);
          pstep(std::integral_constant<int, 11>{}, std::false_type{}, kt + 11);
        }
        for (; kt < nk; kt += 12) {
          pstep(std::integral_constant<int, 0>{}, std::true_type{}, kt);
          if (kt + 1 < nk) pstep(std::integral_constant<int, 1>{}, std::true_type{}, kt + 1);
          if (kt + 2 < nk) pstep(std::integral_constant<int, 2>{}, std::true_type{}, kt + 2);
          if (kt + 3 < nk) pstep(std::integral_constant<int, 3>{}, std::true_type{}, kt + 3);
          if (kt + 4 < nk) pstep(std::integral_constant<int, 4>{}, std::true_type{}, kt + 4);
          if (kt + 5 < nk) pstep(std::integral_constant<int, 5>{}, std::true_type{}, kt + 5);
          if (kt + 6 < nk) pstep(std::integral_constant<int, 6>{}, std::true_type{}, kt + 6);
          if (kt + 7 < nk) pstep(std::integral_constant<int, 7>{}, std::true_type{}, kt + 7);
          if (kt + 8 < nk) pstep(std::integral_constant<int, 8>{}, std::true_type{}, kt + 8);
          if (kt + 9 < nk) pstep(std::integral_constant<int, 9>{}, std::true_type{}, kt + 9);
          if (kt + 10 < nk) pstep(std::integral_constant<int, 10>{}, std::true_type{}, kt + 10);
          if (kt + 11 < nk) pstep(std::integral_constant<int, 11>{}, std::true_type{}, kt + 11);
        }
      } else {
        // (the accumulators live in this branch only: the staging waves' code path must not carry 128 registers of them)
        f32x16 acc[NAM][NA];
#pragma unroll
        for (int i = 0; i < NAM; ++i)
#pragma unroll
          for (int j = 0; j < NA; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
        __syncthreads();
        // operand fragments of v_mfma_f32_32x32x16_bf16: lane (r = lane & 31, h = lane >> 5) holds k = 8h .. 8h+7 of row / column r
        auto frag = [&](const char* base, bool rk, auto tlc, int t0, int pl, int ks) -> bf16x8 {      // ks: 16-k step inside the tile
          constexpr int TLX = decltype(tlc)::value;          // tile edge of this operand
          if (rk) return __builtin_bit_cast(bf16x8, *reinterpret_cast<const uint4*>(base + pl * sp_plane(TLX, true) + (2 * ks + lk) * sp_hs(TLX) + (t0 + li) * 16));
          // K-major image: two transposing reads (k = 8h .. 8h+3 and 8h+4 .. 8h+7) of a 4 k x 16 m block per 16-lane group
          const int g16 = (lane >> 4) & 1, q = (lane & 15) >> 2, pp = lane & 3;
          const char* ad = base + pl * sp_plane(TLX, false) + (16 * ks + 8 * lk + q) * sp_rs(TLX) + (t0 + 16 * g16 + 4 * pp) * 2;
          typedef __attribute__((address_space(3))) s16x4 lds_s16x4;
          const s16x4 v0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(ad));
          const s16x4 v1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(ad + 4 * sp_rs(TLX)));
          struct { s16x4 a, b; } pr = {v0, v1};
          return __builtin_bit_cast(bf16x8, pr);
        };
        constexpr int NKS = BK / 16;        // MFMA k-steps per tile
        struct Frags { bf16x8 a[NKS][NAM][NPL], b[NKS][NA][NPL]; };
        auto fetch = [&](Frags& f, int st) {
          const char* abase = stA(st);
          const char* bbase = stB(st);
#pragma unroll
          for (int ks = 0; ks < NKS; ++ks)
#pragma unroll
            for (int pl = 0; pl < NPL; ++pl) {
#pragma unroll
              for (int i = 0; i < NAM; ++i) f.a[ks][i][pl] = frag(abase, A_RK, std::integral_constant<int, TLM>{}, wm * WTM + 32 * i, pl, ks);
#pragma unroll
              for (int i = 0; i < NA; ++i) f.b[ks][i][pl] = frag(bbase, B_RK, std::integral_constant<int, TL>{}, wn * WT + 32 * i, pl, ks);
            }
        };
        auto mult = [&](const Frags& f) {
#pragma unroll
          for (int ks = 0; ks < NKS; ++ks) {
          if constexpr (PREC == PREC_F16) {
#pragma unroll
            for (int i = 0; i < NAM; ++i)
#pragma unroll
              for (int i2 = 0; i2 < NA; ++i2)
                acc[i][i2] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, f.a[ks][i][0]), __builtin_bit_cast(f16x8, f.b[ks][i2][0]),
                                                                    acc[i][i2], 0, 0, 0);
          } else if constexpr (PREC == PREC_F16X2) {
            constexpr int PA[3] = {1, 0, 0}, PB[3] = {0, 1, 0};      // lo.hi, hi.lo, hi.hi
#pragma unroll
            for (int t = 0; t < 3; ++t)
#pragma unroll
              for (int i = 0; i < NAM; ++i)
#pragma unroll
                for (int i2 = 0; i2 < NA; ++i2)
                  acc[i][i2] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, f.a[ks][i][PA[t]]), __builtin_bit_cast(f16x8, f.b[ks][i2][PB[t]]), acc[i][i2], 0, 0, 0);
          } else {
          // smallest terms first: lo.hi, hi.lo, mid.mid, mid.hi, hi.mid, hi.hi
          constexpr int PA[6] = {2, 0, 1, 1, 0, 0}, PB[6] = {0, 2, 1, 0, 1, 0};
#pragma unroll
          for (int t = 0; t < 6; ++t)
#pragma unroll
            for (int i = 0; i < NAM; ++i)
#pragma unroll
              for (int i2 = 0; i2 < NA; ++i2)
                acc[i][i2] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f.a[ks][i][PA[t]], f.b[ks][i2][PB[t]], acc[i][i2], 0, 0, 0);
          }
        }
        };
        if constexpr (MW == 8 && PREC == PREC_BF16X3) {
        // ---- 256 x 128 tiles, eight multiplying waves (two per SIMD) of 64 x 64: 168 registers per lane, so only the hi planes are double
        // buffered.  Iteration kt: the mid / lo fragments of tile kt are fetched at the top and land behind the four hi.hi MFMAs, whose
        // operands arrived during iteration kt-1; the hi fragments of tile kt+1 are fetched behind them.  (The order of the six term
        // products inside one 16-k block is free: every one of them is added to an accumulator that already holds the sums of all earlier k.)
        struct FragsH { bf16x8 a[NAM], b[NA]; };
        struct FragsML { bf16x8 a[NAM][2], b[NA][2]; };      // [.][0] mid, [.][1] lo
        // Fragment addresses = (stage, plane, sub-tile: constants of the unrolled code) + this lane's offset inside an operand image.  The lane
        // offsets are made OPAQUE at every fetch: with three stages the images span 110-130 KB, beyond the 64 KB a DS instruction's immediate
        // reaches, and hipcc kept lane offset + 64 K (+ 128 K) per operand as loop invariants -- three of them SPILLED in this 168-register
        // kernel, reloaded (scratch_load + s_waitcnt vmcnt(0)) in front of the fragment reads of every k-iteration.  One v_add per window
        // where it is used costs nothing next to that.
        const int g16_ = (lane >> 4) & 1, q_ = (lane & 15) >> 2, pp_ = lane & 3;
        const int lane_a = A_RK ? lk * sp_hs(TLM) + (wm * WTM + li) * 16 : (8 * lk + q_) * sp_rs(TLM) + (wm * WTM + 16 * g16_ + 4 * pp_) * 2;
        const int lane_b = B_RK ? lk * sp_hs(TL) + (wn * WT + li) * 16 : (8 * lk + q_) * sp_rs(TL) + (wn * WT + 16 * g16_ + 4 * pp_) * 2;
        auto frag8 = [&](const char* base, bool rk, auto tlc, int t0, int pl, int lane_off) -> bf16x8 {      // frag(.., ks = 0) with the lane part given
          constexpr int TLX = decltype(tlc)::value;
          if (rk) return __builtin_bit_cast(bf16x8, *reinterpret_cast<const uint4*>(base + pl * sp_plane(TLX, true) + t0 * 16 + lane_off));
          const char* ad = base + pl * sp_plane(TLX, false) + t0 * 2 + lane_off;
          typedef __attribute__((address_space(3))) s16x4 lds_s16x4;
          const s16x4 v0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(ad));
          const s16x4 v1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(ad + 4 * sp_rs(TLX)));
          struct { s16x4 a, b; } pr = {v0, v1};
          return __builtin_bit_cast(bf16x8, pr);
        };
        auto fetch_h = [&](FragsH& f, int st) {
          int la = lane_a, lb = lane_b;
          asm volatile("" : "+v"(la), "+v"(lb));
#pragma unroll
          for (int i = 0; i < NAM; ++i) f.a[i] = frag8(stA(st), A_RK, std::integral_constant<int, TLM>{}, 32 * i, 0, la);
#pragma unroll
          for (int i = 0; i < NA; ++i) f.b[i] = frag8(stB(st), B_RK, std::integral_constant<int, TL>{}, 32 * i, 0, lb);
        };
        auto fetch_ml = [&](FragsML& f, int st) {
          int la = lane_a, lb = lane_b;
          asm volatile("" : "+v"(la), "+v"(lb));
#pragma unroll
          for (int pl = 1; pl < 3; ++pl) {
#pragma unroll
            for (int i = 0; i < NAM; ++i) f.a[i][pl - 1] = frag8(stA(st), A_RK, std::integral_constant<int, TLM>{}, 32 * i, pl, la);
#pragma unroll
            for (int i = 0; i < NA; ++i) f.b[i][pl - 1] = frag8(stB(st), B_RK, std::integral_constant<int, TL>{}, 32 * i, pl, lb);
          }
        };
        static_assert(BK == 16 || !(MW == 8), "the 12-wave kernel is written for 16-deep tiles");
        FragsH h0, h1;
        FragsML ml;
        fetch_h(h0, 0);
        auto cstep8 = [&](auto posc, const int kt_) {
          constexpr int pos = decltype(posc)::value;
          FragsH& hc = (pos & 1) ? h1 : h0;
          FragsH& hn = (pos & 1) ? h0 : h1;
          {
          fetch_ml(ml, pos % 3);
#pragma unroll
          for (int i = 0; i < NAM; ++i)
#pragma unroll
            for (int i2 = 0; i2 < NA; ++i2) acc[i][i2] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(hc.a[i], hc.b[i2], acc[i][i2], 0, 0, 0);
          __builtin_amdgcn_sched_barrier(0);
          fetch_h(hn, (pos + 1) % 3);         // (unconditional: behind the range's last tile this reads a stage nobody writes, and nobody uses the result)
          // mid.hi, hi.mid, mid.mid, lo.hi, hi.lo
#pragma unroll
          for (int t = 0; t < 5; ++t)
#pragma unroll
            for (int i = 0; i < NAM; ++i)
#pragma unroll
              for (int i2 = 0; i2 < NA; ++i2) {
                const bf16x8 av = t == 0 ? ml.a[i][0] : t == 1 ? hc.a[i] : t == 2 ? ml.a[i][0] : t == 3 ? ml.a[i][1] : hc.a[i];
                const bf16x8 bv = t == 0 ? hc.b[i2] : t == 1 ? ml.b[i2][0] : t == 2 ? ml.b[i2][0] : t == 3 ? hc.b[i2] : ml.b[i2][1];
                acc[i][i2] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av, bv, acc[i][i2], 0, 0, 0);
              }
          }
          __builtin_amdgcn_sched_barrier(0);
          __syncthreads();
        };
        for (int kt = 0; kt < nk; kt += 6) {
          cstep8(std::integral_constant<int, 0>{}, kt);
          if (kt + 1 < nk) cstep8(std::integral_constant<int, 1>{}, kt + 1);
          if (kt + 2 < nk) cstep8(std::integral_constant<int, 2>{}, kt + 2);
          if (kt + 3 < nk) cstep8(std::integral_constant<int, 3>{}, kt + 3);
          if (kt + 4 < nk) cstep8(std::integral_constant<int, 4>{}, kt + 4);
          if (kt + 5 < nk) cstep8(std::integral_constant<int, 5>{}, kt + 5);
        }
        } else {
        Frags f0, f1;
        fetch(f0, 0);
        // iteration kt (position pos = kt % 6): fetch tile kt+1 from stage (pos+1) % 3 into the other register set, multiply tile kt
        auto cstep = [&](auto posc, const int kt_) {
          constexpr int pos = decltype(posc)::value;
          if (kt_ + 1 < nk) fetch((pos & 1) ? f0 : f1, (pos + 1) % 3);
          mult((pos & 1) ? f1 : f0);
          // the barrier stays BEHIND the MFMAs: hipcc otherwise hoists it (and the lgkmcnt(0) it needs) to right behind the first MFMA,
          // which puts the fragment reads' latency back on the matrix pipe's critical path
          __builtin_amdgcn_sched_barrier(0);
          __syncthreads();
        };
        for (int kt = 0; kt < nk; kt += 6) {
          cstep(std::integral_constant<int, 0>{}, kt);
          if (kt + 1 < nk) cstep(std::integral_constant<int, 1>{}, kt + 1);
          if (kt + 2 < nk) cstep(std::integral_constant<int, 2>{}, kt + 2);
          if (kt + 3 < nk) cstep(std::integral_constant<int, 3>{}, kt + 3);
          if (kt + 4 < nk) cstep(std::integral_constant<int, 4>{}, kt + 4);
          if (kt + 5 < nk) cstep(std::integral_constant<int, 5>{}, kt + 5);
        }
        }   // MW == 4
        if constexpr (PREC == PREC_F16X2 || PREC == PREC_F16) {
          const float ia = __uint_as_float((unsigned)(254 - seA) << 23), ib = __uint_as_float((unsigned)(254 - seB) << 23);
#pragma unroll
          for (int i = 0; i < NAM; ++i)
#pragma unroll
            for (int j = 0; j < NA; ++j)
#pragma unroll
              for (int r = 0; r < 16; ++r) acc[i][j][r] = acc[i][j][r] * ia * ib;
        }
        epilogue(acc);        // meanwhile the staging waves run the next tile's prologue
      }
    } else {
      f32x16 acc[NAM][NA];
#pragma unroll
      for (int i = 0; i < NAM; ++i)
#pragma unroll
        for (int j = 0; j < NA; ++j)
#pragma unroll
          for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
      // Pipeline: tile kt is multiplied out of LDS buffer kt&1 while tile kt+1 moves registers -> the other buffer and
      // tile kt+2 is in flight global -> registers; one barrier per k-iteration.
      sa.load(A, g.spanA, kend);
      sb.load(B, g.spanB, kend);
      sa.store(As[0]);
      sb.store(Bs[0]);
      if (kbeg + BK > kend) { sa.zero_tail(As[0], kbeg, kend); sb.zero_tail(Bs[0], kbeg, kend); }
      if (nk > 1) {
        sa.load(A, g.spanA, kend);
        sb.load(B, g.spanB, kend);
      }
      __syncthreads();

      // (two k-iterations per trip, so that the LDS buffer index is a compile-time constant and every LDS address an immediate offset)
      auto step = [&](auto curc, const int kt) {
        constexpr int cur = decltype(curc)::value;
        if (kt + 1 < nk) {
          sa.store(As[cur ^ 1]);
          sb.store(Bs[cur ^ 1]);
          if (kbeg + (kt + 2) * BK > kend) { sa.zero_tail(As[cur ^ 1], kbeg + (kt + 1) * BK, kend); sb.zero_tail(Bs[cur ^ 1], kbeg + (kt + 1) * BK, kend); }
        }
        if (kt + 2 < nk) {
          sa.load(A, g.spanA, kend);
          sb.load(B, g.spanB, kend);
        }
        // MFMA 32x32x2 operands: lane (li, lk) supplies row/col li of the 32-wide tile at k = KROW(j) + 2*lk, j = 0..BK/2-1.
        //   RK operand: those 8 values are 8 consecutive floats of its LDS row (two 16-byte reads for the whole k-iteration)
        //   KR operand: one ds_read2st64_b32 per k-pair (both 32-wide tiles), rolling one k-pair ahead of the MFMAs
        float fa[NA][BK / 2], fb[NA][BK / 2];
        const float* ap = As[cur] + (A_RK ? (wm * WT + li) * LDA + (BK / 2) * lk : 2 * lk * LDA + wm * 32 + li);
        const float* bp = Bs[cur] + (B_RK ? (wn * WT + li) * LDB + (BK / 2) * lk : 2 * lk * LDB + wn * 32 + li);
  #pragma unroll
        for (int i = 0; i < NA; ++i) {
          if (A_RK) {
  #pragma unroll
            for (int q = 0; q < BK / 8; ++q) *reinterpret_cast<float4*>(&fa[i][4 * q]) = *reinterpret_cast<const float4*>(ap + 32 * i * LDA + 4 * q);
          } else fa[i][0] = ap[64 * i];
          if (B_RK) {
  #pragma unroll
            for (int q = 0; q < BK / 8; ++q) *reinterpret_cast<float4*>(&fb[i][4 * q]) = *reinterpret_cast<const float4*>(bp + 32 * i * LDB + 4 * q);
          } else fb[i][0] = bp[64 * i];
        }
  #pragma unroll
        for (int j = 0; j < BK / 2; ++j) {
          if (j + 1 < BK / 2) {
  #pragma unroll
            for (int i = 0; i < NA; ++i) {
              if (!A_RK) fa[i][j + 1] = ap[KROW(j + 1) * LDA + 64 * i];
              if (!B_RK) fb[i][j + 1] = bp[KROW(j + 1) * LDB + 64 * i];
            }
          }
  #pragma unroll
          for (int i = 0; i < NA; ++i)
  #pragma unroll
            for (int i2 = 0; i2 < NA; ++i2) acc[i][i2] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[i][j], fb[i2][j], acc[i][i2], 0, 0, 0);
        }
        // pin the interleave of the scalar (KR) operand reads: hipcc otherwise sinks every LDS read directly in front of its MFMAs with
        // lgkmcnt(0).  One LDS read per KR operand and k-pair goes in front of the MFMAs of the previous k-pair.
        if (!A_RK || !B_RK) {
          constexpr int NR = (A_RK ? 0 : 1) + (B_RK ? 0 : 1);
          constexpr int NRK = (A_RK ? NA * (BK / 8) : 0) + (B_RK ? NA * (BK / 8) : 0);
          __builtin_amdgcn_sched_group_barrier(0x100, NR + NRK, 0);
  #pragma unroll
          for (int i = 0; i < BK / 2 - 1; ++i) {
            __builtin_amdgcn_sched_group_barrier(0x100, NR, 0);
            __builtin_amdgcn_sched_group_barrier(0x008, NA * NA, 0);
          }
          __builtin_amdgcn_sched_group_barrier(0x008, NA * NA, 0);
        }
        __syncthreads();
      };
      for (int kt = 0; kt < nk; kt += 2) {
        step(std::integral_constant<int, 0>{}, kt);
        if (kt + 1 < nk) step(std::integral_constant<int, 1>{}, kt + 1);
      }
      epilogue(acc);
    }   // !SPLIT
  }
  }   // phases
}

// ---- fp16x2: absolute maxima of the operands, one pass in front of the launch.  A region is `nb` slices (stride sb) of `rows` rows
// (stride ld) of `inner` contiguous floats; its maximum goes to *slot as (generation << 32 | float bits) with a 64-bit atomic max, so
// that a slot never has to be cleared: a later launch's generation outranks whatever an earlier user left there.
constexpr int AMAX_REGIONS = 4 * GEMM_GROUP_MAX;
constexpr int AMAX_SHARDS = 16;
struct AmaxRegion {
  const float* p;
  long nb, sb, rows, ld;
  int inner;
  unsigned long long* slot;
};
struct AmaxJobs {
  int n;
  unsigned gen;
  int blk_start[AMAX_REGIONS + 1];
  AmaxRegion r[AMAX_REGIONS];
};
__device__ __forceinline__ void absmax_block(const AmaxJobs& jobs) {
  int ri = 0;
  while ((int)blockIdx.x >= jobs.blk_start[ri + 1]) ++ri;
  const AmaxRegion& R = jobs.r[ri];
  const int nblk = jobs.blk_start[ri + 1] - jobs.blk_start[ri], blk = (int)blockIdx.x - jobs.blk_start[ri];
  const int quads = R.inner >> 2, tailn = R.inner & 3;
  int tq = 4;                                    // (at least 4: the up-to-3 elements behind the last whole quad are read by tx < tailn)
  while (tq < quads && tq < 256) tq <<= 1;       // threads along a row (power of two), 256 / tq rows per pass
  const int tx = threadIdx.x & (tq - 1), ty = threadIdx.x / tq, rpp = 256 / tq;
  const long total_rows = R.nb * R.rows;
  float m = 0.f;
  auto amax4 = [](float acc, const float4& v) { return fmaxf(fmaxf(acc, fmaxf(fabsf(v.x), fabsf(v.y))), fmaxf(fabsf(v.z), fabsf(v.w))); };
  // (no 64-bit division on the streaming path: it costs more than the loads)
  const bool one = R.nb == 1;
  const unsigned nrows32 = (unsigned)R.rows;
  auto rowptr = [&](long row) {
    if (one) return R.p + row * R.ld;
    if ((R.nb * R.rows) >> 32) return R.p + (row / R.rows) * R.sb + (row % R.rows) * R.ld;
    const unsigned sl = (unsigned)row / nrows32, rr = (unsigned)row - sl * nrows32;
    return R.p + (long)sl * R.sb + (long)rr * R.ld;
  };
  const long step = (long)nblk * rpp;
  long row = (long)blk * rpp + ty;
  if (quads <= tq) {
    // short rows (one quad per thread): eight rows in flight per thread; a row index past the end re-reads the last row (harmless
    // for a maximum), so that the remainder costs no extra round trip
    const bool has = tx < quads;
    for (; row < total_rows; row += 8 * step) {
      float4 v[8];
      float t[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const float* q = rowptr(min(row + j * step, total_rows - 1));
        v[j] = has ? *reinterpret_cast<const float4*>(q + 4 * tx) : make_float4(0.f, 0.f, 0.f, 0.f);
        t[j] = tx < tailn ? q[4 * quads + tx] : 0.f;
      }
#pragma unroll
      for (int j = 0; j < 8; ++j) m = fmaxf(amax4(m, v[j]), fabsf(t[j]));
    }
  } else {
    for (; row < total_rows; row += step) {
      const float* q = rowptr(row);
      int c = tx;
      for (; c + 3 * tq < quads; c += 4 * tq) {        // long rows: four quads in flight per thread
        const float4 v0 = *reinterpret_cast<const float4*>(q + 4 * c), v1 = *reinterpret_cast<const float4*>(q + 4 * (c + tq));
        const float4 v2 = *reinterpret_cast<const float4*>(q + 4 * (c + 2 * tq)), v3 = *reinterpret_cast<const float4*>(q + 4 * (c + 3 * tq));
        m = amax4(amax4(amax4(amax4(m, v0), v1), v2), v3);
      }
      for (; c < quads; c += tq) m = amax4(m, *reinterpret_cast<const float4*>(q + 4 * c));
      if (tx < tailn) m = fmaxf(m, fabsf(q[4 * quads + tx]));
    }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
  __shared__ float wm[4];
  if ((threadIdx.x & 63) == 0) wm[threadIdx.x >> 6] = m;
  __syncthreads();
  if (threadIdx.x == 0) {
    m = fmaxf(fmaxf(wm[0], wm[1]), fmaxf(wm[2], wm[3]));
    if (!(m <= 3.0e38f)) m = 3.0e38f;            // (an infinite entry: the product is not finite either way)
    // (same-address atomics retire one after the other, ~40 ns each: 16 shards per region, the reader takes the maximum of all)
    atomicMax(amax_shard(R.slot, blk), ((unsigned long long)jobs.gen << 32) | (unsigned long long)__float_as_uint(m));
  }
}
__global__ __launch_bounds__(256) void k_absmax(AmaxJobs jobs) { absmax_block(jobs); }
// The ring: two halves of 512 slots of 16 ADJACENT shards (one 128-byte line: the consuming GEMM reads a slot with 16 scalar loads at
// the start of every tile, and 16 separate lines there cost 0.18 ms per train step -- measured A/B in one call; the producing side's
// same-line atomics cost less than that.  Kernels with thousands of emitting blocks use strided producer slots and have them
// compacted into one line by a small kernel that runs in between anyway: common.h amax_compact.)
constexpr int AMAX_RING_SLOTS = 1024;
constexpr int AMAX_SLOTS = AMAX_RING_SLOTS * AMAX_SLOT_WORDS;      // 64-bit words (128 KB)
__device__ unsigned long long g_amax_ring[AMAX_SLOTS];

// GEMM_STORE + split tiles: zero the tiles that more than one workgroup will accumulate into.  Block b looks at the
// boundary between workgroups b and b+1.
// (16-byte stores when every problem's C rows are 16-byte aligned -- `vec`, decided by the launcher -- else scalar)
// (round 5: a tile several boundaries fall into -- the remainder tiles of a hybrid launch are shared by G / tiles workgroups each -- is
//  zeroed by the block of its FIRST boundary only, and ZERO_PARTS blocks share a tile's rows: 67 -> ~30 us per train step at 256 x 128 tiles)
constexpr int ZERO_PARTS = 4;
__device__ __forceinline__ void zero_split_block(const GemmGroup& grp, int G, int TLM, int TL, int vec, unsigned bb) {
  const unsigned b = bb / ZERO_PARTS, part = bb % ZERO_PARTS;
  const long git = wg_first_iter(grp, b + 1, G);
  int prob = 0;
  while (git >= grp.iter_start[prob + 1]) ++prob;
  const GemmArgs& g = grp.g[prob];
  if (g.mode != GEMM_STORE) return;
  const long it = git - grp.iter_start[prob];
  if (it % g.kt == 0) return;
  if (b > 0) {      // the previous boundary inside the same tile: its block zeroes the tile
    const long pit = wg_first_iter(grp, b, G) - grp.iter_start[prob];
    if (pit >= 0 && pit / g.kt == it / g.kt && pit % g.kt != 0) return;
  }
  const long tile = it / g.kt;
  int zb, m0, n0;
  decode_tile(g, tile, TLM, TL, zb, m0, n0);
  float* C = g.C + (long)zb * g.sC;
  const int w = vec ? 4 : 1;                    // floats per thread and row
  const int tpr = TL / w;                        // threads per row
  const int col = n0 + (threadIdx.x % tpr) * w;
  if (col >= g.N) return;
  const int rows_part = TLM / ZERO_PARTS;
  for (int r = part * rows_part + threadIdx.x / tpr; r < (int)(part + 1) * rows_part; r += 256 / tpr) {
    const int row = m0 + r;
    if (row >= g.M) break;
    const long coff = g.c_tn > 0 ? (long)(row / g.c_tn) * g.c_sg + (long)(row % g.c_tn) * g.c_st : (long)row * g.ldc;
    if (vec && col + 3 < g.N) *reinterpret_cast<float4*>(C + coff + col) = make_float4(0.f, 0.f, 0.f, 0.f);
    else
      for (int c = col; c < min(col + w, g.N); ++c) C[coff + c] = 0.f;
  }
}
__global__ __launch_bounds__(256) void k_zero_split_tiles(GemmGroup grp, int G, int TLM, int TL, int vec) {
  zero_split_block(grp, G, TLM, TL, vec, blockIdx.x);
}
// Both preparations of a fp16x2 GEMM_STORE stream-K launch in ONE launch (a dependent launch costs ~10 us of stream time whatever it
// does): blocks [0, nb) take the operands' absolute maxima, the G - 1 blocks behind them zero the split tiles.  (6 KB of arguments.)
__global__ __launch_bounds__(256) void k_absmax_zero(AmaxJobs jobs, GemmGroup grp, int G, int TLM, int TL, int vec) {
  const unsigned nb = (unsigned)jobs.blk_start[jobs.n];
  if (blockIdx.x < nb) absmax_block(jobs);
  else zero_split_block(grp, G, TLM, TL, vec, blockIdx.x - nb);
}

}  // namespace

// The ticket words of a stream's launches (zero between launches: the kernels leave them so).  One table per stream -- launches on one
// stream run one after the other; it grows by being replaced behind a stream synchronisation.
static unsigned* tick_table(hipStream_t s, size_t words) {
  struct Table { hipStream_t s; unsigned* p; size_t words; };
  static std::mutex mu;
  static std::vector<Table> tables;
  std::lock_guard<std::mutex> lock(mu);
  Table* t = nullptr;
  for (auto& e : tables)
    if (e.s == s) t = &e;
  if (!t) { tables.push_back(Table{s, nullptr, 0}); t = &tables.back(); }
  if (t->words < words) {
    if (t->p) {
      if (hipStreamSynchronize(s) != hipSuccess) return nullptr;
      (void)hipFree(t->p);
      t->p = nullptr; t->words = 0;
    }
    const size_t n = std::max(words, (size_t)1 << 18);
    if (hipMalloc((void**)&t->p, n * sizeof(unsigned)) != hipSuccess) return nullptr;
    if (hipMemset(t->p, 0, n * sizeof(unsigned)) != hipSuccess) return nullptr;
    t->words = n;
  }
  return t->p;
}

static int gemm_prepare(int layout, const GemmArgs& g, GemmArgs& a, bool& twolvl, int TLM, int TL) {
  ASTK_CHECK(g.A.p && g.B.p && g.C, "gemm: null operand");
  ASTK_CHECK(aligned16(g.A.p) && aligned16(g.B.p), "gemm: A/B must be 16-byte aligned");
  ASTK_CHECK((g.A.ld % 4) == 0 && (g.B.ld % 4) == 0 && (g.A.sg % 4) == 0 && (g.A.st % 4) == 0 &&
                 (g.B.sg % 4) == 0 && (g.B.st % 4) == 0 && (g.sA % 4) == 0 && (g.sB % 4) == 0,
             "gemm: leading dimensions / strides must be multiples of 4 floats (lda=%ld ldb=%ld)", g.A.ld, g.B.ld);
  ASTK_CHECK(g.ksplit >= 1 && (g.ksplit == 1 || g.mode == GEMM_ATOMIC), "gemm: split-K needs atomic mode");
  a = g;
  const bool a_kr = layout == GEMM_TN, b_kr = layout != GEMM_NT;
  // the k loop addresses an operand as (uniform pointer) + (32-bit byte offset): each batch slice has to span less than 4 GiB
  auto span = [&](const MatView& v, bool kr, int rows) -> long {   // floats from v.p to the end of the last element touched
    const long k4 = (g.K + 3) & ~3L, r4 = (rows + 3) & ~3L;      // padded to the 16-byte granule (leading dimensions are multiples of 4)
    if (kr) return v.tn > 0 ? 0 : (long)(g.K - 1) * v.ld + r4;   // two-level K-major rows use 64-bit offsets
    if (v.rowidx) return (v.idx_rows > 0 ? v.idx_rows : (long)rows) * v.ld;      // indexed rows: a permutation of [0, rows), or any rows of an array of idx_rows rows
    if (v.tn > 0) return (long)((rows - 1) / v.tn) * v.sg + (long)(v.tn - 1) * v.st + k4;
    return (long)(rows - 1) * v.ld + k4;
  };
  const long spa = span(g.A, a_kr, g.M), spb = span(g.B, b_kr, g.N);
  ASTK_CHECK(spa < (1L << 30) && spb < (1L << 30), "gemm: an operand slice spans more than 4 GiB (M=%d N=%d K=%d lda=%ld ldb=%ld)",
             g.M, g.N, g.K, g.A.ld, g.B.ld);
  ASTK_CHECK(!(a_kr && g.A.tn > 0 && (long)BK * g.A.st + g.M >= (1L << 29)) && !(b_kr && g.B.tn > 0 && (long)BK * g.B.st + g.N >= (1L << 29)),
             "gemm: two-level row stride too large");
  a.spanA = (unsigned)(spa * 4);
  a.spanB = (unsigned)(spb * 4);
  ASTK_CHECK(!(a_kr && g.A.rowidx) && !(b_kr && g.B.rowidx), "gemm: indexed rows are only supported on K-contiguous operands");
  twolvl = (a_kr && g.A.tn > 0) || (b_kr && g.B.tn > 0);
  a.tiles_mn = cdiv(g.M, TLM) * cdiv(g.N, TL);
  a.kt = cdiv(g.K, BK);
  a.iters_total = (long)a.tiles_mn * g.batch * a.kt;
  return 0;
}

static int g_lowp_mode = 0;
static std::atomic<double> g_small_flops{3e9};
// Per-call arithmetic (astk_*_desc.precision / .gemm_operands, astk_gemm_f32_ex): the entry points open a PrecScope, the launches of
// this thread inside it use the descriptor's choice; outside a scope (or with ASTK_PREC_DEFAULT) the process-wide default applies.
static thread_local int tl_prec = -1;       // GemmPrec, or -1 = process default
static thread_local int tl_lowp = -1;       // 0 / 1, or -1 = process default
int low_precision_gemms() { return tl_lowp >= 0 ? tl_lowp : g_lowp_mode; }
static thread_local int tl_forward = 0;
GemmForwardScope::GemmForwardScope() : prev(tl_forward) { tl_forward = 1; }
GemmForwardScope::~GemmForwardScope() { tl_forward = prev; }
static thread_local int tl_det = 0;
DetScope::DetScope(int requested) : prev(tl_det) { tl_det = (requested != 0 || tune_on(TUNE_GEMM_DETERMINISTIC)) ? 1 : 0; }
DetScope::~DetScope() { tl_det = prev; }
bool deterministic_mode() { return tl_det != 0; }
static thread_local int tl_wg_cap = 0;
GemmWgCap::GemmWgCap(int wgs) : prev(tl_wg_cap) { tl_wg_cap = wgs > 0 ? std::max(8, wgs / 8 * 8) : 0; }      // (a multiple of the 8 XCDs)
GemmWgCap::~GemmWgCap() { tl_wg_cap = prev; }

// operand precision of the f32-accurate products.  Process default: bf16x3 (three-term bf16 split: every f32 operand value is represented
// EXACTLY, f32 exponent range, products exact to 2^-26) -- the reference computes in float32 (seq2seq.py:154,302,420), and this is the
// fastest scheme here that is at least as accurate as an f32 fma chain on ANY data.  fp16x2 (two scaled fp16 terms: 22 bits per operand
// value, and only for values within 2^-17 of the operand's maximum) is faster but narrower: opt-in only (astk_set_gemm_precision(0)
// or a descriptor's precision field); f32 = the exact f32 MFMA chain.
static std::atomic<int> g_prec{PREC_BF16X3};
static int default_prec() { return tl_prec >= 0 ? tl_prec : g_prec.load(std::memory_order_relaxed); }
PrecScope::PrecScope(int precision, int operands) : prev_p(tl_prec), prev_l(tl_lowp) {
  if (precision == ASTK_PREC_FP16X2) tl_prec = PREC_F16X2;
  else if (precision == ASTK_PREC_BF16X3) tl_prec = PREC_BF16X3;
  else if (precision == ASTK_PREC_F32) tl_prec = PREC_F32;
  if (operands == ASTK_OPERANDS_F32) tl_lowp = 0;
  else if (operands == ASTK_OPERANDS_FP16) tl_lowp = 1;
}
PrecScope::~PrecScope() { tl_prec = prev_p; tl_lowp = prev_l; }
int gemm_precision_mode() { const int p = default_prec(); return p == PREC_F32 ? 2 : (p == PREC_BF16X3 ? 1 : 0); }
static unsigned long long* amax_ring() {
  static unsigned long long* ring = nullptr;
  if (!ring && hipGetSymbolAddress((void**)&ring, HIP_SYMBOL(g_amax_ring)) != hipSuccess) ring = nullptr;
  return ring;
}
static std::atomic<unsigned> g_amax_counter{1};
// The generation tag of an absolute-maximum pass (high half of the 64-bit atomicMax words: a slot never has to be cleared, a word of an
// older pass loses to any word of a newer one).  The tag is 32 bits wide: at ~40 passes per train step it would wrap after about a
// week of training in one process, and from then on every new word would LOSE to the stale ones (frozen scales: overflow to inf or
// silently lost bits).  So the counter never wraps: when it comes within reach of the end, the whole ring is zeroed ON THE LAUNCH
// STREAM (ordered behind every launch that still reads an old word, in front of every pass that writes a new one) and counting
// restarts at 1.  Like the ring itself this relies on all GEMM launches of a process being ordered on one stream at a time (the
// opt-in side-stream overlap of ast_amd/seq2seq.py keeps its launches on bf16x3 operands, which use no slots: GemmWgCap).
// The restart zeroes the callers' handles too (gemm_amax / gemm_amax_reserve: their words carry old, large generations that would
// outrank every new one), so it must not fall between a caller taking a handle and the launches that read it: ops that hold handles
// across launches call gemm_amax_step_boundary() where none is live (the first op of a train step), which restarts EARLY, at
// AMAX_GEN_SOFT; the hard limit inside next_amax_gen only serves processes that never reach such a boundary (plain astk_gemm_f32
// users, which hold no handles).
constexpr unsigned AMAX_GEN_WRAP = 0xFFFFFF00u;
constexpr unsigned AMAX_GEN_SOFT = 0xFFF00000u;       // ~1M generations (tens of thousands of steps) in front of the hard limit
static std::mutex g_amax_wrap_mutex;
static void amax_restart_locked(hipStream_t s) {
  unsigned long long* ring = nullptr;
  if (hipGetSymbolAddress((void**)&ring, HIP_SYMBOL(g_amax_ring)) == hipSuccess && ring) (void)hipMemsetAsync(ring, 0, sizeof(unsigned long long) * AMAX_SLOTS, s);
  g_amax_counter.store(2);
}
void gemm_amax_step_boundary(hipStream_t s) {
  if (g_amax_counter.load(std::memory_order_relaxed) < AMAX_GEN_SOFT) return;
  std::lock_guard<std::mutex> lock(g_amax_wrap_mutex);
  if (g_amax_counter.load() >= AMAX_GEN_SOFT) amax_restart_locked(s);
}
static unsigned next_amax_gen(hipStream_t s) {
  unsigned gen = g_amax_counter.fetch_add(1);
  if (gen < AMAX_GEN_WRAP) return gen;
  std::lock_guard<std::mutex> lock(g_amax_wrap_mutex);
  if (g_amax_counter.load() >= AMAX_GEN_WRAP) {      // first thread to get here restarts the count
    amax_restart_locked(s);
    return 1;
  }
  return g_amax_counter.fetch_add(1);
}
// slots [0, AMAX_SLOTS / 2) serve the launches' own passes, [AMAX_SLOTS / 2, AMAX_SLOTS) the callers' handles (gemm_amax)
static std::atomic<unsigned> g_amax_handle{0};

// Adds the region(s) of one operand: exact rows x inner when the rows are plain (r * ld), the contiguous span from p to the slice's end
// when they are two-level windows or indexed (every byte of the span is activation data or zero padding of the same array; the
// maximum of a superset can only make the scale more conservative).
static void amax_add(AmaxJobs& J, const MatView& v, bool kr, int rows, int K, int batch, long sbatch, long span_floats, unsigned long long* slot) {
  auto push = [&](const float* p, long nb, long sb, long nrows, long ld, int inner) {
    if (inner <= 0 || nrows <= 0 || J.n >= AMAX_REGIONS) return;
    AmaxRegion& R = J.r[J.n];
    R.p = p; R.nb = nb; R.sb = sb; R.rows = nrows; R.ld = ld; R.inner = inner; R.slot = slot;
    const double bytes = 4.0 * nb * nrows * inner;
    const int blocks = (int)std::min(1024.0, std::max(1.0, bytes / (32.0 * 1024.0)));
    J.blk_start[J.n + 1] = J.blk_start[J.n] + blocks;
    ++J.n;
  };
  const long outer = kr ? K : rows;            // rows of the global array the operand's rows map to
  const int inner = kr ? rows : K;
  if (v.tn > 0 || v.rowidx) {
    const long n = span_floats;                // contiguous: whole 1024-float rows and a tail row
    push(v.p, batch, sbatch, n / 1024, 1024, 1024);
    push(v.p + (n / 1024) * 1024, batch, sbatch, 1, 0, (int)(n % 1024));
  } else if (v.ld == inner && batch == 1) {
    const long n = outer * (long)inner;
    push(v.p, 1, 0, n / 1024, 1024, 1024);
    push(v.p + (n / 1024) * 1024, 1, 0, 1, 0, (int)(n % 1024));
  } else {
    push(v.p, batch, sbatch, outer, v.ld, inner);
  }
}

int gemm_launch_group(int layout, const GemmArgs* list, int n, hipStream_t s) {
  ASTK_CHECK(layout == GEMM_NT || layout == GEMM_NN || layout == GEMM_TN, "gemm: bad layout %d", layout);
  ASTK_CHECK(n >= 0 && n <= GEMM_GROUP_MAX, "gemm: group of %d products (max %d)", n, GEMM_GROUP_MAX);
  GemmGroup grp;
  bool twolvl = false, any_store = false;
  long tiles = 0;
  double flops = 0;
  int min_kt = 0x7fffffff;
  // Tile edge: 128 unless the whole launch is too small to keep the chip busy with 128-tiles (fewer k-iterations than the stream-K
  // threshold below, and either too few tiles for the CUs or tiles so shallow that their ramp dominates): then 64-tiles give 4x the
  // workgroups, each a quarter of the work.  astk_set_tuning("gemm.tile", 64 | 128 | 256) forces one.
  const int force_tl = (int)tune(TUNE_GEMM_TILE);
  // operand scheme of the launch (default_prec: fp16x2 unless the environment asks for bf16x3 or f32)
  int prec = default_prec();
  if (low_precision_gemms() != 0 && n > 0) {        // fp16 operands only if every product of the launch is marked eligible by its caller
    bool all = true;
    for (int i = 0; i < n; ++i) all = all && list[i].lowp != 0;
    if (all) prec = PREC_F16;
  }
  // launches capped to share the CUs with a recurrence kernel are the ones a caller puts on a SIDE stream: the scale slots of the
  // fp16x2 scheme are a process-wide ring ordered by one stream, so these launches take the scheme that needs none
  if (prec == PREC_F16X2 && tl_wg_cap > 0) prec = PREC_BF16X3;
  // deterministic calls: split launches run on the 128 x 128 bf16x3 kernel, the one the fix-up epilogue is instantiated for (fp16x2 asks
  // for less accuracy than that, so it may have it); the f32 and single-term-fp16 schemes keep their kernels with one whole tile per workgroup
  const bool det = tl_det != 0;
  if (det && prec == PREC_F16X2) prec = PREC_BF16X3;
  int TL = 128, TLM = 128;
  for (int pass = 0; pass < 3; ++pass) {
    memset(&grp, 0, sizeof(grp));
    twolvl = any_store = false;
    tiles = 0; flops = 0; min_kt = 0x7fffffff;
    bool tall_ok = true;       // 256-row tiles waste at most 10 % more rows than 128-row tiles on every problem
    for (int i = 0; i < n; ++i) {
      const GemmArgs& g = list[i];
      if (g.M <= 0 || g.N <= 0 || g.K <= 0 || g.batch <= 0) continue;
      GemmArgs& a = grp.g[grp.n];
      bool tl = false;
      ASTK_TRY(gemm_prepare(layout, g, a, tl, TLM, TL));
      twolvl = twolvl || tl;
      grp.iter_start[grp.n] = grp.iters_total;
      grp.iters_total += a.iters_total;
      tiles += (long)a.tiles_mn * g.batch;
      flops += 2.0 * g.M * g.N * (double)g.K * g.batch;
      any_store = any_store || g.mode == GEMM_STORE;
      min_kt = std::min(min_kt, a.kt);
      tall_ok = tall_ok && cdiv(g.M, 256) * 256 * 10 <= cdiv(g.M, 128) * 128 * 11;
      ++grp.n;
    }
    if (pass > 0) break;        // pass 0 sizes the launch with 128-tiles and picks; pass 1 re-plans with the chosen tile
    // (measured, scratch/gemm_bench.py: 40 tiles x 32 k-iterations 33 -> 18 us, 1 tile x 4800 57 -> 38 us, 600 tiles x 8 42 -> 38 us;
    //  200 tiles x 32 is 5 % faster with 128-tiles)
    int max_kt = 0;
    for (int i = 0; i < grp.n; ++i) max_kt = std::max(max_kt, grp.g[i].kt);
    bool small = grp.iters_total < 256L * wgs_per_cu(128) * 10 * (32 / BK) && (tiles < 192 || max_kt <= 12 * (16 / BK > 0 ? 16 / BK : 1));
    // (a capped launch -- work beside a recurrence kernel on a side stream -- is "small" only if its 128-tiles cannot fill the CAPPED grid:
    //  a chunk of the layer-0 projection is 64 tiles for 64 workgroups, one whole tile each; as 64-tiles it ran 240 us instead of 140)
    if (tl_wg_cap > 0) small = small && 2 * tiles < tl_wg_cap;
    // 256 x 128 tiles on the 12-wave kernel (round 5): launches of 20 GFLOP and more whose M extents waste at most 10 % more rows than
    // with 128-row tiles -- measured per launch of the train step (scratch/r5_gemm_t256.sh): 3-6 % faster on every launch above 200 us,
    // slower on the small ones (6400 x 512 x 512: 40 -> 58 us)
    const double big_flops = tune(TUNE_GEMM_T256_ABOVE);
    const bool big = prec == PREC_BF16X3 && BK == 16 && tall_ok && flops >= big_flops && !small;
    int want = force_tl == 64 || force_tl == 128 || force_tl == 256 ? force_tl : (small ? 64 : (big ? 256 : 128));
    if (det && prec == PREC_BF16X3) want = 128;
    if (want == 256 && (prec != PREC_BF16X3 || BK != 16)) want = 128;
    if (want == 64 && prec == PREC_F16) want = 128;        // (the fp16 variant is instantiated for 128-tiles only)
    if (want == 128) break;
    TLM = want;
    TL = want == 256 ? 128 : want;
  }
  // A small launch does not repay an absolute-maximum pass (a launch of its own, 6 us at least): three-term bf16 operands need no scales
  if (prec == PREC_F16X2 && flops < g_small_flops.load()) {
    bool given = true;      // ... unless the caller supplied every maximum already
    for (int i = 0; i < grp.n; ++i) given = given && grp.g[i].amaxA && grp.g[i].amaxB;
    if (!given) prec = PREC_BF16X3;
  }
  const int WGS_PER_CU = wgs_per_cu(TL, prec);
  if (grp.n == 0) return 0;
  for (int i = grp.n; i <= GEMM_GROUP_MAX; ++i) grp.iter_start[i] = grp.iters_total;
  grp.unit = prec != PREC_F32 ? 2 : 1;
  for (int i = 0; i < grp.n; ++i)
    if (grp.g[i].kt & 1) grp.unit = 1;
  if (twolvl) {   // a plain operand next to a two-level one: express it as one group of INT_MAX rows
    const bool a_kr = layout == GEMM_TN, b_kr = layout != GEMM_NT;
    for (int i = 0; i < grp.n; ++i) {
      GemmArgs& a = grp.g[i];
      if (a_kr && a.A.tn <= 0) { a.A.tn = 0x7fffffff; a.A.sg = 0; a.A.st = a.A.ld; }
      if (b_kr && a.B.tn <= 0) { a.B.tn = 0x7fffffff; a.B.sg = 0; a.B.st = a.B.ld; }
    }
  }
  // Grid (measured on MI355X, scratch/gemm_bench.py): a launch with enough k-iterations is split evenly over 256 * wgs_per_cu
  // workgroups (co-resident workgroups hide each other's LDS-store / barrier phases; the grid must be a multiple of 256 or the
  // CUs are loaded unevenly: 600 workgroups for 1200 tiles ran 12 % slower than 768) -- 100-125 TFLOP/s on the train step's big
  // shapes against 60-93 for one-tile-per-workgroup launches.  Smaller products keep one tile per workgroup unless they have too
  // few tiles to occupy the chip; then their k range is split as well.
  const int force_g = (int)tune(TUNE_GEMM_GRID);   // tuning hook
  // (forward ops open a GemmForwardScope; "gemm.forward_pairs" 2 applies their two-contributor rule to EVERY launch of the process -- plain
  //  astk_gemm_f32 calls included: how the tests reach it on the product library with shapes no op produces)
  const bool forward_rule = tl_forward != 0 || tune(TUNE_GEMM_FORWARD_PAIRS) >= 2;
  long G = tiles;
  bool aligned = true;   // workgroup boundaries fall on tile boundaries
  if (grp.iters_total >= 256L * WGS_PER_CU * 10 * (32 / BK)) { G = 256 * WGS_PER_CU; aligned = false; }
  else if (tiles < 160) {
    const long g2 = std::min(256L, grp.iters_total / (4 * (32 / BK)));
    if (g2 > tiles) { G = g2; aligned = false; }
  }
  if (force_g > 0) { G = std::min<long>(force_g, grp.iters_total); aligned = false; }
  if (tl_wg_cap > 0 && G > tl_wg_cap) { G = tl_wg_cap; aligned = false; }
  if (det && prec != PREC_BF16X3) { G = tiles; aligned = true; }        // no fix-up kernel for these schemes: whole tiles only
  if (det && !aligned && G > FIX_WGS) G = FIX_WGS;
  if (aligned && grp.n > 1) {
    // one tile per workgroup needs boundaries on tile boundaries: only true for uniform kt; otherwise fall back to an even split
    bool uniform = true;
    for (int i = 1; i < grp.n; ++i) uniform = uniform && grp.g[i].kt == grp.g[0].kt;
    if (!uniform) { G = std::min(256L * WGS_PER_CU, std::max(1L, grp.iters_total / std::max(1, min_kt))); aligned = false; }
  }
  // Hybrid schedule (kernel: "Hybrid schedule"): a stream-K launch of the split schemes whose tiles all have the same depth and outnumber
  // the grid runs floor(tiles / G) data-parallel waves of whole tiles in XCD-local blocks first; only the rest is split stream-K.
  // ("gemm.hybrid" 0: everything stream-K in row-major tile order, the schedule of rounds 1-4.)
  const bool hybrid_on = tune_on(TUNE_GEMM_HYBRID);
  grp.dp_waves = 0; grp.dp_kt = 0; grp.rem_start = 0;
  if (hybrid_on && prec != PREC_F32 && !aligned && (G % 8) == 0 && tiles >= G) {
    bool uniform = true;
    for (int i = 1; i < grp.n; ++i) uniform = uniform && grp.g[i].kt == grp.g[0].kt;
    // (C += AB goes out as atomic adds: in data-parallel waves all workgroups reach their epilogues together and the atomics arrive in
    //  bursts -- 6400 x 3072 x 1024: 247 -> 271 us; those launches stay all stream-K, where the epilogues are staggered)
    for (int i = 0; i < grp.n; ++i) uniform = uniform && grp.g[i].mode != GEMM_ACCUM;
    if (uniform) {
      grp.dp_waves = (int)(tiles / G);
      // (forward launches: one wave fewer -- the remainder then has >= G tiles, every stream-K range >= one tile, <= 2 contributors per split
      //  tile: run-to-run reproducible sums, see GemmForwardScope)
      if (forward_rule && grp.dp_waves > 0) --grp.dp_waves;
      grp.dp_kt = grp.g[0].kt;
      grp.rem_start = (long)grp.dp_waves * G * grp.dp_kt;
      const int chunk = (int)(G / 8);          // tiles of one XCD and wave
      for (int i = 0; i < grp.n; ++i) {
        GemmArgs& a = grp.g[i];
        const int tiles_n = cdiv(a.N, TL);
        int cols = 1;                          // block width: a power of two <= tiles_n, near sqrt(chunk) from above (rows + cols minimal)
        while (cols * 2 <= tiles_n && cols * 2 * cols * 2 <= 2 * chunk) cols *= 2;
        a.bm = std::max(1, chunk / cols);
      }
    }
  }
  // Forward launches outside the hybrid branch (fewer tiles than workgroups, mixed depths in a group, a capped or odd grid): the same
  // guarantee -- at most TWO contributors per split tile, so that the float atomics commute and the forward pass is bit-reproducible whatever
  // shapes occur -- by shrinking the grid until every stream-K range is at least as deep as the deepest tile, or to exactly two half-tile
  // ranges per tile where the depths allow it (round-5 advice: the guarantee used to depend on which shapes happened to come by).
  if (forward_rule && !aligned && grp.dp_waves == 0 && tune_on(TUNE_GEMM_FORWARD_PAIRS)) {
    int max_kt = 0;
    bool uniform = true;
    for (int i = 0; i < grp.n; ++i) { max_kt = std::max(max_kt, grp.g[i].kt); uniform = uniform && grp.g[i].kt == grp.g[0].kt; }
    const long gmax = std::max(1L, grp.iters_total / std::max(1, max_kt));
    if (G > gmax) G = (uniform && max_kt % (2 * grp.unit) == 0 && G >= 2 * tiles) ? 2 * tiles : gmax;
  }
  // Few tiles, deep K, accumulating output (the weight gradients: TN 512 x 1152 x 38400 is 18 tiles of 2400 k-iterations): chunk-major order,
  // chunk = the divisor of kt nearest to a workgroup's share.  ("gemm.chunk" 0: tile-major always.)
  const bool chunk_on = tune_on(TUNE_GEMM_CHUNK);
  const int chunk_div = (int)tune(TUNE_GEMM_CHUNK_DIV);      // (tuning hook: tiles * div <= G)
  for (int i = 0; i < grp.n; ++i) { grp.g[i].cs = 0; grp.g[i].chunk_iters = 0; }
  if (chunk_on && !det && grp.n == 1 && !aligned && grp.dp_waves == 0 && grp.g[0].mode != GEMM_STORE && prec != PREC_F32 && tiles * chunk_div <= G && grp.g[0].kt >= 128) {
    GemmArgs& a = grp.g[0];
    const long share = grp.iters_total / G;      // k-iterations per workgroup
    int best = 0;
    for (int c = 2; c <= a.kt; c += 2)
      if (a.kt % c == 0 && c >= share / 2 && c <= share * 2 && (best == 0 || labs(c - share) < labs(best - share))) best = c;
    if (best > 0 && best < a.kt) { a.cs = best; a.chunk_iters = tiles * (long)best; }
  }
  const bool log_shapes = tune_on(TUNE_GEMM_LOG);
  if (log_shapes)
    for (int i = 0; i < grp.n; ++i)
      fprintf(stderr, "astk_gemm layout=%d M=%d N=%d K=%d batch=%d mode=%d twolvl=%d group=%d/%d G=%ld kt=%d tile=%d dp_waves=%d bm=%d cs=%d\n", layout, grp.g[i].M,
              grp.g[i].N, grp.g[i].K, grp.g[i].batch, grp.g[i].mode, (int)twolvl, i, grp.n, G, grp.g[i].kt, TLM * 1000 + TL, grp.dp_waves, grp.g[i].bm, grp.g[i].cs);
  ProfScope prof(PROF_GEMM, s, flops);
  if (prof_enabled()) {       // algorithmic bytes of the launch: every operand element read once, every result element written once
    double bytes = 0;
    for (int i = 0; i < grp.n; ++i) {
      const GemmArgs& a = grp.g[i];
      bytes += 4.0 * a.batch * ((double)a.M * a.K + (double)a.N * a.K + (double)a.M * a.N * (a.mode == GEMM_STORE ? 1 : 2));
    }
    prof_add_bytes(PROF_GEMM, bytes);
  }
  dim3 grid((unsigned)G, 1, 1);
  // GEMM_STORE + split tiles: the tiles several workgroups accumulate into are zeroed first (with the maximum pass when there is one).
  // (Builds with -DASTK_GEMM_TICKET=1: the ticket protocol instead -- kernel: "Split tiles without a zeroing launch", measured, off.)
  const bool ticket_on = ASTK_GEMM_TICKET && tune_on(TUNE_GEMM_TICKET);
  grp.tick = nullptr;
  const bool fixup = det && !aligned && prec == PREC_BF16X3 && TL == 128 && TLM == 128;
  bool need_zero = !aligned && any_store && G > 1 && !fixup;
  if (need_zero && ticket_on) {
    long ntiles = 0;
    bool fits = true;
    for (int i = 0; i < grp.n; ++i) {
      grp.tick_base[i] = (int)ntiles;
      ntiles += (grp.iter_start[i + 1] - grp.iter_start[i]) / grp.g[i].kt;
      fits = fits && grp.g[i].kt < (int)TICK_DONE;
    }
    if (fits && ntiles < (1L << 27)) {
      grp.tick = tick_table(s, (size_t)ntiles * TICK_WAVES);
      ASTK_CHECK(grp.tick != nullptr, "gemm: no ticket table");
      need_zero = false;
    }
  }
  bool zero_vec = true;
  for (int i = 0; i < grp.n; ++i) {
    const GemmArgs& a = grp.g[i];
    zero_vec = zero_vec && aligned16(a.C) && (a.sC % 4) == 0 && (a.c_tn > 0 ? (a.c_sg % 4) == 0 && (a.c_st % 4) == 0 : (a.ldc % 4) == 0);
  }
  int amax_blocks = 0;
  if (prec == PREC_F16X2) {
    // one absolute-maximum pass over both operands of every product (part of the GEMM's cost, inside its timing scope)
    unsigned long long* ring = amax_ring();
    ASTK_CHECK(ring != nullptr, "gemm: no absmax ring");
    AmaxJobs J;
    memset(&J, 0, sizeof(J));
    const unsigned gen = next_amax_gen(s);
    J.gen = gen;
    const bool a_kr = layout == GEMM_TN, b_kr = layout != GEMM_NT;
    auto list_view = [](MatView v) { if (v.tn == 0x7fffffff) v.tn = 0; return v; };    // (a plain operand dressed as one group, see above)
    for (int i = 0; i < grp.n; ++i) {
      GemmArgs& a = grp.g[i];
      unsigned long long* sl = ring + (((size_t)gen * (2 * GEMM_GROUP_MAX) + 2 * i) % (AMAX_RING_SLOTS / 2)) * AMAX_SLOT_WORDS;
      unsigned long long* slB = sl + AMAX_SLOT_WORDS;
      const long spa = a.A.tn > 0 && a.A.tn != 0x7fffffff && a_kr ? (long)((a.K - 1) / a.A.tn) * a.A.sg + (long)((a.K - 1) % a.A.tn) * a.A.st + ((a.M + 3) & ~3L) : a.spanA / 4;
      const long spb = a.B.tn > 0 && a.B.tn != 0x7fffffff && b_kr ? (long)((a.K - 1) / a.B.tn) * a.B.sg + (long)((a.K - 1) % a.B.tn) * a.B.st + ((a.N + 3) & ~3L) : a.spanB / 4;
      if (!a.amaxA) {          // (a caller that uses an operand in several launches passes its maximum in: gemm_amax)
        a.amaxA = sl;
        amax_add(J, list_view(a.A), a_kr, a.M, a.K, a.batch, a.sA, spa, sl);
      }
      if (!a.amaxB) {
        a.amaxB = slB;
        amax_add(J, list_view(a.B), b_kr, a.N, a.K, a.batch, a.sB, spb, slB);
      }
    }
    if (log_shapes)
      for (int i = 0; i < J.n; ++i)
        fprintf(stderr, "astk_gemm absmax region %d/%d: %ld x %ld rows (ld %ld) x %d floats = %.1f MB, %d blocks\n", i, J.n, J.r[i].nb, J.r[i].rows, J.r[i].ld,
                J.r[i].inner, 4e-6 * J.r[i].nb * J.r[i].rows * J.r[i].inner, J.blk_start[i + 1] - J.blk_start[i]);
    amax_blocks = J.n > 0 ? J.blk_start[J.n] : 0;
    if (amax_blocks > 0) {
      if (need_zero) hipLaunchKernelGGL(k_absmax_zero, dim3((unsigned)(amax_blocks + (G - 1) * ZERO_PARTS)), dim3(256), 0, s, J, grp, (int)G, TLM, TL, zero_vec ? 1 : 0);
      else hipLaunchKernelGGL(k_absmax, dim3((unsigned)amax_blocks), dim3(256), 0, s, J);
    }
  }
  if (need_zero && amax_blocks == 0) hipLaunchKernelGGL(k_zero_split_tiles, dim3((unsigned)((G - 1) * ZERO_PARTS)), dim3(256), 0, s, grp, (int)G, TLM, TL, zero_vec ? 1 : 0);
#define ASTK_GEMM_LAUNCH(T_, P_, M_)                                                                                              \
  switch (layout) {                                                                                                               \
    case GEMM_NT: hipLaunchKernelGGL((gemm_f32_kernel<T_, true, true, false, P_, M_>), grid, dim3(gemm_threads(P_, M_)), 0, s, grp); break;  \
    case GEMM_NN:                                                                                                                 \
      if (twolvl) hipLaunchKernelGGL((gemm_f32_kernel<T_, true, false, true, P_, M_>), grid, dim3(gemm_threads(P_, M_)), 0, s, grp);  \
      else hipLaunchKernelGGL((gemm_f32_kernel<T_, true, false, false, P_, M_>), grid, dim3(gemm_threads(P_, M_)), 0, s, grp);        \
      break;                                                                                                                      \
    default:                                                                                                                      \
      if (twolvl) hipLaunchKernelGGL((gemm_f32_kernel<T_, false, false, true, P_, M_>), grid, dim3(gemm_threads(P_, M_)), 0, s, grp); \
      else hipLaunchKernelGGL((gemm_f32_kernel<T_, false, false, false, P_, M_>), grid, dim3(gemm_threads(P_, M_)), 0, s, grp);       \
      break;                                                                                                                      \
  }
  if (prec == PREC_F32) {
    if (TL == 64) { ASTK_GEMM_LAUNCH(64, PREC_F32, 64) } else { ASTK_GEMM_LAUNCH(128, PREC_F32, 128) }
  } else if (prec == PREC_F16) { ASTK_GEMM_LAUNCH(128, PREC_F16, 128)
  } else if (prec == PREC_F16X2) {
    if (TL == 64) { ASTK_GEMM_LAUNCH(64, PREC_F16X2, 64) } else { ASTK_GEMM_LAUNCH(128, PREC_F16X2, 128) }
  } else if (TLM == 256) {
    if constexpr (BK == 16) { ASTK_GEMM_LAUNCH(128, PREC_BF16X3, 256) }      // (three 256-row stages of a 32-deep tile do not fit LDS)
  } else if (TL == 64) { ASTK_GEMM_LAUNCH(64, PREC_BF16X3, 64)
  } else if (fixup) {
    switch (layout) {
      case GEMM_NT: hipLaunchKernelGGL((gemm_f32_kernel<128, true, true, false, PREC_BF16X3, 128, true>), grid, dim3(gemm_threads(PREC_BF16X3, 128)), 0, s, grp); break;
      case GEMM_NN:
        if (twolvl) hipLaunchKernelGGL((gemm_f32_kernel<128, true, false, true, PREC_BF16X3, 128, true>), grid, dim3(gemm_threads(PREC_BF16X3, 128)), 0, s, grp);
        else hipLaunchKernelGGL((gemm_f32_kernel<128, true, false, false, PREC_BF16X3, 128, true>), grid, dim3(gemm_threads(PREC_BF16X3, 128)), 0, s, grp);
        break;
      default:
        if (twolvl) hipLaunchKernelGGL((gemm_f32_kernel<128, false, false, true, PREC_BF16X3, 128, true>), grid, dim3(gemm_threads(PREC_BF16X3, 128)), 0, s, grp);
        else hipLaunchKernelGGL((gemm_f32_kernel<128, false, false, false, PREC_BF16X3, 128, true>), grid, dim3(gemm_threads(PREC_BF16X3, 128)), 0, s, grp);
        break;
    }
  } else { ASTK_GEMM_LAUNCH(128, PREC_BF16X3, 128) }
#undef ASTK_GEMM_LAUNCH
  ASTK_LAUNCH_CHECK();
  return 0;
}

int gemm_launch(int layout, const GemmArgs& g, hipStream_t s) { return gemm_launch_group(layout, &g, 1, s); }

void gemm_amax_many(const AmaxMatrix* m, int n, const unsigned long long** out, hipStream_t s) {
  for (int i = 0; i < n; ++i) out[i] = nullptr;
  unsigned long long* ring = amax_ring();
  if (default_prec() != PREC_F16X2 || !ring) return;
  ProfScope prof(PROF_GEMM, s, 0.0);       // part of the GEMMs' cost: timed with them (no flops of its own)
  AmaxJobs J;
  memset(&J, 0, sizeof(J));
  J.gen = next_amax_gen(s);
  for (int i = 0; i < n; ++i) {
    if (!m[i].p || m[i].rows <= 0 || m[i].inner <= 0) continue;
    if (J.n + 2 > AMAX_REGIONS) {          // (two regions per contiguous matrix)
      hipLaunchKernelGGL(k_absmax, dim3((unsigned)J.blk_start[J.n]), dim3(256), 0, s, J);
      const unsigned gen = J.gen;
      memset(&J, 0, sizeof(J));
      J.gen = gen;
    }
    unsigned long long* slot = ring + AMAX_SLOTS / 2 + (size_t)(g_amax_handle.fetch_add(1) % (AMAX_RING_SLOTS / 2)) * AMAX_SLOT_WORDS;
    amax_add(J, mat(m[i].p, m[i].ld), false, (int)m[i].rows, m[i].inner, 1, 0, 0, slot);
    out[i] = slot;
  }
  if (J.n > 0) hipLaunchKernelGGL(k_absmax, dim3((unsigned)J.blk_start[J.n]), dim3(256), 0, s, J);
}

void gemm_amax_reserve(int n, unsigned long long** slots, unsigned* gen, hipStream_t s) {
  unsigned long long* ring = amax_ring();
  const bool on = default_prec() == PREC_F16X2 && ring != nullptr;
  *gen = on ? next_amax_gen(s) : 0;
  for (int i = 0; i < n; ++i)
    slots[i] = on ? ring + AMAX_SLOTS / 2 + (size_t)(g_amax_handle.fetch_add(1) % (AMAX_RING_SLOTS / 2)) * AMAX_SLOT_WORDS : nullptr;
}

// constant maxima (bounded matrices): 8 slots of 16 words; word 0 of a slot holds the bound, the others stay 0
__device__ unsigned long long g_amax_bounds[8 * AMAX_SLOT_WORDS];
const unsigned long long* gemm_amax_bound(float bound, hipStream_t s) {
  if (default_prec() != PREC_F16X2 && low_precision_gemms() == 0) return nullptr;
  static std::mutex mu;
  static float cached[16][8];          // per device: the bound each slot holds (0 = free)
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 16) return nullptr;
  unsigned long long* base = nullptr;
  if (hipGetSymbolAddress((void**)&base, HIP_SYMBOL(g_amax_bounds)) != hipSuccess || !base) return nullptr;
  std::lock_guard<std::mutex> lock(mu);
  for (int i = 0; i < 8; ++i) {
    if (cached[dev][i] == bound) return base + i * AMAX_SLOT_WORDS;
    if (cached[dev][i] == 0.f) {
      unsigned bits;
      memcpy(&bits, &bound, 4);
      // synchronous (once per bound and process): the handle is cached for every later caller on ANY stream, so its one-word
      // initialisation must not be ordered on the first caller's stream only
      const unsigned long long word = (unsigned long long)bits;
      if (hipMemcpy(base + i * AMAX_SLOT_WORDS, &word, sizeof(word), hipMemcpyHostToDevice) != hipSuccess) return nullptr;
      (void)s;
      cached[dev][i] = bound;
      return base + i * AMAX_SLOT_WORDS;
    }
  }
  return nullptr;                      // (more than 8 distinct bounds: the caller's launch measures the matrix itself)
}

const unsigned long long* gemm_amax(const float* p, long rows, long ld, int inner, hipStream_t s) {
  const AmaxMatrix m = {p, rows, ld, inner};
  const unsigned long long* h = nullptr;
  gemm_amax_many(&m, 1, &h, s);
  return h;
}

}  // namespace astk

extern "C" int astk_set_low_precision_gemms(int mode) {
  if (mode != 0 && mode != 1) { astk::set_error("set_low_precision_gemms: mode must be 0 or 1"); return -1; }
  astk::g_lowp_mode = mode;
  return 0;
}
extern "C" int astk_get_low_precision_gemms(void) { return astk::g_lowp_mode; }
extern "C" int astk_set_gemm_precision(int mode) {
  if (mode < 0 || mode > 2) { astk::set_error("set_gemm_precision: mode must be 0 (fp16x2), 1 (bf16x3) or 2 (f32)"); return -1; }
  const int p = astk::g_prec.load();
  const int prev = p == astk::PREC_F32 ? 2 : (p == astk::PREC_BF16X3 ? 1 : 0);
  astk::g_prec.store(mode == 2 ? astk::PREC_F32 : (mode == 1 ? astk::PREC_BF16X3 : astk::PREC_F16X2));
  return prev;
}
extern "C" int astk_get_gemm_precision(void) {
  const int p = astk::g_prec.load();
  return p == astk::PREC_F32 ? 2 : (p == astk::PREC_BF16X3 ? 1 : 0);
}
#ifdef ASTK_TEST_HOOKS
extern "C" int astk_debug_set_amax_generation(unsigned gen) { astk::g_amax_counter.store(gen ? gen : 1u); return 0; }
// libastk_test.so only: ONE grouped launch of n products C_i = A_i B_i (dense row-major operands, leading dimension = inner extent of the
// layout) the way a forward op issues it -- under GemmForwardScope when `forward` is set -- so that the two-contributor rule can be tested
// on shapes no op of the step produces (few tiles and deep K, groups of unequal K).
extern "C" int astk_debug_gemm_group(int layout, int n, const int* M, const int* N, const int* K, const float* const* A, const float* const* B,
                                     float* const* C, int forward, int precision, void* stream) {
  using namespace astk;
  ASTK_CHECK(n >= 1 && n <= GEMM_GROUP_MAX && layout >= GEMM_NT && layout <= GEMM_TN, "debug_gemm_group: bad arguments");
  PrecScope prec_scope(precision, 0);
  GemmArgs g[GEMM_GROUP_MAX];
  for (int i = 0; i < n; ++i) {
    const long lda = layout == GEMM_TN ? M[i] : K[i], ldb = layout == GEMM_NT ? K[i] : N[i];
    g[i] = gemm_args(M[i], N[i], K[i], mat(A[i], lda), mat(B[i], ldb), C[i], N[i]);
  }
  if (forward) { GemmForwardScope fs; return gemm_launch_group(layout, g, n, (hipStream_t)stream); }
  return gemm_launch_group(layout, g, n, (hipStream_t)stream);
}
#endif
extern "C" double astk_set_gemm_bf16_split_below(double flops) { return astk::g_small_flops.exchange(flops < 0 ? 0.0 : flops); }

namespace astk {

}  // namespace astk
