// rsn_wgrad.hip -- weight gradients dW[n][k] += sum_m dY[m][n] * X[m][k], db[n] += sum_m dY[m][n].
//
// A reduction GEMM over ALL sample points (M ~ 1e5..1e6) with a small output (<= 256 x 256): the opposite
// shape of the forward GEMMs, and one the BLAS library serves poorly (hipBLASLt picks 32x64 tiles: 36 ms per
// training step at BASELINE config 3).  Here the OUTPUT is stationary: a 4-wave workgroup keeps the whole
// dW tile in MFMA accumulators (wave w owns 64 output rows x all input columns = 16 blocks of 32x32 = 256
// accumulator registers) and streams its chunk of points once.  For v_mfma_f32_32x32x2_f32 with A = dY^T and
// B = X a fragment is "one float per lane from row m0 + h", straight from the row-major buffers: no LDS and no
// barriers in the main loop.  Which output row / input column a lane's fragment element stands for is a free
// permutation, so lane i takes the NKB CONSECUTIVE columns i*NKB .. i*NKB+NKB-1 (block kb = element kb) and
// the two consecutive rows 2i, 2i+1: one row of X is two global_load_dwordx4 per lane (a half-wave reads
// 1 KiB contiguous) and dY one dwordx2 -- 3 load instructions per 16 MFMAs instead of 10 (every vector-memory
// instruction costs the matrix pipe a few cycles, see rsn_mfma.h).  The permutation is undone once, at the
// flush, through a wave-private 2 KiB LDS tile, so that the fp32 atomics keep the full-rate shape (two 128-B
// row segments per wave instruction, MI355X_MICROARCH.md "Global float atomics").
//
// One launch can reduce over several point segments (the five field evaluations of a training step share
// their weights): the per-launch costs -- a 67 MB atomic flush for a 256 x 256 output on 256 workgroups,
// ~50 us -- are paid once per layer instead of once per layer and level.
//
// MFMA-bound: 2 * N * n_out * k_in FLOP; HBM reads N * (n_out + k_in) * 4 B (each operand once per workgroup).
#include "rsn_mfma.h"

// cache policy of the operand-row loads (buffer-instruction aux bits: 2 = nt); A/B switch of tools/bf16_train_ab.sh
#ifndef WG_LOAD_AUX
#define WG_LOAD_AUX 0
#endif

// split-bf16 stage: VALU instructions pinned in front of the first MFMA / behind every MFMA (sched_group_barrier)
#ifndef WG_X6_HEAD
#define WG_X6_HEAD 160
#endif
#ifndef WG_X6_VALU
#define WG_X6_VALU 4
#endif
#ifndef WG_X6_VALU4
#define WG_X6_VALU4 5  // 4 column blocks (k_in <= 128): measured 253 us against 275 with 6 (256 x 104 over 524,288 points)
#endif
#ifndef WG_X6_VALU2
#define WG_X6_VALU2 4
#endif
#ifndef WG_X6_VMEM
#define WG_X6_VMEM 4
#endif
typedef float f32x2w __attribute__((ext_vector_type(2)));
typedef unsigned u32x4w __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x2w __attribute__((ext_vector_type(2)));
__device__ __forceinline__ unsigned cvt2(float a, float b) {  // v_cvt_pk_bf16_f32: (bf16(b) << 16) | bf16(a), round to nearest even
  return __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2w{a, b}, bf16x2w));
}

#define WG_MAX_SEG 8

struct WGradArgs {
  int n_seg;
  long long seg_begin[WG_MAX_SEG + 1];  // prefix sums of the segment lengths (points): upper bounds when n_dev is set
  const int* n_dev[WG_MAX_SEG];         // optional device-side row count of segment s (rays), times ...
  int per_count[WG_MAX_SEG];            // ... rows per count (samples per ray): n_s = min(bound, *n_dev * per_count)
  const float* dy[WG_MAX_SEG];          // [n_s, ld_dy]
  const float* x[WG_MAX_SEG];           // [n_s, ld_x]
  int ld_dy, ld_x, n_out, k_in, ld_dw;
  const int* col_map;  // optional: packed column k -> destination column (or -1)
  float* dw;           // [n_out, ld_dw], accumulated
  float* db;           // [n_out] or NULL, accumulated
};

// Several reductions of the SAME shape (n_out, k_in, leading dimensions, operand types) and the same segment lengths in
// one launch: workgroup i works on job i % n_jobs with the other gridDim / n_jobs - 1 workgroups of that job.  Every
// workgroup still flushes one tile, so the launch pays ONE flush phase (and one ramp) for n_jobs layers.
#define WG_MAX_JOBS 8
struct WGradJobs {
  int n_jobs;
  WGradArgs j[WG_MAX_JOBS];
};

#ifndef WG_PAIRS
#define WG_PAIRS 4  // point pairs (MFMA K-steps) per software-pipeline stage
#endif

// NKB: input-column blocks of 32 held per wave (8 covers k_in <= 256).
// XV: lane i owns columns i*NKB.. (vector loads of X);  else column kb*32+i (scalar loads, any k_in / alignment).
// DV: lane i owns rows 2i, 2i+1 of the wave's 64 (dwordx2 loads of dY); else rows t*32+i.
// BF != 0: the same reduction on v_mfma_f32_32x32x16_bf16 over the same fp32 row-major operands (fp32 accumulation, fp32
// bias sums); a stage is 16 points, lane half h takes points 8h..8h+7 of it (the K index of the MFMA is the point).
//   BF = 3 (RSN_MMA_BF16X6): both operands split exactly into bf16 triples as they are packed, the 6 leading products
//          (dropped terms <= 2^-24 relative): fp32-equivalent at 2.7x the fp32 MFMA rate;
//   BF = 1 (RSN_MMA_BF16, the opt-in reduced-precision training mode): operands rounded to bf16, one product.
//   XB / DB (BF = 1 only): the X / dY rows ARE bf16 in memory (reduced-precision training saves its wide buffers as
//          bf16: rsn_field_saved, rsn_field_grads_out): half the bytes of this HBM-bound variant, no conversion; a lane's
//          8 points x {2 rows | NKB columns} arrive as packed words and are regrouped per row / column by v_perm_b32.
template <int NKB, bool XV, bool DV, int BF = 0, bool XB = false, bool DB = false>
__global__ __launch_bounds__(256) void rsn_wgrad_kernel(const WGradJobs J) {
  const int n_jobs = J.n_jobs;
  const WGradArgs& a = J.j[blockIdx.x % n_jobs];  // workgroup-uniform
  static_assert(!(XB || DB) || (BF == 1 && XV), "bf16 rows: the plain-bf16 variant with vector loads of X only");
  static_assert(!DB || DV, "bf16 dY rows come as (row 2i, row 2i+1) pairs");
  constexpr int NP = BF ? 8 : WG_PAIRS;  // points per lane and stage
  __shared__ float tr[4][2][NKB * 32];
  const int lane = threadIdx.x & 63;
  const int wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);  // provably wave-uniform: scalar loop control
  const int i = lane & 31, h = lane >> 5;
  // Waves are assigned (row-block pair, point sub-chunk): P = pairs of 32-row blocks the output needs (1, 2 or 4);
  // the 4/P waves that share a pair split the workgroup's chunk of points, so narrow outputs (heads, RGB head,
  // mlp_mid) keep every wave busy.
  const int P = a.n_out <= 64 ? 1 : (a.n_out <= 128 ? 2 : 4);
  const int nsub = 4 / P;
  const int nb0 = (wid % P) * 2;  // this wave's two 32-row output blocks
  const int sub = wid / P;
  // Pipeline stages (2*WG_PAIRS consecutive points) are dealt round-robin to the G = gridDim * nsub wave slots: at any
  // moment the chip reads one contiguous window of G stages (no equal-offset streams from 1 MiB-spaced bases), and
  // the split is even for any number and size of segments.
  // Measured (tools/wgrad_report.py, 256 x 256 output): 281 us per 262,144 points = 122 TFLOP/s in the streaming
  // part, 246 us with the loads removed: the four waves of a workgroup each fetch the whole X row (L1 only partly
  // dedups them), ~10 B/cycle/CU of fill traffic, the per-CU streaming limit.  Sharing X through LDS would halve
  // that; not done yet.
  const long long G = (long long)(gridDim.x / n_jobs) * nsub;  // wave slots of this job
  const long long g = (long long)(blockIdx.x / n_jobs) * nsub + sub;
  const bool t1_live = DV || (nb0 + 1) * 32 < a.n_out;  // wave-uniform: second row block holds live rows

  f32x16 acc[2][NKB];
#pragma unroll
  for (int t = 0; t < 2; ++t)
#pragma unroll
    for (int kb = 0; kb < NKB; ++kb)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[t][kb][r] = 0.0f;
  float bsum[2] = {0.0f, 0.0f};

  // Lane rows / columns are clamped into range instead of masked: a lane whose output row / input column does
  // not exist works on a duplicate of a valid one and its results are never flushed.  Nothing in the main loop
  // touches a loaded value before the MFMAs do, so the loads of stage s+1 stay in flight under the MFMAs of stage s.
  int cdy[2], cx[NKB];
  if (DV) {
    const int n_even = a.n_out + (a.n_out & 1);  // the host checked ld_dy >= n_even
    const int c = nb0 * 32 + 2 * i;
    cdy[0] = c < n_even - 2 ? c : n_even - 2;
    cdy[1] = cdy[0] + 1;
  } else {
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      const int c = (nb0 + t) * 32 + i;
      cdy[t] = c < a.n_out ? c : a.n_out - 1;
    }
  }
  if (XV) {
    const int c = i * NKB;  // the host checked k_in % NKB == 0
    cx[0] = c < a.k_in - NKB ? c : a.k_in - NKB;
#pragma unroll
    for (int kb = 1; kb < NKB; ++kb) cx[kb] = cx[0] + kb;
  } else {
#pragma unroll
    for (int kb = 0; kb < NKB; ++kb) {
      const int c = kb * 32 + i;
      cx[kb] = c < a.k_in ? c : a.k_in - 1;
    }
  }

  // Stage buffers.  Exact fp32 (8 points per stage, 40 registers): FOUR buffers, three stages (12,288 MFMA cycles) in flight
  // -- possible since the buffer loads freed ~150 address registers; 12.75 -> 12.58 ms per step.  Split / plain bf16 over
  // fp32 rows: 80 registers per stage, two fit beside the 256 accumulators (one stage = 512 MFMA cycles of cover).  bf16 rows are 40 registers per stage: FOUR buffers, three stages (1536 MFMA cycles)
  // in flight -- the plain-bf16 reduction is bound by load latency, not by bytes (profiles/r03_wgrad_rows.txt).
  constexpr int NBUF = ((BF == 0 && XV) || (DB && (XB || NKB <= 4))) ? 4 : 2;  // (fp32 X rows of <= 128 columns are <= 40 registers per stage too)
  float fa[NBUF][DB ? 1 : NP][2], fb[NBUF][XB ? 1 : NP][NKB];  // [buffer][point of this lane][block]
  unsigned ua[NBUF][DB ? NP : 1], ub[NBUF][XB ? NP : 1][NKB / 2];  // bf16 rows: packed pairs (rows 2i, 2i+1 | columns 2w, 2w+1)
  auto roff = [&](int p) { return BF ? 8 * h + p : 2 * p + h; };  // row of the stage this lane's p-th point is
  const float* __restrict__ dyp = nullptr;
  const float* __restrict__ xp = nullptr;

  auto load_row = [&](int buf, int p, long long m, bool in) {
    const float* __restrict__ dr = dyp + m * a.ld_dy;
    const float* __restrict__ xr = xp + m * a.ld_x;
    if constexpr (DB) {
      const unsigned v = *reinterpret_cast<const unsigned*>(reinterpret_cast<const __bf16*>(dyp) + m * a.ld_dy + cdy[0]);
      ua[buf][p] = in ? v : 0u;
    } else if constexpr (DV) {
      const float2 v = *reinterpret_cast<const float2*>(dr + cdy[0]);
      fa[buf][p][0] = in ? v.x : 0.0f;
      fa[buf][p][1] = in ? v.y : 0.0f;
    } else {
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        const float v = dr[cdy[t]];
        fa[buf][p][t] = in ? v : 0.0f;
      }
    }
    if constexpr (XB) {
      const __bf16* __restrict__ xb = reinterpret_cast<const __bf16*>(xp) + m * a.ld_x + cx[0];
      if constexpr (NKB == 8) {
        const uint4 v = *reinterpret_cast<const uint4*>(xb);
        ub[buf][p][0] = in ? v.x : 0u; ub[buf][p][1] = in ? v.y : 0u;
        ub[buf][p][2] = in ? v.z : 0u; ub[buf][p][3] = in ? v.w : 0u;
      } else if constexpr (NKB == 4) {
        const uint2 v = *reinterpret_cast<const uint2*>(xb);
        ub[buf][p][0] = in ? v.x : 0u; ub[buf][p][1] = in ? v.y : 0u;
      } else {
        const unsigned v = *reinterpret_cast<const unsigned*>(xb);
        ub[buf][p][0] = in ? v : 0u;
      }
    } else if constexpr (XV && NKB >= 4) {
#pragma unroll
      for (int q = 0; q < NKB / 4; ++q) {
        const float4 v = *reinterpret_cast<const float4*>(xr + cx[0] + 4 * q);
        fb[buf][p][4 * q + 0] = in ? v.x : 0.0f;
        fb[buf][p][4 * q + 1] = in ? v.y : 0.0f;
        fb[buf][p][4 * q + 2] = in ? v.z : 0.0f;
        fb[buf][p][4 * q + 3] = in ? v.w : 0.0f;
      }
    } else if constexpr (XV) {
      const float2 v = *reinterpret_cast<const float2*>(xr + cx[0]);
      fb[buf][p][0] = in ? v.x : 0.0f;
      fb[buf][p][1] = in ? v.y : 0.0f;
    } else {
#pragma unroll
      for (int kb = 0; kb < NKB; ++kb) {
        const float v = xr[cx[kb]];
        fb[buf][p][kb] = in ? v : 0.0f;
      }
    }
  };
  // The same rows through BUFFER loads (the main loop): per stage one scalar descriptor for dY and one for X (base = row m0
  // of the segment, wave-uniform), the point of the stage and the float4 of the row in the scalar offset, and ONE 32-bit
  // lane offset per operand (the lane's half h and its column), loop-invariant -- instead of a 64-bit address per load,
  // rebuilt from the row index with v_mul / v_mad_u64 / v_lshl_add_u64 (57 VALU per stage pair).
  typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
  typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
  constexpr int DBPE = DB ? 2 : 4, XBPE = XB ? 2 : 4;
  const int hrow = BF ? 8 * h : h;  // row of the stage this lane's point p is: hrow + prow(p)
  auto prow = [&](int p) { return BF ? p : 2 * p; };
  unsigned vd[2], vx[NKB];
#pragma unroll
  for (int t = 0; t < 2; ++t) vd[t] = (unsigned)((hrow * a.ld_dy + cdy[t]) * DBPE);
#pragma unroll
  for (int kb = 0; kb < NKB; ++kb) vx[kb] = (unsigned)((hrow * a.ld_x + cx[kb]) * XBPE);
  auto load_row_b = [&](int buf, int p, long long m0) {
    const __amdgpu_buffer_rsrc_t rd = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<char*>(reinterpret_cast<const char*>(dyp)) + m0 * a.ld_dy * DBPE, 0, 0x7fffffff, 0x00020000);
    const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<char*>(reinterpret_cast<const char*>(xp)) + m0 * a.ld_x * XBPE, 0, 0x7fffffff, 0x00020000);
    const unsigned sd = (unsigned)(prow(p) * a.ld_dy * DBPE), sx = (unsigned)(prow(p) * a.ld_x * XBPE);
    if constexpr (DB) {
      ua[buf][p] = __builtin_amdgcn_raw_buffer_load_b32(rd, vd[0], sd, WG_LOAD_AUX);
    } else if constexpr (DV) {
      const u32x2 v = __builtin_amdgcn_raw_buffer_load_b64(rd, vd[0], sd, WG_LOAD_AUX);
      fa[buf][p][0] = __uint_as_float(v.x);
      fa[buf][p][1] = __uint_as_float(v.y);
    } else {
#pragma unroll
      for (int t = 0; t < 2; ++t) fa[buf][p][t] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rd, vd[t], sd, WG_LOAD_AUX));
    }
    if constexpr (XB) {
      if constexpr (NKB == 8) {
        const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rx, vx[0], sx, WG_LOAD_AUX);
        ub[buf][p][0] = v.x; ub[buf][p][1] = v.y; ub[buf][p][2] = v.z; ub[buf][p][3] = v.w;
      } else if constexpr (NKB == 4) {
        const u32x2 v = __builtin_amdgcn_raw_buffer_load_b64(rx, vx[0], sx, WG_LOAD_AUX);
        ub[buf][p][0] = v.x; ub[buf][p][1] = v.y;
      } else {
        ub[buf][p][0] = __builtin_amdgcn_raw_buffer_load_b32(rx, vx[0], sx, WG_LOAD_AUX);
      }
    } else if constexpr (XV && NKB >= 4) {
#pragma unroll
      for (int q = 0; q < NKB / 4; ++q) {
        const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rx, vx[0], sx + 16u * q, WG_LOAD_AUX);
        fb[buf][p][4 * q + 0] = __uint_as_float(v.x);
        fb[buf][p][4 * q + 1] = __uint_as_float(v.y);
        fb[buf][p][4 * q + 2] = __uint_as_float(v.z);
        fb[buf][p][4 * q + 3] = __uint_as_float(v.w);
      }
    } else if constexpr (XV) {
      const u32x2 v = __builtin_amdgcn_raw_buffer_load_b64(rx, vx[0], sx, WG_LOAD_AUX);
      fb[buf][p][0] = __uint_as_float(v.x);
      fb[buf][p][1] = __uint_as_float(v.y);
    } else {
#pragma unroll
      for (int kb = 0; kb < NKB; ++kb) fb[buf][p][kb] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rx, vx[kb], sx, WG_LOAD_AUX));
    }
  };
  // every variant loads through the buffer form.  (Round 3 kept pointer loads for the split-bf16 variant -- 5 % faster then, when its
  // splits stood in blocks in front of the MFMAs; with the splits pinned between the MFMAs the 30 address registers are worth more.
  // -DWG_X6_PTR_LOADS keeps the old form for A/B in diagnostic builds.)
  auto load_row_m = [&](int buf, int p, long long m0) {
#ifdef WG_X6_PTR_LOADS
    if constexpr (BF == 3) load_row(buf, p, m0 + roff(p), true); else load_row_b(buf, p, m0);
#else
    load_row_b(buf, p, m0);
#endif
  };
  // Two neighbouring rows / columns c0, c0 + 1 at once: their values of ONE point sit in adjacent registers (one load), so
  // the subtractions are v_pk_add_f32 over the (c0, c0 + 1) pair while v_cvt_pk_bf16_f32 packs the point pair (2q, 2q + 1)
  // of each: 18 VALU per 2 x 2 values.
  auto split2 = [&](u32x4 (&o0)[3], u32x4 (&o1)[3], auto&& val) {  // val(p): the (c0, c0 + 1) values of the lane's p-th point
    const unsigned HI = 0xffff0000u;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const f32x2w xa = val(2 * q), xb = val(2 * q + 1);
      const unsigned h0 = cvt2(xa[0], xb[0]), h1 = cvt2(xa[1], xb[1]);
      const f32x2w ra = xa - f32x2w{__uint_as_float(h0 << 16), __uint_as_float(h1 << 16)};
      const f32x2w rb = xb - f32x2w{__uint_as_float(h0 & HI), __uint_as_float(h1 & HI)};
      const unsigned m0 = cvt2(ra[0], rb[0]), m1 = cvt2(ra[1], rb[1]);
      const f32x2w sa = ra - f32x2w{__uint_as_float(m0 << 16), __uint_as_float(m1 << 16)};
      const f32x2w sb = rb - f32x2w{__uint_as_float(m0 & HI), __uint_as_float(m1 & HI)};
      o0[0][q] = h0; o1[0][q] = h1;
      o0[1][q] = m0; o1[1][q] = m1;
      o0[2][q] = cvt2(sa[0], sb[0]); o1[2][q] = cvt2(sa[1], sb[1]);
    }
  };
  auto mma6 = [&](f32x16& cacc, const u32x4 (&aw)[3], const u32x4 (&bw)[3]) {  // six products, the smallest terms first
    const bf16x8 a1 = __builtin_bit_cast(bf16x8, aw[0]), a2 = __builtin_bit_cast(bf16x8, aw[1]), a3 = __builtin_bit_cast(bf16x8, aw[2]);
    const bf16x8 b1 = __builtin_bit_cast(bf16x8, bw[0]), b2 = __builtin_bit_cast(bf16x8, bw[1]), b3 = __builtin_bit_cast(bf16x8, bw[2]);
    f32x16 c = cacc;
    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a3, b1, c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a2, b2, c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, b3, c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a2, b1, c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, b2, c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, b1, c, 0, 0, 0);
    cacc = c;
  };
  auto mma_stage = [&](int buf) {
#ifndef WG_X6_OLD
    if constexpr (BF == 3) {
      // Split-bf16, one stage on its own (the masked tail stage of a segment; the main loop is x6_stage below): each fp32 operand
      // value becomes three bf16 pieces (hi, mid, lo; x = hi + mid + lo to 2^-24).  The 8 points of a lane x (2 rows + NKB columns)
      // are 2 + NKB split2 calls per stage against 12 NKB MFMAs; a wave that is alone on its SIMD issues in order, so the splits of
      // column pair j + 1 are written (and pinned, interleave_stage) BETWEEN the MFMAs of pair j.
      u32x4 av[2][3], bv[2][2][3];  // [row] / [buffer][column of the pair]: hi, mid, lo pieces as packed point pairs
#pragma unroll
      for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int p = 0; p < NP; ++p) bsum[t] += fa[buf][p][t];
      split2(av[0], av[1], [&](int p) { return f32x2w{fa[buf][p][0], fa[buf][p][1]}; });
      split2(bv[0][0], bv[0][1], [&](int p) { return f32x2w{fb[buf][p][0], fb[buf][p][1]}; });
#pragma unroll
      for (int j = 0; j < NKB / 2; ++j) {
        if (j + 1 < NKB / 2)
          split2(bv[(j + 1) & 1][0], bv[(j + 1) & 1][1], [&](int p) { return f32x2w{fb[buf][p][2 * j + 2], fb[buf][p][2 * j + 3]}; });
#pragma unroll
        for (int c = 0; c < 2; ++c)
#pragma unroll
          for (int t = 0; t < 2; ++t)
            if (t == 0 || t1_live) mma6(acc[t][2 * j + c], av[t], bv[j & 1][c]);
      }
    } else
#endif
    if constexpr (BF == 3) {
      bf16x8 a1[2], a2[2], a3[2];
#pragma unroll
      for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int p = 0; p < NP; ++p) {
          const float x = fa[buf][p][t];
          bsum[t] += x;
          const __bf16 s1 = (__bf16)x;
          const float r1 = x - (float)s1;
          const __bf16 s2 = (__bf16)r1;
          a1[t][p] = s1;
          a2[t][p] = s2;
          a3[t][p] = (__bf16)(r1 - (float)s2);
        }
#pragma unroll
      for (int kb = 0; kb < NKB; ++kb) {
        bf16x8 b1, b2, b3;
#pragma unroll
        for (int p = 0; p < NP; ++p) {
          const float x = fb[buf][p][kb];
          const __bf16 s1 = (__bf16)x;
          const float r1 = x - (float)s1;
          const __bf16 s2 = (__bf16)r1;
          b1[p] = s1;
          b2[p] = s2;
          b3[p] = (__bf16)(r1 - (float)s2);
        }
#pragma unroll
        for (int t = 0; t < 2; ++t) {
          if (t == 1 && !t1_live) continue;
          f32x16 c = acc[t][kb];  // smallest terms first
          c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a3[t], b1, c, 0, 0, 0);
          c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a2[t], b2, c, 0, 0, 0);
          c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1[t], b3, c, 0, 0, 0);
          c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a2[t], b1, c, 0, 0, 0);
          c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1[t], b2, c, 0, 0, 0);
          c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1[t], b1, c, 0, 0, 0);
          acc[t][kb] = c;
        }
      }
    } else if constexpr (BF != 0) {
      bf16x8 av[2], bv[NKB];
      typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
      if constexpr (DB) {  // packed (row 2i, row 2i+1) pairs of 8 points -> one bf16x8 per row: v_perm_b32 picks the halves
        u32x4 lo, hi;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const unsigned w0 = ua[buf][2 * j], w1 = ua[buf][2 * j + 1];
          lo[j] = __builtin_amdgcn_perm(w1, w0, 0x05040100u);
          hi[j] = __builtin_amdgcn_perm(w1, w0, 0x07060302u);
          bsum[0] += __uint_as_float(w0 << 16) + __uint_as_float(w1 << 16);
          bsum[1] += __uint_as_float(w0 & 0xffff0000u) + __uint_as_float(w1 & 0xffff0000u);
        }
        av[0] = __builtin_bit_cast(bf16x8, lo);
        av[1] = __builtin_bit_cast(bf16x8, hi);
      } else {
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
          for (int p = 0; p < NP; ++p) {
            bsum[t] += fa[buf][p][t];
            av[t][p] = (__bf16)fa[buf][p][t];
          }
      }
      if constexpr (XB) {
#pragma unroll
        for (int w = 0; w < NKB / 2; ++w) {
          u32x4 lo, hi;
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            lo[j] = __builtin_amdgcn_perm(ub[buf][2 * j + 1][w], ub[buf][2 * j][w], 0x05040100u);
            hi[j] = __builtin_amdgcn_perm(ub[buf][2 * j + 1][w], ub[buf][2 * j][w], 0x07060302u);
          }
          bv[2 * w] = __builtin_bit_cast(bf16x8, lo);
          bv[2 * w + 1] = __builtin_bit_cast(bf16x8, hi);
        }
      } else {
#pragma unroll
        for (int kb = 0; kb < NKB; ++kb)
#pragma unroll
          for (int p = 0; p < NP; ++p) bv[kb][p] = (__bf16)fb[buf][p][kb];
      }
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        if (t == 1 && !t1_live) continue;
#pragma unroll
        for (int kb = 0; kb < NKB; ++kb)
          acc[t][kb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av[t], bv[kb], acc[t][kb], 0, 0, 0);
      }
    } else {
#pragma unroll
      for (int p = 0; p < WG_PAIRS; ++p) {
#pragma unroll
        for (int t = 0; t < 2; ++t) {
          if (t == 1 && !t1_live) continue;
          bsum[t] += fa[buf][p][t];
#pragma unroll
          for (int kb = 0; kb < NKB; ++kb)
            acc[t][kb] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[buf][p][t], fb[buf][p][kb], acc[t][kb], 0, 0, 0);
        }
      }
    }
  };

#ifndef WG_X6_OLD
  // The split-bf16 main loop, pipelined ACROSS stages: while stage s multiplies, the splits of its own later column pairs and --
  // under its last pair -- of stage s + 1's rows and first column pair are formed, so that no split stands in front of an MFMA
  // (a wave is alone on its SIMD: what is not slotted between MFMAs is added to them).  xa / xb01: the split rows / first column
  // pair of the stage about to multiply.  Two fp32 stage buffers as before: stage s + 2 is loaded into stage s's buffer behind
  // its last split.
  u32x4 xa[BF == 3 ? 2 : 1][3], xb01[BF == 3 ? 2 : 1][3];
  auto x6_head = [&](int buf) {
    if constexpr (BF == 3) {
      split2(xa[0], xa[1], [&](int p) { return f32x2w{fa[buf][p][0], fa[buf][p][1]}; });
      split2(xb01[0], xb01[1], [&](int p) { return f32x2w{fb[buf][p][0], fb[buf][p][1]}; });
    }
  };
  auto x6_stage = [&](int c, auto&& load_next2) {  // multiplies the stage in buffer c; load_next2(c) refills it
    if constexpr (BF == 3) {
      constexpr int NPR = NKB / 2;
      u32x4 bq[NPR + 1][2][3], an[2][3];
#pragma unroll
      for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int p = 0; p < NP; ++p) bsum[t] += fa[c][p][t];
#pragma unroll
      for (int k = 0; k < 3; ++k) { bq[0][0][k] = xb01[0][k]; bq[0][1][k] = xb01[1][k]; }
#ifdef WG_X6_NEAR
      load_next2(c ^ 1);
#endif
#pragma unroll
      for (int j = 0; j < NPR; ++j) {
        if (j + 1 < NPR) {
          split2(bq[j + 1][0], bq[j + 1][1], [&](int p) { return f32x2w{fb[c][p][2 * j + 2], fb[c][p][2 * j + 3]}; });
        } else {
#ifndef WG_X6_NEAR
          load_next2(c);
#endif
          split2(an[0], an[1], [&](int p) { return f32x2w{fa[c ^ 1][p][0], fa[c ^ 1][p][1]}; });
          split2(bq[NPR][0], bq[NPR][1], [&](int p) { return f32x2w{fb[c ^ 1][p][0], fb[c ^ 1][p][1]}; });
        }
#pragma unroll
        for (int cc = 0; cc < 2; ++cc)
#pragma unroll
          for (int t = 0; t < 2; ++t)
            if (t == 0 || t1_live) mma6(acc[t][2 * j + cc], xa[t], bq[j][cc]);
      }
#pragma unroll
      for (int k = 0; k < 3; ++k) {
        xa[0][k] = an[0][k]; xa[1][k] = an[1][k];
        xb01[0][k] = bq[NPR][0][k]; xb01[1][k] = bq[NPR][1][k];
      }
#ifdef WG_X6_NEAR
      constexpr int NM = 12 * NKB, NL = 8 * (1 + (NKB >= 4 ? NKB / 4 : 1)), G0 = 0;
#else
      constexpr int NM = 12 * NKB, NL = 8 * (1 + (NKB >= 4 ? NKB / 4 : 1)), G0 = NM - 12 - NL > 0 ? NM - 12 - NL : 0;
#endif
#pragma unroll
      for (int g = 0; g < NM; ++g) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                                   // 1 MFMA
        __builtin_amdgcn_sched_group_barrier(0x002, NKB == 8 ? WG_X6_VALU : (NKB == 4 ? WG_X6_VALU4 : WG_X6_VALU2), 0);  // splits
#ifdef WG_X6_EXTRA
        if (NKB == 8 && g % WG_X6_EXTRA == 0) __builtin_amdgcn_sched_group_barrier(0x002, 1, 0);
#endif
        if (g >= G0 && g < G0 + NL) __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);       // one load of stage s + 2
      }
      __builtin_amdgcn_sched_barrier(0);
    }
  };
#endif
  auto interleave_stage = [&]() {
#ifndef WG_X6_OLD
    if constexpr (BF == 3) {
      __builtin_amdgcn_sched_group_barrier(0x002, WG_X6_HEAD, 0);  // the splits of both rows and of column block 0
#pragma unroll
      for (int g = 0; g < 12 * NKB; ++g) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);          // 1 MFMA
        __builtin_amdgcn_sched_group_barrier(0x002, WG_X6_VALU, 0);  // VALU of the next column block's splits / addresses
#if WG_X6_VMEM
        if (g % WG_X6_VMEM == 0) __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);  // one of the next stage's loads
#endif
      }
      return;
    }
#endif
#ifdef WG_F32_SPREAD
    if constexpr (BF == 0 && XV && DV) {  // one load behind each of the first MFMAs (hipcc otherwise issues them in blocks of 4-8)
#pragma unroll
      for (int g = 0; g < WG_PAIRS * 2 * (1 + NKB / 4); ++g) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x006, WG_F32_SPREAD, 0);
      }
      return;
    }
#endif
#pragma unroll
    for (int g = 0; g < (BF == 3 ? 96 : (BF ? 16 : WG_PAIRS * 4)); ++g) {
      __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);  // 1 MFMA
      __builtin_amdgcn_sched_group_barrier(0x026, 2, 0);  // up to 2 of VALU / SALU / VMEM read
    }
  };

  const long long step = 2 * NP;
  long long vprefix = 0;  // stages (full and tail) of the segments in front of this one
  bool any = false;
#pragma unroll 1
  for (int s = 0; s < a.n_seg; ++s) {
    long long n_s = a.seg_begin[s + 1] - a.seg_begin[s];
    if (a.n_dev[s]) {  // the reflected-ray count lives on the device (no host read in the training step): wave-uniform
      const long long nd = (long long)(*a.n_dev[s]) * a.per_count[s];
      n_s = nd < n_s ? (nd > 0 ? nd : 0) : n_s;
    }
    const long long n_full = n_s / step;  // stages with every point in range
    const long long rem = n_s - n_full * step;
    const long long first = ((g - vprefix) % G + G) % G;  // this slot's first stage of the segment
    const long long cnt = first < n_full ? (n_full - first + G - 1) / G : 0;
    dyp = a.dy[s];
    xp = a.x[s];
    const long long gs = G * step;
    // stage t of this slot starts at row (first + t G) step; the prefetch index is clamped, never branched on (straight-line
    // loop bodies let the scheduler slot the loads between the MFMAs)
    auto load_stage = [&](int buf, long long t) {
      const long long tc = t < cnt ? t : cnt - 1;
      const long long m0 = (first + tc * G) * step;
#pragma unroll
      for (int p = 0; p < NP; ++p) load_row_m(buf, p, m0);
    };
    if (cnt > 0) {
      any = true;
      if constexpr (NBUF == 4) {
#pragma unroll
        for (int b = 0; b < NBUF - 1; ++b) load_stage(b, b);
      } else {
#pragma unroll
        for (int p = 0; p < NP; ++p) load_row_m(0, p, first * step);
      }
    }
    long long j = 0;
#if !defined(WG_X6_OLD) && !defined(WG_X6_INSTAGE)
    if constexpr (BF == 3) {
#ifdef WG_X6_NEAR
      if (cnt > 0) {
        x6_head(0);
        __builtin_amdgcn_sched_barrier(0);
      }
#pragma unroll 1
      for (; j + 1 < cnt; j += 2) {
        x6_stage(0, [&](int c) { load_stage(c, j + 1); });
        x6_stage(1, [&](int c) { load_stage(c, j + 2); });
      }
      if (j < cnt) x6_stage(0, [&](int c) { load_stage(c, j + 1); });
#else
      if (cnt > 0) {
        load_stage(1, 1);
        x6_head(0);
        __builtin_amdgcn_sched_barrier(0);
      }
#pragma unroll 1
      for (; j + 1 < cnt; j += 2) {
        x6_stage(0, [&](int c) { load_stage(c, j + 2); });
        x6_stage(1, [&](int c) { load_stage(c, j + 3); });
      }
      if (j < cnt) x6_stage(0, [&](int c) { load_stage(c, j + 2); });  // (its look-ahead works on a repeat of the last stage: unused)
#endif
    } else
#endif
    if constexpr (NBUF == 4) {
#pragma unroll 1
      for (; j + 3 < cnt; j += 4) {
        load_stage(3, j + 3);
        mma_stage(0);
        interleave_stage();
        __builtin_amdgcn_sched_barrier(0);
        load_stage(0, j + 4);
        mma_stage(1);
        interleave_stage();
        __builtin_amdgcn_sched_barrier(0);
        load_stage(1, j + 5);
        mma_stage(2);
        interleave_stage();
        __builtin_amdgcn_sched_barrier(0);
        load_stage(2, j + 6);
        mma_stage(3);
        interleave_stage();
        __builtin_amdgcn_sched_barrier(0);
      }
      if (j < cnt) mma_stage(0);      // the last 0..3 stages are in buffers 0..2 already
      if (j + 1 < cnt) mma_stage(1);
      if (j + 2 < cnt) mma_stage(2);
    } else {
      const long long gs = G * step;
#pragma unroll 1
      for (; j + 1 < cnt; j += 2) {
        const long long m0 = (first + j * G) * step;
        // the next stage's loads (and their address arithmetic) are slotted between this stage's MFMAs: issued as a
        // block between two MFMA bursts they left the matrix pipe idle for ~10 % of the loop.  Straight-line body
        // (the prefetch index is clamped, not branched on) so that the scheduler can interleave.
#pragma unroll
        for (int p = 0; p < NP; ++p) load_row_m(1, p, m0 + gs);
        mma_stage(0);
        interleave_stage();
        __builtin_amdgcn_sched_barrier(0);
        const long long m2 = (j + 2 < cnt) ? m0 + 2 * gs : m0 + gs;
#pragma unroll
        for (int p = 0; p < NP; ++p) load_row_m(0, p, m2);
        mma_stage(1);
        interleave_stage();
        __builtin_amdgcn_sched_barrier(0);
      }
      if (j < cnt) mma_stage(0);
    }
    if (rem > 0 && (vprefix + n_full) % G == g) {  // the segment's tail stage: points beyond n_s contribute zeros
      any = true;
#pragma unroll
      for (int p = 0; p < NP; ++p) {
        const long long m = n_full * step + roff(p);
        const bool in = m < n_s;
        load_row(0, p, in ? m : 0, in);
      }
      mma_stage(0);
    }
    vprefix += n_full + (rem > 0 ? 1 : 0);
  }
  if (!any) return;  // wave-uniform: nothing accumulated, nothing to flush
#ifdef RSN_DIAG_WG_NO_FLUSH  // timing ablation (wrong results): what the atomic flush costs
  if (acc[0][0][0] != 12345.678f) return;
#endif

  // flush: C/D layout col = lane&31 (input-column slot), row = (r&3) + 8*(r>>2) + 4*h (output-row slot).  The
  // wave-private LDS tile turns "lane i holds columns i*NKB+kb" back into "lane i holds column kb*32+i" so that one
  // atomic wave-instruction covers two contiguous 128-B row segments.
  float* trw = &tr[wid][h][0];
  // destination columns first: a load between two atomics would wait (vmcnt counts both) for the atomic in front of it
  int cdst[NKB];
#pragma unroll
  for (int kb = 0; kb < NKB; ++kb) {
    const int k = kb * 32 + i;
    cdst[kb] = -1;
    if (k < a.k_in) cdst[kb] = a.col_map ? a.col_map[k] : k;
  }
  // buffer atomics: descriptor over the n_out live rows of dW; the lane's WHOLE byte offset (row and column) travels in
  // the vector offset -- the one operand the hardware range-checks against num_records (the scalar offset is added after
  // the check), so rows >= n_out (clamped duplicates of live rows, non-zero) are dropped whatever lies behind dW's n_out
  // rows (in the training step: the next parameter's gradient in the flat buffer); padded columns get an offset far
  // past the descriptor (no 32-bit wrap with the row part: rows * ld_dw * 4 < 2^30)
  const __amdgpu_buffer_rsrc_t rdw = __builtin_amdgcn_make_buffer_rsrc(a.dw, 0, a.n_out * a.ld_dw * 4, 0x00020000);
  unsigned vdw[NKB];
#pragma unroll
  for (int kb = 0; kb < NKB; ++kb)
    vdw[kb] = cdst[kb] >= 0 ? (unsigned)(((DV ? 8 : 4) * h * a.ld_dw + cdst[kb]) * 4) : 0x40000000u;
#pragma unroll
  for (int t = 0; t < 2; ++t) {
    if (t == 1 && !t1_live) continue;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int slot0 = (r & 3) + 8 * (r >> 2);  // + 4 h: in vdw
      const int n0 = DV ? nb0 * 32 + 2 * slot0 + t : (nb0 + t) * 32 + slot0;
#pragma unroll
      for (int kb = 0; kb < NKB; ++kb) trw[XV ? i * NKB + kb : kb * 32 + i] = acc[t][kb][r];
#pragma unroll
      for (int kb = 0; kb < NKB; ++kb) {
        const float v = trw[kb * 32 + i];
        __builtin_amdgcn_raw_ptr_buffer_atomic_fadd_f32(v, rdw, vdw[kb] + (unsigned)(n0 * a.ld_dw * 4), 0u, 0);
      }
    }
    if (a.db) {
      const float v = bsum[t] + __shfl_xor(bsum[t], 32, 64);
      const int n = DV ? nb0 * 32 + 2 * i + t : (nb0 + t) * 32 + i;
      if (h == 0 && n < a.n_out) atomicAdd(&a.db[n], v);
    }
  }
}

#if defined(WG_X6_STAGED) || defined(WG_F32_STAGED)
#include "rsn_wgrad_staged_probe.h"  // LDS-staged probe kernels: diagnostic builds only
#endif

static int wgrad_launch(WGradJobs& J, void* stream, int mode = 0, int operand_bf16 = 0) {
  WGradArgs& a = J.j[0];  // the jobs of a launch share shape, leading dimensions and segment lengths (checked by the caller)
  const bool xb = (operand_bf16 & 1) != 0, db = (operand_bf16 & 2) != 0;  // rows that ARE bf16 in memory
  bool bf16 = mode == RSN_MMA_BF16 || mode == RSN_MMA_BF16X6;
  const long long total = a.seg_begin[a.n_seg];
  if (total == 0) return RSN_OK;
  const int cached_cus = rsn_device_cus();
  // grid: every workgroup pays one atomic flush of the output tile (chip-wide ~1.3 TB/s of added bytes) and the
  // waves share the stages; T(G) = stages / (G * nsub) * t_stage + G * t_flush is smallest at G = sqrt(...)
  const int nkb = a.k_in > 128 ? 8 : (a.k_in > 64 ? 4 : 2), nkb_ = nkb;
  // vector-load variants need whole NKB-column groups and aligned rows; anything else takes the scalar-load path
  bool xv = a.k_in % nkb == 0 && a.ld_x % (xb ? nkb : (nkb >= 4 ? 4 : 2)) == 0;
  bool dv = a.n_out > 32 && a.ld_dy % 2 == 0 && a.ld_dy >= a.n_out + (a.n_out & 1);
  for (int jb = 0; jb < J.n_jobs; ++jb)
    for (int s = 0; s < a.n_seg; ++s) {
      xv = xv && ((uintptr_t)J.j[jb].x[s] % (xb ? 2 * nkb : 16) == 0);
      dv = dv && ((uintptr_t)J.j[jb].dy[s] % (db ? 4 : 8) == 0);
    }
  if (xb || db) {  // no other kernel can read bf16 rows: the layout must fit, loudly
    RSN_REQUIRE(mode == RSN_MMA_BF16, RSN_ERR_INVALID_ARGUMENT, "bf16 operand rows need mma_mode RSN_MMA_BF16");
    RSN_REQUIRE(xv && (!db || dv), RSN_ERR_UNSUPPORTED,
                "bf16 operand rows: k_in=%d ld_x=%d n_out=%d ld_dy=%d / alignment do not fit the vector-load layout",
                a.k_in, a.ld_x, a.n_out, a.ld_dy);
  }
  bf16 = bf16 && xv && (dv || (xb && !db));  // the bf16 variants exist for the vector-load layout only
  const int P = a.n_out <= 64 ? 1 : (a.n_out <= 128 ? 2 : 4);
  const int nsub = 4 / P;
  long long stages = 0;
  const int stage_pts = bf16 ? 16 : 2 * WG_PAIRS;
  for (int s = 0; s < a.n_seg; ++s) {
    const long long n_s = a.seg_begin[s + 1] - a.seg_begin[s];
    stages += (n_s + stage_pts - 1) / stage_pts;
  }
  const double t_stage = bf16 ? (a.n_out > 32 ? 2 : 1) * nkb_ * 32 / 1.9e9 * (mode == RSN_MMA_BF16X6 ? 6 : 2.5)  // 1 product: HBM-bound, ~2.5x the MFMA time
                              : WG_PAIRS * (a.n_out > 32 ? 2 : 1) * nkb_ * 64 / 2.1e9;
  const double t_flush = (double)a.n_out * a.k_in * 4.0 / 1.3e12 + 2e-8;
  long long grid = (long long)(sqrt((double)stages * t_stage / (nsub * t_flush)) + 0.5);  // workgroups per job
  if (grid > cached_cus / J.n_jobs) grid = cached_cus / J.n_jobs;
  if (grid < 1) grid = 1;
  grid *= J.n_jobs;  // workgroup i: job i % n_jobs
  hipStream_t st = (hipStream_t)stream;
  // split-bf16 over more than 128 output rows: operand rows staged through LDS (rsn_wgrad_x6s_kernel) where the layout allows
#ifdef WG_X6_STAGED
  bool staged = xv && dv && bf16 && mode == RSN_MMA_BF16X6 && a.n_out > 128 && nkb >= 4 && a.ld_x % 4 == 0 && a.ld_dy % 4 == 0;
  for (int jb = 0; jb < J.n_jobs && staged; ++jb)
    for (int s = 0; s < a.n_seg; ++s) {
      const long long n_s = a.seg_begin[s + 1] - a.seg_begin[s];
      staged = staged && (uintptr_t)J.j[jb].x[s] % 16 == 0 && (uintptr_t)J.j[jb].dy[s] % 16 == 0 &&
               n_s * a.ld_x * 4 < (1ll << 31) - (1ll << 27) && n_s * a.ld_dy * 4 < (1ll << 31) - (1ll << 27);
    }
  if (staged) {
    if (nkb == 8)
      hipLaunchKernelGGL((rsn_wgrad_x6s_kernel<8>), dim3((unsigned)grid), dim3(256), 0, st, J);
    else
      hipLaunchKernelGGL((rsn_wgrad_x6s_kernel<4>), dim3((unsigned)grid), dim3(256), 0, st, J);
    RSN_HIP(hipGetLastError());
    return RSN_OK;
  }
#endif
#ifdef WG_F32_STAGED
  {
    bool st32 = xv && dv && !bf16 && a.n_out > 128 && nkb >= 4 && a.ld_x % 4 == 0 && a.ld_dy % 4 == 0;
    for (int jb = 0; jb < J.n_jobs && st32; ++jb)
      for (int s = 0; s < a.n_seg; ++s) {
        const long long n_s = a.seg_begin[s + 1] - a.seg_begin[s];
        st32 = st32 && (uintptr_t)J.j[jb].x[s] % 16 == 0 && (uintptr_t)J.j[jb].dy[s] % 16 == 0 &&
               n_s * a.ld_x * 4 < (1ll << 31) - (1ll << 27) && n_s * a.ld_dy * 4 < (1ll << 31) - (1ll << 27);
      }
    if (st32) {
      if (nkb == 8)
        hipLaunchKernelGGL((rsn_wgrad_f32s_kernel<8>), dim3((unsigned)grid), dim3(256), 0, st, J);
      else
        hipLaunchKernelGGL((rsn_wgrad_f32s_kernel<4>), dim3((unsigned)grid), dim3(256), 0, st, J);
      RSN_HIP(hipGetLastError());
      return RSN_OK;
    }
  }
#endif
#define RSN_WG(NKBV)                                                                                           \
  do {                                                                                                         \
    if (xv && dv && bf16 && mode == RSN_MMA_BF16X6)                                                            \
      hipLaunchKernelGGL((rsn_wgrad_kernel<NKBV, true, true, 3>), dim3((unsigned)grid), dim3(256), 0, st, J);  \
    else if (bf16 && xb && db)                                                                                 \
      hipLaunchKernelGGL((rsn_wgrad_kernel<NKBV, true, true, 1, true, true>), dim3((unsigned)grid), dim3(256), 0, st, J); \
    else if (bf16 && db)                                                                                       \
      hipLaunchKernelGGL((rsn_wgrad_kernel<NKBV, true, true, 1, false, true>), dim3((unsigned)grid), dim3(256), 0, st, J); \
    else if (bf16 && xb && !dv)                                                                                \
      hipLaunchKernelGGL((rsn_wgrad_kernel<NKBV, true, false, 1, true, false>), dim3((unsigned)grid), dim3(256), 0, st, J); \
    else if (bf16 && xb)                                                                                       \
      { RSN_REQUIRE(false, RSN_ERR_UNSUPPORTED, "bf16 X rows with fp32 dY rows wider than 32 outputs"); }      \
    else if (xv && dv && bf16)                                                                                 \
      hipLaunchKernelGGL((rsn_wgrad_kernel<NKBV, true, true, 1>), dim3((unsigned)grid), dim3(256), 0, st, J);  \
    else if (xv && dv)                                                                                         \
      hipLaunchKernelGGL((rsn_wgrad_kernel<NKBV, true, true>), dim3((unsigned)grid), dim3(256), 0, st, J);     \
    else if (xv)                                                                                               \
      hipLaunchKernelGGL((rsn_wgrad_kernel<NKBV, true, false>), dim3((unsigned)grid), dim3(256), 0, st, J);    \
    else                                                                                                       \
      hipLaunchKernelGGL((rsn_wgrad_kernel<NKBV, false, false>), dim3((unsigned)grid), dim3(256), 0, st, J);   \
  } while (0)
  if (nkb == 8)
    RSN_WG(8);
  else if (nkb == 4)
    RSN_WG(4);
  else
    RSN_WG(2);
#undef RSN_WG
  RSN_HIP(hipGetLastError());
  return RSN_OK;
}

static int weight_grad_multi_impl(int32_t n_segments, const int64_t* n_points, const float* const* dy, int32_t ld_dy,
                                  int32_t n_out, const float* const* x, int32_t ld_x, int32_t k_in,
                                  const int32_t* col_map, float* dw, int32_t ld_dw, float* db, void* stream, int mode,
                                  const int32_t* const* n_dev = nullptr, const int32_t* per_count = nullptr,
                                  int operand_bf16 = 0);

extern "C" int rsn_weight_grad_multi(int32_t n_segments, const int64_t* n_points, const float* const* dy, int32_t ld_dy,
                                     int32_t n_out, const float* const* x, int32_t ld_x, int32_t k_in,
                                     const int32_t* col_map, float* dw, int32_t ld_dw, float* db, void* stream) {
  return weight_grad_multi_impl(n_segments, n_points, dy, ld_dy, n_out, x, ld_x, k_in, col_map, dw, ld_dw, db, stream, RSN_MMA_F32);
}

static int weight_grad_multi_impl(int32_t n_segments, const int64_t* n_points, const float* const* dy, int32_t ld_dy,
                                  int32_t n_out, const float* const* x, int32_t ld_x, int32_t k_in,
                                  const int32_t* col_map, float* dw, int32_t ld_dw, float* db, void* stream, int mode,
                                  const int32_t* const* n_dev, const int32_t* per_count, int operand_bf16) {
  RSN_REQUIRE(n_segments >= 0 && n_segments <= WG_MAX_SEG, RSN_ERR_INVALID_ARGUMENT, "n_segments=%d (at most %d)",
              n_segments, WG_MAX_SEG);
  RSN_REQUIRE(n_out >= 1 && n_out <= 256 && k_in >= 1 && k_in <= 256, RSN_ERR_INVALID_ARGUMENT,
              "n_out=%d k_in=%d (outputs up to 256 x 256)", n_out, k_in);
  RSN_REQUIRE(ld_dy >= n_out && ld_x >= k_in && ld_dw >= 1, RSN_ERR_INVALID_ARGUMENT, "leading dimensions too small");
  if (n_segments == 0) return RSN_OK;
  RSN_REQUIRE(n_points && dy && x && dw, RSN_ERR_INVALID_ARGUMENT, "a pointer is NULL");
  WGradJobs J = {};
  J.n_jobs = 1;
  WGradArgs& a = J.j[0];
  a.seg_begin[0] = 0;
  int ns = 0;
  for (int s = 0; s < n_segments; ++s) {
    RSN_REQUIRE(n_points[s] >= 0, RSN_ERR_INVALID_ARGUMENT, "n_points[%d]=%lld", s, (long long)n_points[s]);
    if (n_points[s] == 0) continue;
    RSN_REQUIRE(dy[s] && x[s], RSN_ERR_INVALID_ARGUMENT, "segment %d: a pointer is NULL", s);
    a.dy[ns] = dy[s];
    a.x[ns] = x[s];
    a.seg_begin[ns + 1] = a.seg_begin[ns] + n_points[s];
    a.n_dev[ns] = n_dev ? n_dev[s] : nullptr;
    a.per_count[ns] = (n_dev && n_dev[s] && per_count) ? per_count[s] : 1;
    RSN_REQUIRE(a.per_count[ns] >= 1, RSN_ERR_INVALID_ARGUMENT, "segment %d: per_count=%d", s, a.per_count[ns]);
    ++ns;
  }
  a.n_seg = ns;
  a.ld_dy = ld_dy; a.ld_x = ld_x; a.n_out = n_out; a.k_in = k_in;
  a.ld_dw = ld_dw; a.col_map = col_map; a.dw = dw; a.db = db;
  return wgrad_launch(J, stream, mode, operand_bf16);
}

extern "C" int rsn_weight_grad_multi_mode(int32_t n_segments, const int64_t* n_points, const float* const* dy,
                                          int32_t ld_dy, int32_t n_out, const float* const* x, int32_t ld_x, int32_t k_in,
                                          const int32_t* col_map, float* dw, int32_t ld_dw, float* db, int32_t mma_mode,
                                          void* stream) {
  RSN_REQUIRE(mma_mode >= RSN_MMA_F32 && mma_mode <= RSN_MMA_BF16, RSN_ERR_INVALID_ARGUMENT, "mma_mode %d", mma_mode);
  return weight_grad_multi_impl(n_segments, n_points, dy, ld_dy, n_out, x, ld_x, k_in, col_map, dw, ld_dw, db, stream,
                                mma_mode);
}

extern "C" int rsn_weight_grad_multi_dev(int32_t n_segments, const int64_t* n_points_max, const int32_t* const* n_dev,
                                         const int32_t* per_count, const float* const* dy, int32_t ld_dy, int32_t n_out,
                                         const float* const* x, int32_t ld_x, int32_t k_in, const int32_t* col_map,
                                         float* dw, int32_t ld_dw, float* db, int32_t mma_mode, int32_t operand_bf16,
                                         void* stream) {
  RSN_REQUIRE(mma_mode >= RSN_MMA_F32 && mma_mode <= RSN_MMA_BF16, RSN_ERR_INVALID_ARGUMENT, "mma_mode %d", mma_mode);
  RSN_REQUIRE(n_segments == 0 || (n_dev && per_count), RSN_ERR_INVALID_ARGUMENT, "n_dev / per_count is NULL");
  RSN_REQUIRE(operand_bf16 >= 0 && operand_bf16 <= 3, RSN_ERR_INVALID_ARGUMENT, "operand_bf16 %d", operand_bf16);
  return weight_grad_multi_impl(n_segments, n_points_max, dy, ld_dy, n_out, x, ld_x, k_in, col_map, dw, ld_dw, db, stream,
                                mma_mode, n_dev, per_count, operand_bf16);
}

// rsn_weight_grad_jobs: n_jobs reductions of one shape over the same segments in ONE launch (see WGradJobs).
extern "C" int rsn_weight_grad_jobs(int32_t n_segments, const int64_t* n_points_max, const int32_t* const* n_dev,
                                    const int32_t* per_count, int32_t n_jobs, const rsn_wgrad_job* jobs, int32_t ld_dy,
                                    int32_t n_out, int32_t ld_x, int32_t k_in, int32_t mma_mode, int32_t operand_bf16,
                                    void* stream) {
  RSN_REQUIRE(mma_mode >= RSN_MMA_F32 && mma_mode <= RSN_MMA_BF16, RSN_ERR_INVALID_ARGUMENT, "mma_mode %d", mma_mode);
  RSN_REQUIRE(operand_bf16 >= 0 && operand_bf16 <= 3, RSN_ERR_INVALID_ARGUMENT, "operand_bf16 %d", operand_bf16);
  RSN_REQUIRE(n_jobs >= 1 && n_jobs <= WG_MAX_JOBS && jobs, RSN_ERR_INVALID_ARGUMENT, "n_jobs=%d (1..%d)", n_jobs, WG_MAX_JOBS);
  RSN_REQUIRE(n_segments >= 0 && n_segments <= WG_MAX_SEG, RSN_ERR_INVALID_ARGUMENT, "n_segments=%d (at most %d)",
              n_segments, WG_MAX_SEG);
  RSN_REQUIRE(n_out >= 1 && n_out <= 256 && k_in >= 1 && k_in <= 256, RSN_ERR_INVALID_ARGUMENT,
              "n_out=%d k_in=%d (outputs up to 256 x 256)", n_out, k_in);
  RSN_REQUIRE(ld_dy >= n_out && ld_x >= k_in, RSN_ERR_INVALID_ARGUMENT, "leading dimensions too small");
  if (n_segments == 0) return RSN_OK;
  RSN_REQUIRE(n_points_max, RSN_ERR_INVALID_ARGUMENT, "n_points_max is NULL");
  WGradJobs J = {};
  J.n_jobs = n_jobs;
  for (int jb = 0; jb < n_jobs; ++jb) {
    const rsn_wgrad_job& q = jobs[jb];
    WGradArgs& a = J.j[jb];
    RSN_REQUIRE(q.dy && q.x && q.dw && q.ld_dw >= 1, RSN_ERR_INVALID_ARGUMENT, "job %d: a pointer is NULL / ld_dw=%d", jb, q.ld_dw);
    a.seg_begin[0] = 0;
    int ns = 0;
    for (int s = 0; s < n_segments; ++s) {
      RSN_REQUIRE(n_points_max[s] >= 0, RSN_ERR_INVALID_ARGUMENT, "n_points_max[%d]=%lld", s, (long long)n_points_max[s]);
      if (n_points_max[s] == 0) continue;
      RSN_REQUIRE(q.dy[s] && q.x[s], RSN_ERR_INVALID_ARGUMENT, "job %d segment %d: a pointer is NULL", jb, s);
      a.dy[ns] = q.dy[s];
      a.x[ns] = q.x[s];
      a.seg_begin[ns + 1] = a.seg_begin[ns] + n_points_max[s];
      a.n_dev[ns] = n_dev ? n_dev[s] : nullptr;
      a.per_count[ns] = (n_dev && n_dev[s] && per_count) ? per_count[s] : 1;
      RSN_REQUIRE(a.per_count[ns] >= 1, RSN_ERR_INVALID_ARGUMENT, "segment %d: per_count=%d", s, a.per_count[ns]);
      ++ns;
    }
    a.n_seg = ns;
    a.ld_dy = ld_dy; a.ld_x = ld_x; a.n_out = n_out; a.k_in = k_in;
    a.ld_dw = q.ld_dw; a.col_map = q.col_map; a.dw = q.dw; a.db = q.db;
  }
  if (J.j[0].n_seg == 0) return RSN_OK;
  return wgrad_launch(J, stream, mma_mode, operand_bf16);
}

extern "C" int rsn_weight_grad(int64_t n_points, const float* dy, int32_t ld_dy, int32_t n_out, const float* x,
                               int32_t ld_x, int32_t k_in, const int32_t* col_map, float* dw, int32_t ld_dw,
                               float* db, void* stream) {
  RSN_REQUIRE(n_points >= 0, RSN_ERR_INVALID_ARGUMENT, "n_points=%lld", (long long)n_points);
  const int64_t n = n_points;
  return rsn_weight_grad_multi(1, &n, &dy, ld_dy, n_out, &x, ld_x, k_in, col_map, dw, ld_dw, db, stream);
}
