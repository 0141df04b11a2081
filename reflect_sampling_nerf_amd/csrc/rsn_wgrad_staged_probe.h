// rsn_wgrad_staged_probe.h -- DIAGNOSTIC BUILDS ONLY (-DWG_X6_STAGED / -DWG_F32_STAGED through tools/_variant.py; rsn_common.h refuses the
// macros without -DRSN_DIAG_BUILD): the weight-gradient reductions with their operand rows STAGED THROUGH LDS.  Both kernels are
// correct (same tests as the product kernels) and neither is faster; they stay as the probes behind DESIGN 4.7's clock finding and
// profiles/r04_wgrad_x6.txt.  Included by rsn_wgrad.hip behind the kernels' shared definitions (WGradJobs, cvt2, f32x2w, u32x4w).
#pragma once

#ifdef WG_X6_STAGED
// ------------------------------------------------------------------------------------------------------------------------
// rsn_wgrad_x6s_kernel (DIAGNOSTIC BUILDS ONLY, -DWG_X6_STAGED: correct, measured, not adopted -- see the note at its end):
// the split-bf16 reduction for outputs of more than 128 rows (the trunk layers: nine tenths of the
// mode's weight-gradient time), operand rows STAGED THROUGH LDS by LDS-DMA instead of through registers.
//
// Why: the register kernel above keeps two stages of fp32 rows (160 registers) beside its 256 accumulators -- one stage
// (~1.5 us) of prefetch distance for a wave that is alone on its SIMD, and every wave of the workgroup fetches the whole X row
// itself (4 x the L1 traffic).  Here the four waves share one copy of a stage (16 points: 16 KiB of dY rows, 16 NKB/8 KiB of
// X rows) in a ring of WGS_NS LDS slots; each wave issues a quarter of a stage's DMA pieces (8 x 1 KiB), three stages
// (~10 us) ahead, at no register cost.  A stage is two scheduling regions with one workgroup barrier between them:
//   region A: MFMAs of the first half of the column pairs | splits of ALL later column pairs of this stage (fp32 from LDS)
//   barrier:  own pieces of stage s + 1 landed (counted vmcnt: stages s + 2, s + 3 stay in flight), everyone is done
//             reading stage s  ->  its slot takes stage s + WGS_NS
//   region B: MFMAs of the second half | splits of stage s + 1's dY rows and first column pair
// so no split stands in front of an MFMA, and sched_group_barrier pins 4-5 VALU behind every MFMA (a wave alone on its SIMD
// issues in order: what is not slotted between MFMAs is added to them).
// Rows beyond a segment's end read as ZEROS through the buffer descriptor's range check (the row offset travels in the
// VECTOR offset, the operand the hardware checks): the tail stage and the look-ahead past the last stage need no masks.
// Column permutation (free, undone at the flush): lane i owns columns 4 i .. 4 i + 3 (blocks 0..3) and 128 + 4 i .. (blocks
// 4..7): its fp32 values of one point are one / two conflict-free ds_read_b128.
#define WGS_NS 4
#ifndef WGS_VALU_A
#define WGS_VALU_A 5
#endif
#ifndef WGS_VALU_B
#define WGS_VALU_B 4
#endif
__device__ __forceinline__ void wgs_dma16(const u32x4w rs, unsigned voff, unsigned lds_dst) {
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, 0 offen lds\n\ts_mov_b32 m0, %0"
               : "=&s"(keep)
               : "v"(voff), "s"(rs), "s"(lds_dst)
               : "memory");
}
template <int N>
__device__ __forceinline__ void wgs_sync() {  // own DMA pieces but the N youngest have landed, own LDS reads returned; workgroup barrier
  asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)\n\ts_barrier" ::"n"(N) : "memory");
}

template <int NKB>
__global__ __launch_bounds__(256) void rsn_wgrad_x6s_kernel(const WGradJobs J) {
  static_assert(NKB == 8 || NKB == 4, "column blocks: 8 (k_in <= 256) or 4 (k_in <= 128)");
  constexpr int XROW = NKB * 128, XOFF = 16 * 1024, SLOT = XOFF + 16 * XROW, NPR = NKB / 2, NPA = NPR / 2;
  constexpr int NDX = NKB == 8 ? 4 : 2, NDMA = 4 + NDX;  // DMA pieces per wave and stage
  __shared__ __attribute__((aligned(1024))) char smem[WGS_NS * SLOT];
  const int n_jobs = J.n_jobs;
  const WGradArgs& a = J.j[blockIdx.x % n_jobs];  // workgroup-uniform
  const int lane = threadIdx.x & 63;
  const int wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int i = lane & 31, h = lane >> 5;
  const int nb0 = wid * 2;  // this wave's two 32-row output blocks (rows 64 wid + 2 i, + 1 in lane i)
  const long long G = gridDim.x / n_jobs, g = blockIdx.x / n_jobs;
  const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) char*)smem;

  f32x16 acc[2][NKB];
#pragma unroll
  for (int t = 0; t < 2; ++t)
#pragma unroll
    for (int kb = 0; kb < NKB; ++kb)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[t][kb][r] = 0.0f;
  float bsum[2] = {0.0f, 0.0f};

  // what this lane reads of a slot: rows 2 i, 2 i + 1 of the wave's 64 (clamped into the live rows: duplicates are never
  // flushed) and its four (eight) columns, of the points 8 h .. 8 h + 7
  const int n_even = a.n_out + (a.n_out & 1);
  int c0 = nb0 * 32 + 2 * i;
  c0 = c0 < n_even - 2 ? c0 : n_even - 2;
  const unsigned rd_dy = (unsigned)(8 * h * 1024 + c0 * 4);
  const unsigned rd_x = (unsigned)(XOFF + 8 * h * XROW + 16 * i);

  // one quarter of split2: the point pair (2 q, 2 q + 1) of two neighbouring rows / columns
  auto split_q = [&](u32x4w (&o0)[3], u32x4w (&o1)[3], int q, const f32x2w xa, const f32x2w xb) {
    const unsigned HI = 0xffff0000u;
#ifdef WGS_NO_SPLIT  // timing ablation (wrong results): the fp32 bits as they are
    o0[0][q] = __float_as_uint(xa[0]); o1[0][q] = __float_as_uint(xa[1]);
    o0[1][q] = __float_as_uint(xb[0]); o1[1][q] = __float_as_uint(xb[1]);
    o0[2][q] = __float_as_uint(xa[0]); o1[2][q] = __float_as_uint(xb[1]);
    return;
#endif
    const unsigned h0 = cvt2(xa[0], xb[0]), h1 = cvt2(xa[1], xb[1]);
    const f32x2w ra = xa - f32x2w{__uint_as_float(h0 << 16), __uint_as_float(h1 << 16)};
    const f32x2w rb = xb - f32x2w{__uint_as_float(h0 & HI), __uint_as_float(h1 & HI)};
    const unsigned m0 = cvt2(ra[0], rb[0]), m1 = cvt2(ra[1], rb[1]);
    const f32x2w sa = ra - f32x2w{__uint_as_float(m0 << 16), __uint_as_float(m1 << 16)};
    const f32x2w sb = rb - f32x2w{__uint_as_float(m0 & HI), __uint_as_float(m1 & HI)};
    o0[0][q] = h0; o1[0][q] = h1;
    o0[1][q] = m0; o1[1][q] = m1;
    o0[2][q] = cvt2(sa[0], sb[0]); o1[2][q] = cvt2(sa[1], sb[1]);
  };
  // the 12 MFMAs of one column block: six products per row block, the two (independent) row blocks alternating
  auto mma12 = [&](int kb, const u32x4w (&a0)[3], const u32x4w (&a1)[3], const u32x4w (&bw)[3]) {
#ifdef WGS_NO_MFMA  // timing ablation (wrong results): the operands are formed, nothing is multiplied
    asm volatile("" ::"v"(a0[0]), "v"(a0[1]), "v"(a0[2]), "v"(a1[0]), "v"(a1[1]), "v"(a1[2]), "v"(bw[0]), "v"(bw[1]), "v"(bw[2]));
    return;
#endif
    constexpr int ia[6] = {2, 1, 0, 1, 0, 0}, ib[6] = {0, 1, 2, 0, 1, 0};  // lo x hi | mid x mid | hi x lo | mid x hi | hi x mid | hi x hi
    f32x16 c0 = acc[0][kb], c1 = acc[1][kb];
#pragma unroll
    for (int k = 0; k < 6; ++k) {
      const bf16x8 bb = __builtin_bit_cast(bf16x8, bw[ib[k]]);
      c0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a0[ia[k]]), bb, c0, 0, 0, 0);
      c1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a1[ia[k]]), bb, c1, 0, 0, 0);
    }
    acc[0][kb] = c0;
    acc[1][kb] = c1;
  };

  // Operands carried from stage to stage: the split dY rows and first column pair, and the fp32 columns 2, 3 of the stage about to
  // multiply.  Two sets, swapped by the twofold-unrolled stage loop (no register copies).
  struct Carry {
    u32x4w xa[2][3], xb01[2][3];
    f32x2w x23[8];
  };
  Carry C0, C1;
#ifndef WGS_DMA
  u32x4w R[2][NDMA];  // staging registers: a wave's pieces of two stages on their way global memory -> LDS
#endif

  long long vprefix = 0;  // stages of the segments in front of this one
  bool any = false;
#pragma unroll 1
  for (int s = 0; s < a.n_seg; ++s) {
    long long n_s = a.seg_begin[s + 1] - a.seg_begin[s];
    if (a.n_dev[s]) {  // the reflected-ray count lives on the device (no host read in the training step): uniform
      const long long nd = (long long)(*a.n_dev[s]) * a.per_count[s];
      n_s = nd < n_s ? (nd > 0 ? nd : 0) : n_s;
    }
    const long long n_st = (n_s + 15) / 16;  // stages, the last one partly beyond the segment (zeros)
    const long long first = ((g - vprefix) % G + G) % G;  // this workgroup's first stage of the segment
    const int cnt = (int)(first < n_st ? (n_st - first + G - 1) / G : 0);
    vprefix += n_st;
    if (cnt == 0) continue;  // workgroup-uniform
    any = true;
    // descriptors over the segment's live rows; the row offset of a piece travels in the lane offset (range-checked)
    auto vsharp = [&](const void* base, long long bytes) {  // raw buffer V#, built by hand so that it can be an inline-asm operand
      const unsigned long long b = (unsigned long long)(size_t)base;
      return u32x4w{(unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)b),
                    (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)((b >> 32) & 0xffffull)),
                    (unsigned)__builtin_amdgcn_readfirstlane((int)bytes), 0x00020000u};
    };
    const u32x4w rs_dy = vsharp(a.dy[s], n_s * a.ld_dy * 4), rs_x = vsharp(a.x[s], n_s * a.ld_x * 4);
    unsigned vdy[4], vx[NDX];
#pragma unroll
    for (int q = 0; q < 4; ++q) vdy[q] = (unsigned)((first * 16 + 4 * wid + q) * a.ld_dy * 4) + 16u * lane;
#pragma unroll
    for (int q = 0; q < NDX; ++q)
      vx[q] = NKB == 8 ? (unsigned)((first * 16 + 4 * wid + q) * a.ld_x * 4) + 16u * lane
                       : (unsigned)((first * 16 + 4 * wid + 2 * q + h) * a.ld_x * 4) + 16u * i;
    const unsigned adv_dy = (unsigned)(G * 16 * a.ld_dy * 4), adv_x = (unsigned)(G * 16 * a.ld_x * 4);
    // piece k (0 .. NDMA - 1) of this wave's quarter of the next stage -> slot; its offset moves on one stage
    auto issue1 = [&](int slot, int k) {
      const unsigned d = lds0 + (unsigned)slot * SLOT;
      if (k < 4) {
        wgs_dma16(rs_dy, vdy[k], d + (unsigned)((4 * wid + k) * 1024));
        vdy[k] += adv_dy;
      } else {
        const int q = k - 4;
        wgs_dma16(rs_x, vx[q], d + XOFF + (unsigned)((4 * wid + (NKB == 8 ? q : 2 * q)) * XROW));
        vx[q] += adv_x;
      }
    };
#ifndef WGS_DMA
    // Register staging instead of LDS-DMA (an LDS-DMA piece costs the issuing wave ~175 cycles wherever it is placed -- measured,
    // profiles/r04_wgrad_x6.txt -- and a wave that is alone on its SIMD pays that in MFMA time): piece k of stage st + 3 is
    // LOADED during stage st, WRITTEN to its slot during stage st + 1, visible behind the barrier of stage st + 2, read from
    // stage st + 2's second half on.
    const __amdgpu_buffer_rsrc_t rb_dy = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.dy[s]), 0, (int)(n_s * a.ld_dy * 4), 0x00020000);
    const __amdgpu_buffer_rsrc_t rb_x = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.x[s]), 0, (int)(n_s * a.ld_x * 4), 0x00020000);
    auto load1 = [&](int buf, int k) {
      if (k < 4) {
        R[buf][k] = __builtin_amdgcn_raw_buffer_load_b128(rb_dy, vdy[k], 0, 0);
        vdy[k] += adv_dy;
      } else {
        R[buf][k] = __builtin_amdgcn_raw_buffer_load_b128(rb_x, vx[k - 4], 0, 0);
        vx[k - 4] += adv_x;
      }
    };
    auto write1 = [&](int slot, int buf, int k) {
      const unsigned d = (unsigned)slot * SLOT + 16u * lane +
                         (k < 4 ? (unsigned)((4 * wid + k) * 1024) : XOFF + (unsigned)((4 * wid + (NKB == 8 ? k - 4 : 2 * (k - 4))) * XROW));
      *reinterpret_cast<u32x4w*>(smem + d) = R[buf][k];
    };
#endif
    auto read_head = [&](unsigned slot_off, f32x2w (&dyv)[8], float4 (&xv)[8]) {  // dY rows and columns 0..3 of the stage in this slot
#pragma unroll
      for (int p = 0; p < 8; ++p) dyv[p] = *reinterpret_cast<const f32x2w*>(smem + slot_off + rd_dy + p * 1024);
#pragma unroll
      for (int p = 0; p < 8; ++p) xv[p] = *reinterpret_cast<const float4*>(smem + slot_off + rd_x + p * XROW);
    };
    auto head_unit = [&](Carry& N, int u, const f32x2w (&dyv)[8], const float4 (&xv)[8]) {  // unit 0..3: rows, 4..7: column pair 0
      const int q = u & 3;
      if (u < 4) {
        bsum[0] += dyv[2 * q][0] + dyv[2 * q + 1][0];
        bsum[1] += dyv[2 * q][1] + dyv[2 * q + 1][1];
        split_q(N.xa[0], N.xa[1], q, dyv[2 * q], dyv[2 * q + 1]);
      } else {
        split_q(N.xb01[0], N.xb01[1], q, f32x2w{xv[2 * q].x, xv[2 * q].y}, f32x2w{xv[2 * q + 1].x, xv[2 * q + 1].y});
        N.x23[2 * q] = f32x2w{xv[2 * q].z, xv[2 * q].w};
        N.x23[2 * q + 1] = f32x2w{xv[2 * q + 1].z, xv[2 * q + 1].w};
      }
    };
#ifdef WGS_DMA
    wgs_sync<0>();  // (the previous segment's look-ahead pieces and reads are done: its slots are free)
#pragma unroll
    for (int k = 0; k < WGS_NS - 1; ++k)
#pragma unroll
      for (int q = 0; q < NDMA; ++q) issue1(k, q);
    wgs_sync<(WGS_NS - 2) * NDMA>();  // stage 0 has landed
#else
    wgs_sync<0>();  // (everyone is done reading the previous segment's slots)
#pragma unroll
    for (int q = 0; q < NDMA; ++q) load1(0, q);
#pragma unroll
    for (int q = 0; q < NDMA; ++q) load1(1, q);
#pragma unroll
    for (int q = 0; q < NDMA; ++q) write1(0, 0, q);
#pragma unroll
    for (int q = 0; q < NDMA; ++q) write1(1, 1, q);
#pragma unroll
    for (int q = 0; q < NDMA; ++q) load1(0, q);  // stage 2
    wgs_sync<0>();
#endif
    {
      f32x2w dyv[8];
      float4 xv[8];
      read_head(0, dyv, xv);
#pragma unroll
      for (int u = 0; u < 8; ++u) head_unit(C0, u, dyv, xv);
    }
    __builtin_amdgcn_sched_barrier(0);

    // One stage: NKB sub-blocks of 12 MFMAs (one column block each).  Behind the MFMAs of a sub-block: its share of the stage's
    // split work (the later column pairs of this stage, then stage st + 1's rows and first pair: "units" of 18 VALU), pinned
    // 1 MFMA : WGS_VALU; then its share of the DMA pieces of stage st + 3 (an LDS-DMA piece costs the issuing wave ~180 cycles
    // back to back, ~60 between MFMAs).
    auto stage = [&](Carry& P, Carry& N, int st, int par) {  // par = st & 1 as a compile-time constant
      constexpr int NSB = NKB, UA = 4 * (NPR - 1), NU = UA + 8, RB = (UA * NSB / NU > 0 ? UA * NSB / NU - 1 : 0);
      const unsigned so = (unsigned)(st & (WGS_NS - 1)) * SLOT, so1 = (unsigned)((st + 1) & (WGS_NS - 1)) * SLOT;
      const int slot3 = (st + 3) & (WGS_NS - 1);
#ifdef WGS_DMA
#ifndef WGS_NO_DMA
      wgs_sync<NDMA>();  // stage st + 1 is in LDS for everyone (st + 2 may be in flight); everyone is done reading stage st - 1
#endif
#else
      asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");  // stage st + 1 is in LDS for everyone, stage st - 1 has been read
#endif
      u32x4w bq[NPR][2][3];
      float4 xh[8];
      f32x2w dyv[8];
      float4 xv[8];
#pragma unroll
      for (int k = 0; k < 3; ++k) { bq[0][0][k] = P.xb01[0][k]; bq[0][1][k] = P.xb01[1][k]; }
#pragma unroll
      for (int b = 0; b < NSB; ++b) {
        int n_ds = 0;
        if (b == 0 && NKB == 8) {
#pragma unroll
          for (int p = 0; p < 8; ++p) xh[p] = *reinterpret_cast<const float4*>(smem + so + rd_x + 512 + p * XROW);
          n_ds += 8;
        }
        if (b == RB) {
          read_head(so1, dyv, xv);
          n_ds += 16;
        }
#pragma unroll
        for (int u = b * NU / NSB; u < (b + 1) * NU / NSB; ++u) {
          if (u < UA) {
            const int j = 1 + u / 4, q = u & 3;
            if (j == 1) split_q(bq[1][0], bq[1][1], q, P.x23[2 * q], P.x23[2 * q + 1]);
            else if (j == 2) split_q(bq[j][0], bq[j][1], q, f32x2w{xh[2 * q].x, xh[2 * q].y}, f32x2w{xh[2 * q + 1].x, xh[2 * q + 1].y});
            else split_q(bq[j][0], bq[j][1], q, f32x2w{xh[2 * q].z, xh[2 * q].w}, f32x2w{xh[2 * q + 1].z, xh[2 * q + 1].w});
          } else {
            head_unit(N, u - UA, dyv, xv);
          }
        }
#ifndef WGS_DMA
        const int k0 = b * NDMA / NSB, k1 = (b + 1) * NDMA / NSB;
#ifndef WGS_NO_ISSUE
#pragma unroll
        for (int k = k0; k < k1; ++k) {
          write1((st + 2) & (WGS_NS - 1), par, k);  // stage st + 2 (loaded during stage st - 1) -> its slot
          load1(par ^ 1, k);                        // stage st + 3
        }
#endif
#endif
        mma12(b, P.xa[0], P.xa[1], bq[b >> 1][b & 1]);
#if !defined(WGS_DMA) && !defined(WGS_NO_ISSUE)
        if (k1 - k0 == 1) { __builtin_amdgcn_sched_group_barrier(0x200, 1, 0); __builtin_amdgcn_sched_group_barrier(0x020, 1, 0); }
        if (k1 - k0 == 2) { __builtin_amdgcn_sched_group_barrier(0x200, 2, 0); __builtin_amdgcn_sched_group_barrier(0x020, 2, 0); }
#endif
        // the LDS reads first
        if (n_ds == 8) __builtin_amdgcn_sched_group_barrier(0x100, 8, 0);
        if (n_ds == 16) __builtin_amdgcn_sched_group_barrier(0x100, 16, 0);
        if (n_ds == 24) __builtin_amdgcn_sched_group_barrier(0x100, 24, 0);
#pragma unroll
        for (int m = 0; m < 12; ++m) {
          __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
          __builtin_amdgcn_sched_group_barrier(0x002, NKB == 8 ? WGS_VALU_B : WGS_VALU_A, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
#ifdef WGS_DMA
#if !defined(WGS_NO_DMA) && !defined(WGS_NO_ISSUE)
#pragma unroll
        for (int k = b * NDMA / NSB; k < (b + 1) * NDMA / NSB; ++k) issue1(slot3, k);
#endif
        __builtin_amdgcn_sched_barrier(0);
#endif
      }
    };
    int st = 0;
#pragma unroll 1
    for (; st + 1 < cnt; st += 2) {
      stage(C0, C1, st, 0);
      stage(C1, C0, st + 1, 1);
    }
    if (st < cnt) {
      stage(C0, C1, st, 0);
    }
  }
  if (!any) return;  // workgroup-uniform
  wgs_sync<0>();     // every look-ahead piece has landed, every read is done: the ring's memory becomes the flush tiles
#ifdef RSN_DIAG_WG_NO_FLUSH  // timing ablation (wrong results): what the atomic flush costs
  if (acc[0][0][0] != 12345.678f) return;
#endif

  // flush (as rsn_wgrad_kernel's, with this kernel's column permutation): C/D layout col = lane & 31 (input-column slot),
  // row = (r & 3) + 8 (r >> 2) + 4 h (output-row slot)
  float* trw = reinterpret_cast<float*>(smem) + (wid * 2 + h) * (NKB * 32);
  int cdst[NKB];
#pragma unroll
  for (int kb = 0; kb < NKB; ++kb) {
    const int k = kb * 32 + i;
    cdst[kb] = -1;
    if (k < a.k_in) cdst[kb] = a.col_map ? a.col_map[k] : k;
  }
  const __amdgpu_buffer_rsrc_t rdw = __builtin_amdgcn_make_buffer_rsrc(a.dw, 0, a.n_out * a.ld_dw * 4, 0x00020000);
  unsigned vdw[NKB];
#pragma unroll
  for (int kb = 0; kb < NKB; ++kb) vdw[kb] = cdst[kb] >= 0 ? (unsigned)((8 * h * a.ld_dw + cdst[kb]) * 4) : 0x40000000u;
#pragma unroll
  for (int t = 0; t < 2; ++t) {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int slot0 = (r & 3) + 8 * (r >> 2);  // + 4 h: in vdw
      const int n0 = nb0 * 32 + 2 * slot0 + t;
#pragma unroll
      for (int kb = 0; kb < NKB; ++kb) trw[(kb < 4 ? 0 : 128) + 4 * i + (kb & 3)] = acc[t][kb][r];
#pragma unroll
      for (int kb = 0; kb < NKB; ++kb) {
        const float v = trw[kb * 32 + i];
        __builtin_amdgcn_raw_ptr_buffer_atomic_fadd_f32(v, rdw, vdw[kb] + (unsigned)(n0 * a.ld_dw * 4), 0u, 0);
      }
    }
    if (a.db) {
      const float v = bsum[t] + __shfl_xor(bsum[t], 32, 64);
      const int n = nb0 * 32 + 2 * i + t;
      if (h == 0 && n < a.n_out) atomicAdd(&a.db[n], v);
    }
  }
}
// What the staged kernel showed (256 x 256 over 524,288 points, one MI355X; profiles/r04_wgrad_x6.txt):
//   register kernel above 449 us | staged, LDS-DMA 457-468 us | staged, loads + ds_write (default of this block) 469 us
//   staged without the operand traffic 372 us, without splits 410 us, MFMAs + LDS reads alone 271 us, flush ~45 us: the parts ADD --
//   and rocprofv3 says why: the kernels need about the same number of CYCLES with and without the operand traffic (683 K / 667 K),
//   but the chip clocks 1.64 GHz with it and 1.90 GHz without (register kernel: 615 K cycles at 1.55 GHz).  At bf16 MFMA rates the
//   chip is power-limited: what a part costs is its energy, not its issue slots, and no placement hides it.
#endif  // WG_X6_STAGED

// ------------------------------------------------------------------------------------------------------------------------
// rsn_wgrad_f32s_kernel (DIAGNOSTIC BUILDS ONLY, -DWG_F32_STAGED: correct, measured, not adopted): the EXACT-fp32 reduction for
// outputs of more than 128 rows with the operand rows staged through LDS.
// Hypothesis: the exact kernel is not power-limited (2.3 GHz), so what its per-wave loads cost (12 %: 246 us without them against
// 281, section header) would be matrix-pipe issue time -- 24 vector-memory instructions per 64 MFMAs and wave, because every wave
// fetches the whole X row for itself.  Here a stage (8 points: 8 KiB of
// dY rows + 8 NKB/8 KiB of X rows) exists once per workgroup in a ring of 4 LDS slots: per stage a wave issues 3-4 global loads
// (its quarter of the rows, three stages ahead, through a two-stage register ring), 3-4 ds_write_b128 and 12 conflict-free LDS
// reads, and meets ONE workgroup barrier.  Same MFMA shape, fragment meaning and flush as rsn_wgrad_kernel<.,true,true,0>
// (lane i: rows 2 i, 2 i + 1 of the wave's 64; columns 4 i .. 4 i + 3 and 128 + 4 i .. -- a free permutation, undone at the flush);
// rows beyond a segment's end read as zeros through the descriptor's range check (row offset in the VECTOR offset).
#ifdef WG_F32_STAGED
template <int NKB>
__global__ __launch_bounds__(256) void rsn_wgrad_f32s_kernel(const WGradJobs J) {
  static_assert(NKB == 8 || NKB == 4, "column blocks: 8 (k_in <= 256) or 4 (k_in <= 128)");
  constexpr int XROW = NKB * 128, XOFF = 8 * 1024, SLOT = XOFF + 8 * XROW, NS = 4;
  constexpr int NDX = NKB == 8 ? 2 : 1, NLD = 2 + NDX;  // 1 KiB pieces per wave and stage
  __shared__ __attribute__((aligned(1024))) char smem[NS * SLOT];
  const int n_jobs = J.n_jobs;
  const WGradArgs& a = J.j[blockIdx.x % n_jobs];  // workgroup-uniform
  const int lane = threadIdx.x & 63;
  const int wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int i = lane & 31, h = lane >> 5;
  const int nb0 = wid * 2;
  const long long G = gridDim.x / n_jobs, g = blockIdx.x / n_jobs;

  f32x16 acc[2][NKB];
#pragma unroll
  for (int t = 0; t < 2; ++t)
#pragma unroll
    for (int kb = 0; kb < NKB; ++kb)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[t][kb][r] = 0.0f;
  float bsum[2] = {0.0f, 0.0f};

  const int n_even = a.n_out + (a.n_out & 1);
  int c0 = nb0 * 32 + 2 * i;
  c0 = c0 < n_even - 2 ? c0 : n_even - 2;  // clamped into the live rows: duplicates are never flushed
  // the lane's point of pair p is 2 p + h
  const unsigned rd_dy = (unsigned)(h * 1024 + c0 * 4);
  const unsigned rd_x = (unsigned)(XOFF + h * XROW + 16 * i);

  struct Frag {
    f32x2w a[4];     // [pair]: rows 2 i, 2 i + 1
    float4 b[4][NKB / 4];
  };
  Frag F0, F1;
  u32x4w R[2][NLD];  // a wave's pieces of two stages on their way global memory -> LDS

  long long vprefix = 0;
  bool any = false;
#pragma unroll 1
  for (int s = 0; s < a.n_seg; ++s) {
    long long n_s = a.seg_begin[s + 1] - a.seg_begin[s];
    if (a.n_dev[s]) {  // device-side row count (no host read in the training step): uniform
      const long long nd = (long long)(*a.n_dev[s]) * a.per_count[s];
      n_s = nd < n_s ? (nd > 0 ? nd : 0) : n_s;
    }
    const long long n_st = (n_s + 7) / 8;
    const long long first = ((g - vprefix) % G + G) % G;
    const int cnt = (int)(first < n_st ? (n_st - first + G - 1) / G : 0);
    vprefix += n_st;
    if (cnt == 0) continue;  // workgroup-uniform
    any = true;
    const __amdgpu_buffer_rsrc_t rb_dy = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.dy[s]), 0, (int)(n_s * a.ld_dy * 4), 0x00020000);
    const __amdgpu_buffer_rsrc_t rb_x = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.x[s]), 0, (int)(n_s * a.ld_x * 4), 0x00020000);
    unsigned vdy[2], vx[NDX];
#pragma unroll
    for (int q = 0; q < 2; ++q) vdy[q] = (unsigned)((first * 8 + 2 * wid + q) * a.ld_dy * 4) + 16u * lane;
#pragma unroll
    for (int q = 0; q < NDX; ++q)
      vx[q] = NKB == 8 ? (unsigned)((first * 8 + 2 * wid + q) * a.ld_x * 4) + 16u * lane
                       : (unsigned)((first * 8 + 2 * wid + h) * a.ld_x * 4) + 16u * i;
    const unsigned adv_dy = (unsigned)(G * 8 * a.ld_dy * 4), adv_x = (unsigned)(G * 8 * a.ld_x * 4);
    auto load1 = [&](int buf, int k) {
      if (k < 2) {
        R[buf][k] = __builtin_amdgcn_raw_buffer_load_b128(rb_dy, vdy[k], 0, 0);
        vdy[k] += adv_dy;
      } else {
        R[buf][k] = __builtin_amdgcn_raw_buffer_load_b128(rb_x, vx[k - 2], 0, 0);
        vx[k - 2] += adv_x;
      }
    };
    auto write1 = [&](int slot, int buf, int k) {
      const unsigned d = (unsigned)slot * SLOT + 16u * lane +
                         (k < 2 ? (unsigned)((2 * wid + k) * 1024) : XOFF + (unsigned)((2 * wid + (k - 2)) * XROW));
      *reinterpret_cast<u32x4w*>(smem + d) = R[buf][k];
    };
    auto read_frag = [&](Frag& F, unsigned so) {
#pragma unroll
      for (int p = 0; p < 4; ++p) {
        F.a[p] = *reinterpret_cast<const f32x2w*>(smem + so + rd_dy + p * 2048);
#pragma unroll
        for (int q = 0; q < NKB / 4; ++q) F.b[p][q] = *reinterpret_cast<const float4*>(smem + so + rd_x + p * 2 * XROW + q * 512);
      }
    };
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");  // (everyone is done reading the previous segment's slots)
#pragma unroll
    for (int q = 0; q < NLD; ++q) load1(0, q);
#pragma unroll
    for (int q = 0; q < NLD; ++q) load1(1, q);
#pragma unroll
    for (int q = 0; q < NLD; ++q) write1(0, 0, q);
#pragma unroll
    for (int q = 0; q < NLD; ++q) write1(1, 1, q);
#pragma unroll
    for (int q = 0; q < NLD; ++q) load1(0, q);  // stage 2
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    read_frag(F0, 0);

    // stage st: multiplies the fragments read during stage st - 1, reads stage st + 1's, writes stage st + 2's pieces (loaded during
    // stage st - 1) to their slot and loads stage st + 3's
    auto stage = [&](Frag& P, Frag& N, int st, int par) {
      asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");  // stage st + 1 is in LDS for everyone, stage st - 2's slot is free
      read_frag(N, (unsigned)((st + 1) & (NS - 1)) * SLOT);
#pragma unroll
      for (int k = 0; k < NLD; ++k) {
        write1((st + 2) & (NS - 1), par, k);
        load1(par ^ 1, k);
      }
#pragma unroll
      for (int p = 0; p < 4; ++p) {
        bsum[0] += P.a[p][0];
        bsum[1] += P.a[p][1];
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
          for (int kb = 0; kb < NKB; ++kb) {
            const float4 bv = P.b[p][kb >> 2];
            const float b = (kb & 3) == 0 ? bv.x : (kb & 3) == 1 ? bv.y : (kb & 3) == 2 ? bv.z : bv.w;
            acc[t][kb] = __builtin_amdgcn_mfma_f32_32x32x2f32(P.a[p][t], b, acc[t][kb], 0, 0, 0);
          }
      }
      constexpr int NDS = 4 * (1 + NKB / 4);
#pragma unroll
      for (int m = 0; m < NDS + 2 * NLD; ++m) {  // one memory instruction behind each of the first MFMAs
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        if (m < NDS) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
        else if (m < NDS + NLD) __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);
        else __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
      }
      __builtin_amdgcn_sched_barrier(0);
    };
    int st = 0;
#pragma unroll 1
    for (; st + 1 < cnt; st += 2) {
      stage(F0, F1, st, 0);
      stage(F1, F0, st + 1, 1);
    }
    if (st < cnt) stage(F0, F1, st, 0);
  }
  if (!any) return;  // workgroup-uniform
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");  // the ring's memory becomes the flush tiles
#ifdef RSN_DIAG_WG_NO_FLUSH  // timing ablation (wrong results): what the atomic flush costs
  if (acc[0][0][0] != 12345.678f) return;
#endif
  float* trw = reinterpret_cast<float*>(smem) + (wid * 2 + h) * (NKB * 32);
  int cdst[NKB];
#pragma unroll
  for (int kb = 0; kb < NKB; ++kb) {
    const int k = kb * 32 + i;
    cdst[kb] = -1;
    if (k < a.k_in) cdst[kb] = a.col_map ? a.col_map[k] : k;
  }
  const __amdgpu_buffer_rsrc_t rdw = __builtin_amdgcn_make_buffer_rsrc(a.dw, 0, a.n_out * a.ld_dw * 4, 0x00020000);
  unsigned vdw[NKB];
#pragma unroll
  for (int kb = 0; kb < NKB; ++kb) vdw[kb] = cdst[kb] >= 0 ? (unsigned)((8 * h * a.ld_dw + cdst[kb]) * 4) : 0x40000000u;
#pragma unroll
  for (int t = 0; t < 2; ++t) {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int slot0 = (r & 3) + 8 * (r >> 2);  // + 4 h: in vdw
      const int n0 = nb0 * 32 + 2 * slot0 + t;
#pragma unroll
      for (int kb = 0; kb < NKB; ++kb) trw[(kb < 4 ? 0 : 128) + 4 * i + (kb & 3)] = acc[t][kb][r];
#pragma unroll
      for (int kb = 0; kb < NKB; ++kb) {
        const float v = trw[kb * 32 + i];
        __builtin_amdgcn_raw_ptr_buffer_atomic_fadd_f32(v, rdw, vdw[kb] + (unsigned)(n0 * a.ld_dw * 4), 0u, 0);
      }
    }
    if (a.db) {
      const float v = bsum[t] + __shfl_xor(bsum[t], 32, 64);
      const int n = nb0 * 32 + 2 * i + t;
      if (h == 0 && n < a.n_out) atomicAdd(&a.db[n], v);
    }
  }
}
// Measured (profiles/r04_wgrad_x6.txt, section 4): 608-614 us against the register kernel's 604 us per 256 x 256 x 524,288-point
// reduction, 359 against 343-352 us at 256 x 104: a sixth of the vector-memory instructions, the same time.  Not adopted.
#endif  // WG_F32_STAGED
