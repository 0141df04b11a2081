// rsn_mfma.h -- device helpers shared by the forward (rsn_field.hip) and backward (rsn_field_bwd.hip) field kernels:
// the lane-local MFMA K loop over packed weight segments and the accumulator <-> LDS-slab epilogues.
// See rsn_field.hip for the layout argument (why activations never cross lanes).
#pragma once
#include "rsn_common.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));

#define RSN_MODE_FRUSTUM 0
#define RSN_MODE_INF 1
#define RSN_MODE_GAUSS 2
#define RSN_MODE_EMB 3

// ------------------------------------------------------------------------------------------------
// small math, written to follow the torch op order of the reference (contraction off: -ffp-contract=off)
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ float softplus_f(float x) { return x > 20.0f ? x : log1pf(expf(x)); }
__device__ __forceinline__ float sigmoid_f(float x) { return 1.0f / (1.0f + expf(-x)); }

// sin / cos of an fp32 argument up to ~1e6 (the IPE reaches 2*pi*2*2^16 = 8.2e5), <= 1.7 ulp -- the quality of ocml's
// sinf (1.57 ulp measured over the same range) at ~1/5 of its instruction count.  Instead of the branchy Payne-Hanek
// path the argument is reduced by multiples of pi/2 with a three-constant Cody-Waite scheme on FUSED multiply-adds:
// pi/2 = C1 + C2 + C3 with C1 = fl32(pi/2); q <= 2^20 is an integer and C1 a multiple of 2^-23, so a - q*C1 is a
// multiple of 2^-23 below 1 in magnitude and the first fma is EXACT; the second rounds once (2^-25), the third term is
// 1e-9.  Reduced argument within 3.1e-8 of the true one (measured over 4e6 arguments up to 8.3e5: max |r| 0.87 --
// the fp32 product a*(2/pi) may pick the neighbouring quadrant, which costs nothing -- and 9.8e-8 / 1.65 ulp on the
// result).  A degree-7/8 minimax pair is evaluated on the reduced argument.  quad = 0: sin, quad = 1: cos.
// Round 1 reduced in fp64 (two v_fma_f64 + five conversions per evaluation): same accuracy, ~2x the issue cycles.
__device__ __forceinline__ float sincos_big(float a, int quad) {
  const float q = rintf(a * 0.63661977236758134308f);
  float r = __builtin_fmaf(-q, 1.5707963705062866f, a);
  r = __builtin_fmaf(-q, -4.371138828673793e-08f, r);
  r = __builtin_fmaf(-q, -1.7151245100058819e-15f, r);
  const int n = (int)q + quad;
  const float s = r * r;
  const float ps = r + r * s * (-1.6666654611e-1f + s * (8.3321608736e-3f + s * (-1.9515295891e-4f)));
  const float pc =
      1.0f + s * (-0.5f + s * (4.166664568298827e-2f + s * (-1.388731625493765e-3f + s * 2.443315711809948e-5f)));
  const float v = (n & 1) ? pc : ps;
  return (n & 2) ? -v : v;
}
__device__ __forceinline__ float sin_big(float a) { return sincos_big(a, 0); }
__device__ __forceinline__ float cos_big(float a) { return sincos_big(a, 1); }

// 34 real-SH polynomial terms, bands l = 1, 2, 4, 8 (reflect_sampling_nerf_components.py:65-127), then the
// per-band roughness attenuation exp(-rho*{1,3,10,36}) (components.py:136-139).
__device__ __forceinline__ void sh34_attenuated(float x, float y, float z, float rho, float sh[34]) {
  const float x2 = x * x, y2 = y * y, z2 = z * z;
  const float xy = x * y, xz = x * z, yz = y * z;
  const float a = x2 - y2;
  const float p = 3.0f * x2 - y2;
  const float q = x2 - 3.0f * y2;
  const float z4 = z2 * z2, x4 = x2 * x2, y4 = y2 * y2;
  const float im5 = y4 - 10.0f * x2 * y2 + 5.0f * x4;
  const float re5 = x4 - 10.0f * x2 * y2 + 5.0f * y4;
  const float im7 = (x2 - 5.0f * y2) * 7.0f * x4 + (21.0f * x2 - y2) * y4;
  const float re7 = (x2 - 21.0f * y2) * x4 + (5.0f * x2 - y2) * 7.0f * y4;
  const float re4 = x2 * q - y2 * p;
  const float t6 = 143.0f * z4 * z2 - 143.0f * z4 + 33.0f * z2 - 1.0f;
  const float t7 = 715.0f * z4 * z2 - 1001.0f * z4 + 385.0f * z2 - 35.0f;
  const float t5 = 39.0f * z4 - 26.0f * z2 + 3.0f;
  const float t4 = 65.0f * z4 - 26.0f * z2 + 1.0f;
  const float e1 = expf(-rho), e2 = expf(-rho * 3.0f), e4 = expf(-rho * 10.0f), e8 = expf(-rho * 36.0f);
  sh[0] = 0.48860251190291992f * y * e1;
  sh[1] = 0.48860251190291992f * z * e1;
  sh[2] = 0.48860251190291992f * x * e1;
  sh[3] = 1.09254843059207907f * xy * e2;
  sh[4] = 1.09254843059207907f * yz * e2;
  sh[5] = 0.31539156525252001f * (3.0f * z2 - 1.0f) * e2;
  sh[6] = 1.09254843059207907f * xz * e2;
  sh[7] = 0.54627421529603953f * a * e2;
  sh[8] = 2.50334294179670453f * xy * a * e4;
  sh[9] = 1.77013076977993053f * yz * p * e4;
  sh[10] = 0.94617469575756001f * xy * (7.0f * z2 - 1.0f) * e4;
  sh[11] = 0.66904654355728916f * yz * (7.0f * z2 - 3.0f) * e4;
  sh[12] = 0.1057855469152043038f * (35.0f * z4 - 30.0f * z2 + 3.0f) * e4;
  sh[13] = 0.66904654355728916f * xz * (7.0f * z2 - 3.0f) * e4;
  sh[14] = 0.473087347878780009f * a * (7.0f * z2 - 1.0f) * e4;
  sh[15] = 1.77013076977993053f * xz * q * e4;
  sh[16] = 0.62583573544917613f * re4 * e4;
  sh[17] = 5.83141328139863895f * xy * (x2 * x4 - 7.0f * x4 * y2 + 7.0f * x2 * y4 - y2 * y4) * e8;
  sh[18] = 5.83141328139863895f * yz * im7 * e8;
  sh[19] = 1.06466553211908514f * xy * (15.0f * z2 - 1.0f) * (3.0f * x4 - 10.0f * x2 * y2 + 3.0f * y4) * e8;
  sh[20] = 3.44991062209810801f * yz * (5.0f * z2 - 1.0f) * im5 * e8;
  sh[21] = 1.91366609903732278f * xy * t4 * a * e8;
  sh[22] = 1.23526615529554407f * yz * t5 * p * e8;
  sh[23] = 0.91230451686981894f * xy * t6 * e8;
  sh[24] = 0.1090412458987799555f * yz * t7 * e8;
  sh[25] = 0.0090867704915649962938f *
           (6435.0f * z4 * z4 - 12012.0f * z4 * z2 + 6930.0f * z4 - 1260.0f * z2 + 35.0f) * e8;
  sh[26] = 0.1090412458987799555f * xz * t7 * e8;
  sh[27] = 0.456152258434909470f * t6 * a * e8;
  sh[28] = 1.23526615529554407f * xz * t5 * q * e8;
  sh[29] = 0.478416524759330697f * t4 * re4 * e8;
  sh[30] = 3.44991062209810801f * xz * (5.0f * z2 - 1.0f) * re5 * e8;
  sh[31] = 0.53233276605954257f * (15.0f * z2 - 1.0f) * (x2 * re5 - y2 * im5) * e8;
  sh[32] = 5.83141328139863895f * xz * re7 * e8;
  sh[33] = 0.72892666017482986f * (x2 * re7 - y2 * im7) * e8;
}

// ReLU as ONE instruction: v_max_i32 on the bits (as signed integers every negative float, and -0, is negative; positive floats
// keep their order) -- from fmaxf / fmed3 hipcc emits v_max(x, x) first (it canonicalises a possible sNaN), two VALU per value in
// every layer epilogue.  Deliberately NOT inline asm (rounds 1-3 had `asm("v_max_f32 %0, 0, %1")`): the compiler's hazard
// recogniser does not see an asm operand, and a ReLU placed right behind the MFMA that writes its input reads the register
// before the matrix pipe has written it (found in rsn_field_x6_train.hip, round 4).  -NaN -> 0, +NaN stays.
__device__ __forceinline__ float relu_f(float x) {
  const int xi = __float_as_int(x);
  return __int_as_float(xi > 0 ? xi : 0);
}

// Rows kept for the backward pass / the weight gradients.  fp32 rows in the exact and the split-bf16 modes; in the
// reduced-precision training mode (RSN_MMA_BF16: the GEMMs round these values to bf16 anyway) the wide buffers
// (activations, bottleneck, mid hidden; layer gradients) are stored AS bf16 -- half the step's HBM stream.  `save`
// stays a float* in the signatures; SBF reinterprets it as a row of bf16 (element offsets, not bytes).
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
template <bool SBF>
__device__ __forceinline__ float* row_ptr(float* base, long long elem) {
  return SBF ? reinterpret_cast<float*>(reinterpret_cast<__bf16*>(base) + elem) : base + elem;
}
template <bool SBF>
__device__ __forceinline__ void put4(float* save, int off, const float4 v) {
  if (SBF) {
    const bf16x4 o = {(__bf16)v.x, (__bf16)v.y, (__bf16)v.z, (__bf16)v.w};
    *reinterpret_cast<bf16x4*>(reinterpret_cast<__bf16*>(save) + off) = o;
  } else {
    *reinterpret_cast<float4*>(save + off) = v;
  }
}

// The same rows through BUFFER stores (the product kernels): the descriptor covers the tile's VALID rows of a row-major
// [N, row_elems] buffer (base = the tile's first row, wave-uniform), the lane sends one 32-bit offset (its row + 16 h
// bytes) and the (block, q) position is a scalar offset.  Lanes past the last valid row fall outside the descriptor's
// range and the hardware drops their stores: no per-lane null pointers, no exec-masked branches around the stores.
// cache policy of the saved-row buffer stores (aux bits: 2 = nt); A/B switch of tools/ (default policy measured best for the
// per-wave-stream kernels in round 3; the LDS-ring bf16 training kernels use nt, rsn_field_bf16_train.hip)
#ifndef RSN_SAVED_ROW_AUX
#define RSN_SAVED_ROW_AUX 0
#endif
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
struct RowBuf {
  __amdgpu_buffer_rsrc_t r;
  unsigned voff;
  unsigned lim;  // bytes of one row (sv_put_it: K-iterations past the row are not stored)
  bool on;
};
template <bool SBF>
__device__ __forceinline__ RowBuf rowbuf(float* base, long long elem, int rows, int row_elems, int m, int h) {
  constexpr int BPE = SBF ? 2 : 4;
  RowBuf b;
  // a literal nullptr (eval instantiations) removes the stores at compile time; a buffer that is absent at run time gets
  // an empty range instead of a branch around every store
  b.on = !(__builtin_constant_p(base == nullptr) && base == nullptr);
#ifdef RSN_DIAG_NO_SAVED_ROWS  // timing ablation (wrong training results): no saved-row store is issued
  b.on = false;
#endif
  b.r = __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<char*>(base) + elem * BPE, 0,
                                          base != nullptr ? rows * row_elems * BPE : 0, 0x00020000);
  b.voff = (unsigned)((m * row_elems + 4 * h) * BPE);
  b.lim = (unsigned)(row_elems * BPE);
  return b;
}
__device__ __forceinline__ bool sv_on(const float* s) { return s != nullptr; }
__device__ __forceinline__ bool sv_on(const RowBuf& b) { return b.on; }
template <bool SBF>
__device__ __forceinline__ void sv_put(float* save, int nb, int q, int h, const float4 v) {
  put4<SBF>(save, (nb * 4 + q) * 8 + 4 * h, v);
}
template <bool SBF>
__device__ __forceinline__ void sv_put(const RowBuf& b, int nb, int q, int, const float4 v) {
  if (SBF) {
    const bf16x4 o = {(__bf16)v.x, (__bf16)v.y, (__bf16)v.z, (__bf16)v.w};
    __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2, o), b.r, b.voff + (unsigned)((nb * 4 + q) * 16), 0, RSN_SAVED_ROW_AUX);
  } else {
    const u32x4 o = {__float_as_uint(v.x), __float_as_uint(v.y), __float_as_uint(v.z), __float_as_uint(v.w)};
    // (offset in the vector offset / immediate, scalar offset 0: a > 8-byte buffer store with a REGISTER soffset is not protected
    // by hipcc against a VALU overwriting its data registers in the next instruction -- see st16, rsn_ringt.h)
    __builtin_amdgcn_raw_buffer_store_b128(o, b.r, b.voff + (unsigned)((nb * 4 + q) * 32), 0, RSN_SAVED_ROW_AUX);
  }
}
// the float4 of K-iteration `it` of the lane's row (features it*8 + 4h ..): the same bytes sv_put(nb = it/4, q = it%4) writes
__device__ __forceinline__ void sv_put_it(const float*, int, const float4) {}
__device__ __forceinline__ void sv_put_it(const RowBuf& b, int it, const float4 v) {
  const u32x4 o = {__float_as_uint(v.x), __float_as_uint(v.y), __float_as_uint(v.z), __float_as_uint(v.w)};
  if ((unsigned)it * 32u < b.lim) __builtin_amdgcn_raw_buffer_store_b128(o, b.r, b.voff + (unsigned)it * 32u, 0, RSN_SAVED_ROW_AUX);  // wave-uniform
}

// ------------------------------------------------------------------------------------------------
// MFMA K loop: acc[nb] += W_seg[nb-block] * X, weights double-buffered in registers.
// ------------------------------------------------------------------------------------------------
// NBT = output blocks of the packed segment ([it][NBT][lane][4]); a wave that takes only NBO < NBT of them passes its
// segment pointer already advanced to its first block (tools/probes/gemm_occ_probe.hip).
// Weight fragments arrive through BUFFER loads: the segment base lives in a scalar buffer descriptor, the K-iteration in
// the scalar offset, and a lane sends ONE 32-bit offset (lane * 16 + nb KiB, loop-invariant registers) instead of a
// 64-bit address that a v_add_co pair advances per load.  The 64-bit-address form (global_load_dwordx4 v[lo:hi]) was the
// unexplained per-load cost of rounds 1-2: with buffer loads the eval kernel went 4.90 -> 4.63 ms (83.8 -> 88.6 % of the fp32-MFMA
// peak) and the training forward 22.5 -> 21.4 ms per step (76.6 -> 80.4 %), same box (profiles/r03_buffer_loads.txt).
// Reads past the descriptor's 2 GiB window return 0 (never reached: a packed segment is < 1 MiB).
struct WBuf {
  __amdgpu_buffer_rsrc_t r;
  unsigned voff;  // lane * 16
};
__device__ __forceinline__ WBuf wbuf_make(const float* __restrict__ wseg, int lane) {
  WBuf b;
  b.r = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(wseg), 0, 0x7fffffff, 0x00020000);
  b.voff = (unsigned)lane * 16u;
  return b;
}
template <int NBO, int NBT = NBO>
__device__ __forceinline__ void load_w(float4 (&w)[NBO], const WBuf& b, int it) {
#pragma unroll
  for (int nb = 0; nb < NBO; ++nb) {
    const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(b.r, b.voff + nb * 1024u, (unsigned)it * (NBT * 1024u), 0);
    w[nb] = make_float4(__uint_as_float(v.x), __uint_as_float(v.y), __uint_as_float(v.z), __uint_as_float(v.w));
  }
}

template <int NBO>
__device__ __forceinline__ void mma4(f32x16 (&acc)[NBO], const float4 (&w)[NBO], const float4 b) {
#pragma unroll
  for (int nb = 0; nb < NBO; ++nb) acc[nb] = __builtin_amdgcn_mfma_f32_32x32x2f32(w[nb].x, b.x, acc[nb], 0, 0, 0);
#pragma unroll
  for (int nb = 0; nb < NBO; ++nb) acc[nb] = __builtin_amdgcn_mfma_f32_32x32x2f32(w[nb].y, b.y, acc[nb], 0, 0, 0);
#pragma unroll
  for (int nb = 0; nb < NBO; ++nb) acc[nb] = __builtin_amdgcn_mfma_f32_32x32x2f32(w[nb].z, b.z, acc[nb], 0, 0, 0);
#pragma unroll
  for (int nb = 0; nb < NBO; ++nb) acc[nb] = __builtin_amdgcn_mfma_f32_32x32x2f32(w[nb].w, b.w, acc[nb], 0, 0, 0);
}

// wseg: packed segment base (global), xl: this lane's slot of the LDS slab (float4 units, stride 64 per it).
// Software pipeline: the loads of K-iteration it+1 are issued while the 32 MFMAs of iteration it run, one load
// after each of the first NBO MFMAs (sched_group_barrier pattern), so every load has a full iteration (2048 MFMA
// cycles) of cover; hipcc on its own sinks the loads to just ahead of their first use.
// Measured with tools/phase_report.py (BASELINE config 2): with global_load_dwordx4 (64-bit VGPR addresses) the trunk K
// loops ran at 93.5 % of the MFMA issue rate and neither the prefetch distance, the position of the loads nor L1 residency
// changed that; with buffer loads (load_w above) they run at 98.4 % (profiles/r03_phase_eval.json): the cost was the
// address path of the load, not its data or its latency.
template <int NBO>
__device__ __forceinline__ void interleave_loads() {
#pragma unroll
  for (int g = 0; g < NBO; ++g) {
    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);  // 1 MFMA
    __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);  // 1 VMEM read
  }
  __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
  __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);  // the LDS read of the activations
}

// First weight fragment of a segment: issued by the caller ahead of the epilogue in front of the GEMM so that its L2
// round trip hides under that epilogue.
template <int NBO, int NBT = NBO>
__device__ __forceinline__ void pre_w(float4 (&wa)[NBO], const float* __restrict__ wseg, int lane) {
  load_w<NBO, NBT>(wa, wbuf_make(wseg, lane), 0);
}

// `save` (optional): the rows of the activations this GEMM READS are kept for the backward pass (rsn_field_saved.act /
// bott / hid; rsn_field_grads_out.dy ...).  The K loop holds float4 `it` of the lane's row in a register anyway (the
// MFMA B operand), so the row leaves as ONE buffer store per K-iteration, 2,048 MFMA cycles apart -- instead of a burst
// of 32 stores in the epilogue that produced it, which blocked the wave at the CU's 64 B/clk write port
// (profiles/r03_store_in_loop.txt).
// PF2: the weight fragments TWO K-iterations ahead (three fragment buffers, 4,096 MFMA cycles of cover).  The training
// kernels stream forward AND transposed weights, 4.5 MB against the 4 MiB L2 of an XCD, so part of their stream comes from
// beyond the L2 and one iteration of cover is short: training forward 20.9 -> 20.3 ms per step (82.2 -> 84.7 % of the
// fp32-MFMA peak), backward 11.97 -> 11.81 ms.  The eval kernel's 2.5 MB stay in L2 and it is 1 % slower this way (4.64 ->
// 4.70 ms): it keeps the two-buffer loop below.
template <int NBO, int NBT = NBO, class SV = const float*>
__device__ __forceinline__ void gemm_run2(f32x16 (&acc)[NBO], float4 (&wa)[NBO], const float* __restrict__ wseg,
                                           const float4* xl, int n_it, int lane, SV save = nullptr) {
  const WBuf wp = wbuf_make(wseg, lane);
  float4 wb[NBO], wc[NBO];
  float4 b0, b1, b2;
  auto cl = [&](int i) { return i < n_it ? i : n_it - 1; };
  load_w<NBO, NBT>(wb, wp, cl(1));
  b0 = xl[0];
  int it = 0;
#pragma unroll 1
  for (; it + 2 < n_it; it += 3) {
    load_w<NBO, NBT>(wc, wp, cl(it + 2));
    b1 = xl[cl(it + 1) * 64];
    mma4<NBO>(acc, wa, b0);
    if (sv_on(save)) sv_put_it(save, it, b0);
    interleave_loads<NBO>();
    __builtin_amdgcn_sched_barrier(0);
    load_w<NBO, NBT>(wa, wp, cl(it + 3));
    b2 = xl[cl(it + 2) * 64];
    mma4<NBO>(acc, wb, b1);
    if (sv_on(save)) sv_put_it(save, it + 1, b1);
    interleave_loads<NBO>();
    __builtin_amdgcn_sched_barrier(0);
    load_w<NBO, NBT>(wb, wp, cl(it + 4));
    b0 = xl[cl(it + 3) * 64];
    mma4<NBO>(acc, wc, b2);
    if (sv_on(save)) sv_put_it(save, it + 2, b2);
    interleave_loads<NBO>();
    __builtin_amdgcn_sched_barrier(0);
  }
  if (it < n_it) {
    mma4<NBO>(acc, wa, b0);
    if (sv_on(save)) sv_put_it(save, it, b0);
  }
  if (it + 1 < n_it) {
    b1 = xl[(it + 1) * 64];
    mma4<NBO>(acc, wb, b1);
    if (sv_on(save)) sv_put_it(save, it + 1, b1);
  }
}

template <int NBO, int NBT = NBO, class SV = const float*>
__device__ __forceinline__ void gemm_run(f32x16 (&acc)[NBO], float4 (&wa)[NBO], const float* __restrict__ wseg,
                                          const float4* xl, int n_it, int lane, SV save = nullptr) {
  const WBuf wp = wbuf_make(wseg, lane);
  float4 wb[NBO];
  float4 ba, bb;
  ba = xl[0];
  int it = 0;
#pragma unroll 1
  for (; it + 1 < n_it; it += 2) {
    load_w<NBO, NBT>(wb, wp, it + 1);
    bb = xl[(it + 1) * 64];
    mma4<NBO>(acc, wa, ba);
    if (sv_on(save)) sv_put_it(save, it, ba);
    interleave_loads<NBO>();
    __builtin_amdgcn_sched_barrier(0);
    {  // unconditional prefetch with a clamped index: one control path => exact vmcnt counts
      const int in = (it + 2 < n_it) ? it + 2 : n_it - 1;
      load_w<NBO, NBT>(wa, wp, in);
      ba = xl[in * 64];
    }
    mma4<NBO>(acc, wb, bb);
    if (sv_on(save)) sv_put_it(save, it + 1, bb);
    interleave_loads<NBO>();
    __builtin_amdgcn_sched_barrier(0);
  }
  if (it < n_it) {
    mma4<NBO>(acc, wa, ba);
    if (sv_on(save)) sv_put_it(save, it, ba);
  }
}

template <int NBO, int NBT = NBO, bool PF2 = false, class SV = const float*>
__device__ __forceinline__ void gemm(f32x16 (&acc)[NBO], const float* __restrict__ wseg, const float4* xl, int n_it,
                                     int lane, SV save = nullptr) {
  float4 wa[NBO];
  pre_w<NBO, NBT>(wa, wseg, lane);
  if (PF2) gemm_run2<NBO, NBT>(acc, wa, wseg, xl, n_it, lane, save); else gemm_run<NBO, NBT>(acc, wa, wseg, xl, n_it, lane, save);
}

// acc[nb][4q+j] = bias[nb*32 + 8q + 4h + j]: the accumulators start from the bias (what torch's addmm does),
// so the 4*NBO bias loads are issued back to back ahead of the K loop and the epilogue needs no memory reads.
template <int NBO>
__device__ __forceinline__ void init_acc(f32x16 (&acc)[NBO], const float* __restrict__ bias, int h) {
#pragma unroll
  for (int nb = 0; nb < NBO; ++nb)
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const float4 bv = *reinterpret_cast<const float4*>(bias + nb * 32 + 8 * q + 4 * h);
      acc[nb][4 * q + 0] = bv.x;
      acc[nb][4 * q + 1] = bv.y;
      acc[nb][4 * q + 2] = bv.z;
      acc[nb][4 * q + 3] = bv.w;
    }
}

// ------------------------------------------------------------------------------------------------
// Split-bf16 K loop (RSN_MMA_BF16X6 / X3): the same GEMM on v_mfma_f32_32x32x16_bf16.
// One K=16 step consumes the lane's two float4's of LDS iterations 2kk, 2kk+1 (same lane-local activations as
// the fp32 loop).  The 8 fp32 activations are split exactly into bf16 triples in registers; the weights arrive
// pre-split ([k16][nb][split][lane][8 bf16], rsn_pack.hip).  NSPLIT = 3: products w1x1, w1x2, w2x1, w1x3, w2x2,
// w3x1 (dropped terms <= 2^-24 relative);  NSPLIT = 2: w1x1, w1x2, w2x1.
// Weight fragments are double-buffered at half-step granularity (half of the output blocks) to fit the VGPR budget.
// ------------------------------------------------------------------------------------------------
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

struct BSplit {
  bf16x8 s1, s2, s3;
};

template <int NSPLIT>
__device__ __forceinline__ BSplit split8(const float4 lo, const float4 hi) {
  const float x[8] = {lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
  BSplit o;
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    const __bf16 b1 = (__bf16)x[e];
    const float r1 = x[e] - (float)b1;
    const __bf16 b2 = (NSPLIT >= 2) ? (__bf16)r1 : (__bf16)0.0f;
    o.s1[e] = b1;
    o.s2[e] = b2;
    if (NSPLIT == 3) {
      const float r2 = r1 - (float)b2;
      o.s3[e] = (__bf16)r2;
    } else {
      o.s3[e] = (__bf16)0.0f;
    }
  }
  return o;
}

template <int NH, int NSPLIT>
__device__ __forceinline__ void load_w16(bf16x8 (&w)[NH][3], const WBuf& wp, int kk, int nbo, int nb0) {
#pragma unroll
  for (int t = 0; t < NH; ++t)
#pragma unroll
    for (int sp = 0; sp < NSPLIT; ++sp) {
#ifdef RSN_DIAG_X6_SAMEW  // timing ablation (wrong results): every K step re-reads step 0's fragments (L1 / L2 hits)
      const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(wp.r, wp.voff, (unsigned)(((0 * nbo + nb0 + t) * 3 + sp) * 1024), 0);
#else
      const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(wp.r, wp.voff, (unsigned)(((kk * nbo + nb0 + t) * 3 + sp) * 1024), 0);
#endif
      w[t][sp] = __builtin_bit_cast(bf16x8, v);
    }
}

template <int NBO, int NH, int NB0, int NSPLIT>
__device__ __forceinline__ void mma16(f32x16 (&acc)[NBO], const bf16x8 (&w)[NH][3], const BSplit& b) {
#pragma unroll
  for (int t = 0; t < NH; ++t) {
    f32x16 c = acc[NB0 + t];
    if (NSPLIT == 3) {
      c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w[t][2], b.s1, c, 0, 0, 0);
      c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w[t][1], b.s2, c, 0, 0, 0);
      c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w[t][0], b.s3, c, 0, 0, 0);
    }
    if (NSPLIT >= 2) {
      c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w[t][1], b.s1, c, 0, 0, 0);
      c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w[t][0], b.s2, c, 0, 0, 0);
    }
    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w[t][0], b.s1, c, 0, 0, 0);
    acc[NB0 + t] = c;
  }
}

template <int NBO, int NSPLIT>
__device__ __forceinline__ void gemm_bf16(f32x16 (&acc)[NBO], const float* __restrict__ wseg, const float4* xl,
                                          int n_k16, int lane) {
  constexpr int H0 = (NBO + 1) / 2, H1 = NBO - H0;
  constexpr int PER = (NSPLIT == 3 ? 6 : (NSPLIT == 2 ? 3 : 1));  // MFMAs per output block and K step
  const WBuf wp = wbuf_make(wseg, lane);
  bf16x8 wa[H0][3], wb[H1 > 0 ? H1 : 1][3];
  load_w16<H0, NSPLIT>(wa, wp, 0, NBO, 0);
  BSplit bc = split8<NSPLIT>(xl[0], xl[64]);
  const int k1 = n_k16 > 1 ? 1 : 0;
  float4 xlo = xl[(2 * k1) * 64], xhi = xl[(2 * k1 + 1) * 64];  // fp32 activations of the NEXT K step
#pragma unroll 1
  for (int kk = 0; kk < n_k16; ++kk) {
    // ---- phase A: first-half MFMAs; the 3-way split of the next step's activations rides in their issue gaps
    //      (matrix and vector pipes are separate; one 32-cycle MFMA leaves room for a few single-issue VALU ops)
    if (H1 > 0) load_w16<(H1 > 0 ? H1 : 1), NSPLIT>(wb, wp, kk, NBO, H0);
    __builtin_amdgcn_sched_barrier(0);
    const BSplit bn = split8<NSPLIT>(xlo, xhi);
    mma16<NBO, H0, 0, NSPLIT>(acc, wa, bc);
#pragma unroll
    for (int g = 0; g < H0 * PER; ++g) {
      __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);  // 1 MFMA
      __builtin_amdgcn_sched_group_barrier(0x002, 2, 0);  // 2 VALU
    }
    __builtin_amdgcn_sched_barrier(0);
    // ---- phase B: prefetch (unconditional, clamped: one control path keeps the s_waitcnt counts exact), then the
    //      second-half MFMAs
    const int kn = (kk + 1 < n_k16) ? kk + 1 : kk;
    const int k2 = (kk + 2 < n_k16) ? kk + 2 : kn;
    load_w16<H0, NSPLIT>(wa, wp, kn, NBO, 0);
    xlo = xl[(2 * k2) * 64];
    xhi = xl[(2 * k2 + 1) * 64];
    __builtin_amdgcn_sched_barrier(0);
    if (H1 > 0) mma16<NBO, (H1 > 0 ? H1 : 1), (H1 > 0 ? H0 : 0), NSPLIT>(acc, wb, bc);
    __builtin_amdgcn_sched_barrier(0);
    bc = bn;
  }
}

// dispatch on the MMA mode: MODE 0 = fp32 MFMA over n_it K-iterations of 8, else split-bf16 over ceil(n_it/2) K=16 steps
template <int MODE, int NBO, int NBT = NBO, bool PF2 = false, class SV = const float*>
__device__ __forceinline__ void gemm_mode(f32x16 (&acc)[NBO], const float* __restrict__ w32, const float* __restrict__ w16,
                                          const float4* xl, int n_it, int lane, SV save = nullptr) {
  static_assert(MODE == 0 || NBT == NBO, "the split-bf16 loops own every output block of their segment");
  if (MODE == 0) {
    gemm<NBO, NBT, PF2>(acc, w32, xl, n_it, lane, save);
  } else {
    gemm_bf16<NBO, (MODE == 1 ? 3 : (MODE == 2 ? 2 : 1))>(acc, w16, xl, (n_it + 1) / 2, lane);
  }
}

// the same with the first fp32 weight fragment fetched early by the caller (pre_mode ... gemm_mode_run); the
// split-bf16 loops fetch their own first fragment (pre_mode is a no-op for them)
template <int MODE, int NBO, int NBT = NBO>
__device__ __forceinline__ void pre_mode(float4 (&wa)[NBO], const float* __restrict__ w32, int lane) {
  if (MODE == 0) pre_w<NBO, NBT>(wa, w32, lane);
}

template <int MODE, int NBO, int NBT = NBO, bool PF2 = false, class SV = const float*>
__device__ __forceinline__ void gemm_mode_run(f32x16 (&acc)[NBO], float4 (&wa)[NBO], const float* __restrict__ w32,
                                              const float* __restrict__ w16, const float4* xl, int n_it, int lane,
                                              SV save = nullptr) {
  static_assert(MODE == 0 || NBT == NBO, "the split-bf16 loops own every output block of their segment");
  if (MODE == 0) {  // (the split-bf16 loops take no saver: their epilogues store)
    if (PF2) gemm_run2<NBO, NBT>(acc, wa, w32, xl, n_it, lane, save); else gemm_run<NBO, NBT>(acc, wa, w32, xl, n_it, lane, save);
  } else {
    gemm_bf16<NBO, (MODE == 1 ? 3 : (MODE == 2 ? 2 : 1))>(acc, w16, xl, (n_it + 1) / 2, lane);
  }
}

template <int NBO>
__device__ __forceinline__ void zero_acc(f32x16 (&acc)[NBO]) {
#pragma unroll
  for (int nb = 0; nb < NBO; ++nb)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[nb][r] = 0.0f;
}

// ReLU sign bits of one lane: bit nb*16 + r = (accumulator register r of block nb, the layer's pre-activation) > 0,
// packed into NBO/2 words (rsn_field_saved.relu_bits).  The dX sweeps mask by these bits instead of re-reading the
// saved fp32 activations (1 KiB per point and layer -> 32 B; 124 fewer live registers in the sweeps).
__device__ __forceinline__ unsigned relu_bits16(const f32x16& a) {
  // pre-activation x > 0  <=>  the bits of relu(x) (relu_f: one v_max_i32, shared with the ReLU epilogue beside this) are non-zero:
  // v_min_u32(., 1), then v_lshl_or_b32.  The C forms (`t > 0`, clamp(t, 0, 1)) become v_cmp_lt_i32 (SGPR pair) + a hazard nop +
  // v_cndmask + v_or3.  The asm takes the relu'd value, NOT the accumulator: an asm operand that is an MFMA result is invisible to
  // the compiler's hazard recogniser (see relu_f); behind the compiler's own v_max_i32 it is an ordinary VALU result.
  unsigned b = 0u;
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const unsigned lo = __float_as_uint(relu_f(a[r]));
    unsigned bit;
    if (r == 0) {
      asm("v_min_u32 %0, %1, 1" : "=v"(bit) : "v"(lo));
      b = bit;
    } else {
      unsigned nb;
      asm("v_min_u32 %0, %2, 1\n\tv_lshl_or_b32 %1, %0, %3, %4" : "=&v"(bit), "=v"(nb) : "v"(lo), "n"(r), "v"(b));
      b = nb;
    }
  }
  return b;
}

// the same for four relu'd values at bit positions 4 q .. 4 q + 3 (the ReLU epilogues below take the bits from the values they
// have just formed: no second v_max per value)
template <int Q>
__device__ __forceinline__ unsigned relu_bits4q(unsigned b, const float4 v) {
  const unsigned lo[4] = {__float_as_uint(v.x), __float_as_uint(v.y), __float_as_uint(v.z), __float_as_uint(v.w)};
#pragma unroll
  for (int t = 0; t < 4; ++t) {
    unsigned bit, nb;
    if (Q == 0 && t == 0) {
      asm("v_min_u32 %0, %1, 1" : "=v"(bit) : "v"(lo[t]));
      b = bit;
    } else if (t == 0) {
      asm("v_min_u32 %0, %2, 1\n\tv_lshl_or_b32 %1, %0, %3, %4" : "=&v"(bit), "=v"(nb) : "v"(lo[t]), "n"(4 * Q), "v"(b));
      b = nb;
    } else if (t == 1) {
      asm("v_min_u32 %0, %2, 1\n\tv_lshl_or_b32 %1, %0, %3, %4" : "=&v"(bit), "=v"(nb) : "v"(lo[t]), "n"(4 * Q + 1), "v"(b));
      b = nb;
    } else if (t == 2) {
      asm("v_min_u32 %0, %2, 1\n\tv_lshl_or_b32 %1, %0, %3, %4" : "=&v"(bit), "=v"(nb) : "v"(lo[t]), "n"(4 * Q + 2), "v"(b));
      b = nb;
    } else {
      asm("v_min_u32 %0, %2, 1\n\tv_lshl_or_b32 %1, %0, %3, %4" : "=&v"(bit), "=v"(nb) : "v"(lo[t]), "n"(4 * Q + 3), "v"(b));
      b = nb;
    }
  }
  return b;
}
__device__ __forceinline__ unsigned relu_bits4(unsigned b, const float4 v, int q) {  // q: a constant after unrolling
  return q == 0 ? relu_bits4q<0>(b, v) : (q == 1 ? relu_bits4q<1>(b, v) : (q == 2 ? relu_bits4q<2>(b, v) : relu_bits4q<3>(b, v)));
}

template <int NBO>
__device__ __forceinline__ void store_relu_bits(const f32x16 (&acc)[NBO], unsigned* bits) {
  if (NBO >= 8) {
#pragma unroll
    for (int w4 = 0; w4 < NBO / 8; ++w4) {
      uint4 v;
      v.x = relu_bits16(acc[w4 * 8 + 0]) | (relu_bits16(acc[w4 * 8 + 1]) << 16);
      v.y = relu_bits16(acc[w4 * 8 + 2]) | (relu_bits16(acc[w4 * 8 + 3]) << 16);
      v.z = relu_bits16(acc[w4 * 8 + 4]) | (relu_bits16(acc[w4 * 8 + 5]) << 16);
      v.w = relu_bits16(acc[w4 * 8 + 6]) | (relu_bits16(acc[w4 * 8 + 7]) << 16);
      *reinterpret_cast<uint4*>(bits + w4 * 4) = v;
    }
  } else {
#pragma unroll
    for (int w = 0; w < NBO / 2; ++w) bits[w] = relu_bits16(acc[2 * w]) | (relu_bits16(acc[2 * w + 1]) << 16);
  }
}

// the lane's NBO/2 mask words of one layer (issued BEFORE the layer's GEMM, consumed by store_masked_bits after it)
template <int NBO>
struct ReluBits {
  unsigned w[NBO / 2];
};

template <int NBO>
__device__ __forceinline__ ReluBits<NBO> load_relu_bits(const unsigned* __restrict__ bits) {
  ReluBits<NBO> m;
  if (NBO >= 8) {
#pragma unroll
    for (int w4 = 0; w4 < NBO / 8; ++w4) {
      const uint4 v = *reinterpret_cast<const uint4*>(bits + w4 * 4);
      m.w[w4 * 4 + 0] = v.x; m.w[w4 * 4 + 1] = v.y; m.w[w4 * 4 + 2] = v.z; m.w[w4 * 4 + 3] = v.w;
    }
  } else {
#pragma unroll
    for (int w = 0; w < NBO / 2; ++w) m.w[w] = bits[w];
  }
  return m;
}

// bit `pos` of `word` as 0 / 0xffffffff: one v_bfe_i32.  Written as (non-volatile) asm because hipcc rewrites
// `x & sext(bit)` into v_and (test) + v_cmp_ne + v_cndmask: three VALU per value and a VCC hazard nop, where bfe + and is two.
__device__ __forceinline__ unsigned bit_mask(int word, int pos) {
  unsigned m;
  asm("v_bfe_i32 %0, %1, %2, 1" : "=v"(m) : "v"(word), "n"(pos));
  return m;
}

// dX-sweep epilogue on mask bits: X[it][lane] = bit ? acc : 0 (v_bfe_i32 gives 0 / -1, one v_and applies it);
// optionally also stored to row `save` (backward pass: the layer's pre-activation gradient for the weight gradients)
template <int NBO, bool SBF = false, class SV = float*>
__device__ __forceinline__ void store_masked_bits(const f32x16 (&acc)[NBO], float4* xl, const ReluBits<NBO>& m, int h,
                                                  SV save = nullptr) {
#pragma unroll
  for (int nb = 0; nb < NBO; ++nb)
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int word = m.w[nb / 2];
      const int base = (nb & 1) * 16 + 4 * q;
      float4 v;
#ifdef RSN_DIAG_NO_EPI_VALU
      (void)word; (void)base;
      v = make_float4(acc[nb][4 * q + 0], acc[nb][4 * q + 1], acc[nb][4 * q + 2], acc[nb][4 * q + 3]);
#else
      v.x = __uint_as_float(__float_as_uint(acc[nb][4 * q + 0]) & bit_mask(word, base + 0));
      v.y = __uint_as_float(__float_as_uint(acc[nb][4 * q + 1]) & bit_mask(word, base + 1));
      v.z = __uint_as_float(__float_as_uint(acc[nb][4 * q + 2]) & bit_mask(word, base + 2));
      v.w = __uint_as_float(__float_as_uint(acc[nb][4 * q + 3]) & bit_mask(word, base + 3));
#endif
      xl[(nb * 4 + q) * 64] = v;
      if (sv_on(save)) sv_put<SBF>(save, nb, q, h, v);
    }
}

// X[it = nb*4+q][lane] = act(acc[nb][4q..4q+3])   (bias already inside acc, see init_acc).
// save (training): the same float4 also goes to row `save` of a row-major [N, 32*NBS] activation buffer
// (this lane's point; the four q of one nb complete one 128-B line per row).
template <int NBO, int NBS, bool RELU, bool SBF = false, class SV = float*>
__device__ __forceinline__ void store_act(const f32x16 (&acc)[NBO], float4* xl, SV save = nullptr, int h = 0,
                                          unsigned* bits = nullptr) {
  unsigned bw[NBS / 2 > 0 ? NBS / 2 : 1];
#pragma unroll
  for (int nb = 0; nb < NBS; ++nb) {
    unsigned b16 = 0u;  // the block's 16 mask bits, taken from the relu'd values as they are formed (dead code without `bits`)
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      float4 v = make_float4(acc[nb][4 * q + 0], acc[nb][4 * q + 1], acc[nb][4 * q + 2], acc[nb][4 * q + 3]);
#ifndef RSN_DIAG_NO_EPI_VALU  // (timing ablation, wrong results: the upper bound of moving this work under the next layer's MFMAs)
      if (RELU) {
        v.x = relu_f(v.x);
        v.y = relu_f(v.y);
        v.z = relu_f(v.z);
        v.w = relu_f(v.w);
        b16 = relu_bits4(b16, v, q);
      }
#endif
      xl[(nb * 4 + q) * 64] = v;
      if (sv_on(save)) sv_put<SBF>(save, nb, q, h, v);
    }
    if (RELU) {
      if (nb & 1) bw[nb / 2] |= b16 << 16; else bw[nb / 2] = b16;
    }
  }
  if (RELU && bits) {
    if (NBS >= 8) {
#pragma unroll
      for (int w4 = 0; w4 < NBS / 8; ++w4)
        *reinterpret_cast<uint4*>(bits + w4 * 4) = make_uint4(bw[w4 * 4], bw[w4 * 4 + 1], bw[w4 * 4 + 2], bw[w4 * 4 + 3]);
    } else {
#pragma unroll
      for (int w = 0; w < NBS / 2; ++w) bits[w] = bw[w];
    }
  }
}

// store_act of one layer fused with init_acc of the next (same NBO): block by block the accumulators are read out and
// immediately re-loaded with the next layer's bias, so the bias round trip hides under the rest of the epilogue.
template <int NBO, bool RELU, bool SBF = false, class SV = float*>
__device__ __forceinline__ void store_act_init(f32x16 (&acc)[NBO], float4* xl, SV save, int h,
                                               const float* __restrict__ bias, unsigned* bits = nullptr) {
  unsigned bw[NBO / 2 > 0 ? NBO / 2 : 1];
#pragma unroll
  for (int nb = 0; nb < NBO; ++nb) {
    unsigned b16 = 0u;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      float4 v = make_float4(acc[nb][4 * q + 0], acc[nb][4 * q + 1], acc[nb][4 * q + 2], acc[nb][4 * q + 3]);
#ifndef RSN_DIAG_NO_EPI_VALU  // (timing ablation, wrong results: the upper bound of moving this work under the next layer's MFMAs)
      if (RELU) {
        v.x = relu_f(v.x);
        v.y = relu_f(v.y);
        v.z = relu_f(v.z);
        v.w = relu_f(v.w);
        b16 = relu_bits4(b16, v, q);
      }
#endif
      xl[(nb * 4 + q) * 64] = v;
      if (sv_on(save)) sv_put<SBF>(save, nb, q, h, v);
      const float4 bv = *reinterpret_cast<const float4*>(bias + nb * 32 + 8 * q + 4 * h);
      acc[nb][4 * q + 0] = bv.x;
      acc[nb][4 * q + 1] = bv.y;
      acc[nb][4 * q + 2] = bv.z;
      acc[nb][4 * q + 3] = bv.w;
    }
    if (RELU) {
      if (nb & 1) bw[nb / 2] |= b16 << 16; else bw[nb / 2] = b16;
    }
  }
  if (RELU && bits) {
    if (NBO >= 8) {
#pragma unroll
      for (int w4 = 0; w4 < NBO / 8; ++w4)
        *reinterpret_cast<uint4*>(bits + w4 * 4) = make_uint4(bw[w4 * 4], bw[w4 * 4 + 1], bw[w4 * 4 + 2], bw[w4 * 4 + 3]);
    } else {
#pragma unroll
      for (int w = 0; w < NBO / 2; ++w) bits[w] = bw[w];
    }
  }
}

// Prefetch of the ReLU-mask rows (saved post-ReLU activations) of a dX-sweep layer: issued BEFORE the layer's GEMM so
// that the HBM round trip hides under its MFMAs (consuming them right after the load parked 30 % of the backward
// kernel's wave cycles in s_waitcnt).
template <int NBO>
__device__ __forceinline__ void load_mask(float4 (&mk)[NBO * 4], const float* __restrict__ xin, int h) {
#pragma unroll
  for (int nb = 0; nb < NBO; ++nb)
#pragma unroll
    for (int q = 0; q < 4; ++q) mk[nb * 4 + q] = *reinterpret_cast<const float4*>(xin + (nb * 4 + q) * 8 + 4 * h);
}

template <int NBO>
__device__ __forceinline__ void store_masked_pre(const f32x16 (&acc)[NBO], float4* xl, const float4 (&mk)[NBO * 4], int h,
                                                 float* save = nullptr) {
#pragma unroll
  for (int nb = 0; nb < NBO; ++nb)
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const float4 x = mk[nb * 4 + q];
      float4 v;
      v.x = x.x > 0.0f ? acc[nb][4 * q + 0] : 0.0f;
      v.y = x.y > 0.0f ? acc[nb][4 * q + 1] : 0.0f;
      v.z = x.z > 0.0f ? acc[nb][4 * q + 2] : 0.0f;
      v.w = x.w > 0.0f ? acc[nb][4 * q + 3] : 0.0f;
      xl[(nb * 4 + q) * 64] = v;
      if (save) *reinterpret_cast<float4*>(save + (nb * 4 + q) * 8 + 4 * h) = v;
    }
}

// dX-sweep epilogue: X[it][lane] = (x_in > 0) ? acc : 0 with x_in = the saved post-ReLU input of the layer
// (row `xin` of a row-major activation buffer); optionally also stored to row `save` (backward pass).
template <int NBO>
__device__ __forceinline__ void store_masked(const f32x16 (&acc)[NBO], float4* xl, const float* __restrict__ xin, int h,
                                             float* save = nullptr) {
#pragma unroll
  for (int nb = 0; nb < NBO; ++nb)
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const float4 x = *reinterpret_cast<const float4*>(xin + (nb * 4 + q) * 8 + 4 * h);
      float4 v;
      v.x = x.x > 0.0f ? acc[nb][4 * q + 0] : 0.0f;
      v.y = x.y > 0.0f ? acc[nb][4 * q + 1] : 0.0f;
      v.z = x.z > 0.0f ? acc[nb][4 * q + 2] : 0.0f;
      v.w = x.w > 0.0f ? acc[nb][4 * q + 3] : 0.0f;
      xl[(nb * 4 + q) * 64] = v;
      if (save) *reinterpret_cast<float4*>(save + (nb * 4 + q) * 8 + 4 * h) = v;
    }
}

