// rsn_ring16.h -- what the plain-bf16 kernels share: packed-bf16 helpers, the fast activations of the bf16 mode, the LDS weight
// ring (LDS-DMA producer / counted-wait consumer) and the 16x16x32 GEMM over it.  Used by rsn_field_bf16.hip (eval) and
// rsn_field_bf16_train.hip (training forward + backward sweeps).  See rsn_field_bf16.hip for the design notes.
#pragma once
#include "rsn_field_common.h"

__device__ __forceinline__ bf16x8 pack8(const float (&v)[8]) {
  bf16x8 o;
#pragma unroll
  for (int e = 0; e < 8; ++e) o[e] = (__bf16)v[e];
  return o;
}

typedef float float2v __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef short short2v __attribute__((ext_vector_type(2)));
typedef unsigned int uint4v __attribute__((ext_vector_type(4)));

// Two fp32 -> one dword of two bf16 (v_cvt_pk_bf16_f32), optionally ReLU'd.  ReLU on packed bf16: as signed 16-bit
// integers negative floats (and -0) are negative, so one v_pk_max_i16 with 0 per PAIR of values does it -- after the
// rounding, which commutes with ReLU (rounding is sign-symmetric and monotone).  The fp32 form costs hipcc two
// v_max (canonicalise + max) per value: a quarter of this kernel's VALU work.
template <bool RELU>
__device__ __forceinline__ unsigned int pack2(float a, float b) {
  const float2v f = {a, b};
  short2v s = __builtin_bit_cast(short2v, __builtin_convertvector(f, bf16x2));
  if (RELU) {
    const short2v z = {0, 0};
    s = __builtin_elementwise_max(s, z);
  }
  return __builtin_bit_cast(unsigned int, s);
}


// bf16-mode activations: sigmoid / softplus on v_exp_f32 / v_log_f32 / v_rcp_f32 (~1e-6 relative) instead of the
// correctly-rounded library forms -- their results are weighed against bf16 GEMM rounding (2^-9) in this mode
__device__ __forceinline__ float fast_sigmoid(float x) {
  return __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-1.4426950408889634f * x));
}
__device__ __forceinline__ float fast_softplus(float x) {
  return x > 20.0f ? x : 0.6931471805599453f * __builtin_amdgcn_logf(1.0f + __builtin_amdgcn_exp2f(1.4426950408889634f * x));
}
// sin / cos with the exact fp32 Cody-Waite reduction of sincos_big and degree-5 / degree-6 least-squares polynomials (4e-6 / 2e-7 absolute
// on |r| <= 0.87): the feature is rounded to bf16 next
__device__ __forceinline__ float sincos_bf16(float a, int quad) {
  const float q = rintf(a * 0.63661977236758134308f);
  float r = __builtin_fmaf(-q, 1.5707963705062866f, a);
  r = __builtin_fmaf(-q, -4.371138828673793e-08f, r);
  const int n = (int)q + quad;
  const float s = r * r;
  const float ps = r + r * s * (-0.16661735f + s * 8.12778e-3f);
  const float pc = 1.0f + s * (-0.49999845f + s * (4.165309e-2f + s * -1.35546e-3f));
  const float v = (n & 1) ? pc : ps;
  return (n & 2) ? -v : v;
}


#define RING_ENC_KS ((RSN_ENC_K16 * 8 + RSN_RING_GROUP_FRAGS - 1) / RSN_RING_GROUP_FRAGS * RSN_RING_GROUP_FRAGS / 8)  // enc K-steps incl. padding
#define RING_RGB_KS (RSN_RING_GROUP_FRAGS == 16 ? 16 : 8)
#ifndef RING_FIFO
#define RING_FIFO 4   // fragments read from the ring ahead of their MFMA (registers: 4 x 4 VGPRs)
#endif
#define RING_GROUP_BYTES (RSN_RING_GROUP_FRAGS * 1024)
#define RING_STASH_BYTES (RSN_ENC_K16 * 1024)          // per wave: encoded inputs as bf16, [k16][lane][8]
#define RING_MAX_LAYERS RSN_RING_MAX_LAYERS            // trunk depth the LDS bias table is sized for
#define RING_BIAS_FLOATS (RING_MAX_LAYERS * 256 + 288 + 128 + 32)

// NW = waves per workgroup.  4: two workgroups per CU, 5-slot ring each (78 KiB);  8: one workgroup per CU whose two
// waves per SIMD share ONE stream (half the LDS-DMA pieces per MFMA, half the L2 traffic), 8-slot ring (130 KiB).
template <int NW>
struct RingCfg {
  static constexpr int SLOTS = RSN_RING_GROUP_FRAGS == 16 ? 4 : (NW == 8 ? 8 : 5);
  static constexpr int LEAD = SLOTS - 1;              // groups in flight ahead of the group being consumed
  static constexpr int PPW = RSN_RING_GROUP_FRAGS / NW;  // LDS-DMA pieces per wave and group
  static constexpr int RING_BYTES = SLOTS * RING_GROUP_BYTES;
  static constexpr int LDS_BYTES = RING_BYTES + NW * RING_STASH_BYTES + RING_BIAS_FLOATS * 4;
};

struct Ring {
  const char* src;     // wave-uniform source pointer into the stream: base + wave * PPW KiB (the lane adds lane * 16)
  unsigned lane16;     // lane * 16
  unsigned lds_dst;    // LDS byte address of this wave's pieces inside slot 0
  int n_groups;        // stream length in groups
  int issue_grp, issue_slot;   // next group to fetch and the slot it goes to
  unsigned rd_base;    // byte offset (inside smem) of this lane's 16 B in fragment 0 of slot 0
  unsigned rd_cur, rd_next;    // the same for the group being consumed / the one after it
  int next_slot;
};

// one LDS-DMA piece: 64 lanes x 16 B from gbase + voff (scalar base + 32-bit lane offset: half the address data of the
// 64-bit-VGPR form goes through the vector-memory issue path) to LDS [lds_dst, lds_dst + 1 KiB)   (M0 = LDS base)
__device__ __forceinline__ void glds16(const void* gbase, unsigned voff, unsigned lds_dst) {
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
               : "=&s"(keep)
               : "v"(voff), "s"(gbase), "s"(lds_dst)
               : "memory");
}

template <int NW>
__device__ __forceinline__ void ring_issue(Ring& r) {
  const char* g = r.src + (size_t)r.issue_grp * RING_GROUP_BYTES;
  const unsigned d = __builtin_amdgcn_readfirstlane(r.lds_dst + (unsigned)r.issue_slot * RING_GROUP_BYTES);
#pragma unroll
  for (int i = 0; i < RingCfg<NW>::PPW; ++i) glds16(g + i * 1024, r.lane16, d + i * 1024);
  r.issue_grp = (r.issue_grp + 1 == r.n_groups) ? 0 : r.issue_grp + 1;
  r.issue_slot = (r.issue_slot + 1 == RingCfg<NW>::SLOTS) ? 0 : r.issue_slot + 1;
}

// group boundary: the group about to be consumed (and the one after it) are in LDS for every wave; the previous
// group's slot is refilled.  vmcnt counts in issue order, so "all but the youngest PPW*(LEAD-2)" covers every DMA of
// the two oldest groups in flight.
// (RSN_RING_NO_*: timing diagnostics of tools/variant_bench.py -- wrong results by construction; they compile only under
// -DRSN_DIAG_BUILD, rsn_common.h, and such a library is refused as librsn_hip.so)
template <int NW>
__device__ __forceinline__ void ring_sync(Ring& r) {
#ifdef RSN_RING_NO_BARRIER
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(RingCfg<NW>::PPW * (RingCfg<NW>::LEAD - 2)) : "memory");
#elif defined(RSN_RING_NO_WAIT)
  asm volatile("s_barrier" ::: "memory");
#else
  asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" ::"n"(RingCfg<NW>::PPW * (RingCfg<NW>::LEAD - 2)) : "memory");
#endif
#ifndef RSN_RING_NO_DMA
  ring_issue<NW>(r);
#endif
  r.rd_cur = r.rd_next;
  r.next_slot = (r.next_slot + 1 == RingCfg<NW>::SLOTS) ? 0 : r.next_slot + 1;
  r.rd_next = r.rd_base + (unsigned)r.next_slot * RING_GROUP_BYTES;
}


typedef float f32x4 __attribute__((ext_vector_type(4)));
// packed row n = 16 b + 4 g + r of the 16x32 stream -> the feature it carries (rsn_pack.hip, rows_perm16)
__device__ __forceinline__ int r16_feature(int n) { return 32 * (n >> 5) + 8 * ((n >> 2) & 3) + 4 * ((n >> 4) & 1) + (n & 3); }
#define R16_STASH_BYTES (4 * 2 * 1024)  // per wave: encoded inputs, [k32 (4)][half (2)][lane][8 bf16]
#define R16_LDS_BYTES (RingCfg<8>::RING_BYTES + 8 * R16_STASH_BYTES + RING_BIAS_FLOATS * 4)

template <int NBO, int KS, int XN>
__device__ __forceinline__ void gemm_ring16(f32x4 (&acc)[NBO][2], const bf16x8 (&X)[XN][2], Ring& r,
                                            bf16x8 (&W)[RING_FIFO], const char* smem) {
  static_assert((NBO * KS) % RSN_RING_GROUP_FRAGS == 0 && KS <= XN, "a GEMM is a whole number of ring groups");
#pragma unroll
  for (int i = 0; i < NBO * KS; ++i) {
    if (i % RSN_RING_GROUP_FRAGS == 0) ring_sync<8>(r);
    const int kk = i / NBO, b = i % NBO;
    const bf16x8 wa = W[i % RING_FIFO];
    const int pos = (i % RSN_RING_GROUP_FRAGS) + RING_FIFO;
#ifndef RSN_R16_NO_LDS_READ  // timing diagnostics (tools/variant_bench.py --define): wrong results by construction
    W[i % RING_FIFO] = *reinterpret_cast<const bf16x8*>(
        smem + (pos < RSN_RING_GROUP_FRAGS ? r.rd_cur + pos * 1024 : r.rd_next + (pos - RSN_RING_GROUP_FRAGS) * 1024));
#endif
#ifndef RSN_R16_NO_MFMA
    acc[b][0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wa, X[kk][0], acc[b][0], 0, 0, 0);
    acc[b][1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wa, X[kk][1], acc[b][1], 0, 0, 0);
#else
    acc[b][0][i % 4] += (float)wa[0] * (float)X[kk][0][0];
#endif
#ifdef RSN_R16_DOUBLE_MFMA  // the MFMA work of a 64-point tile per fragment read (results wrong)
    acc[b][0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wa, X[kk][1], acc[b][0], 0, 0, 0);
    acc[b][1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wa, X[kk][0], acc[b][1], 0, 0, 0);
#endif
    __builtin_amdgcn_sched_barrier(0);
  }
}

// accumulators <- bias[16 b + 4 g + r] (LDS table; the lanes of a group read one address: broadcast)
template <int NBO>
__device__ __forceinline__ void init_acc16(f32x4 (&acc)[NBO][2], const float* bias, int g) {
#pragma unroll
  for (int b = 0; b < NBO; ++b) {
    const float4 bv = *reinterpret_cast<const float4*>(bias + b * 16 + 4 * g);
    const f32x4 v = {bv.x, bv.y, bv.z, bv.w};
    acc[b][0] = v;
    acc[b][1] = v;
  }
}

// blocks 2kk, 2kk+1 -> the next GEMM's B operand of K-step kk; with `bias` the blocks restart from the next bias
template <int NBO, int NKS, bool RELU, int XN>
__device__ __forceinline__ void acc_to_x16(f32x4 (&acc)[NBO][2], bf16x8 (&X)[XN][2], const float* bias = nullptr, int g = 0) {
#pragma unroll
  for (int kk = 0; kk < NKS; ++kk) {
#pragma unroll
    for (int p = 0; p < 2; ++p) {
      uint4v w;
      w[0] = pack2<RELU>(acc[2 * kk][p][0], acc[2 * kk][p][1]);
      w[1] = pack2<RELU>(acc[2 * kk][p][2], acc[2 * kk][p][3]);
      w[2] = pack2<RELU>(acc[2 * kk + 1][p][0], acc[2 * kk + 1][p][1]);
      w[3] = pack2<RELU>(acc[2 * kk + 1][p][2], acc[2 * kk + 1][p][3]);
      X[kk][p] = __builtin_bit_cast(bf16x8, w);
    }
    if (bias) {
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        const float4 bv = *reinterpret_cast<const float4*>(bias + (2 * kk + t) * 16 + 4 * g);
        const f32x4 v = {bv.x, bv.y, bv.z, bv.w};
        acc[2 * kk + t][0] = v;
        acc[2 * kk + t][1] = v;
      }
      if ((kk & 1) == 1) __builtin_amdgcn_sched_barrier(0);
    }
  }
}

