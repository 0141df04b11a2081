// rsn_field_x6_train.hip -- the TRAINING kernels of the split-bf16 mode (RSN_MMA_BF16X6, fp32-equivalent, width 256) on the
// LDS weight ring.
//
// Until round 4 this mode trained on rsn_field_kernel<8, true, 1> / rsn_field_bwd_kernel<8, 1>: the fp32 kernels' structure
// with the K loop on v_mfma_f32_32x32x16_bf16 -- every wave streams its own 24 KiB of pre-split weight fragments per K=16 step
// from L1 / L2 (64 B/clk/CU: the L1's limit) and splits its activations three ways per K-step in registers (3.6 vector
// instructions per MFMA): 0.42 of the bf16 issue rate, MFMA busy 50 %.  Here both sweeps get the design of
// rsn_field_bf16_train.hip with three changes:
//   * the weight stream carries every 16x32 fragment as THREE 1 KiB pieces -- the lo, mid and hi bf16 parts of the fp32 weights
//     (rsn_pack.hip, layout 2; RsnPackedLayout.q_pf = 3) -- pulled once per WORKGROUP through the 4-slot LDS ring; a piece feeds
//     1 / 2 / 3 MFMAs (lo x hi;  mid x {mid, hi};  hi x {lo, mid, hi}: the six products of bf16x6, small ones first), i.e. the
//     same two MFMAs per ds_read_b128 as the plain-bf16 loop (tools/probes/ring16_probe.hip variant C: 0.78 of the 2.5 PF
//     issue peak for the loop alone);
//   * a wave holds ONE 16-point half (128-point tiles per workgroup) and its activations stay FP32 in the accumulator registers
//     that produced them (64 VGPRs, lane-local: blocks 2kk, 2kk+1 are the lane's share of K-step kk); the B operand of K-step
//     kk + 1 -- ReLU / ReLU-mask, 3-way split (11 vector instructions per pair of values), ReLU bits -- is formed a pair of values
//     at a time BETWEEN the MFMAs of K-step kk (gemm_j: 0.6 vector instructions per MFMA, none of them in an MFMA-free phase);
//     a layer hand-off is 64 register copies.  (First version: split once per layer in an epilogue between the GEMMs -- 620 VALU
//     and 18 stores with the matrix pipe idle, both waves of a SIMD in it together: 0.52 of the issue peak, MFMA busy 60 %.)
//   * everything kept for the backward pass / the weight gradients is fp32 (this mode is fp32-equivalent: the same parity
//     bounds as the exact path): a lane's share of a row -- 8 contiguous features = 32 bytes per K-step -- leaves as two
//     non-temporal 16-byte stores at the ring's group boundaries INSIDE the GEMM that reads the row; encode, activations, SH and
//     the chain through the encoding use the exact-fp32 forms of rsn_field_kernel.h (sin_big, expf, log1pf ...), not the fast
//     ones of the bf16 mode.
// Saved-buffer layout: rsn_train_saved_layout (include/rsn.h): enc fp32 [N,128], sh fp32 [N,64] in the ring's slot order,
// ReLU bit words [L+1][N][4 lane groups][2 words], wide buffers fp32 [N,W] in natural feature order.
#include "rsn_ringt.h"

// Ring groups in flight ahead of the one being consumed.  The split-bf16 stream is 3.9 MB forward + 3.0 MB transposed -- beyond an
// XCD's 4 MiB L2 -- so a deeper ring pays in the backward where the plain-bf16 kernels saw nothing: 3 / 5 / 7 groups ahead: 7.53 /
// 6.94 / 7.10 ms per step (the input-gradient launch 2.12 -> 1.61 ms).  The forward has no LDS for more than 3 beside its stashes; with
// the stash cut to 6 KiB per wave (raw-coordinate K-step rebuilt from registers) 4 groups ahead measured the same as 3
// (profiles/r04_x6_ab.txt).
#ifndef X6_LEAD_FWD
#define X6_LEAD_FWD 3
#endif
#ifndef X6_LEAD_BWD
#define X6_LEAD_BWD 5
#endif
#define X6_STASH_BYTES (8 * 1024)   // per wave: 8 float4 per lane -- the encoded inputs [kk (4)][half (2)], later the SH inputs / the
                                    // derivative factors of the encoding

// the three bf16 parts of fp32 values (x = h + m + l exactly: 3 x 8 significant bits), as packed words
typedef uint4v P3[3];   // [0] hi, [1] mid, [2] lo: dword i = the pair of values (2 i, 2 i + 1) of a K-step's eight
#define XH_ 0
#define XM_ 1
#define XL_ 2
__device__ __forceinline__ unsigned split_pair(float a, float b, P3& o, int i) {
  const unsigned h2 = pack2<false>(a, b);
  const float ra = a - __uint_as_float(h2 << 16), rb = b - __uint_as_float(h2 & 0xffff0000u);
  const unsigned m2 = pack2<false>(ra, rb);
  const float sa = ra - __uint_as_float(m2 << 16), sb = rb - __uint_as_float(m2 & 0xffff0000u);
  o[XH_][i] = h2;
  o[XM_][i] = m2;
  o[XL_][i] = pack2<false>(sa, sb);
  return h2;
}

// acc[b] (+)= W-fragment(kk, b) * B(kk), the fragment arriving as the pieces lo, mid, hi (piece j = 3 b + s of K-step kk); six
// products, the small ones first: lo x h | mid x m, mid x h | hi x l, hi x m, hi x h.
// The B operand is formed JUST IN TIME: src(kk, i, a, b) yields the pair of fp32 values (2 i, 2 i + 1) of K-step kk (the caller's
// ReLU / mask applied); the four pairs of K-step kk + 1 are split at the quarter points of K-step kk's pieces, under its MFMAs
// (two operand buffers); note(kk, i, hi word) sees every pair's hi part (ReLU bits).  hook(group) runs behind every ring boundary.
struct NoNote {
  __device__ __forceinline__ void operator()(int, int, unsigned) const {}
};
template <int NBO, int KS, int INIT, class SRC, class NOTE, class RING, class HOOK>
__device__ __forceinline__ void gemm_j(f32x4 (&acc)[NBO], SRC&& src, NOTE&& note, RING& r, bf16x8 (&W)[RING_FIFO], const char* smem,
                                       HOOK&& hook, const float* bias = nullptr, int g = 0) {
  constexpr int PK = NBO * 3;  // pieces per K-step
  static_assert((PK * KS) % RSN_RING_GROUP_FRAGS == 0, "a GEMM is a whole number of ring groups");
  P3 P[2];
#pragma unroll
  for (int q = 0; q < 4; ++q) {  // K-step 0's operand: nothing of this GEMM to hide it under
    float a, b;
    src(0, q, a, b);
    note(0, q, split_pair(a, b, P[0], q));
  }
#pragma unroll
  for (int gi = 0; gi < PK * KS / RSN_RING_GROUP_FRAGS; ++gi) {
    ringt_sync(r);
    hook(gi);
#pragma unroll
    for (int f = 0; f < RSN_RING_GROUP_FRAGS; ++f) {
      const int i = gi * RSN_RING_GROUP_FRAGS + f;
      const int kk = i / PK, j = i % PK, b = j / 3, s = j % 3;
      const bf16x8 wa = W[i % RING_FIFO];
      const int pos = f + RING_FIFO;
      W[i % RING_FIFO] = *reinterpret_cast<const bf16x8*>(
          smem + (pos < RSN_RING_GROUP_FRAGS ? r.rd_cur + pos * 1024 : r.rd_next + (pos - RSN_RING_GROUP_FRAGS) * 1024));
#ifndef RSN_RT_NO_PREP  // (timing ablation: every K-step multiplies K-step 0's operand)
      constexpr int pb = 1;
#pragma unroll
      for (int q = 0; q < 4; ++q)
        if (kk + 1 < KS && j == (q * PK) / 4) {
          float a, b2;
          src(kk + 1, q, a, b2);
          note(kk + 1, q, split_pair(a, b2, P[(kk + 1) & 1], q));
        }
#else
      constexpr int pb = 0;
#endif
      const bf16x8 xh = __builtin_bit_cast(bf16x8, P[kk & pb][XH_]), xm = __builtin_bit_cast(bf16x8, P[kk & pb][XM_]),
                   xl = __builtin_bit_cast(bf16x8, P[kk & pb][XL_]);
      if (s == 0) {
        f32x4 c = acc[b];
        if (INIT != GI_ACC && kk == 0) {
          c = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
          if (INIT == GI_BIAS) {
            const float4 bv = *reinterpret_cast<const float4*>(bias + b * 16 + 4 * g);
            c = f32x4{bv.x, bv.y, bv.z, bv.w};
          }
        }
        acc[b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wa, xh, c, 0, 0, 0);
      } else if (s == 1) {
        acc[b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wa, xm, acc[b], 0, 0, 0);
        acc[b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wa, xh, acc[b], 0, 0, 0);
      } else {
        acc[b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wa, xl, acc[b], 0, 0, 0);
        acc[b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wa, xm, acc[b], 0, 0, 0);
        acc[b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wa, xh, acc[b], 0, 0, 0);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
  }
}

// a lane's four fp32 values of a saved row (counted, non-temporal like st16)
template <class RING>
__device__ __forceinline__ void st16f(const RowD& d, unsigned voff, unsigned soff, float a, float b, float c, float e, RING& r) {
#ifndef RSN_RT_NO_STORES
  const u32x4t v = {__float_as_uint(a), __float_as_uint(b), __float_as_uint(c), __float_as_uint(e)};
  // the constant part goes into the instruction's immediate offset, the scalar offset stays 0: with a REGISTER soffset the
  // compiler's hazard recogniser assumes that a VALU may overwrite the store's data registers in the very next instruction --
  // on gfx950 it may not (measured: the last quarter of every 16 lanes stored the overwriting value); see st16 (rsn_ringt.h)
  __builtin_amdgcn_raw_buffer_store_b128(v, d.r, voff + soff, 0, RT_STORE_AUX);
#ifndef RSN_RT_UNCOUNTED
  r.c0 += 1;
  r.since += 1;
#endif
#endif
}
__device__ __forceinline__ void tie1(u32x2t& a) { asm volatile("" : "+v"(a)::"memory"); }

// The row piece of K-step kk (the lane's features 32 kk + 8 g .. + 7 of its point: 32 bytes at 128 kk + 32 g of the row) leaves ...
template <class SRC, class RING>
__device__ __forceinline__ void store_kstep(const RowD& d, unsigned voff, int kk, SRC&& src, RING& r) {
  float v[8];
#pragma unroll
  for (int i = 0; i < 4; ++i) src(kk, i, v[2 * i], v[2 * i + 1]);
  // (opaque here: otherwise hipcc hoists the sixteen `voff + constant` of a layer out of the layer loop -- where they can no longer
  // become the store's immediate offset -- and carries them through scratch: 60 spill stores + 48 reloads per tile)
  asm volatile("" : "+v"(voff));
  st16f(d, voff, kk * 128, v[0], v[1], v[2], v[3], r);
  st16f(d, voff, kk * 128 + 16, v[4], v[5], v[6], v[7], r);
}
// ... at the group boundary in front of its first piece, inside the GEMM that reads the row (PK pieces per K-step)
template <int PK, int KS, class SRC, class RING>
__device__ __forceinline__ void hook_rows(int gi, const RowD& d, unsigned voff, SRC&& src, RING& r) {
#pragma unroll
  for (int kk = 0; kk < KS; ++kk)
    if ((kk * PK) / RSN_RING_GROUP_FRAGS == gi) store_kstep(d, voff, kk, src, r);
}

// sources of a GEMM's B operand from the accumulators of the GEMM before it: blocks 2 kk, 2 kk + 1 hold K-step kk's eight values
template <bool RELU, int NB>
struct SrcAcc {
  const f32x4 (&A)[NB];
  __device__ __forceinline__ void operator()(int kk, int i, float& a, float& b) const {
    const f32x4 v = A[2 * kk + (i >> 1)];
    a = v[2 * (i & 1)];
    b = v[2 * (i & 1) + 1];
    if (RELU) {
      a = relu_f(a);
      b = relu_f(b);
    }
  }
};
// a layer gradient: masked by the ReLU bits of the layer it enters (bit layout: BitsNote below)
template <int NB>
struct SrcMasked {
  const f32x4 (&A)[NB];
  const unsigned (&bits)[2];
  __device__ __forceinline__ void operator()(int kk, int i, float& a, float& b) const {
    const f32x4 v = A[2 * kk + (i >> 1)];
    const int j = (kk & 3) * 4 + i;
#ifdef RSN_RT_NO_BITS
    const int ma = -1, mb = -1;
#else
    const int ma = __builtin_amdgcn_sbfe((int)bits[kk >> 2], (unsigned)(15 - j), 1u);  // 0 or -1
    const int mb = __builtin_amdgcn_sbfe((int)bits[kk >> 2], (unsigned)(31 - j), 1u);
#endif
    a = __uint_as_float(__float_as_uint(v[2 * (i & 1)]) & (unsigned)ma);
    b = __uint_as_float(__float_as_uint(v[2 * (i & 1) + 1]) & (unsigned)mb);
  }
};
// ReLU bits of post-ReLU values from the hi parts of their splits (hi > 0 <=> value > 0): the pair i of K-step kk is word
// j = 4 (kk & 3) + i of bits[kk >> 2]: (low half > 0) at bit 15 - j, (high half > 0) at bit 31 - j -- the layout of
// rsn_field_bf16_train.hip.  Pairs arrive in ascending (kk, i).
struct BitsNote {
  unsigned (&bw)[2];
  __device__ __forceinline__ void operator()(int kk, int i, unsigned h2) const {
#ifdef RSN_RT_NO_BITS
    bw[kk >> 2] = 0xffffffffu;
#else
    const unsigned t = pk_min_u16(h2, 0x00010001u);
    bw[kk >> 2] = ((kk & 3) == 0 && i == 0) ? t : ((bw[kk >> 2] << 1) | t);
#endif
  }
};

// Gradient w.r.t. this lane's encoded inputs (eacc: packed rows 16 b + 4 g + r = slot (kk = b / 2, e = 4 (b % 2) + r); slot
// u = 8 kk + e: u < 12 the sine feature of (coordinate u / 4, frequency 4 g + u % 4), 12 <= u < 24 its cosine feature, 24..26 the
// raw coordinates on g == 0) folded with per-slot factors fa[u] (same slots):
//   NORMALS: fa = d feature / d (2 pi f x) = e cos(angle):  d raw_density / d x_c += 2 pi f (g_sin fa_sin + g_cos fa_cos)  [+ raw slot]
//   else   : fa = the feature itself:                        d loss / d var_c      += -f^2 / 2 (g_sin fa_sin + g_cos fa_cos)
template <bool NORMALS>
__device__ __forceinline__ void fold_enc3(const f32x4 (&eacc)[8], const float (&fa)[24], const float (&fq)[4], float (&part)[3],
                                          float (&raw)[3]) {
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    float s = 0.0f;
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      const int us = 4 * c + t, uc = 12 + 4 * c + t;
      const float gs = eacc[2 * (us >> 3) + ((us & 7) >> 2)][us & 3];
      const float gc = eacc[2 * (uc >> 3) + ((uc & 7) >> 2)][uc & 3];
      s += (NORMALS ? fq[t] : fq[t] * fq[t]) * (gs * fa[us] + gc * fa[uc]);
    }
    part[c] += s;
    if (NORMALS) raw[c] += eacc[6][c];  // slot u = 24 + c: kk = 3, e = c (meaningful on g == 0 only)
  }
}
template <int NB>
__device__ __forceinline__ void copy_acc(f32x4 (&dst)[NB], const f32x4 (&src)[NB]) {
#pragma unroll
  for (int b = 0; b < NB; ++b) dst[b] = src[b];
}

// ================================================================================================ training forward
template <bool NORMALS>
__global__ __launch_bounds__(512, 2) void rsn_field_x6_train_kernel(const FieldJobs J) {
  constexpr int W = 256;
  constexpr int RB = RT_RING_BYTES(X6_LEAD_FWD, 0);
  __shared__ __attribute__((aligned(1024))) char smem[RB + 8 * X6_STASH_BYTES + RT_TABLE_FLOATS * 4];
  const int lane = threadIdx.x & 63;
  const int wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  float4* ST = reinterpret_cast<float4*>(smem + RB + wid * X6_STASH_BYTES) + lane;   // slot i of this lane: ST[i * 64]
  float* bias = reinterpret_cast<float*>(smem + RB + 8 * X6_STASH_BYTES);
  const float* b_bh = bias + RING_MAX_LAYERS * 256;
  const float* b_mid = b_bh + 288;
  const float* b_rgb = b_mid + 128;
  const float* vden = b_rgb + 32;  // density-head row, natural feature order

  const FieldShared& P = J.s;
  const TileJobs T = tile_space<128>(J);
  if ((long long)blockIdx.x >= T.n_tiles) return;  // workgroup-uniform
  const float* __restrict__ pk = P.packed;
  const int L = P.num_layers;

  for (int i = threadIdx.x; i < L * 256; i += 512) bias[i] = pk[P.L.b[i >> 8] + r16_feature(i & 255)];
  for (int i = threadIdx.x; i < 288; i += 512) bias[RING_MAX_LAYERS * 256 + i] = pk[P.L.b_bh + (i < 256 ? r16_feature(i) : i)];
  if (threadIdx.x < 128) bias[RING_MAX_LAYERS * 256 + 288 + threadIdx.x] = pk[P.L.b_mid + r16_feature(threadIdx.x)];
  if (threadIdx.x < 32) bias[RING_MAX_LAYERS * 256 + 288 + 128 + threadIdx.x] = pk[P.L.b_rgb + threadIdx.x];
  if (threadIdx.x < 256) bias[RING_BIAS_FLOATS + threadIdx.x] = pk[P.L.v_density + threadIdx.x];

  RingT<X6_LEAD_FWD, 0> r;
  bf16x8 Wf[RING_FIFO];
  // the walk: forward stream [0, q_groups); with the normal sweep then [t_g_trunk, t_g_end) of the transposed stream; again
  ring_start(r, pk, P.L, smem, wid, lane, 0, P.L.q_groups, NORMALS ? P.L.t_g_trunk : 0, NORMALS ? P.L.t_g_end : -1, 0, Wf);

  for (long long gtile = blockIdx.x; gtile < T.n_tiles; gtile += gridDim.x) {
    const int jk = (gtile >= T.tb1 ? 1 : 0) + (gtile >= T.tb2 ? 1 : 0);  // workgroup-uniform
    const FieldJob& a = J.j[jk];
    const unsigned n_points = (unsigned)(jk == 0 ? T.np0 : (jk == 1 ? T.np1 : T.np2));
    const unsigned tile = (unsigned)(gtile - (jk == 0 ? 0 : (jk == 1 ? T.tb1 : T.tb2)));
    const unsigned p0 = tile * 128 + wid * 16;   // every wave walks every tile (barriers, DMA shares); rows = 0 past the end
    const int rows = p0 >= n_points ? 0 : (int)(n_points - p0 < 16u ? n_points - p0 : 16u);
    const long long n_max = a.act_stride / W;   // points the saved buffers are sized for
    int ln = lane;
    asm volatile("" : "+v"(ln));  // opaque per-tile lane id (see rsn_field_bf16.hip)
    const int m = ln & 15, g = ln >> 4;
    const unsigned pt = p0 + m;
    const bool valid = pt < n_points;
    const size_t pc = valid ? pt : (n_points ? n_points - 1 : 0);
    const unsigned vrow = (unsigned)m;  // the lane's row inside the wave's 16-row tile

    // ---------------- encode (exact fp32, as rsn_field_kernel.h): the four lanes of a point share its Gaussian, lane group g
    //                  owns frequencies 4g .. 4g+3: slots u = 8 kk + e: 12 sine, 12 cosine features, 3 raw coordinates (g == 0) ---
    float mc[3] = {0.0f, 0.0f, 0.0f}, vc[3] = {0.0f, 0.0f, 0.0f}, vd[3] = {0.0f, 0.0f, 0.0f};
    bool has_dir = true;
    if (a.mode == RSN_MODE_FRUSTUM) {
      const unsigned rayu = (unsigned)pc / (unsigned)a.S;
      const int s = (int)((unsigned)pc - rayu * (unsigned)a.S);
      const size_t ray = rayu;
      float o[3];
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        o[c] = a.origins[ray * 3 + c];
        vd[c] = a.directions[ray * 3 + c];
      }
      frustum_to_contracted(o, vd, a.pixel_area[ray], a.bins[ray * (a.S + 1) + s], a.bins[ray * (a.S + 1) + s + 1], mc, vc);
    } else {  // RSN_MODE_INF
      const float r2 = a.sqradius[pc];
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        vd[c] = a.directions[pc * 3 + c];
        mc[c] = 2.0f * vd[c];
        vc[c] = (0.6f * r2) * (1.0f - vd[c] * vd[c]);
      }
      has_dir = false;  // SH inputs are zeroed (reflect_sampling_nerf_field.py:199)
    }
    float fq[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) fq[t] = P.freqs[4 * g + t];

    auto d_act = [&](int l) { return rowd(a.saved.act, ((long long)l * n_max + p0) * 1024, rows, 1024); };
    auto d_bits = [&](int l) { return rowd(a.saved.relu_bits, ((long long)l * n_max + p0) * 32, rows, 32); };
    // the encoded inputs / SH inputs as a GEMM source: the lane's float4 slots 2 kk, 2 kk + 1 of the stash
    auto src_stash = [&](int kk, int i, float& x, float& y) {
      const float4 q = ST[(2 * kk + (i >> 1)) * 64];
      x = (i & 1) ? q.z : q.x;
      y = (i & 1) ? q.w : q.y;
    };

    f32x4 A[16];      // the pre-activation of the trunk layer just finished (ReLU is applied where it is consumed)
    unsigned bwe[2];  // the embedding's ReLU bits: the seed mask of the normal sweep
    // ---------------- trunk -----------------
    {
      {
        float feat[32];
#pragma unroll
        for (int c = 0; c < 3; ++c) {
          const float sx = 6.283185307179586f * mc[c];
#pragma unroll
          for (int t = 0; t < 4; ++t) {
            const float f = fq[t];
            const float ang = sx * f;
            const float e = expf(-0.5f * (vc[c] * (f * f)));
            feat[c * 4 + t] = e * sin_big(ang);
            feat[12 + c * 4 + t] = e * sin_big(ang + 1.5707963267948966f);
          }
        }
#pragma unroll
        for (int c = 0; c < 3; ++c) feat[24 + c] = g == 0 ? mc[c] : 0.0f;
#pragma unroll
        for (int u = 27; u < 32; ++u) feat[u] = 0.0f;
        const RowD d_enc = rowd(a.saved.enc, (long long)p0 * 512, rows, 512);   // fp32 [N,128]
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
          ST[(2 * kk) * 64] = make_float4(feat[8 * kk], feat[8 * kk + 1], feat[8 * kk + 2], feat[8 * kk + 3]);
          ST[(2 * kk + 1) * 64] = make_float4(feat[8 * kk + 4], feat[8 * kk + 5], feat[8 * kk + 6], feat[8 * kk + 7]);
          st16f(d_enc, vrow * 512 + 32 * g, kk * 128, feat[8 * kk], feat[8 * kk + 1], feat[8 * kk + 2], feat[8 * kk + 3], r);
          st16f(d_enc, vrow * 512 + 32 * g, kk * 128 + 16, feat[8 * kk + 4], feat[8 * kk + 5], feat[8 * kk + 6], feat[8 * kk + 7], r);
        }
        gemm_j<16, 4, GI_BIAS>(A, [&](int kk, int i, float& x, float& y) { x = feat[8 * kk + 2 * i]; y = feat[8 * kk + 2 * i + 1]; },
                               NoNote(), r, Wf, smem, NoHook(), bias, g);
      }
#pragma unroll 1
      for (int l = 1; l < L; ++l) {
        f32x4 B[16];
        unsigned bw[2] = {0u, 0u};
        const RowD da = d_act(l - 1);
        const SrcAcc<true, 16> sa{A};   // act[l-1] = ReLU(A): its rows leave from the GEMM that reads it, one K-step per three groups
        gemm_j<16, 8, GI_BIAS>(B, sa, BitsNote{bw}, r, Wf, smem,
                               [&](int gi) { hook_rows<48, 8>(gi, da, vrow * 1024 + 32 * g, sa, r); }, bias + l * 256, g);
        st8(d_bits(l - 1), vrow * 32 + 8 * g, 0, bw[0], bw[1], r);
        if (l == P.skip_layer) gemm_j<16, 4, GI_ACC>(B, src_stash, NoNote(), r, Wf, smem, NoHook());
        copy_acc(A, B);
      }
    }

    // ---------------- heads: one 16-row block (+ a zero block) on the embedding = ReLU(A) -----------------
    float dcol[3];
    {
      f32x4 acch[2];
      gemm_j<2, 8, GI_BIAS>(acch, SrcAcc<true, 16>{A}, BitsNote{bwe}, r, Wf, smem, NoHook(), b_bh + 256, g);
      const float r0 = acch[0][0], r1 = acch[0][1], r2 = acch[0][2], r3 = acch[0][3];
      // g == 0: r0 raw density, r1..r3 normals;  g == 1: r0..r2 diff;  g == 2: r0 roughness;  g == 3: r0..r2 tint
      const float rough_raw = __shfl(r0, 32 + m, 64);
      const float rho = softplus_f(rough_raw);
      if (g == 0) {  // SH-34 of the view direction: one lane of the point's four writes the slots of all four groups
        float sh[36];
        if (has_dir) {
          sh34_attenuated(vd[0], vd[1], vd[2], rho, sh);
        } else {
#pragma unroll
          for (int i = 0; i < 34; ++i) sh[i] = 0.0f;
        }
        sh[34] = 0.0f; sh[35] = 0.0f;
#pragma unroll
        for (int gq = 0; gq < 4; ++gq) {  // lane group gq owns components 9 gq .. 9 gq + 8 in its slots u = 8 kk + e < 9
          ST[0 * 64 + gq * 16] = make_float4(sh[9 * gq], sh[9 * gq + 1], sh[9 * gq + 2], sh[9 * gq + 3]);
          ST[1 * 64 + gq * 16] = make_float4(sh[9 * gq + 4], sh[9 * gq + 5], sh[9 * gq + 6], sh[9 * gq + 7]);
          ST[2 * 64 + gq * 16] = make_float4(sh[9 * gq + 8], 0.0f, 0.0f, 0.0f);
          ST[3 * 64 + gq * 16] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
        }
      }
      // the other lanes of the point read these slots: hipcc sees only this lane's addresses (constant offsets it can tell apart) and
      // would be free to move a later read of ST[k * 64] above the writes; the wave's LDS operations execute in program order
      asm volatile("" ::: "memory");
      __builtin_amdgcn_wave_barrier();
      dcol[0] = sigmoid_f(r0); dcol[1] = sigmoid_f(r1); dcol[2] = sigmoid_f(r2);
      if (valid) {
        const size_t q = pc;
        if (g == 0) {
          if (a.saved.heads) { a.saved.heads[q * 8 + 0] = r1; a.saved.heads[q * 8 + 1] = r2; a.saved.heads[q * 8 + 2] = r3; }
          if (a.mode != RSN_MODE_INF) {
            float nrm = fmaxf(sqrtf(r1 * r1 + r2 * r2 + r3 * r3), 1e-12f);
            float nx = -(r1 / nrm), ny = -(r2 / nrm), nz = -(r3 / nrm);
            nrm = fmaxf(sqrtf(nx * nx + ny * ny + nz * nz), 1e-12f);
            nx /= nrm; ny /= nrm; nz /= nrm;
            if (a.out.sigma) a.out.sigma[q] = softplus_f(r0 + P.density_bias);
            if (a.out.raw_density) a.out.raw_density[q] = r0;
            if (a.out.pred_normals) {
              a.out.pred_normals[q * 3 + 0] = nx;
              a.out.pred_normals[q * 3 + 1] = ny;
              a.out.pred_normals[q * 3 + 2] = nz;
            }
            if (a.out.n_dot_d) a.out.n_dot_d[q] = vd[0] * nx + vd[1] * ny + vd[2] * nz;
          }
        } else if (g == 1) {
          if (a.mode != RSN_MODE_INF && a.out.diff) {
            a.out.diff[q * 3 + 0] = dcol[0]; a.out.diff[q * 3 + 1] = dcol[1]; a.out.diff[q * 3 + 2] = dcol[2];
          }
        } else if (g == 2) {
          if (a.saved.heads) a.saved.heads[q * 8 + 3] = r0;
          if (a.mode != RSN_MODE_INF) {
            if (a.out.roughness) a.out.roughness[q] = sigmoid_f(r0);
            if (a.out.raw_roughness) a.out.raw_roughness[q] = r0;
          }
        } else {
          if (a.mode != RSN_MODE_INF && a.out.tint) {
            a.out.tint[q * 3 + 0] = dcol[0]; a.out.tint[q * 3 + 1] = dcol[1]; a.out.tint[q * 3 + 2] = dcol[2];
          }
        }
      }
    }
    // ---------------- bottleneck (the embedding's rows and bits leave from this GEMM), mlp_mid, RGB head -----------------
    {
      f32x4 Bt[16];  // bottleneck output (no activation): the x-part of mlp_mid's input
      {
        const RowD da = d_act(L - 1);
        const SrcAcc<true, 16> sa{A};
        gemm_j<16, 8, GI_BIAS>(Bt, sa, NoNote(), r, Wf, smem, [&](int gi) {
          hook_rows<48, 8>(gi, da, vrow * 1024 + 32 * g, sa, r);
          if (gi == 0) st8(d_bits(L - 1), vrow * 32 + 8 * g, 0, bwe[0], bwe[1], r);
        }, b_bh, g);
      }
      f32x4 accm[8];
      {
        const RowD d_sh = rowd(a.saved.sh, (long long)p0 * 256, rows, 256);   // fp32 [N,64]
        gemm_j<8, 2, GI_BIAS>(accm, src_stash, NoNote(), r, Wf, smem,
                              [&](int gi) { hook_rows<24, 2>(gi, d_sh, vrow * 256 + 32 * g, src_stash, r); }, b_mid, g);
        const RowD d_bott = rowd(a.saved.bott, (long long)p0 * 1024, rows, 1024);
        const SrcAcc<false, 16> sb{Bt};
        gemm_j<8, 8, GI_ACC>(accm, sb, NoNote(), r, Wf, smem, [&](int gi) { hook_rows<24, 8>(gi, d_bott, vrow * 1024 + 32 * g, sb, r); });
      }
      f32x4 accr[4];  // block 0 carries the RGB rows 4..6; blocks 1..3 are whole-group padding
      {
        const float4 bv = *reinterpret_cast<const float4*>(b_rgb + 4 * g);
        const f32x4 z = {0.0f, 0.0f, 0.0f, 0.0f};
        accr[0] = f32x4{bv.x, bv.y, bv.z, bv.w};
#pragma unroll
        for (int b = 1; b < 4; ++b) accr[b] = z;
      }
      unsigned bwh[2] = {0u, 0u};
      {
        const RowD d_hid = rowd(a.saved.hid, (long long)p0 * 512, rows, 512);   // hidden (128): K-steps 0..3 = ReLU(accm)
        const SrcAcc<true, 8> sh_{accm};
        gemm_j<4, 4, GI_ACC>(accr, sh_, BitsNote{bwh}, r, Wf, smem, [&](int gi) { hook_rows<12, 4>(gi, d_hid, vrow * 512 + 32 * g, sh_, r); });
        st8(d_bits(L), vrow * 32 + 8 * g, 0, bwh[0], 0u, r);
      }
      const float m0 = sigmoid_f(accr[0][0]), m1 = sigmoid_f(accr[0][1]), m2 = sigmoid_f(accr[0][2]);
      const float t0 = __shfl(dcol[0], 48 + m, 64), t1 = __shfl(dcol[1], 48 + m, 64), t2 = __shfl(dcol[2], 48 + m, 64);
      if (g == 1 && valid) {
        const size_t q = pc;
        if (a.saved.heads) *reinterpret_cast<float4*>(a.saved.heads + q * 8 + 4) = make_float4(m0, m1, m2, 0.0f);
        if (a.out.color) {
          if (a.mode == RSN_MODE_INF) {
            a.out.color[q * 3 + 0] = m0; a.out.color[q * 3 + 1] = m1; a.out.color[q * 3 + 2] = m2;
          } else {
            a.out.color[q * 3 + 0] = dcol[0] + t0 * m0;
            a.out.color[q * 3 + 1] = dcol[1] + t1 * m1;
            a.out.color[q * 3 + 2] = dcol[2] + t2 * m2;
          }
        }
      }
    }

    // ---------------- analytic normals: -normalize(d raw_density / d contracted mean) -----------------
#ifdef RSN_RT_NO_SWEEP
    if (false) {
#else
    if (NORMALS) {
#endif
      // derivative factors of this lane's 24 features w.r.t. their angle, exactly as autograd forms them (the "cosine" features are
      // sines of the ROUNDED angle + pi / 2): parked in the stash (free now) for the two folds of the sweep
      {
        float df[24];
#pragma unroll
        for (int c = 0; c < 3; ++c) {
          // (opaque copies: hipcc otherwise forms these cosines beside the sines of the encode -- they share the argument reduction
          // -- and carries ~100 registers of them across the whole forward through scratch: 259 -> 163 spilled registers)
          float xm = mc[c], xv = vc[c];
          asm volatile("" : "+v"(xm), "+v"(xv));
          const float sx = 6.283185307179586f * xm;
#pragma unroll
          for (int t = 0; t < 4; ++t) {
            const float f = fq[t];
            const float ang = sx * f;
            const float e = expf(-0.5f * (xv * (f * f)));
            df[c * 4 + t] = e * cos_big(ang);
            df[12 + c * 4 + t] = e * cos_big(ang + 1.5707963267948966f);
          }
        }
#pragma unroll
        for (int i = 0; i < 6; ++i) ST[i * 64] = make_float4(df[4 * i], df[4 * i + 1], df[4 * i + 2], df[4 * i + 3]);
      }
      // G: the gradient entering layer l's output BEFORE that layer's ReLU mask bm (applied where it is consumed).  Seed: the
      // density-head row, masked by the embedding's ReLU
      f32x4 G[16];
#pragma unroll
      for (int kk = 0; kk < 8; ++kk) {
        const float4 lo = *reinterpret_cast<const float4*>(vden + 32 * kk + 8 * g);
        const float4 hi = *reinterpret_cast<const float4*>(vden + 32 * kk + 8 * g + 4);
        G[2 * kk] = f32x4{lo.x, lo.y, lo.z, lo.w};
        G[2 * kk + 1] = f32x4{hi.x, hi.y, hi.z, hi.w};
      }
      unsigned bm[2] = {bwe[0], bwe[1]};
      float part[3] = {0.0f, 0.0f, 0.0f}, raw[3] = {0.0f, 0.0f, 0.0f};
      auto enc_part = [&]() {  // eacc = (encoded-input part)^T x gradient, folded at once with the derivative factors
        f32x4 eacc[8];
        gemm_j<8, 8, GI_ZERO>(eacc, SrcMasked<16>{G, bm}, NoNote(), r, Wf, smem, NoHook());
        float df[24];
#pragma unroll
        for (int i = 0; i < 6; ++i) {
          const float4 q = ST[i * 64];
          df[4 * i] = q.x; df[4 * i + 1] = q.y; df[4 * i + 2] = q.z; df[4 * i + 3] = q.w;
        }
        fold_enc3<true>(eacc, df, fq, part, raw);
      };
#pragma unroll 1
      for (int l = L - 1; l >= 1; --l) {
        if (l == P.skip_layer) enc_part();
        const AsyncD db = asyncd(a.saved.relu_bits, ((long long)(l - 1) * n_max + p0) * 32, rows, 32);
        u32x2t b0 = ald8(db, vrow * 32 + 8 * g);
        r.since = 0;
        f32x4 acc[16];
        gemm_j<16, 8, GI_ZERO>(acc, SrcMasked<16>{G, bm}, NoNote(), r, Wf, smem, NoHook());
        wait_loads(r);
        tie1(b0);
        bm[0] = b0.x; bm[1] = b0.y;
        copy_acc(G, acc);
      }
      enc_part();
      float nrm[3];
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        float tot = 6.283185307179586f * part[c] + (g == 0 ? raw[c] : 0.0f);
        tot += __shfl_xor(tot, 16, 64);
        tot += __shfl_xor(tot, 32, 64);
        nrm[c] = tot;
      }
      if (g == 0 && valid && a.saved.normals) {
        const float len = fmaxf(sqrtf(nrm[0] * nrm[0] + nrm[1] * nrm[1] + nrm[2] * nrm[2]), 1e-12f);
        a.saved.normals[pc * 3 + 0] = -(nrm[0] / len);
        a.saved.normals[pc * 3 + 1] = -(nrm[1] / len);
        a.saved.normals[pc * 3 + 2] = -(nrm[2] / len);
      }
    }
  }
  ring_finish(r, wid);
}

// ================================================================================================ backward sweep
template <bool INPUT>
__global__ __launch_bounds__(512, 2) void rsn_field_x6_bwd_kernel(const BwdJobs J) {
  constexpr int W = 256;
  __shared__ __attribute__((aligned(1024))) char smem[RT_RING_BYTES(X6_LEAD_BWD, 0)];
  const int lane = threadIdx.x & 63;
  const int wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const BwdShared& P = J.s;
  const TileJobs T = tile_space<128>(J);
  if ((long long)blockIdx.x >= T.n_tiles) return;  // workgroup-uniform
  const float* __restrict__ pk = P.packed;
  const int L = P.num_layers;

  RingT<X6_LEAD_BWD, 0> r;
  bf16x8 Wf[RING_FIFO];
  // the walk: the whole transposed stream; without an input gradient the two encoded-input pieces are jumped over
  {
    const RsnPackedLayout& Y = P.L;
    int e0, j0, e1, j1;
    if (INPUT) { e0 = Y.t_g_end; j0 = Y.t_g_begin; e1 = -1; j1 = 0; }
    else if (Y.t_g_encskip >= 0) { e0 = Y.t_g_encskip; j0 = Y.t_g_encskip + 4 * Y.q_pf; e1 = Y.t_g_enc0; j1 = Y.t_g_begin; }
    else { e0 = Y.t_g_enc0; j0 = Y.t_g_begin; e1 = -1; j1 = 0; }
    ring_start(r, pk, Y, smem, wid, lane, Y.t_g_begin, e0, j0, e1, j1, Wf);
  }

  for (long long gtile = blockIdx.x; gtile < T.n_tiles; gtile += gridDim.x) {
    const int jk = (gtile >= T.tb1 ? 1 : 0) + (gtile >= T.tb2 ? 1 : 0);  // workgroup-uniform
    const BwdJob& a = J.j[jk];
    const unsigned n_points = (unsigned)(jk == 0 ? T.np0 : (jk == 1 ? T.np1 : T.np2));
    const unsigned tile = (unsigned)(gtile - (jk == 0 ? 0 : (jk == 1 ? T.tb1 : T.tb2)));
    const unsigned p0 = tile * 128 + wid * 16;
    const int rows = p0 >= n_points ? 0 : (int)(n_points - p0 < 16u ? n_points - p0 : 16u);
    const long long n_max = a.act_stride / W;
    int ln = lane;
    asm volatile("" : "+v"(ln));
    const int m = ln & 15, g = ln >> 4;
    const unsigned vrow = (unsigned)m;
    auto d_dy = [&](int l) { return rowd(a.gout.dy, ((long long)l * n_max + p0) * 1024, rows, 1024); };

    // ---------------- per-sample epilogue gradients (reference autograd restated: see rsn_field_bwd.hip) -----------------
    float dz[3];                                       // RGB-head pre-activation gradient
    float4 qh = make_float4(0.0f, 0.0f, 0.0f, 0.0f);  // heads pre-activation gradients, rows 4 g + r
    {
      const unsigned pt = p0 + m;
      const bool valid = pt < n_points;
      const size_t q = valid ? pt : (n_points ? n_points - 1 : 0);
      const float live = valid ? 1.0f : 0.0f;
      float gcol[3] = {0.0f, 0.0f, 0.0f};
      if (a.gin.color) {
#pragma unroll
        for (int c = 0; c < 3; ++c) gcol[c] = a.gin.color[q * 3 + c] * live;
      }
      const float4 hd = *reinterpret_cast<const float4*>(a.saved.heads + q * 8);       // n_raw(3), rough_raw
      const float4 md = *reinterpret_cast<const float4*>(a.saved.heads + q * 8 + 4);   // mid RGB (3)
      const float mid[3] = {md.x, md.y, md.z};
      float dif[3] = {0.0f, 0.0f, 0.0f}, tin[3] = {1.0f, 1.0f, 1.0f};
      if (a.mode != RSN_MODE_INF) {
#pragma unroll
        for (int c = 0; c < 3; ++c) {
          dif[c] = a.fwd.diff[q * 3 + c];
          tin[c] = a.fwd.tint[q * 3 + c];
        }
      }
#pragma unroll
      for (int c = 0; c < 3; ++c) dz[c] = gcol[c] * tin[c] * (mid[c] * (1.0f - mid[c]));
      if (g == 1 && valid && a.gout.dz_rgb) *reinterpret_cast<float4*>(a.gout.dz_rgb + q * 4) = make_float4(dz[0], dz[1], dz[2], 0.0f);
      if (a.mode != RSN_MODE_INF) {
        if (g == 0) {
          const long long ray = (long long)(q / (unsigned)a.S);
          const float rawd = a.fwd.raw_density[q];
          const float gs = a.gin.sigma ? a.gin.sigma[q] * live : 0.0f;
          qh.x = gs * sigmoid_f(rawd + P.density_bias);  // softplus'
          float dir[3], G3[3] = {0.0f, 0.0f, 0.0f};
#pragma unroll
          for (int c = 0; c < 3; ++c) dir[c] = a.directions[ray * 3 + c];
          if (a.gin.pred_normals) {
#pragma unroll
            for (int c = 0; c < 3; ++c) G3[c] = a.gin.pred_normals[q * 3 + c] * live;
          }
          float gd = a.gin.n_dot_d ? a.gin.n_dot_d[q] * live : 0.0f;
          if (a.gin.ray_pn_loss || a.gin.ray_ori_loss) {  // fused normal losses (model.py:403-407)
            const float w = a.gin.weights[q] * live;
            if (a.gin.ray_pn_loss) {
              const float gw = a.gin.ray_pn_loss[ray] * w * -2.0f;
#pragma unroll
              for (int c = 0; c < 3; ++c) G3[c] += gw * (a.saved.normals[q * 3 + c] - a.fwd.pred_normals[q * 3 + c]);
            }
            if (a.gin.ray_ori_loss) gd += a.gin.ray_ori_loss[ray] * w * (2.0f * fmaxf(a.fwd.n_dot_d[q], 0.0f));
          }
#pragma unroll
          for (int c = 0; c < 3; ++c) G3[c] += gd * dir[c];
          const float nraw[3] = {hd.x, hd.y, hd.z};
          const float len = fmaxf(sqrtf(nraw[0] * nraw[0] + nraw[1] * nraw[1] + nraw[2] * nraw[2]), 1e-12f);
          const float v[3] = {-(nraw[0] / len), -(nraw[1] / len), -(nraw[2] / len)};
          float gv[3], gu[3], gn[3];
          normalize_bwd(v, G3, gv);
          gu[0] = -gv[0]; gu[1] = -gv[1]; gu[2] = -gv[2];
          normalize_bwd(nraw, gu, gn);
          qh.y = gn[0]; qh.z = gn[1]; qh.w = gn[2];
        } else if (g == 1) {
          qh.x = gcol[0] * (dif[0] * (1.0f - dif[0]));
          qh.y = gcol[1] * (dif[1] * (1.0f - dif[1]));
          qh.z = gcol[2] * (dif[2] * (1.0f - dif[2]));
        } else if (g == 2) {
          const float sr = sigmoid_f(hd.w);
          const float gr = a.gin.roughness ? a.gin.roughness[q] * live : 0.0f;
          qh.x = gr * sr * (1.0f - sr);
        } else {
          qh.x = gcol[0] * mid[0] * (tin[0] * (1.0f - tin[0]));
          qh.y = gcol[1] * mid[1] * (tin[1] * (1.0f - tin[1]));
          qh.z = gcol[2] * mid[2] * (tin[2] * (1.0f - tin[2]));
        }
      }
      if (valid && a.gout.dz_heads) *reinterpret_cast<float4*>(a.gout.dz_heads + q * 16 + 4 * g) = qh;
    }

    // ---------------- stage 1: d hidden = W_rgb^T dz (before the mid hidden layer's ReLU mask: applied where it is consumed) -------
    f32x4 G8[8];
    unsigned bmh[2];
    {
      const AsyncD db = asyncd(a.saved.relu_bits, ((long long)L * n_max + p0) * 32, rows, 32);
      u32x2t b0 = ald8(db, vrow * 32 + 8 * g);
      r.since = 0;
      gemm_j<8, 2, GI_ZERO>(G8, [&](int kk, int i, float& x, float& y) {  // K-step 0 slot (g = 1, e = 0..2) = dz; K-step 1 zero
        x = (kk == 0 && g == 1) ? (i == 0 ? dz[0] : (i == 1 ? dz[2] : 0.0f)) : 0.0f;
        y = (kk == 0 && g == 1 && i == 0) ? dz[1] : 0.0f;
      }, NoNote(), r, Wf, smem, NoHook());
      wait_loads(r);
      tie1(b0);
      bmh[0] = b0.x; bmh[1] = 0u;
    }
    // ---------------- stage 2: d bottleneck = W_mid[:, 34:]^T d a_mid (the d a_mid rows leave here) -----------------
    f32x4 Gb[16];
    {
      const RowD d_damid = rowd(a.gout.da_mid, (long long)p0 * 512, rows, 512);
      const SrcMasked<8> sm{G8, bmh};
      gemm_j<16, 4, GI_ZERO>(Gb, sm, NoNote(), r, Wf, smem, [&](int gi) { hook_rows<48, 4>(gi, d_damid, vrow * 512 + 32 * g, sm, r); });
    }
    // ---------------- stage 3: d emb = [W_b; W_heads]^T [d b; dz_heads] (the d bottleneck rows leave here) -----------------
    f32x4 G[16];      // the gradient entering a layer's output BEFORE that layer's ReLU mask bm
    unsigned bm[2];
    {
      const AsyncD db = asyncd(a.saved.relu_bits, ((long long)(L - 1) * n_max + p0) * 32, rows, 32);
      u32x2t b0 = ald8(db, vrow * 32 + 8 * g);
      r.since = 0;
      const RowD d_dbott = rowd(a.gout.d_bott, (long long)p0 * 1024, rows, 1024);
      const SrcAcc<false, 16> sb{Gb};
      gemm_j<16, 9, GI_ZERO>(G, [&](int kk, int i, float& x, float& y) {  // ninth K-step: slot (g, e < 4) = heads row 4 g + e
        if (kk < 8) {
          sb(kk, i, x, y);
        } else {
          x = i == 0 ? qh.x : (i == 1 ? qh.z : 0.0f);
          y = i == 0 ? qh.y : (i == 1 ? qh.w : 0.0f);
        }
      }, NoNote(), r, Wf, smem, [&](int gi) { hook_rows<48, 8>(gi, d_dbott, vrow * 1024 + 32 * g, sb, r); });
      wait_loads(r);
      tie1(b0);
      bm[0] = b0.x; bm[1] = b0.y;
    }
    // ---------------- stage 4: trunk, layers L-1 .. 1: dy[l] = G masked by bm; its rows leave from the GEMM that reads it ----------
    float part[3] = {0.0f, 0.0f, 0.0f}, rawu[3] = {0.0f, 0.0f, 0.0f};
    float fq[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) fq[t] = P.freqs[4 * g + t];
    auto enc_part = [&](int l_rows) {  // (encoded-input part)^T x dy, folded into the variance gradient with the saved features
      // ORDINARY loads (the compiler waits for them itself; twice per tile that is cheap): the destination of an inline-asm load is
      // not safe from being spilled before its data has arrived -- seen in a forward experiment of the round, caught by the row test
      const RowD d_enc = rowd(a.saved.enc, (long long)p0 * 512, rows, 512);
      u32x4t fr[8];
#pragma unroll
      for (int k = 0; k < 8; ++k)
        fr[k] = __builtin_amdgcn_raw_buffer_load_b128(d_enc.r, vrow * 512 + 32 * g + (k >> 1) * 128 + (k & 1) * 16, 0, 0);
      f32x4 eacc[8];
      const SrcMasked<16> sm{G, bm};
      if (l_rows >= 0) {
        const RowD dd = d_dy(l_rows);
        gemm_j<8, 8, GI_ZERO>(eacc, sm, NoNote(), r, Wf, smem, [&](int gi) { hook_rows<24, 8>(gi, dd, vrow * 1024 + 32 * g, sm, r); });
      } else {
        gemm_j<8, 8, GI_ZERO>(eacc, sm, NoNote(), r, Wf, smem, NoHook());
      }
      float ft[24];
#pragma unroll
      for (int u = 0; u < 24; ++u) ft[u] = __uint_as_float(fr[u >> 2][u & 3]);
      fold_enc3<false>(eacc, ft, fq, part, rawu);
    };
#pragma unroll 1
    for (int l = L - 1; l >= 1; --l) {
      if (INPUT && l == P.skip_layer) enc_part(-1);
      const AsyncD db = asyncd(a.saved.relu_bits, ((long long)(l - 1) * n_max + p0) * 32, rows, 32);
      u32x2t b0 = ald8(db, vrow * 32 + 8 * g);
      r.since = 0;
      const RowD dd = d_dy(l);
      f32x4 acc[16];
      const SrcMasked<16> sm{G, bm};
      gemm_j<16, 8, GI_ZERO>(acc, sm, NoNote(), r, Wf, smem, [&](int gi) { hook_rows<48, 8>(gi, dd, vrow * 1024 + 32 * g, sm, r); });
      wait_loads(r);
      tie1(b0);
      bm[0] = b0.x; bm[1] = b0.y;
      copy_acc(G, acc);
    }
    if (INPUT) {
      enc_part(0);  // keeps dy[0]
      // ---------------- stage 5: gradient w.r.t. the Gaussian's variance -> pixel_area / sqradius -----------------
      float dvar[3];
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        float tot = -0.5f * part[c];
        tot += __shfl_xor(tot, 16, 64);
        tot += __shfl_xor(tot, 32, 64);
        dvar[c] = tot;
      }
      const unsigned pt = p0 + m;
      if (g == 0 && pt < n_points && a.gout.d_input) {
        const size_t q = pt;
        float gg = 0.0f;
        if (a.mode == RSN_MODE_FRUSTUM) {
          const long long ray = (long long)(q / (unsigned)a.S);
          const int s = (int)(q - (size_t)ray * a.S);
          float o[3], d[3], dv[3];
#pragma unroll
          for (int c = 0; c < 3; ++c) { o[c] = a.origins[ray * 3 + c]; d[c] = a.directions[ray * 3 + c]; }
          frustum_dvar_dpa(o, d, a.pixel_area[ray], a.bins[ray * (a.S + 1) + s], a.bins[ray * (a.S + 1) + s + 1], dv);
          gg = dvar[0] * dv[0] + dvar[1] * dv[1] + dvar[2] * dv[2];
        } else {  // INF: var_c = (0.6 sq)(1 - d_c^2)   (reflect_sampling_nerf_field.py:196)
#pragma unroll
          for (int c = 0; c < 3; ++c) {
            const float dc = a.directions[q * 3 + c];
            gg += dvar[c] * (0.6f * (1.0f - dc * dc));
          }
        }
        a.gout.d_input[q] = gg;
      }
    } else {  // dy[0] has no GEMM behind it on this path: its rows leave here
      const RowD dd = d_dy(0);
      const SrcMasked<16> sm{G, bm};
#pragma unroll
      for (int kk = 0; kk < 8; ++kk) store_kstep(dd, vrow * 1024 + 32 * g, kk, sm, r);
    }
  }
  ring_finish(r, wid);
}

// ------------------------------------------------------------------------------------------------ launchers
int rsn_launch_field_x6_train(long long n_tiles128, hipStream_t st, const FieldJobs& J) {
  bool normals = false, plain = false;
  for (int k = 0; k < J.n_jobs; ++k) {
    const FieldJob& a = J.j[k];
    RSN_REQUIRE(a.mode == RSN_MODE_FRUSTUM || a.mode == RSN_MODE_INF, RSN_ERR_UNSUPPORTED, "job %d: mode %d", k, a.mode);
    RSN_REQUIRE((long long)a.n_rays * a.S < (1LL << 31), RSN_ERR_UNSUPPORTED, "job %d: 2^31 points or more", k);
    if (a.saved.normals) normals = true; else plain = true;
  }
  RSN_REQUIRE(!(normals && plain), RSN_ERR_UNSUPPORTED,
              "evaluations with and without analytic normals cannot share a launch (the weight ring walks one program)");
  RSN_REQUIRE(J.s.L.q_pf == 3 && J.s.L.q_stream != 0, RSN_ERR_INVALID_ARGUMENT, "the packed weights carry no split-bf16 ring stream");
  const int cus = rsn_device_cus();
  const long long grid = n_tiles128 < (long long)cus ? n_tiles128 : (long long)cus;
  if (normals) hipLaunchKernelGGL(rsn_field_x6_train_kernel<true>, dim3((unsigned)grid), dim3(512), 0, st, J);
  else hipLaunchKernelGGL(rsn_field_x6_train_kernel<false>, dim3((unsigned)grid), dim3(512), 0, st, J);
  RSN_HIP(hipGetLastError());
  return RSN_OK;
}

int rsn_launch_field_x6_bwd(long long n_tiles128, hipStream_t st, const BwdJobs& J) {
  bool input = false;
  for (int k = 0; k < J.n_jobs; ++k) {
    RSN_REQUIRE((long long)J.j[k].n_rays * J.j[k].S < (1LL << 31), RSN_ERR_UNSUPPORTED, "job %d: 2^31 points or more", k);
    input = input || J.j[k].need_input_grad != 0;
  }
  for (int k = 0; k < J.n_jobs; ++k)
    RSN_REQUIRE((J.j[k].need_input_grad != 0) == input, RSN_ERR_UNSUPPORTED,
                "evaluations with and without an input gradient cannot share a launch (the weight ring walks one program)");
  RSN_REQUIRE(J.s.L.q_pf == 3 && J.s.L.q_stream != 0, RSN_ERR_INVALID_ARGUMENT, "the packed weights carry no split-bf16 ring stream");
  const int cus = rsn_device_cus();
  const long long grid = n_tiles128 < (long long)cus ? n_tiles128 : (long long)cus;
  if (input) hipLaunchKernelGGL(rsn_field_x6_bwd_kernel<true>, dim3((unsigned)grid), dim3(512), 0, st, J);
  else hipLaunchKernelGGL(rsn_field_x6_bwd_kernel<false>, dim3((unsigned)grid), dim3(512), 0, st, J);
  RSN_HIP(hipGetLastError());
  return RSN_OK;
}
