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
//   * a wave holds ONE 16-point half (128-point tiles per workgroup): its activations live as three bf16 pieces (96 VGPRs)
//     beside 64 accumulator registers; they are split ONCE per layer in the epilogue that produces them (11 vector instructions
//     per pair of values, 0.6 per MFMA), not once per K-step and wave;
//   * everything kept for the backward pass / the weight gradients is fp32 (this mode is fp32-equivalent: the same parity
//     bounds as the exact path): a lane's share of a row -- 8 contiguous features = 32 bytes per K-step -- leaves as two
//     non-temporal 16-byte stores from the epilogue, between the splits; encode, activations, SH and the chain through the
//     encoding use the exact-fp32 forms of rsn_field_kernel.h (sin_big, expf, log1pf ...), not the fast ones of the bf16 mode.
// Saved-buffer layout: rsn_train_saved_layout (include/rsn.h): enc fp32 [N,128], sh fp32 [N,64] in the ring's slot order,
// ReLU bit words [L+1][N][4 lane groups][2 words], wide buffers fp32 [N,W] in natural feature order.
#include "rsn_ringt.h"

#ifndef X6_LEAD_FWD
#define X6_LEAD_FWD 3
#endif
#ifndef X6_LEAD_BWD
#define X6_LEAD_BWD 3
#endif
#define X6_STASH_BYTES (8 * 1024)   // per wave: 8 float4 per lane -- the encoded inputs [kk (4)][half (2)], later the SH inputs / the
                                    // derivative factors of the encoding

// the three bf16 parts of eight fp32 values (x = h + m + l exactly: 3 x 8 significant bits)
typedef bf16x8 X3[3];   // [0] hi, [1] mid, [2] lo  (an array, not a struct: hipcc keeps arrays of them in registers)
#define XH_ 0
#define XM_ 1
#define XL_ 2
__device__ __forceinline__ void split8(const float (&v)[8], X3& o) {
  uint4v wh, wm, wl;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const float a = v[2 * i], b = v[2 * i + 1];
    const unsigned h2 = pack2<false>(a, b);
    const float ra = a - __uint_as_float(h2 << 16), rb = b - __uint_as_float(h2 & 0xffff0000u);
    const unsigned m2 = pack2<false>(ra, rb);
    const float sa = ra - __uint_as_float(m2 << 16), sb = rb - __uint_as_float(m2 & 0xffff0000u);
    wh[i] = h2;
    wm[i] = m2;
    wl[i] = pack2<false>(sa, sb);
  }
  o[XH_] = __builtin_bit_cast(bf16x8, wh);
  o[XM_] = __builtin_bit_cast(bf16x8, wm);
  o[XL_] = __builtin_bit_cast(bf16x8, wl);
}
__device__ __forceinline__ void split8(const float4 lo, const float4 hi, X3& o) {
  const float v[8] = {lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
  split8(v, o);
}

// acc[b] (+)= W-fragment(fr) * X[kk], fragment fr = (K-step kk, block b) arriving as the pieces lo, mid, hi (i = 3 fr + s).
// Six products, the small ones first: lo x h | mid x m, mid x h | hi x l, hi x m, hi x h.
template <int NBO, int KS, int XN, int INIT, class RING, class HOOK>
__device__ __forceinline__ void gemm_x6(f32x4 (&acc)[NBO], const X3 (&X)[XN], RING& r, bf16x8 (&W)[RING_FIFO], const char* smem,
                                        HOOK&& hook, const float* bias = nullptr, int g = 0) {
  static_assert((NBO * KS * 3) % RSN_RING_GROUP_FRAGS == 0 && KS <= XN, "a GEMM is a whole number of ring groups");
#pragma unroll
  for (int gi = 0; gi < NBO * KS * 3 / RSN_RING_GROUP_FRAGS; ++gi) {
    ringt_sync(r);
    hook(gi);
#pragma unroll
    for (int f = 0; f < RSN_RING_GROUP_FRAGS; ++f) {
      const int i = gi * RSN_RING_GROUP_FRAGS + f;
      const int fr = i / 3, s = i % 3, kk = fr / NBO, b = fr % NBO;
      const bf16x8 wa = W[i % RING_FIFO];
      const int pos = f + RING_FIFO;
      W[i % RING_FIFO] = *reinterpret_cast<const bf16x8*>(
          smem + (pos < RSN_RING_GROUP_FRAGS ? r.rd_cur + pos * 1024 : r.rd_next + (pos - RSN_RING_GROUP_FRAGS) * 1024));
      if (s == 0) {
        f32x4 c = acc[b];
        if (INIT != GI_ACC && kk == 0) {
          c = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
          if (INIT == GI_BIAS) {
            const float4 bv = *reinterpret_cast<const float4*>(bias + b * 16 + 4 * g);
            c = f32x4{bv.x, bv.y, bv.z, bv.w};
          }
        }
        acc[b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wa, X[kk][XH_], c, 0, 0, 0);
      } else if (s == 1) {
        acc[b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wa, X[kk][XM_], acc[b], 0, 0, 0);
        acc[b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wa, X[kk][XH_], acc[b], 0, 0, 0);
      } else {
        acc[b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wa, X[kk][XL_], acc[b], 0, 0, 0);
        acc[b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wa, X[kk][XM_], acc[b], 0, 0, 0);
        acc[b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wa, X[kk][XH_], acc[b], 0, 0, 0);
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
  __builtin_amdgcn_raw_buffer_store_b128(v, d.r, voff, soff, RT_STORE_AUX);
#ifndef RSN_RT_UNCOUNTED
  r.c0 += 1;
  r.since += 1;
#endif
#endif
}
template <int OFF>
__device__ __forceinline__ u32x4t aldq(const AsyncD& d, unsigned voff) {
#ifdef RSN_RT_NO_LOADS
  return u32x4t{0u, 0u, 0u, 0u};
#else
  u32x4t v;
  asm volatile("buffer_load_dwordx4 %0, %1, %2, 0 offen offset:%3 sc0" : "=v"(v) : "v"(voff), "s"(d.rs), "n"(OFF) : "memory");
  return v;
#endif
}
__device__ __forceinline__ void tie1(u32x2t& a) { asm volatile("" : "+v"(a)::"memory"); }
__device__ __forceinline__ void tie8(u32x4t (&q)[8]) {
#pragma unroll
  for (int i = 0; i < 8; ++i) asm volatile("" : "+v"(q[i])::"memory");
}

// Epilogue of a GEMM whose result feeds another GEMM: blocks 2kk, 2kk+1 = the lane's features 32 kk + 8 g .. + 7 of its point
// -> (ReLU) -> the fp32 row piece leaves (32 bytes at 128 kk + 32 g of the row) -> split into the next B operand.
template <bool RELU, int NKS, int NBO, int XN, class RING>
__device__ __forceinline__ void epi_rows(const f32x4 (&acc)[NBO], X3 (&X)[XN], const RowD& d, unsigned voff, RING& r) {
#pragma unroll
  for (int kk = 0; kk < NKS; ++kk) {
    float v[8];
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int q = 0; q < 4; ++q) v[4 * t + q] = RELU ? relu_f(acc[2 * kk + t][q]) : acc[2 * kk + t][q];
    st16f(d, voff, kk * 128, v[0], v[1], v[2], v[3], r);
    st16f(d, voff, kk * 128 + 16, v[4], v[5], v[6], v[7], r);
    split8(v, X[kk]);
  }
}
// the same for a layer gradient: masked by the ReLU bits of the layer it enters (bit layout: relu_bits3 below)
template <bool STORE, int NKS, int NBO, int XN, class RING>
__device__ __forceinline__ void epi_masked(const f32x4 (&acc)[NBO], X3 (&X)[XN], const unsigned (&bits)[2], const RowD& d,
                                           unsigned voff, RING& r) {
#pragma unroll
  for (int kk = 0; kk < NKS; ++kk) {
    float v[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const int j = (kk & 3) * 4 + (e >> 1), pos = ((e & 1) ? 31 : 15) - j;
#ifdef RSN_RT_NO_BITS
      const int mk = -1;
#else
      const int mk = __builtin_amdgcn_sbfe((int)bits[kk >> 2], (unsigned)pos, 1u);  // 0 or -1
#endif
      v[e] = __uint_as_float(__float_as_uint(acc[2 * kk + (e >> 2)][e & 3]) & (unsigned)mk);
    }
    if (STORE) {
      st16f(d, voff, kk * 128, v[0], v[1], v[2], v[3], r);
      st16f(d, voff, kk * 128 + 16, v[4], v[5], v[6], v[7], r);
    }
    split8(v, X[kk]);
  }
}
// ReLU bits of the post-ReLU activations X[kk0 .. kk0+3] (from their hi parts: hi > 0 <=> value > 0): word j = 4 (kk - kk0) + wi
// contributes (low half > 0) at bit 15 - j and (high half > 0) at bit 31 - j -- the layout of rsn_field_bf16_train.hip
template <int XN>
__device__ __forceinline__ unsigned relu_bits3(const X3 (&X)[XN], int kk0, unsigned one2) {
#ifdef RSN_RT_NO_BITS
  return 0xffffffffu;
#endif
  unsigned b = 0u;
#pragma unroll
  for (int j = 0; j < 16; ++j) {
    const uint4v w = __builtin_bit_cast(uint4v, X[kk0 + (j >> 2)][XH_]);
    const unsigned t = pk_min_u16(w[j & 3], one2);
    b = (j == 0) ? t : ((b << 1) | t);
  }
  return b;
}

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
  const unsigned one2 = 0x00010001u;

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

    X3 X[8];
    unsigned bwe[2];  // the embedding's ReLU bits: the seed mask of the normal sweep
    // ---------------- trunk -----------------
    {
      f32x4 acc[16];
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
          const float v8[8] = {feat[8 * kk], feat[8 * kk + 1], feat[8 * kk + 2], feat[8 * kk + 3],
                               feat[8 * kk + 4], feat[8 * kk + 5], feat[8 * kk + 6], feat[8 * kk + 7]};
          split8(v8, X[kk]);
        }
      }
      gemm_x6<16, 4, 8, GI_BIAS>(acc, X, r, Wf, smem, NoHook(), bias, g);
#pragma unroll 1
      for (int l = 1; l < L; ++l) {
        epi_rows<true, 8>(acc, X, d_act(l - 1), vrow * 1024 + 32 * g, r);   // X = act[l-1] (post-ReLU); its rows leave here
        st8(d_bits(l - 1), vrow * 32 + 8 * g, 0, relu_bits3<8>(X, 0, one2), relu_bits3<8>(X, 4, one2), r);
        gemm_x6<16, 8, 8, GI_BIAS>(acc, X, r, Wf, smem, NoHook(), bias + l * 256, g);
        if (l == P.skip_layer) {
          X3 XE[4];
#pragma unroll
          for (int kk = 0; kk < 4; ++kk) split8(ST[(2 * kk) * 64], ST[(2 * kk + 1) * 64], XE[kk]);
          gemm_x6<16, 4, 4, GI_ACC>(acc, XE, r, Wf, smem, NoHook());
        }
      }
      epi_rows<true, 8>(acc, X, d_act(L - 1), vrow * 1024 + 32 * g, r);  // out_activation = ReLU: the embedding = act[L-1]
      bwe[0] = relu_bits3<8>(X, 0, one2);
      bwe[1] = relu_bits3<8>(X, 4, one2);
      st8(d_bits(L - 1), vrow * 32 + 8 * g, 0, bwe[0], bwe[1], r);
    }

    // ---------------- heads: one 16-row block (+ a zero block) -----------------
    float dcol[3];
    {
      f32x4 acch[2];
      gemm_x6<2, 8, 8, GI_BIAS>(acch, X, r, Wf, smem, NoHook(), b_bh + 256, g);
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
    // ---------------- bottleneck -----------------
    {
      f32x4 acc[16];
      gemm_x6<16, 8, 8, GI_BIAS>(acc, X, r, Wf, smem, NoHook(), b_bh, g);
      // bottleneck output (no activation): the x-part of mlp_mid's input; its rows leave here
      epi_rows<false, 8>(acc, X, rowd(a.saved.bott, (long long)p0 * 1024, rows, 1024), vrow * 1024 + 32 * g, r);
    }

    // ---------------- mlp_mid + RGB head -----------------
    {
      f32x4 accm[8];
      X3 XS[2];
      {
        const float4 s0 = ST[0 * 64], s1 = ST[1 * 64], s2 = ST[2 * 64], s3 = ST[3 * 64];
        const RowD d_sh = rowd(a.saved.sh, (long long)p0 * 256, rows, 256);   // fp32 [N,64]
        st16f(d_sh, vrow * 256 + 32 * g, 0, s0.x, s0.y, s0.z, s0.w, r);
        st16f(d_sh, vrow * 256 + 32 * g, 16, s1.x, s1.y, s1.z, s1.w, r);
        st16f(d_sh, vrow * 256 + 32 * g, 128, s2.x, s2.y, s2.z, s2.w, r);
        st16f(d_sh, vrow * 256 + 32 * g, 144, s3.x, s3.y, s3.z, s3.w, r);
        split8(s0, s1, XS[0]);
        split8(s2, s3, XS[1]);
      }
      gemm_x6<8, 2, 2, GI_BIAS>(accm, XS, r, Wf, smem, NoHook(), b_mid, g);
      gemm_x6<8, 8, 8, GI_ACC>(accm, X, r, Wf, smem, NoHook());
      epi_rows<true, 4>(accm, X, rowd(a.saved.hid, (long long)p0 * 512, rows, 512), vrow * 512 + 32 * g, r);  // hidden (128): K-steps 0..3
      st8(d_bits(L), vrow * 32 + 8 * g, 0, relu_bits3<8>(X, 0, one2), 0u, r);
    }
    {
      f32x4 accr[4];  // block 0 carries the RGB rows 4..6; blocks 1..3 are whole-group padding
      {
        const float4 bv = *reinterpret_cast<const float4*>(b_rgb + 4 * g);
        const f32x4 z = {0.0f, 0.0f, 0.0f, 0.0f};
        accr[0] = f32x4{bv.x, bv.y, bv.z, bv.w};
#pragma unroll
        for (int b = 1; b < 4; ++b) accr[b] = z;
      }
      gemm_x6<4, 4, 8, GI_ACC>(accr, X, r, Wf, smem, NoHook());
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
          const float sx = 6.283185307179586f * mc[c];
#pragma unroll
          for (int t = 0; t < 4; ++t) {
            const float f = fq[t];
            const float ang = sx * f;
            const float e = expf(-0.5f * (vc[c] * (f * f)));
            df[c * 4 + t] = e * cos_big(ang);
            df[12 + c * 4 + t] = e * cos_big(ang + 1.5707963267948966f);
          }
        }
#pragma unroll
        for (int i = 0; i < 6; ++i) ST[i * 64] = make_float4(df[4 * i], df[4 * i + 1], df[4 * i + 2], df[4 * i + 3]);
      }
      // seed: the density-head row masked by the embedding's ReLU
      {
        f32x4 sd[16];
#pragma unroll
        for (int kk = 0; kk < 8; ++kk) {
          const float4 lo = *reinterpret_cast<const float4*>(vden + 32 * kk + 8 * g);
          const float4 hi = *reinterpret_cast<const float4*>(vden + 32 * kk + 8 * g + 4);
          sd[2 * kk] = f32x4{lo.x, lo.y, lo.z, lo.w};
          sd[2 * kk + 1] = f32x4{hi.x, hi.y, hi.z, hi.w};
        }
        epi_masked<false, 8>(sd, X, bwe, RowD{}, 0u, r);
      }
      float part[3] = {0.0f, 0.0f, 0.0f}, raw[3] = {0.0f, 0.0f, 0.0f};
      auto enc_part = [&]() {  // eacc = (encoded-input part)^T x gradient, folded at once with the derivative factors
        f32x4 eacc[8];
        gemm_x6<8, 8, 8, GI_ZERO>(eacc, X, r, Wf, smem, NoHook());
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
        gemm_x6<16, 8, 8, GI_ZERO>(acc, X, r, Wf, smem, NoHook());
        wait_loads(r);
        tie1(b0);
        const unsigned bm[2] = {b0.x, b0.y};
        epi_masked<false, 8>(acc, X, bm, RowD{}, 0u, r);
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
  const float z8[8] = {0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f};

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
    X3 X[8];
    X3 XH;   // heads pre-activation gradients as the ninth K-step of the [bottleneck; heads]^T GEMM
    X3 X0;   // RGB-head pre-activation gradient
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
      float dz[3];
#pragma unroll
      for (int c = 0; c < 3; ++c) dz[c] = gcol[c] * tin[c] * (mid[c] * (1.0f - mid[c]));
      if (g == 1 && valid && a.gout.dz_rgb) *reinterpret_cast<float4*>(a.gout.dz_rgb + q * 4) = make_float4(dz[0], dz[1], dz[2], 0.0f);
      {
        const float v[8] = {g == 1 ? dz[0] : 0.0f, g == 1 ? dz[1] : 0.0f, g == 1 ? dz[2] : 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f};
        split8(v, X0);
      }
      float4 qh = make_float4(0.0f, 0.0f, 0.0f, 0.0f);  // heads rows 4 g + r
      if (a.mode != RSN_MODE_INF) {
        if (g == 0) {
          const long long ray = (long long)(q / (unsigned)a.S);
          const float rawd = a.fwd.raw_density[q];
          const float gs = a.gin.sigma ? a.gin.sigma[q] * live : 0.0f;
          qh.x = gs * sigmoid_f(rawd + P.density_bias);  // softplus'
          float dir[3], G[3] = {0.0f, 0.0f, 0.0f};
#pragma unroll
          for (int c = 0; c < 3; ++c) dir[c] = a.directions[ray * 3 + c];
          if (a.gin.pred_normals) {
#pragma unroll
            for (int c = 0; c < 3; ++c) G[c] = a.gin.pred_normals[q * 3 + c] * live;
          }
          float gd = a.gin.n_dot_d ? a.gin.n_dot_d[q] * live : 0.0f;
          if (a.gin.ray_pn_loss || a.gin.ray_ori_loss) {  // fused normal losses (model.py:403-407)
            const float w = a.gin.weights[q] * live;
            if (a.gin.ray_pn_loss) {
              const float gw = a.gin.ray_pn_loss[ray] * w * -2.0f;
#pragma unroll
              for (int c = 0; c < 3; ++c) G[c] += gw * (a.saved.normals[q * 3 + c] - a.fwd.pred_normals[q * 3 + c]);
            }
            if (a.gin.ray_ori_loss) gd += a.gin.ray_ori_loss[ray] * w * (2.0f * fmaxf(a.fwd.n_dot_d[q], 0.0f));
          }
#pragma unroll
          for (int c = 0; c < 3; ++c) G[c] += gd * dir[c];
          const float nraw[3] = {hd.x, hd.y, hd.z};
          const float len = fmaxf(sqrtf(nraw[0] * nraw[0] + nraw[1] * nraw[1] + nraw[2] * nraw[2]), 1e-12f);
          const float v[3] = {-(nraw[0] / len), -(nraw[1] / len), -(nraw[2] / len)};
          float gv[3], gu[3], gn[3];
          normalize_bwd(v, G, gv);
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
      {
        const float v[8] = {qh.x, qh.y, qh.z, qh.w, 0.0f, 0.0f, 0.0f, 0.0f};
        split8(v, XH);
      }
    }

    // ---------------- stage 1: d hidden = W_rgb^T dz (pf groups), masked by the mid hidden layer's ReLU; the d a_mid rows leave ---
    {
      const AsyncD db = asyncd(a.saved.relu_bits, ((long long)L * n_max + p0) * 32, rows, 32);
      u32x2t b0 = ald8(db, vrow * 32 + 8 * g);
      r.since = 0;
      X3 XR[2];
#pragma unroll
      for (int q = 0; q < 3; ++q) XR[0][q] = X0[q];
      split8(z8, XR[1]);
      f32x4 acc[8];
      gemm_x6<8, 2, 2, GI_ZERO>(acc, XR, r, Wf, smem, NoHook());
      wait_loads(r);
      tie1(b0);
      const unsigned bm[2] = {b0.x, 0u};
      epi_masked<true, 4>(acc, X, bm, rowd(a.gout.da_mid, (long long)p0 * 512, rows, 512), vrow * 512 + 32 * g, r);  // X[0..3] = d a_mid
    }
    // ---------------- stage 2: d bottleneck = W_mid[:, 34:]^T d a_mid; its rows leave -----------------
    {
      f32x4 acc[16];
      gemm_x6<16, 4, 8, GI_ZERO>(acc, X, r, Wf, smem, NoHook());
      epi_rows<false, 8>(acc, X, rowd(a.gout.d_bott, (long long)p0 * 1024, rows, 1024), vrow * 1024 + 32 * g, r);  // X = d bottleneck
    }
    // ---------------- stage 3: d emb = [W_b; W_heads]^T [d b; dz_heads], masked by the embedding's ReLU: dy[L-1] -----------------
    {
      const AsyncD db = asyncd(a.saved.relu_bits, ((long long)(L - 1) * n_max + p0) * 32, rows, 32);
      u32x2t b0 = ald8(db, vrow * 32 + 8 * g);
      r.since = 0;
      X3 X9[9];
#pragma unroll
      for (int kk = 0; kk < 8; ++kk)
#pragma unroll
        for (int q = 0; q < 3; ++q) X9[kk][q] = X[kk][q];
#pragma unroll
      for (int q = 0; q < 3; ++q) X9[8][q] = XH[q];
      f32x4 acc[16];
      gemm_x6<16, 9, 9, GI_ZERO>(acc, X9, r, Wf, smem, NoHook());
      wait_loads(r);
      tie1(b0);
      const unsigned bm[2] = {b0.x, b0.y};
      epi_masked<true, 8>(acc, X, bm, d_dy(L - 1), vrow * 1024 + 32 * g, r);  // X = dy[L-1]
    }
    // ---------------- stage 4: trunk, layers L-1 .. 1 -----------------
    float part[3] = {0.0f, 0.0f, 0.0f}, rawu[3] = {0.0f, 0.0f, 0.0f};
    float fq[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) fq[t] = P.freqs[4 * g + t];
    auto enc_part = [&]() {  // (encoded-input part)^T x dy, folded into the variance gradient with the saved features
      const AsyncD d_enc = asyncd(a.saved.enc, (long long)p0 * 512, rows, 512);
      u32x4t fr[8];
      fr[0] = aldq<0>(d_enc, vrow * 512 + 32 * g);   fr[1] = aldq<16>(d_enc, vrow * 512 + 32 * g);
      fr[2] = aldq<128>(d_enc, vrow * 512 + 32 * g); fr[3] = aldq<144>(d_enc, vrow * 512 + 32 * g);
      fr[4] = aldq<256>(d_enc, vrow * 512 + 32 * g); fr[5] = aldq<272>(d_enc, vrow * 512 + 32 * g);
      fr[6] = aldq<384>(d_enc, vrow * 512 + 32 * g); fr[7] = aldq<400>(d_enc, vrow * 512 + 32 * g);
      r.since = 0;
      f32x4 eacc[8];
      gemm_x6<8, 8, 8, GI_ZERO>(eacc, X, r, Wf, smem, NoHook());
      wait_loads(r);
      tie8(fr);
      float ft[24];
#pragma unroll
      for (int u = 0; u < 24; ++u) ft[u] = __uint_as_float(fr[u >> 2][u & 3]);
      fold_enc3<false>(eacc, ft, fq, part, rawu);
    };
#pragma unroll 1
    for (int l = L - 1; l >= 1; --l) {
      if (INPUT && l == P.skip_layer) enc_part();
      const AsyncD db = asyncd(a.saved.relu_bits, ((long long)(l - 1) * n_max + p0) * 32, rows, 32);
      u32x2t b0 = ald8(db, vrow * 32 + 8 * g);
      r.since = 0;
      f32x4 acc[16];
      gemm_x6<16, 8, 8, GI_ZERO>(acc, X, r, Wf, smem, NoHook());
      wait_loads(r);
      tie1(b0);
      const unsigned bm[2] = {b0.x, b0.y};
      epi_masked<true, 8>(acc, X, bm, d_dy(l - 1), vrow * 1024 + 32 * g, r);  // X = dy[l-1]; its rows leave
    }
    if (INPUT) {
      enc_part();
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
