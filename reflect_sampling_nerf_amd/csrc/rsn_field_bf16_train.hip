// rsn_field_bf16_train.hip -- the TRAINING kernels of the plain-bf16 mode (RSN_MMA_BF16, width 256) on the LDS weight ring.
//
// The reference trains in reduced precision (reflect_sampling_nerf_config.py:33: mixed_precision=True -> autocast); this
// mode is its analogue here (bf16 MFMA operands, fp32 accumulation, fp32 encode / activations / compositing).  Round 3
// ran it on rsn_field_kernel<8, true, 3> / rsn_field_bwd_kernel<8, 3>: the fp32 kernels' structure with bf16 operands --
// every wave streams all weights from L1 / L2 per 32 points while the MFMAs take 1/16 of the fp32 time (MFMA busy 16 %,
// 43 % of wave cycles parked).  Here both sweeps get the design of rsn_field_bf16_ring16_kernel (rsn_field_bf16.hip):
//   * one 8-wave workgroup per CU walks the network in lockstep, each wave a 32-point tile (two 16-point halves) on
//     v_mfma_f32_16x16x32_bf16; the weight fragments -- forward stream + TRANSPOSED stream (rsn_pack.hip: q_stream and the
//     t_g_* groups behind it) -- are pulled ONCE per workgroup through a 4-slot LDS ring by LDS-DMA;
//   * activations / layer gradients never touch LDS: a lane's share of the next GEMM's B operand is the bf16 pack of its own
//     accumulators (lane-local hand-off; the packed stream permutes output rows so that the share is 8 CONTIGUOUS features
//     per K-step), and the same 16 bytes are what the lane stores of the saved row (row-major bf16 [N, W], the X / dY operand
//     of the weight-gradient kernel): one buffer store per K-step and point, issued at the ring's group boundaries INSIDE the
//     GEMM that reads the row, so the HBM write stream (4.6 KB per sample forward, 4.4 KB backward: this mode is as much
//     HBM-bound as MFMA-bound, 256 FLOP per byte of saved row) drains under the MFMAs;
//   * the ring's consumer wait is COUNTED: vmcnt retires in issue order, so the wait that admits the next weight group
//     names exactly the operations issued behind that group's LDS-DMA (the row stores of the last two groups + one DMA);
//     older stores have had two groups (~1 us) to retire and nothing younger is waited for;
//   * ReLU masks are bit-packed from the packed bf16 activations (v_pk_min_u16 + v_lshl_or_b32 per two values) and applied in
//     the sweeps to the packed gradients (shift, and, v_pk_mul_lo_u16 per two values);
//   * the training forward of the primary levels carries the analytic-normal sweep (reference field.py:125-127,146-147 ->
//     nerfstudio Field.get_normals) through the trunk part of the transposed stream; the chain through the encoding is closed
//     from the saved features themselves (d/dx [e sin a] = 2 pi f [e cos a]: the cosine feature; no trigonometry).
// Saved-buffer layout of this path: rsn_train_saved_layout (include/rsn.h).
#include "rsn_ringt.h"


// acc[b][p] (+)= W-fragment(i) * X[kk][p]; `hook(group)` runs right behind every group boundary (the kernels put their row stores
// there).  INIT: how the accumulators start -- GI_ACC: they are live (a second GEMM onto the same accumulators); GI_ZERO / GI_BIAS:
// the FIRST K-step's MFMAs take the constant 0 / the bias row (LDS table, packed row order; the same vector for both points) as their
// C operand, so no accumulator is written before its first MFMA: 128 v_mov per layer and wave gone (a fifth of the hand-off's VALU).
template <int NBO, int KS, int XN, int INIT, class RING, class HOOK>
__device__ __forceinline__ void gemm_t(f32x4 (&acc)[NBO][2], const bf16x8 (&X)[XN][2], RING& r, bf16x8 (&W)[RING_FIFO],
                                       const char* smem, HOOK&& hook, const float* bias = nullptr, int g = 0) {
  static_assert((NBO * KS) % RSN_RING_GROUP_FRAGS == 0 && KS <= XN, "a GEMM is a whole number of ring groups");
  // (two nested loops, not one with `if (i % 16 == 0)`: hipcc prices the unrolled size BEFORE it folds the wait's if-chain, and
  // refuses to unroll a 128-iteration body that carries the chain in every iteration -- the accumulators then live in scratch)
#pragma unroll
  for (int gi = 0; gi < NBO * KS / RSN_RING_GROUP_FRAGS; ++gi) {
    ringt_sync(r);
    hook(gi);
#pragma unroll
    for (int f = 0; f < RSN_RING_GROUP_FRAGS; ++f) {
      const int i = gi * RSN_RING_GROUP_FRAGS + f;
      const int kk = i / NBO, b = i % NBO;
      const bf16x8 wa = W[i % RING_FIFO];
      const int pos = f + RING_FIFO;
      W[i % RING_FIFO] = *reinterpret_cast<const bf16x8*>(
          smem + (pos < RSN_RING_GROUP_FRAGS ? r.rd_cur + pos * 1024 : r.rd_next + (pos - RSN_RING_GROUP_FRAGS) * 1024));
      if (INIT != GI_ACC && kk == 0) {
        f32x4 c = {0.0f, 0.0f, 0.0f, 0.0f};
        if (INIT == GI_BIAS) {
          const float4 bv = *reinterpret_cast<const float4*>(bias + b * 16 + 4 * g);
          c = f32x4{bv.x, bv.y, bv.z, bv.w};
        }
        acc[b][0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wa, X[0][0], c, 0, 0, 0);
        acc[b][1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wa, X[0][1], c, 0, 0, 0);
      } else {
        acc[b][0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wa, X[kk][0], acc[b][0], 0, 0, 0);
        acc[b][1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wa, X[kk][1], acc[b][1], 0, 0, 0);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
  }
}

// bits of the (non-negative, ReLU'd) packed activations X[kk0 .. kk0+3][p]: word j = 4 (kk - kk0) + wi contributes
// (low half > 0) at bit 15 - j and (high half > 0) at bit 31 - j
template <int XN>
__device__ __forceinline__ unsigned relu_bits_of(const bf16x8 (&X)[XN][2], int p, int kk0, unsigned one2) {
#ifdef RSN_RT_NO_BITS
  return 0xffffffffu;
#endif
  unsigned b = 0u;
#pragma unroll
  for (int j = 0; j < 16; ++j) {
    const uint4v w = __builtin_bit_cast(uint4v, X[kk0 + (j >> 2)][p]);
    const unsigned t = pk_min_u16(w[j & 3], one2);
    b = (j == 0) ? t : ((b << 1) | t);
  }
  return b;
}
// packed gradient word (K-step kk, word wi) masked by those bits
__device__ __forceinline__ unsigned mask_word(unsigned gword, unsigned bits, int j, unsigned one2) {
#ifdef RSN_RT_NO_BITS
  return gword;
#else
  return pk_mul_lo_u16(gword, (bits >> (15 - j)) & one2);
#endif
}

// accumulators -> packed bf16 B operand of the next GEMM (no activation), masked by the layer's ReLU bits (2 words per point)
template <int NBO, int NKS, int XN>
__device__ __forceinline__ void acc_to_x16_masked(const f32x4 (&acc)[NBO][2], bf16x8 (&X)[XN][2], const unsigned (&bits)[2][2],
                                                  unsigned one2) {
#pragma unroll
  for (int kk = 0; kk < NKS; ++kk)
#pragma unroll
    for (int p = 0; p < 2; ++p) {
      uint4v w;
      w[0] = pack2<false>(acc[2 * kk][p][0], acc[2 * kk][p][1]);
      w[1] = pack2<false>(acc[2 * kk][p][2], acc[2 * kk][p][3]);
      w[2] = pack2<false>(acc[2 * kk + 1][p][0], acc[2 * kk + 1][p][1]);
      w[3] = pack2<false>(acc[2 * kk + 1][p][2], acc[2 * kk + 1][p][3]);
#pragma unroll
      for (int wi = 0; wi < 4; ++wi) w[wi] = mask_word(w[wi], bits[p][kk >> 2], (kk & 3) * 4 + wi, one2);
      X[kk][p] = __builtin_bit_cast(bf16x8, w);
    }
}

// Gradient w.r.t. this lane's encoded inputs (eacc: packed rows 16 b + 4 g + r = slot (kk = b / 2, e = 4 (b % 2) + r); slot
// u = 8 kk + e: u < 12 the sine feature of (coordinate u / 4, frequency 4 g + u % 4), 12 <= u < 24 its cosine feature, 24..26 the
// raw coordinates on g == 0) folded with the lane's own features `ft` (same slots, bf16):
//   NORMALS: d raw_density / d x_c += 2 pi f (g_sin * cos_feature - g_cos * sin_feature)   [+ the raw-coordinate slot]
//   else   : d loss / d var_c      += -f^2 / 2 (g_sin * sin_feature + g_cos * cos_feature)
// (the chain through e sin(2 pi f x [+ pi / 2]) exp(-var f^2 / 2), covariance constant for the normals exactly like the
// reference, which sets requires_grad on the mean after the contraction: field.py:125-127)
template <bool NORMALS>
__device__ __forceinline__ void fold_enc(const f32x4 (&eacc)[8][2], const bf16x8 (&ft)[4][2], const float (&fq)[4], int g,
                                         float (&part)[2][3], float (&raw)[2][3]) {
#pragma unroll
  for (int p = 0; p < 2; ++p)
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      float s = 0.0f;
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        const int us = 4 * c + t, uc = 12 + 4 * c + t;
        const float gs = eacc[2 * (us >> 3) + ((us & 7) >> 2)][p][us & 3];
        const float gc = eacc[2 * (uc >> 3) + ((uc & 7) >> 2)][p][uc & 3];
        const float fs = (float)ft[us >> 3][p][us & 7], fc = (float)ft[uc >> 3][p][uc & 7];
        s += NORMALS ? fq[t] * (gs * fc - gc * fs) : (fq[t] * fq[t]) * (gs * fs + gc * fc);
      }
      part[p][c] += s;
      if (NORMALS) raw[p][c] += eacc[6][p][c];  // slot u = 24 + c: kk = 3, e = c (meaningful on g == 0 only)
    }
  (void)g;
}

// ================================================================================================ training forward
template <bool NORMALS>
__global__ __launch_bounds__(512, 2) void rsn_field_bf16_train_kernel(const FieldJobs J) {
  constexpr int W = 256;
  constexpr int RB = RT_RING_BYTES(RT_LEAD_FWD, RT_STAG_FWD);
  __shared__ __attribute__((aligned(1024))) char smem[RB + 8 * R16_STASH_BYTES + RT_TABLE_FLOATS * 4];
  const int lane = threadIdx.x & 63;
  const int wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  char* stash = smem + RB + wid * R16_STASH_BYTES;
  bf16x8* ST = reinterpret_cast<bf16x8*>(stash) + lane;   // fragment (kk, p) of this lane: ST[(kk * 2 + p) * 64]
  float* bias = reinterpret_cast<float*>(smem + RB + 8 * R16_STASH_BYTES);
  const float* b_bh = bias + RING_MAX_LAYERS * 256;
  const float* b_mid = b_bh + 288;
  const float* b_rgb = b_mid + 128;
  const float* vden = b_rgb + 32;  // density-head row, natural feature order

  const FieldShared& P = J.s;
  const TileJobs T = tile_space(J);
  if ((long long)blockIdx.x >= T.n_tiles) return;  // workgroup-uniform
  const float* __restrict__ pk = P.packed;
  const int L = P.num_layers;

  for (int i = threadIdx.x; i < L * 256; i += 512) bias[i] = pk[P.L.b[i >> 8] + r16_feature(i & 255)];
  for (int i = threadIdx.x; i < 288; i += 512) bias[RING_MAX_LAYERS * 256 + i] = pk[P.L.b_bh + (i < 256 ? r16_feature(i) : i)];
  if (threadIdx.x < 128) bias[RING_MAX_LAYERS * 256 + 288 + threadIdx.x] = pk[P.L.b_mid + r16_feature(threadIdx.x)];
  if (threadIdx.x < 32) bias[RING_MAX_LAYERS * 256 + 288 + 128 + threadIdx.x] = pk[P.L.b_rgb + threadIdx.x];
  if (threadIdx.x < 256) bias[RING_BIAS_FLOATS + threadIdx.x] = pk[P.L.v_density + threadIdx.x];

  RingT<RT_LEAD_FWD, RT_STAG_FWD> r;
  bf16x8 Wf[RING_FIFO];
  // the walk: forward stream [0, q_groups); with the normal sweep then [t_g_trunk, t_g_end) of the transposed stream; again
  ring_start(r, pk, P.L, smem, wid, lane, 0, P.L.q_groups, NORMALS ? P.L.t_g_trunk : 0, NORMALS ? P.L.t_g_end : -1, 0, Wf);
  const unsigned one2 = 0x00010001u;

  for (long long gtile = blockIdx.x; gtile < T.n_tiles; gtile += gridDim.x) {
    const int jk = (gtile >= T.tb1 ? 1 : 0) + (gtile >= T.tb2 ? 1 : 0);  // workgroup-uniform
    const FieldJob& a = J.j[jk];
    const unsigned n_points = (unsigned)(jk == 0 ? T.np0 : (jk == 1 ? T.np1 : T.np2));
    const unsigned tile = (unsigned)(gtile - (jk == 0 ? 0 : (jk == 1 ? T.tb1 : T.tb2)));
    const unsigned p0 = tile * 256 + wid * 32;   // every wave walks every tile (barriers, DMA shares); rows = 0 past the end
    const int rows = p0 >= n_points ? 0 : (int)(n_points - p0 < 32u ? n_points - p0 : 32u);
    const long long n_max = a.act_stride / W;   // points the saved buffers are sized for
    int ln = lane;
    asm volatile("" : "+v"(ln));  // opaque per-tile lane id (see rsn_field_bf16.hip)
    const int m = ln & 15, g = ln >> 4;
    bool valid[2];
    size_t pc[2];
    float vd[2][3];
    bool has_dir = true;
    unsigned vrow[2];  // (16 p + m): the lane's row inside the wave's 32-row tile
#pragma unroll
    for (int p = 0; p < 2; ++p) vrow[p] = (unsigned)(16 * p + m);

    // ---------------- encode both points of this lane (fp32) into the stash (as rsn_field_bf16_ring16_kernel) -----------------
    float mcA[2][3], vcA[2][3];
    {
      const int po = g & 1;
      const unsigned pt = p0 + 16 * po + m;
      const size_t pcc = pt < n_points ? pt : (n_points ? n_points - 1 : 0);
      float mc[3] = {0.0f, 0.0f, 0.0f}, vc[3] = {0.0f, 0.0f, 0.0f}, dd[3] = {0.0f, 0.0f, 0.0f};
      if (a.mode == RSN_MODE_FRUSTUM) {
        const unsigned rayu = (unsigned)pcc / (unsigned)a.S;
        const int s = (int)((unsigned)pcc - rayu * (unsigned)a.S);
        const size_t ray = rayu;
        float o[3];
#pragma unroll
        for (int c = 0; c < 3; ++c) {
          o[c] = a.origins[ray * 3 + c];
          dd[c] = a.directions[ray * 3 + c];
        }
        frustum_to_contracted(o, dd, a.pixel_area[ray], a.bins[ray * (a.S + 1) + s], a.bins[ray * (a.S + 1) + s + 1], mc, vc);
      } else {  // RSN_MODE_INF
        const float r2 = a.sqradius[pcc];
#pragma unroll
        for (int c = 0; c < 3; ++c) {
          dd[c] = a.directions[pcc * 3 + c];
          mc[c] = 2.0f * dd[c];
          vc[c] = (0.6f * r2) * (1.0f - dd[c] * dd[c]);
        }
        has_dir = false;  // SH inputs are zeroed (reflect_sampling_nerf_field.py:199)
      }
#pragma unroll
      for (int p = 0; p < 2; ++p) {
        const unsigned ptp = p0 + 16 * p + m;
        valid[p] = ptp < n_points;
        pc[p] = valid[p] ? ptp : (n_points ? n_points - 1 : 0);
#pragma unroll
        for (int c = 0; c < 3; ++c) {
          mcA[p][c] = __shfl(mc[c], 16 * p + m, 64);
          vcA[p][c] = __shfl(vc[c], 16 * p + m, 64);
          vd[p][c] = __shfl(dd[c], 16 * p + m, 64);
        }
      }
    }
    float fq[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) fq[t] = P.freqs[4 * g + t];
#pragma unroll
    for (int p = 0; p < 2; ++p) {
      float feat[24];
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        const float x = mcA[p][c], v = vcA[p][c];
        const float sx = 6.283185307179586f * x;
#pragma unroll
        for (int t = 0; t < 4; ++t) {
          const float f = fq[t];
          const float ang = sx * f;
          const float e = __builtin_amdgcn_exp2f((-0.5f * (v * (f * f))) * 1.4426950408889634f);
          feat[c * 4 + t] = e * sincos_bf16(ang, 0);
          feat[12 + c * 4 + t] = e * sincos_bf16(ang + 1.5707963267948966f, 0);
        }
      }
#pragma unroll
      for (int kk = 0; kk < 3; ++kk) {
        const float v8[8] = {feat[8 * kk], feat[8 * kk + 1], feat[8 * kk + 2], feat[8 * kk + 3],
                             feat[8 * kk + 4], feat[8 * kk + 5], feat[8 * kk + 6], feat[8 * kk + 7]};
        ST[(kk * 2 + p) * 64] = pack8(v8);
      }
      const float rw[8] = {g == 0 ? mcA[p][0] : 0.0f, g == 0 ? mcA[p][1] : 0.0f, g == 0 ? mcA[p][2] : 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f};
      ST[(3 * 2 + p) * 64] = pack8(rw);
    }

    // saved-row descriptors of this tile (scalar)
    // (built where they are used: four scalar registers each, and the kernel is short of those too)
    auto mk_enc = [&]() { return rowd(a.saved.enc, (long long)p0 * 256, rows, 256); };      // bf16 [N,128]
    auto mk_sh = [&]() { return rowd(a.saved.sh, (long long)p0 * 128, rows, 128); };        // bf16 [N,64]
    auto mk_bott = [&]() { return rowd(a.saved.bott, (long long)p0 * 512, rows, 512); };    // bf16 [N,256]
    auto mk_hid = [&]() { return rowd(a.saved.hid, (long long)p0 * 256, rows, 256); };      // bf16 [N,128]
    auto d_act = [&](int l) { return rowd(a.saved.act, ((long long)l * n_max + p0) * 512, rows, 512); };
    auto d_bits = [&](int l) { return rowd(a.saved.relu_bits, ((long long)l * n_max + p0) * 32, rows, 32); };

    bf16x8 X[8][2];
    unsigned bw[2][2];  // ReLU bits of the layer just finished: [point][word]
    // ---------------- trunk -----------------
    {
      f32x4 acc[16][2];
#pragma unroll
      for (int kk = 0; kk < 4; ++kk) { X[kk][0] = ST[(kk * 2) * 64]; X[kk][1] = ST[(kk * 2 + 1) * 64]; }
      const RowD d_enc = mk_enc();
      gemm_t<16, 4, 8, GI_BIAS>(acc, X, r, Wf, smem, [&](int gi) {  // the encoded inputs leave while layer 0 reads them
        st16(d_enc, vrow[0] * 256 + 16 * g, gi * 64, X[gi][0], r);
        st16(d_enc, vrow[1] * 256 + 16 * g, gi * 64, X[gi][1], r);
      }, bias, g);
#pragma unroll 1
      for (int l = 1; l < L; ++l) {
        acc_to_x16<16, 8, true, 8>(acc, X);   // X = act[l-1] (post-ReLU)
#pragma unroll
        for (int p = 0; p < 2; ++p) { bw[p][0] = relu_bits_of<8>(X, p, 0, one2); bw[p][1] = relu_bits_of<8>(X, p, 4, one2); }
        const RowD da = d_act(l - 1), db = d_bits(l - 1);
        gemm_t<16, 8, 8, GI_BIAS>(acc, X, r, Wf, smem, [&](int gi) {  // act[l-1] leaves from the GEMM that reads it, one K-step per group
          st16(da, vrow[0] * 512 + 16 * g, gi * 64, X[gi][0], r);
          st16(da, vrow[1] * 512 + 16 * g, gi * 64, X[gi][1], r);
          if (gi == 0) {
            st8(db, vrow[0] * 32 + 8 * g, 0, bw[0][0], bw[0][1], r);
            st8(db, vrow[1] * 32 + 8 * g, 0, bw[1][0], bw[1][1], r);
          }
        }, bias + l * 256, g);
        if (l == P.skip_layer) {
          bf16x8 XE[4][2];
#pragma unroll
          for (int kk = 0; kk < 4; ++kk) { XE[kk][0] = ST[(kk * 2) * 64]; XE[kk][1] = ST[(kk * 2 + 1) * 64]; }
          gemm_t<16, 4, 4, GI_ACC>(acc, XE, r, Wf, smem, NoHook());
        }
      }
      acc_to_x16<16, 8, true, 8>(acc, X);  // out_activation = ReLU: the embedding = act[L-1]
#pragma unroll
      for (int p = 0; p < 2; ++p) { bw[p][0] = relu_bits_of<8>(X, p, 0, one2); bw[p][1] = relu_bits_of<8>(X, p, 4, one2); }
    }

    // ---------------- heads: one 16-row block (+ a zero block) -----------------
    float dcol[2][3];
    {
      f32x4 acch[2][2];
      gemm_t<2, 8, 8, GI_BIAS>(acch, X, r, Wf, smem, NoHook(), b_bh + 256, g);
#pragma unroll
      for (int p = 0; p < 2; ++p) {
        const float r0 = acch[0][p][0], r1 = acch[0][p][1], r2 = acch[0][p][2], r3 = acch[0][p][3];
        // g == 0: r0 raw density, r1..r3 normals;  g == 1: r0..r2 diff;  g == 2: r0 roughness;  g == 3: r0..r2 tint
        const float rough_raw = __shfl(r0, 32 + m, 64);
        const float rho = fast_softplus(rough_raw);
        if (g == p) {  // SH-34 of the view direction: one lane of the point's four writes the slots of all four groups
          float sh[36];
          if (has_dir) {
            sh34_attenuated(vd[p][0], vd[p][1], vd[p][2], rho, sh);
          } else {
#pragma unroll
            for (int i = 0; i < 34; ++i) sh[i] = 0.0f;
          }
          sh[34] = 0.0f; sh[35] = 0.0f;
#pragma unroll
          for (int gq = 0; gq < 4; ++gq) {
            float v0[8], v1[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) { v0[e] = sh[9 * gq + e]; v1[e] = 0.0f; }
            v1[0] = sh[9 * gq + 8];
            ST[(0 * 2 + p) * 64 + (gq - g) * 16] = pack8(v0);
            ST[(1 * 2 + p) * 64 + (gq - g) * 16] = pack8(v1);
          }
        }
        dcol[p][0] = fast_sigmoid(r0); dcol[p][1] = fast_sigmoid(r1); dcol[p][2] = fast_sigmoid(r2);
        if (valid[p]) {
          const size_t q = pc[p];
          if (g == 0) {
            if (a.saved.heads) { a.saved.heads[q * 8 + 0] = r1; a.saved.heads[q * 8 + 1] = r2; a.saved.heads[q * 8 + 2] = r3; }
            if (a.mode != RSN_MODE_INF) {
              float nrm = fmaxf(sqrtf(r1 * r1 + r2 * r2 + r3 * r3), 1e-12f);
              float nx = -(r1 / nrm), ny = -(r2 / nrm), nz = -(r3 / nrm);
              nrm = fmaxf(sqrtf(nx * nx + ny * ny + nz * nz), 1e-12f);
              nx /= nrm; ny /= nrm; nz /= nrm;
              if (a.out.sigma) a.out.sigma[q] = fast_softplus(r0 + P.density_bias);
              if (a.out.raw_density) a.out.raw_density[q] = r0;
              if (a.out.pred_normals) {
                a.out.pred_normals[q * 3 + 0] = nx;
                a.out.pred_normals[q * 3 + 1] = ny;
                a.out.pred_normals[q * 3 + 2] = nz;
              }
              if (a.out.n_dot_d) a.out.n_dot_d[q] = vd[p][0] * nx + vd[p][1] * ny + vd[p][2] * nz;
            }
          } else if (g == 1) {
            if (a.mode != RSN_MODE_INF && a.out.diff) {
              a.out.diff[q * 3 + 0] = dcol[p][0]; a.out.diff[q * 3 + 1] = dcol[p][1]; a.out.diff[q * 3 + 2] = dcol[p][2];
            }
          } else if (g == 2) {
            if (a.saved.heads) a.saved.heads[q * 8 + 3] = r0;
            if (a.mode != RSN_MODE_INF) {
              if (a.out.roughness) a.out.roughness[q] = fast_sigmoid(r0);
              if (a.out.raw_roughness) a.out.raw_roughness[q] = r0;
            }
          } else {
            if (a.mode != RSN_MODE_INF && a.out.tint) {
              a.out.tint[q * 3 + 0] = dcol[p][0]; a.out.tint[q * 3 + 1] = dcol[p][1]; a.out.tint[q * 3 + 2] = dcol[p][2];
            }
          }
        }
      }
    }
    // ---------------- bottleneck (the embedding's rows and the last layer's bits leave from this GEMM) -----------------
    {
      f32x4 acc[16][2];
      const RowD da = d_act(L - 1), db = d_bits(L - 1);
      gemm_t<16, 8, 8, GI_BIAS>(acc, X, r, Wf, smem, [&](int gi) {
        st16(da, vrow[0] * 512 + 16 * g, gi * 64, X[gi][0], r);
        st16(da, vrow[1] * 512 + 16 * g, gi * 64, X[gi][1], r);
        if (gi == 0) {
          st8(db, vrow[0] * 32 + 8 * g, 0, bw[0][0], bw[0][1], r);
          st8(db, vrow[1] * 32 + 8 * g, 0, bw[1][0], bw[1][1], r);
        }
      }, b_bh, g);
      acc_to_x16<16, 8, false, 8>(acc, X);  // bottleneck output (no activation): the x-part of mlp_mid's input
    }

    // ---------------- mlp_mid + RGB head -----------------
    unsigned bwe[2][2];  // the embedding's bits: the seed mask of the normal sweep (bw is re-used for the mid hidden layer)
#pragma unroll
    for (int p = 0; p < 2; ++p) { bwe[p][0] = bw[p][0]; bwe[p][1] = bw[p][1]; }
    {
      f32x4 accm[8][2];
      bf16x8 XS[2][2];
#pragma unroll
      for (int kk = 0; kk < 2; ++kk) { XS[kk][0] = ST[(kk * 2) * 64]; XS[kk][1] = ST[(kk * 2 + 1) * 64]; }
      const RowD d_sh = mk_sh(), d_bott = mk_bott();
      gemm_t<8, 2, 2, GI_BIAS>(accm, XS, r, Wf, smem, [&](int) {  // the SH inputs' rows
        st16(d_sh, vrow[0] * 128 + 16 * g, 0, XS[0][0], r);
        st16(d_sh, vrow[1] * 128 + 16 * g, 0, XS[0][1], r);
        st16(d_sh, vrow[0] * 128 + 16 * g, 64, XS[1][0], r);
        st16(d_sh, vrow[1] * 128 + 16 * g, 64, XS[1][1], r);
      }, b_mid, g);
      gemm_t<8, 8, 8, GI_ACC>(accm, X, r, Wf, smem, [&](int gi) {  // the bottleneck rows: two K-steps per group
        st16(d_bott, vrow[0] * 512 + 16 * g, (2 * gi) * 64, X[2 * gi][0], r);
        st16(d_bott, vrow[1] * 512 + 16 * g, (2 * gi) * 64, X[2 * gi][1], r);
        st16(d_bott, vrow[0] * 512 + 16 * g, (2 * gi + 1) * 64, X[2 * gi + 1][0], r);
        st16(d_bott, vrow[1] * 512 + 16 * g, (2 * gi + 1) * 64, X[2 * gi + 1][1], r);
      });
      acc_to_x16<8, 4, true, 8>(accm, X);  // hidden (128): K-steps 0..3
#pragma unroll
      for (int p = 0; p < 2; ++p) { bw[p][0] = relu_bits_of<8>(X, p, 0, one2); bw[p][1] = 0u; }
    }
    {
      f32x4 accr[4][2];  // block 0 carries the RGB rows 4..6; blocks 1..3 are whole-group padding
      {
        const float4 bv = *reinterpret_cast<const float4*>(b_rgb + 4 * g);
        const f32x4 v = {bv.x, bv.y, bv.z, bv.w}, z = {0.0f, 0.0f, 0.0f, 0.0f};
        accr[0][0] = v; accr[0][1] = v;
#pragma unroll
        for (int b = 1; b < 4; ++b) { accr[b][0] = z; accr[b][1] = z; }
      }
      const RowD db = d_bits(L), d_hid = mk_hid();
      gemm_t<4, 4, 8, GI_ACC>(accr, X, r, Wf, smem, [&](int) {  // the mid hidden rows and bits
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
          st16(d_hid, vrow[0] * 256 + 16 * g, kk * 64, X[kk][0], r);
          st16(d_hid, vrow[1] * 256 + 16 * g, kk * 64, X[kk][1], r);
        }
        st8(db, vrow[0] * 32 + 8 * g, 0, bw[0][0], bw[0][1], r);
        st8(db, vrow[1] * 32 + 8 * g, 0, bw[1][0], bw[1][1], r);
      });
#pragma unroll
      for (int p = 0; p < 2; ++p) {
        const float m0 = fast_sigmoid(accr[0][p][0]), m1 = fast_sigmoid(accr[0][p][1]), m2 = fast_sigmoid(accr[0][p][2]);
        const float t0 = __shfl(dcol[p][0], 48 + m, 64), t1 = __shfl(dcol[p][1], 48 + m, 64), t2 = __shfl(dcol[p][2], 48 + m, 64);
        if (g == 1 && valid[p]) {
          const size_t q = pc[p];
          if (a.saved.heads) *reinterpret_cast<float4*>(a.saved.heads + q * 8 + 4) = make_float4(m0, m1, m2, 0.0f);
          if (a.out.color) {
            if (a.mode == RSN_MODE_INF) {
              a.out.color[q * 3 + 0] = m0; a.out.color[q * 3 + 1] = m1; a.out.color[q * 3 + 2] = m2;
            } else {
              a.out.color[q * 3 + 0] = dcol[p][0] + t0 * m0;
              a.out.color[q * 3 + 1] = dcol[p][1] + t1 * m1;
              a.out.color[q * 3 + 2] = dcol[p][2] + t2 * m2;
            }
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
      // seed: the density-head row masked by the embedding's ReLU
#pragma unroll
      for (int kk = 0; kk < 8; ++kk) {
        const float4 lo = *reinterpret_cast<const float4*>(vden + 32 * kk + 8 * g);
        const float4 hi = *reinterpret_cast<const float4*>(vden + 32 * kk + 8 * g + 4);
        uint4v w;
        w[0] = pack2<false>(lo.x, lo.y); w[1] = pack2<false>(lo.z, lo.w);
        w[2] = pack2<false>(hi.x, hi.y); w[3] = pack2<false>(hi.z, hi.w);
#pragma unroll
        for (int p = 0; p < 2; ++p) {
          uint4v wm;
#pragma unroll
          for (int wi = 0; wi < 4; ++wi) wm[wi] = mask_word(w[wi], bwe[p][kk >> 2], (kk & 3) * 4 + wi, one2);
          X[kk][p] = __builtin_bit_cast(bf16x8, wm);
        }
      }
      float part[2][3] = {{0.0f, 0.0f, 0.0f}, {0.0f, 0.0f, 0.0f}}, raw[2][3] = {{0.0f, 0.0f, 0.0f}, {0.0f, 0.0f, 0.0f}};
      auto enc_part = [&]() {  // eacc = (encoded-input part)^T x gradient, folded at once with the lane's saved features
        const AsyncD d_enc = asyncd(a.saved.enc, (long long)p0 * 256, rows, 256);
        bf16x8 ft[4][2];
        ft[0][0] = ald16<0>(d_enc, vrow[0] * 256 + 16 * g); ft[0][1] = ald16<0>(d_enc, vrow[1] * 256 + 16 * g);
        ft[1][0] = ald16<64>(d_enc, vrow[0] * 256 + 16 * g); ft[1][1] = ald16<64>(d_enc, vrow[1] * 256 + 16 * g);
        ft[2][0] = ald16<128>(d_enc, vrow[0] * 256 + 16 * g); ft[2][1] = ald16<128>(d_enc, vrow[1] * 256 + 16 * g);
        ft[3][0] = ald16<192>(d_enc, vrow[0] * 256 + 16 * g); ft[3][1] = ald16<192>(d_enc, vrow[1] * 256 + 16 * g);
        r.since = 0;
        f32x4 eacc[8][2];
        gemm_t<8, 8, 8, GI_ZERO>(eacc, X, r, Wf, smem, NoHook());
        wait_loads(r);
        tie(ft);
        fold_enc<true>(eacc, ft, fq, g, part, raw);
      };
#pragma unroll 1
      for (int l = L - 1; l >= 1; --l) {
        if (l == P.skip_layer) enc_part();
        const AsyncD db = asyncd(a.saved.relu_bits, ((long long)(l - 1) * n_max + p0) * 32, rows, 32);
        u32x2t b0 = ald8(db, vrow[0] * 32 + 8 * g), b1 = ald8(db, vrow[1] * 32 + 8 * g);
        r.since = 0;
        f32x4 acc[16][2];
        gemm_t<16, 8, 8, GI_ZERO>(acc, X, r, Wf, smem, NoHook());
        wait_loads(r);
        tie(b0, b1);
        const unsigned bm[2][2] = {{b0.x, b0.y}, {b1.x, b1.y}};
        acc_to_x16_masked<16, 8, 8>(acc, X, bm, one2);
      }
      enc_part();
#pragma unroll
      for (int p = 0; p < 2; ++p) {
        float nrm[3];
#pragma unroll
        for (int c = 0; c < 3; ++c) {
          float tot = 6.283185307179586f * part[p][c] + (g == 0 ? raw[p][c] : 0.0f);
          tot += __shfl_xor(tot, 16, 64);
          tot += __shfl_xor(tot, 32, 64);
          nrm[c] = tot;
        }
        if (g == 0 && valid[p] && a.saved.normals) {
          const float len = fmaxf(sqrtf(nrm[0] * nrm[0] + nrm[1] * nrm[1] + nrm[2] * nrm[2]), 1e-12f);
          a.saved.normals[pc[p] * 3 + 0] = -(nrm[0] / len);
          a.saved.normals[pc[p] * 3 + 1] = -(nrm[1] / len);
          a.saved.normals[pc[p] * 3 + 2] = -(nrm[2] / len);
        }
      }
    }
  }
  ring_finish(r, wid);
}

// ================================================================================================ backward sweep
template <bool INPUT>
__global__ __launch_bounds__(512, 2) void rsn_field_bf16_bwd_kernel(const BwdJobs J) {
  constexpr int W = 256;
  __shared__ __attribute__((aligned(1024))) char smem[RT_RING_BYTES(RT_LEAD_BWD, RT_STAG_BWD)];
  const int lane = threadIdx.x & 63;
  const int wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const BwdShared& P = J.s;
  const TileJobs T = tile_space(J);
  if ((long long)blockIdx.x >= T.n_tiles) return;  // workgroup-uniform
  const float* __restrict__ pk = P.packed;
  const int L = P.num_layers;

  RingT<RT_LEAD_BWD, RT_STAG_BWD> r;
  bf16x8 Wf[RING_FIFO];
  // the walk: the whole transposed stream; without an input gradient the two encoded-input pieces are jumped over
  {
    const RsnPackedLayout& Y = P.L;
    int e0, j0, e1, j1;
    if (INPUT) { e0 = Y.t_g_end; j0 = Y.t_g_begin; e1 = -1; j1 = 0; }
    else if (Y.t_g_encskip >= 0) { e0 = Y.t_g_encskip; j0 = Y.t_g_encskip + 4; e1 = Y.t_g_enc0; j1 = Y.t_g_begin; }
    else { e0 = Y.t_g_enc0; j0 = Y.t_g_begin; e1 = -1; j1 = 0; }
    ring_start(r, pk, Y, smem, wid, lane, Y.t_g_begin, e0, j0, e1, j1, Wf);
  }
  const unsigned one2 = 0x00010001u;
  const bf16x8 zero8 = {};

  for (long long gtile = blockIdx.x; gtile < T.n_tiles; gtile += gridDim.x) {
    const int jk = (gtile >= T.tb1 ? 1 : 0) + (gtile >= T.tb2 ? 1 : 0);  // workgroup-uniform
    const BwdJob& a = J.j[jk];
    const unsigned n_points = (unsigned)(jk == 0 ? T.np0 : (jk == 1 ? T.np1 : T.np2));
    const unsigned tile = (unsigned)(gtile - (jk == 0 ? 0 : (jk == 1 ? T.tb1 : T.tb2)));
    const unsigned p0 = tile * 256 + wid * 32;
    const int rows = p0 >= n_points ? 0 : (int)(n_points - p0 < 32u ? n_points - p0 : 32u);
    const long long n_max = a.act_stride / W;
    int ln = lane;
    asm volatile("" : "+v"(ln));
    const int m = ln & 15, g = ln >> 4;
    unsigned vrow[2];
#pragma unroll
    for (int p = 0; p < 2; ++p) vrow[p] = (unsigned)(16 * p + m);
    auto d_bits = [&](int l) { return rowd(a.saved.relu_bits, ((long long)l * n_max + p0) * 32, rows, 32); };
    auto d_dy = [&](int l) { return rowd(a.gout.dy, ((long long)l * n_max + p0) * 512, rows, 512); };
    const RowD d_damid = rowd(a.gout.da_mid, (long long)p0 * 256, rows, 256);
    const RowD d_dbott = rowd(a.gout.d_bott, (long long)p0 * 512, rows, 512);

    // ---------------- per-sample epilogue gradients (reference autograd restated: see rsn_field_bwd.hip) -----------------
    bf16x8 X[8][2];
    bf16x8 XH[2];   // heads pre-activation gradients as the ninth K-step of the [bottleneck; heads]^T GEMM
    bf16x8 X0[2];   // RGB-head pre-activation gradient
#pragma unroll
    for (int p = 0; p < 2; ++p) {
      const unsigned pt = p0 + 16 * p + m;
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
        uint4v w = {0u, 0u, 0u, 0u};
        if (g == 1) { w[0] = pack2<false>(dz[0], dz[1]); w[1] = pack2<false>(dz[2], 0.0f); }
        X0[p] = __builtin_bit_cast(bf16x8, w);
      }
      float4 qh = make_float4(0.0f, 0.0f, 0.0f, 0.0f);  // heads rows 4 g + r
      if (a.mode != RSN_MODE_INF) {
        if (g == 0) {
          const long long ray = (long long)(q / (unsigned)a.S);
          const float rawd = a.fwd.raw_density[q];
          const float gs = a.gin.sigma ? a.gin.sigma[q] * live : 0.0f;
          qh.x = gs * fast_sigmoid(rawd + P.density_bias);  // softplus'
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
          const float sr = fast_sigmoid(hd.w);
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
        uint4v w = {pack2<false>(qh.x, qh.y), pack2<false>(qh.z, qh.w), 0u, 0u};
        XH[p] = __builtin_bit_cast(bf16x8, w);
      }
    }

    // ---------------- stage 1: d hidden = W_rgb^T dz (one group), masked by the mid hidden layer's ReLU -----------------
    {
      const AsyncD db = asyncd(a.saved.relu_bits, ((long long)L * n_max + p0) * 32, rows, 32);
      u32x2t b0 = ald8(db, vrow[0] * 32 + 8 * g), b1 = ald8(db, vrow[1] * 32 + 8 * g);
      r.since = 0;
      bf16x8 XR[2][2] = {{X0[0], X0[1]}, {zero8, zero8}};
      f32x4 acc[8][2];
      gemm_t<8, 2, 2, GI_ZERO>(acc, XR, r, Wf, smem, NoHook());
      wait_loads(r);
      tie(b0, b1);
      const unsigned bm[2][2] = {{b0.x, 0u}, {b1.x, 0u}};
      acc_to_x16_masked<8, 4, 8>(acc, X, bm, one2);  // X[0..3] = d a_mid
    }
    // ---------------- stage 2: d bottleneck = W_mid[:, 34:]^T d a_mid (4 groups; the d a_mid rows leave here) -----------------
    {
      f32x4 acc[16][2];
      gemm_t<16, 4, 8, GI_ZERO>(acc, X, r, Wf, smem, [&](int gi) {
        st16(d_damid, vrow[0] * 256 + 16 * g, gi * 64, X[gi][0], r);
        st16(d_damid, vrow[1] * 256 + 16 * g, gi * 64, X[gi][1], r);
      });
      acc_to_x16<16, 8, false, 8>(acc, X);  // X = d bottleneck
    }
    // ---------------- stage 3: d emb = [W_b; W_heads]^T [d b; dz_heads] (9 groups), masked by the embedding's ReLU -----------------
    {
      const AsyncD db = asyncd(a.saved.relu_bits, ((long long)(L - 1) * n_max + p0) * 32, rows, 32);
      u32x2t b0 = ald8(db, vrow[0] * 32 + 8 * g), b1 = ald8(db, vrow[1] * 32 + 8 * g);
      r.since = 0;
      bf16x8 X9[9][2];
#pragma unroll
      for (int kk = 0; kk < 8; ++kk) { X9[kk][0] = X[kk][0]; X9[kk][1] = X[kk][1]; }
      X9[8][0] = XH[0]; X9[8][1] = XH[1];
      f32x4 acc[16][2];
      gemm_t<16, 9, 9, GI_ZERO>(acc, X9, r, Wf, smem, [&](int gi) {
        if (gi < 8) {
          st16(d_dbott, vrow[0] * 512 + 16 * g, gi * 64, X9[gi][0], r);
          st16(d_dbott, vrow[1] * 512 + 16 * g, gi * 64, X9[gi][1], r);
        }
      });
      wait_loads(r);
      tie(b0, b1);
      const unsigned bm[2][2] = {{b0.x, b0.y}, {b1.x, b1.y}};
      acc_to_x16_masked<16, 8, 8>(acc, X, bm, one2);  // X = dy[L-1]
    }
    // ---------------- stage 4: trunk, layers L-1 .. 1 -----------------
    float part[2][3] = {{0.0f, 0.0f, 0.0f}, {0.0f, 0.0f, 0.0f}}, rawu[2][3] = {{0.0f, 0.0f, 0.0f}, {0.0f, 0.0f, 0.0f}};
    float fq[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) fq[t] = P.freqs[4 * g + t];
    auto enc_part = [&](int l_rows) {  // (encoded-input part)^T x dy[l_rows], folded into the variance gradient; dy[l_rows] may leave here
      const AsyncD d_enc = asyncd(a.saved.enc, (long long)p0 * 256, rows, 256);
      bf16x8 ft[4][2];
      ft[0][0] = ald16<0>(d_enc, vrow[0] * 256 + 16 * g); ft[0][1] = ald16<0>(d_enc, vrow[1] * 256 + 16 * g);
      ft[1][0] = ald16<64>(d_enc, vrow[0] * 256 + 16 * g); ft[1][1] = ald16<64>(d_enc, vrow[1] * 256 + 16 * g);
      ft[2][0] = ald16<128>(d_enc, vrow[0] * 256 + 16 * g); ft[2][1] = ald16<128>(d_enc, vrow[1] * 256 + 16 * g);
      ft[3][0] = ald16<192>(d_enc, vrow[0] * 256 + 16 * g); ft[3][1] = ald16<192>(d_enc, vrow[1] * 256 + 16 * g);
      r.since = 0;
      f32x4 eacc[8][2];
      if (l_rows >= 0) {
        const RowD dd = d_dy(l_rows);
        gemm_t<8, 8, 8, GI_ZERO>(eacc, X, r, Wf, smem, [&](int gi) {
          st16(dd, vrow[0] * 512 + 16 * g, (2 * gi) * 64, X[2 * gi][0], r);
          st16(dd, vrow[1] * 512 + 16 * g, (2 * gi) * 64, X[2 * gi][1], r);
          st16(dd, vrow[0] * 512 + 16 * g, (2 * gi + 1) * 64, X[2 * gi + 1][0], r);
          st16(dd, vrow[1] * 512 + 16 * g, (2 * gi + 1) * 64, X[2 * gi + 1][1], r);
        });
      } else {
        gemm_t<8, 8, 8, GI_ZERO>(eacc, X, r, Wf, smem, NoHook());
      }
      wait_loads(r);
      tie(ft);
      fold_enc<false>(eacc, ft, fq, g, part, rawu);
    };
#pragma unroll 1
    for (int l = L - 1; l >= 1; --l) {
      if (INPUT && l == P.skip_layer) enc_part(-1);
      const AsyncD db = asyncd(a.saved.relu_bits, ((long long)(l - 1) * n_max + p0) * 32, rows, 32);
      u32x2t b0 = ald8(db, vrow[0] * 32 + 8 * g), b1 = ald8(db, vrow[1] * 32 + 8 * g);
      r.since = 0;
      const RowD dd = d_dy(l);
      f32x4 acc[16][2];
      gemm_t<16, 8, 8, GI_ZERO>(acc, X, r, Wf, smem, [&](int gi) {  // reads (and keeps) dy[l]
        st16(dd, vrow[0] * 512 + 16 * g, gi * 64, X[gi][0], r);
        st16(dd, vrow[1] * 512 + 16 * g, gi * 64, X[gi][1], r);
      });
      wait_loads(r);
      tie(b0, b1);
      const unsigned bm[2][2] = {{b0.x, b0.y}, {b1.x, b1.y}};
      acc_to_x16_masked<16, 8, 8>(acc, X, bm, one2);  // X = dy[l-1]
    }
    if (INPUT) {
      enc_part(0);  // keeps dy[0]
      // ---------------- stage 5: gradient w.r.t. the Gaussian's variance -> pixel_area / sqradius -----------------
#pragma unroll
      for (int p = 0; p < 2; ++p) {
        float dvar[3];
#pragma unroll
        for (int c = 0; c < 3; ++c) {
          float tot = -0.5f * part[p][c];
          tot += __shfl_xor(tot, 16, 64);
          tot += __shfl_xor(tot, 32, 64);
          dvar[c] = tot;
        }
        const unsigned pt = p0 + 16 * p + m;
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
    } else {  // dy[0] has no GEMM behind it on this path: its rows leave here
      const RowD dd = d_dy(0);
#pragma unroll
      for (int kk = 0; kk < 8; ++kk) {
        st16(dd, vrow[0] * 512 + 16 * g, kk * 64, X[kk][0], r);
        st16(dd, vrow[1] * 512 + 16 * g, kk * 64, X[kk][1], r);
      }
    }
  }
  ring_finish(r, wid);
}

// ------------------------------------------------------------------------------------------------ launchers
int rsn_launch_field_bf16_train(long long n_tiles256, hipStream_t st, const FieldJobs& J) {
  bool normals = false, plain = false;
  for (int k = 0; k < J.n_jobs; ++k) {
    const FieldJob& a = J.j[k];
    RSN_REQUIRE(a.mode == RSN_MODE_FRUSTUM || a.mode == RSN_MODE_INF, RSN_ERR_UNSUPPORTED, "job %d: mode %d", k, a.mode);
    RSN_REQUIRE((long long)a.n_rays * a.S < (1LL << 31), RSN_ERR_UNSUPPORTED, "job %d: 2^31 points or more", k);
    if (a.saved.normals) normals = true; else plain = true;
  }
  RSN_REQUIRE(!(normals && plain), RSN_ERR_UNSUPPORTED,
              "evaluations with and without analytic normals cannot share a launch (the weight ring walks one program)");
  const int cus = rsn_device_cus();
  const long long grid = n_tiles256 < (long long)cus ? n_tiles256 : (long long)cus;
  if (normals) hipLaunchKernelGGL(rsn_field_bf16_train_kernel<true>, dim3((unsigned)grid), dim3(512), 0, st, J);
  else hipLaunchKernelGGL(rsn_field_bf16_train_kernel<false>, dim3((unsigned)grid), dim3(512), 0, st, J);
  RSN_HIP(hipGetLastError());
  return RSN_OK;
}

int rsn_launch_field_bf16_bwd(long long n_tiles256, hipStream_t st, const BwdJobs& J) {
  bool input = false;
  for (int k = 0; k < J.n_jobs; ++k) {
    RSN_REQUIRE((long long)J.j[k].n_rays * J.j[k].S < (1LL << 31), RSN_ERR_UNSUPPORTED, "job %d: 2^31 points or more", k);
    input = input || J.j[k].need_input_grad != 0;
  }
  for (int k = 0; k < J.n_jobs; ++k)
    RSN_REQUIRE((J.j[k].need_input_grad != 0) == input, RSN_ERR_UNSUPPORTED,
                "evaluations with and without an input gradient cannot share a launch (the weight ring walks one program)");
  const int cus = rsn_device_cus();
  const long long grid = n_tiles256 < (long long)cus ? n_tiles256 : (long long)cus;
  if (input) hipLaunchKernelGGL(rsn_field_bf16_bwd_kernel<true>, dim3((unsigned)grid), dim3(512), 0, st, J);
  else hipLaunchKernelGGL(rsn_field_bf16_bwd_kernel<false>, dim3((unsigned)grid), dim3(512), 0, st, J);
  RSN_HIP(hipGetLastError());
  return RSN_OK;
}
