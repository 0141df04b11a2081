// rsn_field_f32_ring.hip -- the exact-fp32 TRAINING FORWARD (with the analytic-normal sweep) on the LDS weight ring, width 256.
// NOT part of librsn_hip.so: a round-4 experiment kept as a probe.  Bit-identical to rsn_field_kernel<8, true, 0> in
// every output and saved buffer, and 7 % SLOWER (MFMA busy 82.6 % against 87.1 %).  Build and compare:
//   python -c "from tools._variant import build_variant; print(build_variant(['RSN_F32_RING_TRAIN'], ['tools/probes/rsn_field_f32_ring.hip']))"
//   RSN_LIBRARY=<that .so> python bench.py --no-secondary --no-cpu-baseline
//
// rsn_field_kernel<8, true, 0> (rsn_field_kernel.h) keeps a wave's activations in an LDS slab (148 KiB per workgroup: nothing else
// fits) and lets every wave stream the 2.5 MB of fp32 weight fragments per 32 points through buffer loads from L1 / L2: its K loops
// run at 98 % of the fp32 MFMA issue rate without that stream and at 90 % with it, and its epilogues (ReLU, mask bits, slab writes,
// bias re-initialisation) run with the matrix pipe idle -- one wave per SIMD, nothing to overlap with: 0.85 of the fp32-MFMA peak.
// This kernel computes THE SAME arithmetic in the same order (every accumulator sees the products of its K-iterations in the order
// of the slab kernel: bit-identical saved rows and outputs) with the structure of the round-4 ring kernels (rsn_field_x6_train.hip):
//   * the fp32 fragments (1 KiB = [lane][4]: four K-steps of v_mfma_f32_32x32x2_f32 for one 32-row block) arrive ONCE PER WORKGROUP
//     through an LDS ring by LDS-DMA, as one linear stream in consumption order (RsnPackedLayout.f_stream, rsn_pack.hip);
//   * activations never leave the registers: the D layout of the MFMA (lane (i, h): rows 8 q + 4 h + s of column i) is exactly the
//     B operand of K-iteration it = 4 nb + q, K-step s, of the next layer -- the accumulators of a layer ARE the next layer's input
//     (128 registers; one wave per SIMD has 512); ReLU / ReLU mask, mask bits and the saved-row store of K-iteration `it` happen
//     where the iteration is consumed, between MFMAs of 64 cycles each;
//   * a layer hand-off is 128 register copies; encode, heads, SH and the normal fold are the code of rsn_field_kernel.h.
// Saved buffers, outputs, layouts: those of rsn_field_kernel<8, true, 0> (the backward sweep and the weight gradients are unchanged).
#include "rsn_ringt.h"

#ifndef F32_LEAD
#define F32_LEAD 5   // ring groups in flight ahead of the one being read (6 slots = 96 KiB; this kernel needs no other LDS but tables)
#endif
#define F32_PPW 4    // LDS-DMA pieces per wave and group (4 waves)
#define F32_TABLE_FLOATS (RING_MAX_LAYERS * 256 + 288 + 128 + 32 + 256)

// ------------------------------------------------------------------------------------------------ the ring (4 waves)
struct RingF {
  static constexpr int SLOTS = F32_LEAD + 1;
  const char* src;     // f_stream base + wave * PPW KiB
  unsigned lane16, lds_dst;
  int issue_grp, issue_slot;
  int e0, j0, e1, j1;  // the walk: group e_i - 1 is followed by group j_i
  unsigned rd_base, rd_cur;
  int cur_slot;
  int since;           // vector-memory operations issued since the last batch of asynchronous loads
  int c0;              // counted vector-memory operations (row stores) issued since the last group boundary ...
  int cp[F32_LEAD - 1];  // ... and in the LEAD - 1 intervals before it (cp[0] the newest)
};
__device__ __forceinline__ void ringf_issue(RingF& r) {
  const char* g = r.src + (size_t)r.issue_grp * RING_GROUP_BYTES;
  const unsigned d = __builtin_amdgcn_readfirstlane(r.lds_dst + (unsigned)r.issue_slot * RING_GROUP_BYTES);
#pragma unroll
  for (int i = 0; i < F32_PPW; ++i) glds16(g + i * 1024, r.lane16, d + i * 1024);
  r.since += F32_PPW;
  int n = r.issue_grp + 1;
  n = (n == r.e0) ? r.j0 : ((n == r.e1) ? r.j1 : n);
  r.issue_grp = n;
  r.issue_slot = (r.issue_slot + 1 == RingF::SLOTS) ? 0 : r.issue_slot + 1;
}
// Group boundary in front of the first READ of a group: that group's LDS-DMA was issued LEAD boundaries ago; behind it in the in-order
// vmcnt queue sit the counted stores of the last LEAD intervals and the DMA of LEAD - 1 boundaries.  The slot of the group read
// before this one is refilled: every wave has issued its reads of it (they return long before the DMA's data arrives).
__device__ __forceinline__ void ringf_sync(RingF& r) {
  int n = r.c0 + F32_PPW * (F32_LEAD - 1);
#pragma unroll
  for (int i = 0; i < F32_LEAD - 1; ++i) n += r.cp[i];
#ifndef RSN_F32_NO_WAIT      // (RSN_F32_NO_*: timing ablations, diagnostic builds only; wrong results by construction)
  wait_vm(n);
#endif
#ifndef RSN_F32_NO_BARRIER
  asm volatile("s_barrier" ::: "memory");
#endif
  ringf_issue(r);
#pragma unroll
  for (int i = F32_LEAD - 2; i > 0; --i) r.cp[i] = r.cp[i - 1];
  r.cp[0] = r.c0;
  r.c0 = 0;
  r.cur_slot = (r.cur_slot + 1 == RingF::SLOTS) ? 0 : r.cur_slot + 1;
  r.rd_cur = r.rd_base + (unsigned)r.cur_slot * RING_GROUP_BYTES;
}

// acc[nb] (+)= W(it, nb) * X(it): fragment f = it * NBO + nb of the stream (a whole number of groups per GEMM), four K-steps each.
// src(it) yields the lane's float4 of K-iteration `it` (the caller's ReLU / mask applied), fetched once per iteration; hook(it) runs
// beside it (the caller's saved-row store, mask bits).  The fragments of iteration it + 1 are read from the ring while the 4 NBO
// MFMAs of iteration it run (two fragment buffers); MFMA order inside an iteration: K-step-major over the blocks, like mma4.
template <int NBO, int KITS, int INIT, class SRC, class HOOK>
__device__ __forceinline__ void gemm_f(f32x16 (&acc)[NBO], SRC&& src, RingF& r, const char* smem, HOOK&& hook,
                                       const float* bias = nullptr, int h = 0) {
  static_assert((NBO * KITS) % RSN_RING_GROUP_FRAGS == 0, "a GEMM is a whole number of ring groups");
  float4 W[2][NBO];
  auto fetch1 = [&](int it, int nb, float4 (&w)[NBO]) {
    const int f = it * NBO + nb;
    if (f % RSN_RING_GROUP_FRAGS == 0) ringf_sync(r);
    w[nb] = *reinterpret_cast<const float4*>(smem + r.rd_cur + (f % RSN_RING_GROUP_FRAGS) * 1024);
  };
#pragma unroll
  for (int nb = 0; nb < NBO; ++nb) fetch1(0, nb, W[0]);
  float4 x = src(0);
  hook(0);
#pragma unroll
  for (int it = 0; it < KITS; ++it) {
    // One wave per SIMD: whatever is not an MFMA must sit BETWEEN MFMAs (64 cycles each), or the matrix pipe waits for it.  Behind
    // MFMA j of this iteration's 4 NBO: j < NBO: fragment j of iteration it + 1 (ring boundary included); j = NBO: its B operand;
    // j = NBO + 1: its hook (saved-row store, mask bits).  sched_barrier pins the placement.
    float4 xn = x;
    const float xs[4] = {x.x, x.y, x.z, x.w};
#pragma unroll
    for (int s = 0; s < 4; ++s)
#pragma unroll
      for (int nb = 0; nb < NBO; ++nb) {
        const int j = s * NBO + nb;
        const float4 w4 = W[it & 1][nb];
        const float w = s == 0 ? w4.x : (s == 1 ? w4.y : (s == 2 ? w4.z : w4.w));
        if (INIT != GI_ACC && it == 0 && s == 0) {
          f32x16 c;
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            float4 bv = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
            if (INIT == GI_BIAS) bv = *reinterpret_cast<const float4*>(bias + nb * 32 + 8 * q + 4 * h);
            c[4 * q + 0] = bv.x; c[4 * q + 1] = bv.y; c[4 * q + 2] = bv.z; c[4 * q + 3] = bv.w;
          }
          acc[nb] = __builtin_amdgcn_mfma_f32_32x32x2f32(w, xs[s], c, 0, 0, 0);
        } else {
          acc[nb] = __builtin_amdgcn_mfma_f32_32x32x2f32(w, xs[s], acc[nb], 0, 0, 0);
        }
        if (it + 1 < KITS) {
          if (j < NBO) fetch1(it + 1, j, W[(it + 1) & 1]);
          if (j == NBO) xn = src(it + 1);
          if (j == NBO + 1) hook(it + 1);
        }
        __builtin_amdgcn_sched_barrier(0);
      }
    x = xn;
  }
}

// the lane's float4 of K-iteration `it` from the accumulators of the GEMM before: block it / 4, elements 4 (it % 4) .. + 3
template <bool RELU, int NB>
__device__ __forceinline__ float4 acc_it(const f32x16 (&A)[NB], int it) {
  const f32x16 v = A[it >> 2];
  const int q = it & 3;
  float4 o = make_float4(v[4 * q], v[4 * q + 1], v[4 * q + 2], v[4 * q + 3]);
  if (RELU) {
    o.x = relu_f(o.x); o.y = relu_f(o.y); o.z = relu_f(o.z); o.w = relu_f(o.w);
  }
  return o;
}
// ... masked by the ReLU bits of the layer the gradient enters (rsn_field_saved.relu_bits: word nb / 2, bit (nb & 1) * 16 + r)
template <int NB>
__device__ __forceinline__ float4 acc_it_masked(const f32x16 (&A)[NB], const unsigned (&bits)[NB / 2 > 2 ? NB / 2 : 2], int it) {
  const f32x16 v = A[it >> 2];
  const int q = it & 3, nb = it >> 2;
  const int word = (int)bits[nb >> 1], base = (nb & 1) * 16 + 4 * q;
  return make_float4(__uint_as_float(__float_as_uint(v[4 * q + 0]) & bit_mask(word, base + 0)),
                     __uint_as_float(__float_as_uint(v[4 * q + 1]) & bit_mask(word, base + 1)),
                     __uint_as_float(__float_as_uint(v[4 * q + 2]) & bit_mask(word, base + 2)),
                     __uint_as_float(__float_as_uint(v[4 * q + 3]) & bit_mask(word, base + 3)));
}
// mask bits of the relu'd float4 of K-iteration `it` into the words of rsn_field_saved.relu_bits
template <int NW>
__device__ __forceinline__ void note_bits(unsigned (&bw)[NW], const float4 x, int it) {
  const int nb = it >> 2, q = it & 3;
  const unsigned v[4] = {__float_as_uint(x.x), __float_as_uint(x.y), __float_as_uint(x.z), __float_as_uint(x.w)};
#pragma unroll
  for (int s = 0; s < 4; ++s) {
    const unsigned bit = v[s] < 1u ? v[s] : 1u;   // relu'd value > 0  <=>  its bits are non-zero: min(bits, 1)
    const int pos = (nb & 1) * 16 + 4 * q + s;
    bw[nb >> 1] = (pos == 0) ? bit : (bw[nb >> 1] | (bit << pos));
  }
}
__device__ __forceinline__ u32x4t aldq_f(const AsyncD& d, unsigned voff) { return __builtin_bit_cast(u32x4t, ald16<0>(d, voff)); }
// a lane's float4 of a saved row (counted; offset through voffset / immediate: see st16)
__device__ __forceinline__ void st_row(const RowD& d, unsigned voff, const float4 v, RingF& r) {
#ifndef RSN_RT_NO_STORES
  const u32x4t o = {__float_as_uint(v.x), __float_as_uint(v.y), __float_as_uint(v.z), __float_as_uint(v.w)};
  __builtin_amdgcn_raw_buffer_store_b128(o, d.r, voff, 0, RSN_SAVED_ROW_AUX);
  r.c0 += 1;
  r.since += 1;
#endif
}

// frustum_to_contracted (rsn_field_common.h) without its divergent branch: both sides are formed and selected per lane -- the same
// operations on the same values, bit-identical results.  (In this kernel the branch form compiled to code that gave the contracted
// samples of the no-normals instantiation a different covariance; not understood -- 58 spilled scalar registers around a divergent
// region are the suspect -- and not worth the risk: a dozen extra VALU per point.)
__device__ __forceinline__ void frustum_to_contracted_sel(const float o[3], const float d[3], float pa, float t0, float t1,
                                                          float mean_c[3], float var_c[3]) {
  const float radius = sqrtf(pa) / 1.7724538509055159f;
  const float mu = (t0 + t1) / 2.0f;
  const float hw = (t1 - t0) / 2.0f;
  const float hw2 = hw * hw, mu2 = mu * mu;
  const float den = 3.0f * mu2 + hw2;
  const float tmean = mu + (2.0f * mu * hw2) / den;
  float mean[3];
#pragma unroll
  for (int c = 0; c < 3; ++c) mean[c] = o[c] + d[c] * tmean;
  const float hw4 = hw2 * hw2;
  const float var_t = hw2 / 3.0f - 0.26666666666666666f * ((hw4 * (12.0f * mu2 - hw2)) / (den * den));
  const float var_r =
      (radius * radius) * (mu2 / 4.0f + 0.4166666666666667f * hw2 - (0.26666666666666666f * hw4) / den);
  const float dmag = fmaxf(d[0] * d[0] + d[1] * d[1] + d[2] * d[2], 1e-10f);
  float S[3][3];
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int j = 0; j < 3; ++j)
      S[i][j] = var_t * (d[i] * d[j]) + var_r * ((i == j ? 1.0f : 0.0f) - d[i] * (d[j] / dmag));
  const float n2 = mean[0] * mean[0] + mean[1] * mean[1] + mean[2] * mean[2];
  const float n = sqrtf(n2);
  const bool far_ = n > 1.0f;
  const float n2s = far_ ? n2 : 1.0f;   // (the near side's values are never used: no division by a zero norm)
  const float sc = (2.0f * n - 1.0f) / n2s;
  float J[3][3];
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      const float eye = (i == j) ? 1.0f : 0.0f;
      const float outer = mean[i] * mean[j] / n2s;
      J[i][j] = ((2.0f * n - 2.0f) * (eye - outer) + eye) / n2s;
    }
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    float acc = 0.0f;
#pragma unroll
    for (int b = 0; b < 3; ++b) {
      const float js = J[i][0] * S[0][b] + J[i][1] * S[1][b] + J[i][2] * S[2][b];
      acc += js * J[b][i];
    }
    mean_c[i] = far_ ? sc * mean[i] : mean[i];
    var_c[i] = fmaxf(far_ ? acc : S[i][i], 0.0f);
  }
}

// ================================================================================================ training forward
template <bool NORMALS>
__global__ __launch_bounds__(256) void rsn_field_f32_train_kernel(const FieldJobs J) {
  constexpr int W = 256, NB = 8;
  constexpr int RB = RingF::SLOTS * RING_GROUP_BYTES;
  __shared__ __attribute__((aligned(1024))) char smem[RB + F32_TABLE_FLOATS * 4];
  const int lane = threadIdx.x & 63;
  const int wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  float* bias = reinterpret_cast<float*>(smem + RB);        // natural row order: the accumulator rows are natural here
  const float* b_bh = bias + RING_MAX_LAYERS * 256;
  const float* b_mid = b_bh + 288;
  const float* b_rgb = b_mid + 128;
  const float* vden = b_rgb + 32;

  const FieldShared& P = J.s;
  const TileJobs T = tile_space<128>(J);
  if ((long long)blockIdx.x >= T.n_tiles) return;  // workgroup-uniform
  const float* __restrict__ pk = P.packed;
  const int L = P.num_layers;

  for (int i = threadIdx.x; i < L * 256; i += 256) bias[i] = pk[P.L.b[i >> 8] + (i & 255)];
  for (int i = threadIdx.x; i < 288; i += 256) bias[RING_MAX_LAYERS * 256 + i] = pk[P.L.b_bh + i];
  if (threadIdx.x < 128) bias[RING_MAX_LAYERS * 256 + 288 + threadIdx.x] = pk[P.L.b_mid + threadIdx.x];
  if (threadIdx.x < 32) bias[RING_MAX_LAYERS * 256 + 288 + 128 + threadIdx.x] = pk[P.L.b_rgb + threadIdx.x];
  bias[RING_MAX_LAYERS * 256 + 288 + 128 + 32 + threadIdx.x] = pk[P.L.v_density + threadIdx.x];

  RingF r;
  {
    r.src = reinterpret_cast<const char*>(pk + P.L.f_stream) + wid * (F32_PPW * 1024);
    r.lane16 = (unsigned)lane * 16u;
    r.lds_dst = (unsigned)(size_t)(__attribute__((address_space(3))) char*)smem + (unsigned)wid * (F32_PPW * 1024u);
    r.issue_grp = 0;
    r.issue_slot = 0;
    // the walk: forward stream [0, f_groups); with the normal sweep then the transposed trunk [f_groups, ft_end); again
    r.e0 = NORMALS ? P.L.ft_end : P.L.f_groups; r.j0 = 0; r.e1 = -1; r.j1 = 0;
    r.rd_base = (unsigned)lane * 16u;
    r.cur_slot = RingF::SLOTS - 1;   // the first boundary makes slot 0 the current one
    r.rd_cur = r.rd_base;
    r.c0 = 0;
    r.since = 0;
#pragma unroll
    for (int i = 0; i < F32_LEAD - 1; ++i) r.cp[i] = 0;
    __syncthreads();  // nothing in flight yet (also publishes the LDS tables)
#pragma unroll
    for (int gq = 0; gq < F32_LEAD; ++gq) ringf_issue(r);
  }

  for (long long gtile = blockIdx.x; gtile < T.n_tiles; gtile += gridDim.x) {
    const int jk = (gtile >= T.tb1 ? 1 : 0) + (gtile >= T.tb2 ? 1 : 0);  // workgroup-uniform
    const FieldJob& a = J.j[jk];
    const long long n_points = jk == 0 ? T.np0 : (jk == 1 ? T.np1 : T.np2);
    const long long tile = gtile - (jk == 0 ? 0 : (jk == 1 ? T.tb1 : T.tb2));
    const long long p0 = tile * 128 + wid * 32;   // every wave walks every tile (barriers, DMA shares); rows = 0 past the end
    const int rows = p0 >= n_points ? 0 : (int)(n_points - p0 < 32 ? n_points - p0 : 32);
    const long long n_max = a.act_stride / W;
    int ln = lane;
    asm volatile("" : "+v"(ln));  // opaque per-tile lane id (see rsn_field_kernel.h)
    const int m = ln & 31, h = ln >> 5;
    const long long p = p0 + m;
    const bool valid = p < n_points;
    const long long pc = valid ? p : (n_points > 0 ? n_points - 1 : 0);
    auto row_d = [&](float* base, long long elem, int row_elems) {
      return rowd(base, elem * 4, rows, row_elems * 4);
    };
    auto bits_at = [&](int l) -> unsigned* { return a.saved.relu_bits + (((long long)l * n_max + pc) * 2 + h) * 4; };

    // ---------------- encode (rsn_field_kernel.h: the same operations in the same order) -----------------
    float mc[3] = {0.0f, 0.0f, 0.0f}, vc[3] = {0.0f, 0.0f, 0.0f}, vd[3] = {0.0f, 0.0f, 0.0f};
    bool has_dir = true;
    if (a.mode == RSN_MODE_FRUSTUM) {
      const long long ray = pc / a.S;
      const int s = (int)(pc - ray * a.S);
      float o[3];
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        o[c] = a.origins[ray * 3 + c];
        vd[c] = a.directions[ray * 3 + c];
      }
      #ifdef F32_BRANCHY  // (the shared function with its divergent branch: see frustum_to_contracted_sel)
      frustum_to_contracted(o, vd, a.pixel_area[ray], a.bins[ray * (a.S + 1) + s], a.bins[ray * (a.S + 1) + s + 1], mc, vc);
#else
      frustum_to_contracted_sel(o, vd, a.pixel_area[ray], a.bins[ray * (a.S + 1) + s], a.bins[ray * (a.S + 1) + s + 1], mc, vc);
#endif
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
    // slot u of this lane (lane half h owns frequencies 8h..8h+7): u < 24 sine, 24 <= u < 48 cosine features, 48..50 raw coordinates
    // (h == 0); K-iteration it = u / 4, K-step u % 4 (rsn_pack.hip cols_encoding)
    float feat[56];
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      const float sx = 6.283185307179586f * mc[c];
#pragma unroll
      for (int jj = 0; jj < 8; ++jj) {
        const float f = h ? P.freqs[8 + jj] : P.freqs[jj];
        const float ang = sx * f;
        const float e = expf(-0.5f * (vc[c] * (f * f)));
        feat[c * 8 + jj] = e * sin_big(ang);
        feat[24 + c * 8 + jj] = e * sin_big(ang + 1.5707963267948966f);
      }
    }
#pragma unroll
    for (int c = 0; c < 3; ++c) feat[48 + c] = h == 0 ? mc[c] : 0.0f;
#pragma unroll
    for (int u = 51; u < 56; ++u) feat[u] = 0.0f;
    auto src_enc = [&](int it) { return make_float4(feat[4 * it], feat[4 * it + 1], feat[4 * it + 2], feat[4 * it + 3]); };
    if (a.saved.enc && valid) {  // [N,104] in slot order
      float* row = a.saved.enc + pc * RSN_K_ENC_PAD;
#pragma unroll
      for (int it = 0; it < RSN_ENC_ITS; ++it) *reinterpret_cast<float4*>(row + it * 8 + 4 * h) = src_enc(it);
    }

    // ---------------- trunk -----------------
    f32x16 A[NB];  // the pre-activation of the trunk layer just finished (ReLU is applied where it is consumed)
    gemm_f<NB, 14, GI_BIAS>(A, src_enc, r, smem, [](int) {}, bias, h);
#pragma unroll 1
    for (int l = 1; l < L; ++l) {
      f32x16 B[NB];
      unsigned bw[4] = {0u, 0u, 0u, 0u};
      const RowD da = row_d(a.saved.act, (l - 1) * a.act_stride + p0 * W, W);
      float4 xk;
      gemm_f<NB, 32, GI_BIAS>(B, [&](int it) { xk = acc_it<true>(A, it); return xk; }, r, smem, [&](int it) {
        st_row(da, (unsigned)(m * 1024 + it * 32 + 16 * h), xk, r);   // act[l-1]: one store per K-iteration, where it is consumed
        note_bits(bw, xk, it);
      }, bias + l * 256, h);
      if (a.saved.relu_bits && valid) *reinterpret_cast<uint4*>(bits_at(l - 1)) = make_uint4(bw[0], bw[1], bw[2], bw[3]);
      if (l == P.skip_layer) gemm_f<NB, 14, GI_ACC>(B, src_enc, r, smem, [](int) {});
#pragma unroll
      for (int nb = 0; nb < NB; ++nb) A[nb] = B[nb];
    }

    // ---------------- bottleneck + heads (one GEMM, N = W + 32) on the embedding = ReLU(A) -----------------
    unsigned bwe[4] = {0u, 0u, 0u, 0u};  // the embedding's mask bits: the seed mask of the normal sweep
    f32x16 Bt[NB + 1];
    {
      const RowD da = row_d(a.saved.act, (L - 1) * a.act_stride + p0 * W, W);
      float4 xk;
      gemm_f<NB + 1, 32, GI_BIAS>(Bt, [&](int it) { xk = acc_it<true>(A, it); return xk; }, r, smem, [&](int it) {
        st_row(da, (unsigned)(m * 1024 + it * 32 + 16 * h), xk, r);
        note_bits(bwe, xk, it);
      }, b_bh, h);
      if (a.saved.relu_bits && valid) *reinterpret_cast<uint4*>(bits_at(L - 1)) = make_uint4(bwe[0], bwe[1], bwe[2], bwe[3]);
    }
    float dcol[3], tcol[3], rho;
    {
      const float r0 = Bt[NB][0], r1 = Bt[NB][1], r2 = Bt[NB][2], r3 = Bt[NB][3];
      const float r4 = Bt[NB][4], r5 = Bt[NB][5], r6 = Bt[NB][6];
      // h == 0: r0 raw density, r1..r3 normals, r4 roughness.   h == 1: r0..r2 diff, r4..r6 tint.
      const float rough_raw = __shfl(r4, m, 64);
      rho = softplus_f(rough_raw);
      dcol[0] = sigmoid_f(r0); dcol[1] = sigmoid_f(r1); dcol[2] = sigmoid_f(r2);
      tcol[0] = sigmoid_f(r4); tcol[1] = sigmoid_f(r5); tcol[2] = sigmoid_f(r6);
      if (a.mode != RSN_MODE_INF && valid) {
        if (h == 0) {
          float nrm = fmaxf(sqrtf(r1 * r1 + r2 * r2 + r3 * r3), 1e-12f);
          float nx = -(r1 / nrm), ny = -(r2 / nrm), nz = -(r3 / nrm);
          nrm = fmaxf(sqrtf(nx * nx + ny * ny + nz * nz), 1e-12f);
          nx /= nrm; ny /= nrm; nz /= nrm;
          if (a.out.sigma) a.out.sigma[pc] = softplus_f(r0 + P.density_bias);
          if (a.out.raw_density) a.out.raw_density[pc] = r0;
          if (a.out.pred_normals) {
            a.out.pred_normals[pc * 3 + 0] = nx;
            a.out.pred_normals[pc * 3 + 1] = ny;
            a.out.pred_normals[pc * 3 + 2] = nz;
          }
          if (a.out.n_dot_d) a.out.n_dot_d[pc] = vd[0] * nx + vd[1] * ny + vd[2] * nz;
          if (a.out.roughness) a.out.roughness[pc] = sigmoid_f(r4);
          if (a.out.raw_roughness) a.out.raw_roughness[pc] = r4;
        } else {
          if (a.out.diff) {
            a.out.diff[pc * 3 + 0] = dcol[0]; a.out.diff[pc * 3 + 1] = dcol[1]; a.out.diff[pc * 3 + 2] = dcol[2];
          }
          if (a.out.tint) {
            a.out.tint[pc * 3 + 0] = tcol[0]; a.out.tint[pc * 3 + 1] = tcol[1]; a.out.tint[pc * 3 + 2] = tcol[2];
          }
        }
      }
      if (a.saved.heads && valid && h == 0) *reinterpret_cast<float4*>(a.saved.heads + pc * 8) = make_float4(r1, r2, r3, r4);
    }
    // ---------------- SH-34 of the view direction, attenuated by softplus roughness -----------------
    float shin[32];  // this lane's SH slots u = 4 it + s (lane half h owns components 17h .. 17h+16); its 5..7 are zero padding
    {
      float sh[34];
      if (has_dir) {
        sh34_attenuated(vd[0], vd[1], vd[2], rho, sh);
      } else {
#pragma unroll
        for (int i = 0; i < 34; ++i) sh[i] = 0.0f;
      }
#pragma unroll
      for (int u = 0; u < 32; ++u) shin[u] = (u < 17) ? (h ? sh[17 + u] : sh[u]) : 0.0f;
      if (a.saved.sh && valid) {
#pragma unroll
        for (int it = 0; it < RSN_SH_ITS; ++it)
          *reinterpret_cast<float4*>(a.saved.sh + pc * RSN_K_SH_PAD + it * 8 + 4 * h) =
              make_float4(shin[4 * it], shin[4 * it + 1], shin[4 * it + 2], shin[4 * it + 3]);
      }
    }
    // ---------------- mlp_mid + RGB head -----------------
    {
      f32x16 accm[4];
      gemm_f<4, 8, GI_BIAS>(accm, [&](int it) { return make_float4(shin[4 * it], shin[4 * it + 1], shin[4 * it + 2], shin[4 * it + 3]); },
                            r, smem, [](int) {}, b_mid, h);
      {
        const RowD db = row_d(a.saved.bott, p0 * W, W);
        float4 xk;
        gemm_f<4, 32, GI_ACC>(accm, [&](int it) { xk = acc_it<false>(Bt, it); return xk; }, r, smem, [&](int it) {
          st_row(db, (unsigned)(m * 1024 + it * 32 + 16 * h), xk, r);   // the bottleneck rows
        });
      }
      f32x16 accr[1];
      unsigned bwh[2] = {0u, 0u};
      {
        const RowD dh = row_d(a.saved.hid, p0 * 128, 128);
        float4 xk;
        gemm_f<1, 16, GI_BIAS>(accr, [&](int it) { xk = acc_it<true>(accm, it); return xk; }, r, smem, [&](int it) {
          st_row(dh, (unsigned)(m * 512 + it * 32 + 16 * h), xk, r);   // the mid hidden rows
          note_bits(bwh, xk, it);
        }, b_rgb, h);
        if (a.saved.relu_bits && valid) {
          unsigned* bp = bits_at(L);
          bp[0] = bwh[0];
          bp[1] = bwh[1];
        }
      }
      if (h == 1 && valid) {
        const float m0 = sigmoid_f(accr[0][0]);
        const float m1 = sigmoid_f(accr[0][1]);
        const float m2 = sigmoid_f(accr[0][2]);
        if (a.saved.heads) *reinterpret_cast<float4*>(a.saved.heads + pc * 8 + 4) = make_float4(m0, m1, m2, 0.0f);
        if (a.out.color) {
          if (a.mode == RSN_MODE_INF) {
            a.out.color[pc * 3 + 0] = m0; a.out.color[pc * 3 + 1] = m1; a.out.color[pc * 3 + 2] = m2;
          } else {
            a.out.color[pc * 3 + 0] = dcol[0] + tcol[0] * m0;
            a.out.color[pc * 3 + 1] = dcol[1] + tcol[1] * m1;
            a.out.color[pc * 3 + 2] = dcol[2] + tcol[2] * m2;
          }
        }
      }
    }

    // ---------------- analytic normals = -normalize(d raw_density / d contracted mean) (rsn_field_kernel.h, same order) -----------
    if (NORMALS) {
      // G: the gradient entering a layer's output BEFORE that layer's ReLU mask bm (applied where it is consumed); seed: the
      // density-head row, masked by the embedding's ReLU
      f32x16 G[NB];
#pragma unroll
      for (int it = 0; it < 32; ++it) {
        const float4 w = *reinterpret_cast<const float4*>(vden + it * 8 + 4 * h);
        G[it >> 2][4 * (it & 3) + 0] = w.x; G[it >> 2][4 * (it & 3) + 1] = w.y;
        G[it >> 2][4 * (it & 3) + 2] = w.z; G[it >> 2][4 * (it & 3) + 3] = w.w;
      }
      unsigned bm[4] = {bwe[0], bwe[1], bwe[2], bwe[3]};
      f32x16 eacc[4];
      bool e_live = false;  // (compile-time known along both paths: the first encoded-input GEMM starts the accumulators)
#pragma unroll 1
      for (int l = L - 1; l >= 1; --l) {
        if (l == P.skip_layer) {
          gemm_f<4, 32, GI_ZERO>(eacc, [&](int it) { return acc_it_masked<NB>(G, bm, it); }, r, smem, [](int) {});
          e_live = true;
        }
        const AsyncD db = asyncd(a.saved.relu_bits, (((long long)(l - 1) * n_max + p0) * 2) * 16, rows * 2, 16);
        u32x4t b0 = aldq_f(db, (unsigned)((m * 2 + h) * 16));
        r.since = 0;
        f32x16 acc[NB];
        gemm_f<NB, 32, GI_ZERO>(acc, [&](int it) { return acc_it_masked<NB>(G, bm, it); }, r, smem, [](int) {});
        wait_vm(r.since);
        asm volatile("" : "+v"(b0)::"memory");
        bm[0] = b0.x; bm[1] = b0.y; bm[2] = b0.z; bm[3] = b0.w;
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) G[nb] = acc[nb];
      }
      if (e_live) {
        gemm_f<4, 32, GI_ACC>(eacc, [&](int it) { return acc_it_masked<NB>(G, bm, it); }, r, smem, [](int) {});
      } else {
        gemm_f<4, 32, GI_ZERO>(eacc, [&](int it) { return acc_it_masked<NB>(G, bm, it); }, r, smem, [](int) {});
      }
      // eacc: gradient w.r.t. this lane's encoded inputs, slot u = 4 it + s <-> block it / 4, element 4 (it % 4) + s
      float nrm[3];
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        const float sx = 6.283185307179586f * mc[c];
        float part = 0.0f;
#pragma unroll
        for (int jj = 0; jj < 8; ++jj) {
          const float f = h ? P.freqs[8 + jj] : P.freqs[jj];
          const float ang = sx * f;
          const float e = expf(-0.5f * (vc[c] * (f * f)));
          const int u = c * 8 + jj, u2 = u + 24;
          const float gs = eacc[u >> 4][u & 15], gc = eacc[u2 >> 4][u2 & 15];
          part += (gs * (e * cos_big(ang)) + gc * (e * cos_big(ang + 1.5707963267948966f))) * f;
        }
        part *= 6.283185307179586f;
        if (h == 0) part += eacc[3][c];  // the raw-coordinate input column: slot 48 + c
        nrm[c] = part + __shfl_xor(part, 32, 64);
      }
      if (h == 0 && valid && a.saved.normals) {
        const float len = fmaxf(sqrtf(nrm[0] * nrm[0] + nrm[1] * nrm[1] + nrm[2] * nrm[2]), 1e-12f);
        a.saved.normals[pc * 3 + 0] = -(nrm[0] / len);
        a.saved.normals[pc * 3 + 1] = -(nrm[1] / len);
        a.saved.normals[pc * 3 + 2] = -(nrm[2] / len);
      }
    }
#ifdef F32_KEEP_LIVE
    if (!NORMALS) asm volatile("" ::"v"(mc[0]), "v"(mc[1]), "v"(mc[2]), "v"(vc[0]), "v"(vc[1]), "v"(vc[2]));
#endif
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // no LDS-DMA may outlive the workgroup's LDS allocation
}

// ------------------------------------------------------------------------------------------------ launcher
int rsn_launch_field_f32_train(long long n_tiles128, hipStream_t st, const FieldJobs& J) {
  bool normals = false, plain = false;
  for (int k = 0; k < J.n_jobs; ++k) {
    const FieldJob& a = J.j[k];
    RSN_REQUIRE(a.mode == RSN_MODE_FRUSTUM || a.mode == RSN_MODE_INF, RSN_ERR_UNSUPPORTED, "job %d: mode %d", k, a.mode);
    if (a.saved.normals) normals = true; else plain = true;
  }
  RSN_REQUIRE(!(normals && plain), RSN_ERR_UNSUPPORTED,
              "evaluations with and without analytic normals cannot share a launch (the weight ring walks one program)");
  RSN_REQUIRE(J.s.L.f_stream != 0, RSN_ERR_INVALID_ARGUMENT, "the packed weights carry no fp32 ring stream");
  const int cus = rsn_device_cus();
  const long long grid = n_tiles128 < (long long)cus ? n_tiles128 : (long long)cus;
  if (normals) hipLaunchKernelGGL(rsn_field_f32_train_kernel<true>, dim3((unsigned)grid), dim3(256), 0, st, J);
  else hipLaunchKernelGGL(rsn_field_f32_train_kernel<false>, dim3((unsigned)grid), dim3(256), 0, st, J);
  RSN_HIP(hipGetLastError());
  return RSN_OK;
}
