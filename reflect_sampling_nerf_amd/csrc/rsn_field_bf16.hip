// rsn_field_bf16.hip -- eval field kernel for RSN_MMA_BF16 (BASELINE configs[3]: "bf16 MFMA hidden GEMMs").
//
// Same network, same lane-local activation hand-off as rsn_field.hip, but shaped for the bf16 matrix pipe, which
// is 16x faster than the fp32 one: there the MFMA-free phases (encode, epilogues) and the weight stream cost
// 15 % of the kernel, here they would be 70 %.  So:
//   * activations are parked in LDS already rounded to bf16 (the rounding the bf16 GEMM applies to its B operand
//     anyway -- results are bit-identical to rounding at read time): one ds_read_b128 per K=16 step feeds the MFMAs
//     directly, no conversion inside the K loop, and the slab is 19 KiB per wave instead of 38 KiB;
//   * two workgroups fit a CU (76 KiB LDS, <= 256 registers per wave): while one wave encodes, drains accumulators
//     or waits for weights, the other wave of its SIMD keeps the matrix pipe busy.
// Encode, head activations, SH-34 and compositing inputs stay fp32; accumulation is fp32.
//
// Weights: split 0 of the split-bf16 segments written by rsn_pack_weights ([k16][nb][3][lane][8 bf16]).
// Slab: X[kk][lane] = 8 bf16 = the lane's features kk*16 + {4h..4h+3} and kk*16 + 8 + {4h..4h+3} (K-iterations 2kk, 2kk+1
// of the fp32 kernel), so the packed K order is unchanged.
#include <stdlib.h>

#include "rsn_ring16.h"

// acc -> slab: blocks 0..NBS-1; K=16 step nb*2 + qp holds accumulator registers 8qp..8qp+7 of block nb
template <int NBO, int NBS, bool RELU>
__device__ __forceinline__ void store_h(const f32x16 (&acc)[NBO], bf16x8* xh) {
#pragma unroll
  for (int nb = 0; nb < NBS; ++nb)
#pragma unroll
    for (int qp = 0; qp < 2; ++qp) {
      uint4v w;
#pragma unroll
      for (int e = 0; e < 4; ++e) w[e] = pack2<RELU>(acc[nb][8 * qp + 2 * e], acc[nb][8 * qp + 2 * e + 1]);
      xh[(nb * 2 + qp) * 64] = __builtin_bit_cast(bf16x8, w);
    }
}

// acc[nb] += W1[nb-block] * X over n_k16 K=16 steps; weight fragments double-buffered by halves of the output blocks
template <int NBO>
__device__ __forceinline__ void gemm_h(f32x16 (&acc)[NBO], const float* __restrict__ wseg, const bf16x8* xh, int n_k16,
                                       int lane) {
  constexpr int H0 = (NBO + 1) / 2, H1 = NBO - H0;
  const bf16x8* __restrict__ wp = reinterpret_cast<const bf16x8*>(wseg) + lane;
  bf16x8 wa[H0], wb[H1 > 0 ? H1 : 1];
#pragma unroll
  for (int t = 0; t < H0; ++t) wa[t] = wp[((0 * NBO + t) * 3) * 64];
  bf16x8 bc = xh[0];
#pragma unroll 1
  for (int kk = 0; kk < n_k16; ++kk) {
    const int kn = (kk + 1 < n_k16) ? kk + 1 : kk;
#pragma unroll
    for (int t = 0; t < H1; ++t) wb[t] = wp[((kk * NBO + H0 + t) * 3) * 64];
    const bf16x8 bn = xh[kn * 64];
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int t = 0; t < H0; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wa[t], bc, acc[t], 0, 0, 0);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int t = 0; t < H0; ++t) wa[t] = wp[((kn * NBO + t) * 3) * 64];  // clamped prefetch: one control path
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int t = 0; t < H1; ++t) acc[H0 + t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wb[t], bc, acc[H0 + t], 0, 0, 0);
    __builtin_amdgcn_sched_barrier(0);
    bc = bn;
  }
}

template <int NB>
__global__ __launch_bounds__(256, 2) void rsn_field_bf16_kernel(const FieldArgs a) {
  constexpr int XK = (NB * 2 > 8) ? NB * 2 : 8;  // >= 7 (encoding) and >= 8 (mid hidden)
  constexpr int WAVE_H = (XK + RSN_SH_K16) * 64;
  constexpr int W = NB * 32;
  __shared__ bf16x8 smem[4 * WAVE_H];

  const int lane = threadIdx.x & 63;
  const int wid = threadIdx.x >> 6;
  bf16x8* X = smem + wid * WAVE_H + lane;
  bf16x8* AUX = X + XK * 64;
  __bf16* Xs = reinterpret_cast<__bf16*>(X);  // element (kk, pos) of this lane: Xs[kk * 512 + pos]

  int n_rays = a.n_rays;
  if (a.n_dev) {
    const int nd = *a.n_dev;
    n_rays = nd < n_rays ? nd : n_rays;
  }
  const long long n_points = (long long)n_rays * a.S;
  const long long n_tiles = (n_points + 127) / 128;
  const float* __restrict__ pk = a.packed;

  for (long long tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
    const long long p0 = tile * 128 + wid * 32;
    if (p0 >= n_points) continue;  // wave-uniform; waves never synchronise with each other
    // an opaque copy of the lane id per tile: per-lane weight / output addresses are then not loop-invariant, so hipcc
    // cannot hoist ~30 of them out of the persistent tile loop and spill them (1.2 GB of scratch traffic per launch)
    int ln = lane;
    asm volatile("" : "+v"(ln));
    const int m = ln & 31, h = ln >> 5;
    const long long p = p0 + m;
    const bool valid = p < n_points;
    const long long pc = valid ? p : n_points - 1;

    float mc[3] = {0.0f, 0.0f, 0.0f}, vc[3] = {0.0f, 0.0f, 0.0f}, vd[3] = {0.0f, 0.0f, 0.0f};
    bool has_cov = true, has_dir = true;
    if (a.mode == RSN_MODE_EMB) {
      has_dir = a.view_dirs != nullptr;
#pragma unroll
      for (int c = 0; c < 3; ++c) vd[c] = has_dir ? a.view_dirs[pc * 3 + c] : 0.0f;
#pragma unroll 4
      for (int kk = 0; kk < NB * 2; ++kk) {
        const float4 lo = *reinterpret_cast<const float4*>(a.emb_in + pc * W + kk * 16 + 4 * h);
        const float4 hi = *reinterpret_cast<const float4*>(a.emb_in + pc * W + kk * 16 + 8 + 4 * h);
        const float v[8] = {lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
        X[kk * 64] = pack8(v);
      }
    } else {
      // ---------------- encode (fp32, as rsn_field.hip) -----------------
      if (a.mode == RSN_MODE_FRUSTUM) {
        const long long ray = pc / a.S;
        const int s = (int)(pc - ray * a.S);
        float o[3];
#pragma unroll
        for (int c = 0; c < 3; ++c) {
          o[c] = a.origins[ray * 3 + c];
          vd[c] = a.directions[ray * 3 + c];
        }
        const float pa = a.pixel_area[ray];
        const float t0 = a.bins[ray * (a.S + 1) + s];
        const float t1 = a.bins[ray * (a.S + 1) + s + 1];
        frustum_to_contracted(o, vd, pa, t0, t1, mc, vc);
      } else if (a.mode == RSN_MODE_INF) {
        const float r2 = a.sqradius[pc];
#pragma unroll
        for (int c = 0; c < 3; ++c) {
          vd[c] = a.directions[pc * 3 + c];
          mc[c] = 2.0f * vd[c];
          vc[c] = (0.6f * r2) * (1.0f - vd[c] * vd[c]);
        }
        has_dir = false;  // SH inputs are zeroed (reflect_sampling_nerf_field.py:199)
      } else {
        has_cov = a.cov_diag != nullptr;
        has_dir = a.view_dirs != nullptr;
#pragma unroll
        for (int c = 0; c < 3; ++c) {
          mc[c] = a.means[pc * 3 + c];
          vc[c] = has_cov ? a.cov_diag[pc * 3 + c] : 0.0f;
          vd[c] = has_dir ? a.view_dirs[pc * 3 + c] : 0.0f;
        }
      }
      // slot u of this lane (K-iteration u/4, element u%4) -> K=16 step u/8, position ((u/4)&1)*4 + u%4
#pragma unroll 1
      for (int c = 0; c < 3; ++c) {
        const float x = (c == 0) ? mc[0] : (c == 1 ? mc[1] : mc[2]);
        const float v = (c == 0) ? vc[0] : (c == 1 ? vc[1] : vc[2]);
        const float sx = 6.283185307179586f * x;
#pragma unroll 2
        for (int jj = 0; jj < 8; ++jj) {
          const float f = h ? a.freqs[8 + jj] : a.freqs[jj];
          const float ang = sx * f;
          const float e = has_cov ? expf(-0.5f * (v * (f * f))) : 1.0f;
          const float fs = e * sin_big(ang);
          const float fc = e * sin_big(ang + 1.5707963267948966f);
          const int u = c * 8 + jj, u2 = u + 24;
          Xs[(u >> 3) * 512 + ((u >> 2) & 1) * 4 + (u & 3)] = (__bf16)fs;
          Xs[(u2 >> 3) * 512 + ((u2 >> 2) & 1) * 4 + (u2 & 3)] = (__bf16)fc;
        }
      }
      {  // K-iterations 12 (raw coordinates, h == 0) and 13 (zero weights): K=16 step 6
        const float r[8] = {h == 0 ? mc[0] : 0.0f, h == 0 ? mc[1] : 0.0f, h == 0 ? mc[2] : 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f};
        X[6 * 64] = pack8(r);
      }
      // this lane's encoded inputs (already bf16), re-used by the skip layer
      bf16x8 st[RSN_ENC_K16];
#pragma unroll
      for (int kk = 0; kk < RSN_ENC_K16; ++kk) st[kk] = X[kk * 64];

      // ---------------- trunk -----------------
      f32x16 acc[NB];
      init_acc<NB>(acc, pk + a.L.b[0], h);
      gemm_h<NB>(acc, pk + a.L.h_enc0, X, RSN_ENC_K16, ln);
#pragma unroll 1
      for (int l = 1; l < a.num_layers; ++l) {
        store_h<NB, NB, true>(acc, X);
        init_acc<NB>(acc, pk + a.L.b[l], h);
        gemm_h<NB>(acc, pk + a.L.h_x[l], X, NB * 2, ln);
        if (l == a.skip_layer) {
#pragma unroll
          for (int kk = 0; kk < RSN_ENC_K16; ++kk) X[kk * 64] = st[kk];
          gemm_h<NB>(acc, pk + a.L.h_enc_skip, X, RSN_ENC_K16, ln);
        }
      }
      store_h<NB, NB, true>(acc, X);  // out_activation = ReLU
    }
    if (a.embedding && valid) {  // the embedding as the downstream GEMMs see it (bf16-rounded)
#pragma unroll 4
      for (int kk = 0; kk < NB * 2; ++kk) {
        const bf16x8 f = X[kk * 64];
        *reinterpret_cast<float4*>(a.embedding + pc * W + kk * 16 + 4 * h) =
            make_float4((float)f[0], (float)f[1], (float)f[2], (float)f[3]);
        *reinterpret_cast<float4*>(a.embedding + pc * W + kk * 16 + 8 + 4 * h) =
            make_float4((float)f[4], (float)f[5], (float)f[6], (float)f[7]);
      }
    }

    // ---------------- bottleneck + heads (one GEMM, N = W + 32) -----------------
    float dcol[3], tcol[3], rho;
    {
      f32x16 acc[NB + 1];
      init_acc<NB + 1>(acc, pk + a.L.b_bh, h);
      gemm_h<NB + 1>(acc, pk + a.L.h_bh, X, NB * 2, ln);
      const float r0 = acc[NB][0], r1 = acc[NB][1], r2 = acc[NB][2], r3 = acc[NB][3];
      const float r4 = acc[NB][4], r5 = acc[NB][5], r6 = acc[NB][6];
      // h == 0: r0 raw density, r1..r3 normals, r4 roughness.   h == 1: r0..r2 diff, r4..r6 tint.
      const float rough_raw = __shfl(r4, m, 64);
      rho = (a.mode == RSN_MODE_EMB && a.rough_in) ? a.rough_in[pc] : softplus_f(rough_raw);
      dcol[0] = sigmoid_f(r0); dcol[1] = sigmoid_f(r1); dcol[2] = sigmoid_f(r2);
      tcol[0] = sigmoid_f(r4); tcol[1] = sigmoid_f(r5); tcol[2] = sigmoid_f(r6);
      if (a.mode != RSN_MODE_INF && valid) {
        if (h == 0) {
          float nrm = fmaxf(sqrtf(r1 * r1 + r2 * r2 + r3 * r3), 1e-12f);
          float nx = -(r1 / nrm), ny = -(r2 / nrm), nz = -(r3 / nrm);
          nrm = fmaxf(sqrtf(nx * nx + ny * ny + nz * nz), 1e-12f);
          nx /= nrm; ny /= nrm; nz /= nrm;
          if (a.out.sigma) a.out.sigma[pc] = softplus_f(r0 + a.density_bias);
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
      store_h<NB + 1, NB, false>(acc, X);  // bottleneck output (no activation): the x-part of mlp_mid's input
    }

    // ---------------- SH-34 of the view direction, attenuated by softplus roughness -----------------
    {
      float sh[34];
      if (has_dir) {
        sh34_attenuated(vd[0], vd[1], vd[2], rho, sh);
      } else {
#pragma unroll
        for (int i = 0; i < 34; ++i) sh[i] = 0.0f;
      }
#pragma unroll
      for (int kk = 0; kk < RSN_SH_K16; ++kk) {  // slot u = it*4 + s holds component 17h + u (u < 17)
        float v[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const int u = (2 * kk + (e >> 2)) * 4 + (e & 3);
          v[e] = (u < 17) ? (h ? sh[17 + (u < 17 ? u : 0)] : sh[u < 17 ? u : 0]) : 0.0f;
        }
        AUX[kk * 64] = pack8(v);
      }
    }

    // ---------------- mlp_mid + RGB head -----------------
    {
      f32x16 accm[4];
      init_acc<4>(accm, pk + a.L.b_mid, h);
      gemm_h<4>(accm, pk + a.L.h_mid_sh, AUX, RSN_SH_K16, ln);
      gemm_h<4>(accm, pk + a.L.h_mid_x, X, NB * 2, ln);
      store_h<4, 4, true>(accm, X);
    }
    {
      f32x16 accr[1];
      init_acc<1>(accr, pk + a.L.b_rgb, h);
      gemm_h<1>(accr, pk + a.L.h_rgb, X, 8, ln);
      if (h == 1 && valid && a.out.color) {
        const float m0 = sigmoid_f(accr[0][0]);
        const float m1 = sigmoid_f(accr[0][1]);
        const float m2 = sigmoid_f(accr[0][2]);
        if (a.mode == RSN_MODE_INF || (a.mode == RSN_MODE_EMB && !a.out.diff && !a.out.tint)) {
          a.out.color[pc * 3 + 0] = m0; a.out.color[pc * 3 + 1] = m1; a.out.color[pc * 3 + 2] = m2;
        } else {
          a.out.color[pc * 3 + 0] = dcol[0] + tcol[0] * m0;
          a.out.color[pc * 3 + 1] = dcol[1] + tcol[1] * m1;
          a.out.color[pc * 3 + 2] = dcol[2] + tcol[2] * m2;
        }
      }
    }
  }
}

// ================================================================================================
// Width 256 (the BASELINE network): the weight stream is SHARED by the workgroup through an LDS ring.
//
// The kernel above makes every wave pull its own 1.2 MB of weight fragments per 32-point tile from L1/L2: one 1 KiB
// fragment per 32-cycle MFMA per SIMD = 128 B/clk/CU, twice what a CU's L1 delivers (round 1: 33 % MFMA-busy, more
// vector loads than MFMAs).  Here
//   * the network's fragments form ONE linear stream in consumption order (rsn_pack.hip, L.r_stream), cut into
//     groups of 8 fragments (8 KiB);
//   * the four waves of a workgroup walk the network in lockstep and pull the stream through a 5-slot LDS ring by
//     LDS-DMA (global_load_lds_dwordx4, two 1 KiB pieces per wave and group), 4 groups ahead of the MFMAs; every
//     fragment is read from L2 once per WORKGROUP and tile (4x less L2 traffic) and feeds the MFMAs by
//     ds_read_b128 (128 B/clk/CU of the LDS's 256);
//   * ONE s_barrier per group: behind it group g+1 has landed for every wave and the slot of group g-1 is free for
//     the DMA of group g+4 (counted s_waitcnt vmcnt: other vector-memory operations only make it stricter);
//   * activations never touch LDS: a lane's 256 inputs of the next layer are the bf16 pairs of its own accumulators
//     (same lane-local hand-off as everywhere), kept in 64 VGPRs and indexed statically by the fully unrolled K loop;
//     the LDS holds only the ring (40 KiB), the encoded inputs for the skip layer (7 KiB per wave) and the biases;
//   * two workgroups per CU (78 KiB LDS, <= 256 VGPRs): while one encodes or drains accumulators the other's
//     MFMAs keep the matrix pipe busy.
// ================================================================================================
// acc[nb] += W-fragment(i) * X[kk] over a GEMM of NBO x KS fragments (a whole number of groups); fragment i of the
// stream sits in FIFO register i % RING_FIFO when its MFMA issues, and fragment i + RING_FIFO is read meanwhile.
template <int NW, int NBO, int KS, int XN>
__device__ __forceinline__ void gemm_ring(f32x16 (&acc)[NBO], const bf16x8 (&X)[XN], Ring& r, bf16x8 (&W)[RING_FIFO],
                                          const char* smem) {
  static_assert((NBO * KS) % RSN_RING_GROUP_FRAGS == 0 && KS <= XN, "a GEMM is a whole number of ring groups");
#pragma unroll
  for (int i = 0; i < NBO * KS; ++i) {
    if (i % RSN_RING_GROUP_FRAGS == 0) ring_sync<NW>(r);
    const int kk = i / NBO, nb = i % NBO;
    const bf16x8 wa = W[i % RING_FIFO];
    const int pos = (i % RSN_RING_GROUP_FRAGS) + RING_FIFO;
    W[i % RING_FIFO] = *reinterpret_cast<const bf16x8*>(
        smem + (pos < RSN_RING_GROUP_FRAGS ? r.rd_cur + pos * 1024 : r.rd_next + (pos - RSN_RING_GROUP_FRAGS) * 1024));
#ifdef RSN_RING_NO_MFMA
    acc[nb][i % 16] += (float)wa[0] * (float)X[kk][0];
#elif defined(RSN_RING_MFMA16)  // timing experiment: the same FLOP as two 16x16x32 MFMAs (wrong results)
    acc[nb].lo.lo = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wa, X[kk], acc[nb].lo.lo, 0, 0, 0);
    acc[nb].hi.lo = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wa, X[kk], acc[nb].hi.lo, 0, 0, 0);
#else
    acc[nb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wa, X[kk], acc[nb], 0, 0, 0);
#endif
    // keep the source order (read of fragment i + FIFO, then the MFMA of fragment i): hipcc otherwise sinks the reads
    // to one or two MFMAs ahead of their use and every other MFMA waits out the LDS latency
    __builtin_amdgcn_sched_barrier(0);
  }
}

// accumulators <- bias (LDS table; all lanes of a half-wave read one address: broadcast)
template <int NBO>
__device__ __forceinline__ void init_acc_lds(f32x16 (&acc)[NBO], const float* bias, int h) {
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

// accumulator blocks 0..NBS-1 -> the next GEMM's B operands: K=16 step nb*2 + qp = registers 8qp..8qp+7 of block nb.
// With `bias` the block's accumulators restart from the next layer's bias right after they are packed; the
// sched_barrier keeps hipcc from hoisting all 32 bias loads above the packing (128 extra live registers).
template <int NBO, int NBS, bool RELU, int XN>
__device__ __forceinline__ void acc_to_x(f32x16 (&acc)[NBO], bf16x8 (&X)[XN], const float* bias = nullptr, int h = 0) {
#pragma unroll
  for (int nb = 0; nb < NBS; ++nb) {
#pragma unroll
    for (int qp = 0; qp < 2; ++qp) {
      uint4v w;
#pragma unroll
      for (int e = 0; e < 4; ++e) w[e] = pack2<RELU>(acc[nb][8 * qp + 2 * e], acc[nb][8 * qp + 2 * e + 1]);
      X[nb * 2 + qp] = __builtin_bit_cast(bf16x8, w);
    }
    if (bias) {
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const float4 bv = *reinterpret_cast<const float4*>(bias + nb * 32 + 8 * q + 4 * h);
        acc[nb][4 * q + 0] = bv.x;
        acc[nb][4 * q + 1] = bv.y;
        acc[nb][4 * q + 2] = bv.z;
        acc[nb][4 * q + 3] = bv.w;
      }
      __builtin_amdgcn_sched_barrier(0);
    }
  }
}

#ifdef RSN_DIAG_BUILD  // the 32x32x16 form of the ring kernel: A/B reference of tools/ring_ab.sh only, never in librsn_hip.so
template <int NW>
__global__ __launch_bounds__(NW * 64, 2) void rsn_field_bf16_ring_kernel(const FieldArgs a) {
  constexpr int NB = 8, W = 256;
  constexpr int RING_BYTES = RingCfg<NW>::RING_BYTES;
  __shared__ __attribute__((aligned(1024))) char smem[RingCfg<NW>::LDS_BYTES];
  const int lane = threadIdx.x & 63;
  // wave index as a SCALAR: everything derived from it (this wave's first point of a tile, the running output
  // addresses hipcc strength-reduces out of the tile loop) then lives in SGPRs instead of 13 spilled VGPR pairs
  const int wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  char* stash = smem + RING_BYTES + wid * RING_STASH_BYTES;         // this wave's encoded inputs
  bf16x8* ST = reinterpret_cast<bf16x8*>(stash) + lane;             // [k16][lane]
  __bf16* STs = reinterpret_cast<__bf16*>(ST);                      // element (kk, pos) of this lane: STs[kk*512 + pos]
  float* bias = reinterpret_cast<float*>(smem + RING_BYTES + NW * RING_STASH_BYTES);
  const float* b_bh = bias + RING_MAX_LAYERS * 256;
  const float* b_mid = b_bh + 288;
  const float* b_rgb = b_mid + 128;

  int n_rays = a.n_rays;
  if (a.n_dev) {
    const int nd = *a.n_dev;
    n_rays = nd < n_rays ? nd : n_rays;
  }
  // 32-bit point indices (the launcher sends batches of 2^31 points or more to the per-wave-stream kernel): no 64-bit
  // division per lane, fewer loop-invariant registers
  const unsigned n_points = (unsigned)n_rays * (unsigned)a.S;
  const unsigned n_tiles = (n_points + NW * 32 - 1) / (NW * 32);
  if (blockIdx.x >= n_tiles) return;  // workgroup-uniform: no barrier is skipped by part of a workgroup
  // Every workgroup streams the SAME 1.2 MB in the same order: started together, the 32 CUs of an XCD ask their L2 for
  // the same lines at the same moment.  A start delay that grows with the workgroup's index inside its XCD
  // (blockIdx / 8: workgroups are dealt round-robin over the 8 XCDs) spreads the CUs over the stream.
  for (int i = 0; i < (int)((blockIdx.x >> 3) & 31) * a.stagger; ++i) __builtin_amdgcn_s_sleep(16);
  const float* __restrict__ pk = a.packed;

  // ---- biases -> LDS (once per workgroup)
  for (int i = threadIdx.x; i < a.num_layers * 256; i += NW * 64) bias[i] = pk[a.L.b[i >> 8] + (i & 255)];
  for (int i = threadIdx.x; i < 288; i += NW * 64) bias[RING_MAX_LAYERS * 256 + i] = pk[a.L.b_bh + i];
  if (threadIdx.x < 128) bias[RING_MAX_LAYERS * 256 + 288 + threadIdx.x] = pk[a.L.b_mid + threadIdx.x];
  if (threadIdx.x < 32) bias[RING_MAX_LAYERS * 256 + 288 + 128 + threadIdx.x] = pk[a.L.b_rgb + threadIdx.x];

  // ---- the ring: the first LEAD groups are requested, group 0 is awaited, its first fragments are read
  Ring r;
  r.src = reinterpret_cast<const char*>(pk + a.L.r_stream) + wid * (RingCfg<NW>::PPW * 1024);
  r.lane16 = (unsigned)lane * 16u;
  r.lds_dst = (unsigned)(size_t)(__attribute__((address_space(3))) char*)smem + (unsigned)wid * (RingCfg<NW>::PPW * 1024u);
  r.n_groups = a.L.r_groups;
  r.issue_grp = 0;
  r.issue_slot = 0;
  r.rd_base = (unsigned)lane * 16u;
  r.next_slot = 0;
  r.rd_next = r.rd_base;
  r.rd_cur = r.rd_base;
  __syncthreads();  // nothing in flight yet: a plain barrier (also publishes the bias table)
#pragma unroll
  for (int g = 0; g < RingCfg<NW>::LEAD; ++g) ring_issue<NW>(r);
  asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" ::"n"(RingCfg<NW>::PPW * (RingCfg<NW>::LEAD - 1)) : "memory");
#ifdef RSN_RING_SETPRIO
  if (wid >= NW / 2) __builtin_amdgcn_s_setprio(1);  // the younger half of the workgroup loses every arbitration otherwise
#endif
  bf16x8 Wf[RING_FIFO];
#pragma unroll
  for (int j = 0; j < RING_FIFO; ++j) Wf[j] = *reinterpret_cast<const bf16x8*>(smem + r.rd_next + j * 1024);

  for (unsigned tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
    const unsigned p0 = tile * (NW * 32) + wid * 32;
    // every wave walks every tile (the barriers and the DMA shares are per wave); a wave past the end recomputes the
    // last point and stores nothing
    int ln = lane;
    asm volatile("" : "+v"(ln));  // opaque per-tile lane id: per-lane addresses are not hoisted out of the tile loop
    const int m = ln & 31, h = ln >> 5;
    const unsigned p = p0 + m;
    const bool valid = p < n_points;
    const size_t pc = valid ? p : n_points - 1;

    float mc[3] = {0.0f, 0.0f, 0.0f}, vc[3] = {0.0f, 0.0f, 0.0f}, vd[3] = {0.0f, 0.0f, 0.0f};
    bool has_cov = true, has_dir = true;
    // ---------------- encode (fp32, as rsn_field.hip) into this wave's stash -----------------
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
      const float pa = a.pixel_area[ray];
      const float t0 = a.bins[ray * (a.S + 1) + s];
      const float t1 = a.bins[ray * (a.S + 1) + s + 1];
      frustum_to_contracted(o, vd, pa, t0, t1, mc, vc);
    } else if (a.mode == RSN_MODE_INF) {
      const float r2 = a.sqradius[pc];
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        vd[c] = a.directions[pc * 3 + c];
        mc[c] = 2.0f * vd[c];
        vc[c] = (0.6f * r2) * (1.0f - vd[c] * vd[c]);
      }
      has_dir = false;  // SH inputs are zeroed (reflect_sampling_nerf_field.py:199)
    } else {
      has_cov = a.cov_diag != nullptr;
      has_dir = a.view_dirs != nullptr;
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        mc[c] = a.means[pc * 3 + c];
        vc[c] = has_cov ? a.cov_diag[pc * 3 + c] : 0.0f;
        vd[c] = has_dir ? a.view_dirs[pc * 3 + c] : 0.0f;
      }
    }
#ifdef RSN_RING_NO_ENCODE
    for (int c = 0; c < 0; ++c) {
#else
#pragma unroll 1
    for (int c = 0; c < 3; ++c) {
#endif
      const float x = (c == 0) ? mc[0] : (c == 1 ? mc[1] : mc[2]);
      const float v = (c == 0) ? vc[0] : (c == 1 ? vc[1] : vc[2]);
      const float sx = 6.283185307179586f * x;
#pragma unroll 2
      for (int jj = 0; jj < 8; ++jj) {
        const float f = h ? a.freqs[8 + jj] : a.freqs[jj];
        const float ang = sx * f;
        // exp by v_exp_f32 (2^x): ~1e-6 relative on a feature that is rounded to bf16 (2^-9) next
        const float e = has_cov ? __builtin_amdgcn_exp2f((-0.5f * (v * (f * f))) * 1.4426950408889634f) : 1.0f;
#ifndef RSN_RING_EXACT_MATH
        const float fs = e * sincos_bf16(ang, 0);
        const float fc = e * sincos_bf16(ang + 1.5707963267948966f, 0);
#else
        const float fs = e * sin_big(ang);
        const float fc = e * sin_big(ang + 1.5707963267948966f);
#endif
        const int u = c * 8 + jj, u2 = u + 24;
        STs[(u >> 3) * 512 + ((u >> 2) & 1) * 4 + (u & 3)] = (__bf16)fs;
        STs[(u2 >> 3) * 512 + ((u2 >> 2) & 1) * 4 + (u2 & 3)] = (__bf16)fc;
      }
    }
    {
      const float rw[8] = {h == 0 ? mc[0] : 0.0f, h == 0 ? mc[1] : 0.0f, h == 0 ? mc[2] : 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f};
      ST[6 * 64] = pack8(rw);
    }

    bf16x8 X[16];
    // ---------------- trunk -----------------
    {
      f32x16 acc[NB];
      init_acc_lds<NB>(acc, bias, h);
#pragma unroll
      for (int kk = 0; kk < RSN_ENC_K16; ++kk) X[kk] = ST[kk * 64];
#pragma unroll
      for (int kk = RSN_ENC_K16; kk < RING_ENC_KS; ++kk) X[kk] = bf16x8{};  // padding K-steps (zero weights): finite operands
      gemm_ring<NW, NB, RING_ENC_KS, 16>(acc, X, r, Wf, smem);
#pragma unroll 1
      for (int l = 1; l < a.num_layers; ++l) {
        acc_to_x<NB, NB, true, 16>(acc, X, bias + l * 256, h);
        gemm_ring<NW, NB, 16, 16>(acc, X, r, Wf, smem);
        if (l == a.skip_layer) {
          bf16x8 XE[RING_ENC_KS];
#pragma unroll
          for (int kk = 0; kk < RSN_ENC_K16; ++kk) XE[kk] = ST[kk * 64];
#pragma unroll
          for (int kk = RSN_ENC_K16; kk < RING_ENC_KS; ++kk) XE[kk] = bf16x8{};
          gemm_ring<NW, NB, RING_ENC_KS, RING_ENC_KS>(acc, XE, r, Wf, smem);
        }
      }
      acc_to_x<NB, NB, true, 16>(acc, X);  // out_activation = ReLU: the embedding
    }
    if (a.embedding && valid) {  // the embedding as the downstream GEMMs see it (bf16-rounded)
#pragma unroll
      for (int kk = 0; kk < NB * 2; ++kk) {
        const bf16x8 f = X[kk];
        *reinterpret_cast<float4*>(a.embedding + pc * W + kk * 16 + 4 * h) =
            make_float4((float)f[0], (float)f[1], (float)f[2], (float)f[3]);
        *reinterpret_cast<float4*>(a.embedding + pc * W + kk * 16 + 8 + 4 * h) =
            make_float4((float)f[4], (float)f[5], (float)f[6], (float)f[7]);
      }
    }

    // ---------------- heads (one 32-row block), then the bottleneck -----------------
    float dcol[3], tcol[3], rho;
    {
      f32x16 acch[1];
      init_acc_lds<1>(acch, b_bh + 256, h);
      gemm_ring<NW, 1, 16, 16>(acch, X, r, Wf, smem);
      const float r0 = acch[0][0], r1 = acch[0][1], r2 = acch[0][2], r3 = acch[0][3];
      const float r4 = acch[0][4], r5 = acch[0][5], r6 = acch[0][6];
      // h == 0: r0 raw density, r1..r3 normals, r4 roughness.   h == 1: r0..r2 diff, r4..r6 tint.
      const float rough_raw = __shfl(r4, m, 64);
#ifndef RSN_RING_EXACT_MATH
      rho = fast_softplus(rough_raw);
      dcol[0] = fast_sigmoid(r0); dcol[1] = fast_sigmoid(r1); dcol[2] = fast_sigmoid(r2);
      tcol[0] = fast_sigmoid(r4); tcol[1] = fast_sigmoid(r5); tcol[2] = fast_sigmoid(r6);
#else
      rho = softplus_f(rough_raw);
      dcol[0] = sigmoid_f(r0); dcol[1] = sigmoid_f(r1); dcol[2] = sigmoid_f(r2);
      tcol[0] = sigmoid_f(r4); tcol[1] = sigmoid_f(r5); tcol[2] = sigmoid_f(r6);
#endif
      if (a.mode != RSN_MODE_INF && valid) {
        if (h == 0) {
          float nrm = fmaxf(sqrtf(r1 * r1 + r2 * r2 + r3 * r3), 1e-12f);
          float nx = -(r1 / nrm), ny = -(r2 / nrm), nz = -(r3 / nrm);
          nrm = fmaxf(sqrtf(nx * nx + ny * ny + nz * nz), 1e-12f);
          nx /= nrm; ny /= nrm; nz /= nrm;
          if (a.out.sigma) a.out.sigma[pc] = softplus_f(r0 + a.density_bias);
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
    }
    // ---------------- SH-34 of the view direction, attenuated by softplus roughness -----------------
    // computed HERE, while only the embedding (64 VGPRs) is live, and parked in this wave's stash (free since the skip
    // layer): it costs the bottleneck GEMM below no registers
    {
      float sh[34];
      if (has_dir) {
        sh34_attenuated(vd[0], vd[1], vd[2], rho, sh);
      } else {
#pragma unroll
        for (int i = 0; i < 34; ++i) sh[i] = 0.0f;
      }
#pragma unroll
      for (int kk = 0; kk < 4; ++kk) {  // slot u = it*4 + s holds component 17h + u (u < 17); K=16 step 3 is padding
        float v[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const int u = (2 * kk + (e >> 2)) * 4 + (e & 3);
          v[e] = (u < 17) ? (h ? sh[17 + (u < 17 ? u : 0)] : sh[u < 17 ? u : 0]) : 0.0f;
        }
        ST[kk * 64] = pack8(v);
      }
    }
    {
      f32x16 acc[NB];
      init_acc_lds<NB>(acc, b_bh, h);
      gemm_ring<NW, NB, 16, 16>(acc, X, r, Wf, smem);
      acc_to_x<NB, NB, false, 16>(acc, X);  // bottleneck output (no activation): the x-part of mlp_mid's input
    }

    // ---------------- mlp_mid + RGB head -----------------
    {
      f32x16 accm[4];
      init_acc_lds<4>(accm, b_mid, h);
      bf16x8 XS[4];
#pragma unroll
      for (int kk = 0; kk < 4; ++kk) XS[kk] = ST[kk * 64];
      gemm_ring<NW, 4, 4, 4>(accm, XS, r, Wf, smem);
      gemm_ring<NW, 4, 16, 16>(accm, X, r, Wf, smem);
      acc_to_x<4, 4, true, 16>(accm, X);  // hidden (128): K=16 steps 0..7
#pragma unroll
      for (int kk = 8; kk < RING_RGB_KS; ++kk) X[kk] = bf16x8{};  // padding K-steps of the RGB head
    }
    {
      f32x16 accr[1];
      init_acc_lds<1>(accr, b_rgb, h);
      gemm_ring<NW, 1, RING_RGB_KS, 16>(accr, X, r, Wf, smem);
      if (h == 1 && valid && a.out.color) {
        const float m0 = sigmoid_f(accr[0][0]);
        const float m1 = sigmoid_f(accr[0][1]);
        const float m2 = sigmoid_f(accr[0][2]);
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
  // no LDS-DMA may outlive the workgroup's LDS allocation
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}
#endif  // RSN_DIAG_BUILD

// ================================================================================================
// The same ring on v_mfma_f32_16x16x32_bf16.  Same FLOP per cycle as 32x32x16, but the chip holds a higher clock on
// this shape under load (MI355X_MICROARCH.md, DVFS item 7; measured here by swapping the instruction alone: -6.5 %).
// Lane (m = lane & 15, g = lane >> 4): a wave's 32-point tile is two 16-point halves (p = 0, 1: point p0 + 16 p + m),
// every weight fragment (16 rows x 32 K, 1 KiB) feeds one MFMA per half.  D[row 4g + r][col m]: after a GEMM the lane
// holds packed rows 16 b + 4 g + r (r = 0..3) of its two points in acc[b][p]; the packed stream permutes the output rows
// (rsn_pack.hip, rows_perm16) so that the 8 values of blocks 2kk and 2kk+1 are the contiguous features 32 kk + 8 g .. + 7:
// the next K-step kk takes exactly those from the lane (natural K order) -- activations again never cross lanes.  Encode: lane group g owns frequencies 4g..4g+3 of both points;
// SH: components 9g..9g+8; heads: ONE 16-row block (g = 0: density, normals; 1: diff; 2: roughness; 3: tint).
// ================================================================================================
__global__ __launch_bounds__(512, 2) void rsn_field_bf16_ring16_kernel(const FieldArgs a) {
  constexpr int W = 256;
  constexpr int RING_BYTES = RingCfg<8>::RING_BYTES;
  __shared__ __attribute__((aligned(1024))) char smem[R16_LDS_BYTES];
  const int lane = threadIdx.x & 63;
  const int wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  char* stash = smem + RING_BYTES + wid * R16_STASH_BYTES;
  bf16x8* ST = reinterpret_cast<bf16x8*>(stash) + lane;   // fragment (kk, p) of this lane: ST[(kk * 2 + p) * 64]
  __bf16* STs = reinterpret_cast<__bf16*>(ST);            // element el of it: STs[(kk * 2 + p) * 512 + el]
  float* bias = reinterpret_cast<float*>(smem + RING_BYTES + 8 * R16_STASH_BYTES);
  const float* b_bh = bias + RING_MAX_LAYERS * 256;
  const float* b_mid = b_bh + 288;
  const float* b_rgb = b_mid + 128;

  int n_rays = a.n_rays;
  if (a.n_dev) {
    const int nd = *a.n_dev;
    n_rays = nd < n_rays ? nd : n_rays;
  }
  const unsigned n_points = (unsigned)n_rays * (unsigned)a.S;
  const unsigned n_tiles = (n_points + 255) / 256;
  if (blockIdx.x >= n_tiles) return;  // workgroup-uniform
  const float* __restrict__ pk = a.packed;

  // ---- biases -> LDS, in the PACKED row order of the 16x32 stream (rsn_pack.hip, rows_perm16: packed row 16 b + 4 g + r is
  // feature 32 (b / 2) + 8 g + 4 (b % 2) + r).  Heads: the 32-entry table of the 32x32 layout keeps rows 0..15 = the 16-row
  // heads block here (not permuted, like the RGB rows).
  for (int i = threadIdx.x; i < a.num_layers * 256; i += 512) bias[i] = pk[a.L.b[i >> 8] + r16_feature(i & 255)];
  for (int i = threadIdx.x; i < 288; i += 512) bias[RING_MAX_LAYERS * 256 + i] = pk[a.L.b_bh + (i < 256 ? r16_feature(i) : i)];
  if (threadIdx.x < 128) bias[RING_MAX_LAYERS * 256 + 288 + threadIdx.x] = pk[a.L.b_mid + r16_feature(threadIdx.x)];
  if (threadIdx.x < 32) bias[RING_MAX_LAYERS * 256 + 288 + 128 + threadIdx.x] = pk[a.L.b_rgb + threadIdx.x];

  Ring r;
  r.src = reinterpret_cast<const char*>(pk + a.L.q_stream) + wid * (RingCfg<8>::PPW * 1024);
  r.lane16 = (unsigned)lane * 16u;
  r.lds_dst = (unsigned)(size_t)(__attribute__((address_space(3))) char*)smem + (unsigned)wid * (RingCfg<8>::PPW * 1024u);
  r.n_groups = a.L.q_groups;
  r.issue_grp = 0;
  r.issue_slot = 0;
  r.rd_base = (unsigned)lane * 16u;
  r.next_slot = 0;
  r.rd_next = r.rd_base;
  r.rd_cur = r.rd_base;
  __syncthreads();
#pragma unroll
  for (int gq = 0; gq < RingCfg<8>::LEAD; ++gq) ring_issue<8>(r);
  asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" ::"n"(RingCfg<8>::PPW * (RingCfg<8>::LEAD - 1)) : "memory");
  bf16x8 Wf[RING_FIFO];
#pragma unroll
  for (int j = 0; j < RING_FIFO; ++j) Wf[j] = *reinterpret_cast<const bf16x8*>(smem + r.rd_next + j * 1024);

  for (unsigned tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
    const unsigned p0 = tile * 256 + wid * 32;
    int ln = lane;
    asm volatile("" : "+v"(ln));  // opaque per-tile lane id (see the kernels above)
    const int m = ln & 15, g = ln >> 4;
    bool valid[2];
    size_t pc[2];
    float vd[2][3];
    bool has_dir = true;

    // ---------------- encode both points of this lane (fp32) into the stash -----------------
    // the contracted Gaussian of a point is evaluated ONCE per point-quad: lane group g takes point (g & 1), the
    // other three lanes of the point get it by shuffle
    float mcA[2][3], vcA[2][3];
    bool has_cov = true;
    {
      const int po = g & 1;
      const unsigned pt = p0 + 16 * po + m;
      const size_t pcc = pt < n_points ? pt : n_points - 1;
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
      } else if (a.mode == RSN_MODE_INF) {
        const float r2 = a.sqradius[pcc];
#pragma unroll
        for (int c = 0; c < 3; ++c) {
          dd[c] = a.directions[pcc * 3 + c];
          mc[c] = 2.0f * dd[c];
          vc[c] = (0.6f * r2) * (1.0f - dd[c] * dd[c]);
        }
        has_dir = false;  // SH inputs are zeroed (reflect_sampling_nerf_field.py:199)
      } else {
        has_cov = a.cov_diag != nullptr;
        has_dir = a.view_dirs != nullptr;
#pragma unroll
        for (int c = 0; c < 3; ++c) {
          mc[c] = a.means[pcc * 3 + c];
          vc[c] = has_cov ? a.cov_diag[pcc * 3 + c] : 0.0f;
          dd[c] = has_dir ? a.view_dirs[pcc * 3 + c] : 0.0f;
        }
      }
#pragma unroll
      for (int p = 0; p < 2; ++p) {
        const unsigned ptp = p0 + 16 * p + m;
        valid[p] = ptp < n_points;
        pc[p] = valid[p] ? ptp : n_points - 1;
#pragma unroll
        for (int c = 0; c < 3; ++c) {
          mcA[p][c] = __shfl(mc[c], 16 * p + m, 64);
          vcA[p][c] = __shfl(vc[c], 16 * p + m, 64);
          vd[p][c] = __shfl(dd[c], 16 * p + m, 64);
        }
      }
    }
    // slot u = 8 kk + el of this lane: u < 12 sin(c = u / 4, frequency 4g + u % 4), 12 <= u < 24 the cos, 24..26 raw.
    // The 24 features of a point are collected in registers and leave as three 16-byte stores (round 2 wrote 48 single
    // bf16 per point: ds_write_b16 at a 16-byte lane stride is a 4-way bank conflict).
#pragma unroll
    for (int p = 0; p < 2; ++p) {
      float feat[24];
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        const float x = mcA[p][c], v = vcA[p][c];
        const float sx = 6.283185307179586f * x;
#pragma unroll
        for (int t = 0; t < 4; ++t) {
          const float f = a.freqs[4 * g + t];
          const float ang = sx * f;
          const float e = has_cov ? __builtin_amdgcn_exp2f((-0.5f * (v * (f * f))) * 1.4426950408889634f) : 1.0f;
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

    bf16x8 X[8][2];
    // ---------------- trunk -----------------
    {
      f32x4 acc[16][2];
      init_acc16<16>(acc, bias, g);
#pragma unroll
      for (int kk = 0; kk < 4; ++kk) { X[kk][0] = ST[(kk * 2) * 64]; X[kk][1] = ST[(kk * 2 + 1) * 64]; }
      gemm_ring16<16, 4, 8>(acc, X, r, Wf, smem);
#pragma unroll 1
      for (int l = 1; l < a.num_layers; ++l) {
        acc_to_x16<16, 8, true, 8>(acc, X, bias + l * 256, g);
        gemm_ring16<16, 8, 8>(acc, X, r, Wf, smem);
        if (l == a.skip_layer) {
          bf16x8 XE[4][2];
#pragma unroll
          for (int kk = 0; kk < 4; ++kk) { XE[kk][0] = ST[(kk * 2) * 64]; XE[kk][1] = ST[(kk * 2 + 1) * 64]; }
          gemm_ring16<16, 4, 4>(acc, XE, r, Wf, smem);
        }
      }
      acc_to_x16<16, 8, true, 8>(acc, X);  // out_activation = ReLU: the embedding
    }
    if (a.embedding) {  // the embedding as the downstream GEMMs see it (bf16-rounded); X[kk] = features 32 kk + 8 g .. + 7
#pragma unroll
      for (int p = 0; p < 2; ++p)
        if (valid[p]) {
#pragma unroll
          for (int kk = 0; kk < 8; ++kk) {
            const bf16x8 f = X[kk][p];
            *reinterpret_cast<float4*>(a.embedding + pc[p] * W + 32 * kk + 8 * g) =
                make_float4((float)f[0], (float)f[1], (float)f[2], (float)f[3]);
            *reinterpret_cast<float4*>(a.embedding + pc[p] * W + 32 * kk + 8 * g + 4) =
                make_float4((float)f[4], (float)f[5], (float)f[6], (float)f[7]);
          }
        }
    }

    // ---------------- heads: one 16-row block (+ a zero block: whole-group padding) -----------------
    float dcol[2][3];  // g == 1 lanes: sigmoid(diff)
    {
      f32x4 acch[2][2];
      init_acc16<2>(acch, b_bh + 256, g);
      gemm_ring16<2, 8, 8>(acch, X, r, Wf, smem);
#pragma unroll
      for (int p = 0; p < 2; ++p) {
        const float r0 = acch[0][p][0], r1 = acch[0][p][1], r2 = acch[0][p][2], r3 = acch[0][p][3];
        // g == 0: r0 raw density, r1..r3 normals;  g == 1: r0..r2 diff;  g == 2: r0 roughness;  g == 3: r0..r2 tint
        const float rough_raw = __shfl(r0, 32 + m, 64);
        const float rho = fast_softplus(rough_raw);
        // SH-34 of the view direction, attenuated by softplus roughness: ONE lane of the point's four (group g == p)
        // evaluates the 34 terms and writes the slots of all four groups (slot u < 9 of group g' = component 9 g' + u)
        if (g == p) {
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
            v1[0] = sh[9 * gq + 8];  // (component 35 = 0 for the last group)
            ST[(0 * 2 + p) * 64 + (gq - g) * 16] = pack8(v0);
            ST[(1 * 2 + p) * 64 + (gq - g) * 16] = pack8(v1);
          }
        }
        dcol[p][0] = fast_sigmoid(r0); dcol[p][1] = fast_sigmoid(r1); dcol[p][2] = fast_sigmoid(r2);
        if (a.mode != RSN_MODE_INF && valid[p]) {
          const size_t q = pc[p];
          if (g == 0) {
            float nrm = fmaxf(sqrtf(r1 * r1 + r2 * r2 + r3 * r3), 1e-12f);
            float nx = -(r1 / nrm), ny = -(r2 / nrm), nz = -(r3 / nrm);
            nrm = fmaxf(sqrtf(nx * nx + ny * ny + nz * nz), 1e-12f);
            nx /= nrm; ny /= nrm; nz /= nrm;
            if (a.out.sigma) a.out.sigma[q] = fast_softplus(r0 + a.density_bias);
            if (a.out.raw_density) a.out.raw_density[q] = r0;
            if (a.out.pred_normals) {
              a.out.pred_normals[q * 3 + 0] = nx;
              a.out.pred_normals[q * 3 + 1] = ny;
              a.out.pred_normals[q * 3 + 2] = nz;
            }
            if (a.out.n_dot_d) a.out.n_dot_d[q] = vd[p][0] * nx + vd[p][1] * ny + vd[p][2] * nz;
          } else if (g == 1) {
            if (a.out.diff) {
              a.out.diff[q * 3 + 0] = dcol[p][0]; a.out.diff[q * 3 + 1] = dcol[p][1]; a.out.diff[q * 3 + 2] = dcol[p][2];
            }
          } else if (g == 2) {
            if (a.out.roughness) a.out.roughness[q] = fast_sigmoid(r0);
            if (a.out.raw_roughness) a.out.raw_roughness[q] = r0;
          } else {
            if (a.out.tint) {
              a.out.tint[q * 3 + 0] = dcol[p][0]; a.out.tint[q * 3 + 1] = dcol[p][1]; a.out.tint[q * 3 + 2] = dcol[p][2];
            }
          }
        }
      }
    }
    {
      f32x4 acc[16][2];
      init_acc16<16>(acc, b_bh, g);
      gemm_ring16<16, 8, 8>(acc, X, r, Wf, smem);
      acc_to_x16<16, 8, false, 8>(acc, X);  // bottleneck output (no activation): the x-part of mlp_mid's input
    }

    // ---------------- mlp_mid + RGB head -----------------
    {
      f32x4 accm[8][2];
      init_acc16<8>(accm, b_mid, g);
      bf16x8 XS[2][2];
#pragma unroll
      for (int kk = 0; kk < 2; ++kk) { XS[kk][0] = ST[(kk * 2) * 64]; XS[kk][1] = ST[(kk * 2 + 1) * 64]; }
      gemm_ring16<8, 2, 2>(accm, XS, r, Wf, smem);
      gemm_ring16<8, 8, 8>(accm, X, r, Wf, smem);
      acc_to_x16<8, 4, true, 8>(accm, X);  // hidden (128): K-steps 0..3
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
      gemm_ring16<4, 4, 8>(accr, X, r, Wf, smem);
#pragma unroll
      for (int p = 0; p < 2; ++p) {
        // mid RGB sits on the g == 1 lanes (rows 4..6), like diff; tint comes over from the g == 3 lane of the point
        const float m0 = fast_sigmoid(accr[0][p][0]), m1 = fast_sigmoid(accr[0][p][1]), m2 = fast_sigmoid(accr[0][p][2]);
        const float t0 = __shfl(dcol[p][0], 48 + m, 64), t1 = __shfl(dcol[p][1], 48 + m, 64), t2 = __shfl(dcol[p][2], 48 + m, 64);
        if (g == 1 && valid[p] && a.out.color) {
          const size_t q = pc[p];
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
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // no LDS-DMA may outlive the workgroup's LDS allocation
}

// called by launch_field (rsn_field.hip) for RSN_MMA_BF16 eval launches.  The product library takes NO run-time switches from
// the environment; the A/B switches of tools/ (per-wave stream, start stagger, the 32x32x16 ring) exist in diagnostic
// builds only (-DRSN_DIAG_BUILD, tools/_variant.py).
int rsn_launch_field_bf16(int width, long long grid, hipStream_t st, const FieldArgs& a) {
#ifdef RSN_DIAG_BUILD
  static const bool per_wave_stream = getenv("RSN_BF16_PER_WAVE_STREAM") != nullptr;
  static const int stagger = getenv("RSN_RING_STAGGER") ? atoi(getenv("RSN_RING_STAGGER")) : 0;
  static const bool shape32 = getenv("RSN_RING_SHAPE32") != nullptr;
#else
  constexpr bool per_wave_stream = false;
  constexpr int stagger = 0;
#endif
  switch (width) {
    case 256:
      // full network evaluations run on the shared LDS weight ring; the granular heads-only mode (a caller-supplied
      // embedding skips the trunk, i.e. most of the stream) keeps the per-wave stream
      if (a.mode != RSN_MODE_EMB && a.L.q_stream != 0 && a.num_layers <= RING_MAX_LAYERS && !per_wave_stream &&
          (long long)a.n_rays * a.S < (1LL << 31)) {
        FieldArgs b = a;
        b.stagger = stagger;  // (diagnostic builds) start skew between the workgroups of an XCD (tools/ring_sweep.sh)
        const long long n_points = (long long)a.n_rays * a.S;
        // one 8-wave workgroup per CU, 256-point tiles
        const long long t8 = (n_points + 255) / 256;
        const long long g8 = t8 < (grid + 1) / 2 ? t8 : (grid + 1) / 2;
#ifdef RSN_DIAG_BUILD
        if (shape32 && a.L.r_stream != 0) {
          hipLaunchKernelGGL(rsn_field_bf16_ring_kernel<8>, dim3((unsigned)g8), dim3(512), 0, st, b);
          break;
        }
#endif
        hipLaunchKernelGGL(rsn_field_bf16_ring16_kernel, dim3((unsigned)g8), dim3(512), 0, st, b);
      } else {
        hipLaunchKernelGGL((rsn_field_bf16_kernel<8>), dim3((unsigned)grid), dim3(256), 0, st, a);
      }
      break;
    case 128: hipLaunchKernelGGL((rsn_field_bf16_kernel<4>), dim3((unsigned)grid), dim3(256), 0, st, a); break;
    case 64: hipLaunchKernelGGL((rsn_field_bf16_kernel<2>), dim3((unsigned)grid), dim3(256), 0, st, a); break;
    default: RSN_REQUIRE(false, RSN_ERR_UNSUPPORTED, "width=%d unsupported", width);
  }
  RSN_HIP(hipGetLastError());
  return RSN_OK;
}
