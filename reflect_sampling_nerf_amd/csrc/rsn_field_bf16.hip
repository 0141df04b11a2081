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

// called by launch_field (rsn_field.hip) for RSN_MMA_BF16 eval launches
int rsn_launch_field_bf16(int width, long long grid, hipStream_t st, const FieldArgs& a) {
  switch (width) {
    case 256: hipLaunchKernelGGL((rsn_field_bf16_kernel<8>), dim3((unsigned)grid), dim3(256), 0, st, a); break;
    case 128: hipLaunchKernelGGL((rsn_field_bf16_kernel<4>), dim3((unsigned)grid), dim3(256), 0, st, a); break;
    case 64: hipLaunchKernelGGL((rsn_field_bf16_kernel<2>), dim3((unsigned)grid), dim3(256), 0, st, a); break;
    default: RSN_REQUIRE(false, RSN_ERR_UNSUPPORTED, "width=%d unsupported", width);
  }
  RSN_HIP(hipGetLastError());
  return RSN_OK;
}
