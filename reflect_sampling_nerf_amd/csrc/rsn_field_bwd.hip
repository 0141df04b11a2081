// rsn_field_bwd.hip -- backward sweep of the field for one sampling level (training).
//
// Mirror image of rsn_field.hip: one wavefront owns 32 sample points and walks the network backwards with the
// TRANSPOSED packed weights (rsn_pack.hip, wT_* segments).  dX[k][m] = sum_n W[n][k] * dY[n][m] has the same
// "A = weights, B = points" shape as the forward GEMM, so the lane-local trick carries over unchanged: the
// accumulator registers a lane holds after one W^T GEMM are (after the ReLU mask) the B operand it needs for the
// next one.  ReLU masks come from the post-ReLU activations the training forward saved (x > 0).
//
// What leaves the kernel are the PRE-ACTIVATION gradients of every linear layer, row-major [N, out_features]:
// the weight gradients dW = dY^T X (contraction over all N samples) and the bias gradients are taken by
// rsn_weight_grad (rsn_wgrad.hip).
//
// Autograd semantics restated (reference: reflect_sampling_nerf_model.py:142-344, field.py:122-207):
//   colour = diff + tint * mid          -> d diff = g, d tint = g*mid, d mid = g*tint
//   SH inputs carry no gradient (components.py:52, roughness.detach() at model.py:174)
//   pred_normals = normalize(-normalize(head))      n_dot_d = sum(dir * pred_normals)
//   sigma = softplus(raw + density_bias)            roughness (rendered) = sum w * sigmoid(raw)
//   reflect levels: the Gaussian covariance depends on pixel_area = pi * sqradius (a function of the NON-detached
//   rendered roughness, model.py:225-227,272,286), so the gradient w.r.t. the encoded inputs' variance is carried
//   back to pixel_area (need_input_grad).
#include "rsn_mfma.h"

#include "rsn_field_common.h"
#include "rsn_field_bwd_common.h"

template <int NB, int MODE>
__global__ __launch_bounds__(256) void rsn_field_bwd_kernel(const BwdJobs J) {
  constexpr int XITS = (NB * 4 > 16) ? NB * 4 : 16;
  constexpr int WAVE_F4 = (XITS + RSN_AUX_ITS) * 64;
  constexpr int W = NB * 32;
  constexpr bool SBF = MODE == 3;  // reduced-precision training: wide layer gradients (dy, d_bott, da_mid) stored as bf16
  __shared__ float4 smem[4 * WAVE_F4];

  const int lane = threadIdx.x & 63;
  const int wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);  // a SCALAR (tile bases, buffer descriptors)
  float4* X = smem + wid * WAVE_F4 + lane;  // its XITS.. spill into the SH region (contiguous): NB*4+4 <= XITS+5
  float* Xf = reinterpret_cast<float*>(X);

  const BwdShared& P = J.s;
  // the launch's tile space: job k owns tiles [tb_k, tb_k+1) of 128 points (its ray count may live on the device)
  long long np0 = 0, np1 = 0, np2 = 0, tb1 = 0, tb2 = 0, n_tiles = 0;
#pragma unroll
  for (int k = 0; k < RSN_MAX_JOBS; ++k) {
    if (k < J.n_jobs) {
      int nr = J.j[k].n_rays;
      if (J.j[k].n_dev) {
        const int nd = *J.j[k].n_dev;
        nr = nd < nr ? nd : nr;
      }
      const long long np = (long long)nr * J.j[k].S;
      if (k == 0) np0 = np; else if (k == 1) np1 = np; else np2 = np;
      n_tiles += (np + 127) / 128;
    }
    if (k == 0) tb1 = n_tiles; else if (k == 1) tb2 = n_tiles;
  }
  const float* __restrict__ pk = P.packed;
  const float4 zero4 = make_float4(0.0f, 0.0f, 0.0f, 0.0f);

  for (long long gtile = blockIdx.x; gtile < n_tiles; gtile += gridDim.x) {
    const int jk = (gtile >= tb1 ? 1 : 0) + (gtile >= tb2 ? 1 : 0);  // workgroup-uniform
    const BwdJob& a = J.j[jk];
    const long long n_points = jk == 0 ? np0 : (jk == 1 ? np1 : np2);
    const long long tile = gtile - (jk == 0 ? 0 : (jk == 1 ? tb1 : tb2));
    const long long p0 = tile * 128 + wid * 32;
    if (p0 >= n_points) continue;
    // opaque per-tile copy of the lane id: keeps hipcc from hoisting (and spilling) per-lane addresses out of the loop
    int ln = lane;
    asm volatile("" : "+v"(ln));
    const int m = ln & 31, h = ln >> 5;
    const long long p = p0 + m;
    const bool valid = p < n_points;
    const long long pc = valid ? p : n_points - 1;
    const int rows = (int)(n_points - p0 < 32 ? n_points - p0 : 32);  // valid rows of this wave's tile (wave-uniform)
    // Layer-gradient rows kept for the weight gradients.  Exact fp32: a row leaves from the K loop of the GEMM that READS it
    // (one buffer store per K-iteration, gemm_run2's `save`); the split / plain bf16 loops store from the epilogue that
    // produces it.  History: with the two-buffer K loop this arrangement ran the sweeps 12.1 -> 12.5 ms per step (vmcnt
    // counts stores in order with the loads, so a fragment wait also waited out the store in front of it); with the
    // fragments two iterations ahead (gemm_run2) it is 11.78 -> 11.55 ms.
    constexpr bool LOOPST = MODE == 0;
    auto rb_epi = [&](float* base, long long elem, int row_elems) {
      return rowbuf<SBF>(LOOPST ? nullptr : base, elem, rows, row_elems, m, h);
    };
    auto rb_loop = [&](float* base, long long elem, int row_elems) {
      return rowbuf<false>(LOOPST ? base : nullptr, elem, rows, row_elems, m, h);
    };
    const float live = valid ? 1.0f : 0.0f;  // padded lanes contribute zero gradients

    auto bits_at = [&](int l) -> const unsigned* {  // rsn_field_saved.relu_bits: [L+1][N][2][NB/2] words
      return a.saved.relu_bits + ((((long long)l * (a.act_stride / W)) + pc) * 2 + h) * (NB / 2 > 2 ? NB / 2 : 2);
    };
    // ---------------- per-sample epilogue gradients -----------------
    float gcol[3] = {0.0f, 0.0f, 0.0f};
    if (a.gin.color) {
#pragma unroll
      for (int c = 0; c < 3; ++c) gcol[c] = a.gin.color[pc * 3 + c] * live;
    }
    const float4 hd = *reinterpret_cast<const float4*>(a.saved.heads + pc * 8);       // n_raw(3), rough_raw
    const float4 md = *reinterpret_cast<const float4*>(a.saved.heads + pc * 8 + 4);   // mid RGB (3)
    float mid[3] = {md.x, md.y, md.z};
    float dif[3] = {0.0f, 0.0f, 0.0f}, tin[3] = {1.0f, 1.0f, 1.0f};
    if (a.mode != RSN_MODE_INF) {
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        dif[c] = a.fwd.diff[pc * 3 + c];
        tin[c] = a.fwd.tint[pc * 3 + c];
      }
    }
    // RGB head: colour = diff + tint * mid (INF: colour = mid); mid = sigmoid(z)
    float dz_rgb[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) dz_rgb[c] = gcol[c] * tin[c] * (mid[c] * (1.0f - mid[c]));
    if (h == 1 && valid && a.gout.dz_rgb)
      *reinterpret_cast<float4*>(a.gout.dz_rgb + pc * 4) = make_float4(dz_rgb[0], dz_rgb[1], dz_rgb[2], 0.0f);

    // ---------------- stage 1: d hidden = W_rgb^T dz  (K = 32 with rows 4..6 live), ReLU mask -----------------
    X[0] = (h == 1) ? make_float4(dz_rgb[0], dz_rgb[1], dz_rgb[2], 0.0f) : zero4;
    X[64] = zero4; X[128] = zero4; X[192] = zero4;
    {
      const ReluBits<4> mb = load_relu_bits<4>(bits_at(P.num_layers));
      __builtin_amdgcn_sched_barrier(0);
      f32x16 acc[4];
      zero_acc<4>(acc);
      gemm_mode<MODE, 4, 4, true>(acc, pk + P.L.wT_rgb, pk + P.L.hT_rgb, X, 4, ln);
      store_masked_bits<4, SBF>(acc, X, mb, h, rb_epi(a.gout.da_mid, p0 * 128, 128));
    }
    // ---------------- stage 2: d bottleneck = W_mid[:, 34:]^T d a_mid -----------------
    {
      f32x16 acc[NB];
      zero_acc<NB>(acc);
      gemm_mode<MODE, NB, NB, true>(acc, pk + P.L.wT_mid_x, pk + P.L.hT_mid_x, X, 16, ln, rb_loop(a.gout.da_mid, p0 * 128, 128));
      store_act<NB, NB, false, SBF>(acc, X, rb_epi(a.gout.d_bott, p0 * W, W), h);
    }
    // ---------------- stage 3: heads pre-activation gradients, then d emb = [W_b; W_heads]^T [d b; dz_heads] ------
    {
      float4 q0 = zero4, q1 = zero4;  // heads rows 8q + 4h + j for q = 0, 1
      if (a.mode != RSN_MODE_INF) {
        if (h == 0) {
          const float raw = a.fwd.raw_density[pc];
          const float gs = a.gin.sigma ? a.gin.sigma[pc] * live : 0.0f;
          q0.x = gs * sigmoid_f(raw + P.density_bias);  // softplus'
          // predicted normal: pn = normalize(-normalize(n_raw)); G = g_pn + g_ndd * dir
          float dir[3], G[3] = {0.0f, 0.0f, 0.0f};
          const long long ray = pc / a.S;
#pragma unroll
          for (int c = 0; c < 3; ++c) dir[c] = a.directions[ray * 3 + c];
          if (a.gin.pred_normals) {
#pragma unroll
            for (int c = 0; c < 3; ++c) G[c] = a.gin.pred_normals[pc * 3 + c] * live;
          }
          float gd = a.gin.n_dot_d ? a.gin.n_dot_d[pc] * live : 0.0f;
          if (a.gin.ray_pn_loss || a.gin.ray_ori_loss) {
            // fused normal losses (model.py:403-407): per-ray upstream gradients of sum_s w |n - n_pred|^2 and
            // sum_s w max(0, n.d)^2; the per-sample gradients are formed here and never stored
            const float w = a.gin.weights[pc] * live;
            if (a.gin.ray_pn_loss) {
              const float gw = a.gin.ray_pn_loss[ray] * w * -2.0f;
#pragma unroll
              for (int c = 0; c < 3; ++c) G[c] += gw * (a.saved.normals[pc * 3 + c] - a.fwd.pred_normals[pc * 3 + c]);
            }
            if (a.gin.ray_ori_loss) gd += a.gin.ray_ori_loss[ray] * w * (2.0f * fmaxf(a.fwd.n_dot_d[pc], 0.0f));
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
          q0.y = gn[0]; q0.z = gn[1]; q0.w = gn[2];
          const float sr = sigmoid_f(hd.w);
          const float gr = a.gin.roughness ? a.gin.roughness[pc] * live : 0.0f;
          q1.x = gr * sr * (1.0f - sr);
        } else {
          q0.x = gcol[0] * (dif[0] * (1.0f - dif[0]));
          q0.y = gcol[1] * (dif[1] * (1.0f - dif[1]));
          q0.z = gcol[2] * (dif[2] * (1.0f - dif[2]));
          q1.x = gcol[0] * mid[0] * (tin[0] * (1.0f - tin[0]));
          q1.y = gcol[1] * mid[1] * (tin[1] * (1.0f - tin[1]));
          q1.z = gcol[2] * mid[2] * (tin[2] * (1.0f - tin[2]));
        }
      }
      X[(NB * 4 + 0) * 64] = q0;
      X[(NB * 4 + 1) * 64] = q1;
      X[(NB * 4 + 2) * 64] = zero4;
      X[(NB * 4 + 3) * 64] = zero4;
      if (valid && a.gout.dz_heads) {
        *reinterpret_cast<float4*>(a.gout.dz_heads + pc * 16 + 4 * h) = q0;
        *reinterpret_cast<float4*>(a.gout.dz_heads + pc * 16 + 8 + 4 * h) = q1;
      }
      const int l = P.num_layers - 1;
      const ReluBits<NB> mb = load_relu_bits<NB>(bits_at(l));
      __builtin_amdgcn_sched_barrier(0);
      f32x16 acc[NB];
      zero_acc<NB>(acc);
      gemm_mode<MODE, NB, NB, true>(acc, pk + P.L.wT_bh, pk + P.L.hT_bh, X, NB * 4 + 4, ln, rb_loop(a.gout.d_bott, p0 * W, W));
      store_masked_bits<NB, SBF>(acc, X, mb, h, rb_epi(a.gout.dy, (long long)l * a.act_stride + p0 * W, W));
    }
    // ---------------- stage 4: trunk, layers L-1 .. 1 -----------------
    f32x16 eacc[4];
    zero_acc<4>(eacc);
#pragma unroll 1
    for (int l = P.num_layers - 1; l >= 1; --l) {
      if (a.need_input_grad && l == P.skip_layer)
        gemm_mode<MODE, 4, 4, true>(eacc, pk + P.L.wT_enc_skip, pk + P.L.hT_enc_skip, X, NB * 4, ln);
      const ReluBits<NB> mb = load_relu_bits<NB>(bits_at(l - 1));
      __builtin_amdgcn_sched_barrier(0);
      f32x16 acc[NB];
      zero_acc<NB>(acc);
      gemm_mode<MODE, NB, NB, true>(acc, pk + P.L.wT_x[l], pk + P.L.hT_x[l], X, NB * 4, ln,
                          rb_loop(a.gout.dy, (long long)l * a.act_stride + p0 * W, W));  // reads (and keeps) dy[l]
      store_masked_bits<NB, SBF>(acc, X, mb, h, rb_epi(a.gout.dy, (long long)(l - 1) * a.act_stride + p0 * W, W));
    }
    if (LOOPST && !a.need_input_grad) {  // dy[0] has no GEMM behind it on this path: its rows leave here
      const RowBuf r0 = rb_loop(a.gout.dy, p0 * W, W);
#pragma unroll 4
      for (int it = 0; it < NB * 4; ++it) sv_put_it(r0, it, X[it * 64]);
    }
    // ---------------- stage 5: gradient w.r.t. the Gaussian's variance -> pixel_area / sqradius -----------------
    if (a.need_input_grad) {
      gemm_mode<MODE, 4, 4, true>(eacc, pk + P.L.wT_enc0, pk + P.L.hT_enc0, X, NB * 4, ln, rb_loop(a.gout.dy, p0 * W, W));  // keeps dy[0]
      store_act<4, 4, false>(eacc, X);  // d loss / d encoded input, slot order
      const float* encp = a.saved.enc + pc * RSN_K_ENC_PAD;
      float dvar[3];
#pragma unroll 1
      for (int c = 0; c < 3; ++c) {
        float part = 0.0f;
#pragma unroll 2
        for (int jj = 0; jj < 8; ++jj) {
          const float f = h ? P.freqs[8 + jj] : P.freqs[jj];
          const int u = c * 8 + jj, u2 = u + 24;
          const float gs = Xf[(u >> 2) * 256 + (u & 3)], gc = Xf[(u2 >> 2) * 256 + (u2 & 3)];
          const float fs = encp[(u >> 2) * 8 + 4 * h + (u & 3)], fc = encp[(u2 >> 2) * 8 + 4 * h + (u2 & 3)];
          part += (-0.5f * (f * f)) * (gs * fs + gc * fc);  // d/dvar [exp(-var f^2/2) sin(.)] = -f^2/2 * feature
        }
        const float tot = part + __shfl_xor(part, 32, 64);
        if (c == 0) dvar[0] = tot; else if (c == 1) dvar[1] = tot; else dvar[2] = tot;
      }
      if (h == 0 && valid && a.gout.d_input) {
        float g = 0.0f;
        if (a.mode == RSN_MODE_FRUSTUM) {
          const long long ray = pc / a.S;
          const int s = (int)(pc - ray * a.S);
          float o[3], d[3], dv[3];
#pragma unroll
          for (int c = 0; c < 3; ++c) { o[c] = a.origins[ray * 3 + c]; d[c] = a.directions[ray * 3 + c]; }
          frustum_dvar_dpa(o, d, a.pixel_area[ray], a.bins[ray * (a.S + 1) + s], a.bins[ray * (a.S + 1) + s + 1], dv);
          g = dvar[0] * dv[0] + dvar[1] * dv[1] + dvar[2] * dv[2];
        } else {  // INF: var_c = (0.6 sq)(1 - d_c^2)   (reflect_sampling_nerf_field.py:196)
#pragma unroll
          for (int c = 0; c < 3; ++c) {
            const float dc = a.directions[pc * 3 + c];
            g += dvar[c] * (0.6f * (1.0f - dc * dc));
          }
        }
        a.gout.d_input[pc] = g;
      }
    }
  }
}

static int launch_bwd_jobs(const rsn_field_desc* d, BwdArgs* js, int n, void* stream) {
  RSN_REQUIRE(n >= 1 && n <= RSN_MAX_JOBS, RSN_ERR_INVALID_ARGUMENT, "n_jobs=%d (1..%d)", n, RSN_MAX_JOBS);
  BwdJobs J = {};
  int rc = rsn_compute_layout(d, &J.s.L);
  if (rc != RSN_OK) return rc;
  RSN_REQUIRE(js[0].packed != nullptr, RSN_ERR_INVALID_ARGUMENT, "packed weights pointer is NULL");
  J.s.packed = js[0].packed;
  J.s.num_layers = d->num_layers;
  J.s.skip_layer = d->skip_layer;
  J.s.density_bias = d->density_bias;
  for (int i = 0; i < RSN_NUM_FREQS; ++i) J.s.freqs[i] = d->freqs[i];
  long long n_tiles = 0;
  for (int k = 0; k < n; ++k) {
    BwdArgs& a = js[k];
    RSN_REQUIRE(a.saved.relu_bits && a.saved.heads && a.gout.dy, RSN_ERR_INVALID_ARGUMENT,
                "job %d: saved relu_bits / heads and gout.dy are required", k);
    RSN_REQUIRE(!a.need_input_grad || (a.saved.enc && a.gout.d_input), RSN_ERR_INVALID_ARGUMENT,
                "job %d: need_input_grad needs saved.enc and gout.d_input", k);
    RSN_REQUIRE(a.mode == RSN_MODE_INF || (a.fwd.raw_density && a.fwd.diff && a.fwd.tint), RSN_ERR_INVALID_ARGUMENT,
                "job %d: forward values raw_density/diff/tint are required", k);
    RSN_REQUIRE(!(a.gin.ray_pn_loss || a.gin.ray_ori_loss) || (a.mode == RSN_MODE_FRUSTUM && a.gin.weights),
                RSN_ERR_INVALID_ARGUMENT, "job %d: fused normal losses need the level's weights (frustum levels only)", k);
    RSN_REQUIRE(!a.gin.ray_pn_loss || (a.saved.normals && a.fwd.pred_normals), RSN_ERR_INVALID_ARGUMENT,
                "job %d: ray_pn_loss needs saved.normals and the forward pred_normals", k);
    RSN_REQUIRE(!a.gin.ray_ori_loss || a.fwd.n_dot_d, RSN_ERR_INVALID_ARGUMENT, "job %d: ray_ori_loss needs the forward n_dot_d", k);
    if (a.n_rays <= 0) continue;
    const long long n_points = (long long)a.n_rays * a.S;
    a.act_stride = n_points * (long long)d->width;
    n_tiles += (n_points + 127) / 128;
    J.j[J.n_jobs++] = static_cast<const BwdJob&>(a);
  }
  if (J.n_jobs == 0) return RSN_OK;
  const int cached_cus = rsn_device_cus();
  const long long grid = n_tiles < (long long)cached_cus ? n_tiles : (long long)cached_cus;
  hipStream_t st = (hipStream_t)stream;
  if (rsn_ring_training(d)) {  // plain / split bf16 training at width 256: the LDS-ring kernels (256- / 128-point tiles)
    const int tp = d->mma_mode == RSN_MMA_BF16X6 ? 128 : 256;
    long long tn = 0;
    for (int k = 0; k < J.n_jobs; ++k) tn += ((long long)J.j[k].n_rays * J.j[k].S + tp - 1) / tp;
    return d->mma_mode == RSN_MMA_BF16X6 ? rsn_launch_field_x6_bwd(tn, st, J) : rsn_launch_field_bf16_bwd(tn, st, J);
  }
  const bool x6 = d->mma_mode == RSN_MMA_BF16X6;  // fp32-emulating split-bf16 sweeps (opt-in); else exact fp32
#define RSN_LAUNCH_BWD(NBV)                                                                                  \
  do {                                                                                                       \
    if (x6) hipLaunchKernelGGL((rsn_field_bwd_kernel<NBV, 1>), dim3((unsigned)grid), dim3(256), 0, st, J);    \
    else if (d->mma_mode == RSN_MMA_BF16)  /* reduced-precision training sweeps (bf16 operands, fp32 accumulate) */ \
      hipLaunchKernelGGL((rsn_field_bwd_kernel<NBV, 3>), dim3((unsigned)grid), dim3(256), 0, st, J);              \
    else hipLaunchKernelGGL((rsn_field_bwd_kernel<NBV, 0>), dim3((unsigned)grid), dim3(256), 0, st, J);       \
  } while (0)
  switch (d->width) {
    case 256: RSN_LAUNCH_BWD(8); break;
    case 128: RSN_LAUNCH_BWD(4); break;
    case 64: RSN_LAUNCH_BWD(2); break;
    default: RSN_REQUIRE(false, RSN_ERR_UNSUPPORTED, "width=%d unsupported", d->width);
  }
#undef RSN_LAUNCH_BWD
  RSN_HIP(hipGetLastError());
  return RSN_OK;
}

static int launch_bwd(const rsn_field_desc* d, BwdArgs& a, void* stream) { return launch_bwd_jobs(d, &a, 1, stream); }

// rsn_field_backward_jobs: the backward sweeps of several evaluations of the same field in one launch.
extern "C" int rsn_field_backward_jobs(const rsn_field_desc* desc, const float* packed, int32_t n_jobs,
                                       const rsn_field_bwd_job* jobs, void* stream) {
  RSN_REQUIRE(desc && jobs, RSN_ERR_INVALID_ARGUMENT, "desc/jobs is NULL");
  RSN_REQUIRE(n_jobs >= 1 && n_jobs <= RSN_MAX_JOBS, RSN_ERR_INVALID_ARGUMENT, "n_jobs=%d (1..%d)", n_jobs, RSN_MAX_JOBS);
  BwdArgs js[RSN_MAX_JOBS] = {};
  for (int k = 0; k < n_jobs; ++k) {
    const rsn_field_bwd_job& q = jobs[k];
    BwdArgs& a = js[k];
    RSN_REQUIRE(q.kind == 0 || q.kind == 1, RSN_ERR_INVALID_ARGUMENT, "job %d: kind=%d", k, q.kind);
    RSN_REQUIRE(q.n_rays >= 0 && q.saved && q.gout, RSN_ERR_INVALID_ARGUMENT, "job %d: n_rays / saved / gout", k);
    a.packed = packed;
    a.n_rays = q.n_rays; a.n_dev = q.n_dev; a.need_input_grad = q.need_input_grad;
    a.saved = *q.saved; a.gout = *q.gout;
    if (q.kind == 0) {
      RSN_REQUIRE(q.n_samples >= 1 && q.fwd && q.gin, RSN_ERR_INVALID_ARGUMENT, "job %d: n_samples / fwd / gin", k);
      RSN_REQUIRE(q.n_rays == 0 || (q.origins && q.directions && q.pixel_area && q.euclid_bins), RSN_ERR_INVALID_ARGUMENT,
                  "job %d: a ray input pointer is NULL", k);
      a.mode = RSN_MODE_FRUSTUM; a.S = q.n_samples;
      a.origins = q.origins; a.directions = q.directions; a.pixel_area = q.pixel_area; a.bins = q.euclid_bins;
      a.gin = *q.gin; a.fwd = *q.fwd;
    } else {
      RSN_REQUIRE(q.n_rays == 0 || (q.directions && q.sqradius && q.g_rgb), RSN_ERR_INVALID_ARGUMENT,
                  "job %d: an input pointer is NULL", k);
      a.mode = RSN_MODE_INF; a.S = 1;
      a.directions = q.directions; a.sqradius = q.sqradius;
      a.gin.color = q.g_rgb;
    }
  }
  return launch_bwd_jobs(desc, js, n_jobs, stream);
}

extern "C" int rsn_field_backward_frustum(const rsn_field_desc* desc, const float* packed, int32_t n_rays,
                                          const int32_t* n_dev, int32_t n_samples, const float* origins,
                                          const float* directions, const float* pixel_area, const float* euclid_bins,
                                          const rsn_field_outputs* fwd, const rsn_field_saved* saved,
                                          const rsn_field_grads_in* gin, const rsn_field_grads_out* gout,
                                          int32_t need_input_grad, void* stream) {
  RSN_REQUIRE(desc && fwd && saved && gin && gout, RSN_ERR_INVALID_ARGUMENT, "a struct pointer is NULL");
  RSN_REQUIRE(n_rays >= 0 && n_samples >= 1, RSN_ERR_INVALID_ARGUMENT, "n_rays=%d n_samples=%d", n_rays, n_samples);
  RSN_REQUIRE(n_rays == 0 || (origins && directions && pixel_area && euclid_bins), RSN_ERR_INVALID_ARGUMENT,
              "a ray input pointer is NULL");
  BwdArgs a = {};
  a.packed = packed;
  a.mode = RSN_MODE_FRUSTUM;
  a.n_rays = n_rays; a.n_dev = n_dev; a.S = n_samples; a.need_input_grad = need_input_grad;
  a.origins = origins; a.directions = directions; a.pixel_area = pixel_area; a.bins = euclid_bins;
  a.gin = *gin; a.fwd = *fwd; a.saved = *saved; a.gout = *gout;
  return launch_bwd(desc, a, stream);
}

extern "C" int rsn_field_backward_inf(const rsn_field_desc* desc, const float* packed, int32_t n_rays,
                                      const int32_t* n_dev, const float* directions, const float* sqradius,
                                      const rsn_field_saved* saved, const float* g_rgb,
                                      const rsn_field_grads_out* gout, int32_t need_input_grad, void* stream) {
  RSN_REQUIRE(desc && saved && gout, RSN_ERR_INVALID_ARGUMENT, "a struct pointer is NULL");
  RSN_REQUIRE(n_rays >= 0, RSN_ERR_INVALID_ARGUMENT, "n_rays=%d", n_rays);
  RSN_REQUIRE(n_rays == 0 || (directions && sqradius && g_rgb), RSN_ERR_INVALID_ARGUMENT, "an input pointer is NULL");
  BwdArgs a = {};
  a.packed = packed;
  a.mode = RSN_MODE_INF;
  a.n_rays = n_rays; a.n_dev = n_dev; a.S = 1; a.need_input_grad = need_input_grad;
  a.directions = directions; a.sqradius = sqradius;
  a.gin.color = g_rgb;
  a.saved = *saved; a.gout = *gout;
  return launch_bwd(desc, a, stream);
}
