// rsn_render.hip -- the HBM-side kernels around the field: samplers, compositing, reflection rays.
// All are per-ray, coalesced and tiny next to rsn_field.hip (a few bytes per sample); one wavefront
// (64 lanes) owns one ray wherever a scan along the ray is needed.
//
// Reference semantics restated: nerfstudio SpacedSampler/PDFSampler/get_weights/renderers (SURVEY
// §8(a) N5, N7-N11), reflect_sampling_nerf_components.py:14-36 (reciprocal spacing),
// reflect_sampling_nerf_model.py:215-229,240-241,267-289,312-313,338-339.
#include "rsn_mfma.h"  // shared device math (sin_big, sh34_attenuated); includes rsn_common.h

// ---------------------------------------------------------------------------------------------------
// helpers
// ---------------------------------------------------------------------------------------------------
__device__ __forceinline__ int dyn_count(int n_host, const int* n_dev) {
  if (!n_dev) return n_host;
  const int nd = *n_dev;
  return nd < n_host ? nd : n_host;
}

// torch.linspace(start=0, end, steps) element idx (fp32, symmetric evaluation like ATen's CPU kernel)
__device__ __forceinline__ float linspace0(float end, int steps, int idx) {
  if (steps <= 1) return 0.0f;
  const float step = end / (float)(steps - 1);
  return idx < steps / 2 ? step * (float)idx : end - step * (float)(steps - idx - 1);
}

__device__ __forceinline__ float spacing_fn(int kind, float tan_, float x) {
  return kind == RSN_SPACING_RECIPROCAL ? x / (1.0f / tan_ + x) : x;
}
__device__ __forceinline__ float spacing_inv(int kind, float tan_, float x) {
  return kind == RSN_SPACING_RECIPROCAL ? x / tan_ / (1.0f - x) : x;
}
__device__ __forceinline__ float to_euclid(int kind, float tan_, float b, float s_near, float s_far) {
  return spacing_inv(kind, tan_, b * s_far + (1.0f - b) * s_near);
}

__device__ __forceinline__ float nan_to_num_f(float x) {
  if (x != x) return 0.0f;
  if (x == __builtin_huge_valf()) return 3.4028234663852886e38f;
  if (x == -__builtin_huge_valf()) return -3.4028234663852886e38f;
  return x;
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off, 64);
  return v;
}

// Inclusive scan across the 64 lanes of a wave.  Accumulated in fp64 and rounded once per element, like
// ATen's CPU cumsum (acc_type<float> = double): the prefix sums are then independent of the scan order.
__device__ __forceinline__ double wave_scan_incl(double v, int lane) {
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) {
    const double t = __shfl_up(v, off, 64);
    if (lane >= off) v += t;
  }
  return v;
}

// ---------------------------------------------------------------------------------------------------
// spaced sampler (uniform / reciprocal), N7/N8/F12
// ---------------------------------------------------------------------------------------------------
__global__ void rsn_sample_spaced_kernel(int n_rays, const int* n_dev, int S, int kind, float tan_, const float* nears,
                                         const float* fars, const float* t_rand, float* spacing_bins,
                                         float* euclid_bins) {
  const int R = dyn_count(n_rays, n_dev);
  const long long total = (long long)R * (S + 1);
  for (long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x; e < total;
       e += (long long)gridDim.x * blockDim.x) {
    const int ray = (int)(e / (S + 1));
    const int k = (int)(e - (long long)ray * (S + 1));
    float b = linspace0(1.0f, S + 1, k);
    if (t_rand) {
      const float bl = linspace0(1.0f, S + 1, k > 0 ? k - 1 : 0);
      const float bu = linspace0(1.0f, S + 1, k < S ? k + 1 : S);
      const float lower = (k == 0) ? b : (b + bl) / 2.0f;
      const float upper = (k == S) ? b : (bu + b) / 2.0f;
      b = lower + (upper - lower) * t_rand[e];
    }
    const float s_near = spacing_fn(kind, tan_, nears[ray]);
    const float s_far = spacing_fn(kind, tan_, fars[ray]);
    spacing_bins[e] = b;
    euclid_bins[e] = to_euclid(kind, tan_, b, s_near, s_far);
  }
}

extern "C" int rsn_sample_spaced(int32_t n_rays, const int32_t* n_dev, int32_t n_samples, int32_t spacing, float tan_,
                                 const float* nears, const float* fars, const float* t_rand, float* spacing_bins,
                                 float* euclid_bins, void* stream) {
  RSN_REQUIRE(n_rays >= 0 && n_samples >= 1, RSN_ERR_INVALID_ARGUMENT, "n_rays=%d n_samples=%d", n_rays, n_samples);
  RSN_REQUIRE(spacing == RSN_SPACING_UNIFORM || spacing == RSN_SPACING_RECIPROCAL, RSN_ERR_INVALID_ARGUMENT,
              "spacing=%d", spacing);
  if (n_rays == 0) return RSN_OK;
  RSN_REQUIRE(nears && fars && spacing_bins && euclid_bins, RSN_ERR_INVALID_ARGUMENT, "a pointer is NULL");
  const long long total = (long long)n_rays * (n_samples + 1);
  const int threads = 256;
  long long blocks = (total + threads - 1) / threads;
  if (blocks > 2048) blocks = 2048;
  hipLaunchKernelGGL(rsn_sample_spaced_kernel, dim3((unsigned)blocks), dim3(threads), 0, (hipStream_t)stream, n_rays,
                     n_dev, n_samples, spacing, tan_, nears, fars, t_rand, spacing_bins, euclid_bins);
  RSN_HIP(hipGetLastError());
  return RSN_OK;
}

// ---------------------------------------------------------------------------------------------------
// PDF sampler (inverse CDF), N9.  One wave per ray; CDF and the existing bins live in LDS.
// ---------------------------------------------------------------------------------------------------
#define PDF_MAX_S 1024

__global__ __launch_bounds__(256) void rsn_sample_pdf_kernel(int n_rays, const int* n_dev, int s_in, int s_out, int kind,
                                                             float tan_, float hist_pad, const float* nears,
                                                             const float* fars, const float* weights,
                                                             const float* bins_in, const float* u_rand,
                                                             float* bins_out, float* euclid_out) {
  __shared__ float s_cdf[4][PDF_MAX_S + 1];
  __shared__ float s_bins[4][PDF_MAX_S + 1];
  const int R = dyn_count(n_rays, n_dev);
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  float* cdf = s_cdf[wid];
  float* eb = s_bins[wid];
  for (int ray = blockIdx.x * 4 + wid; ray < R; ray += gridDim.x * 4) {
    const float* w = weights + (long long)ray * s_in;
    // weights + histogram padding, their sum
    float part = 0.0f;
    for (int i = lane; i < s_in; i += 64) part += w[i] + hist_pad;
    float wsum = wave_sum(part);
    const float eps = 1e-5f;
    const float padding = fmaxf(eps - wsum, 0.0f);
    const float padd = padding / (float)s_in;
    wsum += padding;
    // cdf = min(1, cumsum(pdf)), with a leading 0
    double carry = 0.0;
    for (int base = 0; base < s_in; base += 64) {
      const int i = base + lane;
      const float pdf = (i < s_in) ? ((w[i] + hist_pad) + padd) / wsum : 0.0f;
      const double inc = wave_scan_incl((double)pdf, lane) + carry;
      if (i < s_in) cdf[i + 1] = fminf(1.0f, (float)inc);
      carry = __shfl(inc, 63, 64);
    }
    if (lane == 0) cdf[0] = 0.0f;
    for (int i = lane; i <= s_in; i += 64) eb[i] = bins_in[(long long)ray * (s_in + 1) + i];
    __builtin_amdgcn_s_waitcnt(0xc07f);  // lgkmcnt(0): this wave's LDS writes are visible to its own reads
    const int nb = s_out + 1;
    const float u_end = (float)(1.0 - (1.0 / (double)nb));
    const float s_near = spacing_fn(kind, tan_, nears[ray]);
    const float s_far = spacing_fn(kind, tan_, fars[ray]);
    for (int j = lane; j < nb; j += 64) {
      float u = linspace0(u_end, nb, j);
      if (u_rand) {
        u = u + u_rand[(long long)ray * nb + j] / (float)nb;
      } else {
        u = u + (float)(1.0 / (2.0 * (double)nb));
      }
      // searchsorted(cdf, u, side="right"): first index with cdf[idx] > u
      int lo = 0, hi = s_in + 1;
      while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        if (cdf[mid] <= u) lo = mid + 1; else hi = mid;
      }
      int below = lo - 1;
      below = below < 0 ? 0 : (below > s_in ? s_in : below);
      const int above = lo > s_in ? s_in : lo;
      const float c0 = cdf[below], c1 = cdf[above];
      const float b0 = eb[below], b1 = eb[above];
      float t = (u - c0) / (c1 - c0);
      if (t != t) t = 0.0f;                       // nan_to_num(., 0)
      t = nan_to_num_f(t);
      t = fminf(fmaxf(t, 0.0f), 1.0f);
      const float b = b0 + t * (b1 - b0);
      const long long o = (long long)ray * nb + j;
      bins_out[o] = b;
      euclid_out[o] = to_euclid(kind, tan_, b, s_near, s_far);
    }
  }
}

extern "C" int rsn_sample_pdf(int32_t n_rays, const int32_t* n_dev, int32_t s_in, int32_t s_out, int32_t spacing,
                              float tan_, float histogram_padding, const float* nears, const float* fars,
                              const float* weights, const float* spacing_bins_in, const float* u_rand,
                              float* spacing_bins_out, float* euclid_bins_out, void* stream) {
  RSN_REQUIRE(n_rays >= 0 && s_in >= 1 && s_out >= 1, RSN_ERR_INVALID_ARGUMENT, "n_rays=%d s_in=%d s_out=%d", n_rays,
              s_in, s_out);
  RSN_REQUIRE(s_in <= PDF_MAX_S, RSN_ERR_UNSUPPORTED, "s_in=%d > %d", s_in, PDF_MAX_S);
  if (n_rays == 0) return RSN_OK;
  RSN_REQUIRE(nears && fars && weights && spacing_bins_in && spacing_bins_out && euclid_bins_out,
              RSN_ERR_INVALID_ARGUMENT, "a pointer is NULL");
  int blocks = (n_rays + 3) / 4;
  if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(rsn_sample_pdf_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, n_rays, n_dev, s_in, s_out,
                     spacing, tan_, histogram_padding, nears, fars, weights, spacing_bins_in, u_rand, spacing_bins_out,
                     euclid_bins_out);
  RSN_HIP(hipGetLastError());
  return RSN_OK;
}

// ---------------------------------------------------------------------------------------------------
// compositing: weights (N5) + RGB / accumulation / median depth / normals / semantic renderers (N10, N11).
// One wave per ray; exclusive prefix sum of delta*sigma by a wave-level scan with a running carry.
// ---------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void rsn_composite_kernel(int n_rays, const int* n_dev, int S, int background,
                                                            int flags, const rsn_composite_io io) {
  const int eval_mode = flags & RSN_COMP_EVAL;
  const int clip_rgb = flags & (RSN_COMP_EVAL | RSN_COMP_CLIP_RGB);
  const int R = dyn_count(n_rays, n_dev);
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  for (int ray = blockIdx.x * 4 + wid; ray < R; ray += gridDim.x * 4) {
    const long long sbase = (long long)ray * S;
    const float* bins = io.euclid_bins + (long long)ray * (S + 1);
    double carry_dd = 0.0, carry_w = 0.0;
    float acc = 0.0f, c0 = 0.0f, c1 = 0.0f, c2 = 0.0f;
    float d0 = 0.0f, d1 = 0.0f, d2 = 0.0f, t0s = 0.0f, t1s = 0.0f, t2s = 0.0f;
    float n0 = 0.0f, n1 = 0.0f, n2 = 0.0f, rs = 0.0f;
    float pl = 0.0f, ol = 0.0f;  // per-ray partials of the normal losses (model.py:403-407)
    int median_idx = -1;
    for (int base = 0; base < S; base += 64) {
      const int i = base + lane;
      const bool in = i < S;
      float ta = 0.0f, tb = 0.0f, sg = 0.0f;
      if (in) {
        ta = bins[i];
        tb = bins[i + 1];
        sg = io.sigma[sbase + i];
      }
      const float dd = in ? (tb - ta) * sg : 0.0f;
      const double incl = wave_scan_incl((double)dd, lane);
      // exclusive prefix = the previous lane's inclusive one (not incl - dd: an infinite dd must not poison its own T)
      const double prev = __shfl_up(incl, 1, 64);
      const float excl = (float)((lane > 0 ? prev : 0.0) + carry_dd);
      carry_dd += __shfl(incl, 63, 64);
      const float alpha = 1.0f - expf(-dd);
      const float T = expf(-excl);
      float w = in ? nan_to_num_f(alpha * T) : 0.0f;
      if (in && io.weights) io.weights[sbase + i] = w;
      // median depth: first index whose inclusive cumulative weight reaches 0.5
      const double cwd = wave_scan_incl((double)w, lane) + carry_w;
      carry_w = __shfl(cwd, 63, 64);
      const float cw = (float)cwd;
      if (median_idx < 0) {
        const unsigned long long hit = __ballot(in && cw >= 0.5f);
        if (hit) median_idx = base + (int)__builtin_ctzll(hit);
      }
      acc += w;
      if (in) {
        const long long o3 = (sbase + i) * 3;
        float r = io.color[o3 + 0], g = io.color[o3 + 1], b = io.color[o3 + 2];
        if (eval_mode) { r = nan_to_num_f(r); g = nan_to_num_f(g); b = nan_to_num_f(b); }
        c0 += w * r; c1 += w * g; c2 += w * b;
        if (io.diff) {
          float x = io.diff[o3 + 0], y = io.diff[o3 + 1], z = io.diff[o3 + 2];
          if (eval_mode) { x = nan_to_num_f(x); y = nan_to_num_f(y); z = nan_to_num_f(z); }
          d0 += w * x; d1 += w * y; d2 += w * z;
        }
        if (io.tint) {
          float x = io.tint[o3 + 0], y = io.tint[o3 + 1], z = io.tint[o3 + 2];
          if (eval_mode) { x = nan_to_num_f(x); y = nan_to_num_f(y); z = nan_to_num_f(z); }
          t0s += w * x; t1s += w * y; t2s += w * z;
        }
        if (io.pred_normals) {
          const float p0 = io.pred_normals[o3 + 0], p1 = io.pred_normals[o3 + 1], p2 = io.pred_normals[o3 + 2];
          n0 += w * p0; n1 += w * p1; n2 += w * p2;
          if (io.pn_loss_ray) {
            const float e0 = io.normals[o3 + 0] - p0, e1 = io.normals[o3 + 1] - p1, e2 = io.normals[o3 + 2] - p2;
            pl += w * (e0 * e0 + e1 * e1 + e2 * e2);
          }
        }
        if (io.ori_loss_ray) {
          const float nd = fmaxf(io.n_dot_d[sbase + i], 0.0f);
          ol += w * (nd * nd);
        }
        if (io.roughness) rs += w * io.roughness[sbase + i];
      }
    }
    acc = wave_sum(acc);
    c0 = wave_sum(c0); c1 = wave_sum(c1); c2 = wave_sum(c2);
    if (io.diff) { d0 = wave_sum(d0); d1 = wave_sum(d1); d2 = wave_sum(d2); }
    if (io.tint) { t0s = wave_sum(t0s); t1s = wave_sum(t1s); t2s = wave_sum(t2s); }
    if (io.pred_normals) { n0 = wave_sum(n0); n1 = wave_sum(n1); n2 = wave_sum(n2); }
    if (io.roughness) rs = wave_sum(rs);
    if (io.pn_loss_ray) pl = wave_sum(pl);
    if (io.ori_loss_ray) ol = wave_sum(ol);
    if (lane == 0) {
      if (io.pn_loss_ray) io.pn_loss_ray[ray] = pl;
      if (io.ori_loss_ray) io.ori_loss_ray[ray] = ol;
      const float rem = 1.0f - acc;
      float bg0 = 0.0f, bg1 = 0.0f, bg2 = 0.0f;
      if (background == 1) { bg0 = bg1 = bg2 = 1.0f; }
      if (background == 2) { bg0 = io.bg_rgb[ray * 3 + 0]; bg1 = io.bg_rgb[ray * 3 + 1]; bg2 = io.bg_rgb[ray * 3 + 2]; }
      if (background != 0) { c0 = c0 + bg0 * rem; c1 = c1 + bg1 * rem; c2 = c2 + bg2 * rem; }
      if (io.rgb) {  // eval: renderer clamp; CLIP_RGB: the model's own clip(0,1) -- same clamp
        if (clip_rgb) {
          c0 = fminf(fmaxf(c0, 0.0f), 1.0f); c1 = fminf(fmaxf(c1, 0.0f), 1.0f); c2 = fminf(fmaxf(c2, 0.0f), 1.0f);
        }
        io.rgb[ray * 3 + 0] = c0; io.rgb[ray * 3 + 1] = c1; io.rgb[ray * 3 + 2] = c2;
      }
      if (io.accumulation) io.accumulation[ray] = acc;
      if (io.depth) {
        int mi = median_idx < 0 ? S - 1 : median_idx;
        io.depth[ray] = (bins[mi] + bins[mi + 1]) / 2.0f;
      }
      if (io.diff_out) {  // renderer_rgb: white background (reflect_sampling_nerf_model.py:215)
        float x = d0 + rem, y = d1 + rem, z = d2 + rem;
        if (eval_mode) { x = fminf(fmaxf(x, 0.0f), 1.0f); y = fminf(fmaxf(y, 0.0f), 1.0f); z = fminf(fmaxf(z, 0.0f), 1.0f); }
        io.diff_out[ray * 3 + 0] = x; io.diff_out[ray * 3 + 1] = y; io.diff_out[ray * 3 + 2] = z;
      }
      if (io.tint_out) {  // renderer_factor: "random" => no background (model.py:123,217)
        float x = t0s, y = t1s, z = t2s;
        if (eval_mode) { x = fminf(fmaxf(x, 0.0f), 1.0f); y = fminf(fmaxf(y, 0.0f), 1.0f); z = fminf(fmaxf(z, 0.0f), 1.0f); }
        io.tint_out[ray * 3 + 0] = x; io.tint_out[ray * 3 + 1] = y; io.tint_out[ray * 3 + 2] = z;
      }
      if (io.normals_out) {  // NormalsRenderer: n / (|n| + 1e-10)
        const float nn = sqrtf(n0 * n0 + n1 * n1 + n2 * n2) + 1e-10f;
        io.normals_out[ray * 3 + 0] = n0 / nn; io.normals_out[ray * 3 + 1] = n1 / nn; io.normals_out[ray * 3 + 2] = n2 / nn;
      }
      if (io.roughness_out) io.roughness_out[ray] = rs;
    }
  }
}

extern "C" int rsn_composite(int32_t n_rays, const int32_t* n_dev, int32_t n_samples, int32_t background,
                             int32_t flags, const rsn_composite_io* io, void* stream) {
  RSN_REQUIRE(io, RSN_ERR_INVALID_ARGUMENT, "io is NULL");
  RSN_REQUIRE(n_rays >= 0 && n_samples >= 1, RSN_ERR_INVALID_ARGUMENT, "n_rays=%d n_samples=%d", n_rays, n_samples);
  RSN_REQUIRE(background >= 0 && background <= 2, RSN_ERR_INVALID_ARGUMENT, "background=%d", background);
  if (n_rays == 0) return RSN_OK;
  RSN_REQUIRE(io->sigma && io->euclid_bins && io->color, RSN_ERR_INVALID_ARGUMENT, "sigma/bins/color is NULL");
  RSN_REQUIRE(background != 2 || io->bg_rgb, RSN_ERR_INVALID_ARGUMENT, "background=2 needs bg_rgb");
  RSN_REQUIRE((!io->diff_out || io->diff) && (!io->tint_out || io->tint) && (!io->normals_out || io->pred_normals) &&
                  (!io->roughness_out || io->roughness),
              RSN_ERR_INVALID_ARGUMENT, "a surface-attribute output is requested without its per-sample input");
  RSN_REQUIRE((!io->pn_loss_ray || (io->normals && io->pred_normals)) && (!io->ori_loss_ray || io->n_dot_d),
              RSN_ERR_INVALID_ARGUMENT, "a per-ray loss output is requested without its per-sample inputs");
  int blocks = (n_rays + 3) / 4;
  if (blocks > 8192) blocks = 8192;
  hipLaunchKernelGGL(rsn_composite_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, n_rays, n_dev, n_samples,
                     background, flags, *io);
  RSN_HIP(hipGetLastError());
  return RSN_OK;
}

// ---------------------------------------------------------------------------------------------------
// compositing backward (training).  w_i = (1 - e^{-x_i}) T_i, x_i = delta_i sigma_i, T_i = exp(-sum_{j<i} x_j):
//   dL/dx_k = g_w[k] * T_{k+1} - sum_{i>k} g_w[i] w_i          (T_{k+1} = T_k - w_k)
// One wave per ray; the suffix sum is total - inclusive prefix, both in fp64.
// ---------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void rsn_composite_bwd_kernel(int n_rays, const int* n_dev, int S, int background,
                                                                int flags, int detach_weights,
                                                                const rsn_composite_bwd_io io) {
  const int R = dyn_count(n_rays, n_dev);
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  for (int ray = blockIdx.x * 4 + wid; ray < R; ray += gridDim.x * 4) {
    const long long sbase = (long long)ray * S;
    const float* bins = io.euclid_bins + (long long)ray * (S + 1);
    float bg[3] = {0.0f, 0.0f, 0.0f};
    if (background == 1) bg[0] = bg[1] = bg[2] = 1.0f;
    if (background == 2) { bg[0] = io.bg_rgb[ray * 3 + 0]; bg[1] = io.bg_rgb[ray * 3 + 1]; bg[2] = io.bg_rgb[ray * 3 + 2]; }
    // pass 1: unclipped composite and accumulation (clip mask of the model's torch.clip)
    float acc = 0.0f, c0 = 0.0f, c1 = 0.0f, c2 = 0.0f;
    for (int i = lane; i < S; i += 64) {
      const float w = io.weights[sbase + i];
      acc += w;
      c0 += w * io.color[(sbase + i) * 3 + 0];
      c1 += w * io.color[(sbase + i) * 3 + 1];
      c2 += w * io.color[(sbase + i) * 3 + 2];
    }
    acc = wave_sum(acc); c0 = wave_sum(c0); c1 = wave_sum(c1); c2 = wave_sum(c2);
    const float rem = 1.0f - acc;
    if (background != 0) { c0 = c0 + bg[0] * rem; c1 = c1 + bg[1] * rem; c2 = c2 + bg[2] * rem; }
    float g[3] = {io.g_rgb[ray * 3 + 0], io.g_rgb[ray * 3 + 1], io.g_rgb[ray * 3 + 2]};
    if (flags & RSN_COMP_CLIP_RGB) {  // torch.clamp backward: pass where min <= x <= max
      if (!(c0 >= 0.0f && c0 <= 1.0f)) g[0] = 0.0f;
      if (!(c1 >= 0.0f && c1 <= 1.0f)) g[1] = 0.0f;
      if (!(c2 >= 0.0f && c2 <= 1.0f)) g[2] = 0.0f;
    }
    const float g_r = io.g_roughness ? io.g_roughness[ray] : 0.0f;
    const float g_a = io.g_accumulation ? io.g_accumulation[ray] : 0.0f;
    if (lane == 0 && io.g_bg) {
      io.g_bg[ray * 3 + 0] = g[0] * rem; io.g_bg[ray * 3 + 1] = g[1] * rem; io.g_bg[ray * 3 + 2] = g[2] * rem;
    }
    // pass 2: total of g_w[i] * w_i
    double total = 0.0;
    if (!detach_weights) {
      for (int i = lane; i < S; i += 64) {
        const float w = io.weights[sbase + i];
        float gw = g[0] * (io.color[(sbase + i) * 3 + 0] - bg[0]) + g[1] * (io.color[(sbase + i) * 3 + 1] - bg[1]) +
                   g[2] * (io.color[(sbase + i) * 3 + 2] - bg[2]);
        if (io.roughness) gw += g_r * io.roughness[sbase + i];
        gw += g_a;
        total += (double)gw * (double)w;
      }
#pragma unroll
      for (int off = 32; off >= 1; off >>= 1) total += __shfl_xor(total, off, 64);
    }
    // pass 3: per-sample gradients
    double carry_x = 0.0, carry_p = 0.0;
    for (int base = 0; base < S; base += 64) {
      const int i = base + lane;
      const bool in = i < S;
      float w = 0.0f, gw = 0.0f, x = 0.0f, delta = 0.0f;
      if (in) {
        w = io.weights[sbase + i];
        delta = bins[i + 1] - bins[i];
        x = delta * io.sigma[sbase + i];
        const long long o3 = (sbase + i) * 3;
        if (io.g_color) { io.g_color[o3 + 0] = w * g[0]; io.g_color[o3 + 1] = w * g[1]; io.g_color[o3 + 2] = w * g[2]; }
        if (io.g_roughness_sample) io.g_roughness_sample[sbase + i] = w * g_r;
        gw = g[0] * (io.color[o3 + 0] - bg[0]) + g[1] * (io.color[o3 + 1] - bg[1]) + g[2] * (io.color[o3 + 2] - bg[2]);
        if (io.roughness) gw += g_r * io.roughness[sbase + i];
        gw += g_a;
      }
      if (!detach_weights && io.g_sigma) {
        const double xin = wave_scan_incl((double)x, lane) + carry_x;        // sum_{j<=i} x_j
        const double pin = wave_scan_incl((double)gw * (double)w, lane) + carry_p;
        carry_x = __shfl(xin, 63, 64);
        carry_p = __shfl(pin, 63, 64);
        const float t_next = expf(-(float)xin);
        const float gx = gw * t_next - (float)(total - pin);
        if (in) io.g_sigma[sbase + i] = delta * gx;
      } else if (in && io.g_sigma) {
        io.g_sigma[sbase + i] = 0.0f;
      }
    }
  }
}

extern "C" int rsn_composite_backward(int32_t n_rays, const int32_t* n_dev, int32_t n_samples, int32_t background,
                                      int32_t flags, int32_t detach_weights, const rsn_composite_bwd_io* io,
                                      void* stream) {
  RSN_REQUIRE(io, RSN_ERR_INVALID_ARGUMENT, "io is NULL");
  RSN_REQUIRE(n_rays >= 0 && n_samples >= 1, RSN_ERR_INVALID_ARGUMENT, "n_rays=%d n_samples=%d", n_rays, n_samples);
  RSN_REQUIRE(background >= 0 && background <= 2, RSN_ERR_INVALID_ARGUMENT, "background=%d", background);
  RSN_REQUIRE(!(flags & RSN_COMP_EVAL), RSN_ERR_INVALID_ARGUMENT, "backward is defined for training mode only");
  if (n_rays == 0) return RSN_OK;
  RSN_REQUIRE(io->sigma && io->euclid_bins && io->color && io->weights && io->g_rgb, RSN_ERR_INVALID_ARGUMENT,
              "sigma/bins/color/weights/g_rgb is NULL");
  RSN_REQUIRE(background != 2 || io->bg_rgb, RSN_ERR_INVALID_ARGUMENT, "background=2 needs bg_rgb");
  int blocks = (n_rays + 3) / 4;
  if (blocks > 8192) blocks = 8192;
  hipLaunchKernelGGL(rsn_composite_bwd_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, n_rays, n_dev,
                     n_samples, background, flags, detach_weights, *io);
  RSN_HIP(hipGetLastError());
  return RSN_OK;
}

// column sums (bias gradients): each workgroup owns a chunk of rows, thread t owns columns t, t+256, ...
__global__ __launch_bounds__(256) void rsn_colsum_kernel(long long n_rows, int n_cols, int ld, const float* x,
                                                         float* out, long long rows_per_block) {
  const long long r0 = (long long)blockIdx.x * rows_per_block;
  long long r1 = r0 + rows_per_block;
  if (r1 > n_rows) r1 = n_rows;
  for (int c = threadIdx.x; c < n_cols; c += 256) {
    float s0 = 0.0f, s1 = 0.0f, s2 = 0.0f, s3 = 0.0f;
    long long r = r0;
    for (; r + 3 < r1; r += 4) {
      s0 += x[r * ld + c]; s1 += x[(r + 1) * ld + c]; s2 += x[(r + 2) * ld + c]; s3 += x[(r + 3) * ld + c];
    }
    for (; r < r1; ++r) s0 += x[r * ld + c];
    atomicAdd(&out[c], (s0 + s1) + (s2 + s3));
  }
}

extern "C" int rsn_colsum(int64_t n_rows, int32_t n_cols, int32_t ld, const float* x, float* out, int32_t accumulate,
                          void* stream) {
  RSN_REQUIRE(n_rows >= 0 && n_cols >= 1 && ld >= n_cols, RSN_ERR_INVALID_ARGUMENT, "n_rows=%lld n_cols=%d ld=%d",
              (long long)n_rows, n_cols, ld);
  RSN_REQUIRE(out, RSN_ERR_INVALID_ARGUMENT, "out is NULL");
  hipStream_t st = (hipStream_t)stream;
  if (!accumulate) RSN_HIP(hipMemsetAsync(out, 0, sizeof(float) * n_cols, st));
  if (n_rows == 0) return RSN_OK;
  RSN_REQUIRE(x, RSN_ERR_INVALID_ARGUMENT, "x is NULL");
  const long long rows_per_block = 2048;
  const long long blocks = (n_rows + rows_per_block - 1) / rows_per_block;
  hipLaunchKernelGGL(rsn_colsum_kernel, dim3((unsigned)blocks), dim3(256), 0, st, (long long)n_rows, n_cols, ld, x, out,
                     rows_per_block);
  RSN_HIP(hipGetLastError());
  return RSN_OK;
}

__global__ __launch_bounds__(256) void rsn_ray_sum_kernel(int n_rays, const int* n_dev, int S, const float* x,
                                                          float* out) {
  const int R = dyn_count(n_rays, n_dev);
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  for (int ray = blockIdx.x * 4 + wid; ray < R; ray += gridDim.x * 4) {
    float s = 0.0f;
    for (int i = lane; i < S; i += 64) s += x[(long long)ray * S + i];
    s = wave_sum(s);
    if (lane == 0) out[ray] = s;
  }
}

extern "C" int rsn_ray_sum(int32_t n_rays, const int32_t* n_dev, int32_t n_samples, const float* x, float* out,
                           void* stream) {
  RSN_REQUIRE(n_rays >= 0 && n_samples >= 1, RSN_ERR_INVALID_ARGUMENT, "n_rays=%d n_samples=%d", n_rays, n_samples);
  if (n_rays == 0) return RSN_OK;
  RSN_REQUIRE(x && out, RSN_ERR_INVALID_ARGUMENT, "a pointer is NULL");
  int blocks = (n_rays + 3) / 4;
  if (blocks > 8192) blocks = 8192;
  hipLaunchKernelGGL(rsn_ray_sum_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, n_rays, n_dev, n_samples, x,
                     out);
  RSN_HIP(hipGetLastError());
  return RSN_OK;
}

__global__ void rsn_reflect_bwd_kernel(int n_rays, const int* n_masked, const int* ray_index, const float* n_dot_d,
                                       const float* roughness, const float* g_sq, const float* g_pa, float* g_rough) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n_rays) g_rough[i] = 0.0f;
}

__global__ void rsn_reflect_bwd_scatter_kernel(int n_rays, const int* n_masked, const int* ray_index,
                                               const float* n_dot_d, const float* roughness, const float* g_sq,
                                               const float* g_pa, float* g_rough) {
  const int M = dyn_count(n_rays, n_masked);
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= M) return;
  const int r = ray_index[i];
  const float gs = (g_sq ? g_sq[i] : 0.0f) + 3.141592653589793f * (g_pa ? g_pa[i] : 0.0f);
  // sqradius = 2 |n.d| roughness^2
  g_rough[r] = gs * (2.0f * fabsf(n_dot_d[r])) * (2.0f * roughness[r]);
}

extern "C" int rsn_reflect_backward(int32_t n_rays, const int32_t* n_masked, const int32_t* ray_index,
                                    const float* n_dot_d, const float* roughness, const float* g_sqradius,
                                    const float* g_pixel_area, float* g_roughness, void* stream) {
  RSN_REQUIRE(n_rays >= 0, RSN_ERR_INVALID_ARGUMENT, "n_rays=%d", n_rays);
  if (n_rays == 0) return RSN_OK;
  RSN_REQUIRE(n_masked && ray_index && n_dot_d && roughness && g_roughness, RSN_ERR_INVALID_ARGUMENT,
              "a pointer is NULL");
  const int threads = 256, blocks = (n_rays + threads - 1) / threads;
  hipLaunchKernelGGL(rsn_reflect_bwd_kernel, dim3(blocks), dim3(threads), 0, (hipStream_t)stream, n_rays, n_masked,
                     ray_index, n_dot_d, roughness, g_sqradius, g_pixel_area, g_roughness);
  hipLaunchKernelGGL(rsn_reflect_bwd_scatter_kernel, dim3(blocks), dim3(threads), 0, (hipStream_t)stream, n_rays,
                     n_masked, ray_index, n_dot_d, roughness, g_sqradius, g_pixel_area, g_roughness);
  RSN_HIP(hipGetLastError());
  return RSN_OK;
}

// ---------------------------------------------------------------------------------------------------
// reflection rays: mask, stable compaction, secondary-ray construction (model.py:222-229,240-241,267-289).
// A single 1024-thread workgroup walks the rays in order (block scan per 1024-ray chunk) so the
// compaction is stable like the reference's boolean-mask gather.
// ---------------------------------------------------------------------------------------------------
// Two passes over 1024-ray blocks, any number of workgroups (round 1 ran ONE workgroup over all rays):
//   count   : mask, n.d, the default reflect colours; workspace[b] = masked rays of block b;
//   scatter : base_b = sum of workspace[0..b) (every block sums for itself: <= R/1024 integers), then the stable
//             in-block compaction by wave ballots; the last block publishes M.
#define RSN_REFLECT_BLOCK 1024
__global__ __launch_bounds__(RSN_REFLECT_BLOCK) void rsn_reflect_count_kernel(int R, const rsn_reflect_io io) {
  __shared__ int s_wave_tot[16];
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int r = blockIdx.x * RSN_REFLECT_BLOCK + tid;
  bool mk = false;
  if (r < R) {
    float d[3], n[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) { d[c] = io.directions[r * 3 + c]; n[c] = io.pred_normals[r * 3 + c]; }
    const float ndd = n[0] * d[0] + n[1] * d[1] + n[2] * d[2];
    const float accv = io.accumulation[r];
    mk = (accv > 1e-2f) && (ndd < 0.0f);
    io.mask[r] = mk ? 1 : 0;
    if (io.n_dot_d) io.n_dot_d[r] = ndd;
    const float dflt = 1.0f * (1.0f - accv);
#pragma unroll
    for (int c = 0; c < 3; ++c) { io.reflect_coarse[r * 3 + c] = dflt; io.reflect_fine[r * 3 + c] = dflt; }
  }
  const unsigned long long bal = __ballot(mk);
  if (lane == 0) s_wave_tot[wid] = __builtin_popcountll(bal);
  __syncthreads();
  if (tid == 0) {
    int t = 0;
#pragma unroll
    for (int wv = 0; wv < 16; ++wv) t += s_wave_tot[wv];
    io.workspace[blockIdx.x] = t;
  }
}

__global__ __launch_bounds__(RSN_REFLECT_BLOCK) void rsn_reflect_scatter_kernel(int R, float reflect_far,
                                                                                const rsn_reflect_io io) {
  __shared__ int s_wave_tot[16];
  __shared__ int s_part[16];
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  // base of this block: sum of the preceding blocks' counts
  int part = 0;
  for (int j = tid; j < (int)blockIdx.x; j += RSN_REFLECT_BLOCK) part += io.workspace[j];
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) part += __shfl_xor(part, off, 64);
  if (lane == 0) s_part[wid] = part;
  const int r = blockIdx.x * RSN_REFLECT_BLOCK + tid;
  const bool mk = r < R && io.mask[r] != 0;
  const unsigned long long bal = __ballot(mk);
  const int in_wave = __builtin_popcountll(bal & ((1ull << lane) - 1ull));
  if (lane == 0) s_wave_tot[wid] = __builtin_popcountll(bal);
  __syncthreads();
  int base = 0, wave_off = 0, chunk_tot = 0;
#pragma unroll
  for (int wv = 0; wv < 16; ++wv) {
    base += s_part[wv];
    const int t = s_wave_tot[wv];
    if (wv < wid) wave_off += t;
    chunk_tot += t;
  }
  if (mk) {
    float d[3], n[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) { d[c] = io.directions[r * 3 + c]; n[c] = io.pred_normals[r * 3 + c]; }
    const float ndd = n[0] * d[0] + n[1] * d[1] + n[2] * d[2];
    const int i = base + wave_off + in_wave;
    io.ray_index[i] = r;
    const float dep = io.depth[r];
    float rf[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      io.origins2[i * 3 + c] = io.origins[r * 3 + c] + dep * d[c];
      rf[c] = d[c] - 2.0f * ndd * n[c];
    }
    const float nrm = fmaxf(sqrtf(rf[0] * rf[0] + rf[1] * rf[1] + rf[2] * rf[2]), 1e-12f);
#pragma unroll
    for (int c = 0; c < 3; ++c) io.directions2[i * 3 + c] = rf[c] / nrm;
    const float rough = io.roughness[r];
    const float sq = 2.0f * fabsf(ndd) * (rough * rough);
    io.sqradius[i] = sq;
    io.pixel_area2[i] = 3.141592653589793f * sq;
    io.nears2[i] = 0.0f;  // zeros_like(nears) * self.near == 0 (reflect_sampling_nerf_model.py:287)
    io.fars2[i] = reflect_far;
  }
  if (blockIdx.x == gridDim.x - 1 && tid == 0) *io.n_masked = base + chunk_tot;
}

extern "C" size_t rsn_reflect_workspace_bytes(int32_t n_rays) {
  return (size_t)((n_rays > 0 ? n_rays : 0) + RSN_REFLECT_BLOCK - 1) / RSN_REFLECT_BLOCK * sizeof(int32_t) + sizeof(int32_t);
}

extern "C" int rsn_reflect_setup(int32_t n_rays, float reflect_far, const rsn_reflect_io* io, void* stream) {
  RSN_REQUIRE(io, RSN_ERR_INVALID_ARGUMENT, "io is NULL");
  RSN_REQUIRE(n_rays >= 0, RSN_ERR_INVALID_ARGUMENT, "n_rays=%d", n_rays);
  RSN_REQUIRE(io->n_masked, RSN_ERR_INVALID_ARGUMENT, "n_masked is NULL");
  RSN_REQUIRE(n_rays == 0 || (io->origins && io->directions && io->accumulation && io->depth && io->pred_normals &&
                              io->roughness && io->mask && io->ray_index && io->origins2 && io->directions2 &&
                              io->sqradius && io->pixel_area2 && io->nears2 && io->fars2 && io->reflect_coarse &&
                              io->reflect_fine && io->workspace),
              RSN_ERR_INVALID_ARGUMENT, "a pointer is NULL");
  hipStream_t st = (hipStream_t)stream;
  if (n_rays == 0) {
    RSN_HIP(hipMemsetAsync(io->n_masked, 0, sizeof(int32_t), st));
    return RSN_OK;
  }
  const unsigned blocks = (unsigned)((n_rays + RSN_REFLECT_BLOCK - 1) / RSN_REFLECT_BLOCK);
  hipLaunchKernelGGL(rsn_reflect_count_kernel, dim3(blocks), dim3(RSN_REFLECT_BLOCK), 0, st, n_rays, *io);
  hipLaunchKernelGGL(rsn_reflect_scatter_kernel, dim3(blocks), dim3(RSN_REFLECT_BLOCK), 0, st, n_rays, reflect_far, *io);
  RSN_HIP(hipGetLastError());
  return RSN_OK;
}

__global__ void rsn_reflect_combine_kernel(int n_max, const int* n_masked, const int* ray_index, const float* diff,
                                           const float* tint, const float* comp, float* out) {
  const int M = dyn_count(n_max, n_masked);
  const int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= M * 3) return;
  const int i = e / 3, c = e - i * 3;
  const int r = ray_index[i];
  const float v = diff[r * 3 + c] + tint[r * 3 + c] * comp[i * 3 + c];
  out[r * 3 + c] = fminf(fmaxf(v, 0.0f), 1.0f);
}

extern "C" int rsn_reflect_combine(int32_t n_rays_max, const int32_t* n_masked, const int32_t* ray_index,
                                   const float* diff, const float* tint, const float* comp, float* out, void* stream) {
  RSN_REQUIRE(n_rays_max >= 0, RSN_ERR_INVALID_ARGUMENT, "n_rays_max=%d", n_rays_max);
  if (n_rays_max == 0) return RSN_OK;
  RSN_REQUIRE(n_masked && ray_index && diff && tint && comp && out, RSN_ERR_INVALID_ARGUMENT, "a pointer is NULL");
  const int threads = 256;
  const int blocks = (n_rays_max * 3 + threads - 1) / threads;
  hipLaunchKernelGGL(rsn_reflect_combine_kernel, dim3(blocks), dim3(threads), 0, (hipStream_t)stream, n_rays_max,
                     n_masked, ray_index, diff, tint, comp, out);
  RSN_HIP(hipGetLastError());
  return RSN_OK;
}

__global__ void rsn_reflect_combine_bwd_kernel(int n_max, const int* n_masked, const int* ray_index, const float* diff,
                                               const float* tint, const float* comp, const float* g_out,
                                               float* g_comp) {
  const int M = dyn_count(n_max, n_masked);
  const int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= M * 3) return;
  const int i = e / 3, c = e - i * 3;
  const int r = ray_index[i];
  const float v = diff[r * 3 + c] + tint[r * 3 + c] * comp[i * 3 + c];
  g_comp[i * 3 + c] = (v >= 0.0f && v <= 1.0f) ? g_out[r * 3 + c] * tint[r * 3 + c] : 0.0f;
}

extern "C" int rsn_reflect_combine_backward(int32_t n_rays_max, const int32_t* n_masked, const int32_t* ray_index,
                                            const float* diff, const float* tint, const float* comp,
                                            const float* g_out, float* g_comp, void* stream) {
  RSN_REQUIRE(n_rays_max >= 0, RSN_ERR_INVALID_ARGUMENT, "n_rays_max=%d", n_rays_max);
  if (n_rays_max == 0) return RSN_OK;
  RSN_REQUIRE(n_masked && ray_index && diff && tint && comp && g_out && g_comp, RSN_ERR_INVALID_ARGUMENT,
              "a pointer is NULL");
  const int threads = 256;
  const int blocks = (n_rays_max * 3 + threads - 1) / threads;
  hipLaunchKernelGGL(rsn_reflect_combine_bwd_kernel, dim3(blocks), dim3(threads), 0, (hipStream_t)stream, n_rays_max,
                     n_masked, ray_index, diff, tint, comp, g_out, g_comp);
  RSN_HIP(hipGetLastError());
  return RSN_OK;
}

__global__ void rsn_reflect_default_bwd_kernel(int n_rays, const uint8_t* mask, const float* gc, const float* gf,
                                               float* g_acc) {
  const int r = blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= n_rays) return;
  float v = 0.0f;
  if (!mask[r]) {
    if (gc) v -= gc[r * 3 + 0] + gc[r * 3 + 1] + gc[r * 3 + 2];
    if (gf) v -= gf[r * 3 + 0] + gf[r * 3 + 1] + gf[r * 3 + 2];
  }
  g_acc[r] = v;
}

extern "C" int rsn_reflect_default_backward(int32_t n_rays, const uint8_t* mask, const float* g_reflect_coarse,
                                            const float* g_reflect_fine, float* g_accumulation, void* stream) {
  RSN_REQUIRE(n_rays >= 0, RSN_ERR_INVALID_ARGUMENT, "n_rays=%d", n_rays);
  if (n_rays == 0) return RSN_OK;
  RSN_REQUIRE(mask && g_accumulation, RSN_ERR_INVALID_ARGUMENT, "a pointer is NULL");
  const int threads = 256, blocks = (n_rays + threads - 1) / threads;
  hipLaunchKernelGGL(rsn_reflect_default_bwd_kernel, dim3(blocks), dim3(threads), 0, (hipStream_t)stream, n_rays, mask,
                     g_reflect_coarse, g_reflect_fine, g_accumulation);
  RSN_HIP(hipGetLastError());
  return RSN_OK;
}

// ---------------------------------------------------------------------------------------------------
// granular geometry kernels (Field.get_blob / contract / get_reflection): one thread per sample
// ---------------------------------------------------------------------------------------------------
__global__ void rsn_gaussians_kernel(long long n, const float* o_, const float* d_, const float* pa_, const float* t0_,
                                     const float* t1_, float* mean, float* cov) {
  const long long p = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= n) return;
  float o[3], d[3];
#pragma unroll
  for (int c = 0; c < 3; ++c) { o[c] = o_[p * 3 + c]; d[c] = d_[p * 3 + c]; }
  const float radius = sqrtf(pa_[p]) / 1.7724538509055159f;
  const float mu = (t0_[p] + t1_[p]) / 2.0f, hw = (t1_[p] - t0_[p]) / 2.0f;
  const float hw2 = hw * hw, mu2 = mu * mu, den = 3.0f * mu2 + hw2, hw4 = hw2 * hw2;
  const float tmean = mu + (2.0f * mu * hw2) / den;
  const float var_t = hw2 / 3.0f - 0.26666666666666666f * ((hw4 * (12.0f * mu2 - hw2)) / (den * den));
  const float var_r = (radius * radius) * (mu2 / 4.0f + 0.4166666666666667f * hw2 - (0.26666666666666666f * hw4) / den);
  const float dmag = fmaxf(d[0] * d[0] + d[1] * d[1] + d[2] * d[2], 1e-10f);
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    mean[p * 3 + i] = o[i] + d[i] * tmean;
#pragma unroll
    for (int j = 0; j < 3; ++j)
      cov[p * 9 + i * 3 + j] = var_t * (d[i] * d[j]) + var_r * ((i == j ? 1.0f : 0.0f) - d[i] * (d[j] / dmag));
  }
}

__global__ void rsn_contract_kernel(long long n, const float* mean_, const float* cov_, float* mo, float* co) {
  const long long p = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= n) return;
  float m[3], S[3][3];
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    m[i] = mean_[p * 3 + i];
#pragma unroll
    for (int j = 0; j < 3; ++j) S[i][j] = cov_[p * 9 + i * 3 + j];
  }
  const float n2 = m[0] * m[0] + m[1] * m[1] + m[2] * m[2];
  const float nn = sqrtf(n2);
  const bool outside = nn > 1.0f;
  float J[3][3];
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      const float eye = (i == j) ? 1.0f : 0.0f;
      J[i][j] = outside ? ((2.0f * nn - 2.0f) * (eye - m[i] * m[j] / n2) + eye) / n2 : eye;
    }
  const float sc = (2.0f * nn - 1.0f) / n2;
#pragma unroll
  for (int i = 0; i < 3; ++i) mo[p * 3 + i] = outside ? sc * m[i] : m[i];
  float JS[3][3];
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int b = 0; b < 3; ++b) JS[i][b] = J[i][0] * S[0][b] + J[i][1] * S[1][b] + J[i][2] * S[2][b];
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      float v = JS[i][0] * J[0][k] + JS[i][1] * J[1][k] + JS[i][2] * J[2][k];
      if (i == k) v = fmaxf(v, 0.0f);
      co[p * 9 + i * 3 + k] = v;
    }
}

__global__ void rsn_reflection_kernel(long long n, const float* d_, const float* n_, float* refl, float* ndd) {
  const long long p = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= n) return;
  const float d0 = d_[p * 3], d1 = d_[p * 3 + 1], d2 = d_[p * 3 + 2];
  const float n0 = n_[p * 3], n1 = n_[p * 3 + 1], n2 = n_[p * 3 + 2];
  const float dot = d0 * n0 + d1 * n1 + d2 * n2;
  if (ndd) ndd[p] = dot;
  if (refl) {
    const float r0 = d0 - 2.0f * dot * n0, r1 = d1 - 2.0f * dot * n1, r2 = d2 - 2.0f * dot * n2;
    const float len = fmaxf(sqrtf(r0 * r0 + r1 * r1 + r2 * r2), 1e-12f);
    refl[p * 3] = r0 / len; refl[p * 3 + 1] = r1 / len; refl[p * 3 + 2] = r2 / len;
  }
}

// Standalone encoders (the Field's `direction_encoding(...)` / `position_encoding(...)` modules called directly);
// inside the field kernels the same arithmetic is fused (rsn_field.hip).  Output columns in the reference's order.
__global__ void rsn_sh34_kernel(long long n, const float* dirs, const float* rough, float* out) {
  const long long p = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= n) return;
  float sh[34];
  sh34_attenuated(dirs[p * 3], dirs[p * 3 + 1], dirs[p * 3 + 2], rough ? rough[p] : 0.0f, sh);
#pragma unroll
  for (int i = 0; i < 34; ++i) out[p * 34 + i] = sh[i];
}

struct IpeFreqs {
  float f[RSN_NUM_FREQS];
};

// one thread per (point, coordinate): columns c*16+j (sin), 48+c*16+j (sin(. + pi/2)), 96+c (raw input)
__global__ void rsn_ipe_kernel(long long n3, const float* means, const float* cov_diag, IpeFreqs fr, float* out) {
  const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= n3) return;
  const long long p = t / 3;
  const int c = (int)(t - p * 3);
  const float x = means[t];
  const float v = cov_diag ? cov_diag[t] : 0.0f;
  const float sx = 6.283185307179586f * x;
  float* row = out + p * RSN_ENC_DIM;
#pragma unroll 4
  for (int j = 0; j < RSN_NUM_FREQS; ++j) {
    const float f = fr.f[j];
    const float ang = sx * f;
    const float e = cov_diag ? expf(-0.5f * (v * (f * f))) : 1.0f;
    row[c * 16 + j] = e * sin_big(ang);
    row[48 + c * 16 + j] = e * sin_big(ang + 1.5707963267948966f);
  }
  row[96 + c] = x;
}

#define RSN_ELEMENTWISE_LAUNCH(kernel, n, ...)                                                          \
  do {                                                                                                  \
    const int threads = 256;                                                                            \
    const long long blocks = ((n) + threads - 1) / threads;                                             \
    hipLaunchKernelGGL(kernel, dim3((unsigned)blocks), dim3(threads), 0, (hipStream_t)stream, (long long)(n), __VA_ARGS__); \
    RSN_HIP(hipGetLastError());                                                                         \
  } while (0)

extern "C" int rsn_gaussians(int64_t n, const float* origins, const float* directions, const float* pixel_area,
                             const float* starts, const float* ends, float* mean, float* cov, void* stream) {
  RSN_REQUIRE(n >= 0, RSN_ERR_INVALID_ARGUMENT, "n=%lld", (long long)n);
  if (n == 0) return RSN_OK;
  RSN_REQUIRE(origins && directions && pixel_area && starts && ends && mean && cov, RSN_ERR_INVALID_ARGUMENT,
              "a pointer is NULL");
  RSN_ELEMENTWISE_LAUNCH(rsn_gaussians_kernel, n, origins, directions, pixel_area, starts, ends, mean, cov);
  return RSN_OK;
}

extern "C" int rsn_contract(int64_t n, const float* mean, const float* cov, float* mean_out, float* cov_out,
                            void* stream) {
  RSN_REQUIRE(n >= 0, RSN_ERR_INVALID_ARGUMENT, "n=%lld", (long long)n);
  if (n == 0) return RSN_OK;
  RSN_REQUIRE(mean && cov && mean_out && cov_out, RSN_ERR_INVALID_ARGUMENT, "a pointer is NULL");
  RSN_ELEMENTWISE_LAUNCH(rsn_contract_kernel, n, mean, cov, mean_out, cov_out);
  return RSN_OK;
}

extern "C" int rsn_reflection(int64_t n, const float* directions, const float* normals, float* reflections,
                              float* n_dot_d, void* stream) {
  RSN_REQUIRE(n >= 0, RSN_ERR_INVALID_ARGUMENT, "n=%lld", (long long)n);
  if (n == 0) return RSN_OK;
  RSN_REQUIRE(directions && normals, RSN_ERR_INVALID_ARGUMENT, "a pointer is NULL");
  RSN_ELEMENTWISE_LAUNCH(rsn_reflection_kernel, n, directions, normals, reflections, n_dot_d);
  return RSN_OK;
}

extern "C" int rsn_sh34_encode(int64_t n, const float* directions, const float* roughness, float* out, void* stream) {
  RSN_REQUIRE(n >= 0, RSN_ERR_INVALID_ARGUMENT, "n=%lld", (long long)n);
  if (n == 0) return RSN_OK;
  RSN_REQUIRE(directions && out, RSN_ERR_INVALID_ARGUMENT, "a pointer is NULL");
  RSN_ELEMENTWISE_LAUNCH(rsn_sh34_kernel, n, directions, roughness, out);
  return RSN_OK;
}

extern "C" int rsn_ipe_encode(int64_t n, const float* means, const float* cov_diag, const float* freqs16, float* out,
                              void* stream) {
  RSN_REQUIRE(n >= 0, RSN_ERR_INVALID_ARGUMENT, "n=%lld", (long long)n);
  if (n == 0) return RSN_OK;
  RSN_REQUIRE(means && freqs16 && out, RSN_ERR_INVALID_ARGUMENT, "a pointer is NULL");
  IpeFreqs fr;
  for (int i = 0; i < RSN_NUM_FREQS; ++i) fr.f[i] = freqs16[i];  // host array, like rsn_field_desc.freqs
  RSN_ELEMENTWISE_LAUNCH(rsn_ipe_kernel, n * 3, means, cov_diag, fr, out);
  return RSN_OK;
}
