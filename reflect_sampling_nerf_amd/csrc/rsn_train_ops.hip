// rsn_train_ops.hip -- the two steps directly after the hot path in a training iteration:
//   rsn_loss_forward_backward : get_loss_dict (reflect_sampling_nerf_model.py:346-430) -- the 8 loss terms AND their
//                               gradients w.r.t. the model outputs in one pass over the samples;
//   rsn_radam_step            : RAdam (reference config.py:50-53 -> torch.optim.RAdam semantics) over all parameter
//                               tensors in one launch.
// Both are HBM-bound streaming kernels.
#include "rsn_common.h"

// ---------------------------------------------------------------------------------------------------
// losses.  Terms (index): 0 loss_mid_coarse, 1 loss_mid_fine, 2 loss_reflect_mid_coarse, 3 loss_reflect_mid_fine
// (MSE means over R*3), 4/5 predicted_normal_loss_{coarse,fine} = sum w |n - n_pred|^2, 6/7 orientation_loss_
// {coarse,fine} = sum w max(0, n.d)^2.  losses[k] receives the UNSCALED term; gradients are scaled by coef[k].
// ---------------------------------------------------------------------------------------------------
struct LossArgs {
  int R, Sc, Sf;
  const float* image;                       // [R,3] (already blended with the white background if RGBA)
  const float* rgb[4];                      // mid_rgb_coarse, mid_rgb_fine, mid_reflect_coarse, mid_reflect_fine [R,3]
  const float* w[2];                        // weights_coarse [R,Sc], weights_fine [R,Sf]   (detached)
  const float* nrm[2];                      // normals_* [R,S,3]                            (detached)
  const float* pn[2];                       // pred_normals_* [R,S,3]
  const float* ndd[2];                      // n_dot_d_* [R,S]
  float coef[8];
  float* losses;                            // [8], accumulated (zeroed by the launcher)
  float* g_rgb[4];                          // [R,3]
  float* g_pn[2];                           // [R,S,3]
  float* g_ndd[2];                          // [R,S]
};

__device__ __forceinline__ float block_sum_256(float v, float* sh) {
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off, 64);
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  __syncthreads();
  if (lane == 0) sh[wid] = v;
  __syncthreads();
  return sh[0] + sh[1] + sh[2] + sh[3];
}

__global__ __launch_bounds__(256) void rsn_loss_kernel(const LossArgs a) {
  __shared__ float sh[4];
  float part[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  const long long stride = (long long)gridDim.x * blockDim.x;
  const long long tid = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  // MSE terms over R*3 elements
  const long long n3 = (long long)a.R * 3;
  const float inv = 1.0f / (float)n3;
  for (long long e = tid; e < n3; e += stride) {
    const float img = a.image[e];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const float dlt = a.rgb[k][e] - img;
      part[k] += dlt * dlt;
      if (a.g_rgb[k]) a.g_rgb[k][e] = a.coef[k] * 2.0f * dlt * inv;
    }
  }
  // per-sample terms
#pragma unroll
  for (int lv = 0; lv < 2; ++lv) {
    const long long n = (long long)a.R * (lv == 0 ? a.Sc : a.Sf);
    for (long long e = tid; e < n; e += stride) {
      const float w = a.w[lv][e];
      float s2 = 0.0f;
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        const float dlt = a.nrm[lv][e * 3 + c] - a.pn[lv][e * 3 + c];
        s2 += dlt * dlt;
        if (a.g_pn[lv]) a.g_pn[lv][e * 3 + c] = a.coef[4 + lv] * w * (-2.0f * dlt);
      }
      part[4 + lv] += w * s2;
      const float nd = fmaxf(a.ndd[lv][e], 0.0f);
      part[6 + lv] += w * (nd * nd);
      if (a.g_ndd[lv]) a.g_ndd[lv][e] = a.coef[6 + lv] * w * (2.0f * nd);
    }
  }
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    float v = block_sum_256(part[k], sh);
    if (k < 4) v *= inv;
    if (threadIdx.x == 0) atomicAdd(&a.losses[k], v);
  }
}

extern "C" int rsn_loss_forward_backward(int32_t n_rays, int32_t s_coarse, int32_t s_fine, const float* image,
                                         const float* const* rgb4, const float* const* weights2,
                                         const float* const* normals2, const float* const* pred_normals2,
                                         const float* const* n_dot_d2, const float* coef8, float* losses8,
                                         float* const* g_rgb4, float* const* g_pred_normals2, float* const* g_n_dot_d2,
                                         void* stream) {
  RSN_REQUIRE(n_rays >= 1 && s_coarse >= 1 && s_fine >= 1, RSN_ERR_INVALID_ARGUMENT, "n_rays=%d s=%d,%d", n_rays,
              s_coarse, s_fine);
  RSN_REQUIRE(image && rgb4 && weights2 && normals2 && pred_normals2 && n_dot_d2 && coef8 && losses8 && g_rgb4 &&
                  g_pred_normals2 && g_n_dot_d2,
              RSN_ERR_INVALID_ARGUMENT, "a pointer is NULL");
  LossArgs a;
  a.R = n_rays; a.Sc = s_coarse; a.Sf = s_fine; a.image = image; a.losses = losses8;
  for (int k = 0; k < 4; ++k) { a.rgb[k] = rgb4[k]; a.g_rgb[k] = g_rgb4[k]; RSN_REQUIRE(rgb4[k], RSN_ERR_INVALID_ARGUMENT, "rgb[%d] NULL", k); }
  for (int k = 0; k < 2; ++k) {
    a.w[k] = weights2[k]; a.nrm[k] = normals2[k]; a.pn[k] = pred_normals2[k]; a.ndd[k] = n_dot_d2[k];
    a.g_pn[k] = g_pred_normals2[k]; a.g_ndd[k] = g_n_dot_d2[k];
    RSN_REQUIRE(a.w[k] && a.nrm[k] && a.pn[k] && a.ndd[k], RSN_ERR_INVALID_ARGUMENT, "level %d input NULL", k);
  }
  for (int k = 0; k < 8; ++k) a.coef[k] = coef8[k];
  hipStream_t st = (hipStream_t)stream;
  RSN_HIP(hipMemsetAsync(losses8, 0, 8 * sizeof(float), st));
  const long long work = (long long)n_rays * (s_coarse > s_fine ? s_coarse : s_fine);
  long long blocks = (work + 255) / 256;
  if (blocks > 1024) blocks = 1024;
  hipLaunchKernelGGL(rsn_loss_kernel, dim3((unsigned)blocks), dim3(256), 0, st, a);
  RSN_HIP(hipGetLastError());
  return RSN_OK;
}

// In-place chain rule for the loss gradients: g_rgb[k] *= up[k] (k = 0..3), g_pn[lv] *= up[4 + lv], g_ndd[lv] *=
// up[6 + lv] with the eight upstream gradients d total / d loss_k read from DEVICE memory -- one launch instead of a
// dozen elementwise multiplications in the host framework's autograd.
struct LossScaleArgs {
  int R, Sc, Sf;
  const float* up;
  float* g_rgb[4];
  float* g_pn[2];
  float* g_ndd[2];
};

__global__ __launch_bounds__(256) void rsn_loss_scale_kernel(const LossScaleArgs a) {
  const long long stride = (long long)gridDim.x * blockDim.x;
  const long long tid = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  float up[8];
#pragma unroll
  for (int k = 0; k < 8; ++k) up[k] = a.up[k];
  const long long n3 = (long long)a.R * 3;
  for (long long e = tid; e < n3; e += stride) {
#pragma unroll
    for (int k = 0; k < 4; ++k)
      if (a.g_rgb[k]) a.g_rgb[k][e] *= up[k];
  }
#pragma unroll
  for (int lv = 0; lv < 2; ++lv) {
    const long long n = (long long)a.R * (lv == 0 ? a.Sc : a.Sf);
    for (long long e = tid; e < n; e += stride) {
      if (a.g_pn[lv]) {
        a.g_pn[lv][e * 3 + 0] *= up[4 + lv];
        a.g_pn[lv][e * 3 + 1] *= up[4 + lv];
        a.g_pn[lv][e * 3 + 2] *= up[4 + lv];
      }
      if (a.g_ndd[lv]) a.g_ndd[lv][e] *= up[6 + lv];
    }
  }
}

extern "C" int rsn_loss_scale_grads(int32_t n_rays, int32_t s_coarse, int32_t s_fine, const float* upstream8,
                                    float* const* g_rgb4, float* const* g_pred_normals2, float* const* g_n_dot_d2,
                                    void* stream) {
  RSN_REQUIRE(n_rays >= 1 && s_coarse >= 1 && s_fine >= 1, RSN_ERR_INVALID_ARGUMENT, "n_rays=%d s=%d,%d", n_rays,
              s_coarse, s_fine);
  RSN_REQUIRE(upstream8 && g_rgb4 && g_pred_normals2 && g_n_dot_d2, RSN_ERR_INVALID_ARGUMENT, "a pointer is NULL");
  LossScaleArgs a;
  a.R = n_rays; a.Sc = s_coarse; a.Sf = s_fine; a.up = upstream8;
  for (int k = 0; k < 4; ++k) a.g_rgb[k] = g_rgb4[k];
  for (int k = 0; k < 2; ++k) { a.g_pn[k] = g_pred_normals2[k]; a.g_ndd[k] = g_n_dot_d2[k]; }
  const long long work = (long long)n_rays * (s_coarse > s_fine ? s_coarse : s_fine);
  long long blocks = (work + 255) / 256;
  if (blocks > 1024) blocks = 1024;
  hipLaunchKernelGGL(rsn_loss_scale_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, a);
  RSN_HIP(hipGetLastError());
  return RSN_OK;
}

// ---------------------------------------------------------------------------------------------------
// get_loss_dict on per-ray quantities (training step): the per-sample normal terms arrive reduced per ray from the
// compositing epilogue (rsn_composite_io.pn_loss_ray / ori_loss_ray), so one small launch over R rays yields the eight
// terms and the gradients of the four colour terms; the chain rule is a second launch over R rays.
// ---------------------------------------------------------------------------------------------------
struct LossRaysArgs {
  int R;
  const float* image;
  const float* rgb[4];
  const float* pn_ray[2];
  const float* ori_ray[2];
  float coef[8];
  float* losses;
  float* g_rgb[4];
};

// ONE workgroup of 1024 threads: the eight reported loss values are reduced in a FIXED order (per-thread strided sums, a wave
// butterfly, the sixteen wave sums in index order) and are bit-reproducible from run to run -- the data is a few hundred KB,
// a second workgroup would only add float atomics whose arrival order varies (the gradients never depended on it).
__device__ __forceinline__ float block_sum_1024(float v, float* sh) {
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off, 64);
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  __syncthreads();
  if (lane == 0) sh[wid] = v;
  __syncthreads();
  float t = 0.0f;
#pragma unroll
  for (int w = 0; w < 16; ++w) t += sh[w];
  return t;
}

__global__ __launch_bounds__(1024) void rsn_loss_rays_kernel(const LossRaysArgs a) {
  __shared__ float sh[16];
  float part[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  const int stride = gridDim.x * blockDim.x;
  const int tid = blockIdx.x * blockDim.x + threadIdx.x;
  const int n3 = a.R * 3;
  const float inv = 1.0f / (float)n3;
  for (int e = tid; e < n3; e += stride) {
    const float img = a.image[e];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const float dlt = a.rgb[k][e] - img;
      part[k] += dlt * dlt;
      if (a.g_rgb[k]) a.g_rgb[k][e] = a.coef[k] * 2.0f * dlt * inv;
    }
  }
  for (int r = tid; r < a.R; r += stride) {
#pragma unroll
    for (int lv = 0; lv < 2; ++lv) {
      part[4 + lv] += a.pn_ray[lv][r];
      part[6 + lv] += a.ori_ray[lv][r];
    }
  }
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    float v = block_sum_1024(part[k], sh);
    if (k < 4) v *= inv;
    if (threadIdx.x == 0) a.losses[k] = v;
  }
}

extern "C" int rsn_loss_rays_forward(int32_t n_rays, const float* image, const float* const* rgb4,
                                     const float* const* pn_loss_ray2, const float* const* ori_loss_ray2,
                                     const float* coef8, float* losses8, float* const* g_rgb4, void* stream) {
  RSN_REQUIRE(n_rays >= 1, RSN_ERR_INVALID_ARGUMENT, "n_rays=%d", n_rays);
  RSN_REQUIRE(image && rgb4 && pn_loss_ray2 && ori_loss_ray2 && coef8 && losses8 && g_rgb4, RSN_ERR_INVALID_ARGUMENT,
              "a pointer is NULL");
  LossRaysArgs a;
  a.R = n_rays; a.image = image; a.losses = losses8;
  for (int k = 0; k < 4; ++k) { a.rgb[k] = rgb4[k]; a.g_rgb[k] = g_rgb4[k]; RSN_REQUIRE(rgb4[k], RSN_ERR_INVALID_ARGUMENT, "rgb[%d] NULL", k); }
  for (int k = 0; k < 2; ++k) {
    a.pn_ray[k] = pn_loss_ray2[k]; a.ori_ray[k] = ori_loss_ray2[k];
    RSN_REQUIRE(a.pn_ray[k] && a.ori_ray[k], RSN_ERR_INVALID_ARGUMENT, "level %d per-ray loss NULL", k);
  }
  for (int k = 0; k < 8; ++k) a.coef[k] = coef8[k];
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(rsn_loss_rays_kernel, dim3(1), dim3(1024), 0, st, a);  // one workgroup: fixed-order sums (see the kernel)
  RSN_HIP(hipGetLastError());
  return RSN_OK;
}

struct LossRaysBwdArgs {
  int R;
  const float* up;
  float coef[8];
  float* g_rgb[4];
  float* g_pn_ray[2];
  float* g_ori_ray[2];
};

__global__ __launch_bounds__(256) void rsn_loss_rays_bwd_kernel(const LossRaysBwdArgs a) {
  const int stride = gridDim.x * blockDim.x;
  const int tid = blockIdx.x * blockDim.x + threadIdx.x;
  float up[8];
#pragma unroll
  for (int k = 0; k < 8; ++k) up[k] = a.up[k];
  for (int e = tid; e < a.R * 3; e += stride) {
#pragma unroll
    for (int k = 0; k < 4; ++k)
      if (a.g_rgb[k]) a.g_rgb[k][e] *= up[k];
  }
  for (int r = tid; r < a.R; r += stride) {
#pragma unroll
    for (int lv = 0; lv < 2; ++lv) {
      if (a.g_pn_ray[lv]) a.g_pn_ray[lv][r] = a.coef[4 + lv] * up[4 + lv];
      if (a.g_ori_ray[lv]) a.g_ori_ray[lv][r] = a.coef[6 + lv] * up[6 + lv];
    }
  }
}

extern "C" int rsn_loss_rays_backward(int32_t n_rays, const float* upstream8, const float* coef8, float* const* g_rgb4,
                                      float* const* g_pn_ray2, float* const* g_ori_ray2, void* stream) {
  RSN_REQUIRE(n_rays >= 1, RSN_ERR_INVALID_ARGUMENT, "n_rays=%d", n_rays);
  RSN_REQUIRE(upstream8 && coef8 && g_rgb4 && g_pn_ray2 && g_ori_ray2, RSN_ERR_INVALID_ARGUMENT, "a pointer is NULL");
  LossRaysBwdArgs a;
  a.R = n_rays; a.up = upstream8;
  for (int k = 0; k < 8; ++k) a.coef[k] = coef8[k];
  for (int k = 0; k < 4; ++k) a.g_rgb[k] = g_rgb4[k];
  for (int k = 0; k < 2; ++k) { a.g_pn_ray[k] = g_pn_ray2[k]; a.g_ori_ray[k] = g_ori_ray2[k]; }
  int blocks = (n_rays * 3 + 255) / 256;
  if (blocks > 256) blocks = 256;
  hipLaunchKernelGGL(rsn_loss_rays_bwd_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, a);
  RSN_HIP(hipGetLastError());
  return RSN_OK;
}

// ---------------------------------------------------------------------------------------------------
// RAdam, multi-tensor (torch.optim.RAdam: betas, eps, no weight decay, decoupled = false)
// ---------------------------------------------------------------------------------------------------
#define RADAM_MAX_TENSORS 48
struct RAdamArgs {
  float* p[RADAM_MAX_TENSORS];
  const float* g[RADAM_MAX_TENSORS];
  float* m[RADAM_MAX_TENSORS];
  float* v[RADAM_MAX_TENSORS];
  int n[RADAM_MAX_TENSORS];
  int n_tensors;
  float lr, beta1, beta2, eps, bias_c1, bias_c2_sqrt, rect;  // rect < 0: variance not tractable yet (rho_t <= 5)
};

__global__ __launch_bounds__(256) void rsn_radam_kernel(const RAdamArgs a) {
  const int t = blockIdx.y;
  if (t >= a.n_tensors || a.g[t] == nullptr) return;
  float* __restrict__ p = a.p[t];
  const float* __restrict__ g = a.g[t];
  float* __restrict__ m = a.m[t];
  float* __restrict__ v = a.v[t];
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < a.n[t]; i += gridDim.x * blockDim.x) {
    const float gi = g[i];
    const float mi = a.beta1 * m[i] + (1.0f - a.beta1) * gi;   // exp_avg.lerp_(grad, 1 - beta1)
    const float vi = a.beta2 * v[i] + (1.0f - a.beta2) * gi * gi;
    m[i] = mi;
    v[i] = vi;
    const float mhat = mi / a.bias_c1;
    if (a.rect >= 0.0f) {
      const float adaptive = a.bias_c2_sqrt / (sqrtf(vi) + a.eps);
      p[i] = p[i] + mhat * a.lr * adaptive * a.rect * -1.0f;
    } else {
      p[i] = p[i] + mhat * a.lr * -1.0f;
    }
  }
}

extern "C" int rsn_radam_step(int32_t n_tensors, float* const* params, const float* const* grads, float* const* exp_avg,
                              float* const* exp_avg_sq, const int32_t* sizes, int32_t step, float lr, float beta1,
                              float beta2, float eps, void* stream) {
  RSN_REQUIRE(n_tensors >= 1 && n_tensors <= RADAM_MAX_TENSORS, RSN_ERR_INVALID_ARGUMENT, "n_tensors=%d (max %d)",
              n_tensors, RADAM_MAX_TENSORS);
  RSN_REQUIRE(params && grads && exp_avg && exp_avg_sq && sizes && step >= 1, RSN_ERR_INVALID_ARGUMENT,
              "a pointer is NULL or step < 1");
  RAdamArgs a;
  int max_n = 0;
  for (int t = 0; t < n_tensors; ++t) {
    a.p[t] = params[t]; a.g[t] = grads[t]; a.m[t] = exp_avg[t]; a.v[t] = exp_avg_sq[t]; a.n[t] = sizes[t];
    RSN_REQUIRE(a.p[t] && a.m[t] && a.v[t] && a.n[t] >= 0, RSN_ERR_INVALID_ARGUMENT, "tensor %d has NULL state", t);
    if (a.n[t] > max_n) max_n = a.n[t];
  }
  a.n_tensors = n_tensors;
  a.lr = lr; a.beta1 = beta1; a.beta2 = beta2; a.eps = eps;
  // torch.optim.radam._single_tensor_radam, evaluated in double like Python does
  const double b1t = pow((double)beta1, (double)step), b2t = pow((double)beta2, (double)step);
  const double bias1 = 1.0 - b1t, bias2 = 1.0 - b2t;
  const double rho_inf = 2.0 / (1.0 - (double)beta2) - 1.0;
  const double rho_t = rho_inf - 2.0 * step * b2t / bias2;
  a.bias_c1 = (float)bias1;
  a.bias_c2_sqrt = (float)sqrt(bias2);
  a.rect = rho_t > 5.0
               ? (float)sqrt((rho_t - 4.0) * (rho_t - 2.0) * rho_inf / ((rho_inf - 4.0) * (rho_inf - 2.0) * rho_t))
               : -1.0f;
  int bx = (max_n + 255) / 256;
  if (bx > 256) bx = 256;
  if (bx < 1) bx = 1;
  hipLaunchKernelGGL(rsn_radam_kernel, dim3(bx, n_tensors), dim3(256), 0, (hipStream_t)stream, a);
  RSN_HIP(hipGetLastError());
  return RSN_OK;
}
