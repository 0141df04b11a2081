// rsn_field_bwd_common.h -- argument blocks and per-sample gradient helpers shared by the backward sweeps:
// rsn_field_bwd.hip (per-wave weight stream: exact fp32, split-bf16, plain bf16 at widths 64 / 128) and
// rsn_field_bf16_train.hip (plain bf16 at width 256 on the LDS weight ring).
#pragma once
#include "rsn_field_common.h"

struct BwdShared {
  const float* packed;
  RsnPackedLayout L;
  int num_layers, skip_layer;
  float density_bias;
  float freqs[RSN_NUM_FREQS];
};

struct BwdJob {
  int mode, n_rays, S, need_input_grad;
  const int* n_dev;
  const float* origins;
  const float* directions;
  const float* pixel_area;
  const float* bins;
  const float* sqradius;
  rsn_field_grads_in gin;
  rsn_field_outputs fwd;   // forward per-sample values: raw_density, diff, tint
  rsn_field_saved saved;
  rsn_field_grads_out gout;
  long long act_stride;
};

struct BwdArgs : BwdShared, BwdJob {};

// Several evaluations in one launch (see FieldJobs, rsn_field_common.h): the backward sweeps of the two reflect levels and
// of get_inf_color are independent of each other once the compositing backward of both levels has run.
struct BwdJobs {
  BwdShared s;
  int n_jobs;
  BwdJob j[RSN_MAX_JOBS];
};

// d var_c / d pixel_area for a conical-frustum sample after contraction (the mean does not depend on pixel_area):
// var_c = relu(diag(J Sigma J))_c, Sigma = var_t d d^T + var_r (I - d (d/|d|^2)^T), var_r = (pa / 1.7724538509^2) Kr(t).
__device__ __forceinline__ void frustum_dvar_dpa(const float o[3], const float d[3], float pa, float t0, float t1,
                                                 float out[3]) {
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
  const float kr = mu2 / 4.0f + 0.4166666666666667f * hw2 - (0.26666666666666666f * hw4) / den;
  const float var_r = (radius * radius) * kr;
  const float dmag = fmaxf(d[0] * d[0] + d[1] * d[1] + d[2] * d[2], 1e-10f);
  float S[3][3], Nn[3][3];
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      Nn[i][j] = (i == j ? 1.0f : 0.0f) - d[i] * (d[j] / dmag);
      S[i][j] = var_t * (d[i] * d[j]) + var_r * Nn[i][j];
    }
  const float n2 = mean[0] * mean[0] + mean[1] * mean[1] + mean[2] * mean[2];
  const float n = sqrtf(n2);
  float J[3][3];
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      const float eye = (i == j) ? 1.0f : 0.0f;
      J[i][j] = (n > 1.0f) ? ((2.0f * n - 2.0f) * (eye - mean[i] * mean[j] / n2) + eye) / n2 : eye;
    }
  // d(radius^2)/d(pa) = 1 / 1.7724538509^2
  const float dr2 = kr / (1.7724538509055159f * 1.7724538509055159f);
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    float vs = 0.0f, vn = 0.0f;
#pragma unroll
    for (int b = 0; b < 3; ++b) {
      const float js = J[i][0] * S[0][b] + J[i][1] * S[1][b] + J[i][2] * S[2][b];
      const float jn = J[i][0] * Nn[0][b] + J[i][1] * Nn[1][b] + J[i][2] * Nn[2][b];
      vs += js * J[b][i];
      vn += jn * J[b][i];
    }
    out[i] = vs > 0.0f ? vn * dr2 : 0.0f;  // relu on the diagonal (reflect_sampling_nerf_field.py:114-115)
  }
}

// F.normalize backward: y = x / max(|x|, eps);  g_x = (g_y - y (y . g_y)) / |x|
__device__ __forceinline__ void normalize_bwd(const float x[3], const float gy[3], float gx[3]) {
  const float len = fmaxf(sqrtf(x[0] * x[0] + x[1] * x[1] + x[2] * x[2]), 1e-12f);
  const float y0 = x[0] / len, y1 = x[1] / len, y2 = x[2] / len;
  const float dot = y0 * gy[0] + y1 * gy[1] + y2 * gy[2];
  gx[0] = (gy[0] - y0 * dot) / len;
  gx[1] = (gy[1] - y1 * dot) / len;
  gx[2] = (gy[2] - y2 * dot) / len;
}


// rsn_field_bf16_train.hip: the plain-bf16 training kernels on the LDS weight ring (width 256; rsn_ring_training())
int rsn_launch_field_bf16_train(long long n_tiles256, hipStream_t st, const FieldJobs& J);
int rsn_launch_field_bf16_bwd(long long n_tiles256, hipStream_t st, const BwdJobs& J);
// rsn_field_x6_train.hip: the split-bf16 (fp32-equivalent) training kernels on the LDS weight ring (width 256; 128-point tiles)
int rsn_launch_field_x6_train(long long n_tiles128, hipStream_t st, const FieldJobs& J);
int rsn_launch_field_x6_bwd(long long n_tiles128, hipStream_t st, const BwdJobs& J);
// tools/probes/rsn_field_f32_ring.hip (diagnostic builds with -DRSN_F32_RING_TRAIN): the exact-fp32 training forward on the LDS ring
int rsn_launch_field_f32_train(long long n_tiles128, hipStream_t st, const FieldJobs& J);
