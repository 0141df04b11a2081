// rsn_field_common.h -- argument block and geometry helpers shared by the field kernels (rsn_field.hip: exact-fp32 /
// split-bf16 / training; rsn_field_bf16.hip: the dedicated plain-bf16 eval kernel).
#pragma once
#include "rsn_mfma.h"

// What every evaluation of one launch shares (the network) ...
struct FieldShared {
  const float* packed;
  RsnPackedLayout L;
  int num_layers, skip_layer, width;
  float density_bias;
  float freqs[RSN_NUM_FREQS];
};

// ... and what one evaluation ("job") brings: its points, inputs and outputs.
struct FieldJob {
  int mode;
  int n_rays;          // rays (frustum / inf) or points (gauss)
  const int* n_dev;    // optional device-side ray count
  int S;               // samples per ray (1 for inf / gauss)
  const float* origins;
  const float* directions;
  const float* pixel_area;
  const float* bins;
  const float* sqradius;
  const float* means;
  const float* cov_diag;
  const float* view_dirs;
  rsn_field_outputs out;
  float* embedding;
  const float* emb_in;        // RSN_MODE_EMB: [N,W] embedding (post-ReLU trunk output) supplied by the caller
  const float* rough_in;      // RSN_MODE_EMB: optional explicit roughness for the SH attenuation (get_mid's argument)
  rsn_field_saved saved;      // training: activations kept for the backward pass (all NULL in eval)
  long long act_stride;       // floats between consecutive layers in saved.act (= n_points_max * W)
};

// one evaluation per launch (the plain-bf16 eval kernels of rsn_field_bf16.hip)
struct FieldArgs : FieldShared, FieldJob {
  int stagger;                // rsn_field_bf16_ring_kernel: start delay of the second workgroup per CU (x s_sleep 127)
};

// Several evaluations in ONE launch of rsn_field_kernel: the tiles of job 0, then job 1, ... form one tile space that
// the persistent workgroups stride through.  The reflect branch of a training step has three evaluations of a few
// hundred to ~1,250 tiles each on 256 workgroups; launched one by one, each wastes its last partial round of tiles
// (0.3-0.5 ms a round) and get_inf_color's ~20 tiles occupy 20 CUs for a whole launch.
#define RSN_MAX_JOBS 3
struct FieldJobs {
  FieldShared s;
  int n_jobs;
  FieldJob j[RSN_MAX_JOBS];
};

// Conical frustum -> Gaussian (nerfstudio conical_frustum_to_gaussian / compute_3d_gaussian, N3),
// followed by the reference's contraction (reflect_sampling_nerf_field.py:98-119).  Only the diagonal
// of J Sigma J is consumed downstream (N2), so only that is formed.
__device__ __forceinline__ void frustum_to_contracted(const float o[3], const float d[3], float pa, float t0, float t1,
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
  // Sigma = var_t d d^T + var_r (I - d (d/dmag)^T)
  float S[3][3];
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int j = 0; j < 3; ++j)
      S[i][j] = var_t * (d[i] * d[j]) + var_r * ((i == j ? 1.0f : 0.0f) - d[i] * (d[j] / dmag));
  // contraction
  const float n2 = mean[0] * mean[0] + mean[1] * mean[1] + mean[2] * mean[2];
  const float n = sqrtf(n2);
  if (n > 1.0f) {
    const float sc = (2.0f * n - 1.0f) / n2;
    float J[3][3];
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
      for (int j = 0; j < 3; ++j) {
        const float eye = (i == j) ? 1.0f : 0.0f;
        const float outer = mean[i] * mean[j] / n2;
        J[i][j] = ((2.0f * n - 2.0f) * (eye - outer) + eye) / n2;
      }
#pragma unroll
    for (int c = 0; c < 3; ++c) mean_c[c] = sc * mean[c];
    // diag(J S J)
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      float acc = 0.0f;
#pragma unroll
      for (int b = 0; b < 3; ++b) {
        const float js = J[i][0] * S[0][b] + J[i][1] * S[1][b] + J[i][2] * S[2][b];
        acc += js * J[b][i];
      }
      var_c[i] = fmaxf(acc, 0.0f);
    }
  } else {
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      mean_c[c] = mean[c];
      var_c[c] = fmaxf(S[c][c], 0.0f);
    }
  }
}


// rsn_field_bf16.hip: the dedicated RSN_MMA_BF16 eval kernel (two workgroups per CU)
int rsn_launch_field_bf16(int width, long long grid, hipStream_t st, const FieldArgs& a);
// split-bf16 instantiations of rsn_field_kernel (rsn_field_split.hip); mode 1 = BF16X6, 2 = BF16X3 (eval only)
int rsn_launch_field_split(int width, bool train, int mode, long long grid, hipStream_t st, const FieldJobs& J);
