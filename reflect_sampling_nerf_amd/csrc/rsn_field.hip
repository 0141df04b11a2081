// rsn_field.hip -- the dominant kernel: fused per-sample field evaluation on gfx950.
//
// One wavefront owns a tile of 32 sample points for the whole network.  With
// v_mfma_f32_32x32x2_f32 computing D[n][m] = sum_k W[n][k] * X[m][k] (A = weights, B = points) the
// accumulator of lane (m = lane&31, h = lane>>5) holds, for ITS point m, output features
// nb*32 + 8q + 4h + j (register 4q+j).  The K index of the next layer is a free permutation, and the
// packed weights (rsn_pack.hip) are laid out so that K-iteration `it` (8 features) consumes from
// lane (m,h) exactly features it*8 + 4h + {0..3}: the four registers that lane already holds.
// So activations never cross lanes between layers: each lane parks its float4's in a wave-private
// LDS slab X[it][lane] (ds_write_b128 / ds_read_b128, conflict-free, no barriers) purely so that the
// K loop can index them dynamically.  Weights stream L2 -> VGPR as 1 KiB-contiguous
// global_load_dwordx4 per (it, nb), double-buffered one K-iteration (2048 MFMA cycles) ahead.
//
// Roofline: MFMA-bound.  1,230,592 algorithmic FLOP per sample at W=256, L=8;
// the padded instruction stream issues 9,808 MFMAs per 32-point tile vs 9,614 algorithmic.
//
// Reference semantics restated here: reflect_sampling_nerf_field.py:90-207,
// reflect_sampling_nerf_components.py:52-140 and the nerfstudio primitives N1-N3, N6.
#define RSN_FIELD_MAIN_TU
#include "rsn_field_kernel.h"
#include "rsn_field_bwd_common.h"

#ifdef RSN_PHASE_TIMERS
extern "C" int rsn_debug_phase_cycles(unsigned long long* out16, int reset) {
  if (out16) RSN_HIP(hipMemcpyFromSymbol(out16, HIP_SYMBOL(rsn_phase_cycles), sizeof(unsigned long long) * 16));
  if (reset) {
    unsigned long long z[16] = {0};
    RSN_HIP(hipMemcpyToSymbol(HIP_SYMBOL(rsn_phase_cycles), z, sizeof(z)));
  }
  return RSN_OK;
}
#endif

// ------------------------------------------------------------------------------------------------
// One launch over n evaluations (all training or all eval; the plain-bf16 eval kernels take one).
static int launch_field_jobs(const rsn_field_desc* d, FieldArgs* js, int n, void* stream) {
  RSN_REQUIRE(n >= 1 && n <= RSN_MAX_JOBS, RSN_ERR_INVALID_ARGUMENT, "n_jobs=%d (1..%d)", n, RSN_MAX_JOBS);
  FieldJobs J = {};
  int rc = rsn_compute_layout(d, &J.s.L);
  if (rc != RSN_OK) return rc;
  RSN_REQUIRE(js[0].packed != nullptr, RSN_ERR_INVALID_ARGUMENT, "packed weights pointer is NULL");
  J.s.packed = js[0].packed;
  J.s.num_layers = d->num_layers;
  J.s.skip_layer = d->skip_layer;
  J.s.width = d->width;
  J.s.density_bias = d->density_bias;
  for (int i = 0; i < RSN_NUM_FREQS; ++i) J.s.freqs[i] = d->freqs[i];
  const bool train = js[0].saved.act != nullptr || js[0].saved.enc != nullptr || js[0].saved.heads != nullptr;
  long long n_tiles = 0;
  for (int k = 0; k < n; ++k) {
    FieldArgs& a = js[k];
    if (a.n_rays <= 0) continue;
    const bool tk = a.saved.act != nullptr || a.saved.enc != nullptr || a.saved.heads != nullptr;
    RSN_REQUIRE(tk == train, RSN_ERR_INVALID_ARGUMENT, "job %d: training and eval evaluations cannot share a launch", k);
    const long long n_points = (long long)a.n_rays * a.S;
    a.act_stride = n_points * (long long)d->width;
    n_tiles += (n_points + 127) / 128;
    J.j[J.n_jobs++] = static_cast<const FieldJob&>(a);
  }
  if (J.n_jobs == 0) return RSN_OK;
  const int cus = rsn_device_cus();
  // one 4-wave workgroup per CU (one wave per SIMD, LDS slab 148 KiB at W=256): persistent tiles
  const long long grid = n_tiles < (long long)cus ? n_tiles : (long long)cus;
  hipStream_t st = (hipStream_t)stream;
  const int mode = d->mma_mode;
  if (!train && mode == RSN_MMA_BF16) {  // plain bf16 operands: its own kernel, two workgroups per CU
    RSN_REQUIRE(J.n_jobs == 1, RSN_ERR_UNSUPPORTED, "the plain-bf16 eval kernels take one evaluation per launch");
    FieldArgs one = {};
    static_cast<FieldShared&>(one) = J.s;
    static_cast<FieldJob&>(one) = J.j[0];
    const long long g2 = n_tiles < 2LL * cus ? n_tiles : 2LL * cus;
    return rsn_launch_field_bf16(d->width, g2, st, one);
  }
  if (train && rsn_ring_training(d)) {  // plain / split bf16 training at width 256: the LDS-ring kernels (256- / 128-point tiles)
    const int tp = mode == RSN_MMA_BF16X6 ? 128 : 256;
    long long tn = 0;
    for (int k = 0; k < J.n_jobs; ++k) tn += ((long long)J.j[k].n_rays * J.j[k].S + tp - 1) / tp;
    return mode == RSN_MMA_BF16X6 ? rsn_launch_field_x6_train(tn, st, J) : rsn_launch_field_bf16_train(tn, st, J);
  }
#ifdef RSN_F32_RING_TRAIN  // (diagnostic builds only: the ring form of the exact-fp32 training forward)
  if (train && rsn_f32_ring_training(d)) {  // 128-point tiles
    long long tn = 0;
    for (int k = 0; k < J.n_jobs; ++k) tn += ((long long)J.j[k].n_rays * J.j[k].S + 127) / 128;
    return rsn_launch_field_f32_train(tn, st, J);
  }
#endif
  if (mode == RSN_MMA_BF16X6 || (!train && mode == RSN_MMA_BF16X3)) {  // split-bf16 instantiations: rsn_field_split.hip
    rc = rsn_launch_field_split(d->width, train, mode == RSN_MMA_BF16X6 ? 1 : 2, grid, st, J);
    if (rc != RSN_OK) return rc;
    RSN_HIP(hipGetLastError());
    return RSN_OK;
  }
#define RSN_LAUNCH(NBV)                                                                                          \
  do {                                                                                                           \
    if (train && mode == RSN_MMA_BF16)  /* reduced-precision training: plain bf16 operands, fp32 accumulate */  \
      hipLaunchKernelGGL((rsn_field_kernel<NBV, true, 3>), dim3((unsigned)grid), dim3(256), 0, st, J);            \
    else if (train)  /* BF16X3 is an eval-only opt-in: training falls back to exact fp32 */                      \
      hipLaunchKernelGGL((rsn_field_kernel<NBV, true, 0>), dim3((unsigned)grid), dim3(256), 0, st, J);            \
    else                                                                                                         \
      hipLaunchKernelGGL((rsn_field_kernel<NBV, false, 0>), dim3((unsigned)grid), dim3(256), 0, st, J);           \
  } while (0)
  switch (d->width) {
    case 256: RSN_LAUNCH(8); break;
    case 128: RSN_LAUNCH(4); break;
    case 64: RSN_LAUNCH(2); break;
    default: RSN_REQUIRE(false, RSN_ERR_UNSUPPORTED, "width=%d unsupported", d->width);
  }
#undef RSN_LAUNCH
  RSN_HIP(hipGetLastError());
  return RSN_OK;
}

static int launch_field(const rsn_field_desc* d, FieldArgs& a, void* stream) { return launch_field_jobs(d, &a, 1, stream); }

// rsn_field_forward_train_jobs: several training-mode evaluations of the SAME field in one launch.
extern "C" int rsn_field_forward_train_jobs(const rsn_field_desc* desc, const float* packed, int32_t n_jobs,
                                            const rsn_field_job* jobs, void* stream) {
  RSN_REQUIRE(desc && jobs, RSN_ERR_INVALID_ARGUMENT, "desc/jobs is NULL");
  RSN_REQUIRE(n_jobs >= 1 && n_jobs <= RSN_MAX_JOBS, RSN_ERR_INVALID_ARGUMENT, "n_jobs=%d (1..%d)", n_jobs, RSN_MAX_JOBS);
  FieldArgs js[RSN_MAX_JOBS] = {};
  for (int k = 0; k < n_jobs; ++k) {
    const rsn_field_job& q = jobs[k];
    FieldArgs& a = js[k];
    RSN_REQUIRE(q.kind == 0 || q.kind == 1, RSN_ERR_INVALID_ARGUMENT, "job %d: kind=%d", k, q.kind);
    RSN_REQUIRE(q.n_rays >= 0 && q.saved, RSN_ERR_INVALID_ARGUMENT, "job %d: n_rays=%d / saved is NULL", k, q.n_rays);
    RSN_REQUIRE(q.saved->act && q.saved->enc && q.saved->bott && q.saved->sh && q.saved->hid && q.saved->heads &&
                    q.saved->relu_bits,
                RSN_ERR_INVALID_ARGUMENT, "job %d: training needs every saved-activation buffer (normals may be NULL)", k);
    a.packed = packed;
    a.n_rays = q.n_rays; a.n_dev = q.n_dev;
    a.saved = *q.saved;
    if (q.kind == 0) {
      RSN_REQUIRE(q.n_samples >= 1 && q.out, RSN_ERR_INVALID_ARGUMENT, "job %d: n_samples=%d / out is NULL", k, q.n_samples);
      RSN_REQUIRE(q.n_rays == 0 || (q.origins && q.directions && q.pixel_area && q.euclid_bins), RSN_ERR_INVALID_ARGUMENT,
                  "job %d: a ray input pointer is NULL", k);
      a.mode = RSN_MODE_FRUSTUM; a.S = q.n_samples;
      a.origins = q.origins; a.directions = q.directions; a.pixel_area = q.pixel_area; a.bins = q.euclid_bins;
      a.out = *q.out;
    } else {
      RSN_REQUIRE(q.n_rays == 0 || (q.directions && q.sqradius && q.out_rgb), RSN_ERR_INVALID_ARGUMENT,
                  "job %d: an input pointer is NULL", k);
      a.mode = RSN_MODE_INF; a.S = 1;
      a.directions = q.directions; a.sqradius = q.sqradius;
      a.out.color = q.out_rgb;
      a.saved.normals = nullptr;
    }
  }
  return launch_field_jobs(desc, js, n_jobs, stream);
}

extern "C" int rsn_field_forward_gaussians_train(const rsn_field_desc* desc, const float* packed, int32_t n_points,
                                                 const float* means, const float* cov_diag, const float* view_dirs,
                                                 const rsn_field_outputs* out, float* embedding,
                                                 const rsn_field_saved* saved, void* stream) {
  RSN_REQUIRE(desc && out && saved, RSN_ERR_INVALID_ARGUMENT, "desc/out/saved is NULL");
  RSN_REQUIRE(n_points >= 0, RSN_ERR_INVALID_ARGUMENT, "n_points=%d", n_points);
  RSN_REQUIRE(n_points == 0 || means, RSN_ERR_INVALID_ARGUMENT, "means is NULL");
  RSN_REQUIRE(desc->mma_mode == RSN_MMA_F32, RSN_ERR_UNSUPPORTED,
              "training-mode evaluation of explicit Gaussians runs on the exact-fp32 kernels only (mma_mode %d)", desc->mma_mode);
  RSN_REQUIRE(saved->act && saved->enc && saved->bott && saved->sh && saved->hid && saved->heads && saved->relu_bits,
              RSN_ERR_INVALID_ARGUMENT, "training needs every saved-activation buffer (normals may be NULL)");
  FieldArgs a = {};
  a.packed = packed;
  a.mode = RSN_MODE_GAUSS;
  a.n_rays = n_points; a.n_dev = nullptr; a.S = 1;
  a.means = means; a.cov_diag = cov_diag; a.view_dirs = view_dirs;
  a.out = *out;
  a.embedding = embedding;
  a.saved = *saved;
  return launch_field(desc, a, stream);
}

extern "C" int rsn_field_forward_frustum(const rsn_field_desc* desc, const float* packed, int32_t n_rays,
                                         const int32_t* n_dev, int32_t n_samples, const float* origins,
                                         const float* directions, const float* pixel_area, const float* euclid_bins,
                                         const rsn_field_outputs* out, void* stream) {
  RSN_REQUIRE(desc && out, RSN_ERR_INVALID_ARGUMENT, "desc/out is NULL");
  RSN_REQUIRE(n_rays >= 0 && n_samples >= 1, RSN_ERR_INVALID_ARGUMENT, "n_rays=%d n_samples=%d", n_rays, n_samples);
  RSN_REQUIRE(n_rays == 0 || (origins && directions && pixel_area && euclid_bins), RSN_ERR_INVALID_ARGUMENT,
              "a ray input pointer is NULL");
  FieldArgs a = {};
  a.packed = packed;
  a.mode = RSN_MODE_FRUSTUM;
  a.n_rays = n_rays; a.n_dev = n_dev; a.S = n_samples;
  a.origins = origins; a.directions = directions; a.pixel_area = pixel_area; a.bins = euclid_bins;
  a.out = *out;
  return launch_field(desc, a, stream);
}

extern "C" int rsn_field_forward_frustum_train(const rsn_field_desc* desc, const float* packed, int32_t n_rays,
                                               const int32_t* n_dev, int32_t n_samples, const float* origins,
                                               const float* directions, const float* pixel_area,
                                               const float* euclid_bins, const rsn_field_outputs* out,
                                               const rsn_field_saved* saved, void* stream) {
  RSN_REQUIRE(desc && out && saved, RSN_ERR_INVALID_ARGUMENT, "desc/out/saved is NULL");
  RSN_REQUIRE(n_rays >= 0 && n_samples >= 1, RSN_ERR_INVALID_ARGUMENT, "n_rays=%d n_samples=%d", n_rays, n_samples);
  RSN_REQUIRE(n_rays == 0 || (origins && directions && pixel_area && euclid_bins), RSN_ERR_INVALID_ARGUMENT,
              "a ray input pointer is NULL");
  RSN_REQUIRE(saved->act && saved->enc && saved->bott && saved->sh && saved->hid && saved->heads && saved->relu_bits,
              RSN_ERR_INVALID_ARGUMENT, "training needs every saved-activation buffer (normals may be NULL)");
  FieldArgs a = {};
  a.packed = packed;
  a.mode = RSN_MODE_FRUSTUM;
  a.n_rays = n_rays; a.n_dev = n_dev; a.S = n_samples;
  a.origins = origins; a.directions = directions; a.pixel_area = pixel_area; a.bins = euclid_bins;
  a.out = *out;
  a.saved = *saved;
  return launch_field(desc, a, stream);
}

extern "C" int rsn_field_forward_inf(const rsn_field_desc* desc, const float* packed, int32_t n_rays,
                                     const int32_t* n_dev, const float* directions, const float* sqradius,
                                     float* out_rgb, void* stream) {
  RSN_REQUIRE(desc, RSN_ERR_INVALID_ARGUMENT, "desc is NULL");
  RSN_REQUIRE(n_rays >= 0, RSN_ERR_INVALID_ARGUMENT, "n_rays=%d", n_rays);
  RSN_REQUIRE(n_rays == 0 || (directions && sqradius && out_rgb), RSN_ERR_INVALID_ARGUMENT, "an input pointer is NULL");
  FieldArgs a = {};
  a.packed = packed;
  a.mode = RSN_MODE_INF;
  a.n_rays = n_rays; a.n_dev = n_dev; a.S = 1;
  a.directions = directions; a.sqradius = sqradius;
  a.out.color = out_rgb;
  return launch_field(desc, a, stream);
}

extern "C" int rsn_field_forward_inf_train(const rsn_field_desc* desc, const float* packed, int32_t n_rays,
                                           const int32_t* n_dev, const float* directions, const float* sqradius,
                                           float* out_rgb, const rsn_field_saved* saved, void* stream) {
  RSN_REQUIRE(desc && saved, RSN_ERR_INVALID_ARGUMENT, "desc/saved is NULL");
  RSN_REQUIRE(n_rays >= 0, RSN_ERR_INVALID_ARGUMENT, "n_rays=%d", n_rays);
  RSN_REQUIRE(n_rays == 0 || (directions && sqradius && out_rgb), RSN_ERR_INVALID_ARGUMENT, "an input pointer is NULL");
  RSN_REQUIRE(saved->act && saved->enc && saved->bott && saved->sh && saved->hid && saved->heads && saved->relu_bits,
              RSN_ERR_INVALID_ARGUMENT, "training needs every saved-activation buffer");
  FieldArgs a = {};
  a.packed = packed;
  a.mode = RSN_MODE_INF;
  a.n_rays = n_rays; a.n_dev = n_dev; a.S = 1;
  a.directions = directions; a.sqradius = sqradius;
  a.out.color = out_rgb;
  a.saved = *saved;
  a.saved.normals = nullptr;
  return launch_field(desc, a, stream);
}

extern "C" int rsn_field_forward_embedding(const rsn_field_desc* desc, const float* packed, int32_t n_points,
                                           const float* embedding, const float* view_dirs, const float* roughness,
                                           const rsn_field_outputs* out, void* stream) {
  RSN_REQUIRE(desc && out, RSN_ERR_INVALID_ARGUMENT, "desc/out is NULL");
  RSN_REQUIRE(n_points >= 0, RSN_ERR_INVALID_ARGUMENT, "n_points=%d", n_points);
  RSN_REQUIRE(n_points == 0 || embedding, RSN_ERR_INVALID_ARGUMENT, "embedding is NULL");
  FieldArgs a = {};
  a.packed = packed;
  a.mode = RSN_MODE_EMB;
  a.n_rays = n_points; a.n_dev = nullptr; a.S = 1;
  a.emb_in = embedding; a.view_dirs = view_dirs; a.rough_in = roughness;
  a.out = *out;
  return launch_field(desc, a, stream);
}

extern "C" int rsn_field_forward_gaussians(const rsn_field_desc* desc, const float* packed, int32_t n_points,
                                           const float* means, const float* cov_diag, const float* view_dirs,
                                           const rsn_field_outputs* out, float* embedding, void* stream) {
  RSN_REQUIRE(desc && out, RSN_ERR_INVALID_ARGUMENT, "desc/out is NULL");
  RSN_REQUIRE(n_points >= 0, RSN_ERR_INVALID_ARGUMENT, "n_points=%d", n_points);
  RSN_REQUIRE(n_points == 0 || means, RSN_ERR_INVALID_ARGUMENT, "means is NULL");
  FieldArgs a = {};
  a.packed = packed;
  a.mode = RSN_MODE_GAUSS;
  a.n_rays = n_points; a.n_dev = nullptr; a.S = 1;
  a.means = means; a.cov_diag = cov_diag; a.view_dirs = view_dirs;
  a.out = *out;
  a.embedding = embedding;
  return launch_field(desc, a, stream);
}
