/*
 * rsn.h -- C ABI of librsn_hip.so: the MI355X (gfx950) implementation of the
 * reflect-sampling-nerf ray-batch volume-rendering hot path.
 *
 * The reference (236088/reflect-sampling-nerf) is pure Python on PyTorch and has NO FFI for this
 * path; its boundary is the Nerfstudio Model/Field plugin surface.  This header
 * is the boundary the build introduces underneath that surface: each entry point below replaces a
 * group of reference calls, cited as file:line relative to /root/reference/reflect_sampling_nerf/.
 * The Python host mirror (reflect_sampling_nerf_amd/) binds these with ctypes; INTEGRATION.md shows
 * the binding a reference maintainer would add.
 *
 * Conventions
 *   - plain C types only; every pointer is a DEVICE pointer to fp32 (or int32) unless it says host;
 *   - no allocation and no ownership transfer inside the library: the caller (torch) owns inputs,
 *     outputs and workspaces;
 *   - every call is asynchronous on the given hipStream_t (passed as void*), never synchronises
 *     the host, and is re-entrant for distinct streams;
 *   - return 0 on success, a negative rsn_status otherwise; rsn_last_error() gives the message
 *     (thread-local);
 *   - "n_dev": optional device pointer to an int32 ray count that overrides the host count at run
 *     time (dynamic number M of reflected rays, no host sync); pass NULL to use the host count.
 */
#ifndef RSN_H
#define RSN_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define RSN_ABI_VERSION 16
#define RSN_MAX_TRUNK_LAYERS 16
#define RSN_NUM_FREQS 16   /* NeRFEncoding(num_frequencies=16), reflect_sampling_nerf_model.py:98-100 */
#define RSN_ENC_DIM 99     /* 3*16*2 + 3 */
#define RSN_SH_DIM 34      /* IntegratedSHEncoding.get_out_dim, reflect_sampling_nerf_components.py:49-50 */

typedef enum rsn_status {
  RSN_OK = 0,
  RSN_ERR_INVALID_ARGUMENT = -1,
  RSN_ERR_UNSUPPORTED = -2,
  RSN_ERR_HIP = -3,
  RSN_ERR_WORKSPACE = -4
} rsn_status;

typedef enum rsn_spacing { RSN_SPACING_UNIFORM = 0, RSN_SPACING_RECIPROCAL = 1 } rsn_spacing;

/* Shape of the Field: constructor knobs of ReflectSamplingNeRFNerfField
 * (reflect_sampling_nerf_field.py:36-47) plus the frequency table of the position encoding the
 * model builds for it (reflect_sampling_nerf_model.py:98-100: 2**linspace(0,16,16), computed by the
 * caller so that it is bit-identical to the host framework's table). */
typedef struct rsn_field_desc {
  int32_t num_layers;   /* trunk nn.Linear count (base_mlp_num_layers), 2..RSN_MAX_TRUNK_LAYERS */
  int32_t width;        /* base_mlp_layer_width: 64, 128 or 256 */
  int32_t skip_layer;   /* layer whose input is cat([encoding, x]) (4 when num_layers >= 6), or -1 */
  int32_t mid_width;    /* head_mlp_layer_width: 128 */
  float density_bias;   /* 0.5 */
  float freqs[RSN_NUM_FREQS];
  int32_t mma_mode;     /* arithmetic of the dense GEMMs in the eval field kernel (rsn_mma_mode) */
  int32_t param_width;  /* base_mlp_layer_width of the PARAMETER tensors when it is not one of the widths the kernels run at:
                         * any 1 <= param_width <= width; 0 = width.  The kernels run at `width` (the next of 64 / 128 / 256)
                         * with zero-padded units: rsn_pack_weights gives units param_width .. width - 1 zero weights and
                         * biases, their activations and gradients are exact zeros, every wide buffer ([N, W] rows) has
                         * `width` columns of which the first param_width are live (ABI 15) */
} rsn_field_desc;

/* How the field kernel multiplies fp32 operands on the matrix cores:
 *   RSN_MMA_F32     v_mfma_f32_32x32x2_f32: exact fp32 products (157 TFLOP/s peak);
 *   RSN_MMA_BF16X6  fp32 emulation: both operands split exactly into 3 bf16 (8+8+8 mantissa bits), the 6 leading
 *                   cross products on v_mfma_f32_32x32x16_bf16 with fp32 accumulation (dropped terms <= 2^-24
 *                   relative): fp32-equivalent results at 6/16 of the fp32-MFMA cost;
 *   RSN_MMA_BF16X3  2-way split, 3 products (~2^-16 relative): reduced precision, opt-in only;
 *   RSN_MMA_BF16    plain bf16 operands (1 product, 8 mantissa bits), fp32 accumulate: BASELINE configs[3]
 *                   ("bf16 MFMA hidden GEMMs"); encode, heads' activations and compositing stay fp32. */
typedef enum rsn_mma_mode { RSN_MMA_F32 = 0, RSN_MMA_BF16X6 = 1, RSN_MMA_BF16X3 = 2, RSN_MMA_BF16 = 3 } rsn_mma_mode;

/* Parameters in torch.nn.Linear layout: weight [out,in] row-major, bias [out]; names follow the
 * reference Field's state_dict (reflect_sampling_nerf_field.py:54-86).  field_output_low (:67) is
 * never evaluated by the model and is therefore absent. */
typedef struct rsn_field_params {
  const float* trunk_w[RSN_MAX_TRUNK_LAYERS]; /* mlp_base.layers.{i}.weight */
  const float* trunk_b[RSN_MAX_TRUNK_LAYERS]; /* mlp_base.layers.{i}.bias   */
  const float* density_w;    const float* density_b;    /* field_output_density.net    [1,W]      */
  const float* normals_w;    const float* normals_b;    /* field_output_normals.net    [3,W]      */
  const float* roughness_w;  const float* roughness_b;  /* field_output_roughness.net  [1,W]      */
  const float* diff_w;       const float* diff_b;       /* field_output_diff.net       [3,W]      */
  const float* tint_w;       const float* tint_b;       /* field_output_tint.net       [3,W]      */
  const float* bottleneck_w; const float* bottleneck_b; /* field_output_bottleneck.net [W,W]      */
  const float* mid_w;        const float* mid_b;        /* mlp_mid.layers.0            [mid,34+W] */
  const float* rgb_w;        const float* rgb_b;        /* field_output_mid.net        [3,mid]    */
} rsn_field_params;

/* Per-sample outputs of one field evaluation; any pointer may be NULL (output skipped). */
typedef struct rsn_field_outputs {
  float* sigma;        /* [N]    softplus(raw + density_bias)         field.py:133-136 */
  float* color;        /* [N,3]  diff + tint * mid                    model.py:175,209,310,336 */
  float* pred_normals; /* [N,3]  get_pred_normals                     field.py:139-144 */
  float* n_dot_d;      /* [N]    get_reflection's n.d                 field.py:203-204 */
  float* diff;         /* [N,3]  get_diff                             field.py:176-180 */
  float* tint;         /* [N,3]  get_tint                             field.py:182-186 */
  float* roughness;    /* [N]    sigmoid(roughness head)              field.py:150-155 (default act) */
  float* raw_density;  /* [N]    density head before bias/softplus    field.py:133-135 */
  float* raw_roughness;/* [N]    roughness head before its activation field.py:150-155 (caller-chosen act) */
} rsn_field_outputs;

/* Activations the training-mode forward keeps for the backward pass (row-major, N = n_rays*n_samples rows; W = width,
 * L = num_layers).  Written by rsn_field_forward_frustum_train.  fp32 -- except in the reduced-precision training mode
 * (rsn_field_desc.mma_mode == RSN_MMA_BF16, whose GEMMs round these values to bf16 anyway): there the three WIDE buffers
 * act, bott and hid hold bf16 (same shapes, 2 bytes per element; half the step's HBM stream), and so do the wide layer
 * gradients dy, d_bott, da_mid of rsn_field_grads_out; rsn_weight_grad_multi_dev reads them as such (operand_bf16). */
typedef struct rsn_field_saved {
  float* enc;     /* [N,104]  encoded inputs (input of trunk layer 0), kernel slot order                   */
  float* act;     /* [L,N,W]  post-ReLU output of trunk layer l (act[L-1] = embedding)                       */
  float* bott;    /* [N,W]    bottleneck output                                                              */
  float* sh;      /* [N,40]   attenuated SH-34 inputs of mlp_mid, kernel slot order                          */
  float* hid;     /* [N,128]  mlp_mid hidden (post-ReLU)                                                     */
  float* heads;   /* [N,8]    raw normal head (3), raw roughness head (1), mid RGB (3), pad                  */
  float* normals; /* [N,3]    OUT: analytic normals -normalize(d raw_density/d mean) (field.py:146-147),     *
                   *          or NULL to skip the sweep (reflect levels, model.py:295,321)                    */
  uint32_t* relu_bits; /* [L+1,N,2,max(W/64,2)] ReLU masks (pre-activation > 0) of trunk layer l (l < L) and of the mlp_mid *
                   *          hidden layer (l = L), bit-packed per lane half: what the dX sweeps mask by            */
} rsn_field_saved;

int rsn_abi_version(void);
const char* rsn_last_error(void);

/* Layout of the two NARROW saved buffers of a training forward, which depends on the kernel that serves the Field's shape and
 * MMA mode (ABI 15).  Default (rsn_field_kernel): enc = fp32 [N,104], sh = fp32 [N,40] in that kernel's slot order.  With
 * mma_mode == RSN_MMA_BF16 at width 256 the training forward / backward run on the LDS weight ring
 * (rsn_field_bf16_train.hip): enc = bf16 [N,128], sh = bf16 [N,64], slot s = 32 kk + 8 g + e of lane group g, and the
 * ReLU bit words of rsn_field_saved.relu_bits are laid out [L+1][N][4 lane groups][2 words].  mma_mode == RSN_MMA_BF16X6 at
 * width 256 (rsn_field_x6_train.hip, the split-bf16 = fp32-equivalent mode on the same ring): the same slot order and bit
 * layout, enc / sh rows in fp32 (narrow_bf16 = 0); every wide buffer stays fp32 [N,W] in natural feature order.
 * Outputs: *enc_cols / *sh_cols = row length in elements, *narrow_bf16 = 1 if enc / sh rows hold bf16; enc_map[s] / sh_map[s]
 * (caller arrays of 128 / 64 ints, HOST) = column of the reference's 99-wide NeRFEncoding output / 34-wide SH encoding
 * (nerfstudio N2; reflect_sampling_nerf_components.py:38-140) that slot s carries, -1 for padding: the col_map of the
 * weight gradients of trunk layer 0 / the skip layer (rsn_weight_grad_jobs) and of mlp_mid's SH part. */
int rsn_train_saved_layout(const rsn_field_desc* desc, int32_t* enc_cols, int32_t* sh_cols, int32_t* narrow_bf16,
                           int32_t* enc_map, int32_t* sh_map);

/* ---- weights: nn.Linear layout -> MFMA fragment order ---------------------------------------
 * The field kernels stream weights in the exact order the 32x32x2 f32 MFMA consumes them; this
 * re-lays the Field's parameters into one flat buffer (done once per optimiser step).  The split-bf16
 * copies of the segments are written only when desc->mma_mode != RSN_MMA_F32: re-pack after changing it. */
size_t rsn_packed_weights_bytes(const rsn_field_desc* desc);
int rsn_pack_weights(const rsn_field_desc* desc, const rsn_field_params* params, float* packed,
                     size_t packed_bytes, void* stream);

/* The same result in ONE kernel launch per call (plus one for the split-bf16 copies and one for the bf16 ring stream
 * when desc->mma_mode asks for them) instead of one launch per segment: the per-segment job descriptors are kept in a
 * caller-owned device buffer `table` of rsn_pack_table_bytes() bytes.  They depend only on desc and on the parameter /
 * packed POINTERS: pass rebuild_table != 0 on the first call and whenever one of those changed (one asynchronous
 * host-to-device copy on `stream`), 0 otherwise (the optimiser step of a training loop: values change, pointers do
 * not). */
size_t rsn_pack_table_bytes(void);
int rsn_pack_weights_table(const rsn_field_desc* desc, const rsn_field_params* params, float* packed,
                           size_t packed_bytes, void* table, size_t table_bytes, int32_t rebuild_table, void* stream);

/* ---- samplers --------------------------------------------------------------------------------
 * rsn_sample_spaced replaces UniformSampler / ReciprocalSampler.generate_ray_samples
 * (reflect_sampling_nerf_model.py:148,292; reflect_sampling_nerf_components.py:14-36).
 * t_rand: [R,S+1] uniform [0,1) stratified jitter (training) or NULL (eval).
 * Outputs: spacing_bins [R,S+1] (normalised), euclid_bins [R,S+1] (ray parameter t). */
int rsn_sample_spaced(int32_t n_rays, const int32_t* n_dev, int32_t n_samples, int32_t spacing, float tan,
                      const float* nears, const float* fars, const float* t_rand, float* spacing_bins,
                      float* euclid_bins, void* stream);

/* rsn_sample_pdf replaces PDFSampler.generate_ray_samples with include_original=False
 * (reflect_sampling_nerf_model.py:110,112,182,317).  weights [R,S_in], spacing_bins_in [R,S_in+1],
 * u_rand [R,S_out+1] or NULL.  S_in, S_out <= 1024. */
int rsn_sample_pdf(int32_t n_rays, const int32_t* n_dev, int32_t s_in, int32_t s_out, int32_t spacing, float tan,
                   float histogram_padding, const float* nears, const float* fars, const float* weights,
                   const float* spacing_bins_in, const float* u_rand, float* spacing_bins_out,
                   float* euclid_bins_out, void* stream);

/* ---- the field (dominant kernel) -------------------------------------------------------------
 * rsn_field_forward_frustum: for every sample of every ray: conical frustum -> Gaussian ->
 * contraction -> integrated positional encoding -> trunk MLP -> heads -> SH-34 -> mid MLP ->
 * colour.  Replaces get_blob + contract + get_density + get_pred_normals + get_reflection +
 * get_diff + get_tint + get_roughness + get_mid for one sampling level
 * (reflect_sampling_nerf_field.py:90-186; driven from reflect_sampling_nerf_model.py:151-175,
 * 185-209,293-310,319-336).  N = n_rays * n_samples; sample i of ray r is point r*S+i. */
int rsn_field_forward_frustum(const rsn_field_desc* desc, const float* packed, int32_t n_rays, const int32_t* n_dev,
                              int32_t n_samples, const float* origins, const float* directions,
                              const float* pixel_area, const float* euclid_bins, const rsn_field_outputs* out,
                              void* stream);

/* Training-mode variant of rsn_field_forward_frustum: identical outputs, plus the saved activations and
 * (when saved->normals != NULL) the analytic normals of Field.get_normals (reflect_sampling_nerf_field.py:
 * 125-127,134-135,146-147; reflect_sampling_nerf_model.py:159-160,194-195). */
int rsn_field_forward_frustum_train(const rsn_field_desc* desc, const float* packed, int32_t n_rays,
                                    const int32_t* n_dev, int32_t n_samples, const float* origins,
                                    const float* directions, const float* pixel_area, const float* euclid_bins,
                                    const rsn_field_outputs* out, const rsn_field_saved* saved, void* stream);

/* Several training-mode evaluations of the SAME field in ONE launch: the reflect branch of a step evaluates the field on
 * M reflected rays x S samples (model.py:292-297, 317-323) and on M points for get_inf_color (model.py:290) -- a few hundred
 * to ~1,250 tiles of 128 points on 256 persistent workgroups each.  Launched one by one every evaluation wastes its last
 * partial round of tiles and get_inf_color keeps 20 CUs busy for a whole launch; as jobs of one launch their tiles form
 * one tile space.  kind 0 = rsn_field_forward_frustum_train's arguments, kind 1 = rsn_field_forward_inf_train's. */
typedef struct rsn_field_job {
  int32_t kind;              /* 0: conical frustums of rays; 1: get_inf_color */
  int32_t n_rays;
  const int32_t* n_dev;      /* optional device-side ray count */
  int32_t n_samples;         /* kind 0 */
  const float* origins;      /* kind 0 */
  const float* directions;
  const float* pixel_area;   /* kind 0 */
  const float* euclid_bins;  /* kind 0 */
  const float* sqradius;     /* kind 1 */
  float* out_rgb;            /* kind 1: [n_rays,3] */
  const rsn_field_outputs* out;  /* kind 0 */
  const rsn_field_saved* saved;
} rsn_field_job;

int rsn_field_forward_train_jobs(const rsn_field_desc* desc, const float* packed, int32_t n_jobs,
                                 const rsn_field_job* jobs, void* stream);

/* ---- backward of one field level (training) ----------------------------------------------------
 * Upstream gradients per sample (NULL = zero) ... */
typedef struct rsn_field_grads_in {
  const float* sigma;        /* [N]   d loss / d sigma (from rsn_composite_backward)                 */
  const float* color;        /* [N,3] d loss / d colour                                              */
  const float* pred_normals; /* [N,3] d loss / d pred_normals (predicted-normal loss, model.py:403-404) */
  const float* n_dot_d;      /* [N]   d loss / d n_dot_d       (orientation loss, model.py:406-407)  */
  const float* roughness;    /* [N]   d loss / d sigmoid(roughness head) (rendered roughness, model.py:225-226) */
  /* fused normal losses: upstream gradients PER RAY of rsn_composite_io.pn_loss_ray / ori_loss_ray; the kernel forms
   * d/d pred_normals = ray_pn_loss[r] w (-2)(normals - pred_normals) and d/d n_dot_d = ray_ori_loss[r] w 2 max(0, n_dot_d)
   * itself from the level's weights, analytic normals (rsn_field_saved.normals) and forward outputs (pred_normals,
   * n_dot_d), so the per-sample gradient tensors never exist.  Added to pred_normals / n_dot_d above when both are given. */
  const float* ray_pn_loss;  /* [R] or NULL */
  const float* ray_ori_loss; /* [R] or NULL */
  const float* weights;      /* [N] compositing weights of the level (needed by the two above) */
} rsn_field_grads_in;

/* Pre-activation gradients of every linear layer, row-major, consumed by the weight-gradient GEMMs
 * dW = dY^T X (plain library GEMMs on the host) and by rsn_colsum (bias gradients). */
typedef struct rsn_field_grads_out {
  float* dz_rgb;    /* [N,4]   field_output_mid pre-sigmoid (3 live columns)                          */
  float* da_mid;    /* [N,128] mlp_mid pre-activation                      (bf16 under RSN_MMA_BF16)   */
  float* d_bott;    /* [N,W]   bottleneck output                           (bf16 under RSN_MMA_BF16)   */
  float* dz_heads;  /* [N,16]  columns: 0 density, 1-3 normals, 4-6 diff, 8 roughness, 12-14 tint      */
  float* dy;        /* [L,N,W] pre-activation of trunk layer l             (bf16 under RSN_MMA_BF16)   */
  float* d_input;   /* [N]     d loss / d pixel_area (frustum) or d sqradius (inf); need_input_grad only */
} rsn_field_grads_out;

/* The backward sweeps of several evaluations of the same field in ONE launch (see rsn_field_job): kind 0 =
 * rsn_field_backward_frustum's arguments, kind 1 = rsn_field_backward_inf's (g_rgb = upstream gradient of its colour).
 * The sweeps of the two reflect levels and of get_inf_color are independent of each other once both levels' compositing
 * backward has produced the background gradient. */
typedef struct rsn_field_bwd_job {
  int32_t kind;
  int32_t n_rays;
  const int32_t* n_dev;
  int32_t n_samples;         /* kind 0 */
  int32_t need_input_grad;
  const float* origins;      /* kind 0 */
  const float* directions;
  const float* pixel_area;   /* kind 0 */
  const float* euclid_bins;  /* kind 0 */
  const float* sqradius;     /* kind 1 */
  const float* g_rgb;        /* kind 1: [n_rays,3] */
  const rsn_field_outputs* fwd;      /* kind 0 */
  const rsn_field_saved* saved;
  const rsn_field_grads_in* gin;     /* kind 0 */
  const rsn_field_grads_out* gout;
} rsn_field_bwd_job;

int rsn_field_backward_jobs(const rsn_field_desc* desc, const float* packed, int32_t n_jobs,
                            const rsn_field_bwd_job* jobs, void* stream);

/* rsn_field_backward_frustum: backward of rsn_field_forward_frustum_train.  fwd: the forward's per-sample
 * outputs (raw_density, diff, tint are read).  need_input_grad != 0 additionally carries the gradient through
 * the integrated positional encoding's variance back to pixel_area (reflected rays: pixel_area = pi*sqradius
 * depends on the non-detached rendered roughness, reflect_sampling_nerf_model.py:225-227,272,286). */
int rsn_field_backward_frustum(const rsn_field_desc* desc, const float* packed, int32_t n_rays, const int32_t* n_dev,
                               int32_t n_samples, const float* origins, const float* directions,
                               const float* pixel_area, const float* euclid_bins, const rsn_field_outputs* fwd,
                               const rsn_field_saved* saved, const rsn_field_grads_in* gin,
                               const rsn_field_grads_out* gout, int32_t need_input_grad, void* stream);

/* Training variant / backward of rsn_field_forward_inf (get_inf_color, field.py:190-201). */
int rsn_field_forward_inf_train(const rsn_field_desc* desc, const float* packed, int32_t n_rays, const int32_t* n_dev,
                                const float* directions, const float* sqradius, float* out_rgb,
                                const rsn_field_saved* saved, void* stream);
int rsn_field_backward_inf(const rsn_field_desc* desc, const float* packed, int32_t n_rays, const int32_t* n_dev,
                           const float* directions, const float* sqradius, const rsn_field_saved* saved,
                           const float* g_rgb, const rsn_field_grads_out* gout, int32_t need_input_grad,
                           void* stream);

/* rsn_field_forward_inf: get_inf_color (reflect_sampling_nerf_field.py:190-201): mean = 2d,
 * Sigma = 0.6*sqradius*(I - d d^T), no contraction, SH inputs zeroed; out_rgb [M,3]. */
int rsn_field_forward_inf(const rsn_field_desc* desc, const float* packed, int32_t n_rays, const int32_t* n_dev,
                          const float* directions, const float* sqradius, float* out_rgb, void* stream);

/* rsn_field_forward_gaussians: granular Field API (get_density(mean, cov) + heads,
 * reflect_sampling_nerf_field.py:122-186) on explicit, already contracted Gaussians:
 * means [N,3], cov_diag [N,3] (NULL => plain sin/cos encoding), view_dirs [N,3] (NULL => SH zeroed),
 * embedding [N,W] optional output (post-ReLU trunk output). */
int rsn_field_forward_gaussians(const rsn_field_desc* desc, const float* packed, int32_t n_points,
                                const float* means, const float* cov_diag, const float* view_dirs,
                                const rsn_field_outputs* out, float* embedding, void* stream);

/* rsn_field_forward_gaussians_train (ABI 16): the same evaluation in TRAINING mode -- reference field.py:122-137 with
 * requires_density_grad=True followed by get_normals() (field.py:146-147): the saved activations of the points
 * (rsn_field_saved, slab layout: rsn_train_saved_layout) and, in saved->normals [N,3], the analytic normals
 * -normalize(d raw_density / d mean) of the (contracted) means handed in.  Exact-fp32 fields (mma_mode RSN_MMA_F32) only:
 * the reduced / split precision training kernels take conical frustums, not Gaussians (RSN_ERR_UNSUPPORTED otherwise). */
int rsn_field_forward_gaussians_train(const rsn_field_desc* desc, const float* packed, int32_t n_points,
                                      const float* means, const float* cov_diag, const float* view_dirs,
                                      const rsn_field_outputs* out, float* embedding, const rsn_field_saved* saved,
                                      void* stream);

/* rsn_field_forward_embedding: the head getters of the granular Field API on a caller-supplied embedding [N,W]
 * (post-ReLU trunk output, as returned by get_density): get_pred_normals, get_diff, get_tint, get_roughness
 * (out->roughness = sigmoid, out->raw_density = density head), get_mid / get_low (out->color = diff + tint*mid
 * is NOT what get_mid returns: the mid colour itself is written to out->color when out->diff and out->tint are
 * NULL).  view_dirs NULL => SH inputs zeroed (get_low); roughness [N] NULL => softplus(roughness head)
 * (reflect_sampling_nerf_field.py:139-186). */
int rsn_field_forward_embedding(const rsn_field_desc* desc, const float* packed, int32_t n_points,
                                const float* embedding, const float* view_dirs, const float* roughness,
                                const rsn_field_outputs* out, void* stream);

/* ---- granular geometry (Field.get_blob / contract / get_reflection) --------------------------------------
 * rsn_gaussians: conical frustum -> Gaussian per sample (reflect_sampling_nerf_field.py:90-96 ->
 * Frustums.get_gaussian_blob): per-sample origins/directions [N,3], pixel_area/starts/ends [N] ->
 * mean [N,3], cov [N,3,3]. */
int rsn_gaussians(int64_t n, const float* origins, const float* directions, const float* pixel_area,
                  const float* starts, const float* ends, float* mean, float* cov, void* stream);
/* rsn_contract: reflect_sampling_nerf_field.py:98-119: mean' and J cov J with the diagonal clamped >= 0. */
int rsn_contract(int64_t n, const float* mean, const float* cov, float* mean_out, float* cov_out, void* stream);
/* rsn_reflection: reflect_sampling_nerf_field.py:203-207: n_dot_d [N] and normalize(d - 2 (n.d) n) [N,3]. */
int rsn_reflection(int64_t n, const float* directions, const float* normals, float* reflections, float* n_dot_d,
                   void* stream);

/* ---- compositing ------------------------------------------------------------------------------
 * rsn_composite: RaySamples.get_weights + RGB/Accumulation/Depth(median)/Normals/Semantic
 * renderers for one level (reflect_sampling_nerf_model.py:154-156,176-177,188-190,210-227,
 * 296-297,311,322-323,337,341).  One wavefront per ray, wave-level scan for the transmittance.
 * background: 0 = none ("random"), 1 = white, 2 = per-ray tensor bg_rgb [R,3].
 * flags: RSN_COMP_EVAL = RGBRenderer in eval mode (nan_to_num the colours, clamp composites to [0,1]);
 *        RSN_COMP_CLIP_RGB = additionally apply the model's own torch.clip(rgb, 0, 1)
 *        (reflect_sampling_nerf_model.py:177,211; not applied to the reflect composites, :311,337).
 * Outputs (NULL = skip): weights [R,S], rgb [R,3], accumulation
 * [R], depth [R]; surface attributes from the optional per-sample inputs: diff_out [R,3] (white
 * background), tint_out [R,3] (no background), normals_out [R,3] (normalised, eps 1e-10),
 * roughness_out [R]. */
typedef struct rsn_composite_io {
  const float* sigma;        /* [R,S] */
  const float* euclid_bins;  /* [R,S+1] */
  const float* color;        /* [R,S,3] */
  const float* bg_rgb;       /* [R,3] or NULL */
  const float* diff;         /* [R,S,3] or NULL */
  const float* tint;         /* [R,S,3] or NULL */
  const float* pred_normals; /* [R,S,3] or NULL */
  const float* roughness;    /* [R,S] or NULL */
  float* weights;
  float* rgb;
  float* accumulation;
  float* depth;
  float* diff_out;
  float* tint_out;
  float* normals_out;
  float* roughness_out;
  /* training: the per-sample loss terms of get_loss_dict (reflect_sampling_nerf_model.py:395-407) reduced per ray in the
   * compositing epilogue, where the weights are in registers:
   *   pn_loss_ray[r]  = sum_s w |normals - pred_normals|^2      (predicted_normal_loss_*: the sum over r)
   *   ori_loss_ray[r] = sum_s w max(0, n_dot_d)^2               (orientation_loss_*)
   * normals = the analytic normals of the training forward (rsn_field_saved.normals); all NULL = skip. */
  const float* normals;      /* [R,S,3] or NULL */
  const float* n_dot_d;      /* [R,S] or NULL */
  float* pn_loss_ray;        /* [R] or NULL (needs normals and pred_normals) */
  float* ori_loss_ray;       /* [R] or NULL (needs n_dot_d) */
} rsn_composite_io;

#define RSN_COMP_EVAL 1
#define RSN_COMP_CLIP_RGB 2
int rsn_composite(int32_t n_rays, const int32_t* n_dev, int32_t n_samples, int32_t background, int32_t flags,
                  const rsn_composite_io* io, void* stream);

/* rsn_composite_backward: backward of rsn_composite in training mode (no eval clamp).
 * g_rgb [R,3]: gradient w.r.t. the rgb output (after the model's clip when RSN_COMP_CLIP_RGB: masked where the
 * unclipped composite left [0,1]); g_roughness [R] (or NULL): gradient w.r.t. the rendered roughness;
 * weights: the forward's weights [R,S].  detach_weights != 0: the weights were detached (reflect levels,
 * model.py:297,323): no gradient reaches sigma.  Outputs (NULL = skip): g_sigma [R,S], g_color [R,S,3],
 * g_roughness_sample [R,S], g_bg [R,3] (background == 2). */
typedef struct rsn_composite_bwd_io {
  const float* sigma;
  const float* euclid_bins;
  const float* color;
  const float* bg_rgb;
  const float* roughness;       /* [R,S] per-sample sigmoid roughness (or NULL) */
  const float* weights;
  const float* g_rgb;
  const float* g_roughness;
  const float* g_accumulation;  /* [R] or NULL: gradient w.r.t. sum_s w (model.py:240-241: white*(1-acc_fine)) */
  float* g_sigma;
  float* g_color;
  float* g_roughness_sample;
  float* g_bg;
} rsn_composite_bwd_io;

int rsn_composite_backward(int32_t n_rays, const int32_t* n_dev, int32_t n_samples, int32_t background, int32_t flags,
                           int32_t detach_weights, const rsn_composite_bwd_io* io, void* stream);

/* Standalone encoders: what calling the Field's encoding modules directly computes.
 * rsn_sh34_encode: IntegratedSHEncoding.forward (components.py:52-140): 34 real-SH terms of bands l = 1, 2, 4, 8 of
 *   `directions` [n,3], band l attenuated by exp(-l(l+1)/2 * roughness) (roughness [n] or NULL = 0); out [n,34].
 * rsn_ipe_encode: nerfstudio NeRFEncoding(3, 16, 0, 16, include_input=True).forward(means, covs) as the Field calls
 *   it (field.py:129-131): out [n,99] = [sin block 48 | sin(.+pi/2) block 48 | raw 3], each block coord-major /
 *   freq-minor, scaled by exp(-0.5 var f^2) when cov_diag [n,3] is given.  freqs16: HOST array of the 16 frequencies. */
int rsn_sh34_encode(int64_t n, const float* directions, const float* roughness, float* out, void* stream);
int rsn_ipe_encode(int64_t n, const float* means, const float* cov_diag, const float* freqs16, float* out,
                   void* stream);

/* rsn_weight_grad: dW[n][col_map ? col_map[k] : k] += sum_m dY[m][n] * X[m][k]  and  db[n] += sum_m dY[m][n]
 * (n < n_out <= 256, k < k_in <= 256; entries with col_map[k] < 0 are dropped).  The weight-gradient GEMM of
 * every linear layer of the Field: a reduction over all N sample points with the output tile stationary in MFMA
 * accumulators.  dW / db are ACCUMULATED (the caller zeroes them once per step); row-major, leading dims in floats. */
int rsn_weight_grad(int64_t n_points, const float* dy, int32_t ld_dy, int32_t n_out, const float* x, int32_t ld_x,
                    int32_t k_in, const int32_t* col_map, float* dw, int32_t ld_dw, float* db, void* stream);

/* rsn_weight_grad_multi: the same reduction over n_segments (<= 8) point sets in ONE launch -- segment s holds
 * n_points[s] rows of dy[s] / x[s] (HOST arrays of device pointers; common leading dimensions).  The five field
 * evaluations of one training step (model.py:146-292) share their weights, so the per-launch cost (the atomic flush
 * of the stationary output tile) is paid once per layer instead of once per layer and evaluation. */
int rsn_weight_grad_multi(int32_t n_segments, const int64_t* n_points, const float* const* dy, int32_t ld_dy,
                          int32_t n_out, const float* const* x, int32_t ld_x, int32_t k_in, const int32_t* col_map,
                          float* dw, int32_t ld_dw, float* db, void* stream);

/* The same reduction in the Field's MMA mode (rsn_field_desc.mma_mode), over the same fp32 buffers:
 *   RSN_MMA_F32 / RSN_MMA_BF16X3  the exact kernel above;
 *   RSN_MMA_BF16X6  both operands split exactly into bf16 triples inside the kernel, 6 products on
 *                   v_mfma_f32_32x32x16_bf16 with fp32 accumulation: fp32-equivalent (dropped terms <= 2^-24 relative);
 *   RSN_MMA_BF16    operands rounded to bf16 (the opt-in reduced-precision training mode), fp32 accumulation.
 * Bias sums are exact fp32 in every mode.  Shapes / alignments the vector-load layout does not cover take the exact
 * kernel. */
int rsn_weight_grad_multi_mode(int32_t n_segments, const int64_t* n_points, const float* const* dy, int32_t ld_dy,
                               int32_t n_out, const float* const* x, int32_t ld_x, int32_t k_in, const int32_t* col_map,
                               float* dw, int32_t ld_dw, float* db, int32_t mma_mode, void* stream);

/* The same with DEVICE-side segment lengths: segment s holds min(n_points_max[s], *n_dev[s] * per_count[s]) rows when
 * n_dev[s] is not NULL (HOST array of device pointers to int32 counts; per_count = rows per counted unit, i.e. samples
 * per ray), n_points_max[s] rows otherwise.  The reflect branch of a training step runs on the M rays behind the mask
 * (reference model.py:229,259-290); M is produced on the device by rsn_reflect_setup and never read by the host, so the
 * step has no device-to-host synchronisation.  The grid is sized for the upper bounds.
 * operand_bf16 (RSN_MMA_BF16 only): bit 0 = the x rows, bit 1 = the dy rows ARE bf16 in memory (the reduced-precision
 * training mode keeps its wide buffers as bf16, see rsn_field_saved / rsn_field_grads_out); leading dimensions count
 * elements.  bf16 dy rows need n_out > 32; a shape the vector-load layout does not cover is an error, not a fallback. */
int rsn_weight_grad_multi_dev(int32_t n_segments, const int64_t* n_points_max, const int32_t* const* n_dev,
                              const int32_t* per_count, const float* const* dy, int32_t ld_dy, int32_t n_out,
                              const float* const* x, int32_t ld_x, int32_t k_in, const int32_t* col_map, float* dw,
                              int32_t ld_dw, float* db, int32_t mma_mode, int32_t operand_bf16, void* stream);

/* Several weight-gradient reductions of ONE shape over the SAME segments in one launch (the trunk layers of a training step:
 * 256 x 256 each, reference autograd of reflect_sampling_nerf_field.py:54-60's MLP).  The workgroups are dealt to the jobs
 * round-robin; every workgroup still flushes one output tile, so n_jobs layers pay ONE atomic-flush phase and one launch
 * ramp.  dy[s] / x[s]: the job's rows of segment s (HOST arrays of n_segments device pointers); other arguments as
 * rsn_weight_grad_multi_dev. */
typedef struct rsn_wgrad_job {
  const float* const* dy;  /* [n_segments] -> [n_s, ld_dy]                        */
  const float* const* x;   /* [n_segments] -> [n_s, ld_x]                         */
  const int32_t* col_map;  /* optional: packed column k -> destination column / -1 */
  float* dw;               /* [n_out, ld_dw], accumulated                          */
  int32_t ld_dw;
  float* db;               /* [n_out] or NULL, accumulated                         */
} rsn_wgrad_job;
int rsn_weight_grad_jobs(int32_t n_segments, const int64_t* n_points_max, const int32_t* const* n_dev,
                         const int32_t* per_count, int32_t n_jobs, const rsn_wgrad_job* jobs, int32_t ld_dy, int32_t n_out,
                         int32_t ld_x, int32_t k_in, int32_t mma_mode, int32_t operand_bf16, void* stream);

/* get_loss_dict on per-ray quantities only (training step: the per-sample normal terms arrive reduced per ray from
 * rsn_composite): losses8[k] = the UNSCALED terms (0-3 MSE means of rgb4[k] against image; 4,5 = sum_r pn_loss_ray2[lv][r];
 * 6,7 = sum_r ori_loss_ray2[lv][r]); g_rgb4[k] = coef8[k] * d term / d rgb4[k]. */
int rsn_loss_rays_forward(int32_t n_rays, const float* image, const float* const* rgb4, const float* const* pn_loss_ray2,
                          const float* const* ori_loss_ray2, const float* coef8, float* losses8, float* const* g_rgb4,
                          void* stream);

/* its chain rule: g_rgb4[k] *= upstream8[k] in place; g_pn_ray2[lv][r] = coef8[4+lv] * upstream8[4+lv],
 * g_ori_ray2[lv][r] = coef8[6+lv] * upstream8[6+lv] (upstream8: DEVICE pointer). */
int rsn_loss_rays_backward(int32_t n_rays, const float* upstream8, const float* coef8, float* const* g_rgb4,
                           float* const* g_pn_ray2, float* const* g_ori_ray2, void* stream);

/* rsn_colsum: out[c] (+)= sum_r x[r*ld + c], c < n_cols (bias gradients = column sums of dY). */
int rsn_colsum(int64_t n_rows, int32_t n_cols, int32_t ld, const float* x, float* out, int32_t accumulate,
               void* stream);

/* rsn_ray_sum: out[r] = sum_s x[r*S + s]  (per-ray total of a per-sample quantity). */
int rsn_ray_sum(int32_t n_rays, const int32_t* n_dev, int32_t n_samples, const float* x, float* out, void* stream);

/* rsn_reflect_backward: gradient of the secondary-ray construction w.r.t. the rendered roughness
 * (reflect_sampling_nerf_model.py:272,286): sqradius = 2|n.d| roughness^2, pixel_area = pi * sqradius.
 * g_roughness[ray_index[i]] = (g_sqradius[i] + pi * g_pixel_area[i]) * 2|n.d| * 2 roughness, zero elsewhere. */
int rsn_reflect_backward(int32_t n_rays, const int32_t* n_masked, const int32_t* ray_index, const float* n_dot_d,
                         const float* roughness, const float* g_sqradius, const float* g_pixel_area,
                         float* g_roughness, void* stream);

/* rsn_reflect_default_backward: rays that are NOT reflected keep mid_reflect_{coarse,fine} = white*(1-acc_fine)
 * with a live accumulation (reflect_sampling_nerf_model.py:240-241):
 * g_accumulation[r] = mask[r] ? 0 : -sum_c (g_reflect_coarse[r,c] + g_reflect_fine[r,c]). */
int rsn_reflect_default_backward(int32_t n_rays, const uint8_t* mask, const float* g_reflect_coarse,
                                 const float* g_reflect_fine, float* g_accumulation, void* stream);

/* ---- reflection rays --------------------------------------------------------------------------
 * rsn_reflect_setup: reflect_sampling_nerf_model.py:222-229,240-241,267-289: n_dot_d, mask =
 * (acc > 1e-2) & (n_dot_d < 0), stable compaction of the masked rays, secondary-ray origins /
 * directions / sqradius / pixel_area / nears / fars, and the default reflect colour
 * white*(1-acc) for every ray.  Outputs: mask [R] (uint8), n_masked (device int32), ray_index [R]
 * (first M entries = original ray of compacted ray i), compacted [M,*] arrays, reflect_default [R,3]
 * written to BOTH reflect_coarse and reflect_fine. */
typedef struct rsn_reflect_io {
  const float* origins;       /* [R,3] */
  const float* directions;    /* [R,3] */
  const float* accumulation;  /* [R]   accumulation_fine */
  const float* depth;         /* [R]   depth_fine (median) */
  const float* pred_normals;  /* [R,3] rendered, normalised */
  const float* roughness;     /* [R]   rendered */
  uint8_t* mask;              /* [R] */
  int32_t* n_masked;          /* [1] */
  int32_t* ray_index;         /* [R] */
  float* n_dot_d;             /* [R] */
  float* origins2;            /* [R,3] (first M valid) */
  float* directions2;         /* [R,3] */
  float* sqradius;            /* [R]   */
  float* pixel_area2;         /* [R]   pi * sqradius */
  float* nears2;              /* [R]   0 */
  float* fars2;               /* [R]   reflect_far */
  float* reflect_coarse;      /* [R,3] white*(1-acc) */
  float* reflect_fine;        /* [R,3] white*(1-acc) */
  int32_t* workspace;         /* rsn_reflect_workspace_bytes(R) bytes, contents irrelevant (per-block counts) */
} rsn_reflect_io;

size_t rsn_reflect_workspace_bytes(int32_t n_rays);
int rsn_reflect_setup(int32_t n_rays, float reflect_far, const rsn_reflect_io* io, void* stream);

/* rsn_reflect_combine: out[ray_index[i]] = clip(diff[ray_index[i]] + tint[ray_index[i]] * comp[i], 0, 1)
 * for i < *n_masked (reflect_sampling_nerf_model.py:312-313,338-339). */
int rsn_reflect_combine(int32_t n_rays_max, const int32_t* n_masked, const int32_t* ray_index, const float* diff,
                        const float* tint, const float* comp, float* out, void* stream);

/* rsn_reflect_combine_backward: backward of rsn_reflect_combine w.r.t. the composite (diff/tint are detached,
 * reflect_sampling_nerf_model.py:216,218): g_comp[i] = g_out[r] * tint[r] where 0 <= diff[r]+tint[r]*comp[i] <= 1. */
int rsn_reflect_combine_backward(int32_t n_rays_max, const int32_t* n_masked, const int32_t* ray_index,
                                 const float* diff, const float* tint, const float* comp, const float* g_out,
                                 float* g_comp, void* stream);

/* ---- the two steps after the hot path in a training iteration ------------------------
 * rsn_loss_forward_backward: get_loss_dict (reflect_sampling_nerf_model.py:346-430).  Term order of losses8 /
 * coef8: loss_mid_coarse, loss_mid_fine, loss_reflect_mid_coarse, loss_reflect_mid_fine (MSE means against `image`
 * [R,3]), predicted_normal_loss_{coarse,fine} = sum w |normals - pred_normals|^2, orientation_loss_{coarse,fine} =
 * sum w max(0, n_dot_d)^2.  losses8 (device, overwritten) receives the UNSCALED terms; the gradient outputs
 * (w.r.t. rgb4, pred_normals2, n_dot_d2; all other inputs are constants of the loss) are scaled by coef8
 * (host array, the model's loss_coefficients incl. the 50-step warm-up of reflect_sampling_nerf_pipeline.py:79-91).
 * The *4 / *2 arguments are HOST arrays of device pointers (coarse, fine order). */
int rsn_loss_forward_backward(int32_t n_rays, int32_t s_coarse, int32_t s_fine, const float* image,
                              const float* const* rgb4, const float* const* weights2, const float* const* normals2,
                              const float* const* pred_normals2, const float* const* n_dot_d2, const float* coef8,
                              float* losses8, float* const* g_rgb4, float* const* g_pred_normals2,
                              float* const* g_n_dot_d2, void* stream);

/* Chain rule for the gradients rsn_loss_forward_backward wrote: g_rgb4[k] *= upstream8[k], g_pred_normals2[lv] *=
 * upstream8[4 + lv], g_n_dot_d2[lv] *= upstream8[6 + lv], in place, upstream8 = d total / d (scaled loss k) in DEVICE
 * memory (what autograd hands to the backward of get_loss_dict's terms; all ones for loss = sum of the terms). */
int rsn_loss_scale_grads(int32_t n_rays, int32_t s_coarse, int32_t s_fine, const float* upstream8, float* const* g_rgb4,
                         float* const* g_pred_normals2, float* const* g_n_dot_d2, void* stream);

/* rsn_radam_step: one RAdam update (torch.optim.RAdam semantics; the reference's optimiser for the "fields" group,
 * reflect_sampling_nerf_config.py:50-53: lr 1e-3, eps 1e-15, betas 0.9/0.999) over all parameter tensors in one
 * launch.  HOST arrays of device pointers; grads[t] == NULL skips tensor t (unused parameter); step counts from 1. */
int rsn_radam_step(int32_t n_tensors, float* const* params, const float* const* grads, float* const* exp_avg,
                   float* const* exp_avg_sq, const int32_t* sizes, int32_t step, float lr, float beta1, float beta2,
                   float eps, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* RSN_H */
