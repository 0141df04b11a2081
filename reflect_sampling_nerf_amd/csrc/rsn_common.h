// Shared declarations of librsn_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#include "rsn.h"

// Diagnostic builds.  Timing ablations that give WRONG RESULTS BY CONSTRUCTION (RSN_RING_NO_*, RSN_RING_MFMA16,
// RSN_R16_*, RSN_DIAG_NO_SAVED_ROWS), the per-phase cycle counters (RSN_PHASE_TIMERS) and any other -D experiment exist
// only in libraries built with -DRSN_DIAG_BUILD, which tools/_variant.py adds and _build.py's product flags never
// contain; rsn_abi_version() of such a library carries RSN_ABI_DIAG_FLAG, and the Python loader refuses it as the
// product library (reflect_sampling_nerf_amd/_abi.py).
#define RSN_ABI_DIAG_FLAG 0x10000
#ifndef RSN_DIAG_BUILD
#if defined(RSN_RING_NO_BARRIER) || defined(RSN_RING_NO_WAIT) || defined(RSN_RING_NO_DMA) || defined(RSN_RING_NO_MFMA) || \
    defined(RSN_RING_MFMA16) || defined(RSN_RING_NO_ENCODE) || defined(RSN_RING_SETPRIO) || defined(RSN_R16_NO_LDS_READ) || \
    defined(RSN_R16_NO_MFMA) || defined(RSN_R16_DOUBLE_MFMA) || defined(RSN_PHASE_TIMERS) || \
    defined(RSN_DIAG_NO_SAVED_ROWS) || defined(RSN_DIAG_WG_NO_FLUSH) || defined(RSN_DIAG_X6_SAMEW) || \
    defined(RSN_RT_NO_STORES) || defined(RSN_RT_UNCOUNTED) || defined(RSN_RT_NO_LOADS) || defined(RSN_RT_NO_BITS) || \
    defined(RSN_RT_NO_SWEEP) || defined(RSN_RT_NO_PREP) || defined(RSN_RT_SOFFSET_STORES) || defined(RSN_F32_RING_TRAIN) || defined(RSN_F32_NO_WAIT) || defined(RSN_RT_NO_WAIT) || defined(RSN_RT_NO_BARRIER) || defined(RSN_DIAG_NO_EPI_VALU) || defined(RSN_RT_ASM_LOADS) || \
    defined(WG_X6_STAGED) || defined(WG_F32_STAGED) || defined(WG_F32_SPREAD) || defined(WG_X6_EXTRA) || defined(WG_X6_VALU4) || defined(WG_X6_VALU2) || defined(WG_X6_OLD) || defined(WG_X6_INSTAGE) || defined(WG_X6_NEAR) || defined(WG_X6_PTR_LOADS) || defined(WGS_DMA) || \
    defined(WGS_NO_SPLIT) || defined(WGS_NO_MFMA) || defined(WGS_NO_DMA) || defined(WGS_NO_ISSUE) || defined(WGS_NO_BARRIER) || defined(WGS_NO_WAIT) || \
    defined(RSN_F32_NO_BARRIER)
#error "timing-diagnostic macros need -DRSN_DIAG_BUILD (tools/_variant.py): they never go into librsn_hip.so"
#endif
#endif

#define RSN_K_ENC_PAD 104  // 99 encoded inputs padded to 13 K-iterations of 8
#define RSN_K_SH_PAD 40    // 34 SH inputs padded to 5 K-iterations of 8
#define RSN_ENC_ITS (RSN_K_ENC_PAD / 8)
#define RSN_SH_ITS (RSN_K_SH_PAD / 8)
#define RSN_ENC_K16 7      // encoded inputs in K=16 steps (112 >= 104)
#define RSN_SH_K16 3       // SH inputs in K=16 steps (48 >= 40)
#define RSN_AUX_ITS 6      // LDS K-iterations reserved for the SH inputs (even, for the K=16 steps)

void rsn_set_error(const char* fmt, ...);
int rsn_device_cus();                   // CU count of the current device (cached per device)
#ifdef RSN_DIAG_BUILD
bool rsn_env_flag(const char* name);
int rsn_env_int(const char* name, int dflt);    // tools: A/B switches from the environment (diagnostic builds only)
#endif

#define RSN_REQUIRE(cond, code, ...)        \
  do {                                      \
    if (!(cond)) {                          \
      rsn_set_error(__VA_ARGS__);           \
      return (code);                        \
    }                                       \
  } while (0)

#define RSN_HIP(call)                                                                   \
  do {                                                                                  \
    hipError_t e_ = (call);                                                             \
    if (e_ != hipSuccess) {                                                             \
      rsn_set_error("%s failed: %s (%s:%d)", #call, hipGetErrorString(e_), __FILE__, __LINE__); \
      return RSN_ERR_HIP;                                                               \
    }                                                                                   \
  } while (0)

// Offsets (in floats) of every packed segment inside the flat packed-weights buffer.
// A weight segment with n_it K-iterations and nbo 32-row output blocks is laid out
// [it][nb][lane(64)][4]: the float4 that lane (i = lane&31, h = lane>>5) feeds to the four
// v_mfma_f32_32x32x2_f32 K-steps of iteration `it` for output block nb, i.e.
// W[nb*32 + i][col(it*8 + 4h + s)], s = 0..3.
struct RsnPackedLayout {
  int nb;           // width / 32
  int nbm;          // mid_width / 32
  size_t w_x[RSN_MAX_TRUNK_LAYERS];   // x-part of layer l (l >= 1)
  size_t w_enc0;                      // layer 0 (encoding input)
  size_t w_enc_skip;                  // encoding part of the skip layer
  size_t b[RSN_MAX_TRUNK_LAYERS];     // bias of layer l
  size_t w_bh, b_bh;                  // bottleneck (nb blocks) + heads (1 block)
  size_t w_mid_sh, w_mid_x, b_mid;    // mlp_mid: SH part, bottleneck part
  size_t w_rgb, b_rgb;                // field_output_mid (rows 4..6 of one block)
  // transposed segments (training: dX sweeps).  Rows = the layer's INPUT features, K = its output features.
  size_t wT_x[RSN_MAX_TRUNK_LAYERS];  // (x-part of layer l)^T, l >= 1: rows W, K = W
  size_t wT_enc0, wT_enc_skip;        // (encoded-input part)^T: rows = 104 slots padded to 128, K = W
  size_t wT_bh;                       // [bottleneck; heads]^T: rows W, K = W + 32
  size_t wT_mid_x;                    // (bottleneck part of mlp_mid)^T: rows W, K = mid_width
  size_t wT_rgb;                      // (RGB head)^T: rows mid_width, K = 32 (k = 4..6 live)
  size_t v_density;                   // density head weight row [W] (seed of the analytic-normal sweep)
  // split-bf16 copies of the forward segments (RSN_MMA_BF16X6 / X3): [k16][nb][split(3)][lane][8 bf16];
  // offsets in floats like everything else (one (k16, nb, split) chunk = 1 KiB = 256 floats)
  size_t h_x[RSN_MAX_TRUNK_LAYERS];
  size_t h_enc0, h_enc_skip, h_bh, h_mid_sh, h_mid_x, h_rgb;
  size_t hT_x[RSN_MAX_TRUNK_LAYERS];  // split-bf16 copies of the transposed segments (training sweeps)
  size_t hT_enc0, hT_enc_skip, hT_bh, hT_mid_x, hT_rgb;
  // RSN_MMA_BF16 at width 256 only: the whole network's bf16 weight fragments (1 KiB = [lane][8 bf16] each) as ONE
  // linear stream in the exact order rsn_field_bf16_ring_kernel consumes them, in groups of 8 fragments (8 KiB): the
  // kernel's workgroups pull it through an LDS ring by LDS-DMA (rsn_field_bf16.hip)
  size_t r_stream;                    // 0 = absent
  int r_groups;                       // fragment groups per pass over the network
  // the same network as 16x32 fragments for v_mfma_f32_16x16x32_bf16 (rsn_field_bf16_ring16_kernel): lane
  // (i = lane & 15, g = lane >> 4) holds W[16 b + i][feature(kk, g, e)], e = 0..7, of fragment (K-step kk, row block b)
  size_t q_stream;                    // 0 = absent
  int q_groups;
  // ... and, directly behind it (group indices continue), the TRANSPOSED 16x32 fragments of the training sweeps
  // (rsn_field_bf16_train.hip) in consumption order: (RGB head)^T 1 group, (mlp_mid x part)^T 4, [bottleneck; heads]^T 9,
  // then for l = L-1 .. 1: (encoded-input part of the skip layer)^T 4 groups in front of l == skip, (x part of layer l)^T 8;
  // last (layer 0)^T 4 groups.  The backward sweep walks all of it (without the two encoded-input pieces when no input
  // gradient is wanted); the analytic-normal sweep of the training forward walks [t_g_trunk, t_g_end) behind the forward stream.
  int t_g_begin, t_g_trunk, t_g_encskip, t_g_enc0, t_g_end;   // absolute group indices from q_stream; t_g_encskip = -1: no skip layer
  // RSN_MMA_BF16X6 at width 256 (rsn_field_x6_train.hip): the same stream with every fragment as THREE 1 KiB pieces -- the lo, mid
  // and hi bf16 parts of the fp32 weights, in that order (small products first); q_pf = 3 and every group count above is x 3
  int q_pf;                           // 1 KiB pieces per fragment of q_stream: 1 (plain bf16) or 3 (split-bf16)
  // RSN_MMA_F32 at width 256 (tools/probes/rsn_field_f32_ring.hip, the exact-fp32 training forward on the LDS weight ring: diagnostic builds only): the fp32 fragments
  // ([lane][4] = 1 KiB, the (it, nb) chunks of the segments above) as ONE linear stream in consumption order, every GEMM padded to
  // whole groups of 16 fragments: enc0 (13 its x 8 blocks, padded to 14), then per layer l = 1..L-1 the x part (32 x 8) and, behind
  // l == skip, the encoded-input part; [bottleneck; heads] 32 x 9; mlp_mid SH part 5 (padded to 8) x 4; its x part 32 x 4; RGB head
  // 16 x 1.  Directly behind it the TRANSPOSED trunk of the analytic-normal sweep: for l = L-1..1 the (encoded-input part of the skip
  // layer)^T 32 x 4 in front of l == skip and (x part of layer l)^T 32 x 8; last (layer 0)^T 32 x 4.
  size_t f_stream;                    // 0 = absent
  int f_groups, ft_end;               // forward stream = groups [0, f_groups), transposed trunk = [f_groups, ft_end)
  size_t total;                       // floats
};
#ifndef RSN_RING_GROUP_FRAGS
#define RSN_RING_GROUP_FRAGS 16
#endif
#define RSN_RING_MAX_LAYERS 10   // trunk depth the ring kernels' LDS bias table is sized for
// the TRAINING kernels on the LDS weight ring (rsn_field_bf16_train.hip: plain bf16; rsn_field_x6_train.hip: split-bf16) serve
// this shape; every other shape / mode trains on rsn_field_kernel<., true, .> / rsn_field_bwd_kernel
inline bool rsn_ring_training(const rsn_field_desc* d) {
  return (d->mma_mode == RSN_MMA_BF16 || d->mma_mode == RSN_MMA_BF16X6) && d->width == 256 &&
         d->num_layers <= RSN_RING_MAX_LAYERS && RSN_RING_GROUP_FRAGS == 16;
}

// The exact-fp32 TRAINING FORWARD on the LDS weight ring (tools/probes/rsn_field_f32_ring.hip): built in round 4, bit-identical to
// rsn_field_kernel<8, true, 0> and 7 % slower -- NOT part of the product: only diagnostic builds with
// -DRSN_F32_RING_TRAIN (tools/_variant.py, extra source) pack its stream and dispatch to it.
inline bool rsn_f32_ring_training(const rsn_field_desc* d) {
#ifdef RSN_F32_RING_TRAIN
  return d->mma_mode == RSN_MMA_F32 && d->width == 256 && d->num_layers <= RSN_RING_MAX_LAYERS && RSN_RING_GROUP_FRAGS == 16;
#else
  (void)d;
  return false;
#endif
}

int rsn_compute_layout(const rsn_field_desc* desc, RsnPackedLayout* L);
