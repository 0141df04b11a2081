// Weight packing: torch.nn.Linear layout -> the fragment order of v_mfma_f32_32x32x2_f32.
//
// Data layout in HBM (see DESIGN.md "Packed weights"): one flat fp32 buffer; every segment is
// [it][nb][lane][4] so that a wave streams it with 1 KiB-contiguous global_load_dwordx4's.
// The K index is a free permutation (it only changes the summation order); it is chosen so that the
// accumulator registers a lane holds after one layer are exactly the B-operand values the same
// lane needs for the next layer (no cross-lane movement between layers, rsn_field.hip).
#include <stdarg.h>
#include <stdlib.h>
#include <string.h>

#include <string>
#include <vector>

#include "rsn_common.h"

static thread_local std::string g_last_error;

void rsn_set_error(const char* fmt, ...) {
  char buf[1024];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof(buf), fmt, ap);
  va_end(ap);
  g_last_error = buf;
}

extern "C" const char* rsn_last_error(void) { return g_last_error.c_str(); }

// CU count of the CURRENT device (256 on MI355X), cached per device ordinal: a process may drive several devices.
int rsn_device_cus() {
  static int cache[64] = {0};
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return 256;
  if (cache[dev] == 0) {
    int n = 0;
    cache[dev] = (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && n > 0) ? n : 256;
  }
  return cache[dev];
}

#ifdef RSN_DIAG_BUILD  // environment A/B switches exist in diagnostic builds only: the product library reads none
int rsn_env_int(const char* name, int dflt) {
  const char* v = getenv(name);
  return (v != nullptr && v[0] != '\0') ? atoi(v) : dflt;
}

// A/B switches of the tools (RSN_NO_PAIR=1 ...): read once per process and name.
bool rsn_env_flag(const char* name) {
  static thread_local const char* last_name = nullptr;
  static thread_local bool last_val = false;
  if (last_name == name) return last_val;  // call sites pass string literals: pointer identity is enough
  const char* v = getenv(name);
  last_name = name;
  last_val = v != nullptr && v[0] != '\0' && v[0] != '0';
  return last_val;
}
#endif
#ifdef RSN_DIAG_BUILD
extern "C" int rsn_abi_version(void) { return RSN_ABI_VERSION | RSN_ABI_DIAG_FLAG; }  // never the product library
#else
extern "C" int rsn_abi_version(void) { return RSN_ABI_VERSION; }
#endif

int rsn_compute_layout(const rsn_field_desc* d, RsnPackedLayout* L) {
  RSN_REQUIRE(d != nullptr, RSN_ERR_INVALID_ARGUMENT, "desc is NULL");
  RSN_REQUIRE(d->num_layers >= 1 && d->num_layers <= RSN_MAX_TRUNK_LAYERS, RSN_ERR_INVALID_ARGUMENT,
              "num_layers=%d out of range [1,%d]", d->num_layers, RSN_MAX_TRUNK_LAYERS);
  RSN_REQUIRE(d->width == 64 || d->width == 128 || d->width == 256, RSN_ERR_UNSUPPORTED,
              "width=%d unsupported (64, 128 or 256)", d->width);
  RSN_REQUIRE(d->mid_width == 128, RSN_ERR_UNSUPPORTED, "mid_width=%d unsupported (128)", d->mid_width);
  RSN_REQUIRE(d->param_width >= 0 && d->param_width <= d->width, RSN_ERR_INVALID_ARGUMENT,
              "param_width=%d must be 0 (= width) or 1..width=%d", d->param_width, d->width);
  RSN_REQUIRE(d->mma_mode >= RSN_MMA_F32 && d->mma_mode <= RSN_MMA_BF16, RSN_ERR_INVALID_ARGUMENT, "mma_mode=%d",
              d->mma_mode);
  RSN_REQUIRE(d->skip_layer == -1 || (d->skip_layer >= 1 && d->skip_layer <= d->num_layers - 2),
              RSN_ERR_INVALID_ARGUMENT,
              "skip_layer=%d invalid for num_layers=%d (the reference MLP raises a shape error when the skip "
              "index is the last layer)", d->skip_layer, d->num_layers);
  memset(L, 0, sizeof(*L));
  L->nb = d->width / 32;
  L->nbm = d->mid_width / 32;
  const size_t blk = 256;  // floats per (it, nb) chunk: 64 lanes x 4
  size_t off = 0;
  const size_t x_seg = (size_t)(L->nb * 4) * L->nb * blk;
  const size_t enc_seg = (size_t)RSN_ENC_ITS * L->nb * blk;
  for (int l = 0; l < d->num_layers; ++l) {
    if (l == 0) {
      L->w_enc0 = off;
      off += enc_seg;
    } else {
      L->w_x[l] = off;
      off += x_seg;
      if (l == d->skip_layer) {
        L->w_enc_skip = off;
        off += enc_seg;
      }
    }
    L->b[l] = off;
    off += (size_t)d->width;
  }
  L->w_bh = off;
  off += (size_t)(L->nb * 4) * (L->nb + 1) * blk;
  L->b_bh = off;
  off += (size_t)(L->nb + 1) * 32;
  L->w_mid_sh = off;
  off += (size_t)RSN_SH_ITS * L->nbm * blk;
  L->w_mid_x = off;
  off += (size_t)(L->nb * 4) * L->nbm * blk;
  L->b_mid = off;
  off += (size_t)d->mid_width;
  L->w_rgb = off;
  off += (size_t)(L->nbm * 4) * 1 * blk;
  L->b_rgb = off;
  off += 32;
  // transposed segments
  for (int l = 1; l < d->num_layers; ++l) {
    L->wT_x[l] = off;
    off += x_seg;
  }
  const size_t encT_seg = (size_t)(L->nb * 4) * 4 * blk;  // K = W, 4 row blocks (128 >= 104 slots)
  L->wT_enc0 = off;
  off += encT_seg;
  L->wT_enc_skip = off;
  off += (d->skip_layer >= 1) ? encT_seg : 0;
  L->wT_bh = off;
  off += (size_t)(L->nb * 4 + 4) * L->nb * blk;
  L->wT_mid_x = off;
  off += (size_t)(L->nbm * 4) * L->nb * blk;
  L->wT_rgb = off;
  off += (size_t)4 * L->nbm * blk;
  L->v_density = off;
  off += (size_t)d->width;
  // split-bf16 forward segments: n_k16 * nbo * 3 chunks of 256 floats
  const size_t hx_seg = (size_t)(L->nb * 2) * L->nb * 3 * blk;
  const size_t henc_seg = (size_t)RSN_ENC_K16 * L->nb * 3 * blk;
  L->h_enc0 = off;
  off += henc_seg;
  for (int l = 1; l < d->num_layers; ++l) {
    L->h_x[l] = off;
    off += hx_seg;
  }
  L->h_enc_skip = off;
  off += (d->skip_layer >= 1) ? henc_seg : 0;
  L->h_bh = off;
  off += (size_t)(L->nb * 2) * (L->nb + 1) * 3 * blk;
  L->h_mid_sh = off;
  off += (size_t)RSN_SH_K16 * L->nbm * 3 * blk;
  L->h_mid_x = off;
  off += (size_t)(L->nb * 2) * L->nbm * 3 * blk;
  L->h_rgb = off;
  off += (size_t)(L->nbm * 2) * 1 * 3 * blk;
  for (int l = 1; l < d->num_layers; ++l) {
    L->hT_x[l] = off;
    off += hx_seg;
  }
  const size_t hencT_seg = (size_t)(L->nb * 2) * 4 * 3 * blk;
  L->hT_enc0 = off;
  off += hencT_seg;
  L->hT_enc_skip = off;
  off += (d->skip_layer >= 1) ? hencT_seg : 0;
  L->hT_bh = off;
  off += (size_t)(L->nb * 2 + 2) * L->nb * 3 * blk;
  L->hT_mid_x = off;
  off += (size_t)(L->nbm * 2) * L->nb * 3 * blk;
  L->hT_rgb = off;
  off += (size_t)2 * L->nbm * 3 * blk;
  L->q_pf = 1;
  if (d->mma_mode == RSN_MMA_BF16 && d->width == 256) {
    // enc0 7 K-steps x 8 blocks; x layers 16 x 8; enc_skip 7 x 8; heads 16 x 1; bottleneck 16 x 8; mlp_mid SH part
    // 3 (padded to 4) x 4; mlp_mid x part 16 x 4; rgb 8 x 1 -- every GEMM a whole number of 8-fragment groups
    const int G = RSN_RING_GROUP_FRAGS;
    const int enc_ks = (RSN_ENC_K16 * 8 + G - 1) / G * G / 8, rgb_ks = (8 + G - 1) / G * G;
    L->r_groups = (enc_ks * 8 * (d->skip_layer >= 1 ? 2 : 1) + (d->num_layers - 1) * 128 + 16 + 128 + 16 + 64 + rgb_ks) / G;
    L->r_stream = off;
    off += (size_t)L->r_groups * RSN_RING_GROUP_FRAGS * blk;
  }
  if ((d->mma_mode == RSN_MMA_BF16 || d->mma_mode == RSN_MMA_BF16X6) && d->width == 256 && RSN_RING_GROUP_FRAGS == 16) {
    // 16x32 fragments: enc 4 x 16, x 8 x 16, heads 8 x 2, bottleneck 8 x 16, mid SH 2 x 8, mid x 8 x 8, rgb 4 x 4; split-bf16:
    // three pieces per fragment (q_pf), every count below in 16 KiB groups of PIECES
    const int pf = d->mma_mode == RSN_MMA_BF16X6 ? 3 : 1;
    L->q_pf = pf;
    L->q_groups = pf * ((64 * (d->skip_layer >= 1 ? 2 : 1) + (d->num_layers - 1) * 128 + 16 + 128 + 16 + 64 + 16) / 16);
    L->q_stream = off;
    off += (size_t)L->q_groups * 16 * blk;
    // transposed fragments of the training sweeps, directly behind (see RsnPackedLayout)
    int tg = L->q_groups;
    L->t_g_begin = tg;
    tg += pf * (1 + 4 + 9);
    L->t_g_trunk = tg;
    L->t_g_encskip = -1;
    for (int l = d->num_layers - 1; l >= 1; --l) {
      if (l == d->skip_layer) { L->t_g_encskip = tg; tg += pf * 4; }
      tg += pf * 8;
    }
    L->t_g_enc0 = tg;
    tg += pf * 4;
    L->t_g_end = tg;
    off += (size_t)(tg - L->q_groups) * 16 * blk;
  }
  if (rsn_f32_ring_training(d)) {
    const int enc_g = (RSN_ENC_ITS * 8 + 15) / 16, skip = d->skip_layer >= 1 ? 1 : 0;   // 104 fragments -> 7 groups
    L->f_groups = enc_g * (1 + skip) + (d->num_layers - 1) * 16 + 18 + 2 + 8 + 1;
    L->ft_end = L->f_groups + (d->num_layers - 1) * 16 + 8 * (1 + skip);
    L->f_stream = off;
    off += (size_t)L->ft_end * 16 * blk;
  }
  L->total = off;
  return RSN_OK;
}

extern "C" int rsn_train_saved_layout(const rsn_field_desc* d, int32_t* enc_cols, int32_t* sh_cols, int32_t* narrow_bf16,
                                      int32_t* enc_map, int32_t* sh_map) {
  RSN_REQUIRE(d && enc_cols && sh_cols && narrow_bf16 && enc_map && sh_map, RSN_ERR_INVALID_ARGUMENT, "a pointer is NULL");
  RsnPackedLayout L;
  const int rc = rsn_compute_layout(d, &L);
  if (rc != RSN_OK) return rc;
  for (int s = 0; s < 128; ++s) enc_map[s] = -1;
  for (int s = 0; s < 64; ++s) sh_map[s] = -1;
  if (rsn_ring_training(d)) {  // slot s = 32 kk + 8 g + e of lane group g (rsn_field_bf16_train.hip; cols_enc16 / cols_sh16 below)
    *enc_cols = 128; *sh_cols = 64; *narrow_bf16 = d->mma_mode == RSN_MMA_BF16 ? 1 : 0;  // split-bf16: fp32 rows
    for (int s = 0; s < 128; ++s) {
      const int kk = s >> 5, g = (s >> 3) & 3, e = s & 7, u = kk * 8 + e;
      if (u < 12) enc_map[s] = (u / 4) * 16 + 4 * g + (u % 4);
      else if (u < 24) enc_map[s] = 48 + ((u - 12) / 4) * 16 + 4 * g + ((u - 12) % 4);
      else if (u < 27 && g == 0) enc_map[s] = 96 + (u - 24);
    }
    for (int s = 0; s < 64; ++s) {
      const int kk = s >> 5, g = (s >> 3) & 3, e = s & 7, u = kk * 8 + e;
      if (u < 9 && 9 * g + u < RSN_SH_DIM) sh_map[s] = 9 * g + u;
    }
    return RSN_OK;
  }
  *enc_cols = RSN_K_ENC_PAD; *sh_cols = RSN_K_SH_PAD; *narrow_bf16 = 0;
  for (int k = 0; k < RSN_K_ENC_PAD; ++k) {  // slot k = it * 8 + 4 h + s of lane half h (cols_encoding / cols_sh below)
    const int it = k >> 3, h = (k >> 2) & 1, s = k & 3, u = it * 4 + s;
    if (u < 24) enc_map[k] = (u / 8) * 16 + 8 * h + (u % 8);
    else if (u < 48) enc_map[k] = 48 + ((u - 24) / 8) * 16 + 8 * h + ((u - 24) % 8);
    else if (u < 51 && h == 0) enc_map[k] = 96 + (u - 48);
  }
  for (int k = 0; k < RSN_K_SH_PAD; ++k) {
    const int it = k >> 3, h = (k >> 2) & 1, s = k & 3, u = it * 4 + s;
    if (u < 17) sh_map[k] = 17 * h + u;
  }
  return RSN_OK;
}

extern "C" size_t rsn_packed_weights_bytes(const rsn_field_desc* desc) {
  RsnPackedLayout L;
  if (rsn_compute_layout(desc, &L) != RSN_OK) return 0;
  return L.total * sizeof(float);
}

// ---- one pack job = one segment -------------------------------------------------------------------
#define PACK_MAX_ROWS 288
#define PACK_MAX_COLS 256
#define PACK_MAX_SRC 6

struct PackJob {
  const float* src[PACK_MAX_SRC];
  int ld[PACK_MAX_SRC];
  float* dst;
  int n_it, nbo, is_bias, n_rows, transpose;  // transpose: packed row n <- source COLUMN, packed k <- source ROW
  int layout;  // 0: fp32 [it][nb][lane][4] (32x32x2 fragments);  1: bf16 [k32][b16][lane][8] (16x16x32 fragments)
  int16_t row_src[PACK_MAX_ROWS];  // which src a packed row comes from, -1 = zero row
  int16_t row_idx[PACK_MAX_ROWS];  // row inside that src
  int16_t col[PACK_MAX_COLS];      // source column of packed k, -1 = zero
  int16_t col_src[PACK_MAX_COLS];  // transpose == 3: which src packed k comes from (its row col[k]), -1 = zero
};

__device__ __forceinline__ void pack_elem(const PackJob& job, int e) {
  if (job.layout >= 1) {  // one bf16 of a 16x32 fragment: row 16 b + (lane & 15), k = 32 kk + 8 (lane >> 4) + el
    if (e >= job.n_it * job.nbo * 512) return;
    const int el = e & 7, lane = (e >> 3) & 63, frag = e >> 9;
    const int b = frag % job.nbo, kk = frag / job.nbo;
    const int n = b * 16 + (lane & 15), k = kk * 32 + (lane >> 4) * 8 + el;
    const int rs = job.row_src[n], c = job.col[k];
    float v = 0.0f;
    if (job.transpose == 3) {  // transposed, the SOURCE tensor selected by k ([heads]^T)
      const int cs = job.col_src[k];
      if (rs >= 0 && c >= 0 && cs >= 0) v = job.src[cs][(size_t)c * job.ld[cs] + job.row_idx[n]];
    } else if (rs >= 0 && c >= 0) {
      v = job.transpose ? job.src[rs][(size_t)c * job.ld[rs] + job.row_idx[n]]
                        : job.src[rs][(size_t)job.row_idx[n] * job.ld[rs] + c];
    }
    if (job.layout == 2) {  // split-bf16: the fragment as three pieces -- lo, mid, hi parts of v (v = hi + mid + lo up to 2^-24)
      const __bf16 b1 = (__bf16)v;
      const float r1 = v - (float)b1;
      const __bf16 b2 = (__bf16)r1;
      const __bf16 b3 = (__bf16)(r1 - (float)b2);
      __bf16* d3 = reinterpret_cast<__bf16*>(job.dst) + (size_t)frag * 1536 + (e & 511);
      d3[0] = b3; d3[512] = b2; d3[1024] = b1;
      return;
    }
    reinterpret_cast<__bf16*>(job.dst)[e] = (__bf16)v;
    return;
  }
  if (job.is_bias) {
    if (e < job.n_rows) {
      const int rs = job.row_src[e];
      job.dst[e] = rs >= 0 ? job.src[rs][job.row_idx[e]] : 0.0f;
    }
    return;
  }
  const int total = job.n_it * job.nbo * 256;
  if (e >= total) return;
  const int s = e & 3;
  const int lane = (e >> 2) & 63;
  const int chunk = e >> 8;
  const int nb = chunk % job.nbo;
  const int it = chunk / job.nbo;
  const int n = nb * 32 + (lane & 31);
  const int k = it * 8 + 4 * (lane >> 5) + s;
  const int rs = job.row_src[n];
  const int c = job.col[k];
  float v = 0.0f;
  if (job.transpose == 3) {  // transposed, the SOURCE tensor selected by k ([heads]^T: one small tensor per head)
    const int cs = job.col_src[k];
    if (rs >= 0 && c >= 0 && cs >= 0) v = job.src[cs][(size_t)c * job.ld[cs] + job.row_idx[n]];
    job.dst[e] = v;
    return;
  }
  if (rs >= 0 && c >= 0)
    v = job.transpose ? job.src[rs][(size_t)c * job.ld[rs] + job.row_idx[n]]
                      : job.src[rs][(size_t)job.row_idx[n] * job.ld[rs] + c];
  job.dst[e] = v;
}

__global__ void rsn_pack_kernel(const PackJob job) { pack_elem(job, blockIdx.x * blockDim.x + threadIdx.x); }

// Every segment in ONE launch (rsn_pack_weights_table): the job descriptors live in device memory, uploaded once per
// (parameter pointers, shape); workgroup b serves job j with block_start[j] <= b < block_start[j + 1].
#define PACK_MAX_JOBS 96
struct PackTable {
  int n_jobs, n_blocks;
  int block_start[PACK_MAX_JOBS + 1];
  PackJob jobs[PACK_MAX_JOBS];
};

__global__ void rsn_pack_all_kernel(const PackTable* __restrict__ t) {
  const int b = blockIdx.x;
  int lo = 0, hi = t->n_jobs - 1;
  while (lo < hi) {  // last job whose first block is <= b
    const int mid = (lo + hi + 1) >> 1;
    if (t->block_start[mid] <= b) lo = mid; else hi = mid - 1;
  }
  pack_elem(t->jobs[lo], (b - t->block_start[lo]) * 256 + (int)threadIdx.x);
}

// all split-bf16 copies in one launch
#define SPLIT_MAX_SEGS 48
struct SplitSeg { unsigned src, dst; short n_it, nbo; int block0; };
struct SplitJob { float* packed; int n_segs; SplitSeg s[SPLIT_MAX_SEGS]; };

// split-bf16 copy of an already packed fp32 segment [it][nb][lane][4] -> [k16][nb][split(3)][lane][8 bf16]:
// element e of K=16 step kk is the fp32 value of K-iteration 2kk + (e>>2), component e&3 (zero beyond n_it), split
// EXACTLY into three bf16 (v = b1 + b2 + b3 up to 2^-24): the K=16 MFMA step consumes the same lane-local
// activations as the two fp32 K-iterations it replaces.  Works for forward and transposed segments alike.
__device__ __forceinline__ void split_elem(const float* __restrict__ src, int n_it, int nbo, float* dstf, int e) {
  const int n_k16 = (n_it + 1) / 2;
  if (e >= n_k16 * nbo * 512) return;
  const int ee = e & 7;
  const int lane = (e >> 3) & 63;
  const int chunk = e >> 9;
  const int nb = chunk % nbo;
  const int kk = chunk / nbo;
  const int it = 2 * kk + (ee >> 2);
  float v = 0.0f;
  if (it < n_it) v = src[((size_t)(it * nbo + nb) * 64 + lane) * 4 + (ee & 3)];
  const __bf16 b1 = (__bf16)v;
  const float r1 = v - (float)b1;
  const __bf16 b2 = (__bf16)r1;
  const float r2 = r1 - (float)b2;
  const __bf16 b3 = (__bf16)r2;
  __bf16* dst = reinterpret_cast<__bf16*>(dstf);
  const size_t base = (size_t)(kk * nbo + nb) * 3;
  dst[((base + 0) * 64 + lane) * 8 + ee] = b1;
  dst[((base + 1) * 64 + lane) * 8 + ee] = b2;
  dst[((base + 2) * 64 + lane) * 8 + ee] = b3;
}

__global__ void rsn_pack_split_kernel(const float* __restrict__ src, int n_it, int nbo, float* dstf) {
  split_elem(src, n_it, nbo, dstf, blockIdx.x * blockDim.x + threadIdx.x);
}

__global__ void rsn_pack_split_all_kernel(const SplitJob job) {
  int si = 0;
  while (si + 1 < job.n_segs && job.s[si + 1].block0 <= (int)blockIdx.x) ++si;
  const SplitSeg sg = job.s[si];
  split_elem(job.packed + sg.src, sg.n_it, sg.nbo, job.packed + sg.dst, ((int)blockIdx.x - sg.block0) * 256 + (int)threadIdx.x);
}

// ---- ring stream (RSN_MMA_BF16, width 256): split-0 fragments of the h_* segments re-ordered into consumption order
#define RING_MAX_PIECES 48
struct RingPiece {
  unsigned src;      // float offset of the source segment: split-bf16 ([k16][nbo_src][3][lane][8 bf16], mul = 3) or fp32 ([it][nbo_src][lane][4], mul = 1)
  short nbo_src, nb0, nbo, ks_real, ks, mul;
  int frag0;         // first fragment of this piece in the stream
};
struct RingJob {
  const float* packed;
  float* dst;
  int n_pieces, n_frags;
  RingPiece p[RING_MAX_PIECES];
};

__global__ void rsn_pack_ring_kernel(const RingJob job) {  // one 64-thread workgroup per 1 KiB fragment
  const int f = blockIdx.x;
  int pi = 0;
  while (pi + 1 < job.n_pieces && job.p[pi + 1].frag0 <= f) ++pi;
  const RingPiece pc = job.p[pi];
  const int i = f - pc.frag0;
  const int kk = i / pc.nbo, nb = pc.nb0 + i % pc.nbo;
  uint4 v = make_uint4(0u, 0u, 0u, 0u);
  if (kk < pc.ks_real)
    v = reinterpret_cast<const uint4*>(job.packed + pc.src + ((size_t)(kk * pc.nbo_src + nb) * pc.mul) * 256)[threadIdx.x];
  reinterpret_cast<uint4*>(job.dst + (size_t)f * 256)[threadIdx.x] = v;
}

namespace {

void rows_natural(PackJob& j, int n_rows, int src = 0) {
  for (int n = 0; n < n_rows; ++n) {
    j.row_src[n] = (int16_t)src;
    j.row_idx[n] = (int16_t)n;
  }
}

void cols_natural(PackJob& j, int n_cols, int offset) {
  for (int k = 0; k < n_cols; ++k) j.col[k] = (int16_t)(offset + k);
}

// Encoded-input order: lane half h owns frequencies 8h..8h+7.  Slot u of a lane:
//   u in [0,24):  exp*sin  of (coord c = u/8, freq 8h + u%8)   -> reference column c*16 + f
//   u in [24,48): exp*sin(.+pi/2) of the same                 -> reference column 48 + c*16 + f
//   u in [48,51): raw coordinate c (h == 0 only)              -> reference column 96 + c
// packed k = (u/4)*8 + 4h + u%4.            (NeRFEncoding column order: SURVEY §8(a) N2)
void cols_encoding(PackJob& j) {
  for (int k = 0; k < RSN_K_ENC_PAD; ++k) {
    const int it = k >> 3, h = (k >> 2) & 1, s = k & 3;
    const int u = it * 4 + s;
    int c = -1;
    if (u < 24) {
      c = (u / 8) * 16 + 8 * h + (u % 8);
    } else if (u < 48) {
      c = 48 + ((u - 24) / 8) * 16 + 8 * h + ((u - 24) % 8);
    } else if (u < 51 && h == 0) {
      c = 96 + (u - 48);
    }
    j.col[k] = (int16_t)c;
  }
}

// the same slot order used as packed ROWS (transposed segments): packed row r <- reference column enc(r)
int enc_slot_to_column(int k) {
  const int it = k >> 3, h = (k >> 2) & 1, s = k & 3;
  const int u = it * 4 + s;
  if (k >= RSN_K_ENC_PAD) return -1;
  if (u < 24) return (u / 8) * 16 + 8 * h + (u % 8);
  if (u < 48) return 48 + ((u - 24) / 8) * 16 + 8 * h + ((u - 24) % 8);
  if (u < 51 && h == 0) return 96 + (u - 48);
  return -1;
}

// SH order: lane half h owns components 17h .. 17h+16 in slots 0..16 (slots 17..19 zero).
void cols_sh(PackJob& j) {
  for (int k = 0; k < RSN_K_SH_PAD; ++k) {
    const int it = k >> 3, h = (k >> 2) & 1, s = k & 3;
    const int u = it * 4 + s;
    j.col[k] = (int16_t)(u < 17 ? 17 * h + u : -1);
  }
}

// rsn_pack_weights_table collects the jobs instead of launching them one by one
struct PackCollector {
  float* packed;
  std::vector<PackJob> jobs;
  std::vector<SplitSeg> splits;
  bool have_ring = false;
  RingJob ring;
  bool have_fring = false;
  RingJob fring;   // the fp32 consumption-order stream (f_stream)
};
thread_local PackCollector* g_collect = nullptr;

int split_seg(const float* src, int n_it, int nbo, float* dst, hipStream_t st) {
  if (g_collect) {
    SplitSeg sg;
    sg.src = (unsigned)(src - g_collect->packed); sg.dst = (unsigned)(dst - g_collect->packed);
    sg.n_it = (short)n_it; sg.nbo = (short)nbo; sg.block0 = 0;
    g_collect->splits.push_back(sg);
    return RSN_OK;
  }
  const int total = ((n_it + 1) / 2) * nbo * 512;
  const int threads = 256;
  hipLaunchKernelGGL(rsn_pack_split_kernel, dim3((total + threads - 1) / threads), dim3(threads), 0, st, src, n_it, nbo,
                     dst);
  RSN_HIP(hipGetLastError());
  return RSN_OK;
}

int launch(const PackJob& j, hipStream_t st) {
  if (g_collect) {
    g_collect->jobs.push_back(j);
    return RSN_OK;
  }
  const int total = j.is_bias ? j.n_rows : j.n_it * j.nbo * (j.layout >= 1 ? 512 : 256);
  const int threads = 256;
  hipLaunchKernelGGL(rsn_pack_kernel, dim3((total + threads - 1) / threads), dim3(threads), 0, st, j);
  RSN_HIP(hipGetLastError());
  return RSN_OK;
}

void clear_job(PackJob& j) {
  memset(&j, 0, sizeof(j));
  for (int i = 0; i < PACK_MAX_ROWS; ++i) j.row_src[i] = -1;
  for (int i = 0; i < PACK_MAX_COLS; ++i) j.col[i] = -1;
  for (int i = 0; i < PACK_MAX_COLS; ++i) j.col_src[i] = -1;
}

}  // namespace

extern "C" int rsn_pack_weights(const rsn_field_desc* d, const rsn_field_params* p, float* packed,
                                size_t packed_bytes, void* stream) {
  RsnPackedLayout L;
  int rc = rsn_compute_layout(d, &L);
  if (rc != RSN_OK) return rc;
  RSN_REQUIRE(p != nullptr && packed != nullptr, RSN_ERR_INVALID_ARGUMENT, "params/packed is NULL");
  RSN_REQUIRE(packed_bytes >= L.total * sizeof(float), RSN_ERR_WORKSPACE,
              "packed buffer too small: %zu < %zu bytes", packed_bytes, L.total * sizeof(float));
  hipStream_t st = (hipStream_t)stream;
  // WP: the width the kernels run at (64 / 128 / 256); W: the width of the PARAMETER tensors (rsn_field_desc.param_width): units
  // W .. WP - 1 get zero weights and zero biases (their activations and gradients are exact zeros)
  const int WP = d->width, W = (d->param_width > 0 ? d->param_width : d->width), NB = L.nb, NBM = L.nbm;
  PackJob j;

  for (int l = 0; l < d->num_layers; ++l) {
    RSN_REQUIRE(p->trunk_w[l] && p->trunk_b[l], RSN_ERR_INVALID_ARGUMENT, "trunk layer %d has NULL parameters", l);
    const int in_f = (l == 0) ? RSN_ENC_DIM : (l == d->skip_layer ? RSN_ENC_DIM + W : W);
    if (l == 0) {
      clear_job(j);
      j.src[0] = p->trunk_w[l]; j.ld[0] = in_f; j.dst = packed + L.w_enc0; j.n_it = RSN_ENC_ITS; j.nbo = NB;
      rows_natural(j, W); cols_encoding(j);
      if ((rc = launch(j, st)) != RSN_OK) return rc;
    } else {
      clear_job(j);
      j.src[0] = p->trunk_w[l]; j.ld[0] = in_f; j.dst = packed + L.w_x[l]; j.n_it = NB * 4; j.nbo = NB;
      rows_natural(j, W);
      cols_natural(j, W, l == d->skip_layer ? RSN_ENC_DIM : 0);  // cat([encoding, x]): x columns come second
      if ((rc = launch(j, st)) != RSN_OK) return rc;
      if (l == d->skip_layer) {
        clear_job(j);
        j.src[0] = p->trunk_w[l]; j.ld[0] = in_f; j.dst = packed + L.w_enc_skip; j.n_it = RSN_ENC_ITS; j.nbo = NB;
        rows_natural(j, W); cols_encoding(j);
        if ((rc = launch(j, st)) != RSN_OK) return rc;
      }
    }
    clear_job(j);
    j.is_bias = 1; j.src[0] = p->trunk_b[l]; j.dst = packed + L.b[l]; j.n_rows = WP;
    rows_natural(j, W);
    if ((rc = launch(j, st)) != RSN_OK) return rc;
  }

  RSN_REQUIRE(p->density_w && p->normals_w && p->roughness_w && p->diff_w && p->tint_w && p->bottleneck_w &&
                  p->mid_w && p->rgb_w && p->density_b && p->normals_b && p->roughness_b && p->diff_b &&
                  p->tint_b && p->bottleneck_b && p->mid_b && p->rgb_b,
              RSN_ERR_INVALID_ARGUMENT, "a head parameter pointer is NULL");

  // bottleneck (blocks 0..NB-1) + heads block NB.  Heads rows (so that MFMA C rows land where the
  // epilogue wants them): 0 density, 1-3 normals, 4-6 diff, 8 roughness, 12-14 tint.
  auto heads_rows = [&](PackJob& jj, int base) {
    jj.row_src[base + 0] = 1; jj.row_idx[base + 0] = 0;
    for (int c = 0; c < 3; ++c) {
      jj.row_src[base + 1 + c] = 2;  jj.row_idx[base + 1 + c] = (int16_t)c;
      jj.row_src[base + 4 + c] = 3;  jj.row_idx[base + 4 + c] = (int16_t)c;
      jj.row_src[base + 12 + c] = 5; jj.row_idx[base + 12 + c] = (int16_t)c;
    }
    jj.row_src[base + 8] = 4; jj.row_idx[base + 8] = 0;
  };
  clear_job(j);
  j.src[0] = p->bottleneck_w; j.src[1] = p->density_w; j.src[2] = p->normals_w; j.src[3] = p->diff_w;
  j.src[4] = p->roughness_w; j.src[5] = p->tint_w;
  for (int i = 0; i < PACK_MAX_SRC; ++i) j.ld[i] = W;
  j.dst = packed + L.w_bh; j.n_it = NB * 4; j.nbo = NB + 1;
  rows_natural(j, W); heads_rows(j, WP); cols_natural(j, W, 0);
  if ((rc = launch(j, st)) != RSN_OK) return rc;
  clear_job(j);
  j.is_bias = 1; j.n_rows = WP + 32; j.dst = packed + L.b_bh;
  j.src[0] = p->bottleneck_b; j.src[1] = p->density_b; j.src[2] = p->normals_b; j.src[3] = p->diff_b;
  j.src[4] = p->roughness_b; j.src[5] = p->tint_b;
  rows_natural(j, W); heads_rows(j, WP);
  if ((rc = launch(j, st)) != RSN_OK) return rc;

  // mlp_mid: input cat([SH(34), bottleneck(W)])
  clear_job(j);
  j.src[0] = p->mid_w; j.ld[0] = RSN_SH_DIM + W; j.dst = packed + L.w_mid_sh; j.n_it = RSN_SH_ITS; j.nbo = NBM;
  rows_natural(j, d->mid_width); cols_sh(j);
  if ((rc = launch(j, st)) != RSN_OK) return rc;
  clear_job(j);
  j.src[0] = p->mid_w; j.ld[0] = RSN_SH_DIM + W; j.dst = packed + L.w_mid_x; j.n_it = NB * 4; j.nbo = NBM;
  rows_natural(j, d->mid_width); cols_natural(j, W, RSN_SH_DIM);
  if ((rc = launch(j, st)) != RSN_OK) return rc;
  clear_job(j);
  j.is_bias = 1; j.n_rows = d->mid_width; j.src[0] = p->mid_b; j.dst = packed + L.b_mid;
  rows_natural(j, d->mid_width);
  if ((rc = launch(j, st)) != RSN_OK) return rc;

  // field_output_mid (RGB head): rows 4..6 of one 32-row block
  clear_job(j);
  j.src[0] = p->rgb_w; j.ld[0] = d->mid_width; j.dst = packed + L.w_rgb; j.n_it = NBM * 4; j.nbo = 1;
  for (int c = 0; c < 3; ++c) { j.row_src[4 + c] = 0; j.row_idx[4 + c] = (int16_t)c; }
  cols_natural(j, d->mid_width, 0);
  if ((rc = launch(j, st)) != RSN_OK) return rc;
  clear_job(j);
  j.is_bias = 1; j.n_rows = 32; j.src[0] = p->rgb_b; j.dst = packed + L.b_rgb;
  for (int c = 0; c < 3; ++c) { j.row_src[4 + c] = 0; j.row_idx[4 + c] = (int16_t)c; }
  if ((rc = launch(j, st)) != RSN_OK) return rc;

  // ---------------- transposed segments (dX sweeps of the training path) ----------------
  // With transpose=1 a packed ROW selects a source COLUMN (row_idx) and a packed k selects a source ROW (col).
  for (int l = 1; l < d->num_layers; ++l) {
    const int in_f = (l == d->skip_layer) ? RSN_ENC_DIM + W : W;
    clear_job(j);
    j.transpose = 1; j.src[0] = p->trunk_w[l]; j.ld[0] = in_f; j.dst = packed + L.wT_x[l]; j.n_it = NB * 4; j.nbo = NB;
    for (int n = 0; n < W; ++n) { j.row_src[n] = 0; j.row_idx[n] = (int16_t)((l == d->skip_layer ? RSN_ENC_DIM : 0) + n); }
    cols_natural(j, W, 0);
    if ((rc = launch(j, st)) != RSN_OK) return rc;
  }
  for (int which = 0; which < 2; ++which) {
    const int l = which == 0 ? 0 : d->skip_layer;
    if (l < 0) continue;
    const int in_f = (l == 0) ? RSN_ENC_DIM : RSN_ENC_DIM + W;
    clear_job(j);
    j.transpose = 1; j.src[0] = p->trunk_w[l]; j.ld[0] = in_f;
    j.dst = packed + (which == 0 ? L.wT_enc0 : L.wT_enc_skip); j.n_it = NB * 4; j.nbo = 4;
    for (int n = 0; n < 128; ++n) {
      const int c = enc_slot_to_column(n);
      j.row_src[n] = (int16_t)(c >= 0 ? 0 : -1); j.row_idx[n] = (int16_t)(c >= 0 ? c : 0);
    }
    cols_natural(j, W, 0);
    if ((rc = launch(j, st)) != RSN_OK) return rc;
  }
  {  // [bottleneck; heads]^T : rows = W input features; k < W -> bottleneck row k; k >= W -> heads row k - W
    // one source per job row is not enough here (k selects the source), so pack the two K ranges separately
    clear_job(j);
    j.transpose = 1; j.src[0] = p->bottleneck_w; j.ld[0] = W; j.dst = packed + L.wT_bh; j.n_it = NB * 4; j.nbo = NB;
    rows_natural(j, W); cols_natural(j, W, 0);
    if ((rc = launch(j, st)) != RSN_OK) return rc;
    // heads part: 4 K-iterations (k = 0..31 -> heads rows); the source tensor depends on k (one small tensor per head)
    const float* hw[5] = {p->density_w, p->normals_w, p->diff_w, p->roughness_w, p->tint_w};
    const int hbase[5] = {0, 1, 4, 8, 12};
    const int hrows[5] = {1, 3, 3, 1, 3};
    clear_job(j);
    j.transpose = 3;
    for (int t = 0; t < 5; ++t) { j.src[t] = hw[t]; j.ld[t] = W; }
    j.dst = packed + L.wT_bh + (size_t)(NB * 4) * NB * 256; j.n_it = 4; j.nbo = NB;
    rows_natural(j, W);
    for (int t = 0; t < 5; ++t)
      for (int c = 0; c < hrows[t]; ++c) { j.col[hbase[t] + c] = (int16_t)c; j.col_src[hbase[t] + c] = (int16_t)t; }
    if ((rc = launch(j, st)) != RSN_OK) return rc;
  }
  clear_job(j);  // (mlp_mid bottleneck part)^T: rows = W (source columns 34..34+W), K = mid rows
  j.transpose = 1; j.src[0] = p->mid_w; j.ld[0] = RSN_SH_DIM + W; j.dst = packed + L.wT_mid_x; j.n_it = NBM * 4; j.nbo = NB;
  for (int n = 0; n < W; ++n) { j.row_src[n] = 0; j.row_idx[n] = (int16_t)(RSN_SH_DIM + n); }
  cols_natural(j, d->mid_width, 0);
  if ((rc = launch(j, st)) != RSN_OK) return rc;
  clear_job(j);  // (RGB head)^T: rows = mid features, K = 32 with k = 4..6 -> rgb rows 0..2
  j.transpose = 1; j.src[0] = p->rgb_w; j.ld[0] = d->mid_width; j.dst = packed + L.wT_rgb; j.n_it = 4; j.nbo = NBM;
  rows_natural(j, d->mid_width);
  for (int c = 0; c < 3; ++c) j.col[4 + c] = (int16_t)c;
  if ((rc = launch(j, st)) != RSN_OK) return rc;
  clear_job(j);
  j.is_bias = 1; j.n_rows = WP; j.src[0] = p->density_w; j.dst = packed + L.v_density;
  rows_natural(j, W);
  if ((rc = launch(j, st)) != RSN_OK) return rc;

  // ---------------- 16x32 bf16 fragment stream (rsn_field_bf16_ring16_kernel), straight from the nn.Linear tensors ------
  if (L.q_stream != 0) {
    int frag = 0;
    // Output rows of every GEMM whose result feeds another GEMM are PERMUTED: packed row 16 b + 4 g + r (what lane group g
    // holds in accumulator register r of block b after the MFMA) <- source row 32 (b / 2) + 8 g + 4 (b % 2) + r.  The eight
    // values lane (m, g) holds of blocks 2kk, 2kk+1 are then the CONTIGUOUS features 32 kk + 8 g .. + 7: the K order of
    // the next layer is the natural one (slot (kk, g, e) <- feature 32 kk + 8 g + e, lane-local hand-off as before), and a
    // training kernel stores a lane's share of an activation row as ONE 16-byte piece per K-step (the four lanes of a
    // point cover 64 contiguous bytes) instead of two 8-byte pieces 32 bytes apart.
    auto rows_perm16 = [&](PackJob& jj, int n_rows, int n_valid) {  // features >= n_valid (zero-padded units): zero rows
      for (int n = 0; n < n_rows; ++n) {
        const int b = n >> 4, i = n & 15, feat = 32 * (b >> 1) + 8 * (i >> 2) + 4 * (b & 1) + (i & 3);
        jj.row_src[n] = (int16_t)(feat < n_valid ? 0 : -1);
        jj.row_idx[n] = (int16_t)(feat < n_valid ? feat : 0);
      }
    };
    auto cols_x16 = [&](PackJob& jj, int K, int offset) { cols_natural(jj, K, offset); };
    // encoded inputs: lane group g owns frequencies 4g..4g+3; its 32 slots u = 8 kk + e: 12 sin, 12 cos, 3 raw (g == 0)
    auto cols_enc16 = [&](PackJob& jj) {
      for (int k = 0; k < 128; ++k) {
        const int kk = k >> 5, g = (k >> 3) & 3, e = k & 7, u = kk * 8 + e;
        int c = -1;
        if (u < 12) c = (u / 4) * 16 + 4 * g + (u % 4);
        else if (u < 24) c = 48 + ((u - 12) / 4) * 16 + 4 * g + ((u - 12) % 4);
        else if (u < 27 && g == 0) c = 96 + (u - 24);
        jj.col[k] = (int16_t)c;
      }
    };
    // SH inputs: lane group g owns components 9g .. 9g+8 (7 for g == 3) in its slots u = 8 kk + e < 9
    auto cols_sh16 = [&](PackJob& jj) {
      for (int k = 0; k < 64; ++k) {
        const int kk = k >> 5, g = (k >> 3) & 3, e = k & 7, u = kk * 8 + e;
        jj.col[k] = (int16_t)((u < 9 && 9 * g + u < RSN_SH_DIM) ? 9 * g + u : -1);
      }
    };
    auto qpiece = [&](PackJob& jj, int ks, int nbo16) -> int {
      jj.layout = L.q_pf == 3 ? 2 : 1; jj.n_it = ks; jj.nbo = nbo16;
      jj.dst = packed + L.q_stream + (size_t)frag * 256 * L.q_pf;
      frag += ks * nbo16;
      return launch(jj, st);
    };
    clear_job(j);
    j.src[0] = p->trunk_w[0]; j.ld[0] = RSN_ENC_DIM; rows_perm16(j, WP, W); cols_enc16(j);
    if ((rc = qpiece(j, 4, 16)) != RSN_OK) return rc;
    for (int l = 1; l < d->num_layers; ++l) {
      const int in_f = (l == d->skip_layer) ? RSN_ENC_DIM + W : W;
      clear_job(j);
      j.src[0] = p->trunk_w[l]; j.ld[0] = in_f; rows_perm16(j, WP, W); cols_x16(j, W, l == d->skip_layer ? RSN_ENC_DIM : 0);
      if ((rc = qpiece(j, 8, 16)) != RSN_OK) return rc;
      if (l == d->skip_layer) {
        clear_job(j);
        j.src[0] = p->trunk_w[l]; j.ld[0] = in_f; rows_perm16(j, WP, W); cols_enc16(j);
        if ((rc = qpiece(j, 4, 16)) != RSN_OK) return rc;
      }
    }
    clear_job(j);  // heads: ONE 16-row block (0 density, 1-3 normals, 4-6 diff, 8 roughness, 12-14 tint) + a zero block
    j.src[1] = p->density_w; j.src[2] = p->normals_w; j.src[3] = p->diff_w; j.src[4] = p->roughness_w; j.src[5] = p->tint_w;
    for (int i = 0; i < PACK_MAX_SRC; ++i) j.ld[i] = W;
    heads_rows(j, 0); cols_x16(j, W, 0);
    if ((rc = qpiece(j, 8, 2)) != RSN_OK) return rc;
    clear_job(j);
    j.src[0] = p->bottleneck_w; j.ld[0] = W; rows_perm16(j, WP, W); cols_x16(j, W, 0);
    if ((rc = qpiece(j, 8, 16)) != RSN_OK) return rc;
    clear_job(j);
    j.src[0] = p->mid_w; j.ld[0] = RSN_SH_DIM + W; rows_perm16(j, d->mid_width, d->mid_width); cols_sh16(j);
    if ((rc = qpiece(j, 2, 8)) != RSN_OK) return rc;
    clear_job(j);
    j.src[0] = p->mid_w; j.ld[0] = RSN_SH_DIM + W; rows_perm16(j, d->mid_width, d->mid_width); cols_x16(j, W, RSN_SH_DIM);
    if ((rc = qpiece(j, 8, 8)) != RSN_OK) return rc;
    clear_job(j);  // RGB head: rows 4..6 of the first of four 16-row blocks (three of them zero: whole-group padding)
    j.src[0] = p->rgb_w; j.ld[0] = d->mid_width;
    for (int c = 0; c < 3; ++c) { j.row_src[4 + c] = 0; j.row_idx[4 + c] = (int16_t)c; }
    cols_x16(j, d->mid_width, 0);
    if ((rc = qpiece(j, 4, 4)) != RSN_OK) return rc;
    RSN_REQUIRE(frag * L.q_pf == L.q_groups * 16, RSN_ERR_INVALID_ARGUMENT, "16x32 stream: %d fragments, layout says %d groups",
                frag, L.q_groups);
    // ---- transposed pieces (training sweeps): packed ROW 16 b + 4 g + r <- input feature r16_feature (source COLUMN), packed
    //      K natural <- output feature (source ROW); transpose = 1 (see above: row_idx selects the source column)
    auto rowsT_perm16 = [&](PackJob& jj, int n_rows, int n_valid, int col_offset) {
      for (int n = 0; n < n_rows; ++n) {
        const int b = n >> 4, i = n & 15, feat = 32 * (b >> 1) + 8 * (i >> 2) + 4 * (b & 1) + (i & 3);
        jj.row_src[n] = (int16_t)(feat < n_valid ? 0 : -1);
        jj.row_idx[n] = (int16_t)(feat < n_valid ? col_offset + feat : 0);
      }
    };
    // encoded-input slots as packed rows: row 16 b + 4 g + r = slot (kk = b / 2, g, e = 4 (b % 2) + r) of cols_enc16
    auto rowsT_enc16 = [&](PackJob& jj) {
      for (int n = 0; n < 128; ++n) {
        const int b = n >> 4, g = (n >> 2) & 3, r = n & 3, kk = b >> 1, e = 4 * (b & 1) + r, u = kk * 8 + e;
        int c = -1;
        if (u < 12) c = (u / 4) * 16 + 4 * g + (u % 4);
        else if (u < 24) c = 48 + ((u - 12) / 4) * 16 + 4 * g + ((u - 12) % 4);
        else if (u < 27 && g == 0) c = 96 + (u - 24);
        jj.row_src[n] = (int16_t)(c >= 0 ? 0 : -1);
        jj.row_idx[n] = (int16_t)(c >= 0 ? c : 0);
      }
    };
    clear_job(j);  // (RGB head)^T: rows = 128 hidden features; K-step 0 slot (g = 1, e = 0..2) = k 8..10 <- rgb row 0..2; K-step 1 zero
    j.transpose = 1; j.src[0] = p->rgb_w; j.ld[0] = d->mid_width; rowsT_perm16(j, d->mid_width, d->mid_width, 0);
    for (int c = 0; c < 3; ++c) j.col[8 + c] = (int16_t)c;
    if ((rc = qpiece(j, 2, 8)) != RSN_OK) return rc;
    clear_job(j);  // (mlp_mid x part)^T: rows = W bottleneck features (source columns 34..), K = 128 hidden rows
    j.transpose = 1; j.src[0] = p->mid_w; j.ld[0] = RSN_SH_DIM + W; rowsT_perm16(j, WP, W, RSN_SH_DIM); cols_natural(j, d->mid_width, 0);
    if ((rc = qpiece(j, 4, 16)) != RSN_OK) return rc;
    clear_job(j);  // [bottleneck]^T: rows = W embedding features, K = W bottleneck rows ...
    j.transpose = 1; j.src[0] = p->bottleneck_w; j.ld[0] = W; rowsT_perm16(j, WP, W, 0); cols_natural(j, W, 0);
    if ((rc = qpiece(j, 8, 16)) != RSN_OK) return rc;
    {  // ... + [heads]^T as a ninth K-step: slot (g, e < 4) = k 8 g + e <- heads row 4 g + e (0 density, 1-3 normals, 4-6 diff, 8 roughness, 12-14 tint)
      const float* hw[5] = {p->density_w, p->normals_w, p->diff_w, p->roughness_w, p->tint_w};
      const int hbase[5] = {0, 1, 4, 8, 12};
      const int hrows[5] = {1, 3, 3, 1, 3};
      clear_job(j);
      j.transpose = 3;
      for (int t = 0; t < 5; ++t) { j.src[t] = hw[t]; j.ld[t] = W; }
      rowsT_perm16(j, WP, W, 0);
      for (int t = 0; t < 5; ++t)
        for (int c = 0; c < hrows[t]; ++c) {
          const int hr = hbase[t] + c;  // heads row 4 g + e
          j.col[8 * (hr >> 2) + (hr & 3)] = (int16_t)c;
          j.col_src[8 * (hr >> 2) + (hr & 3)] = (int16_t)t;
        }
      if ((rc = qpiece(j, 1, 16)) != RSN_OK) return rc;
    }
    for (int l = d->num_layers - 1; l >= 1; --l) {
      const int in_f = (l == d->skip_layer) ? RSN_ENC_DIM + W : W;
      if (l == d->skip_layer) {
        clear_job(j);
        j.transpose = 1; j.src[0] = p->trunk_w[l]; j.ld[0] = in_f; rowsT_enc16(j); cols_natural(j, W, 0);
        if ((rc = qpiece(j, 8, 8)) != RSN_OK) return rc;
      }
      clear_job(j);
      j.transpose = 1; j.src[0] = p->trunk_w[l]; j.ld[0] = in_f;
      rowsT_perm16(j, WP, W, l == d->skip_layer ? RSN_ENC_DIM : 0); cols_natural(j, W, 0);
      if ((rc = qpiece(j, 8, 16)) != RSN_OK) return rc;
    }
    clear_job(j);
    j.transpose = 1; j.src[0] = p->trunk_w[0]; j.ld[0] = RSN_ENC_DIM; rowsT_enc16(j); cols_natural(j, W, 0);
    if ((rc = qpiece(j, 8, 8)) != RSN_OK) return rc;
    RSN_REQUIRE(frag * L.q_pf == L.t_g_end * 16, RSN_ERR_INVALID_ARGUMENT, "transposed 16x32 stream: %d fragments, layout says %d groups",
                frag, L.t_g_end);
  }

  // ---------------- fp32 consumption-order stream (rsn_field_f32_train.hip): copies of the fp32 fragments above ----------------
  if (L.f_stream != 0) {
    RingJob rj;
    memset(&rj, 0, sizeof(rj));
    rj.packed = packed;
    rj.dst = packed + L.f_stream;
    int frag = 0;
    auto piece = [&](size_t src, int nbo, int its_real, int its) {
      RingPiece& q = rj.p[rj.n_pieces++];
      q.src = (unsigned)src; q.nbo_src = (short)nbo; q.nb0 = 0; q.nbo = (short)nbo;
      q.ks_real = (short)its_real; q.ks = (short)its; q.mul = 1; q.frag0 = frag;
      frag += its * nbo;
    };
    const int enc_its = (RSN_ENC_ITS * 8 + 15) / 16 * 16 / 8;  // 13 -> 14: whole groups
    piece(L.w_enc0, NB, RSN_ENC_ITS, enc_its);
    for (int l = 1; l < d->num_layers; ++l) {
      piece(L.w_x[l], NB, NB * 4, NB * 4);
      if (l == d->skip_layer) piece(L.w_enc_skip, NB, RSN_ENC_ITS, enc_its);
    }
    piece(L.w_bh, NB + 1, NB * 4, NB * 4);
    piece(L.w_mid_sh, NBM, RSN_SH_ITS, 8);
    piece(L.w_mid_x, NBM, NB * 4, NB * 4);
    piece(L.w_rgb, 1, NBM * 4, NBM * 4);
    RSN_REQUIRE(frag == L.f_groups * 16, RSN_ERR_INVALID_ARGUMENT, "fp32 stream: %d fragments, layout says %d groups", frag, L.f_groups);
    for (int l = d->num_layers - 1; l >= 1; --l) {
      if (l == d->skip_layer) piece(L.wT_enc_skip, 4, NB * 4, NB * 4);
      piece(L.wT_x[l], NB, NB * 4, NB * 4);
    }
    piece(L.wT_enc0, 4, NB * 4, NB * 4);
    RSN_REQUIRE(rj.n_pieces <= RING_MAX_PIECES && frag == L.ft_end * 16, RSN_ERR_INVALID_ARGUMENT,
                "fp32 stream: %d fragments in %d pieces, layout says %d groups", frag, rj.n_pieces, L.ft_end);
    rj.n_frags = frag;
    if (g_collect) {
      g_collect->have_fring = true;
      g_collect->fring = rj;
    } else {
      hipLaunchKernelGGL(rsn_pack_ring_kernel, dim3((unsigned)frag), dim3(64), 0, st, rj);
      RSN_HIP(hipGetLastError());
    }
  }

  // ---------------- split-bf16 copies (RSN_MMA_BF16X6 / X3 / BF16) of every GEMM segment ----------------
  if (d->mma_mode == RSN_MMA_F32) return RSN_OK;  // the exact-fp32 kernels never read them
  if ((rc = split_seg(packed + L.w_enc0, RSN_ENC_ITS, NB, packed + L.h_enc0, st)) != RSN_OK) return rc;
  for (int l = 1; l < d->num_layers; ++l) {
    if ((rc = split_seg(packed + L.w_x[l], NB * 4, NB, packed + L.h_x[l], st)) != RSN_OK) return rc;
    if ((rc = split_seg(packed + L.wT_x[l], NB * 4, NB, packed + L.hT_x[l], st)) != RSN_OK) return rc;
  }
  if (d->skip_layer >= 1) {
    if ((rc = split_seg(packed + L.w_enc_skip, RSN_ENC_ITS, NB, packed + L.h_enc_skip, st)) != RSN_OK) return rc;
    if ((rc = split_seg(packed + L.wT_enc_skip, NB * 4, 4, packed + L.hT_enc_skip, st)) != RSN_OK) return rc;
  }
  if ((rc = split_seg(packed + L.wT_enc0, NB * 4, 4, packed + L.hT_enc0, st)) != RSN_OK) return rc;
  if ((rc = split_seg(packed + L.w_bh, NB * 4, NB + 1, packed + L.h_bh, st)) != RSN_OK) return rc;
  if ((rc = split_seg(packed + L.wT_bh, NB * 4 + 4, NB, packed + L.hT_bh, st)) != RSN_OK) return rc;
  if ((rc = split_seg(packed + L.w_mid_sh, RSN_SH_ITS, NBM, packed + L.h_mid_sh, st)) != RSN_OK) return rc;
  if ((rc = split_seg(packed + L.w_mid_x, NB * 4, NBM, packed + L.h_mid_x, st)) != RSN_OK) return rc;
  if ((rc = split_seg(packed + L.wT_mid_x, NBM * 4, NB, packed + L.hT_mid_x, st)) != RSN_OK) return rc;
  if ((rc = split_seg(packed + L.w_rgb, NBM * 4, 1, packed + L.h_rgb, st)) != RSN_OK) return rc;
  if ((rc = split_seg(packed + L.wT_rgb, 4, NBM, packed + L.hT_rgb, st)) != RSN_OK) return rc;
  if (L.r_stream != 0) {
    RingJob rj;
    memset(&rj, 0, sizeof(rj));
    rj.packed = packed;
    rj.dst = packed + L.r_stream;
    int frag = 0;
    auto piece = [&](size_t src, int nbo_src, int nb0, int nbo, int ks_real, int ks) {
      RingPiece& q = rj.p[rj.n_pieces++];
      q.src = (unsigned)src; q.nbo_src = (short)nbo_src; q.nb0 = (short)nb0; q.nbo = (short)nbo;
      q.ks_real = (short)ks_real; q.ks = (short)ks; q.mul = 3; q.frag0 = frag;
      frag += ks * nbo;
    };
    const int G = RSN_RING_GROUP_FRAGS;
    const int enc_ks = (RSN_ENC_K16 * 8 + G - 1) / G * G / 8, rgb_ks = (8 + G - 1) / G * G;  // padded to whole groups
    piece(L.h_enc0, NB, 0, NB, RSN_ENC_K16, enc_ks);
    for (int l = 1; l < d->num_layers; ++l) {
      piece(L.h_x[l], NB, 0, NB, NB * 2, NB * 2);
      if (l == d->skip_layer) piece(L.h_enc_skip, NB, 0, NB, RSN_ENC_K16, enc_ks);
    }
    piece(L.h_bh, NB + 1, NB, 1, NB * 2, NB * 2);  // heads block first (its epilogue feeds the SH encoding)
    piece(L.h_bh, NB + 1, 0, NB, NB * 2, NB * 2);  // bottleneck
    piece(L.h_mid_sh, NBM, 0, NBM, RSN_SH_K16, 4);
    piece(L.h_mid_x, NBM, 0, NBM, NB * 2, NB * 2);
    piece(L.h_rgb, 1, 0, 1, NBM * 2, rgb_ks);
    RSN_REQUIRE(rj.n_pieces <= RING_MAX_PIECES && frag == L.r_groups * RSN_RING_GROUP_FRAGS, RSN_ERR_INVALID_ARGUMENT,
                "ring stream: %d fragments in %d pieces, layout says %d groups", frag, rj.n_pieces, L.r_groups);
    rj.n_frags = frag;
    if (g_collect) {
      g_collect->have_ring = true;
      g_collect->ring = rj;
    } else {
      hipLaunchKernelGGL(rsn_pack_ring_kernel, dim3((unsigned)frag), dim3(64), 0, st, rj);
      RSN_HIP(hipGetLastError());
    }
  }
  return RSN_OK;
}

// ---- the same in (at most) three launches: every fp32 segment, every split-bf16 copy, the ring stream ----------------
extern "C" size_t rsn_pack_table_bytes(void) { return sizeof(PackTable); }

extern "C" int rsn_pack_weights_table(const rsn_field_desc* d, const rsn_field_params* p, float* packed,
                                      size_t packed_bytes, void* table, size_t table_bytes, int32_t rebuild_table,
                                      void* stream) {
  RSN_REQUIRE(table != nullptr && table_bytes >= sizeof(PackTable), RSN_ERR_WORKSPACE,
              "job table buffer too small: %zu < %zu bytes", table_bytes, sizeof(PackTable));
  hipStream_t st = (hipStream_t)stream;
  // the descriptors depend on the shape and on the parameter / packed POINTERS only: collected on the host, laid out
  // as one table and (when the caller says the pointers changed, or on first use) uploaded once
  static thread_local PackCollector col;
  col.packed = packed;
  col.jobs.clear();
  col.splits.clear();
  col.have_ring = false;
  col.have_fring = false;
  g_collect = &col;
  const int rc = rsn_pack_weights(d, p, packed, packed_bytes, stream);
  g_collect = nullptr;
  if (rc != RSN_OK) return rc;
  RSN_REQUIRE((int)col.jobs.size() <= PACK_MAX_JOBS && (int)col.splits.size() <= SPLIT_MAX_SEGS, RSN_ERR_UNSUPPORTED,
              "%zu pack jobs / %zu split segments exceed the table", col.jobs.size(), col.splits.size());
  int blocks = 0;
  static thread_local PackTable host_table;  // stays alive behind the asynchronous upload
  if (rebuild_table) {
    host_table.n_jobs = (int)col.jobs.size();
    for (int i = 0; i < host_table.n_jobs; ++i) {
      const PackJob& j = col.jobs[i];
      host_table.block_start[i] = blocks;
      host_table.jobs[i] = j;
      blocks += ((j.is_bias ? j.n_rows : j.n_it * j.nbo * (j.layout >= 1 ? 512 : 256)) + 255) / 256;
    }
    host_table.block_start[host_table.n_jobs] = blocks;
    host_table.n_blocks = blocks;
    RSN_HIP(hipMemcpyAsync(table, &host_table, sizeof(PackTable), hipMemcpyHostToDevice, st));
  } else {
    for (const PackJob& j : col.jobs)
      blocks += ((j.is_bias ? j.n_rows : j.n_it * j.nbo * (j.layout >= 1 ? 512 : 256)) + 255) / 256;
  }
  hipLaunchKernelGGL(rsn_pack_all_kernel, dim3((unsigned)blocks), dim3(256), 0, st, (const PackTable*)table);
  RSN_HIP(hipGetLastError());
  if (!col.splits.empty()) {
    SplitJob sj;
    memset(&sj, 0, sizeof(sj));
    sj.packed = packed;
    sj.n_segs = (int)col.splits.size();
    int b = 0;
    for (int i = 0; i < sj.n_segs; ++i) {
      sj.s[i] = col.splits[i];
      sj.s[i].block0 = b;
      b += (((sj.s[i].n_it + 1) / 2) * sj.s[i].nbo * 512 + 255) / 256;
    }
    hipLaunchKernelGGL(rsn_pack_split_all_kernel, dim3((unsigned)b), dim3(256), 0, st, sj);
    RSN_HIP(hipGetLastError());
  }
  if (col.have_ring) {
    hipLaunchKernelGGL(rsn_pack_ring_kernel, dim3((unsigned)col.ring.n_frags), dim3(64), 0, st, col.ring);
    RSN_HIP(hipGetLastError());
  }
  if (col.have_fring) {  // (behind rsn_pack_all_kernel on the same stream: it copies what that kernel wrote)
    hipLaunchKernelGGL(rsn_pack_ring_kernel, dim3((unsigned)col.fring.n_frags), dim3(64), 0, st, col.fring);
    RSN_HIP(hipGetLastError());
  }
  return RSN_OK;
}
