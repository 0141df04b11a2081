// rsn_wgrad.hip -- weight gradients dW[n][k] += sum_m dY[m][n] * X[m][k], db[n] += sum_m dY[m][n].
//
// A reduction GEMM over ALL sample points (M ~ 1e5..1e6) with a small output (<= 256 x 256): the opposite
// shape of the forward GEMMs, and one the BLAS library serves poorly (hipBLASLt picks 32x64 tiles: 36 ms per
// training step at BASELINE config 3).  Here the OUTPUT is stationary: a 4-wave workgroup keeps the whole
// dW tile in MFMA accumulators (wave w owns output rows [64w, 64w+64) x all input columns = 16 blocks of
// 32x32 = 256 accumulator registers) and streams its chunk of points once.  For v_mfma_f32_32x32x2_f32 with
// A = dY^T and B = X the two fragments are simply "one float per lane from row m0 + h": lane (i, h) reads
// dY[m0+h][n0+i] and X[m0+h][k0+i] straight from the row-major buffers (32 consecutive floats per half-wave:
// two full 128-B segments per load), so no LDS and no barriers are needed.  The bias gradient is the running sum
// of the A fragments.  Each workgroup flushes its partial tile once with fp32 atomics (two 128-B row segments
// per wave instruction: the full-rate shape, MI355X_MICROARCH.md "Global float atomics").
//
// MFMA-bound: 2 * N * n_out * k_in FLOP; HBM reads N * (n_out + k_in) * 4 B (each operand once per workgroup).
#include "rsn_mfma.h"

struct WGradArgs {
  long long n_points;
  long long chunk;        // points per workgroup (multiple of 8)
  const float* dy;        // [N, ld_dy]
  const float* x;         // [N, ld_x]
  int ld_dy, ld_x, n_out, k_in, ld_dw;
  const int* col_map;     // optional: packed column k -> destination column (or -1)
  float* dw;              // [n_out, ld_dw], accumulated
  float* db;              // [n_out] or NULL, accumulated
};

#define WG_PAIRS 4  // point pairs (MFMA K-steps) per software-pipeline stage

template <int NKB>  // input-column blocks of 32 held per wave (8 covers k_in <= 256)
__global__ __launch_bounds__(256) void rsn_wgrad_kernel(const WGradArgs a) {
  const int lane = threadIdx.x & 63;
  const int wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);  // provably wave-uniform: scalar loop control
  const int i = lane & 31, h = lane >> 5;
  // Waves are assigned (row-block pair, point sub-chunk): P = pairs of 32-row blocks the output needs (1, 2 or 4);
  // the 4/P waves that share a pair split the workgroup's chunk of points, so narrow outputs (heads, RGB head,
  // mlp_mid) keep every wave busy.
  const int P = a.n_out <= 64 ? 1 : (a.n_out <= 128 ? 2 : 4);
  const int nsub = 4 / P;
  const int nb0 = (wid % P) * 2;  // this wave's two 32-row output blocks
  const int sub = wid / P;
  const long long wg_begin = (long long)blockIdx.x * a.chunk;
  const long long sub_chunk = a.chunk / nsub;  // chunk is a multiple of 16
  const long long m_begin = wg_begin + sub * sub_chunk;
  long long m_end = m_begin + sub_chunk;
  if (m_end > a.n_points) m_end = a.n_points;
  if (m_begin >= a.n_points) return;
  const bool t1_live = (nb0 + 1) * 32 < a.n_out;  // wave-uniform: second row block holds live rows

  f32x16 acc[2][NKB];
#pragma unroll
  for (int t = 0; t < 2; ++t)
#pragma unroll
    for (int kb = 0; kb < NKB; ++kb)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[t][kb][r] = 0.0f;
  float bsum[2] = {0.0f, 0.0f};

  // Lane columns are clamped into range instead of masked: a lane whose output row / input column does not exist
  // works on a duplicate of the last valid one and its results are never flushed.  Nothing in the main loop
  // touches a loaded value before the MFMAs do, so the loads of stage s+1 stay in flight under the MFMAs of stage s.
  int cdy[2], cx[NKB];
#pragma unroll
  for (int t = 0; t < 2; ++t) {
    const int c = (nb0 + t) * 32 + i;
    cdy[t] = c < a.n_out ? c : a.n_out - 1;
  }
#pragma unroll
  for (int kb = 0; kb < NKB; ++kb) {
    const int c = kb * 32 + i;
    cx[kb] = c < a.k_in ? c : a.k_in - 1;
  }
  const float* __restrict__ dyp = a.dy;
  const float* __restrict__ xp = a.x;

  float fa[2][WG_PAIRS][2], fb[2][WG_PAIRS][NKB];  // [buffer][pair][block]

  auto load_stage = [&](int buf, long long m0) {  // all 2*WG_PAIRS points of the stage are in range
#pragma unroll
    for (int p = 0; p < WG_PAIRS; ++p) {
      const long long m = m0 + 2 * p + h;
      const float* __restrict__ dr = dyp + m * a.ld_dy;
      const float* __restrict__ xr = xp + m * a.ld_x;
#pragma unroll
      for (int t = 0; t < 2; ++t) fa[buf][p][t] = dr[cdy[t]];
#pragma unroll
      for (int kb = 0; kb < NKB; ++kb) fb[buf][p][kb] = xr[cx[kb]];
    }
  };
  auto load_stage_tail = [&](int buf, long long m0) {  // points beyond m_end contribute zeros
#pragma unroll
    for (int p = 0; p < WG_PAIRS; ++p) {
      const long long m = m0 + 2 * p + h;
      const bool in = m < m_end;
      const long long mc = in ? m : m_begin;
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        const float v = dyp[mc * a.ld_dy + cdy[t]];
        fa[buf][p][t] = in ? v : 0.0f;
      }
#pragma unroll
      for (int kb = 0; kb < NKB; ++kb) {
        const float v = xp[mc * a.ld_x + cx[kb]];
        fb[buf][p][kb] = in ? v : 0.0f;
      }
    }
  };
  auto mma_stage = [&](int buf) {
#pragma unroll
    for (int p = 0; p < WG_PAIRS; ++p) {
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        if (t == 1 && !t1_live) continue;
        bsum[t] += fa[buf][p][t];
#pragma unroll
        for (int kb = 0; kb < NKB; ++kb)
          acc[t][kb] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[buf][p][t], fb[buf][p][kb], acc[t][kb], 0, 0, 0);
      }
    }
  };

  const long long step = 2 * WG_PAIRS;
  const long long n_full = (m_end - m_begin) / step;  // stages with every point in range
  if (n_full > 0) load_stage(0, m_begin);
#pragma unroll 1
  for (long long sidx = 0; sidx < n_full; sidx += 2) {
    const long long m0 = m_begin + sidx * step;
    if (sidx + 1 < n_full) load_stage(1, m0 + step);
    __builtin_amdgcn_sched_barrier(0);
    mma_stage(0);
    __builtin_amdgcn_sched_barrier(0);
    if (sidx + 2 < n_full) load_stage(0, m0 + 2 * step);
    __builtin_amdgcn_sched_barrier(0);
    if (sidx + 1 < n_full) mma_stage(1);
    __builtin_amdgcn_sched_barrier(0);
  }
  if (m_begin + n_full * step < m_end) {
    load_stage_tail(0, m_begin + n_full * step);
    mma_stage(0);
  }

  // flush: C/D layout col = lane&31 (input column), row = (r&3) + 8*(r>>2) + 4*h (output row)
#pragma unroll
  for (int t = 0; t < 2; ++t) {
    if (t == 1 && !t1_live) continue;
#pragma unroll
    for (int kb = 0; kb < NKB; ++kb) {
      const int k = kb * 32 + i;
      int c = -1;
      if (k < a.k_in) c = a.col_map ? a.col_map[k] : k;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int n = (nb0 + t) * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
        if (c >= 0 && n < a.n_out) atomicAdd(&a.dw[(long long)n * a.ld_dw + c], acc[t][kb][r]);
      }
    }
    if (a.db) {
      const float v = bsum[t] + __shfl_xor(bsum[t], 32, 64);
      const int n = (nb0 + t) * 32 + i;
      if (h == 0 && n < a.n_out) atomicAdd(&a.db[n], v);
    }
  }
}

extern "C" int rsn_weight_grad(int64_t n_points, const float* dy, int32_t ld_dy, int32_t n_out, const float* x,
                               int32_t ld_x, int32_t k_in, const int32_t* col_map, float* dw, int32_t ld_dw,
                               float* db, void* stream) {
  RSN_REQUIRE(n_points >= 0 && n_out >= 1 && n_out <= 256 && k_in >= 1 && k_in <= 256, RSN_ERR_INVALID_ARGUMENT,
              "n_points=%lld n_out=%d k_in=%d (outputs up to 256 x 256)", (long long)n_points, n_out, k_in);
  RSN_REQUIRE(ld_dy >= n_out && ld_x >= k_in && ld_dw >= 1, RSN_ERR_INVALID_ARGUMENT, "leading dimensions too small");
  if (n_points == 0) return RSN_OK;
  RSN_REQUIRE(dy && x && dw, RSN_ERR_INVALID_ARGUMENT, "a pointer is NULL");
  WGradArgs a;
  a.n_points = n_points; a.dy = dy; a.x = x; a.ld_dy = ld_dy; a.ld_x = ld_x; a.n_out = n_out; a.k_in = k_in;
  a.ld_dw = ld_dw; a.col_map = col_map; a.dw = dw; a.db = db;
  static int cached_cus = 0;
  if (cached_cus == 0) {
    int dev = 0, n = 0;
    if (hipGetDevice(&dev) == hipSuccess &&
        hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && n > 0)
      cached_cus = n;
    else
      cached_cus = 256;
  }
  long long chunk = (n_points + cached_cus - 1) / cached_cus;
  chunk = ((chunk + 15) / 16) * 16;
  if (chunk < 64) chunk = 64;
  a.chunk = chunk;
  const long long grid = (n_points + chunk - 1) / chunk;
  hipStream_t st = (hipStream_t)stream;
  if (k_in > 128)
    hipLaunchKernelGGL(rsn_wgrad_kernel<8>, dim3((unsigned)grid), dim3(256), 0, st, a);
  else if (k_in > 64)
    hipLaunchKernelGGL(rsn_wgrad_kernel<4>, dim3((unsigned)grid), dim3(256), 0, st, a);
  else
    hipLaunchKernelGGL(rsn_wgrad_kernel<2>, dim3((unsigned)grid), dim3(256), 0, st, a);
  RSN_HIP(hipGetLastError());
  return RSN_OK;
}
