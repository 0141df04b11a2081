// rsn_mfma.h -- device helpers shared by the forward (rsn_field.hip) and backward (rsn_field_bwd.hip) field kernels:
// the lane-local MFMA K loop over packed weight segments and the accumulator <-> LDS-slab epilogues.
// See rsn_field.hip for the layout argument (why activations never cross lanes).
#pragma once
#include "rsn_common.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));

#define RSN_MODE_FRUSTUM 0
#define RSN_MODE_INF 1
#define RSN_MODE_GAUSS 2

// ------------------------------------------------------------------------------------------------
// small math, written to follow the torch op order of the reference (contraction off: -ffp-contract=off)
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ float softplus_f(float x) { return x > 20.0f ? x : log1pf(expf(x)); }
__device__ __forceinline__ float sigmoid_f(float x) { return 1.0f / (1.0f + expf(-x)); }

// ------------------------------------------------------------------------------------------------
// MFMA K loop: acc[nb] += W_seg[nb-block] * X, weights double-buffered in registers.
// ------------------------------------------------------------------------------------------------
template <int NBO>
__device__ __forceinline__ void load_w(float4 (&w)[NBO], const float4* __restrict__ wp, int it) {
#pragma unroll
  for (int nb = 0; nb < NBO; ++nb) w[nb] = wp[(it * NBO + nb) * 64];
}

template <int NBO>
__device__ __forceinline__ void mma4(f32x16 (&acc)[NBO], const float4 (&w)[NBO], const float4 b) {
#pragma unroll
  for (int nb = 0; nb < NBO; ++nb) acc[nb] = __builtin_amdgcn_mfma_f32_32x32x2f32(w[nb].x, b.x, acc[nb], 0, 0, 0);
#pragma unroll
  for (int nb = 0; nb < NBO; ++nb) acc[nb] = __builtin_amdgcn_mfma_f32_32x32x2f32(w[nb].y, b.y, acc[nb], 0, 0, 0);
#pragma unroll
  for (int nb = 0; nb < NBO; ++nb) acc[nb] = __builtin_amdgcn_mfma_f32_32x32x2f32(w[nb].z, b.z, acc[nb], 0, 0, 0);
#pragma unroll
  for (int nb = 0; nb < NBO; ++nb) acc[nb] = __builtin_amdgcn_mfma_f32_32x32x2f32(w[nb].w, b.w, acc[nb], 0, 0, 0);
}

// wseg: packed segment base (global), xl: this lane's slot of the LDS slab (float4 units, stride 64 per it).
// sched_barrier(0) pins the software pipeline: the loads of K-iteration it+1 are issued BEFORE the 32 MFMAs
// of iteration it (hipcc otherwise sinks them to just ahead of their first use, leaving ~500 cycles of cover
// for an L2 round trip instead of 2048).
template <int NBO>
__device__ __forceinline__ void gemm(f32x16 (&acc)[NBO], const float* __restrict__ wseg, const float4* xl, int n_it,
                                     int lane) {
  const float4* __restrict__ wp = reinterpret_cast<const float4*>(wseg) + lane;
  float4 wa[NBO], wb[NBO];
  float4 ba, bb;
  load_w<NBO>(wa, wp, 0);
  ba = xl[0];
  int it = 0;
#pragma unroll 1
  for (; it + 1 < n_it; it += 2) {
    load_w<NBO>(wb, wp, it + 1);
    bb = xl[(it + 1) * 64];
    __builtin_amdgcn_sched_barrier(0);
    mma4<NBO>(acc, wa, ba);
    __builtin_amdgcn_sched_barrier(0);
    if (it + 2 < n_it) {
      load_w<NBO>(wa, wp, it + 2);
      ba = xl[(it + 2) * 64];
    }
    __builtin_amdgcn_sched_barrier(0);
    mma4<NBO>(acc, wb, bb);
    __builtin_amdgcn_sched_barrier(0);
  }
  if (it < n_it) mma4<NBO>(acc, wa, ba);
}

// acc[nb][4q+j] = bias[nb*32 + 8q + 4h + j]: the accumulators start from the bias (what torch's addmm does),
// so the 4*NBO bias loads are issued back to back ahead of the K loop and the epilogue needs no memory reads.
template <int NBO>
__device__ __forceinline__ void init_acc(f32x16 (&acc)[NBO], const float* __restrict__ bias, int h) {
#pragma unroll
  for (int nb = 0; nb < NBO; ++nb)
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const float4 bv = *reinterpret_cast<const float4*>(bias + nb * 32 + 8 * q + 4 * h);
      acc[nb][4 * q + 0] = bv.x;
      acc[nb][4 * q + 1] = bv.y;
      acc[nb][4 * q + 2] = bv.z;
      acc[nb][4 * q + 3] = bv.w;
    }
}

template <int NBO>
__device__ __forceinline__ void zero_acc(f32x16 (&acc)[NBO]) {
#pragma unroll
  for (int nb = 0; nb < NBO; ++nb)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[nb][r] = 0.0f;
}

// X[it = nb*4+q][lane] = act(acc[nb][4q..4q+3])   (bias already inside acc, see init_acc).
// save (training): the same float4 also goes to row `save` of a row-major [N, 32*NBS] activation buffer
// (this lane's point; the four q of one nb complete one 128-B line per row).
template <int NBO, int NBS, bool RELU>
__device__ __forceinline__ void store_act(const f32x16 (&acc)[NBO], float4* xl, float* save = nullptr, int h = 0) {
#pragma unroll
  for (int nb = 0; nb < NBS; ++nb)
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      float4 v = make_float4(acc[nb][4 * q + 0], acc[nb][4 * q + 1], acc[nb][4 * q + 2], acc[nb][4 * q + 3]);
      if (RELU) {
        v.x = fmaxf(v.x, 0.0f);
        v.y = fmaxf(v.y, 0.0f);
        v.z = fmaxf(v.z, 0.0f);
        v.w = fmaxf(v.w, 0.0f);
      }
      xl[(nb * 4 + q) * 64] = v;
      if (save) *reinterpret_cast<float4*>(save + (nb * 4 + q) * 8 + 4 * h) = v;
    }
}

// dX-sweep epilogue: X[it][lane] = (x_in > 0) ? acc : 0 with x_in = the saved post-ReLU input of the layer
// (row `xin` of a row-major activation buffer); optionally also stored to row `save` (backward pass).
template <int NBO>
__device__ __forceinline__ void store_masked(const f32x16 (&acc)[NBO], float4* xl, const float* __restrict__ xin, int h,
                                             float* save = nullptr) {
#pragma unroll
  for (int nb = 0; nb < NBO; ++nb)
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const float4 x = *reinterpret_cast<const float4*>(xin + (nb * 4 + q) * 8 + 4 * h);
      float4 v;
      v.x = x.x > 0.0f ? acc[nb][4 * q + 0] : 0.0f;
      v.y = x.y > 0.0f ? acc[nb][4 * q + 1] : 0.0f;
      v.z = x.z > 0.0f ? acc[nb][4 * q + 2] : 0.0f;
      v.w = x.w > 0.0f ? acc[nb][4 * q + 3] : 0.0f;
      xl[(nb * 4 + q) * 64] = v;
      if (save) *reinterpret_cast<float4*>(save + (nb * 4 + q) * 8 + 4 * h) = v;
    }
}

