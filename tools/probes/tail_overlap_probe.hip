// tail_overlap_probe.hip -- can ONE wave per SIMD hide its own layer epilogue (ReLU, LDS hand-off, bias re-load,
// activation store: 3.5 % of the eval kernel, 6 % of the training forward) under its own MFMAs?
// The product layer is K-major (every K-iteration updates all 8 output blocks), so no block is final before the last
// iteration and the whole epilogue follows the GEMM.  Variant "tail": the last T K-iterations run BLOCK-major with their
// B fragments (the activations) held in registers; block nb's accumulators are final after its tail, and its epilogue
// is issued between the MFMAs of block nb+1's tail.  Only the last block's epilogue stays exposed.
// Layer = 256 x 256 GEMM over 32 points per wave (32 K-iterations of 8) + the epilogue of store_act_init, with or
// without the row-major activation store of the training kernel.
//   hipcc -O3 --offload-arch=gfx950 -ffp-contract=off -std=c++17 -I include -I reflect_sampling_nerf_amd/csrc \
//         tools/probes/tail_overlap_probe.hip -o build/tail_overlap_probe && build/tail_overlap_probe
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "rsn_mfma.h"
void rsn_set_error(const char*, ...) {}

#define NBL 8

// epilogue of ONE block (what store_act_init does per block)
template <bool SAVE>
__device__ __forceinline__ void epi_block(f32x16& a, float4* xl, float* save, int h, const float* __restrict__ bias, int nb) {
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    float4 v = make_float4(relu_f(a[4 * q + 0]), relu_f(a[4 * q + 1]), relu_f(a[4 * q + 2]), relu_f(a[4 * q + 3]));
    xl[(nb * 4 + q) * 64] = v;
    if (SAVE) *reinterpret_cast<float4*>(save + (nb * 4 + q) * 8 + 4 * h) = v;
    const float4 bv = *reinterpret_cast<const float4*>(bias + nb * 32 + 8 * q + 4 * h);
    a[4 * q + 0] = bv.x; a[4 * q + 1] = bv.y; a[4 * q + 2] = bv.z; a[4 * q + 3] = bv.w;
  }
}

// K-major main part + block-major tail of T iterations with the previous block's epilogue in the MFMA shadow
template <int T, bool SAVE>
__device__ __forceinline__ void layer_tail(f32x16 (&acc)[NBL], float4 (&wa)[NBL], const float* __restrict__ wseg, float4* xl,
                                           float* save, int h, const float* __restrict__ bias, int lane) {
  constexpr int NIT = 32;
  const float4* __restrict__ wp = reinterpret_cast<const float4*>(wseg) + lane;
  gemm_run<NBL>(acc, wa, wseg, xl, NIT - T, lane);
  float4 bt[T], wt[T], wn[T];
#pragma unroll
  for (int t = 0; t < T; ++t) {
    bt[t] = xl[(NIT - T + t) * 64];
    wt[t] = wp[((NIT - T + t) * NBL + 0) * 64];
  }
#pragma unroll
  for (int nb = 0; nb < NBL; ++nb) {
    if (nb + 1 < NBL) {
#pragma unroll
      for (int t = 0; t < T; ++t) wn[t] = wp[((NIT - T + t) * NBL + nb + 1) * 64];
    }
#pragma unroll
    for (int t = 0; t < T; ++t) {
      acc[nb] = __builtin_amdgcn_mfma_f32_32x32x2f32(wt[t].x, bt[t].x, acc[nb], 0, 0, 0);
      acc[nb] = __builtin_amdgcn_mfma_f32_32x32x2f32(wt[t].y, bt[t].y, acc[nb], 0, 0, 0);
      acc[nb] = __builtin_amdgcn_mfma_f32_32x32x2f32(wt[t].z, bt[t].z, acc[nb], 0, 0, 0);
      acc[nb] = __builtin_amdgcn_mfma_f32_32x32x2f32(wt[t].w, bt[t].w, acc[nb], 0, 0, 0);
    }
    if (nb > 0) epi_block<SAVE>(acc[nb - 1], xl, save, h, bias, nb - 1);
    // issue order: one MFMA, then a slice of the loads / the epilogue
#pragma unroll
    for (int g = 0; g < 4 * T; ++g) {
      __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                     // 1 MFMA
      if (g < T + 4) __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);      // weight fragments of the next block, bias
      __builtin_amdgcn_sched_group_barrier(0x002, 2, 0);                     // 2 VALU
      if ((g & 3) == 3) __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);   // LDS write
      if (SAVE && (g & 3) == 1) __builtin_amdgcn_sched_group_barrier(0x040, 1, 0);  // activation store
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int t = 0; t < T; ++t) wt[t] = wn[t];
  }
  epi_block<SAVE>(acc[NBL - 1], xl, save, h, bias, NBL - 1);
}

// VAR 0: product layer (gemm_run + store_act_init).  VAR 1: tail overlap, T = 8.  VAR 2: T = 4.  VAR 3: GEMM only.
template <int VAR, bool SAVE>
__global__ __launch_bounds__(256, 1) void klayer(const float* __restrict__ pk, const float* __restrict__ bias, float* out,
                                                 float* act, int n_layers) {
  extern __shared__ float4 smem[];
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  float4* X = smem + wid * 32 * 64 + lane;
  for (int it = 0; it < 32; ++it) X[it * 64] = make_float4(0.001f * lane, 0.002f * it, 1.0f, -1.0f);
  f32x16 acc[NBL];
  init_acc<NBL>(acc, bias, lane >> 5);
  for (int l = 0; l < n_layers; ++l) {
    int ln = lane;
    asm volatile("" : "+v"(ln));
    const int h = ln >> 5;
    const int ph = wid * 3 + blockIdx.x;
    const float* seg = pk + (size_t)((l + ph) & 7) * (32 * 8 * 256);
    float* save = act + (((size_t)blockIdx.x * 4 + wid) * 32 + (ln & 31)) * 256;
    float4 wa[NBL];
    pre_w<NBL>(wa, seg, ln);
    if (VAR == 0) {
      gemm_run<NBL>(acc, wa, seg, X, 32, ln);
      store_act_init<NBL, true>(acc, X, SAVE ? save : nullptr, h, bias);
    } else if (VAR == 1) {
      layer_tail<8, SAVE>(acc, wa, seg, X, save, h, bias, ln);
    } else if (VAR == 2) {
      layer_tail<4, SAVE>(acc, wa, seg, X, save, h, bias, ln);
    } else {
      gemm_run<NBL>(acc, wa, seg, X, 32, ln);
    }
  }
  float s = 0.0f;
  for (int nb = 0; nb < NBL; ++nb) s += acc[nb][0] + acc[nb][5];
  out[blockIdx.x * 256 + threadIdx.x] = s + X[0].x;
}

template <int VAR, bool SAVE>
static void run_layer(const char* name, const float* pk, const float* bias, float* out, float* act) {
  const size_t lds = 150 * 1024;
  const int grid = 256, n_layers = 256;
  hipFuncSetAttribute((const void*)klayer<VAR, SAVE>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  hipLaunchKernelGGL((klayer<VAR, SAVE>), dim3(grid), dim3(256), lds, 0, pk, bias, out, act, n_layers);
  hipDeviceSynchronize();
  hipEventRecord(e0, 0);
  for (int r = 0; r < 5; ++r) hipLaunchKernelGGL((klayer<VAR, SAVE>), dim3(grid), dim3(256), lds, 0, pk, bias, out, act, n_layers);
  hipEventRecord(e1, 0);
  hipEventSynchronize(e1);
  float ms = 0.0f;
  hipEventElapsedTime(&ms, e0, e1);
  ms /= 5;
  const double flop = (double)grid * 4 * n_layers * 32.0 * 2 * 256 * 256;
  float hs[4];
  hipMemcpy(hs, out, sizeof(hs), hipMemcpyDeviceToHost);
  printf("%-58s %8.3f ms  %7.1f TFLOP/s (%.1f %% of 157.3)  out[0] %.6e\n", name, ms, flop / ms / 1e9,
         flop / ms / 1e9 / 157.3 * 100, hs[0]);
}

int main() {
  float *pk, *out, *bias, *act;
  const size_t n = 8ull * 32 * 8 * 256;
  hipMalloc(&pk, n * 4);
  hipMalloc(&out, 1024 * 256 * 4);
  hipMalloc(&bias, 256 * 4);
  hipMalloc(&act, 256ull * 4 * 32 * 256 * 4);
  std::vector<float> h(n), hb(256);
  for (size_t i = 0; i < n; ++i) h[i] = 1.5e-4f * (float)((i * 2654435761u) % 2001) - 0.15f;
  for (int i = 0; i < 256; ++i) hb[i] = 0.01f * (float)(i % 17);
  hipMemcpy(pk, h.data(), n * 4, hipMemcpyHostToDevice);
  hipMemcpy(bias, hb.data(), 256 * 4, hipMemcpyHostToDevice);
  printf("-- 256 x 256 layer over 32 points per wave, one wave per SIMD (the out[0] column checks that the variants agree)\n");
  run_layer<3, false>("GEMM only (no epilogue)", pk, bias, out, act);
  run_layer<0, false>("eval: product layer (K-major GEMM, then epilogue)", pk, bias, out, act);
  run_layer<1, false>("eval: block-major tail of 8, epilogue under MFMAs", pk, bias, out, act);
  run_layer<2, false>("eval: block-major tail of 4, epilogue under MFMAs", pk, bias, out, act);
  run_layer<0, true>("train: product layer + activation store", pk, bias, out, act);
  run_layer<1, true>("train: block-major tail of 8 + activation store", pk, bias, out, act);
  run_layer<2, true>("train: block-major tail of 4 + activation store", pk, bias, out, act);
  return 0;
}
