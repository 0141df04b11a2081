// gemm_occ_probe.hip -- would TWO waves per SIMD hide the MFMA-free phases of the exact-fp32 field kernels?  (round 3:
// "wave pairs" -- two waves share one 32-point tile and its LDS slab, each owning half of the output blocks, two
// 4-wave workgroups per CU -- were built into rsn_field_bwd_kernel, passed the parity tests and ran 5.05 ms against
// 4.97 ms for the one-wave-per-SIMD kernel; this probe isolates why.)
// Part 1/2: the K loop alone (rsn_mfma.h: gemm<NBL, NBT>, weights streamed from an L2-resident packed segment, X from an
// LDS slab, no epilogue, no barriers): NBL output blocks per wave (8 = the product kernels, 4 = wave pairs), 1 or 2
// workgroups per CU, the product's double-buffered loop against a block-major single-buffer loop (fewer bytes in
// flight), with the loads removed one at a time.  Part 3: GEMM + the ReLU / LDS / activation-store epilogue per layer.
// Result (profiles/r03_gemm_occ_probe.txt): with eight waves per CU the weight stream from L2 costs the loop 13-35 %
// (98 % without it, 96 % when it hits L1), and the epilogue is only half hidden: 78 % against 73.6 % with one wave per
// SIMD, whatever the wave priorities.  Not worth the barriers: the product kernels stay at one wave per SIMD.
//   hipcc -O3 --offload-arch=gfx950 -ffp-contract=off -std=c++17 -I include -I reflect_sampling_nerf_amd/csrc \
//         tools/probes/gemm_occ_probe.hip -o build/gemm_occ_probe && build/gemm_occ_probe
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "rsn_mfma.h"
void rsn_set_error(const char*, ...) {}

// LDS-only barrier of a workgroup (weight prefetches and activation stores stay in flight across it)
template <bool PAIR>
__device__ __forceinline__ void pair_sync() {
  if (PAIR) asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

#define LAYERS 64  // GEMMs (256 x 256 over 32 points) per wave and launch

// the K loop of rsn_mfma.h (gemm_run) with switches: VAR bit 0: no weight loads (registers), bit 1: no LDS reads,
// bit 2: every weight load from the same 8 KiB (L1-resident), bit 3: no sched_group_barrier / sched_barrier pinning
template <int NBO, int NBT, int VAR>
__device__ __forceinline__ void gemm_var(f32x16 (&acc)[NBO], const float* __restrict__ wseg, const float4* xl, int n_it, int lane) {
  const float4* __restrict__ wp = reinterpret_cast<const float4*>(wseg) + lane;
  float4 wa[NBO], wb[NBO];
  float4 ba, bb;
  auto ldw = [&](float4 (&w)[NBO], int it) {
    if (VAR & 1) return;
#pragma unroll
    for (int nb = 0; nb < NBO; ++nb) w[nb] = wp[(((VAR & 4) ? 0 : it) * NBT + nb) * 64];
  };
  auto ldx = [&](int it) { return (VAR & 2) ? make_float4(1.0f, 2.0f, 3.0f, 4.0f) : xl[it * 64]; };
#pragma unroll
  for (int nb = 0; nb < NBO; ++nb) wa[nb] = wb[nb] = make_float4(0.5f, 0.25f, -0.5f, 1.0f);
  ldw(wa, 0);
  ba = ldx(0);
  int it = 0;
#pragma unroll 1
  for (; it + 1 < n_it; it += 2) {
    ldw(wb, it + 1);
    bb = ldx(it + 1);
    mma4<NBO>(acc, wa, ba);
    if (!(VAR & 8)) { interleave_loads<NBO>(); __builtin_amdgcn_sched_barrier(0); }
    {
      const int in = (it + 2 < n_it) ? it + 2 : n_it - 1;
      ldw(wa, in);
      ba = ldx(in);
    }
    mma4<NBO>(acc, wb, bb);
    if (!(VAR & 8)) { interleave_loads<NBO>(); __builtin_amdgcn_sched_barrier(0); }
  }
}

// block-major K loop, ONE weight buffer: the four K-steps of a block run back to back (dependent MFMAs: latency = issue
// interval for 32x32x2 f32), and the block's fragment of the NEXT K-iteration is requested right behind them, 3 blocks
// (12 MFMAs) ahead of its use: at most 3 KiB per wave in flight instead of 4-8.
template <int NBO, int NBT, int LAG = 0>
__device__ __forceinline__ void gemm_bm(f32x16 (&acc)[NBO], const float* __restrict__ wseg, const float4* xl, int n_it, int lane) {
  const float4* __restrict__ wp = reinterpret_cast<const float4*>(wseg) + lane;
  float4 w[NBO];
#pragma unroll
  for (int nb = 0; nb < NBO; ++nb) w[nb] = wp[nb * 64];
  float4 b = xl[0], bn;
#pragma unroll 1
  for (int it = 0; it < n_it; ++it) {
    const int in = (it + 1 < n_it) ? it + 1 : it;
    bn = xl[in * 64];
#pragma unroll
    for (int nb = 0; nb < NBO; ++nb) {
      acc[nb] = __builtin_amdgcn_mfma_f32_32x32x2f32(w[nb].x, b.x, acc[nb], 0, 0, 0);
      acc[nb] = __builtin_amdgcn_mfma_f32_32x32x2f32(w[nb].y, b.y, acc[nb], 0, 0, 0);
      acc[nb] = __builtin_amdgcn_mfma_f32_32x32x2f32(w[nb].z, b.z, acc[nb], 0, 0, 0);
      acc[nb] = __builtin_amdgcn_mfma_f32_32x32x2f32(w[nb].w, b.w, acc[nb], 0, 0, 0);
      // reload the fragment of block nb - LAG (its MFMAs of this iteration were issued LAG blocks ago)
      if (nb >= LAG) w[nb - LAG] = wp[(in * NBT + nb - LAG) * 64];
      __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);
      if (nb >= LAG) __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
    }
#pragma unroll
    for (int t = NBO - LAG; t < NBO; ++t) w[t] = wp[(in * NBT + t) * 64];  // the last LAG blocks: behind the iteration
    __builtin_amdgcn_sched_barrier(0);
    b = bn;
  }
}

template <int NBL, int WPS, int VAR, int DESYNC = 0>
__global__ __launch_bounds__(256, WPS) void k(const float* __restrict__ pk, float* out, int pad_unused) {
  // 32 KiB slab per wave; WPS = 1: 4 slabs + padding so that only one workgroup fits a CU
  extern __shared__ float4 smem[];
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  float4* X = smem + (NBL == 8 ? wid : (wid >> 1)) * 32 * 64 + lane;  // wave pairs share their slab
  for (int it = 0; it < 32; ++it) X[it * 64] = make_float4(0.001f * lane, 0.002f * it, 1.0f, -1.0f);
  f32x16 acc[NBL];
  zero_acc<NBL>(acc);
  const int nb0 = (NBL == 8) ? 0 : (wid & 1) * NBL;
  for (int l = 0; l < LAYERS; ++l) {
    int ln = lane;
    asm volatile("" : "+v"(ln));
    const int ph = DESYNC == 0 ? 0 : (DESYNC == 1 ? (wid >> 1) * 3 + blockIdx.x : wid * 3 + blockIdx.x * 5);
    const float* seg = pk + (size_t)((l + ph) & 7) * (32 * 8 * 256) + nb0 * 256;
    if (VAR == 0) gemm<NBL, 8>(acc, seg, X, 32, ln); else if (VAR >= 16) gemm_bm<NBL, 8, VAR - 16>(acc, seg, X, 32, ln); else gemm_var<NBL, 8, VAR>(acc, seg, X, 32, ln);
  }
  float s = 0.0f;
  for (int nb = 0; nb < NBL; ++nb) s += acc[nb][0] + acc[nb][5];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int NBL, int WPS, int VAR, int DESYNC = 0>
static void run(const char* name, const float* pk, float* out, int grid) {
  const size_t lds = WPS == 1 ? 150 * 1024 : 64 * 1024;
  hipFuncSetAttribute((const void*)k<NBL, WPS, VAR, DESYNC>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  hipLaunchKernelGGL((k<NBL, WPS, VAR, DESYNC>), dim3(grid), dim3(256), lds, 0, pk, out, 0);
  hipDeviceSynchronize();
  hipEventRecord(e0, 0);
  for (int r = 0; r < 5; ++r) hipLaunchKernelGGL((k<NBL, WPS, VAR, DESYNC>), dim3(grid), dim3(256), lds, 0, pk, out, 0);
  hipEventRecord(e1, 0);
  hipEventSynchronize(e1);
  float ms = 0.0f;
  hipEventElapsedTime(&ms, e0, e1);
  ms /= 5;
  const double flop = (double)grid * 4 * LAYERS * 32.0 * NBL * 4 * 4096.0;  // MFMAs x 4096 FLOP
  printf("%-44s grid %4d  %8.3f ms  %7.1f TFLOP/s (%.1f %% of 157.3)\n", name, grid, ms, flop / ms / 1e9,
         flop / ms / 1e9 / 157.3 * 100);
}

// ---- layer = GEMM + MFMA-free epilogue (ReLU, LDS hand-off, activation store): does a second wave per SIMD hide the
// epilogue?  PRIO 0: nothing; 1: clock-window priority (the two waves of a SIMD take turns being prio 1, 2048-cycle windows);
// 2: prio 1 while in the epilogue only.
template <int WPS, int PRIO>
__global__ __launch_bounds__(256, WPS) void klayer(const float* __restrict__ pk, float* out, float* act, int n_layers) {
  extern __shared__ float4 smem[];
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  float4* X = smem + (wid >> 1) * 32 * 64 + lane;
  for (int it = 0; it < 32; ++it) X[it * 64] = make_float4(0.001f * lane, 0.002f * it, 1.0f, -1.0f);
  __syncthreads();
  const int nb0 = (wid & 1) * 4;
  const unsigned slot = __builtin_amdgcn_s_getreg((4) | (0 << 6) | (3 << 11)) & 1u;  // HW_ID.wave_id parity
  f32x16 acc[4];
  for (int l = 0; l < n_layers; ++l) {
    int ln = lane;
    asm volatile("" : "+v"(ln));
    const int ph = (wid >> 1) * 3 + blockIdx.x;
    const float* seg = pk + (size_t)((l + ph) & 7) * (32 * 8 * 256) + nb0 * 256;
    zero_acc<4>(acc);
    // K loop (block-major, single buffer) with the priority decision once per K-iteration
    {
      const float4* __restrict__ wp = reinterpret_cast<const float4*>(seg) + ln;
      float4 w[4];
#pragma unroll
      for (int nb = 0; nb < 4; ++nb) w[nb] = wp[nb * 64];
      float4 b = X[0], bn;
#pragma unroll 1
      for (int it = 0; it < 32; ++it) {
        if (PRIO == 1) {
          const unsigned long long t = __builtin_readcyclecounter();
          if ((((unsigned)(t >> 11)) & 1u) == slot) __builtin_amdgcn_s_setprio(1); else __builtin_amdgcn_s_setprio(0);
        }
        const int in = (it + 1 < 32) ? it + 1 : it;
        bn = X[in * 64];
#pragma unroll
        for (int nb = 0; nb < 4; ++nb) {
          acc[nb] = __builtin_amdgcn_mfma_f32_32x32x2f32(w[nb].x, b.x, acc[nb], 0, 0, 0);
          acc[nb] = __builtin_amdgcn_mfma_f32_32x32x2f32(w[nb].y, b.y, acc[nb], 0, 0, 0);
          acc[nb] = __builtin_amdgcn_mfma_f32_32x32x2f32(w[nb].z, b.z, acc[nb], 0, 0, 0);
          acc[nb] = __builtin_amdgcn_mfma_f32_32x32x2f32(w[nb].w, b.w, acc[nb], 0, 0, 0);
          w[nb] = wp[(in * 8 + nb) * 64];
          __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);
          __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
        b = bn;
      }
    }
    if (PRIO == 2) __builtin_amdgcn_s_setprio(1);
    if (PRIO == 1) __builtin_amdgcn_s_setprio(1);
    pair_sync<true>();
    // epilogue: ReLU, LDS hand-off, row-major activation store (what store_act does)
    float* save = act + ((size_t)blockIdx.x * 4 + wid) * 32 * 256 + (size_t)(ln & 31) * 256 + nb0 * 32;
    store_act<4, 4, true>(acc, X + nb0 * 256, save, ln >> 5);
    pair_sync<true>();
    if (PRIO == 2) __builtin_amdgcn_s_setprio(0);
  }
  float s = 0.0f;
  for (int nb = 0; nb < 4; ++nb) s += acc[nb][0] + acc[nb][5];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int WPS, int PRIO>
static void run_layer(const char* name, const float* pk, float* out, float* act, int grid) {
  const size_t lds = WPS == 1 ? 150 * 1024 : 64 * 1024;
  const int total_layers = 128 * 512;  // the same work for every grid
  const int n_layers = total_layers / grid;
  hipFuncSetAttribute((const void*)klayer<WPS, PRIO>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  hipLaunchKernelGGL((klayer<WPS, PRIO>), dim3(grid), dim3(256), lds, 0, pk, out, act, n_layers);
  hipDeviceSynchronize();
  hipEventRecord(e0, 0);
  for (int r = 0; r < 5; ++r) hipLaunchKernelGGL((klayer<WPS, PRIO>), dim3(grid), dim3(256), lds, 0, pk, out, act, n_layers);
  hipEventRecord(e1, 0);
  hipEventSynchronize(e1);
  float ms = 0.0f;
  hipEventElapsedTime(&ms, e0, e1);
  ms /= 5;
  const double flop = (double)grid * 4 * n_layers * 32.0 * 4 * 4 * 4096.0;
  printf("%-52s grid %4d  %8.3f ms  %7.1f TFLOP/s (%.1f %% of 157.3)\n", name, grid, ms, flop / ms / 1e9,
         flop / ms / 1e9 / 157.3 * 100);
}

int main() {
  float *pk, *out;
  const size_t n = 8ull * 32 * 8 * 256;
  hipMalloc(&pk, n * 4);
  hipMalloc(&out, 1024 * 256 * 4);
  std::vector<float> h(n);
  for (size_t i = 0; i < n; ++i) h[i] = 1e-3f * (float)((i * 2654435761u) % 2001) - 1.0f;
  hipMemcpy(pk, h.data(), n * 4, hipMemcpyHostToDevice);
  printf("-- waves in lockstep on the same weights (L1 shares their fetches)\n");
  run<8, 1, 0>("8 blocks, 1 wave/SIMD, product loop", pk, out, 256);
  run<8, 1, 16>("8 blocks, 1 wave/SIMD, block-major lag 0", pk, out, 256);
  run<4, 2, 0>("4 blocks, 2 waves/SIMD, product loop", pk, out, 512);
  run<4, 2, 16>("4 blocks, 2 waves/SIMD, block-major lag 0", pk, out, 512);
  printf("-- every wave (8-block) / wave pair (4-block) on its own phase of the weight set: no sharing\n");
  run<8, 1, 0, 2>("8 blocks, 1 wave/SIMD, product loop", pk, out, 256);
  run<8, 1, 16, 2>("8 blocks, 1 wave/SIMD, block-major lag 0", pk, out, 256);
  run<8, 1, 18, 2>("8 blocks, 1 wave/SIMD, block-major lag 2", pk, out, 256);
  run<8, 1, 20, 2>("8 blocks, 1 wave/SIMD, block-major lag 4", pk, out, 256);
  run<8, 1, 22, 2>("8 blocks, 1 wave/SIMD, block-major lag 6", pk, out, 256);
  run<4, 1, 0, 1>("4 blocks, 1 wave/SIMD, product loop", pk, out, 256);
  run<4, 1, 16, 1>("4 blocks, 1 wave/SIMD, block-major lag 0", pk, out, 256);
  run<4, 2, 0, 1>("4 blocks, 2 waves/SIMD, product loop", pk, out, 512);
  run<4, 2, 16, 1>("4 blocks, 2 waves/SIMD, block-major lag 0", pk, out, 512);
  run<4, 2, 17, 1>("4 blocks, 2 waves/SIMD, block-major lag 1", pk, out, 512);
  run<4, 2, 18, 1>("4 blocks, 2 waves/SIMD, block-major lag 2", pk, out, 512);
  run<4, 2, 1, 1>("4 blocks, 2 waves/SIMD, no weight loads", pk, out, 512);
  printf("-- GEMM + epilogue per layer (wave pairs, 2 pairs per workgroup)\n");
  float* act;
  hipMalloc(&act, 512ull * 4 * 32 * 256 * 4);
  run_layer<1, 0>("1 wave/SIMD (1 WG/CU)", pk, out, act, 256);
  run_layer<2, 0>("2 waves/SIMD (2 WG/CU), age arbitration", pk, out, act, 512);
  run_layer<2, 1>("2 waves/SIMD, clock-window priority", pk, out, act, 512);
  run_layer<2, 2>("2 waves/SIMD, prio 1 in the epilogue", pk, out, act, 512);
  return 0;
}
