// overlap_probe.hip -- does a bf16 MFMA stream hide VALU / LDS-read work on the same SIMD?  
//   hipcc -O3 --offload-arch=gfx950 tools/probes/overlap_probe.hip -o tools/probes/overlap_probe && tools/probes/overlap_probe
// Every variant runs ITER rounds per wave; a round is 16 MFMAs (v_mfma_f32_32x32x16_bf16, 4 independent accumulators)
// and/or 64 VALU ops (v_fma_f32, 8 independent chains) and/or 16 ds_read_b128.
//   mfma        one wave per SIMD, MFMAs only              valu / lds     the other stream alone
//   mfma+valu   one wave per SIMD, interleaved 1 : 4       mfma+lds       interleaved 1 : 1
//   mfma|valu   TWO waves per SIMD: waves 0-3 MFMAs only, waves 4-7 VALU only (same totals)      mfma|lds likewise
// If the streams overlapped, the combined variants would take max(a, b); if they serialise, a + b.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

#define ITER 4096

__device__ __forceinline__ void mfma_round(f32x16 (&acc)[4], const bf16x8& a, const bf16x8& b) {
#pragma unroll
  for (int i = 0; i < 16; ++i) acc[i & 3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[i & 3], 0, 0, 0);
}
__device__ __forceinline__ void valu_round(float (&x)[8], float p, float q) {
#pragma unroll
  for (int i = 0; i < 64; ++i) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x[i & 7]) : "v"(p), "v"(q));
}
__device__ __forceinline__ void lds_round(float4 (&r)[4], const char* base, int lane) {
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    float4 v;
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v) : "v"((unsigned)(size_t)(base + lane * 16)), "n"(0));
    asm volatile("s_waitcnt lgkmcnt(8)" ::: "memory");
    r[i & 3] = v;
  }
}

// peak probes: NCH independent accumulator chains of one MFMA shape, nothing else (SHAPE 0: 32x32x16 bf16, 1: 16x16x32 bf16,
// 2: 32x32x2 f32)
typedef float f32x4 __attribute__((ext_vector_type(4)));
template <int SHAPE, int NCH>
__global__ void peak(float* out) {
  const int lane = threadIdx.x & 63;
  bf16x8 a, b;
  for (int e = 0; e < 8; ++e) {
    a[e] = (__bf16)(0.001f * (lane + e));
    b[e] = (__bf16)(0.002f * (lane - e));
  }
  float s = 0.0f;
  if (SHAPE == 1) {
    f32x4 acc[NCH];
    for (int c = 0; c < NCH; ++c) acc[c] = f32x4{0, 0, 0, 0};
    for (int it = 0; it < ITER; ++it)
#pragma unroll
      for (int i = 0; i < 32; ++i) acc[i % NCH] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc[i % NCH], 0, 0, 0);
    for (int c = 0; c < NCH; ++c) s += acc[c][0];
  } else {
    f32x16 acc[NCH];
    for (int c = 0; c < NCH; ++c)
      for (int r = 0; r < 16; ++r) acc[c][r] = 0.0f;
    for (int it = 0; it < ITER; ++it)
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        if (SHAPE == 0)
          acc[i % NCH] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[i % NCH], 0, 0, 0);
        else
          acc[i % NCH] = __builtin_amdgcn_mfma_f32_32x32x2f32((float)a[0], (float)b[0], acc[i % NCH], 0, 0, 0);
      }
    for (int c = 0; c < NCH; ++c) s += acc[c][0] + acc[c][9];
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int SHAPE, int NCH>
static void run_peak(const char* name, int threads, float* out) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  hipLaunchKernelGGL((peak<SHAPE, NCH>), dim3(256), dim3(threads), 0, 0, out);
  hipDeviceSynchronize();
  hipEventRecord(e0, 0);
  for (int r = 0; r < 5; ++r) hipLaunchKernelGGL((peak<SHAPE, NCH>), dim3(256), dim3(threads), 0, 0, out);
  hipEventRecord(e1, 0);
  hipEventSynchronize(e1);
  float ms = 0.0f;
  hipEventElapsedTime(&ms, e0, e1);
  ms /= 5;
  const double per = SHAPE == 2 ? 32.0 * 32 * 2 * 2 : 32768.0;  // FLOP per MFMA (16x16x32: 16384, 32 per round)
  const double flop = 256.0 * (threads / 64) * ITER * (SHAPE == 1 ? 32 * 16384.0 : 16 * per);
  printf("%-22s %d chains %4d threads/CU  %8.3f ms  %8.1f TFLOP/s\n", name, NCH, threads, ms, flop / ms / 1e9);
}

// MODE: 0 mfma, 1 valu, 2 lds, 3 mfma+valu one wave, 4 mfma+lds one wave, 5 mfma|valu two waves, 6 mfma|lds two waves
template <int MODE>
__global__ void probe(float* out) {
  __shared__ __attribute__((aligned(16))) char smem[8192];
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  for (int i = threadIdx.x; i < 2048; i += blockDim.x) reinterpret_cast<float*>(smem)[i] = (float)i;
  __syncthreads();
  f32x16 acc[4];
  for (int c = 0; c < 4; ++c)
    for (int r = 0; r < 16; ++r) acc[c][r] = 0.0f;
  bf16x8 a, b;
  for (int e = 0; e < 8; ++e) {
    a[e] = (__bf16)(0.001f * (lane + e));
    b[e] = (__bf16)(0.002f * (lane - e));
  }
  float x[8];
  for (int i = 0; i < 8; ++i) x[i] = 0.5f + 0.01f * lane + i;
  float4 rr[4] = {};
  const float p = 0.999f, q = 1e-3f;
  const bool do_mfma = MODE == 0 || MODE == 3 || MODE == 4 || ((MODE == 5 || MODE == 6) && wid < 4);
  const bool do_valu = MODE == 1 || MODE == 3 || (MODE == 5 && wid >= 4);
  const bool do_lds = MODE == 2 || MODE == 4 || (MODE == 6 && wid >= 4);
  if (MODE == 3) {
    for (int it = 0; it < ITER; ++it) {
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        acc[i & 3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[i & 3], 0, 0, 0);
#pragma unroll
        for (int j = 0; j < 4; ++j) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x[(4 * i + j) & 7]) : "v"(p), "v"(q));
        __builtin_amdgcn_sched_barrier(0);
      }
    }
  } else if (MODE == 4) {
    for (int it = 0; it < ITER; ++it) {
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        acc[i & 3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[i & 3], 0, 0, 0);
        float4 v;
        asm volatile("ds_read_b128 %0, %1" : "=v"(v) : "v"((unsigned)(size_t)(smem + lane * 16)));
        asm volatile("s_waitcnt lgkmcnt(8)" ::: "memory");
        rr[i & 3] = v;
        __builtin_amdgcn_sched_barrier(0);
      }
    }
  } else {
    for (int it = 0; it < ITER; ++it) {
      if (do_mfma) mfma_round(acc, a, b);
      if (do_valu) valu_round(x, p, q);
      if (do_lds) lds_round(rr, smem, lane);
    }
  }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  float s = 0.0f;
  for (int c = 0; c < 4; ++c) s += acc[c][0] + acc[c][7] + rr[c].x + rr[c].w;
  for (int i = 0; i < 8; ++i) s += x[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int MODE>
static float run(const char* name, int threads, float* out) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  const int grid = 256;  // one workgroup per CU
  hipLaunchKernelGGL(probe<MODE>, dim3(grid), dim3(threads), 0, 0, out);
  hipDeviceSynchronize();
  hipEventRecord(e0, 0);
  for (int r = 0; r < 5; ++r) hipLaunchKernelGGL(probe<MODE>, dim3(grid), dim3(threads), 0, 0, out);
  hipEventRecord(e1, 0);
  hipEventSynchronize(e1);
  float ms = 0.0f;
  hipEventElapsedTime(&ms, e0, e1);
  ms /= 5;
  printf("%-12s %4d threads/CU  %8.3f ms   %.1f cycles per round and SIMD at 2.0 GHz\n", name, threads, ms,
         ms * 1e-3 * 2.0e9 / ITER);
  return ms;
}

int main() {
  float* out;
  hipMalloc(&out, 256 * 512 * sizeof(float));
  run_peak<0, 4>("32x32x16 bf16", 256, out);
  run_peak<0, 8>("32x32x16 bf16", 256, out);
  run_peak<0, 4>("32x32x16 bf16", 512, out);
  run_peak<1, 4>("16x16x32 bf16", 256, out);
  run_peak<1, 8>("16x16x32 bf16", 256, out);
  run_peak<1, 16>("16x16x32 bf16", 256, out);
  run_peak<1, 8>("16x16x32 bf16", 512, out);
  run_peak<2, 4>("32x32x2 f32", 256, out);
  run_peak<2, 8>("32x32x2 f32", 256, out);
  run_peak<2, 4>("32x32x2 f32", 512, out);
  run<0>("mfma", 256, out);
  run<1>("valu", 256, out);
  run<2>("lds", 256, out);
  run<3>("mfma+valu", 256, out);
  run<4>("mfma+lds", 256, out);
  run<5>("mfma|valu", 512, out);
  run<6>("mfma|lds", 512, out);
  hipFree(out);
  return 0;
}
