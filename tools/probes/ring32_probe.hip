// ring32_probe.hip -- would a workgroup-shared weight stream (LDS-DMA ring, as the bf16 ring kernels have) lift the
// exact-fp32 K loops?  Today every wave streams its own copy of the packed weights from L2 (16 B/clk/CU; the loops run at
// 89-91 % of the MFMA rate in isolation, 98 % without the stream: tools/probes/gemm_occ_probe.hip).  The product kernels keep
// 128 KiB of activation slabs + 20 KiB of stashes in LDS, so a ring could have 12 KiB: three 4-KiB slots (half a
// K-iteration: four 1-KiB fragments = 16 MFMAs = 1,024 MFMA cycles per slot, one workgroup barrier per slot).
// Measured here: the K loop alone (256 x 256 GEMMs over 32 points per wave, X from the wave's LDS slab), weights
//   (a) per wave from L2 (the product loop),
//   (b) through a 3-slot ring of 4-KiB groups (fits the product's LDS),
//   (c) through a 3-slot ring of 8-KiB groups (would need 12 KiB more LDS than the product has).
// Each wave DMAs its quarter of a group (global_load_lds_dwordx4), two groups ahead; after the barrier of group g the
// wave reads g's fragments into registers and issues the MFMAs of group g-1 (so no LDS latency is exposed).
//   hipcc -O3 --offload-arch=gfx950 -ffp-contract=off -std=c++17 -I include -I reflect_sampling_nerf_amd/csrc \
//         tools/probes/ring32_probe.hip -o build/ring32_probe && build/ring32_probe
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "rsn_mfma.h"
void rsn_set_error(const char*, ...) {}

#define GEMMS 64          // 256 x 256 GEMMs per wave and launch
#define STREAM_SEGS 8     // the stream: 8 packed 256 x 256 segments = 2 MiB, walked cyclically

__device__ __forceinline__ void glds16(const void* gbase, unsigned voff, unsigned lds_dst) {
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
               : "=&s"(keep)
               : "v"(voff), "s"(gbase), "s"(lds_dst)
               : "memory");
}

template <int GF>
struct Frag {
  float4 w[GF];
};

template <int GF, int NB0>
__device__ __forceinline__ void mma_group(f32x16 (&acc)[8], const Frag<GF>& R, const float4 b) {
#pragma unroll
  for (int f = 0; f < GF; ++f) acc[NB0 + f] = __builtin_amdgcn_mfma_f32_32x32x2f32(R.w[f].x, b.x, acc[NB0 + f], 0, 0, 0);
#pragma unroll
  for (int f = 0; f < GF; ++f) acc[NB0 + f] = __builtin_amdgcn_mfma_f32_32x32x2f32(R.w[f].y, b.y, acc[NB0 + f], 0, 0, 0);
#pragma unroll
  for (int f = 0; f < GF; ++f) acc[NB0 + f] = __builtin_amdgcn_mfma_f32_32x32x2f32(R.w[f].z, b.z, acc[NB0 + f], 0, 0, 0);
#pragma unroll
  for (int f = 0; f < GF; ++f) acc[NB0 + f] = __builtin_amdgcn_mfma_f32_32x32x2f32(R.w[f].w, b.w, acc[NB0 + f], 0, 0, 0);
}

template <int GF, int SLOTS>
struct RingState {
  const char* src;      // stream base + this wave's quarter of a group
  unsigned lane16;
  unsigned lds_dst;     // LDS byte address of this wave's quarter inside slot 0
  unsigned rd_base;     // LDS byte address of this lane's 16 B of fragment 0 of slot 0
  int n_groups;
  int issue_grp, issue_slot, rd_slot;
};

template <int GF, int SLOTS>
__device__ __forceinline__ void ring_issue(RingState<GF, SLOTS>& r) {
  constexpr int PPW = GF / 4;
  const char* g = r.src + (size_t)r.issue_grp * (GF * 1024);
  const unsigned d = __builtin_amdgcn_readfirstlane(r.lds_dst + (unsigned)r.issue_slot * (GF * 1024));
#pragma unroll
  for (int i = 0; i < PPW; ++i) glds16(g + i * 1024, r.lane16, d + i * 1024);
  r.issue_grp = (r.issue_grp + 1 == r.n_groups) ? 0 : r.issue_grp + 1;
  r.issue_slot = (r.issue_slot + 1 == SLOTS) ? 0 : r.issue_slot + 1;
}

// boundary of group g: g has landed for every wave (own share: all but the youngest group's pieces; the others' by the
// barrier), every wave has finished reading g-1 (its reads were issued one group ago), so g-1's slot takes group g+2;
// then this wave reads g's fragments.
template <int GF, int SLOTS>
__device__ __forceinline__ void ring_step(RingState<GF, SLOTS>& r, Frag<GF>& R, const char* smem) {
  asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)\n\ts_barrier" ::"n"((GF / 4) * (SLOTS - 2)) : "memory");
  ring_issue<GF, SLOTS>(r);
  const unsigned rd = r.rd_base + (unsigned)r.rd_slot * (GF * 1024);
#pragma unroll
  for (int f = 0; f < GF; ++f) R.w[f] = *reinterpret_cast<const float4*>(smem + rd + f * 1024);
  r.rd_slot = (r.rd_slot + 1 == SLOTS) ? 0 : r.rd_slot + 1;
}

template <int GF, int SLOTS>
__global__ __launch_bounds__(256, 1) void kring(const float* __restrict__ pk, float* out) {
  extern __shared__ __attribute__((aligned(1024))) char smem[];
  const int lane = threadIdx.x & 63;
  const int wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  float4* X = reinterpret_cast<float4*>(smem) + wid * 32 * 64 + lane;
  for (int it = 0; it < 32; ++it) X[it * 64] = make_float4(0.001f * lane, 0.002f * it, 1.0f, -1.0f);
  constexpr unsigned RING0 = 4 * 32 * 1024;
  constexpr int PPW = GF / 4;
  RingState<GF, SLOTS> r;
  r.n_groups = STREAM_SEGS * 32 * 8 / GF;
  r.src = reinterpret_cast<const char*>(pk) + wid * PPW * 1024;
  r.lane16 = lane * 16;
  r.lds_dst = RING0 + wid * PPW * 1024;
  r.rd_base = RING0 + lane * 16;
  r.issue_grp = (int)((blockIdx.x * 5u) & 7u) * (32 * 8 / GF);  // workgroups start on different segments
  r.issue_slot = 0;
  r.rd_slot = 0;
  __syncthreads();
#pragma unroll
  for (int i = 0; i < SLOTS - 1; ++i) ring_issue<GF, SLOTS>(r);  // LEAD = SLOTS - 1 groups in flight
  f32x16 acc[8];
  zero_acc<8>(acc);
  Frag<GF> Ra, Rb;
  float4 bp, bc;
  // pipeline prologue: group 0 of the first GEMM is read; its MFMAs follow the next group's boundary
  if (GF == 8) {
    ring_step<GF, SLOTS>(r, Rb, smem);
    bp = X[0];
  } else {
    ring_step<GF, SLOTS>(r, Ra, smem);
    bp = X[0];
    ring_step<GF, SLOTS>(r, Rb, smem);
    mma_group<GF, 0>(acc, Ra, bp);
  }
  // steady state: one K-iteration = 8 fragments.  GF = 8: groups alternate between Ra and Rb (two iterations per trip).
  // GF = 4: half 0 of iteration it goes to Ra, half 1 to Rb; the MFMAs run one group behind the reads.
  const int n_it_total = GEMMS * 32;
#pragma unroll 1
  for (int t = 1; t + 1 < n_it_total; t += (GF == 8 ? 2 : 1)) {
    if (GF == 8) {
      ring_step<GF, SLOTS>(r, Ra, smem);
      bc = X[(t & 31) * 64];
      mma_group<GF, 0>(acc, Rb, bp);
      __builtin_amdgcn_sched_barrier(0);
      ring_step<GF, SLOTS>(r, Rb, smem);
      bp = X[((t + 1) & 31) * 64];
      mma_group<GF, 0>(acc, Ra, bc);
      __builtin_amdgcn_sched_barrier(0);
    } else {
      ring_step<GF, SLOTS>(r, Ra, smem);
      bc = X[(t & 31) * 64];
      mma_group<GF, 4>(acc, Rb, bp);  // half 1 of iteration t-1
      __builtin_amdgcn_sched_barrier(0);
      ring_step<GF, SLOTS>(r, Rb, smem);
      mma_group<GF, 0>(acc, Ra, bc);  // half 0 of iteration t
      __builtin_amdgcn_sched_barrier(0);
      bp = bc;
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // no DMA may land after the workgroup has left
  __syncthreads();
  float s = 0.0f;
  for (int nb = 0; nb < 8; ++nb) s += acc[nb][0] + acc[nb][5];
  out[blockIdx.x * 256 + threadIdx.x] = s + Ra.w[0].x + Rb.w[0].y;
}

__global__ __launch_bounds__(256, 1) void kwave(const float* __restrict__ pk, float* out) {
  extern __shared__ __attribute__((aligned(1024))) char smem[];
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  float4* X = reinterpret_cast<float4*>(smem) + wid * 32 * 64 + lane;
  for (int it = 0; it < 32; ++it) X[it * 64] = make_float4(0.001f * lane, 0.002f * it, 1.0f, -1.0f);
  f32x16 acc[8];
  zero_acc<8>(acc);
  for (int l = 0; l < GEMMS; ++l) {
    int ln = lane;
    asm volatile("" : "+v"(ln));
    const float* seg = pk + (size_t)((l + blockIdx.x * 5) & 7) * (32 * 8 * 256);
    gemm<8, 8>(acc, seg, X, 32, ln);
  }
  float s = 0.0f;
  for (int nb = 0; nb < 8; ++nb) s += acc[nb][0] + acc[nb][5];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <class K>
static void timeit(const char* name, K kern, size_t lds, const float* pk, float* out) {
  hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  hipLaunchKernelGGL(kern, dim3(256), dim3(256), lds, 0, pk, out);
  if (hipDeviceSynchronize() != hipSuccess) { printf("%s: launch failed\n", name); exit(1); }
  hipEventRecord(e0, 0);
  for (int r = 0; r < 5; ++r) hipLaunchKernelGGL(kern, dim3(256), dim3(256), lds, 0, pk, out);
  hipEventRecord(e1, 0);
  hipEventSynchronize(e1);
  float ms = 0.0f;
  hipEventElapsedTime(&ms, e0, e1);
  ms /= 5;
  const double flop = 256.0 * 4 * GEMMS * 32.0 * 2 * 256 * 256;
  printf("%-64s %8.3f ms  %7.1f TFLOP/s (%.1f %% of 157.3)\n", name, ms, flop / ms / 1e9, flop / ms / 1e9 / 157.3 * 100);
}

int main() {
  float *pk, *out;
  const size_t n = (size_t)STREAM_SEGS * 32 * 8 * 256;
  hipMalloc(&pk, n * 4);
  hipMalloc(&out, 1024 * 256 * 4);
  std::vector<float> h(n);
  for (size_t i = 0; i < n; ++i) h[i] = 1e-3f * (float)((i * 2654435761u) % 2001) - 1.0f;
  hipMemcpy(pk, h.data(), n * 4, hipMemcpyHostToDevice);
  printf("-- K loop alone, 256 x 256 GEMMs over 32 points per wave, one 4-wave workgroup per CU, 148+ KiB of LDS\n");
  timeit("(a) per-wave weight stream from L2 (product loop)", kwave, 150 * 1024, pk, out);
  timeit("(b) shared ring, 3 slots x 4 KiB (fits the product's LDS)", kring<4, 3>, 128 * 1024 + 12 * 1024, pk, out);
  timeit("(c) shared ring, 3 slots x 8 KiB", kring<8, 3>, 128 * 1024 + 24 * 1024, pk, out);
  timeit("(d) shared ring, 4 slots x 4 KiB (3 groups ahead)", kring<4, 4>, 128 * 1024 + 16 * 1024, pk, out);
  return 0;
}
