// ring16_probe.hip -- what is the ceiling of the LDS-ring GEMM loop of rsn_field_bf16_ring16_kernel (BASELINE configs[3])?
//
// the kernel sits at 0.52-0.54 of the 2.5 PF dense bf16 peak; either lift it to 0.60 or show, with a
// stand-alone probe of the SAME per-MFMA instruction mix, what the loop itself can reach on this box.  This probe runs the
// product's own machinery (rsn_ring16.h: the 4-slot LDS-DMA ring, the counted wait + one s_barrier per 16-fragment group,
// the 4-deep ds_read_b128 FIFO, two v_mfma_f32_16x16x32_bf16 per fragment) on one 8-wave workgroup per CU with the product's
// LDS footprint (141 KiB: one workgroup per CU, two waves per SIMD), and nothing else:
//   A  pure MFMA: the same accumulator / operand registers, no ring, no LDS read          (the matrix pipe's own rate here)
//   B  the ring GEMM loop alone: 256 x 256 layers back to back, the same X operands re-used (no epilogue)
//   C  the split-bf16 (bf16x6) loop of rsn_field_x6_train.hip: three 1 KiB pieces per fragment, 1 / 2 / 3 MFMAs per piece, one 16-point
//      half per wave (-DSTREAM_GROUPS=n: stream length in 16 KiB groups; 432 = the 7 MB of the split-bf16 forward + transposed stream)
// Output: TFLOP/s of A and B and B / 2500: B is the roofline of the kernel's GEMM structure on this box -- what the product kernel
// would reach with free layer hand-offs, encode, heads, SH and stores.  (Measured, profiles/r04_ring16_probe.txt: A 2,262 TFLOP/s
// = 0.90 of the dense peak, B 1,961 TFLOP/s = 0.78; the product kernel's 1,313 TFLOP/s is 67 % of B: the loop is NOT what holds
// the kernel at 0.52 -- its MFMA-free phases are, which two lockstep waves per SIMD cannot overlap)
//   hipcc -O3 --offload-arch=gfx950 -ffp-contract=off -std=c++17 -I include -I reflect_sampling_nerf_amd/csrc \
//         tools/probes/ring16_probe.hip -o build/ring16_probe && build/ring16_probe
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "rsn_ring16.h"
void rsn_set_error(const char*, ...) {}

#define LAYERS 63             // 256 x 256 layers per tile-pass (the stream wraps around a 7-layer block)
#ifndef STREAM_GROUPS
#define STREAM_GROUPS (7 * 8)  // 7 layers x 128 fragments = 56 groups of 16 KiB: 0.9 MB, L2-resident like the product's 1.3 MB
#endif

template <int variant>
__global__ __launch_bounds__(512, 2) void probe_kernel(const float* __restrict__ stream, float* __restrict__ out, int passes) {
  __shared__ __attribute__((aligned(1024))) char smem[R16_LDS_BYTES];
  const int lane = threadIdx.x & 63;
  const int wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  float* bias = reinterpret_cast<float*>(smem + RingCfg<8>::RING_BYTES + 8 * R16_STASH_BYTES);
  for (int i = threadIdx.x; i < 256; i += 512) bias[i] = 0.001f * (float)(i & 15);
  const int g = lane >> 4;
  bf16x8 X[8][2];
#pragma unroll
  for (int kk = 0; kk < 8; ++kk)
#pragma unroll
    for (int p = 0; p < 2; ++p) {
      const float v[8] = {0.01f * lane, 0.02f, 0.03f * kk, 0.01f, 0.02f * p, 0.01f, 0.03f, 0.02f};
      X[kk][p] = pack8(v);
    }
  f32x4 acc[16][2];
  init_acc16<16>(acc, bias, g);
  if (variant == 0) {  // A: pure MFMA, same register file pressure
    bf16x8 wa = X[0][0];
    for (int ps = 0; ps < passes; ++ps)
#pragma unroll 1
      for (int l = 0; l < LAYERS; ++l) {
#pragma unroll
        for (int i = 0; i < 128; ++i) {
          const int kk = i / 16, b = i % 16;
          acc[b][0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wa, X[kk][0], acc[b][0], 0, 0, 0);
          acc[b][1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wa, X[kk][1], acc[b][1], 0, 0, 0);
        }
      }
  } else if (variant == 2) {
    // C: the split-bf16 (bf16x6) ring loop: every fp32 weight fragment travels as three bf16 pieces (hi, mid, lo: 3 KiB), a wave
    // holds ONE 16-point half with its activations split three ways (96 VGPRs) and issues hi x {hi, mid, lo}, mid x {hi, mid},
    // lo x hi = 6 MFMAs per fragment: the same two MFMAs per 1 KiB ds_read_b128 as the plain-bf16 loop
    Ring r;
    r.src = reinterpret_cast<const char*>(stream) + wid * (RingCfg<8>::PPW * 1024);
    r.lane16 = (unsigned)lane * 16u;
    r.lds_dst = (unsigned)(size_t)(__attribute__((address_space(3))) char*)smem + (unsigned)wid * (RingCfg<8>::PPW * 1024u);
    r.n_groups = STREAM_GROUPS;
    r.issue_grp = 0;
    r.issue_slot = 0;
    r.rd_base = (unsigned)lane * 16u;
    r.next_slot = 0;
    r.rd_next = r.rd_base;
    r.rd_cur = r.rd_base;
    __syncthreads();
#pragma unroll
    for (int gq = 0; gq < RingCfg<8>::LEAD; ++gq) ring_issue<8>(r);
    asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" ::"n"(RingCfg<8>::PPW * (RingCfg<8>::LEAD - 1)) : "memory");
    bf16x8 Wf[RING_FIFO];
#pragma unroll
    for (int j = 0; j < RING_FIFO; ++j) Wf[j] = *reinterpret_cast<const bf16x8*>(smem + r.rd_next + j * 1024);
    bf16x8 X3[3][8];
#pragma unroll
    for (int s3 = 0; s3 < 3; ++s3)
#pragma unroll
      for (int kk = 0; kk < 8; ++kk) X3[s3][kk] = X[(kk + s3) & 7][s3 & 1];
    f32x4 a1[16];
#pragma unroll
    for (int b = 0; b < 16; ++b) a1[b] = acc[b][0];
    for (int ps = 0; ps < passes; ++ps)
#pragma unroll 1
      for (int l = 0; l < LAYERS / 3; ++l) {
#pragma unroll
        for (int gi = 0; gi < 24; ++gi) {
          ring_sync<8>(r);
#pragma unroll
          for (int f = 0; f < 16; ++f) {
            const int i = gi * 16 + f;
            const int fr = i / 3, s3 = i % 3, kk = fr / 16, b = fr % 16;
            const bf16x8 wa = Wf[i % RING_FIFO];
            const int pos = f + RING_FIFO;
            Wf[i % RING_FIFO] = *reinterpret_cast<const bf16x8*>(smem + (pos < 16 ? r.rd_cur + pos * 1024 : r.rd_next + (pos - 16) * 1024));
            a1[b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wa, X3[0][kk], a1[b], 0, 0, 0);
            if (s3 < 2) a1[b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wa, X3[1][kk], a1[b], 0, 0, 0);
            if (s3 < 1) a1[b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wa, X3[2][kk], a1[b], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
          }
        }
      }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
    for (int b = 0; b < 16; ++b) acc[b][0] = a1[b];
  } else {
    Ring r;
    r.src = reinterpret_cast<const char*>(stream) + wid * (RingCfg<8>::PPW * 1024);
    r.lane16 = (unsigned)lane * 16u;
    r.lds_dst = (unsigned)(size_t)(__attribute__((address_space(3))) char*)smem + (unsigned)wid * (RingCfg<8>::PPW * 1024u);
    r.n_groups = STREAM_GROUPS;
    r.issue_grp = 0;
    r.issue_slot = 0;
    r.rd_base = (unsigned)lane * 16u;
    r.next_slot = 0;
    r.rd_next = r.rd_base;
    r.rd_cur = r.rd_base;
    __syncthreads();
#pragma unroll
    for (int gq = 0; gq < RingCfg<8>::LEAD; ++gq) ring_issue<8>(r);
    asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" ::"n"(RingCfg<8>::PPW * (RingCfg<8>::LEAD - 1)) : "memory");
    bf16x8 Wf[RING_FIFO];
#pragma unroll
    for (int j = 0; j < RING_FIFO; ++j) Wf[j] = *reinterpret_cast<const bf16x8*>(smem + r.rd_next + j * 1024);
    for (int ps = 0; ps < passes; ++ps)
#pragma unroll 1
      for (int l = 0; l < LAYERS; ++l) {
        gemm_ring16<16, 8, 8>(acc, X, r, Wf, smem);
      }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  }
  float s = 0.0f;
#pragma unroll
  for (int b = 0; b < 16; ++b) s += acc[b][0][0] + acc[b][1][3];
  if (s == 123.456f) out[blockIdx.x * 512 + threadIdx.x] = s;  // keep the accumulators live
}

int main() {
  int cus = 0;
  hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, 0);
  const size_t stream_floats = (size_t)STREAM_GROUPS * 16 * 256;
  std::vector<float> h(stream_floats);
  for (size_t i = 0; i < stream_floats; ++i) h[i] = 0.0f;  // bf16 pairs of 0: finite results; the timing does not depend on values here
  // small non-zero bf16 weights (0x3c00 = 0.0078125): random-ish sign pattern so that the data toggles like the product's
  unsigned* hw = reinterpret_cast<unsigned*>(h.data());
  unsigned seed = 12345u;
  for (size_t i = 0; i < stream_floats; ++i) {
    seed = seed * 1664525u + 1013904223u;
    const unsigned lo = 0x3c00u | ((seed >> 9) & 0x7fu) | ((seed & 1u) << 15), hi = 0x3b80u | ((seed >> 17) & 0x7fu) | (((seed >> 1) & 1u) << 15);
    hw[i] = lo | (hi << 16);
  }
  float *d_stream, *d_out;
  hipMalloc(&d_stream, stream_floats * 4);
  hipMalloc(&d_out, (size_t)cus * 512 * 4);
  hipMemcpy(d_stream, h.data(), stream_floats * 4, hipMemcpyHostToDevice);
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  const int passes = 8;
  double tf[3];
  const char* names[3] = {"A pure MFMA (no ring, no LDS reads)", "B ring GEMM loop alone", "C split-bf16 (x6) ring loop, 16 points per wave"};
  for (int v = 0; v < 3; ++v) {
    auto launch = [&](int ps) {
      if (v == 0) hipLaunchKernelGGL(probe_kernel<0>, dim3(cus), dim3(512), 0, 0, d_stream, d_out, ps);
      else if (v == 1) hipLaunchKernelGGL(probe_kernel<1>, dim3(cus), dim3(512), 0, 0, d_stream, d_out, ps);
      else hipLaunchKernelGGL(probe_kernel<2>, dim3(cus), dim3(512), 0, 0, d_stream, d_out, ps);
    };
    launch(1);
    hipDeviceSynchronize();
    float best = 1e30f;
    for (int rep = 0; rep < 5; ++rep) {
      hipEventRecord(e0);
      launch(passes);
      hipEventRecord(e1);
      hipEventSynchronize(e1);
      float ms;
      hipEventElapsedTime(&ms, e0, e1);
      if (ms < best) best = ms;
    }
    // per wave and layer: 128 fragments x 2 MFMAs x (16 x 16 x 32 x 2) FLOP
    // (C: LAYERS / 3 layers of 128 fragments x 6 MFMAs -- the same MFMA count per pass)
    const double flop = (double)cus * 8 * passes * (v == 2 ? (LAYERS / 3) * 128.0 * 6.0 : LAYERS * 128.0 * 2.0) * 16384.0;
    tf[v] = flop / (best * 1e-3) / 1e12;
    printf("%-40s %8.3f ms  %8.1f TFLOP/s  %.3f of the 2.5 PF dense bf16 peak\n", names[v], best, tf[v], tf[v] / 2500.0);
  }
  printf("ring loop / pure MFMA: %.3f\n", tf[1] / tf[0]);
  printf("product kernel rsn_field_bf16_ring16_kernel (profiles/r03_config4_bf16_summary.md): 1313 TFLOP/s = %.3f of its own GEMM loop's rate here\n", 1313.0 / tf[1]);
  return 0;
}
