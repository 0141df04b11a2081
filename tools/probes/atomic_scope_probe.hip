// atomic_scope_probe.hip -- what the weight-gradient flush costs, and whether XCD-local atomics are cheaper.
// The flush: every workgroup (one per CU, 256) adds its 256 x 256 fp32 partial tile into ONE tile with device-scope
// atomics (16.7 M atomic adds, 67 MB; 0.76 ms per training step over 14 launches by ablation).  MI355X has 8 XCDs with
// private L2s, so a device-scope atomic cannot resolve in an L2.  Variant: each XCD accumulates into its OWN copy of the
// tile with workgroup-scope atomics (no other XCD touches that copy, so the XCD's L2 is the point of coherence), and a
// small second kernel sums the 8 copies.
//   hipcc -O3 --offload-arch=gfx950 -std=c++17 tools/probes/atomic_scope_probe.hip -o build/atomic_scope_probe
#include <hip/hip_runtime.h>

#include <cstdio>
#include <vector>

#define TILE (256 * 256)

__device__ __forceinline__ int xcc_id() { return (int)(__builtin_amdgcn_s_getreg(20 | (0 << 6) | (3 << 11)) & 0xf); }

// MODE 0: device scope, one tile.  1: workgroup scope, tile of this workgroup's XCD.  2: device scope, per-XCD tiles.
template <int MODE>
__global__ __launch_bounds__(256) void flush(float* dst, int* xcc_seen) {
  const int x = xcc_id();
  if (threadIdx.x == 0) xcc_seen[blockIdx.x] = x;
  float* t = dst + (MODE == 0 ? 0 : (size_t)x * TILE);
  const float v = 1.0f;
  // 256 threads x 256 atomics: thread i takes column i of every row (one wave instruction = 256 contiguous bytes)
  for (int r = 0; r < 256; ++r) {
    if (MODE == 1)
      __hip_atomic_fetch_add(t + r * 256 + threadIdx.x, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    else
      __hip_atomic_fetch_add(t + r * 256 + threadIdx.x, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
}

__global__ void reduce8(const float* part, float* out) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  float s = 0.0f;
  for (int x = 0; x < 8; ++x) s += part[(size_t)x * TILE + i];
  out[i] += s;
}

template <int MODE>
static void run(const char* name, float* dst, float* out, int* seen) {
  hipMemset(dst, 0, 8 * TILE * 4);
  hipMemset(out, 0, TILE * 4);
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  hipLaunchKernelGGL(flush<MODE>, dim3(256), dim3(256), 0, 0, dst, seen);
  hipDeviceSynchronize();
  hipEventRecord(e0, 0);
  const int reps = 20;
  for (int r = 0; r < reps; ++r) {
    hipLaunchKernelGGL(flush<MODE>, dim3(256), dim3(256), 0, 0, dst, seen);
    if (MODE != 0) hipLaunchKernelGGL(reduce8, dim3(256), dim3(256), 0, 0, dst, out);
  }
  hipEventRecord(e1, 0);
  hipEventSynchronize(e1);
  float ms = 0.0f;
  hipEventElapsedTime(&ms, e0, e1);
  std::vector<float> h(8 * TILE);
  hipMemcpy(h.data(), dst, 8 * TILE * 4, hipMemcpyDeviceToHost);
  double total = 0.0;
  for (size_t i = 0; i < h.size(); ++i) total += h[i];
  // every launch adds 256 workgroups x TILE ones
  printf("%-62s %8.1f us per flush%s   sum check %.0f (expect %.0f)\n", name, ms / reps * 1e3, MODE ? " + reduce" : "",
         total, (double)(reps + 1) * 256.0 * TILE);
}

int main() {
  float *dst, *out;
  int* seen;
  hipMalloc(&dst, 8 * TILE * 4);
  hipMalloc(&out, TILE * 4);
  hipMalloc(&seen, 256 * 4);
  run<0>("device-scope atomics, one 256 x 256 tile (the product's flush)", dst, out, seen);
  run<2>("device-scope atomics, one tile per XCD", dst, out, seen);
  run<1>("workgroup-scope atomics, one tile per XCD", dst, out, seen);
  std::vector<int> hs(256);
  hipMemcpy(hs.data(), seen, 256 * 4, hipMemcpyDeviceToHost);
  int cnt[16] = {0};
  for (int i = 0; i < 256; ++i) cnt[hs[i] & 15]++;
  printf("workgroups per XCC_ID:");
  for (int x = 0; x < 16; ++x) if (cnt[x]) printf(" %d:%d", x, cnt[x]);
  printf("   (workgroup 0..7 on XCC %d %d %d %d %d %d %d %d)\n", hs[0], hs[1], hs[2], hs[3], hs[4], hs[5], hs[6], hs[7]);
  return 0;
}
