// rsn_wgrad_slab.hip -- PROBE, not part of librsn_hip.so (tools/wgrad_slab_report.py builds it into a variant library;
// results: profiles/r03_wgrad_slab.txt).  Weight gradients dW[n][k] += sum_m dY[m][n] * X[m][k], db[n] += sum_m dY[m][n]
// over operands kept in SLAB order: would the training kernels' activation / layer-gradient stores become cheap (1 KiB
// contiguous per wave) at no cost to this kernel?  Answer: the stores would, this kernel would not (-8 %).
//
// Slab order is the order the field kernels hold their data in: a 32-point tile of an [N, F] matrix is stored as
// [it = F/8][lane = m + 32 h][4]: element (point 32 t + m, feature 8 it + 4 h + s) -- the float4 lane (m, h) owns after a
// GEMM (rsn_field.hip).  The training forward / backward write one such float4 per lane and K-iteration: ONE KiB
// contiguous per wave store (row-major rows cost them 32 rows x 32 B per store: round 2 measured the activation and
// layer-gradient stores at 3.4 % and 13 % of those kernels).
//
// For this kernel the reduction index (the point) must sit on the MFMA's K, i.e. the operands are needed TRANSPOSED with
// respect to the slab (lanes = features).  The transposition happens in LDS: a workgroup pulls whole tiles (32 points:
// up to 32 KiB of X and 32 KiB of dY) into a double-buffered LDS image by LDS-DMA -- 1 KiB pieces, one per K-iteration
// row, rows padded to 1040 B -- and every wave reads its fragments from there with conflict-free ds_read_b128 /
// ds_read_b64: lane i takes the 8 features of K-iteration row i of X (columns 8 i .. 8 i + 7) and the two rows 2 i,
// 2 i + 1 of its 64-row output slab, exactly the register layout of rsn_wgrad_kernel (rsn_wgrad.hip), so accumulators
// and flush are the same.  Each operand byte is fetched from memory once per workgroup and never passes through VGPRs
// on its way to LDS; one barrier per tile (256 MFMAs per wave).
//
// MFMA-bound: 2 * N * n_out * k_in FLOP; HBM reads N * (n_out + k_in) * 4 B.
#include <utility>

#include "rsn_mfma.h"

// compile-time loop: f(std::integral_constant<int, 0>) ... f(<N-1>).  The staging registers below must be named by
// constants (an indexable register array lands in scratch memory, even when every index is known after unrolling).
template <int... Is, class F>
__device__ __forceinline__ void ws_static_for_impl(std::integer_sequence<int, Is...>, F&& f) {
  (f(std::integral_constant<int, Is>{}), ...);
}
template <int N, class F>
__device__ __forceinline__ void ws_static_for(F&& f) {
  ws_static_for_impl(std::make_integer_sequence<int, N>{}, f);
}

#define WS_MAX_SEG 8
#define WS_ROWB 1040  // bytes per K-iteration row of a tile in LDS: 1 KiB + 16 (stride 260 dwords: conflict-free column reads)

struct WSlabArgs {
  int n_seg;
  long long tiles_max[WS_MAX_SEG];  // 32-point tiles of segment s (upper bound when n_dev is set)
  const int* n_dev[WS_MAX_SEG];     // optional device-side count; points = min(32 tiles_max, *n_dev * per_count)
  int per_count[WS_MAX_SEG];
  const float* dy[WS_MAX_SEG];      // slabs [tile][nit_dy][64][4]
  const float* x[WS_MAX_SEG];       // slabs [tile][nit_x][64][4]
  int nit_dy, nit_x;
  int n_out, k_in, ld_dw;
  const int* col_map;  // optional: slot k -> destination column (or -1)
  float* dw;           // [n_out, ld_dw], accumulated
  float* db;           // [n_out] or NULL, accumulated
};

// NKB: 32-column blocks per wave = 8 features per lane-row: 8 (k_in <= 256: lane i = K-iteration row i, both halves),
// 4 (k_in <= 128: lane i = row i/2, half i&1), 2 (k_in <= 64: lane i = row i/4, half, feature pair).
// NSUB: waves that share a pair of output row blocks (1: n_out > 128, 2: n_out > 64, 4: else) and split a tile's 16 point pairs.
template <int NKB, int NSUB>
__global__ __launch_bounds__(256) void rsn_wgrad_slab_kernel(const WSlabArgs a) {
  constexpr int XROWS = NKB * 4;                   // K-iteration rows of X held per tile
  constexpr int BUF_BYTES = (XROWS + 32) * WS_ROWB;  // X rows then dY rows
  __shared__ __attribute__((aligned(16))) char tiles[2 * BUF_BYTES];
  __shared__ float tr[4][2][NKB * 32];
  const int lane = threadIdx.x & 63;
  const int wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int i = lane & 31, h = lane >> 5;
  constexpr int P = 4 / NSUB;  // pairs of 32-row output blocks
  constexpr int nsub = NSUB;
  const int nb0 = (wid % P) * 2;  // this wave's two 32-row output blocks (rows 32 nb0 + 2 i + t)
  const int sub = wid / P;        // waves that share a block pair split the tile's 16 point pairs

  // ---- the launch's tile list: segment s owns tiles [tb[s], tb[s+1])
  long long tb[WS_MAX_SEG + 1];
  tb[0] = 0;
#pragma unroll
  for (int s = 0; s < WS_MAX_SEG; ++s) {
    long long nt = 0;
    if (s < a.n_seg) {
      nt = a.tiles_max[s];
      if (a.n_dev[s]) {
        const long long np = (long long)(*a.n_dev[s]) * a.per_count[s];
        const long long t = np > 0 ? (np + 31) / 32 : 0;
        nt = t < nt ? t : nt;
      }
    }
    tb[s + 1] = tb[s] + nt;
  }
  const long long T = tb[WS_MAX_SEG];
  if ((long long)blockIdx.x >= T) return;  // workgroup-uniform

  f32x16 acc[2][NKB];
#pragma unroll
  for (int t = 0; t < 2; ++t)
#pragma unroll
    for (int kb = 0; kb < NKB; ++kb)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[t][kb][r] = 0.0f;
  float bsum[2] = {0.0f, 0.0f};

  // ---- this lane's byte offsets inside a tile image (row / column indices clamped into the slab: a lane without an
  //      output row or input column works on a duplicate and is never flushed)
  unsigned off_a, off_b;
  {
    int it_o = nb0 * 4 + (i >> 2);
    it_o = it_o < a.nit_dy ? it_o : a.nit_dy - 1;
    off_a = (unsigned)(XROWS + it_o) * WS_ROWB + (unsigned)(32 * ((i >> 1) & 1)) * 16u + (unsigned)(i & 1) * 8u;
    int it_c = NKB == 8 ? i : (NKB == 4 ? (i >> 1) : (i >> 2));
    it_c = it_c < a.nit_x ? it_c : a.nit_x - 1;
    const int hc = NKB == 8 ? 0 : (NKB == 4 ? (i & 1) : ((i >> 1) & 1));
    off_b = (unsigned)it_c * WS_ROWB + (unsigned)(32 * hc) * 16u + (NKB == 2 ? (unsigned)(i & 1) * 8u : 0u);
  }

  // pieces of one tile (1 KiB each): rows 0..nit_x-1 of X, then rows 0..nit_dy-1 of dY; wave w moves pieces w, w + 4, ...
  // (<= 16 per wave).  They travel global -> VGPR -> LDS: the loads of the NEXT tile are issued during the first half
  // of this tile's point pairs, the LDS writes (into the other buffer) during the second half -- one or two short
  // instructions beside each pair's 16 MFMAs.  (LDS-DMA, global_load_lds, was measured first: its issue costs the
  // wave 100+ cycles per piece, 10-18 % of a tile with one wave per SIMD and nobody to cover for it.)
  const int n_pieces = a.nit_x + a.nit_dy;
  const int pz_last = n_pieces - 1;
  const float* nx_x = nullptr;   // next tile's slabs
  const float* nx_d = nullptr;
  auto locate_tile = [&](long long gt) {
    int s = 0;
    long long lt = gt;
#pragma unroll
    for (int q = 1; q < WS_MAX_SEG; ++q)
      if (gt >= tb[q]) {  // (static indices only: a dynamically indexed register array would live in scratch)
        s = q;
        lt = gt - tb[q];
      }
    nx_x = a.x[s] + lt * (long long)a.nit_x * 256;
    nx_d = a.dy[s] + lt * (long long)a.nit_dy * 256;
  };
  f32x16 sv0, sv1, sv2, sv3;  // 16 staged float4's: piece k = elements 4 (k % 4) .. of vector k / 4
  auto piece_of = [&](int k) { const int pz = wid + 4 * k; return pz < pz_last ? pz : pz_last; };  // clamped: duplicates are harmless
  auto load_piece = [&](auto K) {
    constexpr int k = decltype(K)::value;
    const int pz = piece_of(k);
    const bool is_x = pz < a.nit_x;
    const int row = is_x ? pz : pz - a.nit_x;
    const float* src = (is_x ? nx_x : nx_d) + (long long)row * 256;
    const float4 v = *reinterpret_cast<const float4*>(src + lane * 4);
    f32x16& sv = k / 4 == 0 ? sv0 : (k / 4 == 1 ? sv1 : (k / 4 == 2 ? sv2 : sv3));
    sv[4 * (k % 4) + 0] = v.x; sv[4 * (k % 4) + 1] = v.y; sv[4 * (k % 4) + 2] = v.z; sv[4 * (k % 4) + 3] = v.w;
  };
  auto store_piece = [&](auto K, char* img) {
    constexpr int k = decltype(K)::value;
    const int pz = piece_of(k);
    const bool is_x = pz < a.nit_x;
    const int row = is_x ? pz : XROWS + (pz - a.nit_x);
    const f32x16& sv = k / 4 == 0 ? sv0 : (k / 4 == 1 ? sv1 : (k / 4 == 2 ? sv2 : sv3));
#ifdef WS_DIAG_NO_LDSWRITE
    asm volatile("" ::"v"(sv[4 * (k % 4) + 0]), "v"(sv[4 * (k % 4) + 1]), "v"(sv[4 * (k % 4) + 2]), "v"(sv[4 * (k % 4) + 3]), "v"(row));
#else
    *reinterpret_cast<float4*>(img + (unsigned)row * WS_ROWB + (unsigned)lane * 16u) =
        make_float4(sv[4 * (k % 4) + 0], sv[4 * (k % 4) + 1], sv[4 * (k % 4) + 2], sv[4 * (k % 4) + 3]);
#endif
  };

  float fa[2][2], fb[2][NKB];
  auto read_pair = [&](int rb, const char* img, int pair) {
    const unsigned pt = (unsigned)(2 * pair + h) * 16u;  // this lane half's point of the pair
    const float2 va = *reinterpret_cast<const float2*>(img + off_a + pt);
    fa[rb][0] = va.x;
    fa[rb][1] = va.y;
    if (NKB == 8) {
      const float4 v0 = *reinterpret_cast<const float4*>(img + off_b + pt);
      const float4 v1 = *reinterpret_cast<const float4*>(img + off_b + pt + 512);
      fb[rb][0] = v0.x; fb[rb][1] = v0.y; fb[rb][2] = v0.z; fb[rb][3] = v0.w;
      fb[rb][4 % NKB] = v1.x; fb[rb][5 % NKB] = v1.y; fb[rb][6 % NKB] = v1.z; fb[rb][7 % NKB] = v1.w;
    } else if (NKB == 4) {
      const float4 v0 = *reinterpret_cast<const float4*>(img + off_b + pt);
      fb[rb][0] = v0.x; fb[rb][1] = v0.y; fb[rb][2 % NKB] = v0.z; fb[rb][3 % NKB] = v0.w;
    } else {
      const float2 v0 = *reinterpret_cast<const float2*>(img + off_b + pt);
      fb[rb][0] = v0.x; fb[rb][1] = v0.y;
    }
  };
  auto mma_pair = [&](int rb) {
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      bsum[t] += fa[rb][t];
#pragma unroll
      for (int kb = 0; kb < NKB; ++kb)
        acc[t][kb] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[rb][t], fb[rb][kb], acc[t][kb], 0, 0, 0);
    }
  };

  // ---- persistent tile loop, LDS image double-buffered: one barrier per tile.  Operand traffic is spread evenly over
  // the tile: beside the 16 MFMAs of pair q the wave writes piece q of the NEXT tile (held in registers since the
  // previous tile) into the idle LDS buffer and requests piece q of the tile after next from memory -- every request
  // has a whole tile (~7 us) to land, and the chip sees a steady stream (requesting a tile's 64 KiB within half a
  // tile, every workgroup in the same phase, cost 12 %: profiles/r03_wgrad_slab.txt).
  constexpr int MP = 16 / NSUB;   // point pairs of this wave per tile
  constexpr int PPP = 16 / MP;    // pieces moved per pair
  const long long G = gridDim.x;
  locate_tile(blockIdx.x);
  ws_static_for<16>([&](auto K) { load_piece(K); });
  ws_static_for<16>([&](auto K) { store_piece(K, tiles); });
  locate_tile(blockIdx.x + G < T ? blockIdx.x + G : blockIdx.x);
  ws_static_for<16>([&](auto K) { load_piece(K); });
  int buf = 0;
  for (long long gt = blockIdx.x; gt < T; gt += G) {
#ifndef WS_DIAG_NO_BARRIER
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");  // every wave's rows of this tile are in LDS; the other buffer is free
#endif
#ifdef WS_DIAG_SAME_TILE
    locate_tile(blockIdx.x);
#else
    locate_tile(gt + 2 * G < T ? gt + 2 * G : gt);  // the tile after next (behind the end: any valid tile, never read)
#endif
    const char* img = tiles + buf * BUF_BYTES;
    char* nxt = tiles + (buf ^ 1) * BUF_BYTES;
    read_pair(0, img, sub);
    ws_static_for<MP>([&](auto Q) {
      constexpr int q = decltype(Q)::value;
      constexpr int rb = q & 1;
#ifndef WS_DIAG_NO_READS
      if constexpr (q + 1 < MP) read_pair(rb ^ 1, img, sub + (q + 1) * nsub);
#endif
#ifndef WS_DIAG_NO_FILL
      ws_static_for<PPP>([&](auto Jc) {
        constexpr int k = q * PPP + decltype(Jc)::value;
        store_piece(std::integral_constant<int, k>{}, nxt);   // next tile: registers -> LDS
        load_piece(std::integral_constant<int, k>{});         // tile after next: memory -> registers
      });
#endif
      __builtin_amdgcn_sched_barrier(0);  // the next pair's reads and this pair's piece traffic are issued ahead of its MFMAs
      mma_pair(rb);
      __builtin_amdgcn_sched_barrier(0);
    });
    buf ^= 1;
  }

  // ---- flush (as rsn_wgrad_kernel): C/D layout col = lane&31 (column slot), row = (r&3) + 8*(r>>2) + 4*h (row slot).
  // The wave-private LDS tile turns "lane i holds columns i*NKB+kb" into "lane i holds column kb*32+i" so that one
  // atomic wave-instruction covers two contiguous 128-B row segments.
  float* trw = &tr[wid][h][0];
  int cdst[NKB];
#pragma unroll
  for (int kb = 0; kb < NKB; ++kb) {
    const int k = kb * 32 + i;
    cdst[kb] = -1;
    if (k < a.k_in) cdst[kb] = a.col_map ? a.col_map[k] : k;
  }
#pragma unroll
  for (int t = 0; t < 2; ++t) {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int slot = (r & 3) + 8 * (r >> 2) + 4 * h;
      const int n = nb0 * 32 + 2 * slot + t;
#pragma unroll
      for (int kb = 0; kb < NKB; ++kb) trw[i * NKB + kb] = acc[t][kb][r];
#pragma unroll
      for (int kb = 0; kb < NKB; ++kb) {
        const float v = trw[kb * 32 + i];
        if (cdst[kb] >= 0 && n < a.n_out) atomicAdd(&a.dw[(long long)n * a.ld_dw + cdst[kb]], v);
      }
    }
    if (a.db) {
      const float v = bsum[t] + __shfl_xor(bsum[t], 32, 64);
      const int n = nb0 * 32 + 2 * i + t;
      if (h == 0 && n < a.n_out) atomicAdd(&a.db[n], v);
    }
  }
}

// rsn_weight_grad_slab: slabs [tile = N/32][it = F/8][lane = m + 32 h][4] floats; segment lengths / outputs as rsn_weight_grad_multi_dev
extern "C" int rsn_weight_grad_slab(int32_t n_segments, const int64_t* n_points_max, const int32_t* const* n_dev,
                                    const int32_t* per_count, const float* const* dy, int32_t n_out,
                                    const float* const* x, int32_t k_in, const int32_t* col_map, float* dw, int32_t ld_dw,
                                    float* db, void* stream) {
  RSN_REQUIRE(n_segments >= 0 && n_segments <= WS_MAX_SEG, RSN_ERR_INVALID_ARGUMENT, "n_segments=%d (at most %d)",
              n_segments, WS_MAX_SEG);
  RSN_REQUIRE(n_out >= 1 && n_out <= 256 && k_in >= 1 && k_in <= 256, RSN_ERR_INVALID_ARGUMENT,
              "n_out=%d k_in=%d (outputs up to 256 x 256)", n_out, k_in);
  RSN_REQUIRE(ld_dw >= 1, RSN_ERR_INVALID_ARGUMENT, "ld_dw=%d", ld_dw);
  if (n_segments == 0) return RSN_OK;
  RSN_REQUIRE(n_points_max && dy && x && dw, RSN_ERR_INVALID_ARGUMENT, "a pointer is NULL");
  WSlabArgs a = {};
  long long tiles = 0;
  for (int s = 0; s < n_segments; ++s) {
    RSN_REQUIRE(n_points_max[s] >= 0, RSN_ERR_INVALID_ARGUMENT, "n_points_max[%d]=%lld", s, (long long)n_points_max[s]);
    if (n_points_max[s] == 0) continue;
    RSN_REQUIRE(dy[s] && x[s], RSN_ERR_INVALID_ARGUMENT, "segment %d: a pointer is NULL", s);
    RSN_REQUIRE(((uintptr_t)dy[s] % 16 == 0) && ((uintptr_t)x[s] % 16 == 0), RSN_ERR_INVALID_ARGUMENT,
                "segment %d: slabs must be 16-byte aligned", s);
    const int q = a.n_seg++;
    a.dy[q] = dy[s];
    a.x[q] = x[s];
    a.tiles_max[q] = (n_points_max[s] + 31) / 32;
    a.n_dev[q] = n_dev ? n_dev[s] : nullptr;
    a.per_count[q] = (n_dev && n_dev[s] && per_count) ? per_count[s] : 1;
    RSN_REQUIRE(a.per_count[q] >= 1, RSN_ERR_INVALID_ARGUMENT, "segment %d: per_count=%d", s, a.per_count[q]);
    tiles += a.tiles_max[q];
  }
  if (tiles == 0) return RSN_OK;
  a.nit_dy = (n_out + 7) / 8;
  a.nit_x = (k_in + 7) / 8;
  a.n_out = n_out; a.k_in = k_in; a.ld_dw = ld_dw;
  a.col_map = col_map; a.dw = dw; a.db = db;
  const int cus = rsn_device_cus();
  // every workgroup pays one atomic flush of the output tile (chip-wide ~1.3 TB/s of added bytes): as in rsn_wgrad.hip,
  // T(G) = tiles / G * t_tile + G * t_flush is smallest at G = sqrt(tiles t_tile / t_flush)
  const int nkb = k_in > 128 ? 8 : (k_in > 64 ? 4 : 2);
  const int P = n_out <= 64 ? 1 : (n_out <= 128 ? 2 : 4);
  const double t_tile = 16.0 * 2 * nkb * 64 / 2.1e9 * P / 4.0;
  const double t_flush = (double)n_out * k_in * 4.0 / 1.3e12 + 2e-8;
  long long grid = (long long)(sqrt((double)tiles * t_tile / t_flush) + 0.5);
  if (grid > cus) grid = cus;
  if (grid > tiles) grid = tiles;
  if (grid < 1) grid = 1;
  hipStream_t st = (hipStream_t)stream;
#define WS_LAUNCH(NKBV)                                                                                        \
  do {                                                                                                         \
    if (P == 4) hipLaunchKernelGGL((rsn_wgrad_slab_kernel<NKBV, 1>), dim3((unsigned)grid), dim3(256), 0, st, a); \
    else if (P == 2) hipLaunchKernelGGL((rsn_wgrad_slab_kernel<NKBV, 2>), dim3((unsigned)grid), dim3(256), 0, st, a); \
    else hipLaunchKernelGGL((rsn_wgrad_slab_kernel<NKBV, 4>), dim3((unsigned)grid), dim3(256), 0, st, a);       \
  } while (0)
  if (nkb == 8) WS_LAUNCH(8);
  else if (nkb == 4) WS_LAUNCH(4);
  else WS_LAUNCH(2);
#undef WS_LAUNCH
  RSN_HIP(hipGetLastError());
  return RSN_OK;
}
