// rsn_ringt.h -- what the TRAINING kernels on the LDS weight ring share (rsn_field_bf16_train.hip: plain bf16;
// rsn_field_x6_train.hip: split-bf16, fp32-equivalent): the ring with a walk program and a COUNTED consumer wait, saved-row
// descriptors and counted stores, asynchronous loads behind the ring's own wait arithmetic, the tile space of a multi-job launch.
// Design notes: rsn_field_bf16_train.hip.
#pragma once
#include "rsn_field_bwd_common.h"
#include "rsn_ring16.h"

// ring groups in flight ahead of the one being consumed.  The consumer's counted wait lets the row stores of the last LEAD - 1
// group intervals stay in flight (vmcnt retires in order behind the weight group's LDS-DMA), so LEAD bounds the bytes each wave
// keeps on their way to HBM: 2 KiB per interval.  Measured (profiles/r04_bf16_train_ab.txt): depth is NOT what limits the store
// stream -- forward 3 / 4 groups ahead 1.22 / 1.28 ms, backward 3 / 5 / 7 groups 0.80 / 0.78 / 0.78 ms per primary level.
#ifndef RT_LEAD_FWD
#define RT_LEAD_FWD 3
#endif
#ifndef RT_LEAD_BWD
#define RT_LEAD_BWD 5
#endif
// Cache policy of the saved-row stores (buffer-instruction aux bits: 1 sc0, 2 nt, 16 sc1): NON-TEMPORAL.  The rows (3 GB per
// primary-level launch) are not read again by the kernel that writes them; with the default policy they push the 2.4 MB
// weight stream out of the XCD's 4 MiB L2 and every workgroup's LDS-DMA then comes from beyond it: forward with normals
// 1.83 -> 1.22 ms per launch, backward 1.05 -> 0.78 ms, the step 11.0 -> 8.7 ms (sc1: no change; nt + sc1 as nt).  (The
// exact-fp32 kernels measured the opposite in round 3 -- there the stores share the vector-memory path with a per-wave
// weight stream that is L2-bound either way.)
#ifndef RT_STORE_AUX
#define RT_STORE_AUX 2
#endif
// Stagger: waves 4..7 (the second wave of every SIMD) run STAG ring groups BEHIND waves 0..3 in the same instruction stream, so
// that the two waves of a SIMD are never in a layer epilogue (VALU only: pack, ReLU bits / masks, row addressing) at the same
// time -- one wave's epilogue runs under the other's MFMAs.  The ring holds the groups in between: LEAD + STAG + 1 slots.
#ifndef RT_STAG_FWD
#define RT_STAG_FWD 0
#endif
#ifndef RT_STAG_BWD
#define RT_STAG_BWD 0
#endif
// which waves run behind (the partner of a wave on its SIMD must be in the other set)
#ifndef RT_LATE_MODE
#define RT_LATE_MODE 0
#endif
#if RT_LATE_MODE == 0
#define RT_LATE_WAVE(wid) ((wid) >= 4)
#elif RT_LATE_MODE == 1
#define RT_LATE_WAVE(wid) (((wid) & 1) != 0)
#else
#define RT_LATE_WAVE(wid) ((((wid) >> 1) & 1) != 0)
#endif
#define RT_PPW (RSN_RING_GROUP_FRAGS / 8)
#define RT_TABLE_FLOATS (RING_BIAS_FLOATS + 256)   // biases (packed row order) + the density-head row (normal-sweep seed)
#define RT_RING_BYTES(LEAD, STAG) (((LEAD) + (STAG) + 1) * RING_GROUP_BYTES)

typedef unsigned u32x2t __attribute__((ext_vector_type(2)));
typedef unsigned u32x4t __attribute__((ext_vector_type(4)));

// ------------------------------------------------------------------------------------------------ the ring, with a program
template <int LEAD, int STAG = 0>
struct RingT {
  static constexpr int SLOTS = LEAD + STAG + 1;
  static constexpr int kLead = LEAD, kStag = STAG;
  const char* src;     // q_stream base + wave * PPW KiB
  unsigned lane16, lds_dst;
  int issue_grp, issue_slot;
  int e0, j0, e1, j1;  // the walk: group e_i - 1 is followed by group j_i (two jumps describe every program below)
  unsigned rd_base, rd_cur, rd_next;
  int next_slot;
  int since;           // vector-memory operations issued since the last batch of asynchronous loads (ald8 / ald16; wait_loads)
  int c0;              // counted vector-memory operations (row stores) issued since the last group boundary ...
  int cp[LEAD - 2];    // ... and in the LEAD - 2 intervals before it (cp[0] the newest)
};

template <class RING>
__device__ __forceinline__ void ringt_issue(RING& r) {
  const char* g = r.src + (size_t)r.issue_grp * RING_GROUP_BYTES;
  const unsigned d = __builtin_amdgcn_readfirstlane(r.lds_dst + (unsigned)r.issue_slot * RING_GROUP_BYTES);
#pragma unroll
  for (int i = 0; i < RT_PPW; ++i) glds16(g + i * 1024, r.lane16, d + i * 1024);
  r.since += RT_PPW;
  int n = r.issue_grp + 1;
  n = (n == r.e0) ? r.j0 : ((n == r.e1) ? r.j1 : n);
  r.issue_grp = n;
  r.issue_slot = (r.issue_slot + 1 == RING::SLOTS) ? 0 : r.issue_slot + 1;
}

// s_waitcnt vmcnt(n) with a wave-uniform n that the fully unrolled GEMMs fold to a constant almost everywhere (the field is an
// immediate).  Rounded DOWN to the next available step: a smaller count only waits for more.  An if-chain, not a switch: a
// jump table inside the GEMM loop keeps hipcc from unrolling it (and the accumulators then live in scratch).
#define RT_WAIT_STEP(k) if (n >= k) { asm volatile("s_waitcnt vmcnt(" #k ")" ::: "memory"); return; }
__device__ __forceinline__ void wait_vm(int n) {
  RT_WAIT_STEP(40) RT_WAIT_STEP(36) RT_WAIT_STEP(32) RT_WAIT_STEP(28) RT_WAIT_STEP(26) RT_WAIT_STEP(24) RT_WAIT_STEP(22)
  RT_WAIT_STEP(20) RT_WAIT_STEP(18) RT_WAIT_STEP(16) RT_WAIT_STEP(14) RT_WAIT_STEP(12) RT_WAIT_STEP(10) RT_WAIT_STEP(9)
  RT_WAIT_STEP(8) RT_WAIT_STEP(7) RT_WAIT_STEP(6) RT_WAIT_STEP(5) RT_WAIT_STEP(4) RT_WAIT_STEP(3) RT_WAIT_STEP(2) RT_WAIT_STEP(1)
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}
#undef RT_WAIT_STEP

// Group boundary.  The LDS-DMA of the group after the one about to be consumed was issued LEAD - 1 boundaries ago; behind it
// in the in-order vmcnt queue sit the counted stores of the last LEAD - 1 intervals and the DMA of LEAD - 2 boundaries:
// "at most that many outstanding" = that DMA (and everything older) has landed.  Uncounted operations (per-sample loads /
// stores the compiler issues on its own) only make the wait stricter.
template <class RING>
__device__ __forceinline__ void ringt_sync(RING& r) {
  constexpr int LEAD = RING::kLead;
  int n = r.c0 + RT_PPW * (LEAD - 2);
#pragma unroll
  for (int i = 0; i < LEAD - 2; ++i) n += r.cp[i];
#ifndef RSN_RT_NO_WAIT      // (RSN_RT_NO_WAIT / _NO_BARRIER: timing ablations, diagnostic builds only; wrong results by construction)
  wait_vm(n);
#endif
#ifndef RSN_RT_NO_BARRIER
  asm volatile("s_barrier" ::: "memory");
#endif
  ringt_issue(r);
#pragma unroll
  for (int i = LEAD - 3; i > 0; --i) r.cp[i] = r.cp[i - 1];
  r.cp[0] = r.c0;
  r.c0 = 0;
  r.rd_cur = r.rd_next;
  r.next_slot = (r.next_slot + 1 == RING::SLOTS) ? 0 : r.next_slot + 1;
  r.rd_next = r.rd_base + (unsigned)r.next_slot * RING_GROUP_BYTES;
}
// how a GEMM's accumulators start (gemm_t / gemm_x6): live | the constant 0 | the bias row, as the first MFMA's C operand
enum { GI_ACC = 0, GI_ZERO = 1, GI_BIAS = 2 };
struct NoHook {
  __device__ __forceinline__ void operator()(int) const {}
};


// ------------------------------------------------------------------------------------------------ saved rows
// A row-major buffer [N, row_bytes]: descriptor over the VALID rows of this wave's 32-point tile (rows past the end fall outside
// the range and the hardware drops their stores / returns 0 for their loads); the lane addresses (row 16 p + m, 16 g bytes in).
struct RowD {
  __amdgpu_buffer_rsrc_t r;
};
__device__ __forceinline__ RowD rowd(const void* base, long long byte_off, int rows, int row_bytes) {
  RowD d;
  d.r = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(reinterpret_cast<const char*>(base)) + (base ? byte_off : 0), 0,
                                          base != nullptr ? rows * row_bytes : 0, 0x00020000);
  return d;
}
// (RSN_RT_*: timing ablations of tools/bf16_train_ab.sh -- wrong results by construction; they compile only under -DRSN_DIAG_BUILD)
template <class RING>
__device__ __forceinline__ void st16(const RowD& d, unsigned voff, unsigned soff, const bf16x8 v, RING& r) {
#ifndef RSN_RT_NO_STORES
  // (offset through the VECTOR offset / the instruction's immediate, scalar offset 0.  With a register soffset hipcc's hazard
  // recogniser lets a VALU overwrite the data registers of a > 8-byte buffer store in the very next instruction -- LLVM's
  // createsVALUHazard: "only if not using a register in the soffset field" -- and on gfx950 that store then carries the NEW value
  // in the last quarter of every 16 lanes: found in rsn_field_x6_train.hip, round 4, where a mask temporary followed the store)
  // (opaque: otherwise hipcc hoists `voff + constant` out of the layer loops, where it can no longer become the immediate offset)
  asm volatile("" : "+v"(voff));
#ifdef RSN_RT_SOFFSET_STORES  // (A/B of the round-4 form, diagnostic builds only: the offset in a scalar register)
  __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4t, v), d.r, voff, soff, RT_STORE_AUX);
#else
  __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4t, v), d.r, voff + soff, 0, RT_STORE_AUX);
#endif
#ifndef RSN_RT_UNCOUNTED
  r.c0 += 1;
  r.since += 1;
#endif
#endif
}
template <class RING>
__device__ __forceinline__ void st8(const RowD& d, unsigned voff, unsigned soff, unsigned w0, unsigned w1, RING& r) {
#ifndef RSN_RT_NO_STORES
  const u32x2t v = {w0, w1};
  __builtin_amdgcn_raw_buffer_store_b64(v, d.r, voff + soff, 0, 0);
#ifndef RSN_RT_UNCOUNTED
  r.c0 += 1;
  r.since += 1;
#endif
#endif
}
// Loads of what this kernel (or the forward before it) saved -- ReLU bits, encoded features -- issued a whole GEMM ahead of their use.
// Rounds 4a: ASYNCHRONOUS inline-asm loads with a counted wait of the ring's own arithmetic (`since`), because hipcc guards an
// ordinary load's first use with a wait that knows nothing of the LDS-DMA in flight.  Round 4b: ORDINARY loads after all.  The
// destination of an inline-asm load is an ordinary register to the compiler -- it may spill or copy it BEFORE the data has arrived
// (it did, in a forward experiment: stale rows, caught by the row-level test) -- and the compiler's own wait turned out cheap: it
// counts only the operations it knows (the hooks' buffer stores, <= 16-32 per GEMM), so `vmcnt(k)` at the first use leaves the ring's
// DMA lead (6-10 operations) and the newest stores in flight; measured: no difference (profiles/r04_x6_ab.txt).
// -DRSN_RT_ASM_LOADS (diagnostic builds) keeps the asm form for A/B.
#ifdef RSN_RT_ASM_LOADS
struct AsyncD {
  u32x4t rs;  // buffer descriptor (V#), built by hand so that it can be an inline-asm operand
};
__device__ __forceinline__ AsyncD asyncd(const void* base, long long byte_off, int rows, int row_bytes) {
  const unsigned long long a = (unsigned long long)(size_t)base + (base ? (unsigned long long)byte_off : 0ull);
  AsyncD d;
  d.rs = u32x4t{(unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)a),
                (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)((a >> 32) & 0xffffull)),
                (unsigned)__builtin_amdgcn_readfirstlane(base != nullptr ? rows * row_bytes : 0), 0x00020000u};
  return d;
}
__device__ __forceinline__ u32x2t ald8(const AsyncD& d, unsigned voff) {
  u32x2t v;
  asm volatile("buffer_load_dwordx2 %0, %1, %2, 0 offen sc0" : "=v"(v) : "v"(voff), "s"(d.rs) : "memory");
  return v;
}
template <int OFF>
__device__ __forceinline__ bf16x8 ald16(const AsyncD& d, unsigned voff) {
  u32x4t v;
  asm volatile("buffer_load_dwordx4 %0, %1, %2, 0 offen offset:%3 sc0" : "=v"(v) : "v"(voff), "s"(d.rs), "n"(OFF) : "memory");
  return __builtin_bit_cast(bf16x8, v);
}
#else
struct AsyncD {
  __amdgpu_buffer_rsrc_t r;
};
__device__ __forceinline__ AsyncD asyncd(const void* base, long long byte_off, int rows, int row_bytes) {
  AsyncD d;
  d.r = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(reinterpret_cast<const char*>(base)) + (base ? byte_off : 0), 0,
                                          base != nullptr ? rows * row_bytes : 0, 0x00020000);
  return d;
}
__device__ __forceinline__ u32x2t ald8(const AsyncD& d, unsigned voff) {
#ifdef RSN_RT_NO_LOADS
  return u32x2t{0xffffffffu, 0xffffffffu};
#elif defined(RSN_RT_ASM_LOADS8)
  u32x2t v;
  asm volatile("buffer_load_dwordx2 %0, %1, %2, 0 offen sc0" : "=v"(v) : "v"(voff), "s"(d.r) : "memory");
  return v;
#else
  return __builtin_amdgcn_raw_buffer_load_b64(d.r, voff, 0, 0);
#endif
}
template <int OFF>
__device__ __forceinline__ bf16x8 ald16(const AsyncD& d, unsigned voff) {
#ifdef RSN_RT_NO_LOADS
  return bf16x8{};
#else
  return __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(d.r, voff + OFF, 0, 0));
#endif
}
#endif
// the batch issued before `r.since` was reset has landed behind this
template <class RING>
__device__ __forceinline__ void wait_loads(RING& r) {
#if defined(RSN_RT_ASM_LOADS) || defined(RSN_RT_ASM_LOADS8)
  wait_vm(r.since);
#else
  (void)r;  // ordinary loads: the compiler waits at their first use
#endif
}
__device__ __forceinline__ void tie(u32x2t& a, u32x2t& b) { asm volatile("" : "+v"(a), "+v"(b)::"memory"); }
__device__ __forceinline__ void tie(bf16x8 (&ft)[4][2]) {
#pragma unroll
  for (int kk = 0; kk < 4; ++kk) asm volatile("" : "+v"(ft[kk][0]), "+v"(ft[kk][1])::"memory");
}

// ------------------------------------------------------------------------------------------------ ReLU bits on packed bf16
__device__ __forceinline__ unsigned pk_min_u16(unsigned a, unsigned b) {
  unsigned o;
  asm("v_pk_min_u16 %0, %1, %2" : "=v"(o) : "v"(a), "v"(b));
  return o;
}
__device__ __forceinline__ unsigned pk_mul_lo_u16(unsigned a, unsigned b) {
  unsigned o;
  asm("v_pk_mul_lo_u16 %0, %1, %2" : "=v"(o) : "v"(a), "v"(b));
  return o;
}

// ------------------------------------------------------------------------------------------------ shared tile prologue
struct TileJobs {
  long long np0, np1, np2, tb1, tb2, n_tiles;
};
template <int TILE = 256, class JOBS>
__device__ __forceinline__ TileJobs tile_space(const JOBS& J) {
  TileJobs t = {0, 0, 0, 0, 0, 0};
#pragma unroll
  for (int k = 0; k < RSN_MAX_JOBS; ++k) {
    if (k < J.n_jobs) {
      int nr = J.j[k].n_rays;
      if (J.j[k].n_dev) {
        const int nd = *J.j[k].n_dev;
        nr = nd < nr ? nd : nr;
      }
      const long long np = (long long)nr * J.j[k].S;
      if (k == 0) t.np0 = np; else if (k == 1) t.np1 = np; else t.np2 = np;
      t.n_tiles += (np + TILE - 1) / TILE;
    }
    if (k == 0) t.tb1 = t.n_tiles; else if (k == 1) t.tb2 = t.n_tiles;
  }
  return t;
}

template <class RING>
__device__ __forceinline__ void ring_start(RING& r, const float* pk, const RsnPackedLayout& L, const char* smem, int wid, int lane,
                                           int first, int e0, int j0, int e1, int j1, bf16x8 (&Wf)[RING_FIFO]) {
  constexpr int LEAD = RING::kLead, STAG = RING::kStag;
  r.src = reinterpret_cast<const char*>(pk + L.q_stream) + wid * (RT_PPW * 1024);
  r.lane16 = (unsigned)lane * 16u;
  r.lds_dst = (unsigned)(size_t)(__attribute__((address_space(3))) char*)smem + (unsigned)wid * (RT_PPW * 1024u);
  r.issue_grp = first;
  r.issue_slot = 0;
  r.e0 = e0; r.j0 = j0; r.e1 = e1; r.j1 = j1;
  r.rd_base = (unsigned)lane * 16u;
  // group g lives in slot g mod SLOTS.  A late wave (wid >= 4) makes STAG group boundaries without consuming anything first: its
  // read pointers start STAG slots "before" slot 0, so that its first real boundary finds them where an early wave's start
  const bool late = STAG > 0 && RT_LATE_WAVE(wid);
  r.next_slot = late ? RING::SLOTS - STAG : 0;
  r.rd_next = r.rd_base + (unsigned)r.next_slot * RING_GROUP_BYTES;
  r.rd_cur = r.rd_next;
  r.c0 = 0;
  r.since = 0;
#pragma unroll
  for (int i = 0; i < LEAD - 2; ++i) r.cp[i] = 0;
  __syncthreads();  // nothing in flight yet (also publishes the LDS tables)
#pragma unroll
  for (int gq = 0; gq < LEAD; ++gq) ringt_issue(r);
  asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" ::"n"(RT_PPW * (LEAD - 1)) : "memory");
  if (late) {
#pragma unroll
    for (int i = 0; i < STAG; ++i) ringt_sync(r);
  }
#pragma unroll
  for (int j = 0; j < RING_FIFO; ++j) Wf[j] = *reinterpret_cast<const bf16x8*>(smem + r.rd_next + j * 1024);
}
// the early waves' matching boundaries at the end of the kernel (every wave passes the same number of barriers)
template <class RING>
__device__ __forceinline__ void ring_finish(RING& r, int wid) {
  if (RING::kStag > 0 && !RT_LATE_WAVE(wid)) {
#pragma unroll
    for (int i = 0; i < RING::kStag; ++i) ringt_sync(r);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // no LDS-DMA may outlive the workgroup's LDS allocation
}
