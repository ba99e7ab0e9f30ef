// Tile helpers shared by the fp16-operand contrastive kernels (pixcon_loss_f16.hip: fixed-split sweeps;
// pixcon_loss_f16p.hip: planned, software-pipelined sweeps): MFMA fragment loads from the padded LDS tile, the
// global -> register -> LDS staging of a 32 x 256 contrast tile, accumulator stores.
#pragma once
#include <type_traits>

#include "common.h"
#include "pixcon.h"

namespace ucd {
namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef __fp16 h4 __attribute__((ext_vector_type(4)));

constexpr int kThreads = 256;
constexpr int kWaves = 4;
constexpr int kTI = 32;
constexpr int kBI = kWaves * kTI;
constexpr int kTJ = 32;
constexpr int kN = 256;
constexpr int kPitchH = kN + 24;   // halfs; 560 B = 140 dwords = 12 (mod 64): ds_read_b128 rows conflict-free
constexpr int kMaxSplit = 16;
constexpr float kRescaleTh = 8.f;  // log2 units
constexpr float kFixedShiftMaxK2 = 24.f;   // log2(e)/T <= 24  <=>  T >= 0.0601: the constant-shift form of sweep 1
constexpr float kLog2e = 1.4426950408889634f;
constexpr float kLn2 = 0.6931471805599453f;

__device__ __forceinline__ int tile_row(int reg, int half) { return (reg & 3) + 8 * (reg >> 2) + 4 * half; }

struct TileList {
  int t1a, n1, t2a, n2;
  __device__ __forceinline__ int count() const { return n1 + n2; }
  __device__ __forceinline__ int at(int v) const { return v < n1 ? t1a + v : t2a + (v - n1); }
};

__device__ __forceinline__ void load_anchor_frags(f16x8 (&a16)[16], const _Float16* __restrict__ ch16, int row, bool ok, int half) {
  if (ok) {
    const f16x8* src = reinterpret_cast<const f16x8*>(ch16 + (size_t)row * kN + 8 * half);
#pragma unroll
    for (int ks = 0; ks < 16; ++ks) a16[ks] = src[2 * ks];   // halfs 16 ks + 8 half .. +7
  } else {
#pragma unroll
    for (int ks = 0; ks < 16; ++ks)
#pragma unroll
      for (int j = 0; j < 8; ++j) a16[ks][j] = (_Float16)0.f;
  }
}

// 32 x 256 fp16 contrast tile = 1024 x 16 B: four 16-byte pieces per thread, kept in four named registers
// (an indexed array ends up in scratch memory)
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
struct Stage {
  u32x4 a, b, c, d;
  u32x4 side;          // sweep 2: this thread's 16-byte chunk of the tile's teacher-probability rows
  unsigned lab;        // label byte of contrast row (threadIdx.x & 31) of the tile
};
// The values fetched here are not needed before the commit at the end of the iteration, and the compiler sinks loads
// down to their first use (load ; wait ; write - the whole HBM/L2 round trip exposed once per tile).  A memory-clobbering
// asm statement right after them pins them here, PROVIDED the pointers are not noalias: the kernels therefore take
// ch16 / row_label / p16 without __restrict__.  (Inline-asm loads would hide the pending register writes from the
// compiler - any copy the register allocator inserts before a hand-written s_waitcnt reads stale data; volatile
// loads become system-scope FLAT loads that bypass the L2.)
__device__ __forceinline__ void tile_fetch(Stage& st, const _Float16* ch16, const uint8_t* row_label, int j0) {
  st.lab = row_label[j0 + (threadIdx.x & 31)];
  const int row = threadIdx.x >> 5, c = threadIdx.x & 31;   // piece q covers rows 8q + row
  const _Float16* p0 = ch16 + (size_t)(j0 + row) * kN + c * 8;
  st.a = *reinterpret_cast<const u32x4*>(p0);
  st.b = *reinterpret_cast<const u32x4*>(p0 + 8 * kN);
  st.c = *reinterpret_cast<const u32x4*>(p0 + 16 * kN);
  st.d = *reinterpret_cast<const u32x4*>(p0 + 24 * kN);
}
__device__ __forceinline__ void side_fetch(Stage& st, const _Float16* src) {
  st.side = *reinterpret_cast<const u32x4*>(src);
}
__device__ __forceinline__ void fetch_fence() { asm volatile("" ::: "memory"); }
__device__ __forceinline__ void tile_commit(const Stage& st, _Float16* __restrict__ cs) {
  const int row = threadIdx.x >> 5, c = threadIdx.x & 31;
  _Float16* p = cs + row * kPitchH + c * 8;
  *reinterpret_cast<u32x4*>(p) = st.a;
  *reinterpret_cast<u32x4*>(p + 8 * kPitchH) = st.b;
  *reinterpret_cast<u32x4*>(p + 16 * kPitchH) = st.c;
  *reinterpret_cast<u32x4*>(p + 24 * kPitchH) = st.d;
}

// The compiler's default schedule for the two GEMMs of a tile is `ds_read ; s_waitcnt ; v_mfma` sixteen times over -
// every LDS round trip (~180 cycles) exposed in front of a 32-cycle MFMA, one wave per SIMD and nobody to hide it.
// They are therefore split in a load phase and an MFMA phase: a tile step issues ALL its LDS reads first (the score
// fragments of tile t+1 and the value fragments of tile t, 48 instructions, fenced with sched_barrier so they stay
// there); the LDS pipe then streams them while the score MFMAs and the VALU epilogue of tile t run, and the waitcnt
// pass counts the reads down one MFMA at a time.
struct ScoreFrags {
  f16x8 c[16];
};
struct ValueFrags {
  h4 lo[8][2], hi[8][2];
};

__device__ __forceinline__ void load_score_frags(ScoreFrags& f, const _Float16* __restrict__ cs, int lane) {
  const _Float16* rowp = cs + (lane & 31) * kPitchH + 8 * (lane >> 5);
#pragma unroll
  for (int ks = 0; ks < 16; ++ks) f.c[ks] = *reinterpret_cast<const f16x8*>(rowp + 16 * ks);
}

__device__ __forceinline__ f32x16 mfma_scores(const ScoreFrags& f, const f16x8 (&a16)[16]) {
  f32x16 acc = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int ks = 0; ks < 16; ++ks) acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(f.c[ks], a16[ks], acc, 0, 0, 0);
  return acc;
}

__device__ __forceinline__ void load_value_frags(ValueFrags& f, const _Float16* __restrict__ cs, int lane) {
  const int half = lane >> 5, g = (lane >> 4) & 1, q = (lane & 15) >> 2, p = lane & 3;
  // this lane supplies the address of row (R0 + q), columns 4p..4p+3 of its 16-lane group's 4x16 block
  const _Float16* base = cs + (4 * half + q) * kPitchH + 16 * g + 4 * p;
#pragma unroll
  for (int nt = 0; nt < 8; ++nt) {
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      const _Float16* a0 = base + (16 * s) * kPitchH + 32 * nt;
      f.lo[nt][s] = __builtin_amdgcn_ds_read_tr16_b64_v4f16((__attribute__((address_space(3))) h4*)a0);
      f.hi[nt][s] = __builtin_amdgcn_ds_read_tr16_b64_v4f16((__attribute__((address_space(3))) h4*)(a0 + 8 * kPitchH));
    }
  }
}

// Y[n][i] += sum_j C[j][n] w[j][i]; w (fp32 accumulator layout, values in fp16 range) is the B operand
__device__ __forceinline__ void mfma_values(f32x16 (&acc)[8], const ValueFrags& f, const f32x16& w) {
  f16x8 bfrag[2];
#pragma unroll
  for (int s = 0; s < 2; ++s)
#pragma unroll
    for (int jj = 0; jj < 8; ++jj) bfrag[s][jj] = (_Float16)w[8 * s + jj];
#pragma unroll
  for (int nt = 0; nt < 8; ++nt) {
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      f16x8 a;
      a[0] = (_Float16)f.lo[nt][s][0]; a[1] = (_Float16)f.lo[nt][s][1]; a[2] = (_Float16)f.lo[nt][s][2]; a[3] = (_Float16)f.lo[nt][s][3];
      a[4] = (_Float16)f.hi[nt][s][0]; a[5] = (_Float16)f.hi[nt][s][1]; a[6] = (_Float16)f.hi[nt][s][2]; a[7] = (_Float16)f.hi[nt][s][3];
      acc[nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, bfrag[s], acc[nt], 0, 0, 0);
    }
  }
}

__device__ __forceinline__ f32x16 gemm_scores(const _Float16* __restrict__ cs, const f16x8 (&a16)[16], int lane) {
  ScoreFrags f;
  load_score_frags(f, cs, lane);
  __builtin_amdgcn_sched_barrier(0);
  return mfma_scores(f, a16);
}

__device__ __forceinline__ void gemm_values(f32x16 (&acc)[8], const _Float16* __restrict__ cs, const f32x16& w, int lane) {
  ValueFrags f;
  load_value_frags(f, cs, lane);
  __builtin_amdgcn_sched_barrier(0);
  mfma_values(acc, f, w);
}

__device__ __forceinline__ void store_values(const f32x16 (&acc)[8], float* __restrict__ dst, int half) {
#pragma unroll
  for (int nt = 0; nt < 8; ++nt)
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      float4 v = {acc[nt][4 * g + 0], acc[nt][4 * g + 1], acc[nt][4 * g + 2], acc[nt][4 * g + 3]};
      *reinterpret_cast<float4*>(dst + 32 * nt + 8 * g + 4 * half) = v;
    }
}

}  // namespace
}  // namespace ucd
