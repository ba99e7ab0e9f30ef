// Weight gradients of the stride-1 convolutions of the UCD student on the bf16 matrix cores (the
// backward of modules/residual.py:57-73 conv1 / conv2 / conv3 / proj_conv and modules/deeplab.py:24-37 map_convs / red_conv):
//
//     dW[n][t][k] = sum_m dZ[m][n] * X[shift_t(m)][k]        n < N out channels, k < K in channels, t < taps (1 or 9)
//
// on channels-last maps: dZ is the [M = B*H*W][N] row matrix of the output gradient, X the [M][K] row matrix of the layer's
// input, shift_t the pixel offset ((t/3 - 1) d, (t%3 - 1) d) of tap t of a 3x3 convolution with padding = dilation d (rows
// that fall off the map contribute zero), and dW comes out in the weight's own channels-last order [N][kh][kw][K].
//
// The reduction runs over M, the SLOW index of both operands - the opposite of the forward product - so
//   * both MFMA operands (8 consecutive reduction indices per lane) come from row-major [m][channel] LDS tiles through
//     ds_read_b64_tr_b16 (hardware transpose: a 16-lane group reads a 4-row x 16-column block, lane i receives column i);
//   * the output is small (N x taps K) and the reduction long (26 136 ... 399 384 rows): the rows are cut into chunks, every
//     workgroup owns one (output tile, tap, chunk), writes an fp32 slab, and a second kernel adds the slabs in a fixed order
//     (deterministic; float atomics would cap at 1.3 TB/s of added bytes and change bits from run to run).
// Tiling: 4 waves as 2 x 2 over a BNO x BKO output tile (128 or 64 wide: the 64-channel layers), 64 reduction rows per step,
// two LDS stages filled by LDS-DMA (buffer_load_dwordx4 ... lds: the 3x3 shift, the zero padding and the chunk's end are per-lane
// SOURCE offsets - an out-of-range offset for rows that do not exist, which the descriptor's range check zero-fills), the fill
// of step s+1 in flight under the 16 MFMAs of step s behind a counted vmcnt.  The LDS image is lane-linear, so the bank-conflict swizzle of the transposed reads sits on the
// source address too: 256-byte rows store chunk c of row r at c ^ (((r & 3) << 2) | ((r >> 2) & 3)), 128-byte rows at
// c ^ (((r >> 1) & 1) << 2) - the four rows of a transposed read then lie in four different 64-byte bank ranges.
// The 3x3 layers with 128-aligned channels and large maps take wgrad3_kernel (round 4, further down): one kernel ROW per workgroup -
// three taps from one dZ tile and one X tile of 64 + 2 d rows, 6 MFMA + 2 loader waves; this 9-tap form keeps the rest.
// Workgroup -> work: all tiles and taps of one row chunk sit on one XCD (ids congruent mod 8 share an L2), so a chunk's rows
// of dZ and X leave HBM once and are re-read by its tiles from that L2.
#include <map>
#include <vector>
#include <mutex>
#include <type_traits>

#include "common.h"

namespace ucd {
namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef __hip_bfloat16 bf16;
typedef __attribute__((address_space(1))) const void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

constexpr int kThreads = 256;
constexpr int kRows = 64;            // reduction rows per step

struct WArgs {
  const bf16* dZ; int ldz;
  const bf16* X; int ldx;
  int M, N, K;
  int taps, H, W, dil;
  int chunks, rows_per_chunk;
  int tiles_n, tiles_k, ksplit;     // ksplit: k-tile groups of one chunk that go to different XCDs (large outputs, few chunks)
  float* partial;                   // [chunks][N][taps * K]
  float inv_w, inv_hw;
  // strided layers (STR): dZ rows are the pixels of the [B, oH, oW] output map, X the [B, H, W, K] input map (x_rows rows);
  // output pixel (oy, ox) meets input pixel (oy, ox) * stride + the tap's shift
  int stride, oH, oW, x_rows;
  float inv_ow, inv_ohw;
};

// slot of logical 16-byte chunk c of row r in an LDS image with SLOTS chunks per row (see the header comment)
template <int SLOTS>
__device__ __forceinline__ int swz_slot(int r, int c) {
  return SLOTS == 16 ? (c ^ (((r & 3) << 2) | ((r >> 2) & 3))) : (c ^ (((r >> 1) & 1) << 2));
}

// exact m / d for 0 <= m < 2^22 and d < 2^15 from a float reciprocal and one correction step
__device__ __forceinline__ int fdiv(int m, int d, float inv) {
  int q = (int)((float)m * inv);
  int r = m - q * d;
  if (r < 0) { --q; r += d; }
  if (r >= d) ++q;
  return q;
}

typedef unsigned long long u64;
union Frag { bf16x8 v; u64 h[2]; };

// Transposed fragment reads of one 16-row group as inline asm: behind the builtin hipcc waits vmcnt(0) before the first LDS read
// of a step (it cannot tell the stage being read from the stage the LDS-DMA in flight is filling) and so drains the prefetch
// every step (found in the ISA; 416 TF/s).  An asm read is invisible to that bookkeeping, so its completion is counted by hand:
// a counted lgkmcnt per group (mma_group: the reads of the NEXT group, issued behind the previous MFMAs into the other register
// set, stay in flight).  Nothing else touches LDS or scalar memory between the first read of a stage and its last MFMA.
template <int TNW, int TKW, int IMM_Y, int IMM_X>
__device__ __forceinline__ void read_group(Frag (&fy)[TNW], Frag (&fx)[TKW], const unsigned (&yaddr)[TNW][2],
                                           const unsigned (&xaddr)[TKW][2]) {
#pragma unroll
  for (int a = 0; a < TNW; ++a) {
    asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(fy[a].h[0]) : "v"(yaddr[a][0]), "n"(IMM_Y));
    asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(fy[a].h[1]) : "v"(yaddr[a][1]), "n"(IMM_Y));
  }
#pragma unroll
  for (int b = 0; b < TKW; ++b) {
    asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(fx[b].h[0]) : "v"(xaddr[b][0]), "n"(IMM_X));
    asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(fx[b].h[1]) : "v"(xaddr[b][1]), "n"(IMM_X));
  }
}

// PENDING: transposed reads issued AFTER the ones this group consumes (they stay in flight: LDS returns in order)
template <int TNW, int TKW, int PENDING>
__device__ __forceinline__ void mma_group(f32x16 (&acc)[TNW][TKW], Frag (&fy)[TNW], Frag (&fx)[TKW]) {
  asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(PENDING) : "memory");
  __builtin_amdgcn_sched_barrier(0);                   // no MFMA above the wait (cdna_hip_programming.md 5.4 rule 18)
#pragma unroll
  for (int a = 0; a < TNW; ++a)
#pragma unroll
    for (int b = 0; b < TKW; ++b) acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fy[a].v, fx[b].v, acc[a][b], 0, 0, 0);
  __builtin_amdgcn_sched_barrier(0);                   // ... and the next reads stay behind these MFMAs
}

// the four 16-row groups of stage ST (byte offsets are compile-time immediates of the reads)
template <int BNO, int BKO, int ST, int TNW, int TKW>
__device__ __forceinline__ void compute_stage(f32x16 (&acc)[TNW][TKW], const unsigned (&yaddr)[TNW][2], const unsigned (&xaddr)[TKW][2]) {
  constexpr int kStage = kRows * (BNO + BKO) * 2, RY = BNO * 2, RX = BKO * 2, B = ST * kStage;
  Frag ay[TNW], ax[TKW], by[TNW], bx[TKW];
  constexpr int NR = 2 * (TNW + TKW);                   // reads of one group
  read_group<TNW, TKW, B + 0 * RY, B + 0 * RX>(ay, ax, yaddr, xaddr);
  read_group<TNW, TKW, B + 16 * RY, B + 16 * RX>(by, bx, yaddr, xaddr);
  mma_group<TNW, TKW, NR>(acc, ay, ax);
  read_group<TNW, TKW, B + 32 * RY, B + 32 * RX>(ay, ax, yaddr, xaddr);
  mma_group<TNW, TKW, NR>(acc, by, bx);
  read_group<TNW, TKW, B + 48 * RY, B + 48 * RX>(by, bx, yaddr, xaddr);
  mma_group<TNW, TKW, NR>(acc, ay, ax);
  mma_group<TNW, TKW, 0>(acc, by, bx);
}

// ---- slab sums, deferred (round 6) -----------------------------------------------------------------------------------------------
// dW = sum over the chunks' fp32 slabs in a fixed order (deterministic).  As a launch of its own this is a latency-bound kernel of
// 5 - 7 us behind every weight-gradient product (109 per step at 24 images: 0.65 ms; 106 at 3 images: 0.54 ms + their boundaries).
// Deferred: the sum of layer L rides in the launch of the NEXT weight-gradient product (layer L - 1 of the backward) as extra
// workgroups behind that kernel's own - no dependency between the two, so it hides under the product; the last layer's sum is
// launched alone by ucd_conv_wgrad_flush.  Same arithmetic in the same order: bit-identical gradients.  The caller promises that
// the gradient tensor is not read before the next weight-gradient call or the flush (the autograd nodes: their weight gradients go
// to AccumulateGrad and are first read by the bucket copies - ucd_amd/ddp.py flushes in front of those).
struct SumArgs {
  const float* partial; int chunks; unsigned long long total;
  bf16* dW; float* dW32; int accumulate;
  int cl;        // chunk lanes: 1, 4 or 16 (wgrad_sum_body<CL>)
  int blocks;    // workgroups of the sum (0: nothing pending)
};

// A thread owns 8 consecutive outputs of one chunk lane; CL lanes (1, 4 or 16: small outputs cut into many chunks - 64 x 256 weights
// over 250 chunks would otherwise be eight workgroups walking 250 slabs one after the other) take chunks c = lane, lane + CL, ...
// and are combined through LDS in lane order.  256 threads (further threads of a wider workgroup leave at once); ``red`` = at
// least CL * (256 / CL) * 9 floats of LDS.
template <int CL>
__device__ __forceinline__ void wgrad_sum_body(const SumArgs& q, int bid, float* red_) {
  if (threadIdx.x >= kThreads) return;
  constexpr int OG = kThreads / CL;                      // 8-output groups per workgroup
  float (*red)[OG][9] = reinterpret_cast<float (*)[OG][9]>(red_);
  const float* __restrict__ partial = q.partial;
  const size_t total = (size_t)q.total;
  const int chunks = q.chunks;
  const int og = threadIdx.x % OG, cl = threadIdx.x / OG;
  const size_t i = ((size_t)bid * OG + og) * 8;
  float s[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  if (i < total) {
    int c = cl;
    for (; c + CL < chunks; c += 2 * CL) {               // two slabs (four loads) in flight
      const float4 a0 = *reinterpret_cast<const float4*>(partial + (size_t)c * total + i);
      const float4 a1 = *reinterpret_cast<const float4*>(partial + (size_t)c * total + i + 4);
      const float4 b0 = *reinterpret_cast<const float4*>(partial + (size_t)(c + CL) * total + i);
      const float4 b1 = *reinterpret_cast<const float4*>(partial + (size_t)(c + CL) * total + i + 4);
      s[0] += a0.x; s[1] += a0.y; s[2] += a0.z; s[3] += a0.w; s[4] += a1.x; s[5] += a1.y; s[6] += a1.z; s[7] += a1.w;
      s[0] += b0.x; s[1] += b0.y; s[2] += b0.z; s[3] += b0.w; s[4] += b1.x; s[5] += b1.y; s[6] += b1.z; s[7] += b1.w;
    }
    if (c < chunks) {
      const float4 a0 = *reinterpret_cast<const float4*>(partial + (size_t)c * total + i);
      const float4 a1 = *reinterpret_cast<const float4*>(partial + (size_t)c * total + i + 4);
      s[0] += a0.x; s[1] += a0.y; s[2] += a0.z; s[3] += a0.w; s[4] += a1.x; s[5] += a1.y; s[6] += a1.z; s[7] += a1.w;
    }
  }
  if (CL > 1) {
#pragma unroll
    for (int e = 0; e < 8; ++e) red[cl][og][e] = s[e];
    // the 256 threads of the sum only (a wider workgroup's other waves have left): a named-free barrier over live waves
    __syncthreads();
    if (cl != 0) return;
#pragma unroll
    for (int l = 1; l < CL; ++l)
#pragma unroll
      for (int e = 0; e < 8; ++e) s[e] += red[l][og][e];
  }
  if (i >= total) return;
  if (q.dW) {
    Vec<bf16> o;
#pragma unroll
    for (int e = 0; e < 8; ++e) o.set(e, s[e]);
    o.store(q.dW + i);
  }
  if (q.dW32) {
    float4* d = reinterpret_cast<float4*>(q.dW32 + i);
    if (q.accumulate) {
      const float4 o0 = d[0], o1 = d[1];
      s[0] += o0.x; s[1] += o0.y; s[2] += o0.z; s[3] += o0.w; s[4] += o1.x; s[5] += o1.y; s[6] += o1.z; s[7] += o1.w;
    }
    d[0] = make_float4(s[0], s[1], s[2], s[3]);
    d[1] = make_float4(s[4], s[5], s[6], s[7]);
  }
}
__device__ __forceinline__ void wgrad_sum_dispatch(const SumArgs& q, int bid, float* red) {
  if (q.cl == 16) wgrad_sum_body<16>(q, bid, red);
  else if (q.cl == 4) wgrad_sum_body<4>(q, bid, red);
  else wgrad_sum_body<1>(q, bid, red);
}

template <int BNO, int BKO, bool T9, bool STR = false>
__global__ __launch_bounds__(kThreads, 2) void wgrad_kernel(WArgs p, SumArgs pend, int main_blocks) {
  constexpr int SY = BNO / 8, SX = BKO / 8;                 // 16-byte chunks per LDS row
  constexpr int YB = kRows * BNO * 2, XB = kRows * BKO * 2;  // bytes of the two tiles of a stage
  constexpr int kStage = YB + XB;
  constexpr int CY = YB / 1024 / 4, CX = XB / 1024 / 4;      // LDS-DMA instructions per wave and tile
  constexpr int TNW = BNO / 64, TKW = BKO / 64;              // 32-wide sub-tiles per wave
  extern __shared__ __attribute__((aligned(1024))) unsigned char smem[];
  if ((int)blockIdx.x >= main_blocks) {                      // the previous product's slab sum rides behind this launch's own workgroups
    wgrad_sum_dispatch(pend, (int)blockIdx.x - main_blocks, reinterpret_cast<float*>(smem));
    return;
  }

  // ---- which (chunk, tile, tap) ---------------------------------------------------------------------------------------
  const int tiles_kg = p.tiles_k / p.ksplit;                 // k tiles of one group
  const int per_group = p.tiles_n * tiles_kg * p.taps;
  const int xcd = blockIdx.x & 7, j = blockIdx.x >> 3;
  const int group = (j / per_group) * 8 + xcd;
  if (group >= p.chunks * p.ksplit) return;
  const int chunk = group / p.ksplit, kpart = group - chunk * p.ksplit;
  int rest = j % per_group;
  const int tap = rest % p.taps; rest /= p.taps;
  const int tn = rest % p.tiles_n, tk = kpart * tiles_kg + rest / p.tiles_n;
  const int n0 = tn * BNO, k0 = tk * BKO;
  const int mb = chunk * p.rows_per_chunk, me = min(p.M, mb + p.rows_per_chunk);
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // scalar: LDS-DMA destinations
  const int wn = wave >> 1, wk = wave & 1;
  const int dy = T9 ? (tap / 3 - 1) * p.dil : 0, dx = T9 ? (tap % 3 - 1) * p.dil : 0;
  const int shift = dy * p.W + dx;
  const int hw = p.H * p.W;

  f32x16 acc[TNW][TKW];
#pragma unroll
  for (int a = 0; a < TNW; ++a)
#pragma unroll
    for (int b = 0; b < TKW; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

  // ---- staging: lane l of instruction i of this wave lands on LDS row (wave * C + i) * RPI + l / SLOTS, slot l % SLOTS and
  // fetches the logical chunk slot ^ f(row) of that row ------------------------------------------------------------------
  constexpr int RPY = 64 / SY * 1, RPX = 64 / SX * 1;        // rows per 1 KiB instruction (4 for 256-byte rows, 8 for 128)
  int yrow[CY], ycol[CY], xrow[CX], xcol[CX];
#pragma unroll
  for (int i = 0; i < CY; ++i) {
    yrow[i] = (wave * CY + i) * RPY + lane / SY;
    ycol[i] = n0 + swz_slot<SY>(yrow[i], lane % SY) * 8;
  }
#pragma unroll
  for (int i = 0; i < CX; ++i) {
    xrow[i] = (wave * CX + i) * RPX + lane / SX;
    xcol[i] = k0 + swz_slot<SX>(xrow[i], lane % SX) * 8;
  }
  // Rows that do not exist (past the chunk, or shifted off the map) fetch through an OUT-OF-RANGE offset of a buffer descriptor:
  // the range check zero-fills the LDS bytes (tools/lds_dma_oob_probe.hip) - no branch around a load, no 64-bit select
  const auto rz = __builtin_amdgcn_make_buffer_rsrc((void*)p.dZ, 0, (int)((size_t)p.M * p.ldz * 2), 0x00020000);
  const auto rx = __builtin_amdgcn_make_buffer_rsrc((void*)p.X, 0, (int)((size_t)(STR ? p.x_rows : p.M) * p.ldx * 2), 0x00020000);
  constexpr int kOOB = 0x7FFFFFF0;
  // per-lane byte offsets inside a 64-row step, computed once; the step's first row enters as a SCALAR offset.  Only a step
  // that crosses the chunk's end (the last one) or a shifted tap tests rows.
  unsigned ybase[CY], xbase[CX];
#pragma unroll
  for (int i = 0; i < CY; ++i) ybase[i] = (unsigned)((yrow[i] * p.ldz + ycol[i]) * 2);
#pragma unroll
  for (int i = 0; i < CX; ++i) xbase[i] = (unsigned)((xrow[i] * p.ldx + xcol[i]) * 2);
  // T9: pixel coordinates of this lane's X rows, advanced by 64 rows per step instead of two divisions per row and step
  // (the kernel spent 45-48 % of its cycles issuing instructions, tools/prof_wgrad.sh); fill() is called for s = 0, 1, 2, ...
  // STR: the coordinates are those of the OUTPUT pixel (the dZ row) and xb its image's first X row; the X row of a step is then
  // a per-lane offset of its own (not m + shift)
  int xy[CX], xx[CX], xb[STR ? CX : 1];
  const int cw = STR ? p.oW : p.W, chh = STR ? p.oH : p.H;  // the map the rows m walk over
  const int q64 = (T9 || STR) ? kRows / cw : 0, r64 = (T9 || STR) ? kRows - q64 * cw : 0;
  const bool incremental = (T9 || STR) && q64 + 1 <= chh;   // one wrap per step is enough (always, for real maps)
  if (T9 || STR) {
    const int chw = STR ? p.oH * p.oW : hw;
    const float icw = STR ? p.inv_ow : p.inv_w, ichw = STR ? p.inv_ohw : p.inv_hw;
#pragma unroll
    for (int i = 0; i < CX; ++i) {
      const int m = mb + xrow[i];
      const int b = fdiv(m, chw, ichw);
      const int pix = m - b * chw;
      xy[i] = fdiv(pix, cw, icw);
      xx[i] = pix - xy[i] * cw;
      if (STR) xb[i] = b * hw;
    }
  }
  auto fill = [&](int s, unsigned char* stage) {
    const int m0 = mb + s * kRows;
    const bool whole = m0 + kRows <= me;                                  // wave-uniform
    // the scalar offset must not go negative (a shifted tap at the very first rows): the remainder rides on the lane offset,
    // which stays >= 0 for every row that exists (the range check looks at the lane offset)
    const int sx = max(0, m0 + shift);
    const int zs = m0 * p.ldz * 2, xs = sx * p.ldx * 2, xadj = (m0 + shift - sx) * p.ldx * 2;
#pragma unroll
    for (int i = 0; i < CY; ++i) {
      const unsigned off = (whole || m0 + yrow[i] < me) ? ybase[i] : (unsigned)kOOB;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rz, (lptr_t)(stage + (wave * CY + i) * 1024), 16, (int)off, zs, 0, 0);
    }
#pragma unroll
    for (int i = 0; i < CX; ++i) {
      const int m = m0 + xrow[i];
      bool ok = whole || m < me;
      int y = 0, x = 0, img = 0;
      if (T9 || STR) {
        if (incremental) {
          y = xy[i]; x = xx[i];
          int nx = x + r64, ny = y + q64;                    // the next step's coordinates
          if (nx >= cw) { nx -= cw; ++ny; }
          if (STR) img = xb[i];
          if (ny >= chh) { ny -= chh; if (STR) xb[i] += hw; }
          xx[i] = nx; xy[i] = ny;
        } else if (STR) {
          const int chw = p.oH * p.oW, b = fdiv(m, chw, p.inv_ohw), pix = m - b * chw;
          y = fdiv(pix, cw, p.inv_ow); x = pix - y * cw; img = b * hw;
        } else {
          const int pix = m - fdiv(m, hw, p.inv_hw) * hw;
          y = fdiv(pix, p.W, p.inv_w); x = pix - y * p.W;
        }
        if (STR) { y = y * p.stride; x = x * p.stride; }
        ok = ok && (unsigned)(y + dy) < (unsigned)p.H && (unsigned)(x + dx) < (unsigned)p.W;
      }
      if (STR) {   // the whole row address rides on the lane offset
        const unsigned off = ok ? (unsigned)(((img + (y + dy) * p.W + x + dx) * p.ldx + xcol[i]) * 2) : (unsigned)kOOB;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rx, (lptr_t)(stage + YB + (wave * CX + i) * 1024), 16, (int)off, 0, 0, 0);
        continue;
      }
      const unsigned off = ok ? (unsigned)((int)xbase[i] + xadj) : (unsigned)kOOB;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rx, (lptr_t)(stage + YB + (wave * CX + i) * 1024), 16, (int)off, xs, 0, 0);
    }
  };

  // ---- transposed fragment reads: byte offsets inside a 16-row group (independent of the group: see swz_slot) -----------
  // lane = 32 h + 16 g + 4 q + pp: rows 4 h + q (+ 8 for the second read), columns 16 g + 4 pp .. + 3 of a 32-wide sub-tile
  const int h = lane >> 5, g = (lane >> 4) & 1, q = (lane & 15) >> 2, pp = lane & 3;
  int yoff[TNW][2], xoff[TKW][2];
#pragma unroll
  for (int hi = 0; hi < 2; ++hi) {
    const int row = 4 * h + q + 8 * hi;
#pragma unroll
    for (int a = 0; a < TNW; ++a) {
      const int c = (wn * (BNO / 2) + a * 32 + 16 * g) / 8 + (pp >> 1);
      yoff[a][hi] = row * (BNO * 2) + swz_slot<SY>(row, c) * 16 + 8 * (pp & 1);
    }
#pragma unroll
    for (int b = 0; b < TKW; ++b) {
      const int c = (wk * (BKO / 2) + b * 32 + 16 * g) / 8 + (pp >> 1);
      xoff[b][hi] = YB + row * (BKO * 2) + swz_slot<SX>(row, c) * 16 + 8 * (pp & 1);
    }
  }
  const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)smem;
  unsigned yaddr[TNW][2], xaddr[TKW][2];
#pragma unroll
  for (int hi = 0; hi < 2; ++hi) {
#pragma unroll
    for (int a = 0; a < TNW; ++a) yaddr[a][hi] = lds0 + yoff[a][hi];
#pragma unroll
    for (int b = 0; b < TKW; ++b) xaddr[b][hi] = lds0 + xoff[b][hi];
  }

  const int nsteps = (me - mb + kRows - 1) / kRows;
  fill(0, smem);
  // one step: {all waves done with the stage about to be refilled; issue the next fill; wait for THIS step's fill (counted:
  // the next one stays in flight); barrier; 16 MFMAs}.  Unrolled by the two stages by hand - a runtime stage index makes the
  // compiler keep two register assignments of the accumulators and copy all 64 between them on every loop edge.
#define UCD_WG_STEP(s_, ST_)                                                                     \
  {                                                                                              \
    if (s_) __builtin_amdgcn_s_barrier();                                                        \
    if ((s_) + 1 < nsteps) {                                                                     \
      fill((s_) + 1, smem + (1 - ST_) * kStage);                                                 \
      asm volatile("s_waitcnt vmcnt(%0)" ::"n"(CY + CX) : "memory");                            \
    } else {                                                                                     \
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                                           \
    }                                                                                            \
    __builtin_amdgcn_s_barrier();                                                                \
    compute_stage<BNO, BKO, ST_, TNW, TKW>(acc, yaddr, xaddr);                                   \
  }
  for (int s = 0; s < nsteps; s += 2) {
    UCD_WG_STEP(s, 0)
    if (s + 1 < nsteps) UCD_WG_STEP(s + 1, 1)
  }
#undef UCD_WG_STEP

  // ---- fp32 slab of this (chunk, tile, tap): lanes 0..31 of a register write 32 consecutive k (128 bytes) ---------------
  const size_t ldp = (size_t)p.taps * p.K;
  float* dst = p.partial + ((size_t)chunk * p.N + n0) * ldp + (size_t)tap * p.K + k0;
#pragma unroll
  for (int a = 0; a < TNW; ++a)
#pragma unroll
    for (int b = 0; b < TKW; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int n = wn * (BNO / 2) + a * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
        dst[(size_t)n * ldp + wk * (BKO / 2) + b * 32 + (lane & 31)] = acc[a][b][r];
      }
}

// ---- 3x3 layers: one kernel ROW (three taps) per workgroup (round 4) ------------------------------------------------------------------
// The 9-tap form above stages dZ nine times and X nine times per output tile: 32 KB of LDS-DMA per 64 MFMA instructions, and the
// LDS-DMA path of a CU (~28 B/clk measured, profiles/r04_conv_pipe_probe.txt), not the matrix pipe, sets its time (0.19 - 0.23 of the
// MFMA peak).  Here a workgroup owns (output tile 128 x 128, kernel row kh, row chunk) and computes the taps kw = 0, 1, 2 of that row
// from ONE dZ tile and ONE X tile per 64-row step: tap kw pairs dZ row i with the input pixel d (kw - 1) to the right, which is row
// i + kw d of an X tile that starts d pixels early - 64 + 2 d rows cover all three taps (32.5 KB per 192 MFMA instructions at d = 1).
//   * rows whose pixel leaves the map vertically (y + dy outside [0, H)) or lies past the chunk depend on (m, kh) only: they are
//     zero rows of the dZ tile (out-of-range DMA offset, zero-filled) and so vanish from all three taps;
//   * horizontal validity depends on (m, kw): x + dx in [0, W).  Per step one wave writes, for kw = 0 and kw = 2, a mask dword per
//     PAIR of dZ rows (0xFFFF per valid row) into LDS; a lane's MFMA operand holds 8 reduction rows = 4 such pairs, so masking the
//     dZ fragments of a tap is 2 broadcast ds_read_b64 + 4 v_and per fragment (the centre tap reads a row of ones).
// 6 MFMA waves = (tap kw) x (n half): a 64 x 128 tile of one tap each (2 x 4 accumulators = 128 registers, two fragment sets); 2 LOADER
// waves that do all the staging (csrc/conv1x1.hip, loader-wave form: the issue of an LDS-DMA piece holds a wave for 60 - 185 cycles -
// with every wave loading AND multiplying a step took 3900 cycles for 1536 of MFMA; eight MFMA + four loader waves would leave 168
// registers per lane, which the 96 accumulators + fragments did not fit: 216 B of scratch); THREE LDS stages of 41 - 49 KB ({dZ 16 KB |
// masks | X 96 or 128 rows}), one barrier per step, one workgroup per CU; fragment reads as in the 9-tap form (asm, counted lgkmcnt).
constexpr int k3Threads = 512;          // 6 MFMA waves (tap x n-half) + 2 loader waves
constexpr int k3MaskOff = 16384, k3XOff = 17408;
constexpr int k3Stages = 3;               // two fills in flight: one workgroup per CU has nobody else to cover a fill's latency
__host__ __device__ constexpr int k3_stage_bytes(int nx) { return k3XOff + nx * 8 * 1024; }   // {dZ 16 KB | masks 1 KB | X: 8 waves x NX KB}

struct G3 { Frag y[2]; Frag x[4]; u64 m[2]; };
// reads of 16-row group GI (immediates: 16 rows = 4096 B of a tile, 8 row pairs = 32 B of a mask row): 14 LDS reads.  (Device
// functions, not statements of the kernel: inline asm with VGPR constraints in a __global__ template keeps the HOST pass from
// instantiating the kernel's stub.)
template <int GI>
__device__ __forceinline__ void w3_read(G3& gs, const unsigned (&yaddr)[2][2], const unsigned (&xaddr)[4][2], unsigned maddr) {
#pragma unroll
  for (int a = 0; a < 2; ++a) {
    asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(gs.y[a].h[0]) : "v"(yaddr[a][0]), "n"(GI * 4096));
    asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(gs.y[a].h[1]) : "v"(yaddr[a][1]), "n"(GI * 4096));
  }
#pragma unroll
  for (int b = 0; b < 4; ++b) {
    asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(gs.x[b].h[0]) : "v"(xaddr[b][0]), "n"(GI * 4096));
    asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(gs.x[b].h[1]) : "v"(xaddr[b][1]), "n"(GI * 4096));
  }
  asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(gs.m[0]) : "v"(maddr), "n"(GI * 32));
  asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(gs.m[1]) : "v"(maddr), "n"(GI * 32 + 16));
}
// PENDING = LDS reads issued after this group's (they stay in flight: LDS returns in order); the mask words multiply into the
// dZ operand (one AND pair per n sub-tile, shared by the four k sub-tiles)
template <int PENDING>
__device__ __forceinline__ void w3_mma(f32x16 (&acc)[2][4], G3& gs) {
  asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(PENDING) : "memory");
  __builtin_amdgcn_sched_barrier(0);
#pragma unroll
  for (int a = 0; a < 2; ++a) { gs.y[a].h[0] &= gs.m[0]; gs.y[a].h[1] &= gs.m[1]; }
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 4; ++b) acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(gs.y[a].v, gs.x[b].v, acc[a][b], 0, 0, 0);
  __builtin_amdgcn_sched_barrier(0);
}

template <int NX>   // X-tile LDS rows = 32 NX (96 or 128 >= 64 + 2 d); a loader wave issues 2 NX X pieces + 4 dZ pieces per step
__global__ __launch_bounds__(k3Threads, 1) void wgrad3_kernel(WArgs p, SumArgs pend, int main_blocks) {
  extern __shared__ __attribute__((aligned(1024))) unsigned char smem[];
  if ((int)blockIdx.x >= main_blocks) {
    wgrad_sum_dispatch(pend, (int)blockIdx.x - main_blocks, reinterpret_cast<float*>(smem));
    return;
  }
  constexpr int kStage = k3_stage_bytes(NX);
  constexpr int NXL = 4 * NX;                            // X pieces per loader wave and step
  constexpr int NFILL = 8 + NXL;                         // LDS-DMA instructions of one fill, per loader wave
  // ---- which (chunk, tile, kernel row) ----------------------------------------------------------------------------------
  const int tiles_kg = p.tiles_k / p.ksplit;                 // k tiles of one group (few chunks: their k tiles go to several XCDs)
  const int per_group = p.tiles_n * tiles_kg * 3;
  const int xcd = blockIdx.x & 7, j = blockIdx.x >> 3;
  const int group = (j / per_group) * 8 + xcd;
  if (group >= p.chunks * p.ksplit) return;
  const int chunk = group / p.ksplit, kpart = group - chunk * p.ksplit;
  int rest = j % per_group;
  const int kh = rest % 3; rest /= 3;
  const int tn = rest % p.tiles_n, tk = kpart * tiles_kg + rest / p.tiles_n;
  const int n0 = tn * 128, k0 = tk * 128;
  const int mb = chunk * p.rows_per_chunk, me = min(p.M, mb + p.rows_per_chunk);
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int d = p.dil, dy = (kh - 1) * d;
  const int W = p.W, H = p.H;
  const int nsteps = (me - mb + kRows - 1) / kRows;

  if (wave >= 6) {
    // ================================ loader waves (2) ================================
    const int lw = wave - 6;
    const int shift0 = dy * W - d;                      // X tile row 0 = pixel m0 + shift0
    const int XR = kRows + 2 * d;                       // rows of the X tile the three taps read
    const auto rz = __builtin_amdgcn_make_buffer_rsrc((void*)p.dZ, 0, (int)((size_t)p.M * p.ldz * 2), 0x00020000);
    const auto rx = __builtin_amdgcn_make_buffer_rsrc((void*)p.X, 0, (int)((size_t)p.M * p.ldx * 2), 0x00020000);
    constexpr int kOOB = 0x7FFFFFF0;
    // dZ: 16 pieces of 4 rows; this wave issues 8 lw .. 8 lw + 7
    int zrow[8], zy[8], zx[8];
    unsigned zbase[8];
    const int q64 = kRows / W, r64 = kRows - q64 * W;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      zrow[i] = (lw * 8 + i) * 4 + (lane >> 4);
      const int col = n0 + swz_slot<16>(zrow[i], lane & 15) * 8;
      zbase[i] = (unsigned)((zrow[i] * p.ldz + col) * 2);
      const int m = mb + zrow[i];
      const int b = fdiv(m, H * W, p.inv_hw), pix = m - b * H * W;
      zy[i] = fdiv(pix, W, p.inv_w);
      zx[i] = pix - zy[i] * W;
    }
    // X: pieces lw, lw + 2, ...: LDS row xrow, logical chunk by that row's swizzle
    int xrow[NXL], xlane[NXL];
#pragma unroll
    for (int i = 0; i < NXL; ++i) {
      xrow[i] = (lw + 2 * i) * 4 + (lane >> 4);
      xlane[i] = (xrow[i] * p.ldx + k0 + swz_slot<16>(xrow[i], lane & 15) * 8) * 2;
    }
    // masks: loader 0, lanes 0 .. 31 own the row pairs (2 l, 2 l + 1) of a step; x of row 2 l, advanced by 64 rows per step
    int mx = 0;
    {
      const int m = mb + 2 * (lane & 31);
      const int b = fdiv(m, H * W, p.inv_hw), pix = m - b * H * W;
      mx = pix - fdiv(pix, W, p.inv_w) * W;
    }
    auto fill = [&](int s, int st) {
      unsigned char* stage = smem + st * kStage;
      const int m0 = mb + s * kRows;
      const int zs = m0 * p.ldz * 2;
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const int y = zy[i], x = zx[i];
        int nx = x + r64, ny = y + q64;                   // the next step's coordinates
        if (nx >= W) { nx -= W; ++ny; }
        if (ny >= H) ny -= H;
        zx[i] = nx; zy[i] = ny;
        const bool ok = m0 + zrow[i] < me && (unsigned)(y + dy) < (unsigned)H;
        const int off = ok ? (int)zbase[i] : kOOB;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rz, (lptr_t)(stage + (lw * 8 + i) * 1024), 16, off, zs, 0, 0);
      }
      // X rows: pixel m0 + shift0 + row; the scalar part never negative (the remainder rides on the lane offset)
      const int p0 = m0 + shift0, sx = max(0, p0);
      const int xs = sx * p.ldx * 2, xadj = (p0 - sx) * p.ldx * 2;
#pragma unroll
      for (int i = 0; i < NXL; ++i) {
        // rows of the tile that no tap reads, pixels in front of the first image and past the last one: out-of-range offset (the
        // descriptor's range check sees the lane offset only, not the scalar part)
        const bool ok = xrow[i] < XR && (unsigned)(p0 + xrow[i]) < (unsigned)p.M;
        const int off = ok ? xlane[i] + xadj : kOOB;   // (a named value: with the conditional written in the call the host pass of this
                                                       // compiler silently drops the kernel's stub)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rx, (lptr_t)(stage + k3XOff + (lw + 2 * i) * 1024), 16, off, xs, 0, 0);
      }
      if (lw == 0) {
        const int x0 = mx;
        int x1 = x0 + 1;
        if (x1 >= W) x1 -= W;
        int nx = x0 + r64;
        if (nx >= W) nx -= W;
        mx = nx;
        if (lane < 32) {
          const unsigned m_l = (x0 >= d ? 0x0000FFFFu : 0u) | (x1 >= d ? 0xFFFF0000u : 0u);            // kw = 0: x - d >= 0
          const unsigned m_r = (x0 < W - d ? 0x0000FFFFu : 0u) | (x1 < W - d ? 0xFFFF0000u : 0u);      // kw = 2: x + d < W
          unsigned* mt = reinterpret_cast<unsigned*>(stage + k3MaskOff);
          mt[lane] = m_l;
          mt[32 + lane] = 0xFFFFFFFFu;                 // the centre tap reads a row of ones: one code path for the three taps
          mt[64 + lane] = m_r;
        }
      }
    };
    fill(0, 0);
    if (nsteps > 1) fill(1, 1);
    int fst = 2;                                         // stage of the next fill: (s + 2) % 3
    for (int s = 0; s < nsteps; ++s) {
      // fills issued so far: 0 .. min(s + 1, nsteps - 1); step s's has landed when at most that younger one is in flight
      if (s + 1 < nsteps) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NFILL) : "memory");
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // the mask words this wave wrote
      __builtin_amdgcn_s_barrier();                      // barrier s: the MFMA waves are done with step s - 1; step s is in LDS
      if (s + 2 < nsteps) {
        fill(s + 2, fst);
        fst = fst + 1 == k3Stages ? 0 : fst + 1;
      }
    }
    return;
  }

  // ================================ MFMA waves (6) ================================
  // wave = 3 wn + t: tap kw = t of the kernel row, n half wn: a 64 (n) x 128 (k) tile of ONE tap = 2 x 4 accumulators (128 registers);
  // per 16-row group 2 dZ fragments (masked with the tap's row masks) and 4 X fragments from the tile rows shifted by t d
  const int t = wave % 3, wn = wave / 3;
  f32x16 acc[2][4];
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 4; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;
  // ---- fragment addresses (stage 0; advanced by one stage per step, cyclically) ------------------------------------------------------
  const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)smem;
  const int h = lane >> 5, g = (lane >> 4) & 1, q = (lane & 15) >> 2, pp = lane & 3;
  unsigned yaddr[2][2], xaddr[4][2], maddr;
#pragma unroll
  for (int hi = 0; hi < 2; ++hi) {
    const int row = 4 * h + q + 8 * hi;
#pragma unroll
    for (int a = 0; a < 2; ++a) {
      const int c = (wn * 64 + a * 32 + 16 * g) / 8 + (pp >> 1);
      yaddr[a][hi] = lds0 + row * 256 + swz_slot<16>(row, c) * 16 + 8 * (pp & 1);
    }
    const int r = row + t * d;
#pragma unroll
    for (int b = 0; b < 4; ++b) {
      const int c = (b * 32 + 16 * g) / 8 + (pp >> 1);
      xaddr[b][hi] = lds0 + k3XOff + r * 256 + swz_slot<16>(r, c) * 16 + 8 * (pp & 1);
    }
  }
  maddr = lds0 + k3MaskOff + t * 128 + 8 * h;          // mask row of tap t; pairs 2 h, 2 h + 1 (rows 4 h .. 4 h + 3) of a 16-row group

  int rst = 0;
  for (int s = 0; s < nsteps; ++s) {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // (all reads of the previous step were consumed by its last group)
    __builtin_amdgcn_s_barrier();                      // barrier s
    G3 ga, gb;
    w3_read<0>(ga, yaddr, xaddr, maddr);
    w3_read<1>(gb, yaddr, xaddr, maddr);
    w3_mma<14>(acc, ga);
    w3_read<2>(ga, yaddr, xaddr, maddr);
    w3_mma<14>(acc, gb);
    w3_read<3>(gb, yaddr, xaddr, maddr);
    w3_mma<14>(acc, ga);
    w3_mma<0>(acc, gb);
    // the next stage (cyclic)
    rst = rst + 1 == k3Stages ? 0 : rst + 1;
    const unsigned adv = rst == 0 ? (unsigned)(-(k3Stages - 1) * kStage) : (unsigned)kStage;
#pragma unroll
    for (int hi = 0; hi < 2; ++hi) {
#pragma unroll
      for (int a = 0; a < 2; ++a) yaddr[a][hi] += adv;
#pragma unroll
      for (int b = 0; b < 4; ++b) xaddr[b][hi] += adv;
    }
    maddr += adv;
  }

  // ---- fp32 slab of this (chunk, tile, tap): lanes 0..31 of a register write 32 consecutive k (128 bytes) -----------------------------
  const size_t ldp = (size_t)9 * p.K;
  float* dst = p.partial + ((size_t)chunk * p.N + n0) * ldp + (size_t)(kh * 3 + t) * p.K + k0;
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 4; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int n = wn * 64 + a * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
        dst[(size_t)n * ldp + b * 32 + (lane & 31)] = acc[a][b][r];
      }
}

// the slab sum as a launch of its own (not deferred, or flushed): wgrad_sum_body above
__global__ __launch_bounds__(kThreads) void wgrad_sum_kernel(SumArgs q) {
  __shared__ float red[kThreads * 9];
  wgrad_sum_dispatch(q, (int)blockIdx.x, red);
}

struct Plan { int bno, bko, tiles_n, tiles_k, chunks, rows, ksplit; };

// chunks: about `target` workgroups in all (two per CU), whole 64-row steps, at least 256 rows each; a multiple of 8 chunks
// (one XCD each) where the output is small, else few chunks with the k tiles of a chunk split over the XCDs
Plan make_plan(int M, int N, int K, int taps, int target) {
  Plan pl;
  pl.bno = N % 128 == 0 ? 128 : 64;
  pl.bko = K % 128 == 0 ? 128 : 64;
  pl.tiles_n = N / pl.bno; pl.tiles_k = K / pl.bko;
  const int per_chunk = pl.tiles_n * pl.tiles_k * taps;
  int chunks = target / per_chunk;
  const int max_chunks = (M + 255) / 256;
  if (chunks > max_chunks) chunks = max_chunks;
  pl.ksplit = 1;
  if (chunks >= 8) {
    chunks = chunks / 8 * 8;
  } else {
    if (chunks < 1) chunks = 1;
    while (chunks & (chunks - 1)) --chunks;                        // 1, 2, 4
    int ks = 8 / chunks;
    while (ks > 1 && pl.tiles_k % ks) ks >>= 1;
    pl.ksplit = ks;
  }
  int rows = ceil_div(M, chunks);
  rows = ceil_div(rows, kRows) * kRows;
  pl.rows = rows;
  pl.chunks = ceil_div(M, rows);
  return pl;
}

// Workgroups aimed for (tools/wgrad_probe2.py on MI355X, B = 24): the 1x1 products are bound by the slab traffic
// (chunks x |dW| x 4 bytes written and read back) - one workgroup per CU; the 9-tap products are MFMA work at low per-workgroup
// efficiency - four per CU (two resident, two waiting) were fastest.  UCD_WGRAD_TARGET overrides (probes).
int wgrad_target(int taps, int per_chunk) {
  const char* e = getenv("UCD_WGRAD_TARGET");
  const int v = e ? atoi(e) : 0;
  if (v > 0) return v;
  if (taps == 1) return per_chunk >= 64 ? 512 : 256;
  return per_chunk <= 9 && taps == 9 ? 512 : 1024;
}

// three-tap form: one workgroup (8 waves) per CU; UCD_WGRAD3_TARGET overrides (probes), UCD_WGRAD3=0 keeps the 9-tap form (A/B)
struct Plan3 { int tiles_n, tiles_k, chunks, rows, ksplit; };
Plan3 make_plan3(int M, int N, int K) {
  Plan3 pl;
  pl.tiles_n = N / 128; pl.tiles_k = K / 128;
  const char* e = getenv("UCD_WGRAD3_TARGET");
  const int target = e && atoi(e) > 0 ? atoi(e) : 256;
  const int per_chunk = pl.tiles_n * pl.tiles_k * 3;
  int chunks = target / per_chunk;
  const int max_chunks = (M + 255) / 256;
  if (chunks > max_chunks) chunks = max_chunks;
  pl.ksplit = 1;
  if (chunks >= 8) {
    chunks = chunks / 8 * 8;
  } else {                                                         // few chunks: their k tiles spread over the XCDs
    if (chunks < 1) chunks = 1;
    while (chunks & (chunks - 1)) --chunks;                        // 1, 2, 4
    int ks = 8 / chunks;
    while (ks > 1 && pl.tiles_k % ks) ks >>= 1;
    pl.ksplit = ks;
  }
  int rows = ceil_div(M, chunks);
  rows = ceil_div(rows, kRows) * kRows;
  pl.rows = rows;
  pl.chunks = ceil_div(M, rows);
  return pl;
}
bool wgrad3_enabled() {
  const char* e = getenv("UCD_WGRAD3");
  return !(e && e[0] == '0');
}

// ---- deferred slab sums: host state (see SumArgs above) ---------------------------------------------------------------------------
// One pending sum per stream; ucd_conv_wgrad_ex(flags & 1) under ucd_conv_wgrad_defer(mode & 1) leaves its sum pending and carries the
// previous one of that stream in its launch.  g_defer holds the mode: bit 0 deferral, bit 1 the side stream (below).
std::mutex g_pend_mu;
std::map<hipStream_t, SumArgs> g_pend;
int g_defer = 0;

SumArgs make_sum(const void* workspace, int chunks, size_t total, void* dw, float* dw32, int accumulate32) {
  SumArgs q;
  q.partial = (const float*)workspace; q.chunks = chunks; q.total = total; q.dW = (bf16*)dw; q.dW32 = dw32; q.accumulate = accumulate32;
  // chunk lanes of the sum: enough workgroups to fill the chip when the output is small and the chunks many
  const size_t groups8 = total / 8;
  if (chunks >= 64 && groups8 <= (size_t)16 * 1024) { q.cl = 16; q.blocks = (int)((groups8 + 15) / 16); }
  else if (chunks >= 16 && groups8 <= (size_t)64 * 1024) { q.cl = 4; q.blocks = (int)((groups8 + 63) / 64); }
  else { q.cl = 1; q.blocks = (int)((groups8 + kThreads - 1) / kThreads); }
  return q;
}
SumArgs take_pending(hipStream_t s) {
  std::lock_guard<std::mutex> lock(g_pend_mu);
  SumArgs q{};
  auto it = g_pend.find(s);
  if (it != g_pend.end()) { q = it->second; g_pend.erase(it); }
  return q;
}

// ---- side stream (round 6) ----------------------------------------------------------------------------------------------------------
// Nothing in the backward pass waits for a weight gradient before the optimiser, but on the caller's stream every one of them sits in
// the chain of input-gradient products: at 3 - 6 images per GPU (the per-rank batch of the 8-GPU run) a step is ~850 launches of
// ~10 us, each waiting for the one before, on a chip the small launches fill to a quarter.  A call that allows it (flags & 2, mode
// bit 1) runs on a stream of the library instead: forked behind the caller's stream at the call (the operands are ready there),
// joined back by ucd_conv_wgrad_flush / _drop.  Under stream capture the fork and the join become the graph's edges.  The caller
// keeps dz, x, dw and the workspace alive until the join and does not touch dw before it.
//
// A fork can cost the CALLER's chain: in a replayed graph ROCm keeps the branch whose first node was created FIRST on the hardware
// queue of the fork point and continues the other branch on another queue, behind a cross-queue signal (~8 us).  Launching the side
// work right at the fork makes the caller's chain the branch that hops (kernel trace: runs of 2 - 4 kernels alternating between two
// queues, all weight gradients on the original one): one fork per call made the 3-image step 0.4 ms SLOWER, groups of 32 calls
// per fork 0.5 ms faster.  The form kept: the call only RECORDS its fork point on the caller's stream; the launches go out at the
// NEXT accepted call (or the flush), by when the caller has created its chain's next nodes - the weight gradients are the branch
// that hops, and nothing waits for them: 3 images 8.95 -> 7.95 ms, 6 images 12.0 -> 10.65, 24 images 28.95 -> 28.45
// (profiles/r06_side_stream.md).  UCD_WGRAD_STREAM_GROUP (default 1) calls share a fork point, UCD_WGRAD_STREAM_LATE=0 launches at
// the fork (the slower forms, kept for the A/B).
struct ExArgs {
  const void* dz; int ld_dz; const void* x; int ld_x; int M, N, K, taps, H, W, dilation, stride;
  void* dw; float* dw32; int accumulate32; void* workspace; size_t workspace_bytes; int flags;
};
bool wgrad_three(const ExArgs& q, int oW, int oH);
int wgrad_launch_on(const ExArgs& q, hipStream_t s);

struct Side {
  hipStream_t s = nullptr;
  hipEvent_t fork = nullptr, join = nullptr;
  bool active = false;      // work since the last join
  std::vector<ExArgs> queue;   // accepted calls not launched yet
  std::vector<ExArgs> armed;   // a full group whose fork point is recorded, launched at the next call (UCD_WGRAD_STREAM_LATE)
};
bool side_late() {
  static const bool on = [] { const char* e = getenv("UCD_WGRAD_STREAM_LATE"); return !(e && e[0] == '0'); }();
  return on;
}
int side_group() {
  static const int g = [] { const char* e = getenv("UCD_WGRAD_STREAM_GROUP"); const int v = e ? atoi(e) : 1; return v < 1 ? 1 : v; }();
  return g;
}
std::map<hipStream_t, Side> g_side;     // by the caller's stream; under g_pend_mu

// the side stream of `main`, forked behind its current position; nullptr (with the error set) when HIP refuses
// phase bit 0: record the fork point on `main`; bit 1: let the side stream wait for the recorded point
hipStream_t side_fork(hipStream_t main, const char* fn, int phase = 3) {
  std::lock_guard<std::mutex> lock(g_pend_mu);
  Side& sd = g_side[main];
  if (!sd.s) {
    int least = 0, greatest = 0;
    const char* e = getenv("UCD_WGRAD_STREAM_PRIO");
    const bool low = !(e && e[0] == 'n');                 // "normal": the caller's priority; default: the lowest (the chain goes first)
    hipStream_t st = nullptr;
    hipEvent_t f = nullptr, j = nullptr;
    bool ok = hipDeviceGetStreamPriorityRange(&least, &greatest) == hipSuccess &&
              hipStreamCreateWithPriority(&st, hipStreamNonBlocking, low ? least : 0) == hipSuccess &&
              hipEventCreateWithFlags(&f, hipEventDisableTiming) == hipSuccess &&
              hipEventCreateWithFlags(&j, hipEventDisableTiming) == hipSuccess;
    if (!ok) {
      (void)hipGetLastError();
      set_error("%s: cannot create the side stream of the weight gradients", fn);
      return nullptr;
    }
    sd.s = st; sd.fork = f; sd.join = j;
  }
  if ((phase & 1) && hipEventRecord(sd.fork, main) != hipSuccess) {
    (void)hipGetLastError();
    set_error("%s: cannot record the fork point on the caller's stream", fn);
    return nullptr;
  }
  if (phase & 2) {
    if (hipStreamWaitEvent(sd.s, sd.fork, 0) != hipSuccess) {
      (void)hipGetLastError();
      set_error("%s: cannot fork the side stream behind the caller's", fn);
      return nullptr;
    }
    sd.active = true;
  }
  return sd.s;
}
// the side stream of `main` when it holds work since the last join, else nullptr
hipStream_t side_active(hipStream_t main) {
  std::lock_guard<std::mutex> lock(g_pend_mu);
  auto it = g_side.find(main);
  return it != g_side.end() && it->second.active ? it->second.s : nullptr;
}
int side_join(hipStream_t main, const char* fn) {
  std::lock_guard<std::mutex> lock(g_pend_mu);
  auto it = g_side.find(main);
  if (it == g_side.end() || !it->second.active) return 0;
  it->second.active = false;
  if (hipEventRecord(it->second.join, it->second.s) != hipSuccess || hipStreamWaitEvent(main, it->second.join, 0) != hipSuccess) {
    (void)hipGetLastError();
    set_error("%s: cannot join the side stream of the weight gradients", fn);
    return (int)hipErrorUnknown;
  }
  return 0;
}

// launch the queued calls of `main` on its side stream behind one fork
int side_drain(hipStream_t main, const char* fn) {
  std::vector<ExArgs> todo;
  {
    std::lock_guard<std::mutex> lock(g_pend_mu);
    auto it = g_side.find(main);
    if (it == g_side.end() || it->second.queue.empty()) return 0;
    todo.swap(it->second.queue);
  }
  hipStream_t sd = side_fork(main, fn);
  if (!sd) return (int)hipErrorUnknown;
  for (const ExArgs& q : todo) {
    const int rc = wgrad_launch_on(q, sd);
    if (rc) return rc;
  }
  return 0;
}
// launch a group whose fork point was recorded earlier (the caller's stream has moved on since)
int side_launch_armed(hipStream_t main, const char* fn) {
  std::vector<ExArgs> todo;
  {
    std::lock_guard<std::mutex> lock(g_pend_mu);
    auto it = g_side.find(main);
    if (it == g_side.end() || it->second.armed.empty()) return 0;
    todo.swap(it->second.armed);
  }
  hipStream_t sd = side_fork(main, fn, 2);
  if (!sd) return (int)hipErrorUnknown;
  for (const ExArgs& q : todo) {
    const int rc = wgrad_launch_on(q, sd);
    if (rc) return rc;
  }
  return 0;
}
int side_enqueue(hipStream_t main, const ExArgs& q, const char* fn) {
  const int rc = side_launch_armed(main, fn);
  if (rc) return rc;
  size_t n;
  {
    std::lock_guard<std::mutex> lock(g_pend_mu);
    Side& sd = g_side[main];
    sd.queue.push_back(q);
    n = sd.queue.size();
  }
  if (n < (size_t)side_group()) return 0;
  if (!side_late()) return side_drain(main, fn);
  if (!side_fork(main, fn, 1)) return (int)hipErrorUnknown;      // the fork point now, the launches at the next call
  std::lock_guard<std::mutex> lock(g_pend_mu);
  Side& sd = g_side[main];
  sd.armed.swap(sd.queue);
  return 0;
}
void side_forget(hipStream_t main) {
  std::lock_guard<std::mutex> lock(g_pend_mu);
  auto it = g_side.find(main);
  if (it != g_side.end()) { it->second.queue.clear(); it->second.armed.clear(); }
}

int plan_target(int N, int K, int taps) {
  const int per_chunk = (N / (N % 128 == 0 ? 128 : 64)) * (K / (K % 128 == 0 ? 128 : 64)) * taps;
  return wgrad_target(taps, per_chunk);
}

}  // namespace
}  // namespace ucd

using namespace ucd;

extern "C" {

size_t ucd_conv_wgrad_workspace_bytes(int M, int N, int K, int taps) {
  if (M <= 0 || N <= 0 || K <= 0 || N % 64 || K % 64 || (taps != 1 && taps != 9)) return 0;
  const Plan pl = make_plan(M, N, K, taps, plan_target(N, K, taps));
  int chunks = pl.chunks;
  if (taps == 9 && N % 128 == 0 && K % 128 == 0) {     // the three-tap form may cut the rows differently
    const Plan3 p3 = make_plan3(M, N, K);
    if (p3.chunks > chunks) chunks = p3.chunks;
  }
  return (size_t)chunks * N * taps * K * sizeof(float);
}

int ucd_conv_wgrad(const void* dz, int ld_dz, const void* x, int ld_x, int M, int N, int K, int taps, int H, int W, int dilation,
                   void* dw, float* dw32, int accumulate32, void* workspace, size_t workspace_bytes, ucd_stream_t stream) {
  return ucd_conv_wgrad_strided(dz, ld_dz, x, ld_x, M, N, K, taps, H, W, dilation, 1, dw, dw32, accumulate32, workspace, workspace_bytes,
                                stream);
}

int ucd_conv_wgrad_strided(const void* dz, int ld_dz, const void* x, int ld_x, int M, int N, int K, int taps, int H, int W,
                           int dilation, int stride, void* dw, float* dw32, int accumulate32, void* workspace, size_t workspace_bytes,
                           ucd_stream_t stream) {
  return ucd_conv_wgrad_ex(dz, ld_dz, x, ld_x, M, N, K, taps, H, W, dilation, stride, dw, dw32, accumulate32, workspace, workspace_bytes, 0,
                           stream);
}

int ucd_conv_wgrad_defer(int mode) {
  std::lock_guard<std::mutex> lock(g_pend_mu);
  const int was = g_defer;
  g_defer = mode & 3;
  return was;
}

int ucd_conv_wgrad_mode(void) {
  std::lock_guard<std::mutex> lock(g_pend_mu);
  return g_defer;
}

int ucd_conv_wgrad_flush(ucd_stream_t stream) {
  static const char* fn = "ucd_conv_wgrad_flush";
  const hipStream_t main = (hipStream_t)stream;
  {
    int rc = side_launch_armed(main, fn);              // a group waiting for its launch
    if (!rc) rc = side_drain(main, fn);                // calls still waiting for their group
    if (rc) return rc;
  }
  if (hipStream_t sd = side_active(main)) {            // the side stream: its pending sum, then the join
    const SumArgs q = take_pending(sd);
    if (q.blocks > 0) {
      wgrad_sum_kernel<<<q.blocks, kThreads, 0, sd>>>(q);
      const int rc = check_launch(fn);
      if (rc) { (void)side_join(main, fn); return rc; }
    }
    const int rc = side_join(main, fn);
    if (rc) return rc;
  }
  const SumArgs q = take_pending(main);
  if (q.blocks <= 0) return 0;
  wgrad_sum_kernel<<<q.blocks, kThreads, 0, main>>>(q);
  return check_launch(fn);
}

int ucd_conv_wgrad_drop(ucd_stream_t stream) {
  const hipStream_t main = (hipStream_t)stream;
  (void)take_pending(main);
  side_forget(main);                                   // calls not launched yet never are
  if (hipStream_t sd = side_active(main)) {            // launched work cannot be taken back: join it (a capture must not end forked)
    (void)take_pending(sd);
    return side_join(main, "ucd_conv_wgrad_drop");
  }
  return 0;
}

int ucd_conv_wgrad_ex(const void* dz, int ld_dz, const void* x, int ld_x, int M, int N, int K, int taps, int H, int W, int dilation,
                      int stride, void* dw, float* dw32, int accumulate32, void* workspace, size_t workspace_bytes, int flags,
                      ucd_stream_t stream) {
  static const char* fn = "ucd_conv_wgrad";
  UCD_REQUIRE(dz && x && (dw || dw32) && workspace, UCD_EINVAL, "%s: NULL argument", fn);
  UCD_REQUIRE(M > 0 && N > 0 && K > 0 && N % 64 == 0 && K % 64 == 0, UCD_EUNSUPPORTED, "%s: N (%d) and K (%d) must be multiples of 64", fn, N, K);
  UCD_REQUIRE(taps == 1 || taps == 9, UCD_EINVAL, "%s: taps must be 1 or 9", fn);
  const bool str = stride > 1;
  const int oH = str && H > 0 ? (H - 1) / stride + 1 : H, oW = str && W > 0 ? (W - 1) / stride + 1 : W;
  UCD_REQUIRE((taps == 1 && !str) || (H > 0 && W > 0 && dilation >= 1 && (long long)M % ((long long)oH * oW) == 0 && H < 32768 && W < 32768),
              UCD_EINVAL, "%s: the 3x3 / strided forms need H, W, dilation and M = B*OH*OW", fn);
  UCD_REQUIRE(!str || (N % 128 == 0 && K % 128 == 0), UCD_EUNSUPPORTED, "%s: the strided form takes N and K multiples of 128", fn);
  UCD_REQUIRE(M < (1 << 22), UCD_EUNSUPPORTED, "%s: M = %d rows exceed the index arithmetic of the 3x3 form (2^22)", fn, M);
  const long long x_rows = str ? (long long)(M / (oH * oW)) * H * W : M;
  UCD_REQUIRE((size_t)M * ld_dz * 2 < 0x7FFFFFF0u && (size_t)x_rows * ld_x * 2 < 0x7FFFFFF0u, UCD_EUNSUPPORTED,
              "%s: operands beyond 2 GiB exceed the 32-bit offsets of the staging loads", fn);
  UCD_REQUIRE(aligned16(dz) && aligned16(x) && (!dw || aligned16(dw)) && (!dw32 || aligned16(dw32)) && ld_dz % 8 == 0 && ld_x % 8 == 0 &&
                  ld_dz >= N && ld_x >= K,
              UCD_EALIGN, "%s: operands must be 16-byte aligned with leading dimensions that are multiples of 8", fn);
  ExArgs q{dz, ld_dz, x, ld_x, M, N, K, taps, H, W, dilation, stride, dw, dw32, accumulate32, workspace, workspace_bytes, flags};
  {
    // enough workspace?  (checked here, at the call, also for a launch that waits in the side stream's queue)
    Plan pl = make_plan(M, N, K, taps, plan_target(N, K, taps));
    int chunks = pl.chunks;
    if (wgrad_three(q, oW, oH)) chunks = make_plan3(M, N, K).chunks;
    UCD_REQUIRE(workspace_bytes >= (size_t)chunks * N * taps * K * sizeof(float), UCD_EWORKSPACE, "%s: workspace too small", fn);
  }
  hipStream_t s = (hipStream_t)stream;
  if ((flags & 2) && (ucd_conv_wgrad_mode() & 2)) return side_enqueue(s, q, fn);      // off the caller's chain (see Side above)
  return wgrad_launch_on(q, s);
}

}  // extern "C"

namespace ucd {
namespace {

bool wgrad_three(const ExArgs& q, int oW, int oH) {
  (void)oH;
  // the three-tap form (wgrad3_kernel): 3x3, stride 1, 128-aligned channels, an X tile of 64 + 2 d rows in 128 LDS rows, maps on
  // which a 64-row step wraps at most one image row at a time
  return q.taps == 9 && q.stride <= 1 && q.N % 128 == 0 && q.K % 128 == 0 && q.dilation <= 18 && q.dilation < q.W && kRows / q.W + 1 <= q.H &&
         q.M >= 8192 && wgrad3_enabled();      // (small maps, 3 images per GPU: level with the 9-tap form or behind it)
}

int wgrad_launch_on(const ExArgs& q, hipStream_t s) {
  static const char* fn = "ucd_conv_wgrad";
  const void *dz = q.dz, *x = q.x;
  void *dw = q.dw, *workspace = q.workspace;
  float* dw32 = q.dw32;
  const int ld_dz = q.ld_dz, ld_x = q.ld_x, M = q.M, N = q.N, K = q.K, taps = q.taps, H = q.H, W = q.W, dilation = q.dilation,
            stride = q.stride, accumulate32 = q.accumulate32, flags = q.flags;
  const bool str = stride > 1;
  const int oH = str && H > 0 ? (H - 1) / stride + 1 : H, oW = str && W > 0 ? (W - 1) / stride + 1 : W;
  const long long x_rows = str ? (long long)(M / (oH * oW)) * H * W : M;
  Plan pl = make_plan(M, N, K, taps, plan_target(N, K, taps));
  const size_t total = (size_t)N * taps * K;
  // the three-tap form (wgrad3_kernel): 3x3, stride 1, 128-aligned channels, an X tile of 64 + 2 d rows in 128 LDS rows, maps on
  // which a 64-row step wraps at most one image row at a time
  const bool three = wgrad_three(q, oW, oH);
  Plan3 p3{0, 0, 0, 0, 1};
  if (three) {
    p3 = make_plan3(M, N, K);
    pl.chunks = p3.chunks;
  }
  WArgs a;
  a.dZ = (const bf16*)dz; a.ldz = ld_dz; a.X = (const bf16*)x; a.ldx = ld_x;
  a.M = M; a.N = N; a.K = K; a.taps = taps; a.H = (taps == 9 || str) ? H : 1; a.W = (taps == 9 || str) ? W : M; a.dil = dilation;
  a.stride = str ? stride : 1; a.oH = str ? oH : a.H; a.oW = str ? oW : a.W; a.x_rows = (int)x_rows;
  a.inv_ow = 1.f / (float)a.oW; a.inv_ohw = 1.f / ((float)a.oH * (float)a.oW);
  a.chunks = pl.chunks; a.rows_per_chunk = pl.rows; a.tiles_n = pl.tiles_n; a.tiles_k = pl.tiles_k; a.ksplit = pl.ksplit;
  a.partial = (float*)workspace;
  a.inv_w = 1.f / (float)a.W; a.inv_hw = 1.f / ((float)a.H * (float)a.W);
  const int groups = pl.chunks * pl.ksplit;
  const int per_group = pl.tiles_n * (pl.tiles_k / pl.ksplit) * taps;
  const int grid = ceil_div(groups, 8) * 8 * per_group;
  const size_t lds = (size_t)2 * kRows * (pl.bno + pl.bko) * 2;
  // the pending slab sum of this stream (a deferred earlier call) rides behind this launch's own workgroups; this call's own sum
  // is left pending when the caller allows it (flags & 1) and deferral is on, else launched right behind the product
  const SumArgs pend = take_pending(s);
  const int extra = pend.blocks > 0 ? pend.blocks : 0;
#define UCD_WG_LAUNCH(BN_, BK_)                                                             \
  {                                                                                         \
    if (taps == 9) {                                                                        \
      UCD_TRY_LDS((wgrad_kernel<BN_, BK_, true>), (int)lds);                                \
      wgrad_kernel<BN_, BK_, true><<<grid + extra, kThreads, lds, s>>>(a, pend, grid);      \
    } else {                                                                                \
      UCD_TRY_LDS((wgrad_kernel<BN_, BK_, false>), (int)lds);                               \
      wgrad_kernel<BN_, BK_, false><<<grid + extra, kThreads, lds, s>>>(a, pend, grid);     \
    }                                                                                       \
  }
  if (three) {
    a.chunks = p3.chunks; a.rows_per_chunk = p3.rows; a.tiles_n = p3.tiles_n; a.tiles_k = p3.tiles_k; a.ksplit = p3.ksplit;
    const int grid3 = ceil_div(p3.chunks * p3.ksplit, 8) * 8 * p3.tiles_n * (p3.tiles_k / p3.ksplit) * 3;
    if (kRows + 2 * dilation <= 96) {
      UCD_TRY_LDS((wgrad3_kernel<3>), k3Stages * k3_stage_bytes(3));
      wgrad3_kernel<3><<<grid3 + extra, k3Threads, k3Stages * k3_stage_bytes(3), s>>>(a, pend, grid3);
    } else {
      UCD_TRY_LDS((wgrad3_kernel<4>), k3Stages * k3_stage_bytes(4));
      wgrad3_kernel<4><<<grid3 + extra, k3Threads, k3Stages * k3_stage_bytes(4), s>>>(a, pend, grid3);
    }
  } else if (str) {
    if (taps == 9) {
      UCD_TRY_LDS((wgrad_kernel<128, 128, true, true>), (int)lds);
      wgrad_kernel<128, 128, true, true><<<grid + extra, kThreads, lds, s>>>(a, pend, grid);
    } else {
      UCD_TRY_LDS((wgrad_kernel<128, 128, false, true>), (int)lds);
      wgrad_kernel<128, 128, false, true><<<grid + extra, kThreads, lds, s>>>(a, pend, grid);
    }
  } else if (pl.bno == 128 && pl.bko == 128) UCD_WG_LAUNCH(128, 128)
  else if (pl.bno == 128) UCD_WG_LAUNCH(128, 64)
  else if (pl.bko == 128) UCD_WG_LAUNCH(64, 128)
  else UCD_WG_LAUNCH(64, 64)
#undef UCD_WG_LAUNCH
  int rc = check_launch(fn);
  if (rc) return rc;
  const SumArgs mine = make_sum(workspace, pl.chunks, total, dw, dw32, accumulate32);
  bool defer;
  {
    std::lock_guard<std::mutex> lock(g_pend_mu);
    defer = (g_defer & 1) && (flags & 1);
    if (defer) g_pend[s] = mine;
  }
  if (defer) return 0;
  wgrad_sum_kernel<<<mine.blocks, kThreads, 0, s>>>(mine);
  return check_launch(fn);
}

}  // namespace
}  // namespace ucd
