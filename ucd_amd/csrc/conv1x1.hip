// 1x1 convolutions of the UCD network as row-matrix GEMMs on the bf16 matrix cores, with the ABN work that surrounds
// them fused in (reference call sites modules/residual.py:57-73 conv1 / conv3 / proj_conv,
// modules/deeplab.py:24-37,56-58 map_convs[0] / red_conv).
//
// On channels-last activations a 1x1 convolution IS the GEMM  Y[M, N] = A[M, K] . W[N, K]^T  with M = B*H*W pixels,
// K = C_in, N = C_out: both operands are K-contiguous, which is exactly the operand layout of
// v_mfma_f32_32x32x16_bf16 (lane l holds 8 consecutive k of row / column l & 31).  At the network's shapes these
// products are HBM-bound (M = 26 136 ... 399 384 rows, K and N 64 ... 2048), so what matters is that the activation
// matrix is streamed ONCE and that the elementwise passes around the product never make their own trip to HBM:
//
//   input side   a' = act((a - mean_k) * scale_k + shift_k) applied while the A tile is staged into LDS: the ABN apply
//                of the PRODUCER layer (bn2 in front of conv3) never materialises;
//   output side  (1) evaluation / frozen statistics: y = act((acc - mean_n) * scale_n + shift_n + residual) - the ABN
//                    apply of THIS layer, the residual add and the block activation in the epilogue (the teacher runs
//                    its bottleneck 1x1 layers as one kernel each);
//                (2) training: y = bf16(acc) plus per-tile shifted sums (k, sum(y - k), sum(y - k)^2) per output channel
//                    - the statistics pass of the following ABN costs no read of y;
//                (3) backward of a fused input transform: the product is d a' (gradient w.r.t. the transformed input),
//                    the epilogue turns it into dz = d a' * act'(z(x)) and accumulates sum dz, sum dz * xhat per
//                    channel - the bwd_reduce pass of that ABN costs no read either;
//                (0) plain, optionally accumulating into y (the identity shortcut's gradient, beta = 1).
//   The input gradient of the layer is the same kernel on (dY, W^T); the weight gradient is conv1x1_wgrad below.
//
// Tiling: a workgroup of 4 waves owns a 128 x BN tile (BN = 128, or 64 for the 64-channel layers), each wave a
// 64 x BN/2 sub-tile as 2 x (BN/64) accumulators of 32 x 32; K is walked in steps of 64.  Both operand tiles are filled by
// LDS-DMA (global_load_lds_dwordx4: no registers, no ds_write); its destination is lane-linear, so the bank-conflict
// swizzle sits on the SOURCE address (slot s of row r lives at s ^ ((r >> 1) & 7): every 16-lane group of a ds_read_b128
// hits 16 different 16-byte slots without padding).  Two forms of the K loop:
//   single stage, four workgroups per CU (34 KB LDS, <= 128 VGPRs): {fill; vmcnt(0); barrier; 16 ds_read_b128 + 16 MFMA;
//     barrier} - the fill latency of one workgroup is covered by the MFMAs of the other three;
//   DB, two stages, two workgroups per CU: the fill of step k+1 is issued before the MFMAs of step k and waited for with
//     a counted vmcnt - for grids that give a CU only one or two workgroups (chosen by the host per launch).
//   loader waves (round 4, conv_lw_kernel below): the workgroup's MFMA waves never issue a fill and its loader waves never touch the
//     matrix pipe - three stages, one barrier per K step, one workgroup per CU on 128- or 256-row tiles; chosen by the grid (pick_pipe).
// Only the fused input transform (PRO) stages A through registers (it has to touch the values).  Rows past M are clamped,
// never branched around and never stored.  The output tile goes through LDS (fp32, two 64-row halves) so that global
// stores, residual loads and the per-channel reductions are row-contiguous 16-byte accesses.  blockIdx -> tile: the
// column tiles of one row strip sit on the same XCD (ids congruent mod 8 share an L2), so a strip of A leaves HBM once.
#include "common.h"
#include "abn_finalize.h"

namespace ucd {
namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __hip_bfloat16 bf16;

constexpr int kThreads = 256;
constexpr int kBM = 128;

constexpr int kBK = 64;

struct Args {
  const bf16* A; int lda;
  const bf16* W; int ldw;
  bf16* Y; int ldy;
  int M, N, K;
  const float *in_mean, *in_scale, *in_shift; int in_act; float in_slope;
  const float *out_mean, *out_scale, *out_shift, *out_invstd;
  const bf16* R; int ldr;
  const bf16* Z; int ldz;       // OUT == 4: the producer's pre-norm input (x-hat of its two backward sums)
  int out_act; float out_slope;
  float* partial;
  int accumulate;
  int tiles_m, tiles_n;
  int param_off;     // byte offset of the input-transform constants in LDS
  int iH, iW, dil;   // CONV3: spatial size of the [B, H, W, K] map behind A and the dilation (= padding) of the 3x3 taps
  int stride, oW, ohw;   // stride > 1 (1x1 and CONV3): output row m = (b, oy, ox) of the [B, oH, oW] map reads input pixel (oy, ox) * stride
  int a_rows;        // rows of A (= M unless strided: B * iH * iW)
  int tiles128;      // ceil(M / 128): rows of the per-tile partial buffers
  float* stat_acc;   // != NULL: column sums of OUT 2 / 3 / 4 go into this [2 N] accumulator with fp32 atomics (no partial rows)
  const float* stat_shift;   // OUT 2 with stat_acc: the common shift of the sums (the layer's running mean)
  float* stat_acc2;  // optional second accumulator (the same adds)
  int stat_rep;      // replicas of the accumulator (power of two): row tile t adds into replica t & (stat_rep - 1)
};

// The fused transforms take leaky_relu(slope) only; identity arrives as slope = 1 (elu layers keep the separate ABN
// kernels: a per-element expm1 makes every fused loop a chain of exec-masked branches, measured 2x on the epilogue).
__device__ __forceinline__ float act_rt(float z, float slope) { return z > 0.f ? z : z * slope; }
__device__ __forceinline__ float act_grad_rt(float z, float slope) { return z > 0.f ? 1.f : slope; }

// LDS image of a [rows][64] bf16 operand tile: 128-byte rows, the 16-byte slot s of row r stored at slot
// s ^ ((r >> 1) & 7).  Two rows share a 256-byte bank row; with this XOR the 16 lanes of every ds_read_b128 group
// (rows 0-3,12-15,20-27 / 4-11,16-19,28-31 of a 32-row fragment) hit 16 different slots: conflict-free without padding,
// which is what lets the tile be filled by global_load_lds (lane-linear destination, swizzle on the SOURCE address).
__device__ __forceinline__ int swz(int row, int slot) { return row * 128 + ((slot ^ ((row >> 1) & 7)) << 4); }
// The same for K steps of 32 (64-byte rows, four rows per 256-byte bank row): slot s of row r at s ^ ((r >> 2) & 3) - the rows
// of one ds_read_b128 group that share r & 3 (the same 64-byte quarter of a bank row) differ in (r >> 2) & 3.
template <int BK>
__device__ __forceinline__ int swz_fn(int row) { return BK == 64 ? ((row >> 1) & 7) : ((row >> 2) & 3); }
template <int BK>
__device__ __forceinline__ int swzk(int row, int slot) { return row * (BK * 2) + ((slot ^ swz_fn<BK>(row)) << 4); }

typedef __attribute__((address_space(1))) const void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

// OUT: 0 plain (+accumulate), 1 affine + residual + activation, 2 plain + statistics partials, 3 activation backward + sums,
//      4 block-output backward: acc (+ the shortcut's gradient already in y) is the gradient w.r.t. the OUTPUT of a residual
//        block act(norm(z) + shortcut); with its output R (the sign) and its conv3 output Z: y = d pre = (acc + y) * act'(R) and
//        the two backward sums of the block's last norm (sum d pre, sum d pre * xhat(Z)) - that norm then needs neither its
//        reduction pass nor a separate shortcut gradient (d shortcut = d pre, the same tensor).
//
// Structure: one LDS stage per workgroup and FOUR workgroups per CU (<= 128 VGPRs, 34 KB of LDS): a K step is
// {fill the stage: global_load_lds for W (and for A when it needs no transform), registers + transform for A otherwise;
// wait; barrier; 16 MFMAs per wave; barrier}, and the memory latency of one workgroup's fill is covered by the MFMAs of
// the three others - 128 KB of loads in flight per CU with no software pipeline for the compiler to undo.
// CONV3: the same kernel as an implicit-GEMM 3x3 convolution (stride 1, padding = dilation): the K loop runs over the 9
// taps x K/64 steps, the A tile of tap (kh, kw) is the row tile shifted by ((kh-1) d, (kw-1) d) pixels - with LDS-DMA the
// source address is per lane, so the shift and the zero padding (lanes outside the map fetch a 16-byte zero constant) are
// free; W is the channels-last 4-D weight [N][kh][kw][K] read as 9 [N][K] slices (row pitch 9 K).

// DB: two LDS stages, for grids that give a CU one or two workgroups (N = 256 at 33^2: 410 workgroups) instead of the four
// whose interleaving hides the fill latency of the single-stage form - the 9 K deep, MFMA-bound 3x3 products most of all: the
// fill of step k+1 is issued (LDS-DMA: no registers to carry) before the MFMAs of step k and waited for with a counted
// vmcnt, so it lands under them.  64 KB of LDS, two workgroups per CU.
// barriers executed by the epilogue (conv1x1_epilogue.inc): waves that only load run as many
template <int OUT> constexpr int kEpilogueBarriers = 4 + (OUT >= 2 ? 2 : 0);
template <int OUT, int HALVES> constexpr int kEpilogueBarriersH = 2 * HALVES + (OUT >= 2 ? 2 : 0);

// Pipeline depth of the DB form (round 4): NST stages of K steps of BK columns, fills NST - 1 steps ahead behind counted vmcnt
// waits.  <64, 2> is the original double buffer (64 KB, two workgroups per CU); <32, 4> keeps two workgroups per CU (4 x 16 KB
// stages each: 96 KB of fills in flight per CU instead of 64, three steps of latency cover instead of one) for the grids of
// 257 .. 640 workgroups (every 33 x 33 layer at B = 24); <64, 4> (128 KB, ONE workgroup per CU) is for grids that give a CU at most
// one workgroup anyway (the 3 - 6 images per GPU of the multi-GPU split: 26 - 104 workgroups, where a K step is one exposed
// memory round trip - 36 of them in a row for a 256-channel 3x3 layer).
// CONV3, stride 1: the contiguous range of kernel ROWS of a 3 x 3 tap grid that can meet the map for a tile of consecutive output rows
// [m0, m1) (raster order over the images).  Kernel row kh shifts every pixel by dy = (kh - 1) d; with d >= the tile's distance
// from the map's top / bottom edge all of its rows read padding and the whole kernel row contributes nothing - at the ASPP
// dilations (6, 12, 18 on a 33 x 33 map) that is 12 / 24 / 33 % of the K steps of a 128-row tile.  Returns the first live tap
// (0 or 3) and the number of live taps (3, 6 or 9); uniform integer arithmetic on the tile bounds, no reduction.
__device__ __forceinline__ void live_taps(int m0, int m1, int ohw, int oW, int iH, int dil, int& tap0, int& ntap) {
  tap0 = 0; ntap = 9;
  if (dil <= 1 || m1 - m0 >= ohw) return;                  // a tile that spans a whole image sees every row of the map
  const int first = m0 % ohw, last = (m1 - 1) % ohw;
  int ymin = first / oW, ymax = last / oW;
  if (first > last) { ymin = 0; ymax = iH - 1; }           // the tile crosses an image boundary: bottom rows of one, top rows of the next
  if (ymax < dil) { tap0 = 3; ntap -= 3; }                 // dy = -d: no row has a pixel d rows above it
  if (ymin >= iH - dil) ntap -= 3;                         // dy = +d
}

template <int BN, bool PRO, int OUT, bool CONV3 = false, bool DB = false, int BK = 64, int NST = 2>
__global__ __launch_bounds__(kThreads, DB ? (BK * NST >= 256 ? 1 : 2) : ((PRO || OUT >= 3) && BN == 128 ? 3 : 4)) void conv1x1_kernel(Args p) {
  static_assert(!DB || !PRO, "the double-buffered form has no input transform");
  static_assert(DB || (BK == 64 && NST == 2), "pipeline parameters belong to the DB form");
  static_assert(BK == 64 || BK == 32, "K steps of 64 or 32");
  static_assert(NST == 2 || NST == 4, "two or four stages");
  constexpr int kStage = (kBM + BN) * BK * 2;   // bytes of one LDS stage (A tile + W tile)
  constexpr int WN = BN / 2;           // columns per wave
  constexpr int TN = WN / 32;          // 32-wide accumulator tiles per wave along N
  extern __shared__ __attribute__((aligned(1024))) unsigned char smem[];
  unsigned char* As = smem;                            // [128][64] bf16, swizzled
  unsigned char* Bs = smem + kBM * BK * 2;             // [BN][BK]
  float* Ps = reinterpret_cast<float*>(smem + p.param_off);   // PRO: [3][K] input-transform constants, loaded once

  // tile of this workgroup: column tiles of one strip on the same XCD
  const int xcd = blockIdx.x & 7, j = blockIdx.x >> 3;
  const int tm = (j / p.tiles_n) * 8 + xcd, tn = j % p.tiles_n;
  if (tm >= p.tiles_m) return;
  const int m0 = tm * kBM, n0 = tn * BN;
  // the wave index as a SCALAR: every LDS-DMA destination is wave-uniform, and with a VGPR-derived index the compiler moves each
  // one through a VGPR + v_readfirstlane into M0 (25 of those per two K steps next to 32 MFMAs)
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;

  f32x16 acc[2][TN];
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < TN; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

  // ---- staging addresses -------------------------------------------------------------------------------------------
  // LDS-DMA: wave-instruction c covers tile rows 8c .. 8c+7 (1 KiB); lane l lands on (row 8c + l/8, slot l%8) and
  // therefore FETCHES logical slot (l%8) ^ ((row >> 1) & 7) of that row.  Rows past M are clamped (never stored).
  constexpr int RPC = 1024 / (BK * 2), SPR = BK / 8;    // rows per 1 KiB chunk (8 | 16), 16-byte slots per row (8 | 4)
  constexpr int CA = kBM / RPC / 4, CB = BN / RPC / 4;  // chunks per wave: A kBM / RPC chunks, B BN / RPC chunks, 4 waves
  const bf16* ga[CA];
  const bf16* gb[CB];
  int py[CA], px[CA], pimg[CA], pslot[CA];              // CONV3: pixel coordinates of this lane's staged rows
#pragma unroll
  for (int i = 0; i < CA; ++i) {
    const int row = RPC * (wave * CA + i) + lane / SPR;
    const int slot = ((lane % SPR) ^ swz_fn<BK>(row)) << 3;
    ga[i] = p.A + (size_t)min(m0 + row, p.M - 1) * p.lda + slot;
    if (CONV3 || p.stride > 1) {
      const int m = m0 + row;
      const int mm = min(m, p.M - 1);
      const int b = mm / p.ohw, rem = mm - b * p.ohw;
      const int oy = rem / p.oW, ox = rem - oy * p.oW;
      py[i] = m < p.M ? oy * p.stride : -(1 << 20);        // rows past M: never inside the map -> zeros
      px[i] = ox * p.stride;
      pimg[i] = b * p.iH * p.iW;
      pslot[i] = slot;
      if (!CONV3) ga[i] = p.A + (size_t)(pimg[i] + oy * p.stride * p.iW + px[i]) * p.lda + slot;   // strided 1x1: a row gather
    }
  }
#pragma unroll
  for (int i = 0; i < CB; ++i) {
    const int row = RPC * (wave * CB + i) + lane / SPR;
    gb[i] = p.W + (size_t)(n0 + row) * p.ldw + (((lane % SPR) ^ swz_fn<BK>(row)) << 3);
  }
  // register path of A (PRO): thread -> 16-byte slot ks of rows srow + 32 i
  const int ks = tid & 7, srow = tid >> 3;
  const bf16* arow[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) arow[i] = p.A + (size_t)min(m0 + srow + 32 * i, p.M - 1) * p.lda + ks * 8;

  if (PRO) {
    for (int k = tid; k < p.K; k += kThreads) {
      Ps[k] = p.in_mean[k];
      Ps[p.K + k] = p.in_scale[k];
      Ps[2 * p.K + k] = p.in_shift ? p.in_shift[k] : 0.f;
    }
    // visible after the first barrier of the K loop (the A commit that reads them comes after the global loads)
    __syncthreads();
  }
  const int fr = lane & 31, fh = lane >> 5;
  static_assert(!PRO || BK == 64, "the register-staged input transform walks K in steps of 64");
  const int kpt = p.K / BK;                            // K steps per tap
  int tap0 = 0, ntap = 9;
  if (CONV3 && p.stride == 1) live_taps(m0, min(m0 + kBM, p.M), p.ohw, p.oW, p.iH, p.dil, tap0, ntap);
  const int nk = CONV3 ? ntap * kpt : kpt;             // K steps of the live taps; step kb belongs to tap tap0 + kb / kpt
  uint4 ra[4];
  if (PRO) {
#pragma unroll
    for (int i = 0; i < 4; ++i) ra[i] = *reinterpret_cast<const uint4*>(arow[i]);
  }
  // Every LDS-DMA goes through a buffer descriptor: a per-lane 32-bit byte offset computed once (row, swizzled slot) plus a
  // SCALAR offset for the K position - no vector arithmetic per load in the K loop.
  const auto rsW = __builtin_amdgcn_make_buffer_rsrc((void*)p.W, 0, (int)((size_t)p.N * p.ldw * 2), 0x00020000);
  const auto rsA1 = __builtin_amdgcn_make_buffer_rsrc((void*)p.A, 0, (int)((size_t)p.a_rows * p.lda * 2), 0x00020000);
  unsigned boff[CB], a1off[CA];
#pragma unroll
  for (int i = 0; i < CB; ++i) boff[i] = (unsigned)((const char*)gb[i] - (const char*)p.W);
#pragma unroll
  for (int i = 0; i < CA; ++i) a1off[i] = (unsigned)((const char*)ga[i] - (const char*)p.A);
  // CONV3: the A tile of a tap is fetched through a buffer descriptor with 32-bit byte offsets computed ONCE per tap (shifted
  // pixel, bounds test); pixels outside the map carry an out-of-range offset, which the descriptor's range check zero-fills
  // (tools/lds_dma_oob_probe.hip) - the K steps inside a tap only add k0 (the first version re-derived tap, shift, bounds and a
  // 64-bit select per load and step: 36 % of the wave's cycles were instruction issue, tools/prof_kernel.sh)
  const auto rsA = __builtin_amdgcn_make_buffer_rsrc((void*)p.A, 0, CONV3 ? (int)((size_t)p.a_rows * p.lda * 2) : 0, 0x00020000);
  constexpr unsigned kOOB = 0x7FFFFFF0u;
  unsigned aoff[CA];
  int atap = -1;
  auto set_tap = [&](int tap) {
    const int dy = (tap / 3 - 1) * p.dil, dx = (tap % 3 - 1) * p.dil;
#pragma unroll
    for (int i = 0; i < CA; ++i) {
      const int yy = py[i] + dy, xx = px[i] + dx;
      const bool ok = (unsigned)yy < (unsigned)p.iH && (unsigned)xx < (unsigned)p.iW;
      aoff[i] = ok ? (unsigned)(((pimg[i] + yy * p.iW + xx) * p.lda + pslot[i]) * 2) : kOOB;
    }
    atap = tap;
  };
  auto fill3 = [&](int kb, unsigned char* Ad, unsigned char* Bd) {   // LDS-DMA fill of step kb (double-buffered form)
    const int kt = CONV3 ? kb / kpt : 0, tap = tap0 + kt;
    const int k0 = (kb - kt * kpt) * BK;
    if (CONV3) {
      if (tap != atap) set_tap(tap);
#pragma unroll
      for (int i = 0; i < CA; ++i)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsA, (lptr_t)(Ad + (wave * CA + i) * 1024), 16, (int)aoff[i], k0 * 2, 0, 0);
    } else {
#pragma unroll
      for (int i = 0; i < CA; ++i)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsA1, (lptr_t)(Ad + (wave * CA + i) * 1024), 16, (int)a1off[i], k0 * 2, 0, 0);
    }
#pragma unroll
    for (int i = 0; i < CB; ++i)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsW, (lptr_t)(Bd + (wave * CB + i) * 1024), 16, (int)boff[i], (tap * p.K + k0) * 2, 0, 0);
  };
  if (DB) {
#pragma unroll
    for (int st = 0; st < NST - 1; ++st)
      if (st < nk) fill3(st, As + st * kStage, Bs + st * kStage);
  }
  for (int kb = 0; kb < nk; ++kb) {
    const int kt = CONV3 ? kb / kpt : 0, tap = tap0 + kt;
    const int k0 = (kb - kt * kpt) * BK;
    const int wk0 = CONV3 ? tap * p.K + k0 : k0;       // column offset inside a weight row (pitch 9 K)
    if (kb) {                                          // the previous step's fragment reads are done
      if (DB) {   // raw barrier: __syncthreads() would wait for the fill in flight as well
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
      } else {
        __syncthreads();
      }
    }
    const unsigned char* Ac = As + (DB ? (kb & (NST - 1)) * kStage : 0);
    const unsigned char* Bc = Bs + (DB ? (kb & (NST - 1)) * kStage : 0);
    if (DB) {
      // the stage read in step kb - 1 is free (barrier above): the fill of step kb + NST - 1 goes there; then wait until only
      // the fills BEHIND this step's are still in flight (a counted vmcnt: r younger fills of CA + CB loads each)
      if (kb + NST - 1 < nk) {
        const int st = (kb + NST - 1) & (NST - 1);
        fill3(kb + NST - 1, As + st * kStage, Bs + st * kStage);
      }
      const int r = min(NST - 1, nk - 1 - kb);          // wave-uniform
      if (NST == 4 && r >= 3) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(3 * (CA + CB)) : "memory");
      else if (NST == 4 && r == 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * (CA + CB)) : "memory");
      else if (r >= 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(CA + CB) : "memory");
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
    } else if (PRO) {
#pragma unroll
      for (int i = 0; i < CB; ++i)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsW, (lptr_t)(Bs + (wave * CB + i) * 1024), 16, (int)boff[i], k0 * 2, 0, 0);
      float mu[8], sc[8], sh[8];
      {
        const float* pk = Ps + k0 + ks * 8;
        const float4 a0 = *reinterpret_cast<const float4*>(pk), a1 = *reinterpret_cast<const float4*>(pk + 4);
        const float4 b0 = *reinterpret_cast<const float4*>(pk + p.K), b1 = *reinterpret_cast<const float4*>(pk + p.K + 4);
        const float4 c0 = *reinterpret_cast<const float4*>(pk + 2 * p.K), c1 = *reinterpret_cast<const float4*>(pk + 2 * p.K + 4);
        mu[0] = a0.x; mu[1] = a0.y; mu[2] = a0.z; mu[3] = a0.w; mu[4] = a1.x; mu[5] = a1.y; mu[6] = a1.z; mu[7] = a1.w;
        sc[0] = b0.x; sc[1] = b0.y; sc[2] = b0.z; sc[3] = b0.w; sc[4] = b1.x; sc[5] = b1.y; sc[6] = b1.z; sc[7] = b1.w;
        sh[0] = c0.x; sh[1] = c0.y; sh[2] = c0.z; sh[3] = c0.w; sh[4] = c1.x; sh[5] = c1.y; sh[6] = c1.z; sh[7] = c1.w;
      }
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        Vec<bf16> v, o;
        v.raw = ra[i];
#pragma unroll
        for (int e = 0; e < 8; ++e) o.set(e, act_rt((v.get(e) - mu[e]) * sc[e] + sh[e], p.in_slope));
        *reinterpret_cast<uint4*>(As + swz(srow + 32 * i, ks)) = o.raw;
      }
      // the NEXT step's A rows go out now (always: past the end the last step is re-read and never used) and land under
      // this step's MFMAs; the counted wait below covers the W tile only (the 4 youngest loads stay in flight)
      const int kn = min(k0 + kBK, p.K - kBK);
#pragma unroll
      for (int i = 0; i < 4; ++i) ra[i] = *reinterpret_cast<const uint4*>(arow[i] + kn);
      asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    } else {
      if (CONV3) {
        if (tap != atap) set_tap(tap);
#pragma unroll
        for (int i = 0; i < CA; ++i)
          __builtin_amdgcn_raw_ptr_buffer_load_lds(rsA, (lptr_t)(As + (wave * CA + i) * 1024), 16, (int)aoff[i], k0 * 2, 0, 0);
      } else {
#pragma unroll
        for (int i = 0; i < CA; ++i)
          __builtin_amdgcn_raw_ptr_buffer_load_lds(rsA1, (lptr_t)(As + (wave * CA + i) * 1024), 16, (int)a1off[i], k0 * 2, 0, 0);
      }
#pragma unroll
      for (int i = 0; i < CB; ++i)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsW, (lptr_t)(Bs + (wave * CB + i) * 1024), 16, (int)boff[i], wk0 * 2, 0, 0);
    }
    if (!DB) {
      if (!PRO) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // this wave's ds_writes of the A tile have landed
      __builtin_amdgcn_s_barrier();                      // raw barrier: __syncthreads() would drain the A prefetch too
    }
    if (DB) {
      // double-buffered form (two waves per SIMD): the fragment reads run ONE 16-deep slice ahead of the MFMAs that consume them
      // (two register sets) - left to itself the compiler issues the reads of slice kk + 1 behind the MFMAs of slice kk.  ASPP
      // branches 245-256 -> 219-238 us; the single-stage forms (128-register cap) lose 5-15 % to the extra registers and keep the
      // plain loop.  (Also measured and dropped: four LDS stages at one workgroup per CU - 256->256 57 vs 40 us.)
      bf16x8 af[2][2], bfr[2][TN];
      auto read_slice = [&](int set, int kk) {
#pragma unroll
        for (int a = 0; a < 2; ++a) af[set][a] = *reinterpret_cast<const bf16x8*>(Ac + swzk<BK>(wm * 64 + a * 32 + fr, 2 * kk + fh));
#pragma unroll
        for (int b = 0; b < TN; ++b) bfr[set][b] = *reinterpret_cast<const bf16x8*>(Bc + swzk<BK>(wn * WN + b * 32 + fr, 2 * kk + fh));
      };
      read_slice(0, 0);
#pragma unroll
      for (int kk = 0; kk < BK / 16; ++kk) {
        if (kk + 1 < BK / 16) read_slice((kk + 1) & 1, kk + 1);
        __builtin_amdgcn_sched_barrier(0);                   // the reads above stay above these MFMAs
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
          for (int b = 0; b < TN; ++b)
            acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[kk & 1][a], bfr[kk & 1][b], acc[a][b], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
      }
    } else {
#pragma unroll
      for (int kk = 0; kk < kBK / 16; ++kk) {
        bf16x8 af[2], bfr[TN];
#pragma unroll
        for (int a = 0; a < 2; ++a) af[a] = *reinterpret_cast<const bf16x8*>(Ac + swz(wm * 64 + a * 32 + fr, 2 * kk + fh));
#pragma unroll
        for (int b = 0; b < TN; ++b) bfr[b] = *reinterpret_cast<const bf16x8*>(Bc + swz(wn * WN + b * 32 + fr, 2 * kk + fh));
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
          for (int b = 0; b < TN; ++b) acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[a], bfr[b], acc[a][b], 0, 0, 0);
      }
    }
  }

#include "conv1x1_epilogue.inc"
}

// ---- loader-wave form (round 4) ------------------------------------------------------------------------------------------------
// What bounds the forms above on the MFMA-bound products (3x3: K = 9 x 256 ... 9 x 2048) is not LDS or HBM but the ISSUE of the
// LDS-DMA: a `buffer_load_dwordx4 ... lds` piece (1 KiB) holds the issuing wave for 60 - 185 cycles (MI355X_MICROARCH.md, cycle
// constants), 8 pieces per wave and K step = 800 - 1500 cycles next to 16 MFMAs = 512: a wave of the DB form is ~34 % MFMA-busy, and
// two per SIMD reach ~67 % (profiles/r03_conv3x3_sq.txt).  Here the roles are split: waves 0 - 3 hold the 2 x 2 wave tiles and do
// nothing but {barrier; 16 ds_read_b128 + 16 MFMA} per K step; waves 4 - 7 only stage - they issue the fill of step kb + NST - 1
// right after barrier kb (the stage of step kb - 1 is free then), wait with a counted vmcnt until the fill of step kb + 1 has
// landed and join barrier kb + 1.  ONE barrier per K step, no DMA issue in an MFMA wave's instruction stream.
// BM = 256 (round 4): EIGHT MFMA waves (4 x 2 wave tiles of 64 x 64, two per SIMD) and eight loader waves on a 256 x BN workgroup
// tile, one workgroup per CU: a quarter less staged bytes per MFMA than two 128-row workgroups, and the MFMA waves of a SIMD take
// turns on its matrix pipe with no DMA issue between their instructions.  For the grids of 257 .. 640 128-row tiles (every 33 x 33
// layer at 24 images).  The two 128-row halves run the shared epilogue side by side (conv1x1_epilogue.inc, 256 threads and an LDS
// region each; the per-tile partial rows stay those of 128-row tiles).
// BM = 64 (round 6): TWO MFMA waves (one 64 x BN tile as two column halves) and four loader waves, for grids that leave most of the
// chip without a workgroup (3 images per GPU: 26 row tiles of 128).  What bounds a loader-wave workgroup that has a CU to itself is
// the CU's LDS-DMA rate (~26 - 40 B/clk: tools/probes/fill_rate.hip, lw_timeline.hip; profiles/r06_lw_probe.txt) - a K step of a
// 128 x 64 tile stages 24 KB and takes ~750 cycles with 256 cycles of MFMA in it - so the time of such a launch is the bytes ONE
// workgroup stages, whatever the number of workgroups (3 and 6 images: the same 12.9 / 13.6 us for the 3x3 256 -> 256 layer).
// 64-row tiles stage 16 KB per step on twice as many CUs.
template <int BM, int BN, int OUT, bool CONV3, int BK, int NST>
__global__ __launch_bounds__(BM == 64 ? 384 : BM * 4, (BM + BN) * BK * 2 * NST > 80 * 1024 ? 1 : 2) void conv_lw_kernel(Args p) {
  static_assert(BM == 64 || BM == 128 || BM == 256, "64-, 128- or 256-row workgroup tiles");
  static_assert(BM != 64 || BN == 64, "the 64-row form runs its epilogue on 128 threads: 64-column tiles");
  constexpr int NC = BM / 32, NL = BM == 64 ? 4 : BM / 32;   // MFMA waves (2 per 64 rows), loader waves
  static_assert(BK == 64 || BK == 32, "K steps of 64 or 32");
  static_assert(NST >= 2 && NST <= 4, "two to four stages");
  constexpr int kStage = (BM + BN) * BK * 2;
  constexpr int WN = BN / 2, TN = WN / 32;
  extern __shared__ __attribute__((aligned(1024))) unsigned char smem[];
  unsigned char* As = smem;
  unsigned char* Bs = smem + BM * BK * 2;
  const int xcd = blockIdx.x & 7, j = blockIdx.x >> 3;
  const int tmw = (j / p.tiles_n) * 8 + xcd, tn = j % p.tiles_n;      // workgroup tile (BM rows)
  if (tmw >= p.tiles_m) return;
  const int m0w = tmw * BM, n0 = tn * BN;
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int kpt = p.K / BK;
  int tap0 = 0, ntap = 9;
  if (CONV3 && p.stride == 1) live_taps(m0w, min(m0w + BM, p.M), p.ohw, p.oW, p.iH, p.dil, tap0, ntap);
  const int nk = CONV3 ? ntap * kpt : kpt;             // K steps of the live taps (loader and MFMA waves count the same steps)

  if (wave >= NC) {
    // ================================ loader waves ================================
    const int lw = wave - NC;
    const int m0 = m0w;
    constexpr int RPC = 1024 / (BK * 2), SPR = BK / 8;
    constexpr int CA = BM / RPC / NL, CB = BN / RPC / NL;
    int py[CA], px[CA], pimg[CA], pslot[CA];
    unsigned a1off[CA], boff[CB];
#pragma unroll
    for (int i = 0; i < CA; ++i) {
      const int row = RPC * (lw * CA + i) + lane / SPR;
      const int slot = ((lane % SPR) ^ swz_fn<BK>(row)) << 3;
      a1off[i] = (unsigned)(((size_t)min(m0 + row, p.M - 1) * p.lda + slot) * 2);
      if (CONV3 || p.stride > 1) {
        const int m = m0 + row;
        const int mm = min(m, p.M - 1);
        const int b = mm / p.ohw, rem = mm - b * p.ohw;
        const int oy = rem / p.oW, ox = rem - oy * p.oW;
        py[i] = m < p.M ? oy * p.stride : -(1 << 20);
        px[i] = ox * p.stride;
        pimg[i] = b * p.iH * p.iW;
        pslot[i] = slot;
        if (!CONV3) a1off[i] = (unsigned)(((size_t)(pimg[i] + oy * p.stride * p.iW + px[i]) * p.lda + slot) * 2);
      }
    }
#pragma unroll
    for (int i = 0; i < CB; ++i) {
      const int row = RPC * (lw * CB + i) + lane / SPR;
      boff[i] = (unsigned)(((size_t)(n0 + row) * p.ldw + (((lane % SPR) ^ swz_fn<BK>(row)) << 3)) * 2);
    }
    const auto rsW = __builtin_amdgcn_make_buffer_rsrc((void*)p.W, 0, (int)((size_t)p.N * p.ldw * 2), 0x00020000);
    const auto rsA = __builtin_amdgcn_make_buffer_rsrc((void*)p.A, 0, (int)((size_t)p.a_rows * p.lda * 2), 0x00020000);
    constexpr unsigned kOOB = 0x7FFFFFF0u;
    unsigned aoff[CA];
    int atap = -1;
    auto set_tap = [&](int tap) {
      const int dy = (tap / 3 - 1) * p.dil, dx = (tap % 3 - 1) * p.dil;
#pragma unroll
      for (int i = 0; i < CA; ++i) {
        const int yy = py[i] + dy, xx = px[i] + dx;
        const bool ok = (unsigned)yy < (unsigned)p.iH && (unsigned)xx < (unsigned)p.iW;
        aoff[i] = ok ? (unsigned)(((pimg[i] + yy * p.iW + xx) * p.lda + pslot[i]) * 2) : kOOB;
      }
      atap = tap;
    };
    auto fill = [&](int kb, int st) {
      unsigned char* Ad = As + st * kStage;
      unsigned char* Bd = Bs + st * kStage;
      const int kt = CONV3 ? kb / kpt : 0, tap = tap0 + kt;
      const int k0 = (kb - kt * kpt) * BK;
      if (CONV3) {
        if (tap != atap) set_tap(tap);
#pragma unroll
        for (int i = 0; i < CA; ++i)
          __builtin_amdgcn_raw_ptr_buffer_load_lds(rsA, (lptr_t)(Ad + (lw * CA + i) * 1024), 16, (int)aoff[i], k0 * 2, 0, 0);
      } else {
#pragma unroll
        for (int i = 0; i < CA; ++i)
          __builtin_amdgcn_raw_ptr_buffer_load_lds(rsA, (lptr_t)(Ad + (lw * CA + i) * 1024), 16, (int)a1off[i], k0 * 2, 0, 0);
      }
#pragma unroll
      for (int i = 0; i < CB; ++i)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsW, (lptr_t)(Bd + (lw * CB + i) * 1024), 16, (int)boff[i], (tap * p.K + k0) * 2, 0, 0);
    };
#pragma unroll
    for (int st = 0; st < NST - 1; ++st)
      if (st < nk) fill(st, st);
    int wst = NST - 1;                                  // stage the next fill goes to: (kb + NST - 1) % NST
    for (int kb = 0; kb < nk; ++kb) {
      // fills issued so far: 0 .. min(kb + NST - 2, nk - 1); the fill of step kb has to have landed: r younger ones stay in flight
      const int r = min(NST - 2, nk - 1 - kb);
      if (r >= 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * (CA + CB)) : "memory");
      else if (r == 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(CA + CB) : "memory");
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();                      // barrier kb: the MFMA waves are done with step kb - 1, step kb is in LDS
      if (kb + NST - 1 < nk) fill(kb + NST - 1, wst);
      wst = wst + 1 == NST ? 0 : wst + 1;
    }
#pragma unroll
    for (int b = 0; b < kEpilogueBarriersH<OUT, BM == 64 ? 1 : 2>; ++b) __syncthreads();
    return;
  }

  // ================================ MFMA waves ================================
  const int wmw = wave >> 1, wn = wave & 1;            // wave tile: rows 64 wmw .. + 63 of the workgroup tile
  f32x16 acc[2][TN];
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < TN; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;
  const int fr = lane & 31, fh = lane >> 5;
  int rst = 0;                                          // stage of step kb: kb % NST
  for (int kb = 0; kb < nk; ++kb) {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // the previous step's fragment reads have returned
    __builtin_amdgcn_s_barrier();
    const unsigned char* Ac = As + rst * kStage;
    const unsigned char* Bc = Bs + rst * kStage;
    rst = rst + 1 == NST ? 0 : rst + 1;
    bf16x8 af[2][2], bfr[2][TN];
    auto read_slice = [&](int set, int kk) {
#pragma unroll
      for (int a = 0; a < 2; ++a) af[set][a] = *reinterpret_cast<const bf16x8*>(Ac + swzk<BK>(wmw * 64 + a * 32 + fr, 2 * kk + fh));
#pragma unroll
      for (int b = 0; b < TN; ++b) bfr[set][b] = *reinterpret_cast<const bf16x8*>(Bc + swzk<BK>(wn * WN + b * 32 + fr, 2 * kk + fh));
    };
    read_slice(0, 0);
#pragma unroll
    for (int kk = 0; kk < BK / 16; ++kk) {
      if (kk + 1 < BK / 16) read_slice((kk + 1) & 1, kk + 1);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < TN; ++b)
          acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[kk & 1][a], bfr[kk & 1][b], acc[a][b], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  // epilogue: each group of four MFMA waves (one 128-row half of the workgroup tile) runs the shared 256-thread epilogue on its
  // own LDS region; the barriers inside are workgroup-wide and the groups execute the same sequence of them
  if constexpr (BM == 64) {
    // one 64-row tile, 128 epilogue threads; per-tile partial rows do not exist for 64-row tiles (the host takes this form only with
    // the atomic statistics accumulators or without statistics)
    const int tm = tmw, m0 = m0w, wm = 0;
#define UCD_EPI_THREADS 128
#define UCD_EPI_HALVES 1
#include "conv1x1_epilogue.inc"
#undef UCD_EPI_THREADS
#undef UCD_EPI_HALVES
  } else {
    const int sub = wmw >> 1;
    const int tm = tmw * (BM / 128) + sub, m0 = m0w + sub * 128, wm = wmw & 1;
    constexpr size_t kOutBytes = ((size_t)64 * (BN + 4) * 4 + 1023) / 1024 * 1024;
    unsigned char* smem_all = smem;
    {
      unsigned char* smem = smem_all + sub * kOutBytes;
      const int tid_all = tid;
      {
        const int tid = tid_all & (kThreads - 1);
#define UCD_EPI_TILE_OK (tm < p.tiles128)
#include "conv1x1_epilogue.inc"
#undef UCD_EPI_TILE_OK
      }
    }
  }
}

// (The resident-A form of the short-K, wide-N products - conv_ra_kernel, round 5: A rows resident in LDS, W rows straight from L2 into
// registers, separate epilogue waves - measured slower than the tiled forms, 29.3 vs 22.2 us at 256 -> 1024, and was removed in round 6;
// DESIGN.md section 10, profiles/r05_conv_ra_probe.txt, git history.)

// Per-tile shifted sums (k_t, s1_t, s2_t) -> sums about the common shift K = k_0 -> the usual finalize.
//   sum (y - K) = s1_t + c_t (k_t - K),   sum (y - K)^2 = s2_t + 2 (k_t - K) s1_t + c_t (k_t - K)^2   (exact identities)
// A workgroup of 1024 threads owns 8 channels x 128 tile lanes, four tiles (twelve loads) in flight per thread; fixed
// combination order (deterministic).  MODE 1 finalises, MODE 2 packs (mean_r, M2_r) for the SyncBN all-gather.
// Measured alternatives (tools/conv1x1_probe.py, 3121 row tiles x 64 channels, the 129^2 layers): 256 threads with one tile
// in flight 24-34 us; a two-stage form (64 channels x a group of tiles per workgroup, the last arrival - an atomic ticket
// behind __threadfence - finalising) 20-34 us: the device-scope fences cost more than the second launch they save.
constexpr int kStatThreads = 1024;
template <int MODE>
__global__ __launch_bounds__(kStatThreads) void tile_stats_reduce_kernel(const float* __restrict__ partial, int tiles, int M, int C,
                                                                        float* __restrict__ sums, float* __restrict__ kout,
                                                                        FinalizeArgs fin) {
  constexpr int TL = kStatThreads / 8;
  __shared__ float lds[2][TL][9];
  const int ch = threadIdx.x & 7, tl = threadIdx.x >> 3;
  const int c = blockIdx.x * 8 + ch;
  float S1 = 0.f, S2 = 0.f, K = 0.f;
  FinalizeIn pre{1.f, 0.f, 0.f};
  if (MODE == 1 && tl == 0 && c < C) pre = finalize_prefetch(c, fin);      // in flight under the whole reduction
  if (c < C) {
    K = partial[c];
    auto add = [&](int t, float k, float a, float b) {
      const float cnt = (float)min(kBM, M - t * kBM);
      const float dk = k - K;
      S1 += a + cnt * dk;
      S2 += b + 2.f * dk * a + cnt * dk * dk;
    };
    int t = tl;
    for (; t + 3 * TL < tiles; t += 4 * TL) {
      const float* p0 = partial + (size_t)t * 3 * C + c;
      const float* p1 = p0 + (size_t)TL * 3 * C;
      const float* p2 = p1 + (size_t)TL * 3 * C;
      const float* p3 = p2 + (size_t)TL * 3 * C;
      const float k0 = p0[0], a0 = p0[C], b0 = p0[2 * C], k1 = p1[0], a1 = p1[C], b1 = p1[2 * C];
      const float k2 = p2[0], a2 = p2[C], b2 = p2[2 * C], k3 = p3[0], a3 = p3[C], b3 = p3[2 * C];
      add(t, k0, a0, b0); add(t + TL, k1, a1, b1); add(t + 2 * TL, k2, a2, b2); add(t + 3 * TL, k3, a3, b3);
    }
    for (; t + TL < tiles; t += 2 * TL) {   // the bench's 205 row tiles = two tiles per lane: both in flight (was one round trip each)
      const float* p0 = partial + (size_t)t * 3 * C + c;
      const float* p1 = p0 + (size_t)TL * 3 * C;
      const float k0 = p0[0], a0 = p0[C], b0 = p0[2 * C], k1 = p1[0], a1 = p1[C], b1 = p1[2 * C];
      add(t, k0, a0, b0); add(t + TL, k1, a1, b1);
    }
    for (; t < tiles; t += TL) {
      const float* pt = partial + (size_t)t * 3 * C + c;
      add(t, pt[0], pt[C], pt[2 * C]);
    }
  }
  lds[0][tl][ch] = S1;
  lds[1][tl][ch] = S2;
  __syncthreads();
  if (tl < 8) {   // 128 tile lanes -> 8, then one
    float t1 = 0.f, t2 = 0.f;
#pragma unroll
    for (int i = 0; i < TL / 8; ++i) { t1 += lds[0][tl * (TL / 8) + i][ch]; t2 += lds[1][tl * (TL / 8) + i][ch]; }
    lds[0][tl * (TL / 8)][ch] = t1;
    lds[1][tl * (TL / 8)][ch] = t2;
  }
  __syncthreads();
  if (tl == 0 && c < C) {
    float t1 = 0.f, t2 = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i) { t1 += lds[0][i * (TL / 8)][ch]; t2 += lds[1][i * (TL / 8)][ch]; }
    sums[c] = t1;
    sums[C + c] = t2;
    kout[c] = K;
    if (MODE == 1) {
      finalize_channel_pre(c, t1, t2, K, fin, pre);
    } else {
      const float d = t1 / fin.count;
      fin.pack[c] = K + d;
      fin.pack[C + c] = t2 - t1 * d;
    }
  }
}

// ---- weight gradient: dW[N, K] = dY[M, N]^T . A'[M, K] ------------------------------------------------------------------
// The reduction runs over the M = B*H*W rows, the slow dimension of BOTH operands, so the MFMA fragments (8 consecutive
// reduction indices per lane) are fetched with ds_read_b64_tr_b16 from row-major [m][n] / [m][k] LDS tiles (hardware
// transpose: a 16-lane group reads a 4-row x 16-column block, lane i receives column i).  Both operands go through the
// same read pattern, so the permutation of the reduction index inside a step is the same on both sides.
// Grid: (row chunks, N/128, K/128); every workgroup writes an fp32 [128 x 128] partial of its row chunk, summed by
// wgrad_reduce_kernel (fixed order) into the bf16 weight gradient.  The input transform of the forward (PRO) is
// re-applied to A here, so the transformed activation never exists in memory.
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));

template <bool PRO>
__global__ __launch_bounds__(kThreads, 3) void conv1x1_wgrad_kernel(const bf16* __restrict__ dY, int ldy,
                                                                   const bf16* __restrict__ A, int lda, int M, int N, int K,
                                                                   const float* __restrict__ in_mean,
                                                                   const float* __restrict__ in_scale,
                                                                   const float* __restrict__ in_shift, int in_act,
                                                                   float in_slope, int rows_per_chunk,
                                                                   float* __restrict__ partial) {
  constexpr int BT = 128;              // output tile: 128 (n) x 128 (k)
  constexpr int BMS = 32;              // reduction rows per step
  constexpr int P = BT + 8;            // bf16 pitch of the [m][n] / [m][k] LDS tiles (272 bytes)
  __shared__ __attribute__((aligned(16))) bf16 Ys[1][BMS][P];
  __shared__ __attribute__((aligned(16))) bf16 Xs[1][BMS][P];
  const int n0 = blockIdx.y * BT, k0 = blockIdx.z * BT;
  const int mb = blockIdx.x * rows_per_chunk, me = min(M, mb + rows_per_chunk);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wn = wave >> 1, wk = wave & 1;             // wave tile: 64 (n) x 64 (k)
  const int srow = tid >> 4, scol = (tid & 15) * 8;     // staging: 16 rows x 16 chunks per pass, 2 passes

  f32x16 acc[2][2];
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;
  float mu[8], sc[8], sh[8];
  if (PRO) {
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      mu[e] = in_mean[k0 + scol + e]; sc[e] = in_scale[k0 + scol + e]; sh[e] = in_shift ? in_shift[k0 + scol + e] : 0.f;
    }
  }
  // rows past the chunk end are clamped (no branch around a load) and zeroed with a mask at commit time: they would
  // otherwise enter the reduction
  uint4 ry0[2], rx0[2];
  auto fetch = [&](uint4 (&ry)[2], uint4 (&rx)[2], int m) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int row = min(m + srow + 16 * i, me - 1);
      ry[i] = *reinterpret_cast<const uint4*>(dY + (size_t)row * ldy + n0 + scol);
      rx[i] = *reinterpret_cast<const uint4*>(A + (size_t)row * lda + k0 + scol);
    }
  };
  auto commit = [&](const uint4 (&ry)[2], const uint4 (&rx)[2], int buf, int m) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const unsigned keep = (m + srow + 16 * i < me) ? 0xFFFFFFFFu : 0u;
      uint4 y4 = ry[i];
      y4.x &= keep; y4.y &= keep; y4.z &= keep; y4.w &= keep;
      *reinterpret_cast<uint4*>(&Ys[buf][srow + 16 * i][scol]) = y4;
      if (PRO) {
        Vec<bf16> v, o;
        v.raw = rx[i];
#pragma unroll
        for (int e = 0; e < 8; ++e) o.set(e, act_rt((v.get(e) - mu[e]) * sc[e] + sh[e], in_slope));
        *reinterpret_cast<uint4*>(&Xs[buf][srow + 16 * i][scol]) = o.raw;     // dY is zero on the masked rows
      } else {
        *reinterpret_cast<uint4*>(&Xs[buf][srow + 16 * i][scol]) = rx[i];
      }
    }
  };
  // transposed fragment: 8 reduction rows (two 4-row blocks) of column `col` for this lane
  const int h = lane >> 5, g = (lane >> 4) & 1, q = (lane & 15) >> 2, pp = lane & 3;
  auto frag = [&](const bf16 (*tile)[P], int mstep, int col0) -> bf16x8 {
    typedef __attribute__((address_space(3))) bf16x4* lptr;
    const bf16* a0 = &tile[mstep + 4 * h + q][col0 + 16 * g + 4 * pp];
    bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lptr)a0);
    bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lptr)(a0 + 8 * P));
    bf16x8 r;
    r[0] = lo[0]; r[1] = lo[1]; r[2] = lo[2]; r[3] = lo[3];
    r[4] = hi[0]; r[5] = hi[1]; r[6] = hi[2]; r[7] = hi[3];
    return r;
  };

  auto compute = [&](int cur) {
#pragma unroll
    for (int ms = 0; ms < BMS; ms += 16) {
      bf16x8 fy[2], fx[2];
#pragma unroll
      for (int a = 0; a < 2; ++a) fy[a] = frag(Ys[cur], ms, wn * 64 + a * 32);
#pragma unroll
      for (int b = 0; b < 2; ++b) fx[b] = frag(Xs[cur], ms, wk * 64 + b * 32);
#pragma unroll
      for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b) acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fy[a], fx[b], acc[a][b], 0, 0, 0);
    }
  };
  const int nsteps = (me - mb + BMS - 1) / BMS;
  // one LDS stage, three to four workgroups per CU: the loads of one workgroup land under the MFMAs of the others
  for (int s = 0; s < nsteps; ++s) {
    fetch(ry0, rx0, mb + s * BMS);
    if (s) __syncthreads();                      // the previous step's fragment reads are done
    commit(ry0, rx0, 0, mb + s * BMS);
    __syncthreads();
    compute(0);
  }
  // partial[chunk][n][k] fp32: lanes 0..31 of a register write 32 consecutive k (128 bytes)
  float* dst = partial + ((size_t)blockIdx.x * N + n0) * K + k0;
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int n = wn * 64 + a * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
        dst[(size_t)n * K + wk * 64 + b * 32 + (lane & 31)] = acc[a][b][r];
      }
}

__global__ __launch_bounds__(kThreads) void wgrad_reduce_kernel(const float* __restrict__ partial, int chunks, size_t total,
                                                               bf16* __restrict__ dW) {
  const size_t i = ((size_t)blockIdx.x * kThreads + threadIdx.x) * 8;
  if (i >= total) return;
  float s[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  for (int c = 0; c < chunks; ++c) {
    const float4 a = *reinterpret_cast<const float4*>(partial + (size_t)c * total + i);
    const float4 b = *reinterpret_cast<const float4*>(partial + (size_t)c * total + i + 4);
    s[0] += a.x; s[1] += a.y; s[2] += a.z; s[3] += a.w; s[4] += b.x; s[5] += b.y; s[6] += b.z; s[7] += b.w;
  }
  Vec<bf16> o;
#pragma unroll
  for (int e = 0; e < 8; ++e) o.set(e, s[e]);
  o.store(dW + i);
}

// [R, C] bf16 -> [C, R] (the weight of the input-gradient product): 32 x 32 tiles through LDS
__global__ __launch_bounds__(kThreads) void transpose_bf16_kernel(const uint16_t* __restrict__ src, int R, int C,
                                                                 uint16_t* __restrict__ dst) {
  __shared__ uint16_t t[32][33];
  const int c0 = blockIdx.x * 32, r0 = blockIdx.y * 32;
  const int x = threadIdx.x & 31, y = threadIdx.x >> 5;
  for (int i = y; i < 32; i += 8)
    if (r0 + i < R && c0 + x < C) t[i][x] = src[(size_t)(r0 + i) * C + c0 + x];
  __syncthreads();
  for (int i = y; i < 32; i += 8)
    if (c0 + i < C && r0 + x < R) dst[(size_t)(c0 + i) * R + r0 + x] = t[x][i];
}

// Batched "weight of the input-gradient convolution": for every layer e of a table,
//   dst_e[ci][kh][kw][co] = src_e[co][KH-1-kh][KW-1-kw][ci]       (both channels-last 4-D weights: [out][kh][kw][in] in memory)
// i.e. w.flip(2, 3).transpose(0, 1) of all stride-1 convolutions in ONE launch (the per-layer flip + copy pairs were ~70
// launches per step).  blocks[b] = {entry, spatial index, co tile, ci tile}; entries[e] = {src offset, dst offset, Co, Ci, KH*KW}
// (offsets in elements into the flat bf16 buffers).
__global__ __launch_bounds__(kThreads) void flip_weights_kernel(const uint16_t* __restrict__ src, uint16_t* __restrict__ dst,
                                                               const int4* __restrict__ blocks,
                                                               const long long* __restrict__ entries) {
  __shared__ uint16_t t[32][33];
  const int4 bk = blocks[blockIdx.x];
  const long long* e = entries + (size_t)bk.x * 5;
  const long long so = e[0], dofs = e[1];
  const int Co = (int)e[2], Ci = (int)e[3], S = (int)e[4];
  const int sp = bk.y, co0 = bk.z * 32, ci0 = bk.w * 32;
  const int x = threadIdx.x & 31, y = threadIdx.x >> 5;
  const uint16_t* sp_src = src + so + (size_t)(S - 1 - sp) * Ci;       // flipped spatial tap
  for (int i = y; i < 32; i += 8)
    if (co0 + i < Co && ci0 + x < Ci) t[i][x] = sp_src[(size_t)(co0 + i) * S * Ci + ci0 + x];
  __syncthreads();
  uint16_t* sp_dst = dst + dofs + (size_t)sp * Co;
  for (int i = y; i < 32; i += 8)
    if (ci0 + i < Ci && co0 + x < Co) sp_dst[(size_t)(ci0 + i) * S * Co + co0 + x] = t[x][i];
}

// The same on 64 x 64 tiles with 16-byte accesses on both sides (round 5): the 32 x 32 form moves 64-byte runs with 2-byte
// accesses - 160 MB of weights per step at 1.3 TB/s (122 us).  blocks[b] = {entry, spatial index, co tile of 64, ci tile of 64}.
__global__ __launch_bounds__(kThreads) void flip_weights64_kernel(const uint16_t* __restrict__ src, uint16_t* __restrict__ dst,
                                                                 const int4* __restrict__ blocks,
                                                                 const long long* __restrict__ entries) {
  __shared__ __attribute__((aligned(16))) uint16_t t[64][72];
  const int4 bk = blocks[blockIdx.x];
  const long long* e = entries + (size_t)bk.x * 5;
  const long long so = e[0], dofs = e[1];
  const int Co = (int)e[2], Ci = (int)e[3], S = (int)e[4];
  const int sp = bk.y, co0 = bk.z * 64, ci0 = bk.w * 64;
  const int g = threadIdx.x & 7, r = threadIdx.x >> 3;
  const uint16_t* sp_src = src + so + (size_t)(S - 1 - sp) * Ci;       // flipped spatial tap
#pragma unroll
  for (int pass = 0; pass < 2; ++pass) {
    const int co = co0 + pass * 32 + r, ci = ci0 + g * 8;
    uint4 v = make_uint4(0u, 0u, 0u, 0u);
    if (co < Co && ci < Ci) {
      const uint16_t* q = sp_src + (size_t)co * S * Ci + ci;
      if (ci + 8 <= Ci && (reinterpret_cast<uintptr_t>(q) & 15) == 0) {
        v = *reinterpret_cast<const uint4*>(q);
      } else {
        uint16_t tmp[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) tmp[j] = ci + j < Ci ? q[j] : (uint16_t)0;
        v = make_uint4(tmp[0] | ((unsigned)tmp[1] << 16), tmp[2] | ((unsigned)tmp[3] << 16), tmp[4] | ((unsigned)tmp[5] << 16),
                       tmp[6] | ((unsigned)tmp[7] << 16));
      }
    }
    *reinterpret_cast<uint4*>(&t[pass * 32 + r][g * 8]) = v;
  }
  __syncthreads();
  uint16_t* sp_dst = dst + dofs + (size_t)sp * Co;
#pragma unroll
  for (int pass = 0; pass < 2; ++pass) {
    const int ci = ci0 + pass * 32 + r, co = co0 + g * 8;
    if (ci >= Ci || co >= Co) continue;
    uint16_t tmp[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) tmp[j] = t[g * 8 + j][pass * 32 + r];
    uint16_t* q = sp_dst + (size_t)ci * S * Co + co;
    if (co + 8 <= Co && (reinterpret_cast<uintptr_t>(q) & 15) == 0) {
      *reinterpret_cast<uint4*>(q) = make_uint4(tmp[0] | ((unsigned)tmp[1] << 16), tmp[2] | ((unsigned)tmp[3] << 16),
                                                tmp[4] | ((unsigned)tmp[5] << 16), tmp[6] | ((unsigned)tmp[7] << 16));
    } else {
#pragma unroll
      for (int j = 0; j < 8; ++j)
        if (co + j < Co) q[j] = tmp[j];
    }
  }
}

int pick_wgrad_chunks(int M, int N, int K) {
  // enough workgroups for two per CU, chunks of whole 32-row steps
  const int tiles = (N / 128) * (K / 128);
  int chunks = (512 + tiles - 1) / tiles;
  const int max_chunks = (M + 255) / 256;
  if (chunks > max_chunks) chunks = max_chunks;
  if (chunks < 1) chunks = 1;
  return chunks;
}


// Launch of the DB form with one of its three pipelines (see the kernel's header): 0 = <64, 2>, 1 = <32, 4>, 2 = <64, 4>.
template <int BN, int OUT, bool CONV3>
int launch_db(int pipe, int grid, hipStream_t s, const Args& a, const char* fn) {
  (void)fn;
  constexpr size_t kOut = (size_t)64 * (BN + 4) * 4, kRed = (size_t)(kThreads / (BN / 8)) * 2 * BN * 4;
  auto lds_of = [&](int bk, int nst) {
    size_t l = (size_t)(kBM + BN) * bk * 2 * nst;
    if (l < kOut) l = kOut;
    if (l < kRed) l = kRed;
    return l;
  };
  if (pipe == 5) {                  // loader waves on 256-row workgroup tiles (16 waves, one workgroup per CU): OUT 0 .. 2
    if constexpr (OUT <= 2 && BN == 128) {
      Args b = a;
      b.tiles_m = ceil_div(a.M, 256);
      const int grid256 = ceil_div(b.tiles_m, 8) * 8 * b.tiles_n;
      size_t lds = (size_t)(256 + BN) * 64 * 2 * 3;
      if (lds < kOut) lds = kOut;
      UCD_TRY_LDS((conv_lw_kernel<256, BN, OUT, CONV3, 64, 3>), (int)lds);
      conv_lw_kernel<256, BN, OUT, CONV3, 64, 3><<<grid256, 1024, lds, s>>>(b);
      return 0;
    }
    pipe = 0;
  }
  if (pipe == 6) {                  // loader waves, TWO 32 KB stages, two workgroups (16 waves) per CU: OUT 0 .. 2 (128 registers)
    if constexpr (OUT <= 2) {
      conv_lw_kernel<128, BN, OUT, CONV3, 64, 2><<<grid, 2 * kThreads, lds_of(64, 2), s>>>(a);
      return 0;
    }
    pipe = 0;
  }
  if (pipe == 7) {                  // round 6: loader waves on 64-row tiles (2 MFMA + 4 loader waves), grids of <= 128 tiles of 128 x 64
    if constexpr (BN == 64) {
      Args b = a;
      b.tiles_m = ceil_div(a.M, 64);
      const int grid64 = ceil_div(b.tiles_m, 8) * 8 * b.tiles_n;
      const size_t lds64 = (size_t)(64 + BN) * 64 * 2 * 3;
      conv_lw_kernel<64, BN, OUT, CONV3, 64, 3><<<grid64, 384, lds64, s>>>(b);
      return 0;
    }
    pipe = 4;
  }
  if (pipe == 4) {                  // loader waves, three 32 KB stages, one workgroup (8 waves) per CU: every epilogue fits
    const size_t lds = lds_of(64, 3);
    UCD_TRY_LDS((conv_lw_kernel<128, BN, OUT, CONV3, 64, 3>), (int)lds);
    conv_lw_kernel<128, BN, OUT, CONV3, 64, 3><<<grid, 2 * kThreads, lds, s>>>(a);
    return 0;
  }
  if (pipe == 3) {
    if constexpr (OUT <= 2) {       // two workgroups per CU: the epilogues of OUT 3 / 4 need more registers than 16 waves leave
      conv_lw_kernel<128, BN, OUT, CONV3, 32, 4><<<grid, 2 * kThreads, lds_of(32, 4), s>>>(a);
      return 0;
    }
    pipe = 0;
  }
  if (pipe == 1) {
    conv1x1_kernel<BN, false, OUT, CONV3, true, 32, 4><<<grid, kThreads, lds_of(32, 4), s>>>(a);
  } else if (pipe == 2) {
    const size_t lds = lds_of(64, 4);
    UCD_TRY_LDS((conv1x1_kernel<BN, false, OUT, CONV3, true, 64, 4>), (int)lds);
    conv1x1_kernel<BN, false, OUT, CONV3, true, 64, 4><<<grid, kThreads, lds, s>>>(a);
  } else {
    conv1x1_kernel<BN, false, OUT, CONV3, true, 64, 2><<<grid, kThreads, lds_of(64, 2), s>>>(a);
  }
  return 0;
}

template <int BN, bool CONV3>
int launch_db_out(int out_mode, int pipe, int grid, hipStream_t s, const Args& a, const char* fn) {
  switch (out_mode) {
    case 0: return launch_db<BN, 0, CONV3>(pipe, grid, s, a, fn);
    case 1: return launch_db<BN, 1, CONV3>(pipe, grid, s, a, fn);
    case 2: return launch_db<BN, 2, CONV3>(pipe, grid, s, a, fn);
    case 3: return launch_db<BN, 3, CONV3>(pipe, grid, s, a, fn);
    default:
      if constexpr (CONV3) {
        set_error("%s: out_mode 4 belongs to the 1x1 products", fn);
        return UCD_EINVAL;
      } else {
        return launch_db<BN, 4, false>(pipe, grid, s, a, fn);
      }
  }
}

// Pipeline of a DB launch: UCD_CONV_PIPE = 2x64 | 4x32 | 4x64 overrides (probes / A-B); else by the grid (measured:
// tools/conv3x3_probe.py, tools/conv1x1_probe.py, profiles/r04_conv_pipe_probe.txt).
int pick_pipe(int M, int tiles_n, int BN, int nk64, int out_mode) {
  static int forced = -2;
  if (forced == -2) {
    const char* e = getenv("UCD_CONV_PIPE");
    forced = !e ? -1 : !strcmp(e, "2x64") ? 0 : !strcmp(e, "4x32") ? 1 : !strcmp(e, "4x64") ? 2 : !strcmp(e, "lw32") ? 3 :
             !strcmp(e, "lw64") ? 4 : !strcmp(e, "lw256") ? 5 : !strcmp(e, "lw64x2") ? 6 : -1;
  }
  if (forced >= 0) return forced;
  const long long wg128 = (long long)ceil_div(M, 128) * tiles_n, wg256 = (long long)ceil_div(M, 256) * tiles_n;
  // grids that give a CU at most one workgroup (3 - 6 images per GPU, the multi-GPU split): the loader-wave form - 3x3 256 -> 256 at
  // 3 images 26.8 -> 17.3 us, 512 -> 512 49 -> 30, the ASPP branches 169 -> 100, 1x1 1024 -> 256 12.2 -> 9.5 (profiles/r04_conv_pipe_probe.txt)
  if (wg128 <= 256 && nk64 >= 4) return 4;
  // 257 .. 640 128-row tiles that fit the chip as 256-row tiles (N = 256 at 33 x 33 and 24 images: 410 -> 206 workgroups): the
  // loader-wave form on 256-row tiles - 3x3 256 -> 256 + statistics 44.1 -> 38.0 us, the ASPP branches 238 -> 214, 1x1 2048 -> 256
  // 33.4 -> 31.6; two rounds of it (512 -> 512: 412 workgroups) measured slower (139 vs 132), the four-stage forms slower as well
  if (BN == 128 && out_mode <= 2 && wg256 <= 256 && nk64 >= 4) return 5;
  return 0;
}

}  // namespace
}  // namespace ucd

using namespace ucd;

extern "C" {

int ucd_conv1x1_row_tiles(int M) { return ceil_div(M, kBM); }

int ucd_conv1x1_stat_replicas(int M) {
  // adds to ONE address are serialised at the memory side (~25 ns each: 205 row tiles into one [2 N] accumulator cost the launch
  // +5 us, the 3121 tiles of the 129 x 129 layers +40 us - profiles/r05_step_kernel_summary_atomic_norep.txt): at most 64 per address
  // (every replica is also 2 N more floats that each workgroup of the finalising apply pass reads)
  const int tiles = ceil_div(M, kBM);
  int r = 1;
  while (r < 64 && tiles > 64 * r) r *= 2;
  return r;
}

int ucd_conv1x1(const ucd_conv1x1_desc* d, ucd_stream_t stream) {
  static const char* fn = "ucd_conv1x1";
  UCD_REQUIRE(d && d->a && d->w && d->y, UCD_EINVAL, "%s: NULL operand", fn);
  UCD_REQUIRE(d->M > 0 && d->N > 0 && d->K > 0, UCD_EINVAL, "%s: empty product", fn);
  UCD_REQUIRE(d->K % kBK == 0 && d->N % 64 == 0, UCD_EUNSUPPORTED, "%s: K (%d) and N (%d) must be multiples of 64", fn, d->K, d->N);
  const bool conv3 = d->taps == 9;
  UCD_REQUIRE(d->taps == 0 || d->taps == 1 || conv3, UCD_EINVAL, "%s: taps must be 1 (1x1) or 9 (3x3)", fn);
  const int stride = d->stride > 1 ? d->stride : 1;
  const bool mapped = conv3 || stride > 1;                 // rows are pixels of a [B, H, W, K] map
  const int oH = mapped && d->H > 0 ? (d->H - 1) / stride + 1 : 0, oW = mapped && d->W > 0 ? (d->W - 1) / stride + 1 : 0;
  UCD_REQUIRE(!mapped || (d->H > 0 && d->W > 0 && (long long)d->M % ((long long)oH * oW) == 0 && !d->in_scale), UCD_EINVAL,
              "%s: the 3x3 / strided modes need H, W, M = B*OH*OW (OH = (H - 1) / stride + 1) and no input transform", fn);
  UCD_REQUIRE(!conv3 || (d->dilation >= 1 && d->ldw >= 9 * d->K), UCD_EINVAL, "%s: 3x3 mode needs a dilation and ldw >= 9 K", fn);
  UCD_REQUIRE(stride == 1 || d->out_mode <= 2, UCD_EUNSUPPORTED, "%s: a strided product takes out_mode 0, 1 or 2", fn);
  const long long a_rows = mapped ? (long long)(d->M / (oH * oW)) * d->H * d->W : d->M;
  UCD_REQUIRE((size_t)a_rows * d->lda * 2 < 0x7FFFFFF0u && (size_t)d->N * d->ldw * 2 < 0x7FFFFFF0u, UCD_EUNSUPPORTED,
              "%s: an operand beyond 2 GiB exceeds the 32-bit offsets of the staging loads", fn);
  UCD_REQUIRE(aligned16(d->a) && aligned16(d->w) && aligned16(d->y) && d->lda % 8 == 0 && d->ldw % 8 == 0 && d->ldy % 8 == 0 &&
                  d->lda >= d->K && d->ldw >= d->K && d->ldy >= d->N,
              UCD_EALIGN, "%s: operands must be 16-byte aligned with leading dimensions that are multiples of 8", fn);
  UCD_REQUIRE(d->out_mode >= 0 && d->out_mode <= 4, UCD_EINVAL, "%s: unknown out_mode %d", fn, d->out_mode);
  UCD_REQUIRE(d->out_mode < 2 ? true : d->partial != nullptr, UCD_EINVAL, "%s: partial is NULL", fn);
  UCD_REQUIRE(d->out_mode != 4 || (d->residual && d->side2 && d->out_mean && d->out_invstd && !conv3 && !d->in_scale &&
                                   aligned16(d->side2) && d->ld2 % 8 == 0 && d->ld2 >= d->N),
              UCD_EINVAL, "%s: out_mode 4 (1x1 only) needs the block output (residual), its conv output (side2), out_mean and out_invstd", fn);
  UCD_REQUIRE((d->out_mode != 1 && d->out_mode != 3) || (d->out_mean && d->out_scale && d->out_shift), UCD_EINVAL,
              "%s: out_mode %d needs out_mean, out_scale and out_shift", fn, d->out_mode);
  UCD_REQUIRE(d->out_mode != 3 || (d->residual && d->out_invstd), UCD_EINVAL, "%s: out_mode 3 needs x (residual) and invstd", fn);
  UCD_REQUIRE((d->in_act & UCD_ACT_MASK) != UCD_ACT_ELU && (d->out_act & UCD_ACT_MASK) != UCD_ACT_ELU, UCD_EUNSUPPORTED,
              "%s: the fused transforms take leaky_relu / identity (elu layers use the separate ABN kernels)", fn);
  UCD_REQUIRE(!d->residual || (aligned16(d->residual) && d->ldr % 8 == 0 && d->ldr >= d->N), UCD_EALIGN,
              "%s: residual must be 16-byte aligned", fn);
  UCD_REQUIRE(!d->in_scale || d->in_mean, UCD_EINVAL, "%s: in_scale without in_mean", fn);
  // the epilogues fetch the per-channel constants of a column chunk as two 16-byte loads and store partials as 8-byte pairs
  UCD_REQUIRE(d->out_mode == 0 || ((!d->out_mean || aligned16(d->out_mean)) && (!d->out_scale || aligned16(d->out_scale)) &&
                                   (!d->out_shift || aligned16(d->out_shift)) && (!d->out_invstd || aligned16(d->out_invstd)) &&
                                   (!d->partial || aligned16(d->partial))),
              UCD_EALIGN, "%s: out_mean / out_scale / out_shift / out_invstd / partial must be 16-byte aligned", fn);
  Args a;
  a.A = (const bf16*)d->a; a.lda = d->lda; a.W = (const bf16*)d->w; a.ldw = d->ldw; a.Y = (bf16*)d->y; a.ldy = d->ldy;
  a.M = d->M; a.N = d->N; a.K = d->K;
  a.in_mean = d->in_mean; a.in_scale = d->in_scale; a.in_shift = d->in_shift;
  a.in_act = d->in_act & UCD_ACT_MASK; a.in_slope = a.in_act == UCD_ACT_IDENTITY ? 1.f : d->in_slope;
  a.out_mean = d->out_mean; a.out_scale = d->out_scale; a.out_shift = d->out_shift; a.out_invstd = d->out_invstd;
  a.R = (const bf16*)d->residual; a.ldr = d->ldr;
  a.Z = (const bf16*)d->side2; a.ldz = d->ld2;
  a.out_act = d->out_act & UCD_ACT_MASK; a.out_slope = a.out_act == UCD_ACT_IDENTITY ? 1.f : d->out_slope;
  a.partial = d->partial; a.accumulate = d->accumulate;
  a.iH = d->H; a.iW = d->W; a.dil = d->dilation;
  a.tiles128 = ceil_div(d->M, 128);
  a.stride = stride; a.oW = oW > 0 ? oW : 1; a.ohw = oH * oW > 0 ? oH * oW : 1; a.a_rows = (int)a_rows;
  a.stat_acc = d->out_mode >= 2 ? d->stat_acc : nullptr; a.stat_shift = d->stat_shift; a.stat_acc2 = d->stat_acc2;
  a.stat_rep = d->stat_rep > 1 ? d->stat_rep : 1;
  UCD_REQUIRE((a.stat_rep & (a.stat_rep - 1)) == 0, UCD_EINVAL, "%s: stat_rep must be a power of two", fn);
  UCD_REQUIRE(!a.stat_acc || d->out_mode != 2 || (d->stat_shift && aligned16(d->stat_shift)), UCD_EINVAL,
              "%s: out_mode 2 with stat_acc needs a 16-byte aligned stat_shift", fn);
  // few row tiles (3 - 6 images per GPU, the multi-GPU split): 64-column tiles double the workgroups of a launch that would leave
  // more than half of the CUs without one - a 52-workgroup 3x3 launch is bound by the 36 x 16 MFMAs of ONE workgroup per CU, not by
  // its fills (a fourth LDS stage changed nothing: 18.4 vs 18.4 us).  3 images: 3x3 256 -> 256 18.6 -> 14.0 us, ASPP 102.8 -> 81.4,
  // 1x1 2048 -> 256 15.5 -> 11.1; grids of 129 .. 256 tiles measured level or slower (256 -> 1024: 6.4 -> 7.2).  Same products bit
  // for bit, per-tile statistics equal to fp32 rounding (tests/test_conv1x1_fused_gpu.py).  UCD_CONV_BN64_TILES overrides (0: off).
  static const int bn64_below = getenv("UCD_CONV_BN64_TILES") ? atoi(getenv("UCD_CONV_BN64_TILES")) : 128;
  const int BN = (d->N % 128 == 0 && (long long)ceil_div(d->M, kBM) * (d->N / 128) > bn64_below) ? 128 : 64;
  a.tiles_m = ceil_div(d->M, kBM); a.tiles_n = d->N / BN;
  const int grid = ceil_div(a.tiles_m, 8) * 8 * a.tiles_n;
  // 3x3 with a grid that leaves a CU one or two workgroups (N = 256 at 33^2: 410; tools/conv3x3_probe.py): the
  // double-buffered form (256->256 50.6 -> 47.3 us, the ASPP branches 338-356 -> 308-314); fuller grids are faster with
  // the single stage and four workgroups per CU (512->512 170 vs 179, 128->128 at 65^2 48.7 vs 52.6)
  // (the 1x1 products of such grids gain too: 1024 -> 256 23.3 -> 21.7 us, with statistics 25.1 -> 23.6)
  // out_mode 4 keeps three side tiles in registers next to the accumulators: 168 VGPRs + 96 B of scratch in the single-stage
  // form at three workgroups per CU, 194 VGPRs and no scratch in the double-buffered form (two per CU) - it always takes that
  hipStream_t s = (hipStream_t)stream;
  const bool db = ((conv3 || !d->in_scale) && (long long)ceil_div(d->M, kBM) * (d->N / BN) <= 640) || (d->out_mode == 4 && BN == 128);
  const size_t lds_main = (size_t)(kBM + BN) * 128 * (db ? 2 : 1), lds_out = (size_t)64 * (BN + 4) * 4;
  size_t lds = lds_main > lds_out ? lds_main : lds_out;
  const size_t lds_red = (size_t)(kThreads / (BN / 8)) * 2 * BN * 4;       // statistics reduction scratch
  if (lds_red > lds) lds = lds_red;
  const bool pro = d->in_scale != nullptr;
  a.param_off = (int)align_up(lds, 16);
  if (pro) lds = a.param_off + (size_t)3 * d->K * sizeof(float);
  UCD_REQUIRE(lds <= 64 * 1024, UCD_EUNSUPPORTED, "%s: K = %d is too wide for the fused input transform", fn, d->K);
  int pipe = db ? pick_pipe(d->M, a.tiles_n, BN, (conv3 ? 9 : 1) * (d->K / 64), d->out_mode) : 0;
  // grids of <= UCD_CONV_LW64_TILES (128) tiles of 128 x 64 (3 images per GPU): 64-row tiles on twice as many CUs (conv_lw_kernel,
  // BM = 64) - statistics / link sums only through the atomic accumulators (64-row tiles have no rows in the per-tile partial buffers)
  static const int lw64_below = getenv("UCD_CONV_LW64_TILES") ? atoi(getenv("UCD_CONV_LW64_TILES")) : 128;
  if (db && !pro && pipe == 4 && BN == 64 && (long long)a.tiles_m * a.tiles_n <= lw64_below && (d->out_mode < 2 || a.stat_acc)) pipe = 7;
  if (db && !pro) {
    int rc;
    if (conv3) rc = BN == 128 ? launch_db_out<128, true>(d->out_mode, pipe, grid, s, a, fn) : launch_db_out<64, true>(d->out_mode, pipe, grid, s, a, fn);
    else rc = BN == 128 ? launch_db_out<128, false>(d->out_mode, pipe, grid, s, a, fn) : launch_db_out<64, false>(d->out_mode, pipe, grid, s, a, fn);
    if (rc) return rc;
    return check_launch(fn);
  }
#define UCD_C1_OUT(BNV, PROV)                                                   \
  switch (d->out_mode) {                                                        \
    case 0: conv1x1_kernel<BNV, PROV, 0><<<grid, kThreads, lds, s>>>(a); break; \
    case 1: conv1x1_kernel<BNV, PROV, 1><<<grid, kThreads, lds, s>>>(a); break; \
    case 2: conv1x1_kernel<BNV, PROV, 2><<<grid, kThreads, lds, s>>>(a); break; \
    case 3: conv1x1_kernel<BNV, PROV, 3><<<grid, kThreads, lds, s>>>(a); break; \
    default: conv1x1_kernel<BNV, false, 4><<<grid, kThreads, lds, s>>>(a); break; \
  }
#define UCD_C3_OUT(BNV)                                                                        \
  switch (d->out_mode) {                                                                       \
    case 0: conv1x1_kernel<BNV, false, 0, true, false><<<grid, kThreads, lds, s>>>(a); break;  \
    case 1: conv1x1_kernel<BNV, false, 1, true, false><<<grid, kThreads, lds, s>>>(a); break;  \
    case 2: conv1x1_kernel<BNV, false, 2, true, false><<<grid, kThreads, lds, s>>>(a); break;  \
    case 4: UCD_REQUIRE(false, UCD_EINVAL, "%s: out_mode 4 belongs to the 1x1 products", fn);  \
    default: conv1x1_kernel<BNV, false, 3, true, false><<<grid, kThreads, lds, s>>>(a); break; \
  }
  if (conv3) {
    if (BN == 128) { UCD_C3_OUT(128) } else { UCD_C3_OUT(64) }
  } else if (BN == 128) {
    if (pro) { UCD_C1_OUT(128, true) } else { UCD_C1_OUT(128, false) }
  } else {
    if (pro) { UCD_C1_OUT(64, true) } else { UCD_C1_OUT(64, false) }
  }
#undef UCD_C3_OUT
#undef UCD_C1_OUT
  return check_launch(fn);
}

size_t ucd_conv1x1_stats_partial_bytes(int M, int C) { return (size_t)ceil_div(M, kBM) * 3 * C * sizeof(float); }

int ucd_conv1x1_stats_finalize(const float* partial, int M, int C, const float* weight, float* running_mean,
                               float* running_var, float momentum, float eps, float* buf, float* pack, int flags,
                               ucd_stream_t stream) {
  static const char* fn = "ucd_conv1x1_stats_finalize";
  UCD_REQUIRE(partial && buf && M > 0 && C > 0, UCD_EINVAL, "%s: bad arguments", fn);
  const int tiles = ceil_div(M, kBM);
  float *sums = buf, *kshift = buf + 2 * C, *mean = buf + 3 * C, *invstd = buf + 4 * C, *scale = buf + 5 * C;
  FinalizeArgs fin{kshift, weight, running_mean, running_var, mean, invstd, scale, (float)M, momentum, eps, pack,
                   (flags & UCD_NORM_ABS_GAMMA) != 0};
  hipStream_t s = (hipStream_t)stream;
  if (pack)
    tile_stats_reduce_kernel<2><<<ceil_div(C, 8), kStatThreads, 0, s>>>(partial, tiles, M, C, sums, kshift, fin);
  else
    tile_stats_reduce_kernel<1><<<ceil_div(C, 8), kStatThreads, 0, s>>>(partial, tiles, M, C, sums, kshift, fin);
  return check_launch(fn);
}

size_t ucd_conv1x1_wgrad_workspace_bytes(int M, int N, int K) {
  if (N % 128 || K % 128) return 0;
  return (size_t)pick_wgrad_chunks(M, N, K) * N * K * sizeof(float);
}

int ucd_conv1x1_wgrad(const void* dy, int ld_dy, const void* a, int lda, int M, int N, int K, const float* in_mean,
                      const float* in_scale, const float* in_shift, int in_act, float in_slope, void* dw,
                      void* workspace, size_t workspace_bytes, ucd_stream_t stream) {
  static const char* fn = "ucd_conv1x1_wgrad";
  UCD_REQUIRE(dy && a && dw && workspace, UCD_EINVAL, "%s: NULL argument", fn);
  UCD_REQUIRE(M > 0 && N % 128 == 0 && K % 128 == 0, UCD_EUNSUPPORTED, "%s: N (%d) and K (%d) must be multiples of 128", fn, N, K);
  UCD_REQUIRE(aligned16(dy) && aligned16(a) && aligned16(dw) && ld_dy % 8 == 0 && lda % 8 == 0, UCD_EALIGN,
              "%s: operands must be 16-byte aligned", fn);
  const int chunks = pick_wgrad_chunks(M, N, K);
  UCD_REQUIRE(workspace_bytes >= (size_t)chunks * N * K * 4, UCD_EWORKSPACE, "%s: workspace too small", fn);
  int rows = ceil_div(M, chunks);
  rows = ceil_div(rows, 32) * 32;
  const int grid_x = ceil_div(M, rows);
  hipStream_t s = (hipStream_t)stream;
  const int act = in_act & UCD_ACT_MASK;
  UCD_REQUIRE(act != UCD_ACT_ELU, UCD_EUNSUPPORTED, "%s: the fused input transform takes leaky_relu / identity", fn);
  const float slope = act == UCD_ACT_IDENTITY ? 1.f : in_slope;
  dim3 grid(grid_x, N / 128, K / 128);
  if (in_scale)
    conv1x1_wgrad_kernel<true><<<grid, kThreads, 0, s>>>((const bf16*)dy, ld_dy, (const bf16*)a, lda, M, N, K, in_mean, in_scale,
                                                         in_shift, act, slope, rows, (float*)workspace);
  else
    conv1x1_wgrad_kernel<false><<<grid, kThreads, 0, s>>>((const bf16*)dy, ld_dy, (const bf16*)a, lda, M, N, K, nullptr, nullptr,
                                                          nullptr, act, slope, rows, (float*)workspace);
  int rc = check_launch(fn);
  if (rc) return rc;
  const size_t total = (size_t)N * K;
  wgrad_reduce_kernel<<<(unsigned)((total / 8 + kThreads - 1) / kThreads), kThreads, 0, s>>>((const float*)workspace, grid_x, total,
                                                                                         (bf16*)dw);
  return check_launch(fn);
}

int ucd_flip_weights_batched64(const void* src_flat, void* dst_flat, const int* blocks, int n_blocks, const long long* entries,
                               ucd_stream_t stream) {
  static const char* fn = "ucd_flip_weights_batched64";
  UCD_REQUIRE(src_flat && dst_flat && blocks && entries && n_blocks > 0, UCD_EINVAL, "%s: bad arguments", fn);
  flip_weights64_kernel<<<n_blocks, kThreads, 0, (hipStream_t)stream>>>((const uint16_t*)src_flat, (uint16_t*)dst_flat,
                                                                        (const int4*)blocks, entries);
  return check_launch(fn);
}

int ucd_flip_weights_batched(const void* src_flat, void* dst_flat, const int* blocks, int n_blocks, const long long* entries,
                             ucd_stream_t stream) {
  static const char* fn = "ucd_flip_weights_batched";
  UCD_REQUIRE(src_flat && dst_flat && blocks && entries && n_blocks > 0, UCD_EINVAL, "%s: bad arguments", fn);
  flip_weights_kernel<<<n_blocks, kThreads, 0, (hipStream_t)stream>>>((const uint16_t*)src_flat, (uint16_t*)dst_flat,
                                                                      (const int4*)blocks, entries);
  return check_launch(fn);
}

int ucd_transpose_bf16(const void* src, int rows, int cols, void* dst, ucd_stream_t stream) {
  static const char* fn = "ucd_transpose_bf16";
  UCD_REQUIRE(src && dst && rows > 0 && cols > 0, UCD_EINVAL, "%s: bad arguments", fn);
  transpose_bf16_kernel<<<dim3(ceil_div(cols, 32), ceil_div(rows, 32)), kThreads, 0, (hipStream_t)stream>>>(
      (const uint16_t*)src, rows, cols, (uint16_t*)dst);
  return check_launch(fn);
}

}  // extern "C"
