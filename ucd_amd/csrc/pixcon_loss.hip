// PixelConLossV2 (utils/loss.py:412-466) as a streaming MFMA kernel: no [A, C] matrix is ever
// materialised (the reference builds ~20 of them in float32), and the gradient w.r.t. the
// anchors comes out of the same two sweeps.
//
// Math (i anchors, j contrast rows, T temperature; la/lc labels; the contrast matrix starts with the
// anchors themselves, so j == i is the self pair):
//   S_ij = a_i.c_j / T      E_ij = exp(S_ij) [la_i != lc_j]         neg_i = sum_j E_ij   (un-shifted, :449)
//   m_i = max_j S_ij        S'_ij = S_ij - m_i                        D_ij = exp(S'_ij) + neg_i
//   loss = (1/R) sum_i -(1/num_i) sum_j pos_ij P_ij (S'_ij - log D_ij)                         (:461-466)
//   d loss / d a_i = 1/(T num_i R) * ( (sum_j q_ij / neg_i) * U_i  -  V_i ),
//       q_ij = pos_ij P_ij neg_i / D_ij,   U_i = sum_j E_ij c_j,   V_i = sum_j q_ij c_j
// so a row needs two GEMM pairs of the flash-attention shape (scores, then scores x values, with the
// contrast rows serving as both keys and values):
//   sweep 1 (all tiles):             S -> E -> neg_i, m_i, U_i        4N flop per pair = algorithmic minimum
//   sweep 2 (tiles holding positives): S, P -> q -> loss_i, sum q, V_i
// With rows grouped by label (ucd_pixcon_prep sort_by_label) positives live in two contiguous row
// ranges per anchor tile, so sweep 2 touches only the same-class blocks.
//
// This file is the float32 path: v_mfma_f32_32x32x2_f32 (exact fp32 products and accumulation, the
// parity mode required by the 1e-3 contract; S = cos/0.07 amplifies operand rounding 14x inside exp).
//
// Tiling: a workgroup = 4 waves = 128 anchors; each wave owns 32 anchors and keeps them in registers
// (128 VGPRs) as the B operand of S^T = C_j . A_i^T, so every lane holds ONE anchor (column) and 16
// contrast rows of the 32x32 tile: all per-anchor reductions (neg, max, loss, sum q) are in-lane.  The
// S^T accumulator, turned into E or q in place, is then the B operand of U^T += C_j^T . E with no data
// movement (register r covers contrast rows (r&3)+8(r>>2) and +4 on the upper half-wave, which is the
// k-pair of one 32x32x2 step).  Contrast tiles (32 rows x 256) are staged through LDS once per
// workgroup, double buffered, row pitch 260 floats (conflict-free ds_read_b128 / ds_read_b32).
#include "common.h"
#include "pixcon.h"

namespace ucd {
namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int kThreads = 256;
constexpr int kWaves = 4;
constexpr int kTI = 32;             // anchors per wave
constexpr int kBI = kWaves * kTI;   // anchors per workgroup
constexpr int kTJ = 32;             // contrast rows per tile
constexpr int kN = 256;             // padded feature dimension
constexpr int kPitch = kN + 4;      // LDS row pitch (floats)
constexpr int kMaxSplit = 16;

__device__ __forceinline__ int tile_row(int reg, int half) { return (reg & 3) + 8 * (reg >> 2) + 4 * half; }

struct TileList {  // two ranges of contrast tiles, addressed as one virtual list
  int t1a, n1, t2a, n2;
  __device__ __forceinline__ int count() const { return n1 + n2; }
  __device__ __forceinline__ int at(int v) const { return v < n1 ? t1a + v : t2a + (v - n1); }
};

// Registers <- 32 anchors of this wave: lane (i = lane&31, h = lane>>5) holds a_i[h*128 .. h*128+127].
__device__ __forceinline__ void load_anchor_frags(float (&areg)[128], const float* __restrict__ chat, int ldc, int row,
                                                  bool row_ok, int half) {
  if (row_ok) {
    const float4* src = reinterpret_cast<const float4*>(chat + (size_t)row * ldc + half * 128);
#pragma unroll
    for (int q = 0; q < 32; ++q) {
      float4 v = src[q];
      areg[4 * q + 0] = v.x; areg[4 * q + 1] = v.y; areg[4 * q + 2] = v.z; areg[4 * q + 3] = v.w;
    }
  } else {
#pragma unroll
    for (int q = 0; q < 128; ++q) areg[q] = 0.f;
  }
}

// 32 x 256 contrast tile: global -> registers (8 x float4 per thread) -> LDS
__device__ __forceinline__ void tile_fetch(float4 (&stage)[8], const float* __restrict__ chat, int ldc, int j0) {
#pragma unroll
  for (int q = 0; q < 8; ++q) {
    const int idx = threadIdx.x + kThreads * q;  // 0..2047 float4
    const int row = idx >> 6, c4 = idx & 63;
    stage[q] = *reinterpret_cast<const float4*>(chat + (size_t)(j0 + row) * ldc + c4 * 4);
  }
}
__device__ __forceinline__ void tile_commit(const float4 (&stage)[8], float* __restrict__ cs) {
#pragma unroll
  for (int q = 0; q < 8; ++q) {
    const int idx = threadIdx.x + kThreads * q;
    const int row = idx >> 6, c4 = idx & 63;
    *reinterpret_cast<float4*>(cs + row * kPitch + c4 * 4) = stage[q];
  }
}

// X[j][i] = sum_n C[j][n] a_i[n]   (S^T tile, k order: step s pairs n = s and n = 128 + s)
__device__ __forceinline__ f32x16 gemm_scores(const float* __restrict__ cs, const float (&areg)[128], int lane) {
  f32x16 acc = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  const float* rowp = cs + (lane & 31) * kPitch + (lane >> 5) * 128;
#pragma unroll
  for (int q = 0; q < 32; ++q) {
    const float4 c = *reinterpret_cast<const float4*>(rowp + 4 * q);
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(c.x, areg[4 * q + 0], acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(c.y, areg[4 * q + 1], acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(c.z, areg[4 * q + 2], acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(c.w, areg[4 * q + 3], acc, 0, 0, 0);
  }
  return acc;
}

// Y[n][i] += sum_j C[j][n] w[j][i]   with w = the (transformed) score tile held as the B operand
__device__ __forceinline__ void gemm_values(f32x16 (&acc)[8], const float* __restrict__ cs, const f32x16& w, int lane) {
  const int half = lane >> 5, col = lane & 31;
#pragma unroll
  for (int nt = 0; nt < 8; ++nt) {
#pragma unroll
    for (int reg = 0; reg < 16; ++reg) {
      const float c = cs[tile_row(reg, half) * kPitch + 32 * nt + col];
      acc[nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(c, w[reg], acc[nt], 0, 0, 0);
    }
  }
}

// store the lane's 128 accumulator values of anchor row `dst` (n = 32 nt + 8 g + 4 half + 0..3)
__device__ __forceinline__ void store_values(const f32x16 (&acc)[8], float* __restrict__ dst, int half) {
#pragma unroll
  for (int nt = 0; nt < 8; ++nt)
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      float4 v = {acc[nt][4 * g + 0], acc[nt][4 * g + 1], acc[nt][4 * g + 2], acc[nt][4 * g + 3]};
      *reinterpret_cast<float4*>(dst + 32 * nt + 8 * g + 4 * half) = v;
    }
}

// ---- sweep 1: negatives ----------------------------------------------------------------------------
__global__ __launch_bounds__(kThreads, 1) void pixcon_neg_kernel(const float* __restrict__ chat, int ldc,
                                                                const uint8_t* __restrict__ row_label,
                                                                const ucd_pixcon_meta* __restrict__ meta, float inv_T,
                                                                int nsplit, int maxA, float* __restrict__ negp,
                                                                float* __restrict__ maxp, float* __restrict__ Up) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* cs0 = smem;                                   // [2][32][kPitch]
  int* labs0 = reinterpret_cast<int*>(smem + 2 * kTJ * kPitch);  // [2][32]
  const int A = meta->A, Cpad = meta->Cpad;
  const int i_base = blockIdx.x * kBI;
  if (i_base >= A) return;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, half = lane >> 5;
  const int i_row = i_base + wave * kTI + (lane & 31);
  const bool row_ok = i_row < A;
  const bool wave_ok = i_base + wave * kTI < A;
  const int la = row_ok ? row_label[i_row] : -1;

  const int ntiles = Cpad / kTJ;
  const int per = (ntiles + nsplit - 1) / nsplit;
  const int v_begin = blockIdx.y * per, v_end = min(ntiles, v_begin + per);

  float areg[128];
  load_anchor_frags(areg, chat, ldc, i_row, row_ok, half);
  f32x16 U[8];
#pragma unroll
  for (int nt = 0; nt < 8; ++nt)
#pragma unroll
    for (int r = 0; r < 16; ++r) U[nt][r] = 0.f;
  float neg = 0.f, mx = -INFINITY;

  float4 stage[8];
  int cur = 0;
  if (v_begin < v_end) {
    tile_fetch(stage, chat, ldc, v_begin * kTJ);
    tile_commit(stage, cs0);
    if (threadIdx.x < kTJ) labs0[threadIdx.x] = row_label[v_begin * kTJ + threadIdx.x];
  }
  __syncthreads();
  for (int v = v_begin; v < v_end; ++v) {
    const bool has_next = v + 1 < v_end;
    if (has_next) tile_fetch(stage, chat, ldc, (v + 1) * kTJ);
    const float* cs = cs0 + cur * kTJ * kPitch;
    const int* labs = labs0 + cur * kTJ;
    if (wave_ok) {
      f32x16 x = gemm_scores(cs, areg, lane);
#pragma unroll
      for (int reg = 0; reg < 16; ++reg) {
        const int lc = labs[tile_row(reg, half)];
        const float s = x[reg] * inv_T;
        const bool valid = lc != kPadLabel;
        const bool negm = valid && lc != la;
        const float e = negm ? __expf(s) : 0.f;
        neg += e;
        mx = valid ? fmaxf(mx, s) : mx;
        x[reg] = e;
      }
      gemm_values(U, cs, x, lane);
    }
    if (has_next) {
      tile_commit(stage, cs0 + (cur ^ 1) * kTJ * kPitch);
      if (threadIdx.x < kTJ) labs0[(cur ^ 1) * kTJ + threadIdx.x] = row_label[(v + 1) * kTJ + threadIdx.x];
    }
    __syncthreads();
    cur ^= 1;
  }
  // the two half-waves hold disjoint contrast rows of the same anchor
  neg += __shfl_xor(neg, 32, 64);
  mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
  if (row_ok) {
    if (half == 0) {
      negp[(size_t)blockIdx.y * maxA + i_row] = neg;
      maxp[(size_t)blockIdx.y * maxA + i_row] = mx;
    }
    store_values(U, Up + ((size_t)blockIdx.y * maxA + i_row) * kN, half);
  }
}

// ---- sweep 2: positives ----------------------------------------------------------------------------
__global__ __launch_bounds__(kThreads, 1) void pixcon_pos_kernel(
    const float* __restrict__ chat, int ldc, const uint8_t* __restrict__ row_label, const float* __restrict__ pcat, int ldp,
    int KP2, const ucd_pixcon_meta* __restrict__ meta, float inv_T, int shift_pos, int use_prob, int nsplit1, int nsplit2,
    int maxA, const float* __restrict__ negp, const float* __restrict__ maxp, float* __restrict__ lossp,
    float* __restrict__ qsump, float* __restrict__ Vp, const float* __restrict__ Pmat, int ldP) {
  // Pmat != NULL (ucd_pixcon_loss_given_p): the weight P_ij is read from a caller-materialised [A, C] matrix in the
  // reference's column order (anchors first, teacher rows after them) instead of being formed from the probabilities
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int ppitch = KP2 + 1;
  float* cs0 = smem;                                             // [2][32][kPitch]
  int* labs0 = reinterpret_cast<int*>(smem + 2 * kTJ * kPitch);  // [2][32]
  float* ps0 = smem + 2 * kTJ * kPitch + 2 * kTJ;                // [2][32][ppitch]  contrast probabilities
  float* pa0 = ps0 + 2 * kTJ * ppitch;                           // [4][32][ppitch]  anchor probabilities
  const int A = meta->A, Apad = meta->Apad, Cpad = meta->Cpad, min_new = meta->min_new;
  const int i_base = blockIdx.x * kBI;
  if (i_base >= A) return;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, half = lane >> 5;
  const int i_row = i_base + wave * kTI + (lane & 31);
  const bool row_ok = i_row < A;
  const bool wave_ok = i_base + wave * kTI < A;
  const int la = row_ok ? row_label[i_row] : -1;
  const bool gt_i = la >= min_new;

  // tiles that can hold positives for the anchors of this workgroup
  TileList tl;
  if (meta->sorted) {
    const int Lmin = row_label[i_base], Lmax = row_label[min(i_base + kBI, A) - 1];
    const int r1a = meta->label_start_a[Lmin], r1b = meta->label_start_a[Lmax + 1];
    const int r2a = Apad + meta->label_start_o[Lmin], r2b = Apad + meta->label_start_o[Lmax + 1];
    tl.t1a = r1a / kTJ; tl.n1 = (r1b + kTJ - 1) / kTJ - tl.t1a;
    tl.t2a = r2a / kTJ; tl.n2 = r2b > r2a ? (r2b + kTJ - 1) / kTJ - tl.t2a : 0;
  } else {
    tl.t1a = 0; tl.n1 = Cpad / kTJ; tl.t2a = 0; tl.n2 = 0;
  }
  const int nv = tl.count();
  const int per = (nv + nsplit2 - 1) / nsplit2;
  const int v_begin = blockIdx.y * per, v_end = min(nv, v_begin + per);

  // row constants from sweep 1
  float neg_i = 0.f, m_i = -INFINITY;
  if (row_ok) {
    for (int s = 0; s < nsplit1; ++s) {
      neg_i += negp[(size_t)s * maxA + i_row];
      m_i = fmaxf(m_i, maxp[(size_t)s * maxA + i_row]);
    }
  }
  if (!shift_pos) m_i = 0.f;

  float areg[128];
  load_anchor_frags(areg, chat, ldc, i_row, row_ok, half);
  if (use_prob) {
    float* pa = pa0 + (wave * kTI + (lane & 31)) * ppitch;
    for (int k = half; k < KP2; k += 2) pa[k] = row_ok ? pcat[(size_t)i_row * ldp + k] : 0.f;
  }
  f32x16 V[8];
#pragma unroll
  for (int nt = 0; nt < 8; ++nt)
#pragma unroll
    for (int r = 0; r < 16; ++r) V[nt][r] = 0.f;
  float lossacc = 0.f, qsum = 0.f;

  auto commit_side = [&](int buf, int j0) {
    if (threadIdx.x < kTJ) labs0[buf * kTJ + threadIdx.x] = row_label[j0 + threadIdx.x];
    if (use_prob)
      for (int idx = threadIdx.x; idx < kTJ * KP2; idx += kThreads) {
        const int row = idx / KP2, k = idx - row * KP2;
        ps0[(buf * kTJ + row) * ppitch + k] = pcat[(size_t)(j0 + row) * ldp + k];
      }
  };

  float4 stage[8];
  int cur = 0;
  if (v_begin < v_end) {
    const int j0 = tl.at(v_begin) * kTJ;
    tile_fetch(stage, chat, ldc, j0);
    tile_commit(stage, cs0);
    commit_side(0, j0);
  }
  __syncthreads();
  for (int v = v_begin; v < v_end; ++v) {
    const bool has_next = v + 1 < v_end;
    const int j0 = tl.at(v) * kTJ;
    const int j0n = has_next ? tl.at(v + 1) * kTJ : 0;
    if (has_next) tile_fetch(stage, chat, ldc, j0n);
    const float* cs = cs0 + cur * kTJ * kPitch;
    const int* labs = labs0 + cur * kTJ;
    if (wave_ok) {
      f32x16 x = gemm_scores(cs, areg, lane);
      f32x16 pm;
      if (use_prob) {
        // P^T tile: pm[j][i] = sum_k pc[j][k] pa[i][k]
#pragma unroll
        for (int r = 0; r < 16; ++r) pm[r] = 0.f;
        const float* pc = ps0 + (cur * kTJ + (lane & 31)) * ppitch + half;
        const float* pa = pa0 + (wave * kTI + (lane & 31)) * ppitch + half;
        for (int k = 0; k < KP2; k += 2) pm = __builtin_amdgcn_mfma_f32_32x32x2f32(pc[k], pa[k], pm, 0, 0, 0);
      }
#pragma unroll
      for (int reg = 0; reg < 16; ++reg) {
        const int jl = tile_row(reg, half);
        const int lc = labs[jl];
        const bool pos = row_ok && lc == la && (j0 + jl) != i_row;  // padding rows carry label 255 != la
        float q = 0.f;
        if (pos) {
          float pw;
          if (Pmat) {
            const int j = j0 + jl;
            pw = Pmat[(size_t)i_row * ldP + (j < Apad ? j : A + (j - Apad))];
          } else {
            pw = (use_prob && !(gt_i && lc >= min_new)) ? pm[reg] : 1.f;
          }
          const float sp = x[reg] * inv_T - m_i;
          const float d = __expf(sp) + neg_i;
          lossacc += pw * (sp - __logf(d));
          q = pw * (neg_i / d);
          qsum += q;
        }
        x[reg] = q;
      }
      gemm_values(V, cs, x, lane);
    }
    if (has_next) {
      tile_commit(stage, cs0 + (cur ^ 1) * kTJ * kPitch);
      commit_side(cur ^ 1, j0n);
    }
    __syncthreads();
    cur ^= 1;
  }
  lossacc += __shfl_xor(lossacc, 32, 64);
  qsum += __shfl_xor(qsum, 32, 64);
  if (row_ok) {
    if (half == 0) {
      lossp[(size_t)blockIdx.y * maxA + i_row] = lossacc;
      qsump[(size_t)blockIdx.y * maxA + i_row] = qsum;
    }
    store_values(V, Vp + ((size_t)blockIdx.y * maxA + i_row) * kN, half);
  }
}

// ---- combine: per-row loss and gradient ---------------------------------------------------------------
// one wave per anchor row
__global__ __launch_bounds__(kThreads) void pixcon_finalize_kernel(
    const uint8_t* __restrict__ row_label, const ucd_pixcon_meta* __restrict__ meta, float inv_T, int nsplit1, int nsplit2,
    int maxA, const float* __restrict__ negp, const float* __restrict__ lossp, const float* __restrict__ qsump,
    const float* __restrict__ Up, const float* __restrict__ Vp, float* __restrict__ grad_a, int ldg, int N,
    float* __restrict__ row_stats, float* __restrict__ row_loss) {
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int i = blockIdx.x * (kThreads / 64) + wave;
  const int A = meta->A;
  if (i >= A) return;
  const int num = meta->label_count_c[row_label[i]] - 1;
  const float R = (float)meta->n_valid;
  float neg = 0.f, la = 0.f, qs = 0.f;
  for (int s = 0; s < nsplit1; ++s) neg += negp[(size_t)s * maxA + i];
  for (int s = 0; s < nsplit2; ++s) {
    la += lossp[(size_t)s * maxA + i];
    qs += qsump[(size_t)s * maxA + i];
  }
  const float coef = num > 0 ? inv_T / ((float)num * R) : 0.f;
  const float ratio = neg > 0.f ? qs / neg : 0.f;
  const float rl = num > 0 ? -la / (float)num : 0.f;
  if (grad_a) {
    for (int c = lane * 4; c < ldg; c += 256) {
      float4 u = {0.f, 0.f, 0.f, 0.f}, vv = {0.f, 0.f, 0.f, 0.f};
      if (c < kN) {
        for (int s = 0; s < nsplit1; ++s) {
          const float4 t = *reinterpret_cast<const float4*>(Up + ((size_t)s * maxA + i) * kN + c);
          u.x += t.x; u.y += t.y; u.z += t.z; u.w += t.w;
        }
        for (int s = 0; s < nsplit2; ++s) {
          const float4 t = *reinterpret_cast<const float4*>(Vp + ((size_t)s * maxA + i) * kN + c);
          vv.x += t.x; vv.y += t.y; vv.z += t.z; vv.w += t.w;
        }
      }
      float4 g = {coef * (ratio * u.x - vv.x), coef * (ratio * u.y - vv.y), coef * (ratio * u.z - vv.z),
                  coef * (ratio * u.w - vv.w)};
      *reinterpret_cast<float4*>(grad_a + (size_t)i * ldg + c) = g;
    }
  }
  if (lane == 0) {
    row_loss[i] = rl;
    if (row_stats) {
      row_stats[i] = neg;
      row_stats[(size_t)maxA + i] = (float)num;
      row_stats[(size_t)2 * maxA + i] = rl;
    }
  }
}

// loss = sum_i row_loss_i / R  (fixed summation order: deterministic)
__global__ __launch_bounds__(1024) void pixcon_reduce_kernel(const float* __restrict__ row_loss,
                                                            const ucd_pixcon_meta* __restrict__ meta,
                                                            float* __restrict__ loss_out) {
  __shared__ float part[16];
  const int A = meta->A;
  float s = 0.f;
  for (int i = threadIdx.x; i < A; i += 1024) s += row_loss[i];
  s = wave_sum(s);
  if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) {
    float t = 0.f;
    for (int i = 0; i < 16; ++i) t += part[i];
    const float R = (float)meta->n_valid;
    loss_out[0] = R > 0.f ? t / R : 0.f;
    loss_out[1] = R;
  }
}

struct Plan {
  int nt_i, nsplit1, nsplit2, KP2;
  size_t off_negp, off_maxp, off_lossp, off_qsump, off_rowloss, off_Up, off_Vp, total;
};

Plan make_plan(int BHW, int K) {
  Plan p;
  p.nt_i = ceil_div(BHW, kBI);
  int ns = ceil_div(1024, p.nt_i);
  if (ns > kMaxSplit) ns = kMaxSplit;
  if (ns < 1) ns = 1;
  p.nsplit1 = ns;
  p.nsplit2 = ns;
  p.KP2 = (K + 1) & ~1;
  size_t o = 0;
  const size_t rowvec = align_up((size_t)BHW * 4, 256);
  p.off_negp = o; o += rowvec * ns;
  p.off_maxp = o; o += rowvec * ns;
  p.off_lossp = o; o += rowvec * ns;
  p.off_qsump = o; o += rowvec * ns;
  p.off_rowloss = o; o += rowvec;
  p.off_Up = o; o += (size_t)ns * BHW * kN * 4;
  p.off_Vp = o; o += (size_t)ns * BHW * kN * 4;
  p.total = o;
  return p;
}

}  // namespace

void pixcon_launch_reduce(const float* row_loss, const ucd_pixcon_meta* meta, float* loss_out, hipStream_t s) {
  pixcon_reduce_kernel<<<1, 1024, 0, s>>>(row_loss, meta, loss_out);
}
}  // namespace ucd

using namespace ucd;

extern "C" {

size_t ucd_pixcon_loss_workspace_bytes(int BHW, int N, int K) {
  (void)N;
  const size_t a = make_plan(BHW, K).total, b = pixcon16_workspace_bytes(BHW), c = pixcon16p_workspace_bytes(BHW);
  return a > b ? (a > c ? a : c) : (b > c ? b : c);
}

static int pixcon_loss_impl(const char* fn, const float* chat, int ldc, int N, const uint8_t* row_label, const float* pcat,
                            int ldp, int K, const void* ch16, const void* p16, int precision, const ucd_pixcon_meta* meta,
                            int BHW, float temperature, int shift_pos, int use_prob, const float* Pmat, int ldP,
                            float* loss_out, float* grad_a, int ldg, float* row_stats, void* workspace,
                            size_t workspace_bytes, ucd_stream_t stream) {
  UCD_REQUIRE(row_label && meta && loss_out && workspace, UCD_EINVAL, "%s: NULL argument", fn);
  UCD_REQUIRE(BHW > 0 && N > 0 && temperature > 0.f, UCD_EINVAL, "%s: bad sizes", fn);
  UCD_REQUIRE(precision == UCD_PIXCON_F32 || precision == UCD_PIXCON_F16 || precision == UCD_PIXCON_F16_SPLIT, UCD_EINVAL,
              "%s: unknown precision %d", fn, precision);
  UCD_REQUIRE(!grad_a || (aligned16(grad_a) && ldg % 4 == 0 && ldg >= N && ldg <= kN), UCD_EALIGN,
              "%s: grad_a must be 16-byte aligned, ldg a multiple of 4 in [N, %d]", fn, kN);
  if (precision == UCD_PIXCON_F16 || precision == UCD_PIXCON_F16_SPLIT) {
    UCD_REQUIRE(ch16 && aligned16(ch16) && N <= kN, UCD_EINVAL, "%s: the fp16 path needs ch16 [Cpad, %d]", fn, kN);
    UCD_REQUIRE(!use_prob || (p16 && aligned16(p16) && K > 0 && K <= 112), UCD_EUNSUPPORTED,
                "%s: the fp16 path needs p16 and K <= 112", fn);
    if (precision == UCD_PIXCON_F16 && pixcon16p_eligible(BHW, temperature, use_prob, K))
      return pixcon16p_launch((const _Float16*)ch16, row_label, (const _Float16*)p16, K, meta, BHW, temperature, shift_pos,
                              use_prob, loss_out, grad_a, ldg, row_stats, workspace, workspace_bytes, (hipStream_t)stream);
    return pixcon16_launch((const _Float16*)ch16, row_label, (const _Float16*)p16, K, meta, BHW, temperature, shift_pos,
                           use_prob, loss_out, grad_a, ldg, row_stats, workspace, workspace_bytes, (hipStream_t)stream);
  }
  UCD_REQUIRE(chat, UCD_EINVAL, "%s: chat is NULL", fn);
  UCD_REQUIRE(ldc == kN && N <= kN, UCD_EUNSUPPORTED, "%s: the contrast matrix must be padded to ldc == %d columns (N <= %d)", fn, kN, kN);
  UCD_REQUIRE(aligned16(chat) && (!grad_a || (aligned16(grad_a) && ldg % 4 == 0 && ldg >= N)), UCD_EALIGN,
              "%s: chat / grad_a must be 16-byte aligned, ldg a multiple of 4", fn);
  UCD_REQUIRE(!use_prob || (pcat && K > 0 && ldp >= ((K + 1) & ~1)), UCD_EINVAL,
              "%s: use_prob needs pcat with ldp >= K rounded up to even", fn);
  UCD_REQUIRE(K <= 110, UCD_EUNSUPPORTED, "%s: K = %d teacher classes exceed the LDS budget (K <= 110)", fn, K);
  const Plan p = make_plan(BHW, K);
  UCD_REQUIRE(workspace_bytes >= p.total, UCD_EWORKSPACE, "%s: workspace too small (%zu < %zu)", fn, workspace_bytes, p.total);
  hipStream_t s = (hipStream_t)stream;
  char* ws = (char*)workspace;
  float* negp = (float*)(ws + p.off_negp);
  float* maxp = (float*)(ws + p.off_maxp);
  float* lossp = (float*)(ws + p.off_lossp);
  float* qsump = (float*)(ws + p.off_qsump);
  float* rowloss = (float*)(ws + p.off_rowloss);
  float* Up = (float*)(ws + p.off_Up);
  float* Vp = (float*)(ws + p.off_Vp);
  const float inv_T = 1.f / temperature;
  const int maxA = BHW;

  const size_t lds1 = (size_t)(2 * kTJ * kPitch + 2 * kTJ) * 4;
  // opt in to more than 64 KiB of dynamic LDS (gfx950 has 160 KiB per workgroup); per call, no process-wide state
  UCD_TRY_LDS(pixcon_neg_kernel, 160 * 1024);
  UCD_TRY_LDS(pixcon_pos_kernel, 160 * 1024);
  pixcon_neg_kernel<<<dim3(p.nt_i, p.nsplit1), kThreads, lds1, s>>>(chat, ldc, row_label, meta, inv_T, p.nsplit1, maxA,
                                                                    negp, maxp, Up);
  int rc = check_launch(fn);
  if (rc) return rc;
  const int KP2 = use_prob ? p.KP2 : 0;
  const size_t lds2 = lds1 + (size_t)(2 * kTJ + kBI) * (KP2 + 1) * 4;
  pixcon_pos_kernel<<<dim3(p.nt_i, p.nsplit2), kThreads, lds2, s>>>(chat, ldc, row_label, pcat, ldp, KP2, meta, inv_T,
                                                                    shift_pos, use_prob, p.nsplit1, p.nsplit2, maxA,
                                                                    negp, maxp, lossp, qsump, Vp, Pmat, ldP);
  rc = check_launch(fn);
  if (rc) return rc;
  pixcon_finalize_kernel<<<ceil_div(BHW, kThreads / 64), kThreads, 0, s>>>(row_label, meta, inv_T, p.nsplit1, p.nsplit2,
                                                                           maxA, negp, lossp, qsump, Up, Vp, grad_a,
                                                                           ldg, N, row_stats, rowloss);
  rc = check_launch(fn);
  if (rc) return rc;
  pixcon_reduce_kernel<<<1, 1024, 0, s>>>(rowloss, meta, loss_out);
  return check_launch(fn);
}

int ucd_pixcon_loss(const float* chat, int ldc, int N, const uint8_t* row_label, const float* pcat, int ldp, int K,
                    const void* ch16, const void* p16, int precision, const ucd_pixcon_meta* meta, int BHW,
                    float temperature, int shift_pos, int use_prob, float* loss_out, float* grad_a, int ldg,
                    float* row_stats, void* workspace, size_t workspace_bytes, ucd_stream_t stream) {
  return pixcon_loss_impl("ucd_pixcon_loss", chat, ldc, N, row_label, pcat, ldp, K, ch16, p16, precision, meta, BHW, temperature,
                          shift_pos, use_prob, nullptr, 0, loss_out, grad_a, ldg, row_stats, workspace, workspace_bytes, stream);
}

int ucd_pixcon_loss_given_p(const float* chat, int ldc, int N, const uint8_t* row_label, const float* P, int ld_P,
                            const ucd_pixcon_meta* meta, int max_anchors, float temperature, int shift_pos,
                            float* loss_out, float* grad_a, int ldg, float* row_stats, void* workspace,
                            size_t workspace_bytes, ucd_stream_t stream) {
  static const char* fn = "ucd_pixcon_loss_given_p";
  UCD_REQUIRE(!P || ld_P > 0, UCD_EINVAL, "%s: ld_P must be positive", fn);
  return pixcon_loss_impl(fn, chat, ldc, N, row_label, nullptr, 0, 0, nullptr, nullptr, UCD_PIXCON_F32, meta, max_anchors,
                          temperature, shift_pos, 0, P, ld_P, loss_out, grad_a, ldg, row_stats, workspace, workspace_bytes,
                          stream);
}

}  // extern "C"
