// ABN (BatchNorm + LeakyReLU/identity) kernels for channels-last activations on gfx950.
//
// Replaces the inplace-abn CUDA extension behind the reference's norm_act(...) layers
// (segmentation_module.py:15-20; 107 instances) and the elementwise glue around it
// (residual add + activation, modules/residual.py:90-97; pooled-branch broadcast add,
// modules/deeplab.py:65-68).  All kernels are HBM streams:
//
//   stats       1 read              sum x, sum x^2 per channel           (two-stage, deterministic)
//   apply       1-2 reads, 1 write  y = act((x+pb)*scale + shift + r)
//   bwd_reduce  2-3 reads           sum dz, sum dz*xhat per channel
//   bwd_apply   2-3 reads, 1-2 wr.  dx (and dz for the fused residual)
//
// Layout: an activation is the row-major matrix [M = B*H*W][C]; every lane moves 16 bytes (4 f32 or
// 8 bf16 channels) so a wave reads 1 KiB of consecutive channels/pixels per instruction.  A 256-thread
// block is TX channel-groups wide (TX <= 64) and TY = 256/TX rows tall; blockIdx.x walks channel
// groups, blockIdx.y owns a contiguous band of rows.  Per-channel parameters sit in registers.
#include "common.h"
#include "abn_finalize.h"

namespace ucd {
namespace {

constexpr int kBlock = 256;
constexpr int kMaxBands = 512;  // row bands (= partial sums per channel) of the two-stage reductions

struct Geom {
  int TX, TY, gx, gy, rows_per_band;
};

// Launch geometry.  Measured on MI355X over every layer shape of the network (B = 24, 513^2; tools/abn_bench.py):
// two 256-thread blocks per CU (512 blocks in all) stream fastest - more resident waves make the 13-107 MB layers
// up to 2x SLOWER (DRAM page / TLB thrash of too many concurrent row streams), fewer leave HBM idle - and a band must
// be a whole number of 4-row batches so that no lane falls into the one-row-at-a-time tail loop.
template <int VEC>
Geom make_geom(int M, int C, int target_blocks, int min_iters, int max_bands = 4096) {
  Geom g;
  const int round_rows = 4;
  int CG = C / VEC;
  g.TX = CG < 64 ? CG : 64;
  g.TY = kBlock / g.TX;
  g.gx = ceil_div(CG, g.TX);
  int by_rows = M / (g.TY * min_iters);
  int gy = target_blocks / g.gx;
  if (gy > by_rows) gy = by_rows;
  if (gy > max_bands) gy = max_bands;
  if (gy < 1) gy = 1;
  g.rows_per_band = ceil_div(M, gy);
  g.rows_per_band = ceil_div(g.rows_per_band, round_rows * g.TY) * round_rows * g.TY;
  g.gy = ceil_div(M, g.rows_per_band);
  return g;
}

// Reduce NV floats per thread over the TY rows of a block; result valid in threads with ty == 0.
template <int NV>
__device__ __forceinline__ void block_reduce_rows(float (&v)[NV], int tx, int ty, int TX, int TY, float* lds) {
  // lds: [kBlock][NV]
#pragma unroll
  for (int i = 0; i < NV; ++i) lds[(ty * TX + tx) * NV + i] = v[i];
  __syncthreads();
  int span = 1;
  while (span < TY) span <<= 1;
  for (int s = span >> 1; s > 0; s >>= 1) {
    if (ty < s && ty + s < TY) {
#pragma unroll
      for (int i = 0; i < NV; ++i) lds[(ty * TX + tx) * NV + i] += lds[((ty + s) * TX + tx) * NV + i];
    }
    __syncthreads();
  }
  if (ty == 0) {
#pragma unroll
    for (int i = 0; i < NV; ++i) v[i] = lds[tx * NV + i];
  }
}

// ---- forward statistics -------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(kBlock) void abn_stats_kernel(const T* __restrict__ x, int ld_x, int M, int C,
                                                          const float* __restrict__ plane_bias, int HW,
                                                          int TX, int TY, int rows_per_band,
                                                          float* __restrict__ partial, float* __restrict__ kshift) {
  constexpr int VEC = Vec<T>::N;
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int tx = threadIdx.x % TX, ty = threadIdx.x / TX;
  const int cg = blockIdx.x * TX + tx;
  const bool live = ty < TY && cg * VEC < C;
  float acc[2 * VEC];
#pragma unroll
  for (int i = 0; i < 2 * VEC; ++i) acc[i] = 0.f;
  const int r_begin = blockIdx.y * rows_per_band;
  const int r_end = min(M, r_begin + rows_per_band);
  if (live) {
    const T* xp = x + (size_t)cg * VEC;
    // Sums are taken about k[c] = the channel's value in row 0 ("shifted data" variance): E[x^2]-E[x]^2
    // cancels catastrophically in fp32 when |mean| >> std, e.g. the 1x1 pooled ASPP branch at small batch.
    float k[VEC];
    {
      Vec<T> v0;
      v0.load(xp);
#pragma unroll
      for (int i = 0; i < VEC; ++i) k[i] = v0.get(i) + (plane_bias ? plane_bias[cg * VEC + i] : 0.f);
      if (blockIdx.y == 0 && ty == 0) {
#pragma unroll
        for (int i = 0; i < VEC; ++i) kshift[cg * VEC + i] = k[i];
      }
    }
    auto accumulate = [&](const Vec<T>& v, int r) {
      if (plane_bias) {
        const float* pb = plane_bias + (size_t)(r / HW) * C + cg * VEC;
#pragma unroll
        for (int i = 0; i < VEC; ++i) {
          float f = (v.get(i) + pb[i]) - k[i];
          acc[i] += f;
          acc[VEC + i] += f * f;
        }
      } else {
#pragma unroll
        for (int i = 0; i < VEC; ++i) {
          float f = v.get(i) - k[i];
          acc[i] += f;
          acc[VEC + i] += f * f;
        }
      }
    };
    int r = r_begin + ty;
    // four independent 16-byte loads in flight per lane: with <= 512 row bands (2 blocks per CU) the loop
    // is latency-bound otherwise
    for (; r + 3 * TY < r_end; r += 4 * TY) {
      Vec<T> v0, v1, v2, v3;
      v0.load(xp + (size_t)r * ld_x);
      v1.load(xp + (size_t)(r + TY) * ld_x);
      v2.load(xp + (size_t)(r + 2 * TY) * ld_x);
      v3.load(xp + (size_t)(r + 3 * TY) * ld_x);
      accumulate(v0, r);
      accumulate(v1, r + TY);
      accumulate(v2, r + 2 * TY);
      accumulate(v3, r + 3 * TY);
    }
    for (; r < r_end; r += TY) {
      Vec<T> v;
      v.load(xp + (size_t)r * ld_x);
      accumulate(v, r);
    }
  }
  // idle tail threads (TX does not divide 256) carry zeros through the reduction: ty*TX+tx == threadIdx.x
  block_reduce_rows<2 * VEC>(acc, tx, ty, TX, TY, lds);
  if (ty == 0 && cg * VEC < C) {
    float* p = partial + (size_t)blockIdx.y * 2 * C + cg * VEC;
#pragma unroll
    for (int i = 0; i < VEC; ++i) {
      p[i] = acc[i];
      p[C + i] = acc[VEC + i];
    }
  }
}

// Stage 2 of both reductions: sums[c] / sums[C+c] = sum over bands of the two partial rows of channel c.
// A block owns 8 channels (16 outputs) x 16 band-lanes, four independent loads in flight per thread (the
// loop is latency-bound: a partial row is only 2C floats), fixed combination order (deterministic).
// With FINALIZE the same block turns the two sums into the normalisation constants (one launch less per
// layer than a separate finalize kernel).
// MODE 0: sums only (and an optional second copy), 1: + finalize, 2: + pack [mean_r | M2_r] for the SyncBN gather
template <int MODE, int LANES = 16>
__global__ __launch_bounds__(16 * LANES) void reduce_bands_kernel(const float* __restrict__ partial, int bands, int C,
                                                                  float* __restrict__ sums, FinalizeArgs fin,
                                                                  float* __restrict__ sums2 = nullptr,
                                                                  const float* __restrict__ sign_of = nullptr) {
  // LANES band-lanes per channel row (16, or 64 = 1024 threads when there are hundreds of bands / row tiles: the loop is a chain
  // of dependent ~1 us round trips, 6-7 us per call with 16 lanes on 512 bands - and 220 such calls per step)
  __shared__ float lds[LANES][17];
  const int kl = threadIdx.x & 15, lane = threadIdx.x >> 4;
  const int c = blockIdx.x * 8 + (kl & 7);
  const int k = (kl < 8 ? 0 : C) + c;   // row index inside a [2C] partial
  const int n = 2 * C;
  float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
  FinalizeIn pre{1.f, 0.f, 0.f};
  float pre_k = 0.f;
  if (MODE == 1 && lane == 0 && kl < 8 && c < C) {       // the finalising thread's inputs: in flight under the whole reduction
    pre = finalize_prefetch(c, fin);
    pre_k = fin.kshift ? fin.kshift[c] : 0.f;
  }
  if (c < C) {
    int b = lane;
    for (; b + 3 * LANES < bands; b += 4 * LANES) {
      s0 += partial[(size_t)b * n + k];
      s1 += partial[(size_t)(b + LANES) * n + k];
      s2 += partial[(size_t)(b + 2 * LANES) * n + k];
      s3 += partial[(size_t)(b + 3 * LANES) * n + k];
    }
    for (; b < bands; b += LANES) s0 += partial[(size_t)b * n + k];
  }
  lds[lane][kl] = (s0 + s1) + (s2 + s3);
  __syncthreads();
  if (LANES > 16) {                      // 64 lanes -> 16 (fixed order), then the common tail
    if (lane < 16) {
      float t = 0.f;
#pragma unroll
      for (int i = 0; i < LANES / 16; ++i) t += lds[lane * (LANES / 16) + i][kl];
      lds[lane * (LANES / 16)][kl] = t;
    }
    __syncthreads();
  }
  if (lane == 0 && c < C) {
    float t = 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i) t += lds[i * (LANES / 16)][kl];
    // abs-gamma layers: the second row holds d weight = sign(weight) * sum dz*xhat (bwd_apply undoes the sign)
    if (sign_of && kl >= 8 && sign_of[c] < 0.f) t = -t;
    sums[k] = t;
    if (sums2) sums2[k] = t;
    lds[0][kl] = t;
  }
  if (MODE == 1) {
    __syncthreads();
    if (lane == 0 && kl < 8 && c < C) finalize_channel_pre(c, lds[0][kl], lds[0][kl + 8], pre_k, fin, pre);
  }
  if (MODE == 2) {
    __syncthreads();
    if (lane == 0 && kl < 8 && c < C) {
      const float s1 = lds[0][kl], s2 = lds[0][kl + 8];
      const float d = s1 / fin.count;
      fin.pack[c] = fin.kshift[c] + d;
      fin.pack[C + c] = s2 - s1 * d;
    }
  }
}

// launch: 64 band-lanes (1024 threads) from 96 bands up
#define UCD_REDUCE_BANDS(MODE, bands, ...)                                                        \
  do {                                                                                             \
    if ((bands) >= 96) reduce_bands_kernel<MODE, 64><<<ceil_div(C, 8), 1024, 0, s>>>(__VA_ARGS__);  \
    else reduce_bands_kernel<MODE, 16><<<ceil_div(C, 8), 256, 0, s>>>(__VA_ARGS__);                \
  } while (0)

// SyncBN: Chan's combination of the per-rank (mean, M2) pairs (equal counts per rank) + the usual finalize.
// gathered: [world][2C] = [mean_r | M2_r]; fin.count = world * m_local.
__global__ void abn_combine_finalize_kernel(const float* __restrict__ gathered, int world, int C, float m_local,
                                            FinalizeArgs fin) {
  int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  float mean = 0.f;
  for (int r = 0; r < world; ++r) mean += gathered[(size_t)r * 2 * C + c];
  mean /= (float)world;
  float m2 = 0.f, dev2 = 0.f;
  for (int r = 0; r < world; ++r) {
    const float d = gathered[(size_t)r * 2 * C + c] - mean;
    m2 += gathered[(size_t)r * 2 * C + C + c];
    dev2 += d * d;
  }
  finalize_moments(c, mean, m2 + m_local * dev2, fin);
}

__global__ void abn_finalize_kernel(const float* __restrict__ sums, int C, FinalizeArgs fin) {
  int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  finalize_channel(c, sums[c], sums[C + c], fin);
}

__global__ void abn_eval_params_kernel(const float* __restrict__ weight, const float* __restrict__ rv, float eps, int C,
                                       float* __restrict__ invstd, float* __restrict__ scale, int abs_gamma) {
  int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  float is = 1.f / sqrtf(rv[c] + eps);
  invstd[c] = is;
  scale[c] = (weight ? gamma_eff(weight[c], eps, abs_gamma) : 1.f) * is;
}

// ---- forward apply ------------------------------------------------------------------------------
template <typename T, int ACT>
__global__ __launch_bounds__(kBlock) void abn_apply_kernel(const T* x, int ld_x, T* y, int ld_y,
                                                          const T* __restrict__ res, int ld_r, int M, int C,
                                                          const float* __restrict__ plane_bias, int HW,
                                                          const float* __restrict__ mean, const float* __restrict__ scale,
                                                          const float* __restrict__ beta,
                                                          float slope, int TX, int TY, int rows_per_band) {
  constexpr int VEC = Vec<T>::N;
  const int tx = threadIdx.x % TX, ty = threadIdx.x / TX;
  const int cg = blockIdx.x * TX + tx;
  if (ty >= TY || cg * VEC >= C) return;
  float mu[VEC], sc[VEC], sh[VEC];
#pragma unroll
  for (int i = 0; i < VEC; ++i) {
    mu[i] = mean[cg * VEC + i];
    sc[i] = scale[cg * VEC + i];
    sh[i] = beta ? beta[cg * VEC + i] : 0.f;
  }
  const int r_begin = blockIdx.y * rows_per_band;
  const int r_end = min(M, r_begin + rows_per_band);
  const size_t coff = (size_t)cg * VEC;
  auto emit = [&](const Vec<T>& v, const Vec<T>& rv, int r) {
    Vec<T> o;
    const float* pb = plane_bias ? plane_bias + (size_t)(r / HW) * C + coff : nullptr;
#pragma unroll
    for (int i = 0; i < VEC; ++i) {
      float f = v.get(i);
      if (pb) f += pb[i];
      float z = (f - mu[i]) * sc[i] + sh[i];     // subtract first: exact when x is close to the mean
      if (res) z += rv.get(i);
      o.set(i, act_fwd<ACT>(z, slope));
    }
    o.store(y + (size_t)r * ld_y + coff);
  };
  int r = r_begin + ty;
  // four rows (4-8 independent 16-byte loads) in flight per lane: the small stride-16 layers are a handful
  // of rows per thread, so a one-row-at-a-time loop is a chain of exposed HBM round trips
  for (; r + 3 * TY < r_end; r += 4 * TY) {
    Vec<T> v0, v1, v2, v3, q0, q1, q2, q3;
    v0.load(x + (size_t)r * ld_x + coff);
    v1.load(x + (size_t)(r + TY) * ld_x + coff);
    v2.load(x + (size_t)(r + 2 * TY) * ld_x + coff);
    v3.load(x + (size_t)(r + 3 * TY) * ld_x + coff);
    if (res) {
      q0.load(res + (size_t)r * ld_r + coff);
      q1.load(res + (size_t)(r + TY) * ld_r + coff);
      q2.load(res + (size_t)(r + 2 * TY) * ld_r + coff);
      q3.load(res + (size_t)(r + 3 * TY) * ld_r + coff);
    }
    emit(v0, q0, r);
    emit(v1, q1, r + TY);
    emit(v2, q2, r + 2 * TY);
    emit(v3, q3, r + 3 * TY);
  }
  for (; r < r_end; r += TY) {
    Vec<T> v, q;
    v.load(x + (size_t)r * ld_x + coff);
    if (res) q.load(res + (size_t)r * ld_r + coff);
    emit(v, q, r);
  }
}

// bf16, no plane bias, leaky_relu / identity (slope = 1): the layers of the train step.  Same arithmetic as abn_apply_kernel
// ((x - mean) * scale + shift, + residual, select; no contraction) on float pairs.
// FIN (round 5): the statistics arrive as RAW sums about a shift (fin.acc = [sum (x - k) | sum (x - k)^2], fin.kshift = k: the
// atomic accumulator of the producing GEMM's epilogue) and every thread finalises its eight channels itself - the arithmetic of
// finalize_channel_pre; the first row band also stores mean / invstd / scale for the backward and updates the running statistics
// (what tile_stats_reduce_kernel<1> did in a launch of its own, 103 times per step).
// Sum of the `reps` replicas of a [2 C] accumulator for this thread's eight channels, valid in EVERY thread on return.  The replicas
// were filled by atomics (executed at the memory side: their lines come from memory, ~1-2 us per dependent round trip), so the TY
// row threads of a channel group split them - thread ty takes replicas ty, ty + TY, ... with up to four (16 loads) in flight - and
// combine through LDS in a fixed order (block_reduce_rows).  reps == 1: a plain load.  Called by all 256 threads (barriers inside).
__device__ __forceinline__ void replica_sum(const float* __restrict__ acc, int reps, int C, size_t coff, int tx, int ty, int TX, int TY,
                                            bool live, float* lds, Pack8& s1, Pack8& s2) {
  if (reps <= 1) {
    if (live) { s1 = load_f8(acc + coff); s2 = load_f8(acc + C + coff); }
    return;
  }
  float v[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) v[i] = 0.f;
  if (live) {
    for (int r = ty; r < reps; r += 4 * TY) {
      Pack8 a[4], b[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int rr = r + u * TY;
        if (rr < reps) {
          a[u] = load_f8(acc + (size_t)rr * 2 * C + coff);
          b[u] = load_f8(acc + (size_t)rr * 2 * C + C + coff);
        } else {
#pragma unroll
          for (int j = 0; j < 4; ++j) { a[u].p[j] = f32x2{0.f, 0.f}; b[u].p[j] = f32x2{0.f, 0.f}; }
        }
      }
#pragma unroll
      for (int u = 0; u < 4; ++u)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          v[2 * j] += a[u].p[j].x; v[2 * j + 1] += a[u].p[j].y;
          v[8 + 2 * j] += b[u].p[j].x; v[8 + 2 * j + 1] += b[u].p[j].y;
        }
    }
  }
  block_reduce_rows<16>(v, tx, ty, TX, TY, lds);
  // row 0 of the reduction scratch holds the totals (the last tree level ends with a barrier)
  if (live) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      s1.p[j] = f32x2{lds[tx * 16 + 2 * j], lds[tx * 16 + 2 * j + 1]};
      s2.p[j] = f32x2{lds[tx * 16 + 8 + 2 * j], lds[tx * 16 + 8 + 2 * j + 1]};
    }
  }
}

struct ApplyFin {
  const float* acc;        // [reps][2 C]
  int reps;
  const float* kshift;     // [C]
  const float* weight;     // [C] or NULL
  float* running_mean;     // [C] or NULL
  float* running_var;
  float* mean;             // outputs [C]
  float* invstd;
  float* scale;
  float count, momentum, eps;
  int abs_gamma;
};
__device__ __forceinline__ void apply_fin_channels(const ApplyFin& f, const Pack8& s1, const Pack8& s2, const Pack8& k, const Pack8& w,
                                                   size_t coff, bool writer, Pack8& mu, Pack8& sc) {
  const float inv_n = 1.f / f.count;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    float m[2], is[2], scl[2];
#pragma unroll
    for (int e = 0; e < 2; ++e) {
      const float s = s1.p[j][e], ss = s2.p[j][e];
      const float d = s * inv_n;                      // mean - k
      m[e] = k.p[j][e] + d;
      const float var = fmaxf((ss - s * d) * inv_n, 0.f);
      // v_rsq_f32 (1 ulp) + one Newton step: every thread of the launch finalises its eight channels, and the IEEE 1 / sqrt
      // sequence (~45 instructions per channel) was a microsecond of every apply launch
      const float ve = var + f.eps;
      const float r0 = __builtin_amdgcn_rsqf(ve);
      is[e] = r0 * (1.5f - 0.5f * ve * r0 * r0);
      scl[e] = (f.weight ? gamma_eff(w.p[j][e], f.eps, f.abs_gamma) : 1.f) * is[e];
      if (writer) {
        const int c = (int)coff + 2 * j + e;
        if (f.running_mean) f.running_mean[c] = (1.f - f.momentum) * f.running_mean[c] + f.momentum * m[e];
        if (f.running_var) {
          const float unbiased = f.count > 1.f ? var * (f.count / (f.count - 1.f)) : var;
          f.running_var[c] = (1.f - f.momentum) * f.running_var[c] + f.momentum * unbiased;
        }
        f.mean[c] = m[e];
        f.invstd[c] = is[e];
        f.scale[c] = scl[e];
      }
    }
    mu.p[j] = f32x2{m[0], m[1]};
    sc.p[j] = f32x2{scl[0], scl[1]};
  }
}

template <bool RES, bool FIN = false>
__global__ __launch_bounds__(kBlock) void abn_apply_fast_kernel(const __hip_bfloat16* x, int ld_x, __hip_bfloat16* y, int ld_y,
                                                               const __hip_bfloat16* __restrict__ res, int ld_r, int M, int C,
                                                               const float* __restrict__ mean, const float* __restrict__ scale,
                                                               const float* __restrict__ beta, float slope, int TX, int TY,
                                                               int rows_per_band, ApplyFin fin = ApplyFin{}) {
  extern __shared__ __attribute__((aligned(16))) float fin_lds[];   // FIN with replicas: [256][16] reduction scratch
  const int tx = threadIdx.x % TX, ty = threadIdx.x / TX;
  const int cg = blockIdx.x * TX + tx;
  const bool live = ty < TY && cg * 8 < C;
  if (!FIN && !live) return;
  const size_t coff = (size_t)cg * 8;
  Pack8 mu, sc;
  const int r_begin = blockIdx.y * rows_per_band;
  const int r_end = min(M, r_begin + rows_per_band);
  auto ld = [&](const __hip_bfloat16* p, int ldp, int r) {
    const uint4* q = reinterpret_cast<const uint4*>(p + (size_t)r * ldp + coff);
    if (p == x || p == res) {      // round 6: z is not read again before the backward - a non-temporal load keeps it from displacing what the next
                       // GEMM wants in L2 / the Infinity Cache (same-box A/B 29.77 -> 29.62 ms at 24 images; the residual, read here for the last time in the forward, the same way: 30.30 -> 30.23; profiles/r06_kernel_ab_during.txt)
      typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
      const u32x4 v = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(q));
      return make_uint4(v[0], v[1], v[2], v[3]);
    }
    return *q;
  };
  int r = r_begin + ty;
  // FIN: the first batch of rows is requested BEFORE the statistics (vector memory returns in order: rows first, then the accumulator
  // - both round trips overlap, and the accumulator's lines come from memory: atomics leave nothing in L2)
  const bool pre = FIN && live && r + 3 * TY < r_end;
  uint4 pv0, pv1, pv2, pv3, pq0, pq1, pq2, pq3;
  if (pre) {
    pv0 = ld(x, ld_x, r); pv1 = ld(x, ld_x, r + TY); pv2 = ld(x, ld_x, r + 2 * TY); pv3 = ld(x, ld_x, r + 3 * TY);
    pq0 = pv0; pq1 = pv0; pq2 = pv0; pq3 = pv0;
    if (RES) { pq0 = ld(res, ld_r, r); pq1 = ld(res, ld_r, r + TY); pq2 = ld(res, ld_r, r + 2 * TY); pq3 = ld(res, ld_r, r + 3 * TY); }
  }
  if (FIN) {
    Pack8 s1, s2, kk, ww;
    if (live) {                                         // requested with the rows, ahead of the accumulator's round trip
      kk = load_f8(fin.kshift + coff);
      if (fin.weight) ww = load_f8(fin.weight + coff);
    }
    replica_sum(fin.acc, fin.reps, C, coff, tx, ty, TX, TY, live, fin_lds, s1, s2);
    if (!live) return;
    apply_fin_channels(fin, s1, s2, kk, ww, coff, blockIdx.y == 0 && ty == 0, mu, sc);
  } else {
    mu = load_f8(mean + coff);
    sc = load_f8(scale + coff);
  }
  Pack8 sh;
  if (beta) sh = load_f8(beta + coff);
  else {
#pragma unroll
    for (int j = 0; j < 4; ++j) sh.p[j] = f32x2{0.f, 0.f};
  }
  auto emit = [&](const uint4& xv, const uint4& rv, int r) {
    const Pack8 v = unpack8(xv);
    Pack8 o;
    if (RES) {
      const Pack8 q = unpack8(rv);
#pragma unroll
      for (int j = 0; j < 4; ++j) o.p[j] = leaky2((v.p[j] - mu.p[j]) * sc.p[j] + sh.p[j] + q.p[j], slope);
    } else {
#pragma unroll
      for (int j = 0; j < 4; ++j) o.p[j] = leaky2((v.p[j] - mu.p[j]) * sc.p[j] + sh.p[j], slope);
    }
    *reinterpret_cast<uint4*>(y + (size_t)r * ld_y + coff) = pack8(o);
  };
  if (pre) {
    emit(pv0, pq0, r);
    emit(pv1, pq1, r + TY);
    emit(pv2, pq2, r + 2 * TY);
    emit(pv3, pq3, r + 3 * TY);
    r += 4 * TY;
  }
  for (; r + 3 * TY < r_end; r += 4 * TY) {
    uint4 v0 = ld(x, ld_x, r), v1 = ld(x, ld_x, r + TY), v2 = ld(x, ld_x, r + 2 * TY), v3 = ld(x, ld_x, r + 3 * TY);
    uint4 q0 = v0, q1 = v0, q2 = v0, q3 = v0;
    if (RES) {
      q0 = ld(res, ld_r, r); q1 = ld(res, ld_r, r + TY); q2 = ld(res, ld_r, r + 2 * TY); q3 = ld(res, ld_r, r + 3 * TY);
    }
    emit(v0, q0, r);
    emit(v1, q1, r + TY);
    emit(v2, q2, r + 2 * TY);
    emit(v3, q3, r + 3 * TY);
  }
  for (; r < r_end; r += TY) {
    const uint4 v = ld(x, ld_x, r);
    uint4 q = v;
    if (RES) q = ld(res, ld_r, r);
    emit(v, q, r);
  }
}

// ---- backward reduce ----------------------------------------------------------------------------
template <typename T, int ACT>
__global__ __launch_bounds__(kBlock) void abn_bwd_reduce_kernel(
    const T* __restrict__ x, int ld_x, const T* __restrict__ dy, int ld_dy, const T* __restrict__ yout, int ld_y,
    int M, int C, const float* __restrict__ plane_bias, int HW, const float* __restrict__ mean,
    const float* __restrict__ invstd, const float* __restrict__ scale, const float* __restrict__ shift, float slope,
    int TX, int TY, int rows_per_band, float* __restrict__ partial) {
  constexpr int VEC = Vec<T>::N;
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int tx = threadIdx.x % TX, ty = threadIdx.x / TX;
  const int cg = blockIdx.x * TX + tx;
  const bool live = ty < TY && cg * VEC < C;
  float acc[2 * VEC];
#pragma unroll
  for (int i = 0; i < 2 * VEC; ++i) acc[i] = 0.f;
  if (live) {
    float mu[VEC], is[VEC], sc[VEC], sh[VEC];
#pragma unroll
    for (int i = 0; i < VEC; ++i) {
      mu[i] = mean[cg * VEC + i];
      is[i] = invstd[cg * VEC + i];
      sc[i] = scale[cg * VEC + i];
      sh[i] = shift ? shift[cg * VEC + i] : 0.f;
    }
    const int r_begin = blockIdx.y * rows_per_band;
    const int r_end = min(M, r_begin + rows_per_band);
    const size_t coff = (size_t)cg * VEC;
    auto accumulate = [&](const Vec<T>& v, const Vec<T>& g, const Vec<T>& yo, int r) {
      const float* pb = plane_bias ? plane_bias + (size_t)(r / HW) * C + coff : nullptr;
#pragma unroll
      for (int i = 0; i < VEC; ++i) {
        float f = v.get(i);
        if (pb) f += pb[i];
        float sgn = yout ? yo.get(i) : (f - mu[i]) * sc[i] + sh[i];
        float dz = g.get(i) * act_grad<ACT>(sgn, slope, yout != nullptr);
        float xh = (f - mu[i]) * is[i];
        acc[i] += dz;
        acc[VEC + i] += dz * xh;
      }
    };
    int r = r_begin + ty;
    for (; r + 3 * TY < r_end; r += 4 * TY) {   // four rows (8-12 independent loads) in flight per lane
      Vec<T> v0, g0, y0, v1, g1, y1, v2, g2, y2, v3, g3, y3;
      v0.load(x + (size_t)r * ld_x + coff);
      g0.load(dy + (size_t)r * ld_dy + coff);
      v1.load(x + (size_t)(r + TY) * ld_x + coff);
      g1.load(dy + (size_t)(r + TY) * ld_dy + coff);
      v2.load(x + (size_t)(r + 2 * TY) * ld_x + coff);
      g2.load(dy + (size_t)(r + 2 * TY) * ld_dy + coff);
      v3.load(x + (size_t)(r + 3 * TY) * ld_x + coff);
      g3.load(dy + (size_t)(r + 3 * TY) * ld_dy + coff);
      if (yout) {
        y0.load(yout + (size_t)r * ld_y + coff);
        y1.load(yout + (size_t)(r + TY) * ld_y + coff);
        y2.load(yout + (size_t)(r + 2 * TY) * ld_y + coff);
        y3.load(yout + (size_t)(r + 3 * TY) * ld_y + coff);
      }
      accumulate(v0, g0, y0, r);
      accumulate(v1, g1, y1, r + TY);
      accumulate(v2, g2, y2, r + 2 * TY);
      accumulate(v3, g3, y3, r + 3 * TY);
    }
    for (; r < r_end; r += TY) {
      Vec<T> v, g, yo;
      v.load(x + (size_t)r * ld_x + coff);
      g.load(dy + (size_t)r * ld_dy + coff);
      if (yout) yo.load(yout + (size_t)r * ld_y + coff);
      accumulate(v, g, yo, r);
    }
  }
  block_reduce_rows<2 * VEC>(acc, tx, ty, TX, TY, lds);
  if (ty == 0 && cg * VEC < C) {
    float* p = partial + (size_t)blockIdx.y * 2 * C + cg * VEC;
#pragma unroll
    for (int i = 0; i < VEC; ++i) {
      p[i] = acc[i];
      p[C + i] = acc[VEC + i];
    }
  }
}

// ---- backward apply -----------------------------------------------------------------------------
template <typename T, int ACT>
__global__ __launch_bounds__(kBlock) void abn_bwd_apply_kernel(
    const T* x, int ld_x, const T* dy, int ld_dy, const T* yout, int ld_y,
    T* dx, int ld_dx, T* dz_out, int ld_dz, int M, int C,
    const float* __restrict__ plane_bias, int HW, const float* __restrict__ mean, const float* __restrict__ invstd,
    const float* __restrict__ scale, const float* __restrict__ shift, const float* __restrict__ weight,
    const float* __restrict__ sums, float inv_count, int frozen, int abs_gamma, float slope, int TX, int TY,
    int rows_per_band) {
  constexpr int VEC = Vec<T>::N;
  const int tx = threadIdx.x % TX, ty = threadIdx.x / TX;
  const int cg = blockIdx.x * TX + tx;
  if (ty >= TY || cg * VEC >= C) return;
  float mu[VEC], is[VEC], sc[VEC], sh[VEC], k0[VEC], k1[VEC], gw[VEC];
#pragma unroll
  for (int i = 0; i < VEC; ++i) {
    const int c = cg * VEC + i;
    sc[i] = scale[c];
    sh[i] = shift ? shift[c] : 0.f;
    mu[i] = mean[c];
    if (frozen) {
      is[i] = 0.f; k0[i] = 0.f; k1[i] = 0.f;
      gw[i] = sc[i];
    } else {
      is[i] = invstd[c];
      k0[i] = sums[c] * inv_count;       // mean(dz)
      k1[i] = sums[C + c] * inv_count;   // mean(dz * xhat)
      if (abs_gamma) {                   // sums[C + c] is d weight = sign(weight) * sum dz*xhat; scale = (|w| + eps) * invstd
        if (weight && weight[c] < 0.f) k1[i] = -k1[i];
        gw[i] = sc[i];
      } else {
        gw[i] = (weight ? weight[c] : 1.f) * is[i];
      }
    }
  }
  const int r_begin = blockIdx.y * rows_per_band;
  const int r_end = min(M, r_begin + rows_per_band);
  const size_t coff = (size_t)cg * VEC;
  auto emit = [&](const Vec<T>& v, const Vec<T>& g, const Vec<T>& yo, int r) {
    Vec<T> o, oz;
    const float* pb = plane_bias ? plane_bias + (size_t)(r / HW) * C + coff : nullptr;
#pragma unroll
    for (int i = 0; i < VEC; ++i) {
      float f = v.get(i);
      if (pb) f += pb[i];
      float sgn = yout ? yo.get(i) : (f - mu[i]) * sc[i] + sh[i];
      float dz = g.get(i) * act_grad<ACT>(sgn, slope, yout != nullptr);
      float xh = (f - mu[i]) * is[i];
      o.set(i, (dz - k0[i] - xh * k1[i]) * gw[i]);
      oz.set(i, dz);
    }
    o.store(dx + (size_t)r * ld_dx + coff);
    if (dz_out) oz.store(dz_out + (size_t)r * ld_dz + coff);
  };
  int r = r_begin + ty;
  for (; r + 3 * TY < r_end; r += 4 * TY) {   // four rows (8-12 independent loads) in flight per lane
    Vec<T> v0, v1, v2, v3, g0, g1, g2, g3, y0, y1, y2, y3;
    v0.load(x + (size_t)r * ld_x + coff);
    g0.load(dy + (size_t)r * ld_dy + coff);
    v1.load(x + (size_t)(r + TY) * ld_x + coff);
    g1.load(dy + (size_t)(r + TY) * ld_dy + coff);
    v2.load(x + (size_t)(r + 2 * TY) * ld_x + coff);
    g2.load(dy + (size_t)(r + 2 * TY) * ld_dy + coff);
    v3.load(x + (size_t)(r + 3 * TY) * ld_x + coff);
    g3.load(dy + (size_t)(r + 3 * TY) * ld_dy + coff);
    if (yout) {
      y0.load(yout + (size_t)r * ld_y + coff);
      y1.load(yout + (size_t)(r + TY) * ld_y + coff);
      y2.load(yout + (size_t)(r + 2 * TY) * ld_y + coff);
      y3.load(yout + (size_t)(r + 3 * TY) * ld_y + coff);
    }
    emit(v0, g0, y0, r);
    emit(v1, g1, y1, r + TY);
    emit(v2, g2, y2, r + 2 * TY);
    emit(v3, g3, y3, r + 3 * TY);
  }
  for (; r < r_end; r += TY) {
    Vec<T> v, g, yo;
    v.load(x + (size_t)r * ld_x + coff);
    g.load(dy + (size_t)r * ld_dy + coff);
    if (yout) yo.load(yout + (size_t)r * ld_y + coff);
    emit(v, g, yo, r);
  }
}

// ---- backward, bf16 fast path (no plane bias, leaky_relu / identity): packed-fp32 math, nullable operands as flags --------
template <bool YOUT>
__global__ __launch_bounds__(kBlock) void abn_bwd_reduce_fast_kernel(
    const __hip_bfloat16* __restrict__ x, int ld_x, const __hip_bfloat16* __restrict__ dy, int ld_dy,
    const __hip_bfloat16* __restrict__ yout, int ld_y, int M, int C, const float* __restrict__ mean,
    const float* __restrict__ invstd, const float* __restrict__ scale, const float* __restrict__ shift, float slope, int TX, int TY,
    int rows_per_band, float* __restrict__ partial) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int tx = threadIdx.x % TX, ty = threadIdx.x / TX;
  const int cg = blockIdx.x * TX + tx;
  const bool live = ty < TY && cg * 8 < C;
  f32x2 s1[4], s2[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) { s1[j] = f32x2{0.f, 0.f}; s2[j] = f32x2{0.f, 0.f}; }
  if (live) {
    const size_t coff = (size_t)cg * 8;
    const Pack8 mu = load_f8(mean + coff), is = load_f8(invstd + coff), sc = load_f8(scale + coff);
    Pack8 sh;
    if (shift) sh = load_f8(shift + coff);
    else {
#pragma unroll
      for (int j = 0; j < 4; ++j) sh.p[j] = f32x2{0.f, 0.f};
    }
    const int r_begin = blockIdx.y * rows_per_band;
    const int r_end = min(M, r_begin + rows_per_band);
    auto accumulate = [&](const uint4& xv, const uint4& gv, const uint4& yv) {
      const Pack8 v = unpack8(xv), g = unpack8(gv);
      Pack8 yo;
      if (YOUT) yo = unpack8(yv);
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const f32x2 d = v.p[j] - mu.p[j];
        const f32x2 sgn = YOUT ? yo.p[j] : d * sc.p[j] + sh.p[j];
        const f32x2 dz = leaky_grad2(g.p[j], sgn, slope);
        s1[j] += dz;
        s2[j] += dz * (d * is.p[j]);
      }
    };
    auto ld = [&](const __hip_bfloat16* p, int ldp, int r) { return *reinterpret_cast<const uint4*>(p + (size_t)r * ldp + coff); };
    int r = r_begin + ty;
    for (; r + 3 * TY < r_end; r += 4 * TY) {
      const uint4 v0 = ld(x, ld_x, r), g0 = ld(dy, ld_dy, r), v1 = ld(x, ld_x, r + TY), g1 = ld(dy, ld_dy, r + TY);
      const uint4 v2 = ld(x, ld_x, r + 2 * TY), g2 = ld(dy, ld_dy, r + 2 * TY), v3 = ld(x, ld_x, r + 3 * TY), g3 = ld(dy, ld_dy, r + 3 * TY);
      uint4 y0 = v0, y1 = v0, y2 = v0, y3 = v0;
      if (YOUT) { y0 = ld(yout, ld_y, r); y1 = ld(yout, ld_y, r + TY); y2 = ld(yout, ld_y, r + 2 * TY); y3 = ld(yout, ld_y, r + 3 * TY); }
      accumulate(v0, g0, y0);
      accumulate(v1, g1, y1);
      accumulate(v2, g2, y2);
      accumulate(v3, g3, y3);
    }
    for (; r < r_end; r += TY) {
      const uint4 v = ld(x, ld_x, r), g = ld(dy, ld_dy, r);
      uint4 yo = v;
      if (YOUT) yo = ld(yout, ld_y, r);
      accumulate(v, g, yo);
    }
  }
  float acc[16];
#pragma unroll
  for (int j = 0; j < 4; ++j) { acc[2 * j] = s1[j].x; acc[2 * j + 1] = s1[j].y; acc[8 + 2 * j] = s2[j].x; acc[8 + 2 * j + 1] = s2[j].y; }
  block_reduce_rows<16>(acc, tx, ty, TX, TY, lds);
  if (ty == 0 && cg * 8 < C) {
    float* p = partial + (size_t)blockIdx.y * 2 * C + cg * 8;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      p[i] = acc[i];
      p[C + i] = acc[8 + i];
    }
  }
}

// dx = (dz - mean(dz) - xhat * mean(dz xhat)) * gamma invstd with the same operation order as abn_bwd_apply_kernel:
// ((dz - k0) - xhat k1) gw (the library is built with -ffp-contract=off: no fma anywhere).  frozen: k0 = k1 = invstd = 0, gw = scale.
// RAW (round 5): sums are the atomic accumulator of a link epilogue - [sum dz | sum dz * xhat] with no sign applied; the first row
// band writes the layer's parameter gradients [d bias | d weight] to grad_out from grad_sums (what reduce_bands_kernel did)
template <bool YOUT, bool DZOUT, bool RAW = false>
__global__ __launch_bounds__(kBlock) void abn_bwd_apply_fast_kernel(
    const __hip_bfloat16* x, int ld_x, const __hip_bfloat16* dy, int ld_dy, const __hip_bfloat16* yout, int ld_y,
    __hip_bfloat16* dx, int ld_dx, __hip_bfloat16* dz_out, int ld_dz, int M, int C, const float* __restrict__ mean,
    const float* __restrict__ invstd, const float* __restrict__ scale, const float* __restrict__ shift,
    const float* __restrict__ weight, const float* __restrict__ sums, float inv_count, int frozen, int abs_gamma, float slope,
    int TX, int TY, int rows_per_band, const float* __restrict__ grad_sums = nullptr, float* __restrict__ grad_out = nullptr,
    int reps = 1) {
  extern __shared__ __attribute__((aligned(16))) float raw_lds[];   // RAW with replicas: [256][16] reduction scratch
  const int tx = threadIdx.x % TX, ty = threadIdx.x / TX;
  const int cg = blockIdx.x * TX + tx;
  const bool live = ty < TY && cg * 8 < C;
  if (!RAW && !live) return;
  const size_t coff = (size_t)cg * 8;
  Pack8 rs0, rs1, rg0, rg1;                             // RAW: the replica-summed sums (and this rank's, for the parameter gradients)
  if (RAW) {
    replica_sum(sums, reps, C, coff, tx, ty, TX, TY, live, raw_lds, rs0, rs1);
    const bool want_g = grad_out != nullptr && blockIdx.y == 0;       // block-uniform
    if (want_g && grad_sums != sums) {
      __syncthreads();                                  // the scratch is read by everyone before it is reused
      replica_sum(grad_sums, reps, C, coff, tx, ty, TX, TY, live, raw_lds, rg0, rg1);
    } else {
      rg0 = rs0; rg1 = rs1;
    }
    if (!live) return;
  }
  const Pack8 mu = load_f8(mean + coff), sc = load_f8(scale + coff);
  Pack8 sh, is, k0, k1, gw;
  if (shift) sh = load_f8(shift + coff);
  else {
#pragma unroll
    for (int j = 0; j < 4; ++j) sh.p[j] = f32x2{0.f, 0.f};
  }
  if (frozen) {
#pragma unroll
    for (int j = 0; j < 4; ++j) { is.p[j] = f32x2{0.f, 0.f}; k0.p[j] = f32x2{0.f, 0.f}; k1.p[j] = f32x2{0.f, 0.f}; gw.p[j] = sc.p[j]; }
  } else {
    is = load_f8(invstd + coff);
    Pack8 a0, a1;
    if (RAW) { a0 = rs0; a1 = rs1; }
    else { a0 = load_f8(sums + coff); a1 = load_f8(sums + C + coff); }
    Pack8 w;
    if (weight) w = load_f8(weight + coff);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      k0.p[j] = a0.p[j] * inv_count;                    // mean(dz)
      f32x2 kk = a1.p[j] * inv_count;                   // mean(dz * xhat)
      if (abs_gamma) {                                  // sums[C + c] is d weight = sign(weight) * sum dz*xhat; scale = (|w| + eps) * invstd
        if (weight && !RAW) kk = f32x2{w.p[j].x < 0.f ? -kk.x : kk.x, w.p[j].y < 0.f ? -kk.y : kk.y};
        gw.p[j] = sc.p[j];
      } else {
        gw.p[j] = weight ? w.p[j] * is.p[j] : is.p[j];
      }
      k1.p[j] = kk;
    }
    if (RAW && grad_out && blockIdx.y == 0 && ty == 0) {
      const Pack8 g0 = rg0, g1 = rg1;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        f32x2 dwt = g1.p[j];
        if (abs_gamma && weight) dwt = f32x2{w.p[j].x < 0.f ? -dwt.x : dwt.x, w.p[j].y < 0.f ? -dwt.y : dwt.y};
        *reinterpret_cast<f32x2*>(grad_out + coff + 2 * j) = g0.p[j];
        *reinterpret_cast<f32x2*>(grad_out + C + coff + 2 * j) = dwt;
      }
    }
  }
  const int r_begin = blockIdx.y * rows_per_band;
  const int r_end = min(M, r_begin + rows_per_band);
  auto emit = [&](const uint4& xv, const uint4& gv, const uint4& yv, int r) {
    const Pack8 v = unpack8(xv), g = unpack8(gv);
    Pack8 yo, o, oz;
    if (YOUT) yo = unpack8(yv);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const f32x2 d = v.p[j] - mu.p[j];
      const f32x2 sgn = YOUT ? yo.p[j] : d * sc.p[j] + sh.p[j];
      const f32x2 dz = leaky_grad2(g.p[j], sgn, slope);
      const f32x2 xh = d * is.p[j];
      o.p[j] = (dz - k0.p[j] - xh * k1.p[j]) * gw.p[j];
      oz.p[j] = dz;
    }
    *reinterpret_cast<uint4*>(dx + (size_t)r * ld_dx + coff) = pack8(o);
    if (DZOUT) *reinterpret_cast<uint4*>(dz_out + (size_t)r * ld_dz + coff) = pack8(oz);
  };
  // (non-temporal loads of z and dy - both read here for the last time - measured level: 30.38 vs 30.35 ms; the forward apply keeps its)
  auto ld = [&](const __hip_bfloat16* p, int ldp, int r) { return *reinterpret_cast<const uint4*>(p + (size_t)r * ldp + coff); };
  int r = r_begin + ty;
  for (; r + 3 * TY < r_end; r += 4 * TY) {
    const uint4 v0 = ld(x, ld_x, r), g0 = ld(dy, ld_dy, r), v1 = ld(x, ld_x, r + TY), g1 = ld(dy, ld_dy, r + TY);
    const uint4 v2 = ld(x, ld_x, r + 2 * TY), g2 = ld(dy, ld_dy, r + 2 * TY), v3 = ld(x, ld_x, r + 3 * TY), g3 = ld(dy, ld_dy, r + 3 * TY);
    uint4 y0 = v0, y1 = v0, y2 = v0, y3 = v0;
    if (YOUT) { y0 = ld(yout, ld_y, r); y1 = ld(yout, ld_y, r + TY); y2 = ld(yout, ld_y, r + 2 * TY); y3 = ld(yout, ld_y, r + 3 * TY); }
    emit(v0, g0, y0, r);
    emit(v1, g1, y1, r + TY);
    emit(v2, g2, y2, r + 2 * TY);
    emit(v3, g3, y3, r + 3 * TY);
  }
  for (; r < r_end; r += TY) {
    const uint4 v = ld(x, ld_x, r), g = ld(dy, ld_dy, r);
    uint4 yo = v;
    if (YOUT) yo = ld(yout, ld_y, r);
    emit(v, g, yo, r);
  }
}

// ---- per-(image, channel) plane sums ------------------------------------------------------------
// grid: (channel tiles, B); the block's TY rows stride over the HW pixels of image b.
template <typename T>
__global__ __launch_bounds__(kBlock) void plane_sum_kernel(const T* __restrict__ x, int ld_x, int HW, int C, float alpha,
                                                          int TX, int TY, float* __restrict__ out) {
  constexpr int VEC = Vec<T>::N;
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int tx = threadIdx.x % TX, ty = threadIdx.x / TX;
  const int cg = blockIdx.x * TX + tx;
  const int b = blockIdx.y;
  const bool live = ty < TY && cg * VEC < C;
  float acc[VEC];
#pragma unroll
  for (int i = 0; i < VEC; ++i) acc[i] = 0.f;
  if (live) {
    const T* xp = x + (size_t)b * HW * ld_x + (size_t)cg * VEC;
    int r = ty;
    // four rows in flight per lane, added in row order (the same sums as one row at a time): a lane of the 2048-channel ASPP input
    // walks 272 rows, one exposed round trip each otherwise
    for (; r + 3 * TY < HW; r += 4 * TY) {
      Vec<T> v0, v1, v2, v3;
      v0.load(xp + (size_t)r * ld_x);
      v1.load(xp + (size_t)(r + TY) * ld_x);
      v2.load(xp + (size_t)(r + 2 * TY) * ld_x);
      v3.load(xp + (size_t)(r + 3 * TY) * ld_x);
#pragma unroll
      for (int i = 0; i < VEC; ++i) acc[i] = (((acc[i] + v0.get(i)) + v1.get(i)) + v2.get(i)) + v3.get(i);
    }
    for (; r < HW; r += TY) {
      Vec<T> v;
      v.load(xp + (size_t)r * ld_x);
#pragma unroll
      for (int i = 0; i < VEC; ++i) acc[i] += v.get(i);
    }
  }
  block_reduce_rows<VEC>(acc, tx, ty, TX, TY, lds);
  if (ty == 0 && cg * VEC < C) {
#pragma unroll
    for (int i = 0; i < VEC; ++i) out[(size_t)b * C + cg * VEC + i] = acc[i] * alpha;
  }
}

// ---- sliding-window mean (the image-pooling branch of DeeplabV3 in evaluation mode, modules/deeplab.py:77-83) --------
// out[b, oy, ox, c] = mean over the ph x pw window at (oy, ox), stride 1, no padding (avg_pool2d's "valid" output); the
// replicate padding back to the input size is the caller's (it commutes with the pointwise layers that follow).
// Separable running sums: pass 1 walks every column down (one thread per (b, x, 8/4-channel vector): add the entering row,
// drop the leaving one), pass 2 walks the rows of the intermediate across.  One read of the map instead of ph*pw.
template <typename T>
__global__ __launch_bounds__(kBlock) void window_vsum_kernel(const T* __restrict__ x, int ld_x, int B, int H, int W, int C,
                                                            int ph, float* __restrict__ tmp) {
  constexpr int VEC = Vec<T>::N;
  const int CG = C / VEC, OH = H - ph + 1;
  const size_t idx = (size_t)blockIdx.x * kBlock + threadIdx.x;
  if (idx >= (size_t)B * W * CG) return;
  const int cg = idx % CG, xx = (idx / CG) % W, b = idx / ((size_t)CG * W);
  const T* col = x + ((size_t)b * H * W + xx) * ld_x + (size_t)cg * VEC;
  float acc[VEC];
#pragma unroll
  for (int i = 0; i < VEC; ++i) acc[i] = 0.f;
  for (int y = 0; y < H; ++y) {
    Vec<T> v;
    v.load(col + (size_t)y * W * ld_x);
#pragma unroll
    for (int i = 0; i < VEC; ++i) acc[i] += v.get(i);
    if (y >= ph) {
      Vec<T> o;
      o.load(col + (size_t)(y - ph) * W * ld_x);
#pragma unroll
      for (int i = 0; i < VEC; ++i) acc[i] -= o.get(i);
    }
    if (y >= ph - 1) {
      float* dst = tmp + (((size_t)b * OH + (y - ph + 1)) * W + xx) * C + (size_t)cg * VEC;
#pragma unroll
      for (int i = 0; i < VEC; ++i) dst[i] = acc[i];
    }
  }
}

template <typename T>
__global__ __launch_bounds__(kBlock) void window_hsum_kernel(const float* __restrict__ tmp, int B, int OH, int W, int C, int pw,
                                                            float scale, T* __restrict__ out, int ld_o) {
  constexpr int VEC = Vec<T>::N;
  const int CG = C / VEC, OW = W - pw + 1;
  const size_t idx = (size_t)blockIdx.x * kBlock + threadIdx.x;
  if (idx >= (size_t)B * OH * CG) return;
  const int cg = idx % CG;
  const size_t line = idx / CG;                      // (b, oy)
  const float* row = tmp + line * W * C + (size_t)cg * VEC;
  float acc[VEC];
#pragma unroll
  for (int i = 0; i < VEC; ++i) acc[i] = 0.f;
  for (int xx = 0; xx < W; ++xx) {
#pragma unroll
    for (int i = 0; i < VEC; ++i) acc[i] += row[(size_t)xx * C + i];
    if (xx >= pw) {
#pragma unroll
      for (int i = 0; i < VEC; ++i) acc[i] -= row[(size_t)(xx - pw) * C + i];
    }
    if (xx >= pw - 1) {
      Vec<T> o;
#pragma unroll
      for (int i = 0; i < VEC; ++i) o.set(i, acc[i] * scale);
      o.store(out + (line * OW + (xx - pw + 1)) * ld_o + (size_t)cg * VEC);
    }
  }
}

// ---- attention map (segmentation_module.py:86-94) -----------------------------------------------
// pass 1: a[p] = sum_c x[p,c]^2 (one wave per pixel row), pass 2: per-image sum of a^2, pass 3: y = x*a/||a||
template <typename T>
__global__ __launch_bounds__(kBlock) void attmap_rowsq_kernel(const T* __restrict__ x, int ld_x, int M, int C,
                                                             float* __restrict__ a) {
  constexpr int VEC = Vec<T>::N;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int r = blockIdx.x * (kBlock / 64) + wave;
  if (r >= M) return;
  float s = 0.f;
  for (int c = lane * VEC; c < C; c += 64 * VEC) {
    Vec<T> v;
    v.load(x + (size_t)r * ld_x + c);
#pragma unroll
    for (int i = 0; i < VEC; ++i) s += v.get(i) * v.get(i);
  }
  s = wave_sum(s);
  if (lane == 0) a[r] = s;
}
__global__ __launch_bounds__(kBlock) void attmap_norm_kernel(const float* __restrict__ a, int HW, float* __restrict__ inv) {
  __shared__ float lds[kBlock / 64];
  const int b = blockIdx.x;
  float s = 0.f;
  for (int p = threadIdx.x; p < HW; p += kBlock) {
    float v = a[(size_t)b * HW + p];
    s += v * v;
  }
  s = wave_sum(s);
  if ((threadIdx.x & 63) == 0) lds[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) {
    float t = 0.f;
    for (int i = 0; i < kBlock / 64; ++i) t += lds[i];
    inv[b] = 1.f / sqrtf(t);
  }
}
template <typename T>
__global__ __launch_bounds__(kBlock) void attmap_scale_kernel(const T* __restrict__ x, int ld_x, T* __restrict__ y, int ld_y,
                                                             int M, int HW, int C, const float* __restrict__ a,
                                                             const float* __restrict__ inv) {
  constexpr int VEC = Vec<T>::N;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int r = blockIdx.x * (kBlock / 64) + wave;
  if (r >= M) return;
  const float f = a[r] * inv[r / HW];
  for (int c = lane * VEC; c < C; c += 64 * VEC) {
    Vec<T> v, o;
    v.load(x + (size_t)r * ld_x + c);
#pragma unroll
    for (int i = 0; i < VEC; ++i) o.set(i, v.get(i) * f);
    o.store(y + (size_t)r * ld_y + c);
  }
}

// ---- argument checks ----------------------------------------------------------------------------
int check_act_tensor(const char* fn, const char* name, const void* p, int ld, int dtype, int C, bool optional) {
  if (!p) {
    UCD_REQUIRE(optional, UCD_EINVAL, "%s: %s is NULL", fn, name);
    return 0;
  }
  const int es = dtype == UCD_BF16 ? 2 : 4;
  UCD_REQUIRE(aligned16(p) && ((size_t)ld * es) % 16 == 0, UCD_EALIGN,
              "%s: %s must be 16-byte aligned with a 16-byte multiple row pitch (ld=%d)", fn, name, ld);
  UCD_REQUIRE(ld >= C, UCD_EINVAL, "%s: ld of %s (%d) < C (%d)", fn, name, ld, C);
  return 0;
}
int check_common(const char* fn, int dtype, int M, int C, int act) {
  UCD_REQUIRE(dtype == UCD_F32 || dtype == UCD_BF16, UCD_EINVAL, "%s: unknown dtype %d", fn, dtype);
  UCD_REQUIRE(M > 0 && C > 0, UCD_EINVAL, "%s: empty tensor (M=%d, C=%d)", fn, M, C);
  const int vec = dtype == UCD_BF16 ? 8 : 4;
  UCD_REQUIRE(C % vec == 0, UCD_EALIGN, "%s: C=%d is not a multiple of %d", fn, C, vec);
  UCD_REQUIRE((act & ~(UCD_ACT_MASK | UCD_NORM_ABS_GAMMA)) == 0 && (act & UCD_ACT_MASK) <= UCD_ACT_ELU, UCD_EINVAL,
              "%s: unknown activation / flags 0x%x", fn, act);
  return 0;
}

// bf16 tensors without a plane bias under leaky_relu / identity take the packed-math kernels (UCD_ABN_GENERIC=1: the per-element
// kernels everywhere - the A/B switch of tools/abn_bench.py)
inline bool fast_path(int dtype, const float* plane_bias, int act_kind) {
  static const bool generic = getenv("UCD_ABN_GENERIC") != nullptr && getenv("UCD_ABN_GENERIC")[0] == '1';
  return !generic && dtype == UCD_BF16 && !plane_bias && act_kind != UCD_ACT_ELU;
}

#define UCD_TRY(expr)          \
  do {                         \
    int _rc = (expr);          \
    if (_rc) return _rc;       \
  } while (0)

}  // namespace
}  // namespace ucd

using namespace ucd;

extern "C" {

size_t ucd_abn_workspace_bytes(int M, int C) {
  (void)M;
  return (size_t)kMaxBands * 2 * (size_t)C * sizeof(float);
}

static int abn_stats_impl(const void* x, int ld_x, int dtype, int M, int C, const float* plane_bias, int HW, float* sums,
                          float* kshift, void* workspace, size_t workspace_bytes, ucd_stream_t stream,
                          const FinalizeArgs* fin) {
  static const char* fn = "ucd_abn_stats";
  UCD_TRY(check_common(fn, dtype, M, C, UCD_ACT_IDENTITY));
  UCD_TRY(check_act_tensor(fn, "x", x, ld_x, dtype, C, false));
  UCD_REQUIRE(sums && kshift && workspace, UCD_EINVAL, "%s: sums/kshift/workspace is NULL", fn);
  UCD_REQUIRE(!plane_bias || HW > 0, UCD_EINVAL, "%s: plane_bias needs HW > 0", fn);
  hipStream_t s = (hipStream_t)stream;
  float* partial = (float*)workspace;
  Geom g;
  if (dtype == UCD_BF16) {
    g = make_geom<8>(M, C, 512, 8, kMaxBands);
    UCD_REQUIRE(workspace_bytes >= (size_t)g.gy * 2 * C * 4, UCD_EWORKSPACE, "%s: workspace too small", fn);
    abn_stats_kernel<__hip_bfloat16><<<dim3(g.gx, g.gy), kBlock, kBlock * 16 * 4, s>>>(
        (const __hip_bfloat16*)x, ld_x, M, C, plane_bias, HW, g.TX, g.TY, g.rows_per_band, partial, kshift);
  } else {
    g = make_geom<4>(M, C, 512, 8, kMaxBands);
    UCD_REQUIRE(workspace_bytes >= (size_t)g.gy * 2 * C * 4, UCD_EWORKSPACE, "%s: workspace too small", fn);
    abn_stats_kernel<float><<<dim3(g.gx, g.gy), kBlock, kBlock * 8 * 4, s>>>(
        (const float*)x, ld_x, M, C, plane_bias, HW, g.TX, g.TY, g.rows_per_band, partial, kshift);
  }
  UCD_TRY(check_launch(fn));
  if (fin && fin->pack)
    UCD_REDUCE_BANDS(2, g.gy, partial, g.gy, C, sums, *fin);
  else if (fin)
    UCD_REDUCE_BANDS(1, g.gy, partial, g.gy, C, sums, *fin);
  else
    UCD_REDUCE_BANDS(0, g.gy, partial, g.gy, C, sums, FinalizeArgs{});
  return check_launch(fn);
}

int ucd_abn_stats(const void* x, int ld_x, int dtype, int M, int C, const float* plane_bias, int HW, float* sums,
                  float* kshift, void* workspace, size_t workspace_bytes, ucd_stream_t stream) {
  return abn_stats_impl(x, ld_x, dtype, M, C, plane_bias, HW, sums, kshift, workspace, workspace_bytes, stream, nullptr);
}

int ucd_abn_stats_finalize(const void* x, int ld_x, int dtype, int M, int C, const float* plane_bias, int HW,
                           float* sums, float* kshift, const float* weight, float* running_mean, float* running_var,
                           float momentum, float eps, float* mean, float* invstd, float* scale, int flags,
                           void* workspace, size_t workspace_bytes, ucd_stream_t stream) {
  UCD_REQUIRE(mean && invstd && scale && kshift, UCD_EINVAL, "ucd_abn_stats_finalize: NULL output");
  FinalizeArgs fin{kshift, weight, running_mean, running_var, mean, invstd, scale, (float)M, momentum, eps, nullptr,
                   (flags & UCD_NORM_ABS_GAMMA) != 0};
  return abn_stats_impl(x, ld_x, dtype, M, C, plane_bias, HW, sums, kshift, workspace, workspace_bytes, stream, &fin);
}

int ucd_abn_finalize(const float* sums, const float* kshift, float count, int C, const float* weight,
                     float* running_mean, float* running_var, float momentum, float eps, float* mean, float* invstd,
                     float* scale, int flags, ucd_stream_t stream) {
  static const char* fn = "ucd_abn_finalize";
  UCD_REQUIRE(sums && mean && invstd && scale && C > 0 && count > 0.f, UCD_EINVAL, "%s: bad arguments", fn);
  FinalizeArgs fin{kshift, weight, running_mean, running_var, mean, invstd, scale, count, momentum, eps, nullptr,
                   (flags & UCD_NORM_ABS_GAMMA) != 0};
  abn_finalize_kernel<<<ceil_div(C, 256), 256, 0, (hipStream_t)stream>>>(sums, C, fin);
  return check_launch(fn);
}

int ucd_abn_eval_params(const float* weight, const float* running_var, float eps, int C, float* invstd, float* scale,
                        int flags, ucd_stream_t stream) {
  static const char* fn = "ucd_abn_eval_params";
  UCD_REQUIRE(running_var && invstd && scale && C > 0, UCD_EINVAL, "%s: bad arguments", fn);
  abn_eval_params_kernel<<<ceil_div(C, 256), 256, 0, (hipStream_t)stream>>>(weight, running_var, eps, C, invstd, scale,
                                                                            (flags & UCD_NORM_ABS_GAMMA) != 0);
  return check_launch(fn);
}

int ucd_abn_apply(const void* x, int ld_x, void* y, int ld_y, const void* residual, int ld_r, int dtype, int M, int C,
                  const float* plane_bias, int HW, const float* mean, const float* scale, const float* shift, int act,
                  float slope, ucd_stream_t stream) {
  static const char* fn = "ucd_abn_apply";
  UCD_TRY(check_common(fn, dtype, M, C, act));
  UCD_TRY(check_act_tensor(fn, "x", x, ld_x, dtype, C, false));
  UCD_TRY(check_act_tensor(fn, "y", y, ld_y, dtype, C, false));
  UCD_TRY(check_act_tensor(fn, "residual", residual, ld_r, dtype, C, true));
  UCD_REQUIRE(mean && scale, UCD_EINVAL, "%s: mean/scale is NULL", fn);
  UCD_REQUIRE(!plane_bias || HW > 0, UCD_EINVAL, "%s: plane_bias needs HW > 0", fn);
  hipStream_t s = (hipStream_t)stream;
#define LAUNCH_APPLY(T, VECN, ACT)                                                                             \
  {                                                                                                            \
    Geom g = make_geom<VECN>(M, C, 512, 4);                                                                 \
    abn_apply_kernel<T, ACT><<<dim3(g.gx, g.gy), kBlock, 0, s>>>((const T*)x, ld_x, (T*)y, ld_y,               \
                                                                 (const T*)residual, ld_r, M, C, plane_bias,   \
                                                                 HW, mean, scale, shift, slope, g.TX, g.TY,    \
                                                                 g.rows_per_band);                             \
  }
  const int a = act & UCD_ACT_MASK;
  if (fast_path(dtype, plane_bias, a) && aligned16(mean) && aligned16(scale) && (!shift || aligned16(shift))) {
    const Geom g = make_geom<8>(M, C, 512, 4);
    const float sl = a == UCD_ACT_LEAKY_RELU ? slope : 1.f;
    typedef __hip_bfloat16 B;
    if (residual)
      abn_apply_fast_kernel<true><<<dim3(g.gx, g.gy), kBlock, 0, s>>>((const B*)x, ld_x, (B*)y, ld_y, (const B*)residual, ld_r, M, C,
                                                                      mean, scale, shift, sl, g.TX, g.TY, g.rows_per_band);
    else
      abn_apply_fast_kernel<false><<<dim3(g.gx, g.gy), kBlock, 0, s>>>((const B*)x, ld_x, (B*)y, ld_y, nullptr, 0, M, C, mean, scale,
                                                                       shift, sl, g.TX, g.TY, g.rows_per_band);
    return check_launch(fn);
  }
  if (dtype == UCD_BF16) {
    if (a == UCD_ACT_LEAKY_RELU) LAUNCH_APPLY(__hip_bfloat16, 8, UCD_ACT_LEAKY_RELU)
    else if (a == UCD_ACT_ELU) LAUNCH_APPLY(__hip_bfloat16, 8, UCD_ACT_ELU)
    else LAUNCH_APPLY(__hip_bfloat16, 8, UCD_ACT_IDENTITY)
  } else {
    if (a == UCD_ACT_LEAKY_RELU) LAUNCH_APPLY(float, 4, UCD_ACT_LEAKY_RELU)
    else if (a == UCD_ACT_ELU) LAUNCH_APPLY(float, 4, UCD_ACT_ELU)
    else LAUNCH_APPLY(float, 4, UCD_ACT_IDENTITY)
  }
#undef LAUNCH_APPLY
  return check_launch(fn);
}

int ucd_abn_apply_stats(const void* x, int ld_x, void* y, int ld_y, const void* residual, int ld_r, int M, int C,
                        const float* acc, int reps, const float* kshift, float count, const float* weight, const float* bias,
                        float* running_mean, float* running_var, float momentum, float eps, float* mean, float* invstd,
                        float* scale, int act, float slope, ucd_stream_t stream) {
  static const char* fn = "ucd_abn_apply_stats";
  if (reps < 1) reps = 1;
  UCD_TRY(check_common(fn, UCD_BF16, M, C, act));
  UCD_TRY(check_act_tensor(fn, "x", x, ld_x, UCD_BF16, C, false));
  UCD_TRY(check_act_tensor(fn, "y", y, ld_y, UCD_BF16, C, false));
  UCD_TRY(check_act_tensor(fn, "residual", residual, ld_r, UCD_BF16, C, true));
  const int a = act & UCD_ACT_MASK;
  UCD_REQUIRE(a != UCD_ACT_ELU, UCD_EUNSUPPORTED, "%s: leaky_relu / identity only", fn);
  UCD_REQUIRE(acc && kshift && mean && invstd && scale && count > 0.f, UCD_EINVAL, "%s: NULL statistics argument", fn);
  UCD_REQUIRE(aligned16(acc) && aligned16(kshift) && (C % 4 == 0) && (!weight || aligned16(weight)) && (!bias || aligned16(bias)), UCD_EALIGN,
              "%s: per-channel vectors must be 16-byte aligned", fn);
  const Geom g = make_geom<8>(M, C, 512, 4);
  const float sl = a == UCD_ACT_LEAKY_RELU ? slope : 1.f;
  const ApplyFin fin{acc, reps, kshift, weight, running_mean, running_var, mean, invstd, scale, count, momentum, eps,
                     (act & UCD_NORM_ABS_GAMMA) != 0};
  typedef __hip_bfloat16 B;
  hipStream_t s = (hipStream_t)stream;
  if (residual)
    abn_apply_fast_kernel<true, true><<<dim3(g.gx, g.gy), kBlock, kBlock * 16 * 4, s>>>((const B*)x, ld_x, (B*)y, ld_y, (const B*)residual, ld_r, M, C,
                                                                          nullptr, nullptr, bias, sl, g.TX, g.TY, g.rows_per_band, fin);
  else
    abn_apply_fast_kernel<false, true><<<dim3(g.gx, g.gy), kBlock, kBlock * 16 * 4, s>>>((const B*)x, ld_x, (B*)y, ld_y, nullptr, 0, M, C, nullptr,
                                                                           nullptr, bias, sl, g.TX, g.TY, g.rows_per_band, fin);
  return check_launch(fn);
}

int ucd_abn_bwd_apply_raw(const void* x, int ld_x, const void* dy, int ld_dy, const void* y, int ld_y, void* dx, int ld_dx,
                          void* dz_out, int ld_dz, int M, int C, const float* mean, const float* invstd, const float* scale,
                          const float* shift, const float* weight, const float* sums, const float* grad_sums, int reps,
                          float* grad_out, float count, int act, float slope, ucd_stream_t stream) {
  static const char* fn = "ucd_abn_bwd_apply_raw";
  if (reps < 1) reps = 1;
  UCD_TRY(check_common(fn, UCD_BF16, M, C, act));
  UCD_TRY(check_act_tensor(fn, "x", x, ld_x, UCD_BF16, C, false));
  UCD_TRY(check_act_tensor(fn, "dy", dy, ld_dy, UCD_BF16, C, false));
  UCD_TRY(check_act_tensor(fn, "y", y, ld_y, UCD_BF16, C, true));
  UCD_TRY(check_act_tensor(fn, "dx", dx, ld_dx, UCD_BF16, C, false));
  UCD_TRY(check_act_tensor(fn, "dz_out", dz_out, ld_dz, UCD_BF16, C, true));
  const int a = act & UCD_ACT_MASK;
  UCD_REQUIRE(a != UCD_ACT_ELU, UCD_EUNSUPPORTED, "%s: leaky_relu / identity only", fn);
  UCD_REQUIRE(mean && invstd && scale && sums && count > 0.f, UCD_EINVAL, "%s: training statistics missing", fn);
  UCD_REQUIRE(aligned16(mean) && aligned16(invstd) && aligned16(scale) && aligned16(sums) && C % 4 == 0 && (!shift || aligned16(shift)) &&
                  (!weight || aligned16(weight)) && (!grad_sums || aligned16(grad_sums)) && (!grad_out || aligned16(grad_out)),
              UCD_EALIGN, "%s: per-channel vectors must be 16-byte aligned", fn);
  typedef __hip_bfloat16 B;
  const Geom g = make_geom<8>(M, C, 512, 4);
  const float sl = a == UCD_ACT_LEAKY_RELU ? slope : 1.f;
  const int ag = (act & UCD_NORM_ABS_GAMMA) != 0;
  const float* gs = grad_sums ? grad_sums : sums;
  hipStream_t s = (hipStream_t)stream;
#define UCD_BWD_RAW(YO, DZ)                                                                                                        \
  abn_bwd_apply_fast_kernel<YO, DZ, true><<<dim3(g.gx, g.gy), kBlock, kBlock * 16 * 4, s>>>((const B*)x, ld_x, (const B*)dy, ld_dy, (const B*)y, ld_y, \
                                                                              (B*)dx, ld_dx, (B*)dz_out, ld_dz, M, C, mean, invstd, scale, \
                                                                              shift, weight, sums, 1.f / count, 0, ag, sl, g.TX, g.TY, \
                                                                              g.rows_per_band, gs, grad_out, reps)
  if (y) { if (dz_out) UCD_BWD_RAW(true, true); else UCD_BWD_RAW(true, false); }
  else { if (dz_out) UCD_BWD_RAW(false, true); else UCD_BWD_RAW(false, false); }
#undef UCD_BWD_RAW
  return check_launch(fn);
}

static int bwd_reduce_impl(const char* fn, const void* x, int ld_x, const void* dy, int ld_dy, const void* y, int ld_y,
                           int dtype, int M, int C, const float* plane_bias, int HW, const float* mean,
                           const float* invstd, const float* scale, const float* shift, const float* weight, int act,
                           float slope, float* sums, float* sums_copy, void* workspace, size_t workspace_bytes,
                           ucd_stream_t stream) {
  UCD_TRY(check_common(fn, dtype, M, C, act));
  UCD_TRY(check_act_tensor(fn, "x", x, ld_x, dtype, C, false));
  UCD_TRY(check_act_tensor(fn, "dy", dy, ld_dy, dtype, C, false));
  UCD_TRY(check_act_tensor(fn, "y", y, ld_y, dtype, C, true));
  UCD_REQUIRE(mean && invstd && scale && sums && workspace, UCD_EINVAL, "%s: NULL argument", fn);
  UCD_REQUIRE(!plane_bias || HW > 0, UCD_EINVAL, "%s: plane_bias needs HW > 0", fn);
  hipStream_t s = (hipStream_t)stream;
  float* partial = (float*)workspace;
  Geom g;
#define LAUNCH_RED(T, VECN, ACT)                                                                                   \
  {                                                                                                                \
    g = make_geom<VECN>(M, C, 512, 8, kMaxBands);                                                                \
    UCD_REQUIRE(workspace_bytes >= (size_t)g.gy * 2 * C * 4, UCD_EWORKSPACE, "%s: workspace too small", fn);       \
    abn_bwd_reduce_kernel<T, ACT><<<dim3(g.gx, g.gy), kBlock, kBlock * 2 * VECN * 4, s>>>(                         \
        (const T*)x, ld_x, (const T*)dy, ld_dy, (const T*)y, ld_y, M, C, plane_bias, HW, mean, invstd, scale,     \
        shift, slope, g.TX, g.TY, g.rows_per_band, partial);                                                       \
  }
  const int a = act & UCD_ACT_MASK;
  if (fast_path(dtype, plane_bias, a) && aligned16(mean) && aligned16(invstd) && aligned16(scale) && (!shift || aligned16(shift))) {
    typedef __hip_bfloat16 B;
    g = make_geom<8>(M, C, 512, 8, kMaxBands);
    UCD_REQUIRE(workspace_bytes >= (size_t)g.gy * 2 * C * 4, UCD_EWORKSPACE, "%s: workspace too small", fn);
    const float sl = a == UCD_ACT_LEAKY_RELU ? slope : 1.f;
    if (y)
      abn_bwd_reduce_fast_kernel<true><<<dim3(g.gx, g.gy), kBlock, kBlock * 16 * 4, s>>>(
          (const B*)x, ld_x, (const B*)dy, ld_dy, (const B*)y, ld_y, M, C, mean, invstd, scale, shift, sl, g.TX, g.TY, g.rows_per_band, partial);
    else
      abn_bwd_reduce_fast_kernel<false><<<dim3(g.gx, g.gy), kBlock, kBlock * 16 * 4, s>>>(
          (const B*)x, ld_x, (const B*)dy, ld_dy, nullptr, 0, M, C, mean, invstd, scale, shift, sl, g.TX, g.TY, g.rows_per_band, partial);
  } else if (dtype == UCD_BF16) {
    if (a == UCD_ACT_LEAKY_RELU) LAUNCH_RED(__hip_bfloat16, 8, UCD_ACT_LEAKY_RELU)
    else if (a == UCD_ACT_ELU) LAUNCH_RED(__hip_bfloat16, 8, UCD_ACT_ELU)
    else LAUNCH_RED(__hip_bfloat16, 8, UCD_ACT_IDENTITY)
  } else {
    if (a == UCD_ACT_LEAKY_RELU) LAUNCH_RED(float, 4, UCD_ACT_LEAKY_RELU)
    else if (a == UCD_ACT_ELU) LAUNCH_RED(float, 4, UCD_ACT_ELU)
    else LAUNCH_RED(float, 4, UCD_ACT_IDENTITY)
  }
#undef LAUNCH_RED
  UCD_TRY(check_launch(fn));
  UCD_REDUCE_BANDS(0, g.gy, partial, g.gy, C, sums, FinalizeArgs{}, sums_copy, (act & UCD_NORM_ABS_GAMMA) ? weight : nullptr);
  return check_launch(fn);
}

int ucd_abn_bwd_reduce(const void* x, int ld_x, const void* dy, int ld_dy, const void* y, int ld_y, int dtype, int M,
                       int C, const float* plane_bias, int HW, const float* mean, const float* invstd,
                       const float* scale, const float* shift, const float* weight, int act, float slope, float* sums,
                       void* workspace, size_t workspace_bytes, ucd_stream_t stream) {
  return bwd_reduce_impl("ucd_abn_bwd_reduce", x, ld_x, dy, ld_dy, y, ld_y, dtype, M, C, plane_bias, HW, mean, invstd, scale,
                         shift, weight, act, slope, sums, nullptr, workspace, workspace_bytes, stream);
}

int ucd_abn_reduce_partials(const float* partial, int tiles, int C, float* sums, float* sums_copy, const float* weight,
                            int flags, ucd_stream_t stream) {
  static const char* fn = "ucd_abn_reduce_partials";
  UCD_REQUIRE(partial && sums && tiles > 0 && C > 0, UCD_EINVAL, "%s: bad arguments", fn);
  hipStream_t s = (hipStream_t)stream;
  UCD_REDUCE_BANDS(0, tiles, partial, tiles, C, sums, FinalizeArgs{}, sums_copy, (flags & UCD_NORM_ABS_GAMMA) ? weight : nullptr);
  return check_launch(fn);
}

// ---- SyncBN (one process per GPU): the three library calls around the two collectives of a layer ----
// forward:  ucd_abn_sync_stats -> all_gather(pack) -> ucd_abn_sync_forward
// backward: ucd_abn_sync_bwd_reduce -> all_reduce(sums) -> ucd_abn_bwd_apply
int ucd_abn_sync_stats(const void* x, int ld_x, int dtype, int M, int C, const float* plane_bias, int HW, float* sums,
                       float* kshift, float* pack, void* workspace, size_t workspace_bytes, ucd_stream_t stream) {
  UCD_REQUIRE(sums && kshift && pack, UCD_EINVAL, "ucd_abn_sync_stats: NULL output");
  FinalizeArgs fin{kshift, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, (float)M, 0.f, 0.f, pack, 0};
  return abn_stats_impl(x, ld_x, dtype, M, C, plane_bias, HW, sums, kshift, workspace, workspace_bytes, stream, &fin);
}

int ucd_abn_sync_forward(const void* x, int ld_x, void* y, int ld_y, const void* residual, int ld_r, int dtype, int M, int C,
                         const float* plane_bias, int HW, const float* gathered, int world, const float* weight,
                         const float* bias, float* running_mean, float* running_var, float momentum, float eps, float* buf,
                         int act, float slope, ucd_stream_t stream) {
  static const char* fn = "ucd_abn_sync_forward";
  UCD_REQUIRE(gathered && buf && world >= 1 && C > 0 && M > 0, UCD_EINVAL, "%s: bad arguments", fn);
  float *mean = buf + 3 * C, *invstd = buf + 4 * C, *scale = buf + 5 * C;
  FinalizeArgs fin{nullptr, weight, running_mean, running_var, mean, invstd, scale, (float)M * (float)world, momentum, eps,
                   nullptr, (act & UCD_NORM_ABS_GAMMA) != 0};
  abn_combine_finalize_kernel<<<ceil_div(C, 256), 256, 0, (hipStream_t)stream>>>(gathered, world, C, (float)M, fin);
  UCD_TRY(check_launch(fn));
  return ucd_abn_apply(x, ld_x, y, ld_y, residual, ld_r, dtype, M, C, plane_bias, HW, mean, scale, bias, act, slope, stream);
}

int ucd_abn_sync_finalize(const float* gathered, int world, int M, int C, const float* weight, float* running_mean,
                          float* running_var, float momentum, float eps, float* buf, int flags, ucd_stream_t stream) {
  static const char* fn = "ucd_abn_sync_finalize";
  UCD_REQUIRE(gathered && buf && world >= 1 && C > 0 && M > 0, UCD_EINVAL, "%s: bad arguments", fn);
  float *mean = buf + 3 * C, *invstd = buf + 4 * C, *scale = buf + 5 * C;
  FinalizeArgs fin{nullptr, weight, running_mean, running_var, mean, invstd, scale, (float)M * (float)world, momentum, eps,
                   nullptr, (flags & UCD_NORM_ABS_GAMMA) != 0};
  abn_combine_finalize_kernel<<<ceil_div(C, 256), 256, 0, (hipStream_t)stream>>>(gathered, world, C, (float)M, fin);
  return check_launch(fn);
}

int ucd_abn_sync_forward_comm(ucd_comm_t comm, int world, const void* x, int ld_x, void* y, int ld_y, const void* residual,
                              int ld_r, int dtype, int M, int C, const float* plane_bias, int HW, const float* weight,
                              const float* bias, float* running_mean, float* running_var, float momentum, float eps,
                              float* buf, int act, float slope, void* workspace, size_t workspace_bytes,
                              ucd_stream_t stream) {
  UCD_REQUIRE(comm && buf && world >= 1, UCD_EINVAL, "ucd_abn_sync_forward_comm: bad arguments");
  float *pack = buf + 6 * C, *gathered = buf + 8 * C;
  UCD_TRY(ucd_abn_sync_stats(x, ld_x, dtype, M, C, plane_bias, HW, buf, buf + 2 * C, pack, workspace, workspace_bytes, stream));
  UCD_TRY(comm_all_gather_f32(comm, pack, gathered, (size_t)2 * C, (hipStream_t)stream));
  return ucd_abn_sync_forward(x, ld_x, y, ld_y, residual, ld_r, dtype, M, C, plane_bias, HW, gathered, world, weight, bias,
                              running_mean, running_var, momentum, eps, buf, act, slope, stream);
}

int ucd_abn_sync_backward_comm(ucd_comm_t comm, int world, const void* x, int ld_x, const void* dy, int ld_dy, const void* y,
                               int ld_y, void* dx, int ld_dx, void* dz_out, int ld_dz, int dtype, int M, int C,
                               const float* plane_bias, int HW, const float* mean, const float* invstd, const float* scale,
                               const float* bias, const float* weight, float* sums, float* local_sums, int act, float slope,
                               void* workspace, size_t workspace_bytes, ucd_stream_t stream) {
  UCD_REQUIRE(comm && sums && local_sums && world >= 1, UCD_EINVAL, "ucd_abn_sync_backward_comm: bad arguments");
  UCD_TRY(ucd_abn_sync_bwd_reduce(x, ld_x, dy, ld_dy, y, ld_y, dtype, M, C, plane_bias, HW, mean, invstd, scale, bias, weight,
                                  act, slope, sums, local_sums, workspace, workspace_bytes, stream));
  UCD_TRY(comm_all_reduce_sum_f32(comm, sums, (size_t)2 * C, (hipStream_t)stream));
  return ucd_abn_bwd_apply(x, ld_x, dy, ld_dy, y, ld_y, dx, ld_dx, dz_out, ld_dz, dtype, M, C, plane_bias, HW, mean, invstd,
                           scale, bias, weight, sums, (float)M * (float)world, 0, act, slope, stream);
}

int ucd_abn_sync_bwd_reduce(const void* x, int ld_x, const void* dy, int ld_dy, const void* y, int ld_y, int dtype, int M,
                            int C, const float* plane_bias, int HW, const float* mean, const float* invstd,
                            const float* scale, const float* shift, const float* weight, int act, float slope,
                            float* sums, float* local_sums, void* workspace, size_t workspace_bytes, ucd_stream_t stream) {
  return bwd_reduce_impl("ucd_abn_sync_bwd_reduce", x, ld_x, dy, ld_dy, y, ld_y, dtype, M, C, plane_bias, HW, mean, invstd,
                         scale, shift, weight, act, slope, sums, local_sums, workspace, workspace_bytes, stream);
}

int ucd_abn_bwd_apply(const void* x, int ld_x, const void* dy, int ld_dy, const void* y, int ld_y, void* dx, int ld_dx,
                      void* dz_out, int ld_dz, int dtype, int M, int C, const float* plane_bias, int HW,
                      const float* mean, const float* invstd, const float* scale, const float* shift,
                      const float* weight, const float* sums, float count, int frozen, int act, float slope,
                      ucd_stream_t stream) {
  static const char* fn = "ucd_abn_bwd_apply";
  UCD_TRY(check_common(fn, dtype, M, C, act));
  UCD_TRY(check_act_tensor(fn, "x", x, ld_x, dtype, C, false));
  UCD_TRY(check_act_tensor(fn, "dy", dy, ld_dy, dtype, C, false));
  UCD_TRY(check_act_tensor(fn, "y", y, ld_y, dtype, C, true));
  UCD_TRY(check_act_tensor(fn, "dx", dx, ld_dx, dtype, C, false));
  UCD_TRY(check_act_tensor(fn, "dz_out", dz_out, ld_dz, dtype, C, true));
  UCD_REQUIRE(scale && mean, UCD_EINVAL, "%s: scale/mean is NULL", fn);
  UCD_REQUIRE(frozen || (invstd && sums && count > 0.f), UCD_EINVAL, "%s: training statistics missing", fn);
  UCD_REQUIRE(!plane_bias || HW > 0, UCD_EINVAL, "%s: plane_bias needs HW > 0", fn);
  hipStream_t s = (hipStream_t)stream;
  const float inv_count = frozen ? 0.f : 1.f / count;
#define LAUNCH_BWD(T, VECN, ACT)                                                                                   \
  {                                                                                                                \
    Geom g = make_geom<VECN>(M, C, 512, 4);                                                                    \
    abn_bwd_apply_kernel<T, ACT><<<dim3(g.gx, g.gy), kBlock, 0, s>>>(                                              \
        (const T*)x, ld_x, (const T*)dy, ld_dy, (const T*)y, ld_y, (T*)dx, ld_dx, (T*)dz_out, ld_dz, M, C,        \
        plane_bias, HW, mean, invstd, scale, shift, weight, sums, inv_count, frozen,                              \
        (act & UCD_NORM_ABS_GAMMA) != 0, slope, g.TX, g.TY, g.rows_per_band);                                                                                          \
  }
  const int a = act & UCD_ACT_MASK;
  if (fast_path(dtype, plane_bias, a) && aligned16(mean) && aligned16(scale) && (!shift || aligned16(shift)) &&
      (frozen || (aligned16(invstd) && aligned16(sums) && C % 4 == 0 && (!weight || aligned16(weight))))) {
    typedef __hip_bfloat16 B;
    const Geom g = make_geom<8>(M, C, 512, 4);
    const float sl = a == UCD_ACT_LEAKY_RELU ? slope : 1.f;
    const int ag = (act & UCD_NORM_ABS_GAMMA) != 0;
#define UCD_BWD_FAST(YO, DZ)                                                                                                      \
  abn_bwd_apply_fast_kernel<YO, DZ><<<dim3(g.gx, g.gy), kBlock, 0, s>>>((const B*)x, ld_x, (const B*)dy, ld_dy, (const B*)y, ld_y, \
                                                                        (B*)dx, ld_dx, (B*)dz_out, ld_dz, M, C, mean, invstd, scale, \
                                                                        shift, weight, sums, inv_count, frozen, ag, sl, g.TX, g.TY, \
                                                                        g.rows_per_band)
    if (y) { if (dz_out) UCD_BWD_FAST(true, true); else UCD_BWD_FAST(true, false); }
    else { if (dz_out) UCD_BWD_FAST(false, true); else UCD_BWD_FAST(false, false); }
#undef UCD_BWD_FAST
    return check_launch(fn);
  }
  if (dtype == UCD_BF16) {
    if (a == UCD_ACT_LEAKY_RELU) LAUNCH_BWD(__hip_bfloat16, 8, UCD_ACT_LEAKY_RELU)
    else if (a == UCD_ACT_ELU) LAUNCH_BWD(__hip_bfloat16, 8, UCD_ACT_ELU)
    else LAUNCH_BWD(__hip_bfloat16, 8, UCD_ACT_IDENTITY)
  } else {
    if (a == UCD_ACT_LEAKY_RELU) LAUNCH_BWD(float, 4, UCD_ACT_LEAKY_RELU)
    else if (a == UCD_ACT_ELU) LAUNCH_BWD(float, 4, UCD_ACT_ELU)
    else LAUNCH_BWD(float, 4, UCD_ACT_IDENTITY)
  }
#undef LAUNCH_BWD
  return check_launch(fn);
}

// ---- one-call forward / backward (single process): fewer crossings of the Python/ctypes boundary, which is
// what bounds the step once the per-GPU batch is small (8-GPU regime: ~70 us of host time per layer call)
int ucd_abn_forward(const void* x, int ld_x, void* y, int ld_y, const void* residual, int ld_r, int dtype, int M, int C,
                    const float* plane_bias, int HW, const float* weight, const float* bias, float* running_mean,
                    float* running_var, float momentum, float eps, int training, float* buf, const float* eval_consts,
                    int act, float slope, void* workspace, size_t workspace_bytes, ucd_stream_t stream) {
  static const char* fn = "ucd_abn_forward";
  UCD_REQUIRE(running_mean && running_var, UCD_EINVAL, "%s: running statistics are NULL", fn);
  if (training) {
    UCD_REQUIRE(buf, UCD_EINVAL, "%s: buf is NULL", fn);
    float *sums = buf, *kshift = buf + 2 * C, *mean = buf + 3 * C, *invstd = buf + 4 * C, *scale = buf + 5 * C;
    UCD_TRY(ucd_abn_stats_finalize(x, ld_x, dtype, M, C, plane_bias, HW, sums, kshift, weight, running_mean, running_var,
                                   momentum, eps, mean, invstd, scale, act & UCD_NORM_ABS_GAMMA, workspace, workspace_bytes,
                                   stream));
    return ucd_abn_apply(x, ld_x, y, ld_y, residual, ld_r, dtype, M, C, plane_bias, HW, mean, scale, bias, act, slope, stream);
  }
  const float* scale = nullptr;
  if (eval_consts) {
    scale = eval_consts + C;   // [invstd | scale], computed once for a frozen layer
  } else {
    UCD_REQUIRE(buf, UCD_EINVAL, "%s: buf is NULL", fn);
    UCD_TRY(ucd_abn_eval_params(weight, running_var, eps, C, buf + 4 * C, buf + 5 * C, act & UCD_NORM_ABS_GAMMA, stream));
    scale = buf + 5 * C;
  }
  return ucd_abn_apply(x, ld_x, y, ld_y, residual, ld_r, dtype, M, C, plane_bias, HW, running_mean, scale, bias, act, slope,
                       stream);
}

int ucd_abn_backward(const void* x, int ld_x, const void* dy, int ld_dy, const void* y, int ld_y, void* dx, int ld_dx,
                     void* dz_out, int ld_dz, int dtype, int M, int C, const float* plane_bias, int HW, const float* mean,
                     const float* invstd, const float* scale, const float* bias, const float* weight, float* sums,
                     float count, int training, int need_sums, int act, float slope, void* workspace,
                     size_t workspace_bytes, ucd_stream_t stream) {
  if (training || need_sums)
    UCD_TRY(ucd_abn_bwd_reduce(x, ld_x, dy, ld_dy, y, ld_y, dtype, M, C, plane_bias, HW, mean, invstd, scale, bias, weight, act,
                               slope, sums, workspace, workspace_bytes, stream));
  return ucd_abn_bwd_apply(x, ld_x, dy, ld_dy, y, ld_y, dx, ld_dx, dz_out, ld_dz, dtype, M, C, plane_bias, HW, mean, invstd,
                           scale, bias, weight, sums, count, training ? 0 : 1, act, slope, stream);
}

int ucd_plane_sum(const void* x, int ld_x, int dtype, int B, int HW, int C, float alpha, float* out,
                  ucd_stream_t stream) {
  static const char* fn = "ucd_plane_sum";
  UCD_TRY(check_common(fn, dtype, B * HW, C, UCD_ACT_IDENTITY));
  UCD_TRY(check_act_tensor(fn, "x", x, ld_x, dtype, C, false));
  UCD_REQUIRE(out && B > 0 && HW > 0, UCD_EINVAL, "%s: bad arguments", fn);
  hipStream_t s = (hipStream_t)stream;
  if (dtype == UCD_BF16) {
    Geom g = make_geom<8>(HW, C, 1, 1);
    plane_sum_kernel<__hip_bfloat16><<<dim3(g.gx, B), kBlock, kBlock * 8 * 4, s>>>((const __hip_bfloat16*)x, ld_x, HW,
                                                                                  C, alpha, g.TX, g.TY, out);
  } else {
    Geom g = make_geom<4>(HW, C, 1, 1);
    plane_sum_kernel<float><<<dim3(g.gx, B), kBlock, kBlock * 4 * 4, s>>>((const float*)x, ld_x, HW, C, alpha, g.TX,
                                                                         g.TY, out);
  }
  return check_launch(fn);
}

size_t ucd_window_mean_workspace_bytes(int B, int H, int W, int C, int ph) {
  return (size_t)B * (size_t)(H - ph + 1) * W * C * sizeof(float);
}

int ucd_window_mean(const void* x, int ld_x, int dtype, int B, int H, int W, int C, int ph, int pw, void* out, int ld_out,
                    void* workspace, size_t workspace_bytes, ucd_stream_t stream) {
  static const char* fn = "ucd_window_mean";
  UCD_TRY(check_common(fn, dtype, B * H * W, C, UCD_ACT_IDENTITY));
  UCD_TRY(check_act_tensor(fn, "x", x, ld_x, dtype, C, false));
  UCD_TRY(check_act_tensor(fn, "out", out, ld_out, dtype, C, false));
  UCD_REQUIRE(ph >= 1 && pw >= 1 && ph <= H && pw <= W, UCD_EINVAL, "%s: window %dx%d does not fit the %dx%d map", fn, ph, pw, H, W);
  UCD_REQUIRE(workspace && workspace_bytes >= ucd_window_mean_workspace_bytes(B, H, W, C, ph), UCD_EWORKSPACE,
              "%s: workspace too small", fn);
  hipStream_t s = (hipStream_t)stream;
  const int OH = H - ph + 1;
  const float scale = 1.f / ((float)ph * (float)pw);
  float* tmp = (float*)workspace;
  if (dtype == UCD_BF16) {
    const size_t n1 = (size_t)B * W * (C / 8), n2 = (size_t)B * OH * (C / 8);
    window_vsum_kernel<__hip_bfloat16><<<(unsigned)((n1 + kBlock - 1) / kBlock), kBlock, 0, s>>>((const __hip_bfloat16*)x, ld_x, B, H, W,
                                                                                             C, ph, tmp);
    UCD_TRY(check_launch(fn));
    window_hsum_kernel<__hip_bfloat16><<<(unsigned)((n2 + kBlock - 1) / kBlock), kBlock, 0, s>>>(tmp, B, OH, W, C, pw, scale,
                                                                                             (__hip_bfloat16*)out, ld_out);
  } else {
    const size_t n1 = (size_t)B * W * (C / 4), n2 = (size_t)B * OH * (C / 4);
    window_vsum_kernel<float><<<(unsigned)((n1 + kBlock - 1) / kBlock), kBlock, 0, s>>>((const float*)x, ld_x, B, H, W, C, ph, tmp);
    UCD_TRY(check_launch(fn));
    window_hsum_kernel<float><<<(unsigned)((n2 + kBlock - 1) / kBlock), kBlock, 0, s>>>(tmp, B, OH, W, C, pw, scale, (float*)out,
                                                                                    ld_out);
  }
  return check_launch(fn);
}

size_t ucd_attmap_workspace_bytes(int B, int HW) { return ((size_t)B * HW + B) * sizeof(float); }

int ucd_attmap(const void* x, int ld_x, void* y, int ld_y, int dtype, int B, int HW, int C, void* workspace,
               size_t workspace_bytes, ucd_stream_t stream) {
  static const char* fn = "ucd_attmap";
  UCD_TRY(check_common(fn, dtype, B * HW, C, UCD_ACT_IDENTITY));
  UCD_TRY(check_act_tensor(fn, "x", x, ld_x, dtype, C, false));
  UCD_TRY(check_act_tensor(fn, "y", y, ld_y, dtype, C, false));
  UCD_REQUIRE(workspace && workspace_bytes >= ucd_attmap_workspace_bytes(B, HW), UCD_EWORKSPACE,
              "%s: workspace too small", fn);
  hipStream_t s = (hipStream_t)stream;
  const int M = B * HW;
  float* a = (float*)workspace;
  float* inv = a + M;
  const int rows_per_block = kBlock / 64;
  if (dtype == UCD_BF16)
    attmap_rowsq_kernel<__hip_bfloat16><<<ceil_div(M, rows_per_block), kBlock, 0, s>>>((const __hip_bfloat16*)x, ld_x, M, C, a);
  else
    attmap_rowsq_kernel<float><<<ceil_div(M, rows_per_block), kBlock, 0, s>>>((const float*)x, ld_x, M, C, a);
  UCD_TRY(check_launch(fn));
  attmap_norm_kernel<<<B, kBlock, 0, s>>>(a, HW, inv);
  UCD_TRY(check_launch(fn));
  if (dtype == UCD_BF16)
    attmap_scale_kernel<__hip_bfloat16><<<ceil_div(M, rows_per_block), kBlock, 0, s>>>(
        (const __hip_bfloat16*)x, ld_x, (__hip_bfloat16*)y, ld_y, M, HW, C, a, inv);
  else
    attmap_scale_kernel<float><<<ceil_div(M, rows_per_block), kBlock, 0, s>>>((const float*)x, ld_x, (float*)y, ld_y,
                                                                             M, HW, C, a, inv);
  return check_launch(fn);
}

}  // extern "C"
