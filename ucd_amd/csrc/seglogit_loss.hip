// Fused full-resolution logit losses: bilinear x16 up-sampling of the student and
// teacher logits (segmentation_module.py:133) + UnbiasedCrossEntropy (utils/loss.py:96-109) +
// UnbiasedKnowledgeDistillationLoss (utils/loss.py:148-184) + their gradient w.r.t. the LOW-resolution
// student logits, in one pass.  The reference materialises [B, Ctot, H, W] float32 logits for student and
// teacher (22 MB / image each at VOC 513^2, 158 MB at ADE) and then ~10 full-resolution temporaries per loss;
// here a full-resolution logit only ever exists in a register.
//
// Per full-resolution pixel: z_c = sum of 4 weighted low-res neighbours (torch's align_corners=False
// source-index arithmetic), three log-sum-exps of the student (all classes / old classes [0,K) /
// background + new classes), the teacher soft-max, the two losses and
//   dCE/dz_c = softmax(z)_c - [label is bkg/old] [c < K] exp(z_c - LSE_old) - [label new] [c == label]
//   dKD/dz_c = (softmax(z)_c - q_0 [c in {0} U [K,Ctot)] exp(z_c - LSE_bkgnew) - [1 <= c < K] q_c) / K
// scattered to the 4 neighbours with their bilinear weights.  A 256-thread block owns a 16 x 64 pixel
// tile; the <= 3 x 7 low-res cells under it are staged in LDS (logits in, gradient accumulators out, LDS
// float atomics), then flushed with one global float atomic per cell and class (the summation order of
// float atomics makes the last bits of the gradient run-dependent; the loss sums use a fixed order).
#include "common.h"

namespace ucd {
namespace {

constexpr int kThreads = 256;
constexpr int kTileY = 64, kTileX = 64;     // a thread walks kRows consecutive rows of one pixel column
constexpr int kRows = kTileY / 4;

__device__ __forceinline__ void up_src(int dst, int in_size, float scale, int& i0, int& i1, float& l0, float& l1) {
  float src = scale * ((float)dst + 0.5f) - 0.5f;
  src = src < 0.f ? 0.f : src;
  i0 = min((int)src, in_size - 1);
  i1 = i0 + (i0 < in_size - 1 ? 1 : 0);
  l1 = src - (float)i0;
  l0 = 1.f - l1;
}

// loss_part: [blocks][2] (ce sum, kd sum) ; d_sem accumulates ce_scale*dCE + kd_scale*dKD
// CT > 0: Ctot <= CT and the per-class gradient of a thread's pixel column is accumulated in 2*CT registers
// over consecutive rows that share the same low-res row pair (8x fewer LDS atomics); CT == 0: any Ctot,
// four LDS atomics per pixel and class.
template <int CT, int KT = CT>
__global__ __launch_bounds__(kThreads) void seg_losses_kernel(
    const float* __restrict__ sem_s, int ld_s, const float* __restrict__ sem_t, int ld_t, const int64_t* __restrict__ labels,
    int H, int W, int h, int w, int Ctot, int K, int ignore_index, float scale_h, float scale_w, float ce_scale,
    float kd_scale, float* __restrict__ loss_part, float* __restrict__ d_sem, int ld_d, int tiles_x, int tiles_y) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int b = blockIdx.z, ty0 = blockIdx.y * kTileY, tx0 = blockIdx.x * kTileX;
  // low-res footprint of the tile
  int ya, yb, xa, xb, dummy;
  float f0, f1;
  up_src(ty0, h, scale_h, ya, dummy, f0, f1);
  up_src(min(ty0 + kTileY, H) - 1, h, scale_h, dummy, yb, f0, f1);
  up_src(tx0, w, scale_w, xa, dummy, f0, f1);
  up_src(min(tx0 + kTileX, W) - 1, w, scale_w, dummy, xb, f0, f1);
  const int ny = yb - ya + 1, nx = xb - xa + 1, ncell = ny * nx;
  // CT > 0: kRep copies of the gradient accumulators, a lane adds into copy (lane & 15): the 15-16 consecutive pixel
  // columns under one low-resolution cell would otherwise hit the same LDS word with one ds_add each (a 16-way
  // serialisation per class: 0.98 -> measured below); copies are ncell * Ctot floats apart, an odd count of banks
  constexpr int kRep = CT > 0 ? 16 : 1;
  float* s_log = smem;                      // [ncell][Ctot] student logits
  float* t_log = s_log + ncell * Ctot;      // [ncell][K]    teacher logits
  float* g_acc = t_log + ncell * K;         // [kRep][ncell][Ctot] gradient accumulators
  float* red = g_acc + kRep * ncell * Ctot; // [2][4] block loss partials
  const int gstride = ncell * Ctot;
  for (int i = threadIdx.x; i < ncell * Ctot; i += kThreads) {
    const int cell = i / Ctot, c = i - cell * Ctot;
    const int cy = ya + cell / nx, cx = xa + cell % nx;
    s_log[i] = sem_s[((size_t)(b * h + cy) * w + cx) * ld_s + c];
  }
  for (int i = threadIdx.x; i < kRep * gstride; i += kThreads) g_acc[i] = 0.f;
  if (sem_t)
    for (int i = threadIdx.x; i < ncell * K; i += kThreads) {
      const int cell = i / K, c = i - cell * K;
      const int cy = ya + cell / nx, cx = xa + cell % nx;
      t_log[i] = sem_t[((size_t)(b * h + cy) * w + cx) * ld_t + c];
    }
  __syncthreads();

  float ce_sum = 0.f, kd_sum = 0.f;
  const int X = tx0 + (threadIdx.x & 63);
  const float invK = K > 0 ? 1.f / (float)K : 0.f;
  constexpr int CTA = CT > 0 ? CT : 1;
  float acc0[CTA], acc1[CTA];          // sum over rows of ly0*g / ly1*g for the current low-res row pair
  int cur_y0 = -1, cur_y1 = -1, x0 = 0, x1 = 0;
  float lx0 = 0.f, lx1 = 0.f;
  if (X < W) up_src(X, w, scale_w, x0, x1, lx0, lx1);
  auto flush = [&]() {
    if (CT > 0 && cur_y0 >= 0) {
      const int r0 = (cur_y0 - ya) * nx, r1 = (cur_y1 - ya) * nx;
      float* ga = g_acc + (threadIdx.x & (kRep - 1)) * gstride;
#pragma unroll
      for (int c = 0; c < CTA; ++c)
        if (c < Ctot) {
          atomicAdd(&ga[(r0 + x0 - xa) * Ctot + c], lx0 * acc0[c]);
          atomicAdd(&ga[(r0 + x1 - xa) * Ctot + c], lx1 * acc0[c]);
          atomicAdd(&ga[(r1 + x0 - xa) * Ctot + c], lx0 * acc1[c]);
          atomicAdd(&ga[(r1 + x1 - xa) * Ctot + c], lx1 * acc1[c]);
        }
    }
  };
  for (int it = 0; it < kRows; ++it) {
    const int Y = ty0 + (threadIdx.x >> 6) * kRows + it;
    if (X >= W || Y >= H) continue;
    int y0, y1;
    float ly0, ly1;
    up_src(Y, h, scale_h, y0, y1, ly0, ly1);
    if (CT > 0 && (y0 != cur_y0 || y1 != cur_y1)) {
      flush();
      cur_y0 = y0; cur_y1 = y1;
#pragma unroll
      for (int c = 0; c < CTA; ++c) { acc0[c] = 0.f; acc1[c] = 0.f; }
    }
    const int c00 = (y0 - ya) * nx + (x0 - xa), c01 = (y0 - ya) * nx + (x1 - xa);
    const int c10 = (y1 - ya) * nx + (x0 - xa), c11 = (y1 - ya) * nx + (x1 - xa);
    // torch's up-sampling arithmetic: h0*(w0*v00 + w1*v01) + h1*(w0*v10 + w1*v11)
    auto interp = [&](const float* base, int stride, int c) {
      return ly0 * (lx0 * base[c00 * stride + c] + lx1 * base[c01 * stride + c]) +
             ly1 * (lx0 * base[c10 * stride + c] + lx1 * base[c11 * stride + c]);
    };
    const int64_t lab64 = labels[((size_t)b * H + Y) * W + X];
    const bool ignored = lab64 == ignore_index;
    int lab = ignored ? 0 : (int)lab64;
    if (lab < K) lab = 0;                                    // loss.py:104-105
    if (CT > 0) {
      // Register form (Ctot <= CT, K <= KT).  Every interpolated logit is formed once; the class-set tests (c < K,
      // c == 0 || c >= K, 1 <= c < K) are wave-uniform 0 / 1 multipliers, classes past Ctot / K carry -1e30 (their
      // exponentials are exact zeros, so no guards inside the unrolled loops); e_c = exp(z_c - max) is computed once and
      // every later exponential is a ratio of it:  exp(z_c - LSE_S) = e_c / sum_S e.  (The first version re-evaluated
      // ~120 exponentials and ~1700 vector instructions per pixel and was issue-bound, tools/prof_seglosses.sh: 980 ->
      // 630 us at B = 24, 513^2.  227 VGPRs = two waves per SIMD; capping at three spills and is slower, 790 us.)
      constexpr float kL2e = 1.4426950408889634f, kLn2 = 0.6931471805599453f, kNegBig = -1e30f;
      constexpr int KTA = KT > 0 ? KT : 1;
      float zc[CTA], ec[CTA];
      float mz = kNegBig;
#pragma unroll
      for (int c = 0; c < CTA; ++c) {
        const float z = interp(s_log, Ctot, c);
        zc[c] = c < Ctot ? z : kNegBig;
        mz = fmaxf(mz, zc[c]);
      }
      const float mzl = mz * kL2e;
      float s_all = 0.f, s_old = 0.f, s_bn = 0.f, z_lab = 0.f;
#pragma unroll
      for (int c = 0; c < CTA; ++c) {
        const float e = __builtin_amdgcn_exp2f(__builtin_fmaf(zc[c], kL2e, -mzl));
        ec[c] = e;
        s_all += e;
        s_old = __builtin_fmaf(e, c < K ? 1.f : 0.f, s_old);
        s_bn = __builtin_fmaf(e, (c == 0 || c >= K) ? 1.f : 0.f, s_bn);
        z_lab = c == lab ? zc[c] : z_lab;
      }
      const float den = mz + kLn2 * __builtin_amdgcn_logf(s_all);
      const float lse_old = mz + kLn2 * __builtin_amdgcn_logf(s_old), lse_bn = mz + kLn2 * __builtin_amdgcn_logf(s_bn);
      const float logp = lab == 0 ? lse_old - den : z_lab - den;
      if (!ignored) ce_sum += -logp;
      const float inv_all = 1.f / s_all, inv_old = 1.f / s_old, inv_bn = 1.f / s_bn;
      // teacher soft-max
      float te[KTA];
      float inv_st = 0.f, q0 = 0.f, kd_pix = 0.f;
      if (sem_t) {
        float tcv[KTA];
        float mt = kNegBig, st = 0.f;
#pragma unroll
        for (int c = 0; c < KTA; ++c) {
          const float t = interp(t_log, K, c);
          tcv[c] = c < K ? t : kNegBig;
          mt = fmaxf(mt, tcv[c]);
        }
        const float mtl = mt * kL2e;
#pragma unroll
        for (int c = 0; c < KTA; ++c) {
          te[c] = __builtin_amdgcn_exp2f(__builtin_fmaf(tcv[c], kL2e, -mtl));
          st += te[c];
        }
        inv_st = 1.f / st;
        q0 = te[0] * inv_st;
        kd_pix = q0 * (lse_bn - den);
      }
      const float ce_w = ignored ? 0.f : ce_scale;
      const float kdw = kd_scale * invK;
      const bool lab0 = lab == 0;
      const float q0bn = q0 * inv_bn;
#pragma unroll
      for (int c = 0; c < CTA; ++c) {
        const float p = ec[c] * inv_all;
        const float t_old = ec[c] * (c < K ? inv_old : 0.f);
        const float t_hot = c == lab ? 1.f : 0.f;
        float g = ce_w * (p - (lab0 ? t_old : t_hot));
        if (sem_t) {
          float qc = 0.f;
          if (c >= 1 && c < KTA) {                      // compile-time; the 1 <= c < K test is the uniform multiplier
            qc = te[c] * (c < K ? inv_st : 0.f);
            kd_pix = __builtin_fmaf(qc, zc[c] - den, kd_pix);
          }
          const float bn = ec[c] * ((c == 0 || c >= K) ? q0bn : 0.f);
          g = __builtin_fmaf(kdw, p - bn - qc, g);
        }
        acc0[c] = __builtin_fmaf(ly0, g, acc0[c]);
        acc1[c] = __builtin_fmaf(ly1, g, acc1[c]);
      }
      kd_sum += -kd_pix * invK;
      continue;
    }
    // CT > 0: every interpolated logit is formed ONCE and kept in a register (the three passes below would otherwise
    // redo the 4 LDS reads + 6 flops of the interpolation three times per class: ~450 LDS reads per pixel)
    float zc[CTA], tc[CTA];
    if (CT > 0) {
#pragma unroll
      for (int c = 0; c < CTA; ++c) {
        zc[c] = c < Ctot ? interp(s_log, Ctot, c) : -INFINITY;
        tc[c] = (sem_t && c < K) ? interp(t_log, K, c) : -INFINITY;
      }
    }
    // pass A: maxima
    float mz = -INFINITY;
    if (CT > 0) {
#pragma unroll
      for (int c = 0; c < CTA; ++c) mz = fmaxf(mz, zc[c]);
    } else {
      for (int c = 0; c < Ctot; ++c) mz = fmaxf(mz, interp(s_log, Ctot, c));
    }
    // pass B: the three sums, the labelled logit
    float s_all = 0.f, s_old = 0.f, s_bn = 0.f, z_lab = 0.f;
    if (CT > 0) {
#pragma unroll
      for (int c = 0; c < CTA; ++c)
        if (c < Ctot) {
          const float z = zc[c];
          const float e = __expf(z - mz);
          s_all += e;
          if (c < K) s_old += e;
          if (c == 0 || c >= K) s_bn += e;
          if (c == lab) z_lab = z;
        }
    } else {
      for (int c = 0; c < Ctot; ++c) {
        const float z = interp(s_log, Ctot, c);
        const float e = __expf(z - mz);
        s_all += e;
        if (c < K) s_old += e;
        if (c == 0 || c >= K) s_bn += e;
        if (c == lab) z_lab = z;
      }
    }
    const float den = mz + __logf(s_all);
    const float lse_old = mz + __logf(s_old), lse_bn = mz + __logf(s_bn);
    const float logp = lab == 0 ? lse_old - den : z_lab - den;
    if (!ignored) ce_sum += -logp;
    // teacher soft-max
    float mt = -INFINITY, st = 0.f, q0 = 0.f, kd_pix = 0.f;
    if (sem_t) {
      if (CT > 0) {
#pragma unroll
        for (int c = 0; c < CTA; ++c) mt = fmaxf(mt, tc[c]);
#pragma unroll
        for (int c = 0; c < CTA; ++c) {
          tc[c] = c < K ? __expf(tc[c] - mt) : 0.f;      // from here on: un-normalised teacher probabilities
          st += tc[c];
        }
        q0 = tc[0] / st;
      } else {
        for (int c = 0; c < K; ++c) mt = fmaxf(mt, interp(t_log, K, c));
        for (int c = 0; c < K; ++c) st += __expf(interp(t_log, K, c) - mt);
        q0 = __expf(interp(t_log, K, 0) - mt) / st;
      }
      kd_pix = q0 * (lse_bn - den);
    }
    // pass C: gradients (and the old-class part of the KD loss)
    const float ce_w = ignored ? 0.f : ce_scale;
    const float inv_st = sem_t ? 1.f / st : 0.f;
    auto grad_c = [&](int c, float z, float te) {   // te: exp(teacher_c - mt) (CT > 0) or unused
      const float p = __expf(z - den);
      float g = ce_w * (p - (lab == 0 ? (c < K ? __expf(z - lse_old) : 0.f) : (c == lab ? 1.f : 0.f)));
      if (sem_t) {
        float qc = 0.f;
        if (c >= 1 && c < K) {
          qc = CT > 0 ? te * inv_st : __expf(interp(t_log, K, c) - mt) / st;
          kd_pix += qc * (z - den);
        }
        const float bn = (c == 0 || c >= K) ? q0 * __expf(z - lse_bn) : 0.f;
        g += kd_scale * invK * (p - bn - qc);
      }
      return g;
    };
    if (CT > 0) {
#pragma unroll
      for (int c = 0; c < CTA; ++c)
        if (c < Ctot) {
          const float g = grad_c(c, zc[c], tc[c]);
          acc0[c] += ly0 * g;
          acc1[c] += ly1 * g;
        }
    } else {
      const float w00 = ly0 * lx0, w01 = ly0 * lx1, w10 = ly1 * lx0, w11 = ly1 * lx1;
      for (int c = 0; c < Ctot; ++c) {
        const float g = grad_c(c, interp(s_log, Ctot, c), 0.f);
        atomicAdd(&g_acc[c00 * Ctot + c], w00 * g);
        atomicAdd(&g_acc[c01 * Ctot + c], w01 * g);
        atomicAdd(&g_acc[c10 * Ctot + c], w10 * g);
        atomicAdd(&g_acc[c11 * Ctot + c], w11 * g);
      }
    }
    kd_sum += -kd_pix * invK;
  }
  flush();
  __syncthreads();
  for (int i = threadIdx.x; i < ncell * Ctot; i += kThreads) {
    float v = g_acc[i];
#pragma unroll
    for (int r = 1; r < kRep; ++r) v += g_acc[r * gstride + i];
    if (v != 0.f) {
      const int cell = i / Ctot, c = i - cell * Ctot;
      const int cy = ya + cell / nx, cx = xa + cell % nx;
      atomicAdd(&d_sem[((size_t)(b * h + cy) * w + cx) * ld_d + c], v);
    }
  }
  // block loss sums (fixed order)
  ce_sum = wave_sum(ce_sum);
  kd_sum = wave_sum(kd_sum);
  if ((threadIdx.x & 63) == 0) { red[threadIdx.x >> 6] = ce_sum; red[4 + (threadIdx.x >> 6)] = kd_sum; }
  __syncthreads();
  if (threadIdx.x == 0) {
    const int blk = (blockIdx.z * tiles_y + blockIdx.y) * tiles_x + blockIdx.x;
    loss_part[2 * blk + 0] = red[0] + red[1] + red[2] + red[3];
    loss_part[2 * blk + 1] = red[4] + red[5] + red[6] + red[7];
  }
}

// ---- round 5: the few-class form on packed fp32 math, old and new classes in separate register groups ------------------------------
// The register form above spends ~1400 vector instructions per pixel (ISA count): nine scalar multiplies / adds and a select per
// interpolated logit, a 0 / 1 multiplier or a select per class for every class set, 227 VGPRs with the class-set masks spilled to
// VGPR lanes (v_readlane in the inner loop) - 630 us, 0.01 of any roof.  Here the classes are laid out by SET at staging time: LDS
// slots [0, KT) hold the old classes 0 .. K-1, slots [KT, KT + NT) the new classes K .. Ctot-1, unused slots hold -1e30 (their
// exponentials are exact zeros).  Which set a slot belongs to is then a compile-time fact: sum_old / sum_new are two unmasked sums
// (sum_all = old + new, sum_bkg+new = new + e_0), the gradient coefficient of a slot is one of three per-pixel scalars, the label can
// only hit one of the NT new slots.  All per-class arithmetic runs on class PAIRS (v_pk_mul / v_pk_fma / v_pk_add_f32), logits
// come from LDS four slots at a time (ds_read_b128).  ~330 vector instructions per pixel.
typedef float f32x4 __attribute__((ext_vector_type(4)));
constexpr int kRepPk = 8;          // copies of the LDS gradient accumulators (lane & 7): 45 KB of LDS at the worst-case 7 x 7 cells

__device__ __forceinline__ f32x2 bc2(float v) { return f32x2{v, v}; }
__device__ __forceinline__ f32x4 bc4(float v) { return f32x4{v, v, v, v}; }
__device__ __forceinline__ f32x2 fma2(f32x2 a, f32x2 b, f32x2 c) { return __builtin_elementwise_fma(a, b, c); }
__device__ __forceinline__ f32x4 fma4(f32x4 a, f32x4 b, f32x4 c) { return __builtin_elementwise_fma(a, b, c); }
__device__ __forceinline__ f32x2 exp2_2(f32x2 a) { return f32x2{__builtin_amdgcn_exp2f(a.x), __builtin_amdgcn_exp2f(a.y)}; }

template <int KT, int NT>
__global__ __launch_bounds__(kThreads) void seg_losses_pk_kernel(
    const float* __restrict__ sem_s, int ld_s, const float* __restrict__ sem_t, int ld_t, const int64_t* __restrict__ labels,
    int H, int W, int h, int w, int Ctot, int K, int ignore_index, float scale_h, float scale_w, float ce_scale,
    float kd_scale, float* __restrict__ loss_part, float* __restrict__ d_sem, int ld_d, int tiles_x, int tiles_y, int cell_cap,
    float fx_scale) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  constexpr int CT = KT + NT, NP = CT / 2, KP = KT / 2, kRep = kRepPk;
  // logit rows in LDS are CT + 1 / KT + 1 floats apart: the lanes of a wave read from ~5 cells, sixteen lanes the same address - a
  // ds_read_b32 broadcasts that, a ds_read_b128 serialises it (59 cycles per instruction measured; SQ_LDS_BANK_CONFLICT), and the
  // odd stride keeps the compiler from merging the dword reads
  constexpr int CTS = CT + 1, KTS = KT + 1;
  static_assert(KT % 4 == 0 && NT % 4 == 0, "slot groups are read four at a time");
  constexpr float kL2e = 1.4426950408889634f, kLn2 = 0.6931471805599453f, kNegBig = -1e30f;
  const int b = blockIdx.z, ty0 = blockIdx.y * kTileY, tx0 = blockIdx.x * kTileX;
  int ya, yb, xa, xb, dummy;
  float f0, f1;
  up_src(ty0, h, scale_h, ya, dummy, f0, f1);
  up_src(min(ty0 + kTileY, H) - 1, h, scale_h, dummy, yb, f0, f1);
  up_src(tx0, w, scale_w, xa, dummy, f0, f1);
  up_src(min(tx0 + kTileX, W) - 1, w, scale_w, dummy, xb, f0, f1);
  const int ny = yb - ya + 1, nx = xb - xa + 1, ncell = ny * nx;
  // the host sized the LDS for the exact footprint it computed with the same arithmetic; a disagreement must not become a silent
  // overrun of the accumulators
  if (ncell > cell_cap) __builtin_trap();
  // copies 1 (mod 32) doubles apart: a lane's accumulator word sits at copy (lane & 7) * gstride + cell * CT + slot, and 24 * cell +
  // copy (mod 32) then takes 32 different values over the lanes of a wave - every LDS bank pair is hit by two lanes.  (Copies a
  // multiple of 32 doubles apart put all eight on the same banks: LDS busy 57 % of the kernel, SQ_LDS_BANK_CONFLICT.)
  const int gstride = ((ncell * CT + 31) & ~31) + 1;
  // gradient accumulators in DOUBLE: ds_add_f64 runs at 7.7 cycles per wave instruction, ds_add_f32 at 169 (tools/lds_atomic_probe.hip,
  // profiles/r05_lds_atomics.txt) - the fp32 LDS atomics of the flush were 80 % of the round-3 kernel's 630 us
  double* g_acc = reinterpret_cast<double*>(smem);                     // [kRep][ncell][CT] gradient accumulators by slot
  float* s_log = reinterpret_cast<float*>(g_acc + kRep * gstride + 1);  // [ncell][CTS] student logits by slot
  float* t_log = s_log + ncell * CTS;         // [ncell][KTS] teacher logits
  float* red = t_log + ncell * KTS;           // [2][4] block loss partials
  unsigned char* lab_s = reinterpret_cast<unsigned char*>(red + 8);      // [kRows][kThreads] label codes (0xFF: ignored)
  // all of a thread's labels up front: sixteen independent loads in flight under the staging below.  Loaded inside the row loop
  // the label is a ~2 us HBM round trip at the head of every iteration with two waves per SIMD to hide it - the round-3 form's
  // 630 us were this latency, not its instruction count (packed arithmetic alone: 624 us)
  {
    const int Xl = tx0 + (threadIdx.x & 63), Yl = ty0 + (threadIdx.x >> 6) * kRows;
    int64_t lv[kRows];
#pragma unroll
    for (int it = 0; it < kRows; ++it) lv[it] = (Xl < W && Yl + it < H) ? labels[((size_t)b * H + Yl + it) * W + Xl] : 0;
#pragma unroll
    for (int it = 0; it < kRows; ++it)      // as the generic form reads them: negative -> background, above the classes -> no class
      lab_s[it * kThreads + threadIdx.x] = lv[it] == ignore_index ? 0xFF : (unsigned char)(lv[it] < 0 ? 0 : lv[it] > 254 ? 254 : lv[it]);
  }
  for (int i = threadIdx.x; i < ncell * CT; i += kThreads) {
    const int cell = i / CT, sl = i - cell * CT;
    const int cy = ya + cell / nx, cx = xa + cell % nx;
    const int c = sl < KT ? sl : K + sl - KT;
    const bool valid = sl < KT ? sl < K : c < Ctot;
    s_log[cell * CTS + sl] = valid ? sem_s[((size_t)(b * h + cy) * w + cx) * ld_s + c] : kNegBig;
  }
  for (int i = threadIdx.x; i < kRep * gstride; i += kThreads) g_acc[i] = 0.0;
  if (sem_t)
    for (int i = threadIdx.x; i < ncell * KT; i += kThreads) {
      const int cell = i / KT, c = i - cell * KT;
      const int cy = ya + cell / nx, cx = xa + cell % nx;
      t_log[cell * KTS + c] = c < K ? sem_t[((size_t)(b * h + cy) * w + cx) * ld_t + c] : kNegBig;
    }
  __syncthreads();

  float ce_sum = 0.f, kd_sum = 0.f;
  const int X = tx0 + (threadIdx.x & 63);
  const float invK = 1.f / (float)K;
  const float kdw = sem_t ? kd_scale * invK : 0.f;
  f32x2 acc0[NP], acc1[NP];            // sum over rows of ly0 * g / ly1 * g for the current low-resolution row pair, by slot pair
  int cur_y0 = -1, cur_y1 = -1, x0 = 0, x1 = 0;
  float lx0 = 0.f, lx1 = 0.f;
  if (X < W) up_src(X, w, scale_w, x0, x1, lx0, lx1);
  const int cx0 = x0 - xa, cx1 = x1 - xa;
  auto flush = [&]() {
    if (cur_y0 < 0) return;
    double* ga = g_acc + (threadIdx.x & (kRep - 1)) * gstride;
    double* g00 = ga + ((cur_y0 - ya) * nx + cx0) * CT;
    double* g01 = ga + ((cur_y0 - ya) * nx + cx1) * CT;
    double* g10 = ga + ((cur_y1 - ya) * nx + cx0) * CT;
    double* g11 = ga + ((cur_y1 - ya) * nx + cx1) * CT;
    const f32x2 L0 = bc2(lx0), L1 = bc2(lx1);
#pragma unroll
    for (int p = 0; p < NP; ++p) {
      const f32x2 a0 = L0 * acc0[p], a1 = L1 * acc0[p], b0 = L0 * acc1[p], b1 = L1 * acc1[p];
      atomicAdd(g00 + 2 * p, (double)a0.x); atomicAdd(g00 + 2 * p + 1, (double)a0.y);
      atomicAdd(g01 + 2 * p, (double)a1.x); atomicAdd(g01 + 2 * p + 1, (double)a1.y);
      atomicAdd(g10 + 2 * p, (double)b0.x); atomicAdd(g10 + 2 * p + 1, (double)b0.y);
      atomicAdd(g11 + 2 * p, (double)b1.x); atomicAdd(g11 + 2 * p + 1, (double)b1.y);
    }
  };
  // The x half of the bilinear form is the same for every pixel row that shares a low-resolution row pair: u_y = w0 v(y, x0) +
  // w1 v(y, x1) of the two source rows is formed ONCE per row pair (48 + 32 registers) and a pixel costs z = h0 u_y0 + h1 u_y1 -
  // no LDS read.  (Reading the four corners per pixel - 40 ds_read_b128 of mostly identical addresses, which the LDS serialises
  // lane by lane - kept the LDS pipe busy 80 % of the kernel: SQ_LDS_IDX_ACTIVE, 59 cycles per read; 548 us.)
  f32x2 u0[NP], u1[NP], tu0[KP], tu1[KP];
  for (int it = 0; it < kRows; ++it) {
    const int Y = ty0 + (threadIdx.x >> 6) * kRows + it;
    if (X >= W || Y >= H) continue;
    int y0, y1;
    float ly0, ly1;
    up_src(Y, h, scale_h, y0, y1, ly0, ly1);
    if (y0 != cur_y0 || y1 != cur_y1) {
      flush();
      cur_y0 = y0; cur_y1 = y1;
      const int r0 = (y0 - ya) * nx, r1 = (y1 - ya) * nx;
      const int c00 = r0 + cx0, c01 = r0 + cx1, c10 = r1 + cx0, c11 = r1 + cx1;
#pragma unroll
      for (int p = 0; p < NP; ++p) { acc0[p] = bc2(0.f); acc1[p] = bc2(0.f); }
      const f32x2 L0 = bc2(lx0), L1 = bc2(lx1);
      const float *s00 = s_log + c00 * CTS, *s01 = s_log + c01 * CTS, *s10 = s_log + c10 * CTS, *s11 = s_log + c11 * CTS;
#pragma unroll
      for (int p = 0; p < NP; ++p) {
        u0[p] = fma2(L1, f32x2{s01[2 * p], s01[2 * p + 1]}, L0 * f32x2{s00[2 * p], s00[2 * p + 1]});
        u1[p] = fma2(L1, f32x2{s11[2 * p], s11[2 * p + 1]}, L0 * f32x2{s10[2 * p], s10[2 * p + 1]});
      }
      if (sem_t) {
        const float *t00 = t_log + c00 * KTS, *t01 = t_log + c01 * KTS, *t10 = t_log + c10 * KTS, *t11 = t_log + c11 * KTS;
#pragma unroll
        for (int p = 0; p < KP; ++p) {
          tu0[p] = fma2(L1, f32x2{t01[2 * p], t01[2 * p + 1]}, L0 * f32x2{t00[2 * p], t00[2 * p + 1]});
          tu1[p] = fma2(L1, f32x2{t11[2 * p], t11[2 * p + 1]}, L0 * f32x2{t10[2 * p], t10[2 * p + 1]});
        }
      }
    }
    const f32x2 LY0p = bc2(ly0), LY1p = bc2(ly1);
    const int code = lab_s[it * kThreads + threadIdx.x];
    const bool ignored = code == 0xFF;
    int lab = ignored ? 0 : code;
    if (lab < K) lab = 0;                                    // loss.py:104-105
    const bool lab0 = lab == 0;
    const int jlab = lab - K;                                // the label's new-class slot (negative: background / old / ignored)

    // z = h0 u_y0 + h1 u_y1 is two packed instructions per slot pair: formed for the maximum, formed again for the exponentials
    // (and for the old classes' KD term) instead of living in 24 registers
    auto zpair = [&](int p) { return fma2(LY1p, u1[p], LY0p * u0[p]); };
    f32x2 e[NP];
    float mz = kNegBig;
#pragma unroll
    for (int p = 0; p < NP; ++p) {
      const f32x2 zz = zpair(p);
      mz = fmaxf(mz, fmaxf(zz.x, zz.y));
    }
    const f32x2 L2E = bc2(kL2e), MZL = bc2(-mz * kL2e);
    f32x2 so = bc2(0.f), sn = bc2(0.f);
    float z_lab = 0.f;
#pragma unroll
    for (int p = 0; p < NP; ++p) {
      const f32x2 zz = zpair(p);
      e[p] = exp2_2(fma2(zz, L2E, MZL));
      if (p < KP) {
        so += e[p];
      } else {
        sn += e[p];
        z_lab = jlab == 2 * (p - KP) ? zz.x : z_lab;
        z_lab = jlab == 2 * (p - KP) + 1 ? zz.y : z_lab;
      }
    }
    const float s_old = so.x + so.y, s_new = sn.x + sn.y;
    const float s_all = s_old + s_new, s_bn = s_new + e[0].x;
    const float den = mz + kLn2 * __builtin_amdgcn_logf(s_all);
    const float lse_old = mz + kLn2 * __builtin_amdgcn_logf(s_old), lse_bn = mz + kLn2 * __builtin_amdgcn_logf(s_bn);
    const float logp = lab0 ? lse_old - den : z_lab - den;
    if (!ignored) ce_sum += -logp;
    const float inv_all = 1.f / s_all, inv_old = 1.f / s_old, inv_bn = 1.f / s_bn;
    const float ce_w = ignored ? 0.f : ce_scale;
    // teacher soft-max over the old classes; q_c = te_c / sum te
    f32x2 te[KP];
    float inv_st = 0.f, q0 = 0.f, kd_pix = 0.f;
    if (sem_t) {
      float mt = kNegBig;
#pragma unroll
      for (int p = 0; p < KP; ++p) {
        te[p] = fma2(LY1p, tu1[p], LY0p * tu0[p]);
        mt = fmaxf(mt, fmaxf(te[p].x, te[p].y));
      }
      const f32x2 MTL = bc2(-mt * kL2e);
      f32x2 st2 = bc2(0.f);
#pragma unroll
      for (int p = 0; p < KP; ++p) {
        te[p] = exp2_2(fma2(te[p], L2E, MTL));
        st2 += te[p];
      }
      inv_st = 1.f / (st2.x + st2.y);
      q0 = te[0].x * inv_st;
      kd_pix = q0 * (lse_bn - den);
      // old classes 1 .. K-1: q_c (z_c - den); unused slots carry q = 0 against a finite z
      const f32x2 IST = bc2(inv_st), DEN = bc2(den);
      f32x2 kd2 = f32x2{0.f, te[0].y * inv_st} * (zpair(0) - DEN);
#pragma unroll
      for (int p = 1; p < KP; ++p) kd2 = fma2(te[p] * IST, zpair(p) - DEN, kd2);
      kd_pix += kd2.x + kd2.y;
    } else {
#pragma unroll
      for (int p = 0; p < KP; ++p) te[p] = bc2(0.f);
    }
    // g_c = e_c [(ce_w + kdw) / sum_all - [c old] ce_w [label bkg/old] / sum_old - [c bkg/new] kdw q_0 / sum_bn]
    //       - ce_w [label new][c == label] - kdw q_c [1 <= c < K]
    const float A = (ce_w + kdw) * inv_all, Bo = lab0 ? ce_w * inv_old : 0.f, Bb = kdw * q0 * inv_bn;
    const float coef_old = A - Bo, coef_new = A - Bb, nkq = -kdw * inv_st;
    const f32x2 COLD = bc2(coef_old), CNEW = bc2(coef_new), NKQ = bc2(nkq);
#pragma unroll
    for (int p = 0; p < NP; ++p) {
      f32x2 g;
      if (p == 0) {
        g = e[0] * f32x2{coef_old - Bb, coef_old};
        g = fma2(te[0], f32x2{0.f, nkq}, g);
      } else if (p < KP) {
        g = fma2(te[p], NKQ, e[p] * COLD);
      } else {
        const int j = 2 * (p - KP);
        g = e[p] * CNEW - f32x2{jlab == j ? ce_w : 0.f, jlab == j + 1 ? ce_w : 0.f};
      }
      acc0[p] = fma2(LY0p, g, acc0[p]);
      acc1[p] = fma2(LY1p, g, acc1[p]);
    }
    kd_sum += -kd_pix * invK;
  }
  flush();
  __syncthreads();
  for (int i = threadIdx.x; i < ncell * CT; i += kThreads) {
    double vd = g_acc[i];
#pragma unroll
    for (int r = 1; r < kRep; ++r) vd += g_acc[r * gstride + i];
    const int cell = i / CT, sl = i - cell * CT;
    const int c = sl < KT ? sl : K + sl - KT;
    const bool valid = sl < KT ? sl < K : c < Ctot;
    if (valid && vd != 0.0) {
      // up to four tiles meet in a low-resolution cell: their sums are added as 32-bit FIXED-POINT integers (d_sem's own words,
      // turned into floats by seg_grad_unfix_kernel) - integer addition has no order, so the gradient is the same bit pattern on
      // every run.  (fp32 atomics here made one bf16 rounding of the logit gradient flip in about one run in eight: two discrete
      // training trajectories 5e-4 apart after one update, DESIGN.md "Open" of round 5.)  fx_scale = 2^17 / (largest |gradient| of
      // one pixel): a tile's 4096 pixels stay below 2^29, four tiles below 2^31; resolution 7.6e-6 of one pixel's largest gradient.
      const int cy = ya + cell / nx, cx = xa + cell % nx;
      atomicAdd(reinterpret_cast<int*>(d_sem) + ((size_t)(b * h + cy) * w + cx) * ld_d + c, __double2int_rn(vd * (double)fx_scale));
    }
  }
  // block loss sums (fixed order)
  ce_sum = wave_sum(ce_sum);
  kd_sum = wave_sum(kd_sum);
  if ((threadIdx.x & 63) == 0) { red[threadIdx.x >> 6] = ce_sum; red[4 + (threadIdx.x >> 6)] = kd_sum; }
  __syncthreads();
  if (threadIdx.x == 0) {
    const int blk = (blockIdx.z * tiles_y + blockIdx.y) * tiles_x + blockIdx.x;
    loss_part[2 * blk + 0] = red[0] + red[1] + red[2] + red[3];
    loss_part[2 * blk + 1] = red[4] + red[5] + red[6] + red[7];
  }
}

// ---- the same losses for MANY classes (ADE20K: 151 student / 101 teacher classes; any Ctot the LDS holds) ----------------------
// The generic form above (CT == 0) re-interpolates every logit in each of its passes and adds every pixel's gradient to LDS with
// four float atomics per class - 604 contended LDS atomics and ~900 exponentials per pixel at 151 classes: 4.8 ms for ONE rank's
// 3 x 512^2 batch (26 % of that step; bench.py --dataset ade --task 100-50 --global_batch 3).  Here:
//   * a thread owns one pixel column of RW rows; phase A walks all classes of a row twice (maxima, then sums: student e_c and -
//     for the old classes - teacher te_c in the same pass; exp(z_c - LSE_S) is e_c / sum_S e, so no exponential is evaluated
//     for a ratio) and keeps 8 constants per row in registers; the loss values need nothing else:
//         kd_pix = q_0 (LSE_bn - den) + (sum_{1<=c<K} te_c z_c - den sum_{1<=c<K} te_c) / sum te;
//   * phase B walks the classes again in chunks of 16: g_c = e_c (a_all - a_old [c<K] - b_bn [c in bkg/new]) - hot [c == label]
//     - b_q te_c [1<=c<K] from those constants, accumulated in 2 x 16 registers over the rows that share a low-resolution row
//     pair and flushed with one LDS atomic per class and corner (RW-fold fewer atomics, accumulator copies against column conflicts);
//   * logits are read from LDS four classes at a time (rows padded to a multiple of four with -1e30: their exponentials are 0).
constexpr int kRW = 8;            // rows per thread: a 32 x 64 pixel tile per workgroup
// (LDS sizing: the fp64 accumulator copies take the bytes of four fp32 copies)
constexpr int kRepWD = 2;         // fp64 copies actually kept

__device__ __forceinline__ float4 interp4(const float* base, int o00, int o01, int o10, int o11, int c, float lx0, float lx1,
                                          float ly0, float ly1) {
  const float4 a = *reinterpret_cast<const float4*>(base + o00 + c), b = *reinterpret_cast<const float4*>(base + o01 + c);
  const float4 d = *reinterpret_cast<const float4*>(base + o10 + c), e = *reinterpret_cast<const float4*>(base + o11 + c);
  // torch's up-sampling arithmetic: h0*(w0*v00 + w1*v01) + h1*(w0*v10 + w1*v11)
  return make_float4(ly0 * (lx0 * a.x + lx1 * b.x) + ly1 * (lx0 * d.x + lx1 * e.x), ly0 * (lx0 * a.y + lx1 * b.y) + ly1 * (lx0 * d.y + lx1 * e.y),
                     ly0 * (lx0 * a.z + lx1 * b.z) + ly1 * (lx0 * d.z + lx1 * e.z), ly0 * (lx0 * a.w + lx1 * b.w) + ly1 * (lx0 * d.w + lx1 * e.w));
}

__global__ __launch_bounds__(kThreads) void seg_losses_wide_kernel(
    const float* __restrict__ sem_s, int ld_s, const float* __restrict__ sem_t, int ld_t, const int64_t* __restrict__ labels,
    int H, int W, int h, int w, int Ctot, int K, int ignore_index, float scale_h, float scale_w, float ce_scale,
    float kd_scale, float* __restrict__ loss_part, float* __restrict__ d_sem, int ld_d, int tiles_x, int tiles_y, float fx_scale) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  constexpr int kTY = 4 * kRW;
  constexpr float kL2e = 1.4426950408889634f, kLn2 = 0.6931471805599453f, kNegBig = -1e30f;
  const int CS = (Ctot + 3) & ~3, KS = sem_t ? ((K + 3) & ~3) : 0;
  const int b = blockIdx.z, ty0 = blockIdx.y * kTY, tx0 = blockIdx.x * kTileX;
  int ya, yb, xa, xb, dummy;
  float f0, f1;
  up_src(ty0, h, scale_h, ya, dummy, f0, f1);
  up_src(min(ty0 + kTY, H) - 1, h, scale_h, dummy, yb, f0, f1);
  up_src(tx0, w, scale_w, xa, dummy, f0, f1);
  up_src(min(tx0 + kTileX, W) - 1, w, scale_w, dummy, xb, f0, f1);
  const int ny = yb - ya + 1, nx = xb - xa + 1, ncell = ny * nx;
  float* s_log = smem;                         // [ncell][CS]
  float* t_log = s_log + ncell * CS;           // [ncell][KS]
  // fp64 accumulators, two copies in the bytes of four fp32 ones: ds_add_f64 is 20x the rate of ds_add_f32 on gfx950 (section 3.5 of
  // DESIGN.md, profiles/r05_lds_atomics.txt); copies 4 (mod 32) doubles apart keep them off each other's banks
  const int gstride = ((ncell * CS + 31) & ~31) + 4;
  double* g_acc = reinterpret_cast<double*>(t_log + ncell * KS);   // [kRepWD][gstride]
  float* red = reinterpret_cast<float*>(g_acc + kRepWD * gstride);  // [2][4]
  for (int i = threadIdx.x; i < ncell * CS; i += kThreads) {
    const int cell = i / CS, c = i - cell * CS;
    const int cy = ya + cell / nx, cx = xa + cell % nx;
    s_log[i] = c < Ctot ? sem_s[((size_t)(b * h + cy) * w + cx) * ld_s + c] : kNegBig;
  }
  for (int i = threadIdx.x; i < kRepWD * gstride; i += kThreads) g_acc[i] = 0.0;
  for (int i = threadIdx.x; i < ncell * KS; i += kThreads) {
    const int cell = i / KS, c = i - cell * KS;
    const int cy = ya + cell / nx, cx = xa + cell % nx;
    t_log[i] = c < K ? sem_t[((size_t)(b * h + cy) * w + cx) * ld_t + c] : kNegBig;
  }
  __syncthreads();

  const int X = tx0 + (threadIdx.x & 63), Ybase = ty0 + (threadIdx.x >> 6) * kRW;
  const float invK = K > 0 ? 1.f / (float)K : 0.f;
  const float kdw = sem_t ? kd_scale * invK : 0.f;
  int x0 = 0, x1 = 0;
  float lx0 = 0.f, lx1 = 0.f;
  if (X < W) up_src(X, w, scale_w, x0, x1, lx0, lx1);
  const int cx0 = x0 - xa, cx1 = x1 - xa;
  float ce_sum = 0.f, kd_sum = 0.f;
  // ---- phase A: per row, the normalisers and the loss values --------------------------------------------------------------
  float rmzl[kRW], ra_all[kRW], ra_old[kRW], rhot[kRW], rbbn[kRW], rmtl[kRW], rbq[kRW];
  int rlab[kRW];
#pragma unroll
  for (int it = 0; it < kRW; ++it) {
    const int Y = Ybase + it;
    rmzl[it] = 0.f; ra_all[it] = 0.f; ra_old[it] = 0.f; rhot[it] = 0.f; rbbn[it] = 0.f; rmtl[it] = 0.f; rbq[it] = 0.f; rlab[it] = -1;
    if (X >= W || Y >= H) continue;
    int y0, y1;
    float ly0, ly1;
    up_src(Y, h, scale_h, y0, y1, ly0, ly1);
    const int r0 = (y0 - ya) * nx, r1 = (y1 - ya) * nx;
    const int s00 = (r0 + cx0) * CS, s01 = (r0 + cx1) * CS, s10 = (r1 + cx0) * CS, s11 = (r1 + cx1) * CS;
    const int t00 = (r0 + cx0) * KS, t01 = (r0 + cx1) * KS, t10 = (r1 + cx0) * KS, t11 = (r1 + cx1) * KS;
    const int64_t lab64 = labels[((size_t)b * H + Y) * W + X];
    const bool ignored = lab64 == ignore_index;
    int lab = ignored ? 0 : (int)lab64;
    if (lab < K) lab = 0;                                    // loss.py:104-105
    float mz = kNegBig, mt = kNegBig;
    for (int c = 0; c < CS; c += 4) {
      const float4 v = interp4(s_log, s00, s01, s10, s11, c, lx0, lx1, ly0, ly1);
      mz = fmaxf(fmaxf(mz, fmaxf(v.x, v.y)), fmaxf(v.z, v.w));
    }
    for (int c = 0; c < KS; c += 4) {
      const float4 v = interp4(t_log, t00, t01, t10, t11, c, lx0, lx1, ly0, ly1);
      mt = fmaxf(fmaxf(mt, fmaxf(v.x, v.y)), fmaxf(v.z, v.w));
    }
    const float mzl = mz * kL2e, mtl = mt * kL2e;
    float s_all = 0.f, s_old = 0.f, s_bn = 0.f, z_lab = 0.f, st = 0.f, te0 = 0.f, T0 = 0.f, T1 = 0.f;
    for (int c = 0; c < CS; c += 4) {
      const float4 v4 = interp4(s_log, s00, s01, s10, s11, c, lx0, lx1, ly0, ly1);
      const float v[4] = {v4.x, v4.y, v4.z, v4.w};
      float tv[4] = {kNegBig, kNegBig, kNegBig, kNegBig};
      if (c < KS) {                                          // wave-uniform
        const float4 t4 = interp4(t_log, t00, t01, t10, t11, c, lx0, lx1, ly0, ly1);
        tv[0] = t4.x; tv[1] = t4.y; tv[2] = t4.z; tv[3] = t4.w;
      }
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int cc = c + j;
        const float e = __builtin_amdgcn_exp2f(__builtin_fmaf(v[j], kL2e, -mzl));
        s_all += e;
        s_old += cc < K ? e : 0.f;
        s_bn += (cc == 0 || cc >= K) ? e : 0.f;
        z_lab = cc == lab ? v[j] : z_lab;
        if (c < KS) {
          const float te = __builtin_amdgcn_exp2f(__builtin_fmaf(tv[j], kL2e, -mtl));   // 0 for the padding classes
          st += te;
          te0 = cc == 0 ? te : te0;
          const float tq = (cc >= 1 && cc < K) ? te : 0.f;
          T0 += tq;
          T1 = __builtin_fmaf(tq, v[j], T1);
        }
      }
    }
    const float den = mz + kLn2 * __builtin_amdgcn_logf(s_all);
    const float lse_old = mz + kLn2 * __builtin_amdgcn_logf(s_old), lse_bn = mz + kLn2 * __builtin_amdgcn_logf(s_bn);
    const bool lab0 = lab == 0;
    const float logp = lab0 ? lse_old - den : z_lab - den;
    if (!ignored) ce_sum += -logp;
    const float ce_w = ignored ? 0.f : ce_scale;
    float inv_st = 0.f, q0 = 0.f;
    if (sem_t) {
      inv_st = 1.f / st;
      q0 = te0 * inv_st;
      const float kd_pix = q0 * (lse_bn - den) + inv_st * (T1 - den * T0);
      kd_sum += -kd_pix * invK;
    }
    rmzl[it] = mzl;
    ra_all[it] = (ce_w + kdw) / s_all;
    ra_old[it] = lab0 ? ce_w / s_old : 0.f;
    rhot[it] = lab0 ? 0.f : ce_w;
    rlab[it] = lab;
    rbbn[it] = kdw * q0 / s_bn;
    rmtl[it] = mtl;
    rbq[it] = kdw * inv_st;
  }
  // ---- phase B: gradients, 16 classes at a time ------------------------------------------------------------------------------
  double* ga = g_acc + (threadIdx.x & (kRepWD - 1)) * gstride;
  for (int cb = 0; cb < CS; cb += 16) {
    float acc0[16], acc1[16];
    int cur_y0 = -1, cur_y1 = -1;
    auto flush = [&]() {
      if (cur_y0 < 0) return;
      const int r0 = (cur_y0 - ya) * nx, r1 = (cur_y1 - ya) * nx;
      double* g00 = ga + (r0 + cx0) * CS + cb; double* g01 = ga + (r0 + cx1) * CS + cb;
      double* g10 = ga + (r1 + cx0) * CS + cb; double* g11 = ga + (r1 + cx1) * CS + cb;
#pragma unroll
      for (int k = 0; k < 16; ++k)
        if (cb + k < Ctot) {
          atomicAdd(g00 + k, (double)(lx0 * acc0[k]));
          atomicAdd(g01 + k, (double)(lx1 * acc0[k]));
          atomicAdd(g10 + k, (double)(lx0 * acc1[k]));
          atomicAdd(g11 + k, (double)(lx1 * acc1[k]));
        }
    };
#pragma unroll
    for (int it = 0; it < kRW; ++it) {
      const int Y = Ybase + it;
      if (X >= W || Y >= H) continue;
      int y0, y1;
      float ly0, ly1;
      up_src(Y, h, scale_h, y0, y1, ly0, ly1);
      if (y0 != cur_y0 || y1 != cur_y1) {                    // wave-uniform: a wave's lanes share the row
        flush();
        cur_y0 = y0; cur_y1 = y1;
#pragma unroll
        for (int k = 0; k < 16; ++k) { acc0[k] = 0.f; acc1[k] = 0.f; }
      }
      const int r0 = (y0 - ya) * nx, r1 = (y1 - ya) * nx;
      const int s00 = (r0 + cx0) * CS, s01 = (r0 + cx1) * CS, s10 = (r1 + cx0) * CS, s11 = (r1 + cx1) * CS;
      const int t00 = (r0 + cx0) * KS, t01 = (r0 + cx1) * KS, t10 = (r1 + cx0) * KS, t11 = (r1 + cx1) * KS;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int c = cb + 4 * q;
        if (c >= CS) continue;                               // uniform
        const float4 v4 = interp4(s_log, s00, s01, s10, s11, c, lx0, lx1, ly0, ly1);
        const float v[4] = {v4.x, v4.y, v4.z, v4.w};
        float tv[4] = {kNegBig, kNegBig, kNegBig, kNegBig};
        if (c < KS) {
          const float4 t4 = interp4(t_log, t00, t01, t10, t11, c, lx0, lx1, ly0, ly1);
          tv[0] = t4.x; tv[1] = t4.y; tv[2] = t4.z; tv[3] = t4.w;
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const int cc = c + j;
          const float e = __builtin_amdgcn_exp2f(__builtin_fmaf(v[j], kL2e, -rmzl[it]));
          const float coef = ra_all[it] - (cc < K ? ra_old[it] : 0.f) - ((cc == 0 || cc >= K) ? rbbn[it] : 0.f);
          float g = e * coef - (cc == rlab[it] ? rhot[it] : 0.f);
          if (c < KS) {
            const float te = __builtin_amdgcn_exp2f(__builtin_fmaf(tv[j], kL2e, -rmtl[it]));
            g = __builtin_fmaf(-rbq[it], (cc >= 1 && cc < K) ? te : 0.f, g);
          }
          acc0[4 * q + j] = __builtin_fmaf(ly0, g, acc0[4 * q + j]);
          acc1[4 * q + j] = __builtin_fmaf(ly1, g, acc1[4 * q + j]);
        }
      }
    }
    flush();
  }
  __syncthreads();
  for (int i = threadIdx.x; i < ncell * CS; i += kThreads) {
    const int cell = i / CS, c = i - cell * CS;
    if (c >= Ctot) continue;
    double vd = g_acc[i];
#pragma unroll
    for (int r = 1; r < kRepWD; ++r) vd += g_acc[r * gstride + i];
    if (vd != 0.0) {       // fixed-point words, like the packed form: the same bits on every run (fx_scale 0: fp32 atomics)
      const int cy = ya + cell / nx, cx = xa + cell % nx;
      float* dst = d_sem + ((size_t)(b * h + cy) * w + cx) * ld_d + c;
      if (fx_scale > 0.f) atomicAdd(reinterpret_cast<int*>(dst), __double2int_rn(vd * (double)fx_scale));
      else atomicAdd(dst, (float)vd);
    }
  }
  ce_sum = wave_sum(ce_sum);
  kd_sum = wave_sum(kd_sum);
  if ((threadIdx.x & 63) == 0) { red[threadIdx.x >> 6] = ce_sum; red[4 + (threadIdx.x >> 6)] = kd_sum; }
  __syncthreads();
  if (threadIdx.x == 0) {
    const int blk = (blockIdx.z * tiles_y + blockIdx.y) * tiles_x + blockIdx.x;
    loss_part[2 * blk + 0] = red[0] + red[1] + red[2] + red[3];
    loss_part[2 * blk + 1] = red[4] + red[5] + red[6] + red[7];
  }
}

// ---- validation: up-sampling + arg-max + confusion matrix ---------------------------------------
// The reference's validation (train.py:242-246, metrics/stream_metrics.py:44-47,65-71) up-samples the logits, takes
// outputs.max(dim=1), copies predictions and labels to the host and runs a numpy bincount per image.  Here a block owns a
// 32 x 64 pixel tile like the loss kernel: the low-resolution cells under it are staged in LDS, every pixel's logits are
// interpolated in registers (the same arithmetic as above = torch's), the first maximum wins (torch.max's tie rule), and
// (label, prediction) pairs with 0 <= label < n are counted in an LDS histogram (n <= 64) or straight in global memory,
// as integers: the matrix is exact and order-independent.  pred (optional) receives the arg-max map.
__global__ __launch_bounds__(kThreads) void seg_confusion_kernel(const float* __restrict__ sem, int ld_s,
                                                                const int64_t* __restrict__ labels, int H, int W, int h, int w,
                                                                int Ctot, int n_classes, float scale_h, float scale_w,
                                                                unsigned long long* __restrict__ hist, int use_lds_hist,
                                                                int64_t* __restrict__ pred) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int b = blockIdx.z, ty0 = blockIdx.y * kTileY, tx0 = blockIdx.x * kTileX;
  int ya, yb, xa, xb, dummy;
  float f0, f1;
  up_src(ty0, h, scale_h, ya, dummy, f0, f1);
  up_src(min(ty0 + kTileY, H) - 1, h, scale_h, dummy, yb, f0, f1);
  up_src(tx0, w, scale_w, xa, dummy, f0, f1);
  up_src(min(tx0 + kTileX, W) - 1, w, scale_w, dummy, xb, f0, f1);
  const int ny = yb - ya + 1, nx = xb - xa + 1, ncell = ny * nx;
  float* s_log = smem;                                          // [ncell][Ctot]
  unsigned int* l_hist = reinterpret_cast<unsigned int*>(s_log + ncell * Ctot);   // [n][n] when use_lds_hist
  for (int i = threadIdx.x; i < ncell * Ctot; i += kThreads) {
    const int cell = i / Ctot, c = i - cell * Ctot;
    const int cy = ya + cell / nx, cx = xa + cell % nx;
    s_log[i] = sem[((size_t)(b * h + cy) * w + cx) * ld_s + c];
  }
  if (use_lds_hist)
    for (int i = threadIdx.x; i < n_classes * n_classes; i += kThreads) l_hist[i] = 0u;
  __syncthreads();
  const int X = tx0 + (threadIdx.x & 63);
  int x0 = 0, x1 = 0;
  float lx0 = 0.f, lx1 = 0.f;
  if (X < W) up_src(X, w, scale_w, x0, x1, lx0, lx1);
  for (int it = 0; it < kRows; ++it) {
    const int Y = ty0 + (threadIdx.x >> 6) * kRows + it;
    if (X >= W || Y >= H) continue;
    int y0, y1;
    float ly0, ly1;
    up_src(Y, h, scale_h, y0, y1, ly0, ly1);
    const int c00 = (y0 - ya) * nx + (x0 - xa), c01 = (y0 - ya) * nx + (x1 - xa);
    const int c10 = (y1 - ya) * nx + (x0 - xa), c11 = (y1 - ya) * nx + (x1 - xa);
    float best = -INFINITY;
    int arg = 0;
    for (int c = 0; c < Ctot; ++c) {
      const float z = ly0 * (lx0 * s_log[c00 * Ctot + c] + lx1 * s_log[c01 * Ctot + c]) +
                      ly1 * (lx0 * s_log[c10 * Ctot + c] + lx1 * s_log[c11 * Ctot + c]);
      if (z > best) { best = z; arg = c; }
    }
    const size_t pix = ((size_t)b * H + Y) * W + X;
    if (pred) pred[pix] = arg;
    const int64_t lab = labels[pix];
    if (lab >= 0 && lab < n_classes && arg < n_classes) {
      if (use_lds_hist) atomicAdd(&l_hist[(int)lab * n_classes + arg], 1u);
      else atomicAdd(&hist[(size_t)lab * n_classes + arg], 1ull);
    }
  }
  if (use_lds_hist) {
    __syncthreads();
    for (int i = threadIdx.x; i < n_classes * n_classes; i += kThreads) {
      const unsigned int v = l_hist[i];
      if (v) atomicAdd(&hist[i], (unsigned long long)v);
    }
  }
}

// fixed-point gradient words (seg_losses_pk_kernel) -> fp32, in place
__global__ __launch_bounds__(kThreads) void seg_grad_unfix_kernel(float* __restrict__ d, size_t n, double inv_scale) {
  const size_t i = ((size_t)blockIdx.x * kThreads + threadIdx.x) * 4;
  if (i + 4 <= n) {
    const int4 v = *reinterpret_cast<const int4*>(d + i);
    *reinterpret_cast<float4*>(d + i) = make_float4((float)((double)v.x * inv_scale), (float)((double)v.y * inv_scale),
                                                    (float)((double)v.z * inv_scale), (float)((double)v.w * inv_scale));
  } else {
    for (size_t j = i; j < n; ++j) d[j] = (float)((double)reinterpret_cast<const int*>(d)[j] * inv_scale);
  }
}

__global__ __launch_bounds__(1024) void seg_losses_reduce_kernel(const float* __restrict__ part, int n, float inv_pix,
                                                                float* __restrict__ out) {
  __shared__ double red[2][16];
  double a = 0.0, b = 0.0;
  for (int i = threadIdx.x; i < n; i += 1024) { a += part[2 * i]; b += part[2 * i + 1]; }
  for (int off = 32; off > 0; off >>= 1) { a += __shfl_xor(a, off, 64); b += __shfl_xor(b, off, 64); }
  if ((threadIdx.x & 63) == 0) { red[0][threadIdx.x >> 6] = a; red[1][threadIdx.x >> 6] = b; }
  __syncthreads();
  if (threadIdx.x == 0) {
    double sa = 0.0, sb = 0.0;
    for (int i = 0; i < 16; ++i) { sa += red[0][i]; sb += red[1][i]; }
    out[0] = (float)(sa * inv_pix);   // mean over ALL pixels, ignored ones count as 0 (train.py:116 .mean())
    out[1] = (float)(sb * inv_pix);   // KD mean over all pixels (loss.py:178)
  }
}

}  // namespace
}  // namespace ucd

using namespace ucd;

extern "C" {

size_t ucd_seg_losses_workspace_bytes(int B, int H, int W) {
  return (size_t)B * ceil_div(H, 4 * kRW) * ceil_div(W, kTileX) * 2 * sizeof(float);      // the finer tiling of the two kernels
}

int ucd_seg_losses(const float* sem_s, int ld_s, const float* sem_t, int ld_t, const int64_t* labels, int B, int H, int W,
                   int h, int w, int Ctot, int K, int ignore_index, float ce_weight, float kd_weight, float* loss_out,
                   float* d_sem, int ld_d, void* workspace, size_t workspace_bytes, ucd_stream_t stream) {
  static const char* fn = "ucd_seg_losses";
  UCD_REQUIRE(sem_s && labels && loss_out && d_sem && workspace, UCD_EINVAL, "%s: NULL argument", fn);
  UCD_REQUIRE(B > 0 && H > 0 && W > 0 && h > 0 && w > 0 && Ctot > 0 && K >= 1 && K <= Ctot, UCD_EINVAL, "%s: bad sizes", fn);
  UCD_REQUIRE(ld_s >= Ctot && ld_d >= Ctot && (!sem_t || ld_t >= K), UCD_EINVAL, "%s: bad leading dimension", fn);
  UCD_REQUIRE(H >= 8 * h || H >= h, UCD_EINVAL, "%s: bad scale", fn);
  UCD_REQUIRE((float)H / h >= 4.f && (float)W / w >= 4.f, UCD_EUNSUPPORTED,
              "%s: built for up-sampling factors >= 4 (the model's is 16)", fn);
  // the fixed-point gradient words (scale 2^17 over one pixel's largest gradient) hold the bilinear weights of at most factor^2 pixels
  // per low-resolution cell: 64^2 x 2^17 = 2^29 < 2^31; beyond a factor of 64 an int32 word could wrap silently
  UCD_REQUIRE((float)H / h <= 64.f && (float)W / w <= 64.f, UCD_EUNSUPPORTED, "%s: up-sampling factors above 64 are not supported", fn);
  UCD_REQUIRE(workspace_bytes >= ucd_seg_losses_workspace_bytes(B, H, W), UCD_EWORKSPACE, "%s: workspace too small", fn);
  hipStream_t s = (hipStream_t)stream;
  const bool wide = Ctot > 24;                       // many classes (ADE): seg_losses_wide_kernel on 32 x 64 pixel tiles
  const int tile_y = wide ? 4 * kRW : kTileY;
  const int tiles_x = ceil_div(W, kTileX), tiles_y = ceil_div(H, tile_y);
  // worst-case LDS: (tile/scale + 3) cells per dimension
  int ny = (int)(tile_y * (float)h / H) + 3, nx = (int)(kTileX * (float)w / W) + 3;
  {
    // the exact footprint instead of the bound: the kernels' own source-index arithmetic (up_src, same float operations) over
    // every tile origin - 6 x 6 cells instead of 7 x 7 at 513 / 33, which is what lets two workgroups share a CU's LDS
    auto lo = [](int dst, int in_size, float scale) {
      float src = scale * ((float)dst + 0.5f) - 0.5f;
      src = src < 0.f ? 0.f : src;
      const int i0 = (int)src < in_size - 1 ? (int)src : in_size - 1;
      return i0;
    };
    auto span = [&](int out, int in_size, float scale, int tile) {
      int m = 1;
      for (int t0 = 0; t0 < out; t0 += tile) {
        const int a = lo(t0, in_size, scale);
        const int last = (t0 + tile < out ? t0 + tile : out) - 1;
        const int i0 = lo(last, in_size, scale);
        const int bnd = i0 + (i0 < in_size - 1 ? 1 : 0);
        if (bnd - a + 1 > m) m = bnd - a + 1;
      }
      return m;
    };
    const int ey = span(H, h, (float)h / (float)H, tile_y), ex = span(W, w, (float)w / (float)W, kTileX);
    if (ey < ny) ny = ey;
    if (ex < nx) nx = ex;
  }
  const int CS = (Ctot + 3) & ~3, KS = sem_t ? ((K + 3) & ~3) : 0;
  // few classes: the packed form with old / new classes in separate slot groups (UCD_SEG_PK=0: the round-3 register form, A/B)
  static const int pk_on = getenv("UCD_SEG_PK") ? atoi(getenv("UCD_SEG_PK")) : 1;
  int pk_kt = 0, pk_nt = 0;
  if (pk_on && !wide && (reinterpret_cast<uintptr_t>(d_sem) & 15) == 0) {       // (its fixed-point words are converted four at a time)
    const int nnew = Ctot - K;
    if (K <= 16 && nnew <= 8) { pk_kt = 16; pk_nt = 8; }
    else if (K <= 20 && nnew <= 4) { pk_kt = 20; pk_nt = 4; }
    else if (K <= 12 && nnew <= 12) { pk_kt = 12; pk_nt = 12; }
  }
  const size_t lds = wide ? ((size_t)ny * nx * (CS + KS) + 8) * sizeof(float) + (size_t)kRepWD * (((ny * nx * CS + 31) & ~31) + 4) * 8
                   : pk_kt ? (size_t)kRepPk * (((ny * nx * (pk_kt + pk_nt) + 31) & ~31) + 1) * 8 + 8
                                 + ((size_t)ny * nx * ((pk_kt + pk_nt) + pk_kt + 2) + 8) * sizeof(float) + kRows * kThreads
                          : ((size_t)ny * nx * (17 * Ctot + K) + 8) * sizeof(float);
  UCD_REQUIRE(lds <= 150 * 1024, UCD_EUNSUPPORTED, "%s: %d classes exceed the LDS budget", fn, Ctot);
  UCD_TRY_LDS((seg_losses_pk_kernel<16, 8>), 150 * 1024);
  UCD_TRY_LDS((seg_losses_pk_kernel<20, 4>), 150 * 1024);
  UCD_TRY_LDS((seg_losses_pk_kernel<12, 12>), 150 * 1024);
  UCD_TRY_LDS((seg_losses_kernel<24, 16>), 150 * 1024);
  UCD_TRY_LDS((seg_losses_kernel<24, 24>), 150 * 1024);
  UCD_TRY_LDS(seg_losses_wide_kernel, 150 * 1024);
  hipError_t e = hipMemsetAsync(d_sem, 0, (size_t)B * h * w * ld_d * sizeof(float), s);
  if (e != hipSuccess) { set_error("%s: %s", fn, hipGetErrorString(e)); return (int)e; }
  const float inv_pix = 1.f / ((float)B * H * W);
  float* part = (float*)workspace;
  // torch computes the up-sampling scale as float(in) / out
  // fixed-point scale of the packed form's gradient words: one pixel's gradient is at most ce + 2 kd / K in magnitude
  const float fx_gmax = fabsf(ce_weight * inv_pix) + 2.f * fabsf(kd_weight * inv_pix) / (float)K;
  const float fx_scale = fx_gmax > 0.f ? 131072.f / fx_gmax : 1.f;
  const bool fx_wide = wide && (reinterpret_cast<uintptr_t>(d_sem) & 15) == 0;
#define UCD_SEG_PK_LAUNCH(KT_, NT_)                                                                                              \
  seg_losses_pk_kernel<KT_, NT_><<<dim3(tiles_x, tiles_y, B), kThreads, lds, s>>>(                                               \
      sem_s, ld_s, sem_t, ld_t, labels, H, W, h, w, Ctot, K, ignore_index, (float)h / (float)H, (float)w / (float)W,             \
      ce_weight * inv_pix, kd_weight * inv_pix, part, d_sem, ld_d, tiles_x, tiles_y, ny * nx, fx_scale)
  if (pk_kt == 16) UCD_SEG_PK_LAUNCH(16, 8);
  else if (pk_kt == 20) UCD_SEG_PK_LAUNCH(20, 4);
  else if (pk_kt == 12) UCD_SEG_PK_LAUNCH(12, 12);
#undef UCD_SEG_PK_LAUNCH
  else if (Ctot <= 24 && K <= 16)
    seg_losses_kernel<24, 16><<<dim3(tiles_x, tiles_y, B), kThreads, lds, s>>>(
        sem_s, ld_s, sem_t, ld_t, labels, H, W, h, w, Ctot, K, ignore_index, (float)h / (float)H, (float)w / (float)W,
        ce_weight * inv_pix, kd_weight * inv_pix, part, d_sem, ld_d, tiles_x, tiles_y);
  else if (Ctot <= 24)
    seg_losses_kernel<24, 24><<<dim3(tiles_x, tiles_y, B), kThreads, lds, s>>>(
        sem_s, ld_s, sem_t, ld_t, labels, H, W, h, w, Ctot, K, ignore_index, (float)h / (float)H, (float)w / (float)W,
        ce_weight * inv_pix, kd_weight * inv_pix, part, d_sem, ld_d, tiles_x, tiles_y);
  else
    seg_losses_wide_kernel<<<dim3(tiles_x, tiles_y, B), kThreads, lds, s>>>(
        sem_s, ld_s, sem_t, ld_t, labels, H, W, h, w, Ctot, K, ignore_index, (float)h / (float)H, (float)w / (float)W,
        ce_weight * inv_pix, kd_weight * inv_pix, part, d_sem, ld_d, tiles_x, tiles_y, fx_wide ? fx_scale : 0.f);
  int rc = check_launch(fn);
  if (rc) return rc;
  if (pk_kt || fx_wide) {
    const size_t n = (size_t)B * h * w * ld_d;
    seg_grad_unfix_kernel<<<(unsigned)((n / 4 + kThreads) / kThreads), kThreads, 0, s>>>(d_sem, n, 1.0 / (double)fx_scale);
    rc = check_launch(fn);
    if (rc) return rc;
  }
  seg_losses_reduce_kernel<<<1, 1024, 0, s>>>(part, B * tiles_x * tiles_y, inv_pix, loss_out);
  return check_launch(fn);
}

int ucd_seg_confusion(const float* sem, int ld_s, const int64_t* labels, int B, int H, int W, int h, int w, int Ctot,
                      int n_classes, int64_t* hist, int64_t* pred, ucd_stream_t stream) {
  static const char* fn = "ucd_seg_confusion";
  UCD_REQUIRE(sem && labels && hist, UCD_EINVAL, "%s: NULL argument", fn);
  UCD_REQUIRE(B > 0 && H > 0 && W > 0 && h > 0 && w > 0 && Ctot > 0 && n_classes > 0 && ld_s >= Ctot, UCD_EINVAL, "%s: bad sizes", fn);
  UCD_REQUIRE((float)H / h >= 4.f && (float)W / w >= 4.f, UCD_EUNSUPPORTED,
              "%s: built for up-sampling factors >= 4 (the model's is 16)", fn);
  const int tiles_x = ceil_div(W, kTileX), tiles_y = ceil_div(H, kTileY);
  const int ny = (int)(kTileY * (float)h / H) + 3, nx = (int)(kTileX * (float)w / W) + 3;
  const int use_lds_hist = n_classes <= 64;
  const size_t lds = ((size_t)ny * nx * Ctot) * sizeof(float) + (use_lds_hist ? (size_t)n_classes * n_classes * 4 : 0);
  UCD_REQUIRE(lds <= 150 * 1024, UCD_EUNSUPPORTED, "%s: %d classes exceed the LDS budget", fn, Ctot);
  UCD_TRY_LDS(seg_confusion_kernel, 150 * 1024);
  seg_confusion_kernel<<<dim3(tiles_x, tiles_y, B), kThreads, lds, (hipStream_t)stream>>>(
      sem, ld_s, labels, H, W, h, w, Ctot, n_classes, (float)h / (float)H, (float)w / (float)W, (unsigned long long*)hist,
      use_lds_hist, pred);
  return check_launch(fn);
}

}  // extern "C"
