// The stem's norm + pooling as one pass (models/resnet.py:58-64: mod1 = conv1 7x7/2 -> norm_act -> MaxPool2d(3, stride 2, padding 1)).
//
// The 7x7 convolution leaves the LARGEST activation of the network ([B, 64, 257, 257] at 513^2: 203 MB in bf16 for B = 24); the
// layer-by-layer path writes its normalised copy (203 MB), the pooling reads that back and writes a quarter of it, and the
// backward walks the same maps three more times (pooling backward, the norm's reduction, the norm's apply).  Here
//   forward   out[b, ph, pw, c] = max over the 3x3 window (stride 2, padding 1) of act((z - mean_c) scale_c + beta_c), each value
//             rounded to bf16 before the comparison (so the result - and the first-maximum tie rule - is bit for bit that of
//             max_pool2d(abn(z))); the normalised map is never written; idx keeps the window position (0..8) of the maximum;
//   backward  the gradient of the pooled map reaches z only at the arg-max positions:  dyact = dpool * act'(.) there, 0 elsewhere;
//             phase 1 sums  dyact  and  dyact * xhat  per channel over the pooled map (a quarter of the positions), phase 2 writes
//             dz = gamma~ invstd (dyact - mean(dyact) - xhat mean(dyact xhat)) for every position of z, gathering dyact from the
//             (at most four) windows a position belongs to through idx.  Between the phases SyncBN all-reduces the two sums.
// One thread owns 8 channels (16 bytes) of one pixel; maps are dense channels-last bf16.
#include "common.h"

namespace ucd {
namespace {

typedef __hip_bfloat16 bf16;
constexpr int kThreads = 256;
constexpr int kRedBlocks = 1024;      // workgroups (= partial rows) of the backward reduction
constexpr int kApplyRows = 4;         // the backward apply gives one thread 4 rows x 2 columns x 8 channels of z

__device__ __forceinline__ float leaky(float z, float slope) { return z > 0.f ? z : z * slope; }
__device__ __forceinline__ float bf16_round(float v) { return __bfloat162float(__float2bfloat16(v)); }

struct StemArgs {
  const bf16* z; int B, H, W, C, PH, PW;
  const float *mean, *scale, *beta, *invstd, *weight, *sums;
  float slope, inv_count;
  int abs_gamma, cg_shift;
};

// window position (kh, kw) of pooled pixel (ph, pw) -> input pixel (2 ph - 1 + kh, 2 pw - 1 + kw)
__global__ __launch_bounds__(kThreads) void stem_apply_pool_kernel(StemArgs p, bf16* __restrict__ out, uint8_t* __restrict__ idx) {
  // grid (row chunks, PH, B): 32-bit index arithmetic only, the channel-group count is a power of two
  const int cgs = p.cg_shift, CG = 1 << cgs;
  const int t = blockIdx.x * kThreads + threadIdx.x;
  const int cg = t & (CG - 1), pw = t >> cgs, ph = blockIdx.y, b = blockIdx.z;
  if (pw >= p.PW) return;
  const size_t pix = ((size_t)b * p.PH + ph) * p.PW + pw;
  float mu[8], sc[8], sh[8], best[8];
  int bi[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    mu[e] = p.mean[cg * 8 + e]; sc[e] = p.scale[cg * 8 + e]; sh[e] = p.beta ? p.beta[cg * 8 + e] : 0.f;
    best[e] = -INFINITY; bi[e] = 0;
  }
  // all nine loads are issued before the first use: positions outside the map load a clamped (valid) address and are masked after
  Vec<bf16> v[9];
#pragma unroll
  for (int kh = 0; kh < 3; ++kh) {
    const int y = min(max(2 * ph - 1 + kh, 0), p.H - 1);
#pragma unroll
    for (int kw = 0; kw < 3; ++kw) {
      const int x = min(max(2 * pw - 1 + kw, 0), p.W - 1);
      v[kh * 3 + kw].load(p.z + (((size_t)b * p.H + y) * p.W + x) * p.C + cg * 8);
    }
  }
#pragma unroll
  for (int kh = 0; kh < 3; ++kh) {
    const bool yin = (unsigned)(2 * ph - 1 + kh) < (unsigned)p.H;
#pragma unroll
    for (int kw = 0; kw < 3; ++kw) {
      const bool in = yin && (unsigned)(2 * pw - 1 + kw) < (unsigned)p.W;
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const float a = bf16_round(leaky((v[kh * 3 + kw].get(e) - mu[e]) * sc[e] + sh[e], p.slope));   // what the un-fused apply pass stores
        if (in && a > best[e]) { best[e] = a; bi[e] = kh * 3 + kw; }                                  // first maximum wins (max_pool2d)
      }
    }
  }
  Vec<bf16> o;
#pragma unroll
  for (int e = 0; e < 8; ++e) o.set(e, best[e]);
  o.store(out + pix * p.C + cg * 8);
  if (idx) {
    uint2 w;
    w.x = (unsigned)bi[0] | ((unsigned)bi[1] << 8) | ((unsigned)bi[2] << 16) | ((unsigned)bi[3] << 24);
    w.y = (unsigned)bi[4] | ((unsigned)bi[5] << 8) | ((unsigned)bi[6] << 16) | ((unsigned)bi[7] << 24);
    *reinterpret_cast<uint2*>(idx + pix * p.C + cg * 8) = w;
  }
}

// phase 1: partial[block][0..C) = sum dyact, partial[block][C..2C) = sum dyact * xhat over the pooled pixels of the block
__global__ __launch_bounds__(kThreads) void stem_pool_bwd_reduce_kernel(StemArgs p, const bf16* __restrict__ dpool,
                                                                       const uint8_t* __restrict__ idx, float* __restrict__ partial) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int CG = p.C / 8, PL = kThreads / CG;           // pixel lanes per workgroup
  const int cg = threadIdx.x % CG, pl = threadIdx.x / CG;
  const long long npix = (long long)p.B * p.PH * p.PW;
  float acc[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) acc[i] = 0.f;
  if (pl < PL) {
    float mu[8], sc[8], sh[8], is[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      mu[e] = p.mean[cg * 8 + e]; sc[e] = p.scale[cg * 8 + e]; sh[e] = p.beta ? p.beta[cg * 8 + e] : 0.f; is[e] = p.invstd[cg * 8 + e];
    }
    for (int pix = blockIdx.x * PL + pl; pix < (int)npix; pix += gridDim.x * PL) {
      const int rowi = pix / p.PW;                              // one 32-bit division per pixel and thread
      const int pw = pix - rowi * p.PW, b = rowi / p.PH, ph = rowi - b * p.PH;
      Vec<bf16> g;
      g.load(dpool + (size_t)pix * p.C + cg * 8);
      const uint2 w = *reinterpret_cast<const uint2*>(idx + (size_t)pix * p.C + cg * 8);
      float zsel[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
      Vec<bf16> v[9];                      // an arg-max position is always inside the map, so clamped loads need no mask
#pragma unroll
      for (int kh = 0; kh < 3; ++kh) {
        const int y = min(max(2 * ph - 1 + kh, 0), p.H - 1);
#pragma unroll
        for (int kw = 0; kw < 3; ++kw) {
          const int x = min(max(2 * pw - 1 + kw, 0), p.W - 1);
          v[kh * 3 + kw].load(p.z + (((size_t)b * p.H + y) * p.W + x) * p.C + cg * 8);
        }
      }
#pragma unroll
      for (int pos = 0; pos < 9; ++pos)
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const unsigned id = ((e < 4 ? w.x : w.y) >> (8 * (e & 3))) & 0xffu;
          zsel[e] = id == (unsigned)pos ? v[pos].get(e) : zsel[e];
        }
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const float xc = zsel[e] - mu[e];
        const float dz = g.get(e) * (xc * sc[e] + sh[e] > 0.f ? 1.f : p.slope);
        acc[e] += dz;
        acc[8 + e] += dz * (xc * is[e]);
      }
    }
  }
  // reduce over the pixel lanes (fixed order), lanes past PL carry zeros
#pragma unroll
  for (int i = 0; i < 16; ++i) lds[threadIdx.x * 16 + i] = acc[i];
  __syncthreads();
  if (pl == 0) {
    for (int l = 1; l < PL; ++l)
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[i] += lds[(l * CG + cg) * 16 + i];
    float* dst = partial + (size_t)blockIdx.x * 2 * p.C + cg * 8;
#pragma unroll
    for (int e = 0; e < 8; ++e) { dst[e] = acc[e]; dst[p.C + e] = acc[8 + e]; }
  }
}

// phase 2: dz for every position of z
__global__ __launch_bounds__(kThreads) void stem_pool_bwd_apply_kernel(StemArgs p, const bf16* __restrict__ dpool,
                                                                      const uint8_t* __restrict__ idx, bf16* __restrict__ dz) {
  const int cgs = p.cg_shift, CG = 1 << cgs;
  const int t = blockIdx.x * kThreads + threadIdx.x;
  const int cg = t & (CG - 1), i = t >> cgs, b = blockIdx.z;     // i: column pair (2 i, 2 i + 1)
  if (2 * i >= p.W) return;
  // per-channel constants of this thread's 8 channels, once for the 4 x 2 positions it owns
  float mu[8], sc[8], sh[8], is[8], k0[8], k1[8], gw[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    const int ch = cg * 8 + e;
    mu[e] = p.mean[ch]; sc[e] = p.scale[ch]; sh[e] = p.beta ? p.beta[ch] : 0.f; is[e] = p.invstd[ch];
    k0[e] = p.sums[ch] * p.inv_count;
    k1[e] = p.sums[p.C + ch] * p.inv_count;
    if (p.abs_gamma) {                 // sums[C + c] holds sign(weight) * sum dz xhat; scale = (|w| + eps) invstd  (abn_bwd_apply_kernel)
      if (p.weight && p.weight[ch] < 0.f) k1[e] = -k1[e];
      gw[e] = sc[e];
    } else {
      gw[e] = (p.weight ? p.weight[ch] : 1.f) * is[e];
    }
  }
  // Rows 4 j .. 4 j + 3 lie in the windows of pooled rows 2 j, 2 j + 1, 2 j + 2 only (an even row 2 k in window k at kh = 1, an odd
  // row 2 k + 1 in window k at kh = 2 and in window k + 1 at kh = 0); the same for the column pair.  So 3 x 2 gathers of
  // (idx, dpool) serve 4 x 2 positions, all loads - clamped into the maps, masked afterwards - are issued before the first use.
  const int phb = blockIdx.y * 2, y0 = blockIdx.y * kApplyRows, x0 = 2 * i;
  Vec<bf16> g[3][2], v[kApplyRows][2];
  uint2 w[3][2];
  bool gv[3][2];
#pragma unroll
  for (int a = 0; a < 3; ++a)
#pragma unroll
    for (int c = 0; c < 2; ++c) {
      gv[a][c] = phb + a < p.PH && i + c < p.PW;
      const size_t off = (((size_t)b * p.PH + min(phb + a, p.PH - 1)) * p.PW + min(i + c, p.PW - 1)) * p.C + cg * 8;
      w[a][c] = *reinterpret_cast<const uint2*>(idx + off);
      g[a][c].load(dpool + off);
    }
#pragma unroll
  for (int r = 0; r < kApplyRows; ++r)
#pragma unroll
    for (int cx = 0; cx < 2; ++cx)
      v[r][cx].load(p.z + (((size_t)b * p.H + min(y0 + r, p.H - 1)) * p.W + min(x0 + cx, p.W - 1)) * p.C + cg * 8);
#pragma unroll
  for (int r = 0; r < kApplyRows; ++r) {
    if (y0 + r >= p.H) break;                                    // uniform over the workgroup
#pragma unroll
    for (int cx = 0; cx < 2; ++cx) {
      if (x0 + cx >= p.W) continue;
      float dy[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int ya = 0; ya < 2; ++ya) {                           // the (at most two) windows over the row ...
        if (ya == 1 && !(r & 1)) continue;
        const int a = (r >> 1) + ya, kh = (r & 1) ? (ya ? 0 : 2) : 1;
#pragma unroll
        for (int xa = 0; xa < 2; ++xa) {                         // ... and over the column
          if (xa == 1 && !cx) continue;
          const int kw = cx ? (xa ? 0 : 2) : 1;
          const unsigned pos = gv[a][xa] ? (unsigned)(kh * 3 + kw) : 0xffu;    // 0xff never equals a stored position
#pragma unroll
          for (int e = 0; e < 8; ++e) {
            const unsigned id = ((e < 4 ? w[a][xa].x : w[a][xa].y) >> (8 * (e & 3))) & 0xffu;
            dy[e] += id == pos ? g[a][xa].get(e) : 0.f;
          }
        }
      }
      Vec<bf16> o;
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const float xc = v[r][cx].get(e) - mu[e];
        const float dza = dy[e] * (xc * sc[e] + sh[e] > 0.f ? 1.f : p.slope);
        o.set(e, (dza - k0[e] - (xc * is[e]) * k1[e]) * gw[e]);
      }
      o.store(dz + (((size_t)b * p.H + y0 + r) * p.W + x0 + cx) * p.C + cg * 8);
    }
  }
}

// sums[c] / sums[C + c] over the kRedBlocks partial rows (fixed order); abs-gamma layers carry sign(weight) on the second row
__global__ __launch_bounds__(kThreads) void stem_sum_partials_kernel(const float* __restrict__ partial, int rows, int C,
                                                                    const float* __restrict__ sign_of, float* __restrict__ sums) {
  __shared__ float red[kThreads];
  const int k = blockIdx.x;                       // one workgroup per output value (2 C of them)
  float s = 0.f;
  for (int r = threadIdx.x; r < rows; r += kThreads) s += partial[(size_t)r * 2 * C + k];
  red[threadIdx.x] = s;
  __syncthreads();
  for (int st = kThreads / 2; st > 0; st >>= 1) {
    if (threadIdx.x < st) red[threadIdx.x] += red[threadIdx.x + st];
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    float t = red[0];
    if (sign_of && k >= C && sign_of[k - C] < 0.f) t = -t;
    sums[k] = t;
  }
}

// ---- the stem's 7x7 / 2 convolution (models/resnet.py:58: conv1 = Conv2d(3, 64, 7, stride 2, padding 3), no bias) ---------------------
// z[b, oy, ox, n] = sum_{ky, kx, c} x[b, c, 2 oy - 3 + ky, 2 ox - 3 + kx] * w[n, c, ky, kx]   fp32 image in (any strides), bf16 out.
// An implicit GEMM with K = 7 rows x (7 x 3 = 21 -> 24) = 168 -> 176: a workgroup stages the (2 TH + 5) x (2 TW + 5) x 3 patch of
// its TH x TW = 8 x 32 output pixels in LDS as bf16 [row][pixel][channel] - the 21 values of one kernel row of one output pixel
// are CONTIGUOUS there (6 ox elements from the row start), so an MFMA operand (8 consecutive k of one pixel) is four aligned
// 32-bit LDS reads - and the weights as [n][ky * 24 + kx * 3 + c] (zero where kx * 3 + c >= 21).  Operands are swapped (A =
// weights, B = pixels): a lane then holds 4 consecutive channels of one pixel per accumulator group and stores 8 bytes at a time.
// Persistent workgroups (the 22 KB weight image is built once per workgroup), one tile per iteration.
constexpr int kSTH = 8, kSTW = 32;                    // output tile
constexpr int kSIR = 2 * kSTH + 5;                    // 21 input rows
constexpr int kSIP = 216;                             // bf16 pitch of an input row: (2 * 32 + 5) * 3 = 207, + the over-read of the last pixel
constexpr int kSWP = 184;                             // bf16 pitch of a weight row: 176 + 8 (an odd number of 16-byte slots)
constexpr int kSK = 176;

struct StemConvArgs {
  const float* x; long long sb, sc, sh, sw;           // element strides of the fp32 image [B, 3, H, W]
  const bf16* w;                                      // [64][7][7][3] (channels-last memory order of the [64, 3, 7, 7] weight)
  bf16* z;                                            // [B, OH, OW, 64]
  int B, H, W, OH, OW, tiles_y, tiles_x, ntiles;
};

__global__ __launch_bounds__(kThreads) void stem_conv7x7_kernel(StemConvArgs p) {
  typedef float f32x16 __attribute__((ext_vector_type(16)));
  typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
  __shared__ __attribute__((aligned(16))) unsigned short Wl[64 * kSWP];
  __shared__ __attribute__((aligned(16))) unsigned short In[kSIR * kSIP];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  // weights: zero image, then the 147 real values of every output channel
  for (int i = tid; i < 64 * kSWP; i += kThreads) Wl[i] = 0;
  __syncthreads();
  for (int i = tid; i < 64 * 147; i += kThreads) {
    const int n = i / 147, r = i - n * 147, ky = r / 21, j = r - ky * 21;
    Wl[n * kSWP + ky * 24 + j] = reinterpret_cast<const unsigned short*>(p.w)[i];
  }
  const int col = lane & 31, h = lane >> 5;
  // The patch of a tile: thread e < 216 owns column element e (pixel e / 3, channel e % 3 - constants of the thread) of all 21
  // rows; ALL loads are issued before the first use - and issued for the NEXT tile before this tile's MFMAs, so they land under
  // them (the first version loaded and stored element by element: 18 exposed round trips per tile, 20 us per tile and workgroup).
  const int fe = tid < kSIP ? tid : kSIP - 1, fpx = fe / 3, fc = fe - fpx * 3;
  float pre[kSIR];
  unsigned premask = 0;                                // rows of `pre` that lie inside the image
  auto fetch = [&](int tile) {
    unsigned okm = 0;
    const int b = tile / (p.tiles_y * p.tiles_x), rem = tile - b * p.tiles_y * p.tiles_x;
    const int oy0 = (rem / p.tiles_x) * kSTH, ox0 = (rem % p.tiles_x) * kSTW;
    const int ix = 2 * ox0 - 3 + fpx;
    const bool colok = fpx < 2 * kSTW + 5 && (unsigned)ix < (unsigned)p.W;
    const float* src = p.x + b * p.sb + fc * p.sc + (colok ? ix : 0) * p.sw;
#pragma unroll
    for (int r = 0; r < kSIR; ++r) {
      const int iy = 2 * oy0 - 3 + r;
      const bool ok = colok && (unsigned)iy < (unsigned)p.H;        // clamped address + select: no branch around the load
      const float v = src[(ok ? iy : 0) * p.sh];
      // the select HERE makes the compiler wait for all 21 loads in front of the MFMAs (vmcnt(0)) - measured better for this kernel
      // (102 vs 126 us: three workgroups per CU cover the round trip, and the tile's 16 stores do not queue behind the loads); the
      // one-kernel frozen stem below, two workgroups per CU and no global stores of the tile, defers it (188 -> 142 us)
      pre[r] = ok ? v : 0.f;
      okm |= 1u << r;
    }
    premask = okm;
  };
  if ((int)blockIdx.x < p.ntiles) fetch(blockIdx.x);
  for (int tile = blockIdx.x; tile < p.ntiles; tile += gridDim.x) {
    const int b = tile / (p.tiles_y * p.tiles_x), rem = tile - b * p.tiles_y * p.tiles_x;
    const int oy0 = (rem / p.tiles_x) * kSTH, ox0 = (rem % p.tiles_x) * kSTW;
    __syncthreads();                                   // the previous tile's reads are done (and the weight image is complete)
    if (tid < kSIP) {
#pragma unroll
      for (int r = 0; r < kSIR; ++r) {
        const bf16 v = __float2bfloat16((premask >> r) & 1u ? pre[r] : 0.f);
        In[r * kSIP + tid] = *reinterpret_cast<const unsigned short*>(&v);
      }
    }
    __syncthreads();
    if (tile + (int)gridDim.x < p.ntiles) fetch(tile + gridDim.x);
    f32x16 acc[2][2];                                  // [output row of this wave][channel block]
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
      for (int c = 0; c < 2; ++c)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[a][c][r] = 0.f;
#pragma unroll
    for (int s = 0; s < kSK / 16; ++s) {
      // this lane's 8 k: slot t = 2 s + h of the 22 eight-wide slots; kernel row t / 3, offset 8 (t % 3) inside its 24
      const int t0 = 2 * s, t1 = 2 * s + 1;
      // (slot 21 is all padding - zero weights - and must still read FINITE patch values: kernel row clamped to 6)
      const int ky = h ? (t1 / 3 > 6 ? 6 : t1 / 3) : t0 / 3, j0 = h ? 8 * (t1 % 3) : 8 * (t0 % 3);
      bf16x8 wf[2], pf[2];
#pragma unroll
      for (int c = 0; c < 2; ++c) wf[c] = *reinterpret_cast<const bf16x8*>(&Wl[(c * 32 + col) * kSWP + 16 * s + 8 * h]);
#pragma unroll
      for (int a = 0; a < 2; ++a) {
        const int orow = wave * 2 + a;
        const unsigned* src = reinterpret_cast<const unsigned*>(&In[(2 * orow + ky) * kSIP + 6 * col + j0]);
        union { unsigned u[4]; bf16x8 v; } f;
        f.u[0] = src[0]; f.u[1] = src[1]; f.u[2] = src[2]; f.u[3] = src[3];
        pf[a] = f.v;
      }
#pragma unroll
      for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int c = 0; c < 2; ++c) acc[a][c] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[c], pf[a], acc[a][c], 0, 0, 0);
    }
    // accumulator (row = channel, column = pixel): register r of lane l = channel (r & 3) + 8 (r >> 2) + 4 (l >> 5), pixel l & 31
    const int ox = ox0 + col;
#pragma unroll
    for (int a = 0; a < 2; ++a) {
      const int oy = oy0 + wave * 2 + a;
      if (oy >= p.OH || ox >= p.OW) continue;
      bf16* dst = p.z + (((size_t)b * p.OH + oy) * p.OW + ox) * 64;
#pragma unroll
      for (int c = 0; c < 2; ++c)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          union { bf16 v[4]; uint2 u; } o;
#pragma unroll
          for (int e = 0; e < 4; ++e) o.v[e] = __float2bfloat16(acc[a][c][g * 4 + e]);
          *reinterpret_cast<uint2*>(dst + c * 32 + 8 * g + 4 * h) = o.u;
        }
    }
  }
}

// ---- frozen statistics (the teacher): convolution + norm + activation + 3x3 / 2 max pool in ONE kernel (round 5) --------------------
// Without gradients nobody needs the 257 x 257 x 64 convolution output (203 MB at 24 images): here it exists as one 8 x 32 tile in
// LDS.  A workgroup tile is 3 x 15 POOLED pixels; it computes the 8 x 32 convolution outputs around them with the product code of
// stem_conv7x7_kernel (origin 2 ph0 - 1, 2 pw0 - 1: one row and one column of halo are recomputed by the neighbours, 1.42x the
// MFMA work of a kernel that is bound by its 76 MB image read), rounds them to bf16 as the two-kernel path stores them, applies
// norm + activation with the same expression as stem_apply_pool_kernel - bit-identical results - and pools from LDS.
constexpr int kFPH = 3, kFPW = 15;                    // pooled tile
constexpr int kFTP = 72;                              // bf16 pitch of a pixel of the activated tile: 64 channels + 8 (144 B: the 32 pixels of
                                                      // a row land on 16 different bank offsets instead of 2 - SQ_LDS_BANK_CONFLICT 0.62 at 64)
struct StemFusedArgs {
  StemConvArgs c;                                     // c.z unused; c.tiles_* count POOLED tiles
  const float *mean, *scale, *beta;
  float slope;
  bf16* out;                                          // [B, PH, PW, 64]
  int PH, PW;
};

__global__ __launch_bounds__(kThreads) void stem_conv_pool_kernel(StemFusedArgs q) {
  typedef float f32x16 __attribute__((ext_vector_type(16)));
  typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
  const StemConvArgs& p = q.c;
  extern __shared__ __attribute__((aligned(16))) unsigned char fs_raw[];
  unsigned short* Wl = reinterpret_cast<unsigned short*>(fs_raw);                   // [64][kSWP]
  unsigned short* In = Wl + 64 * kSWP;                                              // [kSIR][kSIP]
  unsigned short* Tl = In + kSIR * kSIP;                                            // [8][32][kFTP] activated tile (bf16 bits)
  float* Pl = reinterpret_cast<float*>(Tl + kSTH * kSTW * kFTP);                    // [3][64] mean | scale | shift
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  for (int i = tid; i < 64 * kSWP; i += kThreads) Wl[i] = 0;
  if (tid < 64) { Pl[tid] = q.mean[tid]; Pl[64 + tid] = q.scale[tid]; Pl[128 + tid] = q.beta ? q.beta[tid] : 0.f; }
  __syncthreads();
  for (int i = tid; i < 64 * 147; i += kThreads) {
    const int n = i / 147, r = i - n * 147, ky = r / 21, j = r - ky * 21;
    Wl[n * kSWP + ky * 24 + j] = reinterpret_cast<const unsigned short*>(p.w)[i];
  }
  const int col = lane & 31, h = lane >> 5;
  const int fe = tid < kSIP ? tid : kSIP - 1, fpx = fe / 3, fc = fe - fpx * 3;
  float pre[kSIR];
  unsigned premask = 0;
  auto origin = [&](int tile, int& b, int& oy0, int& ox0) {
    b = tile / (p.tiles_y * p.tiles_x);
    const int rem = tile - b * p.tiles_y * p.tiles_x;
    oy0 = 2 * (rem / p.tiles_x) * kFPH - 1;
    ox0 = 2 * (rem % p.tiles_x) * kFPW - 1;
  };
  auto fetch = [&](int tile) {
    unsigned okm = 0;
    int b, oy0, ox0;
    origin(tile, b, oy0, ox0);
    const int ix = 2 * ox0 - 3 + fpx;
    const bool colok = fpx < 2 * kSTW + 5 && (unsigned)ix < (unsigned)p.W;
    const float* src = p.x + b * p.sb + fc * p.sc + (colok ? ix : 0) * p.sw;
#pragma unroll
    for (int r = 0; r < kSIR; ++r) {
      const int iy = 2 * oy0 - 3 + r;
      const bool ok = colok && (unsigned)iy < (unsigned)p.H;
      pre[r] = src[(ok ? iy : 0) * p.sh];
      okm |= ok ? (1u << r) : 0u;
    }
    premask = okm;
  };
  if ((int)blockIdx.x < p.ntiles) fetch(blockIdx.x);
  for (int tile = blockIdx.x; tile < p.ntiles; tile += gridDim.x) {
    int b, oy0, ox0;
    origin(tile, b, oy0, ox0);
    __syncthreads();                                   // the previous tile's reads (patch and pooled tile) are done
    if (tid < kSIP) {
#pragma unroll
      for (int r = 0; r < kSIR; ++r) {
        const bf16 v = __float2bfloat16((premask >> r) & 1u ? pre[r] : 0.f);
        In[r * kSIP + tid] = *reinterpret_cast<const unsigned short*>(&v);
      }
    }
    __syncthreads();
    if (tile + (int)gridDim.x < p.ntiles) fetch(tile + gridDim.x);
    f32x16 acc[2][2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
      for (int c = 0; c < 2; ++c)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[a][c][r] = 0.f;
#pragma unroll
    for (int s = 0; s < kSK / 16; ++s) {
      const int t0 = 2 * s, t1 = 2 * s + 1;
      const int ky = h ? (t1 / 3 > 6 ? 6 : t1 / 3) : t0 / 3, j0 = h ? 8 * (t1 % 3) : 8 * (t0 % 3);
      bf16x8 wf[2], pf[2];
#pragma unroll
      for (int c = 0; c < 2; ++c) wf[c] = *reinterpret_cast<const bf16x8*>(&Wl[(c * 32 + col) * kSWP + 16 * s + 8 * h]);
#pragma unroll
      for (int a = 0; a < 2; ++a) {
        const int orow = wave * 2 + a;
        const unsigned* src = reinterpret_cast<const unsigned*>(&In[(2 * orow + ky) * kSIP + 6 * col + j0]);
        union { unsigned u[4]; bf16x8 v; } f;
        f.u[0] = src[0]; f.u[1] = src[1]; f.u[2] = src[2]; f.u[3] = src[3];
        pf[a] = f.v;
      }
#pragma unroll
      for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int c = 0; c < 2; ++c) acc[a][c] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[c], pf[a], acc[a][c], 0, 0, 0);
    }
    // norm + activation of the bf16-rounded outputs into the LDS tile; positions outside the map are the pooling's padding (-inf)
    const int ox = ox0 + col;
#pragma unroll
    for (int a = 0; a < 2; ++a) {
      const int orow = wave * 2 + a, oy = oy0 + orow;
      const bool inside = (unsigned)oy < (unsigned)p.OH && (unsigned)ox < (unsigned)p.OW;
      unsigned short* dst = Tl + (orow * kSTW + col) * kFTP;
#pragma unroll
      for (int c = 0; c < 2; ++c)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const int ch = c * 32 + 8 * g + 4 * h;
          const float4 mu = *reinterpret_cast<const float4*>(Pl + ch), sc = *reinterpret_cast<const float4*>(Pl + 64 + ch);
          const float4 sh = *reinterpret_cast<const float4*>(Pl + 128 + ch);
          const float m4[4] = {mu.x, mu.y, mu.z, mu.w}, s4[4] = {sc.x, sc.y, sc.z, sc.w}, b4[4] = {sh.x, sh.y, sh.z, sh.w};
          union { bf16 v[4]; uint2 u; } o;
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const float zq = bf16_round(acc[a][c][g * 4 + e]);                       // what stem_conv7x7_kernel stores
            o.v[e] = __float2bfloat16(leaky((zq - m4[e]) * s4[e] + b4[e], q.slope));  // what stem_apply_pool_kernel compares
          }
          if (!inside) o.u = make_uint2(0xFF80FF80u, 0xFF80FF80u);
          *reinterpret_cast<uint2*>(dst + ch) = o.u;
        }
    }
    __syncthreads();
    // 3 x 15 pooled pixels x 8 channel groups of 8
    for (int i = tid; i < kFPH * kFPW * 8; i += kThreads) {
      const int cg = i & 7, pp = i >> 3, pr = pp / kFPW, pc = pp - pr * kFPW;
      const int ph = (oy0 + 1) / 2 + pr, pw = (ox0 + 1) / 2 + pc;
      if (ph >= q.PH || pw >= q.PW) continue;
      float best[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) best[e] = -INFINITY;
#pragma unroll
      for (int kh = 0; kh < 3; ++kh)
#pragma unroll
        for (int kw = 0; kw < 3; ++kw) {
          Vec<bf16> v;
          v.raw = *reinterpret_cast<const uint4*>(Tl + ((2 * pr + kh) * kSTW + 2 * pc + kw) * kFTP + cg * 8);
#pragma unroll
          for (int e = 0; e < 8; ++e) best[e] = fmaxf(best[e], v.get(e));
        }
      Vec<bf16> o;
#pragma unroll
      for (int e = 0; e < 8; ++e) o.set(e, best[e]);
      o.store(q.out + (((size_t)b * q.PH + ph) * q.PW + pw) * 64 + cg * 8);
    }
  }
}

bool stem_args_ok(const void* z, int B, int H, int W, int C) {
  return z && B > 0 && B < 65536 && H > 1 && H < 65536 && W > 1 && C >= 8 && (C & (C - 1)) == 0 && C <= 2048 && aligned16(z);
}

}  // namespace
}  // namespace ucd

using namespace ucd;

extern "C" {

int ucd_stem_pooled_size(int n) { return (n + 2 - 3) / 2 + 1; }

size_t ucd_stem_pool_workspace_bytes(int C) { return (size_t)kRedBlocks * 2 * C * sizeof(float); }

int ucd_stem_apply_pool(const void* z, int B, int H, int W, int C, const float* mean, const float* scale, const float* beta, int act,
                        float slope, void* out, uint8_t* idx, ucd_stream_t stream) {
  static const char* fn = "ucd_stem_apply_pool";
  UCD_REQUIRE(stem_args_ok(z, B, H, W, C) && mean && scale && out && aligned16(out) && (!idx || (reinterpret_cast<uintptr_t>(idx) & 7) == 0),
              UCD_EINVAL, "%s: dense channels-last bf16 maps with a power-of-two channel count in 8..2048 expected", fn);
  const int a = act & UCD_ACT_MASK;
  UCD_REQUIRE(a != UCD_ACT_ELU, UCD_EUNSUPPORTED, "%s: leaky_relu / identity only (an elu stem keeps the separate passes)", fn);
  StemArgs p{};
  p.z = (const bf16*)z; p.B = B; p.H = H; p.W = W; p.C = C; p.PH = ucd_stem_pooled_size(H); p.PW = ucd_stem_pooled_size(W);
  p.mean = mean; p.scale = scale; p.beta = beta; p.slope = a == UCD_ACT_IDENTITY ? 1.f : slope;
  p.cg_shift = __builtin_ctz(C / 8);
  stem_apply_pool_kernel<<<dim3(ceil_div(p.PW * (C / 8), kThreads), p.PH, B), kThreads, 0, (hipStream_t)stream>>>(p, (bf16*)out, idx);
  return check_launch(fn);
}

int ucd_stem_conv7x7(const float* x, long long sb, long long sc, long long sh, long long sw, int B, int H, int W, const void* w,
                     void* z, ucd_stream_t stream) {
  static const char* fn = "ucd_stem_conv7x7";
  UCD_REQUIRE(x && w && z && B > 0 && H > 0 && W > 0 && aligned16(z) && (reinterpret_cast<uintptr_t>(w) & 1) == 0, UCD_EINVAL,
              "%s: bad arguments", fn);
  StemConvArgs p{};
  p.x = x; p.sb = sb; p.sc = sc; p.sh = sh; p.sw = sw; p.w = (const bf16*)w; p.z = (bf16*)z;
  p.B = B; p.H = H; p.W = W; p.OH = (H + 6 - 7) / 2 + 1; p.OW = (W + 6 - 7) / 2 + 1;
  p.tiles_y = ceil_div(p.OH, kSTH); p.tiles_x = ceil_div(p.OW, kSTW); p.ntiles = B * p.tiles_y * p.tiles_x;
  const int grid = p.ntiles < 768 ? p.ntiles : 768;          // three resident workgroups per CU, each walking its tiles
  stem_conv7x7_kernel<<<grid, kThreads, 0, (hipStream_t)stream>>>(p);
  return check_launch(fn);
}

int ucd_stem_conv_pool(const float* x, long long sb, long long sc, long long sh, long long sw, int B, int H, int W, const void* w,
                       const float* mean, const float* scale, const float* beta, int act, float slope, void* out, ucd_stream_t stream) {
  static const char* fn = "ucd_stem_conv_pool";
  UCD_REQUIRE(x && w && out && mean && scale && B > 0 && H > 0 && W > 0 && aligned16(out) && (reinterpret_cast<uintptr_t>(w) & 1) == 0 &&
              aligned16(mean) && aligned16(scale) && (!beta || aligned16(beta)), UCD_EINVAL, "%s: bad arguments", fn);
  const int a = act & UCD_ACT_MASK;
  UCD_REQUIRE(a != UCD_ACT_ELU && !(act & UCD_NORM_ABS_GAMMA), UCD_EUNSUPPORTED, "%s: leaky_relu / identity with given scale only", fn);
  StemFusedArgs q{};
  StemConvArgs& p = q.c;
  p.x = x; p.sb = sb; p.sc = sc; p.sh = sh; p.sw = sw; p.w = (const bf16*)w; p.z = nullptr;
  p.B = B; p.H = H; p.W = W; p.OH = (H + 6 - 7) / 2 + 1; p.OW = (W + 6 - 7) / 2 + 1;
  q.PH = ucd_stem_pooled_size(p.OH); q.PW = ucd_stem_pooled_size(p.OW);
  p.tiles_y = ceil_div(q.PH, kFPH); p.tiles_x = ceil_div(q.PW, kFPW); p.ntiles = B * p.tiles_y * p.tiles_x;
  q.mean = mean; q.scale = scale; q.beta = beta; q.slope = a == UCD_ACT_IDENTITY ? 1.f : slope; q.out = (bf16*)out;
  const size_t lds = (size_t)(64 * kSWP + kSIR * kSIP + kSTH * kSTW * kFTP) * 2 + 3 * 64 * 4;
  UCD_TRY_LDS(stem_conv_pool_kernel, 80 * 1024);
  const int grid = p.ntiles < 512 ? p.ntiles : 512;          // two resident workgroups per CU (66 KB of LDS each), each walking its tiles
  stem_conv_pool_kernel<<<grid, kThreads, lds, (hipStream_t)stream>>>(q);
  return check_launch(fn);
}

/* phase 1 (reduce -> sums), 2 (apply with the given sums), 3 (both).  sums [2 C]: sum dyact | sum dyact xhat (x sign(weight) for
 * the |gamma| + eps layers: the layer's d bias | d weight); count = positions per channel over ALL ranks (B H W x world). */
int ucd_stem_pool_backward(const void* z, const void* dpool, const uint8_t* idx, int B, int H, int W, int C, const float* mean,
                           const float* invstd, const float* scale, const float* beta, const float* weight, float* sums, float count,
                           int act, float slope, void* dz, void* workspace, size_t workspace_bytes, int phase, ucd_stream_t stream) {
  static const char* fn = "ucd_stem_pool_backward";
  UCD_REQUIRE(stem_args_ok(z, B, H, W, C) && dpool && idx && mean && invstd && scale && sums && aligned16(dpool), UCD_EINVAL,
              "%s: bad arguments", fn);
  UCD_REQUIRE(phase >= 1 && phase <= 3 && (!(phase & 2) || (dz && aligned16(dz) && count > 0.f)), UCD_EINVAL, "%s: bad phase / dz", fn);
  const int a = act & UCD_ACT_MASK;
  UCD_REQUIRE(a != UCD_ACT_ELU, UCD_EUNSUPPORTED, "%s: leaky_relu / identity only", fn);
  StemArgs p{};
  p.z = (const bf16*)z; p.B = B; p.H = H; p.W = W; p.C = C; p.PH = ucd_stem_pooled_size(H); p.PW = ucd_stem_pooled_size(W);
  p.mean = mean; p.scale = scale; p.beta = beta; p.invstd = invstd; p.weight = weight; p.sums = sums;
  p.slope = a == UCD_ACT_IDENTITY ? 1.f : slope; p.abs_gamma = (act & UCD_NORM_ABS_GAMMA) != 0;
  p.cg_shift = __builtin_ctz(C / 8);
  hipStream_t s = (hipStream_t)stream;
  if (phase & 1) {
    UCD_REQUIRE(workspace && workspace_bytes >= ucd_stem_pool_workspace_bytes(C), UCD_EWORKSPACE, "%s: workspace too small", fn);
    stem_pool_bwd_reduce_kernel<<<kRedBlocks, kThreads, kThreads * 16 * sizeof(float), s>>>(p, (const bf16*)dpool, idx, (float*)workspace);
    int rc = check_launch(fn);
    if (rc) return rc;
    stem_sum_partials_kernel<<<2 * C, kThreads, 0, s>>>((const float*)workspace, kRedBlocks, C, p.abs_gamma ? weight : nullptr, sums);
    rc = check_launch(fn);
    if (rc) return rc;
  }
  if (phase & 2) {
    p.inv_count = 1.f / count;
    stem_pool_bwd_apply_kernel<<<dim3(ceil_div(((W + 1) / 2) * (C / 8), kThreads), ceil_div(H, kApplyRows), B), kThreads, 0, s>>>(p, (const bf16*)dpool, idx, (bf16*)dz);
    return check_launch(fn);
  }
  return 0;
}

}  // extern "C"
