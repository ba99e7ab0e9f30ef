// Contrastive prep: label down-sampling, teacher pseudo-label mixing, stream compaction (optionally
// grouped by label), row gather + L2 normalisation.  Replaces the ~30 small PyTorch kernels, the six
// boolean-mask gathers and the device->host sync of pre_contractive_pixel (utils/utils.py:264-268,
// 349-375).  Nothing here returns a size to the host: counts live in ucd_pixcon_meta.
#include "common.h"
#include "pixcon.h"

namespace ucd {
namespace {

constexpr int kBlock = 256;

// Source index / weights of torch's bilinear resize (align_corners = False), float32 step by step:
// scale = float(in)/out;  src = scale*(dst+0.5) - 0.5, clamped at 0;  i0 = floor(src);  l1 = src - i0.
__device__ __forceinline__ void bilinear_src(int dst, int in_size, float scale, int& i0, int& i1, float& l0, float& l1) {
  float src = __fsub_rn(__fmul_rn(scale, __fadd_rn((float)dst, 0.5f)), 0.5f);
  src = src < 0.f ? 0.f : src;
  i0 = min((int)floorf(src), in_size - 1);
  i1 = i0 + (i0 < in_size - 1 ? 1 : 0);
  l1 = fminf(fmaxf(__fsub_rn(src, (float)i0), 0.f), 1.f);
  l0 = __fsub_rn(1.f, l1);
}

// Stage 1: one thread per low-resolution pixel.
template <typename TL>
__global__ __launch_bounds__(kBlock) void prep_classify_kernel(
    const int64_t* __restrict__ labels, int B, int H, int W, int h, int w, int max_label, float scale_h, float scale_w,
    const TL* __restrict__ tlogits, int ld_t, int K, int sort_by_label, uint8_t* __restrict__ mix_out,
    uint8_t* __restrict__ kind_out, float* __restrict__ prob, int32_t* __restrict__ cnt_a, int32_t* __restrict__ cnt_o,
    int nblk, int32_t* __restrict__ scalars /* [0]=min_new, [1]=n_new, [2..258)=anchors per label, [258..514)=teacher rows per label */) {
  __shared__ int hist_a[256], hist_o[256];
  __shared__ int s_min, s_new;
  const int tid = threadIdx.x;
  hist_a[tid] = 0;
  hist_o[tid] = 0;
  if (tid == 0) { s_min = 0x7fffffff; s_new = 0; }
  __syncthreads();
  const int BHW = B * h * w;
  const int p = blockIdx.x * kBlock + tid;
  if (p < BHW) {
    const int b = p / (h * w), rem = p - b * h * w, oy = rem / w, ox = rem - oy * w;
    int y0, y1, x0, x1;
    float ly0, ly1, lx0, lx1;
    bilinear_src(oy, H, scale_h, y0, y1, ly0, ly1);
    bilinear_src(ox, W, scale_w, x0, x1, lx0, lx1);
    const int64_t* lp = labels + (size_t)b * H * W;
    const float v00 = (float)lp[(size_t)y0 * W + x0], v01 = (float)lp[(size_t)y0 * W + x1];
    const float v10 = (float)lp[(size_t)y1 * W + x0], v11 = (float)lp[(size_t)y1 * W + x1];
    // product weights, then one multiply and three fused multiply-adds in this order: the arithmetic of
    // torch's one-channel bilinear kernel (pinned bit-exactly by tests/test_oracle_golden.py)
    const float w00 = __fmul_rn(ly0, lx0), w01 = __fmul_rn(ly0, lx1), w10 = __fmul_rn(ly1, lx0), w11 = __fmul_rn(ly1, lx1);
    float acc = __fmul_rn(v01, w01);
    acc = __fmaf_rn(v00, w00, acc);
    acc = __fmaf_rn(v10, w10, acc);
    acc = __fmaf_rn(v11, w11, acc);
    int lab = (int)acc;  // truncation toward zero (the reference's .type(torch.int8)), acc >= 0
    if (lab < 0 || lab > max_label) lab = 0;

    // teacher arg-max (first maximum) and softmax over the K old classes
    const TL* tp = tlogits + (size_t)p * ld_t;
    float mx = -INFINITY;
    int arg = 0;
    for (int k = 0; k < K; ++k) {
      float v = (float)tp[k];
      if (v > mx) { mx = v; arg = k; }
    }
    float den = 0.f;
    for (int k = 0; k < K; ++k) den += __expf((float)tp[k] - mx);
    const float inv = 1.f / den;
    float* pp = prob + (size_t)p * K;
    for (int k = 0; k < K; ++k) pp[k] = __expf((float)tp[k] - mx) * inv;

    const int mix = lab != 0 ? lab : arg;
    const int kind = mix > 0 ? (lab > 0 ? 1 : 2) : 0;  // 1: anchor with ground truth, 2: anchor + teacher row
    mix_out[p] = (uint8_t)mix;
    kind_out[p] = (uint8_t)kind;
    if (kind) atomicAdd(&hist_a[mix], 1);
    if (kind == 2) atomicAdd(&hist_o[mix], 1);
    if (lab > 0) {
      atomicMin(&s_min, lab);
      atomicAdd(&s_new, 1);
    }
  }
  __syncthreads();
  // label-major count tables [key][block]; without sorting every row falls under key 0
  if (sort_by_label) {
    cnt_a[(size_t)tid * nblk + blockIdx.x] = hist_a[tid];
    cnt_o[(size_t)tid * nblk + blockIdx.x] = hist_o[tid];
  } else {
    int ta = 0, to = 0;
    if (tid == 0)
      for (int i = 0; i < 256; ++i) { ta += hist_a[i]; to += hist_o[i]; }
    cnt_a[(size_t)tid * nblk + blockIdx.x] = ta;
    cnt_o[(size_t)tid * nblk + blockIdx.x] = to;
  }
  if (hist_a[tid]) atomicAdd(&scalars[2 + tid], hist_a[tid]);
  if (hist_o[tid]) atomicAdd(&scalars[258 + tid], hist_o[tid]);
  if (tid == 0) {
    if (s_new) {
      atomicMin(&scalars[0], s_min);
      atomicAdd(&scalars[1], s_new);
    }
  }
}

// Stage 2: one block turns the count tables into exclusive offsets (label-major, so a stable sort by
// label falls out of the order-preserving compaction) and fills the meta record.
__global__ __launch_bounds__(1024) void prep_scan_kernel(int32_t* __restrict__ cnt_a, int32_t* __restrict__ cnt_o,
                                                         int nblk, const int32_t* __restrict__ scalars,
                                                         ucd_pixcon_meta* __restrict__ meta, int sort_by_label) {
  __shared__ int warp_tot[16];
  __shared__ int carry;
  const int n = 256 * nblk;
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  for (int which = 0; which < 2; ++which) {
    int32_t* cnt = which == 0 ? cnt_a : cnt_o;
    int32_t* lstart = which == 0 ? meta->label_start_a : meta->label_start_o;
    if (tid == 0) carry = 0;
    __syncthreads();
    for (int base = 0; base < n; base += 1024) {
      const int idx = base + tid;
      const int v = idx < n ? cnt[idx] : 0;
      int incl = v;
#pragma unroll
      for (int off = 1; off < 64; off <<= 1) {
        int t = __shfl_up(incl, off, 64);
        if (lane >= off) incl += t;
      }
      if (lane == 63) warp_tot[wv] = incl;
      __syncthreads();
      int woff = 0;
      for (int i = 0; i < wv; ++i) woff += warp_tot[i];
      const int excl = carry + woff + incl - v;
      if (idx < n) {
        cnt[idx] = excl;
        if (idx % nblk == 0) lstart[idx / nblk] = excl;  // first block of a label: start of its segment
      }
      __syncthreads();
      if (tid == 1023) carry = excl + v;
      __syncthreads();
    }
    if (tid == 0) {
      lstart[256] = carry;
      if (which == 0) meta->A = carry; else meta->Co = carry;
    }
    __syncthreads();
  }
  if (tid == 0) {
    const int A = meta->A, Co = meta->Co;
    meta->min_new = scalars[1] ? scalars[0] : 0x7fffffff;
    meta->n_new = scalars[1];
    meta->Apad = (A + kPixTile - 1) / kPixTile * kPixTile;
    meta->Cpad = (meta->Apad + Co + kPixTile - 1) / kPixTile * kPixTile;
    meta->sorted = sort_by_label;
  }
  __syncthreads();
  // per-label counts and the number of rows with at least one positive (anchors whose label occurs
  // at least twice in the contrast set; loss.py:464-466)
  if (tid < 256) {
    meta->label_count_a[tid] = scalars[2 + tid];
    meta->label_count_c[tid] = scalars[2 + tid] + scalars[258 + tid];
  }
  if (tid == 0) {
    int nv = 0;
    for (int L = 0; L < 256; ++L)
      if (scalars[2 + L] + scalars[258 + L] - 1 > 0) nv += scalars[2 + L];
    meta->n_valid = nv;
  }
}

// Stage 2, tables that fit in LDS (round 5): the kernel above walks a table in 1024-entry slices with three barriers per slice -
// 26 slices x 2 tables = 67 us for 2 x 105 KB at the benchmark shape, on the dependent chain in front of the contrastive loss.
// Here a table is copied into LDS with coalesced loads, every thread sums its own contiguous run, ONE block scan orders the runs,
// the runs are rewritten in place and copied back: three barriers per table.
__global__ __launch_bounds__(1024) void prep_scan_lds_kernel(int32_t* __restrict__ cnt_a, int32_t* __restrict__ cnt_o, int nblk,
                                                             const int32_t* __restrict__ scalars,
                                                             ucd_pixcon_meta* __restrict__ meta, int sort_by_label) {
  extern __shared__ int32_t tab[];                     // [n + n / 32]: one pad word per 32 entries (the runs start 26 words apart)
  __shared__ int warp_tot[16];
  const int n = 256 * nblk;
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int per = (n + 1023) / 1024;
  const int i0 = min(n, tid * per), i1 = min(n, i0 + per);
  auto at = [](int i) { return i + (i >> 5); };
  for (int which = 0; which < 2; ++which) {
    int32_t* cnt = which == 0 ? cnt_a : cnt_o;
    int32_t* lstart = which == 0 ? meta->label_start_a : meta->label_start_o;
    for (int i = tid; i < n; i += 1024) tab[at(i)] = cnt[i];
    __syncthreads();
    int sum = 0;
    for (int i = i0; i < i1; ++i) sum += tab[at(i)];
    int incl = sum;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
      const int t = __shfl_up(incl, off, 64);
      if (lane >= off) incl += t;
    }
    if (lane == 63) warp_tot[wv] = incl;
    __syncthreads();
    int run = incl - sum, total = 0;
    for (int i = 0; i < 16; ++i) {
      if (i < wv) run += warp_tot[i];
      total += warp_tot[i];
    }
    for (int i = i0; i < i1; ++i) {
      const int v = tab[at(i)];
      tab[at(i)] = run;
      run += v;
    }
    __syncthreads();
    for (int i = tid; i < n; i += 1024) cnt[i] = tab[at(i)];
    if (tid < 256) lstart[tid] = tab[at(tid * nblk)];  // first block of a label: start of its segment
    if (tid == 0) {
      lstart[256] = total;
      if (which == 0) meta->A = total; else meta->Co = total;
    }
    __syncthreads();
  }
  if (tid == 0) {
    const int A = meta->A, Co = meta->Co;
    meta->min_new = scalars[1] ? scalars[0] : 0x7fffffff;
    meta->n_new = scalars[1];
    meta->Apad = (A + kPixTile - 1) / kPixTile * kPixTile;
    meta->Cpad = (meta->Apad + Co + kPixTile - 1) / kPixTile * kPixTile;
    meta->sorted = sort_by_label;
  }
  // per-label counts and the number of rows with at least one positive (anchors whose label occurs at least twice in the contrast
  // set; loss.py:464-466)
  int nv = 0;
  if (tid < 256) {
    const int ca = scalars[2 + tid], cc = ca + scalars[258 + tid];
    meta->label_count_a[tid] = ca;
    meta->label_count_c[tid] = cc;
    nv = cc - 1 > 0 ? ca : 0;
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) nv += __shfl_xor(nv, off, 64);
  if (lane == 0) warp_tot[wv] = nv;
  __syncthreads();
  if (tid == 0) meta->n_valid = warp_tot[0] + warp_tot[1] + warp_tot[2] + warp_tot[3];
}

// Stage 3: scatter pixel indices / labels to their compacted rows.
__global__ __launch_bounds__(kBlock) void prep_scatter_kernel(const uint8_t* __restrict__ mix, const uint8_t* __restrict__ kind,
                                                             int BHW, int sort_by_label, const int32_t* __restrict__ off_a,
                                                             const int32_t* __restrict__ off_o, int nblk,
                                                             const ucd_pixcon_meta* __restrict__ meta,
                                                             int32_t* __restrict__ anchor_pix, int32_t* __restrict__ old_pix,
                                                             uint8_t* __restrict__ row_label) {
  // rank of a row among the EARLIER rows of its block with the same key (the order-preserving compaction): per wave by ballots over
  // the wave's distinct keys (a handful), across waves by a [wave][key] count table.  (Was: every thread walks all earlier threads
  // of the block, 255 LDS reads for the last one: 38 us on the chain in front of the contrastive loss.)
  __shared__ unsigned short w_a[4][256], w_o[4][256];
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int p = blockIdx.x * kBlock + tid;
  const int m = p < BHW ? mix[p] : 0, kd = p < BHW ? kind[p] : 0;
  const int key = sort_by_label ? m : 0;
  for (int i = tid; i < 4 * 256; i += kBlock) { (&w_a[0][0])[i] = 0; (&w_o[0][0])[i] = 0; }
  __syncthreads();
  int ra = 0, ro = 0;
  {
    const unsigned long long below = lane ? (~0ull >> (64 - lane)) : 0ull;
    unsigned long long todo = __ballot(kd != 0);
    while (todo) {
      const int leader = __ffsll((long long)todo) - 1;
      const int k = __shfl(key, leader, 64);
      const unsigned long long same_a = __ballot(kd != 0 && key == k), same_o = __ballot(kd == 2 && key == k);
      if (kd != 0 && key == k) { ra = __popcll(same_a & below); ro = __popcll(same_o & below); }
      if (lane == leader) { w_a[wv][k] = (unsigned short)__popcll(same_a); w_o[wv][k] = (unsigned short)__popcll(same_o); }
      todo &= ~same_a;
    }
  }
  __syncthreads();
  if (!kd) return;
  for (int i = 0; i < wv; ++i) { ra += w_a[i][key]; ro += w_o[i][key]; }
  const int ia = off_a[(size_t)key * nblk + blockIdx.x] + ra;
  anchor_pix[ia] = p;
  row_label[ia] = (uint8_t)m;
  if (kd == 2) {
    const int io = off_o[(size_t)key * nblk + blockIdx.x] + ro;
    old_pix[io] = p;
    row_label[meta->Apad + io] = (uint8_t)m;
  }
}

// Stage 4: gather + normalise; one wave per contrast row.
template <typename T>
__global__ __launch_bounds__(kBlock) void gather_normalize_kernel(const T* __restrict__ f_n, int ld_n, const T* __restrict__ f_o,
                                                                 int ld_o, int N, const int32_t* __restrict__ anchor_pix,
                                                                 const int32_t* __restrict__ old_pix,
                                                                 const float* __restrict__ prob, int K,
                                                                 const ucd_pixcon_meta* __restrict__ meta,
                                                                 float* __restrict__ chat, int ldc, float* __restrict__ pcat,
                                                                 int ldp, _Float16* __restrict__ ch16,
                                                                 _Float16* __restrict__ p16, int KP16,
                                                                 float* __restrict__ inv_norm) {
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int r = blockIdx.x * (kBlock / 64) + wave;
  const int A = meta->A, Apad = meta->Apad, Co = meta->Co, Cpad = meta->Cpad;
  if (r >= Cpad) return;
  const T* src = nullptr;
  int pix = -1;
  if (r < A) { pix = anchor_pix[r]; src = f_n + (size_t)pix * ld_n; }
  else if (r >= Apad && r < Apad + Co) { pix = old_pix[r - Apad]; src = f_o + (size_t)pix * ld_o; }
  float* dst = chat + (size_t)r * ldc;
  float* pdst = pcat ? pcat + (size_t)r * ldp : nullptr;
  _Float16* hdst = ch16 ? ch16 + (size_t)r * ldc : nullptr;
  _Float16* qdst = p16 ? p16 + (size_t)r * 2 * KP16 : nullptr;
  if (!src) {
    for (int c = lane; c < ldc; c += 64) dst[c] = 0.f;
    if (pdst) for (int k = lane; k < ldp; k += 64) pdst[k] = 0.f;
    if (hdst) for (int c = lane; c < ldc; c += 64) hdst[c] = (_Float16)0.f;
    if (qdst) for (int k = lane; k < 2 * KP16; k += 64) qdst[k] = (_Float16)0.f;
    return;
  }
  float ss = 0.f;
  for (int c = lane; c < N; c += 64) {
    float v = (float)src[c];
    ss += v * v;
  }
  ss = wave_sum(ss);
  const float inv = 1.f / fmaxf(sqrtf(ss), 1e-12f);  // F.normalize: x / max(||x||, eps)
  for (int c = lane; c < ldc; c += 64) {
    const float v = c < N ? (float)src[c] * inv : 0.f;
    dst[c] = v;
    if (hdst) hdst[c] = (_Float16)v;
  }
  const float* pp = prob ? prob + (size_t)pix * K : nullptr;
  if (pdst)
    for (int k = lane; k < ldp; k += 64) pdst[k] = k < K ? pp[k] : 0.f;
  if (qdst)  // hi | lo split: p = hi + lo to ~2^-22
    for (int k = lane; k < KP16; k += 64) {
      const float v = k < K ? pp[k] : 0.f;
      const _Float16 hi = (_Float16)v;
      qdst[k] = hi;
      qdst[KP16 + k] = (_Float16)(v - (float)hi);
    }
  if (r < A && lane == 0) inv_norm[r] = inv;
}

// d f_n[pix(r), :] = gs * inv_norm[r] * (g_r - (g_r . a_r) a_r); zero rows for pixels that are not anchors.
template <typename T>
__global__ __launch_bounds__(kBlock) void scatter_grad_kernel(const float* __restrict__ grad_a, const float* __restrict__ chat,
                                                             int ldc, const float* __restrict__ inv_norm,
                                                             const int32_t* __restrict__ anchor_pix,
                                                             const ucd_pixcon_meta* __restrict__ meta,
                                                             const float* __restrict__ grad_scale, T* __restrict__ d_f_n,
                                                             int ld_d, int N) {
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int r = blockIdx.x * (kBlock / 64) + wave;
  if (r >= meta->A) return;
  const float* g = grad_a + (size_t)r * ldc;
  const float* a = chat + (size_t)r * ldc;
  float dot = 0.f;
  for (int c = lane; c < N; c += 64) dot += g[c] * a[c];
  dot = wave_sum(dot);
  const float k = grad_scale[0] * inv_norm[r];
  T* dst = d_f_n + (size_t)anchor_pix[r] * ld_d;
  for (int c = lane; c < N; c += 64) dst[c] = (T)(k * (g[c] - dot * a[c]));
}

}  // namespace
}  // namespace ucd

using namespace ucd;

extern "C" {

size_t ucd_pixcon_prep_workspace_bytes(int BHW, int K) {
  (void)K;
  const int nblk = ceil_div(BHW, kBlock);
  // mix[BHW] + kind[BHW] (bytes, padded) + two [256][nblk] count tables + 2 scalars + 2x256 label counts
  return align_up((size_t)BHW, 16) * 2 + (size_t)2 * 256 * nblk * 4 + 516 * 4;
}

int ucd_pixcon_prep(const int64_t* labels, int B, int H, int W, int h, int w, int max_label,
                    const void* teacher_logits, int ld_t, int dtype_t, int K, int sort_by_label, int32_t* anchor_pix,
                    int32_t* old_pix, uint8_t* row_label, float* prob, ucd_pixcon_meta* meta, void* workspace,
                    size_t workspace_bytes, ucd_stream_t stream) {
  static const char* fn = "ucd_pixcon_prep";
  UCD_REQUIRE(labels && teacher_logits && anchor_pix && old_pix && row_label && prob && meta && workspace, UCD_EINVAL,
              "%s: NULL argument", fn);
  UCD_REQUIRE(B > 0 && H > 0 && W > 0 && h > 0 && w > 0 && K > 0 && ld_t >= K, UCD_EINVAL, "%s: bad sizes", fn);
  UCD_REQUIRE(max_label >= 1 && max_label <= 254, UCD_EUNSUPPORTED, "%s: max_label must be in [1, 254]", fn);
  UCD_REQUIRE(K <= 255, UCD_EUNSUPPORTED, "%s: K must be <= 255", fn);
  UCD_REQUIRE(dtype_t == UCD_F32 || dtype_t == UCD_BF16, UCD_EINVAL, "%s: unknown dtype", fn);
  const int BHW = B * h * w;
  UCD_REQUIRE(workspace_bytes >= ucd_pixcon_prep_workspace_bytes(BHW, K), UCD_EWORKSPACE, "%s: workspace too small", fn);
  hipStream_t s = (hipStream_t)stream;
  const int nblk = ceil_div(BHW, kBlock);
  uint8_t* mix = (uint8_t*)workspace;
  uint8_t* kind = mix + align_up((size_t)BHW, 16);
  int32_t* cnt_a = (int32_t*)(kind + align_up((size_t)BHW, 16));
  int32_t* cnt_o = cnt_a + (size_t)256 * nblk;
  int32_t* scalars = cnt_o + (size_t)256 * nblk;
  // two ints of state (running minimum starts at 0x7f7f7f7f, count at 0) + the padding labels;
  // all re-initialised on every call by memset nodes on the stream
  hipError_t e = hipMemsetAsync(scalars, 0x7F, 4, s);
  if (e == hipSuccess) e = hipMemsetAsync(scalars + 1, 0, 515 * 4, s);
  if (e == hipSuccess) e = hipMemsetAsync(row_label, 0xFF, (size_t)2 * BHW + 2 * kPixTile, s);
  if (e != hipSuccess) { set_error("%s: %s", fn, hipGetErrorString(e)); return (int)e; }
  // float(in)/out exactly as torch computes the resize scale
  const float scale_h = (float)H / (float)h, scale_w = (float)W / (float)w;
  if (dtype_t == UCD_BF16)
    prep_classify_kernel<__hip_bfloat16><<<nblk, kBlock, 0, s>>>(labels, B, H, W, h, w, max_label, scale_h, scale_w,
                                                                (const __hip_bfloat16*)teacher_logits, ld_t, K,
                                                                sort_by_label, mix, kind, prob, cnt_a, cnt_o, nblk, scalars);
  else
    prep_classify_kernel<float><<<nblk, kBlock, 0, s>>>(labels, B, H, W, h, w, max_label, scale_h, scale_w,
                                                       (const float*)teacher_logits, ld_t, K, sort_by_label, mix, kind,
                                                       prob, cnt_a, cnt_o, nblk, scalars);
  int rc = check_launch(fn);
  if (rc) return rc;
  {
    const size_t n = (size_t)256 * nblk, scan_lds = (n + n / 32 + 1) * sizeof(int32_t);
    if (scan_lds <= 150 * 1024) {
      UCD_TRY_LDS(prep_scan_lds_kernel, 150 * 1024);
      prep_scan_lds_kernel<<<1, 1024, scan_lds, s>>>(cnt_a, cnt_o, nblk, scalars, meta, sort_by_label);
    } else {
      prep_scan_kernel<<<1, 1024, 0, s>>>(cnt_a, cnt_o, nblk, scalars, meta, sort_by_label);
    }
  }
  rc = check_launch(fn);
  if (rc) return rc;
  prep_scatter_kernel<<<nblk, kBlock, 0, s>>>(mix, kind, BHW, sort_by_label, cnt_a, cnt_o, nblk, meta, anchor_pix,
                                              old_pix, row_label);
  return check_launch(fn);
}

int ucd_pixcon_gather(const void* f_n, int ld_n, const void* f_o, int ld_o, int dtype, int BHW, int N,
                      const int32_t* anchor_pix, const int32_t* old_pix, const float* prob, int K,
                      const ucd_pixcon_meta* meta, float* chat, int ldc, float* pcat, int ldp, void* ch16, void* p16,
                      float* inv_norm, ucd_stream_t stream) {
  static const char* fn = "ucd_pixcon_gather";
  UCD_REQUIRE(f_n && f_o && anchor_pix && old_pix && meta && chat && inv_norm, UCD_EINVAL, "%s: NULL argument", fn);
  UCD_REQUIRE(BHW > 0 && N > 0 && ldc >= N && ld_n >= N && ld_o >= N, UCD_EINVAL, "%s: bad sizes", fn);
  UCD_REQUIRE(!pcat || (prob && K > 0 && ldp >= K), UCD_EINVAL, "%s: pcat needs prob, K and ldp >= K", fn);
  UCD_REQUIRE(!p16 || (prob && K > 0), UCD_EINVAL, "%s: p16 needs prob and K", fn);
  const int KP16 = (K + 15) / 16 * 16;
  UCD_REQUIRE(dtype == UCD_F32 || dtype == UCD_BF16, UCD_EINVAL, "%s: unknown dtype", fn);
  hipStream_t s = (hipStream_t)stream;
  const int max_rows = 2 * BHW + 2 * kPixTile;  // worst case; rows past meta->Cpad exit at once
  const int blocks = ceil_div(max_rows, kBlock / 64);
  if (dtype == UCD_BF16)
    gather_normalize_kernel<__hip_bfloat16><<<blocks, kBlock, 0, s>>>((const __hip_bfloat16*)f_n, ld_n,
                                                                     (const __hip_bfloat16*)f_o, ld_o, N, anchor_pix,
                                                                     old_pix, prob, K, meta, chat, ldc, pcat, ldp,
                                                                     (_Float16*)ch16, (_Float16*)p16, KP16, inv_norm);
  else
    gather_normalize_kernel<float><<<blocks, kBlock, 0, s>>>((const float*)f_n, ld_n, (const float*)f_o, ld_o, N,
                                                            anchor_pix, old_pix, prob, K, meta, chat, ldc, pcat, ldp,
                                                            (_Float16*)ch16, (_Float16*)p16, KP16, inv_norm);
  return check_launch(fn);
}

int ucd_pixcon_scatter_grad(const float* grad_a, const float* chat, int ldc, const float* inv_norm,
                            const int32_t* anchor_pix, const ucd_pixcon_meta* meta, const float* grad_scale,
                            void* d_f_n, int ld_d, int dtype, int BHW, int N, ucd_stream_t stream) {
  static const char* fn = "ucd_pixcon_scatter_grad";
  UCD_REQUIRE(grad_a && chat && inv_norm && anchor_pix && meta && grad_scale && d_f_n, UCD_EINVAL, "%s: NULL argument", fn);
  UCD_REQUIRE(BHW > 0 && N > 0 && ldc >= N && ld_d >= N, UCD_EINVAL, "%s: bad sizes", fn);
  UCD_REQUIRE(dtype == UCD_F32 || dtype == UCD_BF16, UCD_EINVAL, "%s: unknown dtype", fn);
  hipStream_t s = (hipStream_t)stream;
  const size_t es = dtype == UCD_BF16 ? 2 : 4;
  hipError_t e = hipMemsetAsync(d_f_n, 0, (size_t)BHW * ld_d * es, s);
  if (e != hipSuccess) { set_error("%s: %s", fn, hipGetErrorString(e)); return (int)e; }
  const int blocks = ceil_div(BHW, kBlock / 64);
  if (dtype == UCD_BF16)
    scatter_grad_kernel<__hip_bfloat16><<<blocks, kBlock, 0, s>>>(grad_a, chat, ldc, inv_norm, anchor_pix, meta,
                                                                 grad_scale, (__hip_bfloat16*)d_f_n, ld_d, N);
  else
    scatter_grad_kernel<float><<<blocks, kBlock, 0, s>>>(grad_a, chat, ldc, inv_norm, anchor_pix, meta, grad_scale,
                                                        (float*)d_f_n, ld_d, N);
  return check_launch(fn);
}

}  // extern "C"
