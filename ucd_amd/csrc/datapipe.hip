// Label path of the training data pipeline on the device (first piece).
//
// The reference transforms every label map on the host, one image at a time: RandomResizedCrop((0.5, 2.0)) =
// crop(i, j, h, w) + PIL NEAREST resize to S x S (dataset/transform.py:481-553 via torchvision's resized_crop),
// RandomHorizontalFlip (dataset/transform.py:300-318), ToTensor, and then the incremental-step remapping as a PYTHON
// LAMBDA PER PIXEL (dataset/voc.py:176-203: t.apply_(lambda x: inverted_order[x] if x in tmp_labels else masking_value),
// 263 k interpreter calls per 513^2 image).  Here the whole batch is one gather kernel: out[b, y, x] =
// lut[ src_b[i + yin[y]][j + xin[x']] ], x' = S-1-x when flipped - integer work, bit-exact.
//
// PIL's NEAREST resize picks source indices by ACCUMULATING in double: xo = a*0.5; xin[x] = (int)xo; xo += a (a = w/S),
// which differs from floor((x+0.5)*a) in ~25 % of the size pairs (pinned against Pillow by tests/golden); the tables are
// therefore built by one sequential thread per image and axis.
#include "common.h"

namespace ucd {
namespace {

__global__ void label_tables_kernel(const int* __restrict__ desc, int B, int S, int* __restrict__ tables) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= 2 * B) return;
  const int b = t >> 1, axis = t & 1;                 // axis 0: rows (h), axis 1: columns (w)
  const int* d = desc + 8 * b;
  const int extent = axis == 0 ? d[4] : d[5];
  const double a = (double)extent / (double)S;
  double xo = a * 0.5;
  int* tab = tables + ((size_t)b * 2 + axis) * S;
  for (int x = 0; x < S; ++x) {
    int v = (int)xo;
    tab[x] = v < extent ? v : extent - 1;
    xo += a;
  }
}

__global__ __launch_bounds__(256) void label_gather_kernel(const uint8_t* const* __restrict__ src,
                                                           const int* __restrict__ desc, int S,
                                                           const uint8_t* __restrict__ lut, const int* __restrict__ tables,
                                                           int64_t* __restrict__ out) {
  const int b = blockIdx.z, y = blockIdx.y;
  const int x = blockIdx.x * blockDim.x + threadIdx.x;
  if (x >= S) return;
  const int* d = desc + 8 * b;
  const int W0 = d[1], i0 = d[2], j0 = d[3], flip = d[6];
  const int* ytab = tables + (size_t)b * 2 * S;
  const int* xtab = ytab + S;
  const int sy = i0 + ytab[y];
  const int sx = j0 + xtab[flip ? S - 1 - x : x];
  const uint8_t v = src[b][(size_t)sy * W0 + sx];
  out[((size_t)b * S + y) * S + x] = (int64_t)lut[v];
}

// ---- image half: crop + Pillow BILINEAR resize (8-bit, separable, 22-bit fixed-point coefficients, support widened
// by the down-scaling factor) + flip + ToTensor (/255) + Normalize ((x - mean) / std), dataset/transform.py:481-553,
// 300-318, 37-86 and run.py:49-55.  Restatement of Pillow's src/libImaging/Resample.c (precompute_coeffs,
// normalize_coeffs_8bpc, ImagingResampleHorizontal/Vertical_8bpc), pinned bit-for-bit by tests/golden.
constexpr int kPrecisionBits = 22;

// coefficient tables: co[b][axis][xx] = {xmin, count, k[0..kmax)}
__global__ void image_coeff_kernel(const int* __restrict__ desc, int B, int S, int kmax, int* __restrict__ co) {
  const int xx = blockIdx.x * blockDim.x + threadIdx.x;
  const int axis = blockIdx.y, b = blockIdx.z;
  if (xx >= S) return;
  const int* d = desc + 8 * b;
  const int in_size = axis == 0 ? d[4] : d[5];          // axis 0: vertical (h), axis 1: horizontal (w)
  int* o = co + (((size_t)b * 2 + axis) * S + xx) * (kmax + 2);
  const double scale = (double)in_size / (double)S;
  const double filterscale = scale < 1.0 ? 1.0 : scale;
  const double support = 1.0 * filterscale;
  const double center = ((double)xx + 0.5) * scale;
  const double ss = 1.0 / filterscale;
  int xmin = (int)(center - support + 0.5);
  if (xmin < 0) xmin = 0;
  int xmax = (int)(center + support + 0.5);
  if (xmax > in_size) xmax = in_size;
  xmax -= xmin;
  if (xmax > kmax) xmax = kmax;                         // cannot happen when kmax = ceil(support) * 2 + 1
  double ww = 0.0;
  for (int x = 0; x < xmax; ++x) {
    double v = ((double)(x + xmin) - center + 0.5) * ss;
    if (v < 0.0) v = -v;
    ww += v < 1.0 ? 1.0 - v : 0.0;
  }
  o[0] = xmin;
  o[1] = xmax;
  for (int x = 0; x < kmax; ++x) {
    double w = 0.0;
    if (x < xmax) {
      double v = ((double)(x + xmin) - center + 0.5) * ss;
      if (v < 0.0) v = -v;
      w = v < 1.0 ? 1.0 - v : 0.0;
      if (ww != 0.0) w /= ww;
    }
    o[2 + x] = w < 0.0 ? (int)(-0.5 + w * (double)(1 << kPrecisionBits)) : (int)(0.5 + w * (double)(1 << kPrecisionBits));
  }
}

__device__ __forceinline__ int clip8(int v) {
  v >>= kPrecisionBits;
  return v < 0 ? 0 : (v > 255 ? 255 : v);
}

// horizontal pass over the rows of the crop: tmp[b][r][xx][c]
__global__ __launch_bounds__(256) void image_hpass_kernel(const uint8_t* const* __restrict__ src, const int* __restrict__ desc,
                                                          int S, int kmax, const int* __restrict__ co, int hmax,
                                                          uint8_t* __restrict__ tmp) {
  const int b = blockIdx.z, r = blockIdx.y;
  const int xx = blockIdx.x * blockDim.x + threadIdx.x;
  const int* d = desc + 8 * b;
  if (xx >= S || r >= d[4]) return;
  const int W0 = d[1], i0 = d[2], j0 = d[3];
  const int* o = co + (((size_t)b * 2 + 1) * S + xx) * (kmax + 2);
  const int xmin = o[0], cnt = o[1];
  const uint8_t* row = src[b] + ((size_t)(i0 + r) * W0 + j0 + xmin) * 3;
  int s0 = 1 << (kPrecisionBits - 1), s1 = s0, s2 = s0;
  for (int x = 0; x < cnt; ++x) {
    const int k = o[2 + x];
    s0 += (int)row[3 * x + 0] * k;
    s1 += (int)row[3 * x + 1] * k;
    s2 += (int)row[3 * x + 2] * k;
  }
  uint8_t* t = tmp + (((size_t)b * hmax + r) * S + xx) * 3;
  t[0] = (uint8_t)clip8(s0);
  t[1] = (uint8_t)clip8(s1);
  t[2] = (uint8_t)clip8(s2);
}

// vertical pass + flip + ToTensor + Normalize: out[b][yy][xx][c] (channels-last storage of [B, 3, S, S])
__global__ __launch_bounds__(256) void image_vpass_kernel(const int* __restrict__ desc, int S, int kmax, const int* __restrict__ co,
                                                          int hmax, const uint8_t* __restrict__ tmp, float m0, float m1, float m2,
                                                          float d0, float d1, float d2, float* __restrict__ out) {
  const int b = blockIdx.z, yy = blockIdx.y;
  const int xx = blockIdx.x * blockDim.x + threadIdx.x;
  if (xx >= S) return;
  const int flip = desc[8 * b + 6];
  const int* o = co + (((size_t)b * 2 + 0) * S + yy) * (kmax + 2);
  const int ymin = o[0], cnt = o[1];
  const int sx = flip ? S - 1 - xx : xx;
  const uint8_t* col = tmp + (((size_t)b * hmax + ymin) * S + sx) * 3;
  int s0 = 1 << (kPrecisionBits - 1), s1 = s0, s2 = s0;
  for (int y = 0; y < cnt; ++y) {
    const int k = o[2 + y];
    const uint8_t* p = col + (size_t)y * S * 3;
    s0 += (int)p[0] * k;
    s1 += (int)p[1] * k;
    s2 += (int)p[2] * k;
  }
  float* q = out + (((size_t)b * S + yy) * S + xx) * 3;
  q[0] = ((float)clip8(s0) / 255.f - m0) / d0;
  q[1] = ((float)clip8(s1) / 255.f - m1) / d1;
  q[2] = ((float)clip8(s2) / 255.f - m2) / d2;
}

}  // namespace
}  // namespace ucd

using namespace ucd;

extern "C" int ucd_image_path(const uint8_t* const* src, const int* desc, int B, int S, int kmax, int hmax, float mean_r,
                              float mean_g, float mean_b, float std_r, float std_g, float std_b, int* coeff_ws, uint8_t* tmp,
                              float* out, ucd_stream_t stream) {
  static const char* fn = "ucd_image_path";
  UCD_REQUIRE(src && desc && coeff_ws && tmp && out && B > 0 && S > 0 && kmax >= 3 && hmax > 0, UCD_EINVAL,
              "%s: bad arguments", fn);
  hipStream_t s = (hipStream_t)stream;
  image_coeff_kernel<<<dim3(ceil_div(S, 64), 2, B), 64, 0, s>>>(desc, B, S, kmax, coeff_ws);
  int rc = check_launch(fn);
  if (rc) return rc;
  image_hpass_kernel<<<dim3(ceil_div(S, 256), hmax, B), 256, 0, s>>>(src, desc, S, kmax, coeff_ws, hmax, tmp);
  rc = check_launch(fn);
  if (rc) return rc;
  image_vpass_kernel<<<dim3(ceil_div(S, 256), S, B), 256, 0, s>>>(desc, S, kmax, coeff_ws, hmax, tmp, mean_r, mean_g, mean_b, std_r,
                                                                  std_g, std_b, out);
  return check_launch(fn);
}

extern "C" int ucd_label_path(const uint8_t* const* src, const int* desc, int B, int S, const uint8_t* lut, int* tables,
                              int64_t* out, ucd_stream_t stream) {
  static const char* fn = "ucd_label_path";
  UCD_REQUIRE(src && desc && lut && tables && out && B > 0 && S > 0, UCD_EINVAL, "%s: bad arguments", fn);
  hipStream_t s = (hipStream_t)stream;
  label_tables_kernel<<<ceil_div(2 * B, 64), 64, 0, s>>>(desc, B, S, tables);
  int rc = check_launch(fn);
  if (rc) return rc;
  label_gather_kernel<<<dim3(ceil_div(S, 256), S, B), 256, 0, s>>>(src, desc, S, lut, tables, out);
  return check_launch(fn);
}
