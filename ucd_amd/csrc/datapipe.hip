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

}  // namespace
}  // namespace ucd

using namespace ucd;

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
