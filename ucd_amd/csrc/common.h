// Shared device/host helpers of libucd_hip (gfx950 only; wave = 64).
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_bf16.h>
#include <stdint.h>
#include <stdio.h>
#include <stdarg.h>
#include <string.h>

#include "../../include/ucd_hip.h"

namespace ucd {

constexpr int kWave = 64;

// ---- error plumbing ----------------------------------------------------------------------------
void set_error(const char* fmt, ...);
int check_launch(const char* what);  // hipGetLastError -> 0 or positive hipError_t (message recorded)

#define UCD_REQUIRE(cond, code, ...)      \
  do {                                    \
    if (!(cond)) {                        \
      ucd::set_error(__VA_ARGS__);        \
      return (code);                      \
    }                                     \
  } while (0)

// Opt a kernel in to `bytes` of dynamic LDS (> 64 KiB needs it).  Done on every call: the attribute belongs to the
// current device's copy of the code object, so caching "already set" in a static would be wrong for a second device
// and racy between threads; the call is a host-side table update (no device work).
#define UCD_TRY_LDS(kernel, bytes)                                                                                  \
  do {                                                                                                              \
    hipError_t _e = hipFuncSetAttribute((const void*)(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (bytes)); \
    if (_e != hipSuccess) {                                                                                         \
      (void)hipGetLastError();                                                                                      \
      ucd::set_error("%s: hipFuncSetAttribute(%s): %s", fn, #kernel, hipGetErrorString(_e));                        \
      return (int)_e;                                                                                               \
    }                                                                                                               \
  } while (0)

// RCCL on the caller's stream (comm.hip)
int comm_all_gather_f32(void* comm, const float* send, float* recv, size_t count, hipStream_t s);
int comm_all_reduce_sum_f32(void* comm, float* buf, size_t count, hipStream_t s);

inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }
inline int ceil_div(int a, int b) { return (a + b - 1) / b; }
inline size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

// ---- 16-byte vectors of activations ------------------------------------------------------------
// One lane always moves 16 bytes: 4 floats or 8 bf16.
template <typename T> struct Vec;
template <> struct Vec<float> {
  static constexpr int N = 4;
  float4 raw;
  __device__ __forceinline__ void load(const float* p) { raw = *reinterpret_cast<const float4*>(p); }
  __device__ __forceinline__ void store(float* p) const { *reinterpret_cast<float4*>(p) = raw; }
  __device__ __forceinline__ float get(int i) const { return (&raw.x)[i]; }
  __device__ __forceinline__ void set(int i, float v) { (&raw.x)[i] = v; }
};
template <> struct Vec<__hip_bfloat16> {
  static constexpr int N = 8;
  uint4 raw;
  __device__ __forceinline__ void load(const __hip_bfloat16* p) { raw = *reinterpret_cast<const uint4*>(p); }
  __device__ __forceinline__ void store(__hip_bfloat16* p) const { *reinterpret_cast<uint4*>(p) = raw; }
  __device__ __forceinline__ float get(int i) const {
    uint32_t w = (&raw.x)[i >> 1];
    return __uint_as_float((i & 1) ? (w & 0xFFFF0000u) : (w << 16));
  }
  __device__ __forceinline__ void set(int i, float v) {
    // round-to-nearest-even via the hardware conversion (keeps NaN a NaN)
    __hip_bfloat16 b = __float2bfloat16(v);
    uint32_t bits = *reinterpret_cast<const uint16_t*>(&b);
    uint32_t& w = (&raw.x)[i >> 1];
    w = (i & 1) ? ((w & 0x0000FFFFu) | (bits << 16)) : ((w & 0xFFFF0000u) | bits);
  }
};

// ---- eight bf16 activations as four float pairs ---------------------------------------------------
// The element-wise passes (ABN apply / backward) are VALU-issue-bound when written per element: ~17 instructions per value
// (the unpack / select / re-insert of Vec<bf16>::get/set, one exec-masked branch per nullable operand) against ~5 with
// packed-fp32 math (v_pk_add_f32 / v_pk_mul_f32; the library is built without fp contraction), one v_cvt_pk_bf16_f32 per PAIR and the nullable operands
// as template flags (round 4: abn_apply on a 13 MB layer 11.7 -> see profiles/r04_abn_fast.txt).
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x2v __attribute__((ext_vector_type(2)));
struct Pack8 {
  f32x2 p[4];          // p[j] = elements (2j, 2j + 1)
};
__device__ __forceinline__ Pack8 unpack8(const uint4& raw) {
  Pack8 o;
  const uint32_t w[4] = {raw.x, raw.y, raw.z, raw.w};
#pragma unroll
  for (int j = 0; j < 4; ++j) o.p[j] = f32x2{__uint_as_float(w[j] << 16), __uint_as_float(w[j] & 0xFFFF0000u)};
  return o;
}
__device__ __forceinline__ uint4 pack8(const Pack8& f) {   // round-to-nearest-even, the same instruction as __float2bfloat16
  uint32_t w[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const bf16x2v b = __builtin_convertvector(f.p[j], bf16x2v);
    w[j] = __builtin_bit_cast(uint32_t, b);
  }
  return make_uint4(w[0], w[1], w[2], w[3]);
}
__device__ __forceinline__ Pack8 load_f8(const float* p) {  // eight per-channel constants (32-byte aligned: channel groups of 8)
  const float4 a = *reinterpret_cast<const float4*>(p), b = *reinterpret_cast<const float4*>(p + 4);
  Pack8 o;
  o.p[0] = f32x2{a.x, a.y}; o.p[1] = f32x2{a.z, a.w}; o.p[2] = f32x2{b.x, b.y}; o.p[3] = f32x2{b.z, b.w};
  return o;
}
// leaky_relu on a pair: z > 0 ? z : z * slope (slope = 1: identity)
__device__ __forceinline__ f32x2 leaky2(f32x2 z, float slope) {
  const f32x2 zs = z * slope;
  return f32x2{z.x > 0.f ? z.x : zs.x, z.y > 0.f ? z.y : zs.y};
}
// g * act'(s) for leaky_relu: s > 0 ? g : g * slope
__device__ __forceinline__ f32x2 leaky_grad2(f32x2 g, f32x2 s, float slope) {
  const f32x2 gs = g * slope;
  return f32x2{s.x > 0.f ? g.x : gs.x, s.y > 0.f ? g.y : gs.y};
}

template <int ACT>
__device__ __forceinline__ float act_fwd(float z, float slope) {
  if (ACT == UCD_ACT_LEAKY_RELU) return z > 0.f ? z : z * slope;
  if (ACT == UCD_ACT_ELU) return z > 0.f ? z : slope * expm1f(z);      // slope = alpha
  return z;
}
// derivative at z, given z itself or (from_y) the activation's output y: leaky_relu needs only the sign
// (sign y == sign z for slope > 0); elu has d/dz = alpha * exp(z) = y + alpha on the negative side
template <int ACT>
__device__ __forceinline__ float act_grad(float z_or_y, float slope, bool from_y = false) {
  if (ACT == UCD_ACT_LEAKY_RELU) return z_or_y > 0.f ? 1.f : slope;
  if (ACT == UCD_ACT_ELU) return z_or_y > 0.f ? 1.f : (from_y ? z_or_y + slope : slope * __expf(z_or_y));
  return 1.f;
}

// inplace_abn's in-place variants normalise with gamma~ = |gamma| + eps (the published forward of InPlaceABN /
// InPlaceABNSync; plain ABN keeps F.batch_norm's raw gamma): selected by the UCD_NORM_ABS_GAMMA bit of `act`
__host__ __device__ __forceinline__ float gamma_eff(float w, float eps, int abs_gamma) {
  return abs_gamma ? fabsf(w) + eps : w;
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v = fmaxf(v, __shfl_xor(v, off, 64));
  return v;
}

}  // namespace ucd
