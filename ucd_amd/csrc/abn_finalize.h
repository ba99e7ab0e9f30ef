// Batch statistics -> normalisation constants: shared by the ABN kernels (abn.hip) and by the statistics epilogue of the
// 1x1-convolution GEMM (conv1x1.hip).
#pragma once
#include "common.h"

namespace ucd {

struct FinalizeArgs {
  const float* kshift;
  const float* weight;
  float* running_mean;
  float* running_var;
  float* mean;
  float* invstd;
  float* scale;
  float count, momentum, eps;
  float* pack;   // PACK mode: [mean_r | M2_r] of this rank for the cross-rank combination
  int abs_gamma; // scale = (|weight| + eps) * invstd (InPlaceABN / InPlaceABNSync) instead of weight * invstd
};

__device__ __forceinline__ void finalize_moments(int c, float mean, float m2, const FinalizeArgs& f) {
  const float var = fmaxf(m2 / f.count, 0.f);
  const float invstd = 1.f / sqrtf(var + f.eps);
  if (f.running_mean) f.running_mean[c] = (1.f - f.momentum) * f.running_mean[c] + f.momentum * mean;
  if (f.running_var) {
    const float unbiased = f.count > 1.f ? var * (f.count / (f.count - 1.f)) : var;
    f.running_var[c] = (1.f - f.momentum) * f.running_var[c] + f.momentum * unbiased;
  }
  f.mean[c] = mean;
  f.invstd[c] = invstd;
  f.scale[c] = (f.weight ? gamma_eff(f.weight[c], f.eps, f.abs_gamma) : 1.f) * invstd;
}

// The finalising thread's inputs, fetched at the START of a reduction kernel (finalize_prefetch) instead of behind its barriers:
// the second-stage kernels are chains of dependent round trips (5 - 7 us for a few KB, ~220 launches per step), and these three
// loads would otherwise add one more trip at the very end.
struct FinalizeIn { float weight, running_mean, running_var; };
__device__ __forceinline__ FinalizeIn finalize_prefetch(int c, const FinalizeArgs& f) {
  FinalizeIn in;
  in.weight = f.weight ? f.weight[c] : 1.f;
  in.running_mean = f.running_mean ? f.running_mean[c] : 0.f;
  in.running_var = f.running_var ? f.running_var[c] : 0.f;
  return in;
}
__device__ __forceinline__ void finalize_channel_pre(int c, float s, float ss, float kshift, const FinalizeArgs& f, const FinalizeIn& in) {
  const float inv_n = 1.f / f.count;
  const float d = s * inv_n;                       // mean - k
  const float mean = kshift + d;
  const float var = fmaxf((ss - s * d) * inv_n, 0.f);
  const float invstd = 1.f / sqrtf(var + f.eps);
  if (f.running_mean) f.running_mean[c] = (1.f - f.momentum) * in.running_mean + f.momentum * mean;
  if (f.running_var) {
    const float unbiased = f.count > 1.f ? var * (f.count / (f.count - 1.f)) : var;
    f.running_var[c] = (1.f - f.momentum) * in.running_var + f.momentum * unbiased;
  }
  f.mean[c] = mean;
  f.invstd[c] = invstd;
  f.scale[c] = (f.weight ? gamma_eff(in.weight, f.eps, f.abs_gamma) : 1.f) * invstd;
}

__device__ __forceinline__ void finalize_channel(int c, float s, float ss, const FinalizeArgs& f) {
  const float inv_n = 1.f / f.count;
  const float d = s * inv_n;                       // mean - k
  const float mean = (f.kshift ? f.kshift[c] : 0.f) + d;
  const float var = fmaxf((ss - s * d) * inv_n, 0.f);
  const float invstd = 1.f / sqrtf(var + f.eps);
  if (f.running_mean) f.running_mean[c] = (1.f - f.momentum) * f.running_mean[c] + f.momentum * mean;
  if (f.running_var) {
    const float unbiased = f.count > 1.f ? var * (f.count / (f.count - 1.f)) : var;
    f.running_var[c] = (1.f - f.momentum) * f.running_var[c] + f.momentum * unbiased;
  }
  f.mean[c] = mean;
  f.invstd[c] = invstd;
  f.scale[c] = (f.weight ? gamma_eff(f.weight[c], f.eps, f.abs_gamma) : 1.f) * invstd;
}

}  // namespace ucd
