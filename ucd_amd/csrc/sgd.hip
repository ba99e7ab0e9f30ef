// The optimiser step of the train iteration (reference train.py:147 `optim.step()` on run.py:175-186's
// torch.optim.SGD(momentum=0.9, nesterov=True, weight_decay)) as ONE launch over every parameter tensor, the bf16 working
// copy of a convolution weight (ucd_amd/master.py) written in the same pass - instead of torch's nine multi-tensor
// launches plus the cast kernel that refreshes the working copies, and without the per-step Python walk over ~320
// parameters that matters at the 8-GPU per-rank batch, where the step is host-bound.
//
// HBM-bound: 22 bytes per weight element (p, g, m read; p, m, bf16 written), ~60 M elements.  One workgroup takes one
// 4096-element chunk of one tensor (block table built once by the host); 16-byte accesses when the chunk's pointers allow.
//
// Arithmetic: torch's update rule (torch/optim/sgd.py; the fused kernel evaluates it with double intermediates, and so
// does this one - the kernel is bandwidth-bound, fp64 costs nothing):
//     g' = g + wd * p ;  m' = mu * m + g'  (zero-initialised m: the first step gives m' = g', torch's clone) ;
//     d  = g' + mu * m' (Nesterov) | m' ;  p' = p - lr * d ;  w16 = bf16(p') (round to nearest even, like .to(bfloat16))
#include "common.h"

namespace ucd {
namespace {

constexpr int kSgdThreads = 256;
constexpr int kSgdChunk = 4096;

struct SgdUpdate {
  double lr, mu, wd;
  bool nesterov, has_m;
  __device__ __forceinline__ float apply(float p, float g, float& m) const {
    if (wd != 0.0) g = (float)((double)g + wd * (double)p);
    if (has_m) {
      const double mb = mu * (double)m + (double)g;
      m = (float)mb;
      g = nesterov ? (float)((double)g + mu * mb) : (float)mb;
    }
    return (float)((double)p - lr * (double)g);
  }
};

__device__ __forceinline__ void sgd_step_block(const ucd_sgd_tensor* __restrict__ table, const int* __restrict__ blocks,
                                               const ucd_sgd_hyper& hyper) {
  const int t = blocks[2 * blockIdx.x], chunk = blocks[2 * blockIdx.x + 1];
  const ucd_sgd_tensor e = table[t];
  const int gi = e.group;
  SgdUpdate u;
  u.lr = hyper.lr[gi]; u.mu = hyper.momentum[gi]; u.wd = hyper.weight_decay[gi];
  u.nesterov = hyper.nesterov[gi] != 0; u.has_m = e.m != nullptr;
  const long long begin = (long long)chunk * kSgdChunk;
  const int n = (int)(e.n - begin < kSgdChunk ? e.n - begin : kSgdChunk);
  float* p = e.p + begin;
  const float* g = e.g + begin;
  float* m = u.has_m ? e.m + begin : nullptr;
  __hip_bfloat16* w = e.w16 ? (__hip_bfloat16*)e.w16 + begin : nullptr;
  const uintptr_t bits = (uintptr_t)p | (uintptr_t)g | (uintptr_t)m | ((uintptr_t)w << 1);
  if ((bits & 15u) == 0) {
    for (int i = threadIdx.x * 4; i + 4 <= n; i += kSgdThreads * 4) {
      float4 pv = *(const float4*)(p + i), gv = *(const float4*)(g + i), mv = make_float4(0.f, 0.f, 0.f, 0.f);
      if (u.has_m) mv = *(const float4*)(m + i);
      pv.x = u.apply(pv.x, gv.x, mv.x); pv.y = u.apply(pv.y, gv.y, mv.y);
      pv.z = u.apply(pv.z, gv.z, mv.z); pv.w = u.apply(pv.w, gv.w, mv.w);
      *(float4*)(p + i) = pv;
      if (u.has_m) *(float4*)(m + i) = mv;
      if (w) {
        union { __hip_bfloat16 h[4]; uint2 raw; } o;
        o.h[0] = __float2bfloat16(pv.x); o.h[1] = __float2bfloat16(pv.y);
        o.h[2] = __float2bfloat16(pv.z); o.h[3] = __float2bfloat16(pv.w);
        *(uint2*)(w + i) = o.raw;
      }
    }
    const int tail = n & ~3;
    for (int i = tail + threadIdx.x; i < n; i += kSgdThreads) {
      float mv = u.has_m ? m[i] : 0.f;
      const float pv = u.apply(p[i], g[i], mv);
      p[i] = pv;
      if (u.has_m) m[i] = mv;
      if (w) w[i] = __float2bfloat16(pv);
    }
  } else {
    for (int i = threadIdx.x; i < n; i += kSgdThreads) {
      float mv = u.has_m ? m[i] : 0.f;
      const float pv = u.apply(p[i], g[i], mv);
      p[i] = pv;
      if (u.has_m) m[i] = mv;
      if (w) w[i] = __float2bfloat16(pv);
    }
  }
}


__global__ __launch_bounds__(kSgdThreads) void sgd_step_kernel(const ucd_sgd_tensor* __restrict__ table,
                                                              const int* __restrict__ blocks, ucd_sgd_hyper hyper) {
  sgd_step_block(table, blocks, hyper);
}

// The same step with the hyper-parameters read from device memory: the form a captured hipGraph replays while the learning
// rate changes every iteration (PolyLR, train.py:150-151) - a by-value kernel argument would be frozen into the graph.
__global__ __launch_bounds__(kSgdThreads) void sgd_step_dev_kernel(const ucd_sgd_tensor* __restrict__ table,
                                                                  const int* __restrict__ blocks,
                                                                  const ucd_sgd_hyper* __restrict__ hyper) {
  sgd_step_block(table, blocks, *hyper);
}

__global__ void sgd_hyper_store_kernel(ucd_sgd_hyper* dst, ucd_sgd_hyper value) { *dst = value; }

}  // namespace
}  // namespace ucd

using namespace ucd;

extern "C" {

int ucd_sgd_chunk(void) { return kSgdChunk; }

int ucd_sgd_step(const ucd_sgd_tensor* table, const int* blocks, int n_blocks, const ucd_sgd_hyper* hyper, ucd_stream_t stream) {
  static const char* fn = "ucd_sgd_step";
  UCD_REQUIRE(n_blocks >= 0, UCD_EINVAL, "%s: n_blocks = %d", fn, n_blocks);
  if (n_blocks == 0) return 0;
  UCD_REQUIRE(table && blocks && hyper, UCD_EINVAL, "%s: table / blocks / hyper is NULL", fn);
  for (int gidx = 0; gidx < UCD_SGD_MAX_GROUPS; ++gidx)
    UCD_REQUIRE(hyper->weight_decay[gidx] >= 0.0 && hyper->momentum[gidx] >= 0.0, UCD_EINVAL,
                "%s: group %d: negative weight decay or momentum", fn, gidx);
  sgd_step_kernel<<<(unsigned)n_blocks, kSgdThreads, 0, (hipStream_t)stream>>>(table, blocks, *hyper);
  return check_launch(fn);
}

int ucd_sgd_hyper_store(ucd_sgd_hyper* device_hyper, const ucd_sgd_hyper* hyper, ucd_stream_t stream) {
  static const char* fn = "ucd_sgd_hyper_store";
  UCD_REQUIRE(device_hyper && hyper, UCD_EINVAL, "%s: device_hyper / hyper is NULL", fn);
  for (int gidx = 0; gidx < UCD_SGD_MAX_GROUPS; ++gidx)
    UCD_REQUIRE(hyper->weight_decay[gidx] >= 0.0 && hyper->momentum[gidx] >= 0.0, UCD_EINVAL,
                "%s: group %d: negative weight decay or momentum", fn, gidx);
  sgd_hyper_store_kernel<<<1, 1, 0, (hipStream_t)stream>>>(device_hyper, *hyper);      // the values travel as a kernel argument
  return check_launch(fn);
}

int ucd_sgd_step_dev(const ucd_sgd_tensor* table, const int* blocks, int n_blocks, const ucd_sgd_hyper* device_hyper,
                     ucd_stream_t stream) {
  static const char* fn = "ucd_sgd_step_dev";
  UCD_REQUIRE(n_blocks >= 0, UCD_EINVAL, "%s: n_blocks = %d", fn, n_blocks);
  if (n_blocks == 0) return 0;
  UCD_REQUIRE(table && blocks && device_hyper, UCD_EINVAL, "%s: table / blocks / device_hyper is NULL", fn);
  sgd_step_dev_kernel<<<(unsigned)n_blocks, kSgdThreads, 0, (hipStream_t)stream>>>(table, blocks, device_hyper);
  return check_launch(fn);
}

}  // extern "C"
