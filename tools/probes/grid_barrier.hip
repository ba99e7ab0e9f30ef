// Round 6 probe: in-kernel cost of a one-counter grid barrier for co-resident grids of <= 256 workgroups (one per CU), as the fused
// GEMM + statistics + apply launch would use it: {__syncthreads; lane 0: release fence, atomic add, relaxed sc1 poll + s_sleep,
// acquire fence; __syncthreads}.  Ten barriers in a row (monotonic targets), cycles per barrier from s_memrealtime (100 MHz).
// Variants: 0 = one counter; 1 = per-XCD counters (blockIdx & 7) + a top counter (the XCD's last arriver adds to it), everyone polls top.
// usage: grid_barrier <variant> <grid> [threads]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>

__device__ __forceinline__ unsigned ld_rlx(const unsigned* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

template <int VAR>
__global__ void bar_kernel(unsigned* ctr, unsigned long long* out, int nbar, int per_xcd_base) {
  __shared__ unsigned long long t[16];
  const int G = gridDim.x;
  for (int b = 0; b < nbar; ++b) {
    __syncthreads();
    if (threadIdx.x == 0) {
      const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      if (VAR == 0) {
        __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const unsigned target = (unsigned)G * (b + 1);
        while (ld_rlx(ctr) < target) __builtin_amdgcn_s_sleep(1);
      } else {
        const int x = blockIdx.x & 7;
        const unsigned mine = (unsigned)((G - x + 7) / 8);            // workgroups with this blockIdx & 7
        const unsigned old = __hip_atomic_fetch_add(ctr + 64 * (1 + x), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (old + 1 == mine * (b + 1)) __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const unsigned target = 8u * (b + 1);
        while (ld_rlx(ctr) < target) __builtin_amdgcn_s_sleep(1);
      }
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      t[b] = __builtin_amdgcn_s_memrealtime() - t0;
    }
    __syncthreads();
  }
  if (threadIdx.x == 0)
    for (int b = 0; b < nbar; ++b) out[(size_t)blockIdx.x * 16 + b] = t[b];
}

int main(int argc, char** argv) {
  const int var = atoi(argv[1]), grid = atoi(argv[2]), threads = argc > 3 ? atoi(argv[3]) : 512;
  unsigned* ctr; unsigned long long* out;
  hipMalloc(&ctr, 64 * 16 * 4); hipMalloc(&out, (size_t)grid * 16 * 8);
  const int nbar = 10;
  std::vector<unsigned long long> h((size_t)grid * 16);
  for (int rep = 0; rep < 3; ++rep) {
    hipMemset(ctr, 0, 64 * 16 * 4);
    if (var == 0) bar_kernel<0><<<grid, threads>>>(ctr, out, nbar, 0);
    else bar_kernel<1><<<grid, threads>>>(ctr, out, nbar, 0);
    hipDeviceSynchronize();
  }
  hipMemcpy(h.data(), out, h.size() * 8, hipMemcpyDeviceToHost);
  printf("variant %d grid %3d x %d threads: per-barrier wait of a workgroup (us; median / max over workgroups):", var, grid, threads);
  for (int b = 0; b < nbar; ++b) {
    std::vector<double> v;
    for (int g = 0; g < grid; ++g) v.push_back(h[(size_t)g * 16 + b] / 100.0);
    std::sort(v.begin(), v.end());
    printf(" %.2f/%.2f", v[grid / 2], v[grid - 1]);
  }
  printf("\n");
  return 0;
}
