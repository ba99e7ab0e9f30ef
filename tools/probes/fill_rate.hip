// Round 6 probe: how many bytes per clock ONE CU can pull from L2 into LDS / registers, by path and shape.
//   mode 0: LDS-DMA (buffer_load_dwordx4 ... lds), 8 rows x 128 B per wave instruction (the GEMM kernels' A / W tile fill)
//   mode 1: buffer_load_dwordx4 to VGPRs (same addresses), values xor-reduced (no LDS write)
//   mode 2: buffer_load_dwordx4 to VGPRs + ds_write_b128 into the swizzled tile
//   mode 3: LDS-DMA, one row of 1 KiB contiguous per wave instruction (fully coalesced)
// One 24 KB "stage" per step (192 rows x 128 B), `waves` loader waves, 3 stages in flight (waits with counted vmcnt).
// Every workgroup streams the same `rows` x pitch byte region (L2 / MALL resident), starting at a per-workgroup row offset.
// usage: fill_rate <mode> <waves> <pitch bytes> <grid> [steps]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
typedef __attribute__((address_space(3))) void* lptr_t;

struct P { const unsigned char* src; size_t bytes; int pitch, rows, steps; unsigned long long* out; unsigned* sink; };

template <int MODE>
__global__ __launch_bounds__(1024, 1) void fill_kernel(P p, int waves) {
  extern __shared__ __attribute__((aligned(1024))) unsigned char smem[];
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const auto rs = __builtin_amdgcn_make_buffer_rsrc((void*)p.src, 0, (int)p.bytes, 0x00020000);
  constexpr int kStage = 24576, NST = 3;
  const int pieces = 24 / waves;                       // 1 KiB pieces per wave and stage
  const int row0 = (blockIdx.x * 192) % p.rows;
  unsigned acc = 0;
  __syncthreads();
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int s = 0; s < p.steps; ++s) {
    unsigned char* dst = smem + (s % NST) * kStage;
    const int kofs = (s * 128) % p.pitch;              // walk along the rows like a K loop
    for (int i = 0; i < pieces; ++i) {
      const int piece = wave * pieces + i;
      unsigned off;
      if (MODE == 3) off = (unsigned)(((size_t)((row0 + piece) % p.rows) * p.pitch + ((kofs * 8) % p.pitch) / 1024 * 1024 + lane * 16) % p.bytes);
      else off = (unsigned)(((size_t)((row0 + piece * 8 + lane / 8) % p.rows) * p.pitch + kofs + (lane % 8) * 16));
      if (MODE == 0 || MODE == 3) {
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lptr_t)(dst + piece * 1024), 16, (int)off, 0, 0, 0);
      } else {
        const auto v = __builtin_amdgcn_raw_buffer_load_b128(rs, (int)off, 0, 0);
        if (MODE == 1) acc ^= v[0] ^ v[1] ^ v[2] ^ v[3];
        else *reinterpret_cast<uint4*>(dst + piece * 1024 + lane * 16) = make_uint4(v[0], v[1], v[2], v[3]);
      }
    }
    if (MODE == 0 || MODE == 3) {
      // keep two stages in flight per wave
      if (s >= 2) {
        if (pieces == 6) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
        else if (pieces == 3) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
        else if (pieces == 12) asm volatile("s_waitcnt vmcnt(24)" ::: "memory");
        else if (pieces == 2) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      }
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  if (threadIdx.x == 0) p.out[blockIdx.x] = t1 - t0;
  if (acc == 0x12345678u) p.sink[0] = acc + smem[lane];
}

int main(int argc, char** argv) {
  const int mode = atoi(argv[1]), waves = atoi(argv[2]), pitch = atoi(argv[3]), grid = atoi(argv[4]);
  const int steps = argc > 5 ? atoi(argv[5]) : 64;
  const int rows = 4096;
  P p; p.pitch = pitch; p.rows = rows; p.steps = steps; p.bytes = (size_t)rows * pitch;
  unsigned char* src; hipMalloc(&src, p.bytes); hipMemset(src, 1, p.bytes); p.src = src;
  hipMalloc(&p.out, grid * 8); hipMalloc(&p.sink, 64);
  auto launch = [&]() {
    const int lds = 3 * 24576;
    switch (mode) {
      case 0: hipFuncSetAttribute((const void*)fill_kernel<0>, hipFuncAttributeMaxDynamicSharedMemorySize, lds); fill_kernel<0><<<grid, waves * 64, lds>>>(p, waves); break;
      case 1: hipFuncSetAttribute((const void*)fill_kernel<1>, hipFuncAttributeMaxDynamicSharedMemorySize, lds); fill_kernel<1><<<grid, waves * 64, lds>>>(p, waves); break;
      case 2: hipFuncSetAttribute((const void*)fill_kernel<2>, hipFuncAttributeMaxDynamicSharedMemorySize, lds); fill_kernel<2><<<grid, waves * 64, lds>>>(p, waves); break;
      default: hipFuncSetAttribute((const void*)fill_kernel<3>, hipFuncAttributeMaxDynamicSharedMemorySize, lds); fill_kernel<3><<<grid, waves * 64, lds>>>(p, waves); break;
    }
  };
  for (int i = 0; i < 3; ++i) launch();
  hipDeviceSynchronize();
  std::vector<unsigned long long> h(grid);
  hipMemcpy(h.data(), p.out, grid * 8, hipMemcpyDeviceToHost);
  std::sort(h.begin(), h.end());
  const double cyc = (double)h[grid / 2];
  printf("mode %d waves %2d pitch %5d grid %3d: %7.0f cycles for %d x 24 KB = %5.1f B/clk/CU (median WG; max %llu)\n", mode, waves, pitch, grid,
         cyc, steps, (double)steps * 24576 / cyc, h[grid - 1]);
  return 0;
}
