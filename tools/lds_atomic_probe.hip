// LDS atomic rates on gfx950: ds_add_f32 / ds_add_u32 / ds_add_u64 / ds_pk_add_bf16?, conflict-free (lane-private words), two lanes per
// word, sixteen lanes per word.  build: hipcc --offload-arch=gfx950 -O3 tools/lds_atomic_probe.hip -o /tmp/lds_atomic_probe
#include <hip/hip_runtime.h>
#include <cstdio>
template <int MODE, int SHARE>
__global__ __launch_bounds__(256) void k(float* out, int iters) {
  __shared__ unsigned long long s[4096];
  for (int i = threadIdx.x; i < 4096; i += 256) s[i] = 0;
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int word = wave * 1024 + (lane / SHARE);
  float* sf = reinterpret_cast<float*>(s);
  unsigned* su = reinterpret_cast<unsigned*>(s);
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int j = 0; j < 16; ++j) {
      if (MODE == 0) atomicAdd(&sf[word + 64 * j], 1.0f);
      if (MODE == 1) atomicAdd(&su[word + 64 * j], 1u);
      if (MODE == 2) atomicAdd(&s[word + 64 * j], 1ull);
      if (MODE == 3) atomicAdd(&reinterpret_cast<double*>(s)[word + 64 * j], 1.0);
    }
  }
  __syncthreads();
  if (threadIdx.x == 0) out[blockIdx.x] = sf[0] + (float)s[1];
}
template <int MODE, int SHARE>
void run(const char* name) {
  float* out; hipMalloc(&out, 4096 * 4);
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  const int iters = 200, blocks = 512;
  k<MODE, SHARE><<<blocks, 256>>>(out, iters);
  hipEventRecord(a);
  k<MODE, SHARE><<<blocks, 256>>>(out, iters);
  hipEventRecord(b); hipEventSynchronize(b);
  float ms; hipEventElapsedTime(&ms, a, b);
  // per CU: 2 blocks x 4 waves x iters x 16 instructions
  const double instr_per_cu = 2.0 * 4 * iters * 16;
  printf("%-10s lanes/word %2d: %8.1f us  -> %6.1f cycles per wave instruction per CU (2.1 GHz)\n", name, SHARE, ms * 1e3,
         ms * 1e-3 * 2.1e9 / instr_per_cu);
  hipFree(out);
}
int main() {
  run<0, 1>("add_f32"); run<0, 2>("add_f32"); run<0, 16>("add_f32");
  run<1, 1>("add_u32"); run<1, 2>("add_u32"); run<1, 16>("add_u32");
  run<2, 1>("add_u64"); run<2, 2>("add_u64"); run<2, 16>("add_u64");
  run<3, 1>("add_f64"); run<3, 2>("add_f64"); run<3, 16>("add_f64");
  return 0;
}
