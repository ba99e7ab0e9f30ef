// Does an out-of-range buffer_load ... lds (LDS-DMA through a buffer descriptor) WRITE ZEROS into LDS, or leave the bytes alone?
// build + run on the GPU box: hipcc --offload-arch=gfx950 tools/lds_dma_oob_probe.hip -o /tmp/oob && /tmp/oob
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __attribute__((address_space(3))) void* lptr_t;
__global__ void k(const float* p, int nbytes, float* out) {
  extern __shared__ __attribute__((aligned(1024))) unsigned char smem[];
  float* f = (float*)smem;
  for (int i = threadIdx.x; i < 256; i += 64) f[i] = -7.f;      // poison
  __syncthreads();
  auto rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)p, 0, nbytes, 0x00020000);
  int off = threadIdx.x * 16;
  if (threadIdx.x & 1) off = 0x7ffffff0;                        // out of range for the odd lanes
  __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (lptr_t)smem, 16, off, 0, 0, 0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  for (int i = threadIdx.x; i < 256; i += 64) out[i] = f[i];
}
int main() {
  float h[256], *d, *o;
  for (int i = 0; i < 256; ++i) h[i] = (float)(i + 1);
  hipMalloc(&d, sizeof(h)); hipMalloc(&o, sizeof(h));
  hipMemcpy(d, h, sizeof(h), hipMemcpyHostToDevice);
  k<<<1, 64, 1024>>>(d, (int)sizeof(h), o);
  hipMemcpy(h, o, sizeof(h), hipMemcpyDeviceToHost);
  printf("lane0 (in range) : %g %g %g %g\nlane1 (OUT of range): %g %g %g %g   (0 = zero-filled, -7 = left alone)\n", h[0], h[1], h[2], h[3], h[4], h[5], h[6], h[7]);
  return 0;
}
