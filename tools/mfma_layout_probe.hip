// GPU probe (not part of the product): verifies on hardware the gfx950 lane maps the fp16 contrastive
// kernel relies on, with exact small-integer data:
//  (1) v_mfma_f32_32x32x16_f16 operand / accumulator layout,
//  (2) ds_read_b64_tr_b16 (4 rows x 16 columns, column-major delivery per 16-lane group),
//  (3) a 32x32 accumulator converted to f16 in place as the B operand of a following MFMA ("A.X" form).
// build: hipcc --offload-arch=gfx950 -O2 tools/mfma_layout_probe.hip -o gpurun_out/mfma_probe && run it.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <cmath>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef __fp16 h4 __attribute__((ext_vector_type(4)));

__global__ void k_mfma(const _Float16* A /*32x16*/, const _Float16* B /*16x32*/, float* D /*32x32*/) {
  int l = threadIdx.x, r = l & 31, h = l >> 5;
  f16x8 a, b;
  for (int j = 0; j < 8; ++j) { a[j] = A[r * 16 + 8 * h + j]; b[j] = B[(8 * h + j) * 32 + r]; }
  f32x16 acc = {0};
  acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc, 0, 0, 0);
  for (int reg = 0; reg < 16; ++reg) D[((reg & 3) + 8 * (reg >> 2) + 4 * h) * 32 + r] = acc[reg];
}

// tile[32 rows j][PITCH halfs]; out[l][0..3] = elements the lane received
__global__ void k_tr(const _Float16* tile, int pitch, int R0, int n0, _Float16* out) {
  extern __shared__ __attribute__((aligned(16))) _Float16 lds[];
  for (int i = threadIdx.x; i < 32 * pitch; i += 64) lds[i] = tile[i];
  __syncthreads();
  int l = threadIdx.x, h = l >> 5, g = (l >> 4) & 1, q = (l & 15) >> 2, p = l & 3;
  const _Float16* addr = lds + (R0 + 4 * h + q) * pitch + n0 + 16 * g + 4 * p;
  h4 v = __builtin_amdgcn_ds_read_tr16_b64_v4f16((__attribute__((address_space(3))) h4*)addr);
  for (int e = 0; e < 4; ++e) out[l * 4 + e] = (_Float16)v[e];
}

// Y[n][i] = sum_j Ct[n][j] * X[j][i], X produced by a first MFMA (X = C1 . A1^T), then fed as B operand.
__global__ void k_chain(const _Float16* Cm /*32 j x 16 k*/, const _Float16* Am /*32 i x 16 k*/, const _Float16* C2 /*32 j x 32 n*/,
                        float* Y /*32 n x 32 i*/, float* Xout) {
  int l = threadIdx.x, r = l & 31, h = l >> 5;
  f16x8 a, b;
  for (int j = 0; j < 8; ++j) { a[j] = Cm[r * 16 + 8 * h + j]; b[j] = Am[r * 16 + 8 * h + j]; }  // B[k][col i] = A_i[k]
  f32x16 x = {0};
  x = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, x, 0, 0, 0);      // x[j][i]
  for (int reg = 0; reg < 16; ++reg) Xout[((reg & 3) + 8 * (reg >> 2) + 4 * h) * 32 + r] = x[reg];
  f32x16 y = {0};
  for (int s = 0; s < 2; ++s) {
    f16x8 bf, af;
    for (int jj = 0; jj < 8; ++jj) {
      bf[jj] = (_Float16)x[8 * s + jj];
      int jrow = 16 * s + 8 * (jj >> 2) + 4 * h + (jj & 3);
      af[jj] = C2[jrow * 32 + r];                                   // A[n = r][k = jrow] = C2[jrow][n]
    }
    y = __builtin_amdgcn_mfma_f32_32x32x16_f16(af, bf, y, 0, 0, 0);
  }
  for (int reg = 0; reg < 16; ++reg) Y[((reg & 3) + 8 * (reg >> 2) + 4 * h) * 32 + r] = y[reg];
}

int main() {
  int bad = 0;
  {  // (1)
    std::vector<_Float16> A(32 * 16), B(16 * 32);
    for (int i = 0; i < 32; ++i) for (int k = 0; k < 16; ++k) A[i * 16 + k] = (_Float16)((i * 3 + k * 5) % 7 - 3);
    for (int k = 0; k < 16; ++k) for (int j = 0; j < 32; ++j) B[k * 32 + j] = (_Float16)((k * 2 + j * 3) % 5 - 2);
    _Float16 *dA, *dB; float* dD;
    hipMalloc(&dA, A.size() * 2); hipMalloc(&dB, B.size() * 2); hipMalloc(&dD, 32 * 32 * 4);
    hipMemcpy(dA, A.data(), A.size() * 2, hipMemcpyHostToDevice); hipMemcpy(dB, B.data(), B.size() * 2, hipMemcpyHostToDevice);
    k_mfma<<<1, 64>>>(dA, dB, dD);
    std::vector<float> D(32 * 32); hipMemcpy(D.data(), dD, D.size() * 4, hipMemcpyDeviceToHost);
    int e = 0;
    for (int i = 0; i < 32; ++i) for (int j = 0; j < 32; ++j) { float s = 0; for (int k = 0; k < 16; ++k) s += (float)A[i * 16 + k] * (float)B[k * 32 + j]; if (s != D[i * 32 + j]) ++e; }
    printf("mfma_f32_32x32x16_f16 layout mismatches: %d\n", e); bad += e;
  }
  {  // (2)
    const int pitch = 280;
    std::vector<_Float16> T(32 * pitch);
    for (int j = 0; j < 32; ++j) for (int n = 0; n < pitch; ++n) T[j * pitch + n] = (_Float16)(j * 64 + (n % 64));  // exact in fp16 (< 2048)
    _Float16 *dT, *dO; hipMalloc(&dT, T.size() * 2); hipMalloc(&dO, 64 * 4 * 2);
    hipMemcpy(dT, T.data(), T.size() * 2, hipMemcpyHostToDevice);
    int e = 0;
    for (int R0 : {0, 8, 16, 24}) for (int n0 : {0, 32, 224}) {
      k_tr<<<1, 64, 32 * pitch * 2>>>(dT, pitch, R0, n0, dO);
      std::vector<_Float16> O(256); hipMemcpy(O.data(), dO, 512, hipMemcpyDeviceToHost);
      for (int l = 0; l < 64; ++l) { int h = l >> 5, g = (l >> 4) & 1, i = l & 15;
        for (int q = 0; q < 4; ++q) { float want = (float)T[(R0 + 4 * h + q) * pitch + n0 + 16 * g + i]; if ((float)O[l * 4 + q] != want) { if (e < 5) printf("  tr mismatch R0=%d n0=%d lane %d elem %d got %g want %g\n", R0, n0, l, q, (float)O[l * 4 + q], want); ++e; } } }
    }
    printf("ds_read_b64_tr_b16 mapping mismatches: %d\n", e); bad += e;
  }
  {  // (3)
    std::vector<_Float16> Cm(32 * 16), Am(32 * 16), C2(32 * 32);
    for (int j = 0; j < 32; ++j) for (int k = 0; k < 16; ++k) Cm[j * 16 + k] = (_Float16)((j + 2 * k) % 3 - 1);
    for (int i = 0; i < 32; ++i) for (int k = 0; k < 16; ++k) Am[i * 16 + k] = (_Float16)((3 * i + k) % 4 - 1);
    for (int j = 0; j < 32; ++j) for (int n = 0; n < 32; ++n) C2[j * 32 + n] = (_Float16)((5 * j + 3 * n) % 5 - 2);
    _Float16 *dC, *dA, *dC2; float *dY, *dX;
    hipMalloc(&dC, Cm.size() * 2); hipMalloc(&dA, Am.size() * 2); hipMalloc(&dC2, C2.size() * 2); hipMalloc(&dY, 4096); hipMalloc(&dX, 4096);
    hipMemcpy(dC, Cm.data(), Cm.size() * 2, hipMemcpyHostToDevice); hipMemcpy(dA, Am.data(), Am.size() * 2, hipMemcpyHostToDevice);
    hipMemcpy(dC2, C2.data(), C2.size() * 2, hipMemcpyHostToDevice);
    k_chain<<<1, 64>>>(dC, dA, dC2, dY, dX);
    std::vector<float> Y(1024), X(1024); hipMemcpy(Y.data(), dY, 4096, hipMemcpyDeviceToHost); hipMemcpy(X.data(), dX, 4096, hipMemcpyDeviceToHost);
    int e = 0;
    std::vector<float> Xr(1024);
    for (int j = 0; j < 32; ++j) for (int i = 0; i < 32; ++i) { float s = 0; for (int k = 0; k < 16; ++k) s += (float)Cm[j * 16 + k] * (float)Am[i * 16 + k]; Xr[j * 32 + i] = s; if (s != X[j * 32 + i]) ++e; }
    for (int n = 0; n < 32; ++n) for (int i = 0; i < 32; ++i) { float s = 0; for (int j = 0; j < 32; ++j) s += (float)C2[j * 32 + n] * Xr[j * 32 + i]; if (s != Y[n * 32 + i]) ++e; }
    printf("accumulator-as-B-operand chain mismatches: %d\n", e); bad += e;
  }
  printf(bad ? "PROBE FAILED\n" : "PROBE OK\n");
  return bad ? 1 : 0;
}
