// Round 6 probe: the loader-wave K loop of csrc/conv1x1.hip (128 x BN tile, K steps of 64, LDS-DMA fills by dedicated waves) as a
// standalone GEMM with s_memtime stamps, to see where a K step of a ONE-workgroup-per-CU grid spends its ~750 cycles.
// Variants (argv[1]): 0 = the library's loop (4 MFMA + 4 loader waves, 3 stages); 1 = 8 MFMA waves as two K-slice groups (each
// group takes two of the four 16-deep slices of a step: two MFMA waves per SIMD, partial sums combined through LDS at the end);
// 2 = variant 0 with all fragment reads of a step issued up front (12 reads in flight, MFMAs behind counted waits).
// build: hipcc -O3 --offload-arch=gfx950 -o lw_timeline tools/probes/lw_timeline.hip ; run: ./lw_timeline <variant> [M N K]
#include <hip/hip_runtime.h>
#include <hip/hip_bf16.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __attribute__((address_space(3))) void* lptr_t;

constexpr int BM = 128, BN = 64, BK = 64, NST = 3;
constexpr int kStage = (BM + BN) * BK * 2;
__device__ __forceinline__ int swz(int row, int slot) { return row * 128 + ((slot ^ ((row >> 1) & 7)) << 4); }

struct P {
  const __hip_bfloat16* A; const __hip_bfloat16* W; float* Y; int M, N, K; unsigned long long* stamps; int nstamp;
};

template <int VAR>
__global__ __launch_bounds__(VAR == 1 ? 768 : 512, 1) void lw_kernel(P p) {
  constexpr int NC = VAR == 1 ? 8 : 4, NL = 4;
  extern __shared__ __attribute__((aligned(1024))) unsigned char smem[];
  unsigned char* As = smem;
  unsigned char* Bs = smem + BM * BK * 2;
  const int tiles_n = p.N / BN;
  const int tm = blockIdx.x / tiles_n, tn = blockIdx.x % tiles_n;
  const int m0 = tm * BM, n0 = tn * BN;
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int nk = p.K / BK;
  unsigned long long* st = p.stamps + (size_t)blockIdx.x * p.nstamp * 8;
  if (wave >= NC) {
    const int lw = wave - NC;
    constexpr int CA = BM / 8 / NL, CB = BN / 8 / NL;
    unsigned aoff[CA], boff[CB];
#pragma unroll
    for (int i = 0; i < CA; ++i) {
      const int row = 8 * (lw * CA + i) + lane / 8;
      const int slot = ((lane % 8) ^ ((row >> 1) & 7)) << 3;
      aoff[i] = (unsigned)(((size_t)min(m0 + row, p.M - 1) * p.K + slot) * 2);
    }
#pragma unroll
    for (int i = 0; i < CB; ++i) {
      const int row = 8 * (lw * CB + i) + lane / 8;
      boff[i] = (unsigned)(((size_t)(n0 + row) * p.K + (((lane % 8) ^ ((row >> 1) & 7)) << 3)) * 2);
    }
    const auto rsA = __builtin_amdgcn_make_buffer_rsrc((void*)p.A, 0, (int)((size_t)p.M * p.K * 2), 0x00020000);
    const auto rsW = __builtin_amdgcn_make_buffer_rsrc((void*)p.W, 0, (int)((size_t)p.N * p.K * 2), 0x00020000);
    auto fill = [&](int kb, int stg) {
      unsigned char* Ad = As + stg * kStage;
      unsigned char* Bd = Bs + stg * kStage;
#pragma unroll
      for (int i = 0; i < CA; ++i)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsA, (lptr_t)(Ad + (lw * CA + i) * 1024), 16, (int)aoff[i], kb * BK * 2, 0, 0);
#pragma unroll
      for (int i = 0; i < CB; ++i)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsW, (lptr_t)(Bd + (lw * CB + i) * 1024), 16, (int)boff[i], kb * BK * 2, 0, 0);
    };
    for (int s = 0; s < NST - 1; ++s)
      if (s < nk) fill(s, s);
    int wst = NST - 1;
    for (int kb = 0; kb < nk; ++kb) {
      const int r = min(NST - 2, nk - 1 - kb);
      if (r >= 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(CA + CB) : "memory");
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      const unsigned long long t0 = __builtin_amdgcn_s_memtime();
      __builtin_amdgcn_s_barrier();
      const unsigned long long t1 = __builtin_amdgcn_s_memtime();
      if (kb + NST - 1 < nk) fill(kb + NST - 1, wst);
      wst = wst + 1 == NST ? 0 : wst + 1;
      const unsigned long long t2 = __builtin_amdgcn_s_memtime();
      if (lw == 0 && lane == 0 && kb < p.nstamp) { st[kb * 8 + 4] = t0; st[kb * 8 + 5] = t1; st[kb * 8 + 6] = t2; }
    }
    if (VAR == 1) { __syncthreads(); __syncthreads(); }
    return;
  }
  // MFMA waves
  const int grp = VAR == 1 ? wave >> 2 : 0, w4 = wave & 3;
  const int wm = w4 >> 1, wn = w4 & 1;
  f32x16 acc[2];
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[a][r] = 0.f;
  const int fr = lane & 31, fh = lane >> 5;
  int rst = 0;
  for (int kb = 0; kb < nk; ++kb) {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    __builtin_amdgcn_s_barrier();
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    const unsigned char* Ac = As + rst * kStage;
    const unsigned char* Bc = Bs + rst * kStage;
    rst = rst + 1 == NST ? 0 : rst + 1;
    if (VAR == 2) {
      bf16x8 af[4][2], bfr[4];
#pragma unroll
      for (int kk = 0; kk < 4; ++kk) {
#pragma unroll
        for (int a = 0; a < 2; ++a) af[kk][a] = *reinterpret_cast<const bf16x8*>(Ac + swz(wm * 64 + a * 32 + fr, 2 * kk + fh));
        bfr[kk] = *reinterpret_cast<const bf16x8*>(Bc + swz(wn * 32 + fr, 2 * kk + fh));
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int kk = 0; kk < 4; ++kk) {
#pragma unroll
        for (int a = 0; a < 2; ++a) acc[a] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[kk][a], bfr[kk], acc[a], 0, 0, 0);
      }
    } else if (VAR == 1) {
      bf16x8 af[2][2], bfr[2];
#pragma unroll
      for (int q = 0; q < 2; ++q) {
        const int kk = grp * 2 + q;
#pragma unroll
        for (int a = 0; a < 2; ++a) af[q][a] = *reinterpret_cast<const bf16x8*>(Ac + swz(wm * 64 + a * 32 + fr, 2 * kk + fh));
        bfr[q] = *reinterpret_cast<const bf16x8*>(Bc + swz(wn * 32 + fr, 2 * kk + fh));
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int q = 0; q < 2; ++q)
#pragma unroll
        for (int a = 0; a < 2; ++a) acc[a] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[q][a], bfr[q], acc[a], 0, 0, 0);
    } else {
      bf16x8 af[2][2], bfr[2];
      auto read_slice = [&](int set, int kk) {
#pragma unroll
        for (int a = 0; a < 2; ++a) af[set][a] = *reinterpret_cast<const bf16x8*>(Ac + swz(wm * 64 + a * 32 + fr, 2 * kk + fh));
        bfr[set] = *reinterpret_cast<const bf16x8*>(Bc + swz(wn * 32 + fr, 2 * kk + fh));
      };
      read_slice(0, 0);
#pragma unroll
      for (int kk = 0; kk < 4; ++kk) {
        if (kk + 1 < 4) read_slice((kk + 1) & 1, kk + 1);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int a = 0; a < 2; ++a) acc[a] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[kk & 1][a], bfr[kk & 1], acc[a], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    const unsigned long long t2 = __builtin_amdgcn_s_memtime();
    if (wave == 0 && lane == 0 && kb < p.nstamp) { st[kb * 8 + 0] = t0; st[kb * 8 + 1] = t1; st[kb * 8 + 2] = t2; }
  }
  if (VAR == 1) {          // combine the two K-slice groups through LDS (fp32 [128][64])
    __syncthreads();
    float* Cs = reinterpret_cast<float*>(smem);
    if (grp == 1) {
#pragma unroll
      for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int row = wm * 64 + a * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
          Cs[row * 68 + wn * 32 + (lane & 31)] = acc[a][r];
        }
    }
    __syncthreads();
    if (grp == 1) return;
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = wm * 64 + a * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
        acc[a][r] += Cs[row * 68 + wn * 32 + (lane & 31)];
      }
  }
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int row = m0 + wm * 64 + a * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
      if (row < p.M) p.Y[(size_t)row * p.N + n0 + wn * 32 + (lane & 31)] = acc[a][r];
    }
}

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

int main(int argc, char** argv) {
  const int var = argc > 1 ? atoi(argv[1]) : 0;
  const int M = argc > 2 ? atoi(argv[2]) : 3267, N = argc > 3 ? atoi(argv[3]) : 256, K = argc > 4 ? atoi(argv[4]) : 2304;
  std::vector<__hip_bfloat16> hA((size_t)M * K), hW((size_t)N * K);
  srand(1);
  for (auto& v : hA) v = __float2bfloat16((rand() % 2001 - 1000) / 1000.f);
  for (auto& v : hW) v = __float2bfloat16((rand() % 2001 - 1000) / 1000.f);
  __hip_bfloat16 *A, *W; float* Y; unsigned long long* S;
  const int grid = ((M + BM - 1) / BM) * (N / BN), nstamp = 40;
  CK(hipMalloc(&A, hA.size() * 2)); CK(hipMalloc(&W, hW.size() * 2)); CK(hipMalloc(&Y, (size_t)M * N * 4));
  CK(hipMalloc(&S, (size_t)grid * nstamp * 8 * 8));
  CK(hipMemcpy(A, hA.data(), hA.size() * 2, hipMemcpyHostToDevice)); CK(hipMemcpy(W, hW.data(), hW.size() * 2, hipMemcpyHostToDevice));
  CK(hipMemset(S, 0, (size_t)grid * nstamp * 8 * 8));
  P p{A, W, Y, M, N, K, S, nstamp};
  const int lds = NST * kStage;
  auto launch = [&]() {
    if (var == 1) { hipFuncSetAttribute((const void*)lw_kernel<1>, hipFuncAttributeMaxDynamicSharedMemorySize, lds); lw_kernel<1><<<grid, 768, lds>>>(p); }
    else if (var == 2) { hipFuncSetAttribute((const void*)lw_kernel<2>, hipFuncAttributeMaxDynamicSharedMemorySize, lds); lw_kernel<2><<<grid, 512, lds>>>(p); }
    else { hipFuncSetAttribute((const void*)lw_kernel<0>, hipFuncAttributeMaxDynamicSharedMemorySize, lds); lw_kernel<0><<<grid, 512, lds>>>(p); }
  };
  for (int i = 0; i < 5; ++i) launch();
  CK(hipDeviceSynchronize());
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipEventRecord(e0);
  for (int i = 0; i < 50; ++i) launch();
  hipEventRecord(e1);
  CK(hipDeviceSynchronize());
  float ms; hipEventElapsedTime(&ms, e0, e1);
  // check one output element
  std::vector<float> hY((size_t)M * N);
  CK(hipMemcpy(hY.data(), Y, hY.size() * 4, hipMemcpyDeviceToHost));
  double worst = 0;
  for (int t = 0; t < 200; ++t) {
    const int m = rand() % M, n = rand() % N;
    double ref = 0;
    for (int k = 0; k < K; ++k) ref += (double)__bfloat162float(hA[(size_t)m * K + k]) * __bfloat162float(hW[(size_t)n * K + k]);
    worst = std::max(worst, std::abs(ref - hY[(size_t)m * N + n]));
  }
  printf("variant %d M=%d N=%d K=%d grid %d: %.2f us per launch (back to back, incl. launch gaps), max abs err %.3g\n", var, M, N, K, grid,
         ms / 50 * 1e3, worst);
  std::vector<unsigned long long> hS((size_t)grid * nstamp * 8);
  CK(hipMemcpy(hS.data(), S, hS.size() * 8, hipMemcpyDeviceToHost));
  const int nk = K / BK, ns = std::min(nk, nstamp);
  // medians over workgroups of: MFMA wave: barrier wait (t1 - t0), body (t2 - t1), step period (t0[k+1] - t0[k]);
  // loader wave: wait-for-landing end -> barrier exit (l1 - l0), fill issue (l2 - l1), step period
  auto med = [&](std::vector<double>& v) { std::sort(v.begin(), v.end()); return v.empty() ? 0.0 : v[v.size() / 2]; };
  printf("step | mfma: barrier  body  period | loader: barrier  issue  period  (cycles, median over %d workgroups)\n", grid);
  for (int k = 0; k < ns; k += (k < 6 ? 1 : 6)) {
    std::vector<double> a, b, c, d, e, f;
    for (int g = 0; g < grid; ++g) {
      const unsigned long long* s = &hS[((size_t)g * nstamp + k) * 8];
      a.push_back((double)(s[1] - s[0])); b.push_back((double)(s[2] - s[1]));
      d.push_back((double)(s[5] - s[4])); e.push_back((double)(s[6] - s[5]));
      if (k + 1 < ns) { const unsigned long long* q = s + 8; c.push_back((double)(q[0] - s[0])); f.push_back((double)(q[4] - s[4])); }
    }
    printf("%4d | %8.0f %6.0f %7.0f | %8.0f %6.0f %7.0f\n", k, med(a), med(b), med(c), med(d), med(e), med(f));
  }
  std::vector<double> life;
  for (int g = 0; g < grid; ++g) life.push_back((double)(hS[((size_t)g * nstamp + ns - 1) * 8 + 2] - hS[(size_t)g * nstamp * 8]));
  printf("main loop (first barrier arrival -> last body end) median %.0f cycles over %d steps = %.0f per step\n", med(life), ns, med(life) / ns);
  return 0;
}
