// PixelConLossV2 streaming kernels, fp16 operands, PLANNED form: the same math as pixcon_loss_f16.hip (constant-shift
// sweep 1, see there), restructured around three measurements of that file's kernels on the benchmark step
// (profiles/r02_pixcon_sq.txt, tools/pixcon_pairs.py):
//
//  1. Rows are grouped by label, so a (128-anchor block x 32-row contrast tile) pair is almost always PURE: every row a
//     negative of every anchor, or every row a positive.  Sweep 2 already visited the same-label ranges only; sweep 1
//     visited everything although a pure-positive tile contributes nothing to it (E = 0).  With one dominant pseudo-label
//     (the usual case: 68 % of the pairs on the benchmark step) that was 2/3 of sweep 1 wasted.  Here a PLAN kernel
//     (one workgroup, device-side: the host never sees the label distribution) builds per anchor block the tile list of
//     each sweep - sweep 1 = everything except the tiles lying wholly inside the block's own label, sweep 2 = the
//     same-label ranges - cuts the lists into work units of CH tiles (CH chosen so that either sweep has ~512 units)
//     and writes the units in chunk-major order (all blocks' chunk 0, then chunk 1, ...: concurrently running units
//     stream the same contrast tiles, which therefore come out of each XCD's L2).  The sweeps are persistent kernels,
//     one workgroup per CU, that draw unit numbers from a counter; results go to the unit's own slot, so the order in
//     which CUs pick units does not change a bit of the result.
//     The row maximum m_i (the shift of the positives, loss.py:446) = max_j S_ij over ALL valid rows is S_ii for unit
//     rows (Cauchy-Schwarz); sweep 1 seeds its running maximum with S_ii computed from the anchor's own fp16 values and
//     still takes the maximum over every row it visits, so only a skipped positive row that beats S_ii by fp16 rounding
//     (an exact duplicate of the anchor's feature vector) is not seen: a change of m_i of the order of 1e-4 / T.
//  2. One wave per SIMD (the accumulators leave no room for two) and the compiler's schedule had nothing overlapping:
//     LDS burst -> 16 MFMA -> 150 VALU -> 16 MFMA -> commit, ~4000 cycles per tile against 1024 of MFMA.  Here a tile
//     step is a fixed sequence of 32 MFMA gaps, pinned with sched_barrier, each carrying its share of the other
//     streams:   gaps 0..15   V/U^T += C(t-1)^T . w(t-1)     | ds_read_b128 of the score fragments of tile t+1
//                gaps 16..31  x(t+1) = S^T of tile t+1       | ds_read_b64_tr of the value fragments of tile t
//                every gap    half of one element (of 16) of the epilogue of tile t  (VALU: mask, exp2 / log2 / rcp)
//     i.e. a three-stage pipeline over tiles (scores of t+1, epilogue of t, values of t-1).  The epilogue is branch-free
//     and general (labels compared per element: 16 label bytes per lane come with ONE broadcast ds_read_b128), so there
//     is a single code path for pure, mixed and padded tiles.
//  3. The tile's label range was reduced with a 10-step ds_bpermute chain on wave 0 while the other waves sat at the
//     barrier; no longer needed.
// Kept from pixcon_loss_f16.hip (and used for what this file does not cover: T < 0.06, more than 32 teacher classes,
// more than 1024 anchor blocks): the fixed-split kernels there.
#include "pixcon_f16_tiles.h"

namespace ucd {
namespace {

constexpr int kRing = 4;                       // LDS tile buffers
constexpr int kBufHalfs = kTJ * kPitchH;       // halfs per buffer
constexpr int kPlanThreads = 1024;             // = the largest number of anchor blocks the plan handles
constexpr int kMaxChunks = 256;                // units per anchor block and sweep
// units per sweep the chunk length aims at: two per CU.  Measured (tools/pixcon_bench.py f16 dom, B = 24 at 513^2): 256 units
// 1.95 ms, 384 1.63, 512 1.52, 1024 1.55, 2048 1.67 - fewer units balance worse, more units write and re-read more partials
constexpr int kTargetUnits = 512;
constexpr int kMinChunk = 8;

struct PlanHdr {
  int ctr1, ctr2;   // work counters of the two sweeps
  int U1, U2;       // units
  int CH1, CH2;     // tiles per unit
  int nblk, pad;
};

typedef unsigned int u32;

// ---- plan ------------------------------------------------------------------------------------------------
// seg1[b] = {a0, n0, a1, n1, a2, n2, 0, 0}: sweep-1 tiles of block b as three ranges (first tile, count)
// seg2[b] = {a0, n0, a1, n1}: sweep-2 tiles
// us1 / us2 [b]: first result slot of block b (slots of a block are consecutive: slot = us[b] + chunk)
// order1 / order2 [u] = block | chunk << 16 of work unit u, chunk-major
__device__ __forceinline__ int block_sum(int v, int* red) {   // 1024 threads
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
  __syncthreads();
  int t = 0;
  for (int w = 0; w < kPlanThreads / 64; ++w) t += red[w];
  return t;
}
__device__ __forceinline__ int block_max(int v, int* red) {
  for (int off = 32; off > 0; off >>= 1) v = max(v, __shfl_xor(v, off, 64));
  __syncthreads();
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
  __syncthreads();
  int t = 0;
  for (int w = 0; w < kPlanThreads / 64; ++w) t = max(t, red[w]);
  return t;
}
__device__ __forceinline__ int block_scan_excl(int v, int* red) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  int inc = v;
  for (int off = 1; off < 64; off <<= 1) {
    const int n = __shfl_up(inc, off, 64);
    if (lane >= off) inc += n;
  }
  __syncthreads();
  if (lane == 63) red[wave] = inc;
  __syncthreads();
  int base = 0;
  for (int w = 0; w < wave; ++w) base += red[w];
  return base + inc - v;
}

__global__ __launch_bounds__(kPlanThreads) void pixcon16p_plan_kernel(const uint8_t* __restrict__ row_label,
                                                                     const ucd_pixcon_meta* __restrict__ meta, int umax,
                                                                     PlanHdr* __restrict__ hdr, int* __restrict__ seg1,
                                                                     int* __restrict__ seg2, int* __restrict__ us1,
                                                                     int* __restrict__ us2, int* __restrict__ order1,
                                                                     int* __restrict__ order2) {
  __shared__ int red[kPlanThreads / 64];
  __shared__ int cnt_s[kPlanThreads];
  __shared__ int hist[kMaxChunks + 1];
  __shared__ int base_s[kMaxChunks + 1];
  __shared__ int ge_s[kMaxChunks + 1];
  const int A = meta->A, Apad = meta->Apad, Cpad = meta->Cpad;
  const int ntiles = Cpad / kTJ;
  const int nblk = (A + kBI - 1) / kBI;
  const int b = threadIdx.x;
  const bool active = b < nblk;
  int l1[6] = {0, 0, 0, 0, 0, 0}, l2[4] = {0, 0, 0, 0};
  if (active) {
    if (meta->sorted) {
      const int Lmin = row_label[b * kBI], Lmax = row_label[min(b * kBI + kBI, A) - 1];
      const int r1a = meta->label_start_a[Lmin], r1b = meta->label_start_a[Lmax + 1];
      const int r2a = Apad + meta->label_start_o[Lmin], r2b = Apad + meta->label_start_o[Lmax + 1];
      l2[0] = r1a / kTJ; l2[1] = (r1b + kTJ - 1) / kTJ - l2[0];
      l2[2] = r2a / kTJ; l2[3] = r2b > r2a ? (r2b + kTJ - 1) / kTJ - l2[2] : 0;
      int sa = 0, ea = 0, so = 0, eo = 0;   // skipped tile ranges [sa, ea) and [so, eo): wholly inside the block's only label
      if (Lmin == Lmax) {
        sa = (r1a + kTJ - 1) / kTJ; ea = max(sa, r1b / kTJ);
        so = (r2a + kTJ - 1) / kTJ; eo = max(so, r2b / kTJ);
      }
      l1[0] = 0; l1[1] = sa;
      l1[2] = ea; l1[3] = so - ea;
      l1[4] = eo; l1[5] = ntiles - eo;
    } else {
      l1[4] = 0; l1[5] = ntiles;
      l2[0] = 0; l2[1] = ntiles;
    }
  }
  const int n1 = l1[1] + l1[3] + l1[5], n2 = l2[1] + l2[3];
  const int T1 = block_sum(n1, red), T2 = block_sum(n2, red);
  const int nmax1 = block_max(n1, red), nmax2 = block_max(n2, red);
  const int CH1 = max(max(kMinChunk, (T1 + kTargetUnits - 1) / kTargetUnits), (nmax1 + kMaxChunks - 1) / kMaxChunks);
  const int CH2 = max(max(kMinChunk, (T2 + kTargetUnits - 1) / kTargetUnits), (nmax2 + kMaxChunks - 1) / kMaxChunks);
  const int c1 = (n1 + CH1 - 1) / CH1, c2 = (n2 + CH2 - 1) / CH2;
  const int o1 = block_scan_excl(c1, red), o2 = block_scan_excl(c2, red);
  const int U1 = block_sum(c1, red), U2 = block_sum(c2, red);
  if (active) {
    for (int q = 0; q < 6; ++q) seg1[b * 8 + q] = l1[q];
    for (int q = 0; q < 4; ++q) seg2[b * 4 + q] = l2[q];
  }
  if (b <= nblk) {   // one past the end closes the last block's range (thread nblk holds c = 0, offset = total)
    us1[b] = o1;
    us2[b] = o2;
  }
  if (threadIdx.x == 0) {
    hdr->ctr1 = 0; hdr->ctr2 = 0;
    hdr->U1 = min(U1, umax); hdr->U2 = min(U2, umax);   // U <= target + nblk <= umax by construction
    hdr->CH1 = CH1; hdr->CH2 = CH2;
    hdr->nblk = nblk; hdr->pad = 0;
  }
  // chunk-major order tables: entries (c, b) with c < cnt[b], sorted by (c, b)
  for (int sweep = 0; sweep < 2; ++sweep) {
    const int cnt = sweep == 0 ? c1 : c2;
    int* __restrict__ order = sweep == 0 ? order1 : order2;
    __syncthreads();
    cnt_s[b] = cnt;
    if (b <= kMaxChunks) hist[b] = 0;
    __syncthreads();
    if (cnt > 0) atomicAdd(&hist[min(cnt, kMaxChunks)], 1);
    __syncthreads();
    if (b <= kMaxChunks) {   // ge_s[c] = #{blocks: cnt > c}
      int n = 0;
      for (int k = b + 1; k <= kMaxChunks; ++k) n += hist[k];
      ge_s[b] = n;
    }
    __syncthreads();
    if (b <= kMaxChunks) {   // base_s[c] = number of entries with a smaller chunk index
      int base = 0;
      for (int cp = 0; cp < b; ++cp) base += ge_s[cp];
      base_s[b] = base;
    }
    __syncthreads();
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int maxc = block_max(cnt, red);
    for (int c = wave; c < maxc; c += kPlanThreads / 64) {
      int run = base_s[c];
      for (int b0 = 0; b0 < nblk; b0 += 64) {
        const int bb = b0 + lane;
        const bool flag = bb < nblk && cnt_s[bb] > c;
        const unsigned long long m = __ballot(flag);
        if (flag) {
          const int idx = run + __popcll(m & ((1ull << lane) - 1ull));
          if (idx < umax) order[idx] = bb | (c << 16);
        }
        run += __popcll(m);
      }
    }
  }
}

// ---- the tile step ----------------------------------------------------------------------------------------
__device__ __forceinline__ f16x8 lds_b128(const _Float16* p) { return *reinterpret_cast<const f16x8*>(p); }
__device__ __forceinline__ h4 lds_tr(const _Float16* p) {
  return __builtin_amdgcn_ds_read_tr16_b64_v4f16((__attribute__((address_space(3))) h4*)p);
}
__device__ __forceinline__ f16x8 join(const h4& lo, const h4& hi) {
  f16x8 a;
  a[0] = (_Float16)lo[0]; a[1] = (_Float16)lo[1]; a[2] = (_Float16)lo[2]; a[3] = (_Float16)lo[3];
  a[4] = (_Float16)hi[0]; a[5] = (_Float16)hi[1]; a[6] = (_Float16)hi[2]; a[7] = (_Float16)hi[3];
  return a;
}
__device__ __forceinline__ u32 label_byte(const u32x4& lw, int e) { return (lw[e >> 2] >> (8 * (e & 3))) & 0xffu; }

struct StageP {
  u32x4 a, b, c, d;
  u32x4 side;
  u32 lab4;
};
__device__ __forceinline__ void tile_fetch_p(StageP& st, const _Float16* ch16, const uint8_t* row_label, int j0) {
  st.lab4 = *reinterpret_cast<const u32*>(row_label + j0 + 4 * (threadIdx.x & 7));
  const int row = threadIdx.x >> 5, c = threadIdx.x & 31;
  const _Float16* p0 = ch16 + (size_t)(j0 + row) * kN + c * 8;
  st.a = *reinterpret_cast<const u32x4*>(p0);
  st.b = *reinterpret_cast<const u32x4*>(p0 + 8 * kN);
  st.c = *reinterpret_cast<const u32x4*>(p0 + 16 * kN);
  st.d = *reinterpret_cast<const u32x4*>(p0 + 24 * kN);
}
// labels of a tile in LDS: 8 dwords, dword (4 half + g) = label bytes of rows 8 g + 4 half + 0..3, so a lane's 16 rows
// (tile_row(reg, half), reg = 4 g + r) are the 16 bytes at 16 half
__device__ __forceinline__ void tile_commit_p(const StageP& st, _Float16* __restrict__ cs, u32* __restrict__ labs) {
  const int row = threadIdx.x >> 5, c = threadIdx.x & 31;
  _Float16* p = cs + row * kPitchH + c * 8;
  *reinterpret_cast<u32x4*>(p) = st.a;
  *reinterpret_cast<u32x4*>(p + 8 * kPitchH) = st.b;
  *reinterpret_cast<u32x4*>(p + 16 * kPitchH) = st.c;
  *reinterpret_cast<u32x4*>(p + 24 * kPitchH) = st.d;
  if (threadIdx.x < 8) labs[(threadIdx.x & 1) * 4 + (threadIdx.x >> 1)] = st.lab4;
}

struct Frags {
  f16x8 sf[16];            // score fragments (A operand of S^T), one per 16-wide k step
  h4 vlo[16], vhi[16];     // value fragments, pair j = 2 nt + s
};

__device__ __forceinline__ void load_all_value_frags(Frags& f, const _Float16* cs, int vbase) {
#pragma unroll
  for (int j = 0; j < 16; ++j) {
    const _Float16* a0 = cs + vbase + (16 * (j & 1)) * kPitchH + 32 * (j >> 1);
    f.vlo[j] = lds_tr(a0);
    f.vhi[j] = lds_tr(a0 + 8 * kPitchH);
  }
}

// MODE 0: sweep 1 (negatives)   w = E = exp2(s2 - m_run) on the negatives           acc = U^T
// MODE 1: sweep 2 (positives)   w = q = pos P neg / (exp2(s2 - m2) + neg)          acc = V^T
struct RowState {
  // per lane = per anchor
  int la;            // anchor label (-1 for a row past A: never equal to a label byte)
  float k2;
  // sweep 1
  float m_run, neg, mx;
  // sweep 2
  float m2, neg_true, lossacc, qsum;
  int self0;         // i_row - 4 * half: element e of the tile at j0 is the self pair when tile_row(e, 0) == self0 - j0
  bool gt_i;
};

struct TileSrc {   // what a fetch needs
  const _Float16* ch16;
  const uint8_t* row_label;
  const _Float16* p16;
  int KP16, s_row, s_c, chunks, ppitch;
};
template <bool PROB>
__device__ __forceinline__ void fetch_tile(StageP& st, const TileSrc& src, int j0) {
  tile_fetch_p(st, src.ch16, src.row_label, j0);
  if (PROB) st.side = *reinterpret_cast<const u32x4*>(src.p16 + (size_t)(j0 + src.s_row) * 2 * src.KP16 + src.s_c * 8);
  fetch_fence();
}

// One tile step = 32 MFMA gaps around ONE barrier in the middle (t = this tile, ring buffer BUF = t mod 4):
//   gaps 0..15   acc += C(t-1)^T . w(t-1)   | score fragments of tile t+1 (buffer BUF+1)   | epilogue elements 0..7 of tile t
//                gaps 11..15 also commit tile t+2 (fetched during the previous step) to buffer BUF+2
//   barrier; fetch tile t+3 into the stage registers (a second set, i.e. two steps of latency, measured slower)
//   gaps 16..31  x(t+1) = S^T of tile t+1    | value fragments of tile t (buffer BUF)         | epilogue elements 8..15
// Between two barriers the workgroup reads buffers BUF-1 (previous step's second half), BUF+1 and writes BUF+2; the
// value-fragment reads of the second half stay in flight across the end of the step (nothing waits for them there).
template <int MODE, bool PROB, int BUF>
__device__ __forceinline__ void tile_step(f32x16 (&acc)[8], const f16x8 (&a16)[16], Frags& f, const f32x16& x_cur,
                                          f32x16& x_next, const f16x8 (&w_prev)[2], f16x8 (&w_new)[2], const u32x4& lw_in,
                                          u32x4& lw_next, RowState& rs, const f32x16& pm, _Float16* cs0, u32* labs0,
                                          _Float16* ps0, int sbase, int vbase, int half, bool self_tile, const u32x4& self_patch,
                                          StageP& stage, const TileSrc& src, int j0_fetch) {
  constexpr int b_cur = BUF, b_nxt = (BUF + 1) % kRing, b_new = (BUF + 2) % kRing;
  const _Float16* cs_cur = cs0 + b_cur * kBufHalfs;
  const _Float16* cs_nxt = cs0 + b_nxt * kBufHalfs;
  _Float16* cs_new = cs0 + b_new * kBufHalfs;
  const f32x16 zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  u32x4 lw_cur = lw_in;
  if (MODE == 1 && self_tile) {   // the tile that holds this wave's own anchors (wave-uniform): the self pair is no positive
    lw_cur[0] |= self_patch[0]; lw_cur[1] |= self_patch[1]; lw_cur[2] |= self_patch[2]; lw_cur[3] |= self_patch[3];
  }
  if (MODE == 0 && self_tile) {
    // sweep 1 passes "the tile holds padding rows" here (the last tiles of either segment, wave-uniform): their label
    // bytes (255) become the lane's own label, so the one compare of the epilogue masks them like positives.  (Subtracting
    // their known contribution afterwards leaves a rounding residue in neg, and neg must be EXACTLY zero for an anchor
    // without negatives: the positives' terms S' - log(exp S' + neg) cancel only then.)
    const u32 la8 = (u32)rs.la & 0xffu;
#pragma unroll
    for (int wd = 0; wd < 4; ++wd)
#pragma unroll
      for (int k = 0; k < 4; ++k)
        if (((lw_cur[wd] >> (8 * k)) & 0xffu) == (u32)kPadLabel)
          lw_cur[wd] = (lw_cur[wd] & ~(0xffu << (8 * k))) | (la8 << (8 * k));
  }
  const int crow = threadIdx.x >> 5, ccol = threadIdx.x & 31;
  _Float16* cdst = cs_new + crow * kPitchH + ccol * 8;
  float arg[16], wv[16], dv[16];
#pragma unroll
  for (int g = 0; g < 32; ++g) {
    if (g < 16) {
      acc[g >> 1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(join(f.vlo[g], f.vhi[g]), w_prev[g & 1], acc[g >> 1], 0, 0, 0);
      f.sf[g] = lds_b128(cs_nxt + sbase + 16 * g);
      if (g == 11) *reinterpret_cast<u32x4*>(cdst) = stage.a;
      if (g == 12) *reinterpret_cast<u32x4*>(cdst + 8 * kPitchH) = stage.b;
      if (g == 13) *reinterpret_cast<u32x4*>(cdst + 16 * kPitchH) = stage.c;
      if (g == 14) *reinterpret_cast<u32x4*>(cdst + 24 * kPitchH) = stage.d;
      if (g == 15) {
        if (threadIdx.x < 8) labs0[b_new * 8 + (threadIdx.x & 1) * 4 + (threadIdx.x >> 1)] = stage.lab4;
        if (PROB && (int)threadIdx.x < kTJ * src.chunks)
          *reinterpret_cast<u32x4*>(ps0 + (b_new * kTJ + src.s_row) * src.ppitch + src.s_c * 8) = stage.side;
      }
    } else {
      const int k = g - 16;
      if (k == 0) {
        __syncthreads();
        fetch_tile<PROB>(stage, src, j0_fetch);
        lw_next = *reinterpret_cast<const u32x4*>(labs0 + b_nxt * 8 + 4 * half);
      }
      x_next = __builtin_amdgcn_mfma_f32_32x32x16_f16(f.sf[k], a16[k], k == 0 ? zero : x_next, 0, 0, 0);
      const _Float16* a0 = cs_cur + vbase + (16 * (k & 1)) * kPitchH + 32 * (k >> 1);
      f.vlo[k] = lds_tr(a0);
      f.vhi[k] = lds_tr(a0 + 8 * kPitchH);
    }
    // The epilogue values are pure arithmetic: instruction selection would place them wherever register pressure is
    // lowest (all at the end of the step).  An empty volatile asm that takes the value in and hands it back is ordered
    // with the sched_barriers, so the arithmetic feeding it lands in this gap.
    const int e = g >> 1;
    if ((g & 1) == 0) {
      if (MODE == 0) {
        // v_max_f32 directly: fmaxf() canonicalises both operands first (two more instructions per element)
        asm("v_max_f32 %0, %1, %2" : "=v"(rs.mx) : "v"(rs.mx), "v"(x_cur[e]));
        arg[e] = __builtin_amdgcn_exp2f(__builtin_fmaf(x_cur[e], rs.k2, -rs.m_run));
        asm volatile("" : "+v"(arg[e]), "+v"(rs.mx));
      } else {
        // with s' = s2 - m2 - log2(neg):  q = neg / (2^(s2-m2) + neg) = 1 / (1 + 2^s'),
        //                                 (s2 - m2) - log2(2^(s2-m2) + neg) = s' - log2(1 + 2^s')
        arg[e] = __builtin_fmaf(x_cur[e], rs.k2, -rs.m2);
        dv[e] = __builtin_amdgcn_exp2f(arg[e]) + 1.f;
        wv[e] = __builtin_amdgcn_rcpf(dv[e]);
        dv[e] = __builtin_amdgcn_logf(dv[e]);
        asm volatile("" : "+v"(arg[e]), "+v"(dv[e]), "+v"(wv[e]));
      }
    } else {
      const u32 lc = label_byte(lw_cur, e);
      if (MODE == 0) {
        const float ev = ((int)lc != rs.la) ? arg[e] : 0.f;
        rs.neg += ev;
        wv[e] = ev;
        asm volatile("" : "+v"(wv[e]), "+v"(rs.neg));
      } else {
        float mf = ((int)lc == rs.la) ? 1.f : 0.f;
        if (PROB) mf = rs.gt_i ? mf : mf * pm[e];
        rs.lossacc = __builtin_fmaf(mf, arg[e] - dv[e], rs.lossacc);    // log2 units; scaled by ln 2 once per row
        wv[e] *= mf;
        rs.qsum += wv[e];
        asm volatile("" : "+v"(wv[e]), "+v"(rs.lossacc), "+v"(rs.qsum));
      }
      if ((e & 7) == 7) {
#pragma unroll
        for (int jj = 0; jj < 8; ++jj) w_new[e >> 3][jj] = (_Float16)wv[8 * (e >> 3) + jj];
        asm volatile("" : "+v"(w_new[e >> 3]));
      }
    }
    __builtin_amdgcn_sched_barrier(0);
  }
}

// P^T tile: pm[j][i] = sum_k pc[j][k] pa[i][k]  (hi/lo split: 3 MFMAs per 16 classes), classes <= 32
__device__ __forceinline__ f32x16 prob_tile_p(const _Float16* ps, int ppitch, int KP16, int nk, const f16x8 (&pah)[2],
                                              const f16x8 (&pal)[2], int lane) {
  f32x16 pm = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  const _Float16* pc = ps + (lane & 31) * ppitch + 8 * (lane >> 5);
#pragma unroll
  for (int kk = 0; kk < 2; ++kk) {
    if (kk < nk) {
      const f16x8 ch = *reinterpret_cast<const f16x8*>(pc + 16 * kk);
      const f16x8 cl = *reinterpret_cast<const f16x8*>(pc + KP16 + 16 * kk);
      pm = __builtin_amdgcn_mfma_f32_32x32x16_f16(ch, pah[kk], pm, 0, 0, 0);
      pm = __builtin_amdgcn_mfma_f32_32x32x16_f16(ch, pal[kk], pm, 0, 0, 0);
      pm = __builtin_amdgcn_mfma_f32_32x32x16_f16(cl, pah[kk], pm, 0, 0, 0);
    }
  }
  return pm;
}

struct UnitList {   // up to three tile ranges addressed as one list
  int a0, n0, a1, n1, a2, n2;
  __device__ __forceinline__ int at(int v) const { return v < n0 ? a0 + v : (v < n0 + n1 ? a1 + (v - n0) : a2 + (v - n0 - n1)); }
};

// One persistent workgroup per CU; MODE as in tile_step; HAS_PROB: the call weights positives with the teacher's joint
// probabilities - a unit then runs the PROB instance of its body only when one of its 128 anchors is an old-class anchor.  negp / mxp / lossp / qsump: [slot][128]; Up / Vp:
// [slot][128][256].
template <int MODE, bool HAS_PROB>
__global__ __launch_bounds__(kThreads, 1) void pixcon16p_sweep_kernel(
    const _Float16* ch16, const uint8_t* row_label, const _Float16* p16, int KP16, const ucd_pixcon_meta* __restrict__ meta,
    PlanHdr* hdr, const int* __restrict__ seg, const int* __restrict__ us, const int* __restrict__ order,
    const int* __restrict__ us1, float k2, int shift_pos, const float* __restrict__ negp, const float* __restrict__ mxp,
    float* __restrict__ out_a, float* __restrict__ out_b, float* __restrict__ out_acc) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  _Float16* cs0 = reinterpret_cast<_Float16*>(smem_raw);                              // [kRing][32][kPitchH]
  u32* labs0 = reinterpret_cast<u32*>(smem_raw + kRing * kBufHalfs * 2);              // [kRing][8]
  int* s_unit = reinterpret_cast<int*>(labs0 + kRing * 8);                            // [4]
  const int ppitch = 2 * KP16 + 8;
  _Float16* ps0 = reinterpret_cast<_Float16*>(s_unit + 4);                            // [kRing][32][ppitch] (PROB)
  const int A = meta->A, min_new = meta->min_new, Apad = meta->Apad, Cend = meta->Apad + meta->Co;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, half = lane >> 5;
  const int sbase = (lane & 31) * kPitchH + 8 * half;
  const int vbase = (4 * half + ((lane & 15) >> 2)) * kPitchH + 16 * ((lane >> 4) & 1) + 4 * (lane & 3);
  const int nk = KP16 / 16;
  // probability rows of a tile: one 16-byte piece per thread (KP16 <= 32: at most 8 pieces per row)
  const int chunks = HAS_PROB ? (2 * KP16) / 8 : 1;
  const int s_idx = min((int)threadIdx.x, kTJ * chunks - 1);
  const int s_row = s_idx / chunks, s_c = s_idx - s_row * chunks;

  for (;;) {
    if (threadIdx.x == 0) s_unit[0] = atomicAdd(MODE == 0 ? &hdr->ctr1 : &hdr->ctr2, 1);
    __syncthreads();
    const int u = __builtin_amdgcn_readfirstlane(s_unit[0]);
    if (u >= (MODE == 0 ? hdr->U1 : hdr->U2)) break;
    const int ent = order[u];
    const int b = ent & 0xffff, c = ent >> 16;
    const int slot = us[b] + c;
    UnitList ul;
    if (MODE == 0) {
      ul.a0 = seg[b * 8 + 0]; ul.n0 = seg[b * 8 + 1]; ul.a1 = seg[b * 8 + 2]; ul.n1 = seg[b * 8 + 3];
      ul.a2 = seg[b * 8 + 4]; ul.n2 = seg[b * 8 + 5];
    } else {
      ul.a0 = seg[b * 4 + 0]; ul.n0 = seg[b * 4 + 1]; ul.a1 = seg[b * 4 + 2]; ul.n1 = seg[b * 4 + 3];
      ul.a2 = 0; ul.n2 = 0;
    }
    const int CH = MODE == 0 ? hdr->CH1 : hdr->CH2;
    const int v_begin = c * CH, v_end = min(ul.n0 + ul.n1 + ul.n2, v_begin + CH);
    const int nt = v_end - v_begin;   // >= 1

    const int i_row = b * kBI + wave * kTI + (lane & 31);
    const bool row_ok = i_row < A;
    const int unit_prob = HAS_PROB ? __syncthreads_or(row_ok && (int)row_label[row_ok ? i_row : 0] < min_new) : 0;
    auto run_unit = [&](auto prob_tag) __attribute__((always_inline)) {
    constexpr bool PROB = decltype(prob_tag)::value;
    RowState rs;
    rs.la = row_ok ? (int)row_label[i_row] : -1;
    rs.k2 = k2;
    rs.m_run = k2 - 14.5f;
    rs.neg = 0.f;
    rs.lossacc = 0.f; rs.qsum = 0.f;
    rs.self0 = i_row - 4 * half;
    rs.gt_i = rs.la >= min_new;
    f16x8 a16[16];
    load_anchor_frags(a16, ch16, i_row, row_ok, half);
    {  // S_ii from the anchor's own fp16 values (fp32 accumulation): seeds the row maximum
      float sii = 0.f;
#pragma unroll
      for (int ks = 0; ks < 16; ++ks)
#pragma unroll
        for (int j = 0; j < 8; ++j) sii = __builtin_fmaf((float)a16[ks][j], (float)a16[ks][j], sii);
      sii += __shfl_xor(sii, 32, 64);
      rs.mx = sii;
    }
    rs.m2 = 0.f; rs.neg_true = 0.f;
    if (MODE == 1) {
      float negs = 0.f, mx = rs.mx;
      if (row_ok) {
        const int ua = us1[b], ub = us1[b + 1];
        for (int s = ua; s < ub; ++s) {
          negs += negp[(size_t)s * kBI + (i_row - b * kBI)];
          mx = fmaxf(mx, mxp[(size_t)s * kBI + (i_row - b * kBI)]);
        }
      }
      // the positives' shift with log2(neg) folded in (neg = negs 2^m_run; rows without negatives: 2^-100 stands in
      // for zero, q rounds to 0 and the log terms cancel to fp32 resolution)
      rs.neg_true = negs * exp2f(rs.m_run);
      rs.m2 = (shift_pos ? mx * k2 : 0.f) + fmaxf(__log2f(negs) + rs.m_run, -100.f);
    }
    f16x8 pah[2], pal[2];
    // positives share the anchor's label, so the pair weight is 1 for an anchor of a new class and p_i . p_j otherwise
    // (loss.py:454-458): a wave whose anchors are all new-class skips the probability products
    const bool need_prob = PROB && __any(row_ok && !rs.gt_i);
    if (PROB) {
      const _Float16* pa = p16 + (size_t)(row_ok ? i_row : 0) * 2 * KP16 + 8 * half;
#pragma unroll
      for (int kk = 0; kk < 2; ++kk) {
        if (kk < nk) {
          pah[kk] = *reinterpret_cast<const f16x8*>(pa + 16 * kk);
          pal[kk] = *reinterpret_cast<const f16x8*>(pa + KP16 + 16 * kk);
        } else {
#pragma unroll
          for (int q = 0; q < 8; ++q) { pah[kk][q] = (_Float16)0.f; pal[kk][q] = (_Float16)0.f; }
        }
      }
    }
    f32x16 acc[8];
#pragma unroll
    for (int q = 0; q < 8; ++q)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[q][r] = 0.f;

    StageP stage;
    TileSrc src;
    src.ch16 = ch16; src.row_label = row_label; src.p16 = p16; src.KP16 = KP16; src.s_row = s_row; src.s_c = s_c;
    src.chunks = chunks; src.ppitch = ppitch;
    auto j0_of = [&](int v) { return ul.at(min(v, v_end - 1)) * kTJ; };
    auto commit = [&](int buf) {
      tile_commit_p(stage, cs0 + buf * kBufHalfs, labs0 + buf * 8);
      if (PROB && (int)threadIdx.x < kTJ * chunks)
        *reinterpret_cast<u32x4*>(ps0 + (buf * kTJ + s_row) * ppitch + s_c * 8) = stage.side;
    };
    fetch_tile<PROB>(stage, src, j0_of(v_begin)); commit(0);
    fetch_tile<PROB>(stage, src, j0_of(v_begin + 1)); commit(1);
    fetch_tile<PROB>(stage, src, j0_of(v_begin + 2));     // committed by step 0
    __syncthreads();

    // the self pair: row lane&31 of the tile that starts at this wave's first anchor; it is one of this lane's 16 elements
    // when bit 2 of the row equals the lane's half: its label byte is overwritten with 255 there (never an anchor label)
    u32x4 self_patch = {0u, 0u, 0u, 0u};
    const int i0w = b * kBI + wave * kTI;
    if (MODE == 1) {
      const int r = lane & 31;
      if (((r >> 2) & 1) == half) {
        const u32 m = 0xffu << (8 * (r & 3));
        self_patch[0] = (r >> 3) == 0 ? m : 0u; self_patch[1] = (r >> 3) == 1 ? m : 0u;
        self_patch[2] = (r >> 3) == 2 ? m : 0u; self_patch[3] = (r >> 3) == 3 ? m : 0u;
      }
    }

    Frags f;
    f32x16 xa, xb;
    f16x8 wa[2], wb[2];
    u32x4 lwa, lwb;
    f32x16 pm = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    {  // prologue: scores of tile 0; the value fragments of tile 0 stand in for "tile -1" with w = 0
      ScoreFrags s0;
      load_score_frags(s0, cs0, lane);
      load_all_value_frags(f, cs0, vbase);
      lwa = *reinterpret_cast<const u32x4*>(labs0 + 4 * half);
      __builtin_amdgcn_sched_barrier(0);
      xa = mfma_scores(s0, a16);
#pragma unroll
      for (int s = 0; s < 2; ++s)
#pragma unroll
        for (int q = 0; q < 8; ++q) { wa[s][q] = (_Float16)0.f; wb[s][q] = (_Float16)0.f; }
    }
    auto step = [&](auto bt, f32x16& x_cur, f32x16& x_next, f16x8 (&w_prev)[2], f16x8 (&w_new)[2], u32x4& lw_cur,
                    u32x4& lw_next, int t) {
      constexpr int B = decltype(bt)::value;
      const int j0 = ul.at(v_begin + t) * kTJ;
      if (PROB && need_prob) pm = prob_tile_p(ps0 + B * kTJ * ppitch, ppitch, KP16, nk, pah, pal, lane);
      tile_step<MODE, PROB, B>(acc, a16, f, x_cur, x_next, w_prev, w_new, lw_cur, lw_next, rs, pm, cs0, labs0, ps0, sbase, vbase,
                               half, MODE == 1 ? j0 == i0w : ((j0 + kTJ > A && j0 < Apad) || j0 + kTJ > Cend), self_patch, stage,
                               src, j0_of(v_begin + t + 3));
#pragma unroll
      for (int q = 0; q < 8; ++q) asm volatile("" : "+a"(acc[q]));   // accumulators stay in the AGPR half
      asm volatile("" : "+v"(x_next));   // the scores are read by VALU instructions next step: keep them out of the AGPRs
    };
    for (int t = 0; t < nt; t += 4) {
      step(std::integral_constant<int, 0>{}, xa, xb, wa, wb, lwa, lwb, t);
      if (t + 1 < nt) step(std::integral_constant<int, 1>{}, xb, xa, wb, wa, lwb, lwa, t + 1);
      if (t + 2 < nt) step(std::integral_constant<int, 2>{}, xa, xb, wa, wb, lwa, lwb, t + 2);
      if (t + 3 < nt) step(std::integral_constant<int, 3>{}, xb, xa, wb, wa, lwb, lwa, t + 3);
    }
    // drain: values of the last tile (its fragments were read during the last step)
    {
      const bool odd = nt & 1;   // after an odd number of steps the newest weights are in wb
#pragma unroll
      for (int j = 0; j < 16; ++j)
        acc[j >> 1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(join(f.vlo[j], f.vhi[j]), odd ? wb[j & 1] : wa[j & 1],
                                                              acc[j >> 1], 0, 0, 0);
    }
    if (MODE == 0) {
      rs.neg += __shfl_xor(rs.neg, 32, 64);
      rs.mx = fmaxf(rs.mx, __shfl_xor(rs.mx, 32, 64));
    } else {
      rs.lossacc += __shfl_xor(rs.lossacc, 32, 64);
      rs.qsum += __shfl_xor(rs.qsum, 32, 64);
    }
    if (row_ok) {
      const size_t o = (size_t)slot * kBI + (i_row - b * kBI);
      if (half == 0) {
        out_a[o] = MODE == 0 ? rs.neg : rs.lossacc * kLn2;   // neg in units of 2^m_run
        out_b[o] = MODE == 0 ? rs.mx : rs.qsum;              // mx: raw cosine units (times k2 = log2 domain)
      }
      store_values(acc, out_acc + o * kN, half);
    }
    };
    if (HAS_PROB && unit_prob) run_unit(std::true_type{});
    else run_unit(std::false_type{});
    __syncthreads();   // the ring and s_unit are reused by the next unit
  }
}

__global__ __launch_bounds__(kThreads) void pixcon16p_finalize_kernel(
    const uint8_t* __restrict__ row_label, const ucd_pixcon_meta* __restrict__ meta, float inv_T, float m_run,
    const int* __restrict__ us1, const int* __restrict__ us2, const float* __restrict__ negp, const float* __restrict__ lossp,
    const float* __restrict__ qsump, const float* __restrict__ Up, const float* __restrict__ Vp, float* __restrict__ grad_a,
    int ldg, float* __restrict__ row_stats, int maxA, float* __restrict__ row_loss) {
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int i = blockIdx.x * (kThreads / 64) + wave;
  const int A = meta->A;
  if (i >= A) return;
  const int b = i / kBI, il = i - b * kBI;
  const int u1a = us1[b], u1b = us1[b + 1], u2a = us2[b], u2b = us2[b + 1];
  const int num = meta->label_count_c[row_label[i]] - 1;
  const float R = (float)meta->n_valid;
  float negs = 0.f, la = 0.f, qs = 0.f;   // negs in units of 2^m_run
  for (int s = u1a; s < u1b; ++s) negs += negp[(size_t)s * kBI + il];
  for (int s = u2a; s < u2b; ++s) {
    la += lossp[(size_t)s * kBI + il];
    qs += qsump[(size_t)s * kBI + il];
  }
  const float coef = num > 0 ? inv_T / ((float)num * R) : 0.f;
  const float ratio = negs > 0.f ? qs / negs : 0.f;   // U is in the same 2^m_run units: the scale cancels
  const float rl = num > 0 ? -la / (float)num : 0.f;
  if (grad_a) {
    for (int c = lane * 4; c < ldg; c += 256) {
      float4 uu = {0.f, 0.f, 0.f, 0.f}, vv = {0.f, 0.f, 0.f, 0.f};
      if (c < kN) {
        for (int s = u1a; s < u1b; ++s) {
          const float4 t = *reinterpret_cast<const float4*>(Up + ((size_t)s * kBI + il) * kN + c);
          uu.x += t.x; uu.y += t.y; uu.z += t.z; uu.w += t.w;
        }
        for (int s = u2a; s < u2b; ++s) {
          const float4 t = *reinterpret_cast<const float4*>(Vp + ((size_t)s * kBI + il) * kN + c);
          vv.x += t.x; vv.y += t.y; vv.z += t.z; vv.w += t.w;
        }
      }
      float4 g = {coef * (ratio * uu.x - vv.x), coef * (ratio * uu.y - vv.y), coef * (ratio * uu.z - vv.z),
                  coef * (ratio * uu.w - vv.w)};
      *reinterpret_cast<float4*>(grad_a + (size_t)i * ldg + c) = g;
    }
  }
  if (lane == 0) {
    row_loss[i] = rl;
    if (row_stats) {
      row_stats[i] = negs > 0.f ? negs * exp2f(m_run) : 0.f;
      row_stats[(size_t)maxA + i] = (float)num;
      row_stats[(size_t)2 * maxA + i] = rl;
    }
  }
}

struct LayoutP {
  int nbmax, umax;
  size_t off_hdr, off_seg1, off_seg2, off_us1, off_us2, off_order1, off_order2, off_negp, off_mxp, off_lossp, off_qsump,
      off_rowloss, off_Up, off_Vp, total;
};
LayoutP make_layout_p(int BHW) {
  LayoutP L;
  L.nbmax = ceil_div(BHW, kBI);
  // units of a sweep <= sum_b ceil(n_b / CH) <= T / CH + blocks with CH >= max(kMinChunk, T / kTargetUnits), T <= blocks x tiles
  const long long tiles_max = (2ll * BHW + 2 * kPixTile) / kTJ + 1;
  const long long by_chunk = ((long long)L.nbmax * tiles_max + kMinChunk - 1) / kMinChunk;
  L.umax = (int)(by_chunk < kTargetUnits ? by_chunk : kTargetUnits) + L.nbmax + 1;
  size_t o = 0;
  auto take = [&](size_t bytes) { const size_t at = o; o += align_up(bytes, 256); return at; };
  L.off_hdr = take(sizeof(PlanHdr));
  L.off_seg1 = take((size_t)L.nbmax * 8 * 4);
  L.off_seg2 = take((size_t)L.nbmax * 4 * 4);
  L.off_us1 = take((size_t)(L.nbmax + 1) * 4);
  L.off_us2 = take((size_t)(L.nbmax + 1) * 4);
  L.off_order1 = take((size_t)L.umax * 4);
  L.off_order2 = take((size_t)L.umax * 4);
  L.off_negp = take((size_t)L.umax * kBI * 4);
  L.off_mxp = take((size_t)L.umax * kBI * 4);
  L.off_lossp = take((size_t)L.umax * kBI * 4);
  L.off_qsump = take((size_t)L.umax * kBI * 4);
  L.off_rowloss = take((size_t)BHW * 4);
  L.off_Up = take((size_t)L.umax * kBI * kN * 4);
  L.off_Vp = take((size_t)L.umax * kBI * kN * 4);
  L.total = o;
  return L;
}

}  // namespace

bool pixcon16p_eligible(int BHW, float temperature, int use_prob, int K) {
  return kLog2e / temperature <= kFixedShiftMaxK2 && ceil_div(BHW, kBI) < kPlanThreads && (!use_prob || K <= 32);
}

size_t pixcon16p_workspace_bytes(int BHW) { return ceil_div(BHW, kBI) < kPlanThreads ? make_layout_p(BHW).total : 0; }

int pixcon16p_launch(const _Float16* ch16, const uint8_t* row_label, const _Float16* p16, int K,
                     const ucd_pixcon_meta* meta, int BHW, float temperature, int shift_pos, int use_prob,
                     float* loss_out, float* grad_a, int ldg, float* row_stats, void* workspace, size_t workspace_bytes,
                     hipStream_t s) {
  static const char* fn = "ucd_pixcon_loss[f16]";
  const LayoutP L = make_layout_p(BHW);
  UCD_REQUIRE(workspace_bytes >= L.total, UCD_EWORKSPACE, "%s: workspace too small (%zu < %zu)", fn, workspace_bytes, L.total);
  char* ws = (char*)workspace;
  PlanHdr* hdr = (PlanHdr*)(ws + L.off_hdr);
  int* seg1 = (int*)(ws + L.off_seg1); int* seg2 = (int*)(ws + L.off_seg2);
  int* us1 = (int*)(ws + L.off_us1); int* us2 = (int*)(ws + L.off_us2);
  int* order1 = (int*)(ws + L.off_order1); int* order2 = (int*)(ws + L.off_order2);
  float* negp = (float*)(ws + L.off_negp); float* mxp = (float*)(ws + L.off_mxp);
  float* lossp = (float*)(ws + L.off_lossp); float* qsump = (float*)(ws + L.off_qsump);
  float* rowloss = (float*)(ws + L.off_rowloss);
  float* Up = (float*)(ws + L.off_Up); float* Vp = (float*)(ws + L.off_Vp);
  const int KP16 = use_prob ? (K + 15) / 16 * 16 : 0;
  const float k2 = kLog2e / temperature;
  int dev = 0, cus = 0;
  UCD_REQUIRE(hipGetDevice(&dev) == hipSuccess && hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && cus > 0,
              UCD_EUNSUPPORTED, "%s: cannot query the device", fn);
  const int grid = cus < L.umax ? cus : L.umax;
  pixcon16p_plan_kernel<<<1, kPlanThreads, 0, s>>>(row_label, meta, L.umax, hdr, seg1, seg2, us1, us2, order1, order2);
  int rc = check_launch(fn);
  if (rc) return rc;
  const size_t lds_base = (size_t)kRing * kBufHalfs * 2 + kRing * 8 * 4 + 16;
  const size_t lds_prob = lds_base + (size_t)kRing * kTJ * (2 * KP16 + 8) * 2;
  // opt in to more than 64 KiB of dynamic LDS, per call (no process-wide state); only what is needed: the block-wide vote of
  // the probability instance keeps a static word of its own
  UCD_TRY_LDS((pixcon16p_sweep_kernel<0, false>), (int)lds_base);
  UCD_TRY_LDS((pixcon16p_sweep_kernel<1, false>), (int)lds_base);
  UCD_TRY_LDS((pixcon16p_sweep_kernel<1, true>), (int)lds_prob);
  pixcon16p_sweep_kernel<0, false><<<grid, kThreads, lds_base, s>>>(ch16, row_label, nullptr, 0, meta, hdr, seg1, us1, order1, us1,
                                                                    k2, shift_pos, nullptr, nullptr, negp, mxp, Up);
  rc = check_launch(fn);
  if (rc) return rc;
  if (use_prob)
    pixcon16p_sweep_kernel<1, true><<<grid, kThreads, lds_prob, s>>>(ch16, row_label, p16, KP16, meta, hdr, seg2, us2, order2, us1,
                                                                    k2, shift_pos, negp, mxp, lossp, qsump, Vp);
  else
    pixcon16p_sweep_kernel<1, false><<<grid, kThreads, lds_base, s>>>(ch16, row_label, nullptr, 0, meta, hdr, seg2, us2, order2,
                                                                      us1, k2, shift_pos, negp, mxp, lossp, qsump, Vp);
  rc = check_launch(fn);
  if (rc) return rc;
  pixcon16p_finalize_kernel<<<ceil_div(BHW, kThreads / 64), kThreads, 0, s>>>(row_label, meta, 1.f / temperature, k2 - 14.5f,
                                                                              us1, us2, negp, lossp, qsump, Up, Vp, grad_a,
                                                                              ldg, row_stats, BHW, rowloss);
  rc = check_launch(fn);
  if (rc) return rc;
  pixcon_launch_reduce(rowloss, meta, loss_out, s);
  return check_launch(fn);
}

}  // namespace ucd
