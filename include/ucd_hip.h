/*
 * ucd_hip.h - C ABI of libucd_hip.so: the MI355X (gfx950) kernels of the UCD train-step hot path.
 *
 * The reference (ygjwd12345/UCD) has no FFI layer of its own: its native work enters through the
 * third-party wheels inplace-abn / apex and through ~40 stock PyTorch kernels per loss.  Each entry
 * point below names the reference call site (file:line under the reference tree) whose device work
 * it replaces; INTEGRATION.md shows the ctypes binding a reference maintainer would add.
 *
 * Conventions
 *   - every pointer is a DEVICE pointer owned by the caller (PyTorch); nothing is allocated, freed or
 *     retained past the call; workspaces are caller-provided and sized by the *_workspace_bytes calls;
 *   - every call is asynchronous on the given hipStream_t (passed as void*), re-entrant, and makes
 *     no host synchronisation (data-dependent sizes such as the anchor count stay on the device);
 *   - return 0 on success, a negative UCD_E* code for argument errors, or the positive hipError_t of
 *     a failed launch; ucd_last_error() returns the thread-local message; no C++ exception crosses;
 *   - activations ("act" tensors) are channels-last: a [B,C,H,W] map is the row-major matrix
 *     [M = B*H*W rows][C channels] with a leading dimension `ld` in ELEMENTS (ld >= C, so a channel
 *     slice of a wider buffer can be addressed); dtype is UCD_F32 or UCD_BF16; base pointers and
 *     ld*sizeof(elem) must be 16-byte aligned and C a multiple of 16/sizeof(elem);
 *   - per-channel parameters and statistics are float32.
 */
#ifndef UCD_HIP_H
#define UCD_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define UCD_VERSION 100 /* 0.1.0 */

typedef void* ucd_stream_t; /* hipStream_t */
typedef void* ucd_comm_t;   /* communicator owned by this library (ucd_comm_init: RCCL, optionally with an IPC mailbox) */

enum ucd_dtype { UCD_F32 = 0, UCD_BF16 = 1 };
enum ucd_act { UCD_ACT_IDENTITY = 0, UCD_ACT_LEAKY_RELU = 1, UCD_ACT_ELU = 2 /* slope = alpha */ };
/* Flag bits OR-ed into an `act` (or `flags`) argument.  UCD_NORM_ABS_GAMMA selects the parameterisation of
 * inplace_abn's InPlaceABN / InPlaceABNSync, gamma~ = |weight| + eps in place of weight (their published forward;
 * d weight = sign(weight) * sum dz*xhat, sign(0) = +1); without it the layer is F.batch_norm (inplace_abn.ABN). */
#define UCD_ACT_MASK 0xff
#define UCD_NORM_ABS_GAMMA 0x100
/* UCD_PIXCON_F16_SPLIT: the fp16 path forced onto its fixed-split kernels (what UCD_PIXCON_F16 falls back to for
 * T < 0.06, more than 32 teacher classes or more than 1023 anchor blocks); kept selectable as the A/B reference. */
enum ucd_pixcon_precision { UCD_PIXCON_F32 = 0, UCD_PIXCON_F16 = 1, UCD_PIXCON_F16_SPLIT = 2 };
enum ucd_error {
  UCD_OK = 0,
  UCD_EINVAL = -1,      /* bad argument (null pointer, negative size, unknown enum) */
  UCD_EALIGN = -2,      /* pointer / leading dimension / channel count not 16-byte friendly */
  UCD_EWORKSPACE = -3,  /* workspace too small */
  UCD_EUNSUPPORTED = -4, /* shape outside what the kernels are built for */
  UCD_ETIMEOUT = -5     /* a mailbox exchange of the communicator timed out earlier (a rank never wrote): the communicator is dead */
#define UCD_ERCCL_BASE 100000 /* RCCL failures are returned as UCD_ERCCL_BASE + ncclResult_t */
#define UCD_EBLAS_BASE 200000 /* hipBLASLt failures are returned as UCD_EBLAS_BASE + hipblasStatus_t */
};

int ucd_version(void);
const char* ucd_last_error(void);
/* `bytes` zero bytes at `ptr`, in stream order (one memset node when the stream is being captured): the per-step fill of the
 * statistics arena (ucd_conv1x1_desc.stat_acc; csrc/abn_node.cpp) - a plain memset, so no tensor version counter moves. */
int ucd_fill_zero(void* ptr, size_t bytes, ucd_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * ABN: fused BatchNorm + LeakyReLU(slope) / identity.  Replaces the inplace-abn==1.0.7 CUDA extension
 * behind every norm_act(...) instance: models/resnet.py:60, modules/residual.py:51,56,64,68,71,81,
 * modules/deeplab.py:30,33,37 (selected at segmentation_module.py:15-20), plus the glue the reference
 * runs as separate kernels around it: residual add + activation (modules/residual.py:90-97), channel
 * concatenation (modules/deeplab.py:56) and the pooled-branch broadcast add (modules/deeplab.py:65-68).
 *
 *   z = (x [+ plane_bias[b, c]] - mean[c]) * scale[c] + shift[c] [+ residual]      y = act(z)
 * ---------------------------------------------------------------------------------------------- */

/* bytes of scratch needed by ucd_abn_stats / ucd_abn_bwd_reduce for an [M, C] map */
size_t ucd_abn_workspace_bytes(int M, int C);

/* Per-channel sums over the M rows, taken about k[c] = x'[row 0, c] ("shifted data": E[x^2] - E[x]^2
 * cancels catastrophically in float32 when |mean| >> std):
 *   sums[0:C] = sum (x' - k), sums[C:2C] = sum (x' - k)^2, kshift[0:C] = k,
 * with x' = x + plane_bias[row / HW, c] (plane_bias may be NULL).  Deterministic two-stage reduction. */
int ucd_abn_stats(const void* x, int ld_x, int dtype, int M, int C,
                  const float* plane_bias, int HW,
                  float* sums /* [2*C] */, float* kshift /* [C] */,
                  void* workspace, size_t workspace_bytes, ucd_stream_t stream);

/* ucd_abn_stats followed by ucd_abn_finalize with count = M, in one launch sequence less (the stage-2
 * reduction finalises its own channels): the single-process training forward. */
int ucd_abn_stats_finalize(const void* x, int ld_x, int dtype, int M, int C, const float* plane_bias, int HW,
                           float* sums, float* kshift, const float* weight,
                           float* running_mean, float* running_var, float momentum, float eps,
                           float* mean, float* invstd, float* scale, int flags /* 0 or UCD_NORM_ABS_GAMMA */,
                           void* workspace, size_t workspace_bytes, ucd_stream_t stream);

/* Batch statistics -> normalisation constants.  With d = sums[c]/count:
 *   mean = kshift + d,  var = (sums[C+c] - sums[c]*d)/count (biased),  invstd = 1/sqrt(var + eps),
 *   scale = weight * invstd ((|weight| + eps) * invstd with flags = UCD_NORM_ABS_GAMMA);  running_mean / running_var
 *   are updated in place with `momentum` (unbiased variance) unless NULL.  kshift NULL means 0; weight NULL means 1.
 * For statistics combined across ranks pass kshift = global mean, sums[0:C] = 0, sums[C:2C] = global M2. */
int ucd_abn_finalize(const float* sums, const float* kshift, float count, int C, const float* weight,
                     float* running_mean, float* running_var, float momentum, float eps,
                     float* mean, float* invstd, float* scale, int flags, ucd_stream_t stream);

/* Evaluation mode (teacher; --fix_bn): invstd = 1/sqrt(running_var + eps), scale = weight * invstd; the
 * mean is running_mean itself. */
int ucd_abn_eval_params(const float* weight, const float* running_var, float eps, int C,
                        float* invstd, float* scale, int flags /* 0 or UCD_NORM_ABS_GAMMA */, ucd_stream_t stream);

/* y = act((x + plane_bias - mean) * scale + shift + residual); shift is the affine bias (NULL = 0);
 * y may alias x (in place) or be a channel slice of a wider buffer (ld_y > C); residual / plane_bias
 * may be NULL.  Subtracting the mean first keeps the result exact when x is close to it.
 * bf16 tensors without a plane bias under leaky_relu / identity (this call, ucd_abn_bwd_reduce and ucd_abn_bwd_apply) run on
 * packed fp32 pairs (round 4: same operations in the same order, bit-identical, 2 - 3x faster on the 13 - 27 MB layers);
 * UCD_ABN_GENERIC=1 in the environment (read once per process) keeps the per-element kernels everywhere. */
int ucd_abn_apply(const void* x, int ld_x, void* y, int ld_y, const void* residual, int ld_r,
                  int dtype, int M, int C, const float* plane_bias, int HW,
                  const float* mean, const float* scale, const float* shift, int act, float slope,
                  ucd_stream_t stream);

/* Backward, stage 1: dz = dy * act'(z) and the two per-channel sums
 *   sums[0:C] = sum dz            (= d bias)
 *   sums[C:2C] = sum dz * xhat    (= d weight),   xhat = (x' - mean) * invstd.
 * The sign of z is taken from y when y != NULL (needed when a residual was fused), else z =
 * (x' - mean) * scale + shift is recomputed from x (shift = the affine bias, NULL = 0).  With UCD_NORM_ABS_GAMMA in
 * `act`, sums[C:2C] is multiplied by sign(weight[c]) (d weight of gamma~ = |weight| + eps); ucd_abn_bwd_apply with the
 * same flag undoes the sign, so the pair stays consistent (also across an all-reduce: the sign is rank-independent). */
int ucd_abn_bwd_reduce(const void* x, int ld_x, const void* dy, int ld_dy, const void* y, int ld_y,
                       int dtype, int M, int C, const float* plane_bias, int HW,
                       const float* mean, const float* invstd, const float* scale, const float* shift,
                       const float* weight /* read only with UCD_NORM_ABS_GAMMA */, int act, float slope,
                       float* sums /* [2*C] */, void* workspace, size_t workspace_bytes, ucd_stream_t stream);

/* Backward, stage 2: dx = (dz - sums[0]/count - xhat * sums[1]/count) * weight * invstd
 * (training statistics) or dx = dz * scale when frozen != 0 (running statistics).  dz_out, when not
 * NULL, receives dz (the gradient of the fused residual input).  dx may alias dy. */
int ucd_abn_bwd_apply(const void* x, int ld_x, const void* dy, int ld_dy, const void* y, int ld_y,
                      void* dx, int ld_dx, void* dz_out, int ld_dz,
                      int dtype, int M, int C, const float* plane_bias, int HW,
                      const float* mean, const float* invstd, const float* scale, const float* shift,
                      const float* weight, const float* sums, float count, int frozen,
                      int act, float slope, ucd_stream_t stream);

/* Apply pass that FINALISES the statistics itself (round 5, bf16 activations, leaky_relu / identity): acc[0..C) = sum (x - k),
 * acc[C..2C) = sum (x - k)^2 over `count` rows (the atomic accumulator of ucd_conv1x1's statistics epilogue, all-reduced over the
 * ranks under SyncBN; `reps` replicas of [2 C], summed here), kshift[c] = k.  Every workgroup derives mean / invstd / scale of its channels in its prologue (2C loads);
 * the first row band also stores them to mean / invstd / scale (saved for the backward) and updates the running statistics
 * (momentum, unbiased variance) - what ucd_conv1x1_stats_finalize did in a launch of its own.  flags: UCD_NORM_ABS_GAMMA. */
int ucd_abn_apply_stats(const void* x, int ld_x, void* y, int ld_y, const void* residual, int ld_r, int M, int C,
                        const float* acc, int reps, const float* kshift, float count, const float* weight, const float* bias,
                        float* running_mean, float* running_var, float momentum, float eps, float* mean, float* invstd,
                        float* scale, int act, float slope, ucd_stream_t stream);

/* Backward apply on RAW sums (round 5, bf16): sums[0..C) = sum dz, sums[C..2C) = sum dz * xhat as accumulated by ucd_conv1x1's
 * out_mode 3 / 4 epilogues with stat_acc (no sign applied, all-reduced over the ranks under SyncBN; count = rows over all ranks).
 * grad_out (optional): [d bias | d weight] of the layer is written by the first row band from grad_sums (this rank's sums; NULL:
 * sums; both `reps` replicas of [2 C], summed in the prologue), d weight with the sign of weight under UCD_NORM_ABS_GAMMA - what ucd_abn_reduce_partials did in a launch of its own.
 * Arguments otherwise as ucd_abn_bwd_apply (training statistics; identity / leaky_relu). */
int ucd_abn_bwd_apply_raw(const void* x, int ld_x, const void* dy, int ld_dy, const void* y, int ld_y, void* dx, int ld_dx,
                          void* dz_out, int ld_dz, int M, int C, const float* mean, const float* invstd, const float* scale,
                          const float* shift, const float* weight, const float* sums, const float* grad_sums, int reps,
                          float* grad_out, float count, int act, float slope, ucd_stream_t stream);

/* Whole-layer forward in one call (single process): training != 0 -> ucd_abn_stats_finalize + ucd_abn_apply with
 * buf = [sums(2C) | kshift(C) | mean(C) | invstd(C) | scale(C)] (kept for the backward); training == 0 ->
 * running statistics, with [invstd | scale] taken from eval_consts when given (frozen teacher) or computed into buf. */
int ucd_abn_forward(const void* x, int ld_x, void* y, int ld_y, const void* residual, int ld_r,
                    int dtype, int M, int C, const float* plane_bias, int HW,
                    const float* weight, const float* bias, float* running_mean, float* running_var,
                    float momentum, float eps, int training, float* buf, const float* eval_consts,
                    int act, float slope, void* workspace, size_t workspace_bytes, ucd_stream_t stream);

/* Whole-layer backward in one call: ucd_abn_bwd_reduce (when training or need_sums) + ucd_abn_bwd_apply.
 * sums[0:C] = d bias, sums[C:2C] = d weight on return. */
int ucd_abn_backward(const void* x, int ld_x, const void* dy, int ld_dy, const void* y, int ld_y,
                     void* dx, int ld_dx, void* dz_out, int ld_dz, int dtype, int M, int C,
                     const float* plane_bias, int HW, const float* mean, const float* invstd,
                     const float* scale, const float* bias, const float* weight, float* sums, float count,
                     int training, int need_sums, int act, float slope,
                     void* workspace, size_t workspace_bytes, ucd_stream_t stream);

/* SyncBN across one-process-per-GPU ranks (InPlaceABNSync, segmentation_module.py:17; the inplace-abn
 * extension all-reduces the statistics of every layer in forward and backward).  The collectives stay with the
 * caller (torch.distributed / RCCL); these are the library calls around them, equal row counts per rank:
 *
 *   forward   ucd_abn_sync_stats    pack[0:C] = mean_r, pack[C:2C] = M2_r = sum (x' - mean_r)^2 of this rank
 *             all_gather(pack)   -> gathered[world][2C]
 *             ucd_abn_sync_forward  Chan's combination (mean = avg mean_r, M2 = sum M2_r + M sum (mean_r - mean)^2),
 *                                   finalize with count = world * M into buf = [. . . | mean | invstd | scale]
 *                                   (same layout as ucd_abn_forward), running statistics update, then apply
 *   backward  ucd_abn_sync_bwd_reduce  sums = local_sums = [sum dz | sum dz xhat] of this rank
 *             all_reduce(sums)
 *             ucd_abn_bwd_apply with count = world * M    (local_sums are the rank's d bias / d weight) */
int ucd_abn_sync_stats(const void* x, int ld_x, int dtype, int M, int C, const float* plane_bias, int HW,
                       float* sums /* [2*C] scratch */, float* kshift /* [C] scratch */, float* pack /* [2*C] */,
                       void* workspace, size_t workspace_bytes, ucd_stream_t stream);
int ucd_abn_sync_forward(const void* x, int ld_x, void* y, int ld_y, const void* residual, int ld_r,
                         int dtype, int M, int C, const float* plane_bias, int HW,
                         const float* gathered /* [world][2*C] */, int world,
                         const float* weight, const float* bias, float* running_mean, float* running_var,
                         float momentum, float eps, float* buf /* [6*C] */, int act, float slope, ucd_stream_t stream);
int ucd_abn_sync_bwd_reduce(const void* x, int ld_x, const void* dy, int ld_dy, const void* y, int ld_y,
                            int dtype, int M, int C, const float* plane_bias, int HW,
                            const float* mean, const float* invstd, const float* scale, const float* shift,
                            const float* weight, int act, float slope, float* sums /* [2*C] */, float* local_sums /* [2*C] */,
                            void* workspace, size_t workspace_bytes, ucd_stream_t stream);

/* ---- row-major bf16 GEMMs of the wide 1x1 convolutions (modules/residual.py:57-63 builds them as nn.Conv2d(k=1);
 * on the channels-last row matrix [B*H*W, C] they are plain GEMMs), fp32 accumulation, bf16 output, through hipBLASLt
 * with the algorithm picked once per shape by timing the library's candidates (tune != 0) and cached:
 *   mode 0:  C[M,N] = A[M,K] . B[N,K]^T     forward          rows x weight^T
 *   mode 1:  C[M,N] = A[M,K] . B[K,N]       input gradient   dY x weight
 *   mode 2:  C[M,N] = A[K,M]^T . B[K,N]     weight gradient  dY^T x rows   (K = B*H*W)
 * lda/ldb/ldc = row pitch in elements.  ucd_gemm_load binds hipBLASLt from the shared object the process already uses. */
int ucd_gemm_load(const char* hipblaslt_path);
size_t ucd_gemm_workspace_bytes(void);
int ucd_gemm_bf16(int mode, int M, int N, int K, const void* A, int lda, const void* B, int ldb, void* C, int ldc,
                  void* workspace, size_t workspace_bytes, int tune, ucd_stream_t stream);
/* C += op(A) op(B) with the plan ucd_gemm_bf16 holds for the shape (the residual-branch gradient folded into the input
 * gradient of a block's first 1x1 convolution); never tunes - warm the shape through ucd_gemm_bf16 first
 * (ucd_gemm_has_plan tells) or it takes the library's first suggestion */
int ucd_gemm_bf16_acc(int mode, int M, int N, int K, const void* A, int lda, const void* B, int ldb, void* C, int ldc,
                      void* workspace, size_t workspace_bytes, ucd_stream_t stream);
int ucd_gemm_has_plan(int mode, int M, int N, int K, int lda, int ldb, int ldc);
float ucd_gemm_last_tuned_us(void);
int ucd_gemm_last_candidates(void);

/* ---- library-owned RCCL communicator (one process per GPU) ------------------------------------------
 * Replaces the per-layer torch.distributed collectives of InPlaceABNSync (the reference reaches NCCL through the
 * inplace-abn extension, segmentation_module.py:17): the exchange runs on the caller's stream inside the layer call.
 *   ucd_comm_load(path)      bind RCCL from the shared object the process already uses (NULL/"" = default search)
 *   rank 0: ucd_comm_unique_id(id, 128); every rank receives the 128 bytes (e.g. one torch.distributed broadcast)
 *   every rank: ucd_comm_init(id, 128, nranks, rank, &comm)         (collective; current HIP device = the rank's GPU) */
int ucd_comm_load(const char* rccl_path);
int ucd_comm_unique_id(void* id_out, size_t bytes);
int ucd_comm_init(const void* id, size_t bytes, int nranks, int rank, ucd_comm_t* comm_out);
int ucd_comm_destroy(ucd_comm_t comm);
int ucd_comm_all_gather(ucd_comm_t comm, const float* send, float* recv /* [nranks*count] */, size_t count,
                        ucd_stream_t stream);
int ucd_comm_all_reduce_sum(ucd_comm_t comm, float* buf, size_t count, ucd_stream_t stream);

/* One-shot mailbox exchange for the small collectives (round 5; SURVEY section 7 hard part 1): every rank owns a mailbox in device
 * memory shared through hipIpcMemHandle; ucd_comm_all_gather / ucd_comm_all_reduce_sum of at most `slot_floats` floats then run as
 * ONE kernel of one workgroup on the caller's stream - peer stores, system-scope fence, sequence flags, bounded spin, sum in rank
 * order (bit-identical on every rank) - instead of an RCCL call; larger messages keep RCCL.  Set-up (every rank, collective):
 *   ucd_comm_init (RCCL underneath) or ucd_comm_init_local (no RCCL: mailbox only - several ranks on ONE GPU, which RCCL refuses)
 *   ucd_comm_ipc_create(comm, slot_floats, timeout_ms, handle)  -> exchange the ucd_comm_ipc_handle_bytes() bytes of every rank
 *   ucd_comm_ipc_connect(comm, handles)      handles = [nranks][handle bytes] in rank order
 * A peer that never writes makes the kernel give up after timeout_ms (<= 0: 2 s).  It then (round 6) writes NaN into its output,
 * latches a word in pinned host memory (ucd_comm_ipc_timeouts; every later host-issued collective on the communicator returns
 * UCD_ETIMEOUT) and poisons the mailbox of EVERY rank: all later exchange kernels of the communicator - replayed graphs included -
 * return NaN at once instead of waiting.  The mailbox lives in fine-grained device memory; where that cannot be allocated
 * ucd_comm_ipc_create fails (no coarse-grained fallback) and the caller keeps RCCL.  ucd_comm_ipc_drop removes the mailbox
 * (the collectives go back to RCCL).  ONE stream at a time per communicator: exchanges of one mailbox must be stream-ordered. */
int ucd_comm_init_local(int nranks, int rank, ucd_comm_t* comm_out);
size_t ucd_comm_ipc_handle_bytes(void);
int ucd_comm_ipc_create(ucd_comm_t comm, int slot_floats, int timeout_ms, void* handle_out);
int ucd_comm_ipc_connect(ucd_comm_t comm, const void* handles);
int ucd_comm_ipc_drop(ucd_comm_t comm);
unsigned ucd_comm_ipc_timeouts(ucd_comm_t comm);

/* Whole SyncBN layer in one call each way, collectives included (comm from ucd_comm_init, world = its size):
 *   forward   ucd_abn_sync_stats -> all-gather -> ucd_abn_sync_forward; buf = [6*C | pack 2*C | gathered world*2*C]
 *   backward  ucd_abn_sync_bwd_reduce -> all-reduce -> ucd_abn_bwd_apply(count = world*M);
 *             sums [2*C] = the global sums on return, local_sums [2*C] = this rank's [d bias | d weight]
 *             (may point straight at the parameters' gradient storage) */
int ucd_abn_sync_forward_comm(ucd_comm_t comm, int world, const void* x, int ld_x, void* y, int ld_y,
                              const void* residual, int ld_r, int dtype, int M, int C, const float* plane_bias, int HW,
                              const float* weight, const float* bias, float* running_mean, float* running_var,
                              float momentum, float eps, float* buf, int act, float slope,
                              void* workspace, size_t workspace_bytes, ucd_stream_t stream);
int ucd_abn_sync_backward_comm(ucd_comm_t comm, int world, const void* x, int ld_x, const void* dy, int ld_dy,
                               const void* y, int ld_y, void* dx, int ld_dx, void* dz_out, int ld_dz, int dtype, int M,
                               int C, const float* plane_bias, int HW, const float* mean, const float* invstd,
                               const float* scale, const float* bias, const float* weight, float* sums,
                               float* local_sums, int act, float slope, void* workspace, size_t workspace_bytes,
                               ucd_stream_t stream);

/* Per-(image, channel) reduction over the HW rows of each image: out[b, c] = alpha * sum_hw x.
 * Global average pooling of the ASPP image-level branch (modules/deeplab.py:72-76) with alpha = 1/HW,
 * and the gradient of a plane_bias (sum of dz over the plane) with alpha = 1. */
int ucd_plane_sum(const void* x, int ld_x, int dtype, int B, int HW, int C, float alpha,
                  float* out /* [B, C] */, ucd_stream_t stream);

/* Sliding-window mean with stride 1 and no padding: out[b, oy, ox, c] = mean of the ph x pw window of x at (oy, ox);
 * out is [B, H-ph+1, W-pw+1, C] channels-last.  Replaces F.avg_pool2d(x, (ph, pw), stride=1) of the image-pooling branch in
 * evaluation mode (modules/deeplab.py:77-83; the teacher at 513^2 and 768^2, --pooling 32) with one read of the map. */
size_t ucd_window_mean_workspace_bytes(int B, int H, int W, int C, int ph);
int ucd_window_mean(const void* x, int ld_x, int dtype, int B, int H, int W, int C, int ph, int pw, void* out, int ld_out,
                    void* workspace, size_t workspace_bytes, ucd_stream_t stream);

/* Spatial attention map (segmentation_module.py:86-94): a[b,p] = sum_c x^2, normalised per image by
 * its Frobenius norm; y = a * x.  workspace: B*HW + B floats. */
size_t ucd_attmap_workspace_bytes(int B, int HW);
int ucd_attmap(const void* x, int ld_x, void* y, int ld_y, int dtype, int B, int HW, int C,
               void* workspace, size_t workspace_bytes, ucd_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * Uncertainty-weighted pixel-contrastive distillation.  Replaces pre_contractive_pixel
 * (utils/utils.py:256-393; twin utils/loss.py:258-395) and PixelConLossV2.forward
 * (utils/loss.py:412-466) - and, with shift_pos = 0, no teacher and no joint-probability weight,
 * PixelConLoss.forward of the dead file utils/loss_new.py:359-400.
 * ---------------------------------------------------------------------------------------------- */

/* Device-resident description of one batch's anchor / contrast sets, written by ucd_pixcon_prep and
 * read by the loss kernels (the host never needs it; tests copy it back). */
typedef struct ucd_pixcon_meta {
  int32_t A;        /* anchors: pixels with mixed label > 0                      (utils.py:358) */
  int32_t Co;       /* teacher-only contrast rows: kept pixels that are not new (utils.py:359) */
  int32_t min_new;  /* smallest down-sampled ground-truth label > 0              (utils.py:353) */
  int32_t n_new;    /* pixels with a ground-truth (new-class) label              (utils.py:352) */
  int32_t Apad;     /* A rounded up to the contrast tile (row offset of the teacher segment) */
  int32_t Cpad;     /* Apad + Co rounded up to the tile: rows of the contrast matrix */
  int32_t n_valid;  /* anchors with at least one positive (rows of the final mean, loss.py:465) */
  int32_t sorted;   /* rows grouped by label (sort_by_label) */
  /* with sorted != 0: rows [label_start_a[L], label_start_a[L+1]) of the anchor segment and rows
     Apad + [label_start_o[L], label_start_o[L+1]) of the teacher segment carry label L */
  int32_t label_start_a[257];
  int32_t label_start_o[257];
  int32_t label_count_a[256]; /* anchors per label */
  int32_t label_count_c[256]; /* contrast rows per label (anchors + teacher rows): num_i + 1 */
  int32_t reserved[2];
} ucd_pixcon_meta;

size_t ucd_pixcon_prep_workspace_bytes(int BHW, int K);

/* Stage 1 (utils/utils.py:264-268,352-359,367-371): per low-resolution pixel p of the [B,h,w] grid
 *   label_ds = trunc(bilinear(labels)) kept when in [0, max_label] else 0   (bit-exact float32
 *              arithmetic of torch's one-channel bilinear kernel, align_corners = False);
 *   mix      = label_ds, or the teacher arg-max where label_ds == 0;
 *   keep     = mix > 0;  keep_o = keep && label_ds == 0;
 * then order-preserving stream compaction.  sort_by_label != 0 additionally groups anchors and
 * teacher rows by label (stable), which is what the fused loss wants; 0 keeps the reference's pixel
 * order.  Outputs (all device):
 *   anchor_pix[BHW], old_pix[BHW]  : pixel index of anchor row r / teacher row r
 *   row_label[2*BHW + 2*tile]      : label of contrast row r (anchors first, teacher rows from
 *                                    meta->Apad), 255 for padding rows
 *   prob[BHW, K]                   : softmax of the teacher logits per pixel (float32)
 *   meta                           : counts, see above
 * teacher_logits is the [B*h*w, K] channels-last low-resolution teacher output ("sem",
 * segmentation_module.py:136) in float32 or bf16 with leading dimension ld_t. */
int ucd_pixcon_prep(const int64_t* labels, int B, int H, int W, int h, int w, int max_label,
                    const void* teacher_logits, int ld_t, int dtype_t, int K, int sort_by_label,
                    int32_t* anchor_pix, int32_t* old_pix, uint8_t* row_label, float* prob,
                    ucd_pixcon_meta* meta, void* workspace, size_t workspace_bytes, ucd_stream_t stream);

/* Stage 2 (utils/utils.py:361-364,372-375): gather the kept rows of the student (anchors) and teacher
 * (teacher-only rows) pre-logit maps, L2-normalise each row (F.normalize, eps 1e-12) and write the
 * contrast matrix chat[Cpad, ldc] (float32): rows [0, A) anchors, [Apad, Apad+Co) teacher rows,
 * zeros elsewhere (also columns N..ldc); inv_norm[r] = 1 / max(||row||, eps) for the anchors (needed
 * by the backward).  When pcat != NULL the teacher probabilities of the same pixels are gathered in
 * the same row order into pcat[Cpad, ldp] (columns K..ldp zero).
 * f_n / f_o are [B*h*w, N] channels-last maps (float32 or bf16). */
int ucd_pixcon_gather(const void* f_n, int ld_n, const void* f_o, int ld_o, int dtype, int BHW, int N,
                      const int32_t* anchor_pix, const int32_t* old_pix, const float* prob, int K,
                      const ucd_pixcon_meta* meta, float* chat, int ldc, float* pcat, int ldp,
                      void* ch16 /* fp16 [Cpad, ldc] copy of chat, or NULL */,
                      void* p16 /* fp16 [Cpad, 2, K rounded up to 16]: hi | lo split of pcat, or NULL */,
                      float* inv_norm, ucd_stream_t stream);

size_t ucd_pixcon_loss_workspace_bytes(int BHW, int N, int K);

/* Loss and its gradient w.r.t. the normalised anchors in one launch sequence (utils/loss.py:435-466):
 *   S_ij = a_i . c_j / T;  neg_i = sum_j [la_i != lc_j] exp(S_ij)                     (un-shifted)
 *   m_i = max_j S_ij (shift_pos != 0) or 0;  S' = S - m_i;  pos_ij = [la_i == lc_j] - [j == i]
 *   P_ij = 1 when use_prob == 0 or (la_i >= min_new and lc_j >= min_new), else p_i . p_j
 *   loss = mean_{i: num_i != 0} ( - sum_j pos_ij P_ij (S'_ij - log(exp S'_ij + neg_i)) / num_i )
 *   grad_a[i, :] = d loss / d a_i  (closed form); grad_a may be NULL.
 * chat [Cpad, ldc] / pcat [Cpad, ldp] / row_label are the outputs of ucd_pixcon_prep + ucd_pixcon_gather
 * (ldc must be 256: feature rows zero-padded to 256 columns; ldp >= K rounded up to even).
 * loss_out[0] = loss, loss_out[1] = number of valid rows.  row_stats (optional, [3, BHW]) receives
 * neg_i, num_i and the per-row loss for inspection.
 * precision = UCD_PIXCON_F32: exact float32 MFMA (v_mfma_f32_32x32x2_f32) on chat / pcat - the parity
 * mode; UCD_PIXCON_F16: fp16 operands ch16 / p16 (from ucd_pixcon_gather), fp32 accumulation
 * (v_mfma_f32_32x32x16_f16, 16x the rate) with an online rescale of the negative sums - the
 * performance mode; loss within ~1e-4, gradients within ~1e-3 of the float32 path. */
int ucd_pixcon_loss(const float* chat, int ldc, int N, const uint8_t* row_label,
                    const float* pcat, int ldp, int K, const void* ch16, const void* p16, int precision,
                    const ucd_pixcon_meta* meta, int BHW,
                    float temperature, int shift_pos, int use_prob,
                    float* loss_out, float* grad_a /* [BHW, ldg] */, int ldg, float* row_stats,
                    void* workspace, size_t workspace_bytes, ucd_stream_t stream);

/* The same loss with the weight matrix P MATERIALISED by the caller: the literal signature of the reference's
 * PixelConLossV2.forward(anchor_features, contrast_feature, anchor_labels, contrast_labels, P) (utils/loss.py:412), for
 * callers that hold plain tensors instead of a prepared batch.  chat [Cpad, 256] holds the anchors in rows [0, A) and the
 * remaining contrast rows (contrast_feature[A:]) in rows [meta->Apad, meta->Apad + Co); contrast_feature[:A] must be the
 * anchors themselves (the reference's construction, utils/utils.py:362; its self-pair mask `mask_p[:, :A] -= eye`
 * assumes it).  P is [A, A + Co] row-major in the reference's column order with leading dimension ld_P, or NULL for
 * P = 1; rows need not be normalised (the row maximum of S is computed, not assumed).  meta: A, Co, Apad, Cpad,
 * n_valid, sorted = 0 and label_count_c must be filled (device memory); max_anchors sizes the workspace
 * (ucd_pixcon_loss_workspace_bytes(max_anchors, N, 0)) and must be >= A.  float32 MFMA path only. */
int ucd_pixcon_loss_given_p(const float* chat, int ldc, int N, const uint8_t* row_label, const float* P, int ld_P,
                            const ucd_pixcon_meta* meta, int max_anchors, float temperature, int shift_pos,
                            float* loss_out, float* grad_a, int ldg, float* row_stats,
                            void* workspace, size_t workspace_bytes, ucd_stream_t stream);

/* Chain rule through F.normalize and scatter to the student map:
 *   d f_n[pix(r), :] = grad_scale[0] * inv_norm[r] * (g_r - (g_r . a_r) a_r),  zero for unkept pixels.
 * grad_scale is a device scalar (the upstream gradient of the loss).  d_f_n has dtype `dtype`. */
int ucd_pixcon_scatter_grad(const float* grad_a, const float* chat, int ldc, const float* inv_norm,
                            const int32_t* anchor_pix, const ucd_pixcon_meta* meta,
                            const float* grad_scale, void* d_f_n, int ld_d, int dtype, int BHW, int N,
                            ucd_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * Fused full-resolution logit losses.  Replaces, in one pass over the label map,
 * the bilinear up-sampling of the student and teacher logits (segmentation_module.py:133),
 * UnbiasedCrossEntropy (utils/loss.py:96-109; reduction 'none' then .mean() over ALL pixels, train.py:116)
 * and UnbiasedKnowledgeDistillationLoss (utils/loss.py:148-184, alpha = 1, reduction 'mean').
 *   sem_s [B*h*w, Ctot] / sem_t [B*h*w, K] : low-resolution student / teacher logits, float32 rows
 *   (sem_t may be NULL: cross entropy only; K = number of old classes incl. background, >= 1)
 *   loss_out[0] = mean CE, loss_out[1] = mean KD
 *   d_sem [B*h*w, ld_d] = d(ce_weight * CE + kd_weight * KD) / d sem_s   (overwritten)
 * The gradient is accumulated with float atomics (LDS, then global): its last bits depend on the
 * execution order; the loss values are summed in a fixed order. */
size_t ucd_seg_losses_workspace_bytes(int B, int H, int W);
int ucd_seg_losses(const float* sem_s, int ld_s, const float* sem_t, int ld_t, const int64_t* labels,
                   int B, int H, int W, int h, int w, int Ctot, int K, int ignore_index,
                   float ce_weight, float kd_weight, float* loss_out, float* d_sem, int ld_d,
                   void* workspace, size_t workspace_bytes, ucd_stream_t stream);

/* Validation on the device: bilinear up-sampling of the low-resolution logits
 * (segmentation_module.py:133), arg-max over the classes (train.py:242 `outputs.max(dim=1)`) and the confusion matrix of
 * metrics/stream_metrics.py:65-71 (`bincount(n * label + pred)` over the pixels with 0 <= label < n_classes) in one pass;
 * the full-resolution logits never exist and nothing is copied to the host.  hist [n_classes, n_classes] int64 is
 * ACCUMULATED into (zero it before the first batch; integer atomics: exact, order-independent); pred (optional,
 * [B, H, W] int64) receives the arg-max map (first maximum, torch.max's rule). */
int ucd_seg_confusion(const float* sem, int ld_s, const int64_t* labels, int B, int H, int W, int h, int w, int Ctot,
                      int n_classes, int64_t* hist, int64_t* pred, ucd_stream_t stream);

/* ---- label path of the training data pipeline on the device (first piece) ------------------------
 * Replaces, for the label maps of a batch, the reference's host-side RandomResizedCrop (crop + PIL NEAREST resize,
 * dataset/transform.py:481-553), RandomHorizontalFlip (:300-318) and the per-pixel Python lambda that remaps labels for
 * the incremental step (dataset/voc.py:176-203).  Bit-exact (index arithmetic of Pillow's NEAREST resize).
 *   src     B device pointers (array in device memory) to uint8 label maps, image b is [H0_b][W0_b] row-major
 *   desc    int32 [B][8] in device memory: {H0, W0, i, j, h, w, flip, 0}  (crop box top-left (i, j), size (h, w))
 *   lut     uint8 [256]: value after remapping for every stored label (inverted_order / masking_value)
 *   tables  int32 workspace [B][2][S];  out  int64 [B][S][S] (the labels tensor the train step takes) */
int ucd_label_path(const uint8_t* const* src, const int* desc, int B, int S, const uint8_t* lut, int* tables,
                   int64_t* out, ucd_stream_t stream);

/* Image half of the same pipeline: crop + Pillow BILINEAR resize (8-bit, separable, anti-aliased when down-scaling; the
 * fixed-point arithmetic of Pillow's Resample.c, bit-exact) + horizontal flip + ToTensor + Normalize
 * (dataset/transform.py:481-553, 300-318, 37-86; run.py:49-55).
 *   src   B device pointers to decoded uint8 RGB images [H0_b][W0_b][3];   desc as in ucd_label_path
 *   kmax  >= ceil(max(1, max crop extent / S)) * 2 + 1;   hmax >= max crop height of the batch
 *   mean_* / std_*  the Normalize constants;   coeff_ws int32 [B][2][S][kmax + 2];   tmp uint8 [B][hmax][S][3]
 *   out   float32 [B][S][S][3] = the channels-last storage of the [B, 3, S, S] batch */
int ucd_image_path(const uint8_t* const* src, const int* desc, int B, int S, int kmax, int hmax, float mean_r, float mean_g,
                   float mean_b, float std_r, float std_g, float std_b, int* coeff_ws, uint8_t* tmp, float* out,
                   ucd_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * 1x1 convolutions as row-matrix GEMMs on the bf16 matrix cores with the neighbouring ABN work fused (SURVEY.md
 * section 8-f4).  Replaces the cuDNN 1x1 convolutions of the bottleneck blocks and of the head -
 * modules/residual.py:57-63 (conv1, conv3), :79-80 (proj_conv), modules/deeplab.py:25,35 (map_convs[0], red_conv) - plus,
 * fused into them, the inplace_abn passes around them (modules/residual.py:64-73,81-82,90-97; modules/deeplab.py:57,69).
 *
 *   y[M, N] = out( in(a)[M, K] . w[N, K]^T )      bf16 operands, fp32 accumulation, bf16 result
 *
 *   in(a)    a itself, or (in_scale != NULL) act_in((a - in_mean[k]) * in_scale[k] + in_shift[k]): the ABN apply of the
 *            layer that PRODUCED a, applied while the tile is staged - that layer's output never exists in memory;
 *   out_mode 0  y = acc (+ y when accumulate != 0: the identity shortcut's gradient, beta = 1)
 *            1  y = act_out((acc - out_mean[n]) * out_scale[n] + out_shift[n] + residual[m, n]): this layer's ABN with
 *               frozen statistics (evaluation mode / --fix_bn), the block's residual add and activation
 *            2  y = acc and partial[t][0..2][n] = (k, sum (y - k), sum (y - k)^2) over the rows of row tile t, k = the
 *               tile's first row: input of ucd_conv1x1_stats_finalize (training: the following ABN's statistics pass)
 *            3  acc is the gradient w.r.t. in'(x) of a layer whose input transform was fused in the forward; with
 *               x = residual[m, n] (its pre-norm input), z = (x - out_mean) * out_scale + out_shift:
 *               y = dz = acc * act_out'(z) and partial[t][0..1][n] = (sum dz, sum dz * (x - out_mean) * out_invstd): input
 *               of ucd_abn_reduce_partials (that ABN's backward reduction)
 *            4  (1x1 only) acc (+ y when accumulate != 0: the identity shortcut's gradient) is the gradient w.r.t. the
 *               OUTPUT out = act(norm(z) + shortcut) of the residual block in front (modules/residual.py:84-97): with
 *               residual = out (the sign of the activation) and side2 = z: y = d pre = (acc + y) * act_out'(out) and
 *               partial[t][0..1][n] = (sum d pre, sum d pre * (z - out_mean) * out_invstd) - the backward sums of the
 *               block's last norm; its backward is then ucd_abn_reduce_partials + ucd_abn_bwd_apply (identity activation)
 *               on d pre, and the shortcut's gradient is d pre itself
 * act_in / act_out: UCD_ACT_LEAKY_RELU or UCD_ACT_IDENTITY (elu layers keep the separate ucd_abn_* kernels); modes 1 and 3
 * need out_mean, out_scale and out_shift (pass zeros / ones for a missing term).
 * Shapes: K and N multiples of 64, any M; pointers 16-byte aligned, leading dimensions (elements) multiples of 8.
 * partial: out_mode 2 needs ucd_conv1x1_stats_partial_bytes(M, N) bytes (the row tiles' [3][N] triples); out_mode 3 needs
 * ucd_conv1x1_row_tiles(M) rows of [2][N].  The input gradient of the layer is the same call on (dY, W^T)
 * (ucd_transpose_bf16 builds W^T); the weight gradient is ucd_conv1x1_wgrad. */
typedef struct ucd_conv1x1_desc {
  const void* a;  int lda;
  const void* w;  int ldw;
  void* y;        int ldy;
  int M, N, K;
  const float* in_mean;  const float* in_scale;  const float* in_shift;  int in_act;  float in_slope;
  int out_mode;
  const float* out_mean;  const float* out_scale;  const float* out_shift;  const float* out_invstd;
  const void* residual;  int ldr;
  int out_act;  float out_slope;
  float* partial;
  int accumulate;
  /* taps = 9: a is the [B, H, W, K] map and the product is the 3x3 convolution with stride 1 and padding = dilation
   * (modules/residual.py:69 conv2, modules/deeplab.py:27-29 the ASPP branches) as an implicit GEMM over 9 K columns: w is
   * the channels-last 4-D weight [N][kh][kw][K] (ldw >= 9 K), M = B*H*W; no input transform, every out_mode.  The input
   * gradient of such a layer is the same call on (dY, w.flip(2, 3).transpose(0, 1)) (ucd_flip_weights_batched).
   * taps = 0 or 1: the 1x1 product above. */
  int taps, H, W, dilation;
  /* out_mode 4 only: the conv output z of the residual block whose OUTPUT is `residual` (x-hat of its last norm's sums) */
  const void* side2;  int ld2;
  /* stride > 1 (0 and 1: none): the strided layers of the first block of a stage (modules/residual.py:57-73 conv2 with
   * stride 2, :79 proj_conv 1x1 stride 2).  a is the [B, H, W, K] map, y the [B, OH, OW, N] one with OH = (H - 1) / stride + 1
   * (padding 0 for taps <= 1, padding = dilation for taps = 9), M = B*OH*OW; out_mode 0, 1 or 2, no input transform. */
  int stride;
  /* Atomic statistics (round 5; NULL: the per-tile partial rows above).  stat_acc != NULL: out_mode 2 / 3 / 4 add their
   * per-channel column sums with fp32 atomics into stat_acc[0..N) / stat_acc[N..2N) (zeroed by the caller before the launch; one
   * accumulator per layer and direction, see ucd_abn_apply_stats / ucd_abn_bwd_apply_raw, which finalise it in their prologue - no
   * second-stage reduction launch).  out_mode 2 then sums about the caller's shift stat_shift[n] (any value near the channel mean:
   * the layer's running mean; NOT the tile's first row) and row tile 0 stores a snapshot of that shift to partial[0..N) (the consumer
   * reads the snapshot, so the running mean may be updated meanwhile).  stat_acc2 (optional): a second accumulator receiving the same
   * adds (SyncBN: one is all-reduced, the other stays this rank's parameter gradient).  Sums are order dependent: results differ
   * in the last bits from run to run. */
  float* stat_acc;  const float* stat_shift;  float* stat_acc2;
  /* stat_rep (0 / 1: one): REPLICAS of the accumulator, a power of two - row tile t adds into replica t % stat_rep (stat_acc holds
   * stat_rep x [2 N] floats); adds to one address are serialised at the memory side (~25 ns each), so a launch of hundreds of row
   * tiles spreads them (ucd_conv1x1_stat_replicas(M) is the library's choice: <= 64 adds per address) and the finalising apply pass
   * sums the replicas in its prologue. */
  int stat_rep;
} ucd_conv1x1_desc;

/* Kernel forms behind ucd_conv1x1 (chosen per launch by the grid; results do not depend on the choice - tests/
 * test_conv1x1_fused_gpu.py::test_every_pipeline_form... holds every form bit-exact on integers): single LDS stage at four workgroups
 * per CU (full grids), double buffer at two (<= 640 tiles, out_mode 4), and since round 4 the loader-wave forms - MFMA waves that
 * only multiply and loader waves that only stage - on 128-row tiles for grids of <= 256 tiles and on 256-row tiles for grids of
 * 257 .. 640 tiles that fit the chip as 256-row tiles.  The environment variable UCD_CONV_PIPE (read once per process: 2x64, 4x32,
 * 4x64, lw32, lw64, lw64x2, lw256) forces one form for every double-buffer-eligible launch; it exists for probes and A/B runs.
 * Launches of at most 128 (128 x 128) tiles (3 - 6 images per GPU) run on 128 x 64 tiles - same outputs bit for bit, twice the
 * workgroups; UCD_CONV_BN64_TILES (read once per process) sets that bound, 0 = never. */
int ucd_conv1x1_row_tiles(int M);
int ucd_conv1x1_stat_replicas(int M);   /* replicas of an atomic statistics accumulator for a product of M rows (1 .. 64) */
size_t ucd_conv1x1_stats_partial_bytes(int M, int C);
int ucd_conv1x1(const ucd_conv1x1_desc* desc, ucd_stream_t stream);

/* out_mode 2 partials -> batch statistics of the [M, C] product -> buf = [sums(2C) | kshift(C) | mean | invstd | scale]
 * (the layout ucd_abn_forward leaves for the backward) and the running statistics, like ucd_abn_stats_finalize; with
 * pack != NULL it writes this rank's [mean_r | M2_r] instead (SyncBN: all-gather, then ucd_abn_sync_forward). */
int ucd_conv1x1_stats_finalize(const float* partial, int M, int C, const float* weight, float* running_mean,
                               float* running_var, float momentum, float eps, float* buf, float* pack, int flags,
                               ucd_stream_t stream);

/* sums[0:C] / sums[C:2C] = sum over the `tiles` rows of partial[tiles][2][C] (fixed order); sums_copy (optional) receives
 * the same; with UCD_NORM_ABS_GAMMA in flags the second row is multiplied by sign(weight) (see ucd_abn_bwd_reduce). */
int ucd_abn_reduce_partials(const float* partial, int tiles, int C, float* sums, float* sums_copy, const float* weight,
                            int flags, ucd_stream_t stream);

/* dw[N, K] (bf16) = dy[M, N]^T . in(a)[M, K] with the forward's input transform re-applied to a (N, K multiples of 128).
 * workspace: ucd_conv1x1_wgrad_workspace_bytes(M, N, K) bytes of fp32 split-M partials, combined in a fixed order. */
size_t ucd_conv1x1_wgrad_workspace_bytes(int M, int N, int K);
int ucd_conv1x1_wgrad(const void* dy, int ld_dy, const void* a, int lda, int M, int N, int K,
                      const float* in_mean, const float* in_scale, const float* in_shift, int in_act, float in_slope,
                      void* dw, void* workspace, size_t workspace_bytes, ucd_stream_t stream);

/* The combination + finalize step of ucd_abn_sync_forward on its own (the stem's fused norm + pooling applies the result itself):
 * gathered [world][2C] = every rank's [mean_r | M2_r] -> buf = [.. | mean | invstd | scale] and the running statistics. */
int ucd_abn_sync_finalize(const float* gathered, int world, int M, int C, const float* weight, float* running_mean,
                          float* running_var, float momentum, float eps, float* buf, int flags, ucd_stream_t stream);

/* ---- stem: norm_act + MaxPool2d(3, stride 2, padding 1) as one pass (models/resnet.py:58-64, mod1.bn1 + mod1.pool1) -----------
 * z [B, H, W, C] dense channels-last bf16 (the 7x7 convolution's output), C a multiple of 8 that divides 2048.
 *   ucd_stem_apply_pool     out [B, PH, PW, C] = max over each 3x3 window of bf16(act((z - mean) scale + beta)) - bit for bit
 *                           max_pool2d(abn_apply(z)), first maximum wins; idx (optional, uint8 [B, PH, PW, C]) = window position
 *                           0..8 of the maximum, kept for the backward; PH = ucd_stem_pooled_size(H)
 *   ucd_stem_pool_backward  phase 1: sums [2C] = (sum dyact, sum dyact xhat) with dyact = dpool * act' at the arg-max positions
 *                           (the layer's d bias / d weight; the second row times sign(weight) for UCD_NORM_ABS_GAMMA layers);
 *                           phase 2: dz [B, H, W, C] = the norm's input gradient for every position (count = B H W x ranks);
 *                           phase 3: both.  SyncBN all-reduces sums between the phases.  workspace:
 *                           ucd_stem_pool_workspace_bytes(C).  leaky_relu / identity only. */
/* The stem's convolution itself (models/resnet.py:58: Conv2d(3, 64, 7, stride 2, padding 3, bias False)): x fp32 [B, 3, H, W] with
 * element strides (sb, sc, sh, sw) - the loader's image in either memory format, converted to bf16 while it is staged -, w the
 * bf16 weight in channels-last memory order [64][7][7][3], z [B, OH, OW, 64] dense channels-last bf16, OH = (H - 1) / 2 + 1. */
int ucd_stem_conv7x7(const float* x, long long sb, long long sc, long long sh, long long sw, int B, int H, int W, const void* w,
                     void* z, ucd_stream_t stream);

/* The same convolution followed by the frozen-statistics norm + activation + 3x3 / 2 max pool (models/resnet.py:58-64 in evaluation
 * mode: the teacher) in ONE kernel: out [B, PH, PW, 64] bf16 = max_pool2d(act((conv(x) - mean) * scale + beta), 3, 2, 1), the
 * convolution output rounded to bf16 in between as ucd_stem_conv7x7 stores it - bit-identical to ucd_stem_conv7x7 followed by
 * ucd_stem_apply_pool, without the 257 x 257 x 64 map in memory.  leaky_relu / identity; mean, scale, beta [64] fp32, 16-byte aligned. */
int ucd_stem_conv_pool(const float* x, long long sb, long long sc, long long sh, long long sw, int B, int H, int W, const void* w,
                       const float* mean, const float* scale, const float* beta, int act, float slope, void* out, ucd_stream_t stream);
int ucd_stem_pooled_size(int n);
size_t ucd_stem_pool_workspace_bytes(int C);
int ucd_stem_apply_pool(const void* z, int B, int H, int W, int C, const float* mean, const float* scale, const float* beta, int act,
                        float slope, void* out, uint8_t* idx, ucd_stream_t stream);
int ucd_stem_pool_backward(const void* z, const void* dpool, const uint8_t* idx, int B, int H, int W, int C, const float* mean,
                           const float* invstd, const float* scale, const float* beta, const float* weight, float* sums, float count,
                           int act, float slope, void* dz, void* workspace, size_t workspace_bytes, int phase, ucd_stream_t stream);

/* Weight gradient of a stride-1 convolution on channels-last maps (backward of modules/residual.py:57-73 conv1 / conv2 / conv3 /
 * proj_conv, modules/deeplab.py:24-37 map_convs / red_conv; replaces MIOpen's weight-gradient solvers and the batched split-M
 * library products on the train step):
 *     dw[n][t][k] = sum_m dz[m][n] * x[shift_t(m)][k]      taps = 1: the 1x1 product dz^T x;  taps = 9: the 3x3 convolution with
 *                                                          padding = dilation over the [B, H, W, K] map behind x (M = B*H*W)
 * dz [M][N] and x [M][K] bf16 row matrices (leading dimensions in elements), N and K multiples of 64.  The result is in the
 * weight's channels-last order [N][kh][kw][K]: dw (bf16, may be NULL) and / or dw32 (fp32, may be NULL; += when accumulate32,
 * e.g. straight into the fp32 gradient bucket).  workspace: ucd_conv_wgrad_workspace_bytes(M, N, K, taps) bytes of fp32 slabs
 * (one per row chunk), added in a fixed order (deterministic).  3x3 layers with 128-aligned channels, M >= 8192 and dilation <= 18
 * run the three-tap form (one kernel row per workgroup, round 4); UCD_WGRAD3=0 in the environment keeps the 9-tap form for them
 * (both are exact on integer operands and agree bit for bit there: tests/test_conv1x1_fused_gpu.py). */
size_t ucd_conv_wgrad_workspace_bytes(int M, int N, int K, int taps);
int ucd_conv_wgrad(const void* dz, int ld_dz, const void* x, int ld_x, int M, int N, int K, int taps, int H, int W, int dilation,
                   void* dw, float* dw32, int accumulate32, void* workspace, size_t workspace_bytes, ucd_stream_t stream);
/* The same for the strided layers (modules/residual.py:57-73 conv2 with stride 2, :79 proj_conv 1x1 stride 2): x is the
 * [B, H, W, K] input map, dz the [M = B*OH*OW][N] output gradient with OH = (H - 1) / stride + 1; N and K multiples of 128.
 * stride <= 1 is ucd_conv_wgrad. */
int ucd_conv_wgrad_strided(const void* dz, int ld_dz, const void* x, int ld_x, int M, int N, int K, int taps, int H, int W,
                           int dilation, int stride, void* dw, float* dw32, int accumulate32, void* workspace, size_t workspace_bytes,
                           ucd_stream_t stream);
/* Round 6: the slab sum of a weight gradient may be DEFERRED into the next weight-gradient launch on the same stream (it rides as
 * extra workgroups behind that product's own: no launch, no kernel boundary; bit-identical result).  ucd_conv_wgrad_ex = the call
 * above plus flags: bit 0 = the caller does not read dw / dw32, nor reuses `workspace`, before the next ucd_conv_wgrad* call on
 * this stream or ucd_conv_wgrad_flush(stream).  Deferral happens only while ucd_conv_wgrad_defer(1) is in force (returns the
 * previous setting; the gradient-bucket wrapper switches it on for its backward passes and flushes in front of its bucket copies);
 * ucd_conv_wgrad_flush launches the stream's pending sum, ucd_conv_wgrad_drop forgets it (an aborted backward).  The autograd
 * nodes of this library replace one launch per convolution of the reference's backward (modules/residual.py:67-73) this way.
 *
 * SIDE STREAM (round 6, mode bit 1 / flags bit 1): nothing in a backward pass waits for a weight gradient before the optimiser, yet on
 * the caller's stream each one sits in the chain of input-gradient products - ~190 of the ~850 dependent launches of a step, which at
 * 3 - 6 images per GPU fill a quarter of the chip each.  Under ucd_conv_wgrad_defer(mode) with mode & 2, a call with flags & 2 runs
 * on a stream owned by the library (lowest priority; UCD_WGRAD_STREAM_PRIO=normal: the default priority), forked behind `stream`
 * and joined back into `stream` by ucd_conv_wgrad_flush(stream) / ucd_conv_wgrad_drop(stream); under stream capture fork and join
 * become the graph's edges.  An accepted call records its fork point on `stream` at once, its launches go out at the NEXT accepted
 * call or at the flush: in a replayed graph the branch created first keeps the fork point's hardware queue and the other one hops
 * behind a cross-queue signal - launched at the fork, the weight gradients make the CALLER's chain the one that hops, ~8 us per
 * call (UCD_WGRAD_STREAM_LATE=0; UCD_WGRAD_STREAM_GROUP=n: n calls per fork point).  flags & 2 promises: dz, x, dw / dw32 and `workspace` stay allocated and untouched by other
 * streams until that flush, `workspace` is not the one of a call without the flag, and dw is not read before the flush; an error
 * of a launch made later is returned by the call that triggers it.  Results are the same bits either way (the same kernels on the
 * same operands).
 * ucd_conv_wgrad_defer(mode): bit 0 deferral, bit 1 side stream; returns the previous mode.  ucd_conv_wgrad_mode(): the mode. */
int ucd_conv_wgrad_ex(const void* dz, int ld_dz, const void* x, int ld_x, int M, int N, int K, int taps, int H, int W, int dilation,
                      int stride, void* dw, float* dw32, int accumulate32, void* workspace, size_t workspace_bytes, int flags,
                      ucd_stream_t stream);
int ucd_conv_wgrad_defer(int mode);
int ucd_conv_wgrad_mode(void);
int ucd_conv_wgrad_flush(ucd_stream_t stream);
int ucd_conv_wgrad_drop(ucd_stream_t stream);

/* The weights of the input-gradient convolutions of ALL stride-1 layers in one launch: for table entry e = {src offset,
 * dst offset, Co, Ci, KH*KW} (elements into the flat bf16 buffers; 4-D weights in channels-last memory order
 * [out][kh][kw][in]), dst_e = src_e.flip(2, 3).transpose(0, 1) in the same memory order.  blocks [n_blocks][4] (device,
 * int32) = {entry, spatial tap, out-channel tile of 32, in-channel tile of 32}, entries [n][5] (device, int64).  Replaces the
 * per-layer flip + copy of the "input gradient on the forward solver" trick (ucd_amd/blocks.py::_StrideOneConvFn). */
int ucd_flip_weights_batched(const void* src_flat, void* dst_flat, const int* blocks, int n_blocks, const long long* entries,
                             ucd_stream_t stream);
/* The same with blocks = {entry, spatial tap, out-channel tile of 64, in-channel tile of 64}: 16-byte accesses on both sides. */
int ucd_flip_weights_batched64(const void* src_flat, void* dst_flat, const int* blocks, int n_blocks, const long long* entries,
                               ucd_stream_t stream);

/* dst[cols, rows] = src[rows, cols]^T (bf16): the [K, N] weight of the input-gradient product. */
int ucd_transpose_bf16(const void* src, int rows, int cols, void* dst, ucd_stream_t stream);

/* ---- optimiser step -------------------------------------------------------------------------------------------------
 * Replaces `optim.step()` (train.py:147) of the reference's torch.optim.SGD(params, lr, momentum=0.9, nesterov=True) with
 * per-group weight decay (run.py:175-186) by ONE launch over every parameter tensor; the bf16 working copy of a
 * convolution weight (apex AMP O1's per-call cast, run.py:199-200) is written in the same pass.
 *   g' = g + wd p;  m' = mu m + g';  d = g' + mu m' (nesterov) | m';  p' = p - lr d;  w16 = bf16(p')
 * (double intermediates like torch's fused kernel; dampening 0; a zero-initialised m makes the first step torch's
 * "momentum buffer = clone of the gradient").  table [n] (device): one entry per tensor - p, g, m fp32 with identical
 * dense layouts of n elements, m NULL for momentum 0, w16 NULL or the bf16 copy of the same layout, group = index into
 * the hyper-parameter arrays.  blocks [n_blocks][2] (device, int32) = {table entry, chunk of ucd_sgd_chunk() elements}. */
#define UCD_SGD_MAX_GROUPS 8
typedef struct ucd_sgd_tensor {
  float* p;  const float* g;  float* m;  void* w16;
  long long n;
  int group;  int pad;
} ucd_sgd_tensor;
typedef struct ucd_sgd_hyper {
  double lr[UCD_SGD_MAX_GROUPS], momentum[UCD_SGD_MAX_GROUPS], weight_decay[UCD_SGD_MAX_GROUPS];
  int nesterov[UCD_SGD_MAX_GROUPS];
} ucd_sgd_hyper;
int ucd_sgd_chunk(void);
int ucd_sgd_step(const ucd_sgd_tensor* table, const int* blocks, int n_blocks, const ucd_sgd_hyper* hyper, ucd_stream_t stream);
/* The same step with the hyper-parameters in DEVICE memory (device_hyper): the form a captured hipGraph of the whole training
 * iteration (train.py:95-151) replays while PolyLR changes the learning rate every iteration (train.py:150-151) - a by-value
 * kernel argument would be frozen into the graph.  ucd_sgd_hyper_store writes host values into that device struct in stream
 * order (the values travel as a kernel argument: no pinned staging buffer that a later step could overwrite early). */
int ucd_sgd_hyper_store(ucd_sgd_hyper* device_hyper, const ucd_sgd_hyper* hyper, ucd_stream_t stream);
int ucd_sgd_step_dev(const ucd_sgd_tensor* table, const int* blocks, int n_blocks, const ucd_sgd_hyper* device_hyper,
                     ucd_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* UCD_HIP_H */
