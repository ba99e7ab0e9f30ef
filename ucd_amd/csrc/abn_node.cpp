// Autograd node of the training-mode ABN layer in C++ (PyTorch extension, host code only).
//
// The kernels live behind the C ABI (include/ucd_hip.h); what this file replaces is the Python wrapper around them on
// the student's 106 training-mode layers: a Python autograd.Function costs ~38 us forward and ~33 us backward of host
// time per layer (tools/host_profile.py), which is what bounds the step once a GPU holds 3-6 images (4-8 GPU runs).
// Here the forward is one pybind call (~8 us) and the backward never enters Python.  It mirrors
// ucd_amd/abn.py::_ABNFunction for the case it is used for - batch statistics, dense channels-last input, optional
// fused residual, optional SyncBN through the library-owned RCCL communicator - and the Python class stays the
// reference implementation for everything else (eval mode, plane bias, slice outputs, torch.distributed fallback).
//
// The reference reaches the same work through inplace_abn's autograd Functions (segmentation_module.py:15-20).
#include <torch/extension.h>
#include <torch/csrc/autograd/engine.h>

#include <cstring>
#include <map>
#include <mutex>

#include "../../include/ucd_hip.h"

namespace {

using torch::autograd::AutogradContext;
using torch::autograd::variable_list;

void check(int rc, const char* what) {
  TORCH_CHECK(rc == 0, what, " failed (code ", rc, "): ", ucd_last_error());
}

// grow-only scratch per (device, stream); calls on one stream are ordered
std::mutex g_mu;
std::map<std::pair<int, int64_t>, at::Tensor> g_ws;

// Tensors that kernels on the library's SIDE stream (weight gradients, include/ucd_hip.h ucd_conv_wgrad_ex flags & 2) still read or
// write: the caching allocator knows nothing of that stream, so the operands stay referenced here until the join
// (wgrad_side_release, called behind ucd_conv_wgrad_flush / _drop) - a block freed on the compute stream before that could be handed
// to a later kernel of that stream while the side stream is still reading it.
std::vector<at::Tensor> g_side_hold;
constexpr int kWsTags = 8;     // tags 0 / 1 scratch of the nodes, 2 / 3 slab workspaces in turn, 4 / 5 the same for side-stream calls

void* workspace(const at::Tensor& like, size_t bytes, int64_t stream, int tag = 0) {
  std::lock_guard<std::mutex> lock(g_mu);
  auto key = std::make_pair((int)like.get_device() * kWsTags + tag, stream);
  auto it = g_ws.find(key);
  if (it == g_ws.end() || (size_t)it->second.numel() < bytes) {
    size_t n = bytes < ((size_t)1 << 20) ? ((size_t)1 << 20) : bytes;
    if (it != g_ws.end() && tag >= 4) g_side_hold.push_back(it->second);     // the side stream may still be in the old buffer
    g_ws[key] = at::empty({(int64_t)n}, like.options().dtype(at::kByte));
    it = g_ws.find(key);
  }
  return it->second.data_ptr();
}

int64_t g_pass_flushes = 0;      // test hook: end-of-pass flushes run so far
bool g_pass_cb_queued = false;   // under g_mu: this backward pass already has its end-of-pass flush queued

void wgrad_side_release() {
  std::vector<at::Tensor> gone;
  {
    std::lock_guard<std::mutex> lock(g_mu);
    gone.swap(g_side_hold);
    g_pass_cb_queued = false;      // (a pass that died before its callbacks ran must not leave the flag behind)
  }
}   // the tensors are released outside the lock

// A weight-gradient call that may be deferred / moved to the side stream is final only behind ucd_conv_wgrad_flush.  The
// gradient-bucket wrapper flushes in front of its copies - but a pass it does not see to its end (torch.autograd.grad under the
// wrapper: no AccumulateGrad, no hook) would hand out gradients whose last slab sum never ran.  So the first such call of a backward
// pass queues the flush as an engine callback: it runs when the pass ends, whoever started it, in stream order before the caller
// gets its gradients (a second flush by the wrapper is a no-op).
void queue_end_of_pass_flush(int64_t stream) {
  {
    std::lock_guard<std::mutex> lock(g_mu);
    if (g_pass_cb_queued) return;
    g_pass_cb_queued = true;
  }
  torch::autograd::Engine::get_default_engine().queue_callback([stream]() {
    ++g_pass_flushes;
    const int rc = ucd_conv_wgrad_flush((ucd_stream_t)stream);
    wgrad_side_release();
    TORCH_CHECK(rc == 0, "ucd_conv_wgrad_flush failed at the end of the backward pass (code ", rc, "): ", ucd_last_error());
  });
}

// test hook (tests/diag/poison_step_diag.py): fill every cached scratch buffer with a byte pattern - a kernel that reads scratch it has
// not written shows up as a changed (or NaN) result
void poison_workspaces(int64_t byte, int64_t tag) {
  std::lock_guard<std::mutex> lock(g_mu);
  for (auto& kv : g_ws)
    if (tag < 0 || kv.first.first % kWsTags == tag) kv.second.fill_(byte);
}

// ---- statistics arena (round 5) ----------------------------------------------------------------------------------------------------
// The per-layer [2 C] accumulators that the GEMM epilogues add their column sums to with fp32 atomics (ucd_conv1x1 stat_acc) live in
// ONE zero-filled buffer per device: a slot is handed out per layer and direction during the forward, and stat_arena_reset - called
// once per training step (ucd_amd/train.py, ucd_amd/ddp.py) - zeroes the used range with ONE fill and starts over.  Correctness never
// depends on the caller: every slot carries the arena's generation, a reset invalidates the slots of older generations (their
// consumers fall back to the reduction kernels), and a full arena resets itself.
struct StatArena {
  at::Tensor buf;
  int64_t off = 0, gen = 1;
  int64_t high = 0;            // high-water mark of `off` over all generations: what a reset zeroes
};
std::map<int, StatArena> g_arena;
constexpr int64_t kArenaFloats = 16 << 20;     // 64 MiB: about four steps' worth at the benchmark's shapes (replicated slots, ~15 MB per step)

int64_t arena_gen(int dev) {
  std::lock_guard<std::mutex> lock(g_mu);
  return g_arena[dev].gen;
}

void stat_arena_reset(int64_t dev, int64_t stream) {
  std::lock_guard<std::mutex> lock(g_mu);
  auto it = g_arena.find((int)dev);
  if (it == g_arena.end() || !it->second.buf.defined()) return;
  StatArena& a = it->second;
  // one fill in stream order (a memset node in a captured step).  A raw memset, not Tensor::zero_(): slots handed out earlier are
  // views of this buffer saved by autograd nodes, and an in-place ATen op would move their shared version counter - a backward that
  // follows a reset must reach the generation check (and fall back), not autograd's "modified by an inplace operation"
  // up to the HIGH-WATER mark, not the last step's footprint: a captured step graph freezes this range into its memset
  // node - had the eager step in front of the capture used fewer floats than the captured one (another batch shape: the replica
  // count of a slot depends on M), the slots beyond would never be zeroed on replay and their atomic sums would grow across replays
  if (a.off > a.high) a.high = a.off;
  if (a.high > 0) check(ucd_fill_zero(a.buf.data_ptr(), (size_t)a.high * sizeof(float), (ucd_stream_t)stream), "ucd_fill_zero");
  a.off = 0;
  a.gen += 1;
}

// n floats (rounded to 64) of zeros, or an undefined tensor (arena switched off by the caller / request larger than the arena)
at::Tensor arena_alloc(const at::Tensor& like, int64_t n, int64_t* gen, int64_t stream) {
  const int dev = (int)like.get_device();
  n = (n + 63) / 64 * 64;
  {
    std::lock_guard<std::mutex> lock(g_mu);
    StatArena& a = g_arena[dev];
    if (!a.buf.defined()) a.buf = at::zeros({kArenaFloats}, like.options().dtype(at::kFloat));
    if (n > kArenaFloats) return at::Tensor();
    if (a.off + n <= kArenaFloats) {
      at::Tensor t = a.buf.narrow(0, a.off, n);
      a.off += n;
      *gen = a.gen;
      return t;
    }
  }
  stat_arena_reset(dev, stream);               // full (nobody resets per step): zero it, invalidate the outstanding slots
  std::lock_guard<std::mutex> lock(g_mu);
  StatArena& a = g_arena[dev];
  at::Tensor t = a.buf.narrow(0, 0, n);
  a.off = n;
  *gen = a.gen;
  return t;
}

bool dense_channels_last(const at::Tensor& x) {
  if (x.dim() != 4) return false;
  const auto B = x.size(0), C = x.size(1), H = x.size(2), W = x.size(3);
  (void)B;
  return H > 1 && W > 1 && x.stride(1) == 1 && x.stride(3) == C && x.stride(2) == W * C && x.stride(0) == H * W * C &&
         (reinterpret_cast<uintptr_t>(x.data_ptr()) & 15) == 0 && ((C * x.element_size()) & 15) == 0;
}

int dtype_code(const at::Tensor& x) {
  if (x.scalar_type() == at::kBFloat16) return UCD_BF16;
  TORCH_CHECK(x.scalar_type() == at::kFloat, "ucd abn node: bf16 or fp32 activations only");
  return UCD_F32;
}

const float* fptr(const at::Tensor& t) { return t.defined() ? t.data_ptr<float>() : nullptr; }

// Weight gradient of a stride-1 convolution on the own kernel (ucd_conv_wgrad, csrc/wgrad.hip): dz [B, N, H, W] and x [B, K, H, W]
// dense channels-last bf16 -> a tensor with the weight's sizes and (channels-last) strides.  dilation 0: 1x1.
bool own_wgrad_ok(const at::Tensor& dz, const at::Tensor& x, const at::Tensor& w4, int64_t dilation) {
  return dz.defined() && dense_channels_last(dz) && dense_channels_last(x) && dz.scalar_type() == at::kBFloat16 &&
         x.scalar_type() == at::kBFloat16 && w4.size(0) % 64 == 0 && w4.size(1) % 64 == 0 && dz.size(1) == w4.size(0) &&
         x.size(1) == w4.size(1) && x.size(0) * x.size(2) * x.size(3) < (1 << 22) &&
         (dilation == 0 ? (w4.size(2) == 1 && w4.size(3) == 1)
                        // every stride-1 3x3 layer (tools/wgrad_probe2.py, profiles/r03_wgrad_probe.txt: 64-256 channel layers
                        // 50-58 vs 80-100 us for MIOpen's solver, 512 -> 512 213 vs 215, the ASPP branches 392 vs 372 - plus the
                        // ~35 us of zero-fill / cast kernels MIOpen wraps around its solver on the step)
                        : (w4.size(2) == 3 && w4.size(3) == 3));
}

// stride > 1 (the first block of a stage: conv2 3x3 / proj_conv 1x1 with stride 2): dz is the smaller output map
// Slab workspace of a weight-gradient call.  The calls of the autograd nodes allow the library to DEFER their slab sum into the next
// weight-gradient launch of the stream (ucd_conv_wgrad_ex, flags bit 0: the gradient goes to AccumulateGrad and is first read by the
// bucket copies, in front of which ucd_amd/ddp.py flushes) - so a call's slabs must outlive the NEXT call: two scratch buffers, taken
// in turn.
void* wgrad_workspace(const at::Tensor& like, size_t bytes, int64_t stream, bool side) {
  static std::map<std::pair<int, int64_t>, int> turn;   // per (device, stream, side): the pending sum is per stream too
  int tag;
  {
    std::lock_guard<std::mutex> lock(g_mu);
    tag = (side ? 4 : 2) + (turn[std::make_pair((int)like.get_device() * 2 + (side ? 1 : 0), stream)] ^= 1);
  }
  return workspace(like, bytes, stream, tag);
}

// flags of a node's weight-gradient call (ucd_conv_wgrad_ex): bit 0 the slab sum may wait, bit 1 the call may leave the compute
// stream - both only for a gradient nothing reads before the wrapper's flush (sum_may_wait below); with bit 1 in force the operands
// are held until that flush
int sum_may_wait(const at::Tensor& w4);
int wgrad_flags(const at::Tensor& w4, bool* side) {
  *side = false;
  if (!sum_may_wait(w4)) return 0;
  *side = (ucd_conv_wgrad_mode() & 2) != 0;
  return *side ? 3 : 1;
}
void hold_for_side(std::initializer_list<at::Tensor> ts) {
  std::lock_guard<std::mutex> lock(g_mu);
  for (const auto& t : ts) g_side_hold.push_back(t);
}

// May the slab sum of this weight gradient wait for the next weight-gradient launch?  Only when nothing reads the gradient before the
// wrapper's flush: the weight is a LEAF (the flat bf16 working copy of ucd_amd/master.py: AccumulateGrad adopts the tensor without
// touching it - the layout contract is met, dw comes out in the weight's own strides).  A weight that is the output of autocast's
// per-call cast has a ToCopyBackward node behind it that reads the gradient at once.
int sum_may_wait(const at::Tensor& w4) { return w4.defined() && w4.is_leaf() && w4.requires_grad() ? 1 : 0; }

at::Tensor own_wgrad(const at::Tensor& dz, const at::Tensor& x, const at::Tensor& w4, int64_t dilation, int64_t stream,
                     int64_t stride = 1) {
  const int64_t B = x.size(0), K = x.size(1), H = x.size(2), W = x.size(3), N = w4.size(0), M = B * dz.size(2) * dz.size(3);
  const int taps = dilation > 0 ? 9 : 1;
  at::Tensor dw = at::empty({N, taps == 9 ? 3 : 1, taps == 9 ? 3 : 1, K}, x.options().memory_format(c10::nullopt));
  const size_t wsb = ucd_conv_wgrad_workspace_bytes((int)M, (int)N, (int)K, taps);
  bool side;
  const int flags = wgrad_flags(w4, &side);
  if (side) hold_for_side({dz, x, dw});
  if (flags && ucd_conv_wgrad_mode()) queue_end_of_pass_flush(stream);
  check(ucd_conv_wgrad_ex(dz.data_ptr(), (int)N, x.data_ptr(), (int)K, (int)M, (int)N, (int)K, taps, (int)H, (int)W,
                          (int)(dilation > 0 ? dilation : 1), (int)stride, dw.data_ptr(), nullptr, 0,
                          wgrad_workspace(x, wsb, stream, side), wsb, flags, (ucd_stream_t)stream),
        "ucd_conv_wgrad");
  return dw.permute({0, 3, 1, 2});      // [N, K, kh, kw] with channels-last strides: the weight's own memory order
}

class ABNTrainNode : public torch::autograd::Function<ABNTrainNode> {
 public:
  static at::Tensor forward(AutogradContext* ctx, at::Tensor x, at::Tensor weight, at::Tensor bias,
                            c10::optional<at::Tensor> residual_, at::Tensor running_mean, at::Tensor running_var,
                            double momentum, double eps, int64_t act, double slope, int64_t comm, int64_t world,
                            int64_t stream, int64_t param_grad) {
    // param_grad != 0: address of this layer's [d bias | d weight] storage (2*C floats, the parameters' .grad views laid
    // out back to back by ucd_amd.ddp); the backward kernels then write the parameter gradients there themselves and the
    // node reports none - no gradient tensors, no AccumulateGrad adds (212 tiny launches per step)
    TORCH_CHECK(dense_channels_last(x), "ucd abn node: input must be a dense channels-last tensor");
    at::Tensor residual = residual_.has_value() ? *residual_ : at::Tensor();
    const bool has_res = residual.defined();
    if (has_res) TORCH_CHECK(dense_channels_last(residual) && residual.sizes() == x.sizes() && residual.scalar_type() == x.scalar_type(),
                             "ucd abn node: residual must match the input");
    const int64_t C = x.size(1), HW = x.size(2) * x.size(3), M = x.size(0) * HW;
    const bool sync = comm != 0;
    at::Tensor y = at::empty_like(x);   // preserves channels-last
    at::Tensor buf = at::empty({(sync ? 8 + 2 * world : 6) * C}, x.options().dtype(at::kFloat));
    const size_t ws_bytes = ucd_abn_workspace_bytes((int)M, (int)C);
    void* ws = workspace(x, ws_bytes, stream);
    const int dt = dtype_code(x);
    if (sync) {
      check(ucd_abn_sync_forward_comm((ucd_comm_t)comm, (int)world, x.data_ptr(), (int)C, y.data_ptr(), (int)C,
                                      has_res ? residual.data_ptr() : nullptr, has_res ? (int)C : 0, dt, (int)M, (int)C, nullptr,
                                      (int)HW, fptr(weight), fptr(bias), running_mean.data_ptr<float>(),
                                      running_var.data_ptr<float>(), (float)momentum, (float)eps, buf.data_ptr<float>(), (int)act,
                                      (float)slope, ws, ws_bytes, (ucd_stream_t)stream),
            "ucd_abn_sync_forward_comm");
    } else {
      check(ucd_abn_forward(x.data_ptr(), (int)C, y.data_ptr(), (int)C, has_res ? residual.data_ptr() : nullptr,
                            has_res ? (int)C : 0, dt, (int)M, (int)C, nullptr, (int)HW, fptr(weight), fptr(bias),
                            running_mean.data_ptr<float>(), running_var.data_ptr<float>(), (float)momentum, (float)eps, 1,
                            buf.data_ptr<float>(), nullptr, (int)act, (float)slope, ws, ws_bytes, (ucd_stream_t)stream),
            "ucd_abn_forward");
    }
    const bool needs_y = has_res && (act & UCD_ACT_MASK) != UCD_ACT_IDENTITY;  // the sign of z is not recoverable from x alone
    ctx->save_for_backward({x, needs_y ? y : at::Tensor(), weight, bias, buf});
    ctx->saved_data["act"] = act;
    ctx->saved_data["slope"] = slope;
    ctx->saved_data["comm"] = comm;
    ctx->saved_data["world"] = world;
    ctx->saved_data["stream"] = stream;
    ctx->saved_data["has_res"] = has_res;
    ctx->saved_data["param_grad"] = param_grad;
    return y;
  }

  static variable_list backward(AutogradContext* ctx, variable_list grads) {
    auto saved = ctx->get_saved_variables();
    at::Tensor x = saved[0], y = saved[1], weight = saved[2], bias = saved[3], buf = saved[4];
    const int64_t act = ctx->saved_data["act"].toInt(), comm = ctx->saved_data["comm"].toInt();
    const int64_t world = ctx->saved_data["world"].toInt(), stream = ctx->saved_data["stream"].toInt();
    const double slope = ctx->saved_data["slope"].toDouble();
    const bool has_res = ctx->saved_data["has_res"].toBool();
    float* param_grad = reinterpret_cast<float*>(ctx->saved_data["param_grad"].toInt());
    const int64_t C = x.size(1), HW = x.size(2) * x.size(3), M = x.size(0) * HW;
    at::Tensor dy = grads[0];
    if (dy.scalar_type() != x.scalar_type()) dy = dy.to(x.scalar_type());
    if (!dense_channels_last(dy)) dy = dy.contiguous(at::MemoryFormat::ChannelsLast);
    at::Tensor dx = at::empty_like(x);
    at::Tensor dz = has_res ? at::empty_like(x) : at::Tensor();
    const bool sync = comm != 0;
    at::Tensor sums;
    if (sync || !param_grad) sums = at::empty({(sync && !param_grad ? 4 : 2) * C}, x.options().dtype(at::kFloat));
    float* b = buf.data_ptr<float>();
    const float *mean = b + 3 * C, *invstd = b + 4 * C, *scale = b + 5 * C;
    const size_t ws_bytes = ucd_abn_workspace_bytes((int)M, (int)C);
    void* ws = workspace(x, ws_bytes, stream);
    const int dt = dtype_code(x);
    const void* yp = y.defined() ? y.data_ptr() : nullptr;
    at::Tensor dbias, dweight;
    if (sync) {
      check(ucd_abn_sync_backward_comm((ucd_comm_t)comm, (int)world, x.data_ptr(), (int)C, dy.data_ptr(), (int)C, yp, yp ? (int)C : 0,
                                       dx.data_ptr(), (int)C, has_res ? dz.data_ptr() : nullptr, has_res ? (int)C : 0, dt, (int)M,
                                       (int)C, nullptr, (int)HW, mean, invstd, scale, fptr(bias), fptr(weight), sums.data_ptr<float>(),
                                       param_grad ? param_grad : sums.data_ptr<float>() + 2 * C, (int)act, (float)slope, ws,
                                       ws_bytes, (ucd_stream_t)stream),
            "ucd_abn_sync_backward_comm");
      if (!param_grad) {
        dbias = sums.narrow(0, 2 * C, C);
        dweight = sums.narrow(0, 3 * C, C);
      }
    } else {
      check(ucd_abn_backward(x.data_ptr(), (int)C, dy.data_ptr(), (int)C, yp, yp ? (int)C : 0, dx.data_ptr(), (int)C,
                             has_res ? dz.data_ptr() : nullptr, has_res ? (int)C : 0, dt, (int)M, (int)C, nullptr, (int)HW, mean,
                             invstd, scale, fptr(bias), fptr(weight), param_grad ? param_grad : sums.data_ptr<float>(), (float)M,
                             1, 1, (int)act, (float)slope, ws, ws_bytes, (ucd_stream_t)stream),
            "ucd_abn_backward");
      if (!param_grad) {
        dbias = sums.narrow(0, 0, C);
        dweight = sums.narrow(0, C, C);
      }
    }
    at::Tensor none;
    return {dx, dweight, dbias, has_res ? dz : none, none, none, none, none, none, none, none, none, none, none};
  }
};

// dw[Co, Ci] = dy[M, Co]^T rows[M, Ci] on the own kernel (2-D row matrices, contiguous bf16)
bool own_wgrad_rows_ok(const at::Tensor& dy, const at::Tensor& rows) {
  return dy.dim() == 2 && rows.dim() == 2 && dy.is_contiguous() && rows.is_contiguous() && dy.scalar_type() == at::kBFloat16 &&
         rows.scalar_type() == at::kBFloat16 && dy.size(1) % 64 == 0 && rows.size(1) % 64 == 0 && dy.size(0) < (1 << 22) &&
         (reinterpret_cast<uintptr_t>(dy.data_ptr()) & 15) == 0 && (reinterpret_cast<uintptr_t>(rows.data_ptr()) & 15) == 0;
}

at::Tensor own_wgrad_rows(const at::Tensor& dy, const at::Tensor& rows, int64_t stream, const at::Tensor& w4) {
  const int64_t M = rows.size(0), Ci = rows.size(1), Co = dy.size(1);
  at::Tensor dw = at::empty({Co, Ci}, rows.options());
  const size_t wsb = ucd_conv_wgrad_workspace_bytes((int)M, (int)Co, (int)Ci, 1);
  bool side;
  const int flags = wgrad_flags(w4, &side);
  if (side) hold_for_side({dy, rows, dw});
  if (flags && ucd_conv_wgrad_mode()) queue_end_of_pass_flush(stream);
  check(ucd_conv_wgrad_ex(dy.data_ptr(), (int)Co, rows.data_ptr(), (int)Ci, (int)M, (int)Co, (int)Ci, 1, 0, 0, 1, 1, dw.data_ptr(), nullptr, 0,
                          wgrad_workspace(rows, wsb, stream, side), wsb, flags, (ucd_stream_t)stream),
        "ucd_conv_wgrad");
  return dw;
}

// ---- wide 1x1 convolution as a row-matrix GEMM (ucd_amd/blocks.py::_Gemm1x1 is the Python twin) ------------------
// y[M, Co] = rows[M, Ci] . w[Co, Ci]^T through ucd_gemm_bf16 (hipBLASLt, tuned once per shape); the weight gradient of a
// long M is eight batched K-chunks + a sum (41 us against 96 for the best single-kernel candidate at M = 26136).
int64_t wgrad_split(int64_t M) {
  if (M >= 8192)
    for (int64_t S : {8, 4, 12, 6, 3, 2})
      if (M % S == 0) return S;
  return 1;
}

class Gemm1x1Node : public torch::autograd::Function<Gemm1x1Node> {
 public:
  static at::Tensor forward(AutogradContext* ctx, at::Tensor rows, at::Tensor w4, int64_t stream, bool own_wg) {
    TORCH_CHECK(rows.dim() == 2 && rows.is_contiguous() && rows.scalar_type() == at::kBFloat16 && w4.dim() == 4 &&
                    w4.scalar_type() == at::kBFloat16 && w4.size(1) == rows.size(1) && w4.size(2) == 1 && w4.size(3) == 1,
                "ucd gemm1x1 node: rows [M, Ci] bf16 contiguous and weight [Co, Ci, 1, 1] bf16 expected");
    ctx->saved_data["own_wgrad"] = own_wg;
    const int64_t M = rows.size(0), Ci = rows.size(1), Co = w4.size(0);
    at::Tensor y = at::empty({M, Co}, rows.options());
    const size_t wsb = ucd_gemm_workspace_bytes();
    check(ucd_gemm_bf16(0, (int)M, (int)Co, (int)Ci, rows.data_ptr(), (int)Ci, w4.data_ptr(), (int)Ci, y.data_ptr(), (int)Co,
                        workspace(rows, wsb, stream, 1), wsb, 1, (ucd_stream_t)stream),
          "ucd_gemm_bf16");
    ctx->save_for_backward({rows, w4});
    ctx->saved_data["stream"] = stream;
    return y;
  }

  static variable_list backward(AutogradContext* ctx, variable_list grads) {
    auto saved = ctx->get_saved_variables();
    at::Tensor rows = saved[0], w4 = saved[1];
    const int64_t stream = ctx->saved_data["stream"].toInt();
    const int64_t M = rows.size(0), Ci = rows.size(1), Co = w4.size(0);
    at::Tensor dy = grads[0].contiguous();
    if (dy.scalar_type() != at::kBFloat16) dy = dy.to(at::kBFloat16);
    const size_t wsb = ucd_gemm_workspace_bytes();
    void* ws = workspace(rows, wsb, stream, 1);
    at::Tensor dx, dw;
    if (ctx->needs_input_grad(0)) {
      dx = at::empty_like(rows);
      check(ucd_gemm_bf16(1, (int)M, (int)Ci, (int)Co, dy.data_ptr(), (int)Co, w4.data_ptr(), (int)Ci, dx.data_ptr(), (int)Ci, ws,
                          wsb, 1, (ucd_stream_t)stream),
            "ucd_gemm_bf16");
    }
    if (ctx->needs_input_grad(1)) {
      const int64_t S = wgrad_split(M);
      if (ctx->saved_data["own_wgrad"].toBool() && own_wgrad_rows_ok(dy, rows)) {
        dw = own_wgrad_rows(dy, rows, stream, w4);
      } else if (S > 1) {
        dw = at::bmm(dy.view({S, M / S, Co}).transpose(1, 2), rows.view({S, M / S, Ci})).sum(0);
      } else {
        dw = at::empty({Co, Ci}, rows.options());
        check(ucd_gemm_bf16(2, (int)Co, (int)Ci, (int)M, dy.data_ptr(), (int)Co, rows.data_ptr(), (int)Ci, dw.data_ptr(), (int)Ci,
                            ws, wsb, 1, (ucd_stream_t)stream),
              "ucd_gemm_bf16");
      }
      dw = dw.as_strided(w4.sizes(), w4.strides());   // [Co, Ci, 1, 1] is one memory order in either format
    }
    return {dx, dw, at::Tensor(), at::Tensor()};
  }
};

at::Tensor gemm1x1(at::Tensor rows, at::Tensor w4, int64_t stream, bool own_wg) { return Gemm1x1Node::apply(rows, w4, stream, own_wg); }

// First 1x1 convolution of an identity-shortcut bottleneck block, together with the shortcut itself: returns (y, rows) so
// that the block input has ONE consumer.  The backward then receives the shortcut's gradient next to dy and folds it
// into the input-gradient GEMM, dx = dskip + dy . w (beta = 1, in place on dskip), instead of autograd adding two
// [M, Ci] tensors afterwards (one 3-pass elementwise kernel per block, 24 blocks).
class Gemm1x1SkipNode : public torch::autograd::Function<Gemm1x1SkipNode> {
 public:
  static variable_list forward(AutogradContext* ctx, at::Tensor rows, at::Tensor w4, int64_t stream, bool own_wg) {
    TORCH_CHECK(rows.dim() == 2 && rows.is_contiguous() && rows.scalar_type() == at::kBFloat16 && w4.dim() == 4 &&
                    w4.scalar_type() == at::kBFloat16 && w4.size(1) == rows.size(1) && w4.size(2) == 1 && w4.size(3) == 1,
                "ucd gemm1x1 skip node: rows [M, Ci] bf16 contiguous and weight [Co, Ci, 1, 1] bf16 expected");
    ctx->saved_data["own_wgrad"] = own_wg;
    const int64_t M = rows.size(0), Ci = rows.size(1), Co = w4.size(0);
    at::Tensor y = at::empty({M, Co}, rows.options());
    const size_t wsb = ucd_gemm_workspace_bytes();
    check(ucd_gemm_bf16(0, (int)M, (int)Co, (int)Ci, rows.data_ptr(), (int)Ci, w4.data_ptr(), (int)Ci, y.data_ptr(), (int)Co,
                        workspace(rows, wsb, stream, 1), wsb, 1, (ucd_stream_t)stream),
          "ucd_gemm_bf16");
    ctx->save_for_backward({rows, w4});
    ctx->saved_data["stream"] = stream;
    return {y, rows};          // an input returned as an output: autograd hands out an alias with this node as grad_fn
  }

  static variable_list backward(AutogradContext* ctx, variable_list grads) {
    auto saved = ctx->get_saved_variables();
    at::Tensor rows = saved[0], w4 = saved[1];
    const int64_t stream = ctx->saved_data["stream"].toInt();
    const int64_t M = rows.size(0), Ci = rows.size(1), Co = w4.size(0);
    const size_t wsb = ucd_gemm_workspace_bytes();
    void* ws = workspace(rows, wsb, stream, 1);
    at::Tensor dy = grads[0], dskip = grads[1], dx, dw;
    if (dy.defined()) {
      dy = dy.contiguous();
      if (dy.scalar_type() != at::kBFloat16) dy = dy.to(at::kBFloat16);
    }
    if (ctx->needs_input_grad(0)) {
      if (dskip.defined() && dy.defined() && dskip.scalar_type() == at::kBFloat16) {
        dskip = dskip.contiguous();
        if (!ucd_gemm_has_plan(1, (int)M, (int)Ci, (int)Co, (int)Co, (int)Ci, (int)Ci)) {   // tune once, into scratch
          at::Tensor scratch = at::empty_like(rows);
          check(ucd_gemm_bf16(1, (int)M, (int)Ci, (int)Co, dy.data_ptr(), (int)Co, w4.data_ptr(), (int)Ci, scratch.data_ptr(),
                              (int)Ci, ws, wsb, 1, (ucd_stream_t)stream),
                "ucd_gemm_bf16");
        }
        check(ucd_gemm_bf16_acc(1, (int)M, (int)Ci, (int)Co, dy.data_ptr(), (int)Co, w4.data_ptr(), (int)Ci, dskip.data_ptr(),
                                (int)Ci, ws, wsb, (ucd_stream_t)stream),
              "ucd_gemm_bf16_acc");
        dx = dskip;
      } else if (dy.defined()) {
        dx = at::empty_like(rows);
        check(ucd_gemm_bf16(1, (int)M, (int)Ci, (int)Co, dy.data_ptr(), (int)Co, w4.data_ptr(), (int)Ci, dx.data_ptr(), (int)Ci, ws,
                            wsb, 1, (ucd_stream_t)stream),
              "ucd_gemm_bf16");
        if (dskip.defined()) dx = dx + dskip;
      } else {
        dx = dskip;
      }
    }
    if (ctx->needs_input_grad(1) && dy.defined()) {
      const int64_t S = wgrad_split(M);
      if (ctx->saved_data["own_wgrad"].toBool() && own_wgrad_rows_ok(dy, rows)) {
        dw = own_wgrad_rows(dy, rows, stream, w4);
      } else if (S > 1) {
        dw = at::bmm(dy.view({S, M / S, Co}).transpose(1, 2), rows.view({S, M / S, Ci})).sum(0);
      } else {
        dw = at::empty({Co, Ci}, rows.options());
        check(ucd_gemm_bf16(2, (int)Co, (int)Ci, (int)M, dy.data_ptr(), (int)Co, rows.data_ptr(), (int)Ci, dw.data_ptr(), (int)Ci,
                            ws, wsb, 1, (ucd_stream_t)stream),
              "ucd_gemm_bf16");
      }
      dw = dw.as_strided(w4.sizes(), w4.strides());
    }
    return {dx, dw, at::Tensor(), at::Tensor()};
  }
};

std::vector<at::Tensor> gemm1x1_skip(at::Tensor rows, at::Tensor w4, int64_t stream, bool own_wg) {
  return Gemm1x1SkipNode::apply(rows, w4, stream, own_wg);
}

// ---- stride-1 convolution (3x3 with padding = dilation, or 1x1) whose input gradient runs on the FORWARD solver --------
// dx = conv2d(dy, w.flip(2,3).transpose(0,1), padding, dilation): MIOpen's backward-data solvers are 1.3-2x slower than its
// forward solvers on these problems (tools/dgrad_probe.py, tools/dgrad1x1_probe.py).  ucd_amd/blocks.py::_StrideOneConvFn is
// the Python twin.
class StrideOneConvNode : public torch::autograd::Function<StrideOneConvNode> {
 public:
  // own_fwd / own_dgrad (3x3 only, dense channels-last bf16, 64-aligned channels): the implicit-GEMM kernel
  // (ucd_conv1x1, taps = 9) instead of MIOpen where it is the faster one (the ASPP branches: 303-315 vs 382-390 us)
  static at::Tensor own3x3(const at::Tensor& a, const at::Tensor& w, int64_t d, int64_t stream) {
    const int64_t B = a.size(0), K = a.size(1), H = a.size(2), W = a.size(3), N = w.size(0);
    at::Tensor y = at::empty({B, N, H, W}, a.options().memory_format(at::MemoryFormat::ChannelsLast));
    ucd_conv1x1_desc dsc;
    memset(&dsc, 0, sizeof(dsc));
    dsc.a = a.data_ptr(); dsc.lda = (int)K; dsc.w = w.data_ptr(); dsc.ldw = (int)(9 * K); dsc.y = y.data_ptr(); dsc.ldy = (int)N;
    dsc.M = (int)(B * H * W); dsc.N = (int)N; dsc.K = (int)K; dsc.out_mode = 0;
    dsc.taps = 9; dsc.H = (int)H; dsc.W = (int)W; dsc.dilation = (int)d;
    check(ucd_conv1x1(&dsc, (ucd_stream_t)stream), "ucd_conv1x1");
    return y;
  }

  static at::Tensor forward(AutogradContext* ctx, at::Tensor x, at::Tensor w, int64_t d, c10::optional<at::Tensor> wt_,
                            bool own_fwd, bool own_dgrad, int64_t stream, bool own_wg) {
    const int64_t pad = d * (w.size(2) / 2);
    // wt = w.flip(2, 3).transpose(0, 1) in channels-last order when the caller keeps it cached (ucd_amd/master.py: one
    // batched kernel per optimiser step instead of a flip + copy per layer and step)
    ctx->save_for_backward({x, w, wt_.has_value() ? *wt_ : at::Tensor()});
    ctx->saved_data["d"] = d;
    ctx->saved_data["own_dgrad"] = own_dgrad;
    ctx->saved_data["own_wgrad"] = own_wg;
    ctx->saved_data["stream"] = stream;
    if (own_fwd) return own3x3(x, w, d, stream);
    return at::conv2d(x, w, {}, {1, 1}, {pad, pad}, {d, d}, 1);
  }

  static variable_list backward(AutogradContext* ctx, variable_list grads) {
    auto saved = ctx->get_saved_variables();
    at::Tensor x = saved[0], w = saved[1], wt = saved[2], dy = grads[0];
    const int64_t d = ctx->saved_data["d"].toInt(), pad = d * (w.size(2) / 2);
    at::Tensor dx, dw;
    if (ctx->needs_input_grad(0)) {
      if (!wt.defined())
        wt = (w.size(2) == 1 ? w.transpose(0, 1) : w.flip({2, 3}).transpose(0, 1)).contiguous(at::MemoryFormat::ChannelsLast);
      if (ctx->saved_data["own_dgrad"].toBool() && dy.scalar_type() == at::kBFloat16) {
        if (!dense_channels_last(dy)) dy = dy.contiguous(at::MemoryFormat::ChannelsLast);
        dx = own3x3(dy, wt, d, ctx->saved_data["stream"].toInt());
      } else {
        dx = at::conv2d(dy, wt, {}, {1, 1}, {pad, pad}, {d, d}, 1);
      }
    }
    if (ctx->needs_input_grad(1)) {
      const int64_t dil = w.size(2) == 3 ? d : 0;
      at::Tensor dyc = dy;
      if (ctx->saved_data["own_wgrad"].toBool() && dyc.scalar_type() == at::kBFloat16 && !dense_channels_last(dyc))
        dyc = dyc.contiguous(at::MemoryFormat::ChannelsLast);
      if (ctx->saved_data["own_wgrad"].toBool() && own_wgrad_ok(dyc, x, w, dil) && (dil == 0 || w.is_contiguous(at::MemoryFormat::ChannelsLast)))
        dw = own_wgrad(dyc, x, w, dil, ctx->saved_data["stream"].toInt());
      else
        dw = std::get<1>(at::convolution_backward(dy, x, w, c10::nullopt, {1, 1}, {pad, pad}, {d, d}, false, {0, 0}, 1,
                                                  {false, true, false}));
    }
    return {dx, dw, at::Tensor(), at::Tensor(), at::Tensor(), at::Tensor(), at::Tensor(), at::Tensor()};
  }
};

at::Tensor conv_stride1(at::Tensor x, at::Tensor w, int64_t d, c10::optional<at::Tensor> wt, bool own_fwd, bool own_dgrad,
                        int64_t stream, bool own_wg) {
  return StrideOneConvNode::apply(x, w, d, wt, own_fwd, own_dgrad, stream, own_wg);
}

// ---- 1x1 convolution + training-mode ABN as ONE node ------------------------------------------------------
// forward   z = x . w^T with the statistics of z accumulated in the GEMM's epilogue (ucd_conv1x1 out_mode 2: no statistics
//           pass over z) -> finalize (or the SyncBN exchange) -> y = act(norm(z) [+ residual])
//           (for the shapes where the tuned library GEMM is faster than the fused kernel by more than the statistics pass
//           costs, `fused` is 0: library GEMM + ucd_abn_forward - the same arithmetic)
// backward  ABN backward (d z, d residual, parameter gradients) -> d x = d z . w (+ the shortcut's gradient, beta = 1)
//           -> d w = d z^T . x as batched split-M products
// with_skip: x is also the block's identity shortcut; the node returns (y, alias of x) so that x has one consumer and the
// shortcut's gradient is folded into the input-gradient product (Gemm1x1SkipNode's trick).
// Reference: conv1 -> bn1, conv3 -> bn3 (+ shortcut, activation), proj_conv -> proj_bn of modules/residual.py:57-97.
// Atomic link (round 5): the producer's `partial` is a zeroed arena slot [2 C] (SyncBN: [4 C] = to-be-reduced | local) instead of
// per-tile rows; flag = {served, address of the consumer's dx, its version, arena generation of the slot (0: tile rows), state, replicas}
// with state 0 empty -> 1 filled by the consumer's product -> 2 read by the producer's backward.
static bool link_is_atomic(const at::Tensor& flag) { return flag.defined() && flag.numel() >= 6 && flag.data_ptr<int64_t>()[3] != 0; }
// may this consumer serve the link?  (tile-row links: always; atomic links: the slot must be of the current generation and empty)
static bool link_servable(const at::Tensor& flag, const at::Tensor& like) {
  if (!link_is_atomic(flag)) return true;
  const int64_t* f = flag.data_ptr<int64_t>();
  TORCH_CHECK(f[4] != 1, "ucd conv+abn node: a backward link's accumulator was filled twice (retain_graph replay of the consumer) - "
                         "run with UCD_STAT_ATOMIC=0");
  return f[4] == 0 && f[3] == arena_gen((int)like.get_device());
}
static void link_atomic(ucd_conv1x1_desc& d, const at::Tensor& lk_partial, const at::Tensor& lk_flag, int64_t C) {
  if (!link_is_atomic(lk_flag)) return;
  const int64_t R = lk_flag.data_ptr<int64_t>()[5];            // replicas (the producer sized the slot: [R][2 C], SyncBN twice)
  d.stat_acc = lk_partial.data_ptr<float>();
  d.stat_rep = (int)R;
  d.stat_acc2 = lk_partial.numel() >= 2 * R * 2 * C ? d.stat_acc + R * 2 * C : nullptr;
  lk_flag.data_ptr<int64_t>()[4] = 1;
}

// B side of the backward link: turn the input-gradient product into out_mode 3 against the producer's statistics
static void link_epilogue(ucd_conv1x1_desc& d, const at::Tensor& lk_z, const at::Tensor& lk_buf, const at::Tensor& lk_bias,
                          const at::Tensor& lk_partial, const at::Tensor& lk_flag, int64_t C, AutogradContext* ctx) {
  const float* b = lk_buf.data_ptr<float>();
  d.out_mode = 3;
  d.out_mean = b + 3 * C; d.out_invstd = b + 4 * C; d.out_scale = b + 5 * C;
  d.out_shift = lk_bias.data_ptr<float>();
  d.residual = lk_z.data_ptr(); d.ldr = (int)C;
  d.out_act = (int)(ctx->saved_data["lk_act"].toInt() & UCD_ACT_MASK);
  d.out_slope = (float)ctx->saved_data["lk_slope"].toDouble();
  d.partial = lk_partial.data_ptr<float>();
  link_atomic(d, lk_partial, lk_flag, C);
  // served: the producer's backward may skip its reduction pass - but ONLY for the gradient this product writes.  The address
  // of that tensor is recorded so the producer can tell "dy is exactly the consumer's dx" from "autograd summed several
  // contributions" (a second consumer of the producer's output: the derivative would be applied to a part of dy only).
  lk_flag.data_ptr<int64_t>()[0] = 1;
  lk_flag.data_ptr<int64_t>()[1] = (int64_t)reinterpret_cast<intptr_t>(d.y);
}

// Block link (kind 3): the producer is the LAST node of a residual block, out = act(norm(z) + shortcut), and this product is
// the input gradient of the next block's first convolution with the identity shortcut's gradient folded in (accumulate): what
// it computes is d out, so its epilogue (out_mode 4) applies the block activation's derivative (sign of out = this node's own
// input x) and accumulates the two backward sums of the producer's norm against x-hat(z).  The producer then runs
// reduce-partials + apply on d pre and hands d pre on as the shortcut's gradient (no second tensor).
static void block_link_epilogue(ucd_conv1x1_desc& d, const at::Tensor& out, const at::Tensor& lk_z, const at::Tensor& lk_buf,
                                const at::Tensor& lk_partial, const at::Tensor& lk_flag, int64_t C, AutogradContext* ctx) {
  const float* b = lk_buf.data_ptr<float>();
  d.out_mode = 4;
  d.out_mean = b + 3 * C; d.out_invstd = b + 4 * C;
  d.residual = out.data_ptr(); d.ldr = (int)C;
  d.side2 = lk_z.data_ptr(); d.ld2 = (int)C;
  d.out_act = UCD_ACT_LEAKY_RELU;
  d.out_slope = (float)ctx->saved_data["lk_slope"].toDouble();
  d.partial = lk_partial.data_ptr<float>();
  link_atomic(d, lk_partial, lk_flag, C);
  lk_flag.data_ptr<int64_t>()[0] = 1;
  lk_flag.data_ptr<int64_t>()[1] = (int64_t)reinterpret_cast<intptr_t>(d.y);
}

// the version counter of the gradient tensor a served link's product wrote: the engine's input buffer may later add a
// second consumer's gradient INTO that tensor in place - same address, so the address check alone would pass - which bumps it
static void link_record_version(const at::Tensor& lk_flag, const at::Tensor& dx) {
  if (lk_flag.defined() && lk_flag.numel() >= 3 && lk_flag.data_ptr<int64_t>()[0] == 1 &&
      lk_flag.data_ptr<int64_t>()[1] == (int64_t)reinterpret_cast<intptr_t>(dx.data_ptr()))
    lk_flag.data_ptr<int64_t>()[2] = (int64_t)dx._version();
}

class ConvABNTrainNode : public torch::autograd::Function<ConvABNTrainNode> {
 public:
  static variable_list forward(AutogradContext* ctx, at::Tensor x, at::Tensor w4, at::Tensor weight, at::Tensor bias,
                               c10::optional<at::Tensor> residual_, at::Tensor running_mean, at::Tensor running_var,
                               double momentum, double eps, int64_t act, double slope, int64_t comm, int64_t world,
                               int64_t stream, int64_t param_grad, bool with_skip, bool fused, int64_t dilation,
                               c10::optional<at::Tensor> wflip_, bool own_dgrad, int64_t wgrad_conv, bool make_link,
                               c10::optional<at::Tensor> lk_z_, c10::optional<at::Tensor> lk_buf_,
                               c10::optional<at::Tensor> lk_bias_, c10::optional<at::Tensor> lk_partial_,
                               c10::optional<at::Tensor> lk_flag_, int64_t lk_act, double lk_slope, int64_t lk_kind,
                               int64_t stride, bool stat_atomic) {
    // stat_atomic (round 5): statistics and link sums through fp32 atomics into arena slots, finalised in the prologue of the apply
    // passes (no tile_stats_reduce / reduce_bands launches); off: the deterministic per-tile rows + second-stage kernels
    // stride > 1: the strided layers of the first block of a stage (conv2 3x3 stride 2 with padding = dilation, proj_conv 1x1
    // stride 2): forward and weight gradient on the own kernels (the strided row gather / implicit GEMM, ucd_conv_wgrad_strided),
    // input gradient through the library's backward-data solver; no shortcut fold, no link consumed (it may still MAKE one).
    // Backward link between two nodes of a chain  A (conv + ABN) -> B (conv + ABN)  where A's output feeds B only:
    // B's input-gradient product applies A's activation derivative and accumulates A's two backward sums in its epilogue
    // (ucd_conv1x1 out_mode 3), so A's backward skips its reduction pass (ucd_abn_bwd_reduce: two reads of the map).
    // make_link (A): also return (z, buf, partial, flag) - partial [row tiles][2][N] on the device, flag a CPU word B sets.
    // lk_* (B): A's z, statistics buffer, bias, partial, flag, activation.  Under SyncBN the producer all-reduces the sums.
    // lk_kind 3 (block link, block_link_epilogue above): A is the previous block's conv3 + bn3 + shortcut + activation node
    // (make_link with a residual), B the first convolution of an identity-shortcut block (with_skip); lk_bias is unused.
    // dilation = 0: 1x1 convolution; dilation >= 1: 3x3, stride 1, padding = dilation (implicit GEMM, taps = 9), weight in
    // channels-last memory order; wflip = w.flip(2, 3).transpose(0, 1) (channels-last; for a 1x1 layer the transposed
    // weight [Ci, Co]) for the input gradient through the own kernel (own_dgrad); wgrad_conv: weight gradient by 0 the batched
    // split-M library products, 1 MIOpen, 2 the own kernel (ucd_conv_wgrad; 3x3 layers: 2 or MIOpen)
    TORCH_CHECK(dense_channels_last(x) && x.scalar_type() == at::kBFloat16, "ucd conv+abn node: x must be dense channels-last bf16");
    const bool conv3 = dilation > 0;
    TORCH_CHECK(w4.dim() == 4 && w4.scalar_type() == at::kBFloat16 && w4.size(1) == x.size(1) &&
                    (conv3 ? (w4.size(2) == 3 && w4.size(3) == 3 && w4.is_contiguous(at::MemoryFormat::ChannelsLast))
                           : (w4.size(2) == 1 && w4.size(3) == 1 && w4.is_contiguous())),
                "ucd conv+abn node: weight [Co, Ci, 1, 1] (contiguous) or [Co, Ci, 3, 3] (channels-last) bf16 expected");
    TORCH_CHECK(!(conv3 && with_skip), "ucd conv+abn node: the shortcut fold belongs to the 1x1 layers");
    at::Tensor wflip = wflip_.has_value() ? *wflip_ : at::Tensor();
    at::Tensor residual = residual_.has_value() ? *residual_ : at::Tensor();
    const bool has_res = residual.defined();
    const int64_t B = x.size(0), K = x.size(1), H = x.size(2), W = x.size(3), N = w4.size(0);
    if (stride < 1) stride = 1;
    TORCH_CHECK(stride == 1 || (!with_skip && !lk_z_.has_value()), "ucd conv+abn node: a strided layer takes no shortcut fold / link");
    const int64_t OH = (H - 1) / stride + 1, OW = (W - 1) / stride + 1;
    const int64_t HW = OH * OW, M = B * HW;                     // rows of the OUTPUT map
    if (has_res)
      TORCH_CHECK(dense_channels_last(residual) && residual.size(1) == N && residual.size(0) == B && residual.size(2) == OH &&
                      residual.size(3) == OW && residual.scalar_type() == at::kBFloat16,
                  "ucd conv+abn node: residual must match the output");
    const bool sync = comm != 0;
    auto opts = x.options().memory_format(at::MemoryFormat::ChannelsLast);
    at::Tensor z = at::empty({B, N, OH, OW}, opts), y = at::empty({B, N, OH, OW}, opts);
    at::Tensor buf = at::empty({(sync ? 8 + 2 * world : 6) * N}, x.options().dtype(at::kFloat));
    float* b = buf.data_ptr<float>();
    const size_t ws_bytes = ucd_abn_workspace_bytes((int)M, (int)N);
    const void* resp = has_res ? residual.data_ptr() : nullptr;
    if (fused) {
      size_t need = ucd_conv1x1_stats_partial_bytes((int)M, (int)N);
      float* partial = (float*)workspace(x, need > ws_bytes ? need : ws_bytes, stream);
      ucd_conv1x1_desc d;
      memset(&d, 0, sizeof(d));
      d.a = x.data_ptr(); d.lda = (int)K; d.w = w4.data_ptr(); d.ldw = (int)(conv3 ? 9 * K : K); d.y = z.data_ptr(); d.ldy = (int)N;
      d.M = (int)M; d.N = (int)N; d.K = (int)K; d.out_mode = 2; d.partial = partial;
      if (conv3) { d.taps = 9; d.H = (int)H; d.W = (int)W; d.dilation = (int)dilation; }
      if (stride > 1) { d.H = (int)H; d.W = (int)W; d.stride = (int)stride; }
      int64_t gen = 0;
      const int64_t R = ucd_conv1x1_stat_replicas((int)M);
      at::Tensor acc = (stat_atomic && N % 8 == 0) ? arena_alloc(x, R * 2 * N, &gen, stream) : at::Tensor();
      if (acc.defined()) {
        // sums about the running mean (equal on every rank), straight into the layer's accumulator; SyncBN: one all-reduce of the
        // 2 N raw sums (they are additive about a common shift) instead of gather + Chan combination
        d.stat_acc = acc.data_ptr<float>();
        d.stat_rep = (int)R;
        d.stat_shift = running_mean.data_ptr<float>();
        d.partial = b + 2 * N;                                   // the snapshot of the shift
        check(ucd_conv1x1(&d, (ucd_stream_t)stream), "ucd_conv1x1");
        if (sync) check(ucd_comm_all_reduce_sum((ucd_comm_t)comm, d.stat_acc, (size_t)(R * 2 * N), (ucd_stream_t)stream), "ucd_comm_all_reduce_sum");
        check(ucd_abn_apply_stats(z.data_ptr(), (int)N, y.data_ptr(), (int)N, resp, has_res ? (int)N : 0, (int)M, (int)N, d.stat_acc,
                                  (int)R, b + 2 * N, (float)M * (sync ? (float)world : 1.f), fptr(weight), fptr(bias),
                                  running_mean.data_ptr<float>(), running_var.data_ptr<float>(), (float)momentum, (float)eps, b + 3 * N,
                                  b + 4 * N, b + 5 * N, (int)act, (float)slope, (ucd_stream_t)stream),
              "ucd_abn_apply_stats");
      } else {
      check(ucd_conv1x1(&d, (ucd_stream_t)stream), "ucd_conv1x1");
      if (!sync) {
        check(ucd_conv1x1_stats_finalize(partial, (int)M, (int)N, fptr(weight), running_mean.data_ptr<float>(),
                                         running_var.data_ptr<float>(), (float)momentum, (float)eps, b, nullptr, (int)act,
                                         (ucd_stream_t)stream),
              "ucd_conv1x1_stats_finalize");
        check(ucd_abn_apply(z.data_ptr(), (int)N, y.data_ptr(), (int)N, resp, has_res ? (int)N : 0, UCD_BF16, (int)M, (int)N, nullptr,
                            (int)HW, b + 3 * N, b + 5 * N, fptr(bias), (int)act, (float)slope, (ucd_stream_t)stream),
              "ucd_abn_apply");
      } else {
        float *pack = b + 6 * N, *gathered = b + 8 * N;
        check(ucd_conv1x1_stats_finalize(partial, (int)M, (int)N, nullptr, nullptr, nullptr, (float)momentum, (float)eps, b, pack, 0,
                                         (ucd_stream_t)stream),
              "ucd_conv1x1_stats_finalize");
        check(ucd_comm_all_gather((ucd_comm_t)comm, pack, gathered, (size_t)2 * N, (ucd_stream_t)stream), "ucd_comm_all_gather");
        check(ucd_abn_sync_forward(z.data_ptr(), (int)N, y.data_ptr(), (int)N, resp, has_res ? (int)N : 0, UCD_BF16, (int)M, (int)N,
                                   nullptr, (int)HW, gathered, (int)world, fptr(weight), fptr(bias), running_mean.data_ptr<float>(),
                                   running_var.data_ptr<float>(), (float)momentum, (float)eps, b, (int)act, (float)slope,
                                   (ucd_stream_t)stream),
              "ucd_abn_sync_forward");
      }
      }
    } else {
      if (conv3 || stride > 1) {
        const int64_t pad = conv3 ? dilation : 0, dil = conv3 ? dilation : 1;
        z = at::conv2d(x, w4, {}, {stride, stride}, {pad, pad}, {dil, dil}, 1);
        if (!dense_channels_last(z)) z = z.contiguous(at::MemoryFormat::ChannelsLast);
      } else {
        const size_t wsb = ucd_gemm_workspace_bytes();
        check(ucd_gemm_bf16(0, (int)M, (int)N, (int)K, x.data_ptr(), (int)K, w4.data_ptr(), (int)K, z.data_ptr(), (int)N,
                            workspace(x, wsb, stream, 1), wsb, 1, (ucd_stream_t)stream),
              "ucd_gemm_bf16");
      }
      void* ws = workspace(x, ws_bytes, stream);
      if (sync)
        check(ucd_abn_sync_forward_comm((ucd_comm_t)comm, (int)world, z.data_ptr(), (int)N, y.data_ptr(), (int)N, resp,
                                        has_res ? (int)N : 0, UCD_BF16, (int)M, (int)N, nullptr, (int)HW, fptr(weight), fptr(bias),
                                        running_mean.data_ptr<float>(), running_var.data_ptr<float>(), (float)momentum, (float)eps, b,
                                        (int)act, (float)slope, ws, ws_bytes, (ucd_stream_t)stream),
              "ucd_abn_sync_forward_comm");
      else
        check(ucd_abn_forward(z.data_ptr(), (int)N, y.data_ptr(), (int)N, resp, has_res ? (int)N : 0, UCD_BF16, (int)M, (int)N, nullptr,
                              (int)HW, fptr(weight), fptr(bias), running_mean.data_ptr<float>(), running_var.data_ptr<float>(),
                              (float)momentum, (float)eps, 1, b, nullptr, (int)act, (float)slope, ws, ws_bytes, (ucd_stream_t)stream),
              "ucd_abn_forward");
    }
    const bool needs_y = has_res && (act & UCD_ACT_MASK) != UCD_ACT_IDENTITY;
    at::Tensor lk_z = lk_z_.has_value() ? *lk_z_ : at::Tensor(), lk_buf = lk_buf_.has_value() ? *lk_buf_ : at::Tensor();
    at::Tensor lk_bias = lk_bias_.has_value() ? *lk_bias_ : at::Tensor();
    at::Tensor lk_partial = lk_partial_.has_value() ? *lk_partial_ : at::Tensor();
    at::Tensor lk_flag = lk_flag_.has_value() ? *lk_flag_ : at::Tensor();
    const bool consume_link = lk_z.defined() && lk_buf.defined() && (lk_bias.defined() || lk_kind == 3) && lk_partial.defined() &&
                              lk_flag.defined() && (lk_kind != 3 || with_skip) && (!link_is_atomic(lk_flag) || (K % 8 == 0));
    at::Tensor my_partial, my_flag;
    // a link needs leaky_relu / identity (the fused epilogues' activations); with a residual it is the block link, whose
    // consumer reads the sign from y - so y must be among the saved tensors (needs_y: leaky_relu with a residual)
    make_link = make_link && bias.defined() && (act & UCD_ACT_MASK) != UCD_ACT_ELU && (!has_res || needs_y);
    if (make_link) {
      my_flag = at::zeros({6}, at::TensorOptions().dtype(at::kLong));   // {served, address of the consumer's dx, its version, generation, state, replicas}
      int64_t gen = 0;
      const int64_t R = ucd_conv1x1_stat_replicas((int)M);
      if (stat_atomic && N % 8 == 0) my_partial = arena_alloc(x, (sync ? 2 : 1) * R * 2 * N, &gen, stream);
      if (my_partial.defined()) {
        my_partial = my_partial.narrow(0, 0, (sync ? 2 : 1) * R * 2 * N);
        my_flag.data_ptr<int64_t>()[3] = gen;
        my_flag.data_ptr<int64_t>()[5] = R;
      } else {
        my_partial = at::empty({(int64_t)ucd_conv1x1_row_tiles((int)M), 2, N}, x.options().dtype(at::kFloat));
      }
    }
    ctx->save_for_backward({x, w4, z, needs_y ? y : at::Tensor(), weight, bias, buf, wflip,
                            consume_link ? lk_z : at::Tensor(), consume_link ? lk_buf : at::Tensor(),
                            consume_link ? lk_bias : at::Tensor(), consume_link ? lk_partial : at::Tensor(),
                            consume_link ? lk_flag : at::Tensor(), my_partial, my_flag});
    ctx->saved_data["lk_act"] = lk_act;
    ctx->saved_data["lk_slope"] = lk_slope;
    ctx->saved_data["lk_kind"] = lk_kind;
    ctx->saved_data["act"] = act;
    ctx->saved_data["slope"] = slope;
    ctx->saved_data["comm"] = comm;
    ctx->saved_data["world"] = world;
    ctx->saved_data["stream"] = stream;
    ctx->saved_data["has_res"] = has_res;
    ctx->saved_data["param_grad"] = param_grad;
    ctx->saved_data["with_skip"] = with_skip;
    ctx->saved_data["dilation"] = dilation;
    ctx->saved_data["stride"] = stride;
    ctx->saved_data["own_dgrad"] = own_dgrad;
    ctx->saved_data["wgrad_conv"] = wgrad_conv;
    if (make_link) {   // outputs: y, [x], z, buf, partial, flag
      // no zero-filled stand-ins for the gradients of the link tensors (the engine would launch a fill per output and step)
      ctx->set_materialize_grads(false);
      ctx->mark_non_differentiable({z, buf, my_partial, my_flag});
      if (with_skip) return {y, x, z, buf, my_partial, my_flag};
      return {y, z, buf, my_partial, my_flag};
    }
    if (with_skip) return {y, x};
    return {y};
  }

  static variable_list backward(AutogradContext* ctx, variable_list grads) {
    auto saved = ctx->get_saved_variables();
    at::Tensor x = saved[0], w4 = saved[1], z = saved[2], y = saved[3], weight = saved[4], bias = saved[5], buf = saved[6];
    at::Tensor wflip = saved[7];
    at::Tensor lk_z = saved[8], lk_buf = saved[9], lk_bias = saved[10], lk_partial = saved[11], lk_flag = saved[12];
    at::Tensor my_partial = saved[13], my_flag = saved[14];
    const int64_t dilation = ctx->saved_data["dilation"].toInt();
    const bool conv3 = dilation > 0, own_dgrad = ctx->saved_data["own_dgrad"].toBool();
    const int64_t act = ctx->saved_data["act"].toInt(), comm = ctx->saved_data["comm"].toInt();
    const int64_t world = ctx->saved_data["world"].toInt(), stream = ctx->saved_data["stream"].toInt();
    const double slope = ctx->saved_data["slope"].toDouble();
    const bool has_res = ctx->saved_data["has_res"].toBool(), with_skip = ctx->saved_data["with_skip"].toBool();
    float* param_grad = reinterpret_cast<float*>(ctx->saved_data["param_grad"].toInt());
    const int64_t stride = ctx->saved_data["stride"].toInt();
    const int64_t B = x.size(0), K = x.size(1), H = x.size(2), W = x.size(3), N = w4.size(0);
    const int64_t HW = z.size(2) * z.size(3), M = B * HW;       // rows of the output map (smaller than x's when strided)
    at::Tensor dy = grads[0];
    at::Tensor dskip = with_skip ? grads[1] : at::Tensor();
    at::Tensor none;
    at::Tensor dx, dw, dweight, dbias, dres;
    at::Tensor dz;   // gradient w.r.t. the convolution output z
    if (my_flag.defined() && my_flag.data_ptr<int64_t>()[0] == 1) {
      // the link's promise (one consumer of y) is checked, not trusted: the incoming gradient must be the very tensor the
      // consumer's input-gradient product wrote.  Anything else (a sum made by the engine, a hook's copy, a stale flag from
      // a backward pass that stopped short of this node) carries a derivative on part of dy only - wrong gradients, silently.
      const int64_t want = my_flag.data_ptr<int64_t>()[1];
      const bool same = dy.defined() && (int64_t)reinterpret_cast<intptr_t>(dy.data_ptr()) == want &&
                        (my_flag.numel() < 3 || (int64_t)dy._version() == my_flag.data_ptr<int64_t>()[2]);
      if (!same) {
        my_flag.data_ptr<int64_t>()[0] = 0;
        TORCH_CHECK(false, "ucd conv+abn node: the backward link was served but the gradient that arrived is not the consumer's "
                           "input gradient, or was added to in place since - the linked map has a second consumer (hook, "
                           "ret_intermediate tap, retain_graph replay).  Run with UCD_BWD_LINK=0.");
      }
    }
    if (dy.defined()) {
      if (dy.scalar_type() != at::kBFloat16) dy = dy.to(at::kBFloat16);
      if (!dense_channels_last(dy)) dy = dy.contiguous(at::MemoryFormat::ChannelsLast);
      dz = at::empty_like(z);
      const bool linked_now = my_flag.defined() && my_flag.data_ptr<int64_t>()[0] == 1;
      if (has_res && !linked_now) dres = at::empty_like(z);
      const bool sync = comm != 0;
      at::Tensor sums;
      if (sync || !param_grad) sums = at::empty({(sync && !param_grad ? 4 : 2) * N}, x.options().dtype(at::kFloat));
      float* b = buf.data_ptr<float>();
      const float *mean = b + 3 * N, *invstd = b + 4 * N, *scale = b + 5 * N;
      const size_t ws_bytes = ucd_abn_workspace_bytes((int)M, (int)N);
      void* ws = workspace(x, ws_bytes, stream);
      const void* yp = y.defined() ? y.data_ptr() : nullptr;
      const bool linked = my_flag.defined() && my_flag.data_ptr<int64_t>()[0] == 1;
      if (linked && link_is_atomic(my_flag)) {
        // atomic link: the consumer's product added the two sums into this layer's arena slot; the apply pass finalises them in its
        // prologue and writes the parameter gradients (SyncBN: the first half is all-reduced, the second stays this rank's)
        int64_t* fl = my_flag.data_ptr<int64_t>();
        fl[0] = 0;
        TORCH_CHECK(fl[4] == 1 && fl[3] == arena_gen((int)x.get_device()),
                    "ucd conv+abn node: the statistics arena was reset between a link's consumer and its producer (a training forward "
                    "inside a backward pass?) - run with UCD_STAT_ATOMIC=0");
        fl[4] = 2;
        const int64_t R = fl[5];
        float* global = my_partial.data_ptr<float>();
        float* local = sync ? global + R * 2 * N : global;
        if (sync) check(ucd_comm_all_reduce_sum((ucd_comm_t)comm, global, (size_t)(R * 2 * N), (ucd_stream_t)stream), "ucd_comm_all_reduce_sum");
        if (!param_grad) sums = at::empty({2 * N}, x.options().dtype(at::kFloat));
        check(ucd_abn_bwd_apply_raw(z.data_ptr(), (int)N, dy.data_ptr(), (int)N, nullptr, 0, dz.data_ptr(), (int)N, nullptr, 0, (int)M,
                                    (int)N, mean, invstd, scale, fptr(bias), fptr(weight), global, local, (int)R,
                                    param_grad ? param_grad : sums.data_ptr<float>(), (float)M * (sync ? (float)world : 1.f),
                                    (int)(UCD_ACT_IDENTITY | (act & UCD_NORM_ABS_GAMMA)), 0.f, (ucd_stream_t)stream),
              "ucd_abn_bwd_apply_raw");
        if (has_res) dres = dy;
        if (!param_grad) {
          dbias = sums.narrow(0, 0, N);
          dweight = sums.narrow(0, N, N);
        }
      } else if (linked) {
        // the consumer's input-gradient product already applied this layer's activation derivative and left the two sums
        // as per-tile partials: combine them (fixed order) and go straight to the apply pass
        my_flag.data_ptr<int64_t>()[0] = 0;
        // single process: the sums are the parameter gradients and the constants of the apply pass.  SyncBN: the apply
        // pass needs the sums over all ranks (all-reduce, like ucd_abn_sync_backward_comm), the parameter gradients stay
        // this rank's (the DDP wrapper averages them).
        float* local = param_grad ? param_grad : sums.data_ptr<float>() + (sync ? 2 * N : 0);
        float* global = sync ? sums.data_ptr<float>() : local;
        check(ucd_abn_reduce_partials(my_partial.data_ptr<float>(), ucd_conv1x1_row_tiles((int)M), (int)N, global,
                                      sync ? local : nullptr, fptr(weight), (int)(act & UCD_NORM_ABS_GAMMA), (ucd_stream_t)stream),
              "ucd_abn_reduce_partials");
        if (sync) check(ucd_comm_all_reduce_sum((ucd_comm_t)comm, global, (size_t)2 * N, (ucd_stream_t)stream), "ucd_comm_all_reduce_sum");
        check(ucd_abn_bwd_apply(z.data_ptr(), (int)N, dy.data_ptr(), (int)N, nullptr, 0, dz.data_ptr(), (int)N, nullptr, 0, UCD_BF16,
                                (int)M, (int)N, nullptr, (int)HW, mean, invstd, scale, fptr(bias), fptr(weight), global,
                                (float)M * (sync ? (float)world : 1.f), 0, (int)(UCD_ACT_IDENTITY | (act & UCD_NORM_ABS_GAMMA)), 0.f,
                                (ucd_stream_t)stream),
              "ucd_abn_bwd_apply");
        // block link: dy already IS d pre = d out * act'(out); the shortcut's gradient is that same tensor (no copy, no write)
        if (has_res) dres = dy;
        if (!param_grad) {
          dbias = sums.narrow(0, sync ? 2 * N : 0, N);
          dweight = sums.narrow(0, sync ? 3 * N : N, N);
        }
      } else if (sync) {
        check(ucd_abn_sync_backward_comm((ucd_comm_t)comm, (int)world, z.data_ptr(), (int)N, dy.data_ptr(), (int)N, yp, yp ? (int)N : 0,
                                         dz.data_ptr(), (int)N, has_res ? dres.data_ptr() : nullptr, has_res ? (int)N : 0, UCD_BF16,
                                         (int)M, (int)N, nullptr, (int)HW, mean, invstd, scale, fptr(bias), fptr(weight),
                                         sums.data_ptr<float>(), param_grad ? param_grad : sums.data_ptr<float>() + 2 * N, (int)act,
                                         (float)slope, ws, ws_bytes, (ucd_stream_t)stream),
              "ucd_abn_sync_backward_comm");
        if (!param_grad) { dbias = sums.narrow(0, 2 * N, N); dweight = sums.narrow(0, 3 * N, N); }
      } else {
        check(ucd_abn_backward(z.data_ptr(), (int)N, dy.data_ptr(), (int)N, yp, yp ? (int)N : 0, dz.data_ptr(), (int)N,
                               has_res ? dres.data_ptr() : nullptr, has_res ? (int)N : 0, UCD_BF16, (int)M, (int)N, nullptr, (int)HW,
                               mean, invstd, scale, fptr(bias), fptr(weight), param_grad ? param_grad : sums.data_ptr<float>(),
                               (float)M, 1, 1, (int)act, (float)slope, ws, ws_bytes, (ucd_stream_t)stream),
              "ucd_abn_backward");
        if (!param_grad) { dbias = sums.narrow(0, 0, N); dweight = sums.narrow(0, N, N); }
      }
    }
    if (stride > 1) {
      const int64_t pad = conv3 ? dilation : 0, dil = conv3 ? dilation : 1;
      if (ctx->needs_input_grad(0) && dz.defined())
        dx = std::get<0>(at::convolution_backward(dz, x, w4, c10::nullopt, {stride, stride}, {pad, pad}, {dil, dil}, false, {0, 0}, 1,
                                                  {true, false, false}));
      if (ctx->needs_input_grad(1) && dz.defined()) {
        if (ctx->saved_data["wgrad_conv"].toInt() == 2 && own_wgrad_ok(dz, x, w4, conv3 ? dilation : 0) && N % 128 == 0 && K % 128 == 0)
          dw = own_wgrad(dz, x, w4, conv3 ? dilation : 0, stream, stride);
        else
          dw = std::get<1>(at::convolution_backward(dz, x, w4, c10::nullopt, {stride, stride}, {pad, pad}, {dil, dil}, false, {0, 0}, 1,
                                                    {false, true, false}));
      }
      return {dx, dw, dweight, dbias, has_res ? dres : none, none, none, none, none, none, none, none, none, none, none, none, none,
              none, none, none, none, none, none, none, none, none, none, none, none, none, none, none};
    }
    if (conv3) {
      // 3x3: input gradient = the same convolution on the flipped + transposed weight (own implicit GEMM, or MIOpen's
      // FORWARD solver when the cached weight is missing / the map is too small to fill the chip); weight gradient: MIOpen
      if (ctx->needs_input_grad(0) && dz.defined()) {
        if (!wflip.defined()) wflip = w4.flip({2, 3}).transpose(0, 1).contiguous(at::MemoryFormat::ChannelsLast);
        if (own_dgrad) {
          dx = at::empty_like(x);
          ucd_conv1x1_desc d;
          memset(&d, 0, sizeof(d));
          d.a = dz.data_ptr(); d.lda = (int)N; d.w = wflip.data_ptr(); d.ldw = (int)(9 * N); d.y = dx.data_ptr(); d.ldy = (int)K;
          d.M = (int)M; d.N = (int)K; d.K = (int)N; d.out_mode = 0;
          d.taps = 9; d.H = (int)H; d.W = (int)W; d.dilation = (int)dilation;
          if (lk_flag.defined() && ctx->saved_data["lk_kind"].toInt() != 3 && link_servable(lk_flag, x)) {
            link_epilogue(d, lk_z, lk_buf, lk_bias, lk_partial, lk_flag, K, ctx);
            link_record_version(lk_flag, dx);
          }
          check(ucd_conv1x1(&d, (ucd_stream_t)stream), "ucd_conv1x1");
        } else {
          dx = at::conv2d(dz, wflip, {}, {1, 1}, {dilation, dilation}, {dilation, dilation}, 1);
        }
      }
      if (ctx->needs_input_grad(1) && dz.defined()) {
        if (ctx->saved_data["wgrad_conv"].toInt() == 2 && own_wgrad_ok(dz, x, w4, dilation))
          dw = own_wgrad(dz, x, w4, dilation, stream);
        else
          dw = std::get<1>(at::convolution_backward(dz, x, w4, c10::nullopt, {1, 1}, {dilation, dilation}, {dilation, dilation}, false,
                                                    {0, 0}, 1, {false, true, false}));
      }
      return {dx, dw, dweight, dbias, has_res ? dres : none, none, none, none, none, none, none, none, none, none, none, none, none,
              none, none, none, none, none, none, none, none, none, none, none, none, none, none, none};
    }
    const size_t wsb = ucd_gemm_workspace_bytes();
    void* gws = workspace(x, wsb, stream, 1);
    if (ctx->needs_input_grad(0)) {
      const bool fold = dskip.defined() && dz.defined() && dskip.scalar_type() == at::kBFloat16;
      if (own_dgrad && dz.defined() && wflip.defined()) {
        // narrow / short-K layers: d x = d z . w through the own kernel on the cached transposed weight [Ci, Co]; the
        // shortcut's gradient is the accumulate operand (no separate add)
        if (fold && !dense_channels_last(dskip)) dskip = dskip.contiguous(at::MemoryFormat::ChannelsLast);
        dx = fold ? dskip : at::empty_like(x);
        ucd_conv1x1_desc d;
        memset(&d, 0, sizeof(d));
        d.a = dz.data_ptr(); d.lda = (int)N; d.w = wflip.data_ptr(); d.ldw = (int)N; d.y = dx.data_ptr(); d.ldy = (int)K;
        d.M = (int)M; d.N = (int)K; d.K = (int)N; d.out_mode = 0; d.accumulate = fold ? 1 : 0;
        const bool block_link = lk_flag.defined() && ctx->saved_data["lk_kind"].toInt() == 3;
        if (lk_flag.defined() && !link_servable(lk_flag, x)) {
          // an atomic link whose arena slot is stale: not served - the producer runs its own reduction pass
        } else if (block_link) {
          if (with_skip && (fold || !dskip.defined())) block_link_epilogue(d, x, lk_z, lk_buf, lk_partial, lk_flag, K, ctx);
        } else if (lk_flag.defined() && !fold && !dskip.defined()) {
          link_epilogue(d, lk_z, lk_buf, lk_bias, lk_partial, lk_flag, K, ctx);
        }
        link_record_version(lk_flag, dx);
        check(ucd_conv1x1(&d, (ucd_stream_t)stream), "ucd_conv1x1");
        if (!fold && dskip.defined()) dx = dx + dskip;
      } else if (fold) {
        if (!dense_channels_last(dskip)) dskip = dskip.contiguous(at::MemoryFormat::ChannelsLast);
        if (!ucd_gemm_has_plan(1, (int)M, (int)K, (int)N, (int)N, (int)K, (int)K)) {   // tune once, into scratch
          at::Tensor scratch = at::empty_like(x);
          check(ucd_gemm_bf16(1, (int)M, (int)K, (int)N, dz.data_ptr(), (int)N, w4.data_ptr(), (int)K, scratch.data_ptr(), (int)K,
                              gws, wsb, 1, (ucd_stream_t)stream),
                "ucd_gemm_bf16");
        }
        check(ucd_gemm_bf16_acc(1, (int)M, (int)K, (int)N, dz.data_ptr(), (int)N, w4.data_ptr(), (int)K, dskip.data_ptr(), (int)K, gws,
                                wsb, (ucd_stream_t)stream),
              "ucd_gemm_bf16_acc");
        dx = dskip;
      } else if (dz.defined()) {
        dx = at::empty_like(x);
        check(ucd_gemm_bf16(1, (int)M, (int)K, (int)N, dz.data_ptr(), (int)N, w4.data_ptr(), (int)K, dx.data_ptr(), (int)K, gws, wsb, 1,
                            (ucd_stream_t)stream),
              "ucd_gemm_bf16");
        if (dskip.defined()) dx = dx + dskip;
      } else {
        dx = dskip;
      }
    }
    if (ctx->needs_input_grad(1) && dz.defined() && ctx->saved_data["wgrad_conv"].toInt() == 2 && own_wgrad_ok(dz, x, w4, 0)) {
      dw = own_wgrad(dz, x, w4, 0, stream);
    } else if (ctx->needs_input_grad(1) && dz.defined() && ctx->saved_data["wgrad_conv"].toInt() == 1) {
      // narrow layers (<= 512 channels at 65^2 / 129^2): MIOpen's weight-gradient solver beats the split-M products
      dw = std::get<1>(at::convolution_backward(dz, x, w4, c10::nullopt, {1, 1}, {0, 0}, {1, 1}, false, {0, 0}, 1,
                                                {false, true, false}));
    } else if (ctx->needs_input_grad(1) && dz.defined()) {
      at::Tensor dz2 = dz.permute({0, 2, 3, 1}).reshape({M, N}), x2 = x.permute({0, 2, 3, 1}).reshape({M, K});
      const int64_t S = wgrad_split(M);
      if (S > 1) {
        dw = at::bmm(dz2.view({S, M / S, N}).transpose(1, 2), x2.view({S, M / S, K})).sum(0);
      } else {
        dw = at::empty({N, K}, x.options().memory_format(c10::nullopt));
        check(ucd_gemm_bf16(2, (int)N, (int)K, (int)M, dz2.data_ptr(), (int)N, x2.data_ptr(), (int)K, dw.data_ptr(), (int)K, gws, wsb, 1,
                            (ucd_stream_t)stream),
              "ucd_gemm_bf16");
      }
      dw = dw.as_strided(w4.sizes(), w4.strides());
    }
    return {dx, dw, dweight, dbias, has_res ? dres : none, none, none, none, none, none, none, none, none, none, none, none, none,
            none, none, none, none, none, none, none, none, none, none, none, none, none, none, none};
  }
};

std::vector<at::Tensor> conv_abn_train(at::Tensor x, at::Tensor w4, at::Tensor weight, at::Tensor bias,
                                       c10::optional<at::Tensor> residual, at::Tensor running_mean, at::Tensor running_var,
                                       double momentum, double eps, int64_t act, double slope, int64_t comm, int64_t world,
                                       int64_t stream, int64_t param_grad, bool with_skip, bool fused, int64_t dilation,
                                       c10::optional<at::Tensor> wflip, bool own_dgrad, int64_t wgrad_conv, bool make_link,
                                       c10::optional<at::Tensor> lk_z, c10::optional<at::Tensor> lk_buf,
                                       c10::optional<at::Tensor> lk_bias, c10::optional<at::Tensor> lk_partial,
                                       c10::optional<at::Tensor> lk_flag, int64_t lk_act, double lk_slope, int64_t lk_kind,
                                       int64_t stride, bool stat_atomic) {
  return ConvABNTrainNode::apply(x, w4, weight, bias, residual, running_mean, running_var, momentum, eps, act, slope, comm, world,
                                 stream, param_grad, with_skip, fused, dilation, wflip, own_dgrad, wgrad_conv, make_link, lk_z,
                                 lk_buf, lk_bias, lk_partial, lk_flag, lk_act, lk_slope, lk_kind, stride, stat_atomic);
}

at::Tensor abn_train(at::Tensor x, at::Tensor weight, at::Tensor bias, c10::optional<at::Tensor> residual,
                     at::Tensor running_mean, at::Tensor running_var, double momentum, double eps, int64_t act, double slope,
                     int64_t comm, int64_t world, int64_t stream, int64_t param_grad) {
  return ABNTrainNode::apply(x, weight, bias, residual, running_mean, running_var, momentum, eps, act, slope, comm, world, stream,
                             param_grad);
}

}  // namespace

PYBIND11_MODULE(TORCH_EXTENSION_NAME, m) {
  m.doc() = "C++ autograd node of the training-mode ABN layer over libucd_hip.so";
  m.def("poison_workspaces", &poison_workspaces);
  m.def("pass_flushes", []() { return g_pass_flushes; }, "number of end-of-pass flushes the nodes' engine callbacks have run (test hook)");
  m.def("wgrad_side_release", &wgrad_side_release, "drop the operands held for the side stream of the weight gradients (after the join)");
  m.def("stat_arena_reset", &stat_arena_reset, "zero the used part of the statistics arena of a device and start a new generation");
  m.def("abn_train", &abn_train, "y = act(BN_batch(x) [+ residual]) with autograd in C++");
  m.def("dense_channels_last", &dense_channels_last);
  m.def("conv_stride1", &conv_stride1, "stride-1 conv (3x3 pad=dilation, or 1x1) with the input gradient on the forward solver");
  m.def("gemm1x1_skip", &gemm1x1_skip, "(rows x w^T, rows): first 1x1 conv of an identity-shortcut block with the shortcut");
  m.def("gemm1x1", &gemm1x1, "rows[M, Ci] x w[Co, Ci, 1, 1]^T with autograd in C++ (call ucd_gemm_load first)");
  m.def("conv_abn_train", &conv_abn_train, "1x1 convolution + training-mode ABN (+ residual) as one node, statistics in the GEMM epilogue");
}
