// RCCL communicator owned by the library: the SyncBN statistics exchanges of a layer run on the caller's compute
// stream, inside the same library call as the kernels around them (no hop to a communication stream and back, no
// framework dispatch per collective - 212 small collectives per step at one process per GPU).
//
// RCCL is bound at run time from the copy the process already has loaded (PyTorch-ROCm ships its own librccl.so; two
// RCCL instances in one process must not be mixed), so libucd_hip.so has no link-time dependency on it.
#include <dlfcn.h>
#include <rccl/rccl.h>

#include "common.h"

namespace ucd {
namespace {

struct RcclApi {
  void* handle = nullptr;
  ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
  ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
  ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
  ncclResult_t (*AllGather)(const void*, void*, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
  const char* (*GetErrorString)(ncclResult_t) = nullptr;
} g_rccl;

int rccl_fail(const char* fn, ncclResult_t r) {
  set_error("%s: RCCL error %d (%s)", fn, (int)r, g_rccl.GetErrorString ? g_rccl.GetErrorString(r) : "?");
  return UCD_ERCCL_BASE + (int)r;
}

#define UCD_RCCL(fn, call)                      \
  do {                                          \
    ncclResult_t r_ = (call);                   \
    if (r_ != ncclSuccess) return rccl_fail(fn, r_); \
  } while (0)

// ---- one-shot mailbox exchange (round 5; SURVEY section 7 hard part 1, N3) ------------------------------------------------------------
// The SyncBN exchanges are 212 collectives of a few KB per step on the dependent chain: a stock RCCL call is a kernel launch plus
// its protocol (~15-20 us at 8 ranks) for 8 KB of payload.  Here every rank owns a MAILBOX - device memory shared with the other
// ranks of the node through hipIpcMemHandle - of [2 parities][world slots][slot floats] plus [2][chunks][world] sequence flags, and an
// exchange is ONE kernel (one workgroup per 4096 floats) on the caller's stream: write my vector into slot `rank` of every peer's mailbox (xGMI
// stores on a node, same-device stores when several ranks share a GPU in the tests), system-scope fence, store the sequence number
// into every peer's flag word, spin (bounded by s_memrealtime) until all `world` flags of my mailbox carry it, then sum the slots in
// rank order - the same order on every rank, so all ranks hold bit-identical results.  Two parities: a rank can be at most one
// exchange ahead of the slowest one (it needs that rank's flag of exchange k + 1 to finish it), so exchange k + 2 never overwrites
// a slot still being read.  The sequence counter lives in device memory and is advanced by the kernel itself: a captured step graph
// replays it.
// A timeout cannot pass silently (round 6): the workgroup that gives up (a) writes NaN into its part of `out` - the step's loss says
// so on every path, a replayed graph included -, (b) latches the exchange number in pinned host memory (the next host-issued call on
// the communicator returns UCD_ETIMEOUT, Trainer.train polls the word at its host synchronisations) and (c) POISONS the
// communicator: a word in its own mailbox and in every peer's.  A poisoned exchange kernel returns NaN at once, without the
// protocol and without waiting - one rank that fell out of step (its flags then carry sequence numbers nobody waits for) would
// otherwise make every later exchange of every rank sit out the full timeout (the "15-minute hang" of profiles/r05_ipc_exchange.md
// was 430 exchanges x 2 s), and sum whatever the slots hold (the "ranks out of lockstep" event of the same note).
constexpr int kIpcMaxWorld = 16;
constexpr int kIpcChunk = 4096;                         // floats of a vector one workgroup of the exchange kernel owns
constexpr unsigned kCommMagic = 0x55434443u;            // "UCDC"

struct IpcState {
  int world = 0, rank = 0, slot = 0, groups = 1;       // groups = ceil(slot / kIpcChunk)
  void* base = nullptr;                                 // my allocation: box | flags | ctl (sequence counters, timeouts, poison)
  size_t bytes = 0, box_bytes = 0;
  void* peer[kIpcMaxWorld] = {};                        // every rank's allocation as mapped here (peer[rank] == base)
  unsigned* host_to = nullptr;                          // pinned host word: sequence number of the first timed-out exchange
  unsigned long long timeout_ticks = 200000000ull;      // s_memrealtime ticks (100 MHz): 2 s
};

struct Comm {
  unsigned magic = kCommMagic;
  ncclComm_t nccl = nullptr;
  IpcState* ipc = nullptr;
  int world = 1, rank = 0;
};

struct IpcArgs {
  float* box[kIpcMaxWorld];
  unsigned* flags[kIpcMaxWorld];
  unsigned* poison[kIpcMaxWorld];                       // every rank's poison word (ctl[groups + 1] of its mailbox)
  unsigned* ctl;
  unsigned* host_to;
  const float* send;
  float* out;
  int world, rank, slot, count, gather, groups, vec;
  unsigned long long timeout;
};

__device__ __forceinline__ void st_sys(float* p, float v) {
  __hip_atomic_store(reinterpret_cast<unsigned*>(p), __float_as_uint(v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}
__device__ __forceinline__ float ld_sys(const float* p) {
  return __uint_as_float(__hip_atomic_load(reinterpret_cast<const unsigned*>(p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM));
}

// One workgroup per 4096-float chunk of the vector (kIpcChunk): workgroup g ALWAYS owns floats [4096 g, 4096 (g + 1)) of every slot,
// with its own sequence counter ctl[g], its own parity and its own flag words flags[par][g][rank] - the single-workgroup protocol
// run independently per chunk, so a 2 C-float message costs one workgroup and an R x 2 C-float one (replicated accumulators) spreads
// its mailbox traffic over up to slot / 4096 of them.  Payload: plain 16-byte stores / loads when the vector allows (the mailbox is
// fine-grained memory; the system-scope release fence before the flag store and the acquire fence after the poll order them - the
// RCCL LL128-free "simple" protocol's discipline), 4-byte system-scope accesses otherwise.
__global__ __launch_bounds__(1024) void ipc_exchange_kernel(IpcArgs a) {
  __shared__ unsigned s_seq;
  __shared__ int s_bad;
  const int tid = threadIdx.x, g = blockIdx.x;
  if (tid == 0) {
    s_seq = a.ctl[g] + 1;
    a.ctl[g] = s_seq;
    // poisoned (an exchange of this communicator timed out here or on a peer): no protocol, NaN out
    s_bad = __hip_atomic_load(a.poison[a.rank], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) != 0u ? 2 : 0;
  }
  __syncthreads();
  const unsigned seq = s_seq;
  const int par = (int)(seq & 1u);
  const int lo = g * kIpcChunk, n = min(a.count - lo, kIpcChunk);          // my chunk: floats [lo, lo + n)
  const bool vec = a.vec != 0;
  if (s_bad == 0) {                                                        // (block-uniform)
    // 1. my chunk into slot `rank` of every rank's mailbox (my own included)
    if (vec) {
      const float4* src = reinterpret_cast<const float4*>(a.send + lo);
      if (tid * 4 < n) {
        const float4 v = src[tid];
        for (int p = 0; p < a.world; ++p)
          reinterpret_cast<float4*>(a.box[p] + ((size_t)par * a.world + a.rank) * a.slot + lo)[tid] = v;
      }
    } else {
      for (int p = 0; p < a.world; ++p) {
        float* dst = a.box[p] + ((size_t)par * a.world + a.rank) * a.slot + lo;
        for (int i = tid; i < n; i += 1024) st_sys(dst + i, a.send[lo + i]);
      }
    }
    __threadfence_system();
    __syncthreads();
    // 2. the sequence number into every rank's flag word for (chunk, me); 3. wait for everyone's in mine
    if (tid < a.world) {
      const size_t fo = ((size_t)par * a.groups + g) * a.world;
      __hip_atomic_store(a.flags[tid] + fo + a.rank, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
      const unsigned* mine = a.flags[a.rank] + fo + tid;
      const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
      while (__hip_atomic_load(mine, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM) != seq) {
        if (__builtin_amdgcn_s_memrealtime() - t0 > a.timeout) {
          s_bad = 1;
          break;
        }
        if (__hip_atomic_load(a.poison[a.rank], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) != 0u) {   // a peer gave up: so do I, now
          s_bad = 2;
          break;
        }
        __builtin_amdgcn_s_sleep(2);
      }
    }
    __syncthreads();
    __threadfence_system();
    if (s_bad == 1 && tid < a.world)                                       // my timeout: poison every rank's communicator, mine included
      __hip_atomic_store(a.poison[tid], seq | 0x80000000u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  }
  if (s_bad) {
    if (tid == 0) {
      atomicAdd(a.ctl + a.groups, 1u);
      if (__hip_atomic_load(a.host_to, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) == 0u)
        __hip_atomic_store(a.host_to, seq ? seq : 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
    // the result of an exchange that did not complete is NaN, never a sum of whatever the slots hold
    const float qnan = __uint_as_float(0x7FC00000u);
    const int reps = a.gather ? a.world : 1;
    for (int r = 0; r < reps; ++r)
      for (int i = tid; i < n; i += 1024) a.out[(size_t)r * a.count + lo + i] = qnan;
    return;
  }
  // 4. combine in rank order (identical on every rank)
  const float* box = a.box[a.rank] + (size_t)par * a.world * a.slot + lo;
  if (vec) {
    if (tid * 4 < n) {
      if (a.gather) {
        for (int r = 0; r < a.world; ++r)
          reinterpret_cast<float4*>(a.out + (size_t)r * a.count + lo)[tid] = reinterpret_cast<const float4*>(box + (size_t)r * a.slot)[tid];
      } else {
        float4 v[4];
        float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
        for (int r0 = 0; r0 < a.world; r0 += 4) {                          // four loads in flight, summed in rank order
#pragma unroll
          for (int j = 0; j < 4; ++j)
            if (r0 + j < a.world) v[j] = reinterpret_cast<const float4*>(box + (size_t)(r0 + j) * a.slot)[tid];
#pragma unroll
          for (int j = 0; j < 4; ++j)
            if (r0 + j < a.world) { s.x += v[j].x; s.y += v[j].y; s.z += v[j].z; s.w += v[j].w; }
        }
        reinterpret_cast<float4*>(a.out + lo)[tid] = s;
      }
    }
  } else if (a.gather) {
    for (int r = 0; r < a.world; ++r)
      for (int i = tid; i < n; i += 1024) a.out[(size_t)r * a.count + lo + i] = ld_sys(box + (size_t)r * a.slot + i);
  } else {
    for (int i = tid; i < n; i += 1024) {
      float s = 0.f;
      for (int r = 0; r < a.world; ++r) s += ld_sys(box + (size_t)r * a.slot + i);
      a.out[lo + i] = s;
    }
  }
}

int ipc_exchange(Comm* c, const float* send, float* out, size_t count, int gather, hipStream_t s, const char* fn) {
  IpcState* st = c->ipc;
  const unsigned bad = *reinterpret_cast<volatile unsigned*>(st->host_to);
  UCD_REQUIRE(bad == 0, UCD_ETIMEOUT, "%s: mailbox exchange %u of this communicator timed out (a rank never wrote its vector)", fn, bad);
  IpcArgs a;
  for (int p = 0; p < st->world; ++p) {
    a.box[p] = reinterpret_cast<float*>(st->peer[p]);
    a.flags[p] = reinterpret_cast<unsigned*>(reinterpret_cast<char*>(st->peer[p]) + st->box_bytes);
    // ctl = [groups sequence counters | timeouts | poison] behind the flags, at the same offset in every rank's mailbox
    a.poison[p] = a.flags[p] + (size_t)2 * st->groups * st->world + st->groups + 1;
  }
  a.ctl = reinterpret_cast<unsigned*>(reinterpret_cast<char*>(st->base) + st->box_bytes + (size_t)2 * st->groups * st->world * sizeof(unsigned));
  a.host_to = st->host_to;
  a.send = send; a.out = out;
  a.world = st->world; a.rank = st->rank; a.slot = st->slot; a.count = (int)count; a.gather = gather; a.groups = st->groups;
  a.vec = (count % 4 == 0 && st->slot % 4 == 0 && (((uintptr_t)send | (uintptr_t)out) & 15) == 0) ? 1 : 0;
  a.timeout = st->timeout_ticks;
  if (count == 0) return 0;
  ipc_exchange_kernel<<<(int)((count + kIpcChunk - 1) / kIpcChunk), 1024, 0, s>>>(a);
  return check_launch(fn);
}

}  // namespace

int comm_all_gather_f32(void* comm, const float* send, float* recv, size_t count, hipStream_t s) {
  Comm* c = reinterpret_cast<Comm*>(comm);
  UCD_REQUIRE(c && c->magic == kCommMagic, UCD_EINVAL, "ucd_comm: not a communicator of this library");
  if (c->ipc && count <= (size_t)c->ipc->slot) return ipc_exchange(c, send, recv, count, 1, s, "ucd_comm_all_gather");
  UCD_REQUIRE(g_rccl.AllGather && c->nccl, UCD_EINVAL, "ucd_comm_all_gather: %zu floats exceed the mailbox slot and there is no RCCL communicator", count);
  UCD_RCCL("ncclAllGather", g_rccl.AllGather(send, recv, count, ncclFloat32, c->nccl, s));
  return 0;
}

int comm_all_reduce_sum_f32(void* comm, float* buf, size_t count, hipStream_t s) {
  Comm* c = reinterpret_cast<Comm*>(comm);
  UCD_REQUIRE(c && c->magic == kCommMagic, UCD_EINVAL, "ucd_comm: not a communicator of this library");
  if (c->ipc && count <= (size_t)c->ipc->slot) return ipc_exchange(c, buf, buf, count, 0, s, "ucd_comm_all_reduce_sum");
  UCD_REQUIRE(g_rccl.AllReduce && c->nccl, UCD_EINVAL, "ucd_comm_all_reduce_sum: %zu floats exceed the mailbox slot and there is no RCCL communicator", count);
  UCD_RCCL("ncclAllReduce", g_rccl.AllReduce(buf, buf, count, ncclFloat32, ncclSum, c->nccl, s));
  return 0;
}

}  // namespace ucd

using namespace ucd;

extern "C" {

int ucd_comm_load(const char* path) {
  if (g_rccl.handle) return 0;
  const char* candidates[] = {path, "librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
  void* h = nullptr;
  for (const char* c : candidates) {
    if (!c || !*c) continue;
    h = dlopen(c, RTLD_NOW | RTLD_LOCAL);
    if (h) break;
  }
  UCD_REQUIRE(h, UCD_EINVAL, "ucd_comm_load: cannot open RCCL (%s)", dlerror());
#define BIND(field, sym)                                                                  \
  g_rccl.field = reinterpret_cast<decltype(g_rccl.field)>(dlsym(h, sym));                 \
  UCD_REQUIRE(g_rccl.field, UCD_EINVAL, "ucd_comm_load: symbol %s not found", sym)
  BIND(GetUniqueId, "ncclGetUniqueId");
  BIND(CommInitRank, "ncclCommInitRank");
  BIND(CommDestroy, "ncclCommDestroy");
  BIND(AllGather, "ncclAllGather");
  BIND(AllReduce, "ncclAllReduce");
  BIND(GetErrorString, "ncclGetErrorString");
#undef BIND
  g_rccl.handle = h;
  return 0;
}

int ucd_comm_unique_id(void* id_out, size_t bytes) {
  UCD_REQUIRE(g_rccl.handle, UCD_EINVAL, "ucd_comm_unique_id: call ucd_comm_load first");
  UCD_REQUIRE(id_out && bytes >= sizeof(ncclUniqueId), UCD_EINVAL, "ucd_comm_unique_id: need %zu bytes", sizeof(ncclUniqueId));
  UCD_RCCL("ncclGetUniqueId", g_rccl.GetUniqueId(reinterpret_cast<ncclUniqueId*>(id_out)));
  return 0;
}

int ucd_comm_init(const void* id, size_t bytes, int nranks, int rank, ucd_comm_t* comm_out) {
  UCD_REQUIRE(g_rccl.handle, UCD_EINVAL, "ucd_comm_init: call ucd_comm_load first");
  UCD_REQUIRE(id && bytes >= sizeof(ncclUniqueId) && comm_out && nranks >= 1 && rank >= 0 && rank < nranks, UCD_EINVAL,
              "ucd_comm_init: bad arguments");
  ncclUniqueId uid;
  memcpy(&uid, id, sizeof uid);
  ncclComm_t c = nullptr;
  UCD_RCCL("ncclCommInitRank", g_rccl.CommInitRank(&c, nranks, uid, rank));
  Comm* cm = new Comm();
  cm->nccl = c; cm->world = nranks; cm->rank = rank;
  *comm_out = cm;
  return 0;
}

int ucd_comm_init_local(int nranks, int rank, ucd_comm_t* comm_out) {
  UCD_REQUIRE(comm_out && nranks >= 1 && rank >= 0 && rank < nranks, UCD_EINVAL, "ucd_comm_init_local: bad arguments");
  Comm* cm = new Comm();
  cm->world = nranks; cm->rank = rank;
  *comm_out = cm;
  return 0;
}

static void ipc_free(IpcState* st) {
  if (!st) return;
  for (int p = 0; p < st->world; ++p)
    if (p != st->rank && st->peer[p] && st->peer[p] != st->base) (void)hipIpcCloseMemHandle(st->peer[p]);
  if (st->base) (void)hipFree(st->base);
  if (st->host_to) (void)hipHostFree(st->host_to);
  (void)hipGetLastError();
  delete st;
}

int ucd_comm_destroy(ucd_comm_t comm) {
  Comm* c = reinterpret_cast<Comm*>(comm);
  if (!c || c->magic != kCommMagic) return 0;
  ipc_free(c->ipc);
  c->ipc = nullptr;
  ncclComm_t n = c->nccl;
  c->magic = 0;
  delete c;
  if (n && g_rccl.CommDestroy) UCD_RCCL("ncclCommDestroy", g_rccl.CommDestroy(n));
  return 0;
}

size_t ucd_comm_ipc_handle_bytes(void) { return sizeof(hipIpcMemHandle_t); }

int ucd_comm_ipc_create(ucd_comm_t comm, int slot_floats, int timeout_ms, void* handle_out) {
  static const char* fn = "ucd_comm_ipc_create";
  Comm* c = reinterpret_cast<Comm*>(comm);
  UCD_REQUIRE(c && c->magic == kCommMagic && handle_out && slot_floats > 0, UCD_EINVAL, "%s: bad arguments", fn);
  UCD_REQUIRE(c->world <= kIpcMaxWorld, UCD_EUNSUPPORTED, "%s: at most %d ranks", fn, kIpcMaxWorld);
  UCD_REQUIRE(!c->ipc, UCD_EINVAL, "%s: the communicator has a mailbox already", fn);
  IpcState* st = new IpcState();
  st->world = c->world; st->rank = c->rank; st->slot = slot_floats;
  st->box_bytes = align_up((size_t)2 * st->world * slot_floats * sizeof(float), 256);
  st->groups = (slot_floats + kIpcChunk - 1) / kIpcChunk;
  st->bytes = st->box_bytes + align_up((size_t)2 * st->groups * st->world * sizeof(unsigned) + (size_t)(st->groups + 2) * sizeof(unsigned) + 64, 256);
  if (timeout_ms > 0) st->timeout_ticks = (unsigned long long)timeout_ms * 100000ull;
  // fine-grained device memory or no mailbox (round 6): the 16-byte payload path uses PLAIN loads and stores between system-scope
  // fences, which is only a protocol on memory the caches do not keep private copies of; on coarse-grained memory a peer's xGMI
  // stores are not guaranteed to be seen by a plain load through this device's L2.  The caller keeps RCCL when this fails.
  hipError_t e = hipExtMallocWithFlags(&st->base, st->bytes, hipDeviceMallocFinegrained);
  if (e == hipSuccess) e = hipMemset(st->base, 0, st->bytes);
  if (e == hipSuccess) e = hipHostMalloc(reinterpret_cast<void**>(&st->host_to), 64, hipHostMallocMapped);
  if (e == hipSuccess) { *st->host_to = 0; e = hipDeviceSynchronize(); }
  hipIpcMemHandle_t h;
  if (e == hipSuccess) e = hipIpcGetMemHandle(&h, st->base);
  if (e != hipSuccess) {
    (void)hipGetLastError();
    set_error("%s: %s", fn, hipGetErrorString(e));
    ipc_free(st);
    return (int)e;
  }
  memcpy(handle_out, &h, sizeof h);
  st->peer[st->rank] = st->base;
  c->ipc = st;
  return 0;
}

int ucd_comm_ipc_connect(ucd_comm_t comm, const void* handles) {
  static const char* fn = "ucd_comm_ipc_connect";
  Comm* c = reinterpret_cast<Comm*>(comm);
  UCD_REQUIRE(c && c->magic == kCommMagic && c->ipc && handles, UCD_EINVAL, "%s: bad arguments (ucd_comm_ipc_create first)", fn);
  IpcState* st = c->ipc;
  for (int p = 0; p < st->world; ++p) {
    if (p == st->rank) continue;
    hipIpcMemHandle_t h;
    memcpy(&h, reinterpret_cast<const char*>(handles) + (size_t)p * sizeof h, sizeof h);
    void* ptr = nullptr;
    hipError_t e = hipIpcOpenMemHandle(&ptr, h, hipIpcMemLazyEnablePeerAccess);
    if (e != hipSuccess) {
      (void)hipGetLastError();
      set_error("%s: hipIpcOpenMemHandle(rank %d): %s", fn, p, hipGetErrorString(e));
      return (int)e;
    }
    st->peer[p] = ptr;
  }
  return 0;
}

int ucd_comm_ipc_drop(ucd_comm_t comm) {
  Comm* c = reinterpret_cast<Comm*>(comm);
  if (!c || c->magic != kCommMagic) return 0;
  ipc_free(c->ipc);
  c->ipc = nullptr;
  return 0;
}

unsigned ucd_comm_ipc_timeouts(ucd_comm_t comm) {
  Comm* c = reinterpret_cast<Comm*>(comm);
  if (!c || c->magic != kCommMagic || !c->ipc) return 0;
  return *reinterpret_cast<volatile unsigned*>(c->ipc->host_to);
}

int ucd_comm_all_gather(ucd_comm_t comm, const float* send, float* recv, size_t count, ucd_stream_t stream) {
  return comm_all_gather_f32(comm, send, recv, count, (hipStream_t)stream);
}

int ucd_comm_all_reduce_sum(ucd_comm_t comm, float* buf, size_t count, ucd_stream_t stream) {
  return comm_all_reduce_sum_f32(comm, buf, count, (hipStream_t)stream);
}

}  // extern "C"
