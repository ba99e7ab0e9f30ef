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

}  // namespace

int comm_all_gather_f32(void* comm, const float* send, float* recv, size_t count, hipStream_t s) {
  UCD_REQUIRE(g_rccl.AllGather && comm, UCD_EINVAL, "ucd_comm: RCCL not loaded or NULL communicator");
  UCD_RCCL("ncclAllGather", g_rccl.AllGather(send, recv, count, ncclFloat32, (ncclComm_t)comm, s));
  return 0;
}

int comm_all_reduce_sum_f32(void* comm, float* buf, size_t count, hipStream_t s) {
  UCD_REQUIRE(g_rccl.AllReduce && comm, UCD_EINVAL, "ucd_comm: RCCL not loaded or NULL communicator");
  UCD_RCCL("ncclAllReduce", g_rccl.AllReduce(buf, buf, count, ncclFloat32, ncclSum, (ncclComm_t)comm, s));
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
  *comm_out = c;
  return 0;
}

int ucd_comm_destroy(ucd_comm_t comm) {
  if (!comm || !g_rccl.CommDestroy) return 0;
  UCD_RCCL("ncclCommDestroy", g_rccl.CommDestroy((ncclComm_t)comm));
  return 0;
}

int ucd_comm_all_gather(ucd_comm_t comm, const float* send, float* recv, size_t count, ucd_stream_t stream) {
  return comm_all_gather_f32(comm, send, recv, count, (hipStream_t)stream);
}

int ucd_comm_all_reduce_sum(ucd_comm_t comm, float* buf, size_t count, ucd_stream_t stream) {
  return comm_all_reduce_sum_f32(comm, buf, count, (hipStream_t)stream);
}

}  // extern "C"
