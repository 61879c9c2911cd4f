// Native RCCL communicator of the engine (one process per GPU, collectives over xGMI): the
// cross-shard sums of the landmark-sharded path and the collectives of the distributed reduced
// solve, called directly from the engine instead of through caller-supplied hooks
// (include/ba_hip.h: ba_hip_comm_*).  Replaces nothing in the reference — arpg/ba is single-process
// (SURVEY.md §8e); what is summed is the reduced pose system of BundleAdjuster.cpp:409-485.
//
// librccl is loaded with dlopen on first use (soname librccl.so.1): the engine has no link-time
// dependency on it, and in a process that already carries RCCL (PyTorch-ROCm) the loaded copy is reused.
#include "engine.h"

#include <dlfcn.h>
#include <rccl/rccl.h>

#include <cstring>
#include <mutex>

namespace bae {

namespace {
struct RcclApi {
  void* lib = nullptr;
  decltype(&ncclGetUniqueId) GetUniqueId = nullptr;
  decltype(&ncclCommInitRank) CommInitRank = nullptr;
  decltype(&ncclCommDestroy) CommDestroy = nullptr;
  decltype(&ncclAllReduce) AllReduce = nullptr;
  decltype(&ncclBroadcast) Broadcast = nullptr;
  decltype(&ncclReduceScatter) ReduceScatter = nullptr;
  decltype(&ncclGetErrorString) GetErrorString = nullptr;
  std::string err;
};

RcclApi* rccl() {
  static RcclApi api;
  static std::once_flag once;
  std::call_once(once, [] {
    for (const char* name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
      api.lib = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
      if (api.lib) break;
    }
    if (!api.lib) { api.err = std::string("dlopen(librccl): ") + dlerror(); return; }
#define BAE_SYM(field, sym)                                                         \
  api.field = reinterpret_cast<decltype(api.field)>(dlsym(api.lib, #sym));          \
  if (!api.field) { api.err = "librccl lacks " #sym; api.lib = nullptr; return; }
    BAE_SYM(GetUniqueId, ncclGetUniqueId)
    BAE_SYM(CommInitRank, ncclCommInitRank)
    BAE_SYM(CommDestroy, ncclCommDestroy)
    BAE_SYM(AllReduce, ncclAllReduce)
    BAE_SYM(Broadcast, ncclBroadcast)
    BAE_SYM(ReduceScatter, ncclReduceScatter)
    BAE_SYM(GetErrorString, ncclGetErrorString)
#undef BAE_SYM
  });
  return api.lib ? &api : nullptr;
}

int rccl_fail(Engine* e, const char* what, ncclResult_t r) {
  RcclApi* a = rccl();
  e->err = std::string(what) + ": " + (a ? a->GetErrorString(r) : "librccl not loaded");
  return -1;
}

// all-reduce hook signature (ba_hip_allreduce_fn): SUM over the ranks, completed on return
int native_allreduce(void* ctx, void* dev_ptr, size_t count, int dtype) {
  Engine* e = static_cast<Engine*>(ctx);
  RcclApi* a = rccl();
  if (!a || !e->comm) return 1;
  const ncclResult_t r = a->AllReduce(dev_ptr, dev_ptr, count, dtype == 0 ? ncclDouble : ncclUint64, ncclSum,
                                      static_cast<ncclComm_t>(e->comm), e->stream);
  if (r != ncclSuccess) { rccl_fail(e, "ncclAllReduce", r); return 1; }
  return hipStreamSynchronize(e->stream) == hipSuccess ? 0 : 1;
}

// collectives hook signature (ba_hip_collective_fn): op 1 broadcast, op 2 in-place reduce-scatter
int native_collective(void* ctx, int op, void* dev_ptr, size_t count, int root) {
  Engine* e = static_cast<Engine*>(ctx);
  RcclApi* a = rccl();
  if (!a || !e->comm) return 1;
  ncclComm_t comm = static_cast<ncclComm_t>(e->comm);
  ncclResult_t r;
  if (op == 1) {
    r = a->Broadcast(dev_ptr, dev_ptr, count, ncclDouble, root, comm, e->stream);
  } else if (op == 2) {
    // in-place form: the receive buffer is this rank's chunk of the send buffer
    double* p = static_cast<double*>(dev_ptr);
    r = a->ReduceScatter(p, p + (size_t)e->rank * count, count, ncclDouble, ncclSum, comm, e->stream);
  } else {
    return 1;
  }
  if (r != ncclSuccess) { rccl_fail(e, op == 1 ? "ncclBroadcast" : "ncclReduceScatter", r); return 1; }
  return hipStreamSynchronize(e->stream) == hipSuccess ? 0 : 1;
}
}  // namespace

int dist_broadcast(Engine* e, double* buf, size_t count, int root, hipStream_t s) {
  if (e->comm && e->coll == native_collective) {
    RcclApi* a = rccl();
    const ncclResult_t r = a->Broadcast(buf, buf, count, ncclDouble, root, static_cast<ncclComm_t>(e->comm), s);
    if (r != ncclSuccess) return rccl_fail(e, "ncclBroadcast", r);
    return 0;  // stream-ordered: the unpack kernels queue up behind it, no host round trip per panel
  }
  BAE_HIP(hipStreamSynchronize(s));
  if (e->coll(e->coll_ctx, 1, buf, count, root) != 0) return e->fail_msg("broadcast hook failed");
  return 0;
}

void comm_release(Engine* e) {
  if (!e->comm) return;
  if (RcclApi* a = rccl()) (void)a->CommDestroy(static_cast<ncclComm_t>(e->comm));
  e->comm = nullptr;
}

}  // namespace bae

using namespace bae;

extern "C" {

int ba_hip_comm_unique_id(void* id128) {
  if (!id128) return -1;
  RcclApi* a = rccl();
  if (!a) return -1;
  static_assert(sizeof(ncclUniqueId) == 128, "ncclUniqueId is 128 bytes");
  ncclUniqueId id;
  if (a->GetUniqueId(&id) != ncclSuccess) return -1;
  memcpy(id128, &id, sizeof(id));
  return 0;
}

int ba_hip_comm_init(ba_hip_engine* h, const void* id128, int rank, int nranks) {
  Engine* e = reinterpret_cast<Engine*>(h);
  if (!id128 || nranks < 1 || rank < 0 || rank >= nranks) return e->fail_msg("ba_hip_comm_init: bad arguments");
  RcclApi* a = rccl();
  if (!a) return e->fail_msg("ba_hip_comm_init: librccl could not be loaded");
  BAE_HIP(hipSetDevice(e->device));
  comm_release(e);
  ncclUniqueId id;
  memcpy(&id, id128, sizeof(id));
  ncclComm_t comm = nullptr;
  const ncclResult_t r = a->CommInitRank(&comm, nranks, id, rank);
  if (r != ncclSuccess) return rccl_fail(e, "ncclCommInitRank", r);
  e->comm = comm;
  e->allreduce = native_allreduce; e->allreduce_ctx = e;
  e->coll = native_collective; e->coll_ctx = e;
  e->rank = rank; e->nranks = nranks;
  e->comm_force = nranks == 1;   // one rank: still run the sharded code paths (test of the RCCL calls)
  e->nzL_valid = false;          // the tile pattern of S is the union over the shards
  return 0;
}

int ba_hip_comm_destroy(ba_hip_engine* h) {
  Engine* e = reinterpret_cast<Engine*>(h);
  comm_release(e);
  e->allreduce = nullptr; e->allreduce_ctx = nullptr; e->coll = nullptr; e->coll_ctx = nullptr;
  e->rank = 0; e->nranks = 1; e->comm_force = false;
  e->nzL_valid = false;
  return 0;
}

}  // extern "C"
