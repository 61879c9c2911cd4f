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
  decltype(&ncclSend) Send = nullptr;
  decltype(&ncclRecv) Recv = nullptr;
  decltype(&ncclGroupStart) GroupStart = nullptr;
  decltype(&ncclGroupEnd) GroupEnd = nullptr;
  decltype(&ncclCommSplit) CommSplit = nullptr;   // optional: absent -> the side transfers share the chain communicator
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
    BAE_SYM(Send, ncclSend)
    BAE_SYM(Recv, ncclRecv)
    BAE_SYM(GroupStart, ncclGroupStart)
    BAE_SYM(GroupEnd, ncclGroupEnd)
#undef BAE_SYM
    api.CommSplit = reinterpret_cast<decltype(api.CommSplit)>(dlsym(api.lib, "ncclCommSplit"));
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
  return hipStreamSynchronize(e->stream) == hipSuccess ? 0 : 1;  // (bytes are counted by the callers' wrapper)
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
  (root == e->rank ? e->cstats.chain_bytes_sent : e->cstats.chain_bytes_recv) += 8.0 * (double)count;
  e->cstats.chain_messages++;
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

int dist_exchange(Engine* e, const std::vector<DistXfer>& x, bool side, hipStream_t s) {
  if (x.empty()) return 0;
  for (const DistXfer& t : x) {
    double& ctr = side ? (t.send ? e->cstats.side_bytes_sent : e->cstats.side_bytes_recv)
                       : (t.send ? e->cstats.chain_bytes_sent : e->cstats.chain_bytes_recv);
    ctr += 8.0 * (double)t.count;
    if (t.send) (side ? e->cstats.side_messages : e->cstats.chain_messages)++;
  }
  if (e->comm && e->coll == native_collective) {
    RcclApi* a = rccl();
    ncclComm_t comm = static_cast<ncclComm_t>(side && e->comm2 ? e->comm2 : e->comm);
    ncclResult_t r = a->GroupStart();
    if (r != ncclSuccess) return rccl_fail(e, "ncclGroupStart", r);
    for (const DistXfer& t : x) {
      r = t.send ? a->Send(t.buf, t.count, ncclDouble, t.peer, comm, s) : a->Recv(t.buf, t.count, ncclDouble, t.peer, comm, s);
      if (r != ncclSuccess) { (void)a->GroupEnd(); return rccl_fail(e, t.send ? "ncclSend" : "ncclRecv", r); }
    }
    r = a->GroupEnd();
    if (r != ncclSuccess) return rccl_fail(e, "ncclGroupEnd", r);
    return 0;
  }
  // hook: sends never block on the receiver (contract of op 3), so "all sends, then all receives" cannot deadlock
  BAE_HIP(hipStreamSynchronize(s));
  for (const DistXfer& t : x)
    if (t.send && e->coll(e->coll_ctx, 3, t.buf, t.count, t.peer) != 0) return e->fail_msg("send hook failed");
  for (const DistXfer& t : x)
    if (!t.send && e->coll(e->coll_ctx, 4, t.buf, t.count, t.peer) != 0) return e->fail_msg("receive hook failed");
  // op 5: end of the exchange — a hook that defers its sends (to batch them with the receives) must have started them
  // all by the time it returns from this call; hooks that send eagerly ignore it
  if (e->coll(e->coll_ctx, 5, nullptr, 0, 0) != 0) return e->fail_msg("exchange flush hook failed");
  return 0;
}

int dist_allreduce_stream(Engine* e, double* buf, size_t count, hipStream_t s) {
  e->cstats.allreduce_bytes += 8.0 * (double)count;
  if (e->comm && e->allreduce == native_allreduce) {
    RcclApi* a = rccl();
    const ncclResult_t r = a->AllReduce(buf, buf, count, ncclDouble, ncclSum, static_cast<ncclComm_t>(e->comm), s);
    if (r != ncclSuccess) return rccl_fail(e, "ncclAllReduce", r);
    return 0;
  }
  BAE_HIP(hipStreamSynchronize(s));
  if (e->allreduce(e->allreduce_ctx, buf, count, 0) != 0) return e->fail_msg("allreduce hook failed");
  return 0;
}

void comm_release(Engine* e) {
  if (!e->comm) return;
  if (RcclApi* a = rccl()) {
    if (e->comm2) (void)a->CommDestroy(static_cast<ncclComm_t>(e->comm2));
    (void)a->CommDestroy(static_cast<ncclComm_t>(e->comm));
  }
  e->comm = nullptr;
  e->comm2 = nullptr;
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
  // side communicator of the distributed solve (the bulk of every panel travels on its own stream, beside
  // the chain's messages): a duplicate of the first one.  Collective over all ranks, like the init itself.
  e->comm2 = nullptr;
  if (a->CommSplit && !getenv("BA_HIP_ONE_COMM")) {
    ncclComm_t c2 = nullptr;
    const ncclResult_t r2 = a->CommSplit(comm, 0, rank, &c2, nullptr);
    if (r2 != ncclSuccess) return rccl_fail(e, "ncclCommSplit", r2);
    e->comm2 = c2;
  }
  e->allreduce = native_allreduce; e->allreduce_ctx = e;
  e->coll = native_collective; e->coll_ctx = e;
  e->rank = rank; e->nranks = nranks;
  e->comm_force = nranks == 1;   // one rank: still run the sharded code paths (test of the RCCL calls)
  e->nzL_valid = false;          // the tile pattern of S is the union over the shards
  e->dist_plan_version = ~0ull;
  e->dog_jrhs_valid = false;
  return 0;
}

int ba_hip_comm_destroy(ba_hip_engine* h) {
  Engine* e = reinterpret_cast<Engine*>(h);
  comm_release(e);
  e->allreduce = nullptr; e->allreduce_ctx = nullptr; e->coll = nullptr; e->coll_ctx = nullptr;
  e->rank = 0; e->nranks = 1; e->comm_force = false;
  e->nzL_valid = false;
  e->dist_plan_version = ~0ull;
  e->dog_jrhs_valid = false;
  return 0;
}

}  // extern "C"
