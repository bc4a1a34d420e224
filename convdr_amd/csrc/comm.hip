// Collectives of the N > 1 paths behind the C ABI: thin wrappers over RCCL's C API (ncclAllGather / ncclAllReduce over
// xGMI), for hosts that are not Python.  The reference's interface for the same exchange is torch.distributed
// (/root/reference/drivers/gen_passage_embeddings.py:314 dist.barrier / DDP; run_convdr_train.py:77-78 nn.DataParallel's
// gather; run_convdr_inference.py:355-368 faiss IndexShards), which is what convdr_amd/parallel.py keeps using; these
// entry points give a torch-free host (tests/capi/ip_search_host.cpp) the same three steps of SURVEY.md section 8(e):
// query all-gather, per-rank top-k all-gather, gradient all-reduce.
//
// librccl.so is NOT a link-time dependency: a process that has torch loaded must use the RCCL torch ships (two copies of
// a collective library in one process do not share communicators), so the library is looked up at the first call --
// whichever librccl.so is already mapped (RTLD_NOLOAD), else the system one.
#include <dlfcn.h>
// RCCL's header is used for its types only (every function is looked up with dlsym): a build host without the header still
// builds the library -- the few ABI-stable declarations the wrappers need are restated below (the header was a
// build dependency of the whole library).
#if __has_include(<rccl/rccl.h>) && !defined(CONVDR_NO_RCCL_HEADER)   // (the macro: to compile the fallback branch on a host that has the header)
#include <rccl/rccl.h>
#else
extern "C" {
typedef struct ncclComm* ncclComm_t;
typedef struct { char internal[128]; } ncclUniqueId;
typedef enum { ncclSuccess = 0 } ncclResult_t;
typedef enum { ncclInt8 = 0, ncclFloat32 = 7 } ncclDataType_t;
typedef enum { ncclSum = 0 } ncclRedOp_t;
}
#endif
#include <stdio.h>
#include <string.h>

#include "common.hpp"
#include "../../include/convdr_hip.h"

namespace convdr {

struct Rccl {
  void* h = nullptr;
  ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
  ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
  ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
  ncclResult_t (*AllGather)(const void*, void*, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*CommCount)(const ncclComm_t, int*) = nullptr;
  ncclResult_t (*CommUserRank)(const ncclComm_t, int*) = nullptr;
  const char* (*GetErrorString)(ncclResult_t) = nullptr;
  bool ok = false;
  char why[256] = "symbols missing";   // the reason `ok` is false (dlerror() is read ONCE, at the failure site: a second call returns NULL)
};

static Rccl load_rccl() {
  Rccl r;
  const char* names[] = {"librccl.so", "librccl.so.1"};
  for (const char* n : names)
    if (!r.h) r.h = dlopen(n, RTLD_NOW | RTLD_NOLOAD | RTLD_GLOBAL);   // the copy this process already uses (torch's)
  for (const char* n : names)
    if (!r.h) r.h = dlopen(n, RTLD_NOW | RTLD_GLOBAL);
  if (!r.h) {
    const char* e = dlerror();
    snprintf(r.why, sizeof r.why, "%s", e ? e : "dlopen failed");
    return r;
  }
  auto sym = [&](const char* s) { return dlsym(r.h, s); };
  r.GetUniqueId = (decltype(r.GetUniqueId))sym("ncclGetUniqueId");
  r.CommInitRank = (decltype(r.CommInitRank))sym("ncclCommInitRank");
  r.CommDestroy = (decltype(r.CommDestroy))sym("ncclCommDestroy");
  r.AllGather = (decltype(r.AllGather))sym("ncclAllGather");
  r.AllReduce = (decltype(r.AllReduce))sym("ncclAllReduce");
  r.CommCount = (decltype(r.CommCount))sym("ncclCommCount");
  r.CommUserRank = (decltype(r.CommUserRank))sym("ncclCommUserRank");
  r.GetErrorString = (decltype(r.GetErrorString))sym("ncclGetErrorString");
  r.ok = r.GetUniqueId && r.CommInitRank && r.CommDestroy && r.AllGather && r.AllReduce && r.CommCount && r.CommUserRank;
  return r;
}

static Rccl& rccl() {
  static Rccl r = load_rccl();   // function-local static: initialised once, thread-safe (C++11)
  return r;
}

#define CONVDR_CHECK_RCCL(expr)                                                                                   \
  do {                                                                                                            \
    ncclResult_t e_ = (expr);                                                                                     \
    if (e_ != ncclSuccess) {                                                                                      \
      set_error("%s failed: %s", #expr, rccl().GetErrorString ? rccl().GetErrorString(e_) : "RCCL error");       \
      return -3;                                                                                                  \
    }                                                                                                             \
  } while (0)

}  // namespace convdr

using namespace convdr;

static_assert(CONVDR_COMM_ID_BYTES == sizeof(ncclUniqueId), "convdr_hip.h: CONVDR_COMM_ID_BYTES must be RCCL's unique-id size");

extern "C" int convdr_comm_unique_id(void* id_out_host) {
  CONVDR_REQUIRE(rccl().ok, "convdr_comm: librccl.so could not be loaded (%s)", rccl().why);
  ncclUniqueId id;
  CONVDR_CHECK_RCCL(rccl().GetUniqueId(&id));
  memcpy(id_out_host, &id, sizeof id);
  return 0;
}

extern "C" int convdr_comm_init(convdr_comm_t* comm, int nranks, int rank, const void* unique_id_host) {
  CONVDR_REQUIRE(rccl().ok, "convdr_comm: librccl.so could not be loaded (%s)", rccl().why);
  CONVDR_REQUIRE(comm && unique_id_host && nranks >= 1 && rank >= 0 && rank < nranks, "convdr_comm_init: bad arguments (rank %d of %d)", rank, nranks);
  ncclUniqueId id;
  memcpy(&id, unique_id_host, sizeof id);
  ncclComm_t c = nullptr;
  CONVDR_CHECK_RCCL(rccl().CommInitRank(&c, nranks, id, rank));
  *comm = (convdr_comm_t)c;
  return 0;
}

extern "C" int convdr_comm_ranks(convdr_comm_t comm, int* nranks, int* rank) {
  CONVDR_REQUIRE(rccl().ok && comm, "convdr_comm_ranks: no communicator");
  if (nranks) CONVDR_CHECK_RCCL(rccl().CommCount((ncclComm_t)comm, nranks));
  if (rank) CONVDR_CHECK_RCCL(rccl().CommUserRank((ncclComm_t)comm, rank));
  return 0;
}

extern "C" int convdr_comm_allgather(convdr_comm_t comm, const void* send, void* recv, size_t bytes_per_rank,
                                     convdr_stream_t stream) {
  CONVDR_REQUIRE(rccl().ok && comm, "convdr_comm_allgather: no communicator");
  CONVDR_CHECK_RCCL(rccl().AllGather(send, recv, bytes_per_rank, ncclInt8, (ncclComm_t)comm, (hipStream_t)stream));
  return 0;
}

extern "C" int convdr_comm_allreduce_f32(convdr_comm_t comm, const float* send, float* recv, size_t count,
                                         convdr_stream_t stream) {
  CONVDR_REQUIRE(rccl().ok && comm, "convdr_comm_allreduce_f32: no communicator");
  CONVDR_CHECK_RCCL(rccl().AllReduce(send, recv, count, ncclFloat32, ncclSum, (ncclComm_t)comm, (hipStream_t)stream));
  return 0;
}

extern "C" int convdr_comm_destroy(convdr_comm_t comm) {
  if (!comm) return 0;
  CONVDR_REQUIRE(rccl().ok, "convdr_comm_destroy: librccl.so is not loaded");
  CONVDR_CHECK_RCCL(rccl().CommDestroy((ncclComm_t)comm));
  return 0;
}
