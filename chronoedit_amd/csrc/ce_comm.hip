// A RCCL communicator OWNED by this library ("one ncclComm_t per process passed in"): the exchanges of the
// sequence-parallel forward (chronoedit_amd/parallel.py) as plain C-ABI calls on the caller's stream - no torch.distributed `Work`
// objects, no process-group watchdog thread polling events, so a sharded denoising step can be captured into a hipGraph
// (profiles/r03_rccl_graph_probe.txt: what dies under capture with torch's "nccl" backend is ProcessGroupNCCL's watchdog, not RCCL).
//
// RCCL is NOT linked: the process already holds one copy (the librccl.so torch ships and loads); ce_comm_load() dlopen()s the path the
// host side hands over and resolves the eight entry points used here.  Everything below is host code; the collectives themselves are
// RCCL's kernels over xGMI.
#include <dlfcn.h>
#include <rccl/rccl.h>  // types only (ncclComm_t, ncclUniqueId, ncclResult_t, ncclDataType_t)

#include <cstdio>
#include <cstring>
#include <mutex>

#include "ce_common.h"

namespace {

struct RcclApi {
  void* handle = nullptr;
  ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
  ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
  ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
  ncclResult_t (*GroupStart)() = nullptr;
  ncclResult_t (*GroupEnd)() = nullptr;
  ncclResult_t (*Send)(const void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*Recv)(void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*AllGather)(const void*, void*, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
  const char* (*GetErrorString)(ncclResult_t) = nullptr;
};
RcclApi g_rccl;
std::mutex g_rccl_mutex;

struct CeComm {
  ncclComm_t comm;
  int rank, world;
};

int fail(const char* what, ncclResult_t r) {
  std::fprintf(stderr, "chronoedit_hip: %s failed: %s\n", what, g_rccl.GetErrorString ? g_rccl.GetErrorString(r) : "(no error string)");
  return CE_ERR_ARG;
}

}  // namespace

CE_API int ce_comm_load(const char* librccl_path) {
  std::lock_guard<std::mutex> lock(g_rccl_mutex);
  if (g_rccl.handle) return CE_OK;
  if (!librccl_path) return CE_ERR_ARG;
  void* h = dlopen(librccl_path, RTLD_NOW | RTLD_LOCAL);
  if (!h) {
    std::fprintf(stderr, "chronoedit_hip: dlopen(%s): %s\n", librccl_path, dlerror());
    return CE_ERR_ARG;
  }
  RcclApi a;
  a.handle = h;
#define CE_SYM(F, N)                                                      \
  a.F = reinterpret_cast<decltype(a.F)>(dlsym(h, N));                     \
  if (!a.F) {                                                             \
    std::fprintf(stderr, "chronoedit_hip: %s lacks %s\n", librccl_path, N); \
    dlclose(h);                                                           \
    return CE_ERR_ARG;                                                    \
  }
  CE_SYM(GetUniqueId, "ncclGetUniqueId")
  CE_SYM(CommInitRank, "ncclCommInitRank")
  CE_SYM(CommDestroy, "ncclCommDestroy")
  CE_SYM(GroupStart, "ncclGroupStart")
  CE_SYM(GroupEnd, "ncclGroupEnd")
  CE_SYM(Send, "ncclSend")
  CE_SYM(Recv, "ncclRecv")
  CE_SYM(AllGather, "ncclAllGather")
  CE_SYM(GetErrorString, "ncclGetErrorString")
#undef CE_SYM
  g_rccl = a;
  return CE_OK;
}

CE_API int ce_comm_unique_id(void* id128) {
  if (!g_rccl.handle || !id128) return CE_ERR_ARG;
  static_assert(sizeof(ncclUniqueId) == 128, "ncclUniqueId is 128 bytes");
  ncclUniqueId id;
  const ncclResult_t r = g_rccl.GetUniqueId(&id);
  if (r != ncclSuccess) return fail("ncclGetUniqueId", r);
  std::memcpy(id128, &id, sizeof id);
  return CE_OK;
}

CE_API int ce_comm_init(void** comm_out, const void* id128, int rank, int world) {
  if (!g_rccl.handle || !comm_out || !id128 || world < 1 || rank < 0 || rank >= world) return CE_ERR_ARG;
  ncclUniqueId id;
  std::memcpy(&id, id128, sizeof id);
  ncclComm_t c;
  const ncclResult_t r = g_rccl.CommInitRank(&c, world, id, rank);  // (collective: every rank of the group calls it; binds the current device)
  if (r != ncclSuccess) return fail("ncclCommInitRank", r);
  *comm_out = new CeComm{c, rank, world};
  return CE_OK;
}

CE_API int ce_comm_destroy(void* comm) {
  if (!comm) return CE_OK;
  CeComm* c = static_cast<CeComm*>(comm);
  const ncclResult_t r = g_rccl.handle ? g_rccl.CommDestroy(c->comm) : ncclSuccess;
  delete c;
  return r == ncclSuccess ? CE_OK : fail("ncclCommDestroy", r);
}

// send / recv: `world` chunks of bytes_per_peer bytes each, chunk p going to / coming from rank p (the layout of
// torch.distributed.all_to_all_single with equal splits).  One grouped batch of ncclSend / ncclRecv pairs on `stream`.
CE_API int ce_comm_all_to_all(void* comm, const void* send, void* recv, size_t bytes_per_peer, hipStream_t stream) {
  if (!g_rccl.handle || !comm || !send || !recv || bytes_per_peer == 0) return CE_ERR_ARG;
  CeComm* c = static_cast<CeComm*>(comm);
  ncclResult_t r = g_rccl.GroupStart();
  if (r != ncclSuccess) return fail("ncclGroupStart", r);
  for (int p = 0; p < c->world && r == ncclSuccess; ++p) {
    r = g_rccl.Send(static_cast<const char*>(send) + (size_t)p * bytes_per_peer, bytes_per_peer, ncclInt8, p, c->comm, stream);
    if (r == ncclSuccess) r = g_rccl.Recv(static_cast<char*>(recv) + (size_t)p * bytes_per_peer, bytes_per_peer, ncclInt8, p, c->comm, stream);
  }
  const ncclResult_t e = g_rccl.GroupEnd();
  if (r != ncclSuccess) return fail("ncclSend / ncclRecv", r);
  if (e != ncclSuccess) return fail("ncclGroupEnd", e);
  return CE_OK;
}

// recv = the ranks' `bytes_per_rank`-byte blocks in rank order.
CE_API int ce_comm_all_gather(void* comm, const void* send, void* recv, size_t bytes_per_rank, hipStream_t stream) {
  if (!g_rccl.handle || !comm || !send || !recv || bytes_per_rank == 0) return CE_ERR_ARG;
  CeComm* c = static_cast<CeComm*>(comm);
  const ncclResult_t r = g_rccl.AllGather(send, recv, bytes_per_rank, ncclInt8, c->comm, stream);
  return r == ncclSuccess ? CE_OK : fail("ncclAllGather", r);
}
