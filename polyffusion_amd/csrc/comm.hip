// comm.hip - the one exchange of the path: start-up broadcast of the packed weight blob over RCCL / xGMI.
//
// librccl.so is NOT linked: it is dlopen()ed the first time a pf_comm_* entry is called, so single-GPU processes (and the
// CPU-only build check) never load it.  Only four RCCL entry points are used - ncclGetUniqueId, ncclCommInitRank,
// ncclBroadcast, ncclCommDestroy - through their documented C signatures (rccl.h; "nccl" names on ROCm).  There is no
// collective in the denoising step loop, hence nothing else here.
#include <dlfcn.h>

#include <mutex>

#include "pf_internal.h"

namespace {

typedef struct ncclComm* ncclComm_t;
typedef struct { char internal[128]; } ncclUniqueId;
enum { ncclSuccess = 0, ncclChar = 0 };

struct Rccl {
  void* handle = nullptr;
  int (*GetUniqueId)(ncclUniqueId*) = nullptr;
  int (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
  int (*Broadcast)(const void*, void*, size_t, int, int, ncclComm_t, hipStream_t) = nullptr;
  int (*CommDestroy)(ncclComm_t) = nullptr;
  const char* (*GetErrorString)(int) = nullptr;
};

Rccl g_rccl;
std::once_flag g_once;
std::string g_load_error;

void load_rccl() {
  const char* names[] = {"librccl.so", "librccl.so.1", "/opt/rocm/lib/librccl.so"};
  for (const char* n : names) {
    g_rccl.handle = dlopen(n, RTLD_NOW | RTLD_LOCAL);
    if (g_rccl.handle) break;
  }
  if (!g_rccl.handle) { g_load_error = std::string("dlopen(librccl.so): ") + dlerror(); return; }
  auto sym = [&](const char* n) { void* p = dlsym(g_rccl.handle, n); if (!p && g_load_error.empty()) g_load_error = std::string("librccl.so lacks ") + n; return p; };
  g_rccl.GetUniqueId = reinterpret_cast<decltype(g_rccl.GetUniqueId)>(sym("ncclGetUniqueId"));
  g_rccl.CommInitRank = reinterpret_cast<decltype(g_rccl.CommInitRank)>(sym("ncclCommInitRank"));
  g_rccl.Broadcast = reinterpret_cast<decltype(g_rccl.Broadcast)>(sym("ncclBroadcast"));
  g_rccl.CommDestroy = reinterpret_cast<decltype(g_rccl.CommDestroy)>(sym("ncclCommDestroy"));
  g_rccl.GetErrorString = reinterpret_cast<decltype(g_rccl.GetErrorString)>(sym("ncclGetErrorString"));
}

int need_rccl() {
  std::call_once(g_once, load_rccl);
  if (!g_load_error.empty()) return pf::set_error(PF_EHIP, "%s", g_load_error.c_str());
  return PF_OK;
}

int rccl_error(const char* what, int rc) {
  return pf::set_error(PF_EHIP, "%s: %s", what, g_rccl.GetErrorString ? g_rccl.GetErrorString(rc) : "RCCL error");
}

}  // namespace

struct pf_comm { ncclComm_t comm; int rank, nranks; };

extern "C" {

int pf_comm_unique_id(void* out128) {
  PF_REQUIRE(out128, "pf_comm_unique_id: null output");
  if (int rc = need_rccl()) return rc;
  ncclUniqueId id;
  if (int rc = g_rccl.GetUniqueId(&id)) return rccl_error("ncclGetUniqueId", rc);
  memcpy(out128, &id, sizeof id);
  return PF_OK;
}

int pf_comm_init(const void* unique_id128, int rank, int nranks, pf_comm** out) {
  PF_REQUIRE(unique_id128 && out && nranks >= 1 && rank >= 0 && rank < nranks, "pf_comm_init: bad arguments (rank %d of %d)", rank, nranks);
  if (int rc = need_rccl()) return rc;
  ncclUniqueId id;
  memcpy(&id, unique_id128, sizeof id);
  ncclComm_t c = nullptr;
  if (int rc = g_rccl.CommInitRank(&c, nranks, id, rank)) return rccl_error("ncclCommInitRank", rc);
  *out = new pf_comm{c, rank, nranks};
  return PF_OK;
}

int pf_comm_bcast(pf_comm* comm, void* dev_buf, size_t bytes, int root, void* stream) {
  PF_REQUIRE(comm && dev_buf && bytes > 0 && root >= 0 && root < comm->nranks, "pf_comm_bcast: bad arguments");
  if (int rc = g_rccl.Broadcast(dev_buf, dev_buf, bytes, ncclChar, root, comm->comm, (hipStream_t)stream)) return rccl_error("ncclBroadcast", rc);
  return PF_OK;
}

int pf_comm_destroy(pf_comm* comm) {
  if (!comm) return PF_OK;
  int rc = g_rccl.CommDestroy ? g_rccl.CommDestroy(comm->comm) : 0;
  delete comm;
  return rc ? rccl_error("ncclCommDestroy", rc) : PF_OK;
}

}  // extern "C"
