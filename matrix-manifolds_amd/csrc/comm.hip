// The one collective of the sharded training path behind the C ABI: a process-lifetime RCCL communicator
// (one process per GPU, xGMI) and an in-place all-reduce(sum) enqueued on the caller's stream.
//
// Replaces the reference's only parallel call site, torch.nn.DataParallel around BatchedObjective
// (graphembed/graphembed/train.py:107-109: parameter broadcast + scalar gather + gradient reduce-add per step) with
// ONE ncclAllReduce of {point gradients, loss, scale gradients} per step.
//
// RCCL is bound at run time (dlopen), not at link time: libmm_manifolds.so stays loadable by a single-GPU client
// without RCCL on its library path, and inside a PyTorch process the copy of librccl that torch has already mapped
// (same SONAME, librccl.so.1) is reused instead of a second one being loaded.  MM_RCCL_LIB overrides the search.
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <dlfcn.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>

#include "../../include/mm_manifolds.h"

struct mm_comm {
  ncclComm_t nccl;
  int rank, world, device;
};

namespace {

struct Rccl {
  void* handle = nullptr;
  decltype(&ncclGetUniqueId) GetUniqueId = nullptr;
  decltype(&ncclCommInitRank) CommInitRank = nullptr;
  decltype(&ncclCommDestroy) CommDestroy = nullptr;
  decltype(&ncclAllReduce) AllReduce = nullptr;
  decltype(&ncclGetErrorString) GetErrorString = nullptr;
  decltype(&ncclGetVersion) GetVersion = nullptr;
  char path[256] = {0};
};

Rccl g_rccl;
std::once_flag g_once;
thread_local char g_err[512] = "";
char g_load_err[512] = "";   // why librccl could not be bound: written once (under g_once), shown to EVERY thread that asks

void set_err(const char* what, const char* detail) { std::snprintf(g_err, sizeof g_err, "%s: %s", what, detail ? detail : ""); }
void set_load_err(const char* what, const char* detail) {
  std::snprintf(g_load_err, sizeof g_load_err, "%s: %s", what, detail ? detail : "");
}

void load_rccl() {
  const char* env = std::getenv("MM_RCCL_LIB");
  void* h = nullptr;
  const char* used = nullptr;
  if (env && *env) {
    h = dlopen(env, RTLD_NOW | RTLD_LOCAL);
    used = env;
  } else {
    // a copy that is already mapped (PyTorch's) first: RTLD_NOLOAD finds it by SONAME without loading anything
    static const char* const names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1", "/opt/rocm/lib/librccl.so"};
    for (const char* nm : names) {
      h = dlopen(nm, RTLD_NOW | RTLD_LOCAL | RTLD_NOLOAD);
      if (h) { used = nm; break; }
    }
    for (int k = 0; !h && k < 4; ++k) {
      h = dlopen(names[k], RTLD_NOW | RTLD_LOCAL);
      used = names[k];
    }
  }
  if (!h) { set_load_err("librccl not found", dlerror()); return; }
  Rccl r;
  r.handle = h;
  r.GetUniqueId = reinterpret_cast<decltype(r.GetUniqueId)>(dlsym(h, "ncclGetUniqueId"));
  r.CommInitRank = reinterpret_cast<decltype(r.CommInitRank)>(dlsym(h, "ncclCommInitRank"));
  r.CommDestroy = reinterpret_cast<decltype(r.CommDestroy)>(dlsym(h, "ncclCommDestroy"));
  r.AllReduce = reinterpret_cast<decltype(r.AllReduce)>(dlsym(h, "ncclAllReduce"));
  r.GetErrorString = reinterpret_cast<decltype(r.GetErrorString)>(dlsym(h, "ncclGetErrorString"));
  r.GetVersion = reinterpret_cast<decltype(r.GetVersion)>(dlsym(h, "ncclGetVersion"));
  if (!r.GetUniqueId || !r.CommInitRank || !r.CommDestroy || !r.AllReduce) {
    set_load_err("librccl lacks a required symbol", used);
    return;
  }
  std::snprintf(r.path, sizeof r.path, "%s", used ? used : "");
  g_rccl = r;
}

const Rccl* rccl() {
  std::call_once(g_once, load_rccl);
  if (g_rccl.handle) return &g_rccl;
  std::snprintf(g_err, sizeof g_err, "%s", g_load_err);   // (the loader ran on whichever thread called first)
  return nullptr;
}

int fail(const Rccl* r, const char* what, ncclResult_t res) {
  set_err(what, (r && r->GetErrorString) ? r->GetErrorString(res) : "RCCL error");
  return MM_ERR_COMM;
}

}  // namespace

extern "C" {

const char* mm_comm_last_error(void) { return g_err; }

int mm_comm_available(void) { return rccl() ? 1 : 0; }

int mm_comm_rccl_version(void) {
  const Rccl* r = rccl();
  int v = 0;
  if (!r || !r->GetVersion || r->GetVersion(&v) != ncclSuccess) return 0;
  return v;
}

int mm_comm_unique_id(void* id_out) {
  if (!id_out) return MM_ERR_ARG;
  const Rccl* r = rccl();
  if (!r) return MM_ERR_COMM;
  static_assert(sizeof(ncclUniqueId) == MM_COMM_ID_BYTES, "mm_manifolds.h states the size of the rendezvous token");
  ncclUniqueId id;
  const ncclResult_t res = r->GetUniqueId(&id);
  if (res != ncclSuccess) return fail(r, "ncclGetUniqueId", res);
  std::memcpy(id_out, &id, sizeof id);
  return MM_OK;
}

int mm_comm_init(mm_comm_t* comm, int rank, int world, const void* unique_id, int device) {
  if (!comm || !unique_id || world < 1 || rank < 0 || rank >= world || device < 0) return MM_ERR_ARG;
  *comm = nullptr;
  const Rccl* r = rccl();
  if (!r) return MM_ERR_COMM;
  hipError_t e = hipSetDevice(device);   // the communicator is bound to the device that is current when it is created; it STAYS current (documented in mm_manifolds.h)
  if (e != hipSuccess) return int(e);
  ncclUniqueId id;
  std::memcpy(&id, unique_id, sizeof id);
  ncclComm_t c = nullptr;
  const ncclResult_t res = r->CommInitRank(&c, world, id, rank);
  if (res != ncclSuccess) return fail(r, "ncclCommInitRank", res);
  mm_comm* out = new (std::nothrow) mm_comm{c, rank, world, device};
  if (!out) { r->CommDestroy(c); return MM_ERR_ARG; }
  *comm = out;
  return MM_OK;
}

int mm_comm_rank(mm_comm_t comm) { return comm ? comm->rank : -1; }
int mm_comm_world(mm_comm_t comm) { return comm ? comm->world : 0; }

int mm_allreduce_sum(mm_comm_t comm, int dtype, void* buf, int64_t count, mm_stream_t stream) {
  if (!comm || count < 0 || (count > 0 && !buf) || (dtype != MM_F32 && dtype != MM_F64)) return MM_ERR_ARG;
  if (count == 0) return MM_OK;
  const Rccl* r = rccl();
  if (!r) return MM_ERR_COMM;
  const ncclResult_t res = r->AllReduce(buf, buf, size_t(count), dtype == MM_F64 ? ncclFloat64 : ncclFloat32, ncclSum, comm->nccl,
                                        static_cast<hipStream_t>(stream));
  if (res != ncclSuccess) return fail(r, "ncclAllReduce", res);
  return MM_OK;
}

int mm_comm_destroy(mm_comm_t comm) {
  if (!comm) return MM_OK;
  const Rccl* r = rccl();
  ncclResult_t res = ncclSuccess;
  if (r) res = r->CommDestroy(comm->nccl);
  delete comm;
  return res == ncclSuccess ? MM_OK : fail(r, "ncclCommDestroy", res);
}

}  // extern "C"
