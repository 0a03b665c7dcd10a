// Gradient exchange of the data-parallel train step over RCCL / xGMI (SURVEY sections 5, 8b, 8e): the reference has no
// collective at all (single process), so this is the one communication step the build adds -- a mean all-reduce of the
// flat fp32 gradient vector before Adam (scripts/train_segmentation.py:133-134 run on every rank's shard).
//
// librccl is bound at run time (dlopen), never at link time: an inference process never loads the 0.5 GB library, and a
// process that already holds an RCCL (a PyTorch-ROCm host brings its own copy) shares that one instead of mapping a second.
// The communicator is created from a 128-byte ncclUniqueId the HOST distributes (rank 0 makes it with
// mgu_comm_get_unique_id, every rank passes the same bytes to mgu_comm_init_rank): the launcher's key-value store is all the
// host framework is used for.
#include <dlfcn.h>
#include <link.h>
#include <rccl/rccl.h>   // types and enums only; every function is resolved with dlsym

#include <mutex>

#include "ctx.h"

using namespace mgud;

namespace {

struct RcclApi {
  void* handle = nullptr;
  std::string err;
  ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
  ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
  ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
  ncclResult_t (*CommCount)(const ncclComm_t, int*) = nullptr;
  ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
  const char* (*GetErrorString)(ncclResult_t) = nullptr;
};

int find_loaded_rccl(struct dl_phdr_info* info, size_t, void* data) {
  if (info->dlpi_name && strstr(info->dlpi_name, "librccl.so")) {
    *static_cast<std::string*>(data) = info->dlpi_name;
    return 1;
  }
  return 0;
}

RcclApi& rccl() {
  static RcclApi api;
  static std::once_flag once;
  std::call_once(once, [] {
    std::string loaded;
    dl_iterate_phdr(find_loaded_rccl, &loaded);   // an RCCL the process already mapped (e.g. the host framework's own copy)
    const char* cands[] = {loaded.empty() ? nullptr : loaded.c_str(), getenv("MGU_RCCL_PATH"), "librccl.so.1", "/opt/rocm/lib/librccl.so.1",
                           "librccl.so"};
    for (const char* p : cands) {
      if (!p || !p[0]) continue;
      api.handle = dlopen(p, RTLD_NOW | RTLD_LOCAL);
      if (api.handle) break;
      api.err = dlerror();
    }
    if (!api.handle) return;
    auto sym = [&](const char* n) {
      void* f = dlsym(api.handle, n);
      if (!f) api.err = std::string("librccl lacks ") + n;
      return f;
    };
    api.GetUniqueId = (decltype(api.GetUniqueId))sym("ncclGetUniqueId");
    api.CommInitRank = (decltype(api.CommInitRank))sym("ncclCommInitRank");
    api.CommDestroy = (decltype(api.CommDestroy))sym("ncclCommDestroy");
    api.CommCount = (decltype(api.CommCount))sym("ncclCommCount");
    api.AllReduce = (decltype(api.AllReduce))sym("ncclAllReduce");
    api.GetErrorString = (decltype(api.GetErrorString))sym("ncclGetErrorString");
    if (!api.GetUniqueId || !api.CommInitRank || !api.CommDestroy || !api.CommCount || !api.AllReduce || !api.GetErrorString) {
      dlclose(api.handle);
      api.handle = nullptr;
    }
  });
  return api;
}

int need_rccl(mgu_ctx* c) {
  RcclApi& r = rccl();
  if (!r.handle) return fail(c, MGU_ERR_STATE, "librccl could not be loaded (%s): the gradient exchange has no other transport", r.err.c_str());
  return MGU_OK;
}

#define NCCLCHK(c, call)                                                                                         \
  do {                                                                                                           \
    ncclResult_t r_ = (call);                                                                                    \
    if (r_ != ncclSuccess) return mgud::fail(c, MGU_ERR_HIP, "%s failed: %s", #call, rccl().GetErrorString(r_)); \
  } while (0)

}  // namespace

// mean all-reduce of flat[lo, hi) on `comm_stream`, ordered after everything enqueued on `s` so far
int mgud::comm_bucket(mgu_ctx* c, float* flat, int64_t lo, int64_t hi, hipStream_t s) {
  if (hi <= lo) return MGU_OK;
  hipEvent_t ev = c->comm_ev[c->comm_ev_next++ % (int)(sizeof c->comm_ev / sizeof c->comm_ev[0])];
  HIPCHK(c, hipEventRecord(ev, s));
  HIPCHK(c, hipStreamWaitEvent(c->comm_stream, ev, 0));
  NCCLCHK(c, rccl().AllReduce(flat + lo, flat + lo, (size_t)(hi - lo), ncclFloat32, ncclAvg, (ncclComm_t)c->comm, c->comm_stream));
  return MGU_OK;
}

// make `s` wait for every bucket issued so far
int mgud::comm_join(mgu_ctx* c, hipStream_t s) {
  hipEvent_t ev = c->comm_ev[c->comm_ev_next++ % (int)(sizeof c->comm_ev / sizeof c->comm_ev[0])];
  HIPCHK(c, hipEventRecord(ev, c->comm_stream));
  HIPCHK(c, hipStreamWaitEvent(s, ev, 0));
  return MGU_OK;
}

extern "C" {

int mgu_comm_get_unique_id(void* id_out) {
  if (!id_out) return fail(nullptr, MGU_ERR_INVALID, "id_out == NULL");
  int rc = need_rccl(nullptr);
  if (rc) return rc;
  static_assert(sizeof(ncclUniqueId) == MGU_COMM_ID_BYTES, "ncclUniqueId size");
  ncclUniqueId id;
  NCCLCHK(nullptr, rccl().GetUniqueId(&id));
  memcpy(id_out, &id, sizeof id);
  return MGU_OK;
}

int mgu_comm_init_rank(mgu_ctx* c, const void* id, int rank, int world_size) {
  if (!c) return MGU_ERR_INVALID;
  if (!id || world_size < 1 || rank < 0 || rank >= world_size) return fail(c, MGU_ERR_INVALID, "bad comm args (rank %d of %d)", rank, world_size);
  if (c->comm) return fail(c, MGU_ERR_STATE, "this context already owns a communicator (mgu_comm_destroy first)");
  int rc = need_rccl(c);
  if (rc) return rc;
  HIPCHK(c, hipSetDevice(c->device));
  ncclUniqueId uid;
  memcpy(&uid, id, sizeof uid);
  ncclComm_t comm = nullptr;
  NCCLCHK(c, rccl().CommInitRank(&comm, world_size, uid, rank));
  c->comm = comm;
  c->comm_world = world_size, c->comm_rank = rank;
  HIPCHK(c, hipStreamCreateWithFlags(&c->comm_stream, hipStreamNonBlocking));
  for (auto& e : c->comm_ev) HIPCHK(c, hipEventCreateWithFlags(&e, hipEventDisableTiming));
  return MGU_OK;
}

int mgu_comm_destroy(mgu_ctx* c) {
  if (!c) return MGU_ERR_INVALID;
  if (!c->comm) return MGU_OK;
  (void)hipSetDevice(c->device);
  (void)hipDeviceSynchronize();
  ncclResult_t r = rccl().CommDestroy((ncclComm_t)c->comm);
  c->comm = nullptr;
  c->comm_world = 1, c->comm_rank = 0;
  if (c->comm_stream) (void)hipStreamDestroy(c->comm_stream);
  c->comm_stream = nullptr;
  for (auto& e : c->comm_ev) {
    if (e) (void)hipEventDestroy(e);
    e = nullptr;
  }
  if (r != ncclSuccess) return fail(c, MGU_ERR_HIP, "ncclCommDestroy failed: %s", rccl().GetErrorString(r));
  return MGU_OK;
}

void* mgu_comm_handle(mgu_ctx* c) { return c ? c->comm : nullptr; }

int mgu_comm_world_size(mgu_ctx* c) { return c && c->comm ? c->comm_world : 1; }

int mgu_allreduce_grads(mgu_ctx* c, void* flat_grad_dev, int64_t n, void* rccl_comm, void* hip_stream) {
  if (!c) return MGU_ERR_INVALID;
  if (n < 0 || (n > 0 && !flat_grad_dev)) return fail(c, MGU_ERR_INVALID, "bad allreduce args");
  void* comm = rccl_comm ? rccl_comm : c->comm;
  if (!comm) return fail(c, MGU_ERR_STATE, "no communicator: call mgu_comm_init_rank or pass an ncclComm_t");
  int rc = need_rccl(c);
  if (rc) return rc;
  if (n == 0) return MGU_OK;
  HIPCHK(c, hipSetDevice(c->device));
  NCCLCHK(c, rccl().AllReduce(flat_grad_dev, flat_grad_dev, (size_t)n, ncclFloat32, ncclAvg, (ncclComm_t)comm, (hipStream_t)hip_stream));
  return MGU_OK;
}

}  // extern "C"
