// RCCL communicator behind the C-ABI: the ONE exchange of the data-parallel path is the gradient sum
// (SURVEY.md §8b `mgd_comm_{init,allreduce_bucket,destroy}`; the reference itself has no multi-GPU path).
// librccl is opened lazily (dlopen) so that the library loads - and the single-GPU path runs - on hosts
// where RCCL is absent; every entry fails loudly (MGD_ELAUNCH + message) if it cannot be opened.
#include <dlfcn.h>
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <mutex>

#include "../../include/mgd_hip.h"

int mgd_set_error(int code, const char* fmt, ...);

namespace {

struct Rccl {
  void* handle = nullptr;
  decltype(&ncclGetUniqueId) get_unique_id = nullptr;
  decltype(&ncclCommInitRank) comm_init_rank = nullptr;
  decltype(&ncclAllReduce) all_reduce = nullptr;
  decltype(&ncclCommDestroy) comm_destroy = nullptr;
  decltype(&ncclGetErrorString) error_string = nullptr;
  bool ok = false;
};

Rccl& rccl() {
  static Rccl r;
  static std::once_flag once;
  std::call_once(once, [] {
    for (const char* name : {"librccl.so.1", "librccl.so"}) {
      r.handle = dlopen(name, RTLD_NOW | RTLD_LOCAL);
      if (r.handle) break;
    }
    if (!r.handle) return;
    r.get_unique_id = (decltype(r.get_unique_id))dlsym(r.handle, "ncclGetUniqueId");
    r.comm_init_rank = (decltype(r.comm_init_rank))dlsym(r.handle, "ncclCommInitRank");
    r.all_reduce = (decltype(r.all_reduce))dlsym(r.handle, "ncclAllReduce");
    r.comm_destroy = (decltype(r.comm_destroy))dlsym(r.handle, "ncclCommDestroy");
    r.error_string = (decltype(r.error_string))dlsym(r.handle, "ncclGetErrorString");
    r.ok = r.get_unique_id && r.comm_init_rank && r.all_reduce && r.comm_destroy && r.error_string;
  });
  return r;
}

int fail(const char* what, ncclResult_t rc) {
  return mgd_set_error(MGD_ELAUNCH, "%s: %s", what, rccl().error_string ? rccl().error_string(rc) : "rccl error");
}

}  // namespace

#define MGD_RCCL_OR_FAIL(what)                                                                   \
  do {                                                                                           \
    if (!rccl().ok) return mgd_set_error(MGD_ELAUNCH, "%s: librccl.so could not be opened", what); \
  } while (0)

extern "C" int mgd_comm_unique_id(void* id128) {
  if (!id128) return mgd_set_error(MGD_EINVAL, "comm_unique_id: null pointer");
  MGD_RCCL_OR_FAIL("comm_unique_id");
  static_assert(sizeof(ncclUniqueId) == MGD_COMM_ID_BYTES, "ncclUniqueId size");
  ncclResult_t rc = rccl().get_unique_id((ncclUniqueId*)id128);
  return rc == ncclSuccess ? MGD_OK : fail("comm_unique_id", rc);
}

extern "C" int mgd_comm_init(void** comm, int rank, int world, const void* id128) {
  if (!comm || !id128 || world < 1 || rank < 0 || rank >= world)
    return mgd_set_error(MGD_EINVAL, "comm_init: rank=%d world=%d", rank, world);
  MGD_RCCL_OR_FAIL("comm_init");
  ncclUniqueId id;
  __builtin_memcpy(&id, id128, sizeof(id));
  ncclComm_t c = nullptr;
  ncclResult_t rc = rccl().comm_init_rank(&c, world, id, rank);      // on the calling thread's current device
  if (rc != ncclSuccess) return fail("comm_init", rc);
  *comm = (void*)c;
  return MGD_OK;
}

extern "C" int mgd_comm_allreduce_bucket(void* comm, float* grads, int64_t count, void* stream) {
  if (!comm || !grads || count < 0) return mgd_set_error(MGD_EINVAL, "comm_allreduce_bucket: bad arguments");
  MGD_RCCL_OR_FAIL("comm_allreduce_bucket");
  if (count == 0) return MGD_OK;
  ncclResult_t rc = rccl().all_reduce(grads, grads, (size_t)count, ncclFloat, ncclSum, (ncclComm_t)comm, (hipStream_t)stream);
  return rc == ncclSuccess ? MGD_OK : fail("comm_allreduce_bucket", rc);
}

extern "C" int mgd_comm_destroy(void* comm) {
  if (!comm) return MGD_OK;
  MGD_RCCL_OR_FAIL("comm_destroy");
  ncclResult_t rc = rccl().comm_destroy((ncclComm_t)comm);
  return rc == ncclSuccess ? MGD_OK : fail("comm_destroy", rc);
}
