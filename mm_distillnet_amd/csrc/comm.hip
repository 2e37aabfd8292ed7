// RCCL entry points of the C ABI (mmd_comm_{init,allreduce_bucket,destroy}) — the data-parallel exchange of the step:
// all-reduce(sum) of contiguous ranges of the flat student gradient buffer and all-reduce(max) of the head_active flag, over xGMI.
// Replaces what DistributedDataParallel's reducer does for the reference (src/optimization/train_methods.py:944-961: gradients
// averaged across ranks; here the 1/N is folded into the optimizer pass, so the collective is a plain sum).
// librccl is opened lazily with dlopen, so the library loads (and every single-GPU path runs) on hosts without it.
#include "common.h"
#include <dlfcn.h>
#include <cstring>

namespace {
typedef struct { char internal[128]; } UniqueId;          // ncclUniqueId (NCCL_UNIQUE_ID_BYTES = 128)
typedef void* Comm;
typedef int (*fn_get_id)(UniqueId*);
typedef int (*fn_init)(Comm*, int, UniqueId, int);
typedef int (*fn_allreduce)(const void*, void*, size_t, int, int, Comm, hipStream_t);
typedef int (*fn_destroy)(Comm);
typedef int (*fn_count)(Comm, int*);
struct Api { void* lib; fn_get_id get_id; fn_init init; fn_allreduce allreduce; fn_destroy destroy; fn_count count; };
Api* api() {
  static Api a{};
  static bool tried = false;
  if (!tried) {
    tried = true;
    a.lib = dlopen("librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
    if (!a.lib) a.lib = dlopen("librccl.so", RTLD_NOW | RTLD_GLOBAL);
    if (a.lib) {
      a.get_id = (fn_get_id)dlsym(a.lib, "ncclGetUniqueId");
      a.init = (fn_init)dlsym(a.lib, "ncclCommInitRank");
      a.allreduce = (fn_allreduce)dlsym(a.lib, "ncclAllReduce");
      a.destroy = (fn_destroy)dlsym(a.lib, "ncclCommDestroy");
      a.count = (fn_count)dlsym(a.lib, "ncclCommCount");
      if (!a.get_id || !a.init || !a.allreduce || !a.destroy) a.lib = nullptr;
    }
  }
  return a.lib ? &a : nullptr;
}
}  // namespace

#define MMD_ENOLIB -38

// 128-byte rendezvous token: created on rank 0, handed to every rank by the host (a file, a TCP store, torch.distributed ...).
extern "C" int mmd_comm_unique_id(void* out128) {
  if (!out128) return MMD_EINVAL;
  Api* a = api();
  if (!a) return MMD_ENOLIB;
  UniqueId id;
  if (a->get_id(&id) != 0) return MMD_ELAUNCH;
  memcpy(out128, &id, sizeof(id));
  return MMD_OK;
}

// One communicator per process / GPU (the caller has made `device` current).  *comm_out is an opaque handle.
extern "C" int mmd_comm_init(void** comm_out, int rank, int world, const void* unique_id128) {
  if (!comm_out || !unique_id128 || world < 1 || rank < 0 || rank >= world) return MMD_EINVAL;
  Api* a = api();
  if (!a) return MMD_ENOLIB;
  UniqueId id;
  memcpy(&id, unique_id128, sizeof(id));
  Comm c = nullptr;
  if (a->init(&c, world, id, rank) != 0) return MMD_ELAUNCH;
  *comm_out = c;
  return MMD_OK;
}

// In-place all-reduce of one bucket on `stream` (asynchronous; ordered like any other work on that stream).
//   dtype 0 = float32 (gradient ranges), 1 = int32 (head_active);  op 0 = sum, 1 = max.
extern "C" int mmd_comm_allreduce_bucket(void* comm, void* buf, long long count, int dtype, int op, hipStream_t stream) {
  if (!comm || !buf || count <= 0 || dtype < 0 || dtype > 1 || op < 0 || op > 1) return MMD_EINVAL;
  Api* a = api();
  if (!a) return MMD_ENOLIB;
  const int nccl_dtype = dtype == 0 ? 7 /* ncclFloat32 */ : 2 /* ncclInt32 */;
  const int nccl_op = op == 0 ? 0 /* ncclSum */ : 2 /* ncclMax */;
  return a->allreduce(buf, buf, (size_t)count, nccl_dtype, nccl_op, comm, stream) == 0 ? MMD_OK : MMD_ELAUNCH;
}

// Number of ranks RCCL itself reports for the communicator (ncclCommCount): what a scaling record cites as "RCCL saw N ranks".
extern "C" int mmd_comm_count(void* comm, void* count_out) {
  if (!comm || !count_out) return MMD_EINVAL;
  Api* a = api();
  if (!a || !a->count) return MMD_ENOLIB;
  int n = 0;
  if (a->count(comm, &n) != 0) return MMD_ELAUNCH;
  *(int*)count_out = n;
  return MMD_OK;
}

extern "C" int mmd_comm_destroy(void* comm) {
  if (!comm) return MMD_EINVAL;
  Api* a = api();
  if (!a) return MMD_ENOLIB;
  return a->destroy(comm) == 0 ? MMD_OK : MMD_ELAUNCH;
}
