// Data-parallel gradient exchange behind the C ABI: an RCCL communicator owned by this library (one rank per process / GPU) and an
// all-reduce over a contiguous range of a flat gradient buffer on a caller-supplied HIP stream.
//
// Replaces torch.nn.parallel.DistributedDataParallel of trainer/train_representation_learning.py:29,39 (its bucketed NCCL all-reduce of the
// trainable gradients) for hosts that do not go through torch.distributed: the caller exchanges the 128-byte unique id over whatever
// control channel it has (pdae_amd/comm.py uses the torch.distributed store), every rank calls pdae_comm_init, and the training step
// enqueues pdae_allreduce_bucket on a side stream as soon as a bucket of gradients is final (event-ordered against the compute stream).
// RCCL is bound at run time (dlopen): libpdae_hip.so itself has no link-time dependency on it, so single-GPU hosts need no RCCL at all and
// a process that already carries PyTorch's copy binds to that one instead of loading a second RCCL.
#include <dlfcn.h>
#include <string.h>

#include "common.h"
#include "kernels.h"

typedef struct { char internal[128]; } rccl_unique_id;           // ncclUniqueId (rccl.h: NCCL_UNIQUE_ID_BYTES 128)
typedef void* rccl_comm;                                          // ncclComm_t
typedef int (*fn_get_id)(rccl_unique_id*);
typedef int (*fn_init_rank)(rccl_comm*, int, rccl_unique_id, int);
typedef int (*fn_allreduce)(const void*, void*, size_t, int, int, rccl_comm, hipStream_t);
typedef int (*fn_destroy)(rccl_comm);
typedef const char* (*fn_errstr)(int);

static struct { void* h; fn_get_id get_id; fn_init_rank init_rank; fn_allreduce allreduce; fn_destroy destroy; fn_errstr errstr; } R;

static int rccl_bind(const char* path) {
  if (R.h) return PDAE_OK;
  const char* cands[] = {path, "librccl.so", "librccl.so.1", "/opt/rocm/lib/librccl.so.1"};
  void* h = nullptr;
  for (int pass = 0; pass < 2 && !h; ++pass)                        // pass 0: only a copy that is already mapped into the process
    for (const char* c : cands) {
      if (!c || !*c) continue;
      h = dlopen(c, RTLD_NOW | RTLD_GLOBAL | (pass == 0 ? RTLD_NOLOAD : 0));
      if (h) break;
    }
  if (!h) { pdae_set_error("comm: cannot load RCCL (%s)", dlerror()); return PDAE_EINVAL; }
  R.get_id = (fn_get_id)dlsym(h, "ncclGetUniqueId"); R.init_rank = (fn_init_rank)dlsym(h, "ncclCommInitRank");
  R.allreduce = (fn_allreduce)dlsym(h, "ncclAllReduce"); R.destroy = (fn_destroy)dlsym(h, "ncclCommDestroy");
  R.errstr = (fn_errstr)dlsym(h, "ncclGetErrorString");
  if (!R.get_id || !R.init_rank || !R.allreduce || !R.destroy) { pdae_set_error("comm: RCCL symbols missing"); return PDAE_EINVAL; }
  R.h = h;
  return PDAE_OK;
}

static int rccl_status(int rc, const char* what) {
  if (rc == 0) return PDAE_OK;
  pdae_set_error("%s: RCCL error %d (%s)", what, rc, R.errstr ? R.errstr(rc) : "?");
  return 1000 + rc;
}

int k_comm_unique_id(const char* librccl_path, void* id128) {
  if (int e = rccl_bind(librccl_path)) return e;
  rccl_unique_id id;
  if (int e = rccl_status(R.get_id(&id), "ncclGetUniqueId")) return e;
  memcpy(id128, &id, sizeof(id));
  return PDAE_OK;
}

int k_comm_init(const char* librccl_path, const void* id128, int nranks, int rank, void** comm) {
  if (int e = rccl_bind(librccl_path)) return e;
  rccl_unique_id id;
  memcpy(&id, id128, sizeof(id));
  rccl_comm c = nullptr;
  if (int e = rccl_status(R.init_rank(&c, nranks, id, rank), "ncclCommInitRank")) return e;
  *comm = c;
  return PDAE_OK;
}

int k_allreduce(void* comm, void* buf, size_t count, int dtype, int op, hipStream_t st) {
  if (!R.h) { pdae_set_error("allreduce_bucket: no communicator (pdae_comm_init first)"); return PDAE_EINVAL; }
  // rccl.h: ncclInt32 = 2, ncclFloat32 = 7; ncclSum = 0, ncclMax = 2
  return rccl_status(R.allreduce(buf, buf, count, dtype == 1 ? 2 : 7, op == 1 ? 2 : 0, (rccl_comm)comm, st), "ncclAllReduce");
}

int k_comm_destroy(void* comm) {
  if (!R.h || !comm) return PDAE_OK;
  return rccl_status(R.destroy((rccl_comm)comm), "ncclCommDestroy");
}
