// Data-parallel gradient exchange behind the C ABI: RCCL all-reduce over xGMI.
//
// Replaces the reference's per-forward torch.nn.DataParallel traffic (replicate + scatter + gather at every
// G/D call: models/dcgan.py:16-17, models/srgan.py:17-19, models/cyclegan.py:19-23) by ONE exchange per optimizer
// step and network: each rank (= one process = one GPU) owns flat fp32 gradient buckets and sums them in place.
//
// RCCL is bound at run time (dlopen/dlsym), not at link time: a PyTorch-ROCm host process already carries its own
// librccl.so, and a second copy of the library in the same process (two sets of globals, two bootstrap threads)
// is what a link-time dependency on /opt/rocm/lib/librccl.so.1 would create.  The copy that is already loaded is
// used; a host without one gets ROCm's.
// The communicator handle is the library's only process-global state besides the conv autotune cache.
#include <dlfcn.h>
#include <rccl/rccl.h>

#include "common.h"

namespace iprgan {

struct RcclApi {
  ncclResult_t (*GetUniqueId)(ncclUniqueId*);
  ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int);
  ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t);
  ncclResult_t (*CommDestroy)(ncclComm_t);
  const char* (*GetErrorString)(ncclResult_t);
  void* handle;
};
static RcclApi g_rccl = {};
static ncclComm_t g_comm = nullptr;
static int g_comm_rank = 0, g_comm_nranks = 0;

static int rccl_load() {
  if (g_rccl.handle) return 0;
  const char* override_path = getenv("IPRGAN_RCCL_LIB");
  void* h = nullptr;
  if (override_path) h = dlopen(override_path, RTLD_NOW | RTLD_GLOBAL);
  if (!h) h = dlopen("librccl.so", RTLD_NOW | RTLD_NOLOAD);        // the host's own copy (PyTorch-ROCm ships one)
  if (!h) h = dlopen("librccl.so.1", RTLD_NOW | RTLD_NOLOAD);
  if (!h) h = dlopen("librccl.so.1", RTLD_NOW | RTLD_GLOBAL);      // ROCm's
  if (!h) h = dlopen("librccl.so", RTLD_NOW | RTLD_GLOBAL);
  IPR_CHECK(h, "comm: cannot load RCCL (librccl.so): %s", dlerror());
#define IPR_SYM(field, name)                                         \
  *(void**)(&g_rccl.field) = dlsym(h, name);                         \
  IPR_CHECK(g_rccl.field, "comm: RCCL lacks %s", name)
  IPR_SYM(GetUniqueId, "ncclGetUniqueId");
  IPR_SYM(CommInitRank, "ncclCommInitRank");
  IPR_SYM(AllReduce, "ncclAllReduce");
  IPR_SYM(CommDestroy, "ncclCommDestroy");
  IPR_SYM(GetErrorString, "ncclGetErrorString");
#undef IPR_SYM
  g_rccl.handle = h;
  return 0;
}

#define IPR_RCCL(call, what)                                                          \
  do {                                                                                \
    const ncclResult_t r__ = (call);                                                  \
    IPR_CHECK(r__ == ncclSuccess, "comm: %s failed: %s", what, g_rccl.GetErrorString(r__)); \
  } while (0)

}  // namespace iprgan

using namespace iprgan;

extern "C" {

int iprgan_comm_unique_id(void* id128) {
  IPR_CHECK(id128, "comm_unique_id: null buffer");
  static_assert(sizeof(ncclUniqueId) == IPRGAN_COMM_ID_BYTES, "ncclUniqueId size");
  if (rccl_load()) return 1;
  ncclUniqueId id;
  IPR_RCCL(g_rccl.GetUniqueId(&id), "ncclGetUniqueId");
  memcpy(id128, &id, sizeof(id));
  return 0;
}

int iprgan_comm_init(int rank, int nranks, const void* id128) {
  IPR_CHECK(!g_comm, "comm_init: a communicator already exists (call iprgan_comm_destroy first)");
  IPR_CHECK(nranks >= 1 && rank >= 0 && rank < nranks && id128, "comm_init: bad rank %d / %d", rank, nranks);
  if (rccl_load()) return 1;
  ncclUniqueId id;
  memcpy(&id, id128, sizeof(id));
  IPR_RCCL(g_rccl.CommInitRank(&g_comm, nranks, id, rank), "ncclCommInitRank");     // binds the CURRENT HIP device
  g_comm_rank = rank;
  g_comm_nranks = nranks;
  return 0;
}

int iprgan_allreduce_bucket(void* buf, size_t n, int dtype, void* stream) {
  IPR_CHECK(g_comm, "allreduce_bucket: no communicator (iprgan_comm_init)");
  IPR_CHECK(dtype == IPRGAN_DTYPE_F32 || dtype == IPRGAN_DTYPE_BF16, "allreduce_bucket: bad dtype %d", dtype);
  if (!n) return 0;
  IPR_RCCL(g_rccl.AllReduce(buf, buf, n, dtype == IPRGAN_DTYPE_F32 ? ncclFloat32 : ncclBfloat16, ncclSum, g_comm,
                            (hipStream_t)stream), "ncclAllReduce");
  return 0;
}

int iprgan_comm_nranks(void) { return g_comm ? g_comm_nranks : 0; }
int iprgan_comm_rank(void) { return g_comm ? g_comm_rank : -1; }

int iprgan_comm_destroy(void) {
  if (!g_comm) return 0;
  const ncclResult_t r = g_rccl.CommDestroy(g_comm);
  g_comm = nullptr;
  g_comm_nranks = 0;
  IPR_CHECK(r == ncclSuccess, "comm_destroy: %s", g_rccl.GetErrorString(r));
  return 0;
}

}  // extern "C"
