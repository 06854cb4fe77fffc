// TEST INFRASTRUCTURE, not product code: a stand-in for librccl.so that implements the five entry points
// ipr-gan_amd/csrc/comm.hip binds at run time (ncclGetUniqueId, ncclCommInitRank, ncclAllReduce, ncclCommDestroy,
// ncclGetErrorString) over a file-backed shared host buffer, so that TWO ranks that share the ONE GPU of a test box can
// execute iprgan_comm_init(nranks = 2), the verified probe, a bucket exchange, the IPRGAN_COMM_TIMEOUT abandon path and the
// collective fallback of iprgan/parallel.py - paths that need two GPUs with the real RCCL.  Selected with
// IPRGAN_RCCL_LIB=<this .so> (comm.hip: rccl_load).  Built by __graft_entry__.build() with hipcc (host code + HIP runtime).
//
// Semantics kept: the rendezvous of ncclCommInitRank blocks until every rank has arrived (no timeout, like the real one);
// ncclAllReduce sums fp32 buffers in rank order (deterministic) and is ENQUEUED on the caller's stream (device-to-host
// copy, a host node that meets the peers and sums, host-to-device copy), so it can wait behind events on a side stream and
// be captured into a HIP graph like the real one.
// STUB_RCCL_FAIL_INIT_RANK=<r>: that rank's ncclCommInitRank fails at once without arriving - its peers stay in the
// rendezvous, which is exactly the situation the abandonable bring-up thread of parallel.RcclTransport.ensure is for.
#include <fcntl.h>
#include <hip/hip_runtime_api.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <time.h>
#include <unistd.h>

#include <atomic>

extern "C" {

typedef struct { char internal[128]; } ncclUniqueId;
typedef struct StubComm* ncclComm_t;
typedef int ncclResult_t;      // 0 = ncclSuccess
typedef int ncclDataType_t;    // 7 = ncclFloat32 (rccl.h)
typedef int ncclRedOp_t;       // 0 = ncclSum

#define STUB_SLOT_FLOATS ((size_t)4 << 20)          // 16 MB per rank and round: larger buffers go in pieces
#define STUB_MAX_RANKS 8

struct StubShared {
  std::atomic<int> arrived;         // ranks inside ncclCommInitRank
  std::atomic<int> barrier_count;
  std::atomic<int> barrier_gen;
  float slot[STUB_MAX_RANKS][STUB_SLOT_FLOATS];
};
struct StubComm {
  StubShared* sh;
  int rank, nranks;
  float* host;                      // pinned staging buffer of this rank
};

static void nap() { struct timespec t = {0, 200000}; nanosleep(&t, nullptr); }

static StubShared* map_shared(const char* token, bool create) {
  char path[256];
  snprintf(path, sizeof(path), "/tmp/iprgan_stub_rccl_%s", token);
  int fd = -1;
  for (int tries = 0; tries < 50000 && fd < 0; ++tries) {           // the creator may not have written the file yet
    fd = open(path, create ? (O_RDWR | O_CREAT) : O_RDWR, 0600);
    if (fd < 0) nap();
  }
  if (fd < 0) return nullptr;
  if (create && ftruncate(fd, sizeof(StubShared)) != 0) { close(fd); return nullptr; }
  struct stat st;
  for (int tries = 0; tries < 50000; ++tries) {                     // ... or sized it
    if (fstat(fd, &st) == 0 && (size_t)st.st_size >= sizeof(StubShared)) break;
    nap();
  }
  void* p = mmap(nullptr, sizeof(StubShared), PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
  close(fd);
  return p == MAP_FAILED ? nullptr : (StubShared*)p;
}

static void barrier(StubComm* c) {
  StubShared* s = c->sh;
  const int gen = s->barrier_gen.load();
  if (s->barrier_count.fetch_add(1) + 1 == c->nranks) {
    s->barrier_count.store(0);
    s->barrier_gen.fetch_add(1);
  } else {
    while (s->barrier_gen.load() == gen) nap();
  }
}

ncclResult_t ncclGetUniqueId(ncclUniqueId* id) {
  memset(id, 0, sizeof(*id));
  snprintf(id->internal, sizeof(id->internal), "%d_%ld", (int)getpid(), (long)time(nullptr));
  return map_shared(id->internal, true) ? 0 : 1;                    // (zero-filled by ftruncate: counters start at 0)
}

ncclResult_t ncclCommInitRank(ncclComm_t* comm, int nranks, ncclUniqueId id, int rank) {
  const char* fail = getenv("STUB_RCCL_FAIL_INIT_RANK");
  if (fail && atoi(fail) == rank) return 2;                         // this rank "cannot enter": its peers keep waiting
  if (nranks > STUB_MAX_RANKS) return 3;
  id.internal[sizeof(id.internal) - 1] = 0;
  StubShared* sh = map_shared(id.internal, false);
  if (!sh) return 4;
  sh->arrived.fetch_add(1);
  while (sh->arrived.load() < nranks) nap();                        // the rendezvous: no timeout, like ncclCommInitRank
  StubComm* c = new StubComm{sh, rank, nranks, nullptr};
  if (hipHostMalloc((void**)&c->host, STUB_SLOT_FLOATS * sizeof(float), hipHostMallocDefault) != hipSuccess) return 5;
  *comm = c;
  return 0;
}

struct StubPiece { StubComm* c; size_t n; };
// host node between the two copies of a piece: publish this rank's piece, meet the peers, sum in rank order
static void stub_exchange(void* arg) {
  StubPiece* p = (StubPiece*)arg;
  StubComm* c = p->c;
  memcpy(c->sh->slot[c->rank], c->host, p->n * sizeof(float));
  barrier(c);                                                       // every rank's piece is in its slot
  for (size_t i = 0; i < p->n; ++i) {
    float s = c->sh->slot[0][i];
    for (int r = 1; r < c->nranks; ++r) s += c->sh->slot[r][i];     // rank order: the same sum on every rank
    c->host[i] = s;
  }
  barrier(c);                                                       // everybody has read the slots: they may be rewritten
}

// Enqueued like the real one (copy out, host node, copy back - all on `stream`, nothing blocks the caller), so the call
// can sit on a side stream behind an event and can be captured into a HIP graph (the argument blocks are never freed: a
// captured graph replays them).
ncclResult_t ncclAllReduce(const void* send, void* recv, size_t count, ncclDataType_t dtype, ncclRedOp_t op, ncclComm_t c,
                           hipStream_t stream) {
  if (dtype != 7 || op != 0) return 6;                              // fp32 SUM only (what comm.hip sends for gradient buckets)
  for (size_t off = 0; off < count; off += STUB_SLOT_FLOATS) {
    const size_t n = count - off < STUB_SLOT_FLOATS ? count - off : STUB_SLOT_FLOATS;
    StubPiece* piece = new StubPiece{c, n};
    if (hipMemcpyAsync(c->host, (const float*)send + off, n * sizeof(float), hipMemcpyDeviceToHost, stream) != hipSuccess) return 7;
    if (hipLaunchHostFunc(stream, stub_exchange, piece) != hipSuccess) return 7;
    if (hipMemcpyAsync((float*)recv + off, c->host, n * sizeof(float), hipMemcpyHostToDevice, stream) != hipSuccess) return 7;
  }
  return 0;
}

ncclResult_t ncclCommDestroy(ncclComm_t c) {
  if (c) {
    if (c->host) (void)hipHostFree(c->host);
    munmap(c->sh, sizeof(StubShared));
    delete c;
  }
  return 0;
}

const char* ncclGetErrorString(ncclResult_t r) {
  switch (r) {
    case 0: return "success";
    case 2: return "stub: this rank was told not to enter the rendezvous (STUB_RCCL_FAIL_INIT_RANK)";
    case 6: return "stub: only fp32 SUM is implemented";
    case 7: return "stub: HIP copy failed";
    default: return "stub: error";
  }
}

}  // extern "C"
