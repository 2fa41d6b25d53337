// k3_comm.hip -- multi-GPU entry points of the path: the decoding graph is read and converted ONCE, on one rank, and broadcast over
// RCCL (xGMI inside a node) to the other ranks' HBM; after that the ranks share nothing (utterances are independent).  RCCL is bound at run time
// (dlopen librccl.so.1) so that single-GPU users of libk3hip.so do not load it.
#include "k3_common.h"
#include <dlfcn.h>
#include <unistd.h>
#include <cstring>
#include <string>
#include <vector>

extern "C" int k3_fst_shape_and_image(const k3_fst *fst, int64_t *shape /* [5]: states, arcs, start, bytes, max_pdf */, void **d_image);
extern "C" int k3_fst_create_shaped(const int64_t *shape, k3_fst **out);

namespace {
typedef struct { char internal[128]; } UniqueId;      // ncclUniqueId (rccl.h:43, NCCL_UNIQUE_ID_BYTES = 128)
struct Rccl {
  void *h = nullptr;
  int (*GetUniqueId)(UniqueId *) = nullptr; int (*CommInitRank)(void **, int, UniqueId, int) = nullptr; int (*CommDestroy)(void *) = nullptr;
  int (*Broadcast)(const void *, void *, size_t, int, int, void *, hipStream_t) = nullptr; const char *(*GetErrorString)(int) = nullptr;
};
int rccl(Rccl **out) {
  static Rccl r; static bool tried = false;
  if (!tried) {
    tried = true;
    for (const char *name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) { r.h = dlopen(name, RTLD_NOW | RTLD_GLOBAL); if (r.h) break; }
    if (r.h) {
      r.GetUniqueId = (int (*)(UniqueId *))dlsym(r.h, "ncclGetUniqueId"); r.CommInitRank = (int (*)(void **, int, UniqueId, int))dlsym(r.h, "ncclCommInitRank");
      r.CommDestroy = (int (*)(void *))dlsym(r.h, "ncclCommDestroy"); r.Broadcast = (int (*)(const void *, void *, size_t, int, int, void *, hipStream_t))dlsym(r.h, "ncclBroadcast");
      r.GetErrorString = (const char *(*)(int))dlsym(r.h, "ncclGetErrorString");
    }
  }
  if (!r.h || !r.GetUniqueId || !r.CommInitRank || !r.Broadcast) { k3::set_error("RCCL (librccl.so.1) is not available: %s", dlerror() ? dlerror() : "symbols missing"); return K3_ERR_UNSUPPORTED; }
  *out = &r; return K3_OK;
}
#define K3_RCCL(R, e) do { const int rc__ = (e); if (rc__ != 0) { k3::set_error("RCCL error %d (%s) in %s", rc__, (R)->GetErrorString ? (R)->GetErrorString(rc__) : "?", #e); return K3_ERR_HIP; } } while (0)
constexpr int kNcclUint8 = 1, kNcclInt64 = 4;      // ncclDataType_t (rccl.h:459-467)
}  // namespace

// One communicator per process (one process per GPU, hipSetDevice done by the caller).  The 128-byte ncclUniqueId travels through a file on a
// file system every rank sees (what a Kaldi recipe has anyway: exp/.../decode/): rank 0 writes <id_file>, the others wait for it.
extern "C" int k3_comm_create(const char *id_file, int32_t rank, int32_t world_size, int32_t timeout_seconds, void **comm) {
  K3_REQUIRE(id_file && comm && world_size >= 1 && rank >= 0 && rank < world_size, "k3_comm_create: bad argument");
  Rccl *R; { const int rc = rccl(&R); if (rc) return rc; }
  UniqueId id; memset(&id, 0, sizeof id);
  if (rank == 0) {
    K3_RCCL(R, R->GetUniqueId(&id));
    const std::string tmp = std::string(id_file) + ".tmp";
    FILE *f = fopen(tmp.c_str(), "wb"); K3_REQUIRE(f && fwrite(&id, sizeof id, 1, f) == 1 && fclose(f) == 0 && rename(tmp.c_str(), id_file) == 0, "k3_comm_create: cannot write the id file");
  } else {
    bool ok = false;
    for (int i = 0; i < 20 * std::max(1, timeout_seconds) && !ok; i++) {
      FILE *f = fopen(id_file, "rb");
      if (f) { ok = fread(&id, sizeof id, 1, f) == 1; fclose(f); }
      if (!ok) usleep(50000);
    }
    K3_REQUIRE(ok, "k3_comm_create: timed out waiting for rank 0's id file");
  }
  void *c = nullptr;
  K3_RCCL(R, R->CommInitRank(&c, world_size, id, rank));
  *comm = c; return K3_OK;
}
extern "C" void k3_comm_destroy(void *comm) { Rccl *R; if (comm && rccl(&R) == K3_OK && R->CommDestroy) (void)R->CommDestroy(comm); }

// k3_fst_csr_bcast(handle, ncclComm_t).  On `root` *fst is the graph (k3_fst_create); on the other ranks *fst is NULL on entry and
// owns a graph of the same shape holding root's image on return.  `comm` is an ncclComm_t (k3_comm_create, or the application's own).
extern "C" int k3_fst_bcast(k3_fst **fst, void *comm, int32_t root, int32_t rank, void *stream) {
  K3_REQUIRE(fst && comm && (rank != root || *fst), "k3_fst_bcast: bad argument (the root rank must pass its graph)");
  Rccl *R; { const int rc = rccl(&R); if (rc) return rc; }
  hipStream_t st = (hipStream_t)stream;
  int64_t shape[5] = {0, 0, 0, 0, -1}; void *image = nullptr;
  if (rank == root) { const int rc = k3_fst_shape_and_image(*fst, shape, &image); if (rc) return rc; }
  int64_t *d_shape = nullptr; K3_HIP_CHECK(hipMalloc((void **)&d_shape, sizeof shape));
  K3_HIP_CHECK(hipMemcpy(d_shape, shape, sizeof shape, hipMemcpyHostToDevice));
  K3_RCCL(R, R->Broadcast(d_shape, d_shape, 5, kNcclInt64, root, comm, st));
  K3_HIP_CHECK(hipStreamSynchronize(st));
  K3_HIP_CHECK(hipMemcpy(shape, d_shape, sizeof shape, hipMemcpyDeviceToHost)); (void)hipFree(d_shape);
  if (rank != root) { const int rc = k3_fst_create_shaped(shape, fst); if (rc) return rc; int64_t s2[5]; const int rc2 = k3_fst_shape_and_image(*fst, s2, &image); if (rc2) return rc2; }
  K3_RCCL(R, R->Broadcast(image, image, (size_t)shape[3], kNcclUint8, root, comm, st));
  K3_HIP_CHECK(hipStreamSynchronize(st));
  return K3_OK;
}
