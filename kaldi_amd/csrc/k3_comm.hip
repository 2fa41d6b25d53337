// k3_comm.hip -- multi-GPU entry points of the path: the decoding graph is read and converted ONCE, on one rank, and broadcast over
// RCCL (xGMI inside a node) to the other ranks' HBM; after that the ranks share nothing (utterances are independent).  RCCL is bound at run time
// (dlopen librccl.so.1) so that single-GPU users of libk3hip.so do not load it.
#include "k3_common.h"
#include <dlfcn.h>
#include <sys/stat.h>
#include <unistd.h>
#include <cstdlib>
#include <cstring>
#include <ctime>
#include <mutex>
#include <string>
#include <vector>

extern "C" int k3_fst_shape_and_image(const k3_fst *fst, int64_t *shape /* [5]: states, arcs, start, bytes, max_pdf */, void **d_image);
extern "C" int k3_fst_create_shaped(const int64_t *shape, k3_fst **out);

namespace {
typedef struct { char internal[128]; } UniqueId;      // ncclUniqueId (rccl.h:43, NCCL_UNIQUE_ID_BYTES = 128)
struct Rccl {
  void *h = nullptr; std::string why;
  int (*GetUniqueId)(UniqueId *) = nullptr; int (*CommInitRank)(void **, int, UniqueId, int) = nullptr; int (*CommDestroy)(void *) = nullptr;
  int (*Broadcast)(const void *, void *, size_t, int, int, void *, hipStream_t) = nullptr; const char *(*GetErrorString)(int) = nullptr;
  int (*AllReduce)(const void *, void *, size_t, int, int, void *, hipStream_t) = nullptr;
};
int rccl(Rccl **out) {
  static Rccl r; static std::once_flag once;
  std::call_once(once, [] {
    for (const char *name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) { r.h = dlopen(name, RTLD_NOW | RTLD_GLOBAL); if (r.h) break; }
    if (!r.h) { const char *e = dlerror(); r.why = e ? e : "dlopen failed"; return; }      // (dlerror() clears the error it returns: read it once)
    r.GetUniqueId = (int (*)(UniqueId *))dlsym(r.h, "ncclGetUniqueId"); r.CommInitRank = (int (*)(void **, int, UniqueId, int))dlsym(r.h, "ncclCommInitRank");
    r.CommDestroy = (int (*)(void *))dlsym(r.h, "ncclCommDestroy"); r.Broadcast = (int (*)(const void *, void *, size_t, int, int, void *, hipStream_t))dlsym(r.h, "ncclBroadcast");
    r.AllReduce = (int (*)(const void *, void *, size_t, int, int, void *, hipStream_t))dlsym(r.h, "ncclAllReduce");
    r.GetErrorString = (const char *(*)(int))dlsym(r.h, "ncclGetErrorString");
    if (!r.GetUniqueId || !r.CommInitRank || !r.Broadcast || !r.AllReduce) r.why = "symbols missing";
  });
  if (!r.h || !r.GetUniqueId || !r.CommInitRank || !r.Broadcast || !r.AllReduce) { k3::set_error("RCCL (librccl.so.1) is not available: %s", r.why.c_str()); return K3_ERR_UNSUPPORTED; }
  *out = &r; return K3_OK;
}
#define K3_RCCL(R, e) do { const int rc__ = (e); if (rc__ != 0) { k3::set_error("RCCL error %d (%s) in %s", rc__, (R)->GetErrorString ? (R)->GetErrorString(rc__) : "?", #e); return K3_ERR_HIP; } } while (0)
constexpr int kNcclUint8 = 1, kNcclInt64 = 4, kNcclFloat32 = 7, kNcclSum = 0;      // ncclDataType_t / ncclRedOp_t (rccl.h:459-467, :441)
struct IdFile { char id[128]; uint64_t magic, nonce; };      // what rank 0 writes
constexpr uint64_t kIdMagic = 0x4b33636f6d6d3031ull;      // "K3comm01"
uint64_t run_nonce() {      // a launcher-supplied run identity (K3_COMM_NONCE, else torchrun's run id): ranks of different runs never accept each other's file
  const char *e = getenv("K3_COMM_NONCE"); if (!e || !*e) e = getenv("TORCHELASTIC_RUN_ID");
  if (!e || !*e) return 0;
  uint64_t h = 1469598103934665603ull; for (; *e; e++) { h ^= (unsigned char)*e; h *= 1099511628211ull; }
  return h ? h : 1;
}
}  // namespace

// The file protocol of the rendezvous by itself (no RCCL: tests/test_parallel_cpu.py runs it with several processes).  Rank 0 removes whatever a
// previous run left at `id_file`, then publishes {id, magic, nonce} atomically (rename); the other ranks poll for a file that (a) carries the magic and
// this run's nonce (K3_COMM_NONCE / TORCHELASTIC_RUN_ID; 0 when the launcher supplied none) AND (b) is not older than `stale_seconds` before their own
// start -- also with a nonce: torchrun's static rendezvous hands every run the same id ("none"), so the nonce alone does not tell two runs apart.  A file a
// crashed or one-rank run left behind is therefore only ever mistaken for the new one inside that window and before rank 0 has replaced it; launchers that
// restart within the window should set K3_COMM_NONCE per run.  k3_comm_create removes the file again once every rank has joined (ncclCommInitRank is
// collective), so a recipe directory can be reused run after run.
extern "C" int k3_comm_exchange_id(const char *id_file, int32_t rank, int32_t timeout_seconds, int32_t stale_seconds, const void *id_in, void *id_out) {
  K3_REQUIRE(id_file && id_out && rank >= 0 && (rank != 0 || id_in), "k3_comm_exchange_id: bad argument");
  const uint64_t nonce = run_nonce(); const time_t t_start = time(nullptr);
  if (rank == 0) {
    (void)unlink(id_file);
    IdFile rec; memcpy(rec.id, id_in, sizeof rec.id); rec.magic = kIdMagic; rec.nonce = nonce;
    const std::string tmp = std::string(id_file) + ".tmp";
    FILE *f = fopen(tmp.c_str(), "wb"); K3_REQUIRE(f && fwrite(&rec, sizeof rec, 1, f) == 1 && fclose(f) == 0 && rename(tmp.c_str(), id_file) == 0, "k3_comm_exchange_id: cannot write the id file");
    memcpy(id_out, id_in, sizeof rec.id); return K3_OK;
  }
  for (int i = 0; i < 20 * std::max(1, timeout_seconds); i++) {
    FILE *f = fopen(id_file, "rb");
    if (f) {
      IdFile rec; struct stat st; const bool got = fread(&rec, sizeof rec, 1, f) == 1 && fstat(fileno(f), &st) == 0; fclose(f);
      if (got && rec.magic == kIdMagic && rec.nonce == nonce && st.st_mtime + stale_seconds >= t_start) { memcpy(id_out, rec.id, sizeof rec.id); return K3_OK; }
    }
    usleep(50000);
  }
  k3::set_error("k3_comm_exchange_id: timed out waiting for rank 0's id file %s (a file of another run is not accepted)", id_file); return K3_ERR_ARG;
}

// One communicator per process (one process per GPU, hipSetDevice done by the caller).  The 128-byte ncclUniqueId travels through a file on a
// file system every rank sees (what a Kaldi recipe has anyway: exp/.../decode/): rank 0 writes <id_file>, the others wait for it.
extern "C" int k3_comm_create(const char *id_file, int32_t rank, int32_t world_size, int32_t timeout_seconds, void **comm) {
  K3_REQUIRE(id_file && comm && world_size >= 1 && rank >= 0 && rank < world_size, "k3_comm_create: bad argument");
  Rccl *R; { const int rc = rccl(&R); if (rc) return rc; }
  UniqueId id, mine; memset(&id, 0, sizeof id); memset(&mine, 0, sizeof mine);
  if (rank == 0) K3_RCCL(R, R->GetUniqueId(&mine));
  { const int rc = k3_comm_exchange_id(id_file, rank, timeout_seconds, 120, &mine, &id); if (rc) return rc; }
  void *c = nullptr;
  K3_RCCL(R, R->CommInitRank(&c, world_size, id, rank));
  if (rank == 0 && world_size > 1) (void)unlink(id_file);      // every rank has joined: nothing of this run stays behind (a one-rank communicator keeps it: nobody else reads it, tests look at it)
  *comm = c; return K3_OK;
}
extern "C" void k3_comm_destroy(void *comm) { Rccl *R; if (comm && rccl(&R) == K3_OK && R->CommDestroy) (void)R->CommDestroy(comm); }

// Sum of `count` floats over the ranks of `comm`, in place (ncclAllReduce over xGMI): the gradient exchange of synchronous data-parallel chain training
// (the reference averages models between jobs, egs/wsj/s5/steps/libs/nnet3/train/chain_objf/acoustic_model.py:121).
// Asynchronous on `stream`; a one-rank communicator leaves the buffer as it is.
extern "C" int k3_comm_allreduce_f32(void *comm, float *d_buf, int64_t count, void *stream) {
  K3_REQUIRE(comm && (d_buf || count == 0) && count >= 0, "k3_comm_allreduce_f32: bad argument");
  Rccl *R; { const int rc = rccl(&R); if (rc) return rc; }
  if (count == 0) return K3_OK;
  K3_RCCL(R, R->AllReduce(d_buf, d_buf, (size_t)count, kNcclFloat32, kNcclSum, comm, (hipStream_t)stream));
  return K3_OK;
}

// k3_fst_csr_bcast(handle, ncclComm_t).  On `root` *fst is the graph (k3_fst_create); on the other ranks *fst is NULL on entry and
// owns a graph of the same shape holding root's image on return.  `comm` is an ncclComm_t (k3_comm_create, or the application's own).
extern "C" int k3_fst_bcast(k3_fst **fst, void *comm, int32_t root, int32_t rank, void *stream) {
  K3_REQUIRE(fst && comm && (rank != root || *fst), "k3_fst_bcast: bad argument (the root rank must pass its graph)");
  Rccl *R; { const int rc = rccl(&R); if (rc) return rc; }
  hipStream_t st = (hipStream_t)stream;
  int64_t shape[5] = {0, 0, 0, 0, -1}; void *image = nullptr;
  if (rank == root) { const int rc = k3_fst_shape_and_image(*fst, shape, &image); if (rc) return rc; }
  int64_t *d_shape = nullptr; K3_HIP_CHECK(hipMalloc((void **)&d_shape, sizeof shape));
  struct Free { int64_t *p; ~Free() { (void)hipFree(p); } } free_shape{d_shape};      // (also on the early returns below)
  K3_HIP_CHECK(hipMemcpy(d_shape, shape, sizeof shape, hipMemcpyHostToDevice));
  K3_RCCL(R, R->Broadcast(d_shape, d_shape, 5, kNcclInt64, root, comm, st));
  K3_HIP_CHECK(hipStreamSynchronize(st));
  K3_HIP_CHECK(hipMemcpy(shape, d_shape, sizeof shape, hipMemcpyDeviceToHost));
  if (rank != root) { const int rc = k3_fst_create_shaped(shape, fst); if (rc) return rc; int64_t s2[5]; const int rc2 = k3_fst_shape_and_image(*fst, s2, &image); if (rc2) return rc2; }
  K3_RCCL(R, R->Broadcast(image, image, (size_t)shape[3], kNcclUint8, root, comm, st));
  K3_HIP_CHECK(hipStreamSynchronize(st));
  return K3_OK;
}
