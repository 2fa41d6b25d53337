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
  if (!r.h || !r.GetUniqueId || !r.CommInitRank || !r.Broadcast || !r.AllReduce) {
    k3::set_error("RCCL (librccl.so.1) is not available: %s", r.why.c_str());
    return K3_ERR_UNSUPPORTED;
  }
  *out = &r; return K3_OK;
}
#define K3_RCCL(R, e) do { const int rc__ = (e); if (rc__ != 0) { k3::set_error("RCCL error %d (%s) in %s", rc__, (R)->GetErrorString ? (R)->GetErrorString(rc__) : "?", #e); return K3_ERR_HIP; } } while (0)
constexpr int kNcclUint8 = 1, kNcclInt64 = 4, kNcclFloat32 = 7, kNcclSum = 0;      // ncclDataType_t / ncclRedOp_t (rccl.h:459-467, :441)
struct IdFile { char id[128]; uint64_t magic, nonce; };      // what rank 0 writes
struct Note { uint64_t magic, nonce, id_hash; };               // <id_file>.arrived.<rank> (a rank has read the id) and <id_file>.go (rank 0 has seen every rank)
constexpr uint64_t kIdMagic = 0x4b33636f6d6d3031ull;      // "K3comm01"
uint64_t fnv(const void *p, size_t n, uint64_t h = 1469598103934665603ull) {
  const unsigned char *c = (const unsigned char *)p;
  for (size_t i = 0; i < n; i++) {
    h ^= c[i];
    h *= 1099511628211ull;
  }
  return h;
}
// A launcher-supplied run identity (K3_COMM_NONCE, else torchrun's run id): ranks of different runs never accept each other's files.  *unique = the identity is
// K3_COMM_NONCE, which the caller makes up per run; torchrun's static rendezvous hands EVERY run the id "none", so that one alone does not tell two runs apart.
uint64_t run_nonce(bool *unique = nullptr) {
  const char *e = getenv("K3_COMM_NONCE"); if (unique) *unique = e && *e;
  if (!e || !*e) e = getenv("TORCHELASTIC_RUN_ID");
  if (!e || !*e) return 0;
  const uint64_t h = fnv(e, strlen(e));
  return h ? h : 1;
}
bool write_atomically(const std::string &path, const void *rec, size_t n) {
  const std::string tmp = path + ".tmp";
  FILE *f = fopen(tmp.c_str(), "wb"); if (!f) return false;
  const bool ok = fwrite(rec, n, 1, f) == 1; return (fclose(f) == 0) && ok && rename(tmp.c_str(), path.c_str()) == 0;
}
// a file of exactly the record's size whose age passes: with a per-run K3_COMM_NONCE the nonce decides alone (a rank that enters late must not reject a valid
// file for its age); otherwise nothing older than max(stale_seconds, timeout_seconds) before the caller's start is taken for this run's
template <typename Rec> bool read_fresh(const std::string &path, Rec *rec, time_t t_start, int window, bool nonce_is_unique, struct timespec *mtime = nullptr) {
  FILE *f = fopen(path.c_str(), "rb"); if (!f) return false;
  // (a truncated or over-long file is not a record)
  struct stat st;
  char extra;
  const bool got = fread(rec, sizeof *rec, 1, f) == 1 && fread(&extra, 1, 1, f) == 0 && fstat(fileno(f), &st) == 0;
  fclose(f);
  if (got && mtime) *mtime = st.st_mtim;
  return got && (nonce_is_unique || st.st_mtime + window >= t_start);
}
bool not_older(const struct timespec &a, const struct timespec &b) { return a.tv_sec > b.tv_sec || (a.tv_sec == b.tv_sec && a.tv_nsec >= b.tv_nsec); }
std::string note_path(const char *id_file, const char *what, int rank = -1) {
  std::string p = std::string(id_file) + "." + what;
  if (rank >= 0) p += "." + std::to_string(rank);
  return p;
}
}  // namespace

// The file protocol of the rendezvous by itself (no RCCL: tests/test_parallel_cpu.py runs it with several processes).  Rank 0 removes whatever a
// previous run left at `id_file`, then publishes {id, magic, nonce} atomically (rename); the other ranks poll for a file that (a) carries the magic and
// this run's nonce (K3_COMM_NONCE / TORCHELASTIC_RUN_ID; 0 when the launcher supplied none) AND (b) unless the nonce is a per-run K3_COMM_NONCE, is not older
// than max(stale_seconds, timeout_seconds) before their own start: torchrun's static rendezvous hands every run the same id ("none"), so that nonce alone does
// not tell two runs apart.  A file a crashed run left behind is therefore only ever mistaken for the new one inside that window and before rank 0 has replaced
// it -- and k3_comm_rendezvous below does not act on an id before rank 0 has confirmed it.
extern "C" int k3_comm_exchange_id(const char *id_file, int32_t rank, int32_t timeout_seconds, int32_t stale_seconds, const void *id_in, void *id_out) {
  K3_REQUIRE(id_file && id_out && rank >= 0 && (rank != 0 || id_in), "k3_comm_exchange_id: bad argument");
  bool unique = false; const uint64_t nonce = run_nonce(&unique); const time_t t_start = time(nullptr); const int window = std::max(stale_seconds, timeout_seconds);
  if (rank == 0) {
    (void)unlink(id_file);
    IdFile rec; memcpy(rec.id, id_in, sizeof rec.id); rec.magic = kIdMagic; rec.nonce = nonce;
    K3_REQUIRE(write_atomically(id_file, &rec, sizeof rec), "k3_comm_exchange_id: cannot write the id file");
    memcpy(id_out, id_in, sizeof rec.id); return K3_OK;
  }
  for (int i = 0; i < 20 * std::max(1, timeout_seconds); i++) {
    IdFile rec;
    if (read_fresh(id_file, &rec, t_start, window, unique) && rec.magic == kIdMagic && rec.nonce == nonce) { memcpy(id_out, rec.id, sizeof rec.id); return K3_OK; }
    usleep(50000);
  }
  k3::set_error("k3_comm_exchange_id: timed out waiting for rank 0's id file %s (a file of another run is not accepted)", id_file); return K3_ERR_ARG;
}

// The whole rendezvous in front of ncclCommInitRank, which has no deadline of its own (a rank that never arrives would leave the others blocked in its bootstrap
// for ever): nobody enters the collective before EVERY rank has been seen.  Rank r > 0 reads the id, announces itself in <id_file>.arrived.<r> (naming the id it
// read; it re-reads and re-announces if rank 0 replaces a stale file under it) and waits for <id_file>.go; rank 0 publishes the id, waits for all world_size - 1
// announcements of THAT id and then writes .go.  A rank takes a .go only if it was written AFTER its own announcement (file times of the shared file system, to the
// nanosecond): the .go a crashed run left next to its id file -- same launcher id, inside the age window -- is older than any announcement of this run, so a rank that
// starts before rank 0 has cleaned up cannot walk into ncclCommInitRank with the dead run's id.  Every wait ends after timeout_seconds with an error that
// names what was missing (the ranks that never arrived / rank 0's confirmation); on failure rank 0 withdraws its files, so late ranks time out too instead of joining a dead run.
extern "C" int k3_comm_rendezvous(const char *id_file, int32_t rank, int32_t world_size, int32_t timeout_seconds, int32_t stale_seconds, const void *id_in, void *id_out) {
  K3_REQUIRE(id_file && id_out && world_size >= 1 && rank >= 0 && rank < world_size && (rank != 0 || id_in), "k3_comm_rendezvous: bad argument");
  bool unique = false; const uint64_t nonce = run_nonce(&unique); const time_t t_start = time(nullptr); const int window = std::max(stale_seconds, timeout_seconds);
  const int polls = 20 * std::max(1, timeout_seconds); const std::string go = note_path(id_file, "go");
  if (world_size == 1) return k3_comm_exchange_id(id_file, 0, timeout_seconds, stale_seconds, id_in, id_out);      // (nobody to wait for)
  if (rank == 0) {
    (void)unlink(go.c_str()); for (int r = 1; r < world_size; r++) (void)unlink(note_path(id_file, "arrived", r).c_str());
    { const int rc = k3_comm_exchange_id(id_file, 0, timeout_seconds, stale_seconds, id_in, id_out); if (rc) return rc; }
    const Note want{kIdMagic, nonce, fnv(id_in, 128)}; std::vector<char> seen(world_size, 0); int missing = world_size - 1;
    for (int i = 0; i < polls && missing > 0; i++) {
      for (int r = 1; r < world_size; r++) {
        Note n;
        if (!seen[r] && read_fresh(note_path(id_file, "arrived", r), &n, t_start, window, true) && n.magic == want.magic && n.nonce == want.nonce &&
            n.id_hash == want.id_hash) {
          seen[r] = 1;
          missing--;
        }
      }
      if (missing > 0) usleep(50000);
    }
    if (missing > 0) {
      std::string who; for (int r = 1; r < world_size; r++) if (!seen[r]) who += (who.empty() ? "" : ", ") + std::to_string(r);
      (void)unlink(id_file); for (int r = 1; r < world_size; r++) (void)unlink(note_path(id_file, "arrived", r).c_str());
      k3::set_error("k3_comm_rendezvous: %d of %d ranks never arrived within %d s (missing: %s); id file %s withdrawn", missing, world_size, timeout_seconds,
          who.c_str(), id_file);
      return K3_ERR_ARG;
    }
    K3_REQUIRE(write_atomically(go, &want, sizeof want), "k3_comm_rendezvous: cannot write the confirmation file");
    return K3_OK;
  }
  uint64_t announced = 0; bool have = false; IdFile rec; struct timespec announced_at = {0, 0};
  for (int i = 0; i < polls; i++) {
    IdFile cur;
    if (read_fresh(id_file, &cur, t_start, window, unique) && cur.magic == kIdMagic && cur.nonce == nonce) {
      const uint64_t h = fnv(cur.id, sizeof cur.id);
      if (!have || h != announced) {
        const Note n{kIdMagic, nonce, h}; K3_REQUIRE(write_atomically(note_path(id_file, "arrived", rank), &n, sizeof n), "k3_comm_rendezvous: cannot write the arrival file");
        { struct stat st; K3_REQUIRE(stat(note_path(id_file, "arrived", rank).c_str(), &st) == 0, "k3_comm_rendezvous: cannot stat the arrival file"); announced_at = st.st_mtim; }
        rec = cur; announced = h; have = true;
      }
      Note g; struct timespec go_at = {0, 0};
      if (read_fresh(go, &g, t_start, window, unique, &go_at) && g.magic == kIdMagic && g.nonce == nonce && g.id_hash == announced && not_older(go_at, announced_at)) {
        memcpy(id_out, rec.id, sizeof rec.id);
        return K3_OK;
      }
    }
    usleep(50000);
  }
  (void)unlink(note_path(id_file, "arrived", rank).c_str());
  k3::set_error(have ? "k3_comm_rendezvous: rank %d read the id but rank 0 never confirmed that all %d ranks arrived within %d s (%s)" :
      "k3_comm_rendezvous: rank %d of %d timed out after %d s waiting for rank 0's id file %s (a file of another run is not accepted)",
                rank, world_size, timeout_seconds, id_file);
  return K3_ERR_ARG;
}

// One communicator per process (one process per GPU, hipSetDevice done by the caller).  The 128-byte ncclUniqueId travels through a file on a
// file system every rank sees (what a Kaldi recipe has anyway: exp/.../decode/): rank 0 writes <id_file>, the others wait for it.
extern "C" int k3_comm_create(const char *id_file, int32_t rank, int32_t world_size, int32_t timeout_seconds, void **comm) {
  K3_REQUIRE(id_file && comm && world_size >= 1 && rank >= 0 && rank < world_size, "k3_comm_create: bad argument");
  Rccl *R; { const int rc = rccl(&R); if (rc) return rc; }
  UniqueId id, mine; memset(&id, 0, sizeof id); memset(&mine, 0, sizeof mine);
  if (rank == 0) K3_RCCL(R, R->GetUniqueId(&mine));
  // (returns once EVERY rank has been seen, or with an error inside the timeout)
  {
    const int rc = k3_comm_rendezvous(id_file, rank, world_size, timeout_seconds, 120, &mine, &id);
    if (rc) return rc;
  }
  void *c = nullptr;
  K3_RCCL(R, R->CommInitRank(&c, world_size, id, rank));
  // every rank has joined: nothing of this run stays behind (a one-rank communicator keeps the id file: nobody else reads it, tests look at it)
  if (rank == 0 && world_size > 1) {
    (void)unlink(id_file); (void)unlink(note_path(id_file, "go").c_str()); for (int r = 1; r < world_size; r++) (void)unlink(note_path(id_file, "arrived", r).c_str());
  }
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
  if (rank != root) {
    const int rc = k3_fst_create_shaped(shape, fst);
    if (rc) return rc;
    int64_t s2[5];
    const int rc2 = k3_fst_shape_and_image(*fst, s2, &image);
    if (rc2) return rc2;
  }
  K3_RCCL(R, R->Broadcast(image, image, (size_t)shape[3], kNcclUint8, root, comm, st));
  K3_HIP_CHECK(hipStreamSynchronize(st));
  return K3_OK;
}
