// TEST DOUBLE for librccl -- lets TWO (or more) processes that share ONE GPU run the engine's data-parallel
// communicator (gt_comm_init with world > 1, engine.hip) for real: RCCL itself refuses two ranks on one device, and
// the build / GPU boxes have a single MI355X.  Selected with GT_RCCL_LIB=<path to this .so> (engine.hip: rccl_api()).
//
// Exports the seven symbols the engine binds: ncclGetUniqueId / CommInitRank / CommDestroy / AllReduce / GroupStart /
// GroupEnd / GetErrorString.  The ranks meet in a POSIX shared-memory segment named after the unique id.  An all-reduce
// honours the stream it is given:   D2H copy (async, pinned)  ->  host function on the stream: publish the rank's slot,
// barrier, sum the slots IN RANK ORDER (every rank computes the same sum: replicas stay bit-identical), barrier
// ->  H2D copy (async).  Nothing here is a product path; it only has to be correct and stream-ordered.
#include <hip/hip_runtime.h>
#include <fcntl.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <time.h>
#include <unistd.h>
#include <atomic>

namespace {
constexpr size_t SLOT_BYTES = (size_t)64 << 20;     // per-rank message capacity
constexpr int MAX_RANKS = 8;
enum { NCCL_FLOAT = 7, NCCL_DOUBLE = 8, NCCL_SUM = 0 };

struct Header {
  std::atomic<uint32_t> arrived;      // barrier counter
  std::atomic<uint32_t> generation;   // barrier generation
  std::atomic<uint32_t> attached;     // ranks that mapped the segment
  std::atomic<uint32_t> detached;
  char pad[4096 - 4 * sizeof(std::atomic<uint32_t>)];
};

struct Comm {
  int rank = 0, world = 1;
  char name[64];
  Header* hdr = nullptr;
  char* slots = nullptr;              // world x SLOT_BYTES behind the header
  size_t map_bytes = 0;
  char* stage = nullptr;              // pinned host staging, SLOT_BYTES
};

struct Job { Comm* c; size_t bytes; size_t count; int dtype; };

bool barrier(Comm* c) {
  Header* h = c->hdr;
  const uint32_t gen = h->generation.load(std::memory_order_acquire);
  if (h->arrived.fetch_add(1, std::memory_order_acq_rel) + 1 == (uint32_t)c->world) {
    h->arrived.store(0, std::memory_order_relaxed);
    h->generation.store(gen + 1, std::memory_order_release);
    return true;
  }
  timespec t0;
  clock_gettime(CLOCK_MONOTONIC, &t0);
  for (unsigned spin = 0;; ++spin) {
    if (h->generation.load(std::memory_order_acquire) != gen) return true;
    if ((spin & 1023) == 1023) {
      timespec t;
      clock_gettime(CLOCK_MONOTONIC, &t);
      if (t.tv_sec - t0.tv_sec > 120) { fprintf(stderr, "[fake_rccl] rank %d: peer never arrived at a barrier\n", c->rank); return false; }
      usleep(50);
    }
  }
}

void host_reduce(void* p) {
  Job* j = (Job*)p;
  Comm* c = j->c;
  memcpy(c->slots + (size_t)c->rank * SLOT_BYTES, c->stage, j->bytes);
  if (barrier(c)) {
    if (j->dtype == NCCL_FLOAT) {
      float* out = (float*)c->stage;
      for (size_t i = 0; i < j->count; ++i) {
        float s = ((const float*)c->slots)[i];
        for (int r = 1; r < c->world; ++r) s += ((const float*)(c->slots + (size_t)r * SLOT_BYTES))[i];
        out[i] = s;
      }
    } else {
      double* out = (double*)c->stage;
      for (size_t i = 0; i < j->count; ++i) {
        double s = ((const double*)c->slots)[i];
        for (int r = 1; r < c->world; ++r) s += ((const double*)(c->slots + (size_t)r * SLOT_BYTES))[i];
        out[i] = s;
      }
    }
    barrier(c);        // nobody overwrites a slot before every rank has read it
  }
  delete j;
}
}  // namespace

extern "C" {

const char* ncclGetErrorString(int r) {
  switch (r) {
    case 0: return "success";
    case 1: return "fake_rccl: unhandled HIP error";
    case 2: return "fake_rccl: system error (shared memory)";
    case 4: return "fake_rccl: invalid argument";
    default: return "fake_rccl: error";
  }
}

int ncclGetUniqueId(void* id) {
  static std::atomic<unsigned> counter{0};
  memset(id, 0, 128);
  timespec t;
  clock_gettime(CLOCK_REALTIME, &t);
  snprintf((char*)id, 64, "/gt_fake_rccl_%d_%ld_%u", (int)getpid(), (long)t.tv_nsec, counter.fetch_add(1));
  return 0;
}

struct FakeId { char internal[128]; };

int ncclCommInitRank(void** comm, int world, FakeId id, int rank) {
  if (!comm || world < 1 || world > MAX_RANKS || rank < 0 || rank >= world) return 4;
  Comm* c = new Comm();
  c->rank = rank; c->world = world;
  memcpy(c->name, id.internal, sizeof(c->name));
  c->name[sizeof(c->name) - 1] = 0;
  c->map_bytes = sizeof(Header) + (size_t)world * SLOT_BYTES;
  const int fd = shm_open(c->name, O_CREAT | O_RDWR, 0600);
  if (fd < 0) { delete c; return 2; }
  if (ftruncate(fd, (off_t)c->map_bytes) != 0) { close(fd); delete c; return 2; }     // a fresh segment reads as zeros
  void* m = mmap(nullptr, c->map_bytes, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
  close(fd);
  if (m == MAP_FAILED) { delete c; return 2; }
  c->hdr = (Header*)m;
  c->slots = (char*)m + sizeof(Header);
  if (hipHostMalloc((void**)&c->stage, SLOT_BYTES) != hipSuccess) { munmap(m, c->map_bytes); delete c; return 1; }
  c->hdr->attached.fetch_add(1);
  if (!barrier(c)) { return 2; }                                                        // collective, like the real call
  *comm = c;
  return 0;
}

int ncclCommDestroy(void* comm) {
  Comm* c = (Comm*)comm;
  if (!c) return 0;
  const bool last = c->hdr->detached.fetch_add(1) + 1 == (uint32_t)c->world;
  munmap((void*)c->hdr, c->map_bytes);
  if (last) shm_unlink(c->name);
  if (c->stage) (void)hipHostFree(c->stage);
  delete c;
  return 0;
}

int ncclAllReduce(const void* send, void* recv, size_t count, int dtype, int op, void* comm, hipStream_t stream) {
  Comm* c = (Comm*)comm;
  if (!c || !send || !recv || op != NCCL_SUM || (dtype != NCCL_FLOAT && dtype != NCCL_DOUBLE)) return 4;
  const size_t bytes = count * (dtype == NCCL_FLOAT ? 4 : 8);
  if (bytes > SLOT_BYTES) return 4;
  if (count == 0) return 0;
  // the single staging buffer is reused by every call: calls of one communicator must be issued on ONE stream (the
  // engine's communicator stream), whose order then serialises them
  if (hipMemcpyAsync(c->stage, send, bytes, hipMemcpyDeviceToHost, stream) != hipSuccess) return 1;
  if (hipLaunchHostFunc(stream, host_reduce, new Job{c, bytes, count, dtype}) != hipSuccess) return 1;
  if (hipMemcpyAsync(recv, c->stage, bytes, hipMemcpyHostToDevice, stream) != hipSuccess) return 1;
  return 0;
}

int ncclGroupStart() { return 0; }
int ncclGroupEnd() { return 0; }

}  // extern "C"
