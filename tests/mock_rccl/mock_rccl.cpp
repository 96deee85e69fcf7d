// TEST INFRASTRUCTURE -- not part of the product.
// A stand-in for librccl that moves data between PROCESSES SHARING ONE GPU through POSIX shared memory, so that the
// library's native transport (csrc/comm.hip: neighbour tables, grouped ncclSend / ncclRecv, all-reduces, the asynchronous
// side stream) can run with 2-4 ranks on the one-GPU test box, where RCCL itself refuses several ranks per device.
// Selected with ALFI_RCCL_LIB=<this library>; exports exactly the entry points comm.hip resolves.
//
// Semantics kept: stream order (every operation first waits for the work queued on its stream and has finished when the
// call returns), message order per (source, destination) pair, grouped calls progress all their sends and receives
// together (no deadlock for symmetric exchanges), sum all-reduce with a fixed (rank) order.  Not kept: asynchrony.
#include <fcntl.h>
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>
#include <sched.h>
#include <sys/mman.h>
#include <unistd.h>

#include <atomic>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

namespace {

constexpr size_t SLOT_BYTES = 16u << 20;
constexpr int MAX_RANKS = 8;
constexpr size_t RED_MAX = 4096;      // doubles per all-reduce

struct Slot {
  std::atomic<uint64_t> written, read;
  uint64_t bytes;
  char pad[40];
  char data[SLOT_BYTES];
};

struct Shared {
  std::atomic<uint64_t> barrier_count;
  std::atomic<uint64_t> barrier_gen;
  char pad[48];
  double red[MAX_RANKS][RED_MAX];
  Slot slots[1];      // nranks * nranks: slot(src, dst)
};

struct Comm {
  int rank = 0, n = 0;
  Shared* sh = nullptr;
  size_t bytes = 0;
  Slot& slot(int src, int dst) { return sh->slots[(size_t)src * n + dst]; }
  void barrier() {
    const uint64_t gen = sh->barrier_gen.load();
    if (sh->barrier_count.fetch_add(1) + 1 == (uint64_t)n) {
      sh->barrier_count.store(0);
      sh->barrier_gen.fetch_add(1);
    } else {
      while (sh->barrier_gen.load() == gen) sched_yield();
    }
  }
};

struct Op {
  bool send;
  void* buf;
  size_t bytes;
  int peer;
  Comm* comm;
  hipStream_t stream;
  bool done;
};

thread_local int g_depth = 0;
thread_local std::vector<Op> g_queue;

size_t type_bytes(ncclDataType_t t) {
  switch (t) {
    case ncclInt8: case ncclUint8: return 1;
    case ncclFloat16: return 2;
    case ncclInt32: case ncclUint32: case ncclFloat32: return 4;
    default: return 8;
  }
}

ncclResult_t progress(std::vector<Op>& ops) {
  // everything queued on the streams involved has to be complete before the data leaves / arrives
  for (const Op& o : ops)
    if (hipStreamSynchronize(o.stream) != hipSuccess) return ncclUnhandledCudaError;
  size_t left = ops.size();
  while (left > 0) {
    bool moved = false;
    for (size_t i = 0; i < ops.size(); ++i) {
      Op& o = ops[i];
      if (o.done) continue;
      // message order per pair: an operation waits for the earlier ones with the same peer and direction
      bool blocked = false;
      for (size_t j = 0; j < i && !blocked; ++j) blocked = !ops[j].done && ops[j].send == o.send && ops[j].peer == o.peer;
      if (blocked) continue;
      if (o.bytes > SLOT_BYTES) return ncclInternalError;
      if (o.send) {
        Slot& s = o.comm->slot(o.comm->rank, o.peer);
        if (s.written.load() != s.read.load()) continue;           // the previous message has not been taken yet
        if (hipMemcpy(s.data, o.buf, o.bytes, hipMemcpyDeviceToHost) != hipSuccess) return ncclUnhandledCudaError;
        s.bytes = o.bytes;
        s.written.fetch_add(1);
      } else {
        Slot& s = o.comm->slot(o.peer, o.comm->rank);
        if (s.written.load() == s.read.load()) continue;           // nothing there yet
        if (s.bytes != o.bytes) {
          fprintf(stderr, "mock_rccl: rank %d expects %zu bytes from %d, message has %llu\n", o.comm->rank, o.bytes, o.peer,
                  (unsigned long long)s.bytes);
          return ncclInvalidArgument;
        }
        if (hipMemcpy(o.buf, s.data, o.bytes, hipMemcpyHostToDevice) != hipSuccess) return ncclUnhandledCudaError;
        s.read.fetch_add(1);
      }
      o.done = true;
      --left;
      moved = true;
    }
    if (!moved) sched_yield();
  }
  return ncclSuccess;
}

}  // namespace

extern "C" {

ncclResult_t ncclGetUniqueId(ncclUniqueId* id) {
  static std::atomic<int> counter{0};
  std::memset(id, 0, sizeof(*id));
  snprintf(id->internal, sizeof(id->internal), "/alfi_mock_rccl_%d_%d", (int)getpid(), counter.fetch_add(1));
  return ncclSuccess;
}

ncclResult_t ncclCommInitRank(ncclComm_t* out, int nranks, ncclUniqueId id, int rank) {
  if (nranks < 1 || nranks > MAX_RANKS || rank < 0 || rank >= nranks) return ncclInvalidArgument;
  Comm* c = new Comm;
  c->rank = rank;
  c->n = nranks;
  c->bytes = sizeof(Shared) + sizeof(Slot) * ((size_t)nranks * nranks - 1);
  const int fd = shm_open(id.internal, O_CREAT | O_RDWR, 0600);
  if (fd < 0 || ftruncate(fd, (off_t)c->bytes) != 0) return ncclSystemError;
  void* p = mmap(nullptr, c->bytes, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
  close(fd);
  if (p == MAP_FAILED) return ncclSystemError;
  c->sh = static_cast<Shared*>(p);          // a fresh segment is zero-filled: counters start at 0
  c->barrier();
  if (rank == 0) shm_unlink(id.internal);   // everybody is attached: the name can go
  c->barrier();
  *out = reinterpret_cast<ncclComm_t>(c);
  return ncclSuccess;
}

ncclResult_t ncclCommCount(const ncclComm_t comm, int* count) {
  *count = reinterpret_cast<const Comm*>(comm)->n;
  return ncclSuccess;
}

ncclResult_t ncclCommUserRank(const ncclComm_t comm, int* rank) {
  *rank = reinterpret_cast<const Comm*>(comm)->rank;
  return ncclSuccess;
}

ncclResult_t ncclCommDestroy(ncclComm_t comm) {
  Comm* c = reinterpret_cast<Comm*>(comm);
  if (!c) return ncclSuccess;
  munmap(c->sh, c->bytes);
  delete c;
  return ncclSuccess;
}

ncclResult_t ncclGroupStart() {
  ++g_depth;
  return ncclSuccess;
}

ncclResult_t ncclGroupEnd() {
  if (--g_depth > 0) return ncclSuccess;
  std::vector<Op> ops;
  ops.swap(g_queue);
  return progress(ops);
}

ncclResult_t ncclSend(const void* buf, size_t count, ncclDataType_t type, int peer, ncclComm_t comm, hipStream_t stream) {
  g_queue.push_back({true, const_cast<void*>(buf), count * type_bytes(type), peer, reinterpret_cast<Comm*>(comm), stream, false});
  if (g_depth > 0) return ncclSuccess;
  std::vector<Op> ops;
  ops.swap(g_queue);
  return progress(ops);
}

ncclResult_t ncclRecv(void* buf, size_t count, ncclDataType_t type, int peer, ncclComm_t comm, hipStream_t stream) {
  g_queue.push_back({false, buf, count * type_bytes(type), peer, reinterpret_cast<Comm*>(comm), stream, false});
  if (g_depth > 0) return ncclSuccess;
  std::vector<Op> ops;
  ops.swap(g_queue);
  return progress(ops);
}

ncclResult_t ncclAllReduce(const void* send, void* recv, size_t count, ncclDataType_t type, ncclRedOp_t op, ncclComm_t comm,
                           hipStream_t stream) {
  Comm* c = reinterpret_cast<Comm*>(comm);
  if (type != ncclFloat64 || op != ncclSum || count > RED_MAX) return ncclInvalidArgument;
  if (hipStreamSynchronize(stream) != hipSuccess) return ncclUnhandledCudaError;
  if (hipMemcpy(c->sh->red[c->rank], send, count * 8, hipMemcpyDeviceToHost) != hipSuccess) return ncclUnhandledCudaError;
  c->barrier();
  std::vector<double> sum(count, 0.0);
  for (int r = 0; r < c->n; ++r)
    for (size_t i = 0; i < count; ++i) sum[i] += c->sh->red[r][i];
  c->barrier();                                    // nobody overwrites its contribution before all have read it
  if (hipMemcpy(recv, sum.data(), count * 8, hipMemcpyHostToDevice) != hipSuccess) return ncclUnhandledCudaError;
  return ncclSuccess;
}

const char* ncclGetErrorString(ncclResult_t r) {
  return r == ncclSuccess ? "no error" : "mock_rccl error";
}

}  // extern "C"
