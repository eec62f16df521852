/* fake_rccl.c -- TEST-ONLY stand-in for librccl, so that the RCCL branch of tsh_search_sharded
 * (tostore_amd/csrc/tsh_host_comm.inl.h: comm_exchange_group, comm_agree, the result all-gather) runs with
 * MORE THAN ONE rank on a box with one GPU.  Real RCCL refuses two ranks on one device; this library implements
 * the four entry points the product resolves with dlsym -- ncclGetUniqueId, ncclCommInitRank, ncclAllGather,
 * ncclCommDestroy (+ ncclGetErrorString) -- between PROCESSES over a POSIX shared-memory segment: device buffers
 * are bounced through the host on the stream the caller passes, so every pitch, slice and offset the product
 * computes for W ranks is exercised for real.  It is selected with TSH_RCCL_LIB=<path> (read where the product
 * dlopens librccl) and is never loaded otherwise.  Not a performance model: an all-gather here costs two
 * process barriers and 1 + W host copies.
 *
 * Semantics kept from NCCL: the call is ordered on `stream` (work queued before it is complete before the send
 * buffer is read; work queued after it sees the receive buffer filled); every rank must call the collectives in
 * the same order; rank r's bytes land at recv + r * bytes.
 * Differences: the call blocks the host until the data arrived (a legal refinement of "asynchronous");
 * a rank that waits longer than TSH_FAKE_RCCL_TIMEOUT_S (default 120) for its peers gets ncclSystemError.
 * TSH_FAKE_RCCL_FAIL_AT=n makes the n-th all-gather (1-based, per communicator) fail on every rank that has the
 * variable set -- the product's TSH_E_RCCL path.
 *
 * Build: gcc -O2 -fPIC -shared -D__HIP_PLATFORM_AMD__ -I/opt/rocm/include fake_rccl.c -o libfake_rccl.so \
 *            -L/opt/rocm/lib -lamdhip64 -lrt
 */
#define _GNU_SOURCE
#include <errno.h>
#include <fcntl.h>
#include <hip/hip_runtime_api.h>
#include <stdatomic.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <time.h>
#include <unistd.h>

typedef struct {
  char internal[128];
} ncclUniqueId;

enum { ncclSuccess = 0, ncclUnhandledCudaError = 1, ncclSystemError = 2, ncclInternalError = 3, ncclInvalidArgument = 4 };

typedef struct {
  _Atomic uint32_t arrive;
  _Atomic uint32_t generation;
  _Atomic uint32_t poisoned; /* a rank gave up: nobody waits any more */
  uint32_t pad0;
  uint8_t pad[4096 - 16];
} ShmHeader;

typedef struct {
  ShmHeader *shm;
  uint8_t *slots;
  size_t map_bytes, slot_bytes;
  int world, rank;
  long calls, fail_at;
  double timeout_s;
} FakeComm;

static double now_s(void) {
  struct timespec ts;
  clock_gettime(CLOCK_MONOTONIC, &ts);
  return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec;
}

/* sense-counting barrier over the segment; 0 = everybody arrived */
static int barrier(FakeComm *c) {
  ShmHeader *h = c->shm;
  const uint32_t gen = atomic_load(&h->generation);
  if (atomic_fetch_add(&h->arrive, 1) == (uint32_t)c->world - 1) {
    atomic_store(&h->arrive, 0);
    atomic_fetch_add(&h->generation, 1);
    return 0;
  }
  const double t0 = now_s();
  for (unsigned spin = 0; atomic_load(&h->generation) == gen; ++spin) {
    if (atomic_load(&h->poisoned)) return -1;
    if (spin > 200) { /* be kind to a small CPU quota shared by all ranks */
      struct timespec ts = {0, 20000};
      nanosleep(&ts, NULL);
      if ((spin & 1023) == 0 && now_s() - t0 > c->timeout_s) {
        atomic_store(&h->poisoned, 1);
        return -1;
      }
    }
  }
  return 0;
}

int ncclGetUniqueId(ncclUniqueId *id) {
  if (!id) return ncclInvalidArgument;
  memset(id, 0, sizeof *id);
  uint64_t r[2] = {0, 0};
  int fd = open("/dev/urandom", O_RDONLY);
  if (fd >= 0) {
    if (read(fd, r, sizeof r) != (ssize_t)sizeof r) r[0] = (uint64_t)now_s();
    close(fd);
  }
  snprintf(id->internal, sizeof id->internal, "/tsh_fake_rccl_%d_%016llx%016llx", (int)getpid(), (unsigned long long)r[0],
           (unsigned long long)r[1]);
  return ncclSuccess;
}

int ncclCommInitRank(void **comm, int nranks, ncclUniqueId id, int rank) {
  if (!comm || nranks < 1 || rank < 0 || rank >= nranks) return ncclInvalidArgument;
  if (strncmp(id.internal, "/tsh_fake_rccl_", 15) != 0) return ncclInvalidArgument; /* an id of the real library */
  id.internal[sizeof id.internal - 1] = 0;
  FakeComm *c = (FakeComm *)calloc(1, sizeof *c);
  if (!c) return ncclSystemError;
  const char *s = getenv("TSH_FAKE_RCCL_SLOT");
  c->slot_bytes = s ? (size_t)strtoull(s, NULL, 10) : ((size_t)4 << 20);
  if (c->slot_bytes < 64) c->slot_bytes = 64;
  s = getenv("TSH_FAKE_RCCL_TIMEOUT_S");
  c->timeout_s = s ? atof(s) : 120.0;
  s = getenv("TSH_FAKE_RCCL_FAIL_AT");
  c->fail_at = s ? atol(s) : 0;
  c->world = nranks;
  c->rank = rank;
  c->map_bytes = sizeof(ShmHeader) + (size_t)nranks * c->slot_bytes;
  /* every rank creates-or-opens and sizes the segment alike; a fresh segment is all zeros = the barrier's start state */
  int fd = shm_open(id.internal, O_CREAT | O_RDWR, 0600);
  if (fd < 0 || ftruncate(fd, (off_t)c->map_bytes) != 0) {
    if (fd >= 0) close(fd);
    free(c);
    return ncclSystemError;
  }
  void *p = mmap(NULL, c->map_bytes, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
  close(fd);
  if (p == MAP_FAILED) {
    free(c);
    return ncclSystemError;
  }
  c->shm = (ShmHeader *)p;
  c->slots = (uint8_t *)p + sizeof(ShmHeader);
  int rc = barrier(c); /* collective, like the real call */
  if (rank == 0) shm_unlink(id.internal); /* everybody has it mapped (or gave up): no name left behind in /dev/shm */
  if (rc) {
    munmap(p, c->map_bytes);
    free(c);
    return ncclSystemError;
  }
  *comm = c;
  return ncclSuccess;
}

static size_t type_bytes(int dt) {
  switch (dt) {
    case 0: case 1: return 1;         /* ncclInt8 / ncclChar, ncclUint8 */
    case 2: case 3: case 7: return 4; /* ncclInt32, ncclUint32, ncclFloat32 */
    case 4: case 5: case 8: return 8; /* ncclInt64, ncclUint64, ncclFloat64 */
    case 6: case 9: return 2;         /* ncclFloat16, ncclBfloat16 */
    default: return 0;
  }
}

int ncclAllGather(const void *send, void *recv, size_t count, int datatype, void *comm, hipStream_t stream) {
  FakeComm *c = (FakeComm *)comm;
  const size_t bytes = count * type_bytes(datatype);
  if (!c || !send || !recv || type_bytes(datatype) == 0) return ncclInvalidArgument;
  c->calls++;
  if (c->fail_at && c->calls == c->fail_at) return ncclInternalError;
  if (hipStreamSynchronize(stream) != hipSuccess) return ncclUnhandledCudaError; /* producers of `send` are done */
  for (size_t off = 0; off < bytes || (bytes == 0 && off == 0); off += c->slot_bytes) {
    const size_t n = bytes - off < c->slot_bytes ? bytes - off : c->slot_bytes;
    if (n && hipMemcpyAsync(c->slots + (size_t)c->rank * c->slot_bytes, (const uint8_t *)send + off, n, hipMemcpyDeviceToHost,
                            stream) != hipSuccess)
      return ncclUnhandledCudaError;
    if (hipStreamSynchronize(stream) != hipSuccess) return ncclUnhandledCudaError;
    if (barrier(c)) return ncclSystemError; /* every rank's piece is in its slot */
    for (int w = 0; w < c->world && n; ++w)
      if (hipMemcpyAsync((uint8_t *)recv + (size_t)w * bytes + off, c->slots + (size_t)w * c->slot_bytes, n,
                         hipMemcpyHostToDevice, stream) != hipSuccess)
        return ncclUnhandledCudaError;
    if (hipStreamSynchronize(stream) != hipSuccess) return ncclUnhandledCudaError;
    if (barrier(c)) return ncclSystemError; /* everybody has read the slots: they may be overwritten */
    if (bytes == 0) break;
  }
  return ncclSuccess;
}

int ncclCommDestroy(void *comm) {
  FakeComm *c = (FakeComm *)comm;
  if (!c) return ncclSuccess;
  munmap(c->shm, c->map_bytes);
  free(c);
  return ncclSuccess;
}

const char *ncclGetErrorString(int rc) {
  switch (rc) {
    case ncclSuccess: return "no error (fake rccl)";
    case ncclUnhandledCudaError: return "unhandled HIP error (fake rccl)";
    case ncclSystemError: return "peer timeout / shared memory error (fake rccl)";
    case ncclInternalError: return "injected failure (fake rccl, TSH_FAKE_RCCL_FAIL_AT)";
    case ncclInvalidArgument: return "invalid argument (fake rccl)";
    default: return "unknown (fake rccl)";
  }
}
