/* librccl_standin.c - TEST INFRASTRUCTURE ONLY (tests/test_gpu_rccl_standin.py builds it as librccl.so.1 in a temp directory and puts that
 * directory in front of a SUBPROCESS's LD_LIBRARY_PATH). Nothing in the product links, loads or ships it: libyolact_hip.so's
 * dlopen("librccl.so.1") is unchanged, and with the real librccl on the loader's path the real one is found.
 *
 * Why it exists: the path's one collective - the weight replication of BASELINE.json configs[3] / [4] (the
 * reference has no counterpart: one process, one device, src/main.rs:63-75) - marshals ncclUniqueId BY VALUE, passes ncclUint8 = 1,
 * orders the broadcast on the handle's stream and owns the communicator's lifetime, all by hand against a dlopen'ed library
 * (csrc/engine.hip: yh_rccl_unique_id, yh_rank_broadcast_weights, yh_group_broadcast_weights). Real RCCL refuses two ranks on one GPU
 * and the test box has one GPU, so that code had only ever run with n = 1. This stand-in implements the eight entry points the library
 * binds, with the real signatures, so that n = 2 executes on ONE device:
 *   - ranks in different processes (ncclCommInitRank) meet in a POSIX shared-memory segment named by the unique id; the broadcast
 *     travels root device -> shared host memory -> receiver device, each leg a hipMemcpyAsync on the CALLER's stream;
 *   - communicators of one process (ncclCommInitAll) broadcast device-to-device on each receiver's stream behind an event recorded on
 *     the root's stream (grouped between ncclGroupStart / ncclGroupEnd, as the library issues them).
 * It checks what the real library would: the id that arrives by value is the one that was handed out, datatype is ncclUint8, every rank
 * passes the same count and root, a communicator is not used after ncclCommDestroy. Fault injection for the error-path tests, read from
 * the environment of the process that loads it (the PRODUCT reads no environment variable; this is not the product):
 *   RCCL_STANDIN_FAIL = "initrank" | "initall" | "broadcast" | "uniqueid"    that call returns ncclSystemError
 *   RCCL_STANDIN_TIMEOUT_S = seconds a rank waits for its peers before giving up with ncclSystemError (default 20)
 */
#define __HIP_PLATFORM_AMD__ 1
#include <errno.h>
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

enum { ncclSuccess = 0, ncclUnhandledCudaError = 1, ncclSystemError = 2, ncclInternalError = 3, ncclInvalidArgument = 4, ncclInvalidUsage = 5 };
typedef struct { char internal[128]; } ncclUniqueId;

#define MAGIC_ID 0x31444953u   /* "SID1" */
#define MAGIC_COMM 0x4D4D4F43u /* "COMM" */
#define MAX_LOCAL 16

typedef struct {   /* the shared-memory rendezvous of one communicator spanning processes */
    volatile uint32_t arrived;      /* ranks that have called ncclCommInitRank */
    volatile uint32_t bcast_seq;    /* broadcasts the root has published */
    volatile uint32_t consumed;     /* receivers that have taken the current broadcast */
    volatile uint32_t root, dtype;
    volatile uint64_t count;
    volatile uint32_t left;         /* ranks that have destroyed their communicator */
} Shared;

struct LocalGroup;
typedef struct Comm {
    uint32_t magic;
    int rank, nranks, dev;
    /* multi-process form */
    Shared* sh;
    char name[64];
    uint32_t seq_seen;
    /* single-process form (ncclCommInitAll) */
    struct LocalGroup* lg;
} Comm;
typedef struct LocalGroup { int n, alive; Comm* c[MAX_LOCAL]; } LocalGroup;
typedef Comm* ncclComm_t;

/* a broadcast recorded between ncclGroupStart and ncclGroupEnd (single-process form) */
typedef struct { const void* send; void* recv; size_t count; int root; Comm* comm; hipStream_t stream; } Pending;
static __thread int g_group_depth = 0;
static __thread Pending g_pending[MAX_LOCAL];
static __thread int g_npending = 0;

static int fail_at(const char* what) {
    const char* f = getenv("RCCL_STANDIN_FAIL");
    return f && strcmp(f, what) == 0;
}
static double now_s(void) { struct timespec t; clock_gettime(CLOCK_MONOTONIC, &t); return t.tv_sec + 1e-9 * t.tv_nsec; }
static double timeout_s(void) { const char* t = getenv("RCCL_STANDIN_TIMEOUT_S"); return t ? atof(t) : 20.0; }
static int wait_until(volatile uint32_t* word, uint32_t at_least) {
    const double t0 = now_s(), lim = timeout_s();
    while (*(volatile uint32_t*)word < at_least) {
        if (now_s() - t0 > lim) return 0;
        usleep(200);
    }
    __sync_synchronize();
    return 1;
}

const char* ncclGetErrorString(int e) {
    switch (e) {
        case ncclSuccess: return "no error";
        case ncclUnhandledCudaError: return "unhandled cuda error (stand-in librccl)";
        case ncclSystemError: return "unhandled system error (stand-in librccl)";
        case ncclInternalError: return "internal error (stand-in librccl)";
        case ncclInvalidArgument: return "invalid argument (stand-in librccl)";
        case ncclInvalidUsage: return "invalid usage (stand-in librccl)";
    }
    return "unknown result code (stand-in librccl)";
}

int ncclGetUniqueId(ncclUniqueId* id) {
    if (!id) return ncclInvalidArgument;
    if (fail_at("uniqueid")) return ncclSystemError;
    memset(id, 0, sizeof *id);
    uint32_t magic = MAGIC_ID;
    memcpy(id->internal, &magic, 4);
    struct timespec t;
    clock_gettime(CLOCK_REALTIME, &t);
    snprintf(id->internal + 8, 56, "/rccl_standin_%d_%lx", (int)getpid(), (unsigned long)(t.tv_nsec ^ (t.tv_sec << 20)));
    /* a checksum over the whole 128 bytes: an id that was truncated or marshalled by pointer instead of by value does not pass */
    uint32_t sum = 0;
    for (int i = 0; i < 120; ++i) sum = sum * 31u + (unsigned char)id->internal[i];
    memcpy(id->internal + 120, &sum, 4);
    return ncclSuccess;
}
static int id_ok(const ncclUniqueId* id) {
    uint32_t magic, sum = 0, want;
    memcpy(&magic, id->internal, 4);
    memcpy(&want, id->internal + 120, 4);
    for (int i = 0; i < 120; ++i) sum = sum * 31u + (unsigned char)id->internal[i];
    return magic == MAGIC_ID && sum == want && id->internal[8] == '/';
}

int ncclCommInitRank(ncclComm_t* out, int nranks, ncclUniqueId id /* BY VALUE, as rccl.h declares it */, int rank) {
    if (!out || nranks < 1 || rank < 0 || rank >= nranks) return ncclInvalidArgument;
    if (fail_at("initrank")) return ncclSystemError;
    if (!id_ok(&id)) return ncclInvalidArgument;
    Comm* c = (Comm*)calloc(1, sizeof(Comm));
    c->magic = MAGIC_COMM; c->rank = rank; c->nranks = nranks;
    if (hipGetDevice(&c->dev) != hipSuccess) { free(c); return ncclUnhandledCudaError; }
    snprintf(c->name, sizeof c->name, "%.56s", id.internal + 8);
    int fd = shm_open(c->name, O_CREAT | O_RDWR, 0600);
    if (fd < 0 || ftruncate(fd, 4096) != 0) { if (fd >= 0) close(fd); free(c); return ncclSystemError; }
    c->sh = (Shared*)mmap(NULL, 4096, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
    close(fd);
    if (c->sh == MAP_FAILED) { free(c); return ncclSystemError; }
    __sync_fetch_and_add(&c->sh->arrived, 1u);
    if (!wait_until(&c->sh->arrived, (uint32_t)nranks)) {   /* real RCCL would block for ever here; the stand-in gives up and says so */
        munmap((void*)c->sh, 4096);
        if (rank == 0) shm_unlink(c->name);
        free(c);
        return ncclSystemError;
    }
    *out = c;
    return ncclSuccess;
}

int ncclCommInitAll(ncclComm_t* comms, int ndev, const int* devlist) {
    if (!comms || ndev < 1 || ndev > MAX_LOCAL) return ncclInvalidArgument;
    if (fail_at("initall")) return ncclSystemError;
    LocalGroup* lg = (LocalGroup*)calloc(1, sizeof(LocalGroup));
    lg->n = ndev; lg->alive = ndev;
    for (int i = 0; i < ndev; ++i) {
        Comm* c = (Comm*)calloc(1, sizeof(Comm));
        c->magic = MAGIC_COMM; c->rank = i; c->nranks = ndev; c->dev = devlist ? devlist[i] : i; c->lg = lg;
        lg->c[i] = c;
        comms[i] = c;
    }
    return ncclSuccess;
}

int ncclCommDestroy(ncclComm_t c) {
    if (!c || c->magic != MAGIC_COMM) return ncclInvalidArgument;
    c->magic = 0;
    if (c->sh) {
        const uint32_t left = __sync_add_and_fetch(&c->sh->left, 1u);
        const int last = left == (uint32_t)c->nranks;
        munmap((void*)c->sh, 4096);
        if (last) {
            char dname[80];
            snprintf(dname, sizeof dname, "%s_d", c->name);
            shm_unlink(dname);
            shm_unlink(c->name);
        }
    }
    if (c->lg && --c->lg->alive == 0) free(c->lg);
    free(c);
    return ncclSuccess;
}

/* ---- the broadcast ------------------------------------------------------------------------------------------------ */
static int bcast_multiprocess(const void* send, void* recv, size_t count, int root, Comm* c, hipStream_t stream) {
    char dname[80];
    snprintf(dname, sizeof dname, "%s_d", c->name);
    Shared* sh = c->sh;
    if (c->rank == root) {
        int fd = shm_open(dname, O_CREAT | O_RDWR, 0600);
        if (fd < 0 || ftruncate(fd, (off_t)count) != 0) { if (fd >= 0) close(fd); return ncclSystemError; }
        void* host = mmap(NULL, count, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
        close(fd);
        if (host == MAP_FAILED) return ncclSystemError;
        /* on the caller's stream: whatever the caller queued in front of the broadcast (the weight upload) comes first */
        hipError_t e = hipMemcpyAsync(host, send, count, hipMemcpyDeviceToHost, stream);
        if (e == hipSuccess) e = hipStreamSynchronize(stream);
        munmap(host, count);
        if (e != hipSuccess) return ncclUnhandledCudaError;
        if (recv != send && hipMemcpyAsync(recv, send, count, hipMemcpyDeviceToDevice, stream) != hipSuccess) return ncclUnhandledCudaError;
        sh->root = (uint32_t)root; sh->dtype = 1; sh->count = count;
        __sync_synchronize();
        __sync_fetch_and_add(&sh->bcast_seq, 1u);
        if (!wait_until(&sh->consumed, (uint32_t)(c->nranks - 1) * (c->seq_seen + 1))) return ncclSystemError;
        c->seq_seen++;
        return ncclSuccess;
    }
    if (!wait_until(&sh->bcast_seq, c->seq_seen + 1)) return ncclSystemError;
    c->seq_seen++;
    if (sh->root != (uint32_t)root || sh->count != count) { __sync_fetch_and_add(&sh->consumed, 1u); return ncclInvalidArgument; }   /* the ranks disagree */
    int fd = shm_open(dname, O_RDONLY, 0600);
    if (fd < 0) { __sync_fetch_and_add(&sh->consumed, 1u); return ncclSystemError; }
    void* host = mmap(NULL, count, PROT_READ, MAP_SHARED, fd, 0);
    close(fd);
    if (host == MAP_FAILED) { __sync_fetch_and_add(&sh->consumed, 1u); return ncclSystemError; }
    hipError_t e = hipMemcpyAsync(recv, host, count, hipMemcpyHostToDevice, stream);
    if (e == hipSuccess) e = hipStreamSynchronize(stream);   /* (the mapping goes away below: the copy must have read it) */
    munmap(host, count);
    __sync_fetch_and_add(&sh->consumed, 1u);
    return e == hipSuccess ? ncclSuccess : ncclUnhandledCudaError;
}

static int run_local(Pending* p, int np) {
    /* every communicator of the group must have posted its part, with one root and one count */
    if (np < 1) return ncclSuccess;
    LocalGroup* lg = p[0].comm->lg;
    if (np != lg->n) return ncclInvalidUsage;
    const Pending* rootp = NULL;
    for (int i = 0; i < np; ++i) {
        if (p[i].comm->lg != lg || p[i].root != p[0].root || p[i].count != p[0].count) return ncclInvalidArgument;
        if (p[i].comm->rank == p[i].root) rootp = &p[i];
    }
    if (!rootp) return ncclInvalidArgument;
    int dev0 = 0;
    hipGetDevice(&dev0);
    hipEvent_t ev;
    hipError_t e = hipSetDevice(rootp->comm->dev);
    if (e == hipSuccess) e = hipEventCreateWithFlags(&ev, hipEventDisableTiming);
    if (e != hipSuccess) return ncclUnhandledCudaError;
    e = hipEventRecord(ev, rootp->stream);   /* the root's buffer is ready when its stream reaches this point */
    for (int i = 0; i < np && e == hipSuccess; ++i) {
        if (&p[i] == rootp) {
            if (p[i].recv != p[i].send) e = hipMemcpyAsync(p[i].recv, p[i].send, p[i].count, hipMemcpyDeviceToDevice, p[i].stream);
            continue;
        }
        e = hipSetDevice(p[i].comm->dev);
        if (e == hipSuccess) e = hipStreamWaitEvent(p[i].stream, ev, 0);
        if (e == hipSuccess) e = hipMemcpyAsync(p[i].recv, rootp->send, p[i].count, hipMemcpyDeviceToDevice, p[i].stream);
    }
    hipSetDevice(rootp->comm->dev);
    hipEventDestroy(ev);
    hipSetDevice(dev0);
    return e == hipSuccess ? ncclSuccess : ncclUnhandledCudaError;
}

int ncclBroadcast(const void* send, void* recv, size_t count, int datatype, int root, ncclComm_t c, hipStream_t stream) {
    if (!c || c->magic != MAGIC_COMM) return ncclInvalidArgument;   /* destroyed or never initialised */
    if (!recv || (c->rank == root && !send) || root < 0 || root >= c->nranks) return ncclInvalidArgument;
    if (datatype != 1 /* ncclUint8 */) return ncclInvalidArgument;   /* the library sends bytes; anything else is a marshalling bug */
    if (fail_at("broadcast")) return ncclSystemError;
    if (c->sh) return bcast_multiprocess(send, recv, count, root, c, stream);
    if (g_npending >= MAX_LOCAL) return ncclInvalidUsage;
    g_pending[g_npending++] = (Pending){ send, recv, count, root, c, stream };
    if (g_group_depth == 0) {   /* one thread driving several communicators WITHOUT a group would deadlock in the real library */
        g_npending = 0;
        return c->nranks == 1 ? ncclSuccess : ncclInvalidUsage;
    }
    return ncclSuccess;
}

int ncclGroupStart(void) { ++g_group_depth; return ncclSuccess; }
int ncclGroupEnd(void) {
    if (g_group_depth < 1) return ncclInvalidUsage;
    if (--g_group_depth > 0) return ncclSuccess;
    const int np = g_npending;
    g_npending = 0;
    return run_local(g_pending, np);
}
