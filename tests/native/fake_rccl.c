/*
 * fake_rccl.c - TEST INFRASTRUCTURE: a stand-in for librccl that lets the library's native collective code
 * (ss_comm_init_rank / ss_search_sharded / ss_find_sharded, ss_comm_init_all / ss_search_sharded_all /
 * ss_find_sharded_all) run with 2, 3, 8 ... ranks ON ONE GPU.  Real RCCL refuses two ranks on one device, so on a
 * one-GPU box that C code had only ever executed with nranks == 1.
 *
 * It implements exactly the nine symbols the library resolves (csrc/ss_comm.hip: rccl()):
 *   ncclGetUniqueId  ncclCommInitRank  ncclCommInitAll  ncclCommDestroy  ncclCommCount
 *   ncclGroupStart   ncclGroupEnd      ncclAllReduce    ncclGetErrorString
 * with the semantics the callers rely on, nothing more:
 *   * one-process-per-rank communicators (ncclCommInitRank) meet in a POSIX shared-memory segment named by the
 *     unique id; ncclAllReduce = wait for `stream`, copy the send buffer to the host, put it in this rank's slot,
 *     barrier, reduce all slots element-wise (MAX / MIN of int32 / uint64), barrier, copy the result to the receive
 *     buffer.  The call returns with the result in place - real RCCL returns after ENQUEUEING; everything the library
 *     orders behind the all-reduce on the same stream sees the same values either way.
 *   * in-process communicators (ncclCommInitAll) are N peers of one heap segment; all-reduces issued inside
 *     ncclGroupStart/End (real RCCL's rule for ONE thread driving several peers) are carried out in ncclGroupEnd; issued
 *     outside a group - one THREAD per peer, real RCCL's other form - the peers meet at the segment's barrier like the
 *     ranks of separate processes.
 *   * FAKE_RCCL_SLOW_US=<us> (+ FAKE_RCCL_SLOW_EVERY=<k>, default 3): every k-th all-reduce of a thread leaves a host function
 *     that sleeps that long on the caller's stream behind its result - a collective kernel that connects lazily or waits for a
 *     late rank, as seen from the caller's bounded spin on what it enqueued behind the all-reduce.
 *   * every wait is bounded (FAKE_RCCL_TIMEOUT_S, default 120 s): a rank that never arrives gives the others
 *     ncclSystemError instead of a hang.
 * Built by sliceslice_rs_amd._build.build_fake_rccl() into tests/native/libfake_rccl.so and handed to the library
 * with SLICESLICE_RCCL_LIB=<path>.  Not part of the product; never loaded unless that variable names it.
 */
#define _GNU_SOURCE
#define __HIP_PLATFORM_AMD__ 1
#include <hip/hip_runtime_api.h>

#include <errno.h>
#include <fcntl.h>
#include <stdatomic.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <time.h>
#include <unistd.h>

enum { kOk = 0, kHipError = 1, kSystemError = 2, kInternalError = 3, kInvalidArgument = 4, kInvalidUsage = 5 };
enum { kInt32 = 2, kUint64 = 5 };       /* ncclDataType_t values the library passes */
enum { kMax = 2, kMin = 3 };            /* ncclRedOp_t */
enum { kMaxRanks = 64, kSlotBytes = 64 };

typedef struct {
    char internal[128];
} ncclUniqueId;

/* what the ranks of one communicator share */
typedef struct {
    _Atomic uint32_t attached;              /* ranks that have mapped the segment */
    _Atomic uint32_t detached;
    _Atomic uint32_t arrived;               /* barrier: ranks in the current round */
    _Atomic uint32_t generation;            /* barrier: completed rounds */
    _Atomic uint32_t broken;                /* a rank gave up waiting: everybody fails from here on */
    uint32_t nranks;
    unsigned char slot[kMaxRanks][kSlotBytes];
} Segment;

typedef struct Pending {
    const void *send;
    void *recv;
    size_t bytes;
    int dtype, op;
    hipStream_t stream;
} Pending;

typedef struct fake_comm {
    Segment *seg;
    int rank, nranks, dev;
    int in_process;                         /* ncclCommInitAll peer: operations are deferred to ncclGroupEnd */
    size_t map_bytes;
    char shm_name[128];
    int has_pending;
    Pending pending;
    struct fake_comm *next_in_group;
} fake_comm;

static _Thread_local int g_group_depth = 0;
static _Thread_local fake_comm *g_group_head = NULL;

static double timeout_s(void)
{
    const char *e = getenv("FAKE_RCCL_TIMEOUT_S");
    const double v = e ? atof(e) : 0.0;
    return v > 0 ? v : 120.0;
}

static double now_s(void)
{
    struct timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec;
}

static void relax(unsigned spins)
{
    if (spins < 2000) {
#if defined(__x86_64__)
        __builtin_ia32_pause();
#endif
    } else {
        struct timespec ts = {0, 50000};    /* 50 us: eight ranks may share fewer cores */
        nanosleep(&ts, NULL);
    }
}

/* generation barrier over the segment; 0 = everybody arrived */
static int barrier(Segment *s)
{
    if (atomic_load(&s->broken)) return kSystemError;
    const uint32_t gen = atomic_load(&s->generation);
    if (atomic_fetch_add(&s->arrived, 1) + 1 == s->nranks) {
        atomic_store(&s->arrived, 0);
        atomic_fetch_add(&s->generation, 1);
        return kOk;
    }
    const double t0 = now_s(), limit = timeout_s();
    for (unsigned spins = 0; atomic_load(&s->generation) == gen; ++spins) {
        if (atomic_load(&s->broken)) return kSystemError;
        relax(spins);
        if ((spins & 1023) == 1023 && now_s() - t0 > limit) {
            atomic_store(&s->broken, 1);
            return kSystemError;
        }
    }
    return kOk;
}

static size_t dtype_bytes(int dtype) { return dtype == kInt32 ? 4 : (dtype == kUint64 ? 8 : 0); }

static void reduce_into(unsigned char *acc, const unsigned char *v, size_t count, int dtype, int op)
{
    for (size_t k = 0; k < count; ++k) {
        if (dtype == kInt32) {
            int32_t a, b;
            memcpy(&a, acc + 4 * k, 4);
            memcpy(&b, v + 4 * k, 4);
            if (op == kMax ? b > a : b < a) memcpy(acc + 4 * k, &b, 4);
        } else {
            uint64_t a, b;
            memcpy(&a, acc + 8 * k, 8);
            memcpy(&b, v + 8 * k, 8);
            if (op == kMax ? b > a : b < a) memcpy(acc + 8 * k, &b, 8);
        }
    }
}

/* (what ss_comm_rccl_info reports for this library: a version no real RCCL has) */
int ncclGetVersion(int *version)
{
    if (!version) return 4;
    *version = 1;
    return 0;
}

const char *ncclGetErrorString(int code)
{
    switch (code) {
    case kOk: return "no error";
    case kHipError: return "fake rccl: unhandled HIP error";
    case kSystemError: return "fake rccl: a rank did not arrive in time (or the shared segment is unavailable)";
    case kInvalidArgument: return "fake rccl: invalid argument";
    case kInvalidUsage: return "fake rccl: invalid usage";
    default: return "fake rccl: internal error";
    }
}

int ncclGetUniqueId(ncclUniqueId *id)
{
    static _Atomic unsigned counter = 0;
    if (!id) return kInvalidArgument;
    memset(id, 0, sizeof *id);
    struct timespec ts;
    clock_gettime(CLOCK_REALTIME, &ts);
    snprintf(id->internal, sizeof id->internal, "/fake_rccl_%d_%u_%lx", (int)getpid(), atomic_fetch_add(&counter, 1),
             (unsigned long)ts.tv_nsec);
    /* the segment exists from here on, zero-filled: ranks attach in any order */
    const int fd = shm_open(id->internal, O_CREAT | O_EXCL | O_RDWR, 0600);
    if (fd < 0) return kSystemError;
    const int rc = ftruncate(fd, (off_t)sizeof(Segment));
    close(fd);
    return rc == 0 ? kOk : kSystemError;
}

int ncclCommInitRank(fake_comm **out, int nranks, ncclUniqueId id, int rank)
{
    if (!out || nranks < 1 || nranks > kMaxRanks || rank < 0 || rank >= nranks) return kInvalidArgument;
    id.internal[sizeof id.internal - 1] = 0;
    if (strncmp(id.internal, "/fake_rccl_", 11) != 0) return kInvalidArgument;
    const int fd = shm_open(id.internal, O_RDWR, 0600);
    if (fd < 0) return kSystemError;
    Segment *s = (Segment *)mmap(NULL, sizeof(Segment), PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
    close(fd);
    if (s == MAP_FAILED) return kSystemError;
    fake_comm *c = (fake_comm *)calloc(1, sizeof *c);
    if (!c) {
        munmap(s, sizeof(Segment));
        return kInternalError;
    }
    c->seg = s;
    c->rank = rank;
    c->nranks = nranks;
    c->map_bytes = sizeof(Segment);
    snprintf(c->shm_name, sizeof c->shm_name, "%s", id.internal);
    if (hipGetDevice(&c->dev) != hipSuccess) c->dev = 0;
    if (rank == 0) s->nranks = (uint32_t)nranks;
    atomic_fetch_add(&s->attached, 1);
    /* a collective, like the real one: returns when every rank has attached */
    const double t0 = now_s(), limit = timeout_s();
    for (unsigned spins = 0; atomic_load(&s->attached) < (uint32_t)nranks; ++spins) {
        relax(spins);
        if ((spins & 1023) == 1023 && now_s() - t0 > limit) {
            atomic_store(&s->broken, 1);
            munmap(s, sizeof(Segment));
            free(c);
            if (rank == 0) shm_unlink(id.internal);
            return kSystemError;
        }
    }
    /* (rank 0 wrote nranks before it attached; everybody is past that now) */
    if (rank == 0) shm_unlink(id.internal);     /* the mappings keep the segment alive; the name is gone */
    *out = c;
    return kOk;
}

int ncclCommInitAll(fake_comm **comms, int ndev, const int *devs)
{
    if (!comms || ndev < 1 || ndev > kMaxRanks) return kInvalidArgument;
    Segment *s = (Segment *)calloc(1, sizeof(Segment));
    if (!s) return kInternalError;
    s->nranks = (uint32_t)ndev;
    atomic_store(&s->attached, (uint32_t)ndev);
    for (int g = 0; g < ndev; ++g) {
        fake_comm *c = (fake_comm *)calloc(1, sizeof *c);
        if (!c) return kInternalError;
        c->seg = s;
        c->rank = g;
        c->nranks = ndev;
        c->dev = devs ? devs[g] : g;        /* the same device may be listed more than once: that is the point */
        c->in_process = 1;
        comms[g] = c;
    }
    return kOk;
}

int ncclCommDestroy(fake_comm *c)
{
    if (!c) return kInvalidArgument;
    Segment *s = c->seg;
    const uint32_t gone = atomic_fetch_add(&s->detached, 1) + 1;
    if (c->in_process) {
        if (gone == (uint32_t)c->nranks) free(s);
    } else {
        munmap(s, c->map_bytes);
    }
    free(c);
    return kOk;
}

int ncclCommCount(const fake_comm *c, int *count)
{
    if (!c || !count) return kInvalidArgument;
    *count = c->nranks;
    return kOk;
}

int ncclGroupStart(void)
{
    ++g_group_depth;
    return kOk;
}

/* stream -> host: everything enqueued on `stream` so far is complete, then the send buffer's bytes */
static int fetch(const Pending *p, int dev, unsigned char *dst)
{
    int cur = -1;
    (void)hipGetDevice(&cur);
    hipError_t e = hipSetDevice(dev);
    if (e == hipSuccess) e = hipStreamSynchronize(p->stream);
    if (e == hipSuccess) e = hipMemcpy(dst, p->send, p->bytes, hipMemcpyDeviceToHost);
    if (cur >= 0) (void)hipSetDevice(cur);
    return e == hipSuccess ? kOk : kHipError;
}

/* FAKE_RCCL_SLOW_US: every k-th delivery is followed, IN STREAM ORDER, by a host function that sleeps - what the caller has
 * enqueued behind the all-reduce (the library's answer-word kernel) waits for it like for a slow collective kernel, while the call
 * itself has long returned (as real RCCL's does). */
static void slow_host_fn(void *arg)
{
    const long us = (long)(intptr_t)arg;
    struct timespec ts = {us / 1000000, (us % 1000000) * 1000};
    nanosleep(&ts, NULL);
}

static void maybe_slow(hipStream_t stream)
{
    static _Thread_local unsigned calls = 0;
    static atomic_int slow_us = -1, every = 3;
    if (atomic_load(&slow_us) < 0) {
        const char *e = getenv("FAKE_RCCL_SLOW_US"), *k = getenv("FAKE_RCCL_SLOW_EVERY");
        atomic_store(&every, k && atoi(k) > 0 ? atoi(k) : 3);
        atomic_store(&slow_us, e ? atoi(e) : 0);
    }
    const int us = atomic_load(&slow_us);
    if (us > 0 && ++calls % (unsigned)atomic_load(&every) == 0) (void)hipLaunchHostFunc(stream, slow_host_fn, (void *)(intptr_t)us);
}

static int deliver(const Pending *p, int dev, const unsigned char *src)
{
    int cur = -1;
    (void)hipGetDevice(&cur);
    hipError_t e = hipSetDevice(dev);
    if (e == hipSuccess) e = hipMemcpy(p->recv, src, p->bytes, hipMemcpyHostToDevice);
    if (e == hipSuccess) maybe_slow(p->stream);
    if (cur >= 0) (void)hipSetDevice(cur);
    return e == hipSuccess ? kOk : kHipError;
}

int ncclGroupEnd(void)
{
    if (g_group_depth <= 0) return kInvalidUsage;
    if (--g_group_depth > 0) return kOk;
    fake_comm *head = g_group_head;
    g_group_head = NULL;
    if (!head) return kOk;
    /* the peers of ONE in-process communicator, each with one pending all-reduce of the same shape */
    Segment *s = head->seg;
    int n = 0, rc = kOk;
    for (fake_comm *c = head; c; c = c->next_in_group) ++n;
    if (n != head->nranks) rc = kInvalidUsage;                       /* a peer is missing: real RCCL would hang */
    unsigned char acc[kSlotBytes];
    for (fake_comm *c = head; c && rc == kOk; c = c->next_in_group) {
        if (c->seg != s || c->pending.bytes != head->pending.bytes || c->pending.dtype != head->pending.dtype ||
            c->pending.op != head->pending.op)
            rc = kInvalidUsage;
        else
            rc = fetch(&c->pending, c->dev, s->slot[c->rank]);
    }
    if (rc == kOk) {
        memcpy(acc, s->slot[head->rank], head->pending.bytes);
        for (fake_comm *c = head->next_in_group; c; c = c->next_in_group)
            reduce_into(acc, s->slot[c->rank], head->pending.bytes / dtype_bytes(head->pending.dtype), head->pending.dtype, head->pending.op);
        for (fake_comm *c = head; c && rc == kOk; c = c->next_in_group) rc = deliver(&c->pending, c->dev, acc);
    }
    for (fake_comm *c = head; c;) {
        fake_comm *nx = c->next_in_group;
        c->has_pending = 0;
        c->next_in_group = NULL;
        c = nx;
    }
    return rc;
}

int ncclAllReduce(const void *send, void *recv, size_t count, int dtype, int op, fake_comm *c, hipStream_t stream)
{
    const size_t eb = dtype_bytes(dtype);
    if (!c || !send || !recv || eb == 0 || (op != kMax && op != kMin) || count == 0 || count * eb > kSlotBytes) return kInvalidArgument;
    Pending p = {send, recv, count * eb, dtype, op, stream};
    if (c->in_process) {
        if (c->nranks == 1 && g_group_depth == 0) {
            unsigned char one[kSlotBytes];
            const int rc = fetch(&p, c->dev, one);
            return rc != kOk ? rc : deliver(&p, c->dev, one);
        }
        if (c->has_pending) return kInvalidUsage;
        if (g_group_depth == 0) goto meet;                                /* one thread per peer: the barrier below */
        c->pending = p;
        c->has_pending = 1;
        c->next_in_group = NULL;
        if (!g_group_head) {
            g_group_head = c;
        } else {
            fake_comm *t = g_group_head;
            while (t->next_in_group) t = t->next_in_group;
            t->next_in_group = c;
        }
        return kOk;
    }
meet:;
    Segment *s = c->seg;
    int rc = fetch(&p, c->dev, s->slot[c->rank]);
    if (rc != kOk) {
        atomic_store(&s->broken, 1);            /* the others must not wait for this rank */
        return rc;
    }
    if ((rc = barrier(s)) != kOk) return rc;    /* every slot is written */
    unsigned char acc[kSlotBytes];
    memcpy(acc, s->slot[0], p.bytes);
    for (int r = 1; r < c->nranks; ++r) reduce_into(acc, s->slot[r], count, dtype, op);
    if ((rc = barrier(s)) != kOk) return rc;    /* every rank has read: the slots are free for the next call */
    return deliver(&p, c->dev, acc);
}
