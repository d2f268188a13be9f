// ss_host.hip - haystacks that start in HOST memory, and the byte histogram.
//   ss_search_host / ss_find_host   search_in(&[u8]) for a host slice: chunked upload, n-1 bytes of carry          (x86.rs:523)
//   ss_search_file                  row f2 of SURVEY.md 8f: the front end of /root/reference/examples/grep.rs:42-56
//   ss_byte_histogram_device        row f3: data for a `position` policy (the reference leaves it to the caller, x86.rs:252-255)
// PCIe- or file-bound by construction; never used for roofline numbers.  The scans are the same kernels (enqueue_scan, ss_scan.hip).
#include "ss_internal.hpp"

#include <fcntl.h>
#include <sys/stat.h>
#include <unistd.h>

#define SS_AUX_HISTOGRAM 1
#include "aux_kernels.hpp"

namespace ssh {

// ---- staging sets of the host-buffer and file front ends ---------------------------------------------------
// Device buffers + streams (+ pinned host buffers for the file reader).  Creating them per call costs ~0.3 ms
// (hipMalloc, stream create/destroy) and pinning 3 x 64 MiB ~10 ms - more than uploading and scanning a small
// haystack - so one set per device is kept for the life of the process and lent to one call at a time; a
// concurrent call builds a private set.
namespace {

constexpr int kStageBuf = 3;
struct Staging {
    uint8_t *h[kStageBuf] = {nullptr, nullptr, nullptr};
    uint8_t *d[kStageBuf] = {nullptr, nullptr, nullptr};
    hipStream_t st[kStageBuf] = {nullptr, nullptr, nullptr};
    size_t cap_d = 0, cap_h = 0;      // bytes per device / pinned buffer
    int nbuf = 0;
    void release()
    {
        for (int b = 0; b < kStageBuf; ++b) {
            if (st[b]) { (void)hipStreamSynchronize(st[b]); (void)hipStreamDestroy(st[b]); }
            if (d[b]) (void)hipFree(d[b]);
            if (h[b]) (void)hipHostFree(h[b]);
            st[b] = nullptr; d[b] = nullptr; h[b] = nullptr;
        }
        cap_d = cap_h = 0;
        nbuf = 0;
    }
    bool ensure(int want_nbuf, size_t want_cap, bool pinned)     // grow-only
    {
        if (want_cap < ((size_t)1 << 20)) want_cap = (size_t)1 << 20;      // do not regrow for every small call
        if (nbuf >= want_nbuf && cap_d >= want_cap && (!pinned || cap_h >= want_cap)) return true;
        if (want_cap < cap_d) want_cap = cap_d;
        if (want_nbuf < nbuf) want_nbuf = nbuf;
        const bool want_pinned = pinned || cap_h > 0;
        release();
        for (int b = 0; b < want_nbuf; ++b) {
            if ((want_pinned && hipHostMalloc((void **)&h[b], want_cap, hipHostMallocDefault) != hipSuccess) ||
                hipMalloc((void **)&d[b], want_cap) != hipSuccess ||
                hipStreamCreateWithFlags(&st[b], hipStreamNonBlocking) != hipSuccess) {
                release();
                return false;
            }
        }
        cap_d = want_cap;
        cap_h = want_pinned ? want_cap : 0;
        nbuf = want_nbuf;
        return true;
    }
};
std::mutex g_staging_mu[kMaxDevices];
Staging g_staging[kMaxDevices];

// Lends the device's cached set when it is free, `mine` otherwise; `mine` is released by its destructor-like
// call site (Lease::done).
struct Lease {
    Staging mine, *set = &mine;
    std::unique_lock<std::mutex> lock;
    explicit Lease(int dev)
    {
        if (dev >= 0 && dev < kMaxDevices) {
            lock = std::unique_lock<std::mutex>(g_staging_mu[dev], std::try_to_lock);
            if (lock.owns_lock()) set = &g_staging[dev];
        }
    }
    ~Lease()
    {
        if (set == &mine) {
            mine.release();
        } else {
            for (int b = 0; b < set->nbuf; ++b) (void)hipStreamSynchronize(set->st[b]);   // nothing of this call in flight
        }
    }
};

// Small host slices skip the upload command altogether: the bytes are copied (by the CPU) into a pinned, device-visible
// buffer that belongs to the calling thread, and the scan reads them straight over PCIe - one launch, one completion
// word, no hipMemcpyAsync (a copy command costs ~6 us whatever its size; 64 KiB over PCIe cost ~1 us).
constexpr size_t kZeroCopyMax = 64u << 10;
struct ThreadPinned {
    uint8_t *p = nullptr;
    hipStream_t st[kMaxDevices] = {nullptr};      // one non-blocking stream per device this thread has searched on
    ~ThreadPinned()
    {
        if (process_exiting()) return;                  // leak: see ExitMark
        if (p) (void)hipHostFree(p);
        for (hipStream_t q : st)
            if (q) (void)hipStreamDestroy(q);
    }
};
thread_local ThreadPinned g_small_host;

// the calling thread's pinned copy of a small host slice and its stream on the current device, or nullptr (too large,
// switched off, no pinned memory / stream)
const uint8_t *small_host_copy(const uint8_t *haystack, size_t len, hipStream_t *stream)
{
    if (len > kZeroCopyMax) return nullptr;
    ThreadPinned &tp = g_small_host;
    int dev = -1;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= kMaxDevices) return nullptr;
    if (!tp.st[dev] && hipStreamCreateWithFlags(&tp.st[dev], hipStreamNonBlocking) != hipSuccess) tp.st[dev] = nullptr;
    if (!tp.p && hipHostMalloc((void **)&tp.p, kZeroCopyMax + 64, hipHostMallocPortable | hipHostMallocMapped) != hipSuccess) tp.p = nullptr;
    if (!tp.p || !tp.st[dev]) return nullptr;
    memcpy(tp.p, haystack, len);
    *stream = tp.st[dev];
    return tp.p;
}

}  // namespace

namespace {

bool parallel_pread(int fd, uint8_t *dst, size_t bytes, off_t off, unsigned threads)
{
    if (threads < 1) threads = 1;
    const size_t part = (bytes + threads - 1) / threads;
    std::vector<std::thread> pool;
    std::atomic<bool> ok{true};
    for (unsigned t = 0; t < threads; ++t) {
        const size_t b = (size_t)t * part;
        if (b >= bytes) break;
        const size_t e = b + part < bytes ? b + part : bytes;
        pool.emplace_back([=, &ok]() {
            size_t done = b;
            while (done < e) {
                const ssize_t r = pread(fd, dst + done, e - done, off + (off_t)done);
                if (r <= 0) {
                    ok = false;
                    return;
                }
                done += (size_t)r;
            }
        });
    }
    for (auto &th : pool) th.join();
    return ok;
}

}  // namespace

}  // namespace ssh

using namespace ssh;

extern "C" {

int ss_search_host(const ss_searcher *s, const uint8_t *haystack, size_t len, int *found)
{
    if (!s || !found) return fail(SS_ERR_ARGUMENT, "NULL argument");
    if (len && !haystack) return fail(SS_ERR_ARGUMENT, "haystack is NULL");
    SearchGate gate(s);                                  // set_filter* are refused while this call runs
    if (s->n == 0) { *found = 1; return SS_OK; }
    if (len < s->n) { *found = 0; return SS_OK; }
    hipStream_t small_st = nullptr;
    if (const uint8_t *pinned = small_host_copy(haystack, len, &small_st))
        return ss_search_device(s, pinned, len, small_st, found);           // the call waits for its own kernel: the buffer is free again
    // Chunked staging: chunk k covers haystack bytes [k*C - carry, (k+1)*C) with carry = n-1, so a
    // match straddling a chunk edge is seen by the later chunk.  Two device buffers / two streams:
    // the upload of chunk k+1 overlaps the scan of chunk k.
    const size_t carry = s->n - 1;
    size_t C = (size_t)64 << 20;
    if (C < 4 * s->n) C = 4 * s->n;
    if (C > len) C = len;
    PerDevice *pd = nullptr;
    if (int rc = get_per_device(s, &pd)) return rc;
    const size_t nbuf = len > C ? 2 : 1;
    Lease lease(pd->dev);
    if (!lease.set->ensure((int)nbuf, C + carry, false)) return fail(SS_ERR_HIP, "staging allocation failed");
    uint8_t **dbuf = lease.set->d;
    hipStream_t *st = lease.set->st;
    const int k = acquire_slot(s, pd);
    const int epoch = next_epoch(pd, k);                 // "found" value of this call (see ss_search_device)
    int rc = SS_OK;
    int result = 0;
    size_t idx = 0;
    for (size_t off = 0; off < len && rc == SS_OK && !result; off += C, ++idx) {
        const int b = (int)(idx % nbuf);
        const size_t lead = off == 0 ? 0 : carry;
        const size_t bytes = (len - off < C ? len - off : C) + lead;
        if (bytes < s->n) break;                         // tail shorter than the needle: nothing new can start here
        hipError_t e = hipStreamSynchronize(st[b]);      // buffer b free again
        if (e == hipSuccess && idx >= nbuf &&          // result of the scan that last used this buffer
            __atomic_load_n(pd->h_flags + k, __ATOMIC_ACQUIRE) == epoch) {
            result = 1;
            break;
        }
        if (e == hipSuccess) e = hipMemcpyAsync(dbuf[b], haystack + off - lead, bytes, hipMemcpyHostToDevice, st[b]);
        if (e != hipSuccess) { rc = fail(SS_ERR_HIP, "upload: %s", hipGetErrorString(e)); break; }
        rc = enqueue_scan(s, pd, dbuf[b], bytes, st[b], pd->d_flags + k, false, 0, pd->h_flags + k, epoch);
    }
    for (size_t b = 0; b < nbuf; ++b)
        if (st[b]) (void)hipStreamSynchronize(st[b]);
    if (rc == SS_OK && !result) result = __atomic_load_n(pd->h_flags + k, __ATOMIC_ACQUIRE) == epoch;
    release_slot(s, pd, k);
    if (rc == SS_OK) *found = result;
    return rc;
}

// find() for a host haystack: the chunked upload of ss_search_host with the uint64 best-offset sink.
// Chunks are issued left to right, so once a finished chunk has reported a match no later chunk can
// improve on it: stop issuing, drain the (at most one) chunk still in flight, read the minimum.
int ss_find_host(const ss_searcher *s, const uint8_t *haystack, size_t len, uint64_t *position)
{
    if (!s || !position) return fail(SS_ERR_ARGUMENT, "NULL argument");
    if (len && !haystack) return fail(SS_ERR_ARGUMENT, "haystack is NULL");
    SearchGate gate(s);                                  // set_filter* are refused while this call runs
    if (s->n == 0) { *position = 0; return SS_OK; }
    if (len < s->n) { *position = SS_NPOS; return SS_OK; }
    hipStream_t small_st = nullptr;
    if (const uint8_t *pinned = small_host_copy(haystack, len, &small_st)) return ss_find_device(s, pinned, len, small_st, position);
    const size_t carry = s->n - 1;
    size_t C = (size_t)64 << 20;
    if (C < 4 * s->n) C = 4 * s->n;
    if (C > len) C = len;
    PerDevice *pd = nullptr;
    if (int rc = get_per_device(s, &pd)) return rc;
    const size_t nbuf = len > C ? 2 : 1;
    Lease lease(pd->dev);
    if (!lease.set->ensure((int)nbuf, C + carry, false)) return fail(SS_ERR_HIP, "staging allocation failed");
    uint8_t **dbuf = lease.set->d;
    hipStream_t *st = lease.set->st;
    const int k = acquire_slot(s, pd);
    int rc = SS_OK;
    size_t idx = 0;
    bool hit = false;
    for (size_t off = 0; off < len && rc == SS_OK && !hit; off += C, ++idx) {
        const int b = (int)(idx % nbuf);
        const size_t lead = off == 0 ? 0 : carry;
        const size_t bytes = (len - off < C ? len - off : C) + lead;
        if (bytes < s->n) break;
        hipError_t e = hipStreamSynchronize(st[b]);
        if (e == hipSuccess && idx >= nbuf) {
            e = hipMemcpy(pd->h_best + k, pd->d_best + k, sizeof(uint64_t), hipMemcpyDeviceToHost);
            if (e == hipSuccess && pd->h_best[k] != SS_NPOS) { hit = true; break; }
        }
        if (e == hipSuccess) e = hipMemcpyAsync(dbuf[b], haystack + off - lead, bytes, hipMemcpyHostToDevice, st[b]);
        if (e != hipSuccess) { rc = fail(SS_ERR_HIP, "upload: %s", hipGetErrorString(e)); break; }
        rc = enqueue_scan(s, pd, dbuf[b], bytes, st[b], pd->d_best + k, true, (uint64_t)(off - lead));
    }
    for (size_t b = 0; b < nbuf; ++b)
        if (st[b]) (void)hipStreamSynchronize(st[b]);
    if (rc == SS_OK) {
        const hipError_t e = hipMemcpy(pd->h_best + k, pd->d_best + k, sizeof(uint64_t), hipMemcpyDeviceToHost);
        if (e != hipSuccess) rc = fail(SS_ERR_HIP, "position read-back: %s", hipGetErrorString(e));
        else *position = pd->h_best[k];
    }
    (void)hipMemset(pd->d_best + k, 0xFF, sizeof(uint64_t));       // slots are all-ones whenever they are free
    release_slot(s, pd, k);
    return rc;
}

// ---- row f2: host-file front end (the shape of examples/grep.rs:42-56: open the file, one search_in) ----
// A three-stage pipeline: reader threads pread() the next chunk into a pinned buffer while the previous
// chunks are in flight as hipMemcpyAsync + scan on their own streams.  Chunk k carries the last n-1 bytes
// of chunk k-1 in front, so a match that straddles a chunk edge is seen by the later chunk.
int ss_search_file(const ss_searcher *s, const char *path, int *found)
{
    if (!s || !path || !found) return fail(SS_ERR_ARGUMENT, "NULL argument");
    SearchGate gate(s);                                  // set_filter* are refused while this call runs
    const int fd = open(path, O_RDONLY);
    if (fd < 0) return fail(SS_ERR_ARGUMENT, "cannot open %s", path);
    struct stat sb;
    if (fstat(fd, &sb) != 0) {
        close(fd);
        return fail(SS_ERR_ARGUMENT, "cannot stat %s", path);
    }
    const size_t len = (size_t)sb.st_size;
    if (s->n == 0 || len < s->n) {                       // answered without reading the file (x86.rs:500, 357-359)
        close(fd);
        *found = s->n == 0;
        return SS_OK;
    }
    const size_t carry = s->n - 1;
    // chunk: 64 MiB for large files, an eighth of the file (>= 8 MiB) for small ones so that reading, upload
    // and scan of a few-hundred-MiB file still overlap
    size_t C = len / 8;
    if (C > ((size_t)64 << 20)) C = (size_t)64 << 20;
    if (C < ((size_t)8 << 20)) C = (size_t)8 << 20;
    if (C < 4 * s->n) C = 4 * s->n;
    if (C > len) C = len;
    const int nbuf = len > C ? kStageBuf : 1;
    unsigned threads = std::thread::hardware_concurrency();
    if (threads > 8) threads = 8;
    if (len < ((size_t)8 << 20)) threads = 1;

    PerDevice *pd = nullptr;
    if (int rc = get_per_device(s, &pd)) {
        close(fd);
        return rc;
    }
    // the device's cached staging set when it is free, a private one otherwise
    Lease lease(pd->dev);
    Staging *fs = lease.set;
    if (!fs->ensure(nbuf, C + carry, true)) {
        close(fd);
        return fail(SS_ERR_HIP, "staging allocation failed");
    }
    uint8_t **hbuf = fs->h, **dbuf = fs->d;
    hipStream_t *st = fs->st;
    const int k = acquire_slot(s, pd);
    const int epoch = next_epoch(pd, k);
    int rc = SS_OK;
    int result = 0;
    size_t idx = 0, prev_total = 0;
    int prev_b = -1;
    for (size_t off = 0; off < len && rc == SS_OK && !result; off += C, ++idx) {
        const int b = (int)(idx % (size_t)nbuf);
        const size_t lead = off == 0 ? 0 : carry;
        const size_t fresh = len - off < C ? len - off : C;
        hipError_t e = hipStreamSynchronize(st[b]);      // the copy + scan that last used buffer b are done
        if (e != hipSuccess) { rc = fail(SS_ERR_HIP, "stream wait: %s", hipGetErrorString(e)); break; }
        if (__atomic_load_n(pd->h_flags + k, __ATOMIC_ACQUIRE) == epoch) { result = 1; break; }
        if (lead) memcpy(hbuf[b], hbuf[prev_b] + prev_total - carry, carry);
        if (!parallel_pread(fd, hbuf[b] + lead, fresh, (off_t)off, threads)) {
            rc = fail(SS_ERR_ARGUMENT, "read error on %s", path);
            break;
        }
        const size_t total = lead + fresh;
        prev_b = b;
        prev_total = total;
        if (total < s->n) break;                         // tail shorter than the needle: nothing new can start here
        e = hipMemcpyAsync(dbuf[b], hbuf[b], total, hipMemcpyHostToDevice, st[b]);
        if (e != hipSuccess) { rc = fail(SS_ERR_HIP, "upload: %s", hipGetErrorString(e)); break; }
        rc = enqueue_scan(s, pd, dbuf[b], total, st[b], pd->d_flags + k, false, 0, pd->h_flags + k, epoch);
    }
    for (int b = 0; b < fs->nbuf; ++b)
        if (st[b]) (void)hipStreamSynchronize(st[b]);
    if (rc == SS_OK && !result) result = __atomic_load_n(pd->h_flags + k, __ATOMIC_ACQUIRE) == epoch;
    close(fd);
    release_slot(s, pd, k);
    if (rc == SS_OK) *found = result;
    return rc;
}

// ---- row f3: data for a `position` policy ---------------------------------------------------------------
int ss_byte_histogram_device(const void *d_haystack, size_t len, size_t sample_bytes, void *hip_stream,
                             uint64_t hist[256])
{
    if (!hist) return fail(SS_ERR_ARGUMENT, "hist is NULL");
    memset(hist, 0, 256 * sizeof(uint64_t));
    if (len < 16) return SS_OK;
    if (!d_haystack) return fail(SS_ERR_ARGUMENT, "haystack is NULL");
    hipStream_t st = static_cast<hipStream_t>(hip_stream);
    int dev = 0;
    HIP_TRY(hipGetDevice(&dev));
    DeviceInfo di;
    if (int rc = device_info(dev, &di)) return rc;
    unsigned long long *d_hist = nullptr;
    HIP_TRY(hipMalloc((void **)&d_hist, 256 * sizeof(unsigned long long)));
    hipError_t e = hipMemsetAsync(d_hist, 0, 256 * sizeof(unsigned long long), st);
    uint64_t stride = 1;
    if (sample_bytes && sample_bytes < len) stride = (len + sample_bytes - 1) / sample_bytes;
    const uint64_t work = len / 16 / stride;
    uint64_t blocks = (work + ss::kBlock - 1) / ss::kBlock;
    if (blocks > (uint64_t)di.cus * 8) blocks = (uint64_t)di.cus * 8;
    if (blocks < 1) blocks = 1;
    if (e == hipSuccess) {
        ss::byte_histogram_kernel<<<dim3((unsigned)blocks), dim3(ss::kBlock), 0, st>>>(
            static_cast<const uint8_t *>(d_haystack), len, stride, d_hist);
        e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipMemcpyAsync(hist, d_hist, 256 * sizeof(uint64_t), hipMemcpyDeviceToHost, st);
    if (e == hipSuccess) e = hipStreamSynchronize(st);
    (void)hipFree(d_hist);
    if (e != hipSuccess) return fail(SS_ERR_HIP, "histogram: %s", hipGetErrorString(e));
    return SS_OK;
}

}  // extern "C"
