/*
 * sliceslice_oracle.c - CPU restatement of the reference's substring-search hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under sliceslice-rs_amd/ may include, link,
 * load or call this file.  The only permitted users are tests/, the
 * __graft_entry__.smoke() checker and bench.py's `cpu_baseline` leg.
 *
 * What it restates (reference = cloudflare/sliceslice-rs @ 2024_08_07, v0.4.3;
 * all citations are paths under /root/reference):
 *
 *   oracle_searcher_init      src/x86.rs:454-459   DynamicAvx2Searcher::new (position = len.wrapping_sub(1))
 *                             src/x86.rs:468-493   ::with_position  (N0 / N1 / N2..N16 / N dispatch, asserts)
 *                             src/x86.rs:297-305   Avx2Searcher::with_position  (assert position < size)
 *   oracle_search_in          src/x86.rs:498-519   DynamicAvx2Searcher::inlined_search_in (N0 -> true)
 *                             src/lib.rs:130-136   MemchrSearcher::inlined_search_in (empty -> false; memchr)
 *                             src/x86.rs:356-376   Avx2Searcher::inlined_search_in (len<=n rule, width ladder)
 *   vector_search_in          src/lib.rs:253-287   chunks_exact(LANES) + one overlapped masked tail chunk
 *   vector_search_in_chunk    src/lib.rs:199-251   two loads, two byte-compares, AND, movemask, & mask,
 *                                                  then LSB-first memcmp of needle[1..] with early exit
 *   chunk_mask_{2,4,8,16,32}  src/x86.rs:26-235    the five `Vector` impls (__m16i/__m32i/__m64i broadcast a
 *                                                  scalar unaligned read and mask the movemask to 0x3/0xF/0xFF)
 *   oracle_naive              src/lib.rs:371-373, tests/i386.rs:6-10   haystack.windows(n).any(|w| w == needle)
 *
 * Third-party arithmetic not under /root/reference: the `memchr` crate
 * (Cargo.toml:15, semver "2.3", no lockfile => exact version unpinned), used at
 * src/lib.rs:135 for one-byte needles.  Its published contract is "index of the
 * first occurrence of a byte, or None"; libc memchr() has the same contract and
 * is used here.  Only `.is_some()` reaches the caller.
 *
 * Parity status: PINNED.  The Rust reference cannot be built in this image (no
 * rustc/cargo, no vendored deps, no network), so there is no oracle/_ref.  The
 * restatement is instead checked (tests/test_oracle.py) against every known-answer
 * vector the reference's own tests hold for this path - tests/golden/kat.json
 * (src/lib.rs:303-331, 422-544 for every `position`), the panic contract
 * (src/x86.rs:533-543) and the two corpus sweeps of tests/i386.rs:46-70 with
 * their hit counts (39,105 of 10,513,405; 4,585 of 4,585) - and against
 * oracle_naive, which is the reference tests' own oracle.
 *
 * Build: see oracle/Makefile (gcc -O2; the 32-lane loop is compiled with
 * __attribute__((target("avx2"))) as a whole, the way the reference's
 * `multiversion!` stamps the whole loop with #[target_feature(enable = "avx2")],
 * src/multiversion.rs:1-40).  A scalar mask builder with the same semantics is
 * always compiled and is used at run time when the host CPU lacks AVX2; tests
 * check that both builders agree.
 */
#ifndef _GNU_SOURCE
#define _GNU_SOURCE            /* sched_getaffinity / pthread_setaffinity_np for the multi-threaded baseline */
#endif
#include <stddef.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <pthread.h>
#include <sched.h>

#if defined(__x86_64__)
#include <immintrin.h>
#define ORACLE_X86 1
#define TARGET_AVX2 __attribute__((target("avx2")))
#endif

#define ORACLE_OK 0
#define ORACLE_EPOSITION 1 /* the reference would panic: position out of range */

enum { KIND_N0 = 0, KIND_N1 = 1, KIND_FIXED = 2 /* N2..N16 */, KIND_DYN = 3 /* N */ };

typedef struct oracle_searcher {
    int kind;
    size_t n;
    size_t position;
    uint8_t first, last;
    uint8_t *needle;      /* owned copy (x86.rs:476-490 copies into [u8;k]; N keeps the caller's N) */
    int force_scalar;     /* test hook: use the scalar mask builders */
} oracle_searcher;

static int have_avx2(void)
{
#if defined(ORACLE_X86)
    static int cached = -1;
    if (cached < 0) cached = __builtin_cpu_supports("avx2") ? 1 : 0;
    return cached;
#else
    return 0;
#endif
}

/* ---- construction: x86.rs:454-493 --------------------------------------------------------- */

int oracle_searcher_init(oracle_searcher **out, const uint8_t *needle, size_t n, size_t position)
{
    *out = NULL;
    if (n == 1 && position != 0) return ORACLE_EPOSITION;      /* assert_eq!(position, 0)  x86.rs:473 */
    if (n >= 2 && !(position < n)) return ORACLE_EPOSITION;    /* assert!(position < size)  x86.rs:300 */
    oracle_searcher *s = (oracle_searcher *)calloc(1, sizeof *s);
    if (!s) return -1;
    s->n = n;
    s->position = position;
    s->kind = n == 0 ? KIND_N0 : n == 1 ? KIND_N1 : n <= 16 ? KIND_FIXED : KIND_DYN;
    s->needle = (uint8_t *)malloc(n ? n : 1);
    if (n) memcpy(s->needle, needle, n);
    if (n) {
        s->first = needle[0];
        s->last = n >= 2 ? needle[position] : needle[0];
    }
    *out = s;
    return ORACLE_OK;
}

/* DynamicAvx2Searcher::new - x86.rs:454-459: position = len.wrapping_sub(1) */
int oracle_searcher_init_default(oracle_searcher **out, const uint8_t *needle, size_t n)
{
    return oracle_searcher_init(out, needle, n, n - 1 /* wraps for n == 0, ignored by N0 */);
}

void oracle_searcher_free(oracle_searcher *s)
{
    if (!s) return;
    free(s->needle);
    free(s);
}

void oracle_searcher_force_scalar(oracle_searcher *s, int on) { s->force_scalar = on; }

/* ---- the five Vector impls: x86.rs:26-235 -------------------------------------------------- */

/* scalar statement of: to_bitmask(lanes_eq(splat(first), load(start)) & lanes_eq(splat(last), load(start+pos))) */
static inline uint32_t chunk_mask_scalar(const uint8_t *start, size_t position, uint8_t first,
                                         uint8_t last, int lanes)
{
    uint32_t m = 0;
    for (int i = 0; i < lanes; ++i)
        if (start[i] == first && start[position + i] == last) m |= 1u << i;
    return m;
}

#if defined(ORACLE_X86)
TARGET_AVX2 static inline uint32_t chunk_mask_32(const uint8_t *start, size_t position,
                                                                     uint8_t first, uint8_t last)
{
    const __m256i a = _mm256_loadu_si256((const __m256i *)start);
    const __m256i b = _mm256_loadu_si256((const __m256i *)(start + position));
    const __m256i ea = _mm256_cmpeq_epi8(_mm256_set1_epi8((char)first), a);
    const __m256i eb = _mm256_cmpeq_epi8(_mm256_set1_epi8((char)last), b);
    return (uint32_t)_mm256_movemask_epi8(_mm256_and_si256(ea, eb));
}

static inline uint32_t sse_mask(__m128i a, __m128i b, uint8_t first, uint8_t last)
{
    const __m128i ea = _mm_cmpeq_epi8(_mm_set1_epi8((char)first), a);
    const __m128i eb = _mm_cmpeq_epi8(_mm_set1_epi8((char)last), b);
    return (uint32_t)_mm_movemask_epi8(_mm_and_si128(ea, eb));
}

static inline uint32_t chunk_mask_16(const uint8_t *start, size_t position, uint8_t first, uint8_t last)
{
    return sse_mask(_mm_loadu_si128((const __m128i *)start),
                    _mm_loadu_si128((const __m128i *)(start + position)), first, last);
}

/* __m64i / __m32i / __m16i: a scalar unaligned read broadcast into an XMM register,
 * movemask masked to the low 8 / 4 / 2 lanes (x86.rs:120-165, 73-118, 26-71). */
static inline uint32_t chunk_mask_8(const uint8_t *start, size_t position, uint8_t first, uint8_t last)
{
    int64_t a, b;
    memcpy(&a, start, 8);
    memcpy(&b, start + position, 8);
    return sse_mask(_mm_set1_epi64x(a), _mm_set1_epi64x(b), first, last) & 0xFF;
}

static inline uint32_t chunk_mask_4(const uint8_t *start, size_t position, uint8_t first, uint8_t last)
{
    int32_t a, b;
    memcpy(&a, start, 4);
    memcpy(&b, start + position, 4);
    return sse_mask(_mm_set1_epi32(a), _mm_set1_epi32(b), first, last) & 0xF;
}

static inline uint32_t chunk_mask_2(const uint8_t *start, size_t position, uint8_t first, uint8_t last)
{
    int16_t a, b;
    memcpy(&a, start, 2);
    memcpy(&b, start + position, 2);
    return sse_mask(_mm_set1_epi16(a), _mm_set1_epi16(b), first, last) & 0x3;
}
#endif

/* ---- vector_search_in_chunk (lib.rs:199-251) + vector_search_in (lib.rs:253-287) ------------ */

/* One stamped copy per (ISA, LANES), the way `multiversion!` + `dispatch!` stamp the reference's
 * loop (src/multiversion.rs:1-56) so that the mask builder inlines into the loop.
 *
 * chunk:  eq = to_bitmask(eq_first & eq_last) & mask; then, lowest set bit first, compare
 *         hay[start+1+off ..][..n-1] with needle[1..]; first equal -> true (early exit).
 *         The reference selects a const-length compare for SIZE = Some(1..=16) (lib.rs:222-241) and a
 *         run-time length otherwise; both are byte equality over n-1 bytes, so memcmp states both.
 * outer:  haystack[..end].chunks_exact(LANES) with mask u32::MAX, then if a remainder exists one
 *         overlapped chunk at end-LANES with mask u32::MAX << (LANES - remainder).                 */
#define DEFINE_VECTOR_SEARCH_IN(NAME, ATTR, LANES, MASK_EXPR)                                        \
    ATTR static int NAME(const oracle_searcher *s, const uint8_t *hay, size_t end)                   \
    {                                                                                                \
        const size_t position = s->position;                                                         \
        const uint8_t first = s->first, last = s->last;                                              \
        const size_t size = s->n - 1;                                                                \
        const uint8_t *needle = s->needle + 1;                                                       \
        (void)position; (void)first; (void)last;                                                     \
        size_t i = 0;                                                                                \
        uint32_t mask = UINT32_MAX;                                                                  \
        const uint8_t *start;                                                                        \
        for (;;) {                                                                                   \
            if (i + (LANES) <= end) {                                                                \
                start = hay + i;                                                                     \
                i += (LANES);                                                                        \
            } else {                                                                                 \
                const size_t remainder = end - i;                                                    \
                if (remainder == 0 || mask != UINT32_MAX) return 0;                                  \
                start = hay + end - (LANES);                                                         \
                mask = UINT32_MAX << ((LANES) - remainder);                                          \
            }                                                                                        \
            uint32_t eq = (MASK_EXPR) & mask;                                                        \
            while (eq != 0) {                                                                        \
                const uint8_t *c = start + 1 + __builtin_ctz(eq);                                    \
                if (memcmp(c, needle, size) == 0) return 1;                                          \
                eq &= eq - 1; /* clear lowest set bit */                                             \
            }                                                                                        \
            if (mask != UINT32_MAX) return 0;                                                        \
        }                                                                                            \
    }

DEFINE_VECTOR_SEARCH_IN(vsi_scalar_2, , 2, chunk_mask_scalar(start, position, first, last, 2))
DEFINE_VECTOR_SEARCH_IN(vsi_scalar_4, , 4, chunk_mask_scalar(start, position, first, last, 4))
DEFINE_VECTOR_SEARCH_IN(vsi_scalar_8, , 8, chunk_mask_scalar(start, position, first, last, 8))
DEFINE_VECTOR_SEARCH_IN(vsi_scalar_16, , 16, chunk_mask_scalar(start, position, first, last, 16))
DEFINE_VECTOR_SEARCH_IN(vsi_scalar_32, , 32, chunk_mask_scalar(start, position, first, last, 32))
#if defined(ORACLE_X86)
DEFINE_VECTOR_SEARCH_IN(vsi_sse2_2, , 2, chunk_mask_2(start, position, first, last))
DEFINE_VECTOR_SEARCH_IN(vsi_sse2_4, , 4, chunk_mask_4(start, position, first, last))
DEFINE_VECTOR_SEARCH_IN(vsi_sse2_8, , 8, chunk_mask_8(start, position, first, last))
DEFINE_VECTOR_SEARCH_IN(vsi_sse2_16, , 16, chunk_mask_16(start, position, first, last))
DEFINE_VECTOR_SEARCH_IN(vsi_avx2_32, TARGET_AVX2, 32, chunk_mask_32(start, position, first, last))
#endif

static int vector_search_in(const oracle_searcher *s, const uint8_t *hay, size_t end, int lanes)
{
#if defined(ORACLE_X86)
    if (!s->force_scalar && have_avx2()) {
        switch (lanes) {
        case 2: return vsi_sse2_2(s, hay, end);
        case 4: return vsi_sse2_4(s, hay, end);
        case 8: return vsi_sse2_8(s, hay, end);
        case 16: return vsi_sse2_16(s, hay, end);
        default: return vsi_avx2_32(s, hay, end);
        }
    }
#endif
    switch (lanes) {
    case 2: return vsi_scalar_2(s, hay, end);
    case 4: return vsi_scalar_4(s, hay, end);
    case 8: return vsi_scalar_8(s, hay, end);
    case 16: return vsi_scalar_16(s, hay, end);
    default: return vsi_scalar_32(s, hay, end);
    }
}

/* ---- Avx2Searcher::inlined_search_in: x86.rs:356-376 --------------------------------------- */

static int avx2_searcher_search_in(const oracle_searcher *s, const uint8_t *hay, size_t len)
{
    if (len <= s->n) return len == s->n && memcmp(hay, s->needle, len) == 0;
    const size_t end = len - s->n + 1;  /* >= 2 */
    if (end < 4) return vector_search_in(s, hay, end, 2);
    if (end < 8) return vector_search_in(s, hay, end, 4);
    if (end < 16) return vector_search_in(s, hay, end, 8);
    if (end < 32) return vector_search_in(s, hay, end, 16);
    return vector_search_in(s, hay, end, 32);
}

/* ---- DynamicAvx2Searcher::inlined_search_in: x86.rs:498-519 -------------------------------- */

int oracle_search_in(const oracle_searcher *s, const uint8_t *hay, size_t len)
{
    switch (s->kind) {
    case KIND_N0: return 1;
    case KIND_N1: /* lib.rs:130-136 */
        if (len == 0) return 0;
        return memchr(hay, s->needle[0], len) != NULL;
    default: return avx2_searcher_search_in(s, hay, len);
    }
}

/* ---- the reference tests' own oracle: lib.rs:371-373, tests/i386.rs:6-10 ------------------- */

/* windows(n).any(|w| w == needle).  Rust's windows(0) panics, so the reference never asks its
 * oracle about an empty needle; for n == 0 this returns what DynamicAvx2Searcher::N0 does (true). */
int oracle_naive(const uint8_t *hay, size_t len, const uint8_t *needle, size_t n)
{
    if (n == 0) return 1;
    if (len < n) return 0;
    for (size_t i = 0; i + n <= len; ++i) {
        size_t j = 0;
        while (j < n && hay[i + j] == needle[j]) ++j;
        if (j == n) return 1;
    }
    return 0;
}

/* ---- corpus sweeps (tests/i386.rs:46-70), looped in C so the 10.5 M pairs finish in seconds - */

/* words are given as one blob + (count+1) offsets, already sorted by length by the caller.
 * mode 0: restated searcher; 1: naive; 2: both, returns -1 on any disagreement. */
long long oracle_sweep_short(const uint8_t *blob, const uint64_t *off, size_t count, int mode)
{
    long long hits = 0;
    for (size_t i = 0; i < count; ++i) {
        const uint8_t *nd = blob + off[i];
        const size_t n = (size_t)(off[i + 1] - off[i]);
        oracle_searcher *s = NULL;
        if (oracle_searcher_init_default(&s, nd, n) != ORACLE_OK) return -2;
        for (size_t j = i; j < count; ++j) {
            const uint8_t *h = blob + off[j];
            const size_t len = (size_t)(off[j + 1] - off[j]);
            int r;
            if (mode == 1) {
                r = oracle_naive(h, len, nd, n);
            } else {
                r = oracle_search_in(s, h, len);
                if (mode == 2 && r != oracle_naive(h, len, nd, n)) {
                    oracle_searcher_free(s);
                    return -1;
                }
            }
            hits += r;
        }
        oracle_searcher_free(s);
    }
    return hits;
}

long long oracle_sweep_long(const uint8_t *hay, size_t len, const uint8_t *blob, const uint64_t *off,
                            size_t count, int mode)
{
    long long hits = 0;
    for (size_t i = 0; i < count; ++i) {
        const uint8_t *nd = blob + off[i];
        const size_t n = (size_t)(off[i + 1] - off[i]);
        oracle_searcher *s = NULL;
        if (oracle_searcher_init_default(&s, nd, n) != ORACLE_OK) return -2;
        int r = mode == 1 ? oracle_naive(hay, len, nd, n) : oracle_search_in(s, hay, len);
        if (mode == 2 && r != oracle_naive(hay, len, nd, n)) {
            oracle_searcher_free(s);
            return -1;
        }
        hits += r;
        oracle_searcher_free(s);
    }
    return hits;
}

/* Timed form of the long sweep for bench.py's config-0 line: searchers prebuilt (untimed in the
 * reference too, bench/benches/i386.rs:246-250), then `iters` passes of one search per needle. */
long long oracle_bench_long(const uint8_t *hay, size_t len, const uint8_t *blob, const uint64_t *off,
                            size_t count, int iters)
{
    oracle_searcher **ss = (oracle_searcher **)calloc(count, sizeof *ss);
    long long hits = 0;
    for (size_t i = 0; i < count; ++i)
        oracle_searcher_init_default(&ss[i], blob + off[i], (size_t)(off[i + 1] - off[i]));
    for (int it = 0; it < iters; ++it)
        for (size_t i = 0; i < count; ++i) hits += oracle_search_in(ss[i], hay, len);
    for (size_t i = 0; i < count; ++i) oracle_searcher_free(ss[i]);
    free(ss);
    return hits;
}

/* Timed form of the short sweep (bench/benches/i386.rs:118-129): searchers prebuilt (untimed), then
 * `iters` passes of: for (i, searcher) in searchers { for haystack in &needles[i..] { search_in } }. */
long long oracle_bench_short(const uint8_t *blob, const uint64_t *off, size_t count, int iters)
{
    oracle_searcher **ss = (oracle_searcher **)calloc(count, sizeof *ss);
    long long hits = 0;
    for (size_t i = 0; i < count; ++i)
        oracle_searcher_init_default(&ss[i], blob + off[i], (size_t)(off[i + 1] - off[i]));
    for (int it = 0; it < iters; ++it)
        for (size_t i = 0; i < count; ++i)
            for (size_t j = i; j < count; ++j)
                hits += oracle_search_in(ss[i], blob + off[j], (size_t)(off[j + 1] - off[j]));
    for (size_t i = 0; i < count; ++i) oracle_searcher_free(ss[i]);
    free(ss);
    return hits;
}

/* ---- multi-threaded CPU baseline (NOT in the reference, which is single-threaded) ---------- */
/* Same range-shard rule the GPU path uses across devices: thread t scans bytes
 * [t*S, (t+1)*S + n-1) clipped to len; OR of the per-thread booleans. */

typedef struct {
    const oracle_searcher *s;
    const uint8_t *hay;
    size_t len;
    int found;
    int t, threads;
} mt_arg;

/* Thread t of T is confined to the CPUs of the NUMA node that holds the same relative share of the machine
 * (node floor(t/T * nodes), from /sys/devices/system/node/node<k>/cpulist, intersected with the process'
 * affinity mask); inside the node the scheduler spreads the threads over cores as usual.  The generator below
 * uses the same rule, so with a block partition of the buffer a thread scans memory that a thread of the same
 * node first touched, whatever the two thread counts are.  Best effort: on any failure nothing is pinned. */
#include <stdio.h>

static int node_cpus(int node, cpu_set_t *out)
{
    char path[96], buf[4096];
    snprintf(path, sizeof path, "/sys/devices/system/node/node%d/cpulist", node);
    FILE *fh = fopen(path, "r");
    if (!fh) return 0;
    const size_t got = fread(buf, 1, sizeof buf - 1, fh);
    fclose(fh);
    buf[got] = 0;
    CPU_ZERO(out);
    int n = 0;
    for (char *p = buf; *p;) {                       /* "0-63,128-191" */
        char *end;
        long lo = strtol(p, &end, 10);
        if (end == p) break;
        long hi = lo;
        if (*end == '-') hi = strtol(end + 1, &end, 10);
        for (long c = lo; c <= hi && c < CPU_SETSIZE; ++c) {
            CPU_SET((int)c, out);
            ++n;
        }
        p = *end == ',' ? end + 1 : end;
        if (*p == '\n') break;
    }
    return n;
}

static void pin_to_share(int t, int threads)
{
    cpu_set_t allowed, node, both;
    if (threads <= 0 || sched_getaffinity(0, sizeof allowed, &allowed) != 0) return;
    int nodes = 0;
    while (nodes < 64 && node_cpus(nodes, &node) > 0) ++nodes;
    if (nodes < 2) return;                           /* one node (or no sysfs): nothing to gain */
    const int k = (int)(((long long)t * nodes) / threads);
    if (node_cpus(k, &node) <= 0) return;
    CPU_AND(&both, &node, &allowed);
    if (CPU_COUNT(&both) > 0) (void)pthread_setaffinity_np(pthread_self(), sizeof both, &both);
}

/* The threads are a POOL that lives from one call to the next (re-built when the thread count changes): created per call, 256
 * threads cost several milliseconds of pthread_create / sysfs reads / affinity calls - as much as the 10 ms scan they were
 * created for, which made the rates at 64 threads and more come out BELOW those at 32 ("what's weak" 9).  Workers
 * park on a barrier; a call sets their shares, releases them, waits on a second barrier and ORs their booleans. */
static struct {
    int threads;
    pthread_t *tid;
    mt_arg *arg;
    pthread_barrier_t start, done;
    int quit;
} g_pool;
static pthread_mutex_t g_pool_mu = PTHREAD_MUTEX_INITIALIZER;

static void *mt_worker(void *p)
{
    mt_arg *a = (mt_arg *)p;
    pin_to_share(a->t, a->threads);
    for (;;) {
        pthread_barrier_wait(&g_pool.start);
        if (g_pool.quit) break;
        a->found = oracle_search_in(a->s, a->hay, a->len);
        pthread_barrier_wait(&g_pool.done);
    }
    return NULL;
}

static void pool_resize(int threads)
{
    if (g_pool.threads == threads) return;
    if (g_pool.threads > 0) {
        g_pool.quit = 1;
        pthread_barrier_wait(&g_pool.start);
        for (int t = 0; t < g_pool.threads; ++t) pthread_join(g_pool.tid[t], NULL);
        pthread_barrier_destroy(&g_pool.start);
        pthread_barrier_destroy(&g_pool.done);
        free(g_pool.tid);
        free(g_pool.arg);
        g_pool.threads = 0;
        g_pool.quit = 0;
    }
    if (threads <= 0) return;
    g_pool.tid = (pthread_t *)calloc((size_t)threads, sizeof *g_pool.tid);
    g_pool.arg = (mt_arg *)calloc((size_t)threads, sizeof *g_pool.arg);
    pthread_barrier_init(&g_pool.start, NULL, (unsigned)threads + 1);
    pthread_barrier_init(&g_pool.done, NULL, (unsigned)threads + 1);
    g_pool.threads = threads;
    for (int t = 0; t < threads; ++t) {
        g_pool.arg[t].t = t;
        g_pool.arg[t].threads = threads;
        pthread_create(&g_pool.tid[t], NULL, mt_worker, &g_pool.arg[t]);
    }
}

int oracle_search_in_mt(const oracle_searcher *s, const uint8_t *hay, size_t len, int threads)
{
    if (threads <= 1 || s->n == 0 || len < s->n * 2 || len < (size_t)threads * 4096)
        return oracle_search_in(s, hay, len);
    pthread_mutex_lock(&g_pool_mu);
    pool_resize(threads);
    const size_t shard = (len + (size_t)threads - 1) / (size_t)threads;
    int found = 0;
    for (int t = 0; t < threads; ++t) {
        size_t b = (size_t)t * shard;
        size_t e = b + shard + s->n - 1;
        if (b > len) b = len;
        if (e > len) e = len;
        g_pool.arg[t].s = s;
        g_pool.arg[t].hay = hay + b;
        g_pool.arg[t].len = e - b;
        g_pool.arg[t].found = 0;
    }
    pthread_barrier_wait(&g_pool.start);
    pthread_barrier_wait(&g_pool.done);
    for (int t = 0; t < threads; ++t) found |= g_pool.arg[t].found;
    pthread_mutex_unlock(&g_pool_mu);
    return found;
}

/* ---- independent restatement of the synthetic-haystack generator -- */
/* byte(i) = (splitmix64(splitmix64(seed) ^ (i >> 3)) >> (8 * (i & 7))) & 0xFF, then 0xFF -> 0x00.
 * (The seed is hashed so that nearby seeds do not yield word-permuted copies of one stream.)
 * The product has its own host and device versions; tests compare all three. */

static inline uint64_t splitmix64(uint64_t x)
{
    uint64_t z = x + 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

void oracle_fill_random(uint8_t *dst, uint64_t global_offset, size_t len, uint64_t seed)
{
    size_t k = 0;
    const uint64_t key = splitmix64(seed);
    while (k < len) {
        const uint64_t i = global_offset + k;
        uint64_t w = splitmix64(key ^ (i >> 3)) >> (8 * (i & 7));
        size_t take = 8 - (size_t)(i & 7);
        if (take > len - k) take = len - k;
        for (size_t j = 0; j < take; ++j, w >>= 8) {
            const uint8_t b = (uint8_t)w;
            dst[k + j] = b == 0xFF ? 0x00 : b;
        }
        k += take;
    }
}

/* The same bytes, written by `threads` pinned threads over a block partition (first touch = local memory for
 * the multi-threaded baseline; see pin_to_share). */
typedef struct {
    uint8_t *dst;
    uint64_t off;
    size_t len;
    uint64_t seed;
    int t, threads;
} fill_arg;

static void *fill_worker(void *p)
{
    fill_arg *a = (fill_arg *)p;
    pin_to_share(a->t, a->threads);
    oracle_fill_random(a->dst, a->off, a->len, a->seed);
    return NULL;
}

void oracle_fill_random_mt(uint8_t *dst, uint64_t global_offset, size_t len, uint64_t seed, int threads)
{
    if (threads <= 1 || len < (size_t)threads * 4096) {
        oracle_fill_random(dst, global_offset, len, seed);
        return;
    }
    pthread_t *tid = (pthread_t *)calloc((size_t)threads, sizeof *tid);
    fill_arg *arg = (fill_arg *)calloc((size_t)threads, sizeof *arg);
    const size_t shard = (len + (size_t)threads - 1) / (size_t)threads;
    for (int t = 0; t < threads; ++t) {
        size_t b = (size_t)t * shard, e = b + shard;
        if (b > len) b = len;
        if (e > len) e = len;
        arg[t].dst = dst + b;
        arg[t].off = global_offset + b;
        arg[t].len = e - b;
        arg[t].seed = seed;
        arg[t].t = t;
        arg[t].threads = threads;
        pthread_create(&tid[t], NULL, fill_worker, &arg[t]);
    }
    for (int t = 0; t < threads; ++t) pthread_join(tid[t], NULL);
    free(tid);
    free(arg);
}

int oracle_have_avx2(void) { return have_avx2(); }
