// ss_scan.hip - a search of ONE device-resident haystack: kernel selection, the Problem of a (searcher, haystack), the launch,
// and the entry points that wait for the answer.
//   ss_search_device / _async    DynamicAvx2Searcher::search_in       /root/reference/src/x86.rs:498-525
//                                (N0 -> true x86.rs:500; N1 = MemchrSearcher lib.rs:130-136;
//                                 len < n -> false / len == n -> equality x86.rs:357-359)
//   ss_find_device / _async      row f1 of SURVEY.md 8f (the Option<usize> shape of bench/sse4-strstr/src/lib.rs:4-15)
// The scan itself lives in scan_filters.hpp / scan_kernels.hpp.  There is no CPU search path in this file.
#include "ss_internal.hpp"

#include <algorithm>

#define SS_AUX_PUBLISH 1
#include "aux_kernels.hpp"
#include "scan_launch.hpp"

namespace ssh {

// Kernel timing (ss_searcher_set_timing): the hipEvent pair that brackets a scan belongs to the CALLING
// THREAD (one pair per thread and device, created on first use), so concurrent calls on one handle never
// share events; ss_searcher_last_kernel_ms reports the calling thread's most recent timed scan.
namespace {
struct ThreadTimer {
    hipEvent_t ev0[kMaxDevices] = {nullptr}, ev1[kMaxDevices] = {nullptr};
    uint64_t owner[kMaxDevices] = {0};              // per device: the uid of the searcher of the most recent timed scan there (0: none)
    int dev = -1;                                   // the device of the most recent timed scan
    ~ThreadTimer()
    {
        if (process_exiting()) return;                  // leak: see ExitMark (ss_core.hip)
        for (int d = 0; d < kMaxDevices; ++d) {
            if (ev0[d]) (void)hipEventDestroy(ev0[d]);
            if (ev1[d]) (void)hipEventDestroy(ev1[d]);
        }
    }
};
thread_local ThreadTimer g_timer;
}  // namespace

void timer_forget(const ss_searcher *s)
{
    for (auto &o : g_timer.owner)
        if (o == s->uid) o = 0;
}

// The calling thread's most recent timed scan through the searcher `uid` on device `dev` (dev < 0: on the device it last launched
// on).  Keyed by the searcher's uid, not its address: a record that outlives its searcher (another thread's) can never pass for
// a later searcher allocated at the same address, and nobody dereferences a searcher to read it.
int thread_last_kernel_ms(uint64_t uid, int dev, float *ms)
{
    ThreadTimer &tm = g_timer;
    if (dev < 0) dev = tm.dev;
    if (dev < 0 || dev >= kMaxDevices || uid == 0 || tm.owner[dev] != uid)
        return fail(SS_ERR_ARGUMENT, "no timed scan has been launched through this searcher by the calling thread");
    HIP_TRY(hipEventSynchronize(tm.ev1[dev]));
    HIP_TRY(hipEventElapsedTime(ms, tm.ev0[dev], tm.ev1[dev]));
    return SS_OK;
}

bool spin_wait_enabled()
{
    static const bool on = []() { const char *v = getenv("SLICESLICE_SPIN_WAIT"); return !(v && v[0] == '0'); }();
    return on;
}

#ifdef SS_TEST_HOOKS
bool cross_exit_enabled()
{
    const char *off = getenv("SLICESLICE_CROSS_EXIT");          // read per call: a test compares both settings in one process
    return !(off && off[0] == '0');
}
#endif

hipError_t launch_signal_flag(hipStream_t st, const int *d_flag, int epoch, long long *h_word, int pair)
{
    ss::signal_flag_kernel<<<1, 1, 0, st>>>(d_flag, epoch, h_word, pair);
    return hipGetLastError();
}

hipError_t launch_publish_best(hipStream_t st, uint64_t *d_best, uint64_t *h_best, int pair)
{
    ss::publish_best_kernel<<<1, 1, 0, st>>>(d_best, h_best, pair);
    return hipGetLastError();
}

namespace {

// ---- kernel selection -------------------------------------------------------------------------------
// variant = 100000*B + 10000*OCC + 1000*LAYOUT + 100*MODE + 10*U + NT.  B: workgroup size (0/2 = 256 threads,
// 1 = 128, 3 = 512); OCC: at most OCC workgroups per CU through unused dynamic LDS (0 = no cap) - both are
// tuning aids (profiles/r01/workgroup_size_sweep.jsonl, occupancy_sweep.jsonl).  LAYOUT 0 = automatic, 1 = 16 bytes per lane throughout,
// 2 = 8-bytes-per-lane first phase (single-stream kernels).  U in {4,8} = pieces (KiB) per wave per tile; NT in {0,1} = plain /
// non-temporal loads; MODE: 0 = single stream (d == 0), 2 = one stream + cross-lane (ds_bpermute) position flags (0 < d <=
// kShiftMaxD; fill_problem never produces a larger d).  variant 0 = automatic: U = 4, non-temporal loads.  The variant digits
// are read in hooks builds only (ss_searcher_set_variant); the product library always launches the automatic choice.
struct Launch {
    int U;
    int nt;
    int mode;   // 0: d == 0, 2: shifted flags + a third byte, 3: shifted flags alone
    bool l8;    // 8-bytes-per-lane first phase (mode 0 / one-byte needles)
    uint32_t dyn_lds;   // unused dynamic LDS per workgroup (caps workgroups per CU; tuning: variant 10000*OCC)
    unsigned block;     // threads per workgroup: 128 / 256 / 512 (tuning: variant 100000*B, B = 1 / 2 / 3)
};

// Measured (tools/tune.py, profiles/r01/l8_short_needles.jsonl, two-tile workgroups): the 8-byte first phase
// is +5 % for one-byte needles (64 GiB: 7.46 vs 7.08 TB/s) but -5 % for two-byte filters on random bytes
// (positions 2, 3: 6.95 vs 7.35 TB/s): there one tile in sixteen holds a candidate and pays for the
// transposition into the 16-byte layout on top of the regular filter.  Automatic choice: one-byte needles
// only; 2xxx variants force it for tuning.

// (Workgroups per CU, the pair-alone kernels and the filter bytes of `new`-built searchers on large haystacks: ss_census.hip.)
constexpr int kAutoU = 4;
constexpr int kAutoTilesPerBlock = 2;    // 32 KiB contiguous per workgroup at U = 4 (profiles/r01/tiles_per_block_sweep.jsonl)

// Unused dynamic LDS that leaves room for exactly `occ` workgroups of `block` threads per CU (160 KiB of LDS).
uint32_t occupancy_pad(int occ, unsigned block)
{
    const uint32_t per = (160u * 1024u) / (uint32_t)occ;
    const uint32_t fixed = (block / ss::kWave) * ss::kNeedleLds;
    // (1 KiB short of the share: the kernels also own a few bytes of static LDS - the completion word's workgroup flag -
    // and a workgroup's allocation is rounded up to the hardware's granule)
    uint32_t pad = per > fixed + 2048 ? ((per - fixed - 1024) & ~1023u) : 0;
    if (pad > 64u * 1024u - fixed) pad = 64u * 1024u - fixed;
    return pad;
}

// Workgroups per CU.  Since the cold half of the Problem left the registers (scan_kernels.hpp, ColdInKernarg) the multi-byte
// kernels need 77-83 VGPRs, so the register file would admit six workgroups of four waves per CU; how many actually run is
// set per launch through unused dynamic LDS.  Measured in one process on one buffer (tools/occ_probe.py,
// profiles/r03/occupancy_probe.jsonl; 16-byte needle on random bytes, 1 / 8 / 32 GiB): FOUR per CU 7.36 / 7.46 / 7.42 TB/s,
// five 7.01 / 7.20 / 7.17, six 7.00 / 7.24 / 7.23 - a streaming scan that rarely sees a candidate wants exactly one
// workgroup per SIMD quartet.  A scan that keeps meeting candidates wants latency hiding instead: on the i386 text, phrases of
// the manual's stock vocabulary run at 6.1-6.2 TB/s with four per CU and 7.1 with six, the reference's own pair (0, n-1) on
// text 3.9-4.4 against 5.0-5.8.  The library cannot see the haystack, so it goes by the needle: when EVERY filter byte is
// text-like (byte_rarity_rank >= 64: letters, digits, blanks, common punctuation, NUL) the haystack is presumably text and
// candidates are to be expected - six per CU; otherwise (a random or binary needle: its rarest bytes are in the filter) four.
// One-byte needles (8-byte loads, 36-68 VGPRs) stay at four: 7.28-7.44 TB/s either way.
Launch pick_variant(int variant, uint64_t d, bool one_byte, int workgroups_per_cu, bool pair_alone)
{
    Launch l;
    l.U = kAutoU;
    l.mode = d == 0 ? 0 : (pair_alone ? 3 : 2);
    l.nt = 1;
    l.l8 = one_byte;
    l.block = ss::kBlock;
    l.dyn_lds = occupancy_pad(workgroups_per_cu, l.block);
#ifdef SS_TEST_HOOKS
    if (variant > 0) {
        l.dyn_lds = 0;
        if (variant >= 100000) {                                // Bxxxxx: workgroup size
            const int b = variant / 100000;
            l.block = b == 1 ? 128 : (b == 3 ? 512 : 256);
            variant %= 100000;
        }
        if (variant >= 10000) {                                 // OCCxxxx: at most OCC workgroups per CU (160 KiB LDS)
            l.dyn_lds = occupancy_pad(variant / 10000, l.block);
            variant %= 10000;
        }
        if (variant >= 1000) l.l8 = variant / 1000 == 2;       // 1xxx: 16-byte layout, 2xxx: 8-byte first phase
        variant %= 1000;
        const int m = variant / 100, u = (variant / 10) % 10;
        if (u == 4 || u == 8) l.U = u;
        if (d != 0 && (m == 2 || m == 3)) l.mode = m;         // x2xx / x3xx: the cross-lane kernels with / without the third byte
        l.nt = (variant % 10) ? 1 : 0;
#ifndef SS_TUNING_VARIANTS
        // a hooks build without the variant kernels holds ONE load flavour (scan_launch.hpp::kernel_built): the launch-shape
        // digits of a variant keep working for every filter pair
        l.nt = 1;
#endif
    }
#else
    (void)variant;
#endif
    return l;
}

}  // namespace

// launch_scan_un<U, NT, FIND> is defined in scan_launch.hpp and explicitly instantiated in the
// scan_inst_*.hip translation units, so that the kernel families compile in parallel.
namespace {
template <int U>
bool launch_scan_u(int nt, const ss::Problem &pr, int q, int mode, bool one_byte, const ss::Shape &shape, hipStream_t st,
                   void *flag, bool l8)
{
#ifdef SS_TUNING_VARIANTS
    if (nt == 0) return ss::launch_scan_un<U, 0, false>(pr, q, mode, one_byte, shape, st, flag, l8);
#else
    if (nt == 0) return false;                  // the product library holds the non-temporal kernels only
#endif
    return ss::launch_scan_un<U, 1, false>(pr, q, mode, one_byte, shape, st, flag, l8);
}

// Builds the Problem for (hay, len) and enqueues the scan.  find == false: *d_sink is an int flag, OR-ed
// (0 -> 1), never cleared.  find == true: *d_sink is a uint64, atomicMin'ed with find_base + offset of
// every match the grid sees (the leftmost one survives).  Preconditions: 1 <= n <= len.
// done_slot >= 0: the call owns flag slot `done_slot` and would like to wait on the slot's completion word
// instead of the stream; granted (*used_done = true) for grids of at most kDoneMaxBlocks workgroups.
constexpr uint64_t kDoneMaxBlocks = 256;         // one atomic per workgroup on ONE address: small grids only (4 MiB);
                                                 // measured: 1 KiB 8.6 vs 11.9 us per call, break-even near 1 MiB
}  // namespace

void fill_problem(const ss_searcher *s, const uint8_t *d_needle, const void *d_hay, size_t len, uint64_t find_base, ss::Problem *out,
                  ProblemShape *shape, const size_t *triple)
{
    ss::Problem &pr = *out;
    const size_t n = s->n;
    const bool one_byte = n == 1;
    // The filter stream starts at the FIRST filter byte: candidate i is tested through hay[fa + i] == needle[fa]
    // and hay[fb + i] == needle[fb], so the kernel's aligned coordinates are those of hay + fa, while matches
    // are verified (and reported) at hay + i.  Bytes in front of hay + fa are never candidates (their index
    // wraps and fails `i < end`), and the last byte either stream touches is hay[len - n + fb] <= hay[len - 1].
    // the triple the DEVICE tests: the searcher's (derive_device_filter), or the one chosen for this haystack (ss_census.hip)
    const size_t fa = one_byte ? 0 : (triple ? triple[0] : s->da), fb = one_byte ? 0 : (triple ? triple[1] : s->db);
    const size_t fc = triple ? triple[2] : s->dc;
    const uint8_t *hf = static_cast<const uint8_t *>(d_hay) + fa;
    pr.hay = static_cast<const uint8_t *>(d_hay);
    pr.mis = (uint32_t)((uintptr_t)hf & 15);
    pr.base = hf - pr.mis;
    pr.needle = d_needle;
    pr.n = n;
    pr.end = (uint64_t)len - n + 1;
    pr.nchunks_all = ((uint64_t)pr.mis + (len - fa) + 15) / 16;
    pr.npieces = (((uint64_t)pr.mis + pr.end + 15) / 16 + 63) / 64;
    size_t position = fb - fa;                          // distance between the two filter bytes
    pr.d = position / 16;
    // third first-phase byte, at most 15 behind the first; "none" (needles of two bytes) = needle[fa + position % 16] once more -
    // any (byte, offset) of the needle is a valid condition
    const bool three = !one_byte && fc > fa && fc - fa <= 15 && fc < n && fc != fb;
    size_t position3 = three ? fc - fa : position % 16;
    // The two further bytes are interchangeable; the kernels are instantiated for "the third byte's dword is not behind
    // the second's" only (10 copies of the first phase instead of 16 - and two of the six others, second byte in dword 0
    // with the third in dword 1 or 3, came out of the compiler waiting for all four loads of a tile before the first
    // xor: 6.3-6.4 instead of 7.4 TB/s, profiles/r02/ab_filter_triples.jsonl).
    if (three && pr.d == 0 && position3 / 4 > position / 4) std::swap(position, position3);
    const uint32_t sh = (uint32_t)(position % 16);
    pr.r = sh % 4;
    pr.n0x4 = 0x01010101u * s->needle[fa];
    pr.nlx4 = 0x01010101u * s->needle[one_byte ? 0 : fa + position];
    pr.q3 = (uint32_t)(position3 / 4);
    pr.r3 = (uint32_t)(position3 % 4);
    pr.n3x4 = 0x01010101u * s->needle[one_byte ? 0 : fa + position3];
    // second-level filter: up to 15 further needle bytes behind the first filter byte
    pr.norder = ss::build_refine_order(s->needle.data() + fa, n - fa, position, pr.order_idx, pr.order_val,
                                       (uint64_t)position3);
    pr.find_base = find_base;
    pr.host_flag = nullptr;
    pr.epoch = 1;
    pr.done_counter = nullptr;
    pr.host_done = nullptr;
    pr.done_target = pr.done_hi = 0;
    pr.flags = 0;
    pr.q = (uint32_t)(sh / 4);
    pr.far_off = one_byte || triple ? 0 : (uint64_t)s->far;
    // exact in-register verification: the needle ends at most 16 bytes behind the first filter byte (lib.rs:222-241)
    pr.exact_len = 0;
    pr.tail16[0] = pr.tail16[1] = pr.tail16[2] = pr.tail16[3] = 0;
    if (!one_byte && pr.d == 0 && n - fa <= 16) {
        // ... plus as many of the fa bytes in front of it as sixteen leave room for: a needle of up to 16 bytes is compared whole
        const size_t behind = n - fa, back = std::min(fa, 16 - behind), el = behind + back;
        pr.exact_len = (uint32_t)(el | (back << 8));
        uint8_t t16[16] = {0};
        memcpy(t16, s->needle.data() + fa - back, el);
        memcpy(pr.tail16, t16, 16);
    }
    shape->position = position;
    shape->position3 = position3;
    shape->fa = fa;
    shape->one_byte = one_byte;
}

int enqueue_scan(const ss_searcher *s, PerDevice *pd, const void *d_hay, size_t len, hipStream_t st, void *d_sink, bool find,
                 uint64_t find_base, int *host_flag, int epoch, int done_slot, bool *used_done)
{
#ifdef SS_TEST_HOOKS
    if (s->debug_fail_scans.load(std::memory_order_relaxed) > 0 && s->debug_fail_scans.fetch_sub(1) > 0)
        return fail(SS_ERR_HIP, "injected scan failure (ss_debug_fail_next_scans)");
#endif
    void *d_flag = d_sink;
    // what the haystack has said so far (ss_census.hip): candidate counts of this searcher's filter, perhaps better filter bytes
    LaunchHints hints;
    launch_hints(s, pd, d_hay, len, st, &hints);
    ss::Problem pr;
    ProblemShape ps;
    fill_problem(s, pd->d_needle, d_hay, len, find_base, &pr, &ps, hints.have_triple ? hints.tri : nullptr);
    if (hints.have_order && pr.d == 0 && s->n >= 2) {
        // the second level's schedule in the order the census MEASURED (ss_census.hip, build_measured_order): the needle byte that
        // kills most of this haystack's candidates first - instead of the static rarity order fill_problem wrote
        pr.norder = hints.norder;
        pr.order_idx[0] = hints.order_idx[0];
        pr.order_idx[1] = hints.order_idx[1];
        pr.order_val[0] = hints.order_val[0];
        pr.order_val[1] = hints.order_val[1];
    }
    pr.host_flag = host_flag;
    pr.epoch = epoch;
    const bool one_byte = ps.one_byte;
    const size_t fa = ps.fa, position = ps.position, position3 = ps.position3;
    const uint32_t sh = (uint32_t)(position % 16);

    // Workgroups per CU.  Without census counts: a guess from the NEEDLE (every filter byte text-like -> the haystack is
    // presumably text; single-stream kernels only) - or four, when the filter bytes have just been chosen for being rare HERE.
    const bool text_like = !one_byte && ss::byte_rarity_rank(s->needle[fa]) >= 64 && ss::byte_rarity_rank(s->needle[fa + position]) >= 64 &&
                           ss::byte_rarity_rank(s->needle[fa + position3]) >= 64;
    int occ = !one_byte && text_like && pr.d == 0 && !hints.have_triple ? 6 : 4;      // (autotune off: this guess is all there is)
    bool pair_alone = false;
    if (hints.have_counts) {
        occ = hints.workgroups_per_cu;
        // a pair 16 or more apart that rarely matches on this haystack needs no third byte in the first phase (MODE 3)
        pair_alone = pr.d != 0 && !find && hints.sparse_pair;
    }
    if (!one_byte && len >= kCensusMinBytes && autotune_enabled() && __atomic_load_n(&pd->last_found, __ATOMIC_RELAXED) != 0) occ = 4;
    Launch l = pick_variant(s->variant, pr.d, one_byte, occ, pair_alone);
    if (find && l.mode == 3) l.mode = 2;                     // find() has no pair-alone kernels
    if (l.mode == 3)                                        // the third byte goes back into the second level's schedule
        pr.norder = ss::build_refine_order(s->needle.data() + fa, s->n - fa, position, pr.order_idx, pr.order_val);
    const uint64_t wpb = l.block / ss::kWave;
    const uint64_t ntiles = (pr.npieces + wpb * l.U - 1) / (wpb * l.U);
    uint64_t blocks, tpb;
    if (s->grid > 0) {
        blocks = (uint64_t)s->grid;
        if (blocks > ntiles) blocks = ntiles;
        tpb = 0;
    } else {
        if (s->grid < 0) {
            tpb = (uint64_t)(-(int64_t)s->grid);
        } else {
            // Short-lived workgroups: two tiles (32 KiB) each from 2 GiB up (from 1 GiB for filter pairs >= 16 apart), one
            // below.  The hardware dispatcher hands out tiles in address order, so the set of lines in flight
            // stays one narrow, advancing window, and a fresh workgroup issues its loads the moment a slot
            // frees up.  Measured (profiles/r01/tiles_per_block_sweep.jsonl, tiles_1_vs_2.txt; 16-byte needle):
            // 64 GiB 7.40-7.43 TB/s at 2 tiles per workgroup vs 7.28 at 4, 7.21 at 8, 7.11 at 64.  One tile is
            // +2 % at 1 GiB, within +-0.7 % from 4 GiB up (and +1.5 % for one-byte needles), but twice as many
            // workgroups have to be drained after an early match (the entry peek in scan_kernel), and the
            // cross-lane kernels (pairs >= 16 apart) lose 5 % with it: each wave re-loads its halo chunks per tile.
            DeviceInfo di;
            if (int rc = device_info(pd->dev, &di)) return rc;
            tpb = ntiles / ((uint64_t)di.cus * (l.mode == 0 ? 256 : 128));
            if (tpb > (uint64_t)kAutoTilesPerBlock) tpb = kAutoTilesPerBlock;
            if (tpb < 1) tpb = 1;
            // ... unless the census has counted candidates (ss_census.hip: two tiles at four workgroups per CU where candidate tiles are
            // no rarity, one at five and six whatever the size; single-stream kernels, launches that follow the census's shape)
            if (l.mode == 0 && s->variant == 0 && hints.have_counts && hints.tiles_per_workgroup != 0 && occ == hints.workgroups_per_cu)
                tpb = (uint64_t)hints.tiles_per_workgroup;
        }
        blocks = (ntiles + tpb - 1) / tpb;
        while (blocks > 0x7fffffffull) {        // gridDim.x limit
            tpb *= 2;
            blocks = (ntiles + tpb - 1) / tpb;
        }
    }
    if (blocks < 1) blocks = 1;
    __atomic_store_n(&pd->last_occ, s->variant == 0 ? occ : 0, __ATOMIC_RELAXED);
    __atomic_store_n(&pd->last_grid, (unsigned)blocks, __ATOMIC_RELAXED);
    __atomic_store_n(&pd->last_mode, l.mode, __ATOMIC_RELAXED);
    const ss::Shape shape = {(unsigned)blocks, l.block, tpb, l.dyn_lds};
    if (used_done) *used_done = false;
    if (done_slot >= 0 && used_done && blocks <= kDoneMaxBlocks && (!find || len < (1ull << ss::kFindOffsetBits))) {
        const int k = done_slot;
        if (pd->done_low[k] > kDoneLowMax || (find && pd->find_tag[k] == 0)) start_over(pd, k);
        pr.done_counter = pd->d_done + k;
        pr.host_done = pd->h_done + k;
        pr.done_target = pd->done_low[k] + (uint32_t)blocks;
        pr.done_hi = pd->done_hi[k];
        pr.flags |= ss::kProblemCounted;
        pd->done_low[k] = pr.done_target;                // (a launch that fails starts the slot over)
        pr.host_flag = nullptr;                          // the completion word carries the answer
        if (find) {                                      // keyed minimum in the slot's own word: see scan_kernel
            d_flag = pd->d_best_done + k;
            pr.find_base += (uint64_t)pd->find_tag[k]-- << ss::kFindOffsetBits;
        }
        *used_done = true;
    }

    ThreadTimer &tm = g_timer;
    const bool timed = s->timing && pd->dev >= 0 && pd->dev < kMaxDevices;
    if (timed) {
        if (!tm.ev0[pd->dev]) {
            HIP_TRY(hipEventCreate(&tm.ev0[pd->dev]));
            HIP_TRY(hipEventCreate(&tm.ev1[pd->dev]));
        }
        HIP_TRY(hipEventRecord(tm.ev0[pd->dev], st));
    }
    const int q = (int)(sh / 4);
    bool launched = false;
    if (find) {   // one tile shape for find(): U = 4
        if (l.U != 4) return fail(SS_ERR_ARGUMENT, "find supports the U = 4 kernels only");
        if (l.nt) launched = ss::launch_scan_un<4, 1, true>(pr, q, l.mode, one_byte, shape, st, d_flag, false);
#ifdef SS_TUNING_VARIANTS
        else launched = ss::launch_scan_un<4, 0, true>(pr, q, l.mode, one_byte, shape, st, d_flag, false);
#endif
    } else if (l.U == 8) {
#ifdef SS_TUNING_VARIANTS
        launched = launch_scan_u<8>(l.nt, pr, q, l.mode, one_byte, shape, st, d_flag, l.l8);
#endif
    } else {
        launched = launch_scan_u<4>(l.nt, pr, q, l.mode, one_byte, shape, st, d_flag, l.l8);
    }
    if (!launched)
        return fail(SS_ERR_ARGUMENT, "kernel variant %d (U = %d, %s loads, mode %d%s) is not part of this build: only the tuning build "
                                     "(-DSS_TUNING_VARIANTS, libsliceslice_hip_tuning.so) holds kernels other than the ones the constructors "
                                     "and ss_searcher_set_filter3 can select",
                    s->variant, l.U, l.nt ? "non-temporal" : "plain", l.mode, l.l8 ? ", 8-byte first phase" : "");
    HIP_TRY(hipGetLastError());
    if (timed) {
        HIP_TRY(hipEventRecord(tm.ev1[pd->dev], st));
        tm.owner[pd->dev] = s->uid;
        tm.dev = pd->dev;
    }
    return SS_OK;
}


// Scans too large for the workgroup count of the completion word still answer through a pinned word when the scan is short
// enough to be waited for by spinning: a one-lane kernel behind the scan (behind the all-reduce, for a sharded search) stores
// epoch << 1 | found.  A host that reaches hipStreamSynchronize before the work is done pays a wake-up on top of it (a 16 MiB
// search: 15.2 us per call with the stream wait, 10.7 with the spin; an 8 GiB shard: 2-7 us per search, box to box).  The spin is bounded by twice the time the scan can
// possibly take at HBM speed (+ 300 us); after that - the stream was busy with other work - the stream wait takes over.
// ... and the sharded entry points only bother from a few MiB per shard: below, the stream wait returns at once (the work is
// done before the host gets there) and the extra launch costs 3-4 us (1 MiB shards: 24.6 -> 27.4 us per search with it).

bool spin_for_word(const long long *word, int epoch, double estimate_us, int *found)
{
    const auto t0 = std::chrono::steady_clock::now();
    const auto budget = std::chrono::microseconds((long long)(2.0 * estimate_us) + 300);
    for (unsigned spins = 0;; ++spins) {
        const long long v = __atomic_load_n(word, __ATOMIC_ACQUIRE);
        if (((uint32_t)v >> 1) == (uint32_t)epoch) {
            *found = (int)(v & 1);
            return true;
        }
        cpu_relax();
        if ((spins & 255) == 255 && std::chrono::steady_clock::now() - t0 > budget) return false;
    }
}

// The answer word of a sharded search (signal_flag_kernel (pair form)): epoch << 2 | "a rank failed" << 1 | found.
bool spin_for_shard_word(const long long *word, int epoch, double estimate_us, int *found, int *failed)
{
    const auto t0 = std::chrono::steady_clock::now();
    const auto budget = std::chrono::microseconds((long long)(2.0 * estimate_us) + 300);
    for (unsigned spins = 0;; ++spins) {
        const unsigned long long v = (unsigned long long)__atomic_load_n(word, __ATOMIC_ACQUIRE);
        if ((v >> 2) == (unsigned long long)(uint32_t)epoch) {
            *found = (int)(v & 1);
            *failed = (int)((v >> 1) & 1);
            return true;
        }
        cpu_relax();
        if ((spins & 255) == 255 && std::chrono::steady_clock::now() - t0 > budget) return false;
    }
}


}  // namespace ssh

using namespace ssh;

extern "C" {

int ss_searcher_last_launch(const ss_searcher *s, int *workgroups_per_cu, unsigned *grid)
{
    if (!s || !workgroups_per_cu || !grid) return fail(SS_ERR_ARGUMENT, "NULL argument");
    PerDevice *pd = nullptr;
    if (int rc = get_per_device(s, &pd)) return rc;
    *workgroups_per_cu = __atomic_load_n(&pd->last_occ, __ATOMIC_RELAXED);
    *grid = __atomic_load_n(&pd->last_grid, __ATOMIC_RELAXED);
    return SS_OK;
}

int ss_searcher_last_kernel_ms(const ss_searcher *s, float *ms)
{
    if (!s || !ms) return fail(SS_ERR_ARGUMENT, "NULL argument");
    return thread_last_kernel_ms(s->uid, -1, ms);
}

int ss_search_device_async(const ss_searcher *s, const void *d_haystack, size_t len, void *hip_stream,
                           int *d_found)
{
    if (!s || !d_found) return fail(SS_ERR_ARGUMENT, "NULL argument");
    if (len && !d_haystack) return fail(SS_ERR_ARGUMENT, "haystack is NULL");
    SearchGate gate(s);                                  // set_filter* are refused while this call runs
    hipStream_t st = static_cast<hipStream_t>(hip_stream);
    PerDevice *pd = nullptr;
    if (int rc = get_per_device(s, &pd)) return rc;
    if (s->n == 0) {                                    // N0: true for every haystack (x86.rs:500)
        static const int one = 1;
        HIP_TRY(hipMemcpyAsync(d_found, &one, sizeof one, hipMemcpyHostToDevice, st));
        return SS_OK;
    }
    if (len < s->n) return SS_OK;                       // cannot occur; flag untouched
    s->used_async.store(true, std::memory_order_release);
    return enqueue_scan(s, pd, d_haystack, len, st, d_found);
}

int ss_search_device(const ss_searcher *s, const void *d_haystack, size_t len, void *hip_stream, int *found)
{
    if (!s || !found) return fail(SS_ERR_ARGUMENT, "NULL argument");
    if (len && !d_haystack) return fail(SS_ERR_ARGUMENT, "haystack is NULL");
    SearchGate gate(s);                                  // set_filter* are refused while this call runs
    if (s->n == 0) { *found = 1; return SS_OK; }        // x86.rs:500
    if (len < s->n) { *found = 0; return SS_OK; }       // x86.rs:357-359 (len == n is decided on the device)
    hipStream_t st = static_cast<hipStream_t>(hip_stream);
    PerDevice *pd = nullptr;
    if (int rc = get_per_device(s, &pd)) return rc;
    const int k = acquire_slot(s, pd);
    // The wave that finds a match stores the call's epoch to the device flag (polled by the grid for the
    // early exit) AND to its pinned-host mirror, so the answer needs neither a device-to-host copy nor a
    // reset of the slot afterwards: launch, wait for the stream, compare.
    const int epoch = next_epoch(pd, k);                // the slot is owned by this call
    const bool spin_ok = spin_wait_enabled();
    // the slot's completion word may hold a find()'s answer (offset + 1), which could pass for 2 * epoch + found
    __atomic_store_n(pd->h_done + k, 0ll, __ATOMIC_RELAXED);
    bool used_done = false;
    int rc = enqueue_scan(s, pd, d_haystack, len, st, pd->d_flags + k, false, 0, pd->h_flags + k, epoch, spin_ok ? k : -1,
                          &used_done);
    bool answered = false;
    // the answer word of a small grid: found-half of the slot's counter << 32 | epoch << 1 | found
    auto take = [&](long long v) {
        if (((uint32_t)v >> 1) != (uint32_t)epoch) return false;
        *found = (int)(v & 1);
        pd->done_hi[k] = (uint32_t)((unsigned long long)v >> 32);
        return true;
    };
    if (rc == SS_OK && used_done) {
        // Small grid: the workgroup that completes the count stores the answer word to the slot's pinned word.  Spin on it
        // for a bounded time (the whole call is a few microseconds); after that - a long kernel behind other work on the
        // stream, or a fault - fall back to the stream wait, which also reports errors.
        const auto t0 = std::chrono::steady_clock::now();
        for (unsigned spins = 0;; ++spins) {
            if (take(__atomic_load_n(pd->h_done + k, __ATOMIC_ACQUIRE))) {
                answered = true;
                // every 256th call still waits for the stream, so that the runtime retires its completed commands
                // in bounded batches instead of whenever the caller next synchronises
                if ((epoch & 255) == 0) (void)hipStreamSynchronize(st);
                break;
            }
            cpu_relax();
            if ((spins & 255) == 255 && std::chrono::steady_clock::now() - t0 > std::chrono::microseconds(200)) break;
        }
    }
    if (rc == SS_OK && !used_done && spin_ok && scan_estimate_us(len) <= kSpinMaxEstimateUs) {
        // larger grid: the word is written by a one-lane kernel behind the scan
        if (launch_signal_flag(st, pd->d_flags + k, epoch, pd->h_done + k, 0) == hipSuccess && spin_for_word(pd->h_done + k, epoch, scan_estimate_us(len), found)) {
            answered = true;
            if ((epoch & 255) == 0) (void)hipStreamSynchronize(st);
        }
    }
    if (rc == SS_OK && !answered) {
        const hipError_t e = hipStreamSynchronize(st);
        if (e != hipSuccess) rc = fail(SS_ERR_HIP, "stream wait: %s", hipGetErrorString(e));
        else if (used_done) { if (!take(__atomic_load_n(pd->h_done + k, __ATOMIC_ACQUIRE))) rc = fail(SS_ERR_HIP, "the completion word was not written"); }
        else *found = __atomic_load_n(pd->h_flags + k, __ATOMIC_ACQUIRE) == epoch;
    }
    if (rc != SS_OK) start_over(pd, k);             // a failed launch may have left a partial workgroup count behind
    if (rc == SS_OK) __atomic_store_n(&pd->last_found, *found, __ATOMIC_RELAXED);      // (launch tuning: see "Workgroups per CU")
    release_slot(s, pd, k);
    return rc;
}

int ss_find_device_async(const ss_searcher *s, const void *d_haystack, size_t len, uint64_t base_offset,
                         void *hip_stream, uint64_t *d_best)
{
    if (!s || !d_best) return fail(SS_ERR_ARGUMENT, "NULL argument");
    if (len && !d_haystack) return fail(SS_ERR_ARGUMENT, "haystack is NULL");
    SearchGate gate(s);                                  // set_filter* are refused while this call runs
    hipStream_t st = static_cast<hipStream_t>(hip_stream);
    PerDevice *pd = nullptr;
    if (int rc = get_per_device(s, &pd)) return rc;
    if (len < s->n) return SS_OK;
    if (s->n == 0) {        // the empty needle matches at offset 0 of every haystack
        if (base_offset != 0) return SS_OK;     // only the first shard reports it
        static const uint64_t zero = 0;
        HIP_TRY(hipMemcpyAsync(d_best, &zero, sizeof zero, hipMemcpyHostToDevice, st));
        return SS_OK;
    }
    s->used_async.store(true, std::memory_order_release);
    return enqueue_scan(s, pd, d_haystack, len, st, d_best, true, base_offset);
}

int ss_find_device(const ss_searcher *s, const void *d_haystack, size_t len, void *hip_stream, uint64_t *position)
{
    if (!s || !position) return fail(SS_ERR_ARGUMENT, "NULL argument");
    if (len && !d_haystack) return fail(SS_ERR_ARGUMENT, "haystack is NULL");
    SearchGate gate(s);                                  // set_filter* are refused while this call runs
    if (s->n == 0) { *position = 0; return SS_OK; }
    if (len < s->n) { *position = SS_NPOS; return SS_OK; }
    hipStream_t st = static_cast<hipStream_t>(hip_stream);
    PerDevice *pd = nullptr;
    if (int rc = get_per_device(s, &pd)) return rc;
    const int k = acquire_slot(s, pd);
    // Small grid: the workgroup that completes the count stores the answer - offset + 1, or all ones - to the slot's
    // completion word (zeroed here first: the word also serves ss_search_device, whose values carry an epoch), and the host
    // spins on the word as ss_search_device does; the minimum lives in the slot's keyed word (enqueue_scan), which needs no
    // re-arming.  Larger grids: slots of d_best are all-ones whenever they are free; a one-lane kernel behind the scan
    // stores the minimum to the slot's pinned mirror (no device-to-host copy command) and re-arms the slot.
    const bool spin_ok = spin_wait_enabled();
    __atomic_store_n(pd->h_done + k, 0ll, __ATOMIC_RELAXED);
    bool used_done = false;
    int rc = enqueue_scan(s, pd, d_haystack, len, st, pd->d_best + k, true, 0, nullptr, 1, spin_ok ? k : -1, &used_done);
    bool answered = false;
    if (rc == SS_OK && used_done) {
        const auto t0 = std::chrono::steady_clock::now();
        for (unsigned spins = 0;; ++spins) {
            const long long v = __atomic_load_n(pd->h_done + k, __ATOMIC_ACQUIRE);
            if (v != 0) {
                *position = v == -1ll ? SS_NPOS : (uint64_t)v - 1;
                answered = true;
                // as in ss_search_device: a real stream wait now and then lets the runtime retire its commands
                if ((next_epoch(pd, k) & 255) == 0) (void)hipStreamSynchronize(st);
                break;
            }
            cpu_relax();
            if ((spins & 255) == 255 && std::chrono::steady_clock::now() - t0 > std::chrono::microseconds(200)) break;
        }
        if (!answered) {
            const hipError_t e = hipStreamSynchronize(st);
            if (e != hipSuccess) {
                rc = fail(SS_ERR_HIP, "stream wait: %s", hipGetErrorString(e));
            } else {
                const long long v = __atomic_load_n(pd->h_done + k, __ATOMIC_ACQUIRE);
                if (v == 0) rc = fail(SS_ERR_HIP, "find: the completion word was not written");
                else *position = v == -1ll ? SS_NPOS : (uint64_t)v - 1;
                answered = rc == SS_OK;
            }
        }
    } else if (rc == SS_OK) {
        // the pinned mirror starts as "pending" (a value no minimum can take), so that a scan short enough to be waited for
        // by spinning (see spin_for_word) is: the one-lane kernel's store ends the wait
        constexpr uint64_t kPending = ~0ull - 1;
        __atomic_store_n(pd->h_best + k, kPending, __ATOMIC_RELAXED);
        hipError_t e = launch_publish_best(st, pd->d_best + k, pd->h_best + k, 0);
        bool have = false;
        if (e == hipSuccess && spin_ok && scan_estimate_us(len) <= kSpinMaxEstimateUs) {
            const auto t0 = std::chrono::steady_clock::now();
            const auto budget = std::chrono::microseconds((long long)(2.0 * scan_estimate_us(len)) + 300);
            for (unsigned spins = 0; !have; ++spins) {
                have = __atomic_load_n(pd->h_best + k, __ATOMIC_ACQUIRE) != kPending;
                if (!have) {
                    cpu_relax();
                    if ((spins & 255) == 255 && std::chrono::steady_clock::now() - t0 > budget) break;
                }
            }
            if (have && (next_epoch(pd, k) & 255) == 0) (void)hipStreamSynchronize(st);
        }
        if (e == hipSuccess && !have) e = hipStreamSynchronize(st);
        if (e != hipSuccess) rc = fail(SS_ERR_HIP, "position read-back: %s", hipGetErrorString(e));
        else *position = __atomic_load_n(pd->h_best + k, __ATOMIC_ACQUIRE);
    }
    if (rc != SS_OK) {
        start_over(pd, k);                                   // a failed launch may have left a partial workgroup count behind
        (void)hipMemset(pd->d_best + k, 0xFF, sizeof(uint64_t));
    }
    release_slot(s, pd, k);
    return rc;
}

}  // extern "C"
