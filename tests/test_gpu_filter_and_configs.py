"""GPU parity of what round 2 added: the filter-pair choice of `new` (results must not depend on the pair -
the reference's own claim for `position`, src/lib.rs:375-378), BASELINE.json configs 3 and 5 at their FULL
shape, the batched position contract, slot exhaustion and the epoch wrap.  Bit-exact booleans / offsets.
Reference citations are paths under /root/reference."""
import ctypes
import os
import random
import threading
import time

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")


@pytest.fixture(scope="module")
def ss():
    import sliceslice_rs_amd as m
    assert torch.cuda.is_available(), "these tests must run on the GPU box"
    m.lib()
    return m


@pytest.fixture(scope="module")
def O():
    from oracle import oracle
    return oracle


def dev(b):
    a = np.frombuffer(b, dtype=np.uint8) if isinstance(b, (bytes, bytearray)) else np.asarray(b, dtype=np.uint8)
    if a.size == 0:
        return torch.empty(0, dtype=torch.uint8, device="cuda")
    return torch.from_numpy(a.copy()).cuda()


def absent_needle(ss, n, seed=0x5EED0002):
    nd = bytearray(ss.fill_random_host(n, seed).tobytes())
    nd[0 if n == 1 else (1 if n == 2 else n // 2)] = 0xFF
    return bytes(nd)


# ---- filter pair ------------------------------------------------------------------------------------------

def test_every_filter_pair_gives_the_reference_answer_on_the_kats(ss, kat):
    """src/lib.rs:375-378 runs every KAT for every `position`; here for every (first, second) pair."""
    for row in kat["generic"]:
        hay, needle = row["haystack"].encode(), row["needle"].encode()
        dh = dev(hay)
        s = ss.DynamicHipSearcher.new(needle)
        assert s.position == (len(needle) - 1) % (1 << 64)
        n = len(needle)
        for a in range(n):
            for b in range(a, n):
                if n < 2 and (a, b) != (0, 0):
                    continue
                s.set_filter(a, b)
                assert s.filter == (a, b) and s.filter3 == (a, b, b)
                assert s.search_in(dh) == row["expected"], (row, a, b)
                assert s.search_in(hay) == row["expected"], (row, a, b)           # host path
                p = s.find(dh)
                want = hay.find(needle)
                assert p == (None if want < 0 else want), (row, a, b)
                # ... and with every third byte the single-stream kernels can take (short needles: all of them)
                if n >= 3 and b > a and b - a <= 15:
                    thirds = [c for c in range(a + 1, min(n, a + 16)) if c != b]
                    for c in (thirds if n <= 12 else thirds[::5]):
                        s.set_filter(a, b, c)
                        assert s.filter3 == (a, b, c)
                        assert s.search_in(dh) == row["expected"], (row, a, b, c)
                        assert s.find(dh) == (None if want < 0 else want), (row, a, b, c)


def test_filter_pair_bounds_are_checked(ss):
    s = ss.DynamicHipSearcher.new(b"abcdef")
    for a, b in ((1, 0), (0, 6), (6, 6), (5, 7)):
        with pytest.raises(ss.PositionError):
            s.set_filter(a, b)
    for a, b, c in ((0, 3, 0), (2, 4, 1), (0, 3, 6), (1, 2, 1)):
        with pytest.raises(ss.PositionError):
            s.set_filter(a, b, c)
    long = ss.DynamicHipSearcher.new(bytes(range(1, 101)))
    long.set_filter(0, 16)                              # 16 apart: cross-lane kernel, no third byte
    with pytest.raises(ss.PositionError):
        long.set_filter(0, 16, 5)
    with pytest.raises(ss.PositionError):
        long.set_filter(0, 5, 16)
    long.set_filter(3, 18, 4)
    assert long.filter3 == (3, 18, 4)
    s1 = ss.DynamicHipSearcher.new(b"a")
    s1.set_filter(0, 0)
    with pytest.raises(ss.PositionError):
        s1.set_filter(0, 1)


def test_new_picks_rare_bytes_and_with_position_keeps_the_callers_byte(ss):
    s = ss.DynamicHipSearcher.new(b" the quick brown fox ")
    assert s.position == 20 and s.filter == ss.choose_filter_pair(b" the quick brown fox ") == (5, 19)   # 'q', 'x'
    assert s.filter3 == ss.choose_filter_triple(b" the quick brown fox ") == (5, 19, 9)                  # ... and 'k'
    for p in (0, 7, 15):
        w = ss.DynamicHipSearcher.with_position(b" the quick brown fox ", p)
        assert w.position == p and w.filter == (0, p)        # up to 15 apart the reference's pair is kept
    assert ss.DynamicHipSearcher.with_position(b" the quick brown fox ", 7).filter3 == (0, 7, 5)        # + 'q' as the third
    # further apart the caller's byte stays, its partner moves next to it: 'q', ' ' (position 20), 'x'
    w = ss.DynamicHipSearcher.with_position(b" the quick brown fox ", 20)
    assert w.position == 20 and w.filter3 == (5, 20, 19)
    with ss.tuning_build():                                  # the pure host form of the same choice (hooks builds)
        assert ss.choose_filter_for_position(b" the quick brown fox ", 20) == (5, 20, 19)
    w.set_filter(0, 20)                                      # the reference's pair, verbatim (cross-lane kernel)
    assert w.filter3 == (0, 20, 20)
    # a 2000-byte needle: the default position 1999 would need two load streams; `new` and `with_position` stay within 15 bytes
    long_needle = (b"lorem ipsum dolor sit amet, " * 80)[:2000]
    a, b = ss.DynamicHipSearcher.new(long_needle).filter
    assert a < b <= a + 15
    for p in (16, 500, 1007, 1008, 1999):
        a, b, c = ss.DynamicHipSearcher.with_position(long_needle, p).filter3
        assert b == p and p - 15 <= a < p and a < c <= a + 15 and c != p
    # the reference's own pair at any distance is one set_filter away
    far = ss.DynamicHipSearcher.new(long_needle)
    far.set_filter(0, 1999)
    assert far.filter3 == (0, 1999, 1999) and far.position == 1999


def test_filter_pairs_too_far_apart_for_any_kernel(ss, O):
    """A pair more than 16 * 62 + 15 bytes apart (only set_filter can ask: e.g. the reference's pair (0, n-1) of a long needle) has
    no kernel of its own since the two-stream kernels went: the device filters with the first byte and two partners behind it,
    and the caller's far byte is the first thing a surviving candidate is tested for in memory.  Answers against the oracle:
    absent, present at both ends and across tile edges, and near misses that differ from the needle ONLY in the far byte, only
    in a partner byte, only in the last byte."""
    rng = np.random.default_rng(11)
    n = 3000
    needle = bytes(rng.integers(1, 255, n, dtype=np.uint8))
    ln = (4 << 20) + 333
    base = torch.empty(ln + 32, dtype=torch.uint8, device="cuda")
    ss.fill_random_device(base, 0x5EED0BEE)
    nd = torch.from_numpy(np.frombuffer(needle, dtype=np.uint8).copy()).cuda()
    for mis in (0, 5):
        hay = base[mis:mis + ln]
        for first, second in ((0, 2999), (0, 1008), (7, 1500), (1900, 2999), (0, 1007)):   # the last one: cross-lane kernel (d = 62)
            s = ss.DynamicHipSearcher.new(needle)
            s.set_filter(first, second)
            assert s.filter3 == (first, second, second)
            host = hay.cpu().numpy()
            assert s.search_in(hay) is O.OracleSearcher(needle).search_in(host) is False
            for at in (0, ln - n, 16384 - first - 1, (2 << 20) + 7):
                saved = hay[at:at + n].clone()
                hay[at:at + n] = nd
                assert s.search_in(hay) is True and s.find(hay) == at, (mis, first, second, at)
                for k in (second, first + 1, first + 2, n - 1, 0):
                    if k == first:
                        continue
                    hay[at + k] ^= 0x5A
                    assert s.search_in(hay) is False and s.find(hay) is None, (mis, first, second, at, k)
                    hay[at + k] = needle[k]
                hay[at:at + n] = saved
            assert s.search_in(hay) is False


def test_filter_pairs_on_text_and_random_haystacks_vs_oracle(ss, O, corpus):
    """Random pairs - near, 16..1008 apart (cross-lane kernels), further (two streams), close to the needle's
    end (no second-level bytes left) - on a text haystack and a random one, misaligned, against the oracle."""
    rng = random.Random(2024)
    text = corpus["i386"]
    buf = torch.empty(len(text) + 64, dtype=torch.uint8, device="cuda")
    rnd_host = np.asarray(ss.fill_random_host(3 << 20, 77))
    rbuf = torch.empty(rnd_host.size + 64, dtype=torch.uint8, device="cuda")
    for mis in (0, 3, 13):
        tview = buf[mis:mis + len(text)]
        tview.copy_(torch.from_numpy(np.frombuffer(text, dtype=np.uint8).copy()))
        rview = rbuf[mis:mis + rnd_host.size]
        rview.copy_(torch.from_numpy(rnd_host))
        cases = []
        for n in (2, 3, 9, 16, 17, 40, 300, 1100, 2500):
            at = rng.randrange(len(text) - n)
            cases.append((tview, text, text[at:at + n]))                               # present in the text
            bad = bytearray(text[at:at + n])
            bad[rng.randrange(n)] ^= 0x15
            cases.append((tview, text, bytes(bad)))                                    # (almost surely) absent
            at = rng.randrange(rnd_host.size - n)
            cases.append((rview, rnd_host.tobytes(), rnd_host[at:at + n].tobytes()))
            cases.append((rview, rnd_host.tobytes(), absent_needle(ss, n)))
        for view, host, needle in cases:
            n = len(needle)
            want = O.OracleSearcher(needle).search_in(np.frombuffer(host, dtype=np.uint8))
            wantp = host.find(needle)
            assert want == (wantp >= 0)
            s = ss.DynamicHipSearcher.new(needle)
            pairs = {s.filter, (0, n - 1), (n - 2, n - 1), (0, 1)}
            for _ in range(6):
                a = rng.randrange(n - 1)
                pairs.add((a, rng.randrange(a + 1, min(n, a + 16))))                   # near
                pairs.add((a, rng.randrange(a, n)))                                    # anywhere
            for a, b in sorted(pairs):
                s.set_filter(a, b)
                assert s.search_in(view) == want, (n, mis, a, b)
                assert s.find(view) == (None if wantp < 0 else wantp), (n, mis, a, b)
                if n >= 3 and 0 < b - a <= 15:
                    for c in {rng.randrange(a + 1, min(n, a + 16)) for _ in range(3)} - {b}:
                        s.set_filter(a, b, c)
                        assert s.search_in(view) == want, (n, mis, a, b, c)
                        assert s.find(view) == (None if wantp < 0 else wantp), (n, mis, a, b, c)


def test_histogram_driven_triple_on_an_adversarial_corpus(ss):
    """A corpus the static ranking is wrong about (all 'a', needle a...ae): `new` filters on three 'a's and every offset
    reaches the second level; the histogram-driven triple contains the 'e' and nothing passes.  Same answers."""
    ln = 64 << 20
    hay = torch.full((ln,), 0x61, dtype=torch.uint8, device="cuda")
    needle = b"a" * 40 + b"e"
    s = ss.DynamicHipSearcher.new(needle)
    assert s.search_in(hay) is False and s.find(hay) is None
    hist = ss.byte_histogram(hay, sample_bytes=1 << 20)
    a, b, c = ss.choose_filter_triple(needle, hist)
    assert 40 in (a, b, c)
    s.set_filter(a, b, c)                                # (what it buys in time: tests/test_gpu_zz_timing.py)
    assert s.search_in(hay) is False and s.find(hay) is None
    hay[ln - 41:] = dev(needle)
    assert s.search_in(hay) is True and s.find(hay) == ln - 41
    assert ss.DynamicHipSearcher.new(needle).find(hay) == ln - 41


def test_filter_stream_never_reads_outside_the_haystack(ss):
    """hay + first is the start of the filter stream: with the haystack flush against BOTH ends of an allocation
    no pair may fault or report the needle copies that sit just outside."""
    hip = ctypes.CDLL("libamdhip64.so.7")
    size = 1 << 16
    p = ctypes.c_void_p()
    assert hip.hipMalloc(ctypes.byref(p), ctypes.c_size_t(size)) == 0
    try:
        assert hip.hipMemset(p, 0x2E, ctypes.c_size_t(size)) == 0
        needle = bytes([0x30 + k for k in range(40)])
        s = ss.DynamicHipSearcher.new(needle)
        for a, b, c in ((0, 39, None), (38, 39, None), (20, 35, 34), (0, 1, 15), (3, 3, None), (39, 39, None), (24, 39, 38)):
            s.set_filter(a, b, c)
            for ln in (40, 41, 4097, size):
                assert s.search_in((p.value + size - ln, ln)) is False, (a, b, ln)
                assert s.search_in((p.value, ln)) is False, (a, b, ln)
    finally:
        hip.hipFree(p)


# ---- BASELINE.json config 3 at full shape -------------------------------------------------------------------

def test_config3_one_gib_needle_length_sweep_full_shape(ss):
    """1 GiB synthetic haystack x needle lengths {1,2,4,8,32,128}: absent by construction -> False / None;
    planted at 0, mid (straddling 2^29) and len-n -> True and find() returns the plant (or an earlier natural
    occurrence, only possible for n <= 4 - checked against the planted bytes)."""
    ln = 1 << 30
    t = torch.empty(ln, dtype=torch.uint8, device="cuda")
    ss.fill_random_device(t, 0x5EED0001)
    for n in (1, 2, 4, 8, 32, 128):
        nd = absent_needle(ss, n)
        s = ss.DynamicHipSearcher.new(nd)
        assert s.search_in(t) is False, n
        assert s.find(t) is None, n
        present = bytes(ss.fill_random_host(n, 0x5EED0003).tobytes())
        assert 0xFF not in present
        sp = ss.DynamicHipSearcher.new(present)
        pn = dev(present)
        naturally = sp.find(t)                             # n <= 4: a random needle occurs somewhere already
        assert naturally is None or n <= 4, n
        for at in (ln - n, (ln // 2) - n // 2, 0):
            saved = t[at:at + n].clone()
            t[at:at + n] = pn
            assert sp.search_in(t) is True, (n, at)
            got = sp.find(t)
            assert got is not None and got <= at, (n, at, got)
            if naturally is None:
                assert got == at, (n, at, got)
            else:
                assert t[got:got + n].cpu().numpy().tobytes() == present
            t[at:at + n] = saved
        assert sp.find(t) == naturally, n
    del t


# ---- BASELINE.json config 5 at full shape -------------------------------------------------------------------

def test_config5_batched_4096_x_1mib_full_shape_with_plants(ss):
    """4096 problems x 1 MiB haystacks x 16-byte needles, ONE launch.  ~200 problems get their needle planted -
    at 0, at len-16, and straddling the edges between the slices a problem is cut into - the flags must equal
    the plant list exactly."""
    count, each = 4096, 1 << 20
    blob = torch.empty(count * each, dtype=torch.uint8, device="cuda")
    ss.fill_random_device(blob, 0x5EED0001)
    nd = bytearray(ss.fill_random_host(16 * count, 0x5EED0077).tobytes())       # 0xFF-free needles
    # unplanted problems must be absent: give them one 0xFF byte; planted ones stay 0xFF-free
    rng = random.Random(5)
    planted = sorted(rng.sample(range(count), 200))
    plant_set = set(planted)
    for i in range(count):
        if i not in plant_set:
            nd[16 * i + 8] = 0xFF
    nblob = dev(bytes(nd))
    # a problem of 1 MiB = 64 tiles of 16 KiB; the launch cuts it into 1..8 contiguous slices: plant at every
    # multiple of 1/8 of the haystack (minus 1, 8, 15 bytes), at 0 and at len-16
    spots = [0, each - 16] + [k * (each // 8) - d for k in range(1, 8) for d in (1, 8, 15)] + [k * (each // 8) for k in range(1, 8)]
    where = {}
    for j, i in enumerate(planted):
        at = spots[j % len(spots)]
        where[i] = at
        blob[i * each + at:i * each + at + 16] = nblob[16 * i:16 * i + 16]
    hay_off = (torch.arange(count + 1, dtype=torch.int64) * each).cuda()
    nd_off = (torch.arange(count + 1, dtype=torch.int64) * 16).cuda()
    found = ss.search_batched(blob, hay_off, nblob, nd_off).cpu().numpy()
    assert sorted(np.nonzero(found)[0].tolist()) == planted
    assert set(np.unique(found).tolist()) <= {0, 1}
    # spot-check with the single-problem entry point (search + find) on a few planted and unplanted problems
    for i in planted[:6] + [k for k in range(count) if k not in plant_set][:3]:
        s = ss.DynamicHipSearcher.new(bytes(nd[16 * i:16 * i + 16]))
        view = blob[i * each:(i + 1) * each]
        assert s.search_in(view) is (i in plant_set)
        assert s.find(view) == where.get(i)
    del blob


def test_batched_position_contract(ss):
    """position[i] follows the with_position rules (x86.rs:300, 473); a violation is reported as
    SS_BATCH_BAD_POSITION for that problem by both many-problem kernels instead of being clamped."""
    hays = [b"hello world", b"hello world", b"x", b"", b"hello world", b"ab"]
    needles = [b"world", b"world", b"x", b"", b"o", b"abc"]
    pos = [4, 5, 1, 99, 0, 3]          # ok, == n (bad), one-byte needle with position 1 (bad), N0 any, ok, bad even though len < n
    want = [1, -1, -1, 1, 1, -1]
    hay_off = np.zeros(len(hays) + 1, dtype=np.int64)
    hay_off[1:] = np.cumsum([len(h) for h in hays])
    nd_off = np.zeros(len(needles) + 1, dtype=np.int64)
    nd_off[1:] = np.cumsum([len(x) for x in needles])
    blob, nblob = dev(b"".join(hays) + b"\0"), dev(b"".join(needles) + b"\0")
    p = torch.tensor(pos, dtype=torch.int64, device="cuda")
    for pairs in (False, True):
        got = ss.search_batched(blob, torch.from_numpy(hay_off).cuda(), nblob, torch.from_numpy(nd_off).cuda(), position=p,
                                pairs=pairs).cpu().tolist()
        assert got == want, (pairs, got)


# ---- host library: slot exhaustion, epoch wrap ---------------------------------------------------------------

def test_128_concurrent_calls_on_one_handle(ss):
    """More callers than flag slots (64) on ONE handle and device: the surplus waits on a condition variable;
    every call gets its own answer (search and find, device and host haystacks mixed)."""
    ln = 2 << 20
    yes = torch.zeros(ln, dtype=torch.uint8, device="cuda")
    no = torch.zeros(ln, dtype=torch.uint8, device="cuda")
    needle = bytes([9, 8, 7, 6, 5, 4, 3])
    yes[ln - 7:] = dev(needle)
    host_yes = yes.cpu().numpy()
    host_no = no.cpu().numpy()
    s = ss.DynamicHipSearcher.new(needle)
    streams = [torch.cuda.Stream() for _ in range(128)]
    errors = []
    barrier = threading.Barrier(128)

    def worker(k):
        try:
            barrier.wait()
            for it in range(12):
                kind = (k + it) % 4
                st = streams[k].cuda_stream
                if kind == 0:
                    assert s.search_in(yes, stream=st) is True
                elif kind == 1:
                    assert s.search_in(no, stream=st) is False
                elif kind == 2:
                    assert s.find(yes, stream=st) == ln - 7
                    assert s.find(no, stream=st) is None
                else:
                    assert s.search_in(host_yes if it % 2 else host_no) is bool(it % 2)
        except Exception as e:                      # noqa: BLE001
            errors.append((k, repr(e)))

    threads = [threading.Thread(target=worker, args=(k,)) for k in range(128)]
    [t.start() for t in threads]
    [t.join(timeout=300) for t in threads]
    assert not any(t.is_alive() for t in threads)
    assert not errors, errors[:3]


def test_epoch_wrap_of_the_flag_slots(ss):
    """"found" is a per-call epoch, never cleared; at 2^31 calls the counters wrap and both copies of the flag
    are reset.  The test hook moves the counters to the edge; answers must stay right across it."""
    ln = 1 << 20
    yes = torch.zeros(ln, dtype=torch.uint8, device="cuda")
    yes[1000:1003] = torch.tensor([7, 7, 9], dtype=torch.uint8)
    no = torch.zeros(ln, dtype=torch.uint8, device="cuda")
    with ss.tuning_build():                           # the hooks (ss_debug_*) exist in builds with -DSS_TEST_HOOKS only
        s = ss.DynamicHipSearcher.new(bytes([7, 7, 9]))
    assert s.search_in(yes) is True and s.search_in(no) is False
    assert s._L.ss_debug_set_epochs(s._h, 2**31 - 3) == 0
    for it in range(8):                               # slot 0 is reused by every sequential call: crosses the wrap
        assert s.search_in(yes) is True, it
        assert s.search_in(no) is False, it
        assert s.search_in(yes.cpu().numpy()) is True, it
    # the completion-word state - a workgroup count that is never reset, the count of workgroups that found the needle,
    # the decreasing key of find()'s minimum - has its own limits; the hook parks all three just short of them
    for state in ((0x7FFF0000 - 700, 5, 1000),          # the workgroup count about to start over
                  (12345, 0xFFFFFFFA, 1000),             # the found count about to wrap
                  (12345, 77, 3),                        # find()'s key about to run out
                  (0x7FFF0000 - 100, 0xFFFFFFFE, 2)):    # all three at once
        assert s._L.ss_debug_set_completion_state(s._h, *state) == 0
        for it in range(16):
            assert s.search_in(yes) is True and s.find(yes) == 1000, (state, it)
            assert s.search_in(no) is False and s.find(no) is None, (state, it)


def test_completion_word_path_behind_a_long_kernel_and_across_sizes(ss):
    """Small grids answer through a pinned completion word the host spins on (bounded), larger ones through the stream
    wait.  A small search queued BEHIND a long scan on the same stream outlives the spin budget and must fall back to
    the stream wait with the right answer; sizes on both sides of the 256-workgroup threshold agree with Python."""
    big = torch.empty(6 << 30, dtype=torch.uint8, device="cuda")
    ss.fill_random_device(big, 0x5EED0001)
    small = torch.zeros(1 << 20, dtype=torch.uint8, device="cuda")
    small[777:780] = torch.tensor([5, 6, 7], dtype=torch.uint8)
    s_long = ss.DynamicHipSearcher.new(absent_needle(ss, 16))
    s_yes, s_no = ss.DynamicHipSearcher.new(bytes([5, 6, 7])), ss.DynamicHipSearcher.new(bytes([5, 6, 8]))
    st = torch.cuda.Stream()
    flag = torch.zeros(1, dtype=torch.int32, device="cuda")
    for _ in range(5):
        with torch.cuda.stream(st):
            for _ in range(4):                                    # ~3 ms of queued work in front of the small searches
                s_long.search_in_async(big, flag)
        assert s_yes.search_in(small, stream=st.cuda_stream) is True
        assert s_no.search_in(small, stream=st.cuda_stream) is False
        with torch.cuda.stream(st):
            for _ in range(4):
                s_long.search_in_async(big, flag)
        assert s_yes.find(small, stream=st.cuda_stream) == 777      # find() answers through the same word
        assert s_no.find(small, stream=st.cuda_stream) is None
    # find and search_in interleaved on one handle share the slots' completion words (offset + 1 vs 2 * epoch + found)
    for k in range(300):
        small[777 + 3 * k:780 + 3 * k] = torch.tensor([5, 6, 7], dtype=torch.uint8)
        assert s_yes.find(small) == 777 and s_yes.search_in(small) is True
        assert s_no.search_in(small) is False and s_no.find(small) is None
    st.synchronize()
    assert int(flag.item()) == 0
    del big
    rng = random.Random(9)
    for ln in (1, 100, 16 << 10, (4 << 20) - 3, 4 << 20, (4 << 20) + 16385, 8 << 20, (8 << 20) + 1, 40 << 20):
        host = np.frombuffer(rng.randbytes(ln), dtype=np.uint8).copy()
        dh = dev(host)
        for _ in range(6):
            n = rng.choice([1, 2, 3, 16, 40])
            if n > ln:
                continue
            at = rng.randrange(ln - n + 1)
            nd = host[at:at + n].tobytes() if rng.random() < 0.5 else rng.randbytes(n)
            want = host.tobytes().find(nd)
            s = ss.DynamicHipSearcher.new(nd)
            for _ in range(3):
                assert s.search_in(dh) == (want >= 0), (ln, n)
            assert s.find(dh) == (None if want < 0 else want), (ln, n)


def test_short_differential_campaign():
    """tools/fuzz_gpu.py for a few seconds in both modes: random haystack kinds / lengths / misalignments / needles /
    positions or filter triples / kernel variants / launch shapes, search_in and find against Python's bytes.find."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    # the campaign draws kernel variants and launch shapes at random when it runs against the tuning build (every variant); against
    # the product library it draws haystacks / needles / positions / filter triples and every search takes the automatic choice
    tuning = sys.modules["sliceslice_rs_amd._build"].build_tuning()
    # small-haystack mode; 2 GiB planted-needle mode; both once more with every wait on the stream instead of the pinned
    # answer words (SLICESLICE_SPIN_WAIT=0: what a service that must not busy-wait runs); and both against the product library
    for extra, env in (([], {"SLICESLICE_HIP_LIB": tuning}), (["2"], {"SLICESLICE_HIP_LIB": tuning}),
                       ([], {"SLICESLICE_SPIN_WAIT": "0", "SLICESLICE_HIP_LIB": tuning}),
                       (["2"], {"SLICESLICE_SPIN_WAIT": "0", "SLICESLICE_HIP_LIB": tuning}), ([], {}), (["2"], {})):
        out = subprocess.run([sys.executable, os.path.join(root, "tools", "fuzz_gpu.py"), "8", "4242"] + extra,
                             capture_output=True, text=True, timeout=600, cwd=root, env=dict(os.environ, **env))
        assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]
        d = json.loads(out.stdout.strip().splitlines()[-1])
        assert d["fuzz"] == "ok" and d["searches"] > 500, d
    out = subprocess.run([sys.executable, os.path.join(root, "tools", "fuzz_batched.py"), "8", "4242"],
                         capture_output=True, text=True, timeout=600, cwd=root)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]
    assert json.loads(out.stdout.strip().splitlines()[-1])["fuzz_batched"] == "ok"


def test_second_level_far_bytes_at_lane_piece_and_tile_edges(ss):
    """The second level looks up to 31 bytes behind the first filter byte: bytes 16..31 come from the next lane's chunk and
    the one after it (two cross-lane hops; for lanes 62 / 63 from the wave's next piece, or - after the wave's last
    piece - from the halo chunk / not at all).  A 40-byte needle is planted so that its candidate falls into lanes
    60..63 and 0..1 of the first and last piece of the first and last wave of a tile; exact copies must be found at their
    offset, copies with ONE byte changed (a near byte, far bytes on both sides of every hop, bytes past the window) must not."""
    ln = 1 << 20
    base = torch.full((ln + 64,), 0x2E, dtype=torch.uint8, device="cuda")
    needle = bytes(range(0x41, 0x41 + 40))
    nd = torch.from_numpy(np.frombuffer(needle, dtype=np.uint8).copy()).cuda()
    s = ss.DynamicHipSearcher.new(needle)
    for mis in (0, 7):
        hay = base[mis:mis + ln]
        for first, second, third in ((0, 5, 9), (3, 18, 11), (7, 8, 9)):
            s.set_filter(first, second, third)
            for tile in (0, 5):
                for wave in (0, 3):
                    for piece in (0, 3):
                        for lane in (60, 61, 62, 63, 0, 1):
                            # aligned coordinates are those of hay + first (16-byte grid of the device pointer)
                            chunk = ((tile * 4 + wave) * 4 + piece) * 64 + lane
                            phase = (hay.data_ptr() + first) & 15
                            at = chunk * 16 - phase - first + 3                 # candidate byte 3 of that chunk
                            if at < 0 or at + 40 > ln:
                                continue
                            hay[at:at + 40] = nd
                            assert s.search_in(hay) is True, (mis, first, tile, wave, piece, lane)
                            assert s.find(hay) == at, (mis, first, tile, wave, piece, lane)
                            for k in (1, 15, 16, 17, 19, 20, 27, 31, 32, 35, 39):
                                if k in (first, second, third):
                                    continue
                                hay[at + k] = 0x7E
                                assert s.search_in(hay) is False, (mis, first, tile, wave, piece, lane, k)
                                hay[at + k] = needle[k]
                            hay[at:at + 40] = 0x2E
        assert s.search_in(hay) is False


def _census_model(host, needle, filt, roles=None):
    """The census kernel's counts (aux_kernels.hpp: census_kernel; ss_census.hip) restated with numpy: 1,024 tiles of 4 KiB
    candidate offsets spread evenly, a 'lane' = 64 consecutive offsets.  `roles` given: also the per-position match counts
    gathered FOR that slot of `filt` (the kernel's pair = the other two slots): for every lane's leftmost pair candidate and leftmost
    triple candidate, at which of the needle's first 64 positions the haystack matches."""
    n, ln = len(needle), host.size
    end = ln - n + 1
    stride = ((end - 4 - 4096) // 1023) & ~4095
    nd = np.frombuffer(needle, dtype=np.uint8)
    if roles is None:
        fa, fb, fc = filt
    else:
        fa, fb, fc = filt[(roles + 1) % 3], filt[(roles + 2) % 3], filt[roles]
    ncheck = min(n, 64)
    tiles3 = tiles2 = match = lanes = 0
    pm, tm, pl, tl, deep = np.zeros(64, dtype=np.int64), np.zeros(64, dtype=np.int64), 0, 0, 0
    amin = min(fa, fb, fc)
    for k in range(1024):
        o = k * stride
        a, b, c = (host[o + f:o + f + 4096] == nd[f] for f in (fa, fb, fc))
        p2 = a & b
        p3 = p2 & c
        tiles2 += bool(p2.any())
        if roles is not None and p2.any():
            per2 = p2.reshape(64, 64)
            for l in np.nonzero(per2.any(axis=1))[0]:
                i = o + 64 * int(l) + int(np.argmax(per2[l]))
                pm[:ncheck] += host[i:i + ncheck] == nd[:ncheck]
                pl += 1
        if p3.any():
            tiles3 += 1
            per_lane = p3.reshape(64, 64)
            lanes += int(per_lane.any(axis=1).sum())
            m = False
            for l in np.nonzero(per_lane.any(axis=1))[0]:
                i = o + 64 * int(l) + int(np.argmax(per_lane[l]))
                eq = host[i:i + ncheck] == nd[:ncheck]
                m = m or bool(eq.all())
                tm[:ncheck] += eq
                tl += 1
                deep += bool(amin < ncheck and eq[amin:].all() and not eq.all())
            match += m
    counts = {"tiles": 1024, "tiles3": tiles3, "tiles2": tiles2, "match_tiles": match, "lanes": lanes}
    if roles is None:
        return counts
    return counts, {"pair_match": [int(x) for x in np.minimum(pm, 0xFFFF)], "triple_match": [int(x) for x in np.minimum(tm, 0xFFFF)],
                    "pair_lanes": pl, "triple_lanes": tl, "deep_lanes": deep}


def _settle(s, h, want_found=False, scans=16):
    """Scans until the handle has stopped looking at its bytes on this haystack (ss_searcher_tuning_state.settled)."""
    for _ in range(scans):
        assert s.search_in(h) is want_found
        st = s.tuning_state(h)
        if st["settled"] and not st["on_trial"]:
            break
    assert s.search_in(h) is want_found                     # (one more: the census a settling scan launched has arrived)
    st = s.tuning_state(h)
    assert st["settled"] == 1 and st["census_state"] == 2, st
    return st


def _rule(st):
    if st["match_tiles"]:
        return 4
    if st["tiles3"] >= 160 or st["lanes"] >= 1024 or st["deep_lanes"] >= 256:
        return 6
    return 5 if st["tiles3"] >= 56 or st["lanes"] >= 256 or st["deep_lanes"] >= 24 else 4


def _tiles_rule(st, default=1):
    """Tiles per workgroup of a single-stream launch that follows the census (ss_census.hip): two at four workgroups per CU from 28
    candidate tiles of 1,024, one at five and six while the candidates are thinly spread, else the launch's own choice (one below 2 GiB)."""
    if st["match_tiles"]:
        return default
    if _rule(st) == 5:
        return 1 if st["tiles3"] < 80 else default
    if _rule(st) == 6:
        return 1 if st["tiles3"] < 330 and st["lanes"] < 1024 and st["deep_lanes"] < 256 else 2
    return 2 if st["tiles3"] >= 28 else default


@pytest.mark.gpu
def test_workgroups_per_cu_follow_the_candidate_census(O):
    """four, five or six workgroups per CU (and one or two tiles per workgroup) is decided by what a census of the HAYSTACK counts (ss_census.hip;
    aux_kernels.hpp: census_kernel), not by the wall-clock time of earlier scans: deterministic for a given haystack and needle,
    nothing timed, observable through ss_searcher_last_launch and ss_searcher_tuning_state.  Hooks build for ss_debug_census.  The
    counts equal a numpy restatement of the sampling - for the searcher's own bytes after the first scan and for the bytes in force
    once the handle has settled (round 6: the census may move bytes the library owns to positions that let fewer candidates through);
    a text-like needle on RANDOM bytes starts at the needle-byte guess (six) and goes to four; scans below 256 MiB take no census; a
    searcher that has just FOUND its needle launches with four; new filter bytes mean a new census."""
    import sliceslice_rs_amd as ss
    gib = 1 << 30
    with ss.tuning_build():
        hay = torch.empty(gib, dtype=torch.uint8, device="cuda")
        ss.fill_random_device(hay, 0x5EED0001)
        raw = np.frombuffer(open(os.path.join(os.path.dirname(__file__), "golden", "data", "i386.txt"), "rb").read(), dtype=np.uint8)
        text_host = np.tile(raw, gib // raw.size + 1)[:gib].copy()
        text = torch.from_numpy(text_host).cuda()
        rnd_host = O.fill_random(gib, 0x5EED0001)
        for needle, h, host in ((b"there is not another one of these", hay, rnd_host),
                                (b"segment descriptor table entries are", text, text_host),
                                (b"privilege level zero!", text, text_host)):
            s = ss.DynamicHipSearcher.new(needle)
            assert s.search_in(h[: 1 << 20]) is False and s.census(h[: 1 << 20]) is None, "a 1 MiB scan takes no census"
            guess = s.last_launch()[0]
            assert s.census(h) is None
            # the FIRST scan of a (searcher, haystack) pair only leaves the pair's name (a searcher that comes once pays for no sampling);
            # the census is taken in front of the SECOND
            assert s.search_in(h) is False and s.census(h) is None and s.tuning_state(h)["census_state"] == 3
            assert s.search_in(h) is False
            assert s.last_launch()[0] == guess, "the first scans of a haystack go by the needle-byte guess"
            got = s.census(h)                               # the census ran in front of that scan: its counts are in
            assert got == _census_model(host, needle, s.filter3), (needle, got)
            assert s.device_filter == s.filter3
            st = _settle(s, h)
            counts = {k: st[k] for k in ("tiles", "tiles3", "tiles2", "match_tiles", "lanes")}
            assert counts["tiles3"] <= got["tiles3"] and counts["lanes"] <= got["lanes"], (got, st)       # bytes only move to fewer candidates
            model = _census_model(host, needle, st["in_force"])
            assert {k: counts[k] for k in ("tiles3", "match_tiles", "lanes")} == {k: model[k] for k in ("tiles3", "match_tiles", "lanes")}, (st, model)
            want = _rule(st)
            if h is hay:
                assert want == 4 and st["tiles3"] == 0
                # a filter that meets no candidates takes its COMPACT form (all three bytes within eight: the first phase's cheapest
                # windows) - on trial like every proposal, kept because the compact bytes meet no candidate on random bytes either
                # (proposed from a span of twelve on: the further dwords of the next lane's chunk are what it saves)
                own_span = max(st["own"]) - min(st["own"])
                assert st["own"] == list(s.filter3) and max(st["in_force"]) - min(st["in_force"]) <= (7 if own_span >= 12 else own_span), st
            picks = set()
            for _ in range(5):
                assert s.search_in(h) is False
                picks.add(s.last_launch()[0])
            assert picks == {want}, (needle, st, picks)
            per_wg = 16384 * _tiles_rule(st)
            assert s.last_launch()[1] in (gib // per_wg, gib // per_wg + 1), "16 KiB tiles: one per workgroup at 1 GiB, two where the census counted 28-55 candidate tiles"
            # the census's shape must not cost an answer: the needle planted flush against the end of the haystack, and across the
            # border of two 16 KiB tiles in the middle (one workgroup's two tiles where the shape says two), found by the launch
            # that follows the census - then taken out again
            nb = torch.from_numpy(np.frombuffer(needle, dtype=np.uint8).copy()).cuda()
            for at in (gib - len(needle), (gib // 2 // 32768) * 32768 + 16384 - 5, (gib // 3 // 32768) * 32768 + 32768 - 7):
                keep = h[at:at + len(needle)].clone()
                h[at:at + len(needle)] = nb
                assert s.search_in(h) is True and s.last_launch()[0] == want, (needle, at)     # (the launch BEFORE this one found nothing)
                assert s.find(h) == at
                h[at:at + len(needle)] = keep
                assert s.search_in(h) is False and s.search_in(h) is False and s.last_launch()[0] == want
            # a needle that is found: the next launch is at four, whatever the census says; absent again: back to the census
            h2 = h[: 300 << 20].clone()
            h2[12345:12345 + len(needle)] = torch.from_numpy(np.frombuffer(needle, dtype=np.uint8).copy()).cuda()
            assert s.search_in(h2) is True and s.find(h2) == 12345
            assert s.search_in(h2) is True and s.last_launch()[0] == 4
            del h2
            assert s.search_in(h) is False and s.search_in(h) is False and s.last_launch()[0] == want
            # new filter bytes: the old counts no longer describe the filter; an explicit triple is the caller's, nothing moves
            a, b, c = s.filter3
            s.set_filter(a, b, c)
            assert s.census(h) is None
            assert s.search_in(h) is False and s.census(h) is None          # (new bytes: a new pair - named by its first scan)
            assert s.search_in(h) is False and s.census(h) == got
            st2 = _settle(s, h)
            assert st2["in_force"] == [a, b, c] and st2["trials"] == 0 and st2["triple_state"] == 1
        # a buffer refilled IN PLACE: everything is looked at again every 256 scans, so the choice follows
        s = ss.DynamicHipSearcher.new(b"segment descriptor table entries are")
        s.set_filter(*s.filter3)                            # (the stock triple pinned: 209 candidate tiles in 1,024 on this text)
        for _ in range(4):
            assert s.search_in(text) is False
        assert s.last_launch()[0] == 6
        ss.fill_random_device(text, 0x5EED0001)            # the same bytes as `hay`: no candidates at all
        for _ in range(260):
            assert s.search_in(text) is False
        assert s.census(text) == _census_model(rnd_host, b"segment descriptor table entries are", s.filter3)
        assert s.search_in(text) is False and s.last_launch()[0] == 4


@pytest.mark.gpu
def test_a_filter_that_meets_no_candidates_takes_its_compact_form():
    """Where the census meets NO candidate the choice of filter bytes decides nothing but what the first phase costs, and that is
    least with all three bytes within eight (ss_census.hip, propose_compact; profiles/r06/headline_triple_probe_windows.jsonl): a
    `new`-built searcher whose own bytes span 12 or more moves them - on trial, kept because the compact bytes meet no candidate
    either - and answers as before; a caller's bytes (with_position: the byte; an explicit triple: all three) stay; with
    ss_set_autotune(0) nothing moves."""
    import sliceslice_rs_amd as ss
    gib = 1 << 30
    hay = torch.empty(gib, dtype=torch.uint8, device="cuda")
    ss.fill_random_device(hay, 0x5EED0001)
    nd = bytearray(ss.fill_random_host(16, 0x5EED0002).tobytes())
    nd[8] = 0xFF                                            # (0xFF never occurs in the generator's bytes: absent)
    nd = bytes(nd)
    s = ss.DynamicHipSearcher.new(nd)
    own = list(s.filter3)
    assert max(own) - min(own) >= 12, own                   # (the static choice: the reference's pair (0, n-1) + a third byte)
    st = _settle(s, hay, scans=24)
    assert st["tiles3"] == 0 and st["lanes"] == 0 and st["own"] == own and st["accepted"] >= 1, st
    assert max(st["in_force"]) - min(st["in_force"]) <= 7 and st["triple_state"] == 2, st
    assert list(s.filter3) == own, "ss_searcher_filter3 keeps reporting the searcher's own choice"
    # ... and the answers are what they were: planted flush against the end and across a tile border, found, leftmost offset right
    nb = torch.from_numpy(np.frombuffer(nd, dtype=np.uint8).copy()).cuda()
    for at in (gib - len(nd), (gib // 2 // 16384) * 16384 - 9, 5):
        keep = hay[at:at + len(nd)].clone()
        hay[at:at + len(nd)] = nb
        assert s.search_in(hay) is True and s.find(hay) == at, at
        hay[at:at + len(nd)] = keep
        assert s.search_in(hay) is False
    # a caller's byte stays: with_position(15) of a 16-byte needle fixes the span at 15 - no compact form
    wp = ss.DynamicHipSearcher.with_position(nd, 15)
    stw = _settle(wp, hay, scans=24)
    assert 0 in stw["in_force"] and 15 in stw["in_force"] and stw["proposal"] != 5, stw
    # an explicit triple is the caller's: nothing is put on trial
    ex = ss.DynamicHipSearcher.new(nd)
    ex.set_filter(*own)
    ste = _settle(ex, hay, scans=24)
    assert ste["in_force"] == own and ste["trials"] == 0, ste
    # launch tuning off: the static bytes
    was = ss.set_autotune(False)
    try:
        off = ss.DynamicHipSearcher.new(nd)
        for _ in range(8):
            assert off.search_in(hay) is False
        sto = off.tuning_state(hay)
        assert sto["in_force"] == own and sto["census_state"] == 0 and sto["trials"] == 0, sto
    finally:
        ss.set_autotune(was)


@pytest.mark.gpu
def test_census_measures_survival_and_moves_the_bytes_the_library_owns(O):
    """what a candidate costs is how deep it survives - measured, not modelled.  The census counts, per needle
    position, how many of the sampled pair / triple candidates MATCH the needle there (checked against a numpy restatement, for
    whichever slot the counts were gathered for); the library moves a byte IT owns to the position that lets the fewest candidates
    through (on trial against the next census), replaces the far byte of a pair 16 or more apart by near ones (one load stream), and
    orders the second level's schedule by the counts.  A caller's bytes stay; answers never change; ss_set_autotune(0) switches it
    all off."""
    import sliceslice_rs_amd as ss
    gib = 1 << 30
    with ss.tuning_build():
        raw = np.frombuffer(open(os.path.join(os.path.dirname(__file__), "golden", "data", "i386.txt"), "rb").read(), dtype=np.uint8)
        host = np.tile(raw, gib // raw.size + 1)[:gib].copy()
        text = torch.from_numpy(host).cuda()
        stock = b"\nSame exceptions as in Reel Address Mode"          # the manual says "Real": hundreds of candidates that survive 24 bytes
        assert stock not in raw.tobytes() and stock.replace(b"Reel", b"Real") in raw.tobytes()
        # (1) the counters themselves, first census: the searcher's own triple, gathered for slot 2
        s = ss.DynamicHipSearcher.new(stock)
        own = list(s.filter3)
        assert s.search_in(text) is False                   # (the first scan names the pair, the second takes the census)
        assert s.search_in(text) is False
        counts, stats = _census_model(host, stock, own, roles=2)
        assert s.census(text) == counts
        got = s.census_stats(text)
        assert got == stats, (got, stats)
        assert stats["triple_lanes"] >= 100 and min(stats["triple_match"][:len(stock)]) == 0
        killer = stats["triple_match"].index(0)
        assert stock[killer:killer + 1] == b"e" and killer == stock.index(b"Reel") + 2      # the byte that tells the needle from the manual's phrase
        # (2) ... the order of the second level follows them: the killer byte first (read off a searcher whose bytes are the caller's and
        #     stay: the same triple, named through ss_searcher_set_filter3)
        ex = ss.DynamicHipSearcher.new(stock)
        ex.set_filter(*own)
        assert ex.search_in(text) is False and ex.search_in(text) is False and ex.search_in(text) is False
        st = ex.tuning_state(text)
        assert st["order_measured"] == 1 and st["order"][0] == killer and st["in_force"] == own, st
        # (3) ... and the bytes move: settled, the filter in force meets (far) fewer candidates than the static one, by the same model
        st = _settle(s, text)
        assert st["triple_state"] == 2 and st["accepted"] >= 1 and st["in_force"] != own and st["own"] == own, st
        model = _census_model(host, stock, st["in_force"])
        assert (st["tiles3"], st["lanes"]) == (model["tiles3"], model["lanes"]) and st["tiles3"] * 4 <= counts["tiles3"], (st, counts)
        assert s.filter3 == tuple(own), "ss_searcher_filter3 keeps reporting the searcher's own choice"
        # (4) a caller's bytes stay: with_position keeps its byte, an explicit triple everything
        wp = ss.DynamicHipSearcher.with_position(stock, len(stock) - 1)
        st = _settle(wp, text)
        assert len(stock) - 1 in st["in_force"], st
        st = _settle(ex, text)
        assert st["in_force"] == own and st["trials"] == 0 and st["order_measured"] == 1, st
        # (5) the reference's pair (0, n-1) of a long needle: 16 or more apart -> the cross-lane kernels at first, then its near form
        rp = ss.DynamicHipSearcher.new(b" the quick brown fox ")
        rp.set_filter(0, 20)
        assert rp.search_in(text) is False
        rp.census(text)
        assert rp.last_mode in (2, 3)
        st = _settle(rp, text)
        assert st["in_force"][0] == 0 and max(st["in_force"]) <= 15 and st["kernel_mode"] == 0 and st["proposal"] in (2, 3), st
        assert rp.filter3[:2] == (0, 20)
        # (6) answers: planted, found and located by every one of them, tuned or not
        for needle, searchers in ((stock, (s, wp, ex)), (b" the quick brown fox ", (rp,))):
            h2 = text[: 300 << 20].clone()
            at = (299 << 20) + 12345
            h2[at:at + len(needle)] = torch.from_numpy(np.frombuffer(needle, dtype=np.uint8).copy()).cuda()
            for t in searchers:
                for _ in range(3):
                    assert t.search_in(h2) is True and t.find(h2) == at
                assert t.search_in(text) is False
            del h2
        # (7) the switch: off, a fresh searcher samples nothing and keeps the static choices; on again, it learns
        assert ss.set_autotune(False) is True
        try:
            off = ss.DynamicHipSearcher.new(stock)
            for _ in range(4):
                assert off.search_in(text) is False
            st = off.tuning_state(text)
            assert st["autotune"] == 0 and st["census_state"] == 0 and st["histogram_state"] in (0, 2) and st["in_force"] == own, st
        finally:
            assert ss.set_autotune(True) is False
        assert off.search_in(text) is False and off.tuning_state(text)["census_state"] >= 1


def _non_latin_haystack(n_bytes, seed):
    """UTF-8-like text in a non-Latin script: two-byte letters (lead 0xD0 / 0xD1, trail 0x80..0xBF with a skewed distribution),
    blanks, a few digits.  Every other byte is a lead byte - which the static rarity ranking takes for rare."""
    rng = np.random.default_rng(seed)
    pairs = n_bytes // 2
    lead = rng.choice(np.array([0xD0, 0xD1], dtype=np.uint8), size=pairs, p=[0.6, 0.4])
    trail = (0x80 + np.minimum(rng.geometric(0.08, size=pairs) - 1, 63)).astype(np.uint8)
    a = np.empty(pairs * 2, dtype=np.uint8)
    a[0::2], a[1::2] = lead, trail
    blanks = rng.integers(0, pairs, size=pairs // 7)
    a[2 * blanks] = 0x20
    a[2 * blanks + 1] = 0x20
    digits = rng.integers(0, pairs * 2, size=pairs // 200)
    a[digits] = rng.integers(0x30, 0x3A, size=digits.size).astype(np.uint8)
    return a


@pytest.mark.gpu
def test_filter_bytes_follow_the_haystacks_histogram(O):
    """Row f3 of SURVEY.md 8f without the caller asking: a searcher built by `new` starts with the static, corpus-free choice of filter
    bytes; on a haystack of 256 MiB or more the library samples the haystack's byte histogram next to the candidate census
    (ss_census.hip) and, when the histogram promises 16 x fewer candidates for other bytes of the needle, filters THIS haystack with
    those - on trial: if the census then counts MORE candidate tiles than the searcher's own triple had, the own triple stays.
    Answers never change; `filter3` keeps reporting the searcher's own choice; `with_position` keeps the caller's byte, an explicit triple everything."""
    import sliceslice_rs_amd as ss
    n_bytes = 512 << 20
    with ss.tuning_build():
        host = _non_latin_haystack(n_bytes, 7)
        hay = torch.from_numpy(host).cuda()
        word = bytes(host[1000:1016]) + b"e" + bytes(host[2000:2008])           # 'e' never occurs in this haystack
        assert b"e"[0] not in host[: 1 << 20]
        s = ss.DynamicHipSearcher.new(word)
        own = s.filter3
        assert 16 not in own, "the static ranking takes 'e' for the needle's most common byte"
        assert s.search_in(hay) is False and s.census(hay) is None    # the first scan of the pair only names it
        assert s.search_in(hay) is False                       # second scan: the searcher's own triple; census + histogram sampled
        first = s.census(hay)
        assert s.device_filter == own and first == _census_model(host, word, own) and first["tiles3"] > 48
        assert s.search_in(hay) is False                       # third scan: the histogram is in -> a triple with the 'e', on trial
        s.census(hay)
        assert 16 in s.device_filter and s.filter3 == own and s.triple_trials == 1 and s.triple_state == 2
        assert s.search_in(hay) is False                       # fourth: its own census is in
        settled = s.census(hay)
        model = _census_model(host, word, s.device_filter)
        assert {k: settled[k] for k in ("tiles3", "match_tiles", "lanes")} == {k: model[k] for k in ("tiles3", "match_tiles", "lanes")} and settled["tiles3"] == 0
        assert 16 in s.device_filter and s.last_launch()[0] == 4
        # answers: planted at the end, found; find() agrees; the caller's choices are not overridden
        host2 = host.copy()
        host2[n_bytes - len(word):] = np.frombuffer(word, dtype=np.uint8)
        hay2 = torch.from_numpy(host2).cuda()
        for _ in range(3):
            assert s.search_in(hay2) is True and s.find(hay2) == n_bytes - len(word)
        wp = ss.DynamicHipSearcher.with_position(word, len(word) - 1)
        sf = ss.DynamicHipSearcher.new(word)
        sf.set_filter(*own)
        for t in (wp, sf):
            for _ in range(3):
                assert t.search_in(hay) is False
            st = _settle(t, hay)
            assert len(word) - 1 in st["in_force"] or t is sf, "with_position keeps the caller's byte in the first phase"
            if t is sf:
                assert tuple(st["in_force"]) == t.filter3 and st["trials"] == 0, "an explicit triple is the caller's: nothing moves"
        del hay2

        # the trial: a byte that is rare overall but comes in RUNS.  The histogram prefers it, the census says no, the own triple stays.
        # (another length: the histogram is remembered per (pointer, length), and the allocator may hand out the same pointer again)
        n_bytes += 1 << 20
        rng = np.random.default_rng(11)
        letters = np.frombuffer(b"bcdfghjklmnopqrstuvwxyz", dtype=np.uint8)
        host = letters[rng.integers(0, letters.size, size=n_bytes)]
        runs = rng.integers(0, n_bytes // 64 - 1, size=n_bytes // 6400)
        for r in runs[:200000]:
            host[64 * r:64 * r + 64] = ord("e")
        hay = torch.from_numpy(host).cuda()
        word = b"eeeb" + bytes(letters[rng.integers(0, letters.size, size=12)])
        want = O.OracleSearcher(word).search_in(host[: 64 << 20])      # (a 16-byte needle of random letters: absent in practice)
        s = ss.DynamicHipSearcher.new(word)
        own = s.filter3
        assert not {0, 1, 2} & set(own), "the static ranking avoids 'e'"
        assert s.search_in(hay) is want and s.census(hay) is None      # (named)
        assert s.search_in(hay) is want
        first = s.census(hay)
        assert first == _census_model(host, word, own)
        assert s.search_in(hay) is want
        s.census(hay)                                       # (the trial is settled by the time a synchronous search has returned)
        assert s.triple_trials >= 1 and s.device_filter[:3] != (0, 1, 2), (s.triple_trials, s.triple_state, s.device_filter)
        st = _settle(s, hay, want_found=want)
        # whatever else the handle tried afterwards: what is in force meets no more candidates than the searcher's own bytes did - by the
        # census's count, which the model confirms - and the triple of the runs ('e', 'e', 'e') is not it
        model = _census_model(host, word, st["in_force"])
        assert (st["tiles3"], st["lanes"]) == (model["tiles3"], model["lanes"]), (st, model)
        assert st["tiles3"] <= first["tiles3"] and st["lanes"] <= first["lanes"] and sorted(st["in_force"]) != [0, 1, 2], (st, first)


@pytest.mark.gpu
@pytest.mark.timing
def test_first_big_search_does_not_wait_for_other_streams():
    """VERDICT r05 item 5c / ADVICE: "a call synchronises only the stream it was given" (include/sliceslice_hip.h).  The first search of
    >= 256 MiB of a process allocates the device's sampling scratch, and the second batched call that names a batch allocates its class
    table - both used to zero them with a NULL-STREAM hipMemset: a device-wide ordering point that waits for every blocking stream.
    Here a blocking stream holds ~100 ms of queued scans; a fresh searcher's first big search and the batched calls on ANOTHER stream
    return in a few milliseconds, answers right.  In a process of its own (the scratch is allocated once per process and device)."""
    code = r'''
import ctypes, os, sys, time
import numpy as np, torch
sys.path.insert(0, %r)
import sliceslice_rs_amd as ss
hip = ctypes.CDLL("libamdhip64.so")
blocking = ctypes.c_void_p()
assert hip.hipStreamCreate(ctypes.byref(blocking)) == 0          # default flags: a BLOCKING stream (synchronises with the null stream)
big = torch.empty(6 << 30, dtype=torch.uint8, device="cuda")
ss.fill_random_device(big, 1)
small = torch.empty(300 << 20, dtype=torch.uint8, device="cuda")
ss.fill_random_device(small, 2)
torch.cuda.synchronize()
L = ss.lib()
absent = bytes([0xFF] * 16)
busy = ss.DynamicHipSearcher.new(absent)
flag = torch.zeros(1, dtype=torch.int32, device="cuda")
hay_off = (torch.arange(257, dtype=torch.int64) * (1 << 20)).cuda()
nblob = torch.from_numpy(np.frombuffer(absent * 256, dtype=np.uint8).copy()).cuda()
nd_off = (torch.arange(257, dtype=torch.int64) * 16).cuda()
side = torch.cuda.Stream()
torch.cuda.synchronize()
def queue_work():
    for _ in range(120):                                          # ~0.85 ms each: ~100 ms of scans queued on the blocking stream
        assert L.ss_search_device_async(busy._h, big.data_ptr(), big.numel(), blocking, flag.data_ptr()) == 0
queue_work()
t0 = time.perf_counter()
with torch.cuda.stream(side):
    fresh = ss.DynamicHipSearcher.new(b"no such needle!!")
    r1 = fresh.search_in(small)                                   # (names the pair)
    r2 = fresh.search_in(small)                                   # first census of the process: census + histogram scratch allocated
    f1 = ss.search_batched(small, hay_off, nblob, nd_off)         # names the batch
    f2 = ss.search_batched(small, hay_off, nblob, nd_off)         # second call: the class table is allocated and sampled
    f3 = ss.search_batched(small, hay_off, nblob, nd_off)
    side.synchronize()
ms = (time.perf_counter() - t0) * 1e3
still_busy = hip.hipStreamQuery(blocking) != 0
assert hip.hipStreamSynchronize(blocking) == 0
total = (time.perf_counter() - t0) * 1e3
assert r1 is False and r2 is False and int(f1.sum().item()) == 0 and int(f3.sum().item()) == 0
print("RESULT", round(ms, 2), round(total, 2), still_busy)
''' % (os.path.dirname(os.path.dirname(os.path.abspath(__file__))),)
    import subprocess
    import sys
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-3000:]
    ms, total, still_busy = [l for l in out.stdout.splitlines() if l.startswith("RESULT")][-1].split()[1:]
    from conftest import timing_log
    timing_log("first_big_search_beside_a_busy_blocking_stream", calls_ms=float(ms), blocking_stream_drained_after_ms=float(total))
    assert still_busy == "True", "the blocking stream's work had ended before the calls returned: the test did not test anything"
    assert float(ms) < 0.5 * float(total), (ms, total)
