"""Parity of the HIP path (through the C ABI) with the CPU oracle and the reference's own vectors.

Every test here needs a real MI355X: run with ``pytest -m gpu``.  The comparisons are bit-exact
(booleans).  Reference citations are paths under /root/reference.
"""
import os
import random
import threading

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")


@pytest.fixture(scope="module")
def ss():
    import sliceslice_rs_amd as m
    assert torch.cuda.is_available(), "these tests must run on the GPU box"
    m.lib()                      # loads the in-tree HIP library; raises if it is missing
    return m


@pytest.fixture(scope="module")
def O():
    from oracle import oracle
    return oracle


def dev(b):
    """host bytes/ndarray -> uint8 device tensor"""
    a = np.frombuffer(b, dtype=np.uint8) if isinstance(b, (bytes, bytearray)) else np.asarray(b, dtype=np.uint8)
    if a.size == 0:
        return torch.empty(0, dtype=torch.uint8, device="cuda")
    return torch.from_numpy(a.copy()).cuda()


def test_library_is_the_in_tree_hip_build(ss):
    assert os.path.samefile(ss.library_path(), os.path.join(os.path.dirname(ss.__file__), "csrc", "libsliceslice_hip.so"))
    info = ss.device_info()
    assert "gfx950" in info["name"], info
    assert info["compute_units"] >= 200


def test_cross_lane_primitives(ss):
    out = [int(v) for v in ss.selftest_dpp()]
    want = [1000 + l + 1 for l in range(63)] + [0]
    assert out[0:64] == want            # DPP wave_shl:1 = "value of lane l+1"; lane 63 keeps `old` (0)
    assert out[64:128] == want          # the same via __shfl_down
    assert out[192:256] == [1000 + (l + 1) % 64 for l in range(64)]          # wave_rol:1
    assert out[256:320] == [1000 + l + 1 for l in range(63)] + [7777]        # lane 63 keeps `old`
    hi, lo = 0x44332211, 0xDDCCBBAA
    for l in range(64):
        r = l & 3
        assert out[128 + l] == (((hi << 32) | lo) >> (8 * r)) & 0xFFFFFFFF


def position_searchers(ss, needle, position):
    """with_position(needle, position) as the library runs it and - for a position 16 or more behind needle[0], where the
    library pairs the caller's byte with one close in front of it - the reference's own pair (needle[0], needle[position])
    too: the cross-lane kernel up to a distance of 1,007, the two-stream kernel beyond."""
    s = ss.DynamicHipSearcher.with_position(needle, position)
    if position >= 16:
        a, b, c = s.filter3
        assert b == position and 1 <= b - a <= 15 and (c == b or 1 <= c - a <= 15), (position, s.filter3)
        e = ss.DynamicHipSearcher.with_position(needle, position)
        e.set_filter(0, position)
        assert e.filter3 == (0, position, position)
        return [s, e]
    return [s]


def test_generic_kats_every_position(ss, kat):
    # src/lib.rs:370-381 - device-resident and host haystack paths
    for row in kat["generic"]:
        hay, needle = row["haystack"].encode(), row["needle"].encode()
        dh = dev(hay)
        for position in range(len(needle)):
            for s in position_searchers(ss, needle, position):
                assert s.search_in(dh) == row["expected"], (row, position, s.filter3)
        s = ss.DynamicHipSearcher.new(needle)
        assert s.search_in(dh) == row["expected"], row
        assert s.search_in(hay) == row["expected"], row         # ss_search_host


def test_memchr_kats(ss, kat):
    # src/lib.rs:303-331
    for row in kat["memchr"]:
        hay, needle = row["haystack"].encode(), row["needle"].encode()
        assert ss.DynamicHipSearcher.new(needle).search_in(dev(hay)) == row["expected"], row


def test_constructor_contract(ss, kat):
    # src/x86.rs:468-475, 533-543
    for row in kat["contract"]:
        needle = row["needle"].encode()
        if row["ok"]:
            ss.DynamicHipSearcher.with_position(needle, row["position"])
        else:
            with pytest.raises(ss.PositionError):
                ss.DynamicHipSearcher.with_position(needle, row["position"])


def test_avx2_and_memchr_searcher_counterparts(ss, kat):
    """The reference's two other searcher types on this path: Avx2Searcher (src/x86.rs:282-382; tests
    avx2_invalid_position, avx2_empty_needle at :533-549) and MemchrSearcher (src/lib.rs:119-142; KATs :303-331)."""
    with pytest.raises(ss.PositionError):
        ss.HipSearcher.with_position(b"foo", 3)                    # avx2_invalid_position
    with pytest.raises(ss.PositionError):
        ss.HipSearcher.new(b"")                                     # avx2_empty_needle
    assert ss.DynamicHipSearcher.new(b"").search_in(dev(b"abc")) is True        # the dynamic searcher's N0 arm instead
    for row in kat["generic"]:
        hay, needle = row["haystack"].encode(), row["needle"].encode()
        if needle:
            assert ss.HipSearcher.new(needle).search_in(dev(hay)) == row["expected"], row
            assert ss.HipSearcher.with_position(needle, len(needle) // 2).search_in(hay) == row["expected"], row
    for row in kat["memchr"]:
        hay, needle = row["haystack"].encode(), row["needle"].encode()
        assert len(needle) == 1
        m = ss.MemchrHipSearcher.new(needle[0])
        assert m.search_in(dev(hay)) == row["expected"] and m.search_in(hay) == row["expected"], row
    assert ss.MemchrHipSearcher.new(0x61).search_in(dev(b"")) is False          # lib.rs:131-133


def test_empty_needle_and_short_haystacks(ss):
    e = dev(b"")
    assert ss.DynamicHipSearcher.new(b"").search_in(e) is True          # x86.rs:500
    assert ss.DynamicHipSearcher.new(b"").search_in(dev(b"abc")) is True
    assert ss.DynamicHipSearcher.new(b"a").search_in(e) is False         # lib.rs:131-133
    assert ss.DynamicHipSearcher.new(b"ab").search_in(e) is False
    assert ss.DynamicHipSearcher.new(b"ab").search_in(dev(b"a")) is False
    assert ss.DynamicHipSearcher.new(b"ab").search_in(dev(b"ab")) is True  # x86.rs:357-359
    assert ss.DynamicHipSearcher.new(b"ab").search_in(dev(b"ba")) is False
    assert ss.DynamicHipSearcher.new(b"").search_in(b"") is True
    assert ss.DynamicHipSearcher.new(b"a").search_in(b"") is False


def test_long_haystack_sweep(ss, corpus, checksums):
    # tests/i386.rs:61-70 (lossy haystack) and bench/benches/i386.rs:281-284 (raw bytes): 4585/4585
    raw = corpus["i386"]
    lossy = raw.decode("utf-8", errors="replace").encode("utf-8")
    for hay, key in ((raw, "long_haystack_hits_raw"), (lossy, "long_haystack_hits_lossy")):
        dh = dev(hay)
        hits = sum(ss.DynamicHipSearcher.new(w).search_in(dh) for w in corpus["words"])
        assert hits == checksums[key] == 4585


def test_long_haystack_absent_words(ss, O, corpus):
    # the same corpus with needles that do NOT occur: candidate-heavy text, full scans
    dh = dev(corpus["i386"])
    rng = random.Random(7)
    for w in rng.sample(corpus["words"], 150):
        for needle in (w + b"\x00", b"\x00" + w, w[:1] + b"\x7f" + w[1:], w + b" qq"):
            want = needle in corpus["i386"]
            assert ss.DynamicHipSearcher.new(needle).search_in(dh) == want, needle
            p = rng.randrange(len(needle))
            for s in position_searchers(ss, needle, p):
                assert s.search_in(dh) == want, (needle, p, s.filter3)


def test_random_grid(ss, corpus, checksums):
    # bench/benches/random.rs:16
    for row in checksums["random_grid"]:
        n, h = corpus["needle"][: row["needle_len"]], corpus["haystack"][: row["haystack_len"]]
        assert ss.DynamicHipSearcher.new(n).search_in(dev(h)) == row["expected"], row


NEEDLE_LENS = [1, 2, 3, 4, 5, 7, 8, 9, 15, 16, 17, 31, 32, 33, 63, 64, 65, 100, 128, 255, 1000, 2048, 2049, 3000]


def test_boundary_sweep_vs_oracle(ss, O):
    """`end` across the reference's width ladder (x86.rs:363-375) and across lane (16 B), piece
    (63*16 B), tile and workgroup edges of the GPU kernel; every pointer misalignment 0..15."""
    rng = random.Random(99)
    big = torch.zeros(1 << 20, dtype=torch.uint8, device="cuda")
    ends = [1, 2, 3, 4, 7, 8, 15, 16, 17, 31, 32, 33, 63, 64, 65, 1007, 1008, 1009, 1024, 4031, 4032, 4033,
            16127, 16128, 16129, 64512, 64513, 100000]
    for n in NEEDLE_LENS:
        for end in rng.sample(ends, 9):
            ln = end + n - 1
            mis = rng.randrange(16)
            hay = np.frombuffer(bytes(rng.choice(b"abc") for _ in range(ln)), dtype=np.uint8).copy()
            needle = bytes(rng.choice(b"abc") for _ in range(n))
            if rng.random() < 0.5:
                at = rng.choice([0, end - 1, end // 2, max(0, end - 2), min(end - 1, 1007), min(end - 1, 1008)])
                hay[at:at + n] = np.frombuffer(needle, dtype=np.uint8)
            want = O.OracleSearcher(needle).search_in(hay)
            assert want == O.naive_contains(hay, needle)
            view = big[mis:mis + ln]
            view.copy_(torch.from_numpy(hay))
            for position in {0, n - 1, n // 2, rng.randrange(n)}:
                for s in position_searchers(ss, needle, position):
                    assert s.search_in(view) == want, (n, end, mis, position, s.filter3)


def test_single_match_at_every_kind_of_edge(ss):
    """One planted match in an otherwise needle-free haystack, at offsets that straddle lane, piece,
    tile and grid-stride edges, for first/last/middle filter positions."""
    ln = 3 << 20
    base = torch.full((ln + 64,), 0x2E, dtype=torch.uint8, device="cuda")
    for n in (2, 5, 16, 17, 40, 200):
        needle = bytes(range(65, 65 + min(n, 26))) + bytes(97 + (k % 26) for k in range(max(0, n - 26)))
        assert len(needle) == n
        nd = torch.from_numpy(np.frombuffer(needle, dtype=np.uint8).copy()).cuda()
        for mis in (0, 5):
            hay = base[mis:mis + ln]
            edges = [0, 1, 15, 16, 1007, 1008, 1009, 16 * 63 * 16 - 3, 16 * 63 * 16, (1 << 20) - 7, (2 << 20) + 1000,
                     ln - n - 1, ln - n]
            for at in edges:
                for shift in (0, -(n // 2), -(n - 1)):
                    a = at + shift
                    if a < 0 or a + n > ln:
                        continue
                    hay[a:a + n] = nd
                    for position in {0, n - 1, n // 2}:
                        for s in position_searchers(ss, needle, position):
                            assert s.search_in(hay) is True, (n, mis, a, position, s.filter3)
                    hay[a:a + n] = 0x2E
            assert ss.DynamicHipSearcher.new(needle).search_in(hay) is False


def test_no_read_or_match_beyond_len(ss):
    """GPU analogue of the reference's ASAN job (.github/workflows/check.yml:42-58): bytes just past
    `len` and just before the pointer are poisoned with the needle, and a copy of the needle straddles
    the end (starts in range, ends out of range); none of them may be reported."""
    for n in (1, 2, 16, 33):
        needle = bytes([0x51 + k for k in range(n)])
        nd = torch.from_numpy(np.frombuffer(needle, dtype=np.uint8).copy()).cuda()
        for ln in (n, n + 1, 100, 1008 * 16 + 3, 70000):
            for mis in (0, 1, 9, 15):
                for k in sorted({0, 1, n // 2, n - 1}):       # k needle bytes lie inside the haystack
                    buf = torch.full((48 + ln + 2 * n + 64,), 0x2E, dtype=torch.uint8, device="cuda")
                    lo = 48 + mis                               # haystack = buf[lo : lo+ln]
                    buf[lo - n:lo] = nd                         # needle right before the start
                    if k < n and ln >= k:
                        buf[lo + ln - k: lo + ln - k + n] = nd   # straddles (k > 0) or follows (k == 0) the end
                    hay = buf[lo:lo + ln]
                    assert ss.DynamicHipSearcher.new(needle).search_in(hay) is False, (n, ln, mis, k)
                    hay[ln - n:ln] = nd                          # a real match flush with the end
                    assert ss.DynamicHipSearcher.new(needle).search_in(hay) is True, (n, ln, mis, k)


def test_haystack_flush_against_end_of_allocation(ss):
    import ctypes
    hip = ctypes.CDLL("libamdhip64.so.7")
    size = 1 << 20
    p = ctypes.c_void_p()
    assert hip.hipMalloc(ctypes.byref(p), ctypes.c_size_t(size)) == 0
    try:
        assert hip.hipMemset(p, 0x2E, ctypes.c_size_t(size)) == 0
        for n in (2, 16, 100):
            needle = bytes([0x30 + (k % 40) for k in range(n)])
            for ln in (n, 4097, 65536 + 5):
                ptr = p.value + size - ln                     # last byte of haystack = last byte of allocation
                s = ss.DynamicHipSearcher.new(needle)
                assert s.search_in((ptr, ln)) is False
    finally:
        hip.hipFree(p)


def test_generator_device_equals_host_and_oracle(ss, O):
    for off, ln in ((0, 4096), (13, 100003), (8, 8), (3, 1), (1 << 33, 70001)):
        t = torch.empty(ln, dtype=torch.uint8, device="cuda")
        ss.fill_random_device(t, 0x5EED0001, off)
        a = t.cpu().numpy()
        assert (a == ss.fill_random_host(ln, 0x5EED0001, off)).all()
        assert (a == O.fill_random(ln, 0x5EED0001, off)).all()
        assert not (a == 0xFF).any()


def absent_needle(ss, n, seed=0x5EED0002):
    """generator bytes with one 0xFF byte (0xFF never occurs in the haystack)."""
    nd = bytearray(ss.fill_random_host(n, seed).tobytes())
    if n == 1:
        nd[0] = 0xFF
    elif n == 2:
        nd[1] = 0xFF
    else:
        nd[n // 2] = 0xFF
    return bytes(nd)


def test_synthetic_vs_oracle_medium(ss, O):
    ln = 32 << 20
    t = torch.empty(ln, dtype=torch.uint8, device="cuda")
    ss.fill_random_device(t, 0x5EED0001)
    host = t.cpu().numpy()
    for n in (1, 2, 4, 8, 16, 32, 128):
        nd = absent_needle(ss, n)
        assert O.OracleSearcher(nd).search_in(host) is False
        assert ss.DynamicHipSearcher.new(nd).search_in(t) is False, n
        # needles cut from the haystack: present
        for at in (0, 12345, ln - n):
            cut = host[at:at + n].tobytes()
            assert ss.DynamicHipSearcher.new(cut).search_in(t) is True, (n, at)
    # random two-byte needles: oracle decides
    rng = random.Random(5)
    for _ in range(40):
        n = rng.choice([2, 3, 4])
        nd = bytes(rng.randrange(255) for _ in range(n))
        assert ss.DynamicHipSearcher.new(nd).search_in(t) == O.OracleSearcher(nd).search_in(host), nd


def test_all_kernel_variants_agree():
    """Every kernel variant ss_searcher_set_variant can name x seven launch shapes against the oracle (tests/_variants_worker.py).
    The product library holds the 18 scan kernels the constructors and set_filter can select and no way to ask for others; the
    variants live in the tuning build (-DSS_TUNING_VARIANTS -DSS_TEST_HOOKS): the full list runs against that one, the automatic
    choice against the product library - where asking for a variant must be REFUSED, not silently ignored."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    worker = os.path.join(root, "tests", "_variants_worker.py")
    build = sys.modules["sliceslice_rs_amd._build"]
    out = subprocess.run([sys.executable, worker, "default"], capture_output=True, text=True, timeout=1200, cwd=root)
    assert out.returncode == 0 and "variants ok" in out.stdout, out.stdout[-2000:] + out.stderr[-3000:]
    env = dict(os.environ, SLICESLICE_HIP_LIB=build.build_tuning())
    out = subprocess.run([sys.executable, worker, "tuning"], capture_output=True, text=True, timeout=1800, cwd=root, env=env)
    assert out.returncode == 0 and "variants ok" in out.stdout, out.stdout[-2000:] + out.stderr[-3000:]


def test_candidate_heavy_inputs(ss, O):
    # adversarial: every offset passes the filter ("why position is caller-tunable")
    ln = 1 << 20
    a = torch.full((ln,), 0x61, dtype=torch.uint8, device="cuda")
    assert ss.DynamicHipSearcher.new(b"a" * 31 + b"b").search_in(a) is False
    assert ss.DynamicHipSearcher.with_position(b"a" * 31 + b"b", 5).search_in(a) is False
    assert ss.DynamicHipSearcher.new(b"ab" + b"a" * 30).search_in(a) is False
    assert ss.DynamicHipSearcher.new(b"a" * 32).search_in(a) is True
    a[ln - 1] = 0x62
    assert ss.DynamicHipSearcher.new(b"a" * 31 + b"b").search_in(a) is True
    # zero bytes + 0x01 bytes: the zero-byte trick's false positives must not leak into the result
    z = torch.zeros(1 << 16, dtype=torch.uint8, device="cuda")
    z[1::2] = 1
    host = z.cpu().numpy()
    for nd in (b"\x01\x01", b"\x00\x00", b"\x01\x00\x01\x01", b"\x01", b"\x02", b"\x00\x01" * 20):
        assert ss.DynamicHipSearcher.new(nd).search_in(z) == O.naive_contains(host, nd), nd


def test_host_haystack_path_chunking(ss, O):
    # ss_search_host stages 64 MiB chunks with n-1 bytes of carry: plant across the chunk edge
    ln = (64 << 20) + 4096
    host = np.zeros(ln, dtype=np.uint8)
    needle = bytes(range(1, 41))
    s = ss.DynamicHipSearcher.new(needle)
    assert s.search_in(host) is False
    at = (64 << 20) - 17
    host[at:at + 40] = np.frombuffer(needle, dtype=np.uint8)
    assert s.search_in(host) is True
    host[at:at + 40] = 0
    host[ln - 40:] = np.frombuffer(needle, dtype=np.uint8)
    assert s.search_in(host) is True


def test_concurrent_search_on_one_searcher(ss):
    t = torch.zeros(4 << 20, dtype=torch.uint8, device="cuda")
    t[-5:] = torch.tensor([9, 8, 7, 6, 5], dtype=torch.uint8)
    s_yes = ss.DynamicHipSearcher.new(bytes([9, 8, 7, 6, 5]))
    s_no = ss.DynamicHipSearcher.new(bytes([9, 8, 7, 6, 6]))
    errs = []

    def work():
        try:
            st = torch.cuda.Stream()
            with torch.cuda.stream(st):
                for _ in range(50):
                    assert s_yes.search_in(t) is True
                    assert s_no.search_in(t) is False
        except Exception as e:     # pragma: no cover
            errs.append(e)
    th = [threading.Thread(target=work) for _ in range(8)]
    [x.start() for x in th]
    [x.join() for x in th]
    assert not errs, errs


def test_concurrent_host_and_file_searches(ss, tmp_path):
    """The host/file front ends borrow ONE cached staging set per device; concurrent callers must fall back
    to private sets and still answer correctly (ctypes releases the GIL during the calls)."""
    rng = np.random.default_rng(9)
    hay = rng.integers(0, 255, size=(3 << 20) + 77, dtype=np.uint8)
    yes = hay[-40:-8].tobytes()
    no = bytearray(yes)
    no[5] = 0xFF
    path = tmp_path / "hay.bin"
    hay.tofile(path)
    s_yes, s_no = ss.DynamicHipSearcher.new(yes), ss.DynamicHipSearcher.new(bytes(no))
    errs = []

    def work(i):
        try:
            for _ in range(20):
                if i % 2:
                    assert s_yes.search_in(hay) is True
                    assert s_no.search_in(hay) is False
                    assert s_yes.find(hay) == hay.size - 40
                else:
                    assert ss.search_file(s_yes, str(path)) is True
                    assert ss.search_file(s_no, str(path)) is False
        except Exception as e:     # pragma: no cover
            errs.append(e)
    th = [threading.Thread(target=work, args=(i,)) for i in range(6)]
    [x.start() for x in th]
    [x.join() for x in th]
    assert not errs, errs


def test_batched_vs_oracle(ss, O):
    rng = random.Random(11)
    hays, needles, want = [], [], []
    for i in range(300):
        ln = rng.choice([0, 1, 5, 16, 100, 1008, 5000, 70000])
        n = rng.choice([0, 1, 2, 3, 8, 16, 17, 40])
        h = np.frombuffer(bytes(rng.choice(b"abcd") for _ in range(ln)), dtype=np.uint8).copy()
        nd = bytes(rng.choice(b"abcd") for _ in range(n))
        if ln >= n > 0 and rng.random() < 0.4:
            at = rng.randrange(ln - n + 1)
            h[at:at + n] = np.frombuffer(nd, dtype=np.uint8)
        hays.append(h)
        needles.append(nd)
        want.append(O.OracleSearcher(nd).search_in(h))
    hay_off = np.zeros(len(hays) + 1, dtype=np.int64)
    hay_off[1:] = np.cumsum([h.size for h in hays])
    nd_off = np.zeros(len(needles) + 1, dtype=np.int64)
    nd_off[1:] = np.cumsum([len(x) for x in needles])
    blob = torch.from_numpy(np.concatenate(hays + [np.zeros(1, dtype=np.uint8)])).cuda()
    nblob = torch.from_numpy(np.frombuffer(b"".join(needles) + b"\0", dtype=np.uint8).copy()).cuda()
    found = ss.search_batched(blob, torch.from_numpy(hay_off).cuda(), nblob, torch.from_numpy(nd_off).cuda())
    got = [bool(x) for x in found.cpu().tolist()]
    assert got == want


def test_batched_matches_straddling_slice_boundaries(ss):
    """Few problems -> many slices each (one tile = 16 KiB per slice): a needle planted across every kind of
    tile/slice edge of a misaligned haystack must be seen by exactly the slice on its left."""
    rng = np.random.default_rng(5)
    hays, needles, want = [], [], []
    for i in range(12):
        ln = 200_000 + 4097 * i
        h = rng.integers(0, 255, size=ln, dtype=np.uint8)          # 0xFF never occurs
        n = [16, 17, 33, 40, 5, 2][i % 6]
        nd = rng.integers(0, 255, size=n, dtype=np.uint8)
        nd[n // 2] = 0xFF                                            # absent unless planted
        mis = sum(x.size for x in hays) % 16                         # torch allocations are 16-byte aligned
        if i % 4 != 3:
            k = 1 + i                                                # tile edge k in aligned coordinates
            at = 16384 * k - mis - [1, n // 2, n - 1][i % 3]        # straddles the edge
            h[at:at + n] = nd
        hays.append(h)
        needles.append(nd.tobytes())
        want.append(needles[-1] in h.tobytes())
    assert want.count(True) == 9
    hay_off = np.zeros(len(hays) + 1, dtype=np.int64)
    hay_off[1:] = np.cumsum([h.size for h in hays])
    nd_off = np.zeros(len(needles) + 1, dtype=np.int64)
    nd_off[1:] = np.cumsum([len(x) for x in needles])
    blob = torch.from_numpy(np.concatenate(hays + [np.zeros(1, dtype=np.uint8)])).cuda()
    nblob = torch.from_numpy(np.frombuffer(b"".join(needles) + b"\0", dtype=np.uint8).copy()).cuda()
    found = ss.search_batched(blob, torch.from_numpy(hay_off).cuda(), nblob, torch.from_numpy(nd_off).cuda())
    assert [bool(x) for x in found.cpu().tolist()] == want


def test_short_haystack_sweep_sample_batched(ss, O, corpus):
    # tests/i386.rs:46-59 shape (word in word), a 200k-pair sample through the batched entry point
    words = sorted(corpus["words"], key=len)
    rng = random.Random(3)
    pairs = []
    for _ in range(200000):
        i = rng.randrange(len(words))
        j = rng.randrange(i, len(words))
        pairs.append((words[i], words[j]))
    want = [n in h for n, h in pairs]
    hay_off = np.zeros(len(pairs) + 1, dtype=np.int64)
    hay_off[1:] = np.cumsum([len(h) for _, h in pairs])
    nd_off = np.zeros(len(pairs) + 1, dtype=np.int64)
    nd_off[1:] = np.cumsum([len(n) for n, _ in pairs])
    blob = torch.from_numpy(np.frombuffer(b"".join(h for _, h in pairs), dtype=np.uint8).copy()).cuda()
    nblob = torch.from_numpy(np.frombuffer(b"".join(n for n, _ in pairs), dtype=np.uint8).copy()).cuda()
    found = ss.search_batched(blob, torch.from_numpy(hay_off).cuda(), nblob, torch.from_numpy(nd_off).cuda())
    got = [bool(x) for x in found.cpu().tolist()]
    assert got == want
    assert sum(got) == sum(want) > 0


def test_full_size_properties_1gib(ss):
    """BASELINE.json config 2 at full size: 1 GiB synthetic haystack, 16-byte needle, position 15.
    Size-independent properties instead of a CPU re-scan: absent by construction -> False; planted at
    len-16 -> True; planted across a 4 GiB/mid boundary of the grid-stride -> True; erased -> False."""
    ln = 1 << 30
    t = torch.empty(ln, dtype=torch.uint8, device="cuda")
    ss.fill_random_device(t, 0x5EED0001)
    nd = absent_needle(ss, 16)
    s = ss.DynamicHipSearcher.new(nd)
    assert s.position == 15
    assert s.search_in(t) is False
    present = bytes(ss.fill_random_host(16, 0x5EED0003).tobytes())
    assert 0xFF not in present
    sp = ss.DynamicHipSearcher.new(present)
    pn = torch.from_numpy(np.frombuffer(present, dtype=np.uint8).copy()).cuda()
    assert sp.search_in(t) is False
    for at in (ln - 16, (ln // 2) - 8, 0, 1008 * 12345 - 5):
        saved = t[at:at + 16].clone()
        t[at:at + 16] = pn
        assert sp.search_in(t) is True, at
        t[at:at + 16] = saved
        assert sp.search_in(t) is False
    assert s.search_in(t) is False


def test_short_haystack_sweep_full_pairs_kernel(ss, corpus, checksums):
    """tests/i386.rs:46-59 in full: every word searched in every word at or after it in length order -
    10,513,405 problems through ss_search_pairs, ranges aliasing into ONE copy of the words; the hit count
    is the committed checksum (39,105)."""
    words = sorted(corpus["words"], key=len)
    W = len(words)
    lens = np.array([len(w) for w in words], dtype=np.int64)
    starts = np.zeros(W, dtype=np.int64)
    starts[1:] = np.cumsum(lens)[:-1]
    blob = torch.from_numpy(np.frombuffer(b"".join(words), dtype=np.uint8).copy()).cuda()
    ni = np.repeat(np.arange(W, dtype=np.int64), W - np.arange(W))
    hj = np.concatenate([np.arange(i, W, dtype=np.int64) for i in range(W)])
    assert ni.size == checksums["short_haystack_pairs"] == 10513405
    nb, ne = torch.from_numpy(starts[ni]).cuda(), torch.from_numpy(starts[ni] + lens[ni]).cuda()
    hb, he = torch.from_numpy(starts[hj]).cuda(), torch.from_numpy(starts[hj] + lens[hj]).cuda()
    found = ss.search_batched(blob, None, blob, None, hay_ranges=(hb, he), needle_ranges=(nb, ne), pairs=True)
    assert int(found.sum().item()) == checksums["short_haystack_hits"] == 39105
    # spot-check individual answers (not only the count)
    got = found.cpu().numpy()
    rng = random.Random(1)
    for _ in range(20000):
        k = rng.randrange(ni.size)
        assert bool(got[k]) == (words[ni[k]] in words[hj[k]])
    # and the same answers from the workgroup-per-problem kernel on a slice of the problems
    sl = slice(0, 50000)
    f2 = ss.search_batched(blob, None, blob, None, hay_ranges=(hb[sl].contiguous(), he[sl].contiguous()),
                           needle_ranges=(nb[sl].contiguous(), ne[sl].contiguous()))
    assert (f2.cpu().numpy() == got[sl]).all()


def test_long_haystack_all_needles_one_launch(ss, corpus):
    """bench/benches/i386.rs:252-256 (`for searcher in &searchers { searcher.search_in(haystack) }`) as ONE
    launch: 4,585 needles, all ranges aliasing the same 857 kB haystack; plus absent variants."""
    hay = corpus["i386"]
    words = list(corpus["words"])
    absent = [w + b"\x00" for w in words[:500]]
    needles = words + absent
    want = [w in hay for w in needles]
    assert sum(want) == 4585
    lens = np.array([len(w) for w in needles], dtype=np.int64)
    nb = np.zeros(len(needles), dtype=np.int64)
    nb[1:] = np.cumsum(lens)[:-1]
    nblob = torch.from_numpy(np.frombuffer(b"".join(needles), dtype=np.uint8).copy()).cuda()
    dh = dev(hay)
    hb = torch.zeros(len(needles), dtype=torch.int64, device="cuda")
    he = torch.full((len(needles),), len(hay), dtype=torch.int64, device="cuda")
    found = ss.search_batched(dh, None, nblob, None, hay_ranges=(hb, he),
                              needle_ranges=(torch.from_numpy(nb).cuda(), torch.from_numpy(nb + lens).cuda()))
    assert [bool(x) for x in found.cpu().tolist()] == want


def test_random_haystack_sweep(ss, corpus, checksums):
    """The reference's THIRD criterion group, search_random_haystack (bench/benches/i386.rs:286-289): the 4,585 words in
    data/haystack (1,000 bytes of noise).  As ONE batched call (4,585 ranges aliasing the same 1,000 bytes), through a plan, and - on a
    100-word sample that holds every hit-or-miss kind - one ss_search_device per needle, the reference's own loop shape
    (bench/benches/i386.rs:252-256).  Expected: the naive count of tests/golden/make_golden.py, word by word."""
    hay = corpus["haystack"]
    words = list(corpus["words"])
    want = [w in hay for w in words]
    assert sum(want) == checksums["random_haystack_hits"] == 106
    assert sorted(w.decode("latin1") for w, h in zip(words, want) if h) == checksums["random_haystack_hit_words"]
    lens = np.array([len(w) for w in words], dtype=np.int64)
    nb = np.zeros(len(words), dtype=np.int64)
    nb[1:] = np.cumsum(lens)[:-1]
    nblob = torch.from_numpy(np.frombuffer(b"".join(words), dtype=np.uint8).copy()).cuda()
    dh = dev(hay)
    hb = torch.zeros(len(words), dtype=torch.int64, device="cuda")
    he = torch.full((len(words),), len(hay), dtype=torch.int64, device="cuda")
    nbd, ned = torch.from_numpy(nb).cuda(), torch.from_numpy(nb + lens).cuda()
    for _ in range(3):                                   # (the second and third call go by the batch's sampled classes)
        found = ss.search_batched(dh, None, nblob, None, hay_ranges=(hb, he), needle_ranges=(nbd, ned))
        assert [bool(x) for x in found.cpu().tolist()] == want
    plan = ss.BatchPlan(dh, None, nblob, None, hay_ranges=(hb, he), needle_ranges=(nbd, ned))
    flags = torch.empty(len(words), dtype=torch.int32, device="cuda")
    for _ in range(2):
        plan.run(flags)
        assert [bool(x) for x in flags.cpu().tolist()] == want
    plan.close()
    at = ss.find_batched(dh, None, nblob, None, hay_ranges=(hb, he), needle_ranges=(nbd, ned)).cpu().tolist()
    assert at == [hay.find(w) for w in words]            # (SS_NPOS reads as -1, which is what bytes.find says too)
    rng = random.Random(3)
    sample = [w for w, h in zip(words, want) if h][:50] + rng.sample([w for w, h in zip(words, want) if not h], 50)
    for w in sample:
        assert ss.DynamicHipSearcher.new(w).search_in(dh) is (w in hay), w
        assert ss.DynamicHipSearcher.with_position(w, 0).search_in(dh) is (w in hay), w


def test_find_leftmost_vs_python(ss, corpus):
    """Row f1: offset of the leftmost occurrence (tests/i386.rs:6-10 `find_subsequence`), bit-exact vs
    Python's bytes.find on text, boundaries, repeated matches and random data."""
    raw = corpus["i386"]
    dh = dev(raw)
    rng = random.Random(21)
    for w in rng.sample(corpus["words"], 400) + [b"", b"\x00\x01\x02", b"zzzzzzzzzzzz"]:
        want = raw.find(w)
        assert ss.DynamicHipSearcher.new(w).find(dh) == (None if want < 0 else want), w
    # many occurrences: the leftmost must win regardless of which workgroup sees which first
    ln = 8 << 20
    t = torch.full((ln + 16,), 0x2E, dtype=torch.uint8, device="cuda")
    for n in (1, 2, 16, 33, 300):
        needle = bytes((37 * k + 11) % 200 + 50 for k in range(n))
        nd = torch.from_numpy(np.frombuffer(needle, dtype=np.uint8).copy()).cuda()
        for mis in (0, 7):
            hay = t[mis:mis + ln]
            s = ss.DynamicHipSearcher.new(needle)
            assert s.find(hay) is None
            spots = sorted(rng.sample(range(0, ln - n), 40) + [ln - n])
            for at in reversed(spots):                 # plant right-to-left; the answer moves left each time
                hay[at:at + n] = nd
                got = s.find(hay)
                assert got is not None and got <= at
                hb = hay[max(0, got - 1):got + n].cpu().numpy().tobytes()
                assert needle in hb
            assert s.find(hay) == hay.cpu().numpy().tobytes().find(needle)
            hay.fill_(0x2E)
    # boundary offsets / positions
    for n in (2, 5, 16, 17, 64):
        needle = bytes(range(100, 100 + n))
        nd = torch.from_numpy(np.frombuffer(needle, dtype=np.uint8).copy()).cuda()
        hay = t[3:3 + 200000]
        hay.fill_(0x2E)
        for at in (0, 1, 15, 16, 1023, 1024, 1025, 16383, 16384, 65535, 65536, 200000 - n):
            hay[at:at + n] = nd
            for position in {0, n - 1, n // 2}:
                for s in position_searchers(ss, needle, position):
                    assert s.find(hay) == at, (n, at, position, s.filter3)
            hay[at:at + n] = 0x2E
    # random data, short needles: first occurrence somewhere in the middle
    r = torch.empty(4 << 20, dtype=torch.uint8, device="cuda")
    ss.fill_random_device(r, 77)
    host = r.cpu().numpy().tobytes()
    for _ in range(60):
        n = rng.choice([1, 2, 3])
        nd = bytes(rng.randrange(255) for _ in range(n))
        want = host.find(nd)
        assert ss.DynamicHipSearcher.new(nd).find(r) == (None if want < 0 else want), nd


def test_search_file_front_end(ss, corpus, tmp_path):
    # examples/grep.rs:42-56: "./grep <backend> <needle> <file>" - map the file, one search_in
    path = tmp_path / "i386.txt"
    path.write_bytes(corpus["i386"])
    for w in (b"Zz", b"privilege", b"PREFACE", b"not to be found anywhere in the manual"):
        assert ss.search_file(ss.DynamicHipSearcher.new(w), str(path)) == (w in corpus["i386"])
    empty = tmp_path / "empty"
    empty.write_bytes(b"")
    assert ss.search_file(ss.DynamicHipSearcher.new(b""), str(empty)) is True
    assert ss.search_file(ss.DynamicHipSearcher.new(b"x"), str(empty)) is False
    with pytest.raises(ss.SlicesliceError):
        ss.search_file(ss.DynamicHipSearcher.new(b"x"), str(tmp_path / "missing"))
    # a file of several 64 MiB chunks: matches straddling chunk edges rely on the n-1 byte carry
    big = tmp_path / "big.bin"
    ln = (3 * 64 << 20) + 12345
    host = np.full(ln, 0x2E, dtype=np.uint8)
    needle = bytes(range(1, 41))
    s = ss.DynamicHipSearcher.new(needle)
    host.tofile(str(big))
    assert ss.search_file(s, str(big)) is False
    for at in ((64 << 20) - 17, (128 << 20) - 39, (128 << 20) - 1, (192 << 20), ln - 40, 0):
        h2 = host.copy()
        h2[at:at + 40] = np.frombuffer(needle, dtype=np.uint8)
        h2.tofile(str(big))
        assert ss.search_file(s, str(big)) is True, at
    host[ln - 39:] = np.frombuffer(needle[:39], dtype=np.uint8)      # only a 39-byte prefix at the very end
    host.tofile(str(big))
    assert ss.search_file(s, str(big)) is False


def test_histogram_and_position_policy(ss, corpus):
    raw = np.frombuffer(corpus["i386"], dtype=np.uint8)
    usable = raw[: (raw.size // 16) * 16]
    hist = ss.byte_histogram(dev(corpus["i386"]))
    assert (hist == np.bincount(usable, minlength=256).astype(np.uint64)).all()
    sampled = ss.byte_histogram(dev(corpus["i386"]), sample_bytes=100000)
    assert 50000 <= int(sampled.sum()) <= 200000
    # misaligned device pointer
    t = torch.zeros(raw.size + 8, dtype=torch.uint8, device="cuda")
    t[3:3 + raw.size] = torch.from_numpy(raw.copy()).cuda()
    h2 = ss.byte_histogram(t[3:3 + raw.size])
    assert (h2 == hist).all()
    needle = b" the quick brown fox "
    pos = ss.choose_position(needle, hist)
    assert pos >= 1 and hist[needle[pos]] == min(hist[b] for b in needle[1:])
    assert needle[pos:pos + 1] in (b"q", b"x", b"k", b"w")
    assert ss.choose_position(needle) == len(needle) - 1            # reference default (x86.rs:285)
    assert ss.choose_position(b"a", hist) == 0 and ss.choose_position(b"", hist) == 0
    # the policy changes speed, never the answer (lib.rs:375-378)
    dh = dev(corpus["i386"])
    assert ss.DynamicHipSearcher.with_position(needle, pos).search_in(dh) == (needle in corpus["i386"])


def test_flag_slots_are_reusable_without_reset(ss):
    """One searcher, alternating haystacks that do / do not contain the needle, device and host paths:
    a stale 'found' value in a reused flag slot would show up as a wrong True."""
    needle = b"slot-reuse-check"
    yes = torch.full((300000,), 0x2E, dtype=torch.uint8, device="cuda")
    no = yes.clone()
    yes[123456:123456 + 16] = torch.from_numpy(np.frombuffer(needle, dtype=np.uint8).copy()).cuda()
    s = ss.DynamicHipSearcher.new(needle)
    hy, hn = yes.cpu().numpy(), no.cpu().numpy()
    for k in range(200):
        assert s.search_in(yes) is True
        assert s.search_in(no) is False
        if k % 20 == 0:
            assert s.search_in(hy) is True and s.search_in(hn) is False
            assert s.find(yes) == 123456 and s.find(no) is None


def test_caller_owned_flags_are_not_seen_stale(ss):
    """Workgroups peek at the flag / best offset through the (non-coherent) scalar cache before they load
    anything.  A caller-owned flag that held "found" in the previous launch and was reset by the caller must
    not be seen stale by the next launch, or late workgroups would leave without scanning their tiles."""
    n_bytes = 192 << 20                                   # 6,144 two-tile workgroups; the peek starts at 1,024
    hay = torch.empty(n_bytes, dtype=torch.uint8, device="cuda")
    ss.fill_random_device(hay, 0xABCD01)
    early = hay[4096:4096 + 16].cpu().numpy().tobytes()
    late = hay[n_bytes - 5000:n_bytes - 5000 + 16].cpu().numpy().tobytes()
    absent = bytearray(late)
    absent[7] = 0xFF
    s_early, s_late, s_abs = (ss.DynamicHipSearcher.new(x) for x in (early, late, bytes(absent)))
    flag = torch.zeros(1, dtype=torch.int32, device="cuda")
    best = torch.empty(1, dtype=torch.int64, device="cuda")
    for _ in range(5):
        flag.zero_()
        s_early.search_in_async(hay, flag)
        assert int(flag.item()) == 1
        flag.zero_()
        s_late.search_in_async(hay, flag)                 # same flag address, match in one of the last workgroups
        assert int(flag.item()) == 1
        flag.zero_()
        s_abs.search_in_async(hay, flag)
        assert int(flag.item()) == 0
        best.fill_(-1)                                    # all ones = none yet
        s_early.find_async(hay, best)
        assert int(best.item()) == 4096
        best.fill_(-1)
        s_late.find_async(hay, best)
        assert int(best.item()) == n_bytes - 5000
        best.fill_(-1)
        s_abs.find_async(hay, best)
        assert int(best.item()) == -1


def test_async_entry_points_capture_into_a_graph(ss):
    """ss_search_device_async / ss_find_device_async only enqueue, so a launch-bound loop of searches can be
    captured once into a hipGraph and replayed."""
    hay = torch.empty(48 << 20, dtype=torch.uint8, device="cuda")
    ss.fill_random_device(hay, 0x77)
    at = hay.numel() - 1000
    yes = hay[at:at + 16].cpu().numpy().tobytes()
    no = bytearray(yes)
    no[3] = 0xFF
    s_yes, s_no = ss.DynamicHipSearcher.new(yes), ss.DynamicHipSearcher.new(bytes(no))
    flags = torch.zeros(2, dtype=torch.int32, device="cuda")
    best = torch.full((1,), -1, dtype=torch.int64, device="cuda")
    side = torch.cuda.Stream()
    with torch.cuda.stream(side):                           # first use outside the capture
        s_yes.search_in_async(hay, flags[0:1])
        s_no.search_in_async(hay, flags[1:2])
        s_yes.find_async(hay, best)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        flags.zero_()
        best.fill_(-1)
        s_yes.search_in_async(hay, flags[0:1])
        s_no.search_in_async(hay, flags[1:2])
        s_yes.find_async(hay, best)
    for _ in range(3):
        flags.fill_(7)
        best.fill_(5)
        g.replay()
        torch.cuda.synchronize()
        assert flags.tolist() == [1, 0] and int(best.item()) == at


def test_offsets_beyond_4gib_and_long_needles(ss):
    """64-bit offsets: matches planted beyond 2^32 and 2^33 in a 9 GiB haystack (search_in and find), and
    needles far longer than the 2 KiB LDS slice / longer than a tile (compare continues from global)."""
    free_b, _ = torch.cuda.mem_get_info()
    ln = 9 << 30
    if free_b < ln + (2 << 30):
        pytest.skip("not enough device memory")
    t = torch.empty(ln, dtype=torch.uint8, device="cuda")
    ss.fill_random_device(t, 0x5EED0001)
    needle = bytes(ss.fill_random_host(16, 0x5EED0003).tobytes())
    s = ss.DynamicHipSearcher.new(needle)
    nd = torch.from_numpy(np.frombuffer(needle, dtype=np.uint8).copy()).cuda()
    assert s.search_in(t) is False and s.find(t) is None
    for at in ((1 << 32) - 8, (1 << 32) + 5, (1 << 33) + 123457, ln - 16):
        saved = t[at:at + 16].clone()
        t[at:at + 16] = nd
        assert s.search_in(t) is True, at
        assert s.find(t) == at, at
        t[at:at + 16] = saved
    # long needles cut from the haystack itself (present) and with one byte flipped (absent)
    for n in (5000, 70000, 1 << 20):
        at = (5 << 30) + 777
        cut = t[at:at + n].cpu().numpy().tobytes()
        sl = ss.DynamicHipSearcher.new(cut)
        assert sl.search_in(t) is True and sl.find(t) == at, n
        bad = bytearray(cut)
        bad[n - 3] ^= 0x55
        for sb in position_searchers(ss, bytes(bad), n // 2):
            assert sb.search_in(t) is False, (n, sb.filter3)
    del t


def test_full_size_properties_64gib(ss):
    """BASELINE.json's target size on one GPU: 64 GiB, 16-byte needle.  Absent by construction -> False;
    planted at len-16, across the middle, at 0 -> True / exact offset; erased -> False again."""
    free_b, _ = torch.cuda.mem_get_info()
    ln = 64 << 30
    if free_b < ln + (4 << 30):
        pytest.skip("not enough device memory for the 64 GiB case")
    t = torch.empty(ln, dtype=torch.uint8, device="cuda")
    ss.fill_random_device(t, 0x5EED0001)
    absent = absent_needle(ss, 16)
    assert ss.DynamicHipSearcher.new(absent).search_in(t) is False
    present = bytes(ss.fill_random_host(16, 0x5EED0003).tobytes())
    sp = ss.DynamicHipSearcher.new(present)
    pn = torch.from_numpy(np.frombuffer(present, dtype=np.uint8).copy()).cuda()
    for at in (ln - 16, (ln // 2) - 8, 0):
        saved = t[at:at + 16].clone()
        t[at:at + 16] = pn
        assert sp.search_in(t) is True, at
        assert sp.find(t) == at, at
        t[at:at + 16] = saved
        assert sp.search_in(t) is False
    # the generator is the same logical haystack whatever the shard: spot-check against the host generator
    for off in (0, (1 << 35) + 4096 - 8, ln - 4096):
        assert (t[off:off + 4096].cpu().numpy() == ss.fill_random_host(4096, 0x5EED0001, off)).all()
    del t


def test_fuzz_batched_and_pairs_vs_python(ss):
    """30,000 random problems (small alphabets, planted and near-miss needles, random `position`s, ragged
    lengths incl. empty needles/haystacks) through BOTH many-problem kernels in one launch each."""
    rng = random.Random(777)
    hays, needles, positions, want = [], [], [], []
    for _ in range(30000):
        alpha = rng.choice([b"ab", b"abc", b"\x00\x01", b"the quick brown fox ", bytes(range(256))])
        n = rng.choice([0, 1, 2, 3, 4, 5, 8, 15, 16, 17, 24, 33, 64, 100, 300 if rng.random() < 0.2 else 40,
                        1100 if rng.random() < 0.1 else 20])
        ln = rng.choice([0, 1, max(n - 1, 0), n, n + 1, n + 7, n + 33, 2 * n + 100, 1500, 20000 if rng.random() < 0.05 else 300])
        hay = bytes(rng.choice(alpha) for _ in range(ln))
        if n and ln >= n and rng.random() < 0.3:
            at = rng.randrange(ln - n + 1)
            nd = bytearray(hay[at:at + n])
            if rng.random() < 0.3:
                nd[rng.randrange(n)] ^= 1                # near miss (may still occur elsewhere: Python decides)
            nd = bytes(nd)
        else:
            nd = bytes(rng.choice(alpha) for _ in range(n))
        hays.append(hay)
        needles.append(nd)
        positions.append(rng.randrange(n) if n else 0)
        want.append(nd in hay)
    hay_off = np.zeros(len(hays) + 1, dtype=np.int64)
    hay_off[1:] = np.cumsum([len(h) for h in hays])
    nd_off = np.zeros(len(needles) + 1, dtype=np.int64)
    nd_off[1:] = np.cumsum([len(x) for x in needles])
    blob = torch.from_numpy(np.frombuffer(b"".join(hays) + b"\0", dtype=np.uint8).copy()).cuda()
    nblob = torch.from_numpy(np.frombuffer(b"".join(needles) + b"\0", dtype=np.uint8).copy()).cuda()
    ho, no = torch.from_numpy(hay_off).cuda(), torch.from_numpy(nd_off).cuda()
    pos = torch.from_numpy(np.array(positions, dtype=np.int64)).cuda()
    for pairs in (False, True):
        for p in (None, pos):
            found = ss.search_batched(blob, ho, nblob, no, position=p, pairs=pairs)
            got = [bool(x) for x in found.cpu().tolist()]
            bad = [k for k in range(len(want)) if got[k] != want[k]]
            assert not bad, (pairs, p is not None, bad[:5], [(hays[k][:40], needles[k]) for k in bad[:2]])


def test_find_on_host_buffers(ss, corpus):
    raw = corpus["i386"]
    for w in (b"Zz", b"privilege", b"PREFACE", b"", b"not in the manual at all"):
        want = raw.find(w)
        assert ss.DynamicHipSearcher.new(w).find(raw) == (None if want < 0 else want), w
    # several 64 MiB chunks: leftmost of many, straddling a chunk edge
    ln = (2 * 64 << 20) + 999
    host = np.full(ln, 0x2E, dtype=np.uint8)
    needle = bytes(range(1, 41))
    s = ss.DynamicHipSearcher.new(needle)
    assert s.find(host) is None
    for at in (ln - 40, (128 << 20) - 7, (64 << 20) - 17, 12345):          # right to left
        host[at:at + 40] = np.frombuffer(needle, dtype=np.uint8)
        assert s.find(host) == at, at
