"""Pins the CPU oracle (oracle/sliceslice_oracle.c) against every known-answer vector the
reference's own tests hold for this path.  CPU only."""
import hashlib
import os
import random

import numpy as np
import pytest

from oracle import oracle as O

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def test_data_files_are_the_reference_ones(checksums):
    for name, want in checksums["sha256"].items():
        with open(os.path.join(GOLDEN, "data", name), "rb") as fh:
            assert hashlib.sha256(fh.read()).hexdigest() == want, name


@pytest.mark.parametrize("force_scalar", [False, True])
def test_generic_kats_every_position(kat, force_scalar):
    # reference src/lib.rs:370-381: result must equal the naive oracle for EVERY position in 0..n
    for row in kat["generic"]:
        hay, needle = row["haystack"].encode(), row["needle"].encode()
        assert O.naive_contains(hay, needle) == row["expected"], row
        for position in range(len(needle)):
            s = O.OracleSearcher.with_position(needle, position, force_scalar=force_scalar)
            assert s.search_in(hay) == row["expected"], (row, position)
        assert O.OracleSearcher(needle, force_scalar=force_scalar).search_in(hay) == row["expected"]


def test_memchr_kats(kat):
    # reference src/lib.rs:303-331
    for row in kat["memchr"]:
        hay, needle = row["haystack"].encode(), row["needle"].encode()
        assert O.OracleSearcher(needle).search_in(hay) == row["expected"], row
        assert O.naive_contains(hay, needle) == row["expected"], row


def test_constructor_contract(kat):
    # reference src/x86.rs:468-475, 533-543
    for row in kat["contract"]:
        needle = row["needle"].encode()
        if row["ok"]:
            O.OracleSearcher.with_position(needle, row["position"])
        else:
            with pytest.raises(O.OraclePositionError):
                O.OracleSearcher.with_position(needle, row["position"])


def test_empty_needle_and_empty_haystack():
    assert O.OracleSearcher(b"").search_in(b"") is True           # x86.rs:500
    assert O.OracleSearcher(b"").search_in(b"abc") is True
    assert O.OracleSearcher(b"a").search_in(b"") is False         # lib.rs:131-133
    assert O.OracleSearcher(b"ab").search_in(b"") is False
    assert O.OracleSearcher(b"ab").search_in(b"a") is False
    assert O.OracleSearcher(b"ab").search_in(b"ab") is True       # x86.rs:357-359
    assert O.OracleSearcher(b"ab").search_in(b"ba") is False


def test_short_haystack_sweep(corpus, checksums):
    # reference tests/i386.rs:46-59: 10,513,405 pairs; mode 2 = restated searcher vs naive on every pair
    words = sorted(corpus["words"], key=len)
    hits = O.sweep_short(words, mode=2)
    assert hits == checksums["short_haystack_hits"] == 39105


def test_long_haystack_sweep(corpus, checksums):
    # reference tests/i386.rs:61-70 (lossy UTF-8 haystack) and bench/benches/i386.rs:281-284 (raw bytes)
    raw = corpus["i386"]
    lossy = raw.decode("utf-8", errors="replace").encode("utf-8")
    assert len(lossy) == checksums["i386_lossy_len"]
    assert O.sweep_long(raw, corpus["words"], mode=2) == checksums["long_haystack_hits_raw"] == 4585
    assert O.sweep_long(lossy, corpus["words"], mode=2) == checksums["long_haystack_hits_lossy"] == 4585


def test_random_haystack_sweep(corpus, checksums):
    # the reference's THIRD criterion group, search_random_haystack (bench/benches/i386.rs:286-289 through :246-256): every word of
    # words.txt in data/haystack (1,000 bytes of noise) - the restated searcher against the naive count of make_golden.py, word by word
    hay = corpus["haystack"]
    assert len(hay) == checksums["random_haystack_len"] == 1000
    hit = [w for w in corpus["words"] if O.OracleSearcher(w).search_in(hay)]
    assert len(hit) == checksums["random_haystack_hits"] == 106
    assert sorted(w.decode("latin1") for w in hit) == checksums["random_haystack_hit_words"]
    assert O.sweep_long(hay, corpus["words"], mode=2) == checksums["random_haystack_hits"]     # (mode 2: restatement == naive on every word)


def test_random_grid(corpus, checksums):
    # bench/benches/random.rs:16 size grid over data/needle, data/haystack
    for row in checksums["random_grid"]:
        n, h = corpus["needle"][: row["needle_len"]], corpus["haystack"][: row["haystack_len"]]
        assert O.OracleSearcher(n).search_in(h) == row["expected"], row


def test_width_ladder_boundaries_vs_naive():
    # end = len - n + 1 straddling every rung of x86.rs:363-375 and the overlapped tail of lib.rs:276-284
    rng = random.Random(1234)
    for n in (2, 3, 4, 5, 8, 15, 16, 17, 31, 32, 33, 64, 65, 100):
        for end in (2, 3, 4, 5, 7, 8, 9, 15, 16, 17, 31, 32, 33, 47, 63, 64, 65, 95, 96, 97, 200):
            ln = end + n - 1
            for trial in range(6):
                hay = bytearray(rng.choice(b"ab") for _ in range(ln))
                needle = bytes(rng.choice(b"ab") for _ in range(n))
                if trial % 3 == 0:                       # plant at a boundary-ish offset
                    at = rng.choice([0, end - 1, max(0, end - 2), end // 2])
                    hay[at:at + n] = needle
                hay = bytes(hay)
                want = needle in hay
                assert O.naive_contains(hay, needle) == want
                for position in {0, n - 1, n // 2}:
                    for fs in (False, True):
                        got = O.OracleSearcher.with_position(needle, position, force_scalar=fs).search_in(hay)
                        assert got == want, (n, end, position, fs)


def test_multithreaded_baseline_agrees():
    hay = O.fill_random(1 << 20, 0x5EED0001)
    needle = bytes(hay[(1 << 19) + 5:(1 << 19) + 21])
    s = O.OracleSearcher(needle)
    assert s.search_in(hay) and s.search_in(hay, threads=4)
    absent = bytearray(needle)
    absent[8] = 0xFF
    s = O.OracleSearcher(bytes(absent))
    assert not s.search_in(hay) and not s.search_in(hay, threads=4)
    # match straddling a thread-shard edge
    h2 = np.zeros(1 << 20, dtype=np.uint8)
    edge = (1 << 20) // 4
    h2[edge - 7:edge + 9] = np.frombuffer(b"0123456789abcdef", dtype=np.uint8)
    assert O.OracleSearcher(b"0123456789abcdef").search_in(h2, threads=4)


def test_generator_has_no_ff_and_is_offset_consistent():
    a = O.fill_random(4096, 0x5EED0001)
    assert not (a == 0xFF).any()
    b = O.fill_random(1000, 0x5EED0001, global_offset=13)
    assert (a[13:1013] == b).all()
    assert O.have_avx2() in (True, False)


def test_fuzz_restatement_vs_naive():
    # 6,000 random (haystack, needle, position) triples over small alphabets (so that matches, near-matches
    # and dense candidates all occur); the restated AVX2 path must equal windows().any() on every one.
    rng = random.Random(2024)
    for case in range(6000):
        alpha = rng.choice([b"ab", b"abc", b"\x00\x01", bytes(range(256))])
        n = rng.choice([1, 2, 3, 4, 5, 8, 15, 16, 17, 24, 33, 64])
        ln = rng.choice([0, 1, n - 1, n, n + 1, n + 2, n + 7, n + 31, n + 32, n + 33, n + 100, 3 * n + 257])
        ln = max(ln, 0)
        hay = bytes(rng.choice(alpha) for _ in range(ln))
        if ln >= n and rng.random() < 0.3:
            at = rng.randrange(ln - n + 1)
            needle = hay[at:at + n]
        else:
            needle = bytes(rng.choice(alpha) for _ in range(n))
        want = needle in hay
        assert O.naive_contains(hay, needle) == want
        position = rng.randrange(n)
        assert O.OracleSearcher.with_position(needle, position, force_scalar=bool(case & 1)).search_in(hay) == want, \
            (case, n, ln, position)
