#!/usr/bin/env python3
"""Regenerates the golden fixtures under tests/golden/.

Run from the repo root in the BUILD container (needs /root/reference):

    python tests/golden/make_golden.py

What it writes
--------------
* ``data/{i386.txt,words.txt,haystack,needle}`` - the data files the reference's
  own tests and benches read (reference ``tests/i386.rs:3-4``,
  ``bench/benches/i386.rs:17,282,287``, ``bench/benches/random.rs:13-14``),
  copied byte-for-byte (MIT, (c) 2020 Cloudflare, Inc. - see data/ATTRIBUTION).
* ``kat.json`` - the known-answer vectors of the reference's unit tests, as
  (family, haystack, needle, expected) rows.  Inputs and expected booleans are
  the ones asserted at reference ``src/lib.rs:303-331`` (memchr KATs) and
  ``src/lib.rs:422-544`` (six generic families, instantiated for
  ``DynamicAvx2Searcher`` at ``src/x86.rs:601-611``).  Each generic row is
  asserted by the reference for EVERY ``position in 0..needle.len()``
  (``src/lib.rs:375-378``); the tests here do the same.
* ``corpus_checksums.json`` - hit counts of the two corpus sweeps of
  ``tests/i386.rs:46-70``, of the third criterion group's loop (the 4,585 words
  in ``data/haystack``, ``bench/benches/i386.rs:286-289``) and of the
  ``bench/benches/random.rs:16`` size grid,
  computed with Python's ``bytes.__contains__`` (a naive, independent scan; no
  reference code involved), plus sha256 of the data files.

No reference source text is stored: only inputs and expected outputs.
"""
import hashlib
import json
import os
import shutil
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference"

LOREM55 = "Lorem ipsum dolor sit amet, consectetur adipiscing elit"
LOREM187 = (
    "Lorem ipsum dolor sit amet, consectetur adipiscing elit. Maecenas commodo "
    "posuere orci a consectetur. Ut mattis turpis ut auctor consequat. Aliquam "
    "iaculis fringilla mi, nec aliquet purus"
)
assert len(LOREM55) == 55 and len(LOREM187) == 187

# (family, haystack, needle, expected) - expected is what the reference asserts.
GENERIC = [
    # search_same, src/lib.rs:422-438
    ("same", "x", "x", True),
    ("same", "xy", "xy", True),
    ("same", "foo", "foo", True),
    ("same", LOREM55, LOREM55, True),
    ("same", LOREM187, LOREM187, True),
    # search_different, src/lib.rs:440-463
    ("different", "x", "y", False),
    ("different", "xy", "xz", False),
    ("different", "bar", "foo", False),
    ("different", LOREM55, "foo", False),
    ("different", LOREM187, "foo", False),
    ("different", LOREM187,
     "foo bar baz qux quux quuz corge grault garply waldo fred plugh xyzzy thud", False),
    # search_prefix, src/lib.rs:465-484
    ("prefix", "xy", "x", True),
    ("prefix", "foobar", "foo", True),
    ("prefix", LOREM55, "Lorem", True),
    ("prefix", LOREM187, "Lorem", True),
    ("prefix", LOREM187, LOREM55, True),
    # search_suffix, src/lib.rs:486-505
    ("suffix", "xy", "y", True),
    ("suffix", "foobar", "bar", True),
    ("suffix", LOREM55, "elit", True),
    ("suffix", LOREM187, "purus", True),
    ("suffix", LOREM187, "Aliquam iaculis fringilla mi, nec aliquet purus", True),
    # search_multiple, src/lib.rs:507-523
    ("multiple", "xx", "x", True),
    ("multiple", "xyxy", "xy", True),
    ("multiple", "foobarfoo", "foo", True),
    ("multiple", LOREM55, "it", True),
    ("multiple", LOREM187, "conse", True),
    # search_middle, src/lib.rs:525-544
    ("middle", "xyz", "y", True),
    ("middle", "wxyz", "xy", True),
    ("middle", "foobarfoo", "bar", True),
    ("middle", LOREM55, "consectetur", True),
    ("middle", LOREM187, "orci", True),
    ("middle", LOREM187, "Maecenas commodo posuere orci a consectetur", True),
]

# MemchrSearcher KATs, src/lib.rs:303-331 (needle is one byte)
MEMCHR = [
    ("memchr_same", "f", "f", True),
    ("memchr_different", "foo", "b", False),
    ("memchr_prefix", "foobar", "f", True),
    ("memchr_suffix", "foobar", "r", True),
    ("memchr_multiple", "foobarfoo", "o", True),
    ("memchr_middle", "foobarfoo", "b", True),
]

# Doc examples: src/x86.rs:1-15 and README.md:12-26
DOC = [
    ("doc", LOREM55, "ipsum", True),
    ("doc", "foo bar baz qux quux quuz corge grault garply waldo fred", "ipsum", False),
]

# Constructor contract, src/x86.rs:468-475,533-543 (DynamicAvx2Searcher only):
#   needle=[] -> N0 for any position (no panic); n==1 -> position must be 0;
#   n>=2 -> position < n, else panic.
CONTRACT = [
    {"needle": "", "position": 0, "ok": True},
    {"needle": "", "position": 7, "ok": True},
    {"needle": "", "position": 2 ** 64 - 1, "ok": True},   # `new` on empty: len.wrapping_sub(1)
    {"needle": "f", "position": 0, "ok": True},
    {"needle": "f", "position": 1, "ok": False},
    {"needle": "foo", "position": 2, "ok": True},
    {"needle": "foo", "position": 3, "ok": False},           # x86.rs:539-543
    {"needle": LOREM55, "position": 54, "ok": True},
    {"needle": LOREM55, "position": 55, "ok": False},
]


def main():
    os.makedirs(os.path.join(HERE, "data"), exist_ok=True)
    sha = {}
    for f in ("i386.txt", "words.txt", "haystack", "needle"):
        src = os.path.join(REF, "data", f)
        dst = os.path.join(HERE, "data", f)
        if os.path.exists(src):
            shutil.copyfile(src, dst)
            os.chmod(dst, 0o644)
        sha[f] = hashlib.sha256(open(dst, "rb").read()).hexdigest()

    with open(os.path.join(HERE, "kat.json"), "w") as fh:
        json.dump({
            "generic": [dict(family=a, haystack=b, needle=c, expected=d) for a, b, c, d in GENERIC + DOC],
            "memchr": [dict(family=a, haystack=b, needle=c, expected=d) for a, b, c, d in MEMCHR],
            "contract": CONTRACT,
        }, fh, indent=1)

    i386 = open(os.path.join(HERE, "data", "i386.txt"), "rb").read()
    words = open(os.path.join(HERE, "data", "words.txt"), "rb").read().split(b"\n")
    if words[-1] == b"":
        words.pop()
    lossy = i386.decode("utf-8", errors="replace").encode("utf-8")   # String::from_utf8_lossy, tests/i386.rs:63
    long_raw = sum(w in i386 for w in words)
    long_lossy = sum(w in lossy for w in words)
    raw_traversed = sum(i386.find(w) + len(w) for w in words)

    srt = sorted(words, key=len)                                      # stable, like sort_unstable_by_key on len only matters for counts
    pairs = 0
    hits = 0
    for i, n in enumerate(srt):
        for h in srt[i:]:
            pairs += 1
            hits += n in h
    hay = open(os.path.join(HERE, "data", "haystack"), "rb").read()
    ndl = open(os.path.join(HERE, "data", "needle"), "rb").read()
    sizes = [1, 5, 10, 20, 50, 100, 1000]                             # bench/benches/random.rs:16
    grid = []
    for i, s in enumerate(sizes):
        for h in sizes[i:]:
            grid.append({"needle_len": s, "haystack_len": h, "expected": ndl[:s] in hay[:h]})

    # the reference's THIRD criterion group, search_random_haystack (bench/benches/i386.rs:286-289): the 4,585 words in data/haystack
    # (1,000 bytes of [0-9A-Za-z] noise) - by the naive scan, like everything here
    random_hits = sum(w in hay for w in words)
    random_hit_words = sorted(w.decode("latin1") for w in words if w in hay)

    with open(os.path.join(HERE, "corpus_checksums.json"), "w") as fh:
        json.dump({
            "sha256": sha,
            "words": len(words),
            "i386_raw_len": len(i386),
            "i386_lossy_len": len(lossy),
            "long_haystack_hits_raw": long_raw,
            "long_haystack_hits_lossy": long_lossy,
            "long_haystack_bytes_to_first_hit_raw": raw_traversed,
            "short_haystack_pairs": pairs,
            "short_haystack_hits": hits,
            "random_haystack_len": len(hay),
            "random_haystack_hits": random_hits,
            "random_haystack_hit_words": random_hit_words,
            "random_grid": grid,
        }, fh, indent=1)
    print("wrote kat.json, corpus_checksums.json;", "pairs", pairs, "hits", hits,
          "long raw/lossy", long_raw, long_lossy, file=sys.stderr)


if __name__ == "__main__":
    main()
