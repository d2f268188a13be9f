#!/usr/bin/env python3
"""Randomised differential test on the GPU: search_in / find through the C ABI against Python's bytes.find over
random (haystack kind, length, misalignment, needle, position or filter triple, kernel variant, launch shape).  Runs for
argv[1] seconds (default 60) with seed argv[2]; a third argument (GiB) selects the large-haystack mode.  Prints a
JSON summary, exits non-zero on the first mismatch."""
import json
import os
import random
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import sliceslice_rs_amd as ss  # noqa: E402

# Kernel variants and launch shapes drawn at random - with the tuning build (SLICESLICE_HIP_LIB=<libsliceslice_hip_tuning.so>: every
# variant, ss_searcher_set_variant / _set_grid).  The product library has neither the variants nor the overrides: there the campaign
# draws haystacks, needles, positions and filter triples only, and every search runs the automatic choice.
TUNING = ss.lib().has_hooks and b"tuning" in ss.lib().ss_version()
VARIANTS = [0, 40, 41, 80, 81, 241, 281, 341, 381, 1041, 2041, 2081, 100041, 300041, 100241, 100341, 40041, 60041] if TUNING else [0]
FIND_VARIANTS = [0, 40, 41, 241] if TUNING else [0]
BIG_VARIANTS = [0, 0, 0, 41, 241, 341, 1041, 60041] if TUNING else [0]
GRIDS = [0, 0, 0, 1, 3, 64, 4096, -1, -2, -3, -7, -64] if TUNING else [0]
BIG_GRIDS = [0, 0, 0, -1, -2, -5, 8192] if TUNING else [0]


def make_haystack(rng, n_bytes):
    kind = rng.choice(["random", "random", "abcd", "ab", "text", "zeros"])
    if kind == "random":
        a = np.frombuffer(rng.randbytes(n_bytes), dtype=np.uint8).copy()
    elif kind == "abcd":
        a = np.frombuffer(bytes(rng.choice(b"abcd") for _ in range(min(n_bytes, 4096))), dtype=np.uint8)
        a = np.resize(a, n_bytes).copy()
        idx = np.array([rng.randrange(n_bytes) for _ in range(max(1, n_bytes // 997))], dtype=np.int64)
        a[idx] = ord("e")
    elif kind == "ab":
        a = np.full(n_bytes, ord("a"), dtype=np.uint8)
        idx = np.array([rng.randrange(n_bytes) for _ in range(max(1, n_bytes // 257))], dtype=np.int64)
        a[idx] = ord("b")
    elif kind == "text":
        t = open(os.path.join(os.path.dirname(__file__), "..", "tests", "golden", "data", "i386.txt"), "rb").read()
        a = np.resize(np.frombuffer(t, dtype=np.uint8), n_bytes).copy()
    else:
        a = np.zeros(n_bytes, dtype=np.uint8)
        a[rng.randrange(n_bytes)] = 1
    return kind, a


def random_filter(rng, s, n):
    """A third of the searchers get a random filter pair / triple instead of what the constructor picked."""
    if n < 2 or rng.random() < 0.66:
        return None
    a = rng.randrange(n - 1) if rng.random() < 0.7 else 0
    near = rng.random() < 0.7
    b = rng.randrange(a + 1, min(n, a + 16)) if near else rng.randrange(a, n)
    c = None
    if n >= 3 and 0 < b - a <= 15 and rng.random() < 0.7:
        c = rng.randrange(a + 1, min(n, a + 16))
        if c == b:
            c = None
    s.set_filter(a, b, c)
    return s.filter3


def big(seconds, seed, gib):
    """Large haystacks (two-tile workgroups, entry peek, early exit): the same needle planted at several random
    offsets of a random haystack; find must return the leftmost, search_in true; then every copy is destroyed
    again and both must say absent."""
    rng = random.Random(seed)
    n_bytes = int(gib * (1 << 30))
    hay = torch.empty(n_bytes, dtype=torch.uint8, device="cuda")
    ss.fill_random_device(hay, 0xF00D + seed)
    t_end = time.time() + seconds
    rounds = 0
    while time.time() < t_end:
        n = rng.choice([1, 2, 5, 16, 17, 40, 300, 1100])
        nd = bytearray(rng.randbytes(n))
        nd[rng.randrange(n)] = 0xFF                                   # 0xFF never occurs in the haystack
        nd = bytes(nd)
        t = torch.from_numpy(np.frombuffer(nd, dtype=np.uint8).copy()).cuda()
        offs = sorted(rng.randrange(n_bytes - n + 1) for _ in range(rng.choice([1, 2, 5])))
        if rng.random() < 0.3:
            offs[0] = rng.choice([0, n_bytes - n, 32768 * 1024 - 3, (1 << 30) - 1])   # edges: start, end, peek threshold ...
            offs.sort()
        saved = [hay[o:o + n].clone() for o in offs]
        for o in offs:
            hay[o:o + n] = t
        pos = None if rng.random() < 0.5 else (0 if n == 1 else rng.randrange(n))
        s = ss.DynamicHipSearcher(nd, pos)
        flt = random_filter(rng, s, n)
        s.set_variant(rng.choice(BIG_VARIANTS))
        s.set_grid(rng.choice(BIG_GRIDS))
        got_b, got_p = s.search_in(hay), s.find(hay)
        for o, sv in zip(reversed(offs), reversed(saved)):            # restore (overlapping plants: reverse order)
            hay[o:o + n] = sv
        absent_b, absent_p = s.search_in(hay), s.find(hay)
        rounds += 1
        if got_b is not True or got_p != offs[0] or absent_b is not False or absent_p is not None:
            print(json.dumps({"MISMATCH": True, "mode": "big", "needle_len": n, "position": pos, "filter": flt, "planted": offs,
                              "search_in": got_b, "find": got_p, "after_restore": [absent_b, absent_p], "seed": seed}))
            sys.exit(1)
    print(json.dumps({"fuzz": "ok", "mode": "big", "gib": gib, "seconds": seconds, "seed": seed, "rounds": rounds,
                      "searches": 4 * rounds}))


def nonlatin(seconds, seed):
    """Haystacks on which the library re-chooses the filter bytes (ss_census.hip): 512 MiB of UTF-8-like text in a non-Latin script
    (the static rarity ranking is wrong about it), needles cut from the haystack - whole, with one byte changed, with a byte that
    never occurs - through `new` (automatic: histogram-driven triple, on trial against the census) and `with_position`; six scans
    per searcher, search_in and find alternating, so that the static triple, the trial and the settled choice all answer."""
    rng = random.Random(seed)
    nrng = np.random.default_rng(seed)
    n_bytes = 512 << 20
    t_end = time.time() + seconds
    rounds = searches = adopted = 0
    while time.time() < t_end:
        pairs = n_bytes // 2
        lead = nrng.choice(np.array([0xD0, 0xD1], dtype=np.uint8), size=pairs, p=[0.6, 0.4])
        trail = (0x80 + np.minimum(nrng.geometric(0.08, size=pairs) - 1, 63)).astype(np.uint8)
        host = np.empty(n_bytes, dtype=np.uint8)
        host[0::2], host[1::2] = lead, trail
        blanks = nrng.integers(0, pairs, size=pairs // 7)
        host[2 * blanks] = 0x20
        host[2 * blanks + 1] = 0x20
        hb = host.tobytes()
        hay = torch.from_numpy(host).cuda()
        for _ in range(10):
            if time.time() >= t_end:
                break
            n = rng.choice([3, 4, 8, 12, 16, 17, 24, 32, 64, 200])
            at = rng.choice([0, n_bytes - n, rng.randrange(n_bytes - n + 1)])
            nd = bytearray(hb[at:at + n])
            r = rng.random()
            if r < 0.35:                                               # near miss: one byte changed
                k = rng.randrange(n)
                nd[k] = (nd[k] + 1 + rng.randrange(254)) & 0xFF
            elif r < 0.55:                                             # a byte that never occurs in the haystack
                nd[rng.randrange(n)] = rng.choice([0xFF, ord("e"), 0x00])
            nd = bytes(nd)
            want = hb.find(nd)
            pos = None if rng.random() < 0.7 else rng.randrange(n)
            s = ss.DynamicHipSearcher(nd, pos)
            for it in range(6):
                got = s.search_in(hay) if it % 2 == 0 else s.find(hay)
                ok = got == (want >= 0) if it % 2 == 0 else got == (want if want >= 0 else None)
                searches += 1
                if not ok:
                    print(json.dumps({"MISMATCH": True, "mode": "nonlatin", "needle_len": n, "position": pos, "at": at, "want": want, "scan": it,
                                      "got": got, "seed": seed, "round": rounds}))
                    sys.exit(1)
            if ss.lib().has_hooks:
                s.census(hay)
                adopted += s.triple_state == 2
            rounds += 1
        del hay
    print(json.dumps({"fuzz": "ok", "mode": "nonlatin", "seconds": seconds, "seed": seed, "searchers": rounds, "searches": searches,
                      "triples_from_the_histogram": adopted if ss.lib().has_hooks else None}))


def compact(seconds, seed):
    """Filters that meet no candidates and take their compact form (ss_census.hip, propose_compact): 512 MiB of random bytes (0xFF-free),
    random needles of 12-200 bytes - absent (a 0xFF inside), or planted at a random offset / flush against either end - through `new`
    and `with_position`; fourteen scans per searcher, search_in and find alternating, so that the static triple, the histogram's, the
    trials and the settled compact form all answer."""
    rng = random.Random(seed)
    n_bytes = 512 << 20
    hay = torch.empty(n_bytes, dtype=torch.uint8, device="cuda")
    ss.fill_random_device(hay, 0xC0DE + seed)
    t_end = time.time() + seconds
    rounds = searches = moved = 0
    while time.time() < t_end:
        n = rng.choice([12, 13, 16, 17, 24, 32, 40, 64, 200])
        nd = bytearray(rng.randbytes(n))
        for k in range(n):
            if nd[k] == 0xFF:
                nd[k] = 0
        plant = rng.random() < 0.5
        if not plant:
            nd[rng.randrange(n)] = 0xFF
        nd = bytes(nd)
        at = rng.choice([0, n_bytes - n, rng.randrange(n_bytes - n + 1)])
        keep = None
        pos = None if rng.random() < 0.7 else rng.randrange(n)
        s = ss.DynamicHipSearcher(nd, pos)
        for it in range(14):
            if plant and it == 7:                                     # from the eighth scan on the needle is there
                keep = hay[at:at + n].clone()
                hay[at:at + n] = torch.from_numpy(np.frombuffer(nd, dtype=np.uint8).copy()).cuda()
            want = at if (plant and it >= 7) else -1
            got = s.search_in(hay) if it % 2 == 0 else s.find(hay)
            ok = got == (want >= 0) if it % 2 == 0 else got == (want if want >= 0 else None)
            searches += 1
            if not ok:
                print(json.dumps({"MISMATCH": True, "mode": "compact", "needle_len": n, "position": pos, "at": at, "want": want, "scan": it,
                                  "got": got, "seed": seed, "round": rounds, "state": s.tuning_state(hay)}))
                sys.exit(1)
            if it == 6:
                st = s.tuning_state(hay)
                moved += st["in_force"] != st["own"]
        if keep is not None:
            hay[at:at + n] = keep
        rounds += 1
    print(json.dumps({"fuzz": "ok", "mode": "compact", "seconds": seconds, "seed": seed, "searchers": rounds, "searches": searches,
                      "searchers_whose_bytes_moved": moved}))


def main():
    seconds = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    if len(sys.argv) > 3 and sys.argv[3] == "compact":                # fuzz_gpu.py SECONDS SEED compact: filters that meet no candidates
        return compact(seconds, seed)
    if len(sys.argv) > 3 and sys.argv[3] == "nonlatin":               # fuzz_gpu.py SECONDS SEED nonlatin: histogram-driven filter bytes
        return nonlatin(seconds, seed)
    if len(sys.argv) > 3:                                             # fuzz_gpu.py SECONDS SEED GIB: the large-haystack mode
        return big(seconds, seed, float(sys.argv[3]))
    rng = random.Random(seed)
    t_end = time.time() + seconds
    cases = searches = 0
    while time.time() < t_end:
        n_bytes = rng.choice([1, 15, 16, 17, 63, 1000, 4097, 70000, 1 << 20, (3 << 20) + 5, (20 << 20) + 123])
        kind, host = make_haystack(rng, n_bytes)
        mis = rng.randrange(16)
        dev = torch.empty(n_bytes + 32, dtype=torch.uint8, device="cuda")
        dev[mis:mis + n_bytes] = torch.from_numpy(host).cuda()
        hay = dev[mis:mis + n_bytes]
        hb = host.tobytes()
        cases += 1
        for _ in range(12):
            n = rng.choice([1, 1, 2, 3, 4, 8, 15, 16, 17, 31, 33, 64, 100, 257, 1000, 1100, 3000])
            if n > n_bytes and rng.random() < 0.8:
                n = rng.randrange(1, n_bytes + 1)
            if n <= n_bytes and rng.random() < 0.6:
                at = rng.choice([0, n_bytes - n, rng.randrange(n_bytes - n + 1)])
                nd = bytearray(hb[at:at + n])
                if rng.random() < 0.4:                               # near miss: one byte changed
                    k = rng.randrange(n)
                    nd[k] = (nd[k] + 1 + rng.randrange(254)) & 0xFF
            else:
                nd = bytearray(rng.randbytes(n))
            nd = bytes(nd)
            pos = None if rng.random() < 0.5 else (0 if n == 1 else rng.randrange(n))
            want = hb.find(nd)
            s = ss.DynamicHipSearcher(nd, pos)
            flt = random_filter(rng, s, n)
            s.set_variant(rng.choice(VARIANTS))
            s.set_grid(rng.choice(GRIDS))
            got_b = s.search_in(hay)
            s.set_variant(rng.choice(FIND_VARIANTS))                 # find() supports the U = 4 kernels
            got_p = s.find(hay)
            searches += 2
            if got_b != (want >= 0) or got_p != (want if want >= 0 else None):
                print(json.dumps({"MISMATCH": True, "kind": kind, "len": n_bytes, "mis": mis, "needle_len": n, "position": pos, "filter": flt,
                                  "want": want, "search_in": got_b, "find": got_p, "seed": seed, "case": cases}))
                sys.exit(1)
    print(json.dumps({"fuzz": "ok", "seconds": seconds, "seed": seed, "haystacks": cases, "searches": searches}))


if __name__ == "__main__":
    main()
