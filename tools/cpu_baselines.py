#!/usr/bin/env python3
"""the CPU side, timed on THIS host with the C/AVX2 restatement of the reference
path (oracle/sliceslice_oracle.c - NOT the Rust binary, which cannot be built in this image).  JSON lines:
config 1 long + short (beside README's 35.181 / 79.416 ms), the synthetic 1 GiB haystack at 1 thread and
all threads, and the needle-length sweep {1,2,4,8,16,32,128} at 1 thread."""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import oracle as O  # noqa: E402

SEED_HAY, SEED_NEEDLE = 0x5EED0001, 0x5EED0002


def absent(n):
    nd = bytearray(O.fill_random(n, SEED_NEEDLE).tobytes())
    nd[0 if n == 1 else (1 if n == 2 else n // 2)] = 0xFF
    return bytes(nd)


def emit(**kw):
    print(json.dumps(kw), flush=True)


def main():
    cores = len(os.sched_getaffinity(0))
    model = "?"
    with open("/proc/cpuinfo") as fh:
        for l in fh:
            if l.startswith("model name"):
                model = l.split(":", 1)[1].strip()
                break
    emit(cpu_model=model, hardware_threads=cores, avx2=bool(O.have_avx2()), note="C restatement of the reference AVX2 path")
    gd = os.path.join(ROOT, "tests", "golden", "data")
    i386 = open(os.path.join(gd, "i386.txt"), "rb").read()
    words = [w for w in open(os.path.join(gd, "words.txt"), "rb").read().split(b"\n") if w]
    O.bench_long(i386, words, 1)
    t = time.perf_counter()
    hits = O.bench_long(i386, words, 10)
    emit(config=1, loop="long (bench/benches/i386.rs:246-256)", threads=1, ms_per_iteration=round((time.perf_counter() - t) * 100, 3),
         hits=hits // 10, readme_ms=35.181)
    ws = sorted(words, key=len)
    t = time.perf_counter()
    hits = O.bench_short(ws, 2)
    emit(config=1, loop="short (bench/benches/i386.rs:118-129)", threads=1, ms_per_iteration=round((time.perf_counter() - t) * 500, 3),
         hits=hits // 2, readme_ms=79.416)
    n_bytes = 1 << 30
    hay = O.fill_random(n_bytes, SEED_HAY)
    for n in (1, 2, 4, 8, 16, 32, 128):
        s = O.OracleSearcher(absent(n))
        best = float("inf")
        for _ in range(3):
            t = time.perf_counter()
            r = s.search_in(hay)
            best = min(best, time.perf_counter() - t)
        assert r is False
        emit(config=3, needle_len=n, threads=1, haystack_bytes=n_bytes, gbps=round(n_bytes / best / 1e9, 2))
    s = O.OracleSearcher(absent(16))
    for th in sorted({1, 8, 64, cores}):
        best = float("inf")
        for _ in range(5):
            t = time.perf_counter()
            r = s.search_in(hay, threads=th)
            best = min(best, time.perf_counter() - t)
        emit(config=2, needle_len=16, threads=th, haystack_bytes=n_bytes, gbps=round(n_bytes / best / 1e9, 2))


if __name__ == "__main__":
    main()
