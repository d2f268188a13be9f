#!/usr/bin/env python3
"""The BASELINE.json configs other than the headline one, as JSON lines (one GPU):

  config 3  1 GiB synthetic haystack, needle lengths {1,2,4,8,16,32,128}, absent + present variants
  config 5  batched: 4096 needles x 1 MiB haystacks in ONE launch (ss_search_batched)
  extra     value distributions of SURVEY.md 8d: English text (i386.txt tiled), adversarial 'a' fill
  config 1  the reference's own CPU-runnable case, searched on the GPU for completeness (latency-bound)

Kernel time = hipEvents on the launch stream (ss_searcher_last_kernel_ms) or torch events on the current
stream for the batched launch.  Haystacks <= 256 MiB are Infinity-Cache-resident and are labelled so.
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import sliceslice_rs_amd as ss  # noqa: E402
from settle import wait_for_vram_reclaim  # noqa: E402

SEED_HAY, SEED_NEEDLE = 0x5EED0001, 0x5EED0002


def absent(n, seed=SEED_NEEDLE):
    nd = bytearray(ss.fill_random_host(n, seed).tobytes())
    nd[0 if n == 1 else (1 if n == 2 else n // 2)] = 0xFF
    return bytes(nd)


def warm(fn, seconds=0.05):
    """Back-to-back launches for ~50 ms: clocks ramp up over the first milliseconds after an idle gap, which
    is longer than a whole series of sub-millisecond launches."""
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < seconds:
        fn()


def timed(searcher, hay, reps):
    searcher.set_timing(True)
    res = searcher.search_in(hay)
    warm(lambda: searcher.search_in(hay))
    ms = []
    for _ in range(reps):
        searcher.search_in(hay)
        ms.append(searcher.last_kernel_ms())
    return res, float(np.median(ms))


def emit(**kw):
    print(json.dumps(kw), flush=True)


def main():
    wait_for_vram_reclaim()
    ap = argparse.ArgumentParser()
    ap.add_argument("--gib", type=float, default=1.0)
    ap.add_argument("--reps", type=int, default=10)
    ap.add_argument("--skip", default="")
    args = ap.parse_args()
    skip = set(args.skip.split(",")) if args.skip else set()
    n_bytes = int(args.gib * (1 << 30))
    hay = torch.empty(n_bytes, dtype=torch.uint8, device="cuda")
    ss.fill_random_device(hay, SEED_HAY)
    torch.cuda.synchronize()
    resident = "hbm" if n_bytes > (256 << 20) else "infinity-cache"

    if "3" not in skip:
        for n in (1, 2, 4, 8, 16, 32, 128):
            nd = absent(n)
            s = ss.DynamicHipSearcher.new(nd)
            res, ms = timed(s, hay, args.reps)
            assert res is False
            emit(config=3, variant="absent", needle_len=n, haystack_bytes=n_bytes, kernel_ms=round(ms, 4),
                 gbps=round(n_bytes / ms / 1e6, 1), frac_of_8tbps=round(n_bytes / ms / 1e6 / 8000, 4), resident=resident)
            # present: a 0xFF-free needle planted at len-n (found only at the very end)
            pres = bytes(ss.fill_random_host(n, 0x5EED0003).tobytes())
            saved = hay[n_bytes - n:].clone()
            hay[n_bytes - n:] = torch.from_numpy(np.frombuffer(pres, dtype=np.uint8).copy()).cuda()
            s2 = ss.DynamicHipSearcher.new(pres)
            t0 = time.perf_counter()
            res2, ms2 = timed(s2, hay, 3)
            hay[n_bytes - n:] = saved
            if n >= 4:
                assert res2 is True
            emit(config=3, variant="present_at_end", needle_len=n, found=res2, kernel_ms=round(ms2, 4),
                 note="n<4: the random needle already occurs earlier, early exit")

    if "find" not in skip:
        nd = absent(16)
        s = ss.DynamicHipSearcher.new(nd)
        s.set_timing(True)
        assert s.find(hay) is None
        ms = []
        for _ in range(args.reps):
            s.find(hay)
            ms.append(s.last_kernel_ms())
        med = float(np.median(ms))
        emit(config="find", variant="absent (full scan)", needle_len=16, haystack_bytes=n_bytes, kernel_ms=round(med, 4),
             gbps=round(n_bytes / med / 1e6, 1), frac_of_8tbps=round(n_bytes / med / 1e6 / 8000, 4))
        pres = bytes(ss.fill_random_host(16, 0x5EED0003).tobytes())
        at = n_bytes // 3
        saved = hay[at:at + 16].clone()
        hay[at:at + 16] = torch.from_numpy(np.frombuffer(pres, dtype=np.uint8).copy()).cuda()
        s2 = ss.DynamicHipSearcher.new(pres)
        s2.set_timing(True)
        assert s2.find(hay) == at
        ms = []
        for _ in range(args.reps):
            s2.find(hay)
            ms.append(s2.last_kernel_ms())
        hay[at:at + 16] = saved
        emit(config="find", variant="present at len/3 (work right of the match is skipped)", position=at,
             kernel_ms=round(float(np.median(ms)), 4))

    if "5" not in skip:
        count, each = 4096, 1 << 20
        total = count * each
        blob = torch.empty(total, dtype=torch.uint8, device="cuda")
        ss.fill_random_device(blob, SEED_HAY)
        needles = b"".join(absent(16, SEED_NEEDLE + 1 + i) for i in range(count))
        nblob = torch.from_numpy(np.frombuffer(needles, dtype=np.uint8).copy()).cuda()
        hay_off = (torch.arange(count + 1, dtype=torch.int64) * each).cuda()
        nd_off = (torch.arange(count + 1, dtype=torch.int64) * 16).cuda()
        found = ss.search_batched(blob, hay_off, nblob, nd_off)
        assert int(found.sum().item()) == 0
        warm(lambda: ss.search_batched(blob, hay_off, nblob, nd_off))
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        ms = []
        for _ in range(args.reps):
            e0.record()
            found = ss.search_batched(blob, hay_off, nblob, nd_off)
            e1.record()
            e1.synchronize()
            ms.append(e0.elapsed_time(e1))
        med = float(np.median(ms))
        emit(config=5, problems=count, haystack_each=each, needle_len=16, total_bytes=total,
             launch_ms=round(med, 4), gbps=round(total / med / 1e6, 1), frac_of_8tbps=round(total / med / 1e6 / 8000, 4),
             note="one launch incl. the flag memset; torch events on the launch stream")
        # the same work as 4096 separate searches (launch + flag read-back each)
        searchers = [ss.DynamicHipSearcher.new(needles[16 * i:16 * i + 16]) for i in range(256)]
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i, s in enumerate(searchers):
            s.search_in(blob[i * each:(i + 1) * each])
        dt = time.perf_counter() - t0
        emit(config=5, variant="unbatched_256_of_them", wall_ms=round(dt * 1e3, 3), gbps=round(256 * each / dt / 1e9, 1))
        del blob

    if "text" not in skip:
        gd = os.path.join(ROOT, "tests", "golden", "data")
        i386 = np.frombuffer(open(os.path.join(gd, "i386.txt"), "rb").read(), dtype=np.uint8)
        reps_t = max(1, n_bytes // i386.size)
        text = torch.from_numpy(i386.copy()).cuda().repeat(reps_t)
        for nd, label in ((b"privilege level zero!", "absent phrase, last byte '!'"),
                          (b" the quick brown fox ", "absent, first/last byte ' ' (27% of the text)"),
                          (b"e" + b"\x00" * 14 + b"e", "absent, first/last byte 'e'"),
                          (b"segment descriptor table entries are", "absent, lower-case letters and spaces only"),
                          (b"the interrupt", "absent? common words only (13 bytes)")):
            s = ss.DynamicHipSearcher.new(nd)
            res, ms = timed(s, text, args.reps)
            emit(config="text", needle=nd.decode("latin1"), label=label + "; new() - filter bytes chosen by the library",
                 filter_bytes=list(s.filter3), filter_chars=[chr(nd[k]) for k in s.filter3], haystack_bytes=text.numel(), found=res,
                 kernel_ms=round(ms, 3), gbps=round(text.numel() / ms / 1e6, 1))
            s = ss.DynamicHipSearcher.with_position(nd, len(nd) - 1)
            res, ms = timed(s, text, args.reps)
            emit(config="text", needle=nd.decode("latin1"), label=label + "; with_position(n-1): the caller's byte, partner chosen next to it",
                 filter_bytes=list(s.filter3), filter_chars=[chr(nd[k]) for k in s.filter3], haystack_bytes=text.numel(), found=res,
                 kernel_ms=round(ms, 3), gbps=round(text.numel() / ms / 1e6, 1))
            s.set_filter(0, len(nd) - 1)
            res, ms = timed(s, text, args.reps)
            emit(config="text", needle=nd.decode("latin1"), label=label + "; set_filter(0, n-1): the reference's pair, verbatim",
                 filter_bytes=list(s.filter3), haystack_bytes=text.numel(), found=res, kernel_ms=round(ms, 3),
                 gbps=round(text.numel() / ms / 1e6, 1))
            # row f3: the same needle with the position chosen from a byte histogram of (a sample of) the haystack
            hist = ss.byte_histogram(text, sample_bytes=64 << 20)
            s = ss.DynamicHipSearcher.new(nd)
            s.set_filter(*ss.choose_filter_triple(nd, hist))
            res, ms = timed(s, text, args.reps)
            emit(config="text", needle=nd.decode("latin1"), label=label + "; filter bytes chosen from a byte histogram of the haystack "
                 "(ss_byte_histogram_device + ss_choose_filter_triple with the histogram)", filter_bytes=list(s.filter3),
                 filter_chars=[chr(nd[k]) for k in s.filter3], haystack_bytes=text.numel(), found=res, kernel_ms=round(ms, 3),
                 gbps=round(text.numel() / ms / 1e6, 1))
        del text
        a = torch.full((n_bytes,), 0x61, dtype=torch.uint8, device="cuda")
        for nd, pos, label in ((b"a" * 15 + b"b", None, "adversarial: every offset passes the first-byte filter"),
                               (b"a" * 15 + b"b", 15, "same needle, with_position(15)"),
                               (b"a" * 15 + b"b", 0, "same needle, position 0"),
                               (b"ab" + b"a" * 14, None, "first byte common, last byte common, fails at byte 1")):
            s = ss.DynamicHipSearcher(nd, pos)
            res, ms = timed(s, a, args.reps)
            emit(config="adversarial", label=label, filter_bytes=list(s.filter3), haystack_bytes=n_bytes, found=res, kernel_ms=round(ms, 3),
                 gbps=round(n_bytes / ms / 1e6, 1))
        del a
        # long needles on random bytes: `new` and `with_position` keep the filter bytes within 15 bytes of each other
        # (single-stream kernel); the reference's pair (0, n-1), verbatim, needs the cross-lane kernel (n <= 1008) or a
        # second load stream (beyond)
        for n in (128, 1000, 2000):
            nd = absent(n)
            ref = ss.DynamicHipSearcher.new(nd)
            ref.set_filter(0, n - 1)
            for how, s in (("new()", ss.DynamicHipSearcher.new(nd)), ("with_position(n-1)", ss.DynamicHipSearcher.with_position(nd, n - 1)),
                           ("set_filter(0, n-1)", ref)):
                res, ms = timed(s, hay, args.reps)
                emit(config="long-needle", needle_len=n, how=how, filter_bytes=list(s.filter3), haystack_bytes=n_bytes, found=res,
                     kernel_ms=round(ms, 4), gbps=round(n_bytes / ms / 1e6, 1))

    if "1" not in skip:
        gd = os.path.join(ROOT, "tests", "golden", "data")
        raw = open(os.path.join(gd, "i386.txt"), "rb").read()
        i386 = torch.from_numpy(np.frombuffer(raw, dtype=np.uint8).copy()).cuda()
        words = [w for w in open(os.path.join(gd, "words.txt"), "rb").read().split(b"\n") if w]
        searchers = [ss.DynamicHipSearcher.new(w) for w in words]
        hits = sum(s.search_in(i386) for s in searchers)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        hits = sum(s.search_in(i386) for s in searchers)
        dt = time.perf_counter() - t0
        emit(config=1, where="gpu, one launch + flag read-back per needle (latency-bound, Infinity-Cache-resident)",
             needles=len(words), hits=hits, ms_per_iteration=round(dt * 1e3, 2),
             us_per_search=round(dt / len(words) * 1e6, 2), reference_published_ms=35.181)
        # the same loop captured ONCE into a hipGraph (4,585 kernel nodes writing 4,585 flags) and replayed
        gflags = torch.zeros(len(words), dtype=torch.int32, device="cuda")
        side = torch.cuda.Stream()
        with torch.cuda.stream(side):
            for i, s in enumerate(searchers[:8]):
                s.search_in_async(i386, gflags[i:i + 1])
        torch.cuda.synchronize()
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph):
            gflags.zero_()
            for i, s in enumerate(searchers):
                s.search_in_async(i386, gflags[i:i + 1])
        graph.replay()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(5):
            graph.replay()
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / 5
        emit(config=1, where="gpu, the per-needle loop captured into one hipGraph and replayed", needles=len(words),
             hits=int(gflags.sum().item()), ms_per_iteration=round(dt * 1e3, 3), us_per_search=round(dt / len(words) * 1e6, 2),
             reference_published_ms=35.181)
        # the same loop as ONE launch: 4,585 needle ranges, all aliasing the one haystack
        lens = np.array([len(w) for w in words], dtype=np.int64)
        nb = np.zeros(len(words), dtype=np.int64)
        nb[1:] = np.cumsum(lens)[:-1]
        nblob = torch.from_numpy(np.frombuffer(b"".join(words), dtype=np.uint8).copy()).cuda()
        nbt, net = torch.from_numpy(nb).cuda(), torch.from_numpy(nb + lens).cuda()
        hb = torch.zeros(len(words), dtype=torch.int64, device="cuda")
        he = torch.full((len(words),), len(raw), dtype=torch.int64, device="cuda")
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        warm(lambda: ss.search_batched(i386, None, nblob, None, hay_ranges=(hb, he), needle_ranges=(nbt, net)))
        ms = []
        for _ in range(args.reps):
            e0.record()
            found = ss.search_batched(i386, None, nblob, None, hay_ranges=(hb, he), needle_ranges=(nbt, net))
            e1.record()
            e1.synchronize()
            ms.append(e0.elapsed_time(e1))
        emit(config=1, where="gpu, ONE batched launch for all needles (bench/benches/i386.rs:252-256 loop)",
             needles=len(words), hits=int(found.sum().item()), ms_per_iteration=round(float(np.median(ms)), 3),
             reference_published_ms=35.181, note="haystack Infinity-Cache/L2-resident, like the reference's L2/L3-resident CPU run")
        # short-haystack loop (bench/benches/i386.rs:118-129): 10,513,405 pairs, lane per pair
        ws = sorted(words, key=len)
        W = len(ws)
        lens = np.array([len(w) for w in ws], dtype=np.int64)
        starts = np.zeros(W, dtype=np.int64)
        starts[1:] = np.cumsum(lens)[:-1]
        blob = torch.from_numpy(np.frombuffer(b"".join(ws), dtype=np.uint8).copy()).cuda()
        ni = np.repeat(np.arange(W, dtype=np.int64), W - np.arange(W))
        hj = np.concatenate([np.arange(i, W, dtype=np.int64) for i in range(W)])
        nbt, net = torch.from_numpy(starts[ni]).cuda(), torch.from_numpy(starts[ni] + lens[ni]).cuda()
        hbt, het = torch.from_numpy(starts[hj]).cuda(), torch.from_numpy(starts[hj] + lens[hj]).cuda()
        ms = []
        for _ in range(args.reps):
            e0.record()
            found = ss.search_batched(blob, None, blob, None, hay_ranges=(hbt, het), needle_ranges=(nbt, net), pairs=True)
            e1.record()
            e1.synchronize()
            ms.append(e0.elapsed_time(e1))
        emit(config="1-short", where="gpu, ss_search_pairs, one lane per pair", pairs=int(ni.size),
             hits=int(found.sum().item()), ms_per_iteration=round(float(np.median(ms)), 3),
             ns_per_search=round(float(np.median(ms)) * 1e6 / ni.size, 3), reference_published_ms=79.416)


if __name__ == "__main__":
    main()
