#!/usr/bin/env python3
"""Kernel rate for every (Q, Q3) = (second / 4, third / 4) combination of the three-byte first phase, on random bytes and on an
all-'a' haystack (needle a...ab variants), for search_in, find and the batched kernel: one process, one buffer per kind."""
import json
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import sliceslice_rs_amd as ss  # noqa: E402


def rate(s, hay, reps=15):
    s.set_timing(True)
    ms = []
    for _ in range(reps):
        assert s.search_in(hay) is False
        ms.append(s.last_kernel_ms())
    return round(hay.numel() / float(np.median(ms)) / 1e6, 1)


def main():
    n_bytes = 1 << 30
    hay = torch.empty(n_bytes, dtype=torch.uint8, device="cuda")
    ss.fill_random_device(hay, 0x5EED0001)
    nd = bytearray(ss.fill_random_host(16, 0x5EED0002).tobytes())
    nd[8] = 0xFF
    nd = bytes(nd)
    out = {}
    for b in (0, 1, 3, 5, 10, 15):
        for c in (1, 2, 6, 9, 14, 15):
            if c == b:
                continue
            s = ss.DynamicHipSearcher.new(nd)
            s.set_filter(0, b, c)
            out[f"random b={b} c={c}"] = rate(s, hay)
    print(json.dumps(out), flush=True)
    # the FIND kernels hold their own copies of the first phase
    out = {}
    for b in (1, 5, 10, 15):
        for c in (2, 6, 9, 14):
            if c == b:
                continue
            s = ss.DynamicHipSearcher.new(nd)
            s.set_filter(0, b, c)
            s.set_timing(True)
            ms = []
            for _ in range(15):
                assert s.find(hay) is None
                ms.append(s.last_kernel_ms())
            out[f"find b={b} c={c}"] = round(hay.numel() / float(np.median(ms)) / 1e6, 1)
    print(json.dumps(out), flush=True)
    # ... and so does the batched kernel: 1,024 problems of 1 MiB, position p for all of them (third byte chosen on the device)
    count, each = 1024, 1 << 20
    nb = bytearray(ss.fill_random_host(16 * count, 0x5EED0003).tobytes())
    for i in range(count):
        nb[16 * i + 8] = 0xFF
    nblob = torch.from_numpy(np.frombuffer(bytes(nb), dtype=np.uint8).copy()).cuda()
    hay_off = (torch.arange(count + 1, dtype=torch.int64) * each).cuda()
    nd_off = (torch.arange(count + 1, dtype=torch.int64) * 16).cuda()
    out = {}
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    for p in (0, 1, 3, 5, 9, 10, 13, 15):
        pos = torch.full((count,), p, dtype=torch.int64, device="cuda")
        ms = []
        for _ in range(15):
            e0.record()
            found = ss.search_batched(hay, hay_off, nblob, nd_off, position=pos)
            e1.record()
            e1.synchronize()
            ms.append(e0.elapsed_time(e1))
        assert int(found.sum().item()) == 0
        out[f"batched position={p}"] = round(count * each / float(np.median(ms)) / 1e6, 1)
    print(json.dumps(out), flush=True)
    hay.fill_(0x61)
    out = {}
    for nd, flt in ((b"a" * 15 + b"b", (0, 15, 14)), (b"a" * 15 + b"b", (0, 0, 15)), (b"a" * 15 + b"b", (0, 14, 15)), (b"a" * 15 + b"b", (0, 1, 15)),
                    (b"a" * 15 + b"b", (0, 15, 1)), (b"ab" + b"a" * 14, (0, 1, 15)), (b"ab" + b"a" * 14, (0, 15, 1)), (b"ab" + b"a" * 14, (0, 1, 2)),
                    (b"a" * 7 + b"b" + b"a" * 8, (0, 7, 15)), (b"a" * 7 + b"b" + b"a" * 8, (0, 15, 7))):
        s = ss.DynamicHipSearcher.new(nd)
        s.set_filter(*flt)
        out[f"aaaa {nd.decode()} {flt}"] = rate(s, hay)
    print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
