#!/usr/bin/env python3
"""Early-exit / present-needle scenarios (kernel ms over a 1 GiB haystack, or argv[1] GiB): a hunt for pathologies, e.g. a
needle that occurs everywhere must not be slower than a full scan."""
import json
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import sliceslice_rs_amd as ss  # noqa: E402
from settle import wait_for_vram_reclaim  # noqa: E402


def med(fn, s, reps=7):
    s.set_timing(True)
    out = fn()
    ms = []
    for _ in range(reps):
        fn()
        ms.append(s.last_kernel_ms())
    return out, float(np.median(ms))


def main():
    wait_for_vram_reclaim()
    n_bytes = int(float(sys.argv[1]) * (1 << 30)) if len(sys.argv) > 1 else 1 << 30   # [GiB] [grid override] [variant]
    grid = int(sys.argv[2]) if len(sys.argv) > 2 else 0
    variant = int(sys.argv[3]) if len(sys.argv) > 3 else 0
    hay = torch.empty(n_bytes, dtype=torch.uint8, device="cuda")
    ss.fill_random_device(hay, 0x5EED0001)
    host_head = hay[:4096].cpu().numpy().tobytes()
    text = None
    rows = []

    def run(label, needle, h=hay, position=None):
        s = ss.DynamicHipSearcher(needle, position)
        s.set_grid(grid)
        s.set_variant(variant)
        r, ms = med(lambda: s.search_in(h), s)
        p, msf = med(lambda: s.find(h), s)
        rows.append((label, len(needle), r, round(ms, 4), p, round(msf, 4)))
        print(json.dumps({"scenario": label, "n": len(needle), "search_in": r, "search_ms": round(ms, 4), "find": p,
                          "find_ms": round(msf, 4)}), flush=True)

    absent16 = bytearray(ss.fill_random_host(16, 0x5EED0002).tobytes())
    absent16[8] = 0xFF
    run("absent (full scan)", bytes(absent16))
    run("1 byte, present everywhere", host_head[100:101])
    run("2 bytes, present every ~64 KiB", host_head[200:202])
    run("3 bytes, present every ~16 MiB", host_head[300:303])
    for n in (16, 32, 200):
        cut0 = host_head[:n]
        run("first %d bytes of the haystack (match at 0)" % n, cut0)
        mid = hay[n_bytes // 2: n_bytes // 2 + n].cpu().numpy().tobytes()
        run("%d bytes cut from the middle" % n, mid)
        endc = hay[n_bytes - n:].cpu().numpy().tobytes()
        run("%d bytes cut from the end" % n, endc)
    a = torch.full((n_bytes,), 0x61, dtype=torch.uint8, device="cuda")
    run("all-'a' haystack, needle 'a'*16 (match at every offset)", b"a" * 16, a)
    run("all-'a' haystack, needle 'a'*300", b"a" * 300, a)
    run("all-'a' haystack, needle 'a'*15+'b' (absent, adversarial)", b"a" * 15 + b"b", a)
    del a


if __name__ == "__main__":
    main()
