#!/usr/bin/env python3
"""Kernel-variant / grid sweep on one GPU: prints kernel-only GB/s (hipEvent time on the launch stream)
for every (needle length, variant, grid) and the plain streaming-read ceiling.  Tuning aid, not a test."""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import sliceslice_rs_amd as ss  # noqa: E402
from settle import wait_for_vram_reclaim  # noqa: E402


def absent(n, ff_at=-1):
    """0xFF never occurs in the haystack: at index 0 the first-byte filter never passes (no phase 2 at all)."""
    nd = bytearray(ss.fill_random_host(n, 0x5EED0002).tobytes())
    nd[ff_at if ff_at >= 0 else (0 if n == 1 else (1 if n == 2 else n // 2))] = 0xFF
    return bytes(nd)


def main():
    wait_for_vram_reclaim()
    ap = argparse.ArgumentParser()
    ap.add_argument("--gib", type=float, default=8.0)
    ap.add_argument("--reps", type=int, default=6)
    ap.add_argument("--needles", default="16")
    ap.add_argument("--variants", default="40,41")
    ap.add_argument("--grids", default="0")
    ap.add_argument("--mis", type=int, default=0)
    ap.add_argument("--ff-at", type=int, default=-1, help="index of the needle's 0xFF byte (default: the middle)")
    args = ap.parse_args()
    n_bytes = int(args.gib * (1 << 30))
    buf = torch.empty(n_bytes + 64, dtype=torch.uint8, device="cuda")
    ss.fill_random_device(buf, 0x5EED0001)
    hay = buf[args.mis:args.mis + n_bytes]
    torch.cuda.synchronize()
    print(json.dumps({"read_ceiling_gbps": round(ss.read_ceiling_gbps(buf[:n_bytes], reps=5), 1)}), flush=True)
    # settle (on top of wait_for_vram_reclaim above): a short run-in before the first measured series
    warm = ss.DynamicHipSearcher.new(absent(16))
    t_end = time.perf_counter() + 0.25
    while time.perf_counter() < t_end:
        warm.search_in(hay)
    for n in [int(x) for x in args.needles.split(",")]:
        nd = absent(n, args.ff_at)
        for v in [int(x) for x in args.variants.split(",")]:
            for g in [int(x) for x in args.grids.split(",")]:
                s = ss.DynamicHipSearcher.new(nd)
                s.set_variant(v)
                s.set_grid(g)
                s.set_timing(True)
                assert s.search_in(hay) is False
                for _ in range(8):                 # run-in of the series
                    s.search_in(hay)
                ms = []
                for _ in range(args.reps):
                    s.search_in(hay)
                    ms.append(s.last_kernel_ms())
                med = float(np.median(ms))
                print(json.dumps({"n": n, "variant": v, "grid": g, "ms": round(med, 4),
                                  "gbps": round(n_bytes / med / 1e6, 1), "min_ms": round(min(ms), 4)}), flush=True)


if __name__ == "__main__":
    main()
