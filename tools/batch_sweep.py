#!/usr/bin/env python3
"""Workgroup-count sweep for ss_search_batched (SLICESLICE_BATCH_WGS) on config 5 (4096 x 1 MiB, 16-byte
needles) and on the config-1 shape (4,585 needles x one 857 kB haystack).  Tuning aid, not a test."""
import json
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import sliceslice_rs_amd as ss  # noqa: E402
from settle import wait_for_vram_reclaim  # noqa: E402

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def absent(n, seed):
    nd = bytearray(ss.fill_random_host(n, seed).tobytes())
    nd[n // 2] = 0xFF
    return bytes(nd)


def timed(fn, reps):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ms = []
    for _ in range(reps):
        e0.record()
        fn()
        e1.record()
        e1.synchronize()
        ms.append(e0.elapsed_time(e1))
    return float(np.median(ms)), float(min(ms))


def main():
    wait_for_vram_reclaim()
    targets = [int(x) for x in (sys.argv[1] if len(sys.argv) > 1 else "0,4096,8192,16384,32768,65536").split(",")]
    count, each = 4096, 1 << 20
    blob = torch.empty(count * each, dtype=torch.uint8, device="cuda")
    ss.fill_random_device(blob, 0x5EED0001)
    needles = b"".join(absent(16, 0x5EED0003 + i) for i in range(count))
    nblob = torch.from_numpy(np.frombuffer(needles, dtype=np.uint8).copy()).cuda()
    hay_off = (torch.arange(count + 1, dtype=torch.int64) * each).cuda()
    nd_off = (torch.arange(count + 1, dtype=torch.int64) * 16).cuda()

    text = open(os.path.join(ROOT, "tests/golden/data/i386.txt"), "rb").read()
    words = [w for w in open(os.path.join(ROOT, "tests/golden/data/words.txt"), "rb").read().split(b"\n") if w]
    thay = torch.from_numpy(np.frombuffer(text, dtype=np.uint8).copy()).cuda()
    wblob = torch.from_numpy(np.frombuffer(b"".join(words), dtype=np.uint8).copy()).cuda()
    woff = np.zeros(len(words) + 1, dtype=np.int64)
    woff[1:] = np.cumsum([len(w) for w in words])
    wb, we = torch.from_numpy(woff[:-1].copy()).cuda(), torch.from_numpy(woff[1:].copy()).cuda()
    hb = torch.zeros(len(words), dtype=torch.int64, device="cuda")
    he = torch.full((len(words),), len(text), dtype=torch.int64, device="cuda")

    for t in targets:
        if t:
            os.environ["SLICESLICE_BATCH_WGS"] = str(t)
        else:
            os.environ.pop("SLICESLICE_BATCH_WGS", None)
        f = ss.search_batched(blob, hay_off, nblob, nd_off)
        assert int(f.sum().item()) == 0
        med, mn = timed(lambda: ss.search_batched(blob, hay_off, nblob, nd_off), 15)
        print(json.dumps({"case": "config5", "wg_target": t, "ms": round(med, 4), "min_ms": round(mn, 4),
                          "gbps": round(count * each / med / 1e6, 1)}), flush=True)
        f = ss.search_batched(thay, None, wblob, None, hay_ranges=(hb, he), needle_ranges=(wb, we))
        assert int(f.sum().item()) == len(words)
        med, mn = timed(lambda: ss.search_batched(thay, None, wblob, None, hay_ranges=(hb, he), needle_ranges=(wb, we)), 15)
        print(json.dumps({"case": "config1-batched", "wg_target": t, "ms": round(med, 4), "min_ms": round(mn, 4)}), flush=True)


if __name__ == "__main__":
    main()
