#!/usr/bin/env python3
"""ss_search_batched on config 5 (4096 x 1 MiB, one launch) and on the config-1 loop as one launch (4,585 needles x i386.txt),
warmed up, median of 15 launches by events on the launch stream.  SLICESLICE_BATCH_WGS=N in the environment sets the total
number of workgroups (one process per setting): `for w in 16384 24576 32768; do SLICESLICE_BATCH_WGS=$w python tools/batch_probe.py; done`."""
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import sliceslice_rs_amd as ss  # noqa: E402


def timed(fn, reps=15):
    t_end = time.perf_counter() + 0.1
    while time.perf_counter() < t_end:
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ms = []
    for _ in range(reps):
        e0.record()
        out = fn()
        e1.record()
        e1.synchronize()
        ms.append(e0.elapsed_time(e1))
    return out, float(np.median(ms))


def main():
    count, each = 4096, 1 << 20
    blob = torch.empty(count * each, dtype=torch.uint8, device="cuda")
    ss.fill_random_device(blob, 0x5EED0001)
    nd = bytearray(ss.fill_random_host(16 * count, 0x5EED0003).tobytes())
    for i in range(count):
        nd[16 * i + 8] = 0xFF
    nblob = torch.from_numpy(np.frombuffer(bytes(nd), dtype=np.uint8).copy()).cuda()
    hay_off = (torch.arange(count + 1, dtype=torch.int64) * each).cuda()
    nd_off = (torch.arange(count + 1, dtype=torch.int64) * 16).cuda()
    found, ms = timed(lambda: ss.search_batched(blob, hay_off, nblob, nd_off))
    assert int(found.sum().item()) == 0
    out = {"wgs": os.environ.get("SLICESLICE_BATCH_WGS", "auto"),
           "config5_ms": round(ms, 4), "config5_gbps": round(count * each / ms / 1e6, 1)}
    gd = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "data")
    raw = open(os.path.join(gd, "i386.txt"), "rb").read()
    words = [w for w in open(os.path.join(gd, "words.txt"), "rb").read().split(b"\n") if w]
    i386 = torch.from_numpy(np.frombuffer(raw, dtype=np.uint8).copy()).cuda()
    lens = np.array([len(w) for w in words], dtype=np.int64)
    nb = np.zeros(len(words), dtype=np.int64)
    nb[1:] = np.cumsum(lens)[:-1]
    wb = torch.from_numpy(np.frombuffer(b"".join(words), dtype=np.uint8).copy()).cuda()
    nbt, net = torch.from_numpy(nb).cuda(), torch.from_numpy(nb + lens).cuda()
    hb = torch.zeros(len(words), dtype=torch.int64, device="cuda")
    he = torch.full((len(words),), len(raw), dtype=torch.int64, device="cuda")
    found, ms = timed(lambda: ss.search_batched(i386, None, wb, None, hay_ranges=(hb, he), needle_ranges=(nbt, net)))
    out.update(i386_hits=int(found.sum().item()), i386_ms=round(ms, 4))
    print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
