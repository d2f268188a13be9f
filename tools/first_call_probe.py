#!/usr/bin/env python3
"""What the FIRST searches of a (searcher, haystack) pair cost as a caller sees them: wall-clock of synchronous search_in calls
1 ... 12 of fresh searchers (median over needles) on haystacks of 256 MiB / 1 GiB / 4 GiB of random bytes and 1 GiB of text, for
one or more library builds (SLICESLICE_HIP_LIB per process: run once per build).
    python tools/first_call_probe.py > out.jsonl"""
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import sliceslice_rs_amd as ss  # noqa: E402


def main():
    gd = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "data")
    raw = open(os.path.join(gd, "i386.txt"), "rb").read()
    big = torch.empty(4 << 30, dtype=torch.uint8, device="cuda")
    ss.fill_random_device(big, 0x5EED0001)
    text = torch.from_numpy(np.tile(np.frombuffer(raw, dtype=np.uint8), (1 << 30) // len(raw) + 1)[: 1 << 30].copy()).cuda()
    warm = ss.DynamicHipSearcher.new(b"\xff" * 16)
    for _ in range(50):
        warm.search_in(big[: 1 << 30])
    phrases = [b"segment descriptor table entries are", b"privilege level zero!", b"there is not another one of these", b" the quick brown fox ",
               b"protection exception handler must", b"no such needle in here", b"the register o", b"eginneng of\na su"]
    for name, hay in (("random 256 MiB", big[: 256 << 20]), ("random 1 GiB", big[: 1 << 30]), ("random 4 GiB", big), ("text 1 GiB", text)):
        calls = [[] for _ in range(12)]
        for k in range(8):
            nd = bytearray(ss.fill_random_host(16, 0xABC0 + k).tobytes())
            nd[8] = 0xFF
            s = ss.DynamicHipSearcher.new(phrases[k] if name.startswith("text") else bytes(nd))
            torch.cuda.synchronize()
            for i in range(12):
                t0 = time.perf_counter()
                s.search_in(hay)
                calls[i].append((time.perf_counter() - t0) * 1e6)
        print(json.dumps({"haystack": name, "lib": os.environ.get("SLICESLICE_HIP_LIB", "product"),
                          "call_us_median": [round(float(np.median(c)), 1) for c in calls]}), flush=True)


if __name__ == "__main__":
    main()
