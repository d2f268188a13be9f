#!/usr/bin/env python3
"""One scan of one (haystack kind, needle) case - a target for rocprofv3 --pmc runs.
  kinds: random | text | a      needle: python bytes literal, e.g. "b'ab' + b'a'*14"   [position]"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import sliceslice_rs_amd as ss  # noqa: E402

kind, needle = sys.argv[1], eval(sys.argv[2])
pos = int(sys.argv[3]) if len(sys.argv) > 3 else None
n_bytes = 1 << 30
if kind == "random":
    hay = torch.empty(n_bytes, dtype=torch.uint8, device="cuda")
    ss.fill_random_device(hay, 0x5EED0001)
elif kind == "a":
    hay = torch.full((n_bytes,), 0x61, dtype=torch.uint8, device="cuda")
else:
    raw = np.frombuffer(open(os.path.join(os.path.dirname(__file__), "..", "tests", "golden", "data", "i386.txt"), "rb").read(), dtype=np.uint8)
    hay = torch.from_numpy(raw.copy()).cuda().repeat(n_bytes // raw.size)
s = ss.DynamicHipSearcher(needle, pos)
s.set_timing(True)
for _ in range(3):
    r = s.search_in(hay)
print(kind, needle[:24], pos, "found", r, "ms", round(s.last_kernel_ms(), 4), "GB/s", round(hay.numel() / s.last_kernel_ms() / 1e6, 1))
