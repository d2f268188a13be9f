#!/usr/bin/env python3
"""Does the scan rate depend on WHERE a buffer sits?  Allocates several haystack-sized buffers in one process and
scans each (kernel-only GB/s, median of 7 after a settle phase).  Investigation aid for the few-percent
process-to-process spread of the headline number."""
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import sliceslice_rs_amd as ss  # noqa: E402
from settle import wait_for_vram_reclaim  # noqa: E402

wait_for_vram_reclaim()

gib = float(sys.argv[1]) if len(sys.argv) > 1 else 64.0
count = int(sys.argv[2]) if len(sys.argv) > 2 else 3
n_bytes = int(gib * (1 << 30))
nd = bytearray(ss.fill_random_host(16, 0x5EED0002).tobytes())
nd[8] = 0xFF
s = ss.DynamicHipSearcher.new(bytes(nd))
s.set_timing(True)
bufs = []
for i in range(count):
    b = torch.empty(n_bytes, dtype=torch.uint8, device="cuda")
    ss.fill_random_device(b, 0x5EED0001)
    bufs.append(b)
torch.cuda.synchronize()
t_end = time.perf_counter() + 0.3
while time.perf_counter() < t_end:
    s.search_in(bufs[0])
for rnd in range(2):
    for i, b in enumerate(bufs):
        ms = []
        for _ in range(7):
            s.search_in(b)
            ms.append(s.last_kernel_ms())
        med = float(np.median(ms))
        print(json.dumps({"round": rnd, "buffer": i, "address": hex(b.data_ptr()), "gib": gib, "ms": round(med, 4),
                          "gbps": round(n_bytes / med / 1e6, 1)}), flush=True)
