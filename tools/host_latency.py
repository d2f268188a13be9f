#!/usr/bin/env python3
"""Per-call latency of ss_search_host (the drop-in `search_in(&[u8])` shape) by haystack size."""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import sliceslice_rs_amd as ss  # noqa: E402

nd = bytearray(ss.fill_random_host(16, 0x5EED0002).tobytes())
nd[8] = 0xFF
s = ss.DynamicHipSearcher.new(bytes(nd))
for size in (1 << 10, 64 << 10, 1 << 20, 16 << 20, 64 << 20, 256 << 20):
    host = ss.fill_random_host(size, 0x5EED0001)
    for _ in range(3):
        assert s.search_in(host) is False
    reps = 200 if size <= (1 << 20) else 20
    t = time.perf_counter()
    for _ in range(reps):
        s.search_in(host)
    dt = (time.perf_counter() - t) / reps
    print(json.dumps({"entry": "ss_search_host", "bytes": size, "us_per_call": round(dt * 1e6, 1),
                      "gbps": round(size / dt / 1e9, 2)}), flush=True)
