#!/usr/bin/env python3
"""Per-call latency of the device-haystack entry points on a small (cache-resident) haystack."""
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import sliceslice_rs_amd as ss  # noqa: E402

hay = torch.empty(1 << 20, dtype=torch.uint8, device="cuda")
ss.fill_random_device(hay, 0x5EED0001)
nd = bytearray(ss.fill_random_host(16, 0x5EED0002).tobytes())
nd[8] = 0xFF
s = ss.DynamicHipSearcher.new(bytes(nd))
present = hay[5000:5016].cpu().numpy().tobytes()
p = ss.DynamicHipSearcher.new(present)
for label, fn in (("search_in absent", lambda: s.search_in(hay)), ("search_in present", lambda: p.search_in(hay)),
                  ("find absent", lambda: s.find(hay)), ("find present", lambda: p.find(hay))):
    for _ in range(20):
        fn()
    t = time.perf_counter()
    for _ in range(500):
        r = fn()
    dt = (time.perf_counter() - t) / 500
    print(json.dumps({"call": label, "haystack_bytes": hay.numel(), "result": r, "us_per_call": round(dt * 1e6, 1)}), flush=True)
