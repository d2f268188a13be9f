#!/usr/bin/env python3
"""30 ss_search_batched launches of ONE shape under `rocprofv3 --kernel-trace` (both forms of the call), to see where a call's time
goes: kernel durations vs the gaps between the dependent commands of one call.
    rocprofv3 --kernel-trace --output-format csv -d gpurun_out/bt -- python tools/batch_trace.py 1024 1024"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import sliceslice_rs_amd as ss  # noqa: E402

count, kib = int(sys.argv[1]), int(sys.argv[2])
each = kib << 10
hay = torch.empty(count * each, dtype=torch.uint8, device="cuda")
ss.fill_random_device(hay, 0x5EED0001)
nb = bytearray(ss.fill_random_host(16 * count, 0x5EED0003).tobytes())
for i in range(count):
    nb[16 * i + 8] = 0xFF
nblob = torch.from_numpy(np.frombuffer(bytes(nb), dtype=np.uint8).copy()).cuda()
hay_off = (torch.arange(count + 1, dtype=torch.int64) * each).cuda()
nd_off = (torch.arange(count + 1, dtype=torch.int64) * 16).cuda()
for plan in ("0", "1"):
    os.environ["SLICESLICE_BATCH_PLAN"] = plan
    t_end = time.perf_counter() + 0.05
    while time.perf_counter() < t_end:
        ss.search_batched(hay, hay_off, nblob, nd_off)
    torch.cuda.synchronize()
    for _ in range(30):
        found = ss.search_batched(hay, hay_off, nblob, nd_off)
        torch.cuda.synchronize()
    assert int(found.sum().item()) == 0
