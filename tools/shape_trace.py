#!/usr/bin/env python3
"""Kernel-only durations (run under `rocprofv3 --kernel-trace`; tools/trace_summary.py groups the trace by kernel and grid):
the single-problem scan at 1..4 tiles per workgroup against both forms of ss_search_batched at several grid sizes, same bytes.
    python tools/shape_trace.py COUNT EACH_KIB"""
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


def spin(fn, n=20):
    t_end = time.perf_counter() + 0.05
    while time.perf_counter() < t_end:
        fn()
    torch.cuda.synchronize()
    for _ in range(n):
        fn()
        torch.cuda.synchronize()


s = ss.DynamicHipSearcher.new(bytes(nb[:16]))
for tpb in (1, 2, 3, 4):
    s.set_grid(-tpb)
    spin(lambda: s.search_in(hay))
for env in ([("SLICESLICE_BATCH_PLAN", "0")],
            [("SLICESLICE_BATCH_PLAN", "1"), ("SLICESLICE_BATCH_WGS", str(96 * 256)), ("SLICESLICE_BATCH_MIN_TILES", "1")],
            [("SLICESLICE_BATCH_PLAN", "1"), ("SLICESLICE_BATCH_WGS", str(160 * 256)), ("SLICESLICE_BATCH_MIN_TILES", "1")],
            [("SLICESLICE_BATCH_PLAN", "1"), ("SLICESLICE_BATCH_WGS", str(256 * 256)), ("SLICESLICE_BATCH_MIN_TILES", "1")],
            [("SLICESLICE_BATCH_PLAN", "1"), ("SLICESLICE_BATCH_WGS", str(256 * 256)), ("SLICESLICE_BATCH_MIN_TILES", "2")],
            [("SLICESLICE_BATCH_PLAN", "1"), ("SLICESLICE_BATCH_WGS", str(512 * 256)), ("SLICESLICE_BATCH_MIN_TILES", "1")]):
    for k in ("SLICESLICE_BATCH_PLAN", "SLICESLICE_BATCH_WGS", "SLICESLICE_BATCH_MIN_TILES"):
        os.environ.pop(k, None)
    os.environ.update(dict(env))
    spin(lambda: ss.search_batched(hay, hay_off, nblob, nd_off))
