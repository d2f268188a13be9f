#!/usr/bin/env python3
"""ss_search_batched across problem shapes, the planned form (batch_plan_kernel + scan grid) against the single-kernel form, IN
ONE PROCESS on one set of buffers: SLICESLICE_BATCH_PLAN / _WGS / _MIN_TILES are read by the library on every call.

    python tools/batch_tune.py [--quick] [--wgs 128,256,512] [--min-tiles 1,2,4]

One JSON line per (shape, setting), plan kernel / memset included, by events on the launch stream after a 50 ms spin: `ms` = one call
bracketed on an otherwise idle stream (median of 15; this includes the HOST's launch path - some 15-20 us of Python, ctypes and
runtime between the first event and the first kernel - during which the GPU waits), `steady_ms` = calls issued back to back, per
call (the launch path overlaps the previous call's kernels: what a pipeline of batches sees).  Plus the config-1 loop as one launch
(4,585 needles x i386.txt) per setting.  Kernel-only durations: tools/shape_trace.py under rocprofv3.

The single-kernel form (SLICESLICE_BATCH_PLAN=0) lives in the tuning build only: run with
SLICESLICE_HIP_LIB=sliceslice-rs_amd/csrc/libsliceslice_hip_tuning.so, or pass --default-only for the shipped library."""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import sliceslice_rs_amd as ss  # noqa: E402

SHAPES = [(4096, 1024), (1024, 1024), (256, 4096), (64, 16384), (16384, 256), (65536, 64), (1, 1 << 20), (8, 1 << 19)]


def timed(fn, reps=15):
    t_end = time.perf_counter() + 0.05
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
    # steady state: calls issued back to back, the host's launch latency overlaps the previous call's execution
    steady = []
    for _ in range(3):
        e0.record()
        for _ in range(10):
            out = fn()
        e1.record()
        e1.synchronize()
        steady.append(e0.elapsed_time(e1) / 10)
    return out, float(np.median(ms)), float(np.median(steady))


def settings(wgs_list, mt_list, occ_list):
    for o in occ_list:
        yield {"SLICESLICE_BATCH_PLAN": "0", "SLICESLICE_BATCH_OCC": str(o)}
        for w in wgs_list:
            for m in mt_list:
                yield {"SLICESLICE_BATCH_PLAN": "1", "SLICESLICE_BATCH_WGS": str(w * 256), "SLICESLICE_BATCH_MIN_TILES": str(m),
                       "SLICESLICE_BATCH_OCC": str(o)}


def apply(env):
    for k in ("SLICESLICE_BATCH_PLAN", "SLICESLICE_BATCH_WGS", "SLICESLICE_BATCH_MIN_TILES", "SLICESLICE_BATCH_OCC"):
        os.environ.pop(k, None)
    os.environ.update(env)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--quick", action="store_true")
    ap.add_argument("--wgs", default="128,256,512", help="workgroups per CU (x 256 CUs) for the planned form")
    ap.add_argument("--min-tiles", default="1,2,4")
    ap.add_argument("--occ", default="4", help="workgroups per CU (SLICESLICE_BATCH_OCC)")
    ap.add_argument("--default-only", action="store_true", help="only the library defaults (the shipped behaviour)")
    args = ap.parse_args()
    wgs = [int(x) for x in args.wgs.split(",")]
    mts = [int(x) for x in args.min_tiles.split(",")]
    occs = [int(x) for x in args.occ.split(",")]
    sets = [{}] if args.default_only else list(settings(wgs, mts, occs))
    total = 4 << 30
    hay = torch.empty(total, dtype=torch.uint8, device="cuda")
    ss.fill_random_device(hay, 0x5EED0001)
    shapes = SHAPES[:3] if args.quick else SHAPES
    for count, kib in shapes:
        each = kib << 10
        used = min(total, count * each)
        count = used // each
        nb = bytearray(ss.fill_random_host(16 * count, 0x5EED0003).tobytes())
        for i in range(count):
            nb[16 * i + 8] = 0xFF
        nblob = torch.from_numpy(np.frombuffer(bytes(nb), dtype=np.uint8).copy()).cuda()
        hay_off = (torch.arange(count + 1, dtype=torch.int64) * each).cuda()
        nd_off = (torch.arange(count + 1, dtype=torch.int64) * 16).cuda()
        for env in sets:
            apply(env)
            found, med, mn = timed(lambda: ss.search_batched(hay[:used], hay_off, nblob, nd_off))
            assert int(found.sum().item()) == 0
            print(json.dumps({"shape": "%dx%dKiB" % (count, kib), "bytes": used, **env, "ms": round(med, 4),
                              "gbps": round(used / med / 1e6, 1), "steady_ms": round(mn, 4), "gbps_steady": round(used / mn / 1e6, 1)}), flush=True)
    # the reference's long-haystack loop as ONE launch: every word occurs in the text (early exit matters)
    gd = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "data")
    raw = open(os.path.join(gd, "i386.txt"), "rb").read()
    words = [w for w in open(os.path.join(gd, "words.txt"), "rb").read().split(b"\n") if w]
    i386 = torch.from_numpy(np.frombuffer(raw, dtype=np.uint8).copy()).cuda()
    lens = np.array([len(w) for w in words], dtype=np.int64)
    nbeg = np.zeros(len(words), dtype=np.int64)
    nbeg[1:] = np.cumsum(lens)[:-1]
    wb = torch.from_numpy(np.frombuffer(b"".join(words), dtype=np.uint8).copy()).cuda()
    nbt, net = torch.from_numpy(nbeg).cuda(), torch.from_numpy(nbeg + lens).cuda()
    hb = torch.zeros(len(words), dtype=torch.int64, device="cuda")
    he = torch.full((len(words),), len(raw), dtype=torch.int64, device="cuda")
    for env in sets:
        apply(env)
        found, med, mn = timed(lambda: ss.search_batched(i386, None, wb, None, hay_ranges=(hb, he), needle_ranges=(nbt, net)))
        print(json.dumps({"shape": "i386 loop, 4585 needles x 857425 B", **env, "hits": int(found.sum().item()),
                          "ms": round(med, 4), "steady_ms": round(mn, 4)}), flush=True)


if __name__ == "__main__":
    main()
