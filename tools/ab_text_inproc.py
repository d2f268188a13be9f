#!/usr/bin/env python3
"""In-process A/B of library builds over MANY text cases: every build is dlopen()ed (RTLD_LOCAL) into one process and scans the same
device buffer (the i386 manual tiled), the (phrase, constructor) cases of a tools/survival_probe.py run; each build's searcher
settles first (launch tuning on), then the builds take turns.  Per case the kernel time of every build (hipEvents on the launch
stream) and what each launched with; at the end the geometric mean of the time ratios against the first build.
    python tools/ab_text_inproc.py --libs cur=...so pre=...so --rows profiles/r06/survival_probe_final.jsonl [--gib 1]"""
import argparse
import ctypes
import json
import math
import os
import statistics
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

vp, sz = ctypes.c_void_p, ctypes.c_size_t


def load(path):
    L = ctypes.CDLL(path, mode=os.RTLD_LOCAL | os.RTLD_NOW)
    L.ss_searcher_new.argtypes = [vp, sz, ctypes.POINTER(vp)]
    L.ss_searcher_with_position.argtypes = [vp, sz, sz, ctypes.POINTER(vp)]
    L.ss_searcher_set_filter3.argtypes = [vp, sz, sz, sz]
    L.ss_search_device.argtypes = [vp, vp, sz, vp, ctypes.POINTER(ctypes.c_int)]
    L.ss_searcher_last_launch.argtypes = [vp, ctypes.POINTER(ctypes.c_int), ctypes.POINTER(ctypes.c_uint)]
    L.ss_searcher_set_timing.argtypes = [vp, ctypes.c_int]
    L.ss_searcher_last_kernel_ms.argtypes = [vp, ctypes.POINTER(ctypes.c_float)]
    L.ss_searcher_free.argtypes = [vp]
    L.ss_last_error.restype = ctypes.c_char_p
    return L


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--libs", nargs="+", required=True)
    ap.add_argument("--rows", required=True)
    ap.add_argument("--gib", type=float, default=1.0)
    ap.add_argument("--rounds", type=int, default=4)
    ap.add_argument("--reps", type=int, default=6)
    ap.add_argument("--limit", type=int, default=0)
    args = ap.parse_args()
    libs = [(l.split("=", 1)[0], load(l.split("=", 1)[1])) for l in args.libs]
    nbytes = int(args.gib * (1 << 30))
    raw = open(os.path.join(ROOT, "tests", "golden", "data", "i386.txt"), "rb").read()
    text = torch.from_numpy(np.tile(np.frombuffer(raw, dtype=np.uint8), nbytes // len(raw) + 1)[:nbytes].copy()).cuda()
    cases, seen = [], set()
    for line in open(args.rows):
        r = json.loads(line)
        if r["kind"] == "text" and (r["needle"], r["mode"]) not in seen:
            seen.add((r["needle"], r["mode"]))
            cases.append((r["needle"].encode("latin1"), r["mode"]))
    if args.limit:
        cases = cases[:args.limit]
    found, ms = ctypes.c_int(0), ctypes.c_float(0)
    logsum = {name: 0.0 for name, _ in libs}
    fracs = {name: [] for name, _ in libs}
    for nd, mode in cases:
        hs = []
        for name, L in libs:
            s = vp()
            if mode == "wp":
                assert L.ss_searcher_with_position(nd, len(nd), len(nd) - 1, ctypes.byref(s)) == 0, L.ss_last_error()
            else:
                assert L.ss_searcher_new(nd, len(nd), ctypes.byref(s)) == 0, L.ss_last_error()
                if mode == "refpair":
                    assert L.ss_searcher_set_filter3(s, 0, len(nd) - 1, len(nd) - 1) == 0, L.ss_last_error()
            L.ss_searcher_set_timing(s, 1)
            hs.append((name, L, s))
        for _ in range(20):                               # every handle settles on the haystack
            for name, L, s in hs:
                assert L.ss_search_device(s, text.data_ptr(), nbytes, None, ctypes.byref(found)) == 0
        t_end = time.perf_counter() + 0.03
        while time.perf_counter() < t_end:
            for name, L, s in hs:
                L.ss_search_device(s, text.data_ptr(), nbytes, None, ctypes.byref(found))
        acc = {name: [] for name, _, _ in hs}
        for _ in range(args.rounds):
            for name, L, s in hs:
                L.ss_search_device(s, text.data_ptr(), nbytes, None, ctypes.byref(found))
                for _ in range(args.reps):
                    assert L.ss_search_device(s, text.data_ptr(), nbytes, None, ctypes.byref(found)) == 0
                    L.ss_searcher_last_kernel_ms(s, ctypes.byref(ms))
                    acc[name].append(ms.value)
        med = {name: statistics.median(v) for name, v in acc.items()}
        shape = {}
        for name, L, s in hs:
            w, g = ctypes.c_int(0), ctypes.c_uint(0)
            L.ss_searcher_last_launch(s, ctypes.byref(w), ctypes.byref(g))
            shape[name] = [w.value, g.value]
            L.ss_searcher_free(s)
        base = med[libs[0][0]]
        for name in med:
            logsum[name] += math.log(med[name] / base)
            fracs[name].append(nbytes / med[name] / 1e6 / 8000.0)
        print(json.dumps({"needle": nd.decode("latin1"), "mode": mode, "found": found.value,
                          "frac": {k: round(nbytes / v / 1e6 / 8000.0, 4) for k, v in med.items()}, "workgroups_per_cu,grid": shape}), flush=True)
    n = max(1, len(cases))
    print(json.dumps({"cases": len(cases), "gib": args.gib, "geomean_time_vs_first": {k: round(math.exp(v / n), 4) for k, v in logsum.items()},
                      "median_frac": {k: round(statistics.median(v), 4) for k, v in fracs.items()},
                      "under_0.90": {k: sum(1 for x in v if x < 0.90) for k, v in fracs.items()},
                      "under_0.88": {k: sum(1 for x in v if x < 0.88) for k, v in fracs.items()},
                      "min_frac": {k: round(min(v), 4) for k, v in fracs.items()}}), flush=True)


if __name__ == "__main__":
    main()
