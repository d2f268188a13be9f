#!/usr/bin/env python3
"""In-process A/B of library builds: every build is dlopen()ed (RTLD_LOCAL) into ONE process and scans the SAME
device buffer, launches interleaved build by build - haystack placement, clocks and box are identical, so
differences of a percent are visible (separate processes differ by +-2 % from placement alone).
    python tools/ab_inproc.py --libs cur=...so two=...so --gib 8 --cases n16,n1,tworst
    python tools/ab_inproc.py --libs a=lib.so#0-15-14 b=lib.so#0-15-1 --cases n16      (one build, two filter triples)"""
import argparse
import ctypes
import json
import os
import statistics
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import sliceslice_rs_amd as ss  # noqa: E402  (the synthetic haystack generator: libsliceslice_hip_tools.so)

vp, sz = ctypes.c_void_p, ctypes.c_size_t


def load(path):
    L = ctypes.CDLL(path, mode=os.RTLD_LOCAL | os.RTLD_NOW)
    L.ss_searcher_new.argtypes = [vp, sz, ctypes.POINTER(vp)]
    L.ss_search_device.argtypes = [vp, vp, sz, vp, ctypes.POINTER(ctypes.c_int)]
    if hasattr(L, "ss_searcher_last_launch"):
        L.ss_searcher_last_launch.argtypes = [vp, ctypes.POINTER(ctypes.c_int), ctypes.POINTER(ctypes.c_uint)]
    L.ss_searcher_set_timing.argtypes = [vp, ctypes.c_int]
    L.ss_searcher_last_kernel_ms.argtypes = [vp, ctypes.POINTER(ctypes.c_float)]
    L.ss_searcher_free.argtypes = [vp]
    L.ss_last_error.restype = ctypes.c_char_p
    return L


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--libs", nargs="+", required=True)
    ap.add_argument("--gib", type=float, default=8.0)
    ap.add_argument("--rounds", type=int, default=6)
    ap.add_argument("--reps", type=int, default=8)
    ap.add_argument("--cases", default="n16,n1,n2000,tworst,tspaces")
    args = ap.parse_args()
    # name=path[@variant[:grid]]: the same build can appear several times with different kernel-variant / grid overrides
    # ...#a-b-c: the filter triple (ss_searcher_set_filter3; a-b: a pair) instead of the constructor's choice
    libs, variants, grids, filters = [], {}, {}, {}
    for l in args.libs:
        name, rest = l.split("=", 1)
        rest, _, flt = rest.partition("#")
        filters[name] = [int(x) for x in flt.split("-")] if flt else None
        path, _, var = rest.partition("@")
        var, _, grid = var.partition(":")
        libs.append((name, load(path)))
        variants[name] = int(var) if var else 0
        grids[name] = int(grid) if grid else 0
    n_bytes = int(args.gib * (1 << 30))
    hay = torch.empty(n_bytes, dtype=torch.uint8, device="cuda")
    ss.fill_random_device(hay, 0x5EED0001)
    torch.cuda.synchronize()
    text = None

    def absent(n):
        a = ss.fill_random_host(n, 0x5EED0002)
        a[0 if n == 1 else (1 if n == 2 else n // 2)] = 0xFF
        return a.tobytes()
    table = {}
    for c in args.cases.split(","):
        if c.startswith("n"):                             # n16, nref2000 (reference pair), nwp2000 (with_position(n-1))
            nd, h = absent(int(c.lstrip("nrefwp"))), hay
        else:
            if text is None:
                raw = np.frombuffer(open(os.path.join(ROOT, "tests", "golden", "data", "i386.txt"), "rb").read(), dtype=np.uint8)
                text = torch.from_numpy(raw.copy()).cuda().repeat((1 << 30) // raw.size)
            nd = {"twpworst": b"segment descriptor table entries are", "twpspaces": b" the quick brown fox ",
                  "tworst": b"segment descriptor table entries are", "tspaces": b" the quick brown fox ", "tpriv": b"privilege level zero!",
                  "tcommon": b"there is not another one of these", "tmid": b"protection exception handler must",
                  "trefspaces": b" the quick brown fox ", "trefworst": b"segment descriptor table entries are", "trefshort": b" quick fox "}[c]
            h = text
        hs = []
        for name, L in libs:
            s = vp()
            if c.startswith("tref") or c.startswith("nref"):    # the reference's pair (0, n-1), verbatim
                assert L.ss_searcher_new(nd, len(nd), ctypes.byref(s)) == 0, L.ss_last_error()
                L.ss_searcher_set_filter3.argtypes = [vp, sz, sz, sz]
                assert L.ss_searcher_set_filter3(s, 0, len(nd) - 1, len(nd) - 1) == 0, L.ss_last_error()
            elif c.startswith("twp") or c.startswith("nwp"):    # with_position(n-1)
                L.ss_searcher_with_position.argtypes = [vp, sz, sz, ctypes.POINTER(vp)]
                assert L.ss_searcher_with_position(nd, len(nd), len(nd) - 1, ctypes.byref(s)) == 0, L.ss_last_error()
            else:
                assert L.ss_searcher_new(nd, len(nd), ctypes.byref(s)) == 0, L.ss_last_error()
            L.ss_searcher_set_timing(s, 1)
            if filters[name]:
                f = filters[name]
                L.ss_searcher_set_filter3.argtypes = [vp, sz, sz, sz]
                rc = L.ss_searcher_set_filter3(s, f[0], f[1], f[1] if len(f) == 2 else f[2])
                assert rc == 0, L.ss_last_error()
            if variants[name]:
                L.ss_searcher_set_variant.argtypes = [vp, ctypes.c_int]
                assert L.ss_searcher_set_variant(s, variants[name]) == 0
            if grids[name]:
                L.ss_searcher_set_grid.argtypes = [vp, ctypes.c_int]
                assert L.ss_searcher_set_grid(s, grids[name]) == 0
            hs.append((name, L, s))
        found, ms = ctypes.c_int(0), ctypes.c_float(0)
        t_end = time.perf_counter() + 0.2
        while time.perf_counter() < t_end:
            for name, L, s in hs:
                L.ss_search_device(s, h.data_ptr(), h.numel(), None, ctypes.byref(found))
        acc = {name: [] for name, _, _ in hs}
        for r in range(args.rounds):
            for name, L, s in hs:
                for _ in range(args.reps):
                    assert L.ss_search_device(s, h.data_ptr(), h.numel(), None, ctypes.byref(found)) == 0
                    L.ss_searcher_last_kernel_ms(s, ctypes.byref(ms))
                    acc[name].append(ms.value)
        row = {name: round(h.numel() / statistics.median(v) / 1e6, 1) for name, v in acc.items()}
        table[c] = row
        occ = {}
        for name, L, s in hs:                             # what the searcher launched with (libraries from round 5 on)
            if hasattr(L, "ss_searcher_last_launch"):
                w, g = ctypes.c_int(0), ctypes.c_uint(0)
                L.ss_searcher_last_launch(s, ctypes.byref(w), ctypes.byref(g))
                occ[name] = [w.value, g.value]
        print(json.dumps({"case": c, "found": found.value, "gbps": row, "workgroups_per_cu,grid": occ}), flush=True)
        for name, L, s in hs:
            L.ss_searcher_free(s)
    print(json.dumps({"median_gbps": table}), flush=True)


if __name__ == "__main__":
    main()
