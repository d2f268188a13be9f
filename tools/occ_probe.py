#!/usr/bin/env python3
"""Workgroups per CU (capped through unused dynamic LDS: variant 10000*OCC + 41) x haystack size x workload, in ONE process on one
buffer: the scan kernels need ~80 VGPRs since the cold half of the Problem left the registers, so the register file no longer
holds them to four workgroups per CU by itself.    python tools/occ_probe.py [--gib 1,8,32] [--occ 0,4,5,6]"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import sliceslice_rs_amd as ss  # noqa: E402


def kernel_ms(s, hay, reps=15):
    s.set_timing(True)
    res = s.search_in(hay)
    t_end = time.perf_counter() + 0.05
    while time.perf_counter() < t_end:
        s.search_in(hay)
    ms = []
    for _ in range(reps):
        s.search_in(hay)
        ms.append(s.last_kernel_ms())
    return res, float(np.median(ms)), float(np.min(ms))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gib", default="1,8,32")
    ap.add_argument("--occ", default="0,4,5,6")
    ap.add_argument("--rounds", type=int, default=1, help="repeat every sweep this many times (A, B, A, B ...: exposes drift)")
    args = ap.parse_args()
    gibs = [float(x) for x in args.gib.split(",")]
    occs = [int(x) for x in args.occ.split(",")]
    big = int(max(gibs) * (1 << 30))
    hay = torch.empty(big, dtype=torch.uint8, device="cuda")
    ss.fill_random_device(hay, 0x5EED0001)
    nd = bytearray(ss.fill_random_host(16, 0x5EED0002).tobytes())
    nd[8] = 0xFF
    cases = [("random16", bytes(nd), None), ("onebyte", b"\xff", None)]
    for gib in gibs:
        n = int(gib * (1 << 30))
        for name, needle, _ in cases:
            for occ in occs * args.rounds:
                s = ss.DynamicHipSearcher.new(needle)
                s.set_variant(occ * 10000 + 41 if occ else 0)
                res, med, mn = kernel_ms(s, hay[:n])
                assert res is False
                print(json.dumps({"case": name, "gib": gib, "occ": occ or "auto", "ms": round(med, 4), "gbps": round(n / med / 1e6, 1),
                                  "gbps_best": round(n / mn / 1e6, 1)}), flush=True)
    del hay
    gd = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "data")
    raw = np.frombuffer(open(os.path.join(gd, "i386.txt"), "rb").read(), dtype=np.uint8)
    reps = (1 << 30) // raw.size + 1
    text = torch.from_numpy(np.tile(raw, reps)[: 1 << 30].copy()).cuda()
    phrases = [b"segment descriptor table entries are", b" the quick brown fox ", b"protection exception handler must", b"privilege level zero!",
               b"there is not another one of these", b"instruction", b"the"]
    for ph in phrases:
        for mode in ("new", "refpair"):
            for occ in occs * args.rounds:
                s = ss.DynamicHipSearcher.new(ph)
                if mode == "refpair":
                    s.set_filter(0, len(ph) - 1)
                s.set_variant(occ * 10000 + 41 if occ else 0)
                res, med, mn = kernel_ms(s, text)
                wg = s.last_launch()[0]
                cen = s.census(text) if ss.lib().has_hooks else None
                print(json.dumps({"case": "text:" + ph.decode(), "mode": mode, "occ": occ or "auto", "found": res, "ms": round(med, 4),
                                  "gbps": round(text.numel() / med / 1e6, 1), "gbps_best": round(text.numel() / mn / 1e6, 1),
                                  "filter": list(s.filter3), "chosen": wg, "census": cen}), flush=True)


if __name__ == "__main__":
    main()
