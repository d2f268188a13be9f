#!/usr/bin/env python3
"""Workgroups per CU: the automatic choice (candidate census, ss_scan.hip) against FORCED four and six (capped through unused
dynamic LDS: variant 10000*OCC + 41) x haystack size x workload, in ONE process on one buffer.  The three searchers of a row take
turns (auto, four, six, auto, ...: rates measured minutes apart differ by 2-3 % from drift alone), so `auto_over_best` compares
like with like; `chosen` is what the automatic searcher launched with (ss_searcher_last_launch).  Needs the hooks build for the
forced settings:

    SLICESLICE_HIP_LIB=sliceslice-rs_amd/csrc/libsliceslice_hip_tuning.so python tools/occ_probe.py [--gib 1,8]"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import sliceslice_rs_amd as ss  # noqa: E402


def paired_ms(searchers, hay, rounds=4, reps=6):
    got = [[] for _ in searchers]
    for s in searchers:
        s.set_timing(True)
        s.search_in(hay)
    t_end = time.perf_counter() + 0.05
    while time.perf_counter() < t_end:
        searchers[0].search_in(hay)
    res = None
    for _ in range(rounds):
        for k, s in enumerate(searchers):
            res = s.search_in(hay)
            for _ in range(reps):
                s.search_in(hay)
                got[k].append(s.last_kernel_ms())
    return res, [float(np.median(g)) for g in got]


def row(make, hay, **kw):
    sa, s4, s6 = make(0), make(40041), make(60041)
    res, (ma, m4, m6) = paired_ms([sa, s4, s6], hay)
    n = hay.numel()
    out = dict(kw, found=res, gbps_auto=round(n / ma / 1e6, 1), gbps_four=round(n / m4 / 1e6, 1), gbps_six=round(n / m6 / 1e6, 1),
               auto_over_best=round(min(m4, m6) / ma, 4), chosen=sa.last_launch()[0], filter=list(sa.filter3),
               census=sa.census(hay) if ss.lib().has_hooks else None)
    print(json.dumps(out), flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gib", default="1,8")
    args = ap.parse_args()
    gibs = [float(x) for x in args.gib.split(",")]
    big = int(max(gibs) * (1 << 30))
    hay = torch.empty(big, dtype=torch.uint8, device="cuda")
    ss.fill_random_device(hay, 0x5EED0001)
    nd = bytearray(ss.fill_random_host(16, 0x5EED0002).tobytes())
    nd[8] = 0xFF

    def maker(needle, refpair=False):
        def make(variant):
            s = ss.DynamicHipSearcher.new(needle)
            if refpair:
                s.set_filter(0, len(needle) - 1)
            s.set_variant(variant)
            return s
        return make
    for gib in gibs:
        n = int(gib * (1 << 30))
        for name, needle in (("random16", bytes(nd)), ("onebyte", b"\xff"), ("text-like needle on random bytes", b"there is not another one of these")):
            row(maker(needle), hay[:n], case=name, gib=gib)
    del hay
    gd = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "data")
    raw = np.frombuffer(open(os.path.join(gd, "i386.txt"), "rb").read(), dtype=np.uint8)
    text = torch.from_numpy(np.tile(raw, (1 << 30) // raw.size + 1)[: 1 << 30].copy()).cuda()
    phrases = [b"segment descriptor table entries are", b" the quick brown fox ", b"protection exception handler must", b"privilege level zero!",
               b"there is not another one of these", b"instruction", b"the"]
    for ph in phrases:
        for mode in ("new", "refpair"):
            row(maker(ph, mode == "refpair"), text, case="text:" + ph.decode(), mode=mode)


if __name__ == "__main__":
    main()
