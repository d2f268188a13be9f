#!/usr/bin/env python3
"""Candidate census vs the better workgroups-per-CU setting: for many absent phrases on 1 GiB of text (the i386 manual tiled) and a
few needles on random bytes, the kernel time at FORCED four and six workgroups per CU (interleaved, in one process, on one buffer)
next to what the searcher's census counted on that haystack (ss_debug_census) and what the automatic choice launches with.  The
data the thresholds of ss_scan.hip::census_choice are read from.  Needs the hooks build:

    SLICESLICE_HIP_LIB=sliceslice-rs_amd/csrc/libsliceslice_hip_tuning.so python tools/occ_census.py [--phrases 40] [--seed 1]
"""
import argparse
import json
import os
import random
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import sliceslice_rs_amd as ss  # noqa: E402


def spin(s, hay, seconds=0.03):
    t_end = time.perf_counter() + seconds
    while time.perf_counter() < t_end:
        s.search_in(hay)


def paired_ms(searchers, hay, rounds=3, reps=7):
    """Median kernel ms per searcher, the searchers taking turns (A B A B ...) so that drift hits both alike."""
    got = [[] for _ in searchers]
    for s in searchers:
        s.set_timing(True)
    spin(searchers[0], hay)
    for _ in range(rounds):
        for k, s in enumerate(searchers):
            s.search_in(hay)
            for _ in range(reps):
                s.search_in(hay)
                got[k].append(s.last_kernel_ms())
    return [float(np.median(g)) for g in got]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--phrases", type=int, default=40)
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--gib", type=float, default=1.0)
    args = ap.parse_args()
    assert ss.lib().has_hooks, "needs the hooks build (SLICESLICE_HIP_LIB=...libsliceslice_hip_tuning.so)"
    rng = random.Random(args.seed)
    nbytes = int(args.gib * (1 << 30))
    gd = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "data")
    raw = open(os.path.join(gd, "i386.txt"), "rb").read()
    text = torch.from_numpy(np.tile(np.frombuffer(raw, dtype=np.uint8), nbytes // len(raw) + 1)[:nbytes].copy()).cuda()
    rnd = torch.empty(nbytes, dtype=torch.uint8, device="cuda")
    ss.fill_random_device(rnd, 0x5EED0001)

    def absent_variant(ph):
        """`ph` with one letter swapped for another common letter so that it no longer occurs in the text."""
        for _ in range(50):
            k = rng.randrange(len(ph))
            for c in b"etaoin srhl":
                cand = ph[:k] + bytes([c]) + ph[k + 1:]
                if cand != ph and cand not in raw:
                    return cand
        return None

    cases = []
    fixed = [b"segment descriptor table entries are", b" the quick brown fox ", b"protection exception handler must", b"privilege level zero!",
             b"there is not another one of these"]
    for ph in fixed:
        cases.append(("text", ph))
    while len(cases) < args.phrases:
        n = rng.choice((4, 6, 8, 10, 12, 16, 16, 20, 24, 32, 40))
        at = rng.randrange(len(raw) - n)
        ph = absent_variant(raw[at:at + n])
        if ph:
            cases.append(("text", ph))
    nd = bytearray(ss.fill_random_host(16, 0x5EED0002).tobytes())
    nd[8] = 0xFF
    cases += [("random", bytes(nd)), ("random", b"privilege level zero!"), ("random", b"\x00\x00\x00\x00\x00\x00\x00\xff")]

    for kind, ph in cases:
        hay = text if kind == "text" else rnd
        for mode in ("new", "refpair"):
            if mode == "refpair" and len(ph) < 3:
                continue
            mk = []
            for variant in (40041, 60041, 0):
                s = ss.DynamicHipSearcher.new(ph)
                if mode == "refpair":
                    s.set_filter(0, len(ph) - 1)
                s.set_variant(variant)
                mk.append(s)
            s4, s6, sa = mk
            found = sa.search_in(hay)
            m4, m6, ma = paired_ms([s4, s6, sa], hay)
            cen = sa.census(hay)
            print(json.dumps({"kind": kind, "needle": ph.decode("latin1"), "n": len(ph), "mode": mode, "found": found, "filter": list(sa.filter3),
                              "ms4": round(m4, 4), "ms6": round(m6, 4), "ms_auto": round(ma, 4), "auto_wg": sa.last_launch()[0],
                              "six_over_four": round(m4 / m6, 4), "gbps4": round(nbytes / m4 / 1e6, 1), "gbps6": round(nbytes / m6 / 1e6, 1),
                              "census": cen}), flush=True)


if __name__ == "__main__":
    main()
