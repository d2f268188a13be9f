#!/usr/bin/env python3
"""Filter bytes chosen from the haystack's own histogram (ss_census.hip) against the static, corpus-free choice of ss_searcher_new:
on a haystack whose most frequent bytes LOOK rare (UTF-8-like text in a non-Latin script) and on the i386 manual, per needle the
kernel time of a `new` searcher (automatic: histogram-driven where it promises 16 x fewer candidates) and of the same needle with
the static triple pinned (set_filter), the two taking turns in one process; with the census counts of both.  Hooks build.
    SLICESLICE_HIP_LIB=...libsliceslice_hip_tuning.so python tools/triple_probe.py [--gib 1]"""
import argparse
import json
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import sliceslice_rs_amd as ss  # noqa: E402
from occ_probe import paired_ms  # noqa: E402


def non_latin(n_bytes, seed):
    rng = np.random.default_rng(seed)
    pairs = n_bytes // 2
    lead = rng.choice(np.array([0xD0, 0xD1], dtype=np.uint8), size=pairs, p=[0.6, 0.4])
    trail = (0x80 + np.minimum(rng.geometric(0.08, size=pairs) - 1, 63)).astype(np.uint8)
    a = np.empty(pairs * 2, dtype=np.uint8)
    a[0::2], a[1::2] = lead, trail
    blanks = rng.integers(0, pairs, size=pairs // 7)
    a[2 * blanks] = 0x20
    a[2 * blanks + 1] = 0x20
    return a


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gib", type=float, default=1.0)
    args = ap.parse_args()
    assert ss.lib().has_hooks
    n = int(args.gib * (1 << 30))
    host = non_latin(n, 7)
    gd = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "data")
    raw = np.frombuffer(open(os.path.join(gd, "i386.txt"), "rb").read(), dtype=np.uint8)
    text = np.tile(raw, n // raw.size + 1)[:n].copy()
    rng = np.random.default_rng(3)
    cases = []
    for k in range(8):                                           # words of the haystack's own alphabet, one trail byte changed: absent
        ln = int(rng.choice([8, 12, 16, 24, 32]))
        at = 2 * int(rng.integers(0, n // 2 - 64))
        w = bytearray(host[at:at + ln].tobytes())
        w[ln // 2 | 1] = w[1] = 0xBF                             # the rarest trail byte of the generator, twice: absent
        cases.append(("non-latin", bytes(w)))
    for ph in (b"segment descriptor table entries are", b"privilege level zero!", b"there is not another one of these"):
        cases.append(("i386 text", ph))
    bufs = {"non-latin": torch.from_numpy(host).cuda(), "i386 text": torch.from_numpy(text).cuda()}
    for kind, nd in cases:
        hay = bufs[kind]
        auto = ss.DynamicHipSearcher.new(nd)
        static = ss.DynamicHipSearcher.new(nd)
        static.set_filter(*static.filter3)                      # the same bytes, pinned: no histogram-driven choice
        for s in (auto, static):
            for _ in range(4):
                s.search_in(hay)
        res, (ma, ms_) = paired_ms([auto, static], hay)
        ca, cs = auto.census(hay), static.census(hay)
        print(json.dumps({"haystack": kind, "needle_len": len(nd), "found": res, "static_triple": list(static.filter3), "device_triple": list(auto.device_filter),
                          "triple_from_histogram": auto.triple_state == 2, "trials": auto.triple_trials,
                          "gbps_auto": round(n / ma / 1e6, 1), "gbps_static": round(n / ms_ / 1e6, 1), "auto_over_static": round(ms_ / ma, 3),
                          "tiles3_auto": ca and ca["tiles3"], "tiles3_static": cs and cs["tiles3"],
                          "wg_auto": auto.last_launch()[0], "wg_static": static.last_launch()[0]}), flush=True)


if __name__ == "__main__":
    main()
