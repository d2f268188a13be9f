#!/usr/bin/env python3
"""Workgroup shape against candidate density (round 6): a candidate holds its WORKGROUP'S slot while one wave works through the
second level - four waves' worth of HBM requests are not issued for as long as that takes.  Pinned filter triples of a few phrases on
1 GiB of the i386 text (candidate tiles from 0 to 250 of 1,024, cheap and deep ones) under forced launch shapes: 256-thread workgroups
at four / five / six per CU, 128-thread workgroups at eight / nine per CU.  One process, one buffer, the shapes taking turns.  Needs the
tuning build (SLICESLICE_HIP_LIB=...libsliceslice_hip_tuning.so)."""
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import sliceslice_rs_amd as ss  # noqa: E402

SHAPES = {"256x4": 40041, "256x5": 50041, "256x6": 60041, "128x8": 180041, "128x9": 190041}


def main():
    assert ss.lib().has_hooks
    nbytes = 1 << 30
    gd = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "data")
    raw = open(os.path.join(gd, "i386.txt"), "rb").read()
    text = torch.from_numpy(np.tile(np.frombuffer(raw, dtype=np.uint8), nbytes // len(raw) + 1)[:nbytes].copy()).cuda()
    rnd = torch.empty(nbytes, dtype=torch.uint8, device="cuda")
    ss.fill_random_device(rnd, 0x5EED0001)
    nd = bytearray(ss.fill_random_host(16, 0x5EED0002).tobytes())
    nd[8] = 0xFF
    seg = b"segment descriptor table entries are"
    cases = [("random16", bytes(nd), (0, 15, 13), rnd), ("seg 11-21-14", seg, (11, 21, 14), text), ("seg 11-21-25", seg, (11, 21, 25), text),
             ("seg 21-35-22", seg, (21, 35, 22), text), ("seg 21-35-28", seg, (21, 35, 28), text),
             ("l regiseer 0-9-4", b"l regiseer", (0, 9, 4), text), ("l regiseer 0-9-7", b"l regiseer", (0, 9, 7), text),
             ("there is not 17-26-19", b"there is not another one of these", (17, 26, 19), text),
             ("eginneng 0-15-10", b"eginneng of\na su", (0, 15, 10), text), ("quick 5-19-9", b" the quick brown fox ", (5, 19, 9), text)]
    for name, needle, tri, hay in cases:
        searchers = {}
        for shape, variant in SHAPES.items():
            s = ss.DynamicHipSearcher.new(needle)
            s.set_filter(*tri)
            s.set_variant(variant)
            s.set_timing(True)
            searchers[shape] = s
        auto = ss.DynamicHipSearcher.new(needle)
        auto.set_filter(*tri)
        auto.set_timing(True)
        for _ in range(4):
            auto.search_in(hay)
        st = auto.tuning_state(hay)
        t_end = time.perf_counter() + 0.05
        while time.perf_counter() < t_end:
            auto.search_in(hay)
        got = {k: [] for k in searchers}
        for _ in range(4):
            for shape, s in searchers.items():
                s.search_in(hay)
                for _ in range(6):
                    s.search_in(hay)
                    got[shape].append(s.last_kernel_ms())
        ms = {k: round(float(np.median(v)), 4) for k, v in got.items()}
        print(json.dumps({"case": name, "triple": tri, "tiles3": st["tiles3"], "lanes": st["lanes"], "deep_lanes": st["deep_lanes"],
                          "auto_wg": auto.last_launch()[0], "ms": ms, "frac": {k: round(nbytes / v / 1e6 / 8000.0, 4) for k, v in ms.items()},
                          "best": min(ms, key=ms.get)}), flush=True)


if __name__ == "__main__":
    main()
