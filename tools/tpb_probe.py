#!/usr/bin/env python3
"""Tiles per workgroup against candidate density: a wave that meets a candidate ends later than its three neighbours, and the
workgroup's slot is held until it does.  With MORE tiles per workgroup a wave's delays add up over its own tiles while the other
waves keep streaming, so the slot is idle for a smaller share of the workgroup's life - if the longer-lived workgroups do not cost
the plain stream more than that (profiles/r01/tiles_per_block_sweep.jsonl).  Pinned triples on 1 GiB of the i386 text and on random
bytes, workgroups per CU four / five / six x tiles per workgroup 1 / 2 / 3 / 4, taking turns in one process.  Tuning build.
    SLICESLICE_HIP_LIB=...libsliceslice_hip_tuning.so python tools/tpb_probe.py"""
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import sliceslice_rs_amd as ss  # noqa: E402

WG = {"4": 40041, "5": 50041, "6": 60041}
TPB = (1, 2, 3, 4)


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
    cases = [("random16", bytes(nd), (0, 15, 13), rnd), ("privilege", b"privilege level zero!", None, text),
             ("seg 11-21-25", seg, (11, 21, 25), text), ("seg 21-35-28", seg, (21, 35, 28), text),
             ("seg 11-21-14", seg, (11, 21, 14), text), ("seg 21-35-22", seg, (21, 35, 22), text),
             ("quick 5-19-9", b" the quick brown fox ", (5, 19, 9), text), ("l regiseer 0-9-4", b"l regiseer", (0, 9, 4), text)]
    for name, needle, tri, hay in cases:
        searchers = {}
        for wg, variant in WG.items():
            for tpb in TPB:
                s = ss.DynamicHipSearcher.new(needle)
                if tri:
                    s.set_filter(*tri)
                s.set_variant(variant)
                s.set_grid(-tpb)
                s.set_timing(True)
                searchers["%sx%d" % (wg, tpb)] = s
        auto = ss.DynamicHipSearcher.new(needle)
        if tri:
            auto.set_filter(*tri)
        for _ in range(6):
            auto.search_in(hay)
        st = auto.tuning_state(hay)
        for s in searchers.values():
            for _ in range(3):
                s.search_in(hay)
        t_end = time.perf_counter() + 0.05
        while time.perf_counter() < t_end:
            auto.search_in(hay)
        got = {k: [] for k in searchers}
        for _ in range(4):
            for k, s in searchers.items():
                s.search_in(hay)
                for _ in range(6):
                    s.search_in(hay)
                    got[k].append(s.last_kernel_ms())
        ms = {k: float(np.median(v)) for k, v in got.items()}
        print(json.dumps({"case": name, "triple": tri, "tiles3": st["tiles3"], "lanes": st["lanes"], "deep_lanes": st["deep_lanes"],
                          "auto_wg": auto.last_launch()[0], "frac": {k: round(nbytes / v / 1e6 / 8000.0, 4) for k, v in ms.items()},
                          "best": min(ms, key=ms.get)}), flush=True)


if __name__ == "__main__":
    main()
