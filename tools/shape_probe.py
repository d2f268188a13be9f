#!/usr/bin/env python3
"""Launch shape against candidate density on MANY phrases: the settled cases of a tools/survival_probe.py run (needle + the bytes
in force, pinned here) under workgroups per CU x tiles per workgroup, taking turns in one process on 1 GiB of the i386 text.
Decides where two tiles per workgroup pay (tools/tpb_probe.py found +2 % around 44 candidate tiles of 1,024).  Tuning build.
    SLICESLICE_HIP_LIB=...libsliceslice_hip_tuning.so python tools/shape_probe.py profiles/r06/survival_probe_final.jsonl [--min 9 --max 260]"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import sliceslice_rs_amd as ss  # noqa: E402

DEFAULT_SHAPES = "4x1,4x2,5x1,5x2,6x1"


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("rows")
    ap.add_argument("--min", type=int, default=9)
    ap.add_argument("--max", type=int, default=260)
    ap.add_argument("--gib", type=float, default=1.0)
    ap.add_argument("--shapes", default=DEFAULT_SHAPES)
    args = ap.parse_args()
    assert ss.lib().has_hooks
    nbytes = int(args.gib * (1 << 30))
    SHAPES = {k: (int(k.split("x")[0]) * 10000 + 41, int(k.split("x")[1])) for k in args.shapes.split(",")}
    gd = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "data")
    raw = open(os.path.join(gd, "i386.txt"), "rb").read()
    text = torch.from_numpy(np.tile(np.frombuffer(raw, dtype=np.uint8), nbytes // len(raw) + 1)[:nbytes].copy()).cuda()
    seen = set()
    for line in open(args.rows):
        r = json.loads(line)
        if r["kind"] != "text":
            continue
        tri = tuple(r["state"]["in_force"])
        key = (r["needle"], tri)
        if key in seen or not (args.min <= r["state"]["tiles3"] <= args.max) or tri[1] - tri[0] > 15 or tri[2] == tri[1]:
            continue
        seen.add(key)
        needle = r["needle"].encode("latin1")
        searchers = {}
        for name, (variant, tpb) in SHAPES.items():
            s = ss.DynamicHipSearcher.new(needle)
            s.set_filter(*tri)
            s.set_variant(variant)
            s.set_grid(-tpb)
            s.set_timing(True)
            searchers[name] = s
        auto = ss.DynamicHipSearcher.new(needle)
        auto.set_filter(*tri)
        for _ in range(6):
            auto.search_in(text)
        st = auto.tuning_state(text)
        for s in searchers.values():
            for _ in range(3):
                s.search_in(text)
        t_end = time.perf_counter() + 0.03
        while time.perf_counter() < t_end:
            auto.search_in(text)
        got = {k: [] for k in searchers}
        for _ in range(4):
            for k, s in searchers.items():
                s.search_in(text)
                for _ in range(6):
                    s.search_in(text)
                    got[k].append(s.last_kernel_ms())
        ms = {k: float(np.median(v)) for k, v in got.items()}
        print(json.dumps({"needle": r["needle"], "triple": tri, "tiles3": st["tiles3"], "lanes": st["lanes"], "deep_lanes": st["deep_lanes"],
                          "gib": args.gib, "auto_wg": auto.last_launch()[0], "frac": {k: round(nbytes / v / 1e6 / 8000.0, 4) for k, v in ms.items()},
                          "best": min(ms, key=ms.get)}), flush=True)


if __name__ == "__main__":
    main()
