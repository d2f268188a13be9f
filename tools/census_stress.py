#!/usr/bin/env python3
"""The census's counters under repetition: the same (needle, pinned triple) asked again and again by fresh searchers on 1 GiB of text
(and on random bytes) must report the SAME candidate counts and per-position match counts every time - the kernel publishes them through
device-scope atomics ordered by waits, not fences (aux_kernels.hpp), and a lost or late counter would show here.  Hooks build.
    SLICESLICE_HIP_LIB=...libsliceslice_hip_tuning.so python tools/census_stress.py [rounds]"""
import json
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import sliceslice_rs_amd as ss  # noqa: E402


def main():
    rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 300
    assert ss.lib().has_hooks
    gd = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "data")
    raw = open(os.path.join(gd, "i386.txt"), "rb").read()
    text = torch.from_numpy(np.tile(np.frombuffer(raw, dtype=np.uint8), (1 << 30) // len(raw) + 1)[: 1 << 30].copy()).cuda()
    rnd = torch.empty(1 << 30, dtype=torch.uint8, device="cuda")
    ss.fill_random_device(rnd, 0x5EED0001)
    cases = [(b"segment descriptor table entries are", (11, 21, 14), text), (b"\nSame exceptions as in Reel Address Mode", (1, 5, 12), text),
             (b"        e ", (0, 9, 8), text), (b" the quick brown fox ", (0, 20, 20), text), (b"there is not another one of these", (17, 26, 19), rnd)]
    censuses = 0
    for needle, tri, hay in cases:
        first = None
        for r in range(rounds):
            s = ss.DynamicHipSearcher.new(needle)
            s.set_filter(*tri)
            s.search_in(hay)                                # names the pair
            s.search_in(hay)                                # the census, in front of this scan
            got = (s.census(hay), s.census_stats(hay))
            censuses += 1
            assert got[0] is not None and got[1] is not None, (needle, r)
            if first is None:
                first = got
            elif got != first:
                print(json.dumps({"MISMATCH": True, "needle": needle.decode("latin1"), "round": r, "first": first, "got": got}))
                sys.exit(1)
    print(json.dumps({"census_stress": "ok", "cases": len(cases), "rounds": rounds, "censuses": censuses}))


if __name__ == "__main__":
    main()
