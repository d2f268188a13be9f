#!/usr/bin/env python3
"""The headline workload (random bytes, 16-byte absent needle) under different first-phase triples, taking turns in ONE process on
ONE buffer: what ss_searcher_new settles on with launch tuning on, the searcher's own static triple pinned, and other pinned
triples of different SPAN (how far the two further bytes lie behind the first: the next lane's dwords the first phase has to fetch)
- next to the plain-read ceiling of the same buffer.  What the compact form (ss_census.hip, propose_compact) stands on.
    python tools/headline_triple_probe.py [--gib 64]"""
import argparse
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import sliceslice_rs_amd as ss  # noqa: E402
from occ_probe import paired_ms  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gib", type=float, default=64.0)
    ap.add_argument("--rounds", type=int, default=4)
    args = ap.parse_args()
    n = int(args.gib * (1 << 30))
    hay = torch.empty(n, dtype=torch.uint8, device="cuda")
    ss.fill_random_device(hay, 0x5EED0001)
    nd = bytearray(ss.fill_random_host(16, 0x5EED0002).tobytes())
    nd[8] = 0xFF
    nd = bytes(nd)
    print(json.dumps({"gib": args.gib, "read_ceiling_gbps_before": round(ss.read_ceiling_gbps(hay), 1)}), flush=True)
    auto = ss.DynamicHipSearcher.new(nd)
    for _ in range(16):                                         # the handle settles
        auto.search_in(hay)
    names, searchers = ["auto"], [auto]
    for tri in ((0, 15, 13), (1, 8, 15), (0, 1, 2), (0, 3, 2), (0, 4, 5), (0, 7, 6), (0, 4, 8), (0, 8, 9), (0, 11, 10), (0, 8, 12), (0, 12, 13), (0, 15, 12), (8, 9, 10)):
        s = ss.DynamicHipSearcher.new(nd)
        s.set_filter(*tri)
        names.append("pinned %d,%d,%d" % tri)
        searchers.append(s)
    for rep in range(2):
        res, ms = paired_ms(searchers, hay, rounds=args.rounds)
        row = {"found": res, "rep": rep}
        for k, m in zip(names, ms):
            row[k] = {"ms": round(m, 4), "gbps": round(n / m / 1e6, 1), "frac": round(n / m / 1e6 / 8000, 4)}
        row["auto_state"] = {k: v for k, v in auto.tuning_state(hay).items() if k in ("own", "in_force", "trials", "accepted", "tiles3", "tiles2", "workgroups_per_cu")} \
            if isinstance(auto.tuning_state(hay), dict) else str(auto.tuning_state(hay))
        row["launch"] = {k: list(s.last_launch()) for k, s in zip(names, searchers)}
        print(json.dumps(row), flush=True)
    print(json.dumps({"read_ceiling_gbps_after": round(ss.read_ceiling_gbps(hay), 1)}), flush=True)


if __name__ == "__main__":
    main()
