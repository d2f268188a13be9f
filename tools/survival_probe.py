#!/usr/bin/env python3
"""What the census's MEASURED survival counts buy: for many absent phrases on 1 GiB of text (the i386 manual
tiled; the phrase set of tools/occ_census.py, same seed) and a few needles on random bytes, the kernel time of a searcher with
launch tuning ON (third byte / near bytes by the census's per-position match counts, second-level schedule in measured order,
workgroups per CU by the candidate count) next to the SAME searcher with it OFF (ss_set_autotune(0): static triple, static
schedule, needle-byte guess) - the two taking turns in one process on one buffer, hipEvents on the launch stream - and what the
handle reports about the haystack afterwards (ss_searcher_tuning_state).

    python tools/survival_probe.py [--phrases 48] [--seed 1] [--modes new,refpair,wp] [--kinds text,random]
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

PEAK = 8000.0


def make(ph, mode):
    if mode == "wp":
        return ss.DynamicHipSearcher.with_position(ph, len(ph) - 1)
    s = ss.DynamicHipSearcher.new(ph)
    if mode == "refpair":
        s.set_filter(0, len(ph) - 1)
    return s


def turns(on, off, pin, hay, rounds=4, reps=6):
    got = {"on": [], "off": [], "pin": []}
    for s in (on, off, pin):
        s.set_timing(True)

    def call(s, tuned):
        ss.set_autotune(tuned)
        r = s.search_in(hay)
        ss.set_autotune(True)
        return r
    res = None
    for _ in range(16):                      # the tuned searchers settle: census, trials, order
        res = call(on, True)
        call(pin, True)
    call(off, False)
    t_end = time.perf_counter() + 0.05
    while time.perf_counter() < t_end:
        call(on, True)
    for _ in range(rounds):
        for name, s, tuned in (("on", on, True), ("off", off, False), ("pin", pin, True)):
            call(s, tuned)
            for _ in range(reps):
                call(s, tuned)
                got[name].append(s.last_kernel_ms())
    return res, float(np.median(got["on"])), float(np.median(got["off"])), float(np.median(got["pin"]))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--phrases", type=int, default=48)
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--modes", default="new,refpair,wp")
    ap.add_argument("--kinds", default="text,random")
    args = ap.parse_args()
    rng = random.Random(args.seed)
    nbytes = 1 << 30
    gd = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "data")
    raw = open(os.path.join(gd, "i386.txt"), "rb").read()
    text = torch.from_numpy(np.tile(np.frombuffer(raw, dtype=np.uint8), nbytes // len(raw) + 1)[:nbytes].copy()).cuda()
    rnd = torch.empty(nbytes, dtype=torch.uint8, device="cuda")
    ss.fill_random_device(rnd, 0x5EED0001)

    def absent_variant(ph):
        for _ in range(50):
            k = rng.randrange(len(ph))
            for c in b"etaoin srhl":
                cand = ph[:k] + bytes([c]) + ph[k + 1:]
                if cand != ph and cand not in raw:
                    return cand
        return None

    cases = [("text", ph) for ph in (b"segment descriptor table entries are", b" the quick brown fox ", b"protection exception handler must",
                                     b"privilege level zero!", b"there is not another one of these")]
    while len(cases) < args.phrases:
        n = rng.choice((4, 6, 8, 10, 12, 16, 16, 20, 24, 32, 40))
        at = rng.randrange(len(raw) - n)
        ph = absent_variant(raw[at:at + n])
        if ph:
            cases.append(("text", ph))
    nd = bytearray(ss.fill_random_host(16, 0x5EED0002).tobytes())
    nd[8] = 0xFF
    cases += [("random", bytes(nd)), ("random", b"privilege level zero!"), ("random", b"there is not another one of these")]
    modes = args.modes.split(",")
    cases = [c for c in cases if c[0] in args.kinds.split(",")]
    for kind, ph in cases:
        hay = text if kind == "text" else rnd
        for mode in modes:
            if mode != "new" and len(ph) < 3:
                continue
            on, off, pin = make(ph, mode), make(ph, mode), make(ph, mode)
            a, b, c = pin.device_triple()
            if b - a <= 15 and c != b:
                pin.set_filter(a, b, c)      # the same bytes, named by the caller: nothing moves; census order and workgroups per CU only
            found, m_on, m_off, m_pin = turns(on, off, pin, hay)
            st, sp = on.tuning_state(hay), pin.tuning_state(hay)
            print(json.dumps({"kind": kind, "needle": ph.decode("latin1"), "n": len(ph), "mode": mode, "found": found, "filter": list(on.filter3),
                              "ms_on": round(m_on, 4), "ms_off": round(m_off, 4), "ms_pin": round(m_pin, 4), "pin": {k: sp[k] for k in ("tiles3", "lanes", "in_force", "order_measured")},
                              "wg_pin": pin.last_launch()[0], "frac_on": round(nbytes / m_on / 1e6 / PEAK, 4),
                              "frac_off": round(nbytes / m_off / 1e6 / PEAK, 4), "on_over_off": round(m_off / m_on, 4),
                              "wg_on": on.last_launch()[0], "wg_off": off.last_launch()[0],
                              "state": {k: st[k] for k in ("census_state", "tiles3", "tiles2", "lanes", "pair_lanes", "triple_lanes", "triple_state",
                                                           "trials", "accepted", "settled", "proposal", "own", "in_force", "order_measured", "order", "kernel_mode")}}),
                  flush=True)


if __name__ == "__main__":
    main()
