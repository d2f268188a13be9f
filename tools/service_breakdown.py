#!/usr/bin/env python3
"""Where a request to the resident search service spends its time: run against the hooks build with
SLICESLICE_SERVICE_DEBUG=1, whose library prints - every 16,384 requests - the average microseconds of the mailbox write (CPU
stores through the PCIe BAR + the HDP flush) and of `post + wait` (mailbox write -> answer word seen).  This script drives
requests of increasing content so that the differences isolate the steps:

    1 KiB absent, bound      one workgroup, no acquire, one tile, no count-out  -> round trip + poll latency + one tile
    1 KiB absent, unbound    + the acquire (cache invalidate) per request
    64 KiB absent, bound     four workgroups: + the count-out atomics and the last workgroup's answer
    857 KB text, bound       the reference's bench shape (bench/benches/i386.rs:246-256): words of the text, early exit
    SLICESLICE_SERVICE_HDP_FLUSH=0 (second run)   the same without the flush-register write per request

    SLICESLICE_HIP_LIB=<libsliceslice_hip_tuning.so> SLICESLICE_SERVICE_DEBUG=1 python tools/service_breakdown.py
Prints one JSON line per case with the wall time per request as Python sees it (ctypes call overhead included: ~1 us); the
library's own lines go to stderr."""
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import sliceslice_rs_amd as ss  # noqa: E402

N = 32768   # requests per case: two of the library's reporting periods


def main():
    assert ss.lib().has_hooks, "run with SLICESLICE_HIP_LIB=<libsliceslice_hip_tuning.so>"
    gd = os.path.join(os.path.dirname(__file__), "..", "tests", "golden", "data")
    raw = open(os.path.join(gd, "i386.txt"), "rb").read()
    words = [w for w in open(os.path.join(gd, "words.txt"), "rb").read().split(b"\n") if w]
    text = torch.from_numpy(np.frombuffer(raw, dtype=np.uint8).copy()).cuda()
    rnd = torch.empty(1 << 20, dtype=torch.uint8, device="cuda")
    ss.fill_random_device(rnd, 0x5EED0001)
    torch.cuda.synchronize()
    absent = ss.DynamicHipSearcher.new(bytes([255] * 16))
    wsearchers = [ss.DynamicHipSearcher.new(w) for w in words]
    with ss.SearchService(lease_ms=50.0) as sv:
        def run(name, hay, searchers, bound):
            if bound:
                sv.bind(hay)
            else:
                sv.unbind()
            for s in searchers[:64]:
                sv.search_in(s, hay)
            sys.stderr.write("---- %s ----\n" % name)
            t0 = time.perf_counter()
            k = 0
            for i in range(N):
                sv.search_in(searchers[k], hay)
                k = k + 1 if k + 1 < len(searchers) else 0
            us = (time.perf_counter() - t0) / N * 1e6
            print(json.dumps({"case": name, "requests": N, "wall_us_per_request_python": round(us, 3), "counters": sv.counters()}), flush=True)
        run("1KiB_absent_bound", rnd[:1024], [absent], True)
        run("1KiB_absent_unbound", rnd[:1024], [absent], False)
        run("64KiB_absent_bound", rnd[:65536], [absent], True)
        run("64KiB_absent_unbound", rnd[:65536], [absent], False)
        run("1MiB_absent_bound", rnd, [absent], True)
        run("i386_words_bound", text, wsearchers, True)
        run("i386_words_unbound", text, wsearchers, False)


if __name__ == "__main__":
    main()
