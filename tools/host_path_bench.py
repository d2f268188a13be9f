#!/usr/bin/env python3
"""PCIe-inclusive rate of the host-buffer entry points (ss_search_host, ss_search_file): never the
headline `value` - noted in DESIGN.md.  One GPU."""
import json
import os
import sys
import tempfile
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import sliceslice_rs_amd as ss  # noqa: E402
from settle import wait_for_vram_reclaim  # noqa: E402


def main():
    wait_for_vram_reclaim()
    gib = float(sys.argv[1]) if len(sys.argv) > 1 else 4.0
    n_bytes = int(gib * (1 << 30))
    host = ss.fill_random_host(n_bytes, 0x5EED0001)
    nd = bytearray(ss.fill_random_host(16, 0x5EED0002).tobytes())
    nd[8] = 0xFF
    s = ss.DynamicHipSearcher.new(bytes(nd))
    assert s.search_in(host[: 1 << 20]) is False
    for label, buf in (("pageable numpy buffer", host),):
        best = float("inf")
        for _ in range(3):
            t = time.perf_counter()
            r = s.search_in(buf)
            best = min(best, time.perf_counter() - t)
            assert r is False
        print(json.dumps({"entry": "ss_search_host", "source": label, "bytes": n_bytes, "s": round(best, 4),
                          "gbps": round(n_bytes / best / 1e9, 2)}), flush=True)
    if hasattr(ss, "search_file"):
        d = "/dev/shm" if os.path.isdir("/dev/shm") else tempfile.gettempdir()
        path = os.path.join(d, "ss_host_path_bench.bin")
        host.tofile(path)
        for threads in [int(x) for x in os.environ.get("SS_FILE_THREADS_SWEEP", "0").split(",")]:
            if threads:
                os.environ["SLICESLICE_FILE_THREADS"] = str(threads)
            best = float("inf")
            for _ in range(3):
                t = time.perf_counter()
                r = ss.search_file(s, path)
                best = min(best, time.perf_counter() - t)
                assert r is False
            print(json.dumps({"entry": "ss_search_file", "source": "file in " + d, "bytes": n_bytes,
                              "read_threads": threads or "default", "s": round(best, 4),
                              "gbps": round(n_bytes / best / 1e9, 2)}), flush=True)
        os.unlink(path)


if __name__ == "__main__":
    main()
