#!/usr/bin/env python3
"""ss_search_batched over `count` problems of `each` bytes (random bytes, absent 16-byte needles, the `new` position):
kernel + flag memset by events on the launch stream, median of 15.  SLICESLICE_BATCH_WGS=N overrides the total number of
workgroups (one process per setting).    python tools/batch_shape_probe.py COUNT EACH_KIB [COUNT EACH_KIB ...]"""
import json
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import sliceslice_rs_amd as ss  # noqa: E402


def main():
    args = [int(a) for a in sys.argv[1:]] or [4096, 1024]
    out = {"wgs": os.environ.get("SLICESLICE_BATCH_WGS", "auto")}
    for count, kib in zip(args[0::2], args[1::2]):
        each = kib << 10
        hay = torch.empty(count * each, dtype=torch.uint8, device="cuda")
        ss.fill_random_device(hay, 0x5EED0001)
        nb = bytearray(ss.fill_random_host(16 * count, 0x5EED0003).tobytes())
        for i in range(count):
            nb[16 * i + 8] = 0xFF
        nblob = torch.from_numpy(np.frombuffer(bytes(nb), dtype=np.uint8).copy()).cuda()
        hay_off = (torch.arange(count + 1, dtype=torch.int64) * each).cuda()
        nd_off = (torch.arange(count + 1, dtype=torch.int64) * 16).cuda()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        ms = []
        for _ in range(20):
            e0.record()
            found = ss.search_batched(hay, hay_off, nblob, nd_off)
            e1.record()
            e1.synchronize()
            ms.append(e0.elapsed_time(e1))
        assert int(found.sum().item()) == 0
        m = float(np.median(ms[5:]))
        out[f"{count}x{kib}KiB"] = {"ms": round(m, 4), "gbps": round(count * each / m / 1e6, 1)}
        del hay
    print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
