#!/usr/bin/env python3
"""Cross-device early exit of ss_search_sharded_all (the host relays the first device's match into the other devices' flags
through the BAR), on and off (SLICESLICE_CROSS_EXIT=0), with three 3 GiB shards on ONE GPU (a test set:
SLICESLICE_COMM_SET_NO_RCCL=1): ms per search_in with the needle in shard 0 / 1 / 2, and absent.    python tools/cross_exit_probe.py"""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import sliceslice_rs_amd as ss
os.environ["SLICESLICE_COMM_SET_NO_RCCL"] = "1"
G, each = 3, 3 << 30
needle = bytes(range(200, 216))
pn = torch.from_numpy(np.frombuffer(needle, dtype=np.uint8).copy()).cuda()
sh = []
for g in range(G):
    t = torch.empty(each, dtype=torch.uint8, device="cuda"); ss.fill_random_device(t, 0x5EED0100 + g); sh.append(t)
for where in (0, 1, 2):
    for g in range(G): sh[g][4096:4112] = 0
    sh[where][4096:4112] = pn
    torch.cuda.synchronize()
    row = {"needle_in_shard": where}
    for relay in ("1", "0"):
        os.environ["SLICESLICE_CROSS_EXIT"] = relay
        node = ss.NodeSearcher(needle, devices=[0] * G)
        for _ in range(5): assert node.search_in(sh) is True
        t0 = time.perf_counter()
        for _ in range(30): assert node.search_in(sh) is True
        row["relay" if relay == "1" else "no_relay"] = round((time.perf_counter() - t0) / 30 * 1e3, 4)
        node.close()
    print(row)
for g in range(G): sh[g][4096:4112] = 0
torch.cuda.synchronize()
node = ss.NodeSearcher(needle, devices=[0] * G)
t0 = time.perf_counter()
for _ in range(10): assert node.search_in(sh) is False
print({"absent_ms": round((time.perf_counter() - t0) / 10 * 1e3, 4)})
