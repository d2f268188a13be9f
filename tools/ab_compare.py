#!/usr/bin/env python3
"""Same-box A/B of library builds: for every library given (name=path), a fresh process measures the kernel
GB/s of the listed cases; the whole sequence is repeated --rounds times, interleaved, so that box-to-box and
minute-to-minute drift cancels.  JSON lines + a median table on stdout.
    python tools/ab_compare.py --libs cur=...so two=...so --gib 8 --rounds 3"""
import argparse
import json
import os
import statistics
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
WORKER = r'''
import json, os, sys, time
import numpy as np, torch
sys.path.insert(0, %r)
import sliceslice_rs_amd as ss
from tools_settle import wait_for_vram_reclaim
wait_for_vram_reclaim()
gib = float(sys.argv[1]); cases = sys.argv[2].split(",")
n_bytes = int(gib * (1 << 30))
hay = torch.empty(n_bytes, dtype=torch.uint8, device="cuda"); ss.fill_random_device(hay, 0x5EED0001); torch.cuda.synchronize()
def absent(n):
    nd = bytearray(ss.fill_random_host(n, 0x5EED0002).tobytes()); nd[0 if n == 1 else (1 if n == 2 else n // 2)] = 0xFF; return bytes(nd)
text = None
out = {}
for c in cases:
    if c.startswith("n"):
        s = ss.DynamicHipSearcher.new(absent(int(c[1:]))); h = hay
    else:
        if text is None:
            raw = np.frombuffer(open(os.path.join(%r, "tests", "golden", "data", "i386.txt"), "rb").read(), dtype=np.uint8)
            text = torch.from_numpy(raw.copy()).cuda().repeat((1 << 30) // raw.size)
        nd = {"tworst": b"segment descriptor table entries are", "tspaces": b" the quick brown fox ", "tpriv": b"privilege level zero!"}[c]
        s = ss.DynamicHipSearcher.new(nd); h = text
    s.set_timing(True)
    t_end = time.perf_counter() + 0.2
    while time.perf_counter() < t_end:
        s.search_in(h)
    ms = []
    for _ in range(12):
        s.search_in(h); ms.append(s.last_kernel_ms())
    out[c] = round(h.numel() / float(np.median(ms)) / 1e6, 1)
print(json.dumps(out))
'''


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--libs", nargs="+", required=True)
    ap.add_argument("--gib", type=float, default=8.0)
    ap.add_argument("--rounds", type=int, default=3)
    ap.add_argument("--cases", default="n16,n1,n2000,tworst,tspaces")
    args = ap.parse_args()
    libs = [l.split("=", 1) for l in args.libs]
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    worker = WORKER % (ROOT, ROOT)
    worker = worker.replace("from tools_settle import", "sys.path.insert(0, %r)\nfrom settle import" % os.path.join(ROOT, "tools"))
    acc = {}
    for r in range(args.rounds):
        for name, path in libs:
            env = dict(os.environ, SLICESLICE_HIP_LIB=path)
            p = subprocess.run([sys.executable, "-c", worker, str(args.gib), args.cases], capture_output=True, text=True, env=env, timeout=900)
            line = [l for l in p.stdout.splitlines() if l.startswith("{")]
            if p.returncode != 0 or not line:
                print(json.dumps({"lib": name, "round": r, "error": p.stderr[-500:]}), flush=True)
                continue
            d = json.loads(line[-1])
            print(json.dumps({"lib": name, "round": r, **d}), flush=True)
            for k, v in d.items():
                acc.setdefault((name, k), []).append(v)
    table = {}
    for (name, k), v in acc.items():
        table.setdefault(name, {})[k] = round(statistics.median(v), 1)
    print(json.dumps({"median_gbps": table}), flush=True)


if __name__ == "__main__":
    main()
