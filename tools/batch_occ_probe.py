#!/usr/bin/env python3
"""Workgroups per CU for the BATCHED scan (SLICESLICE_BATCH_OCC, hooks build: read at every launch) on haystacks with and without
candidates: one plan per workload, its runs under 4 / 5 / 6 workgroups per CU taking turns in one process.
    SLICESLICE_HIP_LIB=...libsliceslice_hip_tuning.so python tools/batch_occ_probe.py [--mib 1024]"""
import argparse
import json
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import sliceslice_rs_amd as ss  # noqa: E402
from batch_probe import events_ms  # noqa: E402
from triple_probe import non_latin  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--mib", type=int, default=1024)
    ap.add_argument("--reps", type=int, default=30)
    args = ap.parse_args()
    assert ss.lib().has_hooks
    count, each = args.mib, 1 << 20
    n = count * each
    gd = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "data")
    raw = np.frombuffer(open(os.path.join(gd, "i386.txt"), "rb").read(), dtype=np.uint8)
    hoff = (torch.arange(count + 1, dtype=torch.int64) * each).cuda()
    work = []
    text = torch.from_numpy(np.tile(raw, n // raw.size + 1)[:n].copy()).cuda()
    for name, phrases in (("i386 text, stock phrases", [b"segment descriptor table entries are", b"protection exception handler must", b"tione as in Real", b"l regiseer"]),
                          ("i386 text, rare-byte phrases", [b"privilege level zero!", b" the quick brown fox ", b"there is not another one of these"])):
        nd = b"".join(phrases[i % len(phrases)] for i in range(count))
        lens = np.array([0] + [len(phrases[i % len(phrases)]) for i in range(count)], dtype=np.int64)
        work.append((name, text, torch.from_numpy(np.frombuffer(nd, dtype=np.uint8).copy()).cuda(), torch.from_numpy(np.cumsum(lens)).cuda()))
    blob = torch.empty(n, dtype=torch.uint8, device="cuda")
    ss.fill_random_device(blob, 0x5EED0001)
    ndr = bytearray(ss.fill_random_host(16 * count, 0x5EED0003).tobytes())
    ndr[8::16] = b"\xff" * count
    work.append(("random bytes", blob, torch.from_numpy(np.frombuffer(bytes(ndr), dtype=np.uint8).copy()).cuda(), (torch.arange(count + 1, dtype=torch.int64) * 16).cuda()))
    # the reference's i386 loop: 4,585 words, one text (aliased ranges), every word present - bound by how many short-lived workgroups
    # the chip keeps in flight, not by bytes
    words = [w for w in open(os.path.join(gd, "words.txt"), "rb").read().split(b"\n") if w]
    i386 = torch.from_numpy(raw.copy()).cuda()
    hb = torch.zeros(len(words), dtype=torch.int64, device="cuda")
    he = torch.full((len(words),), raw.size, dtype=torch.int64, device="cuda")
    wb = torch.from_numpy(np.frombuffer(b"".join(words), dtype=np.uint8).copy()).cuda()
    wo = torch.from_numpy(np.cumsum(np.array([0] + [len(w) for w in words], dtype=np.int64))).cuda()
    work.append(("the reference's i386 loop", i386, wb, wo, (hb, he), len(words), raw.size * len(words)))
    for item in work:
        name, hay, nb, noff = item[:4]
        ranges = item[4] if len(item) > 4 else None
        count = item[5] if len(item) > 4 else args.mib
        n = item[6] if len(item) > 4 else count * each
        if ranges is not None:
            plan = ss.BatchPlan(hay, None, nb, noff, hay_ranges=ranges)
            out = torch.empty(count, dtype=torch.int32, device="cuda")
            row = {"workload": name, "problems": count, "bytes": n}
            for rnd in range(2):
                for occ in (4, 5, 6):
                    os.environ["SLICESLICE_BATCH_OCC"] = str(occ)
                    row["occ%d_ms_%d" % (occ, rnd)] = round(events_ms(lambda: plan.run(out), args.reps)[0], 4)
                    row["call_occ%d_ms_%d" % (occ, rnd)] = round(events_ms(lambda: ss.search_batched(hay, None, nb, noff, hay_ranges=ranges), args.reps)[0], 4)
            del os.environ["SLICESLICE_BATCH_OCC"]
            row["found"] = int(out.sum().item())
            print(json.dumps(row), flush=True)
            plan.close()
            continue
        plan = ss.BatchPlan(hay, hoff, nb, noff)
        out = torch.empty(count, dtype=torch.int32, device="cuda")
        row = {"workload": name, "problems": count, "bytes": n}
        for rnd in range(2):
            for occ in (4, 5, 6):
                os.environ["SLICESLICE_BATCH_OCC"] = str(occ)
                row["occ%d_ms_%d" % (occ, rnd)] = round(events_ms(lambda: plan.run(out), args.reps)[0], 4)
        # the unplanned call's scan kernel needs 80 vector registers (six waves per SIMD fit), the plan's 81-82 (five)
        for _ in range(4):
            ss.search_batched(hay, hoff, nb, noff)
        for rnd in range(2):
            for occ in (4, 5, 6):
                os.environ["SLICESLICE_BATCH_OCC"] = str(occ)
                row["call_occ%d_ms_%d" % (occ, rnd)] = round(events_ms(lambda: ss.search_batched(hay, hoff, nb, noff), args.reps)[0], 4)
        del os.environ["SLICESLICE_BATCH_OCC"]
        for occ in (4, 5, 6):
            row["occ%d_gbps" % occ] = round(n / min(row["occ%d_ms_0" % occ], row["occ%d_ms_1" % occ]) / 1e6, 1)
            row["call_occ%d_gbps" % occ] = round(n / min(row["call_occ%d_ms_0" % occ], row["call_occ%d_ms_1" % occ]) / 1e6, 1)
        row["found"] = int(out.sum().item())
        print(json.dumps(row), flush=True)
        plan.close()


if __name__ == "__main__":
    main()
