#!/usr/bin/env python3
"""Randomised differential run of the many-problem kernels (ss_search_batched, ss_search_pairs, and ss_batch_plan_run for bool and
find plans: two runs each into garbage-filled outputs, so the self-resetting state is exercised) against Python's `in` / find:
ragged problems over small alphabets and random bytes, planted and near-miss needles, every kind of `position` (none, near,
16 or more behind needle[0]), haystacks from empty to a few MiB so that slices, the tile floor and surplus workgroups all
occur.    python tools/fuzz_batched.py SECONDS SEED      (SLICESLICE_BATCH_WGS=N varies the grid)"""
import json
import os
import random
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import sliceslice_rs_amd as ss  # noqa: E402


def one_round(rng, count):
    hays, needles, positions, want, where = [], [], [], [], []
    big_budget = 3
    for _ in range(count):
        alpha = rng.choice([b"ab", b"abc", b"\x00\x01", b"the quick brown fox ", bytes(range(256))])
        n = rng.choice([0, 1, 2, 3, 5, 8, 15, 16, 17, 18, 24, 31, 32, 33, 64, 100, 300, 1100, 2100])
        ln = rng.choice([0, 1, max(n - 1, 0), n, n + 1, n + 7, 2 * n + 100, 1500, 20000, 70000])
        if big_budget and rng.random() < 0.01:
            ln = rng.choice([(1 << 20) + 5, (3 << 20) - 7, 200000, (9 << 20) + 3])   # 9 MiB: a plan scans those round robin, four tiles per workgroup
            big_budget -= 1
        if ln > 4096:
            base = bytes(rng.choice(alpha) for _ in range(4096))
            hay = (base * (ln // 4096 + 1))[:ln]
        else:
            hay = bytes(rng.choice(alpha) for _ in range(ln))
        if n and ln >= n and rng.random() < 0.35:
            at = rng.choice([0, ln - n, rng.randrange(ln - n + 1)])
            nd = bytearray(hay[at:at + n])
            if rng.random() < 0.4:
                nd[rng.randrange(n)] ^= rng.choice([1, 0x80, 0x55])
            nd = bytes(nd)
        else:
            nd = bytes(rng.choice(alpha) for _ in range(n))
        hays.append(hay)
        needles.append(nd)
        positions.append(rng.choice([0, n - 1, n // 2, rng.randrange(n)]) if n else 0)
        want.append(nd in hay)
        where.append(hay.find(nd))
    hay_off = np.zeros(count + 1, dtype=np.int64)
    hay_off[1:] = np.cumsum([len(h) for h in hays])
    nd_off = np.zeros(count + 1, dtype=np.int64)
    nd_off[1:] = np.cumsum([len(x) for x in needles])
    blob = torch.from_numpy(np.frombuffer(b"".join(hays) + b"\0", dtype=np.uint8).copy()).cuda()
    nblob = torch.from_numpy(np.frombuffer(b"".join(needles) + b"\0", dtype=np.uint8).copy()).cuda()
    ho, no = torch.from_numpy(hay_off).cuda(), torch.from_numpy(nd_off).cuda()
    pos = torch.from_numpy(np.array(positions, dtype=np.int64)).cuda()
    for pairs in (False, True):
        for p in (None, pos):
            got = [bool(x) for x in ss.search_batched(blob, ho, nblob, no, position=p, pairs=pairs).cpu().tolist()]
            bad = [k for k in range(count) if got[k] != want[k]]
            if bad:
                k = bad[0]
                print(json.dumps({"MISMATCH": True, "pairs": pairs, "with_positions": p is not None, "problem": k, "needle_len": len(needles[k]),
                                  "haystack_len": len(hays[k]), "position": positions[k], "want": want[k], "needle": needles[k][:40].hex()}))
                sys.exit(1)
    # the unplanned call twice more on the SAME batch: the library samples the haystacks in front of the second call that names a batch
    # and later calls choose their filter bytes by the sampled classes (ss_batched.hip, batch_classes) - the answers must not move
    for rep in range(2):
        got = [bool(x) for x in ss.search_batched(blob, ho, nblob, no).cpu().tolist()]
        gotf = ss.find_batched(blob, ho, nblob, no).cpu().tolist()
        bad = [k for k in range(count) if got[k] != want[k] or gotf[k] != where[k]]
        if bad:
            k = bad[0]
            print(json.dumps({"MISMATCH": True, "repeated_call": rep, "problem": k, "needle_len": len(needles[k]), "haystack_len": len(hays[k]),
                              "want": where[k], "got": gotf[k], "got_bool": got[k], "needle": needles[k][:40].hex()}))
            sys.exit(1)
    # plans: the same problems, set up once, run twice; outputs start as garbage
    for find, with_pos in ((False, False), (True, False), (False, True)):
        # (plans with the caller's positions too: a first filter byte that is not needle[0], cold parts with bytes in front of it)
        plan = ss.BatchPlan(blob, ho, nblob, no, find=find, position=pos if with_pos else None)
        out = torch.full((count,), 0x5a5a5a5a, dtype=torch.int64 if find else torch.int32, device="cuda")
        for run in range(2):
            plan.run(out)
            got = out.cpu().tolist()
            exp = where if find else [1 if w else 0 for w in want]
            bad = [k for k in range(count) if got[k] != exp[k]]
            if bad:
                k = bad[0]
                print(json.dumps({"MISMATCH": True, "plan": True, "find": find, "with_positions": with_pos, "run": run, "problem": k, "needle_len": len(needles[k]),
                                  "haystack_len": len(hays[k]), "want": exp[k], "got": got[k], "needle": needles[k][:40].hex()}))
                sys.exit(1)
            out.fill_(-7)
        plan.close()
    return 14 * count


def main():
    seconds = float(sys.argv[1]) if len(sys.argv) > 1 else 30.0
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    rng = random.Random(seed)
    t_end = time.time() + seconds
    rounds = problems = 0
    while time.time() < t_end:
        problems += one_round(rng, rng.choice([1, 7, 300, 3000]))
        rounds += 1
    print(json.dumps({"fuzz_batched": "ok", "seconds": seconds, "seed": seed, "rounds": rounds, "problems_checked": problems,
                      "wgs": os.environ.get("SLICESLICE_BATCH_WGS", "auto")}))


if __name__ == "__main__":
    main()
