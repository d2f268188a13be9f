#!/usr/bin/env python3
"""Filter bytes of a PLAN chosen by the haystacks' own sampled histogram (ss_batch_plan_create: batch_sample_kernel + 16 classes)
against the static, corpus-free classes (SLICESLICE_BATCH_STATIC_CLASSES=1 at plan creation, hooks build): per workload one plan
of each kind on the same problems, their runs taking turns in one process (events around single runs, median of `--reps`).
Workloads: non-Latin UTF-8-like text cut into 1 MiB haystacks with absent words of its own alphabet; the i386 manual tiled, stock
phrases; the reference's i386 loop (4,585 words, one text: aliased ranges, every word present); random bytes (config 5's shape).
    SLICESLICE_HIP_LIB=...libsliceslice_hip_tuning.so python tools/batch_triple_probe.py [--mib 1024] [--reps 30]"""
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


def plans(*a, **kw):
    auto = ss.BatchPlan(*a, **kw)
    os.environ["SLICESLICE_BATCH_STATIC_CLASSES"] = "1"
    try:
        static = ss.BatchPlan(*a, **kw)
    finally:
        del os.environ["SLICESLICE_BATCH_STATIC_CLASSES"]
    return auto, static


def measure(name, nbytes, auto, static, reps, want=None, call=None):
    out = torch.empty(auto.count, dtype=torch.int32, device="cuda")
    row = {"workload": name, "problems": auto.count, "bytes": nbytes}
    if call is not None:
        # the unplanned call: the library samples in front of the second call that names the batch; the static classes on request
        for _ in range(4):
            ref = call()
        torch.cuda.synchronize()
        for rnd in range(2):
            row["call_hist_ms_%d" % rnd] = round(events_ms(call, reps)[0], 4)
            os.environ["SLICESLICE_BATCH_STATIC_CLASSES"] = "1"
            try:
                row["call_static_ms_%d" % rnd] = round(events_ms(call, reps)[0], 4)
                assert torch.equal(call(), ref), name
            finally:
                del os.environ["SLICESLICE_BATCH_STATIC_CLASSES"]
        row["call_hist_over_static"] = round(min(row["call_static_ms_0"], row["call_static_ms_1"]) / min(row["call_hist_ms_0"], row["call_hist_ms_1"]), 3)
    answers = {}
    for rnd in range(2):
        for kind, p in (("hist", auto), ("static", static)):
            ms, mn = events_ms(lambda: p.run(out), reps)
            row["%s_ms_%d" % (kind, rnd)] = round(ms, 4)
            torch.cuda.synchronize()
            answers[kind] = out.clone()
            if want is not None:
                assert int(out.sum().item()) == want, (name, kind, int(out.sum().item()), want)
    assert torch.equal(answers["hist"], answers["static"]), name          # whichever table chose the bytes: the same answers
    row["found"] = int(answers["hist"].sum().item())
    for kind in ("hist", "static"):
        row[kind + "_gbps"] = round(nbytes / min(row[kind + "_ms_0"], row[kind + "_ms_1"]) / 1e6, 1)
    row["hist_over_static"] = round(row["hist_gbps"] / row["static_gbps"], 3)
    changed = sum(auto.filter_of(i)[0] != static.filter_of(i)[0] for i in range(0, auto.count, max(1, auto.count // 256)))
    row["descriptors_changed_of_sampled"] = "%d of %d" % (changed, len(range(0, auto.count, max(1, auto.count // 256))))
    print(json.dumps(row), flush=True)
    auto.close()
    static.close()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--mib", type=int, default=1024)
    ap.add_argument("--reps", type=int, default=30)
    args = ap.parse_args()
    assert ss.lib().has_hooks, "hooks build needed (SLICESLICE_HIP_LIB=...libsliceslice_hip_tuning.so)"
    count, each = args.mib, 1 << 20
    n = count * each
    gd = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "data")
    rng = np.random.default_rng(3)
    hoff = (torch.arange(count + 1, dtype=torch.int64) * each).cuda()

    # 1. non-Latin text, absent words of its own alphabet (one trail byte replaced by the generator's rarest)
    host = non_latin(n, 7)
    for nlen in (8, 16, 32):
        nd = bytearray()
        for i in range(count):
            at = 2 * int(rng.integers(0, each // 2 - 64)) + i * each
            w = bytearray(host[at:at + nlen].tobytes())
            w[nlen // 2 | 1] = w[1] = 0xBF
            nd += w
        nb = torch.from_numpy(np.frombuffer(bytes(nd), dtype=np.uint8).copy()).cuda()
        noff = (torch.arange(count + 1, dtype=torch.int64) * nlen).cuda()
        hay = torch.from_numpy(host).cuda()
        measure("non-latin text, %d x 1 MiB, (all but) absent %d-byte words" % (count, nlen), n, *plans(hay, hoff, nb, noff), args.reps,
                call=lambda: ss.search_batched(hay, hoff, nb, noff))
        del hay

    # 2. the i386 manual tiled, absent phrases in its own vocabulary
    raw = np.frombuffer(open(os.path.join(gd, "i386.txt"), "rb").read(), dtype=np.uint8)
    text = np.tile(raw, n // raw.size + 1)[:n].copy()
    phrases = [b"segment descriptor table entries are", b"privilege level zero!", b"there is not another one of these", b"the quick brown fox", b"instruction prefetch queues"]
    nd = b"".join(phrases[i % len(phrases)] for i in range(count))
    lens = np.array([0] + [len(phrases[i % len(phrases)]) for i in range(count)], dtype=np.int64)
    nb = torch.from_numpy(np.frombuffer(nd, dtype=np.uint8).copy()).cuda()
    noff = torch.from_numpy(np.cumsum(lens)).cuda()
    hay = torch.from_numpy(text).cuda()
    measure("i386 text tiled, %d x 1 MiB, absent phrases" % count, n, *plans(hay, hoff, nb, noff), args.reps, call=lambda: ss.search_batched(hay, hoff, nb, noff))
    del hay

    # 3. the reference's i386 loop: 4,585 words, one text (aliased ranges), every word present
    words = [w for w in open(os.path.join(gd, "words.txt"), "rb").read().split(b"\n") if w]
    i386 = torch.from_numpy(raw.copy()).cuda()
    hb = torch.zeros(len(words), dtype=torch.int64, device="cuda")
    he = torch.full((len(words),), raw.size, dtype=torch.int64, device="cuda")
    wb = torch.from_numpy(np.frombuffer(b"".join(words), dtype=np.uint8).copy()).cuda()
    wo = torch.from_numpy(np.cumsum(np.array([0] + [len(w) for w in words], dtype=np.int64))).cuda()
    measure("the reference's i386 loop: %d words, one text" % len(words), raw.size * len(words),
            *plans(i386, None, wb, wo, hay_ranges=(hb, he)), args.reps, want=len(words),
            call=lambda: ss.search_batched(i386, None, wb, wo, hay_ranges=(hb, he)))

    # 4. random bytes, config 5's shape: absent 16-byte needles
    blob = torch.empty(n, dtype=torch.uint8, device="cuda")
    ss.fill_random_device(blob, 0x5EED0001)
    ndr = bytearray(ss.fill_random_host(16 * count, 0x5EED0003).tobytes())
    ndr[8::16] = b"\xff" * count
    nb = torch.from_numpy(np.frombuffer(bytes(ndr), dtype=np.uint8).copy()).cuda()
    noff = (torch.arange(count + 1, dtype=torch.int64) * 16).cuda()
    measure("random bytes, %d x 1 MiB, absent 16-byte needles" % count, n, *plans(blob, hoff, nb, noff), args.reps, want=0,
            call=lambda: ss.search_batched(blob, hoff, nb, noff))


if __name__ == "__main__":
    main()
