#!/usr/bin/env python3
"""One cut of a random blob into `count` haystacks of `each` bytes, `count` absent 16-byte needles: ss_search_batched against
ss_batch_plan_run on the same problems, alternating in one process (events on the launch stream around single calls).  Under
`rocprofv3 --kernel-trace --stats` the two show up as scan_batched_plan_kernel<4, false, false> (+ batch_plan_kernel) and
<4, false, true>.      python tools/batch_probe.py 1024x1048576 [256x4194304 ...] [--find] [--reps 30]"""
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import sliceslice_rs_amd as ss  # noqa: E402


def events_ms(fn, reps):
    t_end = time.perf_counter() + 0.05
    while time.perf_counter() < t_end:
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ms = []
    for _ in range(reps):
        e0.record()
        fn()
        e1.record()
        e1.synchronize()
        ms.append(e0.elapsed_time(e1))
    return float(np.median(ms)), float(np.min(ms))


def main():
    args = [a for a in sys.argv[1:] if "x" in a and not a.startswith("--")]
    find = "--find" in sys.argv
    reps = int(sys.argv[sys.argv.index("--reps") + 1]) if "--reps" in sys.argv else 30
    shapes = [tuple(int(x) for x in a.split("x")) for a in args] or [(1024, 1 << 20)]
    total = max(c * e for c, e in shapes)
    blob = torch.empty(total, dtype=torch.uint8, device="cuda")
    ss.fill_random_device(blob, 0x5EED0001)
    for count, each in shapes:
        nd = bytearray(ss.fill_random_host(16 * count, 0x5EED0003).tobytes())
        nd[8::16] = b"\xff" * count
        nblob = torch.from_numpy(np.frombuffer(bytes(nd), dtype=np.uint8).copy()).cuda()
        hay_off = (torch.arange(count + 1, dtype=torch.int64) * each).cuda()
        nd_off = (torch.arange(count + 1, dtype=torch.int64) * 16).cuda()
        hay = blob[:count * each]
        plan = ss.BatchPlan(hay, hay_off, nblob, nd_off, find=find)
        out = torch.empty(count, dtype=torch.int64 if find else torch.int32, device="cuda")
        call = (lambda: ss.find_batched(hay, hay_off, nblob, nd_off)) if find else (lambda: ss.search_batched(hay, hay_off, nblob, nd_off))
        row = {"problems": count, "each": each, "find": find}
        for rnd in range(2):
            row["call_ms_%d" % rnd], row["call_min_%d" % rnd] = [round(x, 4) for x in events_ms(call, reps)]
            row["plan_ms_%d" % rnd], row["plan_min_%d" % rnd] = [round(x, 4) for x in events_ms(lambda: plan.run(out), reps)]
        # the yardstick: the SAME bytes as ONE haystack through the single-problem kernel (ss_search_device_async: launch only, like
        # a plan run), bracketed the same way - what a batched call could at best cost
        if not find:
            one = ss.DynamicHipSearcher.new(bytes(nd[:16]))
            flag = torch.zeros(1, dtype=torch.int32, device="cuda")
            row["single_problem_ms"], row["single_problem_min"] = [round(x, 4) for x in events_ms(lambda: one.search_in_async(hay, flag), reps)]
            row["single_problem_gbps"] = round(count * each / row["single_problem_ms"] / 1e6, 1)
        nb = count * each
        row["call_gbps"] = round(nb / min(row["call_ms_0"], row["call_ms_1"]) / 1e6, 1)
        row["plan_gbps"] = round(nb / min(row["plan_ms_0"], row["plan_ms_1"]) / 1e6, 1)
        plan.close()
        print(json.dumps(row), flush=True)


if __name__ == "__main__":
    main()
