#!/usr/bin/env python3
"""One named workload, launches of ONE kernel - a target for `rocprofv3 --kernel-trace --stats` and the separate `--pmc` passes
(tools/profile_kernels.sh).  Like bench.py: a 50 ms spin of back-to-back launches first (clocks ramp up over the first
milliseconds after an idle gap - longer than a whole series of sub-millisecond launches), then `launches` (default 24) measured
ones; tools/collect_kernel_profiles.py takes the LAST `launches` rows of the kernel from rocprofv3's trace, so that the
rocprofv3 median is the median of the same launches the hipEvent median printed here covers.  Prints one JSON line: the case,
the kernel family expected, the algorithmic bytes per launch and the kernel time by hipEvents (median, min).

  cases:  headline1g   1 GiB random, 16-byte absent needle, new()              (scan_kernel<3,0,...>)
          near16       the same needle with set_filter(0, 1, 2); with variant 42041 (tuning build) the 8-byte first phase's three-byte filter
          onebyte      1 GiB random, 1-byte absent needle (8-byte loads)       (scan_kernel<0,0,true,...,L8>)
          mode2        1 GiB random, 128-byte needle, set_filter(0, 127)       (scan_kernel<.,2,...> cross-lane)
          far_pair     1 GiB random, 2000-byte needle, set_filter(0, 1999)     (rounds 1-3: the two-stream kernels, case "mode1"; now the
                       single-stream kernels with the caller's far byte checked in memory)
          long_wp      1 GiB random, 2000-byte needle, with_position(1999)     (single stream: partner byte next to 1999)
          long_new     1 GiB random, 2000-byte needle, new()                   (single stream again)
          find         1 GiB random, 16-byte absent needle, find()             (FIND kernel)
          batched      4096 x 1 MiB, 4096 absent 16-byte needles, one call     (scan_batched_plan_kernel<4, false, false>)
          batched_plan the same problems through an ss_batch_plan: one run() per launch  (scan_batched_plan_kernel<4, false, true>)
          random_text_needle   'there is not another one of these' through new() on RANDOM bytes: the needle looks like text, the
                       haystack is not - workgroups per CU from the previous scan's candidate rate, not from the needle
          text_worst   i386.txt tiled to 1 GiB, letters-only absent phrase, new()
          text_refpair the same phrase with the reference's pair (0, n-1), set verbatim
          text_wp      the same phrase through with_position(n-1)
          text_spaces  ' the quick brown fox ' with the reference's pair (' ', ' '), set verbatim
          text_spaces_new   the same needle through new(): filter bytes 'q', 'x', 'k'
          text_common_new   'there is not another one of these' through new(): common letters only
          <case>_static     any of the above with launch tuning OFF (ss_set_autotune(0)): the constructors' static bytes, the static
                            schedule order, the needle-byte guess for workgroups per CU - e.g. text_refpair_static keeps the caller's
                            pair in the cross-lane kernels, where text_refpair (tuning on) runs its near form in the single-stream ones
"""
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import sliceslice_rs_amd as ss  # noqa: E402

SEED_HAY, SEED_NEEDLE = 0x5EED0001, 0x5EED0002


def absent(n, seed=SEED_NEEDLE):
    nd = bytearray(ss.fill_random_host(n, seed).tobytes())
    nd[0 if n == 1 else (1 if n == 2 else n // 2)] = 0xFF
    return bytes(nd)


def main():
    case = sys.argv[1]
    launches = int(sys.argv[2]) if len(sys.argv) > 2 else 24
    static = case.endswith("_static")
    if static:
        case = case[:-len("_static")]
        ss.set_autotune(False)

    def spin(fn):
        t_end = time.perf_counter() + 0.05
        while time.perf_counter() < t_end:
            fn()
        torch.cuda.synchronize()
    n_bytes = 1 << 30
    out = {"case": case + ("_static" if static else ""), "launches": launches, "autotune": not static}
    if case.startswith("text"):
        raw = np.frombuffer(open(os.path.join(os.path.dirname(__file__), "..", "tests", "golden", "data", "i386.txt"), "rb").read(), dtype=np.uint8)
        hay = torch.from_numpy(raw.copy()).cuda().repeat(n_bytes // raw.size)
    elif case.startswith("batched"):
        hay = torch.empty(4096 << 20, dtype=torch.uint8, device="cuda")
        ss.fill_random_device(hay, SEED_HAY)
    else:
        hay = torch.empty(n_bytes, dtype=torch.uint8, device="cuda")
        ss.fill_random_device(hay, SEED_HAY)
    torch.cuda.synchronize()
    out["algorithmic_bytes_per_launch"] = hay.numel()
    if case.startswith("batched"):
        count, each = 4096, 1 << 20
        nd = bytearray(ss.fill_random_host(16 * count, SEED_NEEDLE + 1).tobytes())
        for i in range(count):
            nd[16 * i + 8] = 0xFF
        nblob = torch.from_numpy(np.frombuffer(bytes(nd), dtype=np.uint8).copy()).cuda()
        hay_off = (torch.arange(count + 1, dtype=torch.int64) * each).cuda()
        nd_off = (torch.arange(count + 1, dtype=torch.int64) * 16).cuda()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        plan = ss.BatchPlan(hay, hay_off, nblob, nd_off) if case == "batched_plan" else None
        flags = torch.empty(count, dtype=torch.int32, device="cuda")
        call = (lambda: plan.run(flags)) if plan is not None else (lambda: ss.search_batched(hay, hay_off, nblob, nd_off))
        spin(call)
        ms = []
        for _ in range(launches):
            e0.record()
            found = call()
            e1.record()
            e1.synchronize()
            ms.append(e0.elapsed_time(e1))
        assert int(found.sum().item()) == 0
        out.update(kernel="scan_batched_plan_kernel", ms=round(float(np.median(ms)), 4), ms_min=round(float(np.min(ms)), 4),
                   note="ms = the whole call by events (plan kernel + scan grid + the host's launch path); the kernel alone: rocprofv3")
    else:
        phrase = b"segment descriptor table entries are"
        def exact(nd):                                   # the reference's pair (needle[0], needle[n-1]), verbatim
            e = ss.DynamicHipSearcher.new(nd)
            e.set_filter(0, len(nd) - 1)
            return e
        def near(nd):                                    # all three filter bytes within four bytes (the L8 three-byte experiment)
            e = ss.DynamicHipSearcher.new(nd)
            e.set_filter(0, 1, 2)
            return e
        s = {
            "headline1g": lambda: ss.DynamicHipSearcher.new(absent(16)),
            "near16": lambda: near(absent(16)),
            "onebyte": lambda: ss.DynamicHipSearcher.new(absent(1)),
            "mode2": lambda: exact(absent(128)),
            "mode1": lambda: exact(absent(2000)),
            "far_pair": lambda: exact(absent(2000)),
            "random_text_needle": lambda: ss.DynamicHipSearcher.new(b"there is not another one of these"),
            "long_wp": lambda: ss.DynamicHipSearcher.with_position(absent(2000), 1999),
            "text_wp": lambda: ss.DynamicHipSearcher.with_position(phrase, len(phrase) - 1),
            "long_new": lambda: ss.DynamicHipSearcher.new(absent(2000)),
            "find": lambda: ss.DynamicHipSearcher.new(absent(16)),
            "text_worst": lambda: ss.DynamicHipSearcher.new(phrase),
            "text_refpair": lambda: exact(phrase),
            "text_spaces": lambda: exact(b" the quick brown fox "),
            "text_spaces_new": lambda: ss.DynamicHipSearcher.new(b" the quick brown fox "),
            "text_common_new": lambda: ss.DynamicHipSearcher.new(b"there is not another one of these"),
        }[case]()
        s.set_timing(True)
        if len(sys.argv) > 3:
            s.set_variant(int(sys.argv[3]))              # tuning build: e.g. 40041 / 60041 = at most 4 / 6 workgroups per CU
            out["variant"] = int(sys.argv[3])
        spin(lambda: s.find(hay) if case == "find" else s.search_in(hay))
        ms = []
        for _ in range(launches):
            r = s.find(hay) if case == "find" else s.search_in(hay)
            ms.append(s.last_kernel_ms())
        assert r in (False, None), r
        out.update(kernel="scan_kernel", filter_bytes=list(s.filter3), ms=round(float(np.median(ms)), 4), ms_min=round(float(np.min(ms)), 4))
        out["workgroups_per_cu"], out["grid"] = s.last_launch()
        st = s.tuning_state(hay)
        out["tuning"] = {k: st[k] for k in ("census_state", "tiles3", "lanes", "deep_lanes", "triple_state", "in_force", "own", "order_measured", "kernel_mode", "settled")}
        if ss.lib().has_hooks:
            out["census"] = s.census(hay)
    out["gbps"] = round(hay.numel() / out["ms"] / 1e6, 1)
    print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
