#!/usr/bin/env python3
"""What a wave spends in the candidate path (instrumented A/B build, -DSS_CAND_PROF: s_memtime stamps at the path's entry, behind
the cold part, around the second level and the compare, summed with atomics by lane 0): per settled (phrase, triple) case of a
survival-probe run the number of wave-tiles with candidates and the average ticks of each part.
    SLICESLICE_HIP_LIB=.../libsliceslice_hip_prof.so python tools/cand_prof.py profiles/r06/survival_probe_final.jsonl"""
import ctypes
import json
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import sliceslice_rs_amd as ss  # noqa: E402


def main():
    L = ctypes.CDLL(os.environ["SLICESLICE_HIP_LIB"])
    L.ss_debug_cand_prof.argtypes = [ctypes.POINTER(ctypes.c_ulonglong), ctypes.c_int]
    nbytes = 1 << 30
    gd = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "data")
    raw = open(os.path.join(gd, "i386.txt"), "rb").read()
    text = torch.from_numpy(np.tile(np.frombuffer(raw, dtype=np.uint8), nbytes // len(raw) + 1)[:nbytes].copy()).cuda()
    buf = (ctypes.c_ulonglong * 8)()
    seen = set()
    for line in open(sys.argv[1]):
        r = json.loads(line)
        if r["kind"] != "text":
            continue
        tri = tuple(r["state"]["in_force"])
        key = (r["needle"], tri)
        if key in seen or r["state"]["tiles3"] < 4 or tri[1] - tri[0] > 15 or tri[2] == tri[1]:
            continue
        seen.add(key)
        needle = r["needle"].encode("latin1")
        s = ss.DynamicHipSearcher.new(needle)
        s.set_filter(*tri)
        s.set_timing(True)
        for _ in range(6):
            s.search_in(text)
        st = s.tuning_state(text)
        torch.cuda.synchronize()
        assert L.ss_debug_cand_prof(buf, 1) == 0
        reps = 5
        ms = []
        for _ in range(reps):
            s.search_in(text)
            ms.append(s.last_kernel_ms())
        torch.cuda.synchronize()
        assert L.ss_debug_cand_prof(buf, 1) == 0
        c = [int(x) for x in buf]
        n = max(1, c[0])
        print(json.dumps({"needle": r["needle"][:30], "n": len(needle), "triple": tri, "tiles3": st["tiles3"], "lanes": st["lanes"], "deep": st["deep_lanes"],
                          "wg": s.last_launch()[0], "frac": round(nbytes / float(np.median(ms)) / 1e6 / 8000, 4),
                          "cand_wave_tiles_per_scan": c[0] // reps, "share_of_wave_tiles": round(c[0] / reps / (nbytes / 4096), 4), "per_piece_share": round(c[6] / n, 3), "ticks_ballots": round(c[5] / n, 1),
                          "ticks_total": round(c[1] / n, 1), "ticks_cold": round(c[2] / n, 1), "ticks_second_level": round(c[3] / n, 1),
                          "ticks_compare": round(c[4] / n, 1)}), flush=True)


if __name__ == "__main__":
    main()
