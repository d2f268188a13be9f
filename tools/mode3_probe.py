#!/usr/bin/env python3
"""Cross-lane kernels with (MODE 2) and without (MODE 3) the third first-phase byte, at four and six workgroups per CU, against the
automatic choice, interleaved in one process: filter pairs 16 or more apart on random bytes and on text.  Hooks build.
    SLICESLICE_HIP_LIB=...libsliceslice_hip_tuning.so python tools/mode3_probe.py [--gib 1,8]"""
import argparse
import json
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import sliceslice_rs_amd as ss  # noqa: E402
from occ_probe import paired_ms  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gib", default="1,8")
    args = ap.parse_args()
    gibs = [float(x) for x in args.gib.split(",")]
    big = int(max(gibs) * (1 << 30))
    hay = torch.empty(big, dtype=torch.uint8, device="cuda")
    ss.fill_random_device(hay, 0x5EED0001)
    gd = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "data")
    raw = np.frombuffer(open(os.path.join(gd, "i386.txt"), "rb").read(), dtype=np.uint8)
    text = torch.from_numpy(np.tile(raw, (1 << 30) // raw.size + 1)[: 1 << 30].copy()).cuda()

    def absent(n):
        a = ss.fill_random_host(n, 0x5EED0002)
        a[n // 2] = 0xFF
        return a.tobytes()
    cases = [("random", absent(n), g) for g in gibs for n in (24, 200, 700)]
    cases += [("text", ph, 1.0) for ph in (b"segment descriptor table entries are", b" the quick brown fox ", b"privilege level zero!",
                                            b"there is not another one of these", b"protection exception handler must")]
    variants = {"auto": 0, "m2w4": 40241, "m3w4": 40341, "m2w6": 60241, "m3w6": 60341}
    for kind, nd, gib in cases:
        h = hay[: int(gib * (1 << 30))] if kind == "random" else text
        ss_ = []
        for name, v in variants.items():
            s = ss.DynamicHipSearcher.new(nd)
            s.set_filter(0, len(nd) - 1)
            s.set_variant(v)
            ss_.append(s)
        res, ms = paired_ms(ss_, h)
        cen = ss_[0].census(h)
        row = {"kind": kind, "n": len(nd), "gib": gib, "found": res, "census": cen, "auto_mode": ss_[0].last_mode, "auto_wg": ss_[0].last_launch()[0]}
        row.update({name: round(h.numel() / m / 1e6, 1) for name, m in zip(variants, ms)})
        row["auto_over_best"] = round(min(ms[1:]) / ms[0], 4)
        print(json.dumps(row), flush=True)


if __name__ == "__main__":
    main()
