#!/usr/bin/env python3
"""In-process A/B of library builds on BATCH PLANS: every build is dlopen()ed (RTLD_LOCAL) into one process, a plan of each is
made on the same device buffers, and their runs take turns (events around single runs; median / min of --reps per round, two
rounds).  Shapes as tools/batch_probe.py (COUNTxBYTES); `--present K`: every K-th needle is cut out of its haystack (early
exit inside a problem matters), else all absent.
    python tools/ab_batch_inproc.py --libs cur=...so old=...so --shapes 16384x65536 65536x16384 1024x1048576 [--find] [--present 2]"""
import argparse
import ctypes
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import sliceslice_rs_amd as ss  # noqa: E402
from batch_probe import events_ms  # noqa: E402

vp, sz = ctypes.c_void_p, ctypes.c_size_t


def load(path):
    L = ctypes.CDLL(path, mode=os.RTLD_LOCAL | os.RTLD_NOW)
    L.ss_batch_plan_create.argtypes = [vp, vp, vp, vp, vp, vp, vp, sz, ctypes.c_int, vp, ctypes.POINTER(vp)]
    L.ss_batch_plan_run.argtypes = [vp, vp, vp]
    L.ss_batch_plan_free.argtypes = [vp]
    L.ss_search_batched.argtypes = [vp, vp, vp, vp, vp, vp, vp, sz, vp, vp]
    L.ss_find_batched.argtypes = [vp, vp, vp, vp, vp, vp, sz, vp, vp]
    L.ss_last_error.restype = ctypes.c_char_p
    return L


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--libs", nargs="+", required=True)
    ap.add_argument("--shapes", nargs="+", default=[])
    ap.add_argument("--i386", action="store_true")
    ap.add_argument("--call", action="store_true", help="the UNPLANNED calls (ss_search_batched / ss_find_batched) instead of plan runs")
    ap.add_argument("--find", action="store_true")
    ap.add_argument("--present", type=int, default=0)
    ap.add_argument("--at", choices=["random", "start", "middle", "end"], default="random", help="where a present needle is cut out of its haystack")
    ap.add_argument("--reps", type=int, default=30)
    args = ap.parse_args()
    libs = [(s.split("=", 1)[0], load(s.split("=", 1)[1])) for s in args.libs]
    shapes = [tuple(int(x) for x in a.split("x")) for a in args.shapes]
    st = torch.cuda.current_stream().cuda_stream
    blob = torch.empty(max([c * e for c, e in shapes] + [16]), dtype=torch.uint8, device="cuda")
    ss.fill_random_device(blob, 0x5EED0001)
    if args.i386:
        # the reference's i386 loop: 4,585 words, one text (aliased ranges), every word present
        gd = os.path.join(ROOT, "tests", "golden", "data")
        raw = np.frombuffer(open(os.path.join(gd, "i386.txt"), "rb").read(), dtype=np.uint8)
        words = [w for w in open(os.path.join(gd, "words.txt"), "rb").read().split(b"\n") if w]
        text = torch.from_numpy(raw.copy()).cuda()
        hb = torch.zeros(len(words), dtype=torch.int64, device="cuda")
        he = torch.full((len(words),), raw.size, dtype=torch.int64, device="cuda")
        wb = torch.from_numpy(np.frombuffer(b"".join(words), dtype=np.uint8).copy()).cuda()
        wo = torch.from_numpy(np.cumsum(np.array([0] + [len(w) for w in words], dtype=np.int64))).cuda()
        out = torch.empty(len(words), dtype=torch.int64 if args.find else torch.int32, device="cuda")
        plans = []
        for name, L in libs:
            h = vp()
            if not args.call:
                rc = L.ss_batch_plan_create(text.data_ptr(), hb.data_ptr(), he.data_ptr(), wb.data_ptr(), wo.data_ptr(), wo.data_ptr() + 8,
                                            None, len(words), int(args.find), st, ctypes.byref(h))
                assert rc == 0, L.ss_last_error()
            plans.append((name, L, h))

        def run386(L, h):
            if not args.call:
                return L.ss_batch_plan_run(h, st, out.data_ptr())
            if args.find:
                return L.ss_find_batched(text.data_ptr(), hb.data_ptr(), he.data_ptr(), wb.data_ptr(), wo.data_ptr(), wo.data_ptr() + 8, len(words), st, out.data_ptr())
            return L.ss_search_batched(text.data_ptr(), hb.data_ptr(), he.data_ptr(), wb.data_ptr(), wo.data_ptr(), wo.data_ptr() + 8, None, len(words), st, out.data_ptr())
        row = {"workload": "the reference's i386 loop", "problems": len(words), "find": args.find, "unplanned_call": args.call}
        for rnd in range(3):
            for name, L, h in plans:
                ms, mn = events_ms(lambda: run386(L, h), args.reps)
                row["%s_ms_%d" % (name, rnd)] = round(ms, 4)
                torch.cuda.synchronize()
                assert os.environ.get("AB_NO_CHECK") == "1" or int((out >= 0).sum().item() if args.find else out.sum().item()) == len(words)
        for name, L, h in plans:
            if not args.call:
                L.ss_batch_plan_free(h)
        print(json.dumps(row), flush=True)
    for count, each in shapes:
        hay = blob[:count * each]
        nd = bytearray(ss.fill_random_host(16 * count, 0x5EED0003).tobytes())
        nd[8::16] = b"\xff" * count
        want = 0
        if args.present:
            rng = np.random.default_rng(5)
            idx = np.arange(0, count, args.present)
            at = rng.integers(0, each - 16, size=idx.size)
            if args.at != "random":
                at[:] = {"start": 0, "middle": each // 2, "end": each - 16}[args.at]
            src = (idx * each + at)[:, None] + np.arange(16)[None, :]
            cut = hay.cpu().numpy()[src.reshape(-1)].reshape(-1, 16)
            arr = np.frombuffer(bytes(nd), dtype=np.uint8).reshape(count, 16).copy()
            arr[idx] = cut
            nd = bytearray(arr.tobytes())
            want = idx.size
        nblob = torch.from_numpy(np.frombuffer(bytes(nd), dtype=np.uint8).copy()).cuda()
        hoff = (torch.arange(count + 1, dtype=torch.int64) * each).cuda()
        noff = (torch.arange(count + 1, dtype=torch.int64) * 16).cuda()
        out = torch.empty(count, dtype=torch.int64 if args.find else torch.int32, device="cuda")
        plans = []
        for name, L in libs:
            h = vp()
            if not args.call:
                rc = L.ss_batch_plan_create(hay.data_ptr(), hoff.data_ptr(), hoff.data_ptr() + 8, nblob.data_ptr(), noff.data_ptr(), noff.data_ptr() + 8,
                                            None, count, int(args.find), st, ctypes.byref(h))
                assert rc == 0, L.ss_last_error()
            plans.append((name, L, h))

        def run(L, h):
            if not args.call:
                return L.ss_batch_plan_run(h, st, out.data_ptr())
            if args.find:
                return L.ss_find_batched(hay.data_ptr(), hoff.data_ptr(), hoff.data_ptr() + 8, nblob.data_ptr(), noff.data_ptr(), noff.data_ptr() + 8,
                                         count, st, out.data_ptr())
            return L.ss_search_batched(hay.data_ptr(), hoff.data_ptr(), hoff.data_ptr() + 8, nblob.data_ptr(), noff.data_ptr(), noff.data_ptr() + 8,
                                       None, count, st, out.data_ptr())
        row = {"problems": count, "each": each, "find": args.find, "present_every": args.present, "at": args.at, "unplanned_call": args.call}
        for rnd in range(2):
            for name, L, h in plans:
                ms, mn = events_ms(lambda: run(L, h), args.reps)
                row["%s_ms_%d" % (name, rnd)], row["%s_min_%d" % (name, rnd)] = round(ms, 4), round(mn, 4)
                torch.cuda.synchronize()
                got = int((out >= 0).sum().item()) if args.find else int(out.sum().item())
                assert os.environ.get("AB_NO_CHECK") == "1" or got == want, (name, got, want)     # (AB_NO_CHECK=1: timing-only builds that leave parts of the protocol out)
        for name, L, h in plans:
            row[name + "_gbps"] = round(count * each / min(row[name + "_ms_0"], row[name + "_ms_1"]) / 1e6, 1)
            if not args.call:
                L.ss_batch_plan_free(h)
        print(json.dumps(row), flush=True)


if __name__ == "__main__":
    main()
