#!/usr/bin/env python3
"""The noise model of the timing tests: runs `pytest -m "gpu and timing"` N times (default 10) with SS_TIMING_LOG set, and
summarises what the tests measured - min / median / max per quantity, and whether every run was green.
    python tools/timing_spread.py [runs=10] > profiles/r05/timing_test_spread.jsonl"""
import json
import os
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main():
    runs = int(sys.argv[1]) if len(sys.argv) > 1 else 10
    log = tempfile.mktemp(prefix="ss_timing_", suffix=".jsonl", dir="/tmp")
    outcomes = []
    for k in range(runs):
        t0 = time.time()
        r = subprocess.run([sys.executable, "-m", "pytest", os.path.join(ROOT, "tests"), "-q", "-m", "gpu and timing", "-p", "no:cacheprovider"],
                           cwd=ROOT, env=dict(os.environ, SS_TIMING_LOG=log), capture_output=True, text=True)
        tail = [l for l in r.stdout.splitlines() if " passed" in l or " failed" in l]
        outcomes.append({"run": k, "rc": r.returncode, "seconds": round(time.time() - t0, 1), "summary": tail[-1] if tail else r.stdout[-200:]})
        print(json.dumps(outcomes[-1]), flush=True)
    rows = [json.loads(l) for l in open(log)] if os.path.exists(log) else []
    by = {}
    for r in rows:
        for k, v in r.items():
            if k != "test" and isinstance(v, (int, float)):
                by.setdefault((r["test"], k), []).append(v)
    for (test, k), v in sorted(by.items()):
        v.sort()
        print(json.dumps({"test": test, "quantity": k, "samples": len(v), "min": v[0], "median": v[len(v) // 2], "max": v[-1]}), flush=True)
    for r in rows:
        if "line" in r:
            print(json.dumps(r), flush=True)
    print(json.dumps({"runs": runs, "all_green": all(o["rc"] == 0 for o in outcomes)}), flush=True)


if __name__ == "__main__":
    main()
