#!/usr/bin/env python3
"""rocprofv3 --kernel-trace csv -> one row per (kernel, grid): launches, median / min duration, GB/s for `bytes`.
    python tools/trace_summary.py TRACE_DIR BYTES [skip_first_n_per_group]"""
import collections
import csv
import glob
import json
import sys

import numpy as np

d, nbytes = sys.argv[1], float(sys.argv[2])
skip = int(sys.argv[3]) if len(sys.argv) > 3 else 0
groups = collections.OrderedDict()
for f in glob.glob(d + "/**/*kernel_trace.csv", recursive=True):
    rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
    for r in rows:
        key = (r["Kernel_Name"].split("(")[0][-70:], int(r.get("Grid_Size_X") or r.get("Grid_Size") or 0))
        groups.setdefault(key, []).append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for (name, grid), us in groups.items():
    us = us[skip:] if len(us) > skip + 3 else us
    med, mn = float(np.median(us)), float(np.min(us))
    print(json.dumps({"kernel": name, "grid_threads": grid, "launches": len(us), "median_us": round(med, 2), "min_us": round(mn, 2),
                      "gbps_median": round(nbytes / med / 1e3, 1), "gbps_best": round(nbytes / mn / 1e3, 1)}))
