#!/usr/bin/env python3
"""gpurun_out/<tag>_* (written by tools/capture_profiles.sh on the GPU box) -> profiles/<round>/ summaries:
the bench line, rocprofv3's kernel stats, per-launch FETCH_SIZE (x2 correction, KiB units - see
MI355X_MICROARCH.md's HBM section) and the SQ instruction mix per KiB piece."""
import csv
import json
import os
import shutil
import statistics
import sys

tag, rnd = sys.argv[1], sys.argv[2]
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src, dst = os.path.join(root, "gpurun_out"), os.path.join(root, "profiles", rnd)
os.makedirs(dst, exist_ok=True)
shutil.copy(os.path.join(src, f"{tag}_bench.json"), os.path.join(dst, "bench64g.json"))
shutil.copy(os.path.join(src, f"{tag}_kt", "r_kernel_stats.csv"), os.path.join(dst, "bench64g_kernel_stats.csv"))
shutil.copy(os.path.join(src, f"{tag}_configs.jsonl"), os.path.join(dst, "configs_1_3_5_text.jsonl"))
# the bench line printed by the SAME command rocprofv3 traced (its hipEvent average must agree with
# rocprofv3's; whole runs differ by a few percent from process to process)
for line in open(os.path.join(src, f"{tag}_kt.log"), errors="replace"):
    if line.startswith('{"metric"'):
        open(os.path.join(dst, "bench64g_under_rocprofv3.json"), "w").write(line)
bench = json.load(open(os.path.join(dst, "bench64g.json")))
hay = bench["config"]["haystack_bytes"]


def scan_rows(path):
    return [r for r in csv.DictReader(open(path)) if "scan_kernel" in r["Kernel_Name"]]


rows = scan_rows(os.path.join(src, f"{tag}_pmc", "r_counter_collection.csv"))
keep = ["Kernel_Name", "Grid_Size", "Workgroup_Size", "VGPR_Count", "SGPR_Count", "LDS_Block_Size", "Counter_Name", "Counter_Value"]
with open(os.path.join(dst, "bench64g_pmc_fetch_size.csv"), "w", newline="") as fh:
    w = csv.writer(fh)
    w.writerow(keep)
    for r in rows:
        w.writerow([r[k] for k in keep])
fs = statistics.median(float(r["Counter_Value"]) for r in rows)
traffic = 2 * fs * 1024
json.dump({"source": f"rocprofv3 --pmc FETCH_SIZE, bench.py --steps 5 --warmup 2 ({len(rows)} launches)",
           "kernel": rows[0]["Kernel_Name"], "fetch_size_kib_median": fs, "correction": "x2 (gfx950), KiB units",
           "hbm_read_bytes_per_launch": traffic, "haystack_bytes": hay,
           "hbm_read_bytes_per_haystack_byte": traffic / hay},
          open(os.path.join(root, "profiles", "pmc_traffic.json"), "w"), indent=1)
acc = {}
for r in scan_rows(os.path.join(src, f"{tag}_sq", "r_counter_collection.csv")):
    acc.setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
with open(os.path.join(dst, "bench64g_pmc_sq.csv"), "w", newline="") as fh:
    w = csv.writer(fh)
    w.writerow(["Counter_Name", "median_per_launch", "per_KiB_piece"])
    for k, v in sorted(acc.items()):
        m = statistics.median(v)
        w.writerow([k, m, m / (hay / 1024)])
        print(k, round(m / (hay / 1024), 3))
ks = [r for r in csv.DictReader(open(os.path.join(dst, "bench64g_kernel_stats.csv"))) if "scan_kernel" in r["Name"]]
print("traffic/haystack", round(traffic / hay, 4), "| bench value", bench["value"], "kernel_ms_avg", bench["roofline"]["kernel_ms_avg"],
      "| rocprof avg ns", ks[0]["AverageNs"] if ks else None, "calls", ks[0]["Calls"] if ks else None)

# rocprofv3's own durations of the TIMED launches (the last `steps` scan launches of the traced command) next to the
# hipEvent average bench.py printed in that same command
try:
    under = json.load(open(os.path.join(dst, "bench64g_under_rocprofv3.json")))
    tr = [r for r in csv.DictReader(open(os.path.join(src, f"{tag}_kt", "r_kernel_trace.csv"))) if "scan_kernel" in r["Kernel_Name"]]
    tr.sort(key=lambda r: int(r["Start_Timestamp"]))
    last = tr[-under["steps"]:]
    durs = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6 for r in last]
    agree = {"command": "rocprofv3 --kernel-trace --stats -- python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-configs",
             "rocprofv3_avg_ms_of_the_timed_launches": round(sum(durs) / len(durs), 4), "launches": len(durs),
             "rocprofv3_avg_ms_all_launches_incl_warmup": round(float(ks[0]["AverageNs"]) / 1e6, 4) if ks else None,
             "bench_hipevent_kernel_ms_avg": under["roofline"]["kernel_ms_avg"],
             "achieved_gbps_from_rocprofv3": round(hay / (sum(durs) / len(durs)) / 1e6, 1)}
    json.dump(agree, open(os.path.join(dst, "rocprof_vs_hipevents.json"), "w"), indent=1)
    print(agree)
except Exception as e:
    print("agreement file not written:", e)
