#!/usr/bin/env python3
"""gpurun_out/<tag>_<case>_{kt,pmc,sq} (tools/profile_kernels.sh) -> profiles/<round>/kernels.json + kernels.md:
per case the dominant kernel's name, rocprofv3 duration of the MEASURED launches (the last `launches` rows of the kernel in the
trace - the 50 ms spin in front of them is excluded: median, min, mean), achieved GB/s by the median and fraction of the 8 TB/s peak,
HBM bytes fetched per launch (FETCH_SIZE x 2 x 1024: the gfx950 correction and KiB unit of
MI355X_MICROARCH.md's HBM section) over the algorithmic bytes, and the SQ instruction mix per KiB piece."""
import csv
import glob
import json
import os
import statistics
import sys

tag, rnd = sys.argv[1], sys.argv[2]
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src, dst = os.path.join(root, "gpurun_out"), os.path.join(root, "profiles", rnd)
os.makedirs(dst, exist_ok=True)
# registers as the compiler allocated them (build() -> csrc/kernel_resources.json, snapshot in profiles/<round>/): rocprofv3's
# VGPR_Count column is the architectural half of the unified file (64 for all of these), not what limits the occupancy
res_path = os.path.join(dst, "kernel_resources.json")
resources = json.load(open(res_path)) if os.path.exists(res_path) else []


def allocated(kernel_name):
    key = kernel_name.replace(" ", "")
    for r in resources:
        if r["name"].replace(" ", "").startswith(key.split("(")[0]):
            return r
    return None


rows = []
for kt in sorted(glob.glob(os.path.join(src, f"{tag}_*_kt"))):
    case = os.path.basename(kt)[len(tag) + 1:-3]
    line = None
    for l in open(kt + ".log", errors="replace"):
        if l.startswith('{"case"'):
            line = json.loads(l)
    if line is None:
        continue
    want = line["kernel"]
    stats = [r for r in csv.DictReader(open(os.path.join(kt, "r_kernel_stats.csv"))) if want in r["Name"]]
    stats.sort(key=lambda r: -float(r["TotalDurationNs"]))
    if not stats:
        continue
    k = stats[0]
    alg = line["algorithmic_bytes_per_launch"]
    avg_ns = float(k["AverageNs"])
    # the measured launches: the last `launches` rows of this kernel in the trace (what the hipEvent median covers)
    tr = [r for r in csv.DictReader(open(os.path.join(kt, "r_kernel_trace.csv"))) if r["Kernel_Name"] == k["Name"]]
    tr.sort(key=lambda r: int(r["Start_Timestamp"]))
    durs = [int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in tr[-int(line["launches"]):]]
    med_ns, min_ns, mean_ns = statistics.median(durs), min(durs), statistics.mean(durs)
    # the other processes of the case (profile_kernels.sh runs three): their medians; the figure of the table is the median of
    # the processes' medians, and a difference between rounds inside `spread` is placement noise, not a regression
    proc = [med_ns]
    for rep in ("b", "c"):
        t2 = os.path.join(kt + rep, "r_kernel_trace.csv")
        if os.path.exists(t2):
            tr2 = [r for r in csv.DictReader(open(t2)) if r["Kernel_Name"] == k["Name"]]
            tr2.sort(key=lambda r: int(r["Start_Timestamp"]))
            d2 = [int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in tr2[-int(line["launches"]):]]
            if d2:
                proc.append(statistics.median(d2))
    first_process_med_ns = med_ns
    med_ns = statistics.median(proc)
    row = {"case": case, "kernel": k["Name"], "calls_in_trace": int(k["Calls"]), "measured_launches": len(durs),
           "rocprof_median_ms": round(med_ns / 1e6, 4), "rocprof_min_ms": round(min_ns / 1e6, 4), "rocprof_mean_ms": round(mean_ns / 1e6, 4),
           "rocprof_avg_ms_all_calls_incl_spin": round(avg_ns / 1e6, 4),
           "hipevent_median_ms": line["ms"], "hipevent_min_ms": line.get("ms_min"), "algorithmic_bytes_per_launch": alg,
           "gbps": round(alg / med_ns, 1), "frac_of_8tbps": round(alg / med_ns / 8000, 4), "gbps_best": round(alg / min_ns, 1),
           "filter_bytes": line.get("filter_bytes"), "processes": len(proc), "process_medians_ms": [round(x / 1e6, 4) for x in proc],
           "frac_min_max_over_processes": [round(alg / max(proc) / 8000, 4), round(alg / min(proc) / 8000, 4)],
           "workgroups_per_cu": line.get("workgroups_per_cu"), "census": line.get("census")}
    # the second clock: hipEvents around the same launches in a process WITHOUT the profiler (what a caller's own timing sees)
    np_log = os.path.join(src, f"{tag}_{case}_noprof.log")
    if os.path.exists(np_log):
        for l in open(np_log, errors="replace"):
            if l.startswith('{"case"'):
                j = json.loads(l)
                row["hipevent_no_profiler_ms"] = j["ms"]
                row["frac_by_hipevents_no_profiler"] = round(alg / (j["ms"] * 1e6) / 8000, 4)
                row["workgroups_per_cu_no_profiler"] = j.get("workgroups_per_cu")
    pmc = os.path.join(src, f"{tag}_{case}_pmc", "r_counter_collection.csv")
    if os.path.exists(pmc):
        vals = [float(r["Counter_Value"]) for r in csv.DictReader(open(pmc)) if want in r["Kernel_Name"] and r["Counter_Name"] == "FETCH_SIZE"]
        if vals:
            fetched = 2 * statistics.median(vals) * 1024
            row.update(vgpr=None, hbm_bytes_fetched_per_launch=fetched, fetched_over_algorithmic=round(fetched / alg, 4))
        regs = [r for r in csv.DictReader(open(pmc)) if want in r["Kernel_Name"]]
        if regs:
            row["vgpr"] = int(regs[0]["VGPR_Count"])
            row["lds_block_size"] = int(regs[0]["LDS_Block_Size"])
    sq = os.path.join(src, f"{tag}_{case}_sq", "r_counter_collection.csv")
    if os.path.exists(sq):
        acc = {}
        for r in csv.DictReader(open(sq)):
            if want in r["Kernel_Name"]:
                acc.setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
        pieces = alg / 1024
        med = {c: statistics.median(v) for c, v in acc.items()}
        row["sq_per_kib"] = {c.replace("SQ_INSTS_", "").lower(): round(v / pieces, 2) for c, v in med.items() if c.startswith("SQ_INSTS_")}
        if med.get("SQ_WAVE_CYCLES"):
            row["wait_fraction"] = round(med.get("SQ_WAIT_ANY", 0) / med["SQ_WAVE_CYCLES"], 3)
    a = allocated(k["Name"])
    if a:
        row.update(vgpr=a["vgprs"], waves_per_simd=a["waves_per_simd"], sgprs=a.get("sgprs"), scratch_bytes_per_lane=a.get("scratch_bytes_per_lane"))
    rows.append(row)
    for kind in ("kt",):
        os.makedirs(os.path.join(dst, "kernel_stats"), exist_ok=True)
        import shutil
        shutil.copy(os.path.join(kt, "r_kernel_stats.csv"), os.path.join(dst, "kernel_stats", f"{case}_kernel_stats.csv"))
json.dump(rows, open(os.path.join(dst, "kernels.json"), "w"), indent=1)
with open(os.path.join(dst, "kernels.md"), "w") as fh:
    fh.write("Three processes per case; ms = median over the processes of each process's median over its 24 measured launches; "
             "`frac` the same, with the slowest and the fastest process behind it.  TWO clocks: rocprofv3's kernel durations (kernel-trace: "
             "begin to end of the dispatch), and hipEvents on the launch stream around the same launches in a process WITHOUT the profiler - "
             "what a caller's own timing sees (bench.py's `configs.*` rows are that clock).\n\n")
    fh.write("| case | kernel | median ms (per process) | GB/s | frac of 8 TB/s (min - max over the processes) | hipEvents, no profiler: ms / frac | fetched / algorithmic | VGPRs allocated (waves per SIMD) | workgroups per CU | VALU / SALU / LDS / VMEM per KiB | wait |\n|---|---|---|---|---|---|---|---|---|---|---|\n")
    for r in rows:
        sqk = r.get("sq_per_kib", {})
        fh.write("| %s | `%s` | %.4f (%s) | %.0f | %.3f (%.3f - %.3f) | %s / %s | %s | %s | %s | %s / %s / %s / %s | %s |\n" % (
            r["case"], r["kernel"].replace("void ", "")[:70], r["rocprof_median_ms"], ", ".join("%.4f" % x for x in r["process_medians_ms"]), r["gbps"], r["frac_of_8tbps"],
            r["frac_min_max_over_processes"][0], r["frac_min_max_over_processes"][1],
            r.get("hipevent_no_profiler_ms", "-"), r.get("frac_by_hipevents_no_profiler", "-"),
            r.get("fetched_over_algorithmic", "-"), "%s (%s)" % (r.get("vgpr", "-"), r.get("waves_per_simd", "-")), r.get("workgroups_per_cu") or "-",
            sqk.get("valu", "-"), sqk.get("salu", "-"), sqk.get("lds", "-"),
            sqk.get("vmem_rd", "-"), r.get("wait_fraction", "-")))
print(open(os.path.join(dst, "kernels.md")).read())
