#!/usr/bin/env python3
"""Condense a tools/profile_round.sh output directory into the tracked profiles/ files.

usage: python tools/summarize_profile.py gpurun_out/prof_<tag> <tag>
writes profiles/<tag>_kernel_stats.csv   (rocprofv3 --kernel-trace --stats summary, verbatim top rows)
       profiles/<tag>_pmc.json           (per-launch counters of the RK45 kernel + corrected HBM bytes)
       profiles/traffic_latest.json      (what bench.py reports as roofline.traffic)

HBM bytes follow /opt/skills/guides/MI355X_MICROARCH.md "HBM": FETCH_SIZE and WRITE_SIZE are
in KiB, collected in SEPARATE --pmc passes; on gfx950 FETCH_SIZE reports half of the bytes
of coalesced streaming reads, so it is doubled (cross-check: the doubled figure equals the
bytes the kernel must read -- forcing block + SpatialParams + y0 -- to 0.01 %); WRITE_SIZE is
exact.
"""
import csv
import glob
import json
import os
import sys

src, tag = sys.argv[1], sys.argv[2]
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
prof = os.path.join(root, "profiles")
os.makedirs(prof, exist_ok=True)


def rows(sub):
    f = sorted(glob.glob(os.path.join(src, sub, "*", "*_counter_collection.csv")), key=os.path.getmtime)
    return list(csv.DictReader(open(f[-1]))) if f else []   # (the newest: gpurun merges the directories of several calls)


stats = sorted(glob.glob(os.path.join(src, "trace", "*", "*_kernel_stats.csv")), key=os.path.getmtime)[-1]
lines = open(stats).read().splitlines()
with open(os.path.join(prof, f"{tag}_kernel_stats.csv"), "w") as f:
    f.write("\n".join(lines[:12]) + "\n")
# the headline kernel: the queue kernel (persistent grid over (time piece, tile) items) or, with --time-pieces 0,
# the one-wave-per-tile kernel; the busiest of them in this trace
kstat = max((r for r in csv.DictReader(open(stats)) if "rk45_queue_kernel" in r["Name"] or "rk45_multi_kernel" in r["Name"]),
            key=lambda r: float(r["Percentage"]))

out = {"kernel": kstat["Name"], "calls": int(kstat["Calls"]), "avg_duration_ms": float(kstat["AverageNs"]) / 1e6,
       "min_duration_ms": float(kstat["MinNs"]) / 1e6, "max_duration_ms": float(kstat["MaxNs"]) / 1e6,
       "percentage_of_gpu_time": float(kstat["Percentage"])}
counters = {}
for sub in ("pmc_fetch", "pmc_write", "pmc_grbm"):
    for r in rows(sub):
        if r["Kernel_Name"] == kstat["Name"]:  # the dominant kernel only (bench.py also runs the dense section)
            counters.setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
            out.setdefault("vgpr_count", int(r["VGPR_Count"]))
            out.setdefault("accum_vgpr_count", int(r.get("Accum_VGPR_Count") or 0))
            out.setdefault("sgpr_count", int(r["SGPR_Count"]))
            out.setdefault("lds_block_size", int(r["LDS_Block_Size"]))
            out.setdefault("grid_size", int(r["Grid_Size"]))
            out.setdefault("workgroup_size", int(r["Workgroup_Size"]))
avg = {k: sum(v) / len(v) for k, v in counters.items()}
out["counters_per_launch_avg"] = avg
if "FETCH_SIZE" in avg and "WRITE_SIZE" in avg:
    rd = avg["FETCH_SIZE"] * 1024.0 * 2.0   # KiB -> B, gfx950 half-count correction
    wr = avg["WRITE_SIZE"] * 1024.0
    out["hbm_read_bytes_per_launch"] = rd
    out["hbm_write_bytes_per_launch"] = wr
    out["hbm_bytes_per_launch"] = rd + wr
    out["hbm_GBps"] = (rd + wr) / (out["avg_duration_ms"] * 1e-3) / 1e9
if "GRBM_GUI_ACTIVE" in avg:
    out["effective_clock_GHz"] = avg["GRBM_GUI_ACTIVE"] / 8.0 / (out["avg_duration_ms"] * 1e-3) / 1e9
p = os.path.join(src, "bench_trace.json")
if os.path.exists(p):
    try:
        out["bench_line_under_trace"] = json.loads(open(p).read().strip().splitlines()[-1])
    except Exception:
        pass
json.dump(out, open(os.path.join(prof, f"{tag}_pmc.json"), "w"), indent=1)
if "hbm_bytes_per_launch" in out:
    json.dump({"tag": tag, "hbm_bytes_per_launch": out["hbm_bytes_per_launch"],
               "hbm_read_bytes_per_launch": out["hbm_read_bytes_per_launch"],
               "hbm_write_bytes_per_launch": out["hbm_write_bytes_per_launch"],
               "workload": out.get("bench_line_under_trace", {}).get("config", {}).get("workload"),
               "hillslopes_per_gpu": out.get("bench_line_under_trace", {}).get("config", {}).get("hillslopes_per_gpu"),
               "per_gpu_key": out.get("bench_line_under_trace", {}).get("config", {}).get("per_gpu_key")},
              open(os.path.join(prof, "traffic_latest.json"), "w"), indent=1)
print(json.dumps({k: v for k, v in out.items() if k != "bench_line_under_trace"}, indent=1))
