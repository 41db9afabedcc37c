#!/usr/bin/env python3
"""Condense a tools/pmc_issue.sh output directory into profiles/<tag>_pmc_issue.json.

usage: python tools/summarize_pmc_issue.py gpurun_out/pmc_issue_<tag> <tag>

Per workload (headline / wet / cold): the counters of the busiest RK45 kernel averaged over its launches, the
bench line of the same process (lane-attempts per launch), and derived figures:
  valu_insts_per_wave_attempt = SQ_INSTS_VALU / (lane-attempts / 64)   (a wave-attempt = 64 lane-attempts: lower
                                bound on the wave-level loop trips, exact while the lanes of a wave stay together)
  issue_busy = SQ_ACTIVE_INST_VALU * 4 / (1024 SIMDs * cycles), cycles = GRBM_GUI_ACTIVE / 8 (sum over the 8 XCDs):
               the fraction of the SIMDs' issue time spent issuing VALU instructions (SQ_ACTIVE_INST_* count
               quad-cycles, /opt/skills/guides/MI355X_MICROARCH.md "Per-instruction cycle constants")
  active_lanes_per_valu_inst = SQ_THREAD_CYCLES_VALU / SQ_ACTIVE_INST_VALU / 4 ... as in profiles/r02_*_pmc.json
"""
import csv
import glob
import json
import os
import sys

src, tag = sys.argv[1], sys.argv[2]
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
out = {"tag": tag, "source": "tools/pmc_issue.sh (rocprofv3 --pmc, two SQ passes per workload, bench.py --steps 3 --warmup 1)",
       "units": "SQ_*_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* count quad-cycles; GRBM_GUI_ACTIVE is summed over the 8 XCDs",
       "workloads": {}}
for w in ("headline", "wet", "saturating", "active_surface", "cold", "cold_radau", "cold_radau_continue"):
    counters, kname, meta = {}, None, {}
    for p in ("A", "B"):
        files = sorted(glob.glob(os.path.join(src, f"{w}_{p}", "*", "*_counter_collection.csv")), key=os.path.getmtime)   # (gpurun merges the directories of several calls: take the newest)
        if not files:
            continue
        want = ("radau_fallback_kernel",) if w.startswith("cold_radau") else ("rk45_queue_kernel", "rk45_multi_kernel")
        rows = [r for r in csv.DictReader(open(files[-1])) if any(k in r["Kernel_Name"] for k in want)]
        if not rows:
            continue
        by = {}
        for r in rows:
            by.setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
        kname = rows[0]["Kernel_Name"]
        meta = {"vgpr_count": int(rows[0]["VGPR_Count"]), "sgpr_count": int(rows[0]["SGPR_Count"]),
                "scratch_size": int(rows[0].get("Scratch_Size") or 0), "grid_size": int(rows[0]["Grid_Size"]),
                "workgroup_size": int(rows[0]["Workgroup_Size"]), "launches_averaged": len(next(iter(by.values())))}
        for k, v in by.items():
            counters[k if k not in counters else k + "_pass" + p] = sum(v) / len(v)
    if not counters:
        continue
    line = None
    for p in ("A", "B"):
        f = os.path.join(src, f"bench_{w}_{p}.json")
        if os.path.exists(f):
            try:
                line = json.loads(open(f).read().strip().splitlines()[-1])
                break
            except Exception:
                pass
    d = {"kernel": kname, **meta, "counters_per_launch_avg": counters}
    if line:
        att = float(line["attempts_per_pass"])
        if w.startswith("cold_radau"):   # the pass's own (implicit) attempts: total minus the RK45 launch's (the "cold" workload)
            att -= out["workloads"].get("cold", {}).get("derived", {}).get("lane_attempts", 0.0)
        wa = att / 64.0
        d["bench_line_under_pmc"] = {k: line[k] for k in ("value", "ms_per_step", "attempts_per_pass", "flagged_fraction") if k in line}
        d["bench_line_under_pmc"]["per_gpu_key"] = line["config"]["per_gpu_key"]
        d["bench_line_under_pmc"]["kernel_ms"] = line["roofline"]["kernel_ms"]
        c = counters
        der = {"lane_attempts": att, "wave_attempts_lower_bound": wa}
        if "SQ_INSTS_VALU" in c:
            der["valu_insts_per_wave_attempt"] = c["SQ_INSTS_VALU"] / wa
        if "SQ_INSTS_SALU" in c:
            der["salu_insts_per_wave_attempt"] = c["SQ_INSTS_SALU"] / wa
        if "SQ_INSTS_BRANCH" in c:
            der["branches_per_wave_attempt"] = c["SQ_INSTS_BRANCH"] / wa
        if "SQ_INSTS_VMEM_RD" in c:
            der["vmem_loads_per_wave_attempt"] = c["SQ_INSTS_VMEM_RD"] / wa
        if "SQ_INSTS_VMEM_WR" in c:
            der["vmem_stores_per_wave_attempt"] = c["SQ_INSTS_VMEM_WR"] / wa
        if "GRBM_GUI_ACTIVE" in c:
            cyc = c["GRBM_GUI_ACTIVE"] / 8.0
            der["shader_cycles_per_launch"] = cyc
            kms = line["roofline"]["kernel_ms"]
            der["effective_clock_GHz_vs_bench_kernel_ms"] = cyc / (kms * 1e-3) / 1e9
            if "SQ_ACTIVE_INST_VALU" in c:
                der["issue_busy"] = c["SQ_ACTIVE_INST_VALU"] * 4.0 / (1024.0 * cyc)
            if "SQ_INSTS_VALU" in c:
                der["valu_issue_slots_at_4_cycles"] = c["SQ_INSTS_VALU"] * 4.0 / (1024.0 * cyc)
        if "SQ_THREAD_CYCLES_VALU" in c and "SQ_ACTIVE_INST_VALU" in c:
            der["active_lanes_per_valu_inst"] = c["SQ_THREAD_CYCLES_VALU"] / c["SQ_ACTIVE_INST_VALU"]
        if "SQ_WAVE_CYCLES" in c:
            wc = c["SQ_WAVE_CYCLES"]
            split = {}
            if "SQ_ACTIVE_INST_ANY" in c:
                split["issuing"] = c["SQ_ACTIVE_INST_ANY"] / wc
            if "SQ_WAIT_INST_ANY" in c:
                split["issue_stalled"] = c["SQ_WAIT_INST_ANY"] / wc
            if "SQ_WAIT_ANY" in c:
                split["waiting"] = c["SQ_WAIT_ANY"] / wc
            der["wave_time_split"] = split
        if "SQ_INSTS_VALU" in c:
            der["flop_equivalents_note"] = ("606 algorithmic flop per attempt (SURVEY 8d) vs the VALU instructions actually executed per "
                                            "wave-attempt above: an FMA counts 2 flop and 1 instruction")
        d["derived"] = der
    out["workloads"][w] = d
os.makedirs(os.path.join(root, "profiles"), exist_ok=True)
dst = os.path.join(root, "profiles", f"{tag}_pmc_issue.json")
json.dump(out, open(dst, "w"), indent=1)
print(json.dumps({w: d.get("derived") for w, d in out["workloads"].items()}, indent=1))
