#!/usr/bin/env python3
"""Histogram of rocprofv3 PC samples per instruction (and per stall reason where the method is stochastic).
usage: tools/pcsample_hist.py gpurun_out/pcs_<tag>   -> writes hist.txt there (development aid)"""
import collections
import csv
import glob
import os
import sys

out = sys.argv[1]
files = glob.glob(os.path.join(out, "raw", "**", "*pc_sampling*.csv"), recursive=True)
lines = []
for f in files:
    rows = list(csv.DictReader(open(f)))
    lines.append("%s: %d samples, columns %s" % (os.path.relpath(f, out), len(rows), list(rows[0].keys()) if rows else []))
    if not rows:
        continue
    key_i = "Instruction" if "Instruction" in rows[0] else None
    key_c = "Instruction_Comment" if "Instruction_Comment" in rows[0] else None
    by = collections.Counter()
    reason = collections.defaultdict(collections.Counter)
    for r in rows:
        k = (r.get(key_c, "") if key_c else "", r.get(key_i, "") if key_i else "")
        by[k] += 1
        for col in ("Stall_Reason", "Wave_Issued", "Instruction_Type"):
            if col in r:
                reason[col][r[col]] += 1
    total = sum(by.values())
    for col, c in reason.items():
        lines.append("%s: %s" % (col, ", ".join("%s=%.1f%%" % (k, 100.0 * v / total) for k, v in c.most_common(12))))
    for (cm, ins), n in by.most_common(150):
        lines.append("%6.2f%%  %-70s %s" % (100.0 * n / total, ins[:70], cm[-60:]))
open(os.path.join(out, "hist.txt"), "w").write("\n".join(lines) + "\n")
print("\n".join(lines[:40]))
