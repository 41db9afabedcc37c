#!/usr/bin/env python3
"""Development probe: bench.py's driver_e2e section by itself, three times (tools/rk45_solver end to end on the
headline workload; phase timers of the driver).   python tools/e2e_probe.py"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.argv = [sys.argv[0]]
import bench  # noqa: E402

bench._load()
args = bench.parse()
for _ in range(3):
    r = bench.driver_e2e_section(args)
    print(json.dumps({k: r.get(k) for k in ("process_wall_s", "phases_s", "error")}))
