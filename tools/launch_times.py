#!/usr/bin/env python3
"""Per-launch durations (ms) of the kernels whose name contains one of the given substrings, from a
rocprofv3 --kernel-trace --output-format csv directory:  python tools/launch_times.py <dir> k_scan k_edge_round"""
import csv
import glob
import sys

f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
for pat in sys.argv[2:]:
    d = [round((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6, 3) for r in rows if pat in r["Kernel_Name"]]
    print(pat, round(sum(d), 3), d)
