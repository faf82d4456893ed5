#!/usr/bin/env python3
"""Diagnostic: durations and inter-kernel gaps from a rocprofv3 --kernel-trace CSV."""
import csv, glob, sys
files = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)
if not files:
    sys.exit("no kernel_trace.csv under " + sys.argv[1])
rows = list(csv.DictReader(open(files[0])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
prev = None
for r in rows[-30:]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    gap = (s - prev) / 1e3 if prev else 0.0
    print("%-44s dur %8.1f us  gap-before %6.1f us" % (r["Kernel_Name"][:44], (e - s) / 1e3, gap))
    prev = e
