#!/usr/bin/env python3
"""Prints the last step of the trace tools/timeline_r02.sh collected: start (us), duration, idle gap before it, kernel."""
import csv
import os
import sys

d = sys.argv[1] if len(sys.argv) > 1 else os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out", "prof_tl")
rows = list(csv.DictReader(open(os.path.join(d, "kernel_trace.csv"))))
ev = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"][:56]) for r in rows]
mc = list(csv.DictReader(open(os.path.join(d, "memcpy_trace.csv"))))
ev += [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "MEMCPY " + r["Direction"][12:]) for r in mc]
ev.sort()
i0 = [i for i, e in enumerate(ev) if "k_pack_reads" in e[2]][-1]
t0 = prev_end = ev[i0][0]
gaps = 0.0
for s, e, n in ev[i0:]:
    gap = (s - prev_end) / 1000
    gaps += max(gap, 0.0)
    print("%9.1f us  dur %8.1f  gap %7.1f  %s" % ((s - t0) / 1000, (e - s) / 1000, gap, n))
    prev_end = max(prev_end, e)
print("idle between the first and the last event: %.1f us" % gaps)
