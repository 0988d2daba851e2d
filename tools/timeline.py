#!/usr/bin/env python3
"""Prints the last step of a rocprofv3 kernel + memory-copy trace (tools/timeline.sh): start (us), duration, idle gap before
it, kernel; then the step's length, busy time and the largest gaps."""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
ev = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"][:60]) for r in rows]
if len(sys.argv) > 2 and sys.argv[2]:
    try:
        mc = list(csv.DictReader(open(sys.argv[2])))
        ev += [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "MEMCPY " + r["Direction"][12:]) for r in mc]
    except OSError:
        pass
ev.sort()
packs = [i for i, e in enumerate(ev) if "k_pack_reads" in e[2]]
i0 = packs[-2] if len(packs) > 1 else packs[-1]      # the last complete step: from its pack kernel to the next one
i1 = packs[-1] if len(packs) > 1 else len(ev)
t0 = prev_end = ev[i0][0]
gaps = []
for s, e, n in ev[i0:i1]:
    gap = (s - prev_end) / 1000
    if gap > 0:
        gaps.append((gap, (s - t0) / 1000, n))
    print("%9.1f us  dur %8.1f  gap %7.1f  %s" % ((s - t0) / 1000, (e - s) / 1000, gap, n))
    prev_end = max(prev_end, e)
print("step: %.1f us from the first to the last event, idle %.1f us" % ((prev_end - t0) / 1000, sum(g[0] for g in gaps)))
print("largest gaps:")
for g in sorted(gaps, reverse=True)[:15]:
    print("   %7.1f us before %-50s (at %.1f)" % (g[0], g[2], g[1]))
if len(packs) > 1:
    print("step period (pack to pack): %.1f us" % ((ev[packs[-1]][0] - ev[packs[-2]][0]) / 1000))
