#!/usr/bin/env python3
"""Developer aid: per-kernel averages of one rocprofv3 --pmc pass (any counters) with the kernel durations beside them.
Usage: tools/dev/pmc_quick.py <rocprofv3 output dir> [kernel-name filter]"""
import csv, glob, os, sys
from collections import defaultdict
d = sys.argv[1]; pat = sys.argv[2] if len(sys.argv) > 2 else ""
def find(suffix):
    hits = sorted(glob.glob(os.path.join(d, "**", "*" + suffix), recursive=True), key=os.path.getmtime)
    return hits[-1] if hits else None
tot, n, dur = defaultdict(lambda: defaultdict(float)), defaultdict(lambda: defaultdict(int)), defaultdict(list)
p = find("_counter_collection.csv")
if p:
    for r in csv.DictReader(open(p)):
        tot[r["Kernel_Name"]][r["Counter_Name"]] += float(r["Counter_Value"]); n[r["Kernel_Name"]][r["Counter_Name"]] += 1
p = find("_kernel_trace.csv")
if p:
    for r in csv.DictReader(open(p)):
        dur[r["Kernel_Name"]].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-6)
for k in sorted(set(tot) | set(dur), key=lambda k: -sum(dur.get(k, [0]))):
    if pat not in k: continue
    ms = dur.get(k, [])
    print("%s\n   launches %d  avg %.3f ms  total %.1f ms" % (k[:110], len(ms), sum(ms) / max(1, len(ms)), sum(ms)))
    for c in sorted(tot[k]): print("   %-28s %.4g" % (c, tot[k][c] / n[k][c]))
