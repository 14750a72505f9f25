#!/usr/bin/env python3
"""Developer aid: what the GPU was doing over a rocprofv3 --kernel-trace run.  Takes the timed tail of the trace (after the
last gap longer than --split ms, i.e. bench.py's timed step), and prints the wall time covered by >= 1 kernel, by >= 2, idle time,
and per kernel: launches, total and average duration, and the time it ran ALONE.
Usage: tools/dev/timeline.py <rocprofv3 output dir> [--last-ms N]"""
import csv, glob, os, sys
from collections import defaultdict
d = sys.argv[1]
last_ms = float(sys.argv[sys.argv.index("--last-ms") + 1]) if "--last-ms" in sys.argv else None
p = sorted(glob.glob(os.path.join(d, "**", "*_kernel_trace.csv"), recursive=True), key=os.path.getmtime)[-1]
ev = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in csv.DictReader(open(p))]
ev.sort()
t_end = max(e[1] for e in ev)
if last_ms: ev = [e for e in ev if e[0] >= t_end - last_ms * 1e6]
t0 = ev[0][0]
pts = []
for i, (s, e, k) in enumerate(ev): pts += [(s, 1, i), (e, -1, i)]
pts.sort()
live, prev = set(), t0
cover = defaultdict(float); alone = defaultdict(float)
for t, kind, i in pts:
    dt = (t - prev) * 1e-6
    cover[min(len(live), 3)] += dt
    if len(live) == 1: alone[ev[next(iter(live))][2]] += dt
    prev = t
    if kind == 1: live.add(i)
    else: live.discard(i)
tot = (t_end - t0) * 1e-6
print("window %.1f ms: idle %.1f, one kernel %.1f, two %.1f, three+ %.1f" % (tot, cover[0], cover[1], cover[2], cover[3]))
dur = defaultdict(list)
for s, e, k in ev: dur[k].append((e - s) * 1e-6)
for k in sorted(dur, key=lambda k: -sum(dur[k])):
    print("%8.1f ms total %6d x %8.3f ms avg  alone %8.1f  %s" % (sum(dur[k]), len(dur[k]), sum(dur[k]) / len(dur[k]), alone[k], k[:90]))
