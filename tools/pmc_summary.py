#!/usr/bin/env python3
"""Summarise the rocprofv3 passes of one workload into profiles/r02_pmc_<workload>.json (read by bench.py for
`roofline.traffic` and `roofline.pmc`) and profiles/r02_<workload>_kernel_stats.csv.

Passes (separate runs, as MI355X_MICROARCH.md prescribes: FETCH_SIZE and WRITE_SIZE do not fit one pass; PMC passes carry
--kernel-trace only), each `rocprofv3 ... --output-format csv -d <dir> -- python3 bench.py --workload W --steps 2 --warmup 1 --cpu-sample 0`:
    <dir>/stats    --kernel-trace --stats
    <dir>/fetch    --pmc FETCH_SIZE --kernel-trace
    <dir>/write    --pmc WRITE_SIZE --kernel-trace
    <dir>/sq       --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES GRBM_GUI_ACTIVE --kernel-trace

HBM bytes per launch = (2 * FETCH_SIZE + WRITE_SIZE) * 1024: both counters are in KiB, and on gfx950 FETCH_SIZE tallies the
128-byte requests of wide coalesced reads at 64 bytes (the guide's correction; other access widths are uncalibrated).
Clock = GRBM_GUI_ACTIVE / 8 XCDs / kernel duration.  SQ_ACTIVE_INST_VALU counts quad-cycles summed over the chip's 1024
SIMDs: valu_busy_frac = 4 * SQ_ACTIVE_INST_VALU / (1024 * duration * clock).

Usage: pmc_summary.py <dir> <workload> [cells-per-launch of the dominant kernel, optional: kernel=cells ...]"""
import csv
import glob
import json
import os
import sys
from collections import defaultdict


def find(d, suffix):
    hits = sorted(glob.glob(os.path.join(d, "**", "*" + suffix), recursive=True), key=os.path.getmtime)   # (gpurun merges: older runs' files stay)
    return hits[-1] if hits else None


def counters(d):
    path = find(d, "_counter_collection.csv")
    tot, n = defaultdict(lambda: defaultdict(float)), defaultdict(lambda: defaultdict(int))
    if path:
        for row in csv.DictReader(open(path)):
            tot[row["Kernel_Name"]][row["Counter_Name"]] += float(row["Counter_Value"])
            n[row["Kernel_Name"]][row["Counter_Name"]] += 1
    return {k: {c: tot[k][c] / n[k][c] for c in tot[k]} for k in tot}, {k: max(n[k].values()) for k in n}


def durations(d):
    path = find(d, "_kernel_trace.csv")
    t = defaultdict(list)
    if path:
        for row in csv.DictReader(open(path)):
            t[row["Kernel_Name"]].append((int(row["End_Timestamp"]) - int(row["Start_Timestamp"])) * 1e-6)
    return t


def main():
    d, workload = sys.argv[1], sys.argv[2]
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    cells = dict(kv.split("=") for kv in sys.argv[3:])
    fetch, nf = counters(os.path.join(d, "fetch"))
    write, _ = counters(os.path.join(d, "write"))
    sq, _ = counters(os.path.join(d, "sq"))
    sq_ms = durations(os.path.join(d, "sq"))
    st_ms = durations(os.path.join(d, "stats"))
    rows = []
    for k in sorted(set(fetch) | set(write) | set(sq) | set(st_ms)):
        fk, wk = fetch.get(k, {}).get("FETCH_SIZE", 0.0), write.get(k, {}).get("WRITE_SIZE", 0.0)
        r = {"kernel": k, "launches": len(st_ms.get(k, [])) or nf.get(k, 0), "fetch_kib_raw": fk, "write_kib": wk,
             "hbm_bytes_per_launch": (2 * fk + wk) * 1024}
        if k in st_ms:
            ms = st_ms[k]
            r["ms_avg"], r["ms_min"], r["ms_max"] = sum(ms) / len(ms), min(ms), max(ms)
            r["hbm_GBs"] = r["hbm_bytes_per_launch"] / (r["ms_avg"] * 1e-3) / 1e9
        if k in sq:
            s = sq[k]
            r.update({c: s[c] for c in s})
            ms = sum(sq_ms[k]) / len(sq_ms[k])
            r["ms_avg_sq_pass"] = ms
            if "GRBM_GUI_ACTIVE" in s and ms > 0:
                r["clock_GHz"] = s["GRBM_GUI_ACTIVE"] / 8 / (ms * 1e-3) / 1e9
                if "SQ_ACTIVE_INST_VALU" in s:
                    r["valu_busy_frac"] = 4 * s["SQ_ACTIVE_INST_VALU"] / (1024 * ms * 1e-3 * r["clock_GHz"] * 1e9)
        for pat, c in cells.items():
            if pat in k.replace(" ", "") and "SQ_INSTS_VALU" in r:
                r["cells_per_launch"] = float(c)
                r["valu_insts_per_cell"] = r["SQ_INSTS_VALU"] * 64 / float(c)
        rows.append(r)
    rows.sort(key=lambda r: -(r.get("ms_avg", 0) * max(r["launches"], 1)))
    for r in rows[:14]:
        print("%-64s n %3d  %8.3f ms  HBM %8.3f GB/launch %7.1f GB/s  VALU busy %5s  clock %5s" % (
            r["kernel"][:64], r["launches"], r.get("ms_avg", 0), r["hbm_bytes_per_launch"] / 1e9, r.get("hbm_GBs", 0),
            "%.2f" % r["valu_busy_frac"] if "valu_busy_frac" in r else "-", "%.2f" % r["clock_GHz"] if "clock_GHz" in r else "-"))
    out = {"workload": workload, "note": __doc__.split("Usage")[0].strip().split("\n\n", 1)[1], "kernels": rows[:24]}
    json.dump(out, open(os.path.join(root, "profiles", "r02_pmc_%s.json" % workload), "w"), indent=1)
    stats = find(os.path.join(d, "stats"), "_kernel_stats.csv")
    if stats:
        open(os.path.join(root, "profiles", "r02_%s_kernel_stats.csv" % workload), "w").write(open(stats).read())


if __name__ == "__main__":
    main()
