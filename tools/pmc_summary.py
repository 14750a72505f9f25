#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes (separate runs, MI355X_MICROARCH.md "HBM" section) into
per-launch HBM bytes per kernel.  FETCH_SIZE/WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE counts 128-B requests at 64 B
for wide coalesced streams, so the read side is doubled (the guide's correction; other access widths are uncalibrated).
Usage: pmc_summary.py <fetch counter_collection.csv> <write counter_collection.csv> [out.json] [reads] [read_len]"""
import csv
import json
import sys
from collections import defaultdict


def per_kernel(path, counter):
    tot, n = defaultdict(float), defaultdict(int)
    for row in csv.DictReader(open(path)):
        if row["Counter_Name"] == counter:
            tot[row["Kernel_Name"]] += float(row["Counter_Value"])
            n[row["Kernel_Name"]] += 1
    return {k: (tot[k] / n[k], n[k]) for k in tot}


def main():
    f = per_kernel(sys.argv[1], "FETCH_SIZE")
    w = per_kernel(sys.argv[2], "WRITE_SIZE")
    rows = []
    for k in sorted(set(f) | set(w)):
        fk, wk = f.get(k, (0, 0))[0], w.get(k, (0, 0))[0]
        rows.append({"kernel": k, "launches": f.get(k, w.get(k))[1], "fetch_kib_raw": fk, "write_kib": wk,
                     "hbm_bytes_per_launch": (2 * fk + wk) * 1024})
    rows.sort(key=lambda r: -r["hbm_bytes_per_launch"])
    for r in rows[:12]:
        print("%-70s launches %3d  fetch(raw) %12.0f KiB  write %12.0f KiB  HBM/launch %8.3f GB" %
              (r["kernel"][:70], r["launches"], r["fetch_kib_raw"], r["write_kib"], r["hbm_bytes_per_launch"] / 1e9))
    if len(sys.argv) > 3:
        dom = next(r for r in rows if "k_viterbi_fill2<16, 5" in r["kernel"] or "k_viterbi_fill<16, 5" in r["kernel"])
        json.dump({"kernel": "k_viterbi_fill<16,5>", "reads": int(sys.argv[4]), "read_len": int(sys.argv[5]),
                   "hbm_bytes_per_launch": dom["hbm_bytes_per_launch"], "fetch_kib_raw": dom["fetch_kib_raw"],
                   "write_kib": dom["write_kib"],
                   "note": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes; bytes = (2*FETCH_SIZE + WRITE_SIZE)*1024 "
                           "(gfx950 FETCH_SIZE half-count correction; 4-B-per-lane access width uncalibrated)",
                   "kernels": rows[:12]}, open(sys.argv[3], "w"), indent=1)


if __name__ == "__main__":
    main()
