#!/usr/bin/env python3
"""Summarise the rocprofv3 passes of one workload (tools/profile.sh) into profiles/r<NN>_pmc_<workload>.json (replayed, labelled, by
bench.py as `roofline.traffic` / `roofline.replayed_counters` while `source_hash` still matches the device code) and
profiles/r<NN>_<workload>_kernel_stats.csv.  ROUND=r04 (default) names the files.

Passes (separate runs, as MI355X_MICROARCH.md prescribes: FETCH_SIZE and WRITE_SIZE do not fit one pass; PMC passes carry
--kernel-trace only), each `rocprofv3 ... --output-format csv -d <dir> -- python3 bench.py --workload W --steps 2 --warmup 1 --cpu-sample 0`:
    <dir>/stats    --kernel-trace --stats
    <dir>/fetch    --pmc FETCH_SIZE --kernel-trace
    <dir>/write    --pmc WRITE_SIZE --kernel-trace
    <dir>/sq       --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES GRBM_GUI_ACTIVE --kernel-trace
    <dir>/stall    --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_WAIT_INST_LDS SQ_INSTS_LDS --kernel-trace

HBM bytes per launch = (2 * FETCH_SIZE + WRITE_SIZE) * 1024: both counters are in KiB, and on gfx950 FETCH_SIZE tallies the
128-byte requests of wide coalesced reads at 64 bytes (the guide's correction; other access widths are uncalibrated).
Clock = GRBM_GUI_ACTIVE / 8 XCDs / kernel duration.  SQ_ACTIVE_INST_VALU counts quad-cycles summed over the chip's 1024
SIMDs: valu_busy_frac = 4 * SQ_ACTIVE_INST_VALU / (1024 * duration * clock).  Where a kernel's wave-cycles go (stall pass):
wave_cycles_parked_frac = SQ_WAIT_ANY / SQ_WAVE_CYCLES (waiting on a counter: memory, LDS, export), wave_cycles_issue_stalled_frac =
SQ_WAIT_INST_ANY / SQ_WAVE_CYCLES (has an instruction, cannot issue it), wave_cycles_lds_stalled_frac = SQ_WAIT_INST_LDS / SQ_WAVE_CYCLES,
wave_cycles_issuing_frac = SQ_ACTIVE_INST_ANY / SQ_WAVE_CYCLES.

Usage: pmc_summary.py <dir> <workload> [cells-per-launch of the dominant kernel, optional: kernel=cells ...]"""
import csv
import glob
import json
import os
import sys
from collections import defaultdict

PROFILED_STEPS = 3   # tools/profile.sh: --steps 2 --warmup 1


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
    stall, _ = counters(os.path.join(d, "stall"))
    sq_ms = durations(os.path.join(d, "sq"))
    st_ms = durations(os.path.join(d, "stats"))
    rows = []
    for k in sorted(set(fetch) | set(write) | set(sq) | set(st_ms) | set(stall)):
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
        if k in stall and stall[k].get("SQ_WAVE_CYCLES"):
            w = stall[k]["SQ_WAVE_CYCLES"]
            r["stall_pass"] = {c: stall[k][c] for c in stall[k]}
            r["wave_cycles_parked_frac"] = stall[k].get("SQ_WAIT_ANY", 0.0) / w
            r["wave_cycles_issue_stalled_frac"] = stall[k].get("SQ_WAIT_INST_ANY", 0.0) / w
            r["wave_cycles_lds_stalled_frac"] = stall[k].get("SQ_WAIT_INST_LDS", 0.0) / w
            r["wave_cycles_issuing_frac"] = stall[k].get("SQ_ACTIVE_INST_ANY", 0.0) / w
        # train: the cells given are a fill class's per E-step; a class may run as several launches of its kernel per step (the
        # dominant class's Backward is two: DESIGN.md 4), so per-pass figures = per-launch averages x launches per step
        lpp = 1
        if workload == "train" and r["launches"] and r["launches"] % PROFILED_STEPS == 0:
            lpp = max(1, r["launches"] // PROFILED_STEPS)
        if lpp > 1:
            r["launches_per_pass"] = lpp
            r["hbm_bytes_per_pass"] = r["hbm_bytes_per_launch"] * lpp
        for pat, c in cells.items():
            if pat in k.replace(" ", "") and "SQ_INSTS_VALU" in r:
                r["cells_per_launch"] = float(c)
                r["valu_insts_per_cell"] = r["SQ_INSTS_VALU"] * 64 * lpp / float(c)
        rows.append(r)
    rows.sort(key=lambda r: -(r.get("ms_avg", 0) * max(r["launches"], 1)))
    for r in rows[:14]:
        print("%-64s n %3d  %8.3f ms  HBM %8.3f GB/launch %7.1f GB/s  VALU busy %5s  clock %5s" % (
            r["kernel"][:64], r["launches"], r.get("ms_avg", 0), r["hbm_bytes_per_launch"] / 1e9, r.get("hbm_GBs", 0),
            "%.2f" % r["valu_busy_frac"] if "valu_busy_frac" in r else "-", "%.2f" % r["clock_GHz"] if "clock_GHz" in r else "-"))
    sys.path.insert(0, root)
    from quaff_amd import api
    import subprocess
    rnd = os.environ.get("ROUND", "r04")
    try:
        head = subprocess.check_output(["git", "-C", root, "rev-parse", "--short", "HEAD"], text=True).strip()
    except Exception:
        head = None
    bench_line = None
    bl = os.path.join(d, "bench_line.json")
    if os.path.exists(bl) and os.path.getsize(bl):
        bench_line = json.load(open(bl))
    out = {"workload": workload, "source_hash": api.kernel_source_hash(), "git_head": head,
           "command": "tools/profile.sh %s (+ the flags in bench_line.config)" % workload,
           "note": __doc__.split("Usage")[0].strip().split("\n\n", 1)[1], "kernels": rows[:24]}
    if bench_line:
        out["bench_line_of_the_stats_pass"] = {"value": bench_line.get("value"), "ms_per_step": bench_line.get("ms_per_step"),
                                               "workload": bench_line.get("config", {}).get("workload")}
    json.dump(out, open(os.path.join(root, "profiles", "%s_pmc_%s.json" % (rnd, workload)), "w"), indent=1)
    stats = find(os.path.join(d, "stats"), "_kernel_stats.csv")
    if stats:
        open(os.path.join(root, "profiles", "%s_%s_kernel_stats.csv" % (rnd, workload)), "w").write(open(stats).read())


if __name__ == "__main__":
    main()
