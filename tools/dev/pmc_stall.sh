#!/bin/bash
# One rocprofv3 counter pass that says where a kernel's wave-cycles go (issue vs parked vs LDS): tools/dev/pmc_stall.sh <workload> [bench flags]
set -e
W=$1; shift || true
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
OUT=$ROOT/gpurun_out/stall_$W
rm -rf "$OUT"; mkdir -p "$OUT"
cd /tmp; export TMPDIR=/tmp
timeout -k 10 500 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_WAIT_INST_LDS SQ_INSTS_VALU SQ_INSTS_LDS --kernel-trace --output-format csv -d "$OUT/p" -- python3 $ROOT/bench.py --workload $W --steps 1 --warmup 1 --cpu-sample 0 "$@" > "$OUT/log" 2>&1
python3 - "$OUT" <<'PY'
import csv, glob, sys, collections
d = sys.argv[1]
f = sorted(glob.glob(d + "/p/**/*_counter_collection.csv", recursive=True))[-1]
tot = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
for r in csv.DictReader(open(f)):
    tot[r["Kernel_Name"]][r["Counter_Name"]] += float(r["Counter_Value"])
for k, c in sorted(tot.items(), key=lambda kv: -kv[1].get("SQ_WAVE_CYCLES", 0))[:8]:
    w = c.get("SQ_WAVE_CYCLES", 1) or 1
    print("%-60s wave-cyc %.3g  parked %.2f  issue-stall %.2f  active %.2f  valu %.2f  lds-issue-stall %.2f | insts valu %.4g lds %.4g" % (
        k[:60], w, c["SQ_WAIT_ANY"] / w, c["SQ_WAIT_INST_ANY"] / w, c["SQ_ACTIVE_INST_ANY"] / w, c["SQ_ACTIVE_INST_VALU"] / w,
        c["SQ_WAIT_INST_LDS"] / w, c["SQ_INSTS_VALU"], c["SQ_INSTS_LDS"]))
PY
