#!/bin/bash
# rocprofv3 kernel stats of one bench workload: tools/dev/stats.sh <workload> [bench flags] -> gpurun_out/stats_<workload>.txt
W=$1; shift || true
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
OUT=$ROOT/gpurun_out/stats_$W
rm -rf "$OUT"; mkdir -p "$OUT"
cd /tmp; export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT" -- python3 $ROOT/bench.py --workload $W --steps 2 --warmup 1 --cpu-sample 0 "$@" > "$OUT/log" 2>&1
python3 - "$OUT" <<'PY'
import csv, glob, sys
f = sorted(glob.glob(sys.argv[1] + "/**/*_kernel_stats.csv", recursive=True))[-1]
for r in list(csv.DictReader(open(f)))[:14]:
    print("%-70s calls %4s  avg %10.3f us  total %10.3f ms  %5s%%" % (r["Name"][:70], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["TotalDurationNs"]) / 1e6, r["Percentage"]))
PY
