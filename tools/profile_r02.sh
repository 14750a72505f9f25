#!/bin/bash
# The four rocprofv3 passes of one workload on the GPU box (run through gpurun from the repo root):
#   tools/profile_r02.sh align|train|overlap|fulldp [extra bench.py flags]
# Output under gpurun_out/prof_<workload>/{stats,fetch,write,sq}; summarise afterwards (here or in the container) with
#   python3 tools/pmc_summary.py gpurun_out/prof_<workload> <workload> [kernel=cells ...]
# The profiler's command line starts the program itself (python3 bench.py ...), PMC passes carry --kernel-trace only.
set -e
W=$1; shift || true
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/prof_$W
rm -rf "$OUT"; mkdir -p "$OUT"
cd /tmp; export TMPDIR=/tmp
ARGS="$ROOT/bench.py --workload $W --steps 2 --warmup 1 --cpu-sample 0 $*"
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats" -- python3 $ARGS > "$OUT/stats.log" 2>&1
echo "stats done"
timeout -k 10 400 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d "$OUT/fetch" -- python3 $ARGS > "$OUT/fetch.log" 2>&1
echo "fetch done"
timeout -k 10 400 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d "$OUT/write" -- python3 $ARGS > "$OUT/write.log" 2>&1
echo "write done"
timeout -k 10 400 rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d "$OUT/sq" -- python3 $ARGS > "$OUT/sq.log" 2>&1
echo "sq done"
grep -h '^{' "$OUT/stats.log" | tail -1 > "$OUT/bench_line.json" || true
