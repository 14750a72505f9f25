#!/bin/bash
# The five rocprofv3 passes of one workload on the GPU box (run through gpurun from the repo root):
#   tools/profile.sh align|train|overlap|fulldp [extra bench.py flags]
# Output under gpurun_out/prof_<workload>/{stats,fetch,write,sq,stall}; summarise afterwards (here or in the container) with
#   python3 tools/pmc_summary.py gpurun_out/prof_<workload> <workload> [kernel=cells ...]      -> profiles/r<NN>_*
# The profiler's command line starts the program itself (python3 bench.py ...); PMC passes carry --kernel-trace only;
# FETCH_SIZE and WRITE_SIZE in separate passes (MI355X_MICROARCH.md).
set -e
W=$1; shift || true
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/prof_$W
rm -rf "$OUT"; mkdir -p "$OUT"
cd /tmp; export TMPDIR=/tmp
ARGS="$ROOT/bench.py --workload $W --steps 2 --warmup 1 --cpu-sample 0 $*"
pass() {  # name, rocprofv3 flags...
  local name=$1; shift
  timeout -k 10 500 rocprofv3 "$@" --output-format csv -d "$OUT/$name" -- python3 $ARGS > "$OUT/$name.log" 2>&1
  echo "$name done"
}
pass stats --kernel-trace --stats
pass fetch --pmc FETCH_SIZE --kernel-trace
pass write --pmc WRITE_SIZE --kernel-trace
pass sq --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES GRBM_GUI_ACTIVE --kernel-trace
pass stall --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_WAIT_INST_LDS SQ_INSTS_LDS --kernel-trace
grep -h '^{' "$OUT/stats.log" | tail -1 > "$OUT/bench_line.json" || true
