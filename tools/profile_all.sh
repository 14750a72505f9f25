#!/bin/bash
# Everything profiles/r<NN>_* is made of, in one go on the GPU box (through gpurun, from the repo root):
#   bash tools/profile_all.sh            # the four workloads' rocprofv3 passes (tools/profile.sh) + summaries + full-size bench lines
# Profiled sizes are cut down where a full step is long (overlap: the first 680 rows = 10 internal blocks; fulldp: 512 reads = two rounds
# of resident workgroups); the bench lines are the full stated configurations.
set -e
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd "$ROOT"
export ROUND=${ROUND:-r04}
summ() {  # workload: kernel=cells pairs from the stats pass's bench line, then the summary
  python3 - "$1" <<'PY' > gpurun_out/prof_$1/cells.txt
import json, sys
w = sys.argv[1]
d = json.load(open("gpurun_out/prof_%s/bench_line.json" % w))
c, r = d["config"], d["roofline"]
out = {}
if w in ("align", "fulldp"):
    for k, v in c.get("fill_kernels", {}).items():
        out[k.replace("qf::", "").replace(" ", "")] = v["cells"]
    out.setdefault(r["kernel"].replace("qf::", "").replace(" ", ""), r["cells_per_launch"])
elif w == "train":
    for g, v in c["kernels"].items():
        out["k_forward_" + g.replace(" ", "")[:-1] + ","] = v["cells"]
        out["k_backward_" + g.replace(" ", "")[:-1] + ","] = v["cells"] * r["cells_per_launch"] / max(1, r["forward_kernel"]["cells_per_launch"])
elif w == "overlap":
    n = max(1, r.get("launches_per_step", 1))
    for k, v in c["fill_kernels"].items():
        out[k.replace("qf::", "").replace(" ", "")] = v["cells"] / n
print(" ".join("%s=%d" % (k, v) for k, v in out.items()))
PY
  python3 tools/pmc_summary.py gpurun_out/prof_$1 $1 $(cat gpurun_out/prof_$1/cells.txt) > gpurun_out/prof_$1/summary.txt
  cp gpurun_out/prof_$1/bench_line.json profiles/${ROUND}_$1_profiled_bench.json
}
# (a gpurun call is at most 20 minutes: `bash tools/profile_all.sh a` = align + train, `... b` = overlap + fulldp, no argument = both)
PART=${1:-ab}
if [[ $PART == *a* ]]; then
bash tools/profile.sh align && summ align
bash tools/profile.sh train && summ train
python3 bench.py > profiles/${ROUND}_align_bench.json 2> gpurun_out/bench_align.err && echo "align line"
python3 bench.py --workload train > profiles/${ROUND}_train_bench.json 2> gpurun_out/bench_train.err && echo "train line"
python3 bench.py --workload train --band 80 > profiles/${ROUND}_train_band80_bench.json 2> gpurun_out/bench_train80.err && echo "train band-80 line"
fi
if [[ $PART == *b* ]]; then
bash tools/profile.sh overlap --overlap-rows 680 --inflight 1 --serial-classes && summ overlap
bash tools/profile.sh fulldp --reads 512 && summ fulldp
python3 bench.py --workload fulldp > profiles/${ROUND}_fulldp_bench.json 2> gpurun_out/bench_fulldp.err && echo "fulldp line"
python3 bench.py --workload overlap > profiles/${ROUND}_overlap_bench.json 2> gpurun_out/bench_overlap.err && echo "overlap line"
python3 bench.py --workload overlap --reads 1500 > profiles/${ROUND}_overlap_dense_bench.json 2> gpurun_out/bench_dense.err && echo "dense line"
fi
echo "profiles done"
mkdir -p gpurun_out/profiles_${ROUND} && cp profiles/${ROUND}_* gpurun_out/profiles_${ROUND}/
