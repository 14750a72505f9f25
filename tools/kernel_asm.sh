#!/bin/bash
# Device assembly + register / instruction-mix summary for the fill kernels (developer aid).
# usage: tools/kernel_asm.sh [source.hip] [extra hipcc flags...]   -> gpurun_out/scratch/<name>.s
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
SRC=${1:-qf_kernels.hip}; shift || true
OUT=$ROOT/gpurun_out/scratch; mkdir -p "$OUT"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off --offload-device-only -S "$@" \
  -o "$OUT/${SRC%.hip}.s" "$ROOT/quaff_amd/csrc/$SRC" 2>&1 | grep -v "hip-link" || true
python3 - "$OUT/${SRC%.hip}.s" <<'PY'
import re, sys, collections
s = open(sys.argv[1]).read()
for m in re.finditer(r'\.name:\s+(\S+)\n(?:.*\n)*?\s+\.sgpr_count:\s+(\d+)(?:.*\n)*?\s+\.vgpr_count:\s+(\d+)\n\s+\.vgpr_spill_count:\s+(\d+)', s):
    name, sg, vg, sp = m.groups()
    i = s.find(name + ':'); j = s.find('.Lfunc_end', i)
    c = collections.Counter()
    for l in s[i:j].split('\n'):
        l = l.strip()
        if not l or l[0] in ';.': continue
        op = l.split()[0]
        c['v' if op.startswith('v_') else 's' if op.startswith('s_') else 'm'] += 1
    scr = s[i:j].count('scratch_')
    print(f"{vg:>4} vgpr {sp:>3} spill {scr:>3} scratch-ops  valu {c['v']:>5} salu {c['s']:>4} mem {c['m']:>3}  {name[:90]}")
PY
