#!/bin/bash
# Developer A/B aid: build a copy of libquaffhip.so with extra -D flags for ONE kernel file, under build/variants/ (git-ignored,
# travels with gpurun), and run any program against it with QUAFF_HIP_LIBRARY=build/variants/libquaffhip_<name>.so.
#   tools/dev/variant.sh <name> <file.hip> [-DMACRO=... ...]
set -e
NAME=$1; FILE=$2; shift 2
ROOT=$(cd "$(dirname "$0")/../.." && pwd)
SRC=$ROOT/quaff_amd/csrc
OUT=$ROOT/build/variants
mkdir -p "$OUT"
EXTRA=""
[ "$FILE" = qf_fb.hip ] && EXTRA="-munsafe-fp-atomics"
[ "$FILE" = qf_sort.hip ] && EXTRA="-Wno-deprecated-declarations"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -Wall -Wno-unused-result $EXTRA "$@" -c -o "$OUT/${FILE%.hip}_$NAME.o" "$SRC/$FILE" 2>&1 | grep -v "hip-link" || true
OBJS=""
for f in qf_api qf_kernels qf_fb qf_overlap qf_sort qf_model; do
  if [ "$f.hip" = "$FILE" ]; then OBJS="$OBJS $OUT/${f}_$NAME.o"; else OBJS="$OBJS $SRC/$f.o"; fi
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -o "$OUT/libquaffhip_$NAME.so" $OBJS -ldl 2>&1 | grep -v "hip-link" || true
ls -la "$OUT/libquaffhip_$NAME.so"
