#!/bin/bash
# Incremental build of liblass_hip.so: one object per translation unit (cached under lass_amd/csrc/.obj, rebuilt when the
# .hip or any header is newer), compiled in parallel, then linked; keeps __graft_entry__'s staleness stamp in step.
set -e
cd "$(dirname "$0")/../lass_amd/csrc"
mkdir -p .obj
SRC="api.hip conv.hip wino.hip wino32.hip wino4.hip conv_bf16.hip conv_bf16_fused.hip stft.hip misc.hip"
NEWEST_H=$(ls -t *.h ../../include/lass_hip.h | head -1)
pids=()
for s in $SRC; do
  o=.obj/${s%.hip}.o
  if [ ! -f $o ] || [ $s -nt $o ] || [ $NEWEST_H -nt $o ]; then
    ( hipcc -O3 -std=c++17 --offload-arch=gfx950 -fPIC -fno-slp-vectorize -c $s -o $o 2> .obj/${s%.hip}.log || { grep -E "error" .obj/${s%.hip}.log | head -20; exit 1; } ) &
    pids+=($!)
  fi
done
for p in "${pids[@]}"; do wait $p; done
OBJS=""; for s in $SRC; do OBJS="$OBJS .obj/${s%.hip}.o"; done
hipcc --offload-arch=gfx950 -shared -o liblass_hip.so $OBJS
if [ "$1" = "diag" ]; then  # diagnostic library (-DLASS_CONV_DIAG in the files named in $2, default "api wino32"): liblass_hip_diag.so
  mkdir -p .obj/diag; DOBJS=""
  for s in $SRC; do
    b=${s%.hip}
    if echo " ${2:-api wino32} " | grep -q " $b "; then
      hipcc -O3 -std=c++17 --offload-arch=gfx950 -fPIC -fno-slp-vectorize -DLASS_CONV_DIAG -c $s -o .obj/diag/$b.o 2> .obj/diag/$b.log || { grep error .obj/diag/$b.log | head; exit 1; }
      DOBJS="$DOBJS .obj/diag/$b.o"
    else DOBJS="$DOBJS .obj/$b.o"; fi
  done
  hipcc --offload-arch=gfx950 -shared -o liblass_hip_diag.so $DOBJS
fi
python3 - <<'PY'
import sys; sys.path.insert(0, "../..")
import __graft_entry__ as g
open(g.STAMP, "w").write(g._src_hash())
PY
echo built
