#!/bin/bash
# Diagnostic builds of wino32.hip with compile-time timing experiments: bash tools/build_diag32.sh "0 1 2 8" [extra -D flags]
set -e
cd "$(dirname "$0")/../lass_amd/csrc"
bash ../../tools/quick_build.sh > /dev/null
mkdir -p .obj/diag
[ -f .obj/diag/api.o ] && [ .obj/diag/api.o -nt api.hip ] || hipcc -O3 -std=c++17 --offload-arch=gfx950 -fPIC -fno-slp-vectorize -DLASS_CONV_DIAG -c api.hip -o .obj/diag/api.o
for E in $1; do
  ( hipcc -O3 -std=c++17 --offload-arch=gfx950 -fPIC -fno-slp-vectorize -DLASS_CONV_DIAG -DW32_EXP=$E $2 -c wino32.hip -o .obj/diag/wino32_e$E.o &&
    hipcc --offload-arch=gfx950 -shared -o liblass_hip_diag_e$E.so .obj/diag/api.o .obj/diag/wino32_e$E.o .obj/conv.o .obj/wino.o .obj/conv_bf16.o .obj/conv_bf16_fused.o .obj/stft.o .obj/misc.o ) &
done
wait
ls liblass_hip_diag_e*.so
