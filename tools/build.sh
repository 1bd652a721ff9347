#!/bin/bash
# Build liblass_hip.so (and optionally the diagnostic variant / ISA) from anywhere.  Usage: tools/build.sh [diag] [asm FILE]
set -e
cd "$(dirname "$0")/../lass_amd/csrc"
SRC="api.hip conv.hip wino.hip wino32.hip wino4.hip conv_bf16.hip conv_bf16_fused.hip stft.hip misc.hip"
hipcc -O3 -std=c++17 --offload-arch=gfx950 -fPIC -fno-slp-vectorize -shared -o liblass_hip.so $SRC -Rpass-analysis=kernel-resource-usage 2> build.log || { grep -E "error" build.log | head -20; exit 1; }
python3 - <<'PY'
import sys; sys.path.insert(0, "../..")
import __graft_entry__ as g
open(g.STAMP, "w").write(g._src_hash())   # keep build()'s staleness stamp in step with this manual build
PY
if [ "$1" = "diag" ]; then hipcc -O3 -std=c++17 --offload-arch=gfx950 -fPIC -fno-slp-vectorize -shared -DLASS_CONV_DIAG -o liblass_hip_diag.so $SRC 2>&1 | grep -E "error" || true; fi
python3 - <<'PY'
import re
txt=open('build.log').read()
for b in re.split(r'remark: Function Name: ', txt)[1:]:
    name=b.split()[0]
    m=re.search(r'(conv_bf16_kernel|wino_kernel|conv_kernel_sb|conv_kernel_db)I(.*?)EEv', name)
    if not m or ("wino" not in m.group(1) and "bf16" not in m.group(1)): continue
    g=lambda k: re.search(k+r': (\d+)', b).group(1)
    print(m.group(1), re.findall(r'Li(\d+)E', name), 'VGPR',g('VGPRs'),'AGPR',g('AGPRs'),'occ',g(r'Occupancy \[waves/SIMD\]'),'spill',g('VGPRs Spill'),'scratch',g(r'ScratchSize \[bytes/lane\]'),'LDS',g(r'LDS Size \[bytes/block\]'))
PY
