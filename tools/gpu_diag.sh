# Phase-stamp diagnostics of the Winograd kernels (GPU box).  Build the diagnostic library first, here or there:
#   cd lass_amd/csrc && hipcc -O3 -std=c++17 --offload-arch=gfx950 -fPIC -fno-slp-vectorize -shared -DLASS_CONV_DIAG -o liblass_hip_diag.so *.hip
# usage: EXPS="0 7 8 15 16" ONLY=encoder_block3,decoder_block3 bash tools/gpu_diag.sh TAG
#   LASS_EXP bits (results are WRONG when set, timing only): 1 no weight DMA, 2 no input transform, 4 no raw staging,
#   8 no MFMA, 16 / 32: +16 / +32 dummy VALU instructions per chunk
TAG=${1:-diag}; O=$GRAFT_REPO_ROOT/gpurun_out/$TAG; mkdir -p $O; cd $GRAFT_REPO_ROOT
for E in ${EXPS:-0}; do
echo "== LASS_EXP=$E"
LASS_EXP=$E LASS_HIP_LIB=$GRAFT_REPO_ROOT/lass_amd/csrc/liblass_hip_diag.so timeout -k 10 120 python tools/conv_bench.py --iters 1 --only ${ONLY:-encoder_block1,encoder_block3,decoder_block3,decoder_block5} 2>&1 | grep -E "wino-diag|ms" | grep -v "\.up" | tee -a $O/diag.log
done
