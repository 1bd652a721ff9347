# Phase stamps / timing experiments of wino32.hip.  Build the diagnostic libraries HERE first (compile-time switches):
#   bash tools/build_diag32.sh "0 1 2 8 16 32"      -> lass_amd/csrc/liblass_hip_diag_e<N>.so
# then on the GPU box:  EXPS="0 1 2 8 16 32" ONLY=encoder_block1,decoder_block6 bash tools/gpu_diag32.sh TAG
#   W32_EXP bits (results are WRONG when set, timing only): 1 no patch loads, 2 no epilogue, 8 no MFMA, 16 no patch transform,
#   32 one wave per SIMD
TAG=${1:-diag32}; O=$GRAFT_REPO_ROOT/gpurun_out/$TAG; mkdir -p $O; cd $GRAFT_REPO_ROOT
for E in ${EXPS:-0}; do
echo "== W32_EXP=$E" | tee -a $O/diag.log
LASS_HIP_LIB=$GRAFT_REPO_ROOT/lass_amd/csrc/liblass_hip_diag_e$E.so timeout -k 10 120 python tools/conv_bench.py --iters 2 --only ${ONLY:-encoder_block1,decoder_block6} 2>&1 | grep -E "wino32-diag|ms" | grep -v "\.up" | sed 's/\[wino32-diag\] //; s/1024x512 B=16 grid=256 | cycles per strip and wave: //; s/(strips per wave [0-9.]*) | //' | awk '/flags=/{k=$2" "$3" "$4; if(seen[k]++) next} {print}' | tee -a $O/diag.log
done
