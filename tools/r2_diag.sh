TAG=$1; O=$GRAFT_REPO_ROOT/gpurun_out/$TAG; mkdir -p $O; cd $GRAFT_REPO_ROOT
true
for E in ${EXPS:-0}; do
echo "== LASS_EXP=$E"
LASS_EXP=$E LASS_HIP_LIB=$GRAFT_REPO_ROOT/lass_amd/csrc/liblass_hip_diag.so timeout -k 10 120 python tools/conv_bench.py --iters 1 --only ${ONLY:-encoder_block1,encoder_block3,decoder_block3,decoder_block5} 2>&1 | grep -E "wino-diag|ms" | grep -v "\.up" | tee -a $O/diag.log
done
