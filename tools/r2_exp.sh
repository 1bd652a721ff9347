TAG=$1; O=$GRAFT_REPO_ROOT/gpurun_out/$TAG; mkdir -p $O; cd $GRAFT_REPO_ROOT
for E in 0 1 2 4 8 3 7 15; do
  echo "== LASS_EXP=$E"
  LASS_WINO8=0 LASS_EXP=$E timeout -k 10 120 python tools/conv_bench.py --iters 5 --only encoder_block1,encoder_block3,decoder_block3,decoder_block5,decoder_block6 2>/dev/null | grep -v "\.up" | tee -a $O/exp.log || exit 1
done
