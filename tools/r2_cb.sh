TAG=$1; O=$GRAFT_REPO_ROOT/gpurun_out/$TAG; mkdir -p $O; cd $GRAFT_REPO_ROOT
LASS_COMPUTE=${MODE:-bf16} timeout -k 10 300 python tools/conv_bench.py --iters 7 ${CB_ARGS} 2>/dev/null | tee $O/cb_${MODE:-bf16}.log
