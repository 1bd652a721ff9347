# A/B of two library builds on the same box: usage: bash tools/gpu_ab.sh TAG libA.so libB.so
TAG=$1; O=$GRAFT_REPO_ROOT/gpurun_out/$TAG; mkdir -p $O; cd $GRAFT_REPO_ROOT
for L in $2 $3; do
  echo "== $L"
  LASS_HIP_LIB=$GRAFT_REPO_ROOT/lass_amd/csrc/$L timeout -k 10 300 python tools/conv_bench.py --iters 7 ${CB_ARGS} 2>/dev/null | tee $O/cb_$L.log || exit 1
done
