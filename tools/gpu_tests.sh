# full GPU test suite on the GPU box, log to gpurun_out/gputest/
O=$GRAFT_REPO_ROOT/gpurun_out/gputest; mkdir -p $O; cd $GRAFT_REPO_ROOT
timeout -k 10 1100 python -m pytest tests -m gpu -q --durations=8 > $O/gputest.log 2>&1; echo "pytest rc $?" >> $O/gputest.log
tail -14 $O/gputest.log
