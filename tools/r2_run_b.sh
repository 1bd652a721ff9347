O=$GRAFT_REPO_ROOT/gpurun_out/r2b
mkdir -p $O
cd $GRAFT_REPO_ROOT
timeout -k 10 1100 python -m pytest tests -m gpu -q --durations=15 > $O/gputest.log 2>&1; echo "pytest rc $?" >> $O/gputest.log
tail -30 $O/gputest.log
