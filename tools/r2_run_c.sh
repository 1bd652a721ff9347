O=$GRAFT_REPO_ROOT/gpurun_out/r2c; mkdir -p $O; cd $GRAFT_REPO_ROOT
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "full_size" -s > $O/test.log 2>&1; tail -5 $O/test.log
timeout -k 10 600 python tools/longform_bench.py 2>/dev/null | tee $O/longform.log
