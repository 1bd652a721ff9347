# round-2 first GPU pass: tests, bench, FETCH/WRITE calibration
O=$GRAFT_REPO_ROOT/gpurun_out/r2a
mkdir -p $O
cd $GRAFT_REPO_ROOT
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $O/gputest.log 2>&1; echo "pytest rc $?" >> $O/gputest.log
tail -5 $O/gputest.log
timeout -k 10 400 python bench.py --steps 10 --warmup 3 > $O/bench.json 2> $O/bench.err; echo "bench rc $?"
hipcc -O3 --offload-arch=gfx950 tools/fetch_calib.hip -o $O/fetch_calib || exit 1
cd /tmp && export TMPDIR=/tmp
timeout -k 10 120 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/calib_fetch -o x -- $O/fetch_calib > $O/calib_fetch.log 2>&1 || exit 1
timeout -k 10 120 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/calib_write -o x -- $O/fetch_calib > $O/calib_write.log 2>&1 || exit 1
cd $GRAFT_REPO_ROOT
python tools/fetch_calib_summary.py $(find $O/calib_fetch -name '*counter_collection.csv') $(find $O/calib_write -name '*counter_collection.csv') $O/fetch_calibration.json
