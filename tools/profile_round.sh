# Round-end profile set (run on the GPU box: `bash tools/profile_round.sh`); outputs under gpurun_out/prof_round/.
#  1. rocprofv3 --kernel-trace --stats of the default bench (f32) and of --dtype bf16  -> kernel stats CSVs
#  2. separate --pmc passes (FETCH_SIZE, WRITE_SIZE) of `bench.py --steps 1 --warmup 1` -> HBM traffic per conv launch
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/prof_round
mkdir -p $O
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_f32 -o x -- python3 $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline > $O/bench_f32_under_rocprof.json 2> $O/stats_f32.log || exit 1
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_bf16 -o x -- python3 $R/bench.py --steps 5 --warmup 2 --dtype bf16 --no-cpu-baseline > $O/bench_bf16_under_rocprof.json 2> $O/stats_bf16.log || exit 1
timeout -k 10 120 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -o x -- python3 $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline > $O/pmc_fetch.json 2> $O/pmc_fetch.log || exit 1
timeout -k 10 120 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/pmc_write -o x -- python3 $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline > $O/pmc_write.json 2> $O/pmc_write.log || exit 1
echo done
