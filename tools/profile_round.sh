# Round profile set (run on the GPU box: `bash tools/profile_round.sh`); outputs under gpurun_out/prof_round/, the
# summaries to commit are copied into profiles/<round>/ by hand afterwards.
#  1. rocprofv3 --kernel-trace --stats of the default bench (f32 headline + bf16 / bf16x3 modes in one run; LASS_SPLIT=0 in every
#     rocprofv3 pass: per-kernel durations and counters are only attributable when the two half-batches do not overlap)
#  2. separate --pmc passes (FETCH_SIZE, WRITE_SIZE) of `bench.py --steps 1 --warmup 1` for f32 and bf16
#  3. FETCH_SIZE / WRITE_SIZE calibration of the access shapes the kernels use (tools/fetch_calib.hip)
#  4. SQ counters per conv launch of the pipeline itself (bench.py --steps 1, last step) for f32 and bf16 -> per-launch MFMA
#     utilisation
#  5. the evaluator end to end on 256 clips (tools/eval_bench.py) and the long-form cases (tools/longform_bench.py)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/prof_round
# two calls fit gpurun's 20-minute limit: `profile_round.sh 1` = the rocprofv3 passes, `profile_round.sh 2` = the un-profiled runs
PART=${1:-1}
if [ "$PART" = "1" ]; then
rm -rf $O; mkdir -p $O
LASS_SPLIT=0 timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o x -- python3 $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline > $O/bench_under_rocprof.json 2> $O/stats.log || exit 1
# f32 alone (the headline's kernels by name, no mode legs mixed in): rocprof average per launch vs the in-bench HIP events
LASS_SPLIT=0 timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_f32only -o x -- python3 $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline --modes none > $O/bench_f32only_under_rocprof.json 2> $O/stats_f32only.log || exit 1
cp $(find $O/stats_f32only -name '*kernel_stats.csv') $O/kernel_stats_f32only.csv
for D in f32 bf16; do
  LASS_SPLIT=0 timeout -k 10 150 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch_$D -o x -- python3 $R/bench.py --steps 1 --warmup 1 --dtype $D --modes none --no-cpu-baseline > $O/pmc_fetch_$D.json 2> $O/pmc_fetch_$D.log || exit 1
  LASS_SPLIT=0 timeout -k 10 150 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/pmc_write_$D -o x -- python3 $R/bench.py --steps 1 --warmup 1 --dtype $D --modes none --no-cpu-baseline > $O/pmc_write_$D.json 2> $O/pmc_write_$D.log || exit 1
done
hipcc -O3 --offload-arch=gfx950 $R/tools/fetch_calib.hip -o $O/fetch_calib 2>/dev/null || exit 1
timeout -k 10 120 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/calib_fetch -o x -- $O/fetch_calib > $O/calib_fetch.log 2>&1 || exit 1
timeout -k 10 120 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/calib_write -o x -- $O/fetch_calib > $O/calib_write.log 2>&1 || exit 1
for D in f32 bf16; do  # the SHIPPED pipeline, launch by launch (not the stage API)
  LASS_SPLIT=0 timeout -k 10 150 rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU --output-format csv -d $O/pmc_sq_$D -o x -- python3 $R/bench.py --steps 1 --warmup 1 --dtype $D --modes none --no-cpu-baseline > $O/pmc_sq_$D.log 2>&1 || exit 1
done
for D in f32 bf16; do  # second SQ pass: LDS / VMEM shares (with pass 1: where the wave cycles go, tools/sq_wait_table.py)
  LASS_SPLIT=0 timeout -k 10 150 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM --output-format csv -d $O/pmc_sq2_$D -o x -- python3 $R/bench.py --steps 1 --warmup 1 --dtype $D --modes none --no-cpu-baseline > $O/pmc_sq2_$D.log 2>&1 || exit 1
done
cd $R
for D in f32 bf16; do
  python3 tools/sq_wait_table.py $(find $O/pmc_sq_$D -name '*counter_collection.csv') $(find $O/pmc_sq2_$D -name '*counter_collection.csv') > $O/wave_cycles_$D.md
done
F=$(find $O/calib_fetch -name '*counter_collection.csv'); W=$(find $O/calib_write -name '*counter_collection.csv')
python3 tools/fetch_calib_summary.py $F $W $O/fetch_calibration.json > /dev/null
for D in f32 bf16; do
  python3 tools/traffic_summary.py $(find $O/pmc_fetch_$D -name '*counter_collection.csv') $(find $O/pmc_write_$D -name '*counter_collection.csv') $O/conv_traffic_$D.json $D $O/fetch_calibration.json
  python3 tools/pmc_table.py $(find $O/pmc_sq_$D -name '*counter_collection.csv') $D > $O/mfma_util_$D.md
  cat $O/mfma_util_$D.md
  python3 tools/traffic_table.py $(find $O/pmc_fetch_$D -name '*counter_collection.csv') $(find $O/pmc_write_$D -name '*counter_collection.csv') $D > $O/traffic_per_launch_$D.md
done
cp $(find $O/stats -name '*kernel_stats.csv') $O/kernel_stats.csv
rm -rf $O/stats $O/stats_f32only $O/pmc_fetch_* $O/pmc_write_* $O/calib_fetch $O/calib_write $O/pmc_sq_* $O/pmc_sq2_* $O/fetch_calib
echo done part 1; exit 0
fi
mkdir -p $O; cd $R
EVAL_BENCH_REPS=5 timeout -k 10 200 python3 tools/eval_bench.py 260 > $O/eval_bench_260clips.log 2>&1
timeout -k 10 200 python3 tools/longform_bench.py > $O/longform_bench.log 2>&1
# the driver-style default run (no profiler): the line BENCH_rNN.json will hold, incl. cpu_baseline and the evaluator legs
timeout -k 10 600 python3 bench.py > $O/bench_default.json 2> $O/bench_default.err
# the STFT hazard probe (needs lass_amd/csrc/liblass_hip_stftdbg.so = stft.hip with -DLASS_STFT_DBG, no -fno-slp-vectorize)
if [ -f lass_amd/csrc/liblass_hip_stftdbg.so ]; then
  LASS_HIP_LIB=$R/lass_amd/csrc/liblass_hip_stftdbg.so timeout -k 10 300 python3 tools/stft_hazard_probe.py 10 4 > $O/stft_hazard_probe.log 2>&1
fi
timeout -k 10 200 python3 tools/coresident_stress.py bf16 60 > $O/coresident_stress.log 2>&1
echo done part 2
