cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r5m; rm -rf $O; mkdir -p $O
for D in f32 bf16; do
  LASS_SPLIT=0 timeout -k 10 150 rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU --output-format csv -d $O/pmc_sq_$D -o x -- python3 $R/bench.py --steps 1 --warmup 1 --dtype $D --modes none --no-cpu-baseline > $O/pmc_sq_$D.log 2>&1 || exit 1
  LASS_SPLIT=0 timeout -k 10 150 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM --output-format csv -d $O/pmc_sq2_$D -o x -- python3 $R/bench.py --steps 1 --warmup 1 --dtype $D --modes none --no-cpu-baseline > $O/pmc_sq2_$D.log 2>&1 || exit 1
  cp $(find $O/pmc_sq_$D -name '*counter_collection.csv') $O/sq_$D.csv
  cp $(find $O/pmc_sq2_$D -name '*counter_collection.csv') $O/sq2_$D.csv
  rm -rf $O/pmc_sq_$D $O/pmc_sq2_$D
done
ls -la $O
