# SQ counters of the conv kernels on a few layers in one compute mode (LASS_COMPUTE=f32|bf16|bf16x3).
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
M=${LASS_COMPUTE:-bf16}
timeout -k 10 90 rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU --output-format csv -d $R/gpurun_out/pmc_sq_$M -o x -- python3 $R/tools/conv_bench.py --iters 1 --only encoder_block1,encoder_block3,decoder_block3,decoder_block5 > $R/gpurun_out/pmc_sq_$M.log 2>&1 || exit 1
timeout -k 10 90 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM --output-format csv -d $R/gpurun_out/pmc_sq2_$M -o x -- python3 $R/tools/conv_bench.py --iters 1 --only encoder_block1,encoder_block3,decoder_block3,decoder_block5 > $R/gpurun_out/pmc_sq2_$M.log 2>&1 || exit 1
echo done
