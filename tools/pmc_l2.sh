# L1 (TCP) / L2 (TCC) counters of the Winograd kernels on two layers, old (LASS_WINO=1) vs wave-specialised (2) schedule.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for W in 1 2; do
export LASS_WINO=$W
timeout -k 10 60 rocprofv3 --kernel-trace --pmc TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_PENDING_STALL_CYCLES_sum --output-format csv -d $R/gpurun_out/pmc_tcp_w$W -o x -- python3 $R/tools/conv_bench.py --iters 1 --only decoder_block3,encoder_block3 > $R/gpurun_out/pmc_tcp_w$W.log 2>&1 || exit 1
timeout -k 10 60 rocprofv3 --kernel-trace --pmc TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum TCC_TAG_STALL_sum --output-format csv -d $R/gpurun_out/pmc_tcc_w$W -o x -- python3 $R/tools/conv_bench.py --iters 1 --only decoder_block3,encoder_block3 > $R/gpurun_out/pmc_tcc_w$W.log 2>&1 || exit 1
done
echo done
