cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/bfprof; mkdir -p $O
for N in 2 4; do
LASS_BF16_NPX=$N timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/npx$N -o x -- python3 $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --dtype bf16 --modes none > $O/npx$N.json 2> $O/npx$N.log || exit 1
cp $(find $O/npx$N -name '*kernel_stats.csv') $O/stats_npx$N.csv
done
