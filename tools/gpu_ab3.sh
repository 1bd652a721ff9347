# Plain A/B of two library builds on one box: bench runs alternating A B A B A B, f32 headline + bf16 mode in each run.
# usage: bash tools/gpu_ab3.sh TAG libA.so libB.so [rounds, default 3]
TAG=$1; A=$2; B=$3; N=${4:-3}
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$TAG; mkdir -p $O; cd $R
for i in $(seq 1 $N); do for L in A B; do
  LIB=$A; [ $L = B ] && LIB=$B
  LASS_HIP_LIB=$R/lass_amd/csrc/$LIB timeout -k 10 300 python3 bench.py --steps 20 --warmup 3 --no-cpu-baseline --modes bf16 > $O/bench_$L$i.json 2> $O/err_$L$i.log || { tail -5 $O/err_$L$i.log; exit 1; }
  python3 -c "
import json; v = json.load(open('$O/bench_$L$i.json')); print('$L$i $LIB f32 %.1f clips/s (conv %.3f ms)  bf16 %.1f clips/s (conv %.3f ms)' % (v['value'], v['roofline']['class_ms_per_step'], v['modes']['bf16']['clips_s'], v['modes']['bf16']['conv_ms']), flush=True)"
done; done
