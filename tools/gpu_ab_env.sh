# A/B of one environment switch on the same box: bench.py (f32 headline + the modes given) alternating A B A B.
# usage: bash tools/gpu_ab_env.sh TAG VAR A_VALUE B_VALUE [bench args...]
TAG=$1; VAR=$2; VA=$3; VB=$4; shift 4
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$TAG; mkdir -p $O; cd $R
for i in 1 2; do for L in A B; do
  V=$VA; [ $L = B ] && V=$VB
  env $VAR=$V timeout -k 10 300 python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline "$@" > $O/bench_$L$i.json 2> $O/err_$L$i.log || { tail -5 $O/err_$L$i.log; exit 1; }
  python3 - <<PY
import json
v = json.load(open("$O/bench_$L$i.json"))
m = v.get("modes", {})
print("$L$i $VAR=$V", "f32 clips/s %.1f conv_ms %.2f" % (v["value"], v["roofline"]["class_ms_per_step"]), " ".join("%s %.1f (conv %.2f ms)" % (k, d["clips_s"], d["conv_ms"]) for k, d in m.items()), flush=True)
PY
done; done
