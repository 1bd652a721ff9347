# kernel iteration loop on the GPU box: conv parity tests, per-layer timing, short bench.  usage: bash tools/gpu_iter.sh TAG [extra bench args]
TAG=${1:-k}
O=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $O
cd $GRAFT_REPO_ROOT
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_stages.py -m gpu -x -q -k "convblock or upconv or separate or taps or encoder_block or fusion or bf16 or random_shapes" > $O/test.log 2>&1; echo "pytest rc $?" >> $O/test.log
tail -4 $O/test.log
grep -q "pytest rc 0" $O/test.log || exit 1
timeout -k 10 300 python tools/conv_bench.py --iters 5 > $O/conv_bench.log 2>&1 || exit 1
cat $O/conv_bench.log
timeout -k 10 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline ${BENCH_ARGS:---modes none} > $O/bench.json 2> $O/bench.err || exit 1
python - <<PY
import json
d=json.loads([l for l in open("$O/bench.json") if l.startswith("{")][0])
print("clips/s", round(d["value"],1), "ms/step", round(d["ms_per_step"],3), "frac", round(d["roofline"]["frac"],4), "conv ms", round(d["roofline"]["class_ms_per_step"],3))
for m,v in d.get("modes",{}).items(): print(m, round(v["clips_s"],1), "conv ms", round(v["conv_ms"],3), "frac", round(v["frac"],4))
PY
